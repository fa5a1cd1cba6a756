#!/usr/bin/env python3
"""Which launches of a rocprofv3 kernel trace of bench.py belong to the STEP and which to setup?  Cuts the trace at the grad-norm
kernel (`sumsq_partial_kernel`: once per optimiser step) and prints, for the last N steps, every kernel that is not one of the
in-tree kernels ((anonymous namespace)::...) with its launches per step; everything in front of the first cut is setup.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline ...
    python tools/step_launch_census.py DIR/**/t_kernel_trace.csv [N]"""
import collections
import csv
import glob
import sys


def main():
    paths = glob.glob(sys.argv[1], recursive=True)
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = []
    for p in paths:
        rows += list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    cuts = [i for i, n in enumerate(names) if "sumsq_partial_kernel" in n]
    print(f"{len(rows)} launches, {len(cuts)} optimiser steps in the trace")
    if len(cuts) < n_last + 1:
        print("not enough steps")
        return
    foreign_setup = collections.Counter(n.split("(")[0][:90] for n in names[:cuts[0]] if "(anonymous namespace)" not in n and "_GLOBAL__N_" not in n)
    print("launches that are not in-tree kernels, IN FRONT OF the first optimiser step (setup):")
    for k, v in foreign_setup.most_common():
        print(f"  {v:6d}  {k}")
    per_step = []
    for a, b in zip(cuts[-n_last - 1:-1], cuts[-n_last:]):
        seg = names[a:b]
        per_step.append((len(seg), collections.Counter(n.split("(")[0][:90] for n in seg if "(anonymous namespace)" not in n and "_GLOBAL__N_" not in n)))
    print(f"last {n_last} steps: launches per step {[p[0] for p in per_step]}; launches that are not in-tree kernels, per step:")
    keys = sorted(set(k for _, c in per_step for k in c))
    for k in keys:
        print(f"  {[c[k] for _, c in per_step]}  {k}")
    if not keys:
        print("  none")


if __name__ == "__main__":
    main()
