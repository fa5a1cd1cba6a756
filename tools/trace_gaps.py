#!/usr/bin/env python3
"""Idle time of the GPU inside the timed steps, from a rocprofv3 kernel trace (csv).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 4 --warmup 2 ...
    python tools/trace_gaps.py gpurun_out/trace/**/*_kernel_trace.csv [--last-ms 100]

Prints, for the last part of the trace (the timed steps): the span, the union of kernel intervals over all streams
(= time the GPU had at least one kernel resident), the idle remainder, a histogram of idle gaps and the kernels that
most often precede a gap.  A launch-bound schedule shows many 2-10 us gaps; a dependency stall shows few long ones."""
import argparse
import collections
import csv
import glob
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--last-ms", type=float, default=100.0, help="analyse only this trailing window of the trace (the timed steps)")
    a = ap.parse_args()
    files = [f for p in a.paths for f in glob.glob(p, recursive=True)]
    if not files:
        sys.exit("no trace file")
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t1 - a.last_ms * 1e6
    rows = [r for r in rows if r[0] >= cut]
    span = max(r[1] for r in rows) - rows[0][0]
    busy, gaps, cur_end, prev_name = 0, [], rows[0][0], None
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, prev_name, n))
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
            prev_name = n
    idle = span - busy
    print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f} ms  idle {idle / 1e6:.3f} ms ({100 * idle / span:.1f} %)")
    print(f"sum of kernel durations {sum(e - s for s, e, _ in rows) / 1e6:.3f} ms (overlap = sum - busy = {(sum(e - s for s, e, _ in rows) - busy) / 1e6:.3f} ms)")
    edges = [0, 2e3, 5e3, 1e4, 2e4, 5e4, 1e5, 1e6, 1e12]
    for lo, hi in zip(edges, edges[1:]):
        g = [x[0] for x in gaps if lo <= x[0] < hi]
        if g:
            print(f"  gaps {lo / 1e3:7.0f}-{hi / 1e3:<9.0f} us: {len(g):6d}  total {sum(g) / 1e6:7.3f} ms")
    by = collections.Counter()
    for g, p, n in gaps:
        by[(p or "")[:60] + "  ->  " + n[:60]] += g
    print("largest idle by (kernel before -> kernel after):")
    for k, v in by.most_common(12):
        print(f"  {v / 1e6:7.3f} ms  {k}")


if __name__ == "__main__":
    main()
