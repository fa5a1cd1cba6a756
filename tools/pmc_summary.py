#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o f -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o w -- python bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv out.json

Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KiB and counts 128-B requests as 64 B on gfx950 -> x 1024 x 2;
WRITE_SIZE is in KiB and exact for 16-B-per-lane stores -> x 1024."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def load(path, counter):
    tot, seen, dur = collections.Counter(), collections.defaultdict(set), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return tot, seen, dur


def main():
    f_tot, f_seen, f_dur = load(sys.argv[1], "FETCH_SIZE")
    w_tot, w_seen, _ = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(f_tot, key=lambda k: -f_dur[k]):
        n = len(f_seen[k])
        nw = max(1, len(w_seen.get(k, ())))
        out[k] = {"launches": n, "avg_us": f_dur[k] / n / 1e3, "fetch_bytes_corrected": f_tot[k] / n * 1024 * 2,
                  "write_bytes": w_tot.get(k, 0.0) / nw * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out.items())[:16]:
        print(f"{k[:44]:44s} x{v['launches']:4d} {v['avg_us']:8.1f} us  fetch {v['fetch_bytes_corrected'] / 1e6:8.1f} MB  write {v['write_bytes'] / 1e6:8.1f} MB")


if __name__ == "__main__":
    main()
