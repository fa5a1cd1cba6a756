#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 PMC counters (one or more counter_collection csv files).

    rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT ... --kernel-trace --output-format csv -d DIR -o x -- python3 tools/bench_attn.py
    python tools/pmc_generic.py out.json DIR/**/x_counter_collection.csv [more.csv]

Values are summed over the dispatch's shader engines / XCDs as rocprofv3 reports them and divided by the launch count."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name)


def main():
    out_path, pats = sys.argv[1], sys.argv[2:]
    tot = collections.defaultdict(collections.Counter)
    seen = collections.defaultdict(lambda: collections.defaultdict(set))
    dur = collections.Counter()
    durseen = collections.defaultdict(set)
    for pat in pats:
        for path in glob.glob(pat, recursive=True):
            for r in csv.DictReader(open(path)):
                k, c = short(r["Kernel_Name"]), r["Counter_Name"]
                tot[k][c] += float(r["Counter_Value"])
                seen[k][c].add((path, r["Dispatch_Id"]))
                if (path, r["Dispatch_Id"]) not in durseen[k]:
                    durseen[k].add((path, r["Dispatch_Id"]))
                    dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for k in sorted(tot, key=lambda k: -dur[k]):
        out[k] = {"avg_us_under_pmc": dur[k] / max(1, len(durseen[k])) / 1e3}
        for c, v in tot[k].items():
            out[k][c] = v / max(1, len(seen[k][c]))
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in list(out.items())[:12]:
        print(k[:50], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
