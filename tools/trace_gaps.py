#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
# steady state: from the first adamw_kernel to the last one
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
if len(idx) >= 2:
    rows = rows[idx[0] + 1: idx[-1] + 1]
    nsteps = len(idx) - 1
else:
    rows = rows[skip:]
    nsteps = 1
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy_end = int(rows[0]["Start_Timestamp"])
idle = 0
gaps = []
dur = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur += e - s
    if s > busy_end:
        idle += s - busy_end
        gaps.append((s - busy_end, r["Kernel_Name"][:60]))
    busy_end = max(busy_end, e)
print(f"steps {nsteps}  kernels/step {len(rows) / nsteps:.0f}  span/step {span / nsteps / 1e6:.3f} ms  kernel time/step {dur / nsteps / 1e6:.3f} ms  "
      f"idle/step {idle / nsteps / 1e6:.3f} ms in {len(gaps) / nsteps:.0f} gaps (mean {idle / max(len(gaps), 1) / 1e3:.2f} us)")
h = collections.Counter()
for g, _ in gaps:
    h[min(int(g / 1000), 20)] += 1
print("gap histogram (us: count/step):", {k: round(v / nsteps, 1) for k, v in sorted(h.items())})
big = sorted(gaps, reverse=True)[:12]
print("largest gaps:", [(round(g / 1e3, 1), n) for g, n in big])
by = collections.Counter()
for g, n in gaps:
    by[n] += g
print("idle by the kernel that follows (us/step):", [(n, round(v / nsteps / 1e3, 1)) for n, v in by.most_common(12)])
