# coding=utf-8
"""Cuts a rocprofv3 --kernel-trace (csv) into phases separated by idle gaps (> 100 ms) and reports, per phase: number of
dispatches, sum of kernel durations, idle time BETWEEN consecutive kernels inside the phase, wall (first start .. last
end), and the per-kernel mean durations — the evidence for "what does a hipGraph replay add to the eager sequence".

    python tools/trace_gaps.py <dir-with-*kernel_trace.csv> [--min-dispatches N]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
min_disp = int(sys.argv[sys.argv.index("--min-dispatches") + 1]) if "--min-dispatches" in sys.argv else 8
files = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
if not files:
    sys.exit("no *kernel_trace.csv under " + root)
rows = []
for fn in files:
    with open(fn) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
phases, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 100e6:
        phases.append(cur)
        cur = []
    cur.append(r)
if cur:
    phases.append(cur)
print("| phase | dispatches | kernel time ms | idle between kernels ms | wall ms | largest gap us |")
print("|---|---|---|---|---|---|")
detail = []
for i, p in enumerate(phases):
    if len(p) < min_disp:
        continue
    busy = sum(b - a for a, b, _ in p) / 1e6
    gaps = [max(0, p[j + 1][0] - p[j][1]) for j in range(len(p) - 1)]
    wall = (p[-1][1] - p[0][0]) / 1e6
    print("| {} | {} | {:.3f} | {:.3f} | {:.3f} | {:.1f} |".format(i, len(p), busy, sum(gaps) / 1e6, wall, max(gaps) / 1e3 if gaps else 0))
    per = defaultdict(list)
    for a, b, k in p:
        per[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]].append((b - a) / 1e3)
    detail.append((i, per, gaps))
for i, per, gaps in detail:
    print("\nphase {}: mean kernel durations (us) / calls; median gap {:.1f} us".format(
        i, sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("  {:>10.1f} x{:<4d} {}".format(sum(v) / len(v), len(v), k))
