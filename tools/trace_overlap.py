# coding=utf-8
"""From a rocprofv3 --kernel-trace csv: the LAST busy phase (after the last idle gap > 100 ms) split into exchange kernels
(RCCL device kernels + the halo pack) and reduce passes; reports their busy time, the time both are running at once, and the
wall — evidence that the halo exchange runs on a second stream UNDER the passes.

    python tools/trace_overlap.py <dir-with-*kernel_trace.csv>
"""
import csv
import glob
import json
import os
import sys

files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))
rows = []
for fn in files:
    with open(fn) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 100e6:
        cut = i
ph = rows[cut:]


def is_comm(name):
    return "nccl" in name.lower() or "rccl" in name.lower()


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    t = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            t += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return t


comm = union([(a, b) for a, b, nme, _ in ph if is_comm(nme)])
comp = union([(a, b) for a, b, nme, _ in ph if "seg_reduce_kernel" in nme])
names = {}
for a, b, nme, q in ph:
    key = nme.replace("(anonymous namespace)::", "").split("(")[0][-70:]
    names.setdefault(key, [0, 0.0, set()])
    names[key][0] += 1
    names[key][1] += (b - a) / 1e6
    names[key][2].add(q)
print(json.dumps({
    "dispatches": len(ph), "wall_ms": (ph[-1][1] - ph[0][0]) / 1e6,
    "exchange_kernels_busy_ms": total(comm) / 1e6, "reduce_passes_busy_ms": total(comp) / 1e6,
    "both_running_ms": intersect(comm, comp) / 1e6,
    "exchange_time_hidden_under_passes": round(intersect(comm, comp) / max(total(comm), 1), 3),
    "kernels": {k: {"calls": v[0], "total_ms": round(v[1], 3), "queues": sorted(v[2])} for k, v in names.items()}}, indent=1))
