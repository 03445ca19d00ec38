# coding=utf-8
"""Per-dispatch (kernel name, start ns, end ns, duration us) rows of ONE kernel out of a rocprofv3 --kernel-trace result
(rocpd .db or kernel_trace.csv), written as a small CSV under profiles/ so that `roofline.frac` can be re-derived without
the raw database.

    python tools/kernel_dispatch_csv.py <results.db | dir with *kernel_trace.csv> "<kernel substring>" <out.csv>
"""
import csv
import glob
import os
import sqlite3
import sys

src, needle, out = sys.argv[1:4]
rows = []
if os.path.isdir(src):
    for fn in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        with open(fn) as fh:
            for r in csv.DictReader(fh):
                if needle in r["Kernel_Name"]:
                    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
else:
    con = sqlite3.connect(src)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    for name, start, end in cur.execute("select {n}, start, end from kernels order by start".format(n=name_col)):
        if needle in name:
            rows.append((name, int(start), int(end)))
rows.sort(key=lambda r: r[1])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "start_ns", "end_ns", "duration_us"])
    for name, a, b in rows:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        w.writerow([short, a, b, "{:.3f}".format((b - a) / 1e3)])
d = [(b - a) / 1e3 for _, a, b in rows]
print("{} dispatches, avg {:.3f} us, min {:.3f}, max {:.3f} -> {}".format(len(d), sum(d) / max(len(d), 1), min(d or [0]), max(d or [0]), out))
