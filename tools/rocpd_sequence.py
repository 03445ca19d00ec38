# coding=utf-8
"""The LAST n kernel dispatches of a rocprofv3 rocpd (sqlite) trace, in launch order, with durations and the gap to the previous
dispatch's end — one training step's launch sequence.  usage: python tools/rocpd_sequence.py <results.db> [n]"""
import sqlite3
import sys

db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select {n}, start, end from kernels order by start desc limit {k}".format(n=name_col, k=n)).fetchall()[::-1]
prev = None
print("| # | kernel | us | gap_us |")
print("|---|---|---|---|")
tot = 0.0
for i, (name, s, e) in enumerate(rows):
    name = name.replace("void tfgx::(anonymous namespace)::", "").replace("tfgx::(anonymous namespace)::", "")
    print("| {} | {} | {:.1f} | {:.1f} |".format(i, name[:100], (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
    tot += (e - s) / 1e3
    prev = e
print("kernel time {:.1f} us, span {:.1f} us".format(tot, (rows[-1][2] - rows[0][1]) / 1e3))
