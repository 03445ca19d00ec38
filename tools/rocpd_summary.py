# coding=utf-8
"""Summarise rocprofv3 rocpd (sqlite) outputs: per-kernel calls / avg / total duration (the --kernel-trace --stats
view) and per-kernel PMC counter sums.  usage: python tools/rocpd_summary.py <results.db> [...]"""
import sqlite3
import sys


def summarise(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    out = ["## {}".format(db)]
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select {n}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                       "from kernels group by {n} order by sum(end-start) desc".format(n=name_col)).fetchall()
    total = sum(r[5] for r in rows) or 1
    out.append("| kernel | calls | avg_us | min_us | max_us | total_ms | pct |")
    out.append("|---|---|---|---|---|---|---|")
    for name, calls, avg, mn, mx, tot in rows[:25]:
        out.append("| {} | {} | {:.1f} | {:.1f} | {:.1f} | {:.3f} | {:.1f} |".format(
            name[:110], calls, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e6, 100.0 * tot / total))
    try:
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        if ccols:
            kn = "kernel_name" if "kernel_name" in ccols else name_col
            rows = cur.execute("select {k}, counter_name, count(*), sum(value), avg(value) from counters_collection "
                               "group by {k}, counter_name order by sum(value) desc".format(k=kn)).fetchall()
            if rows:
                out.append("")
                out.append("| kernel | counter | dispatches | sum | avg per dispatch |")
                out.append("|---|---|---|---|---|")
                for k, c, n, s, a in rows[:40]:
                    out.append("| {} | {} | {} | {:.6g} | {:.6g} |".format(k[:90], c, n, s, a))
    except sqlite3.Error as ex:
        out.append("(no counters: {})".format(ex))
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(summarise(p))
        print()
