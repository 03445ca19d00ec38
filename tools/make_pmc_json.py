# coding=utf-8
"""rocprofv3 PMC passes -> profiles/<tag>_<workload>_pmc.json (what bench.py imports as roofline.traffic).

    python tools/make_pmc_json.py <fetch.db> <write.db> <stats.db> <kernel substring> <out.json> [workload]

FETCH_SIZE / WRITE_SIZE are reported in KiB and summed over the XCDs; on gfx950 FETCH_SIZE tallies a 128-byte request
as 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is used as reported.
kernel_ms is the kernel's average duration in the --kernel-trace --stats pass of the SAME command."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["tf_geometric_amd/csrc/tfgx_reduce.hip", "tf_geometric_amd/csrc/tfgx_common.h"]


def kernel_source_sha():
    """sha256[:16] of the sources of the measured kernel: bench.py drops an imported `roofline.traffic` whose profile was
    taken on other sources (the counters describe the binary that ran, not the one in the tree)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _per_dispatch(db, counter, needle):
    con = sqlite3.connect(db)
    cur = con.cursor()
    ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in ccols else "name"
    rows = cur.execute("select {k}, count(*), sum(value) from counters_collection where counter_name = ? group by {k}"
                       .format(k=kn), (counter,)).fetchall()
    rows = [r for r in rows if needle in r[0]]
    if not rows:
        raise SystemExit("no {} rows for a kernel containing {!r} in {}".format(counter, needle, db))
    name, n, total = max(rows, key=lambda r: r[2])
    return name, total / n, n


def _avg_us(db, needle):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select {n}, count(*), avg(end-start) from kernels group by {n}".format(n=name_col)).fetchall()
    rows = [r for r in rows if needle in r[0]]
    name, calls, avg = max(rows, key=lambda r: r[1] * r[2])
    return name, avg / 1e3, calls


if __name__ == "__main__":
    fetch_db, write_db, stats_db, needle, out = sys.argv[1:6]
    workload = sys.argv[6] if len(sys.argv) > 6 else "products"
    kname, fetch_kib, nf = _per_dispatch(fetch_db, "FETCH_SIZE", needle)
    _, write_kib, nw = _per_dispatch(write_db, "WRITE_SIZE", needle)
    _, avg_us, calls = _avg_us(stats_db, needle)
    short = kname[kname.find("seg_reduce_kernel"):] if "seg_reduce_kernel" in kname else kname
    short = short.split("(")[0]
    blob = {
        "_comment": "Per-launch HBM-side traffic from separate rocprofv3 --pmc passes of `python bench.py` (FETCH_SIZE, "
                    "WRITE_SIZE; KiB, summed over XCDs). gfx950: FETCH_SIZE counts a 128-B request as 64 B -> doubled "
                    "(MI355X_MICROARCH.md); WRITE_SIZE as reported. kernel_ms = rocprofv3 --kernel-trace --stats average "
                    "of the same command on the same box.",
        "workload": workload, "kernel": short,
        "fetch_size_kib_per_launch": fetch_kib, "write_size_kib_per_launch": write_kib, "fetch_correction": 2.0,
        "traffic_bytes_per_launch": int(fetch_kib * 1024 * 2 + write_kib * 1024),
        "kernel_ms": avg_us / 1e3, "kernel_avg_us_rocprof": avg_us,
        "dispatches": {"fetch_pass": nf, "write_pass": nw, "stats_pass": calls},
        "kernel_source_sha16": kernel_source_sha(), "kernel_sources": KERNEL_SOURCES,
    }
    with open(out, "w") as fh:
        json.dump(blob, fh, indent=2)
    print(json.dumps(blob))
