# coding=utf-8
"""Assemble profiles/r04_rmat_pmc.json from the rocprofv3 passes of tools/rmat_pmc.py (tools/rmat_pmc.sh).

    python tools/rmat_pmc_collect.py <dir with <variant>/{stats,hit,fetch}/ and <variant>.json> <out.json>
Per variant and per CALL of the headline launch (all kernels of the call together: the main walk, the hub-chunk walk, the hub
finalize): kernel time (rocprofv3 --kernel-trace), TCC_HIT / TCC_MISS (L2 hit rate), FETCH_SIZE (KiB; x2 on gfx950 per
MI355X_MICROARCH.md = bytes that came from beyond L2: MALL or HBM)."""
import glob
import json
import os
import sqlite3
import sys

src, out = sys.argv[1:3]
NEEDLES = ("seg_reduce", "hub_")


def calls_of(path, default):
    try:
        with open(path) as fh:
            return json.loads([ln for ln in fh if ln.startswith("{")][-1])["calls_total"]
    except (OSError, IndexError, KeyError, ValueError):
        return default


def db_of(d):
    hits = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    return hits[0] if hits else None


def kernel_rows(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    return [(n_, c, t) for n_, c, t in cur.execute("select {n}, count(*), sum(end-start) from kernels group by {n}".format(n=name_col))
            if any(k in n_ for k in NEEDLES)]


def counter_rows(db):
    cur = sqlite3.connect(db).cursor()
    ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kn = "kernel_name" if "kernel_name" in ccols else "name"
    return [(k, c, n_, s) for k, c, n_, s in cur.execute(
        "select {k}, counter_name, count(*), sum(value) from counters_collection group by {k}, counter_name".format(k=kn))
        if any(x in k for x in NEEDLES)]


res = {"_comment": __doc__.strip().splitlines()[-4:], "variants": {}}
for meta_path in sorted(glob.glob(os.path.join(src, "*.json"))):
    with open(meta_path) as fh:
        meta = json.loads([ln for ln in fh if ln.startswith("{")][-1])
    v, calls = meta["variant"], meta["calls_total"]
    rec = dict(meta)
    sdb, hdb, fdb = (db_of(os.path.join(src, v, k)) for k in ("stats", "hit", "fetch"))
    if sdb:
        rows = kernel_rows(sdb)
        rec["kernel_ms_per_call_rocprof"] = sum(t for _, _, t in rows) / 1e6 / calls
        rec["kernels"] = {n_.split("(")[0].replace("void tfgx::(anonymous namespace)::", ""): {"dispatches": c, "total_ms": t / 1e6}
                          for n_, c, t in rows}
    if hdb:
        calls = calls_of(os.path.join(src, v + ".hit.meta"), 3)
        tot = {}
        for k, c, n_, s in counter_rows(hdb):
            tot[c] = tot.get(c, 0.0) + s
        if "TCC_HIT_sum" in tot and "TCC_MISS_sum" in tot:
            rec["tcc_hit_per_call"], rec["tcc_miss_per_call"] = tot["TCC_HIT_sum"] / calls, tot["TCC_MISS_sum"] / calls
            rec["l2_hit_rate"] = tot["TCC_HIT_sum"] / max(tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"], 1.0)
    if fdb:
        calls = calls_of(os.path.join(src, v + ".fetch.meta"), 3)
        tot = sum(s for k, c, n_, s in counter_rows(fdb) if c == "FETCH_SIZE")
        rec["fetch_size_kib_per_call"] = tot / calls
        rec["bytes_from_beyond_l2_per_call"] = tot / calls * 1024 * 2
    res["variants"][v] = rec
with open(out, "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps({v: {k: r.get(k) for k in ("ms_per_call_events", "kernel_ms_per_call_rocprof", "l2_hit_rate",
                                            "bytes_from_beyond_l2_per_call", "hub_rows", "hub_threshold")}
                  for v, r in res["variants"].items()}, indent=1))
