# coding=utf-8
"""F = 256 on the products graph: the headline launch (weighted sum + implicit self-loops) by lanes per row of the column-block
walk (TFGX_REDUCE_WIDE_G256 = 32: two 128-column blocks, 16: four 64-column blocks) and one burst per row; the row stride as
the layer lays it out (288) and the caller's dense 256 (re-laid per table).  One process per setting (the switch is read once)."""
import sys, os, json, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import torch
    from tf_geometric_amd import synthetic, _lib as L, plan as P
    from tf_geometric_amd.plan import CsrPlan, segment_reduce
    n, e, _ = synthetic.WORKLOADS["products"]
    ei = L.as_i32(synthetic.synthetic_edge_stripe(n, e, seed=0))
    plan = CsrPlan.build(ei, n, n)
    w = torch.rand(int(ei.shape[1]), device="cuda") + 0.5
    sc = torch.rand(n, device="cuda")
    res = {"G256": os.environ.get("TFGX_REDUCE_WIDE_G256", "32"), "wide_blocks": sys.argv[2]}
    wb = None if sys.argv[2] == "policy" else int(sys.argv[2])
    for F in (256, 384, 128):
        for ld in ("friendly", "dense"):
            x = P.gather_friendly_empty(n, F, torch.device("cuda")) if ld == "friendly" else torch.empty(n, F, device="cuda")
            x.normal_()
            out = torch.empty(n, F, device="cuda")
            fn = lambda: segment_reduce(plan, x, L.SUM, w_csr=w, self_coef=sc, out=out, wide_blocks=wb)
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); a.record()
                for _ in range(5):
                    fn()
                b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
            res["F{}_{}_ms".format(F, ld)] = round(min(ts), 3)
            res["F{}_{}_kernel".format(F, ld)] = segment_reduce(plan, x, L.SUM, w_csr=w, self_coef=sc, out=out, wide_blocks=wb, describe=True)
            del x, out
    print(json.dumps(res))
else:
    for rep in range(2):
        for g256, wbs in (("32", "policy"), ("16", "policy"), ("32", "-1")):
            env = dict(os.environ, TFGX_REDUCE_WIDE_G256=g256)
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", wbs], env=env, capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:])
