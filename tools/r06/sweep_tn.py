# coding=utf-8
"""x^T g (weight + bias gradient in one reduction over the node dimension) at the shapes of the products / Reddit layers, by tile
size S, waves along M (WM) and workgroup count — one process per setting (the switches are read once).  JSON lines."""
import sys, os, json, subprocess
SHAPES = [(2449029, 512, 128, False), (2449029, 100, 128, True), (2449029, 100, 256, True), (232965, 602, 80, True), (169343, 128, 256, True)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import torch
    from tf_geometric_amd.plan import gemm_tn
    res = {"S": os.environ.get("TFGX_TN_S", "policy"), "WM": os.environ.get("TFGX_TN_WM", "policy"), "WGS": os.environ.get("TFGX_TN_WGS_ENV", "policy")}
    for M, Ka, N, bias in SHAPES:
        x = torch.randn(M, Ka, device="cuda"); g = torch.randn(M, N, device="cuda")
        fn = lambda: gemm_tn(x, g, want_bias=bias)
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); a.record()
            for _ in range(5):
                fn()
            b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
        ms = min(ts)
        res["{}x{}->{}{}".format(M, Ka, N, "+b" if bias else "")] = {"ms": round(ms, 3), "TFLOPs": round(2.0 * M * Ka * N / ms / 1e9, 1),
                                                                      "GBps": round((M * (Ka + N) * 4) / ms / 1e6, 0)}
        del x, g
    print(json.dumps(res))
else:
    for s in ("0", "16", "32"):
        for wm in ("0", "1", "2", "4"):
            for wgs in ("0", "1024"):
                if s == "0" and (wm != "0"):
                    continue
                env = dict(os.environ)
                if s != "0": env["TFGX_TN_S"] = s
                if wm != "0": env["TFGX_TN_WM"] = wm
                if wgs != "0": env["TFGX_TN_WGS_ENV"] = wgs
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
                print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else json.dumps({"error": r.stderr[-300:]}), flush=True)
