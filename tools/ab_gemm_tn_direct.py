import sys, json, torch, os
sys.path.insert(0, ".")
from tf_geometric_amd.plan import gemm_tn
n = 2400000
out = {"direct": os.environ.get("TFGX_TN_DIRECT", "0"), "wgs": os.environ.get("TFGX_TN_WGS_ENV", "")}
for (ka, nn) in [(100, 256), (256, 40), (128, 128), (100, 16), (256, 256), (602, 80)]:
    m = n if ka * nn < 60000 else n // 4
    x = torch.randn(m, ka, device="cuda"); g = torch.randn(m, nn, device="cuda")
    dw, db = gemm_tn(x, g, want_bias=True)
    ref = (x[:200000].double().t() @ g[:200000].double())
    dw2, db2 = gemm_tn(x[:200000], g[:200000], want_bias=True)
    err = float((dw2.double() - ref).abs().max() / ref.abs().max())
    errb = float((db2.double() - g[:200000].double().sum(0)).abs().max())
    for _ in range(3): gemm_tn(x, g, want_bias=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): gemm_tn(x, g, want_bias=True)
    b.record(); torch.cuda.synchronize()
    out["{}x{}".format(ka, nn)] = [round(a.elapsed_time(b) / 10, 3), "%.1e" % err, "%.1e" % errb]
print(json.dumps(out))
