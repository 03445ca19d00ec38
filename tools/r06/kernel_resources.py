# coding=utf-8
"""VGPRs / SGPRs / scratch / spills / occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/r06/kernel_resources.py tf_geometric_amd/csrc/tfgx_attn.hip [name filter]"""
import re, subprocess, sys, os, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o",
                        os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage"] + os.environ.get("TFGX_EXTRA_HIPCC_FLAGS", "").split(),
                       capture_output=True, text=True)
txt = r.stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n
print("{:<100s} {:>5s} {:>5s} {:>7s} {:>6s} {:>6s} {:>4s}".format("kernel", "VGPR", "SGPR", "scratch", "vspill", "sspill", "occ"))
for b in blocks:
    name = demangle(b.split("\n")[0].strip().split(" ")[0]).replace("void tfgx::(anonymous namespace)::", "")
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1)) if re.search(k + r": (\d+)", b) else -1
    row = (g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"Occupancy \[waves/SIMD\]"))
    if flt in name and (flt or row[2] > 0 or row[3] > 0):
        print("{:<100s} {:>5d} {:>5d} {:>7d} {:>6d} {:>6d} {:>4d}".format(name[:100], *row))
