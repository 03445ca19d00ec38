# coding=utf-8
"""Builds tf_geometric_amd/lib/libtfgx.so (the C-ABI HIP library, gfx950 only) in-tree with hipcc."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtfgx.so")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
SOURCES = ["tfgx_plan.hip", "tfgx_reduce.hip", "tfgx_norm.hip", "tfgx_attn.hip", "tfgx_gemm.hip", "tfgx_misc.hip", "tfgx_backward.hip",
           "tfgx_topk.hip", "tfgx_fused.hip", "tfgx_poolgrad.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("TFGX_EXTRA_HIPCC_FLAGS", "").split()      # developer A/B switches (e.g. -DTFGX_PREFETCH_INDEX=0)


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, "tfgx_common.h"), os.path.join(_HERE, "..", "include", "tfgx.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(LIB_PATH):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_dist(force=force or bool(jobs), verbose=verbose)
    build_line_rate_probe(force=force, verbose=verbose)
    build_c_abi_demo(force=force or bool(jobs), verbose=verbose)
    build_tf_shim_mock(force=force or bool(jobs), verbose=verbose)
    return LIB_PATH


DIST_SRC = os.path.join(CSRC, "tfgx_dist.cpp")
DIST_LIB = os.path.join(LIB_DIR, "libtfgx_dist.so")
HALO_DEMO_SRC = os.path.join(_HERE, "..", "examples", "c_abi_halo_demo.cpp")
HALO_DEMO_BIN = os.path.join(LIB_DIR, "c_abi_halo_demo")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_dist(force=False, verbose=True):
    """lib/libtfgx_dist.so: the halo-exchange C ABI (include/tfgx_dist.h) = host code over libtfgx.so + librccl, kept out
    of the compute library so libtfgx.so has no communication dependency; and its torch-free demo program."""
    inc = os.path.join(_HERE, "..", "include")
    hdr = os.path.join(inc, "tfgx_dist.h")
    if force or _newer(DIST_SRC, DIST_LIB) or _newer(hdr, DIST_LIB) or _newer(LIB_PATH, DIST_LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROCM, "include"),
               DIST_SRC, "-o", DIST_LIB, "-L", LIB_DIR, "-ltfgx", "-L", os.path.join(ROCM, "lib"), "-lrccl",
               "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if os.path.exists(HALO_DEMO_SRC) and (force or _newer(HALO_DEMO_SRC, HALO_DEMO_BIN) or _newer(DIST_LIB, HALO_DEMO_BIN)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-I", inc, "-I", os.path.join(ROCM, "include"), HALO_DEMO_SRC,
               "-L", LIB_DIR, "-ltfgx_dist", "-ltfgx", "-L", os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath,$ORIGIN",
               "-o", HALO_DEMO_BIN]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return DIST_LIB


PROBE_SRC = os.path.join(_HERE, "..", "tools", "line_rate_probe.cpp")
PROBE_BIN = os.path.join(LIB_DIR, "line_rate_probe")


def build_line_rate_probe(force=False, verbose=True):
    """lib/line_rate_probe: the random-line-rate probe (tools/line_rate_probe.cpp — a measurement utility, no part of the
    library).  bench.py runs it on the SAME box for the roofs of the cache-resident configs: random aligned spans served by
    a table of the size the timed kernel gathers from (an L2-sized source block, an Infinity-Cache-sized feature table)."""
    if not os.path.exists(PROBE_SRC):
        return None
    if force or _newer(PROBE_SRC, PROBE_BIN):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", PROBE_SRC, "-o", PROBE_BIN]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return PROBE_BIN


DEMO_SRC = os.path.join(_HERE, "..", "examples", "c_abi_demo.cpp")
DEMO_BIN = os.path.join(LIB_DIR, "c_abi_demo")


def build_c_abi_demo(force=False, verbose=True):
    """examples/c_abi_demo.cpp: a HIP host program (no Python, no torch) linked against libtfgx.so through
    include/tfgx.h — run by tests/test_gpu_c_abi.py on the GPU box."""
    if not os.path.exists(DEMO_SRC):
        return None
    if force or _newer(DEMO_SRC, DEMO_BIN) or _newer(LIB_PATH, DEMO_BIN):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-I", os.path.join(_HERE, "..", "include"), DEMO_SRC,
               "-L", LIB_DIR, "-ltfgx", "-Wl,-rpath,$ORIGIN", "-o", DEMO_BIN]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return DEMO_BIN


SHIM_DIR = os.path.join(_HERE, "..", "integration", "tf_shim")
SHIM_MOCK_BIN = os.path.join(LIB_DIR, "tf_shim_mock_driver")


def build_tf_shim_mock(force=False, verbose=True):
    """lib/tf_shim_mock_driver: integration/tf_shim/tfgx_tf_ops.cc (the tf.load_op_library binding) LINKED against the mock
    TensorFlow runtime of integration/tf_shim/mock/ and libtfgx.so / libtfgx_dist.so, with a script-driven main — every
    Compute() body of the shim runs on the GPU in tests/test_gpu_tf_shim.py.  (TensorFlow itself is not in the image.)"""
    srcs = [os.path.join(SHIM_DIR, "tfgx_tf_ops.cc"), os.path.join(SHIM_DIR, "mock", "mock_runtime.cc"),
            os.path.join(SHIM_DIR, "mock", "mock_driver.cc")]
    if not all(os.path.exists(s) for s in srcs):
        return None
    fw = os.path.join(SHIM_DIR, "mock", "tensorflow", "core", "framework")
    deps = srcs + [os.path.join(SHIM_DIR, "mock", "mock_runtime.h"), os.path.join(fw, "op.h"), os.path.join(fw, "op_kernel.h"),
                   LIB_PATH, DIST_LIB]
    if force or any(_newer(d, SHIM_MOCK_BIN) for d in deps):
        inc = os.path.join(_HERE, "..", "include")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", inc, "-I", os.path.join(SHIM_DIR, "mock")] + srcs + \
              ["-L", LIB_DIR, "-ltfgx_dist", "-ltfgx", "-L", os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath,$ORIGIN",
               "-o", SHIM_MOCK_BIN]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SHIM_MOCK_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)


ASAN_DIR = os.path.join(LIB_DIR, "asan")
ASAN_LIB = os.path.join(ASAN_DIR, "libtfgx.so")


def asan_runtime():
    """Path of clang's shared AddressSanitizer runtime (must be LD_PRELOADed into an uninstrumented host process)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROCM, "lib", "llvm", "lib", "clang", "*", "lib", "linux",
                                         "libclang_rt.asan-x86_64.so")))
    return hits[-1] if hits else None


def build_asan(force=False, verbose=True):
    """lib/asan/libtfgx.so: the same sources with the HOST side instrumented by AddressSanitizer (the
    memory-error detector of the plan; device code is left alone: -fno-gpu-sanitize).  Used by
    tests/test_abi.py::test_host_code_under_address_sanitizer and tools/asan_host_check.py."""
    os.makedirs(ASAN_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, x) for x in SOURCES]
    deps = srcs + [os.path.join(CSRC, "tfgx_common.h"), os.path.join(_HERE, "..", "include", "tfgx.h")]
    if not force and os.path.exists(ASAN_LIB) and not any(_newer(d, ASAN_LIB) for d in deps):
        return ASAN_LIB
    flags = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize",
             "-shared-libsan", "-Wno-unused-function"]

    def compile_one(src):
        obj = os.path.join(ASAN_DIR, os.path.basename(src).replace(".hip", ".o"))
        cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan",
           "-o", ASAN_LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return ASAN_LIB
