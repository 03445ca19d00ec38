# coding=utf-8
"""bench.py as the driver runs it: the PLAIN command (`python3 bench.py --gpus N ...`, no launcher around it)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(argv, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TFGX_BENCH_BACKEND"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def _json_line(res):
    lines = res.stdout.decode().splitlines()          # stdout carries the ONE JSON line and nothing else (no gloo / RCCL banners)
    assert len(lines) == 1 and lines[0].startswith("{"), (res.stdout.decode()[-2000:], res.stderr.decode()[-3000:])
    return json.loads(lines[0])


def test_plain_command_starts_its_own_ranks(tfg):
    """`python3 bench.py --gpus 2` with WORLD_SIZE unset launches its two ranks itself (torch.distributed.run on
    127.0.0.1) and rank 0 prints the one JSON line.  On a one-GPU box the exchange cannot be on RCCL (two ranks per device
    are refused), so this is the explicitly requested plumbing mode: gloo control plane, rows staged through the host,
    and the line says so."""
    import torch
    res = _run(["--gpus", "2", "--workload", "tiny", "--steps", "3", "--warmup", "1"], env={"TFGX_BENCH_BACKEND": "gloo"})
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    line = _json_line(res)
    cfg = line["config"]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "strong"
    assert cfg["transport"] == "torch" and cfg["rccl_ranks"] == 0 and "plumbing_check" in cfg
    assert cfg["devices_visible"] == torch.cuda.device_count()
    ranks = line["roofline"]["per_rank"]
    assert [d["rank"] for d in ranks] == [0, 1] and sum(d["edges"] for d in ranks) == cfg["edges"]
    assert all(d["halo_rows_received"] > 0 and d["exchange_GBps_received"] > 0 for d in ranks)


def test_more_ranks_than_gpus_is_refused_loudly(tfg):
    """Without the plumbing switch a rank count above the visible GPUs must fail (non-zero, a message that says why):
    a scaling line is only ever carried by the tfgx_dist RCCL communicator, one device per rank."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has a GPU per rank")
    res = _run(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"], timeout=600)
    assert res.returncode != 0
    assert "RCCL needs one device per rank" in res.stderr.decode(), res.stderr.decode()[-2000:]
    assert not [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]


def test_single_gpu_line_shape(tfg):
    """N = 1 on a small workload: the contract keys, the roofline / cpu_baseline objects, parity of the timed output."""
    res = _run(["--workload", "tiny", "--steps", "3", "--warmup", "1"])
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    line = _json_line(res)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["dtype"] == "f32" and line["roofline"]["bound"] == "hbm"
    assert 0 < line["roofline"]["frac"] < 1.0 and line["cpu_baseline"]["kind"] == "port"
    assert line["parity_vs_cpu_port_max_abs_err"] < 1e-4


def test_watchdog_turns_a_hang_in_the_first_exchange_into_a_reason(tfg):
    """A rank stuck inside its first step (what a blocked grouped ncclSend / ncclRecv looks like from Python) must end the
    run within the watchdog's limit: exit code != 0, ONE JSON line with "error" naming the phase on stdout — not the
    driver's 1800 s timeout with nothing to read."""
    res = _run(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1"],
               env={"TFGX_BENCH_BACKEND": "gloo", "TFGX_BENCH_TEST_HANG": "first-step", "TFGX_BENCH_TEST_HANG_RANK": "1",
                    "TFGX_BENCH_WATCHDOG_S": "20"}, timeout=600)
    assert res.returncode != 0
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (res.stdout.decode()[-2000:], res.stderr.decode()[-3000:])
    line = json.loads(lines[0])
    assert line["value"] is None and "made no progress" in line["error"] and line["n_gpus"] == 2
    assert "made no progress in phase" in res.stderr.decode()  # whichever rank's timer fires first reports (the stuck one, or its peer waiting for it)


def test_products_shaped_eight_rank_plumbing_run(tfg):
    """The products-shaped graph through the `--gpus 8` code path on ONE GPU (host-staged transport, explicitly requested):
    eight stripes generated, routed and sharded, every peer dense, one line with eight per-rank records.  Not a scaling
    number — the plumbing the driver's 8-GPU run goes through, at the driver's own shape."""
    res = _run(["--gpus", "8", "--steps", "2", "--warmup", "1"], env={"TFGX_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    line = _json_line(res)
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["nodes"] == 2400000 and cfg["transport"] == "torch" and "plumbing_check" in cfg
    ranks = line["roofline"]["per_rank"]
    assert [d["rank"] for d in ranks] == list(range(8)) and sum(d["edges"] for d in ranks) == cfg["edges"]
    assert max(d["edges"] for d in ranks) <= 1.05 * cfg["edges"] / 8           # edge-balanced destination ranges
    assert cfg["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_default_line_configs_carry_a_real_roof_each(tfg):
    """The driver's default command (bounded here: fewer steps, no CPU legs, no R-MAT line): every BASELINE config sits in the
    `configs` block with a `roofline` object against the resource that bounds it — HBM for the tables far beyond the caches
    (C4, C5), the L2s for the source-blocked attention (C3, both attention widths), the Infinity Cache's probed line rate for
    the arxiv-shaped table (C2) — and NO fraction anywhere in the line exceeds 1 (VERDICT r5 missing #4 / #5: a `frac` above 1
    means the roof is wrong)."""
    res = _run(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-rmat"], timeout=1200)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    line = _json_line(res)
    cfgs = line["configs"]
    want = {"C2_arxiv_gcn_2layer": "mall", "C3_reddit_gat_H8_A8": "l2", "C3_reddit_gat_H8_A64": "l2",
            "C4_mean_sage_256_concat": "hbm", "C4_maxpool_sage_256_concat": "hbm", "C5_papers100M_one_shard_of_8": "hbm"}
    for key, bound in want.items():
        assert key in cfgs and "error" not in cfgs[key] and "skipped" not in cfgs[key], (key, cfgs.get(key))
        r = cfgs[key]["roofline"]
        assert r["bound"] == bound and r["peak"] and 0.0 < r["frac"] <= 1.0, (key, r)
    fracs = []

    def walk(node, path):
        if isinstance(node, dict):
            for k, v in node.items():
                assert "frac_of_hbm_peak" not in k, path + "/" + k
                if k.startswith("frac") and isinstance(v, (int, float)):
                    fracs.append((path + "/" + k, v))
                walk(v, path + "/" + k)
        elif isinstance(node, list):
            for i, v in enumerate(node):
                walk(v, "{}[{}]".format(path, i))
    walk(line, "")
    assert fracs and all(0.0 <= v <= 1.0 for _, v in fracs), [f for f in fracs if not 0.0 <= f[1] <= 1.0]
    assert cfgs["C3_reddit_gat_H8_A8"]["fwd_bwd_ms"] > cfgs["C3_reddit_gat_H8_A8"]["forward_ms"] > 0
