# coding=utf-8
"""Host-side behaviour of bench.py that needs no GPU: the watchdog and the environment it hands to its ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env, timeout=120):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TFGX_BENCH_BACKEND", "HSA_ENABLE_IPC_MODE_LEGACY"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_watchdog_reports_a_stuck_phase_and_exits_nonzero():
    """A phase that never returns (here: a test hook before HIP is touched) ends the process within the limit with ONE
    JSON line carrying "error" and the phase name on stdout, exit code 3 — never a silent hang."""
    res = _run(["--workload", "tiny"], {"TFGX_BENCH_TEST_HANG": "start", "TFGX_BENCH_WATCHDOG_S": "1"})
    assert res.returncode == 3, (res.returncode, res.stderr.decode()[-2000:])
    lines = res.stdout.decode().splitlines()
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] is None and "TEST HANG in 'start'" in line["error"] and line["n_gpus"] == 1
    assert line["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # defaulted by bench.py itself before torch is imported
    assert "made no progress" in res.stderr.decode()


def test_ipc_mode_is_defaulted_where_the_multi_rank_paths_live():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC): defaulted at import of tf_geometric_amd.dist, a value
    the user exported wins, and transport.ipc_mode_note() says what the process runs with."""
    code = ("import os; os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None); import tf_geometric_amd.dist.transport as t; "
            "print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'], t.ipc_mode_note())")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode().split()
    assert out[0] == "0" and out[1] == "None"
    code = ("import os; os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '1'; import tf_geometric_amd.dist.transport as t; "
            "print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY']); print(t.ipc_mode_note())")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    assert out[0] == "1" and "dmabuf" in out[1]


def test_a_stale_error_lock_of_an_earlier_job_does_not_silence_the_watchdog(tmp_path):
    """the rank that writes the error line is elected with an O_EXCL file.  Named after the launcher's pid alone, a
    file left by an earlier job whose launcher had the same (recycled) pid made every rank of a later job stand back: exit 3
    and NO line.  The name now carries the launcher's start time and the rendezvous port; a stale file of the old form, or of
    another job, changes nothing."""
    me = os.getpid()                                   # the launcher of the rank started below
    stale = [tmp_path / "tfgx_bench_error_{}.lock".format(me), tmp_path / "tfgx_bench_error_{}_0_29400.lock".format(me)]
    for p in stale:
        p.write_text("")
    env = {"TFGX_BENCH_TEST_HANG": "start", "TFGX_BENCH_WATCHDOG_S": "1", "RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2",
           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29400", "TMPDIR": str(tmp_path)}
    res = _run(["--workload", "tiny", "--gpus", "2"], env)
    assert res.returncode == 3, (res.returncode, res.stderr.decode()[-2000:])
    lines = res.stdout.decode().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["reported_by_rank"] == 1
    # the SAME job's second rank (same launcher, same port) finds the first one's file and stays quiet
    res2 = _run(["--workload", "tiny", "--gpus", "2"], dict(env, RANK="0", LOCAL_RANK="0"))
    assert res2.returncode == 3 and res2.stdout.decode().strip() == ""
