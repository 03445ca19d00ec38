# coding=utf-8
"""The block-seeded synthetic generators bench.py shards its workload with (tf_geometric_amd/synthetic.py): any stripe can be
generated alone, and the union over the stripes is the same edge multiset for every number of stripes."""
import numpy as np

from tf_geometric_amd import synthetic as S


def _keys(ei, n):
    return np.sort(ei[0].astype(np.int64) * n + ei[1])


def test_edge_stripes_union_is_independent_of_the_stripe_count():
    n, e = 50000, 5 * S.PAIR_BLOCK + 12345            # 3 blocks of pairs (e/2), the last one partial
    whole = S.synthetic_edge_stripe(n, e, seed=4)
    h = whole.shape[1] // 2
    assert whole.dtype == np.int32 and whole.shape[0] == 2 and (whole[0] != whole[1]).all()
    assert np.array_equal(whole[0, :h], whole[1, h:]) and np.array_equal(whole[1, :h], whole[0, h:])    # [all (a,b) | all (b,a)]
    assert abs(whole.shape[1] - e) < 400 and whole.min() >= 0 and whole.max() < n       # only the a == b pairs are dropped
    for parts in (2, 3, 8):
        stripes = [S.synthetic_edge_stripe(n, e, seed=4, stripe=r, num_stripes=parts) for r in range(parts)]
        assert np.array_equal(_keys(np.concatenate(stripes, axis=1), n), _keys(whole, n))
    assert not np.array_equal(_keys(S.synthetic_edge_stripe(n, e, seed=5), n), _keys(whole, n))
    assert S.synthetic_edge_stripe(n, 0).shape == (2, 0)


def test_feature_rows_are_the_same_matrix_from_any_row_range():
    n, f = 3 * S.ROW_BLOCK + 17, 5
    x = S.synthetic_feature_rows(n, f, seed=9)
    assert x.shape == (n, f) and x.dtype == np.float32
    for lo, hi in ((0, 10), (S.ROW_BLOCK - 3, S.ROW_BLOCK + 3), (2 * S.ROW_BLOCK, n), (n - 1, n), (7, 7)):
        assert np.array_equal(S.synthetic_feature_rows(n, f, seed=9, row_lo=lo, row_hi=hi), x[lo:hi])
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1.0) < 0.01


def test_bench_plain_command_builds_its_own_launcher_line(monkeypatch):
    """`python3 bench.py --gpus N` without a launcher around it re-executes itself under torch.distributed.run on the loopback
    interface with its own arguments (no GPU needed to check the command line)."""
    import os
    import sys
    import bench
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit:
        pass
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in a and a[a.index("--nproc-per-node") + 1] == "4" and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(a[a.index("--master-port") + 1]) < 65536
    assert a[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and a[-7].endswith("bench.py")
