# coding=utf-8
"""Synthetic inputs of SURVEY.md §8d (shared by bench.py and the examples; numpy only)."""
import numpy as np


def synthetic_edges(num_nodes, num_edges, seed=0):
    """E/2 uniform pairs on [0,N)^2, self pairs dropped, both directions emitted as [all (a,b) | all (b,a)] — the
    layout tf_geometric's convert_edge_to_directed produces (utils/graph_utils.py:186-190). Duplicates are kept
    (they sum). int32 [2, ~E]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    half = num_edges // 2
    a = rng.integers(0, num_nodes, size=half, dtype=np.int32)
    b = rng.integers(0, num_nodes, size=half, dtype=np.int32)
    keep = a != b
    a, b = a[keep], b[keep]
    return np.stack([np.concatenate([a, b]), np.concatenate([b, a])]).astype(np.int32)


PAIR_BLOCK = 1 << 20          # pairs per independently seeded block of synthetic_edge_stripe
ROW_BLOCK = 1 << 16           # rows per independently seeded block of synthetic_feature_rows


def _pair_block(num_nodes, count, seed, k):
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), int(k)])))
    a = rng.integers(0, num_nodes, size=count, dtype=np.int32)
    b = rng.integers(0, num_nodes, size=count, dtype=np.int32)
    keep = a != b
    return a[keep], b[keep]


def synthetic_edge_stripe(num_nodes, num_edges, seed=0, stripe=0, num_stripes=1):
    """The same family of graphs as synthetic_edges (E/2 uniform pairs, self pairs dropped, both directions emitted), but
    drawn in blocks of PAIR_BLOCK pairs, block k from its own stream SeedSequence([seed, k]) — so stripe `stripe` of
    `num_stripes` (a contiguous range of blocks) can be generated ALONE: rank r of an N-GPU job builds only its E/N edges,
    and the union over the stripes is the same edge multiset for every num_stripes (bench.py: the same graph at every N).
    A stripe is [all (a,b) of its blocks | all (b,a) of its blocks]; with num_stripes = 1 that is synthetic_edges'
    layout.  int32 [2, ~E/num_stripes]."""
    half = num_edges // 2
    n_blocks = -(-half // PAIR_BLOCK) if half else 0
    lo, hi = (n_blocks * stripe) // num_stripes, (n_blocks * (stripe + 1)) // num_stripes
    blocks = [_pair_block(num_nodes, min(PAIR_BLOCK, half - k * PAIR_BLOCK), seed, k) for k in range(lo, hi)]
    tot = sum(int(a.shape[0]) for a, _ in blocks)
    out = np.empty((2, 2 * tot), dtype=np.int32)           # filled in place: one pass over the 1 GB at products size
    pos = 0
    for a, b in blocks:
        m = int(a.shape[0])
        out[0, pos:pos + m], out[1, pos:pos + m] = a, b
        out[0, tot + pos:tot + pos + m], out[1, tot + pos:tot + pos + m] = b, a
        pos += m
    return out


def synthetic_feature_rows(num_nodes, num_features, seed=1, row_lo=0, row_hi=None):
    """Rows [row_lo, row_hi) of a standard-normal [N, F] float32 matrix drawn in blocks of ROW_BLOCK rows (block k from
    SeedSequence([seed, k])): any row range costs only its own blocks, and every rank sees the same matrix."""
    row_hi = num_nodes if row_hi is None else row_hi
    out = np.empty((max(row_hi - row_lo, 0), num_features), dtype=np.float32)
    for k in range(row_lo // ROW_BLOCK, -(-row_hi // ROW_BLOCK) if row_hi > row_lo else 0):
        b0, b1 = k * ROW_BLOCK, min((k + 1) * ROW_BLOCK, num_nodes)
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), int(k)])))
        blk = rng.standard_normal((b1 - b0, num_features), dtype=np.float32)
        s0, s1 = max(b0, row_lo), min(b1, row_hi)
        out[s0 - row_lo:s1 - row_lo] = blk[s0 - b0:s1 - b0]
    return out


def synthetic_features(num_nodes, num_features, seed=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal((num_nodes, num_features), dtype=np.float32)


def synthetic_edge_weight(num_edges, seed=2):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.uniform(0.5, 1.5, size=num_edges).astype(np.float32)


def glorot_uniform(fan_in, fan_out, seed=3):
    rng = np.random.Generator(np.random.PCG64(seed))
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=(fan_in, fan_out)).astype(np.float32)


# R-MAT variant of SURVEY.md §8d: (a,b,c,d) = (0.57,0.19,0.19,0.05), generated on the GPU (torch; imported lazily)
def rmat_edges(n, e, seed, dev):
    """E/2 R-MAT pairs over 2^ceil(log2 n) ids, pairs with an id >= n or a == b dropped, both directions emitted
    [all (a,b) | all (b,a)] like the uniform generator.  int32 [2, ~E] on the device."""
    import torch
    k = max(1, int(np.ceil(np.log2(max(n, 2)))))
    half = e // 2
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    got_a, got_b, have = [], [], 0
    a_, b_, c_ = 0.57, 0.19, 0.19
    for _ in range(8):
        m = int((half - have) * 1.7) + 1024
        src = torch.zeros(m, dtype=torch.int64, device=dev)
        dst = torch.zeros(m, dtype=torch.int64, device=dev)
        for _lvl in range(k):
            r = torch.rand(m, generator=g, device=dev)
            sbit = r >= (a_ + b_)
            dbit = ((r >= a_) & (r < a_ + b_)) | (r >= a_ + b_ + c_)
            src = src * 2 + sbit
            dst = dst * 2 + dbit
        keep = (src < n) & (dst < n) & (src != dst)
        got_a.append(src[keep])
        got_b.append(dst[keep])
        have += int(keep.sum().item())
        if have >= half:
            break
    a = torch.cat(got_a)[:half].to(torch.int32)
    b = torch.cat(got_b)[:half].to(torch.int32)
    return torch.stack([torch.cat([a, b]), torch.cat([b, a])]).contiguous()


WORKLOADS = {
    # name: (nodes, edges, features)  — BASELINE.json configs
    "products": (2400000, 123000000, 100),   # ogbn-products-shaped: north-star target (segment-sum, 1 GPU)
    "arxiv": (170000, 1200000, 128),         # ogbn-arxiv-shaped (configs[1])
    "reddit": (233000, 114000000, 602),      # Reddit-shaped (configs[2]: 8-head GAT)
    "cora": (2708, 10556, 1433),             # Cora-shaped (configs[0])
    "tiny": (20000, 400000, 100),            # plumbing check
}
