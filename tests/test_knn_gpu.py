"""GPU parity: HIP kNN(k=2)+ratio through the C-ABI vs the CPU oracle. Bit-exact (indices, f32
distances, ratio decisions) -- integer work, no tolerance."""
import os

import numpy as np
import pytest

import oracle
from image_matching_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(ctx, q, t):
    fq, ft = ctx.features(q), ctx.features(t)
    assert fq.integral and ft.integral
    idx, dist = ctx.knn2(fq, ft)
    ridx, rdist = oracle.knn2(q, t, threads=8)
    assert np.array_equal(idx, ridx), np.nonzero((idx != ridx).any(1))[0][:10]
    assert np.array_equal(dist, rdist)
    return idx, dist


def test_golden_fixture(ctx):
    g = np.load(os.path.join(GOLD, "knn_q256_n4096.npz"))
    fq, ft = ctx.features(g["q"]), ctx.features(g["t"])
    idx, dist = ctx.knn2(fq, ft)
    assert np.array_equal(idx, g["idx"]) and np.array_equal(dist, g["dist"])
    # f32 storage (what cv2 SIFT returns) gives the same bits
    idx2, dist2 = ctx.knn2(ctx.features(g["q"].astype(np.float32)), ctx.features(g["t"].astype(np.float32)))
    assert np.array_equal(idx2, g["idx"]) and np.array_equal(dist2, g["dist"])


@pytest.mark.parametrize("nq,nt", [(1, 2), (3, 1), (5, 0), (0, 7), (31, 33), (64, 64), (129, 1000), (700, 5000),
                                   (1025, 4097), (2500, 20000)])
def test_ragged_sizes(ctx, nq, nt):
    t = synth.synth_descriptors(max(nt, 1), seed=11 + nt)[:nt]
    q = synth.synth_descriptors(max(nq, 1), seed=5 + nq)[:nq]
    if nt >= 8 and nq >= 8:
        q, _, _ = synth.synth_queries(t, nq, seed=nq, dup_frac=0.05)
    _check(ctx, q, t)


@pytest.mark.parametrize("nq,nt", [(513, 129), (6144, 16384), (6145, 30000), (8192, 16385), (7000, 100001), (12000, 50000),
                                   (600, 127), (5000, 1 << 17), (6000, 140001)])
def test_kernel_path_boundaries_with_ties(ctx, nq, nt):
    """Sizes either side of the launcher's switches (512 / 6144 queries, 128-row stages, 16 k-row chunks, 8e8 pairs: the
    256-query form below, the short-chunk 512-query form above) with a fifth of
    the train rows copied from other rows: equal distances across chunk and row-quarter boundaries, lowest index first."""
    t = synth.synth_descriptors(nt, seed=nt)
    rng = np.random.default_rng(nq)
    src = rng.integers(0, nt, nt // 5)
    dst = rng.integers(0, nt, nt // 5)
    t[dst] = t[src]
    q, _, _ = synth.synth_queries(t, nq, seed=nq, dup_frac=0.3)
    _check(ctx, q, t)


def test_extreme_values_exact(ctx):
    """All-255 queries vs (almost) all-0 rows: ssd = 128*255^2 = 8 323 200, the largest value the path can see."""
    t = np.zeros((300, 128), np.uint8)
    t[7, 100] = 1
    q = np.full((40, 128), 255, np.uint8)
    q[5, 17] = 254
    idx, dist = _check(ctx, q, t)
    assert idx[0].tolist() == [7, 0]
    assert dist[0, 1] == np.float32(np.sqrt(np.float32(8323200.0)))
    # and the other extreme mix: half-255 rows on both sides
    t[1::3, :64] = 255
    t[2::3] = 255
    q[::2, 64:] = 0
    _check(ctx, q, t)


def test_all_equal_ties_lowest_index(ctx):
    t = np.tile(synth.synth_descriptors(1, seed=2), (777, 1))
    q = np.tile(t[:1], (70, 1))
    idx, dist = _check(ctx, q, t)
    assert (idx[:, 0] == 0).all() and (idx[:, 1] == 1).all()


def test_ratio_and_dev_path(ctx):
    import torch
    t = synth.synth_descriptors(30000, seed=21)
    q, _, _ = synth.synth_queries(t, 3000, seed=22, noise=40)
    fq, ft = ctx.features(q), ctx.features(t)
    out4 = torch.empty((3000, 4), dtype=torch.int32, device="cuda:0")
    ctx.knn2_dev(fq, ft, out4.data_ptr(), idx_offset=1000)
    ctx.synchronize()
    o = out4.cpu().numpy()
    ridx, rdist = oracle.knn2(q, t, threads=8)
    assert np.array_equal(o[:, 1] - 1000, ridx[:, 0]) and np.array_equal(o[:, 3] - 1000, ridx[:, 1])
    assert np.array_equal(np.sqrt(o[:, 0].astype(np.float32)), rdist[:, 0])
    for r in (0.7, 0.9, 1.0):
        gq, gt, gd = ctx.ratio_dev(out4.data_ptr(), 3000, r)
        rq, rt, rd = oracle.ratio(ridx, rdist, r)
        assert np.array_equal(gq, rq) and np.array_equal(gt - 1000, rt) and np.array_equal(gd, rd)
    assert 0 < len(rq) < 3000


def test_sharded_merge_equals_whole(ctx):
    """top-2 over a union == merge of per-shard top-2, with ties across shard borders."""
    import torch
    t = synth.synth_descriptors(9000, seed=31)
    q, _, _ = synth.synth_queries(t, 1500, seed=32, dup_frac=0.1)
    fq = ctx.features(q)
    bounds = [0, 2500, 2501, 7000, 9000]
    lists = torch.empty((len(bounds) - 1, 1500, 4), dtype=torch.int32, device="cuda:0")
    keep = []
    for s in range(len(bounds) - 1):
        ft = ctx.features(t[bounds[s]:bounds[s + 1]])
        keep.append(ft)
        ctx.knn2_dev(fq, ft, lists[s].data_ptr(), idx_offset=bounds[s])
    out = torch.empty((1500, 4), dtype=torch.int32, device="cuda:0")
    ctx.merge_top2_dev(lists.data_ptr(), len(bounds) - 1, 1500, out.data_ptr())
    ctx.synchronize()
    o = out.cpu().numpy()
    ridx, rdist = oracle.knn2(q, t, threads=8)
    assert np.array_equal(o[:, [1, 3]], ridx)
    assert np.array_equal(np.sqrt(o[:, [0, 2]].astype(np.float32)), rdist)


def test_large_map_properties(ctx):
    """Size-independent checks at a size the oracle cannot finish: every query's source row (planted
    exactly) must come back at distance 0, planted ties in index order; a sampled subset is checked
    against the oracle."""
    import torch
    n, nq = 2_000_000, 4096
    g = torch.Generator(device="cuda:0").manual_seed(5)
    t_dev = torch.randint(0, 256, (n, 128), dtype=torch.uint8, device="cuda:0", generator=g)
    # sources on even rows, duplicates 12345 rows later (odd rows): the two sets cannot collide
    src = (torch.randperm((n - 20000) // 2, device="cuda:0", generator=g)[:nq] * 2).sort().values
    q_dev = t_dev[src].clone()
    hi = src + 12345
    t_dev[hi] = q_dev  # a duplicate at a higher index: exact tie at distance 0
    torch.cuda.synchronize()   # torch filled these on ITS stream; the context's stream reads them next
    ft = ctx.features_from_device_u8(t_dev.data_ptr(), n)
    fq = ctx.features_from_device_u8(q_dev.data_ptr(), nq)
    out4 = torch.empty((nq, 4), dtype=torch.int32, device="cuda:0")
    ctx.knn2_dev(fq, ft, out4.data_ptr())
    ctx.synchronize()
    o = out4.cpu().numpy()
    src_h, hi_h = src.cpu().numpy(), hi.cpu().numpy()
    ok = hi_h > src_h
    assert (o[ok, 0] == 0).all() and (o[ok, 2] == 0).all()
    assert np.array_equal(o[ok, 1], src_h[ok]) and np.array_equal(o[ok, 3], hi_h[ok])
    # oracle on a window of the map that contains the first 64 sources
    w = int(src_h[63]) + 20000
    w = min(w, 300000)
    sel = np.nonzero(hi_h < w)[0][:64]
    if len(sel):
        ft_w = ctx.features_from_device_u8(t_dev.data_ptr(), w)
        fq_w = ctx.features(q_dev[sel].cpu().numpy())
        idx, dist = ctx.knn2(fq_w, ft_w)
        ridx, rdist = oracle.knn2(q_dev[sel].cpu().numpy(), t_dev[:w].cpu().numpy(), threads=8)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)


def test_fractional_rows_are_refused_by_the_mfma_entry_point_and_rerouted_by_the_facade(ctx):
    """128 floats per row that are not integers 0..255 (RootSIFT, normalised rows): imatch_knn2 must not answer with
    clamped int8 neighbours and rc 0 -- the C call errors, the facade sends the matrices through the float kernel and
    gives BFMatcher's f32 answer (match_finder.py:182-183)."""
    import image_matching_amd as im
    from image_matching_amd.match_finder import match_finder
    rng = np.random.default_rng(12)
    t = np.sqrt(synth.synth_descriptors(3000, seed=3).astype(np.float32) / 512.0)      # RootSIFT-like rows
    q = (t[rng.integers(0, 3000, 200)] + rng.normal(0, 0.01, (200, 128))).astype(np.float32)
    fq, ft = ctx.features(q), ctx.features(t)
    assert not fq.integral and not ft.integral
    with pytest.raises(im.ImatchError, match="not integer valued"):
        ctx.knn2(fq, ft)
    ints = ctx.features(synth.synth_descriptors(100, seed=4))
    with pytest.raises(im.ImatchError, match="not integer valued"):
        ctx.knn2(ints, ft)                                 # one fractional side is enough
    with pytest.raises(im.ImatchError, match="not integer valued"):
        import torch
        out = torch.empty((100, 4), dtype=torch.int32, device="cuda:0")
        ctx.knn2_dev(ints, ft, out.data_ptr())
    mf = match_finder(ctx)
    idx, dist = mf.knn_match(q, t)
    ridx, rdist = oracle.knn2(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    # integer rows handed over as plain float matrices still take the MFMA path and its exact answer
    ti = synth.synth_descriptors(3000, seed=3)
    qi, _, _ = synth.synth_queries(ti, 200, seed=5, dup_frac=0.1)
    idx, dist = mf.knn_match(qi.astype(np.float32), ti.astype(np.float32))
    ridx, rdist = oracle.knn2(qi, ti)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
