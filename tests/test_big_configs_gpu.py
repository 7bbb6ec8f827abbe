"""GPU parity at BASELINE.json's full sizes, through the C-ABI.

configs[1]  CLAHE + SIFT on the whole 8192 x 8192 map against the oracle on the same image: every keypoint, in order.
configs[3]  CLAHE + SIFT on a whole 16384 x 16384 map (the map leg): pyramid bands, list order, a central crop through the oracle.
configs[3]  kNN 4096 queries x 5 M descriptors (the whole map of the 8-GPU layout on one GPU).
configs[4]  kNN 131072 queries x 2.5 M descriptors (one of the eight shards of the 20 M-row map).
            For both: SURVEY 8(d) descriptor statistics, planted exact ties (lowest index first), every reported
            (index, ssd) pair recomputed from the rows, nobody may be worse than his own source row, an exact full-width
            top-2 (f32 GEMM of integer rows: every sum is an integer below 2^24) for a sample of the queries, and the CPU
            oracle on a window of <= 300 k rows."""
import numpy as np
import pytest

import oracle
from image_matching_amd import synth

pytestmark = pytest.mark.gpu


# ---- configs[1] --------------------------------------------------------------------------------------------------------
def _big_map():
    """8192 x 8192 from four seeded 2048 x 2048 synthetic maps, turned and mirrored (the 8192^2 generator itself takes 12 s)."""
    base = [synth.synth_map(2048, seed=40 + i) for i in range(4)]
    rows = []
    for r in range(4):
        row = []
        for c in range(4):
            b = base[(r + c) % 4]
            k = (r * 4 + c) % 8
            b = np.rot90(b, k % 4)
            row.append(b.T if k >= 4 else b)
        rows.append(np.concatenate(row, axis=1))
    return np.ascontiguousarray(np.concatenate(rows, axis=0))


def test_sift_whole_8192_map_against_the_oracle(ctx):
    """find_kp_dp's extraction (CLAHE + SIFT, match_finder.py:199-221) of the whole 8192 x 8192 map -- the large-image path:
    16384^2 base octave, staggered layers, two streams, one wave per keypoint in the descriptor stage -- against the CPU
    oracle on the same image (about 10 s on the box's host cores)."""
    m = _big_map()
    f = ctx.sift_extract(m)
    kp, desc = f.download()
    eq = ctx.clahe(m)
    assert np.array_equal(eq, oracle.clahe(m))
    res = oracle.sift(eq)
    assert len(kp) > 150_000 and len(kp) == len(res.kp)
    # same order, same positions, same packed octave | layer | xi field and response, bit for bit
    assert np.array_equal(kp[:, :2], res.kp[:, :2])
    assert np.array_equal(kp[:, 4:], res.kp[:, 4:])
    assert np.abs(kp[:, 2] - res.kp[:, 2]).max() <= 1e-4                         # size: last place of an exp2f
    dang = np.abs((kp[:, 3] - res.kp[:, 3] + 180.0) % 360.0 - 180.0)
    assert dang.max() < 1.0 and np.percentile(dang, 99) < 1e-2                   # device atan2f / expf against glibc's
    l2 = np.linalg.norm(desc - res.desc, axis=1)
    assert (l2 <= 16).mean() >= 0.99 and (l2 == 0).mean() >= 0.995, ((l2 <= 16).mean(), (l2 == 0).mean(), l2.max())
    f.free()


# ---- configs[3], the map leg: a 16384 x 16384 orthomap -------------------------------------------------------------------------
def _map_16k():
    """16384 x 16384: an 8 x 8 tiling of four seeded 2048^2 synthetic maps (turned and mirrored) with seeded +-3 noise on top, so
    that no two tiles are alike (the 16384^2 generator itself needs a minute and several GB)."""
    base = [synth.synth_map(2048, seed=60 + i) for i in range(4)]
    rows = []
    for r in range(8):
        row = []
        for c in range(8):
            b = base[(r + 2 * c) % 4]
            k = (r * 8 + c) % 8
            b = np.rot90(b, k % 4)
            row.append(b.T if k >= 4 else b)
        rows.append(np.concatenate(row, axis=1))
    img = np.concatenate(rows, axis=0).astype(np.int16)
    rng = np.random.default_rng(77)
    for y in range(0, 16384, 2048):          # in bands: the noise of the whole image at once is 0.5 GB of int16
        img[y:y + 2048] += rng.integers(-3, 4, (2048, 16384), dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def _kp_order_violations(kp):
    """KeyPoint12_LessThan (x, y ascending; size descending; angle ascending; response, octave descending) between
    neighbours of the list: number of pairs out of order, and of pairs removeDuplicatedSorted would have dropped."""
    keys = [kp[:, 0], kp[:, 1], -kp[:, 2], kp[:, 3], -kp[:, 4], -kp[:, 5]]
    eq = np.ones(len(kp) - 1, bool)
    gt = np.zeros(len(kp) - 1, bool)
    for k in keys:
        a, b = k[:-1], k[1:]
        gt |= eq & (a > b)
        eq &= a == b
    dup = np.all(kp[:-1, :4] == kp[1:, :4], axis=1)
    return int(gt.sum()), int(dup.sum())


def test_sift_whole_16384_map_bands_order_and_a_central_crop(ctx):
    """configs[3]'s map leg: find_kp_dp's extraction (match_finder.py:199-221, as position_finder.py:248-262 drives it over a
    map) of a whole 16384 x 16384 image -- a 34 GB arena, 32768^2 layers of 4.3 GB staggered in memory, 64-row scan walks, two
    streams. The oracle cannot hold that pyramid on the host, so:
      * CLAHE of the whole image: bit-equal to the oracle's;
      * octaves 0 and 1, every Gaussian layer, in three row bands (top edge, middle, bottom edge): bit-equal to
        oracle_gauss_blur of the SAME rows (+- the kernel radius) of the layer below, which is exact for a band -- the base
        layer from the equalised image's rows through the oracle's x2 resize, octave 1's layer 0 = every second pixel of
        octave 0's layer 3;
      * the keypoint list: in bounds, in KeyPoint12_LessThan order, no duplicates;
      * the oracle's SIFT on a central 4096^2 crop of the equalised image (origin a multiple of 2048, so the octave grids
        coincide): each of its keypoints of octaves <= 3 at least 512 pixels inside the crop -- further than the blur chain's and
        the descriptor window's reach, so the crop's border cannot have touched it -- is in the GPU's list with the same
        octave | layer | xi field and response bit for bit, position within the float rounding of the offset (the full image
        forms (c + xc) * 2^o with c near 30 000), size within an ulp, angle as elsewhere, descriptors >= 99.5 % identical; and
        the other way round, every GPU keypoint of that region is one of the oracle's."""
    from scipy.spatial import cKDTree
    side = 16384
    m = _map_16k()
    f = ctx.sift_extract(m)
    kp, desc = f.download()
    assert len(kp) > 600_000
    eq = f.image(side, side)
    want_eq = oracle.clahe(m)
    assert np.array_equal(eq, want_eq)
    del want_eq, m
    # ---- bands of octaves 0 and 1
    sigma, nl = 1.6, 3
    kf = 2.0 ** (1.0 / nl)
    sig = [sigma] + [float(np.sqrt((sigma * kf ** i) ** 2 - (sigma * kf ** (i - 1)) ** 2)) for i in range(1, nl + 3)]
    sig_diff = float(np.sqrt(np.float32(max(np.float32(sigma) * np.float32(sigma) - np.float32(1.0), np.float32(0.01)))))
    radius = lambda s: (int(round(s * 8 + 1)) | 1) // 2
    assert ctx.sift_debug_octaves() == 14
    nb = 96
    for o in (0, 1):
        rows = (2 * side) >> o
        for r0 in (0, rows // 2 - 40, rows - nb):
            # layer 0
            if o == 0:
                R = radius(sig_diff)
                a, b = max(r0 // 2 - R, 0), min((r0 + nb) // 2 + R, side)            # rows of the equalised image
                up = oracle.upsample2(eq[a:b])                                          # rows 2a .. 2b of the x2 image
                # exact rows of the band's x2 image: all but the first and last (which would need rows a - 1 and b)
                lo_ok = 2 * a + (1 if a > 0 else 0)
                hi_ok = 2 * b - (1 if b < side else 0)
                want = oracle.gauss_blur(up[lo_ok - 2 * a:hi_ok - 2 * a], sig_diff)
                # ... and of its blur: R rows further in, except at the image's own edges
                t0 = lo_ok + (R if lo_ok > 0 else 0)
                t1 = hi_ok - (R if hi_ok < rows else 0)
                t0, t1 = max(t0, r0), min(t1, r0 + nb)
                got = ctx.sift_debug_layer_rows(0, 0, t0, t1 - t0)
                assert t1 - t0 >= nb - 2 and np.array_equal(got, want[t0 - lo_ok:t1 - lo_ok]), (o, 0, r0)
            else:
                got = ctx.sift_debug_layer_rows(1, 0, r0, nb)
                src = ctx.sift_debug_layer_rows(0, nl, 2 * r0, 2 * nb)
                assert np.array_equal(got, src[::2, ::2]), (o, 0, r0)
            for i in range(1, nl + 3):
                R = radius(sig[i])
                a, b = max(r0 - R, 0), min(r0 + nb + R, rows)
                src = ctx.sift_debug_layer_rows(o, i - 1, a, b - a)
                want = oracle.gauss_blur(src, sig[i])[r0 - a:r0 - a + nb]
                got = ctx.sift_debug_layer_rows(o, i, r0, nb)
                assert np.array_equal(got, want), (o, i, r0)
    # ---- the list
    assert (kp[:, 0] >= 0).all() and (kp[:, 0] < side).all() and (kp[:, 1] >= 0).all() and (kp[:, 1] < side).all()
    assert (kp[:, 2] > 0).all() and (kp[:, 3] >= 0).all() and (kp[:, 3] < 360).all() and (kp[:, 4] > 0).all()
    assert _kp_order_violations(kp) == (0, 0)
    assert desc.min() >= 0 and desc.max() <= 255 and np.array_equal(desc, np.rint(desc))
    # ---- a central crop through the oracle
    c0, cs, margin = 6144, 4096, 512
    res = oracle.sift(eq[c0:c0 + cs, c0:c0 + cs])
    okp = res.kp.copy()
    okp[:, 0] += c0
    okp[:, 1] += c0

    def octave_of(k):       # pyramid octave: the packed field's low byte is signed, first octave -1
        o = k[:, 5].astype(np.int64) & 255
        return np.where(o >= 128, o - 256, o) + 1

    def inner(k, shrink):
        lo, hi = c0 + margin + shrink, c0 + cs - margin - shrink
        return (k[:, 0] >= lo) & (k[:, 0] < hi) & (k[:, 1] >= lo) & (k[:, 1] < hi) & (octave_of(k) <= 3)

    def match(a, a_desc, b, b_desc):
        """every record of a has a partner in b: -> (number unmatched, identical-descriptor share, worst angle difference)"""
        tree = cKDTree(b[:, :2].astype(np.float64))
        near = tree.query_ball_point(a[:, :2].astype(np.float64), r=8e-3)
        missing, same, worst = 0, 0, 0.0
        for j, cand in enumerate(near):
            best = None
            for c in cand:
                if b[c, 5] != a[j, 5] or b[c, 4] != a[j, 4] or abs(b[c, 2] - a[j, 2]) > 1e-4:
                    continue
                d = abs((b[c, 3] - a[j, 3] + 180.0) % 360.0 - 180.0)
                if best is None or d < best[0]:
                    best = (d, c)
            if best is None or best[0] > 0.5:
                missing += 1
                continue
            worst = max(worst, best[0])
            same += int(np.array_equal(a_desc[j], b_desc[best[1]]))
        return missing, same / max(len(a) - missing, 1), worst

    sel_o = inner(okp, 0)
    assert int(sel_o.sum()) > 20_000
    near_crop = (kp[:, 0] >= c0) & (kp[:, 0] < c0 + cs) & (kp[:, 1] >= c0) & (kp[:, 1] < c0 + cs)
    g_kp, g_desc = kp[near_crop], desc[near_crop]
    missing, same, worst = match(okp[sel_o], res.desc[sel_o], g_kp, g_desc)
    assert missing == 0 and same >= 0.995 and worst < 0.5, (missing, same, worst)
    sel_g = inner(g_kp, 1)
    missing, same, worst = match(g_kp[sel_g], g_desc[sel_g], okp, res.desc)
    assert missing == 0 and same >= 0.995, (missing, same, worst)
    assert abs(int(sel_g.sum()) - int(inner(okp, 1).sum())) <= 4    # same region, same count (float rounding at its border)
    f.free()


# ---- configs[3] / configs[4] -----------------------------------------------------------------------------------------------
def _ssd_rows(torch, q, t_rows):
    d = q.to(torch.int32) - t_rows.to(torch.int32)
    return (d * d).sum(dim=1)


def _exact_top2(torch, q, t, block=1 << 20):
    """Exact (ssd, idx) top-2 with the lowest-index tie-break over ALL rows of t for a few queries: f32 GEMM of integer
    rows (exact: every partial sum is an integer below 2^24), first-occurrence minima."""
    qf = q.to(torch.float32)
    qn = (qf * qf).sum(dim=1)
    best = torch.full((q.shape[0], 2), 2 ** 31 - 1, dtype=torch.int64, device=q.device)
    bidx = torch.full((q.shape[0], 2), -1, dtype=torch.int64, device=q.device)
    for s in range(0, t.shape[0], block):
        tf = t[s:s + block].to(torch.float32)
        d = (qn[:, None] + (tf * tf).sum(dim=1)[None, :] - 2.0 * (qf @ tf.T)).to(torch.int64)
        for _ in range(2):
            m, _unused = d.min(dim=1)
            first = (d == m[:, None]).to(torch.int8).argmax(dim=1)       # first occurrence = lowest index
            cand_d = torch.cat([best, m[:, None]], dim=1)
            cand_i = torch.cat([bidx, (first + s)[:, None]], dim=1)
            # order by (ssd, idx): idx of earlier blocks is lower, stable sort keeps them first on ties
            order = torch.sort(cand_d, dim=1, stable=True).indices[:, :2]
            best, bidx = torch.gather(cand_d, 1, order), torch.gather(cand_i, 1, order)
            d[torch.arange(q.shape[0], device=q.device), first] = 2 ** 40
    return best, bidx


def _check_knn_at_scale(ctx, n, nq, n_dup, seed):
    import torch
    dev = torch.device("cuda:0")
    t = synth.synth_descriptors_dev(0, n, dev, seed=7)
    q, src, dup = synth.synth_queries_dev(t, nq, seed=seed, n_dup=n_dup)
    ft = ctx.features_from_device_u8(t.data_ptr(), n)
    fq = ctx.features_from_device_u8(q.data_ptr(), nq)
    out4 = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    ctx.knn2_dev(fq, ft, out4.data_ptr())
    ctx.synchronize()
    o = out4.to(torch.int64)
    # every reported pair is a real pair: ssd recomputed from the rows
    assert (o[:, 1] >= 0).all() and (o[:, 3] >= 0).all() and (o[:, 1] != o[:, 3]).all()
    assert torch.equal(_ssd_rows(torch, q, t[o[:, 1]]).to(torch.int64), o[:, 0])
    assert torch.equal(_ssd_rows(torch, q, t[o[:, 3]]).to(torch.int64), o[:, 2])
    assert (o[:, 0] <= o[:, 2]).all()
    # nobody is worse than his own source row (noise of at most 6 per value: ssd <= 128 * 36)
    own = _ssd_rows(torch, q, t[src]).to(torch.int64)
    assert (o[:, 0] <= own).all() and (own <= 128 * 36).all()
    # planted exact ties: (0, source) then (0, duplicate), lower index first
    has = dup >= 0
    assert int(has.sum()) >= n_dup // 2
    assert (o[has, 0] == 0).all() and (o[has, 2] == 0).all()
    assert torch.equal(o[has, 1], src[has]) and torch.equal(o[has, 3], dup[has])
    # exact full-width answer for a sample of the queries (ties included)
    pick = torch.cat([torch.nonzero(has).flatten()[:16], torch.arange(nq - 48, nq, device=dev)])
    eb, ei = _exact_top2(torch, q[pick], t)
    assert torch.equal(o[pick][:, [0, 2]], eb) and torch.equal(o[pick][:, [1, 3]], ei)
    # the CPU oracle on a window of the map: the queries whose source lies in it, against those rows only
    w0, w1 = n // 2, min(n, n // 2 + 200_000)
    inside = torch.nonzero((src >= w0) & (src < w1) & ((dup < 0) | (dup < w1))).flatten()[:256]
    if len(inside):
        fw = ctx.features_from_device_u8(t[w0:w1].data_ptr(), w1 - w0)
        fqw = ctx.features_from_device_u8(q[inside].contiguous().data_ptr(), len(inside))
        idx, dist = ctx.knn2(fqw, fw)
        ridx, rdist = oracle.knn2(q[inside].cpu().numpy(), t[w0:w1].cpu().numpy(), threads=16)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
        fw.free()
        fqw.free()
    ft.free()
    fq.free()


def test_knn_one_frame_x_300k_rows_short_chunk_form(ctx):
    """A frame's 3000 queries against 300 k rows: 9e8 pairs, the 512-query form with short chunks (merged first exchange)."""
    _check_knn_at_scale(ctx, 300_000, 3000, 48, seed=10)


def test_knn_config3_4096_queries_x_5M_rows(ctx):
    _check_knn_at_scale(ctx, 5_000_000, 4096, 64, seed=8)


def test_knn_config4_shard_131072_queries_x_2_5M_rows(ctx):
    _check_knn_at_scale(ctx, 2_500_000, 131072, 512, seed=9)
