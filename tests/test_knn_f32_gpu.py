"""GPU parity of the float kNN (csrc/knn_f32.hip) with oracle_knn2 through the C-ABI: the same f32 sum in index order on
both sides, so indices, distances and tie-breaks are bit-exact (match_finder.py:182-183 on AKAZE descriptors)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _unit(n, d, seed):
    v = np.random.default_rng(seed).normal(size=(n, d)).astype(np.float32)
    return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12).astype(np.float32)


@pytest.mark.parametrize("nq,nt,d", [(300, 5000, 64), (1, 1, 64), (5, 2, 64), (64, 63, 64), (65, 65, 64), (130, 4097, 64),
                                     (77, 1000, 128), (40, 300, 7), (3000, 50000, 64)])
def test_bit_exact(ctx, nq, nt, d):
    q, t = _unit(nq, d, nq + d), _unit(nt, d, nt + 1)
    gi, gd = ctx.knn2_f32(q, t)
    wi, wd = oracle.knn2(q, t)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)


def test_resident_train_set(ctx):
    """imatch_fset_from_host + imatch_knn2_fset: the train rows travel once; results equal the uploading call's."""
    t = _unit(7000, 64, 3)
    fs = ctx.fset(t)
    assert len(fs) == 7000
    for seed, nq in ((1, 300), (2, 1), (3, 2049)):     # a larger query set after a smaller one: the workspace regrows
        q = _unit(nq, 64, seed)
        gi, gd = ctx.knn2_f32(q, fs)
        wi, wd = oracle.knn2(q, t)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    fs.free()
    empty = ctx.fset(np.zeros((0, 64), np.float32))
    gi, gd = ctx.knn2_f32(_unit(5, 64, 9), empty)
    assert (gi == -1).all() and np.isinf(gd).all()


def test_last_akaze_extraction_matched_where_it_lies(ctx):
    """imatch_knn2_fset_akaze: the queries are the descriptors the last AKAZE extraction left on the device -- the
    result equals the call that uploads the downloaded matrix, bit for bit; a later extraction retires the serial, and
    the facade (match_finder.knn_match, match_finder.py:182-183) then takes the upload path by itself."""
    import image_matching_amd as im
    from image_matching_amd import synth
    from image_matching_amd.match_finder import match_finder
    from image_matching_amd.cvtypes import FloatDescriptors
    img = synth.synth_map(1024, seed=5)
    crop = np.ascontiguousarray(img[:768, :768])
    frame = np.ascontiguousarray(img[100:580, 200:840])
    p = im.AkazeParams()
    ckp, cdesc = ctx.akaze_extract(crop, p)
    fs = ctx.fset(cdesc)
    kp, desc = ctx.akaze_extract(frame, p)
    serial, n = ctx.akaze_last()
    assert serial != 0 and n == len(kp) > 100
    gi, gd = ctx.knn2_fset_akaze(serial, fs)
    wi, wd = ctx.knn2_f32(desc, fs)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    oi, od = oracle.knn2(desc, cdesc)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    ctx.akaze_extract(crop, p)                       # another extraction: the frame's descriptors are gone from the device
    assert ctx.akaze_last()[0] != serial
    with pytest.raises(im.ImatchError):
        ctx.knn2_fset_akaze(serial, fs)
    # the facade: device path while the frame is the last extraction, upload path afterwards -- same matches either way
    mf = match_finder(ctx=ctx, detector="akaze")
    rkp, rdp, _ = mf.find_kp_dp(crop)
    fkp, fdp, _ = mf.find_kp_dp(frame)
    assert isinstance(fdp, FloatDescriptors) and fdp.source is not None and fdp.source[1] == ctx.akaze_last()[0]
    a = mf.knn_match(fdp, rdp)                        # device path
    mf.find_kp_dp(crop)                               # retires the frame's serial
    b = mf.knn_match(fdp, rdp)                        # upload path
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], oi) and np.array_equal(a[1], od)
    fs.free()


@pytest.mark.parametrize("nq,nt,d,kind", [(3000, 30000, 64, "unit"), (1, 4096, 64, "unit"), (777, 50001, 64, "unit"),
                                          (500, 8192, 128, "unit"), (300, 6000, 7, "unit"), (257, 5000, 33, "scaled"),
                                          (400, 9000, 64, "tiny"), (400, 9000, 64, "sift"), (16384, 4100, 64, "unit")])
def test_resident_set_filter_path_is_bit_exact(ctx, nq, nt, d, kind):
    """Resident sets of 4096 rows and more go through the f16 matrix-core filter + exact refinement (kf_filter_kernel /
    kf_refine_kernel): the answer must be the exhaustive kernel's and the oracle's bit for bit -- unit rows, rows of
    other scales (norms 1e-3 .. 30), rows of tiny values (f16 subnormals), integer-valued rows up to 255, planted exact
    duplicates (index order on ties) and clusters of near-duplicates around the second-best distance."""
    rng = np.random.default_rng(nq + nt + d)
    q, t = _unit(nq, d, nq + d), _unit(nt, d, nt + 1)
    if kind == "scaled":
        t *= rng.uniform(1e-3, 30.0, size=(nt, 1)).astype(np.float32)
        q *= rng.uniform(1e-3, 30.0, size=(nq, 1)).astype(np.float32)
    elif kind == "tiny":
        t *= np.float32(3e-5); q *= np.float32(3e-5)
    elif kind == "sift":
        t = np.rint(np.abs(t) * 400).clip(0, 255).astype(np.float32); q = np.rint(np.abs(q) * 400).clip(0, 255).astype(np.float32)
    k = min(nq, 64)
    src = rng.choice(nt, k, replace=False)
    q[:k] = t[src]                                        # exact hits
    dup = rng.choice(nt, k, replace=False)
    t[dup] = t[src]                                       # ... that exist twice: ties at distance 0, lower index first
    q[:k] = t[src]
    if nq > 200:                                           # clusters of near-duplicates of a query's neighbours
        for i in range(100, 110):
            near = rng.choice(nt, 40, replace=False)
            t[near] = q[i] + (rng.normal(size=(40, d)) * 1e-3 * max(float(np.linalg.norm(q[i])), 1e-6)).astype(np.float32)
    fs = ctx.fset(t)
    gi, gd = ctx.knn2_f32(q, fs)
    assert ctx.knn2_f32_last_path() == 1                  # the filter answered, nobody fell back
    n_chk = nq if nq * nt <= 1.6e8 else 400
    pick = np.r_[0:min(n_chk, nq)] if n_chk == nq else np.unique(np.r_[0:128, rng.choice(nq, n_chk, replace=False)])
    wi, wd = oracle.knn2(q[pick], t)
    assert np.array_equal(gi[pick], wi) and np.array_equal(gd[pick], wd)
    if n_chk != nq:                                        # the rest against the exhaustive kernel on the GPU
        ei, ed = ctx.knn2_f32(q, t)
        assert ctx.knn2_f32_last_path() == 0
        assert np.array_equal(gi, ei) and np.array_equal(gd, ed)
    fs.free()


def test_filter_path_hands_over_when_it_cannot_answer(ctx):
    """Values beyond the f16 path's range (either side), non-finite values and a candidate list that overflows (a set of
    identical rows: every row is a candidate) all end in the exhaustive kernel's answer."""
    t = _unit(5000, 64, 1)
    q = _unit(50, 64, 2)
    t_big = t.copy(); t_big[17, 3] = 5000.0
    for tt, qq in ((t_big, q), (t, q * np.float32(3000.0))):
        fs = ctx.fset(tt)
        gi, gd = ctx.knn2_f32(qq, fs)
        assert ctx.knn2_f32_last_path() in (0, 2)         # the set did not qualify (0) or the query flagged itself (2)
        wi, wd = oracle.knn2(qq, tt)
        assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
        fs.free()
    same = np.repeat(_unit(1, 64, 3), 6000, axis=0)           # 6000 identical rows: more candidates than a list holds
    fs = ctx.fset(same)
    gi, gd = ctx.knn2_f32(q, fs)
    assert ctx.knn2_f32_last_path() == 2                  # lists overflowed: the exhaustive kernel answered
    assert (gi[:, 0] == 0).all() and (gi[:, 1] == 1).all()
    wi, wd = oracle.knn2(q, same)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    fs.free()


def test_empty_and_ties(ctx):
    q = _unit(10, 64, 1)
    gi, gd = ctx.knn2_f32(q, np.zeros((0, 64), np.float32))
    assert (gi == -1).all() and np.isinf(gd).all()
    gi, gd = ctx.knn2_f32(np.zeros((0, 64), np.float32), q)
    assert gi.shape == (0, 2)
    t = _unit(500, 64, 2)
    t[400] = t[7]; t[123] = t[7]; t[300] = q[3]; t[20] = q[3]      # exact duplicates: the lower train index ranks first
    q2 = np.vstack([q, t[7:8]])
    gi, gd = ctx.knn2_f32(q2, t)
    wi, wd = oracle.knn2(q2, t)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    assert tuple(gi[10]) == (7, 123) and tuple(gi[3]) == (20, 300) and gd[3, 0] == 0


def test_large_train_set_properties(ctx):
    """A million train rows (six whole-map AKAZE extractions): planted exact duplicates come back as (lower index, higher
    index) at distance 0, and a sample of the queries is compared with the oracle in full."""
    nt, nq = 1 << 20, 2048
    t = _unit(nt, 64, 11)
    rows = np.random.default_rng(12).choice(nt - 60000, nq // 2, replace=False)
    t[rows + 54321] = t[rows]
    q = np.vstack([t[rows], _unit(nq // 2, 64, 13)])
    gi, gd = ctx.knn2_f32(q, t)
    first = np.minimum(rows, rows + 54321)
    # a planted row may itself have been overwritten by another plant: judge by content, not by index
    assert (gd[:nq // 2] == 0).all()
    assert np.array_equal(t[gi[:nq // 2, 0]], q[:nq // 2]) and np.array_equal(t[gi[:nq // 2, 1]], q[:nq // 2])
    assert (gi[:nq // 2, 0] < gi[:nq // 2, 1]).all() and (gi[:nq // 2, 0] <= first).all()
    assert (gd[:, 0] <= gd[:, 1]).all()
    pick = np.r_[0:8, nq - 8:nq]
    wi, wd = oracle.knn2(q[pick], t)
    assert np.array_equal(gi[pick], wi) and np.array_equal(gd[pick], wd)


def test_akaze_chain_through_the_facade(ctx):
    """find_kp_dp -> find_matches -> find_keypoints_transform with detector='akaze' recovers a planted homography."""
    from image_matching_amd import synth
    from image_matching_amd.match_finder import match_finder, roi as roi_t
    from image_matching_amd.image_processing import image_processing
    mf = match_finder(ctx, detector="akaze")
    m = synth.synth_map(2048, seed=5)
    frame, H = synth.synth_frame(m, 3, size=(480, 640), scale_range=(0.9, 1.1))
    kp1, dp1, img1 = mf.find_kp_dp(frame)
    kp2, dp2, img2 = mf.find_kp_dp(m)
    assert dp1.shape[1] == 64 and len(kp1) == len(dp1) > 50
    cadr = image_processing(img=img1)
    cadr.kp, cadr.dp = kp1, dp1
    r = roi_t(img2, kp2, dp2, 1.0, 0, 0)
    good, img3 = mf.find_matches(r, cadr)
    assert len(good) > 10 and img3.shape == (2048, 640 + 2048, 3)
    # the same good list from the oracle's matcher on the same descriptors
    wi, wd = oracle.knn2(np.asarray(dp1), np.asarray(dp2))
    keep = wd[:, 0].astype(np.float64) < mf.points_quality * wd[:, 1].astype(np.float64)
    assert np.array_equal(good.query_idx, np.nonzero(keep)[0]) and np.array_equal(good.train_idx, wi[keep, 0])
    out = mf.find_keypoints_transform(good, r, cadr, None)
    M = out[5]
    assert M is not None
    corners = np.array([[0, 0, 1], [639, 0, 1], [639, 479, 1], [0, 479, 1]], np.float64).T
    a = M @ corners; b = H @ corners
    err = np.hypot(a[0] / a[2] - b[0] / b[2], a[1] / a[2] - b[1] / b[2])
    assert err.max() < 2.0, err
