"""GPU parity: CLAHE, INTER_AREA and SIFT through the C-ABI vs the CPU oracle.

Bars (BASELINE.json north_star): byte stages (CLAHE, resize) bit-exact; the f32 pyramid is bit-exact
because both sides use the same fma order; keypoints within 0.5 px for >= 99 % of the oracle's
keypoints; descriptors within L2 <= 16 (of norm 512; a few +-1 LSB flips from exp/atan2 and parallel
summation order) for >= 99 % of matched keypoints."""
import numpy as np
import pytest

import oracle
import image_matching_amd as im
from image_matching_amd import synth

pytestmark = pytest.mark.gpu

DESC_L2_TOL = 16.0
KP_TOL_PX = 0.5


@pytest.fixture(scope="module")
def small_map():
    return synth.synth_map(768, seed=99)


@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (243, 325), (97, 131), (64, 64), (1000, 700)])
def test_clahe_bit_exact(ctx, small_map, shape):
    img = small_map[:shape[0], :shape[1]] if shape[0] <= 768 and shape[1] <= 768 else np.tile(small_map, (2, 1))[:shape[0], :shape[1]]
    got = ctx.clahe(img)
    want = oracle.clahe(img)
    assert np.array_equal(got, want), int((got != want).sum())


def test_clahe_strided_view_and_flat_image(ctx, small_map):
    view = small_map[10:310, 20:420]  # non-contiguous rows
    assert np.array_equal(ctx.clahe(view), oracle.clahe(view))
    flat = np.full((120, 160), 77, np.uint8)
    assert np.array_equal(ctx.clahe(flat), oracle.clahe(flat))


@pytest.mark.parametrize("shape,dsize", [((720, 1280), (640, 360)), ((480, 640), (427, 320)), ((768, 768), (500, 333)),
                                         ((300, 500), (499, 299)), ((600, 600), (200, 200)), ((97, 131), (17, 13)),
                                         ((480, 640), (640, 480))])
def test_resize_area_bit_exact(ctx, small_map, shape, dsize):
    img = np.tile(small_map, (1, 2))[:shape[0], :shape[1]]
    got = ctx.resize_area(img, dsize)
    want = oracle.resize_area(img, dsize)
    assert got.shape == want.shape == (dsize[1], dsize[0])
    assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("shape,dsize", [((100, 100), (200, 200)), ((97, 131), (140, 103)), ((240, 320), (640, 480)),
                                         ((50, 70), (233, 51)), ((60, 80), (79, 121)), ((1, 1), (5, 3))])
def test_resize_enlarge_bit_exact(ctx, small_map, shape, dsize):
    """An enlarged axis: cv2.resize(INTER_AREA) runs its bilinear kernel with the area coefficients (a frame narrower
    than image_size_sift makes rescale_cadr enlarge, match_finder.py:171-175)."""
    img = np.ascontiguousarray(small_map[:shape[0], :shape[1]])
    got = ctx.resize_area(img, dsize)
    want = oracle.resize_area(img, dsize)
    assert got.shape == want.shape == (dsize[1], dsize[0])
    assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("shape", [(240, 320), (201, 333), (64, 96)])
def test_pyramid_bit_exact(ctx, small_map, shape):
    img = small_map[:shape[0], :shape[1]]
    want = oracle.sift(img, keep_pyramid=True)
    f = ctx.sift_extract(img, im.SiftParams(clahe=False))
    assert ctx.sift_debug_octaves() == len(want.gauss)
    for o in range(len(want.gauss)):
        for i in range(6):
            got = ctx.sift_debug_layer(0, o, i)
            assert got.shape == want.gauss[o][i].shape
            assert np.array_equal(got, want.gauss[o][i]), ("gauss", o, i, float(np.abs(got - want.gauss[o][i]).max()))
        for i in range(5):
            got = ctx.sift_debug_layer(1, o, i)
            assert np.array_equal(got, want.dog[o][i]), ("dog", o, i)
    f.free()


def _match_keypoints(kp_g, kp_o):
    """For every oracle keypoint the nearest GPU keypoint with the same octave/layer word and an angle
    within 1 degree; returns (index into GPU or -1, distance)."""
    from scipy.spatial import cKDTree
    tree = cKDTree(kp_g[:, :2])
    idx = np.full(len(kp_o), -1)
    dist = np.full(len(kp_o), np.inf)
    for i, k in enumerate(kp_o):
        for j in tree.query_ball_point(k[:2], KP_TOL_PX):
            g = kp_g[j]
            da = abs(g[3] - k[3])
            da = min(da, 360 - da)
            if (int(g[5]) & 0xffff) == (int(k[5]) & 0xffff) and da < 1.0:
                d = float(np.hypot(*(g[:2] - k[:2])))
                if d < dist[i]:
                    idx[i], dist[i] = j, d
    return idx, dist


@pytest.mark.parametrize("shape,clahe", [((480, 640), True), ((240, 320), False), ((333, 517), True)])
def test_keypoints_and_descriptors(ctx, small_map, shape, clahe):
    img = small_map[:shape[0], :shape[1]]
    pre = oracle.clahe(img) if clahe else img
    want = oracle.sift(pre)
    f = ctx.sift_extract(img, im.SiftParams(clahe=clahe))
    kp, desc = f.download()
    if clahe:
        assert np.array_equal(f.image(*shape), pre)
    assert len(want.kp) > 100
    # integer-valued descriptors, SIFT-normalised
    assert np.array_equal(desc, np.rint(desc)) and desc.min() >= 0 and desc.max() <= 255
    # OpenCV's final order: sorted by (x, y, -size, angle)
    key = np.lexsort((kp[:, 3], -kp[:, 2], kp[:, 1], kp[:, 0]))
    assert np.array_equal(key, np.arange(len(kp)))
    idx, dist = _match_keypoints(kp, want.kp)
    found = idx >= 0
    assert found.mean() >= 0.99, (found.mean(), len(kp), len(want.kp))
    assert abs(len(kp) - len(want.kp)) <= 0.01 * len(want.kp) + 2
    l2 = np.linalg.norm(desc[idx[found]] - want.desc[found], axis=1)
    assert (l2 <= DESC_L2_TOL).mean() >= 0.99, (float(np.percentile(l2, 99)), float(l2.max()))
    print("kp %d/%d matched %.4f  exact-pos %.4f  desc L2 p50 %.2f p99 %.2f max %.2f  identical %.3f" % (
        len(kp), len(want.kp), found.mean(), (dist[found] == 0).mean(), np.median(l2), np.percentile(l2, 99), l2.max(),
        (l2 == 0).mean()))
    f.free()


def test_degenerate_images(ctx):
    flat = np.full((100, 120), 128, np.uint8)
    f = ctx.sift_extract(flat)
    assert len(f) == 0
    kp, desc = f.download()
    assert kp.shape == (0, 7) and desc.shape == (0, 128)
    tiny = (np.arange(12 * 9) % 251).astype(np.uint8).reshape(9, 12)
    f2 = ctx.sift_extract(tiny, im.SiftParams(clahe=False))
    assert len(f2) == len(oracle.sift(tiny).kp)


def test_extract_is_deterministic(ctx, small_map):
    img = small_map[:300, :400]
    a = ctx.sift_extract(img).download()
    b = ctx.sift_extract(img).download()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_non_default_params(ctx, small_map):
    img = small_map[:200, :260]
    want = oracle.sift(img, n_octave_layers=4, contrast_threshold=0.03, edge_threshold=12, sigma=1.4)
    f = ctx.sift_extract(img, im.SiftParams(n_octave_layers=4, contrast_threshold=0.03, edge_threshold=12, sigma=1.4,
                                            clahe=False))
    kp, desc = f.download()
    idx, dist = _match_keypoints(kp, want.kp)
    assert (idx >= 0).mean() >= 0.99


def test_calls_right_after_an_extraction_do_not_disturb_its_descriptors(ctx):
    """imatch_sift_extract returns with the upload of its sorted keypoint records still queued (they sit in the context's
    pinned block); a call that fills the same block next -- homography, the float kNN, the canvas -- must wait for it."""
    from image_matching_amd import synth
    m = synth.synth_map(1024, seed=9)
    ref_kp, ref_desc = ctx.sift_extract(m).download()
    rng = np.random.default_rng(0)
    src = (rng.random((3000, 2)) * 500).astype(np.float32)
    for follow in ("homography", "knn_f32", "canvas"):
        f = ctx.sift_extract(m)
        if follow == "homography":
            ctx.homography(src, src * 1.5 + 3, 0.1, seed=1)
        elif follow == "knn_f32":
            ctx.knn2_f32(rng.random((4000, 64), dtype=np.float32), rng.random((500, 64), dtype=np.float32))
        else:
            ctx.draw_matches(m[:300, :400], src[:500], m[:200, :300], src[:500])
        kp, desc = f.download()
        assert np.array_equal(kp, ref_kp) and np.array_equal(desc, ref_desc), follow
        f.free()


def test_device_sort_and_its_host_fallback_agree_with_the_oracle():
    """Frames are sorted, deduplicated and rescaled on the device (kp_sort_kernel), launched for the
    previous extraction's keypoint count plus a margin; an image with more records than that goes through the host's
    radix sort in the same call. Both must give the oracle's list, record for record: a fresh context sees a small image
    first (few keypoints: a small launch), then a 1024^2 image (several thousand: host fallback), then the same image
    again (the launch is large enough now: device sort)."""
    c = im.Context(0)
    m = synth.synth_map(1024, seed=21)
    small = np.ascontiguousarray(m[:96, :128])
    p = im.SiftParams(clahe=True)
    f0 = c.sift_extract(small, p)
    want0 = oracle.sift(oracle.clahe(small))
    kp0 = f0.download(descriptors=False)[0]
    assert len(kp0) == len(want0.kp) < 400 and np.array_equal(kp0[:, :2], want0.kp[:, :2])
    want = oracle.sift(oracle.clahe(m))
    assert 2000 < len(want.kp) <= 8192, len(want.kp)
    runs = []
    for _ in range(2):
        f = c.sift_extract(m, p)
        kp, desc = f.download()
        runs.append((kp, desc))
        f.free()
    for kp, desc in runs:
        assert len(kp) == len(want.kp)
        assert np.array_equal(kp[:, :2], want.kp[:, :2]) and np.array_equal(kp[:, 4:], want.kp[:, 4:])
        assert np.abs(kp[:, 2] - want.kp[:, 2]).max() <= 1e-4
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    f0.free()
    c.close() if hasattr(c, "close") else None


def test_octave_hand_off_time_out_falls_back_to_the_one_workgroup_walk(small_map, capfd):
    """A single frame's small octaves run as one workgroup per octave, each waiting for a flag of the octave before it.
    IMATCH_DEBUG_TAIL_WITHHOLD=0 makes the first of them never publish its flag: the next workgroup gives up after about
    a second, the library says so on stderr, runs the extraction again with the one-workgroup walk (which waits for
    nobody) and keeps to that form on the context. The result is the normal one, record for record."""
    import os
    img = np.ascontiguousarray(small_map[:480, :640])
    c = im.Context(0)
    try:
        want_kp, want_d = c.sift_extract(img).download()
        os.environ["IMATCH_DEBUG_TAIL_WITHHOLD"] = "0"
        try:
            got_kp, got_d = c.sift_extract(img).download()
        finally:
            del os.environ["IMATCH_DEBUG_TAIL_WITHHOLD"]
        err = capfd.readouterr().err
        assert "waited in vain" in err and "one-workgroup walk" in err, err
        assert len(want_kp) > 500
        assert np.array_equal(got_kp, want_kp) and np.array_equal(got_d, want_d)
        # the context keeps to the one-workgroup walk: the knob no longer has anything to withhold
        os.environ["IMATCH_DEBUG_TAIL_WITHHOLD"] = "0"
        try:
            again_kp, again_d = c.sift_extract(img).download()
        finally:
            del os.environ["IMATCH_DEBUG_TAIL_WITHHOLD"]
        assert "waited in vain" not in capfd.readouterr().err
        assert np.array_equal(again_kp, want_kp) and np.array_equal(again_d, want_d)
    finally:
        c.close()


def test_stage_timers_cover_the_chain(small_map):
    """SURVEY section 5, tracing: with profiling on every stage of the path is bracketed by events (and a roctx range);
    imatch_stage_times gives the last bracket of each. One frame through extract -> kNN -> ratio -> homography."""
    import torch
    c = im.Context(0)
    try:
        c.set_profiling(True)
        assert c.stage_times() == {}
        fm = c.sift_extract(small_map)
        ff = c.sift_extract(np.ascontiguousarray(small_map[100:580, 60:700]))
        out4 = torch.empty((len(ff), 4), dtype=torch.int32, device="cuda:0")
        c.knn2_dev(ff, fm, out4.data_ptr())
        gq, gt, gd = c.ratio_dev(out4.data_ptr(), len(ff), 0.7)
        kf, km = ff.download(descriptors=False)[0], fm.download(descriptors=False)[0]
        assert len(gq) > 10
        H, _ = c.homography(kf[gq, :2], km[gt, :2], 0.1, seed=1)
        assert H is not None
        st = c.stage_times()
        assert set(st) == {"clahe", "scale_space", "keypoints", "describe", "knn", "knn_merge", "ratio", "homography_score"}
        assert all(0 < v < 50 for v in st.values()), st
        assert abs(st["scale_space"] - c.sift_pyramid_ms()) < 1e-6 and abs(st["knn"] - c.knn2_kernel_ms()) < 1e-6
        c.set_profiling(False)
        c.sift_extract(small_map).free()     # no brackets are recorded with profiling off: the old values stay
        assert c.stage_times()["scale_space"] == st["scale_space"]
    finally:
        c.close()
