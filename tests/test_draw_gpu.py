"""GPU parity of the match canvas (csrc/draw.hip) with the CPU oracle under "Spec D", through the C-ABI: integer
rasterisation on both sides, so bit-exact (reference call site: match_finder.py:196)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _case(h1, w1, h2, w2, n, seed, spread=1.0):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h1, w1), dtype=np.uint8)
    b = rng.integers(0, 256, (h2, w2), dtype=np.uint8)
    xy1 = (rng.random((n, 2)) * [w1, h1] * spread).astype(np.float32)
    xy2 = (rng.random((n, 2)) * [w2, h2] * spread).astype(np.float32)
    return a, xy1, b, xy2


@pytest.mark.parametrize("shape", [(480, 640, 2048, 2048, 900), (48, 64, 20, 30, 17), (5, 7, 9, 3, 0), (1, 1, 1, 1, 1),
                                   (300, 301, 200, 513, 5000)])
def test_canvas_bit_exact(ctx, shape):
    a, xy1, b, xy2 = _case(*shape, seed=sum(shape))
    assert np.array_equal(ctx.draw_matches(a, xy1, b, xy2), oracle.draw_matches(a, xy1, b, xy2))


def test_canvas_overlaps_clipping_and_views(ctx):
    a, xy1, b, xy2 = _case(120, 160, 90, 100, 400, 3, spread=1.3)     # points beyond the images: clipped primitives
    xy1[:50] = xy1[0]                                                  # many matches through one pixel: highest index wins
    xy2[10:20] = [np.nan, np.inf]
    assert np.array_equal(ctx.draw_matches(a, xy1, b, xy2), oracle.draw_matches(a, xy1, b, xy2))
    big = np.random.default_rng(4).integers(0, 256, (200, 300), dtype=np.uint8)
    va, vb = big[3:103, 5:155], big[50:190, 100:290]                  # strided views
    assert np.array_equal(ctx.draw_matches(va, xy1 * 0.5, vb, xy2 * 0.5),
                          oracle.draw_matches(np.ascontiguousarray(va), xy1 * 0.5, np.ascontiguousarray(vb), xy2 * 0.5))


def test_find_matches_returns_the_gpu_canvas(ctx):
    """The facade's img3 (match_finder.py:196-197) is this canvas for the good matches, in order."""
    from image_matching_amd import synth
    from image_matching_amd.match_finder import match_finder, roi as roi_t
    from image_matching_amd.image_processing import image_processing
    mf = match_finder(ctx)
    m = synth.synth_map(512, seed=5)
    frame, _ = synth.synth_frame(m, 0, size=(120, 160))
    kp1, dp1, img1 = mf.find_kp_dp(frame)
    kp2, dp2, img2 = mf.find_kp_dp(m)
    cadr = image_processing(img=img1)
    cadr.kp, cadr.dp = kp1, dp1
    good, img3 = mf.find_matches(roi_t(img2, kp2, dp2, 1.0, 0, 0), cadr)
    want = oracle.draw_matches(img1, np.asarray(kp1.xy)[good.query_idx], img2, np.asarray(kp2.xy)[good.train_idx])
    assert img3.shape == (512, 160 + 512, 3) and np.array_equal(img3, want)
