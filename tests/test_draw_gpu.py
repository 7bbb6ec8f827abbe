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


# ---- overlay primitives: pose overlay (match_finder.py:249-250) and map marker (utils.py:62-77) -----------------
def _prims(rng, n, w, h):
    out = []
    for i in range(n):
        col = tuple(int(v) for v in rng.integers(0, 256, 3)) if i % 3 else int(rng.integers(0, 256))
        if i % 2:
            out.append(("circle", (int(rng.integers(-10, w + 10)), int(rng.integers(-10, h + 10))), int(rng.integers(0, 40)), col,
                        int(rng.choice([1, 2, 3, 5, 9, -1]))))
        else:
            out.append(("line", (int(rng.integers(-30, w + 30)), int(rng.integers(-30, h + 30))),
                        (int(rng.integers(-30, w + 30)), int(rng.integers(-30, h + 30))), col, int(rng.choice([1, 2, 3, 5, 8]))))
    return out


@pytest.mark.parametrize("shape", [(480, 1120, 3), (300, 301), (64, 64, 3), (1, 1), (9, 200, 3)])
def test_overlay_bit_exact(ctx, shape):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    prims = _prims(rng, 40, shape[1], shape[0])
    assert np.array_equal(ctx.draw_overlay(img, prims), oracle.draw_overlay(img, prims))
    assert np.array_equal(ctx.draw_overlay(img, []), img)


def test_overlay_far_end_points_same_pixels_bounded_work(ctx):
    """A wild homography puts the quadrilateral's corners millions of pixels away. The kernel walks only the DDA steps
    whose brush can touch the image: the pixels are the oracle's (which walks every step), and the time no longer grows
    with the line's length or explodes with its thickness."""
    import time
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (240, 320, 3), dtype=np.uint8)
    far = 1 << 20
    prims = [("line", (-far, -far + 37), (far, far), (255, 0, 0), 5),            # through the image, both ends far away
             ("line", (far, 100), (-far + 11, 131), (0, 255, 0), 3),              # nearly horizontal, right to left
             ("line", (160, -far), (171, far - 5), (0, 0, 255), 9),               # nearly vertical
             ("line", (-far, -far), (-far + 50, -far + 3), 77, 2),                # nowhere near the image
             ("line", (-40, 300), (far, -far), (9, 9, 9), 1),                     # one end near, outside
             ("line", (10, 10), (300, 200), (200, 200, 0), 255)]                  # the thickest brush
    got = ctx.draw_overlay(img, prims)
    assert np.array_equal(got, oracle.draw_overlay(img, prims))
    # beyond the library's coordinate range the wrapper clips the SEGMENT (its direction stays): a steep diagonal through
    # the image centre still passes through it
    wild = [("line", (160 - (1 << 33), 120 - (1 << 34)), (160 + (1 << 33), 120 + (1 << 34)), (255, 255, 255), 3)]
    t0 = time.perf_counter()
    out = ctx.draw_overlay(img, wild + [("line", (-(1 << 27), -(1 << 27) + 5), (1 << 27, 1 << 27), (1, 2, 3), 255)])
    assert time.perf_counter() - t0 < 2.0
    drawn = (ctx.draw_overlay(img, wild) != img).any(axis=2)
    ys, xs = np.nonzero(drawn)
    assert len(ys) > 300 and abs(np.polyfit(xs - 160.0, ys - 120.0, 1)[0] - 2.0) < 0.05 and drawn[120, 160]
    assert (out != img).any()


def test_overlay_rejects_what_it_cannot_draw(ctx):
    import image_matching_amd as im
    img = np.zeros((8, 8), np.uint8)
    with pytest.raises(im.ImatchError):
        ctx.draw_overlay(img, [("circle", (1, 1), 5000, 1, 1)])
    with pytest.raises(im.ImatchError):
        ctx.draw_overlay(img, [("line", (0, 0), (1, 1), 1, 0)])


def test_pose_overlay_of_find_keypoints_transform(ctx):
    """img_for_pub comes back with cv2.circle(.., 10, 255, 5) + cv2.polylines(.., 255, 3) drawn (match_finder.py:243-250)."""
    from image_matching_amd import synth
    from image_matching_amd.match_finder import line_intersection, match_finder, perspective_transform, roi as roi_t
    from image_matching_amd.image_processing import image_processing
    mf = match_finder(ctx)
    m = synth.synth_map(768, seed=5)
    frame, _ = synth.synth_frame(m, 1, size=(240, 320))
    kp1, dp1, img1 = mf.find_kp_dp(frame)
    kp2, dp2, img2 = mf.find_kp_dp(m)
    cadr = image_processing(img=img1)
    cadr.kp, cadr.dp = kp1, dp1
    r = roi_t(img2, kp2, dp2, 1.0, 0, 0)
    good, img3 = mf.find_matches(r, cadr)
    assert len(good) > 10
    res = mf.find_keypoints_transform(good, r, cadr, img3.copy())
    M, pub = res[5], res[6]
    assert M is not None and pub.shape == img3.shape
    pts = np.float32([[0, 0], [0, 239], [319, 239], [319, 0]]).reshape(-1, 1, 2)
    dst = perspective_transform(pts, M)
    dst[:, 0, 0] += 320
    xc, yc = line_intersection((dst[0][0], dst[2][0]), (dst[1][0], dst[3][0]))
    q = np.int32(dst).reshape(4, 2)
    prims = [("circle", (int(xc), int(yc)), 10, 255, 5)]
    prims += [("line", tuple(map(int, q[k])), tuple(map(int, q[(k + 1) % 4])), 255, 3) for k in range(4)]
    assert np.array_equal(pub, oracle.draw_overlay(img3, prims))
    assert (pub != img3).any()


def test_map_marker(ctx):
    """utils.draw_circle_on_map_by_coord_and_angles (utils.py:62-77): gray input becomes 3-channel, marker at the geodetic
    point's pixel, heading line of 20 pixels."""
    from image_matching_amd import utils
    from image_matching_amd.geodetic_conv import GeodeticConvert
    utils.set_default_context(ctx)
    m = np.random.default_rng(1).integers(0, 255, (300, 400), dtype=np.uint8)
    origin = (55.75, 37.61)
    g = GeodeticConvert()
    g.initialiseReference(origin[0], origin[1], 0)
    lat, lon, _ = g.ned2Geodetic(north=-100.0, east=150.0, down=0)
    out = utils.draw_circle_on_map_by_coord_and_angles(m, origin, (lat, lon), 0.5 * 2, 0.5, (0, 0, 255))
    y, x, _ = g.geodetic2Ned(lat, lon, 0)
    px_, py_ = int(x / 1.0), int(-y / 1.0)
    want = oracle.draw_overlay(np.repeat(m[:, :, None], 3, 2),
                               [("line", (px_, py_), (int(px_ + 20 * np.sin(0.5)), int(py_ - 20 * np.cos(0.5))), (0, 0, 255), 2),
                                ("circle", (px_, py_), 4, (0, 0, 255), 2)])
    assert out.shape == (300, 400, 3) and np.array_equal(out, want)
    assert tuple(out[py_ - 4, px_]) == (0, 0, 255) and abs(px_ - 150) <= 1 and abs(py_ - 100) <= 1
    col = np.zeros((50, 60, 3), np.uint8)
    assert utils.draw_circle_on_map_by_coord_and_angles(col, origin, origin, 1.0, 0.0, (1, 2, 3)).shape == (50, 60, 3)
