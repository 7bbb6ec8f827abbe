"""CPU checks of the match-canvas oracle (oracle/draw_oracle.c, "Spec D"): layout and dtype of cv2.drawMatchesKnn's
canvas (match_finder.py:196), and the geometry of the primitives. Pixel parity with OpenCV's anti-aliased drawing is not
a goal (see the file header)."""
import numpy as np

import oracle


def test_canvas_layout_without_matches():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    b = rng.integers(0, 256, (100, 90), dtype=np.uint8)
    c = oracle.draw_matches(a, np.zeros((0, 2)), b, np.zeros((0, 2)))
    assert c.shape == (100, 154, 3) and c.dtype == np.uint8
    assert np.array_equal(c[:48, :64], np.repeat(a[..., None], 3, 2))
    assert np.array_equal(c[:, 64:], np.repeat(b[..., None], 3, 2))
    assert not c[48:, :64].any()


def test_line_and_circles():
    a = np.zeros((40, 50), np.uint8)
    b = np.zeros((40, 60), np.uint8)
    c = oracle.draw_matches(a, [[10.4, 20.5]], b, [[30.6, 9.5]])       # -> (10, 20) and (31 + 50, 10): half to even
    drawn = c.any(axis=2)
    ys, xs = np.nonzero(drawn)
    colour = c[20, 10]
    assert (c[drawn] == colour).all() and colour.min() >= 64
    # line: one pixel per column between the circles, end points included
    assert drawn[20, 10] and drawn[10, 81]
    for x in range(15, 77):
        col = np.nonzero(drawn[:, x])[0]
        assert len(col) == 1 and abs(col[0] - (20 + (x - 10) * (10 - 20) / (81 - 10))) <= 0.5 + 1e-9
    # circles of radius 3 around both end points
    for cx, cy in ((10, 20), (81, 10)):
        assert drawn[cy, cx + 3] and drawn[cy, cx - 3] and drawn[cy + 3, cx] and drawn[cy - 3, cx]
        assert not drawn[cy + 1, cx + 1]
    assert xs.min() == 7 and xs.max() == 84


def test_later_match_wins_and_clipping():
    a = np.full((30, 30), 9, np.uint8)
    c = oracle.draw_matches(a, [[5, 5], [5, 5]], a, [[5, 5], [5, 5]])
    first = oracle.draw_matches(a, [[5, 5]], a, [[5, 5]])
    assert (c[5, 5] != first[5, 5]).any()              # the second match's colour covers the first
    far = oracle.draw_matches(a, [[-1e9, np.nan]], a, [[1e9, 50]])     # clipped, no crash
    assert far.shape == (30, 60, 3)


# ---- overlay primitives (match_finder.py:249-250, utils.py:75-76) against an independent numpy statement ----------
def _np_overlay(img, prims):
    """Spec D restated with numpy distance fields: ring / disc by squared distance, thick line = union of discs on the
    rounded-half-up integer DDA."""
    out = img.copy()
    h, w = img.shape[:2]
    yy, xx = np.mgrid[0:h, 0:w]
    for p in prims:
        col = np.zeros(4, np.int64)
        c = p[3]
        col[:len(np.atleast_1d(c))] = np.atleast_1d(c)[:4]
        t = p[4]
        if p[0] == "circle":
            d4 = 4 * ((xx - p[1][0]) ** 2 + (yy - p[1][1]) ** 2)
            r = p[2]
            m = (d4 <= 4 * r * r) if t < 0 else ((d4 < (2 * r + t) ** 2) & ((d4 >= (2 * r - t) ** 2) | (2 * r <= t)))
        else:
            (x0, y0), (x1, y1) = p[1], p[2]
            n = max(abs(x1 - x0), abs(y1 - y0))
            m = np.zeros((h, w), bool)
            for k in range(n + 1):
                cx = x0 + (int(np.floor(k * (x1 - x0) / n + 0.5)) if n else 0)
                cy = y0 + (int(np.floor(k * (y1 - y0) / n + 0.5)) if n else 0)
                m |= (4 * ((xx - cx) ** 2 + (yy - cy) ** 2) <= t * t) & (abs(xx - cx) <= t // 2) & (abs(yy - cy) <= t // 2)
        out[m] = col[:1] if img.ndim == 2 else col[:3]
    return out


def test_overlay_against_numpy_statement():
    rng = np.random.default_rng(3)
    for shape in ((60, 80), (60, 80, 3), (7, 5, 3)):
        img = rng.integers(0, 200, shape, dtype=np.uint8)
        prims = [("circle", (30, 20), 10, 255, 5), ("line", (2, 3), (70, 50), (0, 255, 9), 3),
                 ("line", (70, 10), (-20, 58), (7, 7, 7), 2), ("circle", (40, 40), 4, (1, 2, 3), 2),
                 ("circle", (10, 50), 6, (90, 80, 70), -1), ("line", (5, 5), (5, 5), 200, 1),
                 ("circle", (0, 0), 1, 250, 3), ("line", (60, 59), (79, 0), (255, 0, 0), 5)]
        assert np.array_equal(oracle.draw_overlay(img, prims), _np_overlay(img, prims))


def test_overlay_order_and_pose_marker_shape():
    img = np.zeros((50, 50, 3), np.uint8)
    a = oracle.draw_overlay(img, [("circle", (25, 25), 10, 255, 5), ("line", (0, 25), (49, 25), (0, 9, 0), 3)])
    assert tuple(a[25, 15]) == (0, 9, 0)                  # the later primitive covers the ring
    assert tuple(a[15, 25]) == (255, 0, 0)                # cv2's scalar colour 255 = Scalar(255, 0, 0)
    assert not a[25 - 5, 25].any() and not a[0, 0].any()  # inside the ring's hole / outside: untouched
