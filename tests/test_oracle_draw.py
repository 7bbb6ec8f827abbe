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
