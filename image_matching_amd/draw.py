"""Minimal numpy rasterisers for the debug canvases the reference builds with cv2 drawing calls
(drawMatchesKnn match_finder.py:196; circle/polylines match_finder.py:249-250). The callers only
publish these images when a publish_* flag is set (position_finder.py:319-320,351-352); pixels are
not part of the parity contract, the array shape and dtype are."""
import numpy as np


def _line_points(x0, y0, x1, y1):
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    return np.linspace(x0, x1, n).round().astype(np.int64), np.linspace(y0, y1, n).round().astype(np.int64)


def draw_line(img, p0, p1, color, thickness=1):
    xs, ys = _line_points(p0[0], p0[1], p1[0], p1[1])
    r = max(int(thickness) // 2, 0)
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            x, y = xs + dx, ys + dy
            ok = (x >= 0) & (x < img.shape[1]) & (y >= 0) & (y < img.shape[0])
            img[y[ok], x[ok]] = color
    return img


def draw_circle(img, center, radius, color, thickness=1):
    t = np.linspace(0, 2 * np.pi, max(16, int(8 * radius)))
    for rr in np.arange(radius - thickness / 2.0, radius + thickness / 2.0 + 0.5, 0.5):
        x = np.round(center[0] + rr * np.cos(t)).astype(np.int64)
        y = np.round(center[1] + rr * np.sin(t)).astype(np.int64)
        ok = (x >= 0) & (x < img.shape[1]) & (y >= 0) & (y < img.shape[0])
        img[y[ok], x[ok]] = color
    return img


def draw_polyline(img, pts, color, thickness=1, closed=True):
    pts = np.asarray(pts).reshape(-1, 2)
    n = len(pts)
    for i in range(n if closed else n - 1):
        draw_line(img, pts[i], pts[(i + 1) % n], color, thickness)
    return img


def matches_canvas(img1, xy1, img2, xy2, pairs, max_lines=400):
    """Side-by-side RGB canvas [max(h1,h2), w1+w2, 3] with one line per good match
    (the layout of cv2.drawMatchesKnn with NOT_DRAW_SINGLE_POINTS)."""
    h1, w1 = img1.shape[:2]
    h2, w2 = img2.shape[:2]
    out = np.zeros((max(h1, h2), w1 + w2, 3), np.uint8)
    out[:h1, :w1] = img1[..., None] if img1.ndim == 2 else img1
    out[:h2, w1:] = img2[..., None] if img2.ndim == 2 else img2
    rng = np.random.default_rng(0)
    for q, t in pairs[:max_lines]:
        color = rng.integers(64, 256, size=3)
        draw_line(out, xy1[q], (xy2[t][0] + w1, xy2[t][1]), color)
    return out
