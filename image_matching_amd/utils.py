"""Host helpers of the hot path with the reference's behaviour (reference utils.py): resize_img
(utils.py:17-22, the INTER_AREA resize now runs on the GPU), line_intersection (utils.py:46-60),
CrossProduct (utils.py:79-84), isConvex (utils.py:86-143, messages included: the callers log them,
position_finder.py:396-397,639), angle (utils.py:145-152), calculate_euqlidian_dist (utils.py:154-156), and the three
names the reference's nodes import besides them: check_exist (utils.py:8-14), rotate_image (utils.py:25-44),
draw_circle_on_map_by_coord_and_angles (utils.py:62-77, drawn on the GPU)."""
from decimal import Decimal
from math import atan2, pi
from os.path import exists

import numpy as np

from .geodetic_conv import GeodeticConvert

_ctx = None


def default_context():
    """The process-wide imatch context (device 0), created on first use."""
    global _ctx
    if _ctx is None:
        from ._lib import Context
        _ctx = Context(0)
    return _ctx


def set_default_context(ctx):
    global _ctx
    _ctx = ctx


def resize_img(img, scale):
    # dsize exactly as the reference computes it: int(dim * Decimal(scale))
    width = int(img.shape[1] * Decimal(scale))
    height = int(img.shape[0] * Decimal(scale))
    return default_context().resize_area(img, (width, height))


def check_exist(path, i=1):
    """The loggers' unused-file-name probe (utils.py:8-14; image_logger.py:41,50,189, command_logger.py:21). As in the
    reference the recursive call's result is dropped: a path that exists gives None, a free path comes back unchanged."""
    print(path)
    if exists(path) is True:
        path = str(i) + '_' + path
        check_exist(path, i=i + 1)
    else:
        return path


def rotate_image(mat, angle):
    """utils.py:25-44: rotate about the centre by `angle` degrees (counter-clockwise) into the bounding box of the
    rotated image -- cv2.getRotationMatrix2D + cv2.warpAffine (bilinear, constant zero border). Imported by the
    reference's match_finder.py:13 and never called; host numpy (float bilinear weights, not OpenCV's 5-bit fixed-point
    ones: parity unpinned, as for everything OpenCV)."""
    mat = np.asarray(mat)
    height, width = mat.shape[:2]
    cx, cy = width / 2, height / 2
    a = np.deg2rad(angle)
    alpha, beta = np.cos(a), np.sin(a)
    M = np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]], np.float64)
    abs_cos, abs_sin = abs(M[0, 0]), abs(M[0, 1])
    bound_w = int(height * abs_sin + width * abs_cos)
    bound_h = int(height * abs_cos + width * abs_sin)
    M[0, 2] += bound_w / 2 - cx
    M[1, 2] += bound_h / 2 - cy
    # warpAffine maps destination pixels through the inverse transform
    A = np.linalg.inv(np.vstack([M, [0, 0, 1]]))
    xs, ys = np.meshgrid(np.arange(bound_w, dtype=np.float64), np.arange(bound_h, dtype=np.float64))
    sx = A[0, 0] * xs + A[0, 1] * ys + A[0, 2]
    sy = A[1, 0] * xs + A[1, 1] * ys + A[1, 2]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = sx - x0, sy - y0
    src = mat.astype(np.float64).reshape(height, width, -1)
    out = np.zeros((bound_h, bound_w, src.shape[2]), np.float64)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx < width) & (yy >= 0) & (yy < height)
            v = src[np.clip(yy, 0, height - 1), np.clip(xx, 0, width - 1)]
            out += np.where(ok, wx * wy, 0.0)[..., None] * v
    out = np.clip(np.rint(out), 0, 255).astype(mat.dtype) if mat.dtype == np.uint8 else out.astype(mat.dtype)
    return out.reshape((bound_h, bound_w) + mat.shape[2:])


def draw_circle_on_map_by_coord_and_angles(img, coord, circle_coord, pixel_size, yaw, color):
    """utils.py:62-77 (callers position_finder.py:333-346,382-388): mark the geodetic point circle_coord on the map
    image whose top-left corner is at coord -- a circle of radius 4 and a 20-pixel heading line in `color`, both of
    thickness 2 -- on a 3-channel copy of img. Drawn by imatch_draw_overlay on the GPU ("Spec D" pixels)."""
    img = np.asarray(img)
    if img.ndim == 2:                       # cv2.cvtColor(img, COLOR_GRAY2BGR); a colour image is kept as it is
        img = np.repeat(img[:, :, None], 3, axis=2)
    g_c = GeodeticConvert()
    g_c.initialiseReference(coord[0], coord[1], 0)
    y, x, z = g_c.geodetic2Ned(circle_coord[0], circle_coord[1], 0)
    pixel_x = int(x / float(pixel_size))
    pixel_y = int(-y / float(pixel_size))
    x2 = int(pixel_x + 20 * np.sin(yaw))
    y2 = int(pixel_y - 20 * np.cos(yaw))
    return default_context().draw_overlay(img, [("line", (pixel_x, pixel_y), (x2, y2), color, 2),
                                                ("circle", (pixel_x, pixel_y), 4, color, 2)])


def line_intersection(line1, line2):
    (ax, ay), (bx, by) = line1
    (cx, cy), (dx, dy) = line2
    xdiff = (ax - bx, cx - dx)
    ydiff = (ay - by, cy - dy)
    div = xdiff[0] * ydiff[1] - xdiff[1] * ydiff[0]
    if div == 0:
        raise Exception('lines do not intersect')
    d = (ax * by - ay * bx, cx * dy - cy * dx)
    x = (d[0] * xdiff[1] - d[1] * xdiff[0]) / div
    y = (d[0] * ydiff[1] - d[1] * ydiff[0]) / div
    return x, y


def CrossProduct(A):
    x1, y1 = A[1][0] - A[0][0], A[1][1] - A[0][1]
    x2, y2 = A[2][0] - A[0][0], A[2][1] - A[0][1]
    return x1 * y2 - y1 * x2


def calculate_euqlidian_dist(point1, point2):
    return np.sqrt((point1[0] - point2[0]) ** 2 + (point1[1] - point2[1]) ** 2)


def angle(A, B, C, /):
    a = atan2(A[1] - B[1], A[0] - B[0])
    c = atan2(C[1] - B[1], C[0] - B[0])
    if a < 0:
        a += pi * 2
    if c < 0:
        c += pi * 2
    return (pi * 2 + c - a) if a > c else (c - a)


def isConvex(points, shape, angle_restriction_homography, low_scale_restriction_homography,
             high_scale_restriction_homography):
    """Plausibility gate on the projected frame corners: convex, side ratios within [ls, hs], every
    interior angle within `angle_restriction_homography` of a right angle. Returns (bool, message)."""
    p = [[pt[0][0], pt[0][1]] for pt in points]
    n = len(points)
    prev = 0
    for i in range(n):
        curr = CrossProduct([p[i], p[(i + 1) % n], p[(i + 2) % n]])
        if curr != 0:
            if curr * prev < 0:
                return False, "ANSWER IS NOT CONVEX"
            prev = curr
    dist1 = calculate_euqlidian_dist(p[0], p[1])
    dist2 = calculate_euqlidian_dist(p[1], p[2])
    ls, hs = low_scale_restriction_homography, high_scale_restriction_homography
    if shape[0] / dist1 < ls:
        return False, "SHAPE NOT FIT, shape[0]/dist1: " + str(shape[0] / dist1) + " ls: " + str(ls)
    if shape[1] / dist2 < ls:
        return False, "SHAPE NOT FIT, shape[1]/dist2: " + str(shape[1] / dist2) + " ls: " + str(ls)
    if shape[0] / dist1 > hs:
        return False, "SHAPE NOT FIT, shape[0]/dist1: " + str(shape[0] / dist1) + " hs: " + str(hs)
    if shape[1] / dist2 > hs:
        # the reference prints ls after "hs:" here (utils.py:117); the text is kept as logged
        return False, "SHAPE NOT FIT, shape[1]/dist2: " + str(shape[1] / dist2) + " hs: " + str(ls)
    corners = ((p[0], p[1], p[2]), (p[1], p[2], p[3]), (p[2], p[3], p[0]), (p[3], p[0], p[1]))
    for k, (a, b, c) in enumerate(corners, start=1):
        ang = angle(a, b, c)
        if abs(ang - np.pi / 2) > angle_restriction_homography:
            return False, "ANGLE NOT FIT, angle_%d delta: " % k + str(ang - np.pi / 2)
    return True, "TRANSFROM IS GOOOOOOOD!"
