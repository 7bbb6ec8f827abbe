"""Host helpers of the hot path with the reference's behaviour (reference utils.py): resize_img
(utils.py:17-22, the INTER_AREA resize now runs on the GPU), line_intersection (utils.py:46-60),
CrossProduct (utils.py:79-84), isConvex (utils.py:86-143, messages included: the callers log them,
position_finder.py:396-397,639), angle (utils.py:145-152), calculate_euqlidian_dist (utils.py:154-156)."""
from decimal import Decimal
from math import atan2, pi

import numpy as np

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
