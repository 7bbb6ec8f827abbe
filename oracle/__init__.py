"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (image_matching_amd/) never does: it fails loudly when libimatch.so is missing.

PARITY UNPINNED: the reference (Fricodelco/image_matching) ships no tests or golden vectors and
its arithmetic is OpenCV's, which is absent here; the C files restate the published algorithms
(see each file's header for the reference call site they follow).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so from the C restatement (gcc, a few seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _declare(_LIB)
    return _LIB


_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i64 = ctypes.c_int64


def _declare(L):
    L.oracle_knn2.argtypes = [_f32p, _i64, _f32p, _i64, ctypes.c_int, _i32p, _f32p, ctypes.c_int]
    L.oracle_knn2.restype = None
    L.oracle_knn2_u8.argtypes = [_u8p, _i64, _u8p, _i64, ctypes.c_int, _i32p, _f32p, ctypes.c_int]
    L.oracle_knn2_u8.restype = None
    L.oracle_knn2_fast.argtypes = [_f32p, _i64, _f32p, _i64, ctypes.c_int, _i32p, _f32p, ctypes.c_int]
    L.oracle_knn2_fast.restype = None
    L.oracle_ratio.argtypes = [_i32p, _f32p, _i64, ctypes.c_double, _i32p, _i32p, _f32p]
    L.oracle_ratio.restype = _i64


def _p(a, t):
    return a.ctypes.data_as(t)


def knn2(q, t, threads=0):
    """BFMatcher(NORM_L2).knnMatch(q, t, k=2) -> (idx int32 [Q,2], dist f32 [Q,2])."""
    if q.dtype == np.uint8 and t.dtype == np.uint8:
        q = np.ascontiguousarray(q)
        t = np.ascontiguousarray(t)
        fn, pt = lib().oracle_knn2_u8, _u8p
    else:
        q = np.ascontiguousarray(q, dtype=np.float32)
        t = np.ascontiguousarray(t, dtype=np.float32)
        fn, pt = lib().oracle_knn2, _f32p
    nq, d = q.shape if q.ndim == 2 else (0, 128)
    nt = t.shape[0]
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.float32)
    fn(_p(q, pt), nq, _p(t, pt), nt, d, _p(idx, _i32p), _p(dist, _f32p), threads)
    return idx, dist


def knn2_fast(q, t, threads=0):
    """The CPU BASELINE's matcher (knn_fast.c): f32 rows, SIMD squared distances, cache-tiled, square roots of the winners
    only -- the shape of cv2.BFMatcher(NORM_L2).knnMatch's own computation. Same result as knn2 on integer-valued rows."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    nq, d = q.shape if q.ndim == 2 else (0, 128)
    idx = np.empty((nq, 2), np.int32)
    dist = np.empty((nq, 2), np.float32)
    lib().oracle_knn2_fast(_p(q, _f32p), nq, _p(t, _f32p), t.shape[0], d, _p(idx, _i32p), _p(dist, _f32p), threads)
    return idx, dist


def ratio(idx, dist, r):
    """Lowe ratio test in the reference's float semantics -> (good_q, good_t, good_d)."""
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    nq = idx.shape[0]
    gq = np.empty(nq, np.int32)
    gt = np.empty(nq, np.int32)
    gd = np.empty(nq, np.float32)
    g = lib().oracle_ratio(_p(idx, _i32p), _p(dist, _f32p), nq, float(r), _p(gq, _i32p), _p(gt, _i32p), _p(gd, _f32p))
    return gq[:g].copy(), gt[:g].copy(), gd[:g].copy()


# ---- CLAHE / INTER_AREA / SIFT (sift_oracle.c) ---------------------------------------------
def _declare_sift(L):
    c_int, c_double = ctypes.c_int, ctypes.c_double
    L.oracle_clahe.argtypes = [_u8p, c_int, c_int, c_double, c_int, c_int, _u8p]
    L.oracle_clahe.restype = None
    L.oracle_resize_area.argtypes = [_u8p, c_int, c_int, _u8p, c_int, c_int]
    L.oracle_resize_area.restype = c_int
    L.oracle_gauss_kernel.argtypes = [c_double, _f32p]
    L.oracle_gauss_kernel.restype = c_int
    L.oracle_gauss_blur.argtypes = [_f32p, c_int, c_int, c_double, _f32p]
    L.oracle_gauss_blur.restype = None
    L.oracle_upsample2_linear.argtypes = [_u8p, c_int, c_int, _f32p]
    L.oracle_upsample2_linear.restype = None
    L.oracle_sift_run.argtypes = [_u8p, c_int, c_int, c_int, c_double, c_double, c_double, c_int]
    L.oracle_sift_run.restype = ctypes.c_void_p
    L.oracle_sift_free.argtypes = [ctypes.c_void_p]
    L.oracle_sift_free.restype = None
    L.oracle_sift_count.argtypes = [ctypes.c_void_p]
    L.oracle_sift_count.restype = c_int
    L.oracle_sift_octaves.argtypes = [ctypes.c_void_p]
    L.oracle_sift_octaves.restype = c_int
    L.oracle_sift_octave_size.argtypes = [ctypes.c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]
    L.oracle_sift_octave_size.restype = None
    L.oracle_sift_get.argtypes = [ctypes.c_void_p, _f32p, _f32p]
    L.oracle_sift_get.restype = None
    L.oracle_sift_layer.argtypes = [ctypes.c_void_p, c_int, c_int, c_int, _f32p]
    L.oracle_sift_layer.restype = c_int


_orig_declare = _declare


def _declare(L):  # noqa: F811
    _orig_declare(L)
    _declare_sift(L)


def clahe(img, clip_limit=30.0, tiles=(8, 8)):
    """cv2.createCLAHE(clipLimit, tileGridSize=tiles).apply(img) for uint8 gray."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty_like(img)
    lib().oracle_clahe(_p(img, _u8p), img.shape[0], img.shape[1], float(clip_limit), int(tiles[0]), int(tiles[1]),
                       _p(out, _u8p))
    return out


def resize_area(img, dsize):
    """cv2.resize(img, dsize=(width, height), interpolation=cv2.INTER_AREA)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    dw, dh = int(dsize[0]), int(dsize[1])
    out = np.empty((dh, dw), np.uint8)
    rc = lib().oracle_resize_area(_p(img, _u8p), img.shape[0], img.shape[1], _p(out, _u8p), dh, dw)
    if rc != 0:
        raise ValueError("oracle.resize_area: empty destination")
    return out


def gauss_kernel(sigma):
    k = np.zeros(64, np.float32)
    n = lib().oracle_gauss_kernel(float(sigma), _p(k, _f32p))
    return k[:n].copy()


def gauss_blur(img, sigma):
    img = np.ascontiguousarray(img, dtype=np.float32)
    out = np.empty_like(img)
    lib().oracle_gauss_blur(_p(img, _f32p), img.shape[0], img.shape[1], float(sigma), _p(out, _f32p))
    return out


def upsample2(img):
    """createInitialImage's resize: u8 [h,w] -> f32 [2h,2w], INTER_LINEAR (Spec S1)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((2 * img.shape[0], 2 * img.shape[1]), np.float32)
    lib().oracle_upsample2_linear(_p(img, _u8p), img.shape[0], img.shape[1], _p(out, _f32p))
    return out


class SiftResult:
    """keypoints [K,7] = (x, y, size, angle, response, octave, class_id) in OpenCV's final order,
    descriptors [K,128] float32 (integer valued), and optionally the pyramids."""

    def __init__(self, kp, desc, gauss=None, dog=None):
        self.kp, self.desc, self.gauss, self.dog = kp, desc, gauss, dog


def sift(img, n_octave_layers=3, contrast_threshold=0.04, edge_threshold=10.0, sigma=1.6, keep_pyramid=False):
    """cv2.SIFT_create(0, nOctaveLayers, contrastThreshold, edgeThreshold, sigma).detectAndCompute(img, None)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    L = lib()
    h = L.oracle_sift_run(_p(img, _u8p), img.shape[0], img.shape[1], int(n_octave_layers), float(contrast_threshold),
                          float(edge_threshold), float(sigma), 1 if keep_pyramid else 0)
    try:
        n = L.oracle_sift_count(h)
        kp = np.empty((n, 7), np.float32)
        desc = np.empty((n, 128), np.float32)
        L.oracle_sift_get(h, _p(kp, _f32p), _p(desc, _f32p))
        gauss = dog = None
        if keep_pyramid:
            gauss, dog = [], []
            for o in range(L.oracle_sift_octaves(h)):
                r, c = ctypes.c_int(), ctypes.c_int()
                L.oracle_sift_octave_size(h, o, ctypes.byref(r), ctypes.byref(c))
                g, d = [], []
                for i in range(n_octave_layers + 3):
                    a = np.empty((r.value, c.value), np.float32)
                    L.oracle_sift_layer(h, 0, o, i, _p(a, _f32p))
                    g.append(a)
                for i in range(n_octave_layers + 2):
                    a = np.empty((r.value, c.value), np.float32)
                    L.oracle_sift_layer(h, 1, o, i, _p(a, _f32p))
                    d.append(a)
                gauss.append(g)
                dog.append(d)
        return SiftResult(kp, desc, gauss, dog)
    finally:
        L.oracle_sift_free(h)


# ---- homography (homography_oracle.c) -------------------------------------------------------
def _declare_homography(L):
    L.oracle_homography.argtypes = [_f32p, _f32p, ctypes.c_int, ctypes.c_double, ctypes.c_uint64,
                                    ctypes.POINTER(ctypes.c_double), _u8p]
    L.oracle_homography.restype = ctypes.c_int


_declare_before_homography = _declare


def _declare(L):  # noqa: F811
    _declare_before_homography(L)
    _declare_homography(L)


def homography(src, dst, thr=0.1, seed=0):
    """Spec H' estimator -> (H [3,3] float64, mask uint8 [n,1]) or (None, None)."""
    src = np.ascontiguousarray(np.asarray(src, dtype=np.float32).reshape(-1, 2))
    dst = np.ascontiguousarray(np.asarray(dst, dtype=np.float32).reshape(-1, 2))
    n = src.shape[0]
    H = np.zeros((3, 3), np.float64)
    mask = np.zeros((max(n, 1), 1), np.uint8)
    ok = lib().oracle_homography(_p(src, _f32p), _p(dst, _f32p), n, float(thr), ctypes.c_uint64(int(seed)),
                                 H.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _p(mask, _u8p))
    if not ok:
        return None, None
    return H, mask[:n]


# ---- camera pre-step (camera_oracle.c) ---------------------------------------------------------
_f64p = ctypes.POINTER(ctypes.c_double)
_i16p = ctypes.POINTER(ctypes.c_int16)
_u16p = ctypes.POINTER(ctypes.c_uint16)


def _declare_camera(L):
    c_int = ctypes.c_int
    L.oracle_cvt_gray.argtypes = [_u8p, c_int, c_int, c_int, _u8p]
    L.oracle_cvt_gray.restype = None
    L.oracle_fisheye_new_K.argtypes = [_f64p, _f64p, c_int, c_int, ctypes.c_double, _f64p]
    L.oracle_fisheye_new_K.restype = None
    L.oracle_fisheye_maps.argtypes = [_f64p, _f64p, _f64p, c_int, c_int, _i16p, _u16p]
    L.oracle_fisheye_maps.restype = None
    L.oracle_remap_linear.argtypes = [_u8p, c_int, c_int, c_int, _i16p, _u16p, c_int, c_int, _u8p]
    L.oracle_remap_linear.restype = None


_declare_before_camera = _declare


def _declare(L):  # noqa: F811
    _declare_before_camera(L)
    _declare_camera(L)


def cvt_gray(img, rgb=False):
    """cv2.cvtColor(img, COLOR_BGR2GRAY) (rgb: COLOR_RGB2GRAY)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty(img.shape[:2], np.uint8)
    lib().oracle_cvt_gray(_p(img, _u8p), img.shape[0], img.shape[1], 1 if rgb else 0, _p(out, _u8p))
    return out


def fisheye_new_K(K, D, size, balance=0.0):
    """cv2.fisheye.estimateNewCameraMatrixForUndistortRectify(K, D, size, eye(3), balance=balance)."""
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).reshape(4)
    out = np.empty(9, np.float64)
    lib().oracle_fisheye_new_K(_p(K, _f64p), _p(D, _f64p), int(size[0]), int(size[1]), float(balance), _p(out, _f64p))
    return out.reshape(3, 3)


def fisheye_maps(K, D, new_K, size):
    """cv2.fisheye.initUndistortRectifyMap(K, D, eye(3), new_K, size, CV_16SC2) -> (map1 i16 [h,w,2], map2 u16 [h,w])."""
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).reshape(4)
    new_K = np.ascontiguousarray(new_K, np.float64).reshape(9)
    w, h = int(size[0]), int(size[1])
    m1 = np.empty((h, w, 2), np.int16)
    m2 = np.empty((h, w), np.uint16)
    lib().oracle_fisheye_maps(_p(K, _f64p), _p(D, _f64p), _p(new_K, _f64p), w, h, _p(m1, _i16p), _p(m2, _u16p))
    return m1, m2


def remap_linear(img, map1, map2):
    """cv2.remap(img, map1, map2, INTER_AREA (linear inside remap), BORDER_CONSTANT)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    map1 = np.ascontiguousarray(map1, np.int16)
    map2 = np.ascontiguousarray(map2, np.uint16)
    dh, dw = map2.shape
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, ch), np.uint8)
    lib().oracle_remap_linear(_p(img, _u8p), img.shape[0], img.shape[1], ch, _p(map1, _i16p), _p(map2, _u16p), dh, dw,
                              _p(out, _u8p))
    return out


def undistort(img, K, D, DIM, balance=0.0, scale=0.5):
    """csi_camera.undistort (csi_camera.py:138-167) with dim2 = dim3 = the input size."""
    dim1 = img.shape[:2][::-1]
    scaled_K = np.array(K, np.float64) * dim1[0] / DIM[0]
    scaled_K[2][2] = 1.0
    new_K = fisheye_new_K(scaled_K, D, dim1, balance)
    m1, m2 = fisheye_maps(scaled_K, D, new_K, dim1)
    und = remap_linear(img, m1, m2)
    s0, s1 = int((dim1[0] * (1 - scale)) / 2), int((dim1[1] * (1 - scale)) / 2)
    t0, t1 = int(dim1[0] * scale), int(dim1[1] * scale)
    return und[s1:s1 + t1, s0:s0 + t0]


# ---- match canvas (draw_oracle.c) ---------------------------------------------------------------
def _declare_draw(L):
    c_int = ctypes.c_int
    L.oracle_draw_matches.argtypes = [_u8p, c_int, c_int, _u8p, c_int, c_int, _f32p, _f32p, c_int, _u8p]
    L.oracle_draw_matches.restype = None
    L.oracle_draw_overlay.argtypes = [_u8p, c_int, c_int, c_int, _i32p, c_int, _u8p]
    L.oracle_draw_overlay.restype = None


_declare_before_draw = _declare


def _declare(L):  # noqa: F811
    _declare_before_draw(L)
    _declare_draw(L)


def draw_matches(img1, xy1, img2, xy2):
    """The match canvas of match_finder.py:196 under "Spec D" (draw_oracle.c): xy1[i] <-> xy2[i] are the good matches."""
    img1 = np.ascontiguousarray(img1, np.uint8)
    img2 = np.ascontiguousarray(img2, np.uint8)
    xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
    out = np.empty((max(img1.shape[0], img2.shape[0]), img1.shape[1] + img2.shape[1], 3), np.uint8)
    lib().oracle_draw_matches(_p(img1, _u8p), img1.shape[0], img1.shape[1], _p(img2, _u8p), img2.shape[0], img2.shape[1],
                              _p(xy1, _f32p), _p(xy2, _f32p), len(xy1), _p(out, _u8p))
    return out


def overlay_records(prims):
    """("line", p0, p1, color, thickness) / ("circle", centre, radius, color, thickness) -> int32 [n][8] records."""
    rec = np.zeros((len(prims), 8), np.int32)
    col = rec.view(np.uint8).reshape(len(prims), 32)
    for i, p in enumerate(prims):
        c = p[3]
        c = (list(c) + [0, 0, 0, 0])[:4] if np.ndim(c) else [c, 0, 0, 0]
        rec[i, :6] = (0, p[1][0], p[1][1], p[2][0], p[2][1], p[4]) if p[0] == "line" else (1, p[1][0], p[1][1], p[2], 0, p[4])
        col[i, 24:28] = np.clip(c, 0, 255)
    return rec


def draw_overlay(img, prims):
    """Primitives drawn in order on a copy of img (u8 [h,w] or [h,w,3]) under "Spec D" (draw_oracle.c); follows
    match_finder.py:249-250 and utils.py:75-76."""
    img = np.ascontiguousarray(img, np.uint8)
    ch = 1 if img.ndim == 2 else img.shape[2]
    rec = overlay_records(prims)
    out = np.empty(img.shape, np.uint8)
    lib().oracle_draw_overlay(_p(img, _u8p), img.shape[0], img.shape[1], ch, _p(rec, _i32p), len(rec), _p(out, _u8p))
    return out


# ---- AKAZE (akaze_oracle.c) -------------------------------------------------------------------------
def _declare_akaze(L):
    c_int, c_vp = ctypes.c_int, ctypes.c_void_p
    ip = ctypes.POINTER(c_int)
    L.oracle_akaze_run.argtypes = [_u8p, c_int, c_int, ctypes.c_double, c_int, c_int]
    L.oracle_akaze_run.restype = c_vp
    L.oracle_akaze_free.argtypes = [c_vp]
    L.oracle_akaze_free.restype = None
    L.oracle_akaze_count.argtypes = [c_vp]
    L.oracle_akaze_count.restype = c_int
    L.oracle_akaze_levels.argtypes = [c_vp]
    L.oracle_akaze_levels.restype = c_int
    L.oracle_akaze_kcontrast.argtypes = [c_vp]
    L.oracle_akaze_kcontrast.restype = ctypes.c_float
    L.oracle_akaze_level_info.argtypes = [c_vp, c_int, ip, ip, ip, ip, ip, _f32p]
    L.oracle_akaze_level_info.restype = None
    L.oracle_akaze_get.argtypes = [c_vp, _f32p, _f32p]
    L.oracle_akaze_get.restype = None
    L.oracle_akaze_layer.argtypes = [c_vp, c_int, c_int, _f32p]
    L.oracle_akaze_layer.restype = c_int
    L.oracle_fed_tau.argtypes = [ctypes.c_float, ctypes.c_float, _f32p]
    L.oracle_fed_tau.restype = c_int
    L.oracle_akaze_gauss.argtypes = [_f32p, c_int, c_int, c_int, ctypes.c_double, _f32p]
    L.oracle_akaze_gauss.restype = None


_declare_before_akaze = _declare


def _declare(L):  # noqa: F811
    _declare_before_akaze(L)
    _declare_akaze(L)


class AkazeResult:
    """kp [K,7] = x, y, size, angle, response, octave, class_id; desc [K,64] f32; levels: list of dicts with
    rows/cols/sigma_size/border/nsteps/esigma and, when kept, the Lt/Lsmooth/Lx/Ly/Ldet planes."""
    def __init__(self, kp, desc, kcontrast, levels):
        self.kp, self.desc, self.kcontrast, self.levels = kp, desc, kcontrast, levels


def fed_tau(T, tau_max=0.25):
    out = np.empty(256, np.float32)
    n = lib().oracle_fed_tau(float(T), float(tau_max), _p(out, _f32p))
    return out[:max(n, 0)].copy()


def akaze(img, threshold=0.0005, n_octaves=4, n_octave_layers=5, keep_levels=False):
    """cv2.AKAZE_create(descriptor_type=AKAZE_DESCRIPTOR_KAZE, threshold=threshold, nOctaves=n_octaves,
    nOctaveLayers=n_octave_layers, diffusivity=KAZE_DIFF_PM_G2).detectAndCompute(img, None) (match_finder.py:210-220)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    L = lib()
    h = L.oracle_akaze_run(_p(img, _u8p), img.shape[0], img.shape[1], float(threshold), int(n_octaves), int(n_octave_layers))
    if not h:
        raise ValueError("oracle.akaze: level sizes must halve exactly (see akaze_oracle.c)")
    try:
        n = L.oracle_akaze_count(h)
        kp = np.empty((n, 7), np.float32)
        desc = np.empty((n, 64), np.float32)
        L.oracle_akaze_get(h, _p(kp, _f32p), _p(desc, _f32p))
        levels = []
        ci = ctypes.c_int
        for i in range(L.oracle_akaze_levels(h)):
            r, c, s, b, ns, es = ci(), ci(), ci(), ci(), ci(), ctypes.c_float()
            L.oracle_akaze_level_info(h, i, ctypes.byref(r), ctypes.byref(c), ctypes.byref(s), ctypes.byref(b), ctypes.byref(ns),
                                      ctypes.byref(es))
            lv = dict(rows=r.value, cols=c.value, sigma_size=s.value, border=b.value, nsteps=ns.value, esigma=es.value)
            if keep_levels:
                for kind, name in enumerate(("Lt", "Lsmooth", "Lx", "Ly", "Ldet")):
                    a = np.empty((r.value, c.value), np.float32)
                    L.oracle_akaze_layer(h, kind, i, _p(a, _f32p))
                    lv[name] = a
            levels.append(lv)
        return AkazeResult(kp, desc, float(L.oracle_akaze_kcontrast(h)), levels)
    finally:
        L.oracle_akaze_free(h)
