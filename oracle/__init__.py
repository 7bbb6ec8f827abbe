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
