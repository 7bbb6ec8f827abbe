"""ctypes binding of include/imatch.h. There is no CPU fallback: a missing libimatch.so or a
missing GPU raises."""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMATCH_LIB: another build of the library (development builds made with IMATCH_LIB_OUT / IMATCH_BUILD_DEFINES, see
# build.py); the product is the in-tree libimatch.so
_LIB_PATH = os.environ.get("IMATCH_LIB") or os.path.join(_HERE, "libimatch.so")
_lib = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p


class AkazeParams(ctypes.Structure):
    """imatch_akaze_params (include/imatch.h); defaults are the reference's call (match_finder.py:200-211)."""
    _fields_ = [("threshold", ctypes.c_double), ("n_octaves", ctypes.c_int), ("n_octave_layers", ctypes.c_int),
                ("clahe", ctypes.c_int), ("clahe_clip", ctypes.c_double), ("clahe_tiles_x", ctypes.c_int),
                ("clahe_tiles_y", ctypes.c_int)]

    def __init__(self, threshold=0.0005, n_octaves=4, n_octave_layers=5, clahe=True, clahe_clip=30.0, clahe_tiles=(8, 8)):
        super().__init__(float(threshold), int(n_octaves), int(n_octave_layers), 1 if clahe else 0, float(clahe_clip),
                         int(clahe_tiles[0]), int(clahe_tiles[1]))


class ImatchError(RuntimeError):
    pass


class SiftParams(ctypes.Structure):
    """imatch_sift_params (include/imatch.h); defaults = OpenCV's SIFT_create defaults + the
    reference's CLAHE(30, (8,8)) (match_finder.py:200)."""
    _fields_ = [("n_octave_layers", ctypes.c_int), ("contrast_threshold", ctypes.c_double),
                ("edge_threshold", ctypes.c_double), ("sigma", ctypes.c_double), ("clahe", ctypes.c_int),
                ("clahe_clip", ctypes.c_double), ("clahe_tiles_x", ctypes.c_int), ("clahe_tiles_y", ctypes.c_int)]

    def __init__(self, n_octave_layers=3, contrast_threshold=0.04, edge_threshold=10.0, sigma=1.6, clahe=True,
                 clahe_clip=30.0, clahe_tiles=(8, 8)):
        super().__init__(int(n_octave_layers), float(contrast_threshold), float(edge_threshold), float(sigma),
                         1 if clahe else 0, float(clahe_clip), int(clahe_tiles[0]), int(clahe_tiles[1]))


# name -> (restype, argtypes); every symbol include/imatch.h declares
SIGNATURES = {
    "imatch_device_count": (ctypes.c_int, []),
    "imatch_init": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_destroy": (None, [c_vp]),
    "imatch_last_error": (ctypes.c_char_p, [c_vp]),
    "imatch_stream": (c_vp, [c_vp]),
    "imatch_synchronize": (ctypes.c_int, [c_vp]),
    "imatch_set_profiling": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "imatch_knn2_kernel_ms": (ctypes.c_int, [c_vp, c_f32p]),
    "imatch_sift_pyramid_ms": (ctypes.c_int, [c_vp, c_f32p]),
    "imatch_stage_times": (ctypes.c_int, [c_vp, c_f32p, ctypes.c_int]),
    "imatch_stage_name": (ctypes.c_char_p, [ctypes.c_int]),
    "imatch_features_from_host_f32": (ctypes.c_int, [c_vp, c_f32p, c_i64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_features_from_host_u8": (ctypes.c_int, [c_vp, c_u8p, c_i64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_features_from_device_u8": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_features_from_device_u8_seg": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.POINTER(c_i64), ctypes.c_size_t,
                                                          ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_features_count": (c_i64, [c_vp]),
    "imatch_features_is_integral": (ctypes.c_int, [c_vp]),
    "imatch_features_free": (None, [c_vp]),
    "imatch_clahe": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                    ctypes.c_int, ctypes.c_int, c_u8p]),
    "imatch_resize_area": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u8p,
                                          ctypes.c_int, ctypes.c_int]),
    "imatch_cvt_gray": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u8p]),
    "imatch_fisheye_new_camera_matrix": (ctypes.c_int, [c_f64p, c_f64p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                        c_f64p]),
    "imatch_fisheye_undistort": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                c_f64p, c_f64p, c_f64p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, c_u8p]),
    "imatch_draw_matches": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_u8p, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, c_f32p, c_f32p, ctypes.c_int, c_u8p]),
    "imatch_draw_overlay": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                           ctypes.c_int, c_u8p]),
    "imatch_knn2_f32": (ctypes.c_int, [c_vp, c_f32p, c_i64, c_f32p, c_i64, ctypes.c_int, c_i32p, c_f32p]),
    "imatch_fset_from_host": (ctypes.c_int, [c_vp, c_f32p, c_i64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_fset_count": (c_i64, [c_vp]),
    "imatch_fset_free": (None, [c_vp]),
    "imatch_knn2_fset": (ctypes.c_int, [c_vp, c_f32p, c_i64, c_vp, c_i32p, c_f32p]),
    "imatch_knn2_f32_debug_last_path": (ctypes.c_int, [c_vp]),
    "imatch_akaze_last": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_i64)]),
    "imatch_knn2_fset_akaze": (ctypes.c_int, [c_vp, ctypes.c_uint64, c_vp, c_i32p, c_f32p]),
    "imatch_akaze_extract": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(AkazeParams), ctypes.POINTER(c_i64)]),
    "imatch_akaze_extract_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                ctypes.POINTER(AkazeParams), ctypes.POINTER(c_i64)]),
    "imatch_akaze_download": (ctypes.c_int, [c_vp, c_f32p, c_f32p]),
    "imatch_akaze_image": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int]),
    "imatch_akaze_debug_levels": (ctypes.c_int, [c_vp]),
    "imatch_akaze_debug_layer": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, c_f32p, ctypes.POINTER(ctypes.c_int),
                                                ctypes.POINTER(ctypes.c_int)]),
    "imatch_sift_extract": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(SiftParams), ctypes.POINTER(c_vp)]),
    "imatch_sift_extract_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(SiftParams), ctypes.POINTER(c_vp)]),
    "imatch_sift_extract_batch_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.POINTER(SiftParams), ctypes.POINTER(c_vp),
                                                     ctypes.POINTER(c_i64)]),
    "imatch_ratio_batch_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_i64), ctypes.c_int, ctypes.c_double, c_i32p,
                                              c_i32p, c_f32p, ctypes.POINTER(c_i64)]),
    "imatch_homography_batch": (ctypes.c_int, [c_vp, c_f32p, c_f32p, ctypes.POINTER(c_i64), ctypes.c_int, ctypes.c_double,
                                               ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), c_u8p,
                                               ctypes.POINTER(ctypes.c_int)]),
    "imatch_features_desc_u8_dev": (ctypes.c_int, [c_vp, c_vp]),
    "imatch_features_download": (ctypes.c_int, [c_vp, c_f32p, c_f32p]),
    "imatch_features_keypoints_xy": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_int64), ctypes.c_int64, c_f32p]),
    "imatch_features_image": (ctypes.c_int, [c_vp, c_u8p, ctypes.c_int, ctypes.c_int]),
    "imatch_features_set_keypoints": (ctypes.c_int, [c_vp, c_f32p]),
    "imatch_sift_debug_layer": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p,
                                               ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "imatch_sift_debug_octaves": (ctypes.c_int, [c_vp]),
    "imatch_sift_debug_layer_rows": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f32p]),
    "imatch_homography": (ctypes.c_int, [c_vp, c_f32p, c_f32p, ctypes.c_int, ctypes.c_double, ctypes.c_uint64,
                                         ctypes.POINTER(ctypes.c_double), c_u8p, ctypes.POINTER(ctypes.c_int)]),
    "imatch_match_localize": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_uint64,
                                             c_i32p, c_i32p, c_f32p, ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double),
                                             c_u8p, ctypes.POINTER(ctypes.c_int)]),
    "imatch_match_localize_debug_mismatches": (ctypes.c_int, [c_vp]),
    "imatch_knn2": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32p, c_f32p]),
    "imatch_knn2_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, c_vp]),
    "imatch_knn2_noalloc_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, c_vp]),
    "imatch_knn2_debug_last_chunks": (ctypes.c_int, [c_vp]),
    "imatch_merge_top2_dev": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_i64, c_vp]),
    "imatch_ratio_dev": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_double, c_i32p, c_i32p, c_f32p,
                                        ctypes.POINTER(c_i64)]),
}


def load_library():
    """Load libimatch.so (built in-tree by image_matching_amd/build.py) and declare its symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImatchError(
            "libimatch.so is not built (%s). Run `python -m image_matching_amd.build`; "
            "there is no CPU fallback." % _LIB_PATH)
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64.so.7; importing torch
    # first makes libimatch.so bind to that copy instead of loading /opt/rocm's beside it (two
    # runtimes in one process cannot both own the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a, t):
    return a.ctypes.data_as(t)


def _clip_segment(x0, y0, x1, y1, box):
    """Liang-Barsky on Python integers (exact): the part of the segment inside [-box, box]^2 with end points rounded to
    the nearest integer, or None if it misses the box."""
    from fractions import Fraction
    t0, t1 = Fraction(0), Fraction(1)
    dx, dy = x1 - x0, y1 - y0
    for p, q in ((-dx, x0 + box), (dx, box - x0), (-dy, y0 + box), (dy, box - y0)):
        if p == 0:
            if q < 0:
                return None
            continue
        r = Fraction(q, p)
        if p < 0:
            t0 = max(t0, r)
        else:
            t1 = min(t1, r)
    if t0 > t1:
        return None
    rnd = lambda v: int(round(v))
    return rnd(x0 + t0 * dx), rnd(y0 + t0 * dy), rnd(x0 + t1 * dx), rnd(y0 + t1 * dy)


class Features:
    """A descriptor set resident in HBM (int8 rows = value - 128, + squared norms). Outlives the call that made
    it, so `find_matches` need not upload again (the reference re-uploads on every call,
    match_finder.py:187-188)."""

    def __init__(self, ctx, handle, n):
        self.ctx = ctx
        self._h = handle
        self.n = int(n)

    def __len__(self):
        return self.n

    @property
    def integral(self):
        return bool(self.ctx.lib.imatch_features_is_integral(self._h))

    def download(self, keypoints=True, descriptors=True):
        """-> (kp float32 [K,7] = x, y, size, angle, response, octave, class_id; desc float32 [K,128])."""
        kp = np.empty((self.n, 7), np.float32) if keypoints else None
        desc = np.empty((self.n, 128), np.float32) if descriptors else None
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.imatch_features_download(
                self._h, _ptr(kp, c_f32p) if keypoints else None, _ptr(desc, c_f32p) if descriptors else None))
        return kp, desc

    def keypoints_xy(self, rows):
        """-> float32 [n,2]: (x, y) of the given keypoint rows (kp[m.queryIdx].pt of match_finder.py:231-232)."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        out = np.empty((len(rows), 2), np.float32)
        # reads the set's own host-side keypoint table only: no device work, no context state, so no context lock (another
        # thread may sit in a long call on the context meanwhile); the caller keeps the set alive
        rc = self.ctx.lib.imatch_features_keypoints_xy(
            self._h, rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(rows), _ptr(out, c_f32p))
        if rc != 0:
            raise ImatchError("imatch error %d: keypoint rows out of range or a set without keypoints" % rc)
        return out

    def set_keypoints(self, kp):
        """Keypoint records float32 [n,7] (as `download` returns them) for a set made from descriptor rows alone, e.g. a
        map database read from disk: afterwards Context.match_localize takes the set."""
        kp = np.ascontiguousarray(kp, dtype=np.float32)
        if kp.shape != (self.n, 7):
            raise ValueError("keypoint table must be [%d, 7]" % self.n)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.imatch_features_set_keypoints(self._h, _ptr(kp, c_f32p)))

    def desc_u8_dev(self, out_ptr):
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.imatch_features_desc_u8_dev(self._h, c_vp(out_ptr)))

    def image(self, h, w):
        out = np.empty((h, w), np.uint8)
        with self.ctx.lock:
            self.ctx._check(self.ctx.lib.imatch_features_image(self._h, _ptr(out, c_u8p), h, w))
        return out

    def free(self):
        # the pool the blocks go back to is the context's: same lock as every other call (re-entrant, so a finaliser
        # running on the thread that holds it does not deadlock)
        with self.ctx.lock:
            if self._h is not None and self.ctx._h is not None:
                self.ctx.lib.imatch_features_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FSet:
    """A float descriptor set resident in HBM (the train side of knn2_f32): made once, matched against many times."""

    def __init__(self, ctx, handle, n, d):
        self.ctx, self._h, self.n, self.d = ctx, handle, int(n), int(d)

    def __len__(self):
        return self.n

    def free(self):
        with self.ctx.lock:
            if self._h is not None and self.ctx._h is not None:
                self.ctx.lib.imatch_fset_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One HIP device + one stream. Calls are serialised with a lock (rospy runs callbacks on
    several threads, position_finder.py:127,146)."""

    def __init__(self, device=0):
        self.lib = load_library()
        n = self.lib.imatch_device_count()
        if n <= 0:
            raise ImatchError("no HIP device visible (imatch_device_count() = %d); there is no CPU fallback" % n)
        h = c_vp()
        rc = self.lib.imatch_init(int(device), ctypes.byref(h))
        if rc != 0:
            raise ImatchError("imatch_init(device=%d) failed with code %d" % (device, rc))
        self._h = h
        self.device = int(device)
        self.lock = threading.RLock()

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.imatch_last_error(self._h)
            raise ImatchError("imatch error %d: %s" % (rc, msg.decode() if msg else "?"))

    def close(self):
        if self._h is not None:
            self.lib.imatch_destroy(self._h)
            self._h = None

    def __del__(self):
        # features may still reference the context; leave the device objects to process exit
        pass

    @property
    def stream(self):
        return self.lib.imatch_stream(self._h)

    def synchronize(self):
        with self.lock:
            self._check(self.lib.imatch_synchronize(self._h))

    def set_profiling(self, on=True):
        with self.lock:
            self._check(self.lib.imatch_set_profiling(self._h, 1 if on else 0))

    def knn2_kernel_ms(self):
        ms = ctypes.c_float(0)
        with self.lock:
            self._check(self.lib.imatch_knn2_kernel_ms(self._h, ctypes.byref(ms)))
        return ms.value

    def stage_times(self):
        """{stage name: ms of its last bracket} for every stage that has run since set_profiling(True) (imatch_stage_times):
        clahe, scale_space, keypoints, describe, knn, knn_merge, ratio, homography_score."""
        n = 8
        ms = np.zeros(n, np.float32)
        with self.lock:
            self._check(self.lib.imatch_stage_times(self._h, _ptr(ms, c_f32p), n))
        return {self.lib.imatch_stage_name(i).decode(): float(ms[i]) for i in range(n) if ms[i] >= 0}

    def sift_pyramid_ms(self):
        ms = ctypes.c_float(0)
        with self.lock:
            self._check(self.lib.imatch_sift_pyramid_ms(self._h, ctypes.byref(ms)))
        return ms.value

    # -- descriptor sets ----------------------------------------------------------------
    def features(self, desc):
        """Upload a [n,128] float32 or uint8 descriptor matrix."""
        desc = np.asarray(desc)
        if desc.ndim != 2:
            desc = desc.reshape(-1, 128) if desc.size else np.zeros((0, 128), np.float32)
        h = c_vp()
        with self.lock:
            if desc.dtype == np.uint8:
                a = np.ascontiguousarray(desc)
                rc = self.lib.imatch_features_from_host_u8(self._h, _ptr(a, c_u8p), a.shape[0], a.shape[1],
                                                           ctypes.byref(h))
            else:
                a = np.ascontiguousarray(desc, dtype=np.float32)
                rc = self.lib.imatch_features_from_host_f32(self._h, _ptr(a, c_f32p), a.shape[0], a.shape[1],
                                                            ctypes.byref(h))
            self._check(rc)
        return Features(self, h, desc.shape[0])

    def features_from_device_u8(self, dev_ptr, n, d=128):
        """Wrap a u8 [n,128] matrix already in device memory (e.g. torch tensor.data_ptr())."""
        h = c_vp()
        with self.lock:
            self._check(self.lib.imatch_features_from_device_u8(self._h, c_vp(dev_ptr), int(n), int(d),
                                                                ctypes.byref(h)))
        return Features(self, h, n)

    def features_from_device_u8_seg(self, dev_ptr, counts, seg_pitch_rows):
        """One set from len(counts) row blocks in device memory: block b = counts[b] rows at dev_ptr + b * seg_pitch_rows
        * 128 bytes (an all-gathered array of per-rank slots, each filled from its start). Queues work only."""
        cnt = (c_i64 * len(counts))(*[int(c) for c in counts])
        h = c_vp()
        with self.lock:
            self._check(self.lib.imatch_features_from_device_u8_seg(self._h, c_vp(dev_ptr), len(counts), cnt,
                                                                    int(seg_pitch_rows) * 128, 128, ctypes.byref(h)))
        return Features(self, h, sum(int(c) for c in counts))

    # -- pre-processing and SIFT ----------------------------------------------------------
    @staticmethod
    def _gray(img):
        img = np.asarray(img)
        if img.ndim != 2 or img.dtype != np.uint8:
            raise ValueError("expected a 2-D uint8 image, got %s %s" % (img.dtype, img.shape))
        if img.strides[1] != 1 or img.strides[0] < img.shape[1]:
            img = np.ascontiguousarray(img)
        return img

    def clahe(self, img, clip_limit=30.0, tiles=(8, 8)):
        img = self._gray(img)
        out = np.empty(img.shape, np.uint8)
        with self.lock:
            self._check(self.lib.imatch_clahe(self._h, _ptr(img, c_u8p), img.shape[0], img.shape[1], img.strides[0],
                                              float(clip_limit), int(tiles[0]), int(tiles[1]), _ptr(out, c_u8p)))
        return out

    def resize_area(self, img, dsize):
        """cv2.resize(img, dsize=(width, height), interpolation=INTER_AREA)."""
        img = self._gray(img)
        dw, dh = int(dsize[0]), int(dsize[1])
        out = np.empty((dh, dw), np.uint8)
        with self.lock:
            self._check(self.lib.imatch_resize_area(self._h, _ptr(img, c_u8p), img.shape[0], img.shape[1],
                                                    img.strides[0], _ptr(out, c_u8p), dh, dw))
        return out

    # -- camera pre-step --------------------------------------------------------------------
    @staticmethod
    def _color(img):
        img = np.asarray(img)
        if img.ndim != 3 or img.shape[2] != 3 or img.dtype != np.uint8:
            raise ValueError("expected a [h, w, 3] uint8 image, got %s %s" % (img.dtype, img.shape))
        if img.strides[2] != 1 or img.strides[1] != 3 or img.strides[0] < 3 * img.shape[1]:
            img = np.ascontiguousarray(img)
        return img

    def cvt_gray(self, img, rgb=False):
        """cv2.cvtColor(img, COLOR_BGR2GRAY) (rgb=True: COLOR_RGB2GRAY)."""
        img = self._color(img)
        out = np.empty(img.shape[:2], np.uint8)
        with self.lock:
            self._check(self.lib.imatch_cvt_gray(self._h, _ptr(img, c_u8p), img.shape[0], img.shape[1], img.strides[0],
                                                 1 if rgb else 0, _ptr(out, c_u8p)))
        return out

    def fisheye_undistort(self, img, K, D, new_K, window=None, to_gray=False):
        """initUndistortRectifyMap(K, D, I, new_K, size, CV_16SC2) + remap(INTER_AREA, BORDER_CONSTANT), restricted to
        window = (x0, y0, width, height) of the rectified image (default: all of it)."""
        img = np.asarray(img)
        if img.ndim == 2:
            img = self._gray(img)
            ch = 1
        else:
            img = self._color(img)
            ch = 3
        h, w = img.shape[:2]
        x0, y0, cw, chh = window if window is not None else (0, 0, w, h)
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        D = np.ascontiguousarray(D, np.float64).reshape(4)
        new_K = np.ascontiguousarray(new_K, np.float64).reshape(9)
        out = np.empty((chh, cw) if (to_gray or ch == 1) else (chh, cw, 3), np.uint8)
        with self.lock:
            self._check(self.lib.imatch_fisheye_undistort(self._h, _ptr(img, c_u8p), h, w, ch, img.strides[0],
                                                          _ptr(K, c_f64p), _ptr(D, c_f64p), _ptr(new_K, c_f64p), int(x0),
                                                          int(y0), int(cw), int(chh), 1 if to_gray else 0,
                                                          _ptr(out, c_u8p)))
        return out

    def draw_matches(self, img1, xy1, img2, xy2):
        """The canvas of cv2.drawMatchesKnn(img1, kp1, img2, kp2, good, None, NOT_DRAW_SINGLE_POINTS): xy1[i] <-> xy2[i]
        are the good matches' keypoint positions."""
        img1, img2 = self._gray(img1), self._gray(img2)
        xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
        xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
        if len(xy1) != len(xy2):
            raise ValueError("one point of each image per match")
        out = np.empty((max(img1.shape[0], img2.shape[0]), img1.shape[1] + img2.shape[1], 3), np.uint8)
        with self.lock:
            self._check(self.lib.imatch_draw_matches(self._h, _ptr(img1, c_u8p), img1.shape[0], img1.shape[1],
                                                     img1.strides[0], _ptr(img2, c_u8p), img2.shape[0], img2.shape[1],
                                                     img2.strides[0], _ptr(xy1, c_f32p), _ptr(xy2, c_f32p), len(xy1),
                                                     _ptr(out, c_u8p)))
        return out

    def draw_overlay(self, img, prims):
        """Draw primitives on a copy of img (uint8 [h,w] or [h,w,3]) in order, on the GPU (imatch_draw_overlay).
        prims: iterable of ("line", (x0, y0), (x1, y1), color, thickness) / ("circle", (x, y), radius, color, thickness)
        with integer coordinates; color is a scalar or a per-channel tuple as cv2 takes it."""
        img = np.asarray(img)
        if img.ndim == 2:
            img = self._gray(img)
            ch = 1
        else:
            img = self._color(img)
            ch = 3
        rec = np.zeros((len(prims), 8), np.int32)
        col = rec.view(np.uint8).reshape(len(prims), 32)
        for i, p in enumerate(prims):
            c = p[3]
            c = (list(c) + [0, 0, 0, 0])[:4] if np.ndim(c) else [c, 0, 0, 0]     # cv2: a scalar colour is Scalar(v, 0, 0, 0)
            lim = 1 << 28     # the library's coordinate limit
            x0, y0 = (int(v) for v in p[1])
            if p[0] == "line":
                x1, y1 = (int(v) for v in p[2])
                if max(abs(x0), abs(y0), abs(x1), abs(y1)) > lim:
                    # a wild homography: the SEGMENT is clipped (exact integer arithmetic, as cv2.clipLine clips it) to a
                    # box far outside any image, so its direction stays what it was; clamping the end points one by one
                    # would turn the line
                    seg = _clip_segment(x0, y0, x1, y1, 1 << 24)
                    if seg is None:
                        seg = (lim, lim, lim, lim)       # nowhere near the image: a point that is dropped
                    x0, y0, x1, y1 = seg
                rec[i, :6] = (0, x0, y0, x1, y1, p[4])
            elif p[0] == "circle":
                x0, y0 = (min(max(v, -lim), lim) for v in (x0, y0))     # a circle that far away touches no image
                rec[i, :6] = (1, x0, y0, p[2], 0, p[4])
            else:
                raise ValueError("unknown primitive %r" % (p[0],))
            col[i, 24:28] = np.clip(c, 0, 255)
        out = np.empty(img.shape, np.uint8)
        with self.lock:
            self._check(self.lib.imatch_draw_overlay(self._h, _ptr(img, c_u8p), img.shape[0], img.shape[1], ch, img.strides[0],
                                                     rec.ctypes.data_as(c_vp), len(prims), _ptr(out, c_u8p)))
        return out

    # -- AKAZE ------------------------------------------------------------------------------
    def akaze_extract(self, img, params=None, image=False):
        """[CLAHE ->] AKAZE(DESCRIPTOR_KAZE, PM_G2) detectAndCompute -> (kp f32 [K,7], desc f32 [K,64]) and, with
        image=True, the CLAHE output as the third item."""
        img = self._gray(img)
        params = params or AkazeParams()
        n = c_i64()
        with self.lock:
            self._check(self.lib.imatch_akaze_extract(self._h, _ptr(img, c_u8p), img.shape[0], img.shape[1], img.strides[0],
                                                      ctypes.byref(params), ctypes.byref(n)))
            kp = np.empty((n.value, 7), np.float32)
            desc = np.empty((n.value, 64), np.float32)
            self._check(self.lib.imatch_akaze_download(self._h, _ptr(kp, c_f32p), _ptr(desc, c_f32p)))
            if not image:
                return kp, desc
            out = np.empty(img.shape, np.uint8) if params.clahe else img.copy()
            if params.clahe:
                self._check(self.lib.imatch_akaze_image(self._h, _ptr(out, c_u8p), img.shape[0], img.shape[1]))
        return kp, desc, out

    def fset(self, desc):
        """Upload a float descriptor matrix [n, d <= 128] once; pass the result as `t` of knn2_f32."""
        desc = np.ascontiguousarray(desc, np.float32)
        if desc.ndim != 2:
            raise ValueError("expected a 2-D descriptor matrix")
        h = c_vp()
        with self.lock:
            self._check(self.lib.imatch_fset_from_host(self._h, _ptr(desc, c_f32p), desc.shape[0], desc.shape[1],
                                                       ctypes.byref(h)))
        return FSet(self, h, desc.shape[0], desc.shape[1])

    def knn2_f32(self, q, t):
        """BFMatcher(NORM_L2).knnMatch(q, t, k=2) for float descriptors -> (idx int32 [Q,2], dist f32 [Q,2]).
        t: a matrix (uploaded for this call) or an FSet (resident)."""
        if isinstance(t, FSet):
            q = np.ascontiguousarray(q, np.float32).reshape(-1, t.d)
            idx = np.empty((len(q), 2), np.int32)
            dist = np.empty((len(q), 2), np.float32)
            with self.lock:
                self._check(self.lib.imatch_knn2_fset(self._h, _ptr(q, c_f32p), len(q), t._h, _ptr(idx, c_i32p),
                                                      _ptr(dist, c_f32p)))
            return idx, dist
        q = np.ascontiguousarray(q, np.float32)
        t = np.ascontiguousarray(t, np.float32)
        d = q.shape[1] if q.ndim == 2 else (t.shape[1] if t.ndim == 2 else 64)
        q, t = q.reshape(-1, d), t.reshape(-1, d)
        idx = np.empty((len(q), 2), np.int32)
        dist = np.empty((len(q), 2), np.float32)
        with self.lock:
            self._check(self.lib.imatch_knn2_f32(self._h, _ptr(q, c_f32p), len(q), _ptr(t, c_f32p), len(t), d,
                                                 _ptr(idx, c_i32p), _ptr(dist, c_f32p)))
        return idx, dist

    def knn2_f32_last_path(self):
        """Test hook: 0 = the last float kNN call ran the exhaustive kernel, 1 = filter + refinement, 2 = filter, then exhaustive."""
        return int(self.lib.imatch_knn2_f32_debug_last_path(self._h))

    def akaze_last(self):
        """(serial, n) of this context's last AKAZE extraction, whose descriptors are still on the device; serial 0: none."""
        serial, n = ctypes.c_uint64(0), c_i64(0)
        with self.lock:
            self._check(self.lib.imatch_akaze_last(self._h, ctypes.byref(serial), ctypes.byref(n)))
        return int(serial.value), int(n.value)

    def knn2_fset_akaze(self, serial, t):
        """knn2_f32 with the descriptors of AKAZE extraction `serial` as the queries, taken where the extraction left
        them on the device (no download / upload round trip), against the resident set t. Raises ImatchError when the
        context has run another AKAZE extraction since."""
        _, n = self.akaze_last()
        idx = np.empty((n, 2), np.int32)
        dist = np.empty((n, 2), np.float32)
        with self.lock:
            self._check(self.lib.imatch_knn2_fset_akaze(self._h, ctypes.c_uint64(int(serial)), t._h, _ptr(idx, c_i32p),
                                                        _ptr(dist, c_f32p)))
        return idx, dist

    def akaze_extract_dev(self, img_ptr, h, w, stride, params=None):
        """The same for a u8 image already in device memory (a torch tensor's data_ptr()) -> (kp, desc)."""
        params = params or AkazeParams()
        n = c_i64()
        with self.lock:
            self._check(self.lib.imatch_akaze_extract_dev(self._h, c_vp(img_ptr), int(h), int(w), int(stride),
                                                          ctypes.byref(params), ctypes.byref(n)))
            kp = np.empty((n.value, 7), np.float32)
            desc = np.empty((n.value, 64), np.float32)
            self._check(self.lib.imatch_akaze_download(self._h, _ptr(kp, c_f32p), _ptr(desc, c_f32p)))
        return kp, desc

    def akaze_debug_layer(self, kind, level):
        """Test hook: plane `kind` (0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet, 5 contrast factor) of evolution level `level`."""
        r, c = ctypes.c_int(), ctypes.c_int()
        with self.lock:
            self._check(self.lib.imatch_akaze_debug_layer(self._h, kind, level, None, ctypes.byref(r), ctypes.byref(c)))
            out = np.empty((r.value, c.value), np.float32)
            self._check(self.lib.imatch_akaze_debug_layer(self._h, kind, level, _ptr(out, c_f32p), ctypes.byref(r),
                                                          ctypes.byref(c)))
        return out

    def akaze_levels(self):
        return int(self.lib.imatch_akaze_debug_levels(self._h))

    def sift_extract(self, img, params=None):
        """[CLAHE ->] SIFT detectAndCompute; the result is a Features set resident in HBM."""
        img = self._gray(img)
        params = params or SiftParams()
        h = c_vp()
        with self.lock:
            self._check(self.lib.imatch_sift_extract(self._h, _ptr(img, c_u8p), img.shape[0], img.shape[1],
                                                     img.strides[0], ctypes.byref(params), ctypes.byref(h)))
            n = self.lib.imatch_features_count(h)
        return Features(self, h, n)

    def sift_extract_dev(self, img_ptr, h, w, stride=None, params=None):
        """SIFT on a u8 image already in device memory (e.g. a torch tensor's data_ptr())."""
        params = params or SiftParams()
        hnd = c_vp()
        with self.lock:
            self._check(self.lib.imatch_sift_extract_dev(self._h, c_vp(img_ptr), int(h), int(w),
                                                         int(stride if stride is not None else w),
                                                         ctypes.byref(params), ctypes.byref(hnd)))
            n = self.lib.imatch_features_count(hnd)
        return Features(self, hnd, n)

    def sift_extract_batch_dev(self, img_ptr, n_img, img_pitch, h, w, stride=None, params=None):
        """SIFT on n_img equally sized u8 images in device memory (image b at img_ptr + b*img_pitch bytes), every
        kernel launched once for the batch. -> (Features holding all rows, offsets int64 [n_img+1])."""
        params = params or SiftParams()
        hnd = c_vp()
        offs = np.zeros(int(n_img) + 1, np.int64)
        with self.lock:
            self._check(self.lib.imatch_sift_extract_batch_dev(
                self._h, c_vp(img_ptr), int(n_img), int(img_pitch), int(h), int(w), int(stride if stride is not None else w),
                ctypes.byref(params), ctypes.byref(hnd), offs.ctypes.data_as(ctypes.POINTER(c_i64))))
            n = self.lib.imatch_features_count(hnd)
        return Features(self, hnd, n), offs

    def sift_debug_layer(self, kind, o, i):
        r, c = ctypes.c_int(), ctypes.c_int()
        with self.lock:
            self._check(self.lib.imatch_sift_debug_layer(self._h, kind, o, i, None, ctypes.byref(r), ctypes.byref(c)))
            out = np.empty((r.value, c.value), np.float32)
            self._check(self.lib.imatch_sift_debug_layer(self._h, kind, o, i, _ptr(out, c_f32p), ctypes.byref(r),
                                                         ctypes.byref(c)))
        return out

    def sift_debug_layer_rows(self, o, i, row0, nrows):
        """Rows [row0, row0 + nrows) of Gaussian layer i of octave o of the last extraction (test hook)."""
        r, c = ctypes.c_int(), ctypes.c_int()
        with self.lock:
            self._check(self.lib.imatch_sift_debug_layer(self._h, 0, int(o), int(i), None, ctypes.byref(r), ctypes.byref(c)))
            out = np.empty((int(nrows), c.value), np.float32)
            self._check(self.lib.imatch_sift_debug_layer_rows(self._h, int(o), int(i), int(row0), int(nrows), _ptr(out, c_f32p)))
        return out

    def sift_debug_octaves(self):
        return self.lib.imatch_sift_debug_octaves(self._h)

    # -- homography ------------------------------------------------------------------------
    def homography(self, src, dst, thr=0.1, seed=0):
        """cv2.findHomography(src, dst, cv2.USAC_MAGSAC, thr) -> (H float64 [3,3], mask uint8 [n,1]) or
        (None, None) when no model exists."""
        src = np.ascontiguousarray(np.asarray(src, dtype=np.float32).reshape(-1, 2))
        dst = np.ascontiguousarray(np.asarray(dst, dtype=np.float32).reshape(-1, 2))
        if src.shape != dst.shape:
            raise ValueError("src and dst must have the same number of points")
        n = src.shape[0]
        H = np.zeros((3, 3), np.float64)
        mask = np.zeros((max(n, 1), 1), np.uint8)
        found = ctypes.c_int(0)
        with self.lock:
            self._check(self.lib.imatch_homography(self._h, _ptr(src, c_f32p), _ptr(dst, c_f32p), n, float(thr),
                                                   ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
                                                   H.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                   _ptr(mask, c_u8p), ctypes.byref(found)))
        if not found.value:
            return None, None
        return H, mask[:n]

    def homography_batch(self, src, dst, offsets, thr=0.1, seeds=None):
        """len(offsets)-1 independent findHomography problems in one launch: problem b is rows
        offsets[b]..offsets[b+1] of src/dst. -> (H float64 [n,3,3], mask uint8 [total], found bool [n])."""
        src = np.ascontiguousarray(np.asarray(src, dtype=np.float32).reshape(-1, 2))
        dst = np.ascontiguousarray(np.asarray(dst, dtype=np.float32).reshape(-1, 2))
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        n_seg = len(offs) - 1
        if src.shape != dst.shape or src.shape[0] != int(offs[-1]):
            raise ValueError("src, dst and offsets disagree")
        sd = np.ascontiguousarray(seeds if seeds is not None else np.zeros(n_seg), dtype=np.uint64)
        H = np.zeros((n_seg, 3, 3), np.float64)
        mask = np.zeros(max(src.shape[0], 1), np.uint8)
        found = np.zeros(n_seg, np.int32)
        with self.lock:
            self._check(self.lib.imatch_homography_batch(
                self._h, _ptr(src, c_f32p), _ptr(dst, c_f32p), offs.ctypes.data_as(ctypes.POINTER(c_i64)), n_seg, float(thr),
                sd.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), H.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                _ptr(mask, c_u8p), found.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return H, mask[:src.shape[0]], found.astype(bool)

    # -- kNN ---------------------------------------------------------------------------
    def knn2(self, q, t):
        """-> (idx int32 [Q,2], dist f32 [Q,2]); missing neighbours are (-1, inf)."""
        idx = np.empty((q.n, 2), np.int32)
        dist = np.empty((q.n, 2), np.float32)
        with self.lock:
            self._check(self.lib.imatch_knn2(self._h, q._h, t._h, _ptr(idx, c_i32p), _ptr(dist, c_f32p)))
        return idx, dist

    def knn2_dev(self, q, t, out4_ptr, idx_offset=0):
        with self.lock:
            self._check(self.lib.imatch_knn2_dev(self._h, q._h, t._h, int(idx_offset), c_vp(out4_ptr)))

    def knn2_noalloc_dev(self, q, t, out4_ptr, idx_offset=0):
        """knn2_dev with the scratch in the library's fixed per-GPU block: no allocation (imatch_knn2_noalloc_dev)."""
        with self.lock:
            self._check(self.lib.imatch_knn2_noalloc_dev(self._h, q._h, t._h, int(idx_offset), c_vp(out4_ptr)))

    def knn2_last_chunks(self):
        """Train chunks of this context's last kNN launch (test hook)."""
        return int(self.lib.imatch_knn2_debug_last_chunks(self._h))

    def merge_top2_dev(self, in4_ptr, n_lists, nq, out4_ptr):
        with self.lock:
            self._check(self.lib.imatch_merge_top2_dev(self._h, c_vp(in4_ptr), int(n_lists), int(nq), c_vp(out4_ptr)))

    def ratio_dev(self, out4_ptr, nq, ratio):
        gq = np.empty(nq, np.int32)
        gt = np.empty(nq, np.int32)
        gd = np.empty(nq, np.float32)
        n = c_i64(0)
        with self.lock:
            self._check(self.lib.imatch_ratio_dev(self._h, c_vp(out4_ptr), int(nq), float(ratio), _ptr(gq, c_i32p),
                                                  _ptr(gt, c_i32p), _ptr(gd, c_f32p), ctypes.byref(n)))
        g = n.value
        return gq[:g].copy(), gt[:g].copy(), gd[:g].copy()

    def match_localize(self, q, t, ratio=0.7, min_good=10, thr=0.1, seed=0):
        """One frame's match and localisation in one call (imatch_match_localize): kNN of the frame's set `q` against the
        map's set `t` (both from sift_extract*), ratio test, and -- with more than `min_good` survivors,
        position_finder.py:316 -- the homography of the matched keypoints. -> (good_q, good_t, good_dist, H or None,
        mask or None): what knn2_dev + ratio_dev + homography give, bit for bit, without the host round trip in between."""
        nq = len(q)
        gq = np.empty(max(nq, 1), np.int32)
        gt = np.empty(max(nq, 1), np.int32)
        gd = np.empty(max(nq, 1), np.float32)
        mask = np.zeros((max(nq, 1), 1), np.uint8)
        H = np.zeros((3, 3), np.float64)
        n = c_i64(0)
        found = ctypes.c_int(0)
        with self.lock:
            self._check(self.lib.imatch_match_localize(self._h, q._h, t._h, float(ratio), int(min_good), float(thr),
                                                       ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), _ptr(gq, c_i32p),
                                                       _ptr(gt, c_i32p), _ptr(gd, c_f32p), ctypes.byref(n),
                                                       H.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _ptr(mask, c_u8p),
                                                       ctypes.byref(found)))
        g = n.value
        if not found.value:
            return gq[:g], gt[:g], gd[:g], None, None
        return gq[:g], gt[:g], gd[:g], H, mask[:g]

    def match_localize_mismatches(self):
        return int(self.lib.imatch_match_localize_debug_mismatches(self._h))

    def ratio_batch_dev(self, out4_ptr, offsets, ratio):
        """Ratio test on the rows of a batch: -> (good_q local to each segment, good_t, good_d, good_offsets)."""
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        n_seg = len(offs) - 1
        nq = int(offs[-1])
        gq = np.empty(max(nq, 1), np.int32)
        gt = np.empty(max(nq, 1), np.int32)
        gd = np.empty(max(nq, 1), np.float32)
        goff = np.zeros(n_seg + 1, np.int64)
        with self.lock:
            self._check(self.lib.imatch_ratio_batch_dev(self._h, c_vp(out4_ptr), offs.ctypes.data_as(ctypes.POINTER(c_i64)),
                                                        n_seg, float(ratio), _ptr(gq, c_i32p), _ptr(gt, c_i32p),
                                                        _ptr(gd, c_f32p), goff.ctypes.data_as(ctypes.POINTER(c_i64))))
        g = int(goff[-1])
        return gq[:g].copy(), gt[:g].copy(), gd[:g].copy(), goff
