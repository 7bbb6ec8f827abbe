"""ctypes binding of include/imatch.h. There is no CPU fallback: a missing libimatch.so or a
missing GPU raises."""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libimatch.so")
_lib = None

c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_i64 = ctypes.c_int64
c_vp = ctypes.c_void_p


class ImatchError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/imatch.h declares
SIGNATURES = {
    "imatch_device_count": (ctypes.c_int, []),
    "imatch_init": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_destroy": (None, [c_vp]),
    "imatch_last_error": (ctypes.c_char_p, [c_vp]),
    "imatch_stream": (c_vp, [c_vp]),
    "imatch_synchronize": (ctypes.c_int, [c_vp]),
    "imatch_features_from_host_f32": (ctypes.c_int, [c_vp, c_f32p, c_i64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_features_from_host_u8": (ctypes.c_int, [c_vp, c_u8p, c_i64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_features_from_device_u8": (ctypes.c_int, [c_vp, c_vp, c_i64, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "imatch_features_count": (c_i64, [c_vp]),
    "imatch_features_is_integral": (ctypes.c_int, [c_vp]),
    "imatch_features_free": (None, [c_vp]),
    "imatch_knn2": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32p, c_f32p]),
    "imatch_knn2_dev": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int32, c_vp]),
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


class Features:
    """A descriptor set resident in HBM (bf16 rows + squared norms). Outlives the call that made
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

    def free(self):
        if self._h is not None and self.ctx._h is not None:
            self.ctx.lib.imatch_features_free(self._h)
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
        self.lock = threading.Lock()

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
