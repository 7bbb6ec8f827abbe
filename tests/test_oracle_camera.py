"""CPU checks of the camera pre-step oracle (oracle/camera_oracle.c): independent numpy statements of the same
published algorithms, structural properties, and the committed fixture. Parity with OpenCV itself is unpinned (no cv2
here, no fixtures in the reference); see the file header of camera_oracle.c."""
import ctypes
import os

import numpy as np

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIM = (1920, 1080)
K = np.array([[980.112887385314, 0.0, 941.3416228048669], [0.0, 984.9820720751916, 539.6096400309403], [0.0, 0.0, 1.0]])
D = np.array([-0.027361240466794334, 0.02977144485814498, -0.06669059605260248, 0.038761129156165954])


def test_gray_matches_integer_and_float_statement():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    g = oracle.cvt_gray(img)
    b, gg, r = (img[..., k].astype(np.int64) for k in range(3))
    assert np.array_equal(g, ((b * 3735 + gg * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8))
    assert np.abs(g.astype(np.float64) - (0.114 * b + 0.587 * gg + 0.299 * r)).max() <= 0.5 + 1e-2
    assert np.array_equal(oracle.cvt_gray(img, rgb=True), oracle.cvt_gray(img[..., ::-1]))
    flat = np.full((4, 4, 3), 200, np.uint8)
    assert np.all(oracle.cvt_gray(flat) == 200)


def test_identity_calibration_is_identity():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
    k = np.array([[100.0, 0, 80], [0, 100.0, 45], [0, 0, 1]])
    m1, m2 = oracle.fisheye_maps(k, np.zeros(4), k, (160, 90))
    # D = 0 is still the equidistant fisheye model (theta instead of tan(theta)): only the principal point is fixed
    assert tuple(m1[45, 80]) == (80, 45) and m2[45, 80] == 0
    # a remap through integer maps copies pixels, and taps outside read the constant border
    yy, xx = np.mgrid[0:90, 0:160]
    ident = np.stack([xx, yy], -1).astype(np.int16)
    assert np.array_equal(oracle.remap_linear(img, ident, np.zeros((90, 160), np.uint16)), img)
    shifted = oracle.remap_linear(img, ident + np.int16(-3), np.zeros((90, 160), np.uint16))
    assert np.array_equal(shifted[3:, 3:], img[:-3, :-3]) and not shifted[:3].any() and not shifted[:, :3].any()


def test_remap_fraction_weights():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (20, 30), dtype=np.uint8)
    yy, xx = np.mgrid[0:19, 0:29]
    m1 = np.stack([xx, yy], -1).astype(np.int16)
    a = img[:-1, :-1].astype(np.int64); b = img[:-1, 1:].astype(np.int64)
    c = img[1:, :-1].astype(np.int64); d = img[1:, 1:].astype(np.int64)
    for fx, fy in ((16, 0), (0, 16), (8, 24), (31, 31), (1, 0)):
        m2 = np.full((19, 29), fy * 32 + fx, np.uint16)
        want = (a * (32 - fx) * (32 - fy) + b * fx * (32 - fy) + c * (32 - fx) * fy + d * fx * fy) * 32
        want = ((want + 16384) >> 15).astype(np.uint8)
        assert np.array_equal(oracle.remap_linear(img, m1, m2), want), (fx, fy)
    # half a pixel to the right is the rounded-up mean
    m2 = np.full((19, 29), 16, np.uint16)
    assert np.array_equal(oracle.remap_linear(img, m1, m2), ((a + b + 1) >> 1).astype(np.uint8))


def _maps_numpy(k, d, nk, w, h):
    """The same model evaluated per pixel without OpenCV's running sums."""
    j, i = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    x = (j - nk[0, 2]) / nk[0, 0]
    y = (i - nk[1, 2]) / nk[1, 1]
    r = np.hypot(x, y)
    th = np.arctan(r)
    thd = th * (1 + d[0] * th**2 + d[1] * th**4 + d[2] * th**6 + d[3] * th**8)
    s = np.where(r == 0, 1.0, thd / np.where(r == 0, 1.0, r))
    return k[0, 0] * x * s + k[0, 2], k[1, 1] * y * s + k[1, 2]


def test_maps_match_direct_evaluation():
    nk = oracle.fisheye_new_K(K, D, DIM, 1.0)
    m1, m2 = oracle.fisheye_maps(K, D, nk, DIM)
    u, v = _maps_numpy(K, D, nk, *DIM)
    iu = m1[..., 0].astype(np.int64) * 32 + (m2 & 31)
    iv = m1[..., 1].astype(np.int64) * 32 + (m2 >> 5)
    du = np.abs(iu - np.rint(u * 32)); dv = np.abs(iv - np.rint(v * 32))
    assert du.max() <= 1 and dv.max() <= 1
    assert (du > 0).mean() < 1e-4 and (dv > 0).mean() < 1e-4   # only values on a rounding boundary may differ
    # the rectified image magnifies the centre: source steps are below one pixel per pixel there
    assert 0 < iu[540, 961] - iu[540, 960] < 32 * 2


def test_new_camera_matrix_properties():
    for bal in (0.0, 0.3, 1.0):
        nk = oracle.fisheye_new_K(K, D, DIM, bal)
        assert nk[0, 1] == 0 and nk[2, 2] == 1 and abs(nk[0, 0] / nk[1, 1] - K[0, 0] / K[1, 1]) < 1e-12
    nk1, nk0 = oracle.fisheye_new_K(K, D, DIM, 1.0), oracle.fisheye_new_K(K, D, DIM, 0.0)
    assert nk1[0, 0] < nk0[0, 0]            # balance 1 keeps every source pixel: the shortest focal length
    # with balance 1 the four edge mid-points of the source land inside the rectified frame, one of them on its edge
    w, h = DIM
    # decode the maps back to source coordinates
    m1, m2 = oracle.fisheye_maps(K, D, nk1, DIM)
    u = m1[..., 0] + (m2 & 31) / 32.0
    v = m1[..., 1] + (m2 >> 5) / 32.0
    assert u.min() <= 1.0 or v.min() <= 1.0 or u.max() >= w - 1 or v.max() >= h - 1
    inside = (u >= -1) & (u <= w) & (v >= -1) & (v <= h)
    assert inside[h // 2, :].sum() > 0.95 * w * min(1.0, nk1[0, 0] / nk0[0, 0])
    # a centred principal point stays centred: the four edge mid-points undistort symmetrically
    k = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    z = oracle.fisheye_new_K(k, np.zeros(4), (640, 480), 0.5)
    assert z[0, 0] > 0 and abs(z[0, 2] - 320) < 1e-9 and abs(z[1, 2] - 240) < 1e-9


def test_library_new_camera_matrix_equals_oracle():
    """imatch_fisheye_new_camera_matrix is host arithmetic: callable without a GPU."""
    lib = ctypes.CDLL(os.path.join(ROOT, "image_matching_amd", "libimatch.so"))
    f64p = ctypes.POINTER(ctypes.c_double)
    lib.imatch_fisheye_new_camera_matrix.argtypes = [f64p, f64p, ctypes.c_int, ctypes.c_int, ctypes.c_double, f64p]
    for size, bal in ((DIM, 1.0), (DIM, 0.0), ((1280, 720), 0.4)):
        k = (K * size[0] / DIM[0]).copy()
        k[2, 2] = 1.0
        kk = np.ascontiguousarray(k.reshape(9))
        out = np.empty(9)
        rc = lib.imatch_fisheye_new_camera_matrix(kk.ctypes.data_as(f64p), D.ctypes.data_as(f64p), size[0], size[1], bal,
                                                  out.ctypes.data_as(f64p))
        assert rc == 0
        assert np.array_equal(out.reshape(3, 3), oracle.fisheye_new_K(k, D, size, bal))
    assert lib.imatch_fisheye_new_camera_matrix(None, None, 1, 1, 0.0, None) < 0


def test_golden_fixture():
    g = np.load(os.path.join(ROOT, "tests", "golden", "camera_small.npz"))
    und = oracle.undistort(g["img"], K, D, DIM, balance=1.0, scale=0.625)
    assert np.array_equal(und, g["undistorted"])
    assert np.array_equal(oracle.cvt_gray(und), g["gray"])
    assert np.array_equal(oracle.fisheye_new_K(K, D, DIM, 1.0), g["new_K_full_b1"])
    assert np.array_equal(oracle.fisheye_new_K(K, D, DIM, 0.0), g["new_K_full_b0"])
    assert und.shape == (84, 150, 3)      # int(135 * 0.625), int(240 * 0.625): csi_camera.py:164-166
