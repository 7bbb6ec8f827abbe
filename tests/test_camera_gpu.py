"""GPU parity of the camera pre-step (csrc/camera.hip) against the CPU oracle, through the C-ABI: integer arithmetic on
both sides, so the bar is bit-exact (csi_camera.py:138-167,184; position_finder.py:171)."""
import os

import numpy as np
import pytest

import oracle
from image_matching_amd import camera

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _img(h, w, seed, ch=3):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)


@pytest.mark.parametrize("h,w", [(1, 1), (5, 7), (3, 1024), (480, 641), (720, 1280)])
def test_cvt_gray_bit_exact(ctx, h, w):
    img = _img(h, w, h * 1000 + w)
    assert np.array_equal(ctx.cvt_gray(img), oracle.cvt_gray(img))
    assert np.array_equal(ctx.cvt_gray(img, rgb=True), oracle.cvt_gray(img, rgb=True))


def test_cvt_gray_strided_view(ctx):
    big = _img(100, 200, 5)
    view = big[10:90, 17:150]                     # rows 600 bytes apart, start not 4-byte aligned
    assert np.array_equal(ctx.cvt_gray(view), oracle.cvt_gray(np.ascontiguousarray(view)))
    with pytest.raises(ValueError):
        ctx.cvt_gray(big[..., 0])


def test_undistort_reference_frame(ctx):
    """The capture node's call: 1920x1080 BGR, balance=1.0, scale=0.625 (csi_camera.py:184)."""
    img = _img(1080, 1920, 11)
    cam = camera.Camera(ctx)
    got = cam.undistort(img, balance=1.0, scale=0.625)
    want = oracle.undistort(img, camera.K, camera.D, camera.DIM, balance=1.0, scale=0.625)
    assert got.shape == want.shape == (675, 1200, 3)
    assert np.array_equal(got, want)
    # the fused gray output equals cvtColor of the colour result (position_finder.py:171)
    assert np.array_equal(cam.undistort(img, balance=1.0, scale=0.625, to_gray=True), oracle.cvt_gray(want))
    # second frame, same calibration: the maps kept in HBM are reused
    img2 = _img(1080, 1920, 12)
    assert np.array_equal(cam.undistort(img2, balance=1.0, scale=0.625),
                          oracle.undistort(img2, camera.K, camera.D, camera.DIM, balance=1.0, scale=0.625))


@pytest.mark.parametrize("size,balance,scale", [((960, 540), 0.0, 0.5), ((1280, 720), 0.4, 1.0), ((640, 360), 1.0, 0.3)])
def test_undistort_other_calls(ctx, size, balance, scale):
    """Other sizes of the calibration's aspect ratio, balances and crops; a changed calibration rebuilds the maps."""
    img = _img(size[1], size[0], size[0])
    cam = camera.Camera(ctx)
    want = oracle.undistort(img, camera.K, camera.D, camera.DIM, balance=balance, scale=scale)
    assert np.array_equal(cam.undistort(img, balance=balance, scale=scale), want)
    gray = np.ascontiguousarray(img[..., 1])
    assert np.array_equal(cam.undistort(gray, balance=balance, scale=scale),
                          oracle.undistort(gray, camera.K, camera.D, camera.DIM, balance=balance, scale=scale))
    d2 = camera.D * 1.5
    cam2 = camera.Camera(ctx, D_=d2)
    assert np.array_equal(cam2.undistort(img, balance=balance, scale=scale),
                          oracle.undistort(img, camera.K, d2, camera.DIM, balance=balance, scale=scale))


def test_undistort_border_is_constant_zero(ctx):
    """balance 0 keeps the shortest field of view... balance 1 at full size shows the black border cv2.remap fills
    with BORDER_CONSTANT."""
    img = np.full((540, 960, 3), 255, np.uint8)
    cam = camera.Camera(ctx)
    got = cam.undistort(img, balance=1.0, scale=1.0)
    want = oracle.undistort(img, camera.K, camera.D, camera.DIM, balance=1.0, scale=1.0)
    assert np.array_equal(got, want)
    assert got.min() == 0 and got.max() == 255      # border taps read 0, interior reads the source


def test_undistort_golden_and_errors(ctx):
    g = np.load(os.path.join(ROOT, "tests", "golden", "camera_small.npz"))
    cam = camera.Camera(ctx)
    assert np.array_equal(cam.undistort(g["img"], balance=1.0, scale=0.625), g["undistorted"])
    assert np.array_equal(cam.undistort(g["img"], balance=1.0, scale=0.625, to_gray=True), g["gray"])
    assert np.array_equal(camera.estimate_new_camera_matrix(camera.K, camera.D, camera.DIM, 1.0), g["new_K_full_b1"])
    with pytest.raises(AssertionError):           # the reference's own assertion (csi_camera.py:151)
        cam.undistort(np.zeros((100, 100, 3), np.uint8))
    from image_matching_amd import ImatchError
    with pytest.raises(ImatchError):
        ctx.fisheye_undistort(np.zeros((90, 160, 3), np.uint8), camera.K, camera.D, camera.K, window=(100, 0, 100, 50))
