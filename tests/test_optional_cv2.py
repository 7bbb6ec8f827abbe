"""Pins the CPU oracle against real OpenCV wherever `cv2` is importable ("If the GPU box has cv2, an optional
test regenerates (i)-(iv) with real cv2 and reports deltas -- never required"). In this image cv2 is absent, so the
whole module is skipped and parity stays UNPINNED; a maintainer with opencv-python >= 4.5 runs
`python -m pytest tests/test_optional_cv2.py -v -s` to pin it (the reference's call sites are cited per test).
These tests have never run here: they state the bars of BASELINE.json's north star, not a measured result."""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

import oracle  # noqa: E402
from image_matching_amd import synth  # noqa: E402


@pytest.fixture(scope="module")
def frame():
    return np.ascontiguousarray(synth.synth_map(1024, seed=3)[100:580, 200:840])     # 480 x 640


def test_clahe_bit_exact(frame):
    """match_finder.py:200,203"""
    want = cv2.createCLAHE(clipLimit=30.0, tileGridSize=(8, 8)).apply(frame)
    got = oracle.clahe(frame, 30.0, (8, 8))
    print("CLAHE: %d of %d pixels differ, max |d| = %d" % ((got != want).sum(), got.size, np.abs(got.astype(int) - want).max()))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dsize", [(320, 240), (427, 320), (213, 160), (800, 600)])
def test_inter_area_bit_exact(frame, dsize):
    """utils.py:17-22"""
    want = cv2.resize(frame, dsize, interpolation=cv2.INTER_AREA)
    got = oracle.resize_area(frame, dsize)
    print("INTER_AREA %s: %d pixels differ" % (dsize, (got != want).sum()))
    assert np.array_equal(got, want)


def test_gray_bit_exact():
    """position_finder.py:171, image_processing.py:54"""
    img = np.random.default_rng(1).integers(0, 256, (97, 131, 3), dtype=np.uint8)
    assert np.array_equal(oracle.cvt_gray(img), cv2.cvtColor(img, cv2.COLOR_BGR2GRAY))
    assert np.array_equal(oracle.cvt_gray(img, rgb=True), cv2.cvtColor(img, cv2.COLOR_RGB2GRAY))


def test_fisheye_undistort_bit_exact():
    """csi_camera.py:138-167,184"""
    from image_matching_amd import camera
    img = np.random.default_rng(2).integers(0, 256, (540, 960, 3), dtype=np.uint8)
    dim1 = (960, 540)
    k = camera.K * dim1[0] / camera.DIM[0]
    k[2][2] = 1.0
    new_k = cv2.fisheye.estimateNewCameraMatrixForUndistortRectify(k, camera.D, dim1, np.eye(3), balance=1.0)
    assert np.allclose(new_k, oracle.fisheye_new_K(k, camera.D, dim1, 1.0), rtol=1e-12, atol=1e-9)
    m1, m2 = cv2.fisheye.initUndistortRectifyMap(k, camera.D, np.eye(3), new_k, dim1, cv2.CV_16SC2)
    want = cv2.remap(img, m1, m2, interpolation=cv2.INTER_AREA, borderMode=cv2.BORDER_CONSTANT)
    o1, o2 = oracle.fisheye_maps(k, camera.D, new_k, dim1)
    print("maps: %d of %d entries differ" % ((o1 != m1).any(axis=2).sum() + (o2 != m2).sum(), o2.size))
    got = oracle.remap_linear(img, m1, m2)         # the remap alone, on cv2's own maps
    assert np.array_equal(got, want)
    assert ((o1 != m1).any(axis=2) | (o2 != m2)).mean() < 1e-4      # the inverse goes through an SVD in cv2


def test_sift_keypoints_and_descriptors(frame):
    """match_finder.py:205-209,220: >= 99 % of cv2's keypoints within 0.5 px, descriptors within L2 16 of 512"""
    eq = cv2.createCLAHE(clipLimit=30.0, tileGridSize=(8, 8)).apply(frame)
    kp, desc = cv2.SIFT_create(0, 3, 0.04, 10, 1.6).detectAndCompute(eq, None)
    res = oracle.sift(eq)
    ref = np.array([[k.pt[0], k.pt[1], k.size, k.angle] for k in kp], np.float32)
    hit = 0
    dl2 = []
    for i, r in enumerate(ref):
        d = np.hypot(res.kp[:, 0] - r[0], res.kp[:, 1] - r[1])
        ok = (d < 0.5) & (np.abs(res.kp[:, 2] - r[2]) < 0.1 * r[2]) & (np.abs((res.kp[:, 3] - r[3] + 180) % 360 - 180) < 2.0)
        if ok.any():
            hit += 1
            j = int(np.argmin(np.where(ok, d, 1e9)))
            dl2.append(np.linalg.norm(res.desc[j] - desc[i]))
    print("SIFT: cv2 %d keypoints, oracle %d, matched %.2f %%, descriptor L2 p50 %.1f p99 %.1f max %.1f" % (
        len(ref), len(res.kp), 100.0 * hit / max(len(ref), 1), np.median(dl2), np.percentile(dl2, 99), np.max(dl2)))
    assert hit >= 0.99 * len(ref)
    assert np.percentile(dl2, 99) <= 16.0


def test_knn_and_ratio_bit_exact():
    """match_finder.py:182-194"""
    t = synth.synth_descriptors(4096, seed=1)
    q, _, _ = synth.synth_queries(t, 256, seed=2, dup_frac=0.05)
    m = cv2.BFMatcher(cv2.NORM_L2).knnMatch(q.astype(np.float32), t.astype(np.float32), k=2)
    idx, dist = oracle.knn2(q, t)
    assert all(a.trainIdx == idx[i, 0] and b.trainIdx == idx[i, 1] for i, (a, b) in enumerate(m))
    assert all(a.distance == dist[i, 0] and b.distance == dist[i, 1] for i, (a, b) in enumerate(m))
    good = [i for i, (a, b) in enumerate(m) if a.distance < 0.7 * b.distance]
    assert np.array_equal(oracle.ratio(idx, dist, 0.7)[0], np.array(good, np.int32))


def test_akaze_keypoints_and_descriptors(frame):
    """match_finder.py:210-211,220"""
    eq = cv2.createCLAHE(clipLimit=30.0, tileGridSize=(8, 8)).apply(frame)
    det = cv2.AKAZE_create(descriptor_type=cv2.AKAZE_DESCRIPTOR_KAZE, threshold=0.0005, nOctaves=4, nOctaveLayers=5,
                           diffusivity=cv2.KAZE_DIFF_PM_G2)
    kp, desc = det.detectAndCompute(eq, None)
    res = oracle.akaze(eq)
    ref = np.array([[k.pt[0], k.pt[1], k.size, k.angle, k.class_id] for k in kp], np.float32)
    hit, dl2 = 0, []
    for i, r in enumerate(ref):
        d = np.hypot(res.kp[:, 0] - r[0], res.kp[:, 1] - r[1]) + 100.0 * (res.kp[:, 6] != r[4])
        j = int(np.argmin(d))
        if d[j] < 0.5:
            hit += 1
            dl2.append(np.linalg.norm(res.desc[j] - desc[i]))
    print("AKAZE: cv2 %d keypoints, oracle %d, matched %.2f %%, descriptor L2 p50 %.3f p99 %.3f" % (
        len(ref), len(res.kp), 100.0 * hit / max(len(ref), 1), np.median(dl2), np.percentile(dl2, 99)))
    assert hit >= 0.99 * len(ref)
    assert np.percentile(dl2, 99) <= 0.2


def test_homography_within_one_pixel(frame):
    """match_finder.py:233: same geometric answer within 1 px corner reprojection"""
    rng = np.random.default_rng(5)
    H = np.array([[0.9, -0.3, 120.0], [0.28, 0.95, 60.0], [1e-5, -2e-5, 1.0]])
    src = rng.uniform(0, [640, 480], (300, 2)).astype(np.float32)
    p = np.c_[src, np.ones(len(src))] @ H.T
    dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
    dst[:60] += rng.uniform(-80, 80, (60, 2)).astype(np.float32)          # outliers
    want, _ = cv2.findHomography(src.reshape(-1, 1, 2), dst.reshape(-1, 1, 2), cv2.USAC_MAGSAC, 0.1)
    got, _ = oracle.homography(src, dst, 0.1, seed=0)
    corners = np.array([[0, 0, 1], [639, 0, 1], [639, 479, 1], [0, 479, 1]], np.float64).T
    a, b = want @ corners, got @ corners
    err = np.hypot(a[0] / a[2] - b[0] / b[2], a[1] / a[2] - b[1] / b[2]).max()
    print("homography corner reprojection vs cv2: %.4f px" % err)
    assert err <= 1.0
