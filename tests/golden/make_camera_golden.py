"""Golden fixture of the camera pre-step, made by the CPU oracle (oracle/camera_oracle.c) in this container.
PARITY UNPINNED against OpenCV (absent here); the fixture pins the oracle and the HIP path to each other and to
this recorded state. Usage: python tests/golden/make_camera_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402

# the reference's calibration (csi_camera.py:142-144)
DIM = (1920, 1080)
K = np.array([[980.112887385314, 0.0, 941.3416228048669], [0.0, 984.9820720751916, 539.6096400309403], [0.0, 0.0, 1.0]])
D = np.array([-0.027361240466794334, 0.02977144485814498, -0.06669059605260248, 0.038761129156165954])


def image(h, w, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = 96 + 64 * np.sin(x / 7.0) * np.cos(y / 5.0)
    img = np.stack([base + 20 * k + rng.normal(0, 12, (h, w)) for k in range(3)], axis=-1)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    img = image(135, 240, 77)      # 16:9 like DIM
    und = oracle.undistort(img, K, D, DIM, balance=1.0, scale=0.625)   # csi_camera.py:184
    gray = oracle.cvt_gray(und)
    new_K = oracle.fisheye_new_K(K, D, DIM, 1.0)
    new_K0 = oracle.fisheye_new_K(K, D, DIM, 0.0)
    np.savez_compressed(os.path.join(HERE, "camera_small.npz"), img=img, undistorted=und, gray=gray, new_K_full_b1=new_K,
                        new_K_full_b0=new_K0)
    print(und.shape, gray.shape, new_K)


main()
