"""Generates tests/golden/sift_small.npz from the CPU oracle: small images -> CLAHE / INTER_AREA outputs,
SIFT keypoints + descriptors, and one homography case. The reference holds no vectors of its own
(parity unpinned): these pin the ORACLE (and through it the HIP path) against regressions.
Run from the repo root."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from image_matching_amd import synth  # noqa: E402

m = synth.synth_map(512, seed=4321)
imgs = {"tex": m[40:200, 100:340].copy(), "odd": m[300:421, 17:200].copy()}
yy, xx = np.mgrid[0:96, 0:128]
blob = np.full((96, 128), 90, np.float32)
for cx, cy, s, a in [(30, 30, 3.0, 90), (90, 50, 5.0, -60), (60, 75, 2.2, 110)]:
    blob += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
imgs["blob"] = np.clip(np.rint(blob), 0, 255).astype(np.uint8)
out = {}
for name, img in imgs.items():
    out[name + "_img"] = img
    out[name + "_clahe"] = oracle.clahe(img)
    r = oracle.sift(out[name + "_clahe"])
    out[name + "_kp"] = r.kp
    out[name + "_desc"] = r.desc.astype(np.uint8)
    dw, dh = int(img.shape[1] * 0.61), int(img.shape[0] * 0.61)
    out[name + "_area"] = oracle.resize_area(img, (dw, dh))
    print(name, img.shape, "kp", len(r.kp))
rng = np.random.default_rng(3)
Ht = np.array([[0.8, -0.5, 100], [0.55, 0.85, 40], [2e-5, -3e-5, 1]])
src = rng.uniform(0, 600, (120, 2)).astype(np.float32)
p = np.c_[src, np.ones(120)] @ Ht.T
dst = (p[:, :2] / p[:, 2:] + rng.normal(0, 0.3, (120, 2))).astype(np.float32)
dst[::5] = rng.uniform(0, 800, (24, 2)).astype(np.float32)
H, mask = oracle.homography(src, dst, 0.1, seed=11)
out.update(h_src=src, h_dst=dst, h_H=H, h_mask=mask, h_true=Ht)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sift_small.npz"), **out)
