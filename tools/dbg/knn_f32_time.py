"""Development: the float kNN against a resident set (the AKAZE path's matcher), host pointers for the queries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import image_matching_amd as im
ctx = im.Context(0)
rng = np.random.default_rng(0)
for nq, nt in ((2853, 26792), (3584, 35235), (3000, 200000)):
    t = rng.random((nt, 64), dtype=np.float32)
    q = rng.random((nq, 64), dtype=np.float32)
    fs = ctx.fset(t)
    for _ in range(3):
        ctx.knn2_f32(q, fs)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.knn2_f32(q, fs)
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("knn2_f32 %d x %d x 64 against a resident set: %.3f ms per call (query upload and result download included), %.1f GFMA/s"
          % (nq, nt, ms, nq * nt * 64 / ms / 1e6))
