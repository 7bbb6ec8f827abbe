"""Where does a homography call's time go between Python and the library? Usage: homo_call.py"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import image_matching_amd as im
from image_matching_amd import synth

ctx = im.Context(0)
ctx.set_profiling(os.environ.get("FS_NOPROF") is None)
params = im.SiftParams()
map_img = synth.synth_map(4096, seed=1234)
f_map = ctx.sift_extract(map_img, params)
map_xy = np.ascontiguousarray(f_map.download(descriptors=False)[0][:, :2])
frame = synth.synth_frame(map_img, 3, size=(480, 640))[0]
acc = np.zeros(6)
reps = 200
for it in range(reps + 5):
    f = ctx.sift_extract(frame, params)
    idx, dist = ctx.knn2(f, f_map)
    keep = np.nonzero(dist[:, 0].astype(np.float64) < 0.7 * dist[:, 1].astype(np.float64))[0]
    kp = f.download(descriptors=False)[0]
    t0 = time.perf_counter()
    src, dst = kp[keep, :2], map_xy[idx[keep, 0]]
    t1 = time.perf_counter()
    H, _ = ctx.homography(src, dst, 0.1, seed=it)
    t2 = time.perf_counter()
    H, _ = ctx.homography(src, dst, 0.1, seed=it)
    t3 = time.perf_counter()
    time.sleep(0.0002)
    t4 = time.perf_counter()
    H, _ = ctx.homography(src, dst, 0.1, seed=it)
    t5 = time.perf_counter()
    ctx.synchronize()
    t6 = time.perf_counter()
    H, _ = ctx.homography(src, dst, 0.1, seed=it)
    t7 = time.perf_counter()
    f.free()
    if it >= 5:
        acc += [t1 - t0, t2 - t1, t3 - t2, t5 - t4, t6 - t5, t7 - t6]
for n, a in zip(["gather", "first call", "second call", "call after a 0.2 ms sleep", "synchronize (idle stream)", "call after it"], acc / reps * 1e6):
    print("%-28s %.1f us" % (n, a))
print(len(keep), "correspondences")
