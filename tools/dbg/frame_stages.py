"""One frame at a time through the single-frame entry points, as bench.py's `single_frame` does it: wall time per
frame beside the library's own event brackets of the stages (imatch_stage_times) -- the difference is launch gaps and
host round trips. Usage: frame_stages.py [map_side] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth

side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
ctx = im.Context(0)
prof = os.environ.get("FS_NOPROF") is None      # FS_NOPROF=1: no event brackets (what they cost: compare the wall times)
ctx.set_profiling(prof)
params = im.SiftParams()
map_img = synth.synth_map(side, seed=1234)
f_map = ctx.sift_extract(map_img, params)
map_xy = np.ascontiguousarray(f_map.download(descriptors=False)[0][:, :2])
frames = [torch.from_numpy(synth.synth_frame(map_img, i, size=(480, 640))[0]).to(dev) for i in range(8)]
out4 = torch.empty((8192, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
import gc
if os.environ.get("FS_GC") == "freeze":
    gc.collect(); gc.freeze()
elif os.environ.get("FS_GC") == "disable":
    gc.disable()
acc = {}
phases = np.zeros(5)
wall = 0.0
for it in range(reps + 5):
    fr = frames[it % 8]
    t0 = time.perf_counter()
    f = ctx.sift_extract_dev(fr.data_ptr(), 480, 640, 640, params)
    t1 = time.perf_counter()
    if os.environ.get("FS_ONE_CALL"):
        gq, gt, gd, H, _ = ctx.match_localize(f, f_map, 0.7, 10, 0.1, seed=it)
        f.free()
        t5 = time.perf_counter()
        if it >= 5:
            wall += t5 - t0
            phases += np.array([t1 - t0, t5 - t1, 0, 0, 0])
            if prof:
                for k, v in ctx.stage_times().items():
                    acc[k] = acc.get(k, 0.0) + v
        continue
    ctx.knn2_dev(f, f_map, out4.data_ptr())
    gq, gt, gd = ctx.ratio_dev(out4.data_ptr(), len(f), 0.7)
    if os.environ.get("FS_SYNC"):
        ts = time.perf_counter()
        ctx.synchronize()
        if it >= 5:
            acc["(synchronize after ratio_dev, wall)"] = acc.get("(synchronize after ratio_dev, wall)", 0.0) + (time.perf_counter() - ts) * 1e3
    t2 = time.perf_counter()
    kp = f.download(descriptors=False)[0]
    src, dst = kp[gq, :2], map_xy[gt]
    t3 = time.perf_counter()
    if os.environ.get("FS_RAWCALL"):
        import ctypes
        from image_matching_amd._lib import _ptr, c_f32p, c_u8p
        q0 = time.perf_counter()
        s32 = np.ascontiguousarray(src, dtype=np.float32); d32 = np.ascontiguousarray(dst, dtype=np.float32)
        q1 = time.perf_counter()
        Hh = np.zeros((3, 3)); mk = np.zeros((len(s32), 1), np.uint8); fnd = ctypes.c_int(0)
        q2 = time.perf_counter()
        args = (ctx._h, _ptr(s32, c_f32p), _ptr(d32, c_f32p), len(s32), 0.1, ctypes.c_uint64(it),
                Hh.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _ptr(mk, c_u8p), ctypes.byref(fnd))
        q3 = time.perf_counter()
        if it >= 5:
            for nm, v in (("imports", q0 - t3), ("ascontiguousarray", q1 - q0), ("zeros", q2 - q1), ("pointers", q3 - q2)):
                acc["(raw set-up: %s)" % nm] = acc.get("(raw set-up: %s)" % nm, 0.0) + v * 1e3
        ta = time.perf_counter()
        ctx.lib.imatch_homography(*args)
        tb = time.perf_counter()
        if it >= 5:
            acc["(raw: argument set-up, wall)"] = acc.get("(raw: argument set-up, wall)", 0.0) + (ta - t3) * 1e3
            acc["(raw: the C call, wall)"] = acc.get("(raw: the C call, wall)", 0.0) + (tb - ta) * 1e3
    else:
        H, _ = ctx.homography(src, dst, 0.1, seed=it)
    t4 = time.perf_counter()
    if os.environ.get("FS_TWICE"):
        H, _ = ctx.homography(src, dst, 0.1, seed=it)
        if it >= 5:
            acc["(second homography call, wall)"] = acc.get("(second homography call, wall)", 0.0) + (time.perf_counter() - t4) * 1e3
    f.free()
    t5 = time.perf_counter()
    if it >= 5:
        wall += t5 - t0
        phases += np.diff([t0, t1, t2, t3, t4, t5])
        if prof:
            for k, v in ctx.stage_times().items():
                acc[k] = acc.get(k, 0.0) + v
print("wall %.3f ms per frame (%d keypoints, %d good, %d map rows)" % (wall / reps * 1e3, len(f), len(gq), len(f_map)))
for n, a in zip(["extract", "knn+ratio", "gather", "homography", "free"], phases / reps * 1e3):
    print("  host phase %-12s %.3f ms" % (n, a))
tot = 0.0
for k, v in acc.items():
    print("  stage %-18s %.3f ms" % (k, v / reps))
    tot += v / reps
print("  stages sum %.3f ms" % tot)
