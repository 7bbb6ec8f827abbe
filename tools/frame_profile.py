"""Host-side timing of one frame's chain, phase by phase (each phase ends with a stream sync, so the sum
is an upper bound of the pipelined step). Usage: frame_profile.py [map_side] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import sharded, synth

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
ctx = im.Context(0)
ctx.set_profiling(True)
params = im.SiftParams()
map_img = synth.synth_map(side, seed=1234)
f_map = ctx.sift_extract(map_img, params)
map_xy = np.ascontiguousarray(f_map.download(descriptors=False)[0][:, :2])
matcher = sharded.ShardedMatcher(ctx, f_map, 0, dev)
frames = [torch.from_numpy(synth.synth_frame(map_img, i, size=(480, 640))[0]).to(dev) for i in range(8)]
names = ["extract", "extract+sync", "match_dp", "match+sync", "ratio_dev", "kp download", "homography", "free"]
acc = np.zeros(len(names))
pyr = knn = 0.0
for it in range(reps + 5):
    fr = frames[it % 8]
    t = [time.perf_counter()]
    f = ctx.sift_extract_dev(fr.data_ptr(), 480, 640, 640, params); t.append(time.perf_counter())
    ctx.synchronize(); t.append(time.perf_counter())
    out4 = matcher.match_dp(f, len(f), 8192); t.append(time.perf_counter())
    ctx.synchronize(); t.append(time.perf_counter())
    gq, gt, gd = ctx.ratio_dev(out4.data_ptr(), out4.shape[0], 0.7); t.append(time.perf_counter())
    kp = f.download(descriptors=False)[0]; t.append(time.perf_counter())
    H, _ = ctx.homography(kp[gq, :2], map_xy[gt], 0.1, seed=it); t.append(time.perf_counter())
    f.free(); t.append(time.perf_counter())
    if it >= 5:
        acc += np.diff(t)
        pyr += ctx.sift_pyramid_ms(); knn += ctx.knn2_kernel_ms()
for n, a in zip(names, acc / reps * 1e3):
    print("%-14s %8.3f ms" % (n, a))
print("total          %8.3f ms   (scale-space events %.3f ms, kNN kernel %.3f ms, %d map descriptors)" % (
    acc.sum() / reps * 1e3, pyr / reps, knn / reps, len(f_map)))
