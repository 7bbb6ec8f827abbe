"""Host-side timing of one batched step, phase by phase (each phase ends with a stream sync).
Usage: batch_profile.py [batch] [map_side] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
side = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
ctx = im.Context(0)
ctx.set_profiling(True)
params = im.SiftParams()
map_img = synth.synth_map(side, seed=1234)
f_map = ctx.sift_extract(map_img, params)
map_xy = np.ascontiguousarray(f_map.download(descriptors=False)[0][:, :2])
frames = np.stack([synth.synth_frame(map_img, i, size=(480, 640))[0] for i in range(B)])
fr = torch.from_numpy(frames).to(dev)
out4 = torch.empty((B * 8192, 4), dtype=torch.int32, device=dev)
names = ["extract (host returns)", "extract tail (describe+pack)", "knn", "ratio", "kp download+gather", "homography", "free"]
acc = np.zeros(len(names))
pyr = knn = 0.0
for it in range(reps + 3):
    t = [time.perf_counter()]
    f, offs = ctx.sift_extract_batch_dev(fr.data_ptr(), B, 480 * 640, 480, 640, 640, params); t.append(time.perf_counter())
    ctx.synchronize(); t.append(time.perf_counter())
    ctx.knn2_dev(f, f_map, out4.data_ptr()); ctx.synchronize(); t.append(time.perf_counter())
    gq, gt, gd, goff = ctx.ratio_batch_dev(out4.data_ptr(), offs, 0.7); t.append(time.perf_counter())
    kp = f.download(descriptors=False)[0]
    src = kp[np.repeat(offs[:-1], np.diff(goff)) + gq, :2]; dst = map_xy[gt]; t.append(time.perf_counter())
    H, _, found = ctx.homography_batch(src, dst, goff, 0.1, seeds=np.arange(B)); t.append(time.perf_counter())
    f.free(); t.append(time.perf_counter())
    if it >= 3:
        acc += np.diff(t)
        pyr += ctx.sift_pyramid_ms(); knn += ctx.knn2_kernel_ms()
for n, a in zip(names, acc / reps * 1e3):
    print("%-30s %8.3f ms  (%.3f per frame)" % (n, a, a / B))
print("total %.3f ms = %.3f per frame; scale-space events %.3f ms, kNN kernel %.3f ms, Q=%d, found %d/%d" % (
    acc.sum() / reps * 1e3, acc.sum() / reps * 1e3 / B, pyr / reps, knn / reps, len(f), found.sum(), B))
