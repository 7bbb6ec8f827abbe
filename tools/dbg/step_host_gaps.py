"""Development: host-side time stamps of the batched step's calls (no extra synchronisation), to find where the GPU idles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth

B = 64
dev = torch.device("cuda", 0)
ctx = im.Context(0)
params = im.SiftParams()
map_img = synth.synth_map(8192, seed=1234)
f_map = ctx.sift_extract(map_img, params)
frames = np.stack([synth.synth_frame(map_img, i, size=(480, 640))[0] for i in range(B)])
fr = torch.from_numpy(frames).to(dev)
out4 = torch.empty((B * 8192, 4), dtype=torch.int32, device=dev)
acc = np.zeros(5)
n = 0
t_prev = None
for it in range(25):
    t0 = time.perf_counter()
    f, offs = ctx.sift_extract_batch_dev(fr.data_ptr(), B, 480 * 640, 480, 640, 640, params)
    t1 = time.perf_counter()
    ctx.knn2_dev(f, f_map, out4.data_ptr())
    t2 = time.perf_counter()
    gq, gt, gd, goff = ctx.ratio_batch_dev(out4.data_ptr(), offs, 0.7)
    t3 = time.perf_counter()
    f.free()
    t4 = time.perf_counter()
    if it >= 5:
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, (t0 - t_prev)]
        n += 1
    t_prev = t4
for name, v in zip(["extract call", "knn2_dev call (queues only)", "ratio_batch_dev call (waits for the kNN)", "free", "python between steps"], acc / n * 1e3):
    print("%-45s %8.3f ms" % (name, v))
print("step %.3f ms" % (acc.sum() / n * 1e3))
