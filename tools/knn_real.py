"""The bench's kNN launch on the bench's own data (64 real 640x480 frames against the 8192^2 map's descriptors), alone and
repeated, beside a synthetic set of the same shape: separates what the data costs from what the bench's surroundings cost."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import image_matching_amd as im
from image_matching_amd import synth

ctx = im.Context(0)
ctx.set_profiling(True)
dev = torch.device("cuda:0")
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = synth.synth_map(side, seed=1234)
f_map = ctx.sift_extract(m)
frames = np.stack([synth.synth_frame(m, i, size=(480, 640))[0] for i in range(64)])
fd = torch.from_numpy(frames).to(dev)
f, offs = ctx.sift_extract_batch_dev(fd.data_ptr(), 64, 480 * 640, 480, 640, 640)
nq, n = len(f), len(f_map)
out4 = torch.empty((nq, 4), dtype=torch.int32, device=dev)


def run(fq, ft, label):
    ms = []
    for _ in range(12):
        ctx.knn2_dev(fq, ft, out4.data_ptr())
        ms.append(ctx.knn2_kernel_ms())
    ms = np.array(ms[2:])
    print("%-34s Q=%d N=%d  launch %.3f ms (min %.3f max %.3f)  %.1f%% of 2.5 PF" % (
        label, len(fq), len(ft), ms.mean(), ms.min(), ms.max(), len(fq) * len(ft) * 256 / ms.mean() / 1e9 / 2500 * 100), flush=True)


run(f, f_map, "real frames vs real map")
t = synth.synth_descriptors_dev(0, n, dev, seed=7)
q, _, _ = synth.synth_queries_dev(t.clone(), nq, seed=8, n_dup=nq // 1000)
ft, fq = ctx.features_from_device_u8(t.data_ptr(), n), ctx.features_from_device_u8(q.data_ptr(), nq)
run(fq, ft, "synthetic, same shape")
run(f, ft, "real frames vs synthetic map")
run(fq, f_map, "synthetic queries vs real map")
# the bench's sequence: extraction, then the launch
ms = []
for i in range(8):
    f2, _ = ctx.sift_extract_batch_dev(fd.data_ptr(), 64, 480 * 640, 480, 640, 640)
    ctx.knn2_dev(f2, f_map, out4.data_ptr())
    ms.append(ctx.knn2_kernel_ms())
    f2.free()
print("launch right after an extraction: %.3f ms (min %.3f max %.3f)" % (np.mean(ms[2:]), min(ms[2:]), max(ms[2:])))
