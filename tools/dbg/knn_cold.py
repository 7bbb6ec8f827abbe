"""Development: the kNN launch at the bench shape hot (back to back) and cold (2 GB of other traffic before each launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth

nq, n = 188136, 235864
ctx = im.Context(0)
ctx.set_profiling(True)
dev = torch.device("cuda:0")
st = torch.cuda.ExternalStream(ctx.stream, device=dev)
t = synth.synth_descriptors_dev(0, n, dev, seed=7)
q, _, _ = synth.synth_queries_dev(t.clone(), nq, seed=8, n_dup=100)
ft = ctx.features_from_device_u8(t.data_ptr(), n)
fq = ctx.features_from_device_u8(q.data_ptr(), nq)
out4 = torch.empty((nq, 4), dtype=torch.int32, device=dev)
big = torch.empty(1 << 29, dtype=torch.float32, device=dev)   # 2 GB
for _ in range(3):
    ctx.knn2_dev(fq, ft, out4.data_ptr())
ctx.synchronize()
hot, cold, idle = [], [], []
for _ in range(8):
    ctx.knn2_dev(fq, ft, out4.data_ptr()); ctx.synchronize(); hot.append(ctx.knn2_kernel_ms())
for _ in range(8):
    with torch.cuda.stream(st):
        big.add_(1.0)
    ctx.knn2_dev(fq, ft, out4.data_ptr()); ctx.synchronize(); cold.append(ctx.knn2_kernel_ms())
import time
for gap in (0.0002, 0.001, 0.003, 0.02):
    idle = []
    for _ in range(8):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < gap:
            pass
        ctx.knn2_dev(fq, ft, out4.data_ptr()); ctx.synchronize(); idle.append(ctx.knn2_kernel_ms())
    print("idle %.1f ms before the launch: %.3f ms (min %.3f)" % (gap * 1e3, np.mean(idle), min(idle)))
# a memory-bound phase before the launch, as the extraction is
for _ in range(8):
    with torch.cuda.stream(st):
        for _ in range(3):
            big.add_(1.0)
    ctx.knn2_dev(fq, ft, out4.data_ptr()); ctx.synchronize(); cold.append(ctx.knn2_kernel_ms())
print("after 3 ms of streaming kernels: %.3f ms" % np.mean(cold[8:]))
print("hot  %.3f ms (min %.3f)" % (np.mean(hot), min(hot)))
print("cold %.3f ms (min %.3f)  -- after 4 GB of streaming traffic" % (np.mean(cold), min(cold)))
