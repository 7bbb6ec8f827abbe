"""kNN launch times over the shapes that matter (HIP events on the context's stream), SURVEY 8(d) descriptor statistics.
Usage: python tools/knn_shapes.py [shape ...] with shape = Q x N, e.g. 188136x235864; env knobs of csrc/knn.hip apply."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import image_matching_amd as im
from image_matching_amd import synth

DEFAULT = ["188136x235864", "2951x235864", "2951x58717", "4096x5000000", "131072x2500000", "94068x1000000"]


def main():
    shapes = [tuple(int(v) for v in a.split("x")) for a in (sys.argv[1:] or DEFAULT)]
    ctx = im.Context(0)
    ctx.set_profiling(True)
    dev = torch.device("cuda:0")
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    n_max = max(n for _, n in shapes)
    t_all = synth.synth_descriptors_dev(0, n_max, dev, seed=7)
    for nq, n in shapes:
        t = t_all[:n]
        q, _, _ = synth.synth_queries_dev(t.clone(), nq, seed=8, n_dup=max(1, nq // 1000))
        ft = ctx.features_from_device_u8(t.data_ptr(), n)
        fq = ctx.features_from_device_u8(q.data_ptr(), nq)
        out4 = torch.empty((nq, 4), dtype=torch.int32, device=dev)
        for _ in range(2):
            ctx.knn2_dev(fq, ft, out4.data_ptr())
        ctx.synchronize()
        reps = 20 if nq * n < 2e11 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kern = []
        e0.record(st)
        for _ in range(reps):
            ctx.knn2_dev(fq, ft, out4.data_ptr())
        e1.record(st)
        ctx.synchronize()
        for _ in range(3):
            ctx.knn2_dev(fq, ft, out4.data_ptr())
            kern.append(ctx.knn2_kernel_ms())
        ms = e0.elapsed_time(e1) / reps
        k = min(kern)
        # (against the bf16 peak, as every earlier table in profiles/ is; the int8 peak the kernel runs on is twice that)
        print("Q=%-7d N=%-9d call %.3f ms  mfma launch(es) %.3f ms  %.1f Gdist/s  %.1f%% of 2.5 PF (launches alone %.1f%%)"
              % (nq, n, ms, k, nq * n / ms / 1e6, nq * n * 256 / ms / 1e9 / 2500 * 100, nq * n * 256 / k / 1e9 / 2500 * 100), flush=True)
        ft.free()
        fq.free()


main()
