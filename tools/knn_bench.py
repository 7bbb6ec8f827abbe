"""Quick kNN timing (HIP events on the context's stream via torch ExternalStream)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_matching_amd as im


def gen(n, seed):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    # gamma-ish SIFT-like u8: square of uniform scaled
    x = torch.rand((n, 128), device="cuda:0", generator=g)
    x = (x ** 3) * 140
    return x.to(torch.uint8)


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    ctx = im.Context(0)
    t = gen(n, 1)
    q = t[torch.randperm(n, device="cuda:0")[:nq]].clone()
    q = (q.to(torch.int16) + torch.randint(-6, 7, q.shape, device="cuda:0", dtype=torch.int16)).clamp(0, 255).to(torch.uint8)
    ft = ctx.features_from_device_u8(t.data_ptr(), n)
    fq = ctx.features_from_device_u8(q.data_ptr(), nq)
    out4 = torch.empty((nq, 4), dtype=torch.int32, device="cuda:0")
    s = torch.cuda.ExternalStream(ctx.stream, device="cuda:0")
    ctx.knn2_dev(fq, ft, out4.data_ptr())
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        ctx.knn2_dev(fq, ft, out4.data_ptr())
    e1.record(s)
    ctx.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gd = nq * n / ms / 1e6
    print("Q=%d N=%d: %.3f ms  %.1f Gdist/s  %.1f TFLOP/s (%.1f%% of 2.5 PF)" % (nq, n, ms, gd, gd * 256 / 1e3, gd * 256 / 1e3 / 2500 * 100))


main()
