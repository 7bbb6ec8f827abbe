"""Extract an N x N synthetic map a few times (for rocprofv3 --kernel-trace).
Usage: extract_profile.py [side] [reps] [warm]   warm = 1: keep the GPU busy with matmuls right before each extraction
(shows how much of the scale-space time is clock ramp-up after the host-side pause between extractions)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_matching_amd as im
from image_matching_amd import synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = im.Context(0)
ctx.set_profiling(True)
rng = np.random.default_rng(0)
base = synth.synth_map(1024, seed=1)
img = np.tile(base, (side // 1024, side // 1024))
img = (img.astype(np.int16) + rng.integers(-3, 4, img.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)
if warm:
    import torch
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
for i in range(reps):
    if warm:
        for _ in range(40):
            b = a @ a
    t = time.time()
    f = ctx.sift_extract(img)
    print("extract %d: %.3f s, %d kp, scale-space %.2f ms" % (i, time.time() - t, len(f), ctx.sift_pyramid_ms()), flush=True)
    f.free()
