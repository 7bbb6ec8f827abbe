"""Extract an N x N synthetic map a few times (for rocprofv3 --kernel-trace)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import image_matching_amd as im
from image_matching_amd import synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = im.Context(0)
ctx.set_profiling(True)
rng = np.random.default_rng(0)
base = synth.synth_map(1024, seed=1)
img = np.tile(base, (side // 1024, side // 1024))
img = (img.astype(np.int16) + rng.integers(-3, 4, img.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)
for i in range(reps):
    t = time.time()
    f = ctx.sift_extract(img)
    print("extract %d: %.3f s, %d kp, scale-space %.2f ms" % (i, time.time() - t, len(f), ctx.sift_pyramid_ms()), flush=True)
    f.free()
