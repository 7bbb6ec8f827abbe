"""kNN tuning on the bench workload: one frame's descriptors vs the 8192^2 map's, env knobs per run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import image_matching_amd as im
from image_matching_amd import synth
ctx = im.Context(0); ctx.set_profiling(True)
side = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
map_img = synth.synth_map(side, seed=1234)
f_map = ctx.sift_extract(map_img)
fr = synth.synth_frame(map_img, 3, size=(480, 640))[0]
f = ctx.sift_extract(fr)
nq = len(f)
out = torch.empty((8192, 4), dtype=torch.int32, device="cuda:0")
ref = None
def run(label, **env):
    global ref
    for k in ("IMATCH_KNN_QB", "IMATCH_KNN_XCH", "IMATCH_KNN_CHUNKS"):
        os.environ.pop(k, None)
    for k, v in env.items():
        os.environ["IMATCH_KNN_" + k] = str(v)
    ms = []
    for i in range(6):
        ctx.knn2_dev(f, f_map, out.data_ptr()); ctx.synchronize()
        ms.append(ctx.knn2_kernel_ms())
    got = out[:nq].cpu().numpy().copy()
    if ref is None: ref = got
    print("%-28s %.3f ms (min %.3f)  %s" % (label, np.median(ms[1:]), min(ms), "same" if np.array_equal(got, ref) else "DIFFERENT"), flush=True)
print("Q=%d N=%d" % (nq, len(f_map)))
run("default")
for c in (48, 64, 80, 96, 128, 160, 240):
    run("chunks %d" % c, CHUNKS=c)
