"""Host-side phases of the AKAZE single-frame chain of bench.py's `akaze_path` (extract, kNN vs the resident crop set,
ratio, homography). Usage: akaze_chain_phases.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import image_matching_amd as im
from image_matching_amd import synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = im.Context(0)
map_img = synth.synth_map(4096, seed=1234)
crop = np.ascontiguousarray(map_img[:2048, :2048])
p = im.AkazeParams()
ckp, cdesc = ctx.akaze_extract(crop, p)
cset = ctx.fset(cdesc)
frames = [synth.synth_frame(crop, 500 + i, size=(480, 640), scale_range=(0.9, 1.1))[0] for i in range(8)]
acc = np.zeros(5)
n = 0
for r in range(reps + 1):
    for i, f in enumerate(frames):
        t = [time.perf_counter()]
        kp, desc = ctx.akaze_extract(f, p); t.append(time.perf_counter())
        idx, dist = ctx.knn2_fset_akaze(ctx.akaze_last()[0], cset); t.append(time.perf_counter())
        idx2, dist2 = ctx.knn2_f32(desc, cset); t.append(time.perf_counter())
        keep = dist[:, 0].astype(np.float64) < 0.7 * dist[:, 1].astype(np.float64)
        q = np.nonzero(keep)[0]; t.append(time.perf_counter())
        H = None
        if len(q) > 10:
            H, _ = ctx.homography(kp[q, :2], ckp[idx[q, 0], :2], 0.1, seed=i)
        t.append(time.perf_counter())
        if r > 0:
            acc += np.diff(t); n += 1
for name, a in zip(["akaze_extract (host in/out)", "knn2_fset_akaze (device queries)", "knn2_f32 (uploaded queries)", "ratio (numpy)", "homography"], acc / n * 1e3):
    print("%-34s %.3f ms" % (name, a))
print("%d keypoints per frame, %d crop descriptors" % (len(kp), len(ckp)))
