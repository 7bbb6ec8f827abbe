import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import image_matching_amd as im
from image_matching_amd import sharded, synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 55
dev = torch.device("cuda", 0)
ctx = im.Context(0); ctx.set_profiling(True)
params = im.SiftParams()
map_img = synth.synth_map(int(sys.argv[2]) if len(sys.argv) > 2 else 4096, seed=1234)
f_map = ctx.sift_extract(map_img, params)
map_xy = np.ascontiguousarray(f_map.download(descriptors=False)[0][:, :2])
matcher = sharded.ShardedMatcher(ctx, f_map, 0, dev, local_only=True)
frames_h = [synth.synth_frame(map_img, i, size=(480, 640))[0] for i in range(64)]
frames = [torch.from_numpy(f).to(dev) for f in frames_h[:8]]
batch = torch.from_numpy(np.stack(frames_h)).to(dev)
def single(reps):
    names = ["extract", "sync", "match_dp", "sync2", "ratio_dev", "kp download", "homography", "free"]
    acc = np.zeros(len(names))
    for it in range(reps):
        fr = frames[it % 8]
        t = [time.perf_counter()]
        f = ctx.sift_extract_dev(fr.data_ptr(), 480, 640, 640, params); t.append(time.perf_counter())
        ctx.synchronize(); t.append(time.perf_counter())
        out4 = matcher.match_dp(f, len(f), 8192); t.append(time.perf_counter())
        ctx.synchronize(); t.append(time.perf_counter())
        gq, gt, gd = ctx.ratio_dev(out4.data_ptr(), out4.shape[0], 0.7); t.append(time.perf_counter())
        kp = f.download(descriptors=False)[0]; t.append(time.perf_counter())
        H, _ = ctx.homography(kp[gq, :2], map_xy[gt], 0.1, seed=it); t.append(time.perf_counter())
        f.free(); t.append(time.perf_counter())
        acc += np.diff(t)
    print("  ".join("%s %.3f" % (n, a) for n, a in zip(names, acc / reps * 1e3)), " total %.3f ms" % (acc.sum() / reps * 1e3), flush=True)
single(5); single(20)
for i in range(nb):
    f, offs = ctx.sift_extract_batch_dev(batch.data_ptr(), 64, 480 * 640, 480, 640, 640, params)
    out4 = matcher.match_dp(f, len(f), 64 * 3200)
    gq, gt, gd, goff = ctx.ratio_batch_dev(out4.data_ptr(), offs, 0.7)
    ng = np.diff(goff)
    src = f.keypoints_xy(np.repeat(offs[:-1], ng) + gq)
    H, _, found = ctx.homography_batch(src, map_xy[gt], goff, 0.1, seeds=np.arange(64) + i * 64)
    f.free()
ctx.synchronize()
print("after %d batch steps" % nb)
single(5); single(20); single(20)
