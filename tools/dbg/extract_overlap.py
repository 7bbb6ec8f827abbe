"""Do two batch extractions on two contexts (streams) overlap usefully? The scale space is HBM-bound and the descriptor
kernel VALU-bound, so one batch's descriptors beside the next batch's scale space could cost less than their sum.
Measures extractions per second with 1, 2 and 3 contexts, each driven by its own host thread (no kNN anywhere).
Usage: extract_overlap.py [batch] [reps]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
params = im.SiftParams()
map_img = synth.synth_map(2048, seed=1234)
frames = np.stack([synth.synth_frame(map_img, i, size=(480, 640))[0] for i in range(B)])
fr = torch.from_numpy(frames).to(dev)
torch.cuda.synchronize()


def run(n_ctx):
    ctxs = [im.Context(0) for _ in range(n_ctx)]

    def work(cx, n):
        for _ in range(n):
            f, offs = cx.sift_extract_batch_dev(fr.data_ptr(), B, 480 * 640, 480, 640, 640, params)
            cx.synchronize()
            f.free()
    for cx in ctxs:
        work(cx, 2)
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(cx, reps)) for cx in ctxs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    print("%d context(s): %.2f ms per extraction of %d frames (%.0f frames/s)" % (n_ctx, dt / (reps * n_ctx) * 1e3, B, B * reps * n_ctx / dt), flush=True)


for n in (1, 2, 3, 1):
    run(n)
