"""Development: host phases of the AKAZE frame extraction (IMATCH_TIMING=1 prints them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import image_matching_amd as im
from image_matching_amd import synth
ctx = im.Context(0)
m = synth.synth_map(2048, seed=5)
frame, _ = synth.synth_frame(m, 3, size=(480, 640), scale_range=(0.9, 1.1))
p = im.AkazeParams()
for _ in range(3):
    ctx.akaze_extract(frame, p)
t = time.perf_counter()
for _ in range(10):
    kp, desc = ctx.akaze_extract(frame, p)[:2]
print("frame %.3f ms, %d keypoints" % ((time.perf_counter() - t) / 10 * 1e3, len(kp)))
