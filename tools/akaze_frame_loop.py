"""AKAZE extraction of one 640x480 frame, repeated (for rocprofv3 --kernel-trace). Usage: akaze_frame_loop.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import image_matching_amd as im  # noqa: E402
from image_matching_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = im.Context(0)
m = synth.synth_map(2048, seed=5)
frame, _ = synth.synth_frame(m, 3, size=(480, 640), scale_range=(0.9, 1.1))
for _ in range(reps):
    kp, desc = ctx.akaze_extract(frame, im.AkazeParams())
print(len(kp))
