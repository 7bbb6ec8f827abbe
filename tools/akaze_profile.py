"""AKAZE path timing on one MI355X: extraction of a 640x480 frame and of a 2048^2 crop, the float kNN between them.
Usage: python tools/akaze_profile.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import image_matching_amd as im  # noqa: E402
from image_matching_amd import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = im.Context(0)
    m = synth.synth_map(2048, seed=5)
    frame, _ = synth.synth_frame(m, 3, size=(480, 640), scale_range=(0.9, 1.1))
    p = im.AkazeParams()
    for name, img in (("frame 640x480", frame), ("crop 2048x2048", m)):
        ctx.akaze_extract(img, p)
        t = time.perf_counter()
        for _ in range(reps):
            kp, desc = ctx.akaze_extract(img, p)
        ms = (time.perf_counter() - t) / reps * 1e3
        print("%-16s %8.3f ms  %6d keypoints  (host pointers in and out)" % (name, ms, len(kp)), flush=True)
        if name.startswith("frame"):
            fk, fd = kp, desc
        else:
            mk, md = kp, desc
    ctx.knn2_f32(fd, md)
    t = time.perf_counter()
    for _ in range(reps):
        ctx.knn2_f32(fd, md)
    ms = (time.perf_counter() - t) / reps * 1e3
    print("knn2_f32 %d x %d x 64: %.3f ms (uploads included)" % (len(fd), len(md), ms), flush=True)


main()
