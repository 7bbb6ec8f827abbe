"""Development: phase time stamps of the small-octave kernel (library built with -DTAIL_TRACE for sift.hip)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth, _lib

ctx = im.Context(0)
m = synth.synth_map(2048, seed=3)
fr = torch.from_numpy(synth.synth_frame(m, 1, size=(480, 640))[0]).cuda()
for _ in range(4):
    f = ctx.sift_extract_dev(fr.data_ptr(), 480, 640, 640, im.SiftParams())
    ctx.synchronize()
    f.free()
lib = ctypes.CDLL(_lib._LIB_PATH)
buf = np.zeros(256, dtype=np.int64)
assert lib.imatch_dbg_tail_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf[buf > 0]
d = np.diff(t) / 100.0
names = ["load", "mirror"]
for i in range(1, 6):
    names += ["row%d" % i, "col%d" % i] + (["mirror%d" % i] if i < 5 else [])
    if i >= 3:
        names += ["scan%d" % (i - 2)]
per = len(names)
print("stamps", len(t), "phases per octave", per, "total %.1f us" % ((t[-1] - t[0]) / 100.0))
for o in range(len(d) // per):
    seg = d[o * per:(o + 1) * per]
    print("octave +%d: %.1f us | " % (o, seg.sum()) + " ".join("%s %.2f" % (n, v) for n, v in zip(names, seg)))
