"""bench.py's parity block reports descriptor_l2_max > 0 on its frames while p99 is 0: which record is it, and is it a
pairing artefact (two records at one location) or a descriptor that differs? Pairs records by exact (x, y, size, angle)
first. Usage: parity_outlier.py [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import image_matching_amd as im
from image_matching_amd import synth
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = im.Context(0)
params = im.SiftParams()
map_img = synth.synth_map(8192, seed=1234)
oracle.lib()
for i in range(n):
    frame = synth.synth_frame(map_img, i, size=(480, 640))[0]
    res = oracle.sift(oracle.clahe(frame))
    f = ctx.sift_extract(frame, params)
    gkp, gdesc = f.download()
    f.free()
    okp, odesc = res.kp, res.desc
    print("frame %d: oracle %d records, HIP %d; record tables identical: %s" %
          (i, len(okp), len(gkp), okp.shape == gkp.shape and np.array_equal(okp[:, :6], gkp[:, :6])), flush=True)
    if okp.shape == gkp.shape and np.array_equal(okp[:, :6], gkp[:, :6]):
        l2 = np.linalg.norm(gdesc.astype(np.float64) - odesc.astype(np.float64), axis=1)
        bad = np.nonzero(l2 > 0)[0]
        print("  rows whose descriptors differ: %d" % len(bad))
        for r in bad[:8]:
            d = np.nonzero(gdesc[r] != odesc[r])[0]
            print("  row %d kp %s l2 %.3f elements %s hip %s oracle %s" % (r, okp[r, :6], l2[r], d, gdesc[r][d], odesc[r][d]))
    else:
        key = lambda k: [tuple(np.asarray(r[:6]).view(np.uint32) if False else r[:6]) for r in k]
        so, sg = set(key(okp)), set(key(gkp))
        print("  only in oracle: %d, only in HIP: %d" % (len(so - sg), len(sg - so)))
        for r in list(so - sg)[:5]:
            print("   oracle only", r)
        for r in list(sg - so)[:5]:
            print("   HIP only   ", r)
