"""Development: print where the kNN launch and the oracle differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
import image_matching_amd as im
from image_matching_amd import synth
nq, nt = int(sys.argv[1]), int(sys.argv[2])
ctx = im.Context(0)
t = synth.synth_descriptors(nt, seed=11 + nt)
q, _, _ = synth.synth_queries(t, nq, seed=nq, dup_frac=0.05)
idx, dist = ctx.knn2(ctx.features(q), ctx.features(t))
ridx, rdist = oracle.knn2(q, t, threads=8)
bad = np.nonzero((idx != ridx).any(1) | (dist != rdist).any(1))[0]
print("mismatching queries:", len(bad), "of", nq)
for i in bad[:12]:
    d = ((q[i].astype(np.int64) - t[idx[i]].astype(np.int64)) ** 2).sum(1) if (idx[i] >= 0).all() else None
    print(i, "got", idx[i], dist[i] ** 2, "true ssd of got", d, "want", ridx[i], rdist[i] ** 2, " rows%16:", idx[i] % 16, ridx[i] % 16, "tile", idx[i] // 16 % 8, ridx[i] // 16 % 8)
