"""Development check: the kNN launch against itself -- the same call twice, and shards merged against the whole set."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth

ctx = im.Context(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
for nq, nt in [(24000, 150001), (24000, 75001), (43000, 75000), (3000, 150001), (24000, 14200), (8000, 300000)]:
    t = synth.synth_descriptors(nt, seed=nt)
    t[rng.integers(0, nt, nt // 5)] = t[rng.integers(0, nt, nt // 5)]
    q, _, _ = synth.synth_queries(t, nq, seed=nq, dup_frac=0.3)
    fq, ft = ctx.features(q), ctx.features(t)
    outs = []
    for rep in range(3):
        o = torch.empty((nq, 4), dtype=torch.int32, device=dev)
        ctx.knn2_dev(fq, ft, o.data_ptr())
        ctx.synchronize()
        outs.append(o.cpu().numpy())
    same = [np.array_equal(outs[0], x) for x in outs[1:]]
    half = nt // 2 + 1
    fa, fb = ctx.features(t[:half]), ctx.features(t[half:])
    lists = torch.empty((2, nq, 4), dtype=torch.int32, device=dev)
    ctx.knn2_dev(fq, fa, lists[0].data_ptr(), idx_offset=0)
    ctx.knn2_dev(fq, fb, lists[1].data_ptr(), idx_offset=half)
    m = torch.empty((nq, 4), dtype=torch.int32, device=dev)
    ctx.merge_top2_dev(lists.data_ptr(), 2, nq, m.data_ptr())
    ctx.synchronize()
    mm = m.cpu().numpy()
    bad = np.nonzero((mm != outs[0]).any(1))[0]
    print(nq, nt, "repeat identical:", same, "shard-merge mismatches:", len(bad), flush=True)
    for i in bad[:5]:
        print("   q", i, "whole", outs[0][i], "merged", mm[i], "again", outs[1][i], flush=True)
    for f in (fq, ft, fa, fb):
        f.free()
