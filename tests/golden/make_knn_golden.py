"""Generates tests/golden/knn_q256_n4096.npz from the CPU oracle (the reference has no vectors
of its own: SURVEY.md 8c; parity unpinned). Run from the repo root."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from image_matching_amd import synth  # noqa: E402

t = synth.synth_descriptors(4096, seed=7)
q, src, dup = synth.synth_queries(t, 256, seed=8, dup_frac=0.05)
idx, dist = oracle.knn2(q, t)
gq, gt, gd = oracle.ratio(idx, dist, 0.7)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "knn_q256_n4096.npz"),
                    q=q, t=t, idx=idx, dist=dist, good_q=gq, good_t=gt, good_d=gd, src=src, dup=dup)
print("queries", q.shape, "train", t.shape, "good", len(gq), "ties", int((dup >= 0).sum()))
