"""Development: per-workgroup time stamps of the kNN kernel (library built with -DKNN_TRACE for knn.hip, tools/dbg/build_trace.sh).
Stamps of wave 0: 0 start, 1 prologue done, 2 last stage done, 3 end; per stage k: 4+3k top (compute of the stage before is
done), 5+3k its own loads have landed, 6+3k the workgroup's barrier has released.
Usage: python tools/dbg/knn_trace.py Q N"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_matching_amd as im
from image_matching_amd import synth, _lib

nq, n = int(sys.argv[1]), int(sys.argv[2])
ctx = im.Context(0)
dev = torch.device("cuda:0")
t = synth.synth_descriptors_dev(0, n, dev, seed=7)
q, _, _ = synth.synth_queries_dev(t.clone(), nq, seed=8, n_dup=2)
ft = ctx.features_from_device_u8(t.data_ptr(), n)
fq = ctx.features_from_device_u8(q.data_ptr(), nq)
out4 = torch.empty((nq, 4), dtype=torch.int32, device=dev)
for _ in range(5):
    ctx.knn2_dev(fq, ft, out4.data_ptr())
ctx.synchronize()
lib = ctypes.CDLL(_lib._LIB_PATH)
SL = 40
buf = np.zeros(2048 * SL, dtype=np.int64)
rc = lib.imatch_dbg_knn_trace(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
tr = buf.reshape(2048, SL)
tr = tr[tr[:, 3] > 0]
t0 = tr[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("workgroups traced:", len(tr))
print("start   : min %.1f  median %.1f  max %.1f us" % (us(tr[:, 0]).min(), np.median(us(tr[:, 0])), us(tr[:, 0]).max()))
print("prologue: median %.2f us (max %.2f)" % (np.median(tr[:, 1] - tr[:, 0]) / 100, (tr[:, 1] - tr[:, 0]).max() / 100))
K = (SL - 4) // 3
for k in range(K):
    top, landed, rel = tr[:, 4 + 3 * k], tr[:, 5 + 3 * k], tr[:, 6 + 3 * k]
    ok = (top > 0) & (rel > 0)
    if not ok.any():
        break
    nxt = tr[:, 4 + 3 * (k + 1)] if k + 1 < K else tr[:, 2]
    ok2 = ok & (nxt > 0)
    print("stage %2d: wait for loads %.2f us, barrier %.2f us, exchange+compute %.2f us   [n=%d]"
          % (k, np.median((landed - top)[ok]) / 100, np.median((rel - landed)[ok]) / 100,
             np.median((nxt - rel)[ok2]) / 100 if ok2.any() else float("nan"), ok.sum()))
print("workgroup total: median %.1f us, max %.1f us; last end at %.1f us" % (np.median(tr[:, 3] - tr[:, 0]) / 100, (tr[:, 3] - tr[:, 0]).max() / 100, us(tr[:, 3]).max()))
