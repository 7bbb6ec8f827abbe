"""Development: per-workgroup time stamps of the 16x16x32 kNN kernel (library built with -DKNN_TRACE for knn.hip).
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
live = tr[:, 0] > 0
tr = tr[live]
t0 = tr[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("workgroups traced:", len(tr))
print("start   : min %.1f  median %.1f  max %.1f us" % (us(tr[:, 0]).min(), np.median(us(tr[:, 0])), us(tr[:, 0]).max()))
print("prologue: median %.2f us (max %.2f)" % (np.median(tr[:, 1] - tr[:, 0]) / 100, (tr[:, 1] - tr[:, 0]).max() / 100))
nst = ((tr[:, 4::2] > 0).sum(axis=1))
print("stages per workgroup: min %d max %d" % (nst.min(), nst.max()))
for k in range(int(nst.max())):
    ok = nst > k
    top = tr[ok, 4 + 2 * k]
    cmp_ = tr[ok, 5 + 2 * k]
    nxt = np.where(nst[ok] > k + 1, tr[ok, 4 + 2 * (k + 1)] if 4 + 2 * (k + 1) < SL else tr[ok, 2], tr[ok, 2])
    prev = tr[ok, 1] if k == 0 else tr[ok, 5 + 2 * (k - 1)]
    print("stage %2d: wait+barrier (from prev compute start) %.2f us, exchange %.2f us, compute->next top %.2f us   [n=%d]"
          % (k, np.median(top - prev) / 100, np.median(cmp_ - top) / 100, np.median(nxt - cmp_) / 100, ok.sum()))
print("merge+write: median %.2f us" % (np.median(tr[:, 3] - tr[:, 2]) / 100))
print("workgroup total: median %.1f us, max %.1f us; last end at %.1f us" % (np.median(tr[:, 3] - tr[:, 0]) / 100, (tr[:, 3] - tr[:, 0]).max() / 100, us(tr[:, 3]).max()))
