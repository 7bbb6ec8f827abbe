"""N > 1 path on CPU: world_size-2 gloo processes run ShardedMatcher (partitioning, all_gather /
all_to_all plumbing, merge semantics, tie-break across shard borders). The HIP kernels cannot run here,
so the test injects a stand-in backend built on the CPU oracle -- test infrastructure, not a product
fallback; on the GPU box the same class is driven by the real Context (bench.py --gpus N)."""
import ctypes
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _view(ptr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = (ctypes.c_char * n).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class _FakeFeatures:
    def __init__(self, rows):
        self.rows = np.ascontiguousarray(rows, dtype=np.uint8)
        self.n = len(self.rows)

    def __len__(self):
        return self.n

    def desc_u8_dev(self, ptr):
        _view(ptr, (self.n, 128), np.uint8)[:] = self.rows

    def free(self):
        pass


class _FakeBackend:
    """knn2_dev / merge_top2_dev / features_from_device_u8 with the C-ABI's semantics, on host memory."""
    stream = None

    def features(self, rows):
        return _FakeFeatures(rows)

    def features_from_device_u8(self, ptr, n, d=128):
        return _FakeFeatures(_view(ptr, (n, 128), np.uint8).copy())

    def features_from_device_u8_seg(self, ptr, counts, seg_pitch_rows):
        slots = _view(ptr, (len(counts), seg_pitch_rows, 128), np.uint8)
        return _FakeFeatures(np.concatenate([slots[b, :c] for b, c in enumerate(counts)] + [np.zeros((0, 128), np.uint8)]))

    def knn2_dev(self, q, t, out_ptr, idx_offset=0):
        import oracle
        out = _view(out_ptr, (q.n, 4), np.int32)
        idx, _ = oracle.knn2(q.rows, t.rows)
        for k in range(2):
            ok = idx[:, k] >= 0
            d = q.rows[ok].astype(np.int64) - t.rows[idx[ok, k]].astype(np.int64)
            out[:, 2 * k] = np.iinfo(np.int32).max
            out[:, 2 * k + 1] = -1
            out[ok, 2 * k] = (d * d).sum(1)
            out[ok, 2 * k + 1] = idx[ok, k] + idx_offset

    def merge_top2_dev(self, in_ptr, n_lists, nq, out_ptr):
        lists = _view(in_ptr, (n_lists, nq, 4), np.int32)
        out = _view(out_ptr, (nq, 4), np.int32)
        cand = np.concatenate([lists[:, :, 0:2], lists[:, :, 2:4]], axis=0).astype(np.int64)  # [2L, nq, 2]
        key = np.where(cand[:, :, 1] >= 0, cand[:, :, 0] * (1 << 32) + cand[:, :, 1], np.iinfo(np.int64).max)
        order = np.argsort(key, axis=0, kind="stable")[:2]
        for k in range(2):
            sel = np.take_along_axis(cand, order[k][None, :, None].repeat(2, 2), 0)[0]
            valid = np.take_along_axis(key, order[k][None], 0)[0] != np.iinfo(np.int64).max
            out[:, 2 * k] = np.where(valid, sel[:, 0], np.iinfo(np.int32).max)
            out[:, 2 * k + 1] = np.where(valid, sel[:, 1], -1)

    def synchronize(self):
        pass


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from image_matching_amd import sharded, synth
        train = synth.synth_descriptors(3001, seed=13)               # odd size: ragged last shard
        queries = []
        for r in range(world):
            q, _, _ = synth.synth_queries(train, 150 + 37 * r, seed=20 + r, dup_frac=0.1)
            queries.append(q)
        # a planted tie that straddles the shard border: rows 1500 and 1501 identical to query 0 of rank 0
        train[1500] = queries[0][0]
        train[1501] = queries[0][0]
        lo, hi = sharded.partition(len(train), world)[rank]
        be = _FakeBackend()
        m = sharded.ShardedMatcher(be, be.features(train[lo:hi]), lo, "cpu")
        # replicated: everybody matches rank 0's frame
        out = m.match_replicated(be.features(queries[0]), len(queries[0])).numpy().copy()
        ridx, _ = oracle.knn2(queries[0], train)
        ok_rep = np.array_equal(out[:, [1, 3]], ridx) and out[0, 0] == 0 and out[0, 2] == 0
        # data-parallel frames: each rank matches its own frame
        mine = queries[rank]
        out2 = m.match_dp(be.features(mine), len(mine), q_cap=256).numpy().copy()
        ridx2, rdist2 = oracle.knn2(mine, train)
        ok_dp = np.array_equal(out2[:, [1, 3]], ridx2) and np.array_equal(
            np.sqrt(out2[:, [0, 2]].astype(np.float32)), rdist2)
        # a map that fits one GPU is replicated: the whole set on every rank, no exchange (bench.py --map-mode replicated)
        m_rep = sharded.ShardedMatcher(be, be.features(train), 0, "cpu", local_only=True)
        out3 = m_rep.match_dp(be.features(mine), len(mine), q_cap=256).numpy().copy()
        ok_local = m_rep.world == 1 and np.array_equal(out3, out2)
        # ragged exchange: rank 1 sends no rows at all, rank 0 fills its slot to the brim; stale rows of the earlier,
        # larger call are still in the slots and must not be matched
        mine4 = queries[0][:256 if rank == 0 else 0] if len(queries[0]) >= 256 else np.repeat(queries[0], 2, 0)[:256 if rank == 0 else 0]
        out4 = m.match_dp(be.features(mine4), len(mine4), q_cap=256).numpy().copy()
        ok_ragged = out4.shape == (len(mine4), 4) and m.last_counts == [256, 0]
        if len(mine4):
            ridx4, _ = oracle.knn2(mine4, train)
            ok_ragged = ok_ragged and np.array_equal(out4[:, [1, 3]], ridx4)
        ok_dp = ok_dp and ok_ragged
        ret[rank] = (bool(ok_rep), bool(ok_dp and ok_local), out2.shape[0])
    finally:
        dist.destroy_process_group()


def test_partition_is_contiguous_and_complete():
    from image_matching_amd import sharded
    for n, w in [(10, 3), (8, 8), (5, 8), (0, 2), (5_000_000, 8)]:
        p = sharded.partition(n, w)
        assert len(p) == w and p[0][0] == 0 and p[-1][1] == n
        assert all(p[i][1] == p[i + 1][0] for i in range(w - 1)) and all(a <= b for a, b in p)


def test_sharded_matcher_world2_gloo():
    import oracle
    oracle.lib()  # build before forking workers
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0][:2] == (True, True) and ret[1][:2] == (True, True)
    assert ret[0][2] == 150 and ret[1][2] == 187
