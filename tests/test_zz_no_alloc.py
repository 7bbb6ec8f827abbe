"""The entry point that allocates nothing (imatch_knn2_noalloc_dev) against the entry point it is a variant of
(imatch_knn2_dev), through the C-ABI, bit for bit: two calls in a row, the second with a smaller input, so that what the
first call left in the shared per-GPU scratch block must not leak into the second."""
import numpy as np
import pytest

from image_matching_amd import synth

pytestmark = pytest.mark.gpu


def _both(ctx, torch, fq, ft, nq, offset=0):
    a = torch.full((max(nq, 1), 4), -7, dtype=torch.int32, device="cuda:0")
    b = torch.full((max(nq, 1), 4), -9, dtype=torch.int32, device="cuda:0")
    ctx.knn2_dev(fq, ft, a.data_ptr(), idx_offset=offset)
    ctx.knn2_noalloc_dev(fq, ft, b.data_ptr(), idx_offset=offset)
    ctx.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


def test_no_alloc_entry_point_equals_knn2_dev_twice_in_a_row(ctx):
    import torch
    t = synth.synth_descriptors(60000, seed=21)
    t[40000:41000] = t[100:1100]                               # ties across chunks: lowest index first
    q1, _, _ = synth.synth_queries(t, 9000, seed=22, dup_frac=0.2)
    ft, fq1 = ctx.features(t), ctx.features(q1)
    a, b = _both(ctx, torch, fq1, ft, 9000)
    assert np.array_equal(a, b) and (a[:, 1] >= 0).all()
    # second call, smaller on both sides (another kernel variant, fewer chunks): stale lists and bounds must not show
    t2 = t[:5000].copy()
    q2, _, _ = synth.synth_queries(t2, 300, seed=23, dup_frac=0.2)
    ft2, fq2 = ctx.features(t2), ctx.features(q2)
    a2, b2 = _both(ctx, torch, fq2, ft2, 300, offset=1000)
    assert np.array_equal(a2, b2) and (a2[:, 1] >= 1000).all()
    # and against the oracle, so that "equal" is not "equally wrong"
    import oracle
    ridx, rdist = oracle.knn2(q2, t2)
    assert np.array_equal(b2[:, [1, 3]] - 1000, ridx)
    assert np.array_equal(np.sqrt(b2[:, [0, 2]].astype(np.float32)), rdist)
    # a single query row against two rows, and an empty query set
    a3, b3 = _both(ctx, torch, ctx.features(q2[:1]), ctx.features(t2[:2]), 1)
    assert np.array_equal(a3, b3)
    ctx.knn2_noalloc_dev(ctx.features(q2[:0]), ft2, torch.empty((1, 4), dtype=torch.int32, device="cuda:0").data_ptr())


def test_no_alloc_entry_point_many_queries_fewer_chunks_and_limit(ctx):
    """Many queries: the fixed block forces fewer, longer chunks than knn2_dev picks -- same result. 200 000 queries x
    300 000 rows: the launcher picks 24 chunks (300 032 padded rows in chunks of at most 16 k, a multiple of 8); one
    partial list per chunk is 200 704 x 16 B = 3.2 MB and the 64 MiB block, less the 2.4 MB of bound words, has room for
    20, so the call runs on 16 -- the only code it does not share with knn2_dev. The call runs on a FRESH context (its
    workspace is empty, so any use of it would have to allocate) and the device's free memory must be the same before and
    after it. Beyond the stated limit the call is refused with a clear error."""
    import torch
    import image_matching_amd as im
    import oracle
    nq, nt = 200_000, 300_000
    g = torch.Generator(device="cuda:0").manual_seed(3)
    t_dev = torch.randint(0, 256, (nt, 128), dtype=torch.uint8, device="cuda:0", generator=g)
    rows = torch.randint(0, nt, (nq,), device="cuda:0", generator=g)
    q_dev = t_dev[rows].clone()
    q_dev[:, :4] = torch.randint(0, 256, (nq, 4), dtype=torch.uint8, device="cuda:0", generator=g)
    b = torch.full((nq, 4), -9, dtype=torch.int32, device="cuda:0")
    a = torch.full((nq, 4), -7, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()   # torch's stream made these; the contexts' streams read them
    fresh = im.Context(0)
    ft = fresh.features_from_device_u8(t_dev.data_ptr(), nt)
    fq = fresh.features_from_device_u8(q_dev.data_ptr(), nq)
    # a first small call on the fresh context's stream: whatever the HIP runtime sets up lazily for a new queue is in place
    # before the measurement (the call's own scratch would be 79 MB at this size, were it to come from the workspace)
    warm_t = fresh.features_from_device_u8(t_dev.data_ptr(), 5000)
    warm_q = fresh.features_from_device_u8(q_dev.data_ptr(), 300)
    fresh.knn2_noalloc_dev(warm_q, warm_t, b.data_ptr())
    fresh.synchronize()
    free_before = torch.cuda.mem_get_info(0)[0]
    fresh.knn2_noalloc_dev(fq, ft, b.data_ptr())
    fresh.synchronize()
    delta = free_before - torch.cuda.mem_get_info(0)[0]
    print("free device memory across the no-alloc call: %d bytes less" % delta)
    assert delta == 0, "the entry point that allocates nothing changed the free device memory by %d bytes" % delta
    assert fresh.knn2_last_chunks() == 16, "the fixed block should have forced 16 chunks"
    fresh.knn2_dev(fq, ft, a.data_ptr())       # 24 chunks, lists in the context's workspace (grown here)
    fresh.synchronize()
    assert fresh.knn2_last_chunks() == 24
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert np.array_equal(a, b) and (b[:, 1] >= 0).all() and (b[:, 3] >= 0).all()
    # the oracle on a sample of the queries against all rows
    pick = np.random.default_rng(5).choice(nq, 64, replace=False)
    ridx, rdist = oracle.knn2(q_dev.cpu().numpy()[pick], t_dev.cpu().numpy())
    assert np.array_equal(b[pick][:, [1, 3]], ridx)
    assert np.array_equal(np.sqrt(b[pick][:, [0, 2]].astype(np.float32)), rdist)
    big = torch.zeros((262_145, 128), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    fbig = fresh.features_from_device_u8(big.data_ptr(), 262_145)
    out = torch.empty((262_145, 4), dtype=torch.int32, device="cuda:0")
    with pytest.raises(im.ImatchError, match="at most 262144"):
        fresh.knn2_noalloc_dev(fbig, ft, out.data_ptr())
    for f in (fbig, fq, ft, warm_q, warm_t):
        f.free()
    fresh.close()
