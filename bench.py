#!/usr/bin/env python3
"""bench.py -- the reference's headline path on MI355X.

Metric (BASELINE.json): frames/sec of the full chain CLAHE -> SIFT -> kNN(k=2)+ratio -> homography of one
640x480 gray frame against the full map whose descriptors stay resident in HBM (configs[1]; the
reference's map.png is not in its tree, so the "full map" is the seeded synthetic 8192x8192 orthomap of
SURVEY.md 8d). Gdist/s of the kNN stage is reported beside it.

A step = every rank takes ONE BATCH of --batch frames (default 64, already in HBM as one [B,480,640] tensor)
through the batched entry points: extract (every kernel launched once for the batch) -> one kNN launch
over all the batch's keypoints against the whole map -> ratio -> homography per frame (iff > 10 good
matches, position_finder.py:316). Per frame the results are identical to the single-frame entry points
(tests/test_batch_gpu.py); --batch 1 runs those instead, and their latency is reported beside the
headline as "single_frame".

At N > 1 every rank takes its OWN batches (frames are independent units: data-parallel, per-GPU work per step
constant, total frames per step = N * batch: "scaling": "weak"; value = N * batch * steps / wall time, wall =
max over ranks between two barriers). What happens to the map depends on its size (--map-mode):
  replicated  a map whose descriptor set fits one GPU many times over (configs[1]'s does: 60 MB) lives on every
              rank whole; no data-path collective (auto picks this below 16 GiB of packed descriptors);
  sharded     the map's descriptors are split over the ranks by contiguous index ranges, queries are
              all-gathered, every rank matches all batches against its shard, and an all_to_all returns the
              per-shard top-2 to the batch's owner (image_matching_amd/sharded.py: the layout of configs[3]/[4],
              maps of 5M-20M descriptors).
In replicated mode a short sharded loop is run AFTER the headline is final and reported beside it as "sharded_map":
one batch compared bit for bit with the whole-map result, then 10 timed steps. It runs under a watchdog
(IMATCH_BENCH_EXTRA_TIMEOUT seconds, default 120): a failure or a stuck collective there is reported in that
field and cannot change or withhold `value`. IMATCH_BENCH_SHARDED_EXTRA=0 skips it.

One JSON line on stdout (rank 0). Progress goes to stderr."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA (spec)
PYRAMID_BYTES_PER_INPUT_PIXEL = 475.7


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(frames, map_desc_sample, n_map, map_xy_sample, ratio, threads):
    """The same chain through the CPU oracle on this box's host cores, frame after frame for about two seconds of wall
    time (`threads` OpenMP threads: tens of core-seconds). The kNN leg runs against the whole map when it holds at most
    300 000 descriptors, else against a slice and is scaled linearly (it is a dense scan)."""
    import oracle
    oracle.lib()
    oracle.sift(oracle.clahe(frames[0]))          # first touch: page in the library, spin up the OpenMP pool
    t_ext = t_knn_s = t_h = 0.0
    pairs = 0
    done = 0
    kept = []
    t_start = time.perf_counter()
    for frame in frames:
        t0 = time.perf_counter()
        res = oracle.sift(oracle.clahe(frame))
        t1 = time.perf_counter()
        idx, dist = oracle.knn2(res.desc.astype(np.uint8), map_desc_sample, threads=threads)
        t2 = time.perf_counter()
        gq, gt, _ = oracle.ratio(idx, dist, ratio)
        H = None
        if len(gq) >= 4:
            H, _ = oracle.homography(res.kp[gq, :2], map_xy_sample[gt], 0.1, seed=0)
        t3 = time.perf_counter()
        kept.append((res.kp, res.desc, idx, dist, gq, gt, H))
        t_ext += t1 - t0; t_knn_s += t2 - t1; t_h += t3 - t2
        pairs += len(res.desc) * len(map_desc_sample)
        done += 1
        if time.perf_counter() - t_start > 2.0:
            break
    scale = n_map / max(len(map_desc_sample), 1)
    total = t_ext + t_knn_s * scale + t_h
    return {
        "value": done / total, "unit": "frames/s", "cores": threads, "kind": "port",
        "sample": "%d frames 640x480, %.1f s of wall time on %d threads: extract %.3f s/frame (OpenMP) + kNN vs %d map "
                  "descriptors %.3f s/frame%s + ratio and homography %.4f s/frame; restatement of OpenCV, not OpenCV"
                  % (done, t_ext + t_knn_s + t_h, threads, t_ext / done, len(map_desc_sample), t_knn_s / done,
                     "" if scale == 1 else " (scaled x%.1f to %d)" % (scale, n_map), t_h / done),
        "knn_gdist_per_s": pairs / t_knn_s / 1e9,
    }, kept


def parity_block(ctx, params, frames, truth_H, kept, f_map, map_xy, ratio):
    """SURVEY 8(d) "parity checks run with every measurement": the frames the CPU baseline leg just ran through the
    oracle, run through the HIP entry points and compared with what the oracle gave (the oracle is the checker here;
    nothing below is timed)."""
    try:
        from scipy.spatial import cKDTree
        corners = np.array([[0, 0, 1], [639, 0, 1], [639, 479, 1], [0, 479, 1]], np.float64).T
        centre = np.array([319.5, 239.5, 1.0])

        def proj(H, p):
            v = H @ p
            return v[:2] / v[2]
        kp_hit, kp_n, l2 = 0, 0, []
        knn_same = ratio_same = True
        h_err, pos_err = [], []
        for (okp, odesc, oidx, odist, ogq, ogt, oH), frame, Ht in zip(kept, frames, truth_H):
            f = ctx.sift_extract(frame, params)
            gkp, gdesc = f.download()
            d, j = cKDTree(gkp[:, :2]).query(okp[:, :2])
            # a location carries one record per orientation peak: pair records by location first, then by angle
            hit = d < 0.5
            kp_hit += int(hit.sum())
            kp_n += len(okp)
            same = hit & (np.abs((gkp[j, 3] - okp[:, 3] + 180.0) % 360.0 - 180.0) < 1.0)
            l2.extend(np.linalg.norm(gdesc[j[same]] - odesc[same], axis=1).tolist())
            # the matcher on the oracle's own descriptors: indices, distances and ratio decisions bit for bit
            fq = ctx.features(odesc.astype(np.uint8))
            gidx, gdist = ctx.knn2(fq, f_map)
            knn_same &= bool(np.array_equal(gidx, oidx) and np.array_equal(gdist, odist))
            keep = np.nonzero(gdist[:, 0].astype(np.float64) < ratio * gdist[:, 1].astype(np.float64))[0]
            ratio_same &= bool(np.array_equal(keep, ogq) and np.array_equal(gidx[keep, 0], ogt))
            fq.free()
            f.free()
            if oH is not None and len(ogq) > 10:
                gH, _ = ctx.homography(okp[ogq, :2], map_xy[ogt], 0.1, seed=0)
                if gH is not None:
                    h_err.append(float(max(np.hypot(*(proj(gH, c) - proj(oH, c))) for c in corners.T)))
                    pos_err.append(float(np.hypot(*(proj(gH, centre) - proj(Ht, centre)))))
        l2 = np.asarray(l2) if l2 else np.zeros(1)
        return {"frames": len(kept), "keypoints_within_0.5px": kp_hit / max(kp_n, 1),
                "descriptor_l2_p99": float(np.percentile(l2, 99)), "descriptor_l2_max": float(l2.max()),
                "knn_idx_and_dist_bit_exact": knn_same, "ratio_decisions_identical": ratio_same,
                "homography_corner_px_vs_oracle_max": max(h_err) if h_err else None,
                "frame_centre_map_px_vs_planted_truth_median": float(np.median(pos_err)) if pos_err else None,
                "note": "HIP entry points against the CPU oracle on the cpu_baseline leg's frames; after the timed region"}
    except Exception as e:      # must not take the headline down with it
        return {"error": repr(e)[:300]}


def akaze_extra(ctx, map_img, ratio):
    """Beside the headline (not part of `value`): the detector the checked-in reference calls (match_finder.py:210-211),
    one frame at a time as BASELINE configs[0] describes it -- CLAHE + AKAZE of a 640x480 frame, float kNN against the
    AKAZE descriptors of a 2048x2048 crop of the map, ratio test, homography. A failure here is reported in the block."""
    try:
        import image_matching_amd as im
        from image_matching_amd import synth
        crop = np.ascontiguousarray(map_img[:2048, :2048])
        p = im.AkazeParams()
        t0 = time.perf_counter()
        ckp, cdesc = ctx.akaze_extract(crop, p)
        ckp, cdesc = ctx.akaze_extract(crop, p)
        t_crop = (time.perf_counter() - t0) / 2
        cset = ctx.fset(cdesc)      # the crop's descriptors stay in HBM, as the ROI's do between re-extractions
        frames = [synth.synth_frame(crop, 500 + i, size=(480, 640), scale_range=(0.9, 1.1))[0] for i in range(8)]
        found = 0
        n_kp = 0

        def one(f, i):
            kp, desc = ctx.akaze_extract(f, p)
            idx, dist = ctx.knn2_f32(desc, cset)
            keep = dist[:, 0].astype(np.float64) < ratio * dist[:, 1].astype(np.float64)
            q = np.nonzero(keep)[0]
            H = None
            if len(q) > 10:
                H, _ = ctx.homography(kp[q, :2], ckp[idx[q, 0], :2], 0.1, seed=i)
            return len(kp), H is not None
        for i, f in enumerate(frames[:2]):
            one(f, i)
        t0 = time.perf_counter()
        reps = 3
        for r in range(reps):
            for i, f in enumerate(frames):
                n, ok = one(f, i)
                n_kp += n
                found += ok
        ms = (time.perf_counter() - t0) / (reps * len(frames)) * 1e3
        return {"detector": "AKAZE(DESCRIPTOR_KAZE, 0.0005, 4, 5, PM_G2), the call of match_finder.py:210-211",
                "ms_per_frame": ms, "frames_per_s": 1e3 / ms, "keypoints_per_frame": n_kp / (reps * len(frames)),
                "crop": "2048x2048, %d keypoints, extracted in %.1f ms" % (len(ckp), t_crop * 1e3),
                "homographies_found": "%d/%d" % (found, reps * len(frames)),
                "note": "single-frame calls with host pointers in and out (imatch_akaze_extract, imatch_knn2_fset against the "
                        "resident crop set, imatch_homography), measured after the timed region; not part of value"}
    except Exception as e:      # must not take the headline down with it
        return {"error": repr(e)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--map-side", type=int, default=8192)
    ap.add_argument("--frames", type=int, default=128, help="distinct synthetic frames per rank, cycled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=64,
                    help="frames per step and GPU, taken through the batched entry points (one launch per stage for "
                         "the whole batch, one kNN launch over all frames' keypoints); 1 = the single-frame calls")
    ap.add_argument("--map-mode", choices=["auto", "replicated", "sharded"], default="auto",
                    help="N > 1: whole map on every rank (no collective) or map rows sharded over the ranks (all_gather + "
                         "all_to_all per step); auto = replicated when the packed descriptor set is below 16 GiB")
    ap.add_argument("--streams", type=int, default=1,
                    help="frames in flight per GPU: each has its own context (stream + arena) and host thread; a step is "
                         "then one frame per stream")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal aids for a one-GPU box (never set by the driver): IMATCH_BENCH_DEVICE puts every rank on that device,
    # IMATCH_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one GPU.
    backend = os.environ.get("IMATCH_BENCH_BACKEND", "nccl")
    if "IMATCH_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["IMATCH_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world:
        log("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import image_matching_amd as im
    from image_matching_amd import sharded, synth
    ctx = im.Context(local_rank)
    ctx.set_profiling(True)
    ratio = 0.7
    params = im.SiftParams()

    # ---- the map: generated and extracted once; descriptors stay in HBM
    t0 = time.time()
    map_img = synth.synth_map(args.map_side, seed=1234)
    log("[rank %d] synthetic map %dx%d in %.1fs" % (rank, args.map_side, args.map_side, time.time() - t0))
    t0 = time.time()
    f_map = ctx.sift_extract(map_img, params)
    t_map_extract = time.time() - t0
    pyr_ms_map = ctx.sift_pyramid_ms()
    t0 = time.time()
    f_map2 = ctx.sift_extract(map_img, params)   # second pass: arena already allocated
    t_map_extract2 = time.time() - t0
    pyr_ms_map = min(pyr_ms_map, ctx.sift_pyramid_ms())
    f_map2.free()
    n_map = len(f_map)
    map_kp = f_map.download(descriptors=False)[0]
    map_xy = np.ascontiguousarray(map_kp[:, :2])
    log("[rank %d] map: %d descriptors, extract %.2fs (first) %.2fs (warm), scale-space stage %.2f ms"
        % (rank, n_map, t_map_extract, t_map_extract2, pyr_ms_map))
    map_mode = args.map_mode
    if map_mode == "auto":
        map_mode = "replicated" if n_map * 264 < (16 << 30) else "sharded"
    if world == 1:
        map_mode = "replicated"
    lo, hi = sharded.partition(n_map, world)[rank]
    f_shard = None
    if world > 1:
        all_u8 = torch.empty((n_map, 128), dtype=torch.uint8, device=dev)
        f_map.desc_u8_dev(all_u8.data_ptr())
        ctx.synchronize()
        shard_u8 = all_u8[lo:hi].contiguous()
        f_shard = ctx.features_from_device_u8(shard_u8.data_ptr(), hi - lo)
        del all_u8
    if map_mode == "replicated":
        matcher = sharded.ShardedMatcher(ctx, f_map, 0, dev, local_only=True)
        n_local = n_map
    else:
        matcher = sharded.ShardedMatcher(ctx, f_shard, lo, dev)
        n_local = hi - lo
    f_local = f_map if map_mode == "replicated" else f_shard

    # ---- frames: each rank has its own set, resident in HBM before the timed region
    frames_t = [synth.synth_frame(map_img, 100 * rank + i, size=(480, 640)) for i in range(args.frames)]
    frames_h = [f for f, _ in frames_t]
    truth_H = [h for _, h in frames_t]
    frames_d = [torch.from_numpy(f).to(dev) for f in frames_h]
    q_cap = 4096                     # rows per frame in the exchange / result buffers (frames here have ~3000 keypoints)

    stats = {"kp": [], "good": [], "found": 0, "knn_ms": [], "pyr_ms": []}

    # extra frames in flight: one context (stream, arena, workspace) and one matcher per stream
    n_streams = max(1, args.streams)
    if world > 1 and n_streams > 1:
        log("note: --streams applies to single-GPU runs (the collectives are issued from one thread); using 1")
        n_streams = 1
    ctxs = [ctx] + [im.Context(local_rank) for _ in range(n_streams - 1)]
    for c in ctxs[1:]:
        c.set_profiling(True)
    matchers = [matcher] + [sharded.ShardedMatcher(c, f_local, 0 if map_mode == "replicated" else lo, dev,
                                                    local_only=map_mode == "replicated") for c in ctxs[1:]]
    import threading
    stats_lock = threading.Lock()

    def step(i, record=False, lane=0):
        cx, mt = ctxs[lane], matchers[lane]
        fr = frames_d[(i * n_streams + lane) % len(frames_d)]
        f = cx.sift_extract_dev(fr.data_ptr(), 480, 640, 640, params)
        nq = len(f)
        pyr_ms = cx.sift_pyramid_ms() if record else 0.0
        if nq > q_cap:
            raise RuntimeError('frame has %d keypoints, exchange buffers hold %d' % (nq, q_cap))
        out4 = mt.match_dp(f, nq, q_cap)
        gq, gt, gd = cx.ratio_dev(out4.data_ptr(), out4.shape[0], ratio)
        knn_ms = cx.knn2_kernel_ms() if record else 0.0
        H = None
        if len(gq) > 10:                                   # position_finder.py:316
            kp = f.download(descriptors=False)[0]
            H, _ = cx.homography(kp[gq, :2], map_xy[gt], 0.1, seed=i)
        f.free()
        if record:
            with stats_lock:
                stats["pyr_ms"].append(pyr_ms)
                stats["knn_ms"].append(knn_ms)
                stats["kp"].append(nq)
                stats["good"].append(len(gq))
                stats["found"] += H is not None
        return H

    # batched steps: B frames resident as one [B,480,640] tensor per step
    B = max(1, args.batch)
    bq_cap = ((B * 3200 + 1023) // 1024) * 1024      # rows per batch in the exchange / result buffers (frames: ~2940 each)
    if B > 1:
        n_sets = max(2, (len(frames_h) + B - 1) // B)
        batches_d = [torch.from_numpy(np.stack([frames_h[(j * B + k) % len(frames_h)] for k in range(B)])).to(dev)
                     for j in range(n_sets)]

    def step_batch(i, record=False, lane=0, mt=None):
        cx, mt = ctxs[lane], (mt if mt is not None else matchers[lane])
        fr = batches_d[(i * n_streams + lane) % len(batches_d)]
        f, offs = cx.sift_extract_batch_dev(fr.data_ptr(), B, 480 * 640, 480, 640, 640, params)
        nq = len(f)
        pyr_ms = cx.sift_pyramid_ms() if record else 0.0
        if nq > bq_cap:
            raise RuntimeError('batch has %d keypoints, buffers hold %d' % (nq, bq_cap))
        out4 = mt.match_dp(f, nq, bq_cap)
        gq, gt, gd, goff = cx.ratio_batch_dev(out4.data_ptr(), offs, ratio)
        knn_ms = cx.knn2_kernel_ms() if record else 0.0
        ng = np.diff(goff)
        use = ng > 10                                      # position_finder.py:316
        src = f.keypoints_xy(np.repeat(offs[:-1], ng) + gq)
        src_off = goff
        gt_u = gt
        if not use.all():                                  # frames below the gate take no part
            keep = np.repeat(use, ng)
            src, gt_u = src[keep], gt[keep]
            src_off = np.concatenate([[0], np.cumsum(np.where(use, ng, 0))])
        H, _, found = cx.homography_batch(src, map_xy[gt_u], src_off, 0.1, seeds=np.arange(B) + i * B)
        f.free()
        if record:
            with stats_lock:
                stats["pyr_ms"].append(pyr_ms)
                stats["knn_ms"].append(knn_ms)
                stats["kp"].append(nq)
                stats["good"].append(float(ng.mean()))
                stats["found"] += int(found.sum())
        return H

    def run_steps(n, record):
        one = step_batch if B > 1 else step
        if n_streams == 1:
            for i in range(n):
                one(i, record)
            return
        errs = []

        def worker(lane):
            try:
                for i in range(n):
                    one(i, record, lane)
            except Exception as e:      # surfaced after the join
                errs.append(e)
        ts = [threading.Thread(target=worker, args=(l,)) for l in range(n_streams)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]

    def barrier():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # A generational collection over this process's object graph (frame lists, torch tensors) takes ~35 ms; one landing
    # inside a 20-frame latency measurement reads as +1.7 ms per frame. Collect now, then keep the collector off while
    # timing (reference counting still frees everything the steps allocate).
    import gc
    run_steps(args.warmup, False)
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    # N > 1, replicated map: the sharded-map exchange path, run AFTER the headline is final (see the watchdog below)
    def sharded_map_extra():
        try:
            mt_sh = sharded.ShardedMatcher(ctx, f_shard, lo, dev)
            n_sh = 10
            # the exchange path must give what the whole map gives: one batch through both, compared bit for bit
            f_chk, _ = ctx.sift_extract_batch_dev(batches_d[0].data_ptr(), B, 480 * 640, 480, 640, 640, params)
            a = matcher.match_dp(f_chk, len(f_chk), bq_cap)
            ctx.synchronize()
            a = a.clone()
            b = mt_sh.match_dp(f_chk, len(f_chk), bq_cap)
            ctx.synchronize()
            same = torch.tensor([int(torch.equal(a, b))], dtype=torch.int64, device=dev)
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            f_chk.free()
            for i in range(2):
                step_batch(i, False, 0, mt_sh)
            barrier()
            t1 = time.perf_counter()
            for i in range(n_sh):
                step_batch(i, False, 0, mt_sh)
            barrier()
            tw = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            return {"frames_per_s": world * B * n_sh / float(tw.item()), "ms_per_step": float(tw.item()) / n_sh * 1e3,
                    "steps": n_sh, "knn_launch_ms": ctx.knn2_kernel_ms(),
                    "per_launch": "%d query rows (all ranks' batches, padded) x %d shard rows" % (world * bq_cap, hi - lo),
                    "exchange": "all_gather of u8 query rows + all_to_all of per-shard top-2 (RCCL), merge_top2",
                    "identical_to_whole_map": bool(same.item())}
        except Exception as e:    # must not take the headline down with it
            return {"error": repr(e)[:300]}

    # single-frame latency, outside the timed region: the per-frame calls, one frame at a time
    single_ms = None
    if world == 1:
        for i in range(5):
            step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(20):
            step(i)
        barrier()
        single_ms = (time.perf_counter() - t1) / 20 * 1e3

    gc.enable()
    if rank == 0:
        fps = world * n_streams * B * args.steps / wall
        q_mean = float(np.mean(stats["kp"]))
        knn_ms = float(np.mean(stats["knn_ms"]))
        cap_rows = bq_cap if B > 1 else q_cap
        exchange = world > 1 and map_mode == "sharded"
        q_per_launch = (world * cap_rows) if exchange else q_mean     # one launch holds all B frames' keypoints
        flops = 256.0 * q_per_launch * n_local            # 256 flop per (query, train) pair
        achieved_tf = flops / (knn_ms * 1e-3) / 1e12
        pyr_bytes = PYRAMID_BYTES_PER_INPUT_PIXEL * args.map_side * args.map_side
        out = {
            "metric": "frames/sec (SIFT+kNN+RANSAC) vs map",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "640x480 gray frame vs full %dx%d synthetic orthomap, %d map descriptors resident "
                                   "in HBM (configs[1])" % (args.map_side, args.map_side, n_map),
                       "frame": "640x480 u8", "map_descriptors": n_map, "keypoints_per_frame": q_mean / B,
                       "frames_per_step_per_gpu": B * n_streams,
                       "entry_points": "batched (imatch_sift_extract_batch_dev, one kNN launch, imatch_ratio_batch_dev, "
                                       "imatch_homography_batch)" if B > 1 else "single-frame",
                       "good_matches_per_frame": float(np.mean(stats["good"])),
                       "homographies_found": "%d/%d" % (stats["found"], args.steps * B * n_streams), "ratio": ratio,
                       "frames_in_flight_per_gpu": n_streams,
                       "parallelism": "single GPU" if world == 1 else
                                      ("map sharded x%d, frames data-parallel" % world if exchange else
                                       "map replicated on %d GPUs, frames data-parallel, no collective" % world)},
            "knn_gdist_per_s": world * q_per_launch * float(n_local) / (knn_ms * 1e-3) / 1e9,
            "roofline": {"kernel": ("knn2_kernel16<8,128> (bf16 MFMA 16x16x32, fused top-2)" if q_per_launch > 512
                                    else "knn2_kernel (bf16 MFMA 32x32x16, fused top-2)"), "bound": "mfma", "achieved": achieved_tf,
                         "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": achieved_tf / MFMA_BF16_PEAK_TF,
                         "traffic": None, "launch_ms": knn_ms,
                         "per_launch": "%d queries x %d train rows x 256 flop" % (int(q_per_launch), n_local)},
            "roofline_pyramid": {"kernel": "scale-space stage of the %dx%d map extraction (upsample, blur+DoG, "
                                           "decimate, extrema scan)" % (args.map_side, args.map_side),
                                 "bound": "hbm", "achieved": pyr_bytes / (pyr_ms_map * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": pyr_bytes / (pyr_ms_map * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "stage_ms": pyr_ms_map, "frame_stage_ms": float(np.mean(stats["pyr_ms"]))},
        }
        # HBM-side traffic of the dominant kernel comes from separate rocprofv3 --pmc passes over this same command
        # (profiles/r1_pmc.md); it is attached only when that measurement was taken on the same launch shape
        try:
            with open(os.path.join(ROOT, "profiles", "r1_pmc_knn.json")) as fh:
                pmc = json.load(fh)
            if world == 1 and B == pmc.get("batch", 32) and abs(pmc.get("n_queries", q_mean) - q_mean) < 0.05 * q_mean:
                out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
                out["roofline"]["traffic_note"] = "bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE, " + pmc["source"]
        except (OSError, ValueError, KeyError):
            pass
        if single_ms is not None:
            out["single_frame"] = {"ms_per_frame": single_ms, "frames_per_s": 1e3 / single_ms,
                                   "note": "one frame at a time through the single-frame entry points, measured after "
                                           "the timed region; not part of value"}
        if world == 1:
            out["akaze_path"] = akaze_extra(ctx, map_img, ratio)
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            n_s = n_map if n_map <= 300_000 else 100_000
            d_s = f_local.download(keypoints=False)[1][:n_s].astype(np.uint8)
            out["cpu_baseline"], kept = cpu_baseline(frames_h[:16], d_s, n_map, map_xy[:n_s], ratio, threads)
            if n_s == n_map:    # the oracle matched against the whole map: its lists are comparable as they are
                out["parity"] = parity_block(ctx, params, frames_h, truth_H, kept, f_local, map_xy, ratio)
        else:
            out["cpu_baseline"] = None
    else:
        out = None

    # The headline is final here. At N > 1 with a replicated map the sharded-map loop runs now, under a watchdog: this
    # is the only part of the run that depends on the all_gather / all_to_all exchange, and if it does not come back
    # (every rank would then sit in the same collective) each rank's watchdog ends its process with status 0 after rank 0
    # has printed the line without it.
    printed = threading.Lock()

    def emit(extra):
        with printed:
            if rank == 0 and out is not None and "_done" not in out:
                if extra is not None:
                    out["sharded_map"] = extra
                out["_done"] = True
                print(json.dumps({k: v for k, v in out.items() if k != "_done"}), flush=True)

    if world > 1 and map_mode == "replicated" and B > 1 and os.environ.get("IMATCH_BENCH_SHARDED_EXTRA", "1") != "0":
        finished = threading.Event()

        def watchdog():
            if not finished.wait(float(os.environ.get("IMATCH_BENCH_EXTRA_TIMEOUT", "120"))):
                emit({"error": "the sharded-map loop did not finish in time; the headline does not depend on it"})
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        extra = sharded_map_extra()
        finished.set()
        emit(extra)
    else:
        emit(None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
