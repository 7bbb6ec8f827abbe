#!/usr/bin/env python3
"""bench.py -- the reference's headline path on MI355X.

Metric (BASELINE.json): frames/sec of the full chain CLAHE -> SIFT -> kNN(k=2)+ratio -> homography of one
640x480 gray frame against the full map whose descriptors stay resident in HBM (configs[1]; the
reference's map.png is not in its tree, so the "full map" is the seeded synthetic 8192x8192 orthomap of
SURVEY.md 8d). Gdist/s of the kNN stage is reported beside it.

A step = every rank takes ONE BATCH of --batch frames (default 256 at c1, 64 at c3, 32 at c5; already in HBM as one
[B,480,640] tensor)
through the batched entry points: extract (every kernel launched once for the batch) -> one kNN launch
over all the batch's keypoints against the whole map -> ratio -> homography per frame (iff > 10 good
matches, position_finder.py:316). Per frame the results are identical to the single-frame entry points
(tests/test_batch_gpu.py); --batch 1 runs those instead, and their latency is reported beside the
headline as "single_frame".

Workloads (--workload; "auto" = c1 at N = 1, c3 at N > 1):
  c1  BASELINE configs[1]: the map's own descriptors (236 k rows for the 8192^2 orthomap), whole on every GPU. At N > 1
      frames are data-parallel and there is no collective.
  c3  BASELINE configs[3]: a 5 M-row map (--map-rows) = the orthomap's real descriptors followed by SURVEY 8(d)'s synthetic
      ones, SHARDED over the ranks by contiguous index ranges (image_matching_amd/sharded.py, match_dp): every rank takes
      its own batch, the batches' descriptor rows are all-gathered as bytes, every rank matches all of them against its
      shard (one MFMA launch), an all_to_all returns the per-shard top-2 to the batch's owner, merge_top2 picks the
      winners. Per-GPU work per step is constant in N: "scaling": "weak"; value = N * batch * steps / wall.
  c5  BASELINE configs[4]: 32 frames per step against a 20 M-row map sharded the same way, the frames REPLICATED (every
      rank extracts the same batch: north_star's pattern, match_replicated): per-shard top-2 -> all_gather -> merge.
      Total work per step is fixed: "scaling": "strong"; value = batch * steps / wall.
wall = max over ranks between two barriers. At N > 1 the line also carries the per-rank kNN roofline fractions, the
exchange time per step, and the world size RCCL reports.

Launch: under torchrun (WORLD_SIZE set) every process is one rank. Without it, `python bench.py --gpus N` with N > 1 starts
the N ranks itself as child processes -- before anything in this process has touched torch or the GPU -- relays rank 0's
line and exits with the first non-zero child status.

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
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA (spec): BASELINE.json's kNN target is quoted on it
MFMA_I8_PEAK_TF = 5000.0     # dense int8 MFMA: twice the bf16 multiply-adds per clock (MI355X_MICROARCH.md, matrix cores)
PYRAMID_BYTES_PER_INPUT_PIXEL = 475.7
# What the scale-space kernels actually move: per octave pixel 5 blurs x (4 B read + 4 B write) + the scan's
# six layer reads (24 B) = 64 B, octave pixels sum to (4/3) * 4 per input pixel; the base blur reads 1 B per input pixel
# and writes 4 B per doubled pixel (16 B per input pixel): 64 * 16 / 3 + 17 = 358.3 B per input pixel.
PYRAMID_BYTES_MOVED_PER_INPUT_PIXEL = 64.0 * 16.0 / 3.0 + 17.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(frames, map_desc_sample, n_map, map_xy_sample, ratio, threads):
    """The same chain through the CPU oracle on this box's host cores, frame after frame: up to 16 frames, bounded at about
    25 s of wall time (`threads` OpenMP threads). The matcher is oracle.knn2_fast -- f32 rows, SIMD squared distances,
    cache-tiled, square roots of the winners only: the shape of cv2.BFMatcher's own computation (oracle/knn_fast.c) --
    and is checked against the scalar checker oracle.knn2 on the first frame, outside the timing. The kNN leg runs against
    the whole map when it holds at most 300 000 descriptors, else against a slice and is scaled linearly (a dense scan)."""
    import oracle
    oracle.lib()
    first = oracle.sift(oracle.clahe(frames[0]))          # first touch: page in the library, spin up the OpenMP pool
    map_f32 = map_desc_sample.astype(np.float32)          # what cv2 holds: CV_32F rows
    chk_i, chk_d = oracle.knn2(first.desc.astype(np.uint8), map_desc_sample, threads=threads)
    fast_i, fast_d = oracle.knn2_fast(first.desc, map_f32, threads=threads)
    if not (np.array_equal(chk_i, fast_i) and np.array_equal(chk_d, fast_d)):
        raise RuntimeError("cpu_baseline: oracle.knn2_fast disagrees with the checker oracle.knn2")
    t_ext = t_knn_s = t_h = 0.0
    pairs = 0
    done = 0
    kept = []
    t_start = time.perf_counter()
    for frame in frames:
        t0 = time.perf_counter()
        res = oracle.sift(oracle.clahe(frame))
        t1 = time.perf_counter()
        idx, dist = oracle.knn2_fast(res.desc, map_f32, threads=threads)
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
        if time.perf_counter() - t_start > 25.0:
            break
    scale = n_map / max(len(map_desc_sample), 1)
    total = t_ext + t_knn_s * scale + t_h
    gdist = pairs / t_knn_s / 1e9
    return {
        "value": done / total, "unit": "frames/s", "cores": threads, "kind": "port",
        "sample": "%d frames 640x480, %.1f s of wall time on %d threads: extract %.3f s/frame (OpenMP) + kNN vs %d map "
                  "descriptors %.3f s/frame%s (SIMD, cache-tiled, f32 rows as cv2 holds them: %.2f Gdist/s = %.3f per core; "
                  "checked against the scalar checker on the first frame) + ratio and homography %.4f s/frame; restatement of "
                  "OpenCV, not OpenCV"
                  % (done, t_ext + t_knn_s + t_h, threads, t_ext / done, len(map_desc_sample), t_knn_s / done,
                     "" if scale == 1 else " (scaled x%.1f to %d)" % (scale, n_map), gdist, gdist / threads, t_h / done),
        "knn_gdist_per_s": gdist, "knn_gdist_per_s_per_core": gdist / threads,
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
            idx, dist = ctx.knn2_fset_akaze(ctx.akaze_last()[0], cset)     # the frame's descriptors where the extraction left them
            keep = dist[:, 0].astype(np.float64) < ratio * dist[:, 1].astype(np.float64)
            q = np.nonzero(keep)[0]
            H = None
            if len(q) > 10:
                H, _ = ctx.homography(kp[q, :2], ckp[idx[q, 0], :2], 0.1, seed=i)
            return len(kp), H is not None
        for i, f in enumerate(frames[:2]):
            one(f, i)
        # as for single_frame: the collector off (a generational collection over this process's objects is tens of
        # milliseconds) and the library's stage events off while timing
        import gc
        gc.collect()
        gc.disable()
        ctx.set_profiling(False)
        try:
            t0 = time.perf_counter()
            reps = 5
            for r in range(reps):
                for i, f in enumerate(frames):
                    n, ok = one(f, i)
                    n_kp += n
                    found += ok
            ms = (time.perf_counter() - t0) / (reps * len(frames)) * 1e3
        finally:
            ctx.set_profiling(True)
            gc.enable()
        return {"detector": "AKAZE(DESCRIPTOR_KAZE, 0.0005, 4, 5, PM_G2), the call of match_finder.py:210-211",
                "ms_per_frame": ms, "frames_per_s": 1e3 / ms, "keypoints_per_frame": n_kp / (reps * len(frames)),
                "crop": "2048x2048, %d keypoints, extracted in %.1f ms" % (len(ckp), t_crop * 1e3),
                "homographies_found": "%d/%d" % (found, reps * len(frames)),
                "note": "single-frame calls (imatch_akaze_extract with host pointers in and out, imatch_knn2_fset_akaze: the frame's "
                        "descriptors matched where the extraction left them against the resident crop set, imatch_homography), "
                        "measured after the timed region; not part of value"}
    except Exception as e:      # must not take the headline down with it
        return {"error": repr(e)[:300]}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["auto", "c1", "c3", "c5"], default="auto",
                    help="c1: configs[1], the map's own descriptors whole on every GPU; c3: configs[3], a 5 M-row map sharded "
                         "over the ranks, frames data-parallel; c5: configs[4], 32 replicated frames vs a 20 M-row sharded map; "
                         "auto = c1 at N = 1, c3 at N > 1")
    ap.add_argument("--map-side", type=int, default=8192)
    ap.add_argument("--map-rows", type=int, default=0, help="total descriptor rows of the c3 / c5 map (default 5 M / 20 M)")
    ap.add_argument("--frames", type=int, default=128, help="distinct synthetic frames per rank, cycled")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=0,
                    help="frames per step and GPU, taken through the batched entry points (one launch per stage for "
                         "the whole batch, one kNN launch over all frames' keypoints); 1 = the single-frame calls; "
                         "default 256 (c3: 64, c5: 32); at most 256")
    ap.add_argument("--streams", type=int, default=1,
                    help="frames in flight per GPU: each has its own context (stream + arena) and host thread; a step is "
                         "then one frame per stream")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process -- which has not
    imported torch or touched the GPU, and never will -- relay rank 0's stdout, exit with the first non-zero status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while any(pr.poll() is None for pr in procs):
        failed = [pr.returncode for pr in procs if pr.poll() not in (None, 0)]
        if failed:   # a rank died: the others would sit in a collective until RCCL's own timeout -- end exactly these children
            rc = failed[0]
            for pr in procs:
                if pr.poll() is None:
                    pr.terminate()
            t_end = time.time() + 10
            for pr in procs:
                try:
                    pr.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    pr.kill()
                    pr.wait()
            break
        time.sleep(0.2)
    for pr in procs:
        rc = rc or (pr.returncode or 0)
    reader.join(timeout=5)
    if got:
        sys.stdout.write(got[0].decode())
        sys.stdout.flush()
    return rc


def filler_xy(idx, side):
    """Map position of a synthetic (keypoint-less) descriptor row: a fixed pseudo-random pixel, so that a false match with
    such a row is an outlier for the homography like any other."""
    idx = np.asarray(idx, dtype=np.int64)
    return np.stack([(idx * 2654435761 % 1000003) % side, (idx * 40503 % 999983) % side], axis=1).astype(np.float32)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    if os.environ.get("IMATCH_BENCH_HANG_DUMP"):      # debugging aid: every thread's stack on stderr after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["IMATCH_BENCH_HANG_DUMP"]), exit=True)
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
    workload = args.workload if args.workload != "auto" else ("c1" if world == 1 else "c3")
    replicated_frames = workload == "c5"
    B = args.batch if args.batch > 0 else (32 if workload == "c5" else 256 if workload == "c1" else 64)

    if os.environ.get("IMATCH_BENCH_STACKS"):      # development: every thread's Python stack on stderr after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["IMATCH_BENCH_STACKS"]), exit=False)
    import image_matching_amd as im
    from image_matching_amd import sharded, synth
    ctx = im.Context(local_rank)
    ctx.set_profiling(True)
    ratio = 0.7
    params = im.SiftParams()

    # ---- the orthomap: generated and extracted once; its descriptors are the map (c1) or the map's real rows (c3, c5)
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
    n_real = len(f_map)
    map_kp = f_map.download(descriptors=False)[0]
    map_xy = np.ascontiguousarray(map_kp[:, :2])
    log("[rank %d] map: %d descriptors, extract %.2fs (first) %.2fs (warm), scale-space stage %.2f ms"
        % (rank, n_real, t_map_extract, t_map_extract2, pyr_ms_map))

    def build_shard(lo_, hi_):
        """Rows [lo_, hi_) of the c3 / c5 descriptor map as a resident set: rows [0, n_real) are the orthomap's own
        descriptors, the rest SURVEY 8(d)'s synthetic ones (any row range is reproducible on its own)."""
        shard_u8 = torch.empty((hi_ - lo_, 128), dtype=torch.uint8, device=dev)
        if lo_ < n_real:
            real_u8 = torch.empty((n_real, 128), dtype=torch.uint8, device=dev)
            f_map.desc_u8_dev(real_u8.data_ptr())
            ctx.synchronize()
            shard_u8[:min(hi_, n_real) - lo_] = real_u8[lo_:min(hi_, n_real)]
            del real_u8
        if hi_ > n_real:
            a = max(lo_, n_real)
            shard_u8[a - lo_:] = synth.synth_descriptors_dev(a - n_real, hi_ - n_real, dev, seed=7)
        torch.cuda.synchronize(dev)
        f = ctx.features_from_device_u8(shard_u8.data_ptr(), hi_ - lo_)
        ctx.synchronize()
        del shard_u8
        return f

    # ---- the descriptor map of the workload, resident in HBM
    if workload == "c1":
        n_map, lo, hi = n_real, 0, n_real
        f_local = f_map
        matcher = sharded.ShardedMatcher(ctx, f_map, 0, dev, local_only=True)
    else:
        n_map = args.map_rows if args.map_rows > 0 else (5_000_000 if workload == "c3" else 20_000_000)
        n_map = max(n_map, n_real)
        lo, hi = sharded.partition(n_map, world)[rank]
        t0 = time.time()
        f_local = build_shard(lo, hi)
        matcher = sharded.ShardedMatcher(ctx, f_local, lo, dev)
        log("[rank %d] descriptor map: %d rows (%d real + synthetic), shard rows %d..%d resident in %.1fs"
            % (rank, n_map, n_real, lo, hi, time.time() - t0))
    n_local = hi - lo

    def xy_of(gt):
        gt = np.asarray(gt, dtype=np.int64)
        if n_map == n_real:
            return map_xy[gt]
        out = filler_xy(gt, args.map_side)
        real = gt < n_real
        out[real] = map_xy[gt[real]]
        return out

    # ---- frames: each rank has its own set (c5: every rank the same), resident in HBM before the timed region
    frame_seed0 = 0 if replicated_frames else 100 * rank
    frames_t = [synth.synth_frame(map_img, frame_seed0 + i, size=(480, 640)) for i in range(args.frames)]
    frames_h = [f for f, _ in frames_t]
    truth_H = [h for _, h in frames_t]
    frames_d = [torch.from_numpy(f).to(dev) for f in frames_h]
    q_cap = 4096                     # rows per frame in the exchange / result buffers (frames here have ~3000 keypoints)

    stats = {"kp": [], "good": [], "found": 0, "knn_ms": [], "pyr_ms": [], "q_launch": []}

    # extra frames in flight: one context (stream, arena, workspace) and one matcher per stream
    n_streams = max(1, args.streams)
    if world > 1 and n_streams > 1:
        log("note: --streams applies to single-GPU runs (the collectives are issued from one thread); using 1")
        n_streams = 1
    ctxs = [ctx] + [im.Context(local_rank) for _ in range(n_streams - 1)]
    for c in ctxs[1:]:
        c.set_profiling(True)
    matchers = [matcher] + [sharded.ShardedMatcher(c, f_local, lo, dev, local_only=workload == "c1") for c in ctxs[1:]]
    import threading
    stats_lock = threading.Lock()

    def match(mt, f, nq, cap):
        if replicated_frames:
            return mt.match_replicated(f, nq)
        return mt.match_dp(f, nq, cap)

    # a frame alone on one GPU against the real map: the per-frame tail is one library call (IMATCH_BENCH_ONE_CALL=0: the
    # separate calls, for A/B runs)
    one_call = (world == 1 and workload == "c1" and n_map == n_real and n_streams == 1
                and os.environ.get("IMATCH_BENCH_ONE_CALL", "1") != "0")

    def step(i, record=False, lane=0):
        cx, mt = ctxs[lane], matchers[lane]
        fr = frames_d[(i * n_streams + lane) % len(frames_d)]
        f = cx.sift_extract_dev(fr.data_ptr(), 480, 640, 640, params)
        nq = len(f)
        pyr_ms = cx.sift_pyramid_ms() if record else 0.0
        if nq > q_cap:
            raise RuntimeError('frame has %d keypoints, exchange buffers hold %d' % (nq, q_cap))
        if one_call:
            # the whole tail in one call of the library (imatch_match_localize): same results as the calls below
            gq, gt, gd, H, _ = cx.match_localize(f, f_local, ratio, 10, 0.1, seed=i)
            knn_ms = cx.knn2_kernel_ms() if record else 0.0
        else:
            out4 = match(mt, f, nq, q_cap)
            gq, gt, gd = cx.ratio_dev(out4.data_ptr(), nq, ratio)
            knn_ms = cx.knn2_kernel_ms() if record else 0.0
            H = None
            if len(gq) > 10:                                   # position_finder.py:316
                kp = f.download(descriptors=False)[0]
                H, _ = cx.homography(kp[gq, :2], xy_of(gt), 0.1, seed=i)
        f.free()
        if record:
            with stats_lock:
                stats["pyr_ms"].append(pyr_ms)
                stats["knn_ms"].append(knn_ms)
                stats["kp"].append(nq)
                stats["q_launch"].append(sum(mt.last_counts) if (mt.world > 1 and not replicated_frames) else nq)
                stats["good"].append(len(gq))
                stats["found"] += H is not None
        return H

    # batched steps: B frames resident as one [B,480,640] tensor per step
    bq_cap = ((B * 3200 + 1023) // 1024) * 1024      # rows per batch in the exchange / result buffers (frames: ~2940 each)
    if B > 1:
        n_sets = max(2, (len(frames_h) + B - 1) // B)
        batches_d = [torch.from_numpy(np.stack([frames_h[(j * B + k) % len(frames_h)] for k in range(B)])).to(dev)
                     for j in range(n_sets)]

    # The tail of a batched step -- gathering the matched keypoints, the hypothesis scoring launch and the host refits --
    # needs nothing from the extraction arena, so it runs on a second context (own stream) from a worker thread while the
    # main thread already extracts the next batch: the GPU no longer idles through the tail's host work (about 1 ms of a
    # 13.8 ms step). One tail in flight at most; every tail has finished before a timed region ends (drain_tail).
    pipeline = B > 1 and n_streams == 1 and os.environ.get("IMATCH_BENCH_PIPELINE", "1") != "0"
    tail_ctx = im.Context(local_rank) if pipeline else None
    if tail_ctx is not None:
        tail_ctx.set_profiling(True)
    tail_err = []
    import queue
    tail_q = queue.Queue()

    def tail_worker():          # one long-lived thread: starting a thread per step cost ~0.1 ms of GPU idling per step
        while True:
            job = tail_q.get()
            try:
                if job is None:
                    return
                batch_tail(*job)
            except Exception as e:      # surfaced by the next drain_tail
                tail_err.append(e)
            finally:
                tail_q.task_done()

    def drain_tail():
        if pipeline:
            tail_q.join()
        if tail_err:
            raise tail_err[0]

    def batch_tail(cx, f, offs, gq, gt, goff, i, record, pyr_ms, knn_ms, q_launch_now, nq):
        ng = np.diff(goff)
        use = ng > 10                                      # position_finder.py:316
        src = f.keypoints_xy(np.repeat(offs[:-1], ng) + gq)
        src_off = goff
        gt_u = gt
        if not use.all():                                  # frames below the gate take no part
            keep = np.repeat(use, ng)
            src, gt_u = src[keep], gt[keep]
            src_off = np.concatenate([[0], np.cumsum(np.where(use, ng, 0))])
        H, _, found = cx.homography_batch(src, xy_of(gt_u), src_off, 0.1, seeds=np.arange(B) + i * B)
        f.free()
        if record:
            with stats_lock:
                stats["pyr_ms"].append(pyr_ms)
                stats["knn_ms"].append(knn_ms)
                stats["kp"].append(nq)
                stats["q_launch"].append(q_launch_now)
                stats["good"].append(float(ng.mean()))
                stats["found"] += int(found.sum())
        return H

    if pipeline:
        threading.Thread(target=tail_worker, daemon=True).start()

    def step_batch(i, record=False, lane=0):
        cx, mt = ctxs[lane], matchers[lane]
        fr = batches_d[(i * n_streams + lane) % len(batches_d)]
        f, offs = cx.sift_extract_batch_dev(fr.data_ptr(), B, 480 * 640, 480, 640, 640, params)
        nq = len(f)
        pyr_ms = cx.sift_pyramid_ms() if record else 0.0
        if nq > bq_cap:
            raise RuntimeError('batch has %d keypoints, buffers hold %d' % (nq, bq_cap))
        if os.environ.get("IMATCH_BENCH_TRACE"):
            log("[rank %d] step %d: %d query rows" % (rank, i, nq))
        out4 = match(mt, f, nq, bq_cap)
        gq, gt, gd, goff = cx.ratio_batch_dev(out4.data_ptr(), offs, ratio)
        knn_ms = cx.knn2_kernel_ms() if record else 0.0
        q_launch_now = sum(mt.last_counts) if (mt.world > 1 and not replicated_frames) else nq
        if not pipeline:
            return batch_tail(cx, f, offs, gq, gt, goff, i, record, pyr_ms, knn_ms, q_launch_now, nq)
        drain_tail()
        tail_q.put((tail_ctx, f, offs, gq, gt, goff, i, record, pyr_ms, knn_ms, q_launch_now, nq))
        return None

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
        drain_tail()
        for c in ctxs + ([tail_ctx] if tail_ctx is not None else []):
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
        sharded.all_reduce(tw, dist.ReduceOp.MAX)
        wall = float(tw.item())

    # ---- after the timed region -------------------------------------------------------------------------------------
    # the library's own stage brackets of the LAST step (HIP events on the streams the stages ran on)
    stages_ms = dict(ctx.stage_times())
    if tail_ctx is not None:
        stages_ms.update({k: v for k, v in tail_ctx.stage_times().items() if k == "homography_score"})
    knn_ms = float(np.mean(stats["knn_ms"]))
    q_launch = float(np.mean(stats["q_launch"]))
    achieved_tf = 256.0 * q_launch * n_local / (knn_ms * 1e-3) / 1e12     # 256 flop per (query, train) pair
    multi = None
    if world > 1:
        # every rank's kNN launch (fraction of the int8 MFMA peak), and the exchange alone: the step's collectives on
        # buffers of the step's size, bracketed by events on the context's stream
        mine = torch.tensor([achieved_tf / MFMA_I8_PEAK_TF, knn_ms, q_launch, float(n_local)], dtype=torch.float64, device=dev)
        allr = torch.empty((world, 4), dtype=torch.float64, device=dev)
        sharded.all_gather(allr.view(-1), mine)
        allr = allr.cpu().numpy()
        ex_us = None
        if workload != "c1":
            nq_x = int(np.mean(stats["kp"]))
            cap_x = bq_cap if B > 1 else q_cap
            st = torch.cuda.ExternalStream(ctx.stream, device=dev)
            reps = 20
            with torch.cuda.stream(st):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if replicated_frames:
                    a = torch.zeros((nq_x, 4), dtype=torch.int32, device=dev)
                    b = torch.zeros((world * nq_x, 4), dtype=torch.int32, device=dev)
                    sharded.all_gather(b, a)
                    e0.record()
                    for _ in range(reps):
                        sharded.all_gather(b, a)
                    e1.record()
                else:
                    a = torch.zeros((cap_x, 128), dtype=torch.uint8, device=dev)
                    b = torch.zeros((world * cap_x, 128), dtype=torch.uint8, device=dev)
                    c = torch.zeros((world * cap_x, 4), dtype=torch.int32, device=dev)
                    d = torch.zeros((world * cap_x, 4), dtype=torch.int32, device=dev)
                    sharded.all_gather(b, a)
                    sharded.all_to_all(d, c)
                    e0.record()
                    for _ in range(reps):
                        sharded.all_gather(b, a)
                        sharded.all_to_all(d, c)
                    e1.record()
            e1.synchronize()
            ex = torch.tensor([e0.elapsed_time(e1) / reps * 1e3], dtype=torch.float64, device=dev)
            sharded.all_reduce(ex, dist.ReduceOp.MAX)
            ex_us = float(ex.item())
        # IMATCH_BENCH_VERIFY=1 (tests, rehearsals at small --map-rows): this rank also builds the WHOLE map and checks that
        # the exchange path gave one batch exactly what a single launch over all rows gives
        same_as_whole = None
        if workload != "c1" and os.environ.get("IMATCH_BENCH_VERIFY") == "1":
            whole = torch.empty((n_map, 128), dtype=torch.uint8, device=dev)
            f_map.desc_u8_dev(whole.data_ptr())
            ctx.synchronize()
            if n_map > n_real:
                whole[n_real:] = synth.synth_descriptors_dev(0, n_map - n_real, dev, seed=7)
            torch.cuda.synchronize(dev)
            f_whole = ctx.features_from_device_u8(whole.data_ptr(), n_map)
            if B > 1:
                f_chk, _ = ctx.sift_extract_batch_dev(batches_d[0].data_ptr(), B, 480 * 640, 480, 640, 640, params)
            else:
                f_chk = ctx.sift_extract_dev(frames_d[0].data_ptr(), 480, 640, 640, params)
            a = match(matcher, f_chk, len(f_chk), bq_cap if B > 1 else q_cap)
            ctx.synchronize()
            a = a.clone()
            b = torch.empty((len(f_chk), 4), dtype=torch.int32, device=dev)
            ctx.knn2_dev(f_chk, f_whole, b.data_ptr())
            ctx.synchronize()
            same = torch.tensor([int(torch.equal(a, b))], dtype=torch.int64, device=dev)
            sharded.all_reduce(same, dist.ReduceOp.MIN)
            same_as_whole = bool(same.item())
            f_chk.free()
            f_whole.free()
            del whole
        multi = {"identical_to_whole_map": same_as_whole, "per_rank_knn_frac": [float(v) for v in allr[:, 0]], "per_rank_knn_launch_ms": [float(v) for v in allr[:, 1]],
                 "per_rank_queries_per_launch": [float(v) for v in allr[:, 2]], "per_rank_shard_rows": [int(v) for v in allr[:, 3]],
                 "exchange_us_per_step": ex_us,
                 "exchange": None if workload == "c1" else
                             ("all_gather of per-shard top-2, 16 B per query and rank" if replicated_frames else
                              "all_gather of u8 query rows (128 B per row) + all_to_all of per-shard top-2 (16 B per query and rank)"),
                 "collective_backend": dist.get_backend(), "world_size_seen": dist.get_world_size()}

    # ---- the OTHER workload of the default series, measured in this run (after the timed region, not part of `value`).
    # The driver's series runs `--gpus 1` as c1 and `--gpus N > 1` as c3, so value(N) / value(1) of the headline compares two
    # workloads; with both points on every line each workload has its own N-series: c1 is data-parallel (every rank the
    # whole 236 k-row map, no collective), c3 the sharded 5 M-row map (match_dp).
    def side_point(wl2, steps2, warm2):
        nonlocal workload, replicated_frames, B, bq_cap, batches_d, matchers, n_map, stats
        saved = (workload, replicated_frames, B, bq_cap, batches_d, matchers, n_map, stats)
        f_side = None
        try:
            workload, replicated_frames = wl2, False
            B = args.batch if args.batch > 1 else (256 if wl2 == "c1" else 64)
            bq_cap = ((B * 3200 + 1023) // 1024) * 1024
            n_sets2 = max(2, (len(frames_h) + B - 1) // B)
            batches_d = [torch.from_numpy(np.stack([frames_h[(j * B + k) % len(frames_h)] for k in range(B)])).to(dev)
                         for j in range(n_sets2)]
            if wl2 == "c1":
                n_map = n_real
                mt = sharded.ShardedMatcher(ctx, f_map, 0, dev, local_only=True)
                rows_here = n_real
            else:
                n_map = max(args.map_rows if args.map_rows > 0 else 5_000_000, n_real)
                lo2, hi2 = sharded.partition(n_map, world)[rank]
                f_side = build_shard(lo2, hi2)
                mt = sharded.ShardedMatcher(ctx, f_side, lo2, dev)
                rows_here = hi2 - lo2
            matchers = [mt]
            stats = {"kp": [], "good": [], "found": 0, "knn_ms": [], "pyr_ms": [], "q_launch": []}
            run_steps(warm2, False)
            barrier()
            t_s = time.perf_counter()
            run_steps(steps2, True)
            barrier()
            w2 = time.perf_counter() - t_s
            if world > 1:
                tw2 = torch.tensor([w2], dtype=torch.float64, device=dev)
                sharded.all_reduce(tw2, dist.ReduceOp.MAX)
                w2 = float(tw2.item())
            k_ms = float(np.mean(stats["knn_ms"]))
            ql = float(np.mean(stats["q_launch"]))
            return {"workload_id": wl2, "value": world * B * steps2 / w2, "unit": "frames/s", "n_gpus": world,
                    "steps": steps2, "warmup": warm2, "ms_per_step": w2 / steps2 * 1e3, "frames_per_step_per_gpu": B,
                    "map_descriptors": n_map, "scaling": "weak", "knn_launch_ms": k_ms,
                    "knn_frac_of_int8_peak": 256.0 * ql * rows_here / (k_ms * 1e-3) / 1e12 / MFMA_I8_PEAK_TF,
                    "homographies_found": "%d/%d" % (stats["found"], steps2 * B),
                    "note": "measured in this run after the timed region, same step code and barriers as the headline; "
                            "not part of value"}
        finally:
            try:
                drain_tail()      # (a failure may have left a tail in flight: it reads B and the matchers)
            except Exception:
                pass
            workload, replicated_frames, B, bq_cap, batches_d, matchers, n_map, stats = saved
            if f_side is not None:
                ctx.synchronize()
                f_side.free()

    series = None
    if n_streams == 1 and B > 1 and workload in ("c1", "c3") and os.environ.get("IMATCH_BENCH_SERIES", "1") != "0":
        other = "c3" if workload == "c1" else "c1"
        try:
            series = side_point(other, 6 if other == "c3" else 8, 2)
        except Exception as e:      # must not take the headline down with it (every rank fails alike: sizes are the same)
            series = {"workload_id": other, "error": repr(e)[:300]}

    # single-frame latency, outside the timed region: the per-frame calls, one frame at a time
    # (the library's event brackets -- two events per stage, sixteen per frame -- are switched off for it: they are this
    # harness's instrument for roofline / stages_ms, cost a frame's chain 0.04 ms, and a caller of the per-frame entry
    # points does not have them on; the collector is still off, see above)
    single_ms = single_prof_ms = None
    if world == 1:
        def frames_alone(n):
            for i in range(5):
                step(i)
            blocks = []
            for b in range(5):      # the median of five blocks: a box's host side wanders by 5-10 % between blocks
                barrier()
                t1 = time.perf_counter()
                for i in range(n // 5):
                    step(b * (n // 5) + i)
                barrier()
                blocks.append((time.perf_counter() - t1) / (n // 5) * 1e3)
            return float(np.median(blocks))
        single_prof_ms = frames_alone(50)
        for c in ctxs:
            c.set_profiling(False)
        single_ms = frames_alone(150)
        for c in ctxs:
            c.set_profiling(True)

    gc.enable()
    out = None
    if rank == 0:
        frames_per_step = (1 if replicated_frames else world) * n_streams * B
        fps = frames_per_step * args.steps / wall
        q_mean = float(np.mean(stats["kp"]))
        pyr_bytes = PYRAMID_BYTES_PER_INPUT_PIXEL * args.map_side * args.map_side
        what = {"c1": "640x480 gray frame vs full %dx%d synthetic orthomap, %d map descriptors resident in HBM (configs[1])"
                      % (args.map_side, args.map_side, n_map),
                "c3": "640x480 gray frames vs a %d-row descriptor map (the %dx%d orthomap's %d real rows + SURVEY 8(d) synthetic "
                      "rows) sharded over %d GPU(s), RCCL top-2 merge (configs[3])" % (n_map, args.map_side, args.map_side, n_real, world),
                "c5": "%d replicated 640x480 frames per step vs a %d-row descriptor map sharded over %d GPU(s), all-gather "
                      "top-2 merge (configs[4])" % (B, n_map, world)}[workload]
        par = {"c1": "single GPU" if world == 1 else "map replicated on %d GPUs, frames data-parallel, no collective" % world,
               "c3": "map sharded x%d by contiguous row ranges, frames data-parallel (match_dp)" % world,
               "c5": "map sharded x%d by contiguous row ranges, frames replicated (match_replicated)" % world}[workload]
        out = {
            "metric": "frames/sec (SIFT+kNN+RANSAC) vs map",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if replicated_frames else "weak",
            "vs_baseline": None, "dtype": "i8", "data": "synthetic",
            "config": {"workload": what, "workload_id": workload,
                       "frame": "640x480 u8", "map_descriptors": n_map, "keypoints_per_frame": q_mean / B,
                       "frames_per_step_per_gpu": B * n_streams, "frames_per_step": frames_per_step,
                       "entry_points": "batched (imatch_sift_extract_batch_dev, one kNN launch, imatch_ratio_batch_dev, "
                                       "imatch_homography_batch)" if B > 1 else "single-frame",
                       "good_matches_per_frame": float(np.mean(stats["good"])),
                       "homographies_found": "%d/%d" % (stats["found"], args.steps * B * n_streams), "ratio": ratio,
                       "frames_in_flight_per_gpu": n_streams, "parallelism": par,
                       "step_tail": ("match gathering + homography of batch i on a second context / worker thread while batch "
                                     "i+1 is being extracted" if pipeline else "in line")},
            "knn_gdist_per_s": (world if multi else 1) * q_launch * float(n_local) / (knn_ms * 1e-3) / 1e9,
            "roofline": {"kernel": "knn2_kernel8 (int8 MFMA 16x16x64, int32 accumulate, fused top-2)", "bound": "mfma",
                         "achieved": achieved_tf, "peak": MFMA_I8_PEAK_TF, "unit": "TFLOP/s", "frac": achieved_tf / MFMA_I8_PEAK_TF,
                         "frac_of_bf16_peak": achieved_tf / MFMA_BF16_PEAK_TF,
                         "traffic": None, "launch_ms": knn_ms,
                         "per_launch": "%d queries x %d train rows x 256 flop%s" % (int(q_launch), n_local,
                                                                                   " (rank 0's launch)" if world > 1 else "")},
            "roofline_pyramid": {"kernel": "scale-space stage of the %dx%d map extraction (upsample, blur+DoG, "
                                           "decimate, extrema scan)" % (args.map_side, args.map_side),
                                 "bound": "hbm", "achieved": pyr_bytes / (pyr_ms_map * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": pyr_bytes / (pyr_ms_map * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                 "stage_ms": pyr_ms_map, "frame_stage_ms": float(np.mean(stats["pyr_ms"])),
                                 "bytes_moved": PYRAMID_BYTES_MOVED_PER_INPUT_PIXEL * args.map_side * args.map_side,
                                 "frac_moved": PYRAMID_BYTES_MOVED_PER_INPUT_PIXEL * args.map_side * args.map_side
                                               / (pyr_ms_map * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "achieved / frac: SURVEY 8(d)'s algorithmic 475.7 B per input pixel (counts DoG planes "
                                         "written and re-read, which the kernels never store); frac_moved: the bytes the "
                                         "kernels move (each blur reads and writes a layer once, the scan reads six layers, "
                                         "the base blur reads the u8 image; PMC-confirmed, profiles/)"},
        }
        out["stages_ms"] = dict(stages_ms, note="event brackets of the last timed step's stages inside the library "
                                                "(imatch_stage_times), per step of %d frames" % (B * n_streams))
        if multi:
            out["multi_gpu"] = multi
        # both workloads of the default series at THIS N: value(N) / value(1) exists per workload without a kept file
        head = {"workload_id": workload, "value": fps, "unit": "frames/s", "n_gpus": world, "ms_per_step": wall / args.steps * 1e3,
                "frames_per_step_per_gpu": B * n_streams, "map_descriptors": n_map, "note": "the headline (value)"}
        out["series"] = {workload: head}
        if series is not None:
            out["series"][series["workload_id"]] = series
        # HBM-side traffic of the dominant kernel: NOT measured in this run -- it comes from separate rocprofv3 --pmc passes
        # over this same command, kept under profiles/, and is attached only when that measurement was taken on the
        # same launch shape
        for pmc_file in ("r3_pmc_knn.json", "r2_pmc_knn.json", "r1_pmc_knn.json"):
            try:
                with open(os.path.join(ROOT, "profiles", pmc_file)) as fh:
                    pmc = json.load(fh)
                if workload == "c1" and world == 1 and B == pmc.get("batch", 32) and \
                        abs(pmc.get("n_queries", q_mean) - q_mean) < 0.05 * q_mean:
                    out["roofline"]["traffic"] = pmc["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_note"] = ("from profiles/%s, not measured in this run: bytes per launch, "
                                                       "2 x FETCH_SIZE + WRITE_SIZE, " % pmc_file) + pmc["source"]
                    break
            except (OSError, ValueError, KeyError):
                pass
        try:      # the same for the scale-space stage of the map extraction
            with open(os.path.join(ROOT, "profiles", "r3_pmc_pyramid.json")) as fh:
                pp = json.load(fh)
            if args.map_side == 8192:
                out["roofline_pyramid"]["traffic"] = pp["traffic_bytes_per_extraction"]
                out["roofline_pyramid"]["traffic_note"] = "from profiles/r3_pmc_pyramid.json, not measured in this run: " + pp["source"]
        except (OSError, ValueError, KeyError):
            pass
        if single_ms is not None:
            out["single_frame"] = {"ms_per_frame": single_ms, "frames_per_s": 1e3 / single_ms,
                                   "ms_per_frame_with_stage_events": single_prof_ms,
                                   "calls": ("imatch_sift_extract_dev + imatch_match_localize" if one_call else
                                             "imatch_sift_extract_dev + imatch_knn2_dev + imatch_ratio_dev + imatch_homography"),
                                   "note": "one frame at a time through the single-frame entry points, measured after "
                                           "the timed region (median of five blocks of 30 frames, the library's stage-event brackets off; with them "
                                           "on: ms_per_frame_with_stage_events); not part of value"}
        if world == 1 and workload == "c1":
            out["akaze_path"] = akaze_extra(ctx, map_img, ratio)
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            if n_map == n_real and n_map <= 300_000:
                d_s = f_local.download(keypoints=False)[1].astype(np.uint8)
                xy_s = map_xy
            else:   # the oracle's kNN leg runs against the first 100 000 rows and is scaled to the map (a dense scan)
                d_s = f_map.download(keypoints=False)[1][:100_000].astype(np.uint8)
                xy_s = map_xy[:100_000]
            out["cpu_baseline"], kept = cpu_baseline(frames_h[:16], d_s, n_map, xy_s, ratio, threads)
            if len(d_s) == n_map:    # the oracle matched against the whole map: its lists are comparable as they are
                out["parity"] = parity_block(ctx, params, frames_h, truth_H, kept, f_local, map_xy, ratio)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
