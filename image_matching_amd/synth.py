"""Seeded synthetic inputs of the shapes SURVEY.md 8(d) names (the reference ships no images,
maps or descriptor sets: .MISSING_LARGE_BLOBS). numpy only; used by tests and bench.py."""
import numpy as np


def sift_normalise_u8(v):
    """SIFT descriptor normalisation (Spec S10): L2 -> clamp 0.2 -> x512 -> round-half-even -> clip 0..255."""
    v = np.asarray(v, dtype=np.float32)
    nrm = np.sqrt((v * v).sum(axis=1, keepdims=True))
    v = np.minimum(v, 0.2 * nrm)
    nrm2 = 512.0 / np.maximum(np.sqrt((v * v).sum(axis=1, keepdims=True)), np.finfo(np.float32).eps)
    return np.clip(np.rint(v * nrm2), 0, 255).astype(np.uint8)


def synth_descriptors(n, seed=7, chunk=1 << 18):
    """u8 [n,128] SIFT-distributed descriptors: Gamma(0.6,1)^128, SIFT-normalised."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 128), np.uint8)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        out[s:e] = sift_normalise_u8(rng.gamma(0.6, 1.0, size=(e - s, 128)))
    return out


def synth_queries(train, nq, seed=8, noise=6, dup_frac=0.001):
    """Queries = random train rows + integer noise U{-noise..noise}. For a fraction dup_frac the
    query is planted exactly at its source row AND at a higher train row (an exact tie at distance
    0: the lower index must rank first). `train` is modified in place. Returns (queries, src_rows, dup_rows)
    where dup_rows[i] = -1 when query i has no planted tie."""
    rng = np.random.default_rng(seed)
    n = train.shape[0]
    src = rng.integers(0, n, size=nq)
    q = train[src].astype(np.int16) + rng.integers(-noise, noise + 1, size=(nq, 128), dtype=np.int16)
    q = np.clip(q, 0, 255).astype(np.uint8)
    dup = np.full(nq, -1, np.int64)
    ndup = int(round(nq * dup_frac))
    if n >= 4 and ndup > 0:
        who = rng.choice(nq, size=ndup, replace=False)
        taken = set(int(s) for s in src)
        for i in who:
            lo = int(src[i])
            if lo >= n - 1:
                continue
            hi = int(rng.integers(lo + 1, n))
            if hi in taken:
                continue
            taken.add(hi)
            train[lo] = q[i]
            train[hi] = q[i]
            dup[i] = hi
    return q, src, dup


# ---- the same descriptor statistics, generated on the device (BASELINE configs[3]/[4]: 5 M / 20 M rows) ---------------
DEV_CHUNK = 1 << 18


def synth_descriptors_dev(lo, hi, device, seed=7):
    """Rows [lo, hi) of the SURVEY 8(d) descriptor map (Gamma(0.6, 1)^128, SIFT-normalised, u8) as a torch tensor on
    `device`. Chunk c = rows [c*2^18, (c+1)*2^18) has its own generator seed, so any row range of the map -- a rank's
    shard -- is generated without the rows before it and is the same on every rank. (numpy's Gamma sampler takes ~25 ns
    per value: 5 M rows would cost a minute of host time per rank; the statistics are the same.)"""
    import torch
    out = torch.empty((hi - lo, 128), dtype=torch.uint8, device=device)
    alpha = torch.full((DEV_CHUNK, 128), 0.6, dtype=torch.float32, device=device)
    for c in range(lo // DEV_CHUNK, (max(hi, lo + 1) - 1) // DEV_CHUNK + 1):
        g = torch.Generator(device=device).manual_seed(int(seed) * 1000003 + c)
        v = torch._standard_gamma(alpha, generator=g)
        nrm = v.square().sum(dim=1, keepdim=True).sqrt()
        v = torch.minimum(v, 0.2 * nrm)
        nrm2 = 512.0 / v.square().sum(dim=1, keepdim=True).sqrt().clamp_min(1.1920929e-07)
        u = (v * nrm2).round().clamp(0, 255).to(torch.uint8)       # torch.round: half to even
        a, b = max(lo, c * DEV_CHUNK), min(hi, (c + 1) * DEV_CHUNK)
        if b > a:
            out[a - lo:b - lo] = u[a - c * DEV_CHUNK:b - c * DEV_CHUNK]
    if out.is_cuda:
        torch.cuda.synchronize(out.device)   # made on torch's stream; its readers (a Context's own stream) come after this
    return out


def synth_queries_dev(train, nq, seed=8, noise=6, n_dup=0):
    """Device version of synth_queries: queries = random rows of `train` (a u8 [n,128] torch tensor) + integer noise
    U{-noise..noise}; the first n_dup queries are planted exactly at their source row AND at a higher row (exact ties at
    distance 0: the lower index must rank first). `train` is modified in place. -> (queries u8 [nq,128], src int64 [nq],
    dup int64 [nq], -1 where the query has no planted tie). Sources sit on even rows, duplicates on odd rows."""
    import torch
    n = train.shape[0]
    g = torch.Generator(device=train.device).manual_seed(int(seed))
    src = torch.randperm(max((n - 2) // 2, 1), device=train.device, generator=g)[:nq] * 2
    if src.numel() < nq:      # more queries than even rows: repeat
        src = src.repeat((nq + src.numel() - 1) // src.numel())[:nq]
    q = train[src].to(torch.int16) + torch.randint(-noise, noise + 1, (nq, 128), dtype=torch.int16, device=train.device,
                                                   generator=g)
    q = q.clamp(0, 255).to(torch.uint8)
    dup = torch.full((nq,), -1, dtype=torch.int64, device=train.device)
    n_dup = min(n_dup, nq)
    if n_dup > 0:
        room = ((n - 1 - src[:n_dup]) // 2).clamp_min(1)                                     # odd offsets that stay inside
        step = (torch.rand(n_dup, device=train.device, generator=g, dtype=torch.float64) * room).long().clamp_max(room - 1) * 2 + 1
        hi_rows = src[:n_dup] + step
        ok = hi_rows < n
        # distinct sources on even rows, duplicates on odd rows: the two sets cannot collide; drop repeated odd rows
        uniq = torch.ones_like(ok)
        srt, order = torch.sort(hi_rows)
        uniq[order[1:]] = srt[1:] != srt[:-1]
        ok &= uniq
        sel = torch.nonzero(ok).flatten()
        train[src[sel]] = q[sel]
        train[hi_rows[sel]] = q[sel]
        dup[sel] = hi_rows[sel]
    if q.is_cuda:
        torch.cuda.synchronize(q.device)   # as above: complete before a Context's stream reads q and train
    return q, src, dup


# ---- images (the reference's map.png and flight videos are not in its tree) -------
def _upsample_bilinear(a, side):
    n = a.shape[0]
    t = np.linspace(0, n - 1, side)
    i0 = np.minimum(t.astype(np.int64), n - 2)
    f = (t - i0).astype(np.float32)
    rows = a[i0] * (1 - f)[:, None] + a[i0 + 1] * f[:, None]
    return rows[:, i0] * (1 - f)[None, :] + rows[:, i0 + 1] * f[None, :]


def synth_map(side, seed=1234):
    """u8 [side,side] orthomap stand-in: 6 octaves of value noise (amplitude ~ 1/f), random filled
    rectangles and lines ("fields", "roads"), per-pixel Gaussian noise sigma 2."""
    rng = np.random.default_rng(seed)
    img = np.zeros((side, side), np.float32)
    amp = 1.0
    for o in range(6):
        n = 4 * (2 ** o) + 1
        img += amp * _upsample_bilinear(rng.random((n, n), dtype=np.float32), side)
        amp *= 0.5
    img = (img - img.min()) / (img.max() - img.min()) * 150 + 50
    nrect = int(2000 * (side / 2048) ** 2)
    for _ in range(nrect):
        w, h = rng.integers(4, 60, size=2)
        x, y = rng.integers(0, side - 4, size=2)
        v = rng.uniform(-45, 45)
        if rng.random() < 0.25:  # a thin "road"
            if rng.random() < 0.5:
                w, h = int(rng.integers(60, 300)), int(rng.integers(2, 5))
            else:
                w, h = int(rng.integers(2, 5)), int(rng.integers(60, 300))
        img[y:y + h, x:x + w] += v
    img += rng.normal(0, 2, size=img.shape).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp_bilinear(src, H, out_h, out_w):
    """dst(y,x) = src(H @ (x,y,1)) with bilinear sampling (H maps dst pixels to src pixels)."""
    ys, xs = np.mgrid[0:out_h, 0:out_w].astype(np.float64)
    den = H[2, 0] * xs + H[2, 1] * ys + H[2, 2]
    sx = (H[0, 0] * xs + H[0, 1] * ys + H[0, 2]) / den
    sy = (H[1, 0] * xs + H[1, 1] * ys + H[1, 2]) / den
    x0 = np.clip(np.floor(sx).astype(np.int64), 0, src.shape[1] - 2)
    y0 = np.clip(np.floor(sy).astype(np.int64), 0, src.shape[0] - 2)
    fx = np.clip(sx - x0, 0, 1)
    fy = np.clip(sy - y0, 0, 1)
    s = src.astype(np.float32)
    top = s[y0, x0] * (1 - fx) + s[y0, x0 + 1] * fx
    bot = s[y0 + 1, x0] * (1 - fx) + s[y0 + 1, x0 + 1] * fx
    return top * (1 - fy) + bot * fy


def synth_frame(map_img, i, size=(480, 640), scale_range=(0.8, 1.25)):
    """Frame i cut from the map by a random homography (rotation, scale, slight perspective), with
    gain/bias/noise. Returns (frame u8, H) where H maps FRAME pixel (x,y,1) to MAP pixel."""
    rng = np.random.default_rng(1000 + i)
    h, w = size
    side = map_img.shape[0]
    ang = rng.uniform(-np.pi, np.pi)
    s = rng.uniform(*scale_range)
    margin = int(0.75 * s * np.hypot(h, w)) + 8
    cx, cy = rng.uniform(margin, side - margin, size=2)
    c, sn = np.cos(ang) * s, np.sin(ang) * s
    A = np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1.0]])
    P = np.eye(3)
    P[2, 0], P[2, 1] = rng.uniform(-1e-4, 1e-4, size=2)
    T0 = np.array([[1, 0, -w / 2], [0, 1, -h / 2], [0, 0, 1.0]])
    T1 = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]])
    H = T1 @ A @ P @ T0
    f = warp_bilinear(map_img, H, h, w)
    f = f * rng.uniform(0.8, 1.2) + rng.uniform(-15, 15) + rng.normal(0, 3, size=f.shape)
    return np.clip(np.rint(f), 0, 255).astype(np.uint8), H / H[2, 2]


def synth_flight(map_img, n, size=(720, 1280), map_pixel_size=0.5, frame_pixel_size=0.375, fps=5.0, speed=20.0, seed=0):
    """A straight-ish flight over the map for the replay loop (BASELINE configs[2]: 1280x720 frames at the
    reference's ~5 Hz publishing cadence, photo_publisher.py:110,130-131). Returns a list of (frame u8, truth)
    with truth = dict(stamp, cx, cy: map pixel under the frame centre, yaw: frame rotation against the map,
    H: frame pixel -> map pixel)."""
    rng = np.random.default_rng(5000 + seed)
    h, w = size
    side_y, side_x = map_img.shape
    s = frame_pixel_size / map_pixel_size
    margin = 0.75 * s * np.hypot(h, w) * 2.2 + 16           # room for the search window around the frame
    step = speed / fps / map_pixel_size                      # map pixels per frame
    heading = rng.uniform(-np.pi, np.pi)
    cx = rng.uniform(margin, side_x - margin)
    cy = rng.uniform(margin, side_y - margin)
    yaw = rng.uniform(-0.4, 0.4)
    out = []
    for i in range(n):
        nx, ny = cx + step * np.cos(heading), cy + step * np.sin(heading)
        if not (margin <= nx <= side_x - margin and margin <= ny <= side_y - margin):
            heading += np.pi / 2 + rng.uniform(0, np.pi / 2)   # turn away from the border
            nx, ny = cx + step * np.cos(heading), cy + step * np.sin(heading)
            nx, ny = np.clip(nx, margin, side_x - margin), np.clip(ny, margin, side_y - margin)
        cx, cy = nx, ny
        heading += rng.normal(0, 0.02)
        yaw = float(np.clip(yaw + rng.normal(0, 0.01), -0.5, 0.5))
        c, sn = np.cos(yaw) * s, np.sin(yaw) * s
        H = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1.0]]) @ np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1.0]]) @ \
            np.array([[1, 0, -(w - 1) / 2], [0, 1, -(h - 1) / 2], [0, 0, 1.0]])
        f = warp_bilinear(map_img, H, h, w)
        f = f * rng.uniform(0.9, 1.1) + rng.uniform(-10, 10) + rng.normal(0, 3, size=f.shape)
        out.append((np.clip(np.rint(f), 0, 255).astype(np.uint8),
                    {"stamp": i / fps, "cx": float(cx), "cy": float(cy), "yaw": yaw, "H": H}))
    return out
