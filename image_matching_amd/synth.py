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
