"""Map descriptor database: the map's features extracted ONCE, kept on disk, loaded and
pinned in HBM -- what BASELINE.json's configs[1], [3] and [4] call "map descriptors resident in HBM".

The reference has only the in-memory form of this idea: in the no-GPS cold start it tiles the whole map
(match_finder.py:86-128), extracts every tile once and keeps the list `self.rois` (position_finder.py:248-262),
then matches each frame against each tile in turn (:265-272); while tracking it re-extracts a crop around the
last fix every 12th frame (:232-240,275-286). Here the tiles' features are concatenated into ONE set in
full-map pixel coordinates:

    desc   u8  [N,128]   SIFT descriptors (integers 0..255, the exact kNN operand)
    kp     f32 [N,7]     x, y (map pixels), size, angle, response, octave, class_id
    tiles  i32 [T,6]     core x0, y0, x1, y1 and the tile's row range [r0, r1) in desc/kp
    meta   json          pixel size, geo-referenced corners, SIFT parameters, tile geometry, format version

Tiles are independent units (the CLAHE + SIFT of find_kp_dp runs on each tile with a margin, as the reference
runs it on each crop), so a build shards over GPUs with no collective: rank g takes tiles g, g+W, ...; only
the row offsets need the per-tile counts (`merge`). A keypoint is kept by the tile whose CORE contains it,
so the margins add no duplicates. Rows are ordered tile by tile, keypoints inside a tile in the extraction's
order: the global row index is what kNN's lowest-index tie-break refers to, and contiguous row ranges are
what `sharded.partition` hands to the ranks.

On disk: a directory with desc.npy, kp.npy, tiles.npy (plain .npy, memory-mapped on load) and meta.json."""
import json
import os

import numpy as np

FORMAT_VERSION = 1


class MapDB:
    def __init__(self, desc, kp, tiles, meta):
        self.desc = desc
        self.kp = kp
        self.tiles = tiles
        self.meta = meta
        self._resident = {}

    def __len__(self):
        return int(self.desc.shape[0])

    @property
    def xy(self):
        return self.kp[:, :2]

    # ---- build ---------------------------------------------------------------------------------------------
    @staticmethod
    def tile_grid(h, w, tile):
        return [(x0, y0, min(x0 + tile, w), min(y0 + tile, h)) for y0 in range(0, h, tile) for x0 in range(0, w, tile)]

    @classmethod
    def build(cls, ctx, img, tile=4096, margin=256, params=None, pixel_size=None, corners=None, rank=0, world=1):
        """Extract the tiles rank, rank + world, ... of `img` (u8 [H,W]). With world == 1 the result is the whole
        database; otherwise a partial one (tile table complete, rows only for this rank's tiles) for `merge`."""
        from ._lib import SiftParams
        params = params if params is not None else SiftParams()
        h, w = img.shape
        grid = cls.tile_grid(h, w, tile)
        descs, kps, rows = [], [], []
        for t, (x0, y0, x1, y1) in enumerate(grid):
            if t % world != rank:
                rows.append(0)
                continue
            xa, ya, xb, yb = max(0, x0 - margin), max(0, y0 - margin), min(w, x1 + margin), min(h, y1 + margin)
            f = ctx.sift_extract(np.ascontiguousarray(img[ya:yb, xa:xb]), params)
            kp, desc = f.download()
            f.free()
            kp[:, 0] += xa
            kp[:, 1] += ya
            keep = (kp[:, 0] >= x0) & (kp[:, 0] < x1) & (kp[:, 1] >= y0) & (kp[:, 1] < y1)
            kps.append(kp[keep])
            descs.append(desc[keep].astype(np.uint8))
            rows.append(int(keep.sum()))
        meta = {"format": FORMAT_VERSION, "height": int(h), "width": int(w), "tile": int(tile), "margin": int(margin),
                "pixel_size": None if pixel_size is None else float(pixel_size),
                "corners": corners,   # [[pixel_y, pixel_x, lat, lon] x 4], the .@@@ file's rows (image_processing.py:57-69)
                "sift": {"n_octave_layers": int(params.n_octave_layers), "contrast_threshold": float(params.contrast_threshold),
                         "edge_threshold": float(params.edge_threshold), "sigma": float(params.sigma), "clahe": bool(params.clahe)},
                "rank": int(rank), "world": int(world)}
        tiles = np.zeros((len(grid), 6), np.int32)
        tiles[:, :4] = np.asarray(grid, np.int32).reshape(-1, 4)
        tiles[:, 5] = rows        # counts for now; merge() turns them into ranges
        db = cls(np.concatenate(descs) if descs else np.zeros((0, 128), np.uint8),
                 np.concatenate(kps) if kps else np.zeros((0, 7), np.float32), tiles, meta)
        return cls.merge([db]) if world == 1 else db

    @classmethod
    def merge(cls, parts):
        """Partial databases of one build (any order) -> the whole one: rows tile by tile, ranges by a prefix sum."""
        world = parts[0].meta["world"]
        by_rank = {p.meta["rank"]: p for p in parts}
        if sorted(by_rank) != list(range(world)):
            raise ValueError("merge needs the parts of ranks 0..%d, got %s" % (world - 1, sorted(by_rank)))
        n_tiles = len(parts[0].tiles)
        counts = np.zeros(n_tiles, np.int64)
        for p in parts:
            counts += p.tiles[:, 5]
        start = np.concatenate([[0], np.cumsum(counts)])
        desc = np.zeros((int(start[-1]), 128), np.uint8)
        kp = np.zeros((int(start[-1]), 7), np.float32)
        for r, p in by_rank.items():
            at = 0
            for t in range(r, n_tiles, world):
                c = int(p.tiles[t, 5])
                desc[start[t]:start[t] + c] = p.desc[at:at + c]
                kp[start[t]:start[t] + c] = p.kp[at:at + c]
                at += c
        tiles = parts[0].tiles.copy()
        tiles[:, 4] = start[:-1]
        tiles[:, 5] = start[1:]
        meta = dict(parts[0].meta)
        meta["rank"], meta["world"] = 0, 1
        return cls(desc, kp, tiles, meta)

    # ---- disk ----------------------------------------------------------------------------------------------
    def save(self, path):
        os.makedirs(path, exist_ok=True)
        np.save(os.path.join(path, "desc.npy"), np.ascontiguousarray(self.desc, dtype=np.uint8))
        np.save(os.path.join(path, "kp.npy"), np.ascontiguousarray(self.kp, dtype=np.float32))
        np.save(os.path.join(path, "tiles.npy"), np.ascontiguousarray(self.tiles, dtype=np.int32))
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(self.meta, f, indent=1)

    @classmethod
    def load(cls, path, mmap=True):
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        if meta.get("format") != FORMAT_VERSION:
            raise ValueError("map database format %r, this build reads %d" % (meta.get("format"), FORMAT_VERSION))
        mode = "r" if mmap else None
        desc = np.load(os.path.join(path, "desc.npy"), mmap_mode=mode)       # allow_pickle stays False
        kp = np.load(os.path.join(path, "kp.npy"), mmap_mode=mode)
        tiles = np.load(os.path.join(path, "tiles.npy"), mmap_mode=mode)
        if desc.ndim != 2 or desc.shape[1] != 128 or desc.dtype != np.uint8 or kp.shape != (desc.shape[0], 7):
            raise ValueError("map database arrays do not fit together")
        return cls(desc, kp, tiles, meta)

    # ---- HBM -----------------------------------------------------------------------------------------------
    def to_device(self, ctx, rank=0, world=1, chunk=1 << 20):
        """This rank's contiguous row range as a descriptor set pinned in HBM (kept until `release`). Returns
        (features, first_row). A 20 M-row map is 5.3 GB packed: far inside 288 GB, so it is loaded whole."""
        from .sharded import partition
        key = (id(ctx), rank, world)
        if key not in self._resident:
            lo, hi = partition(len(self), world)[rank]
            rows = np.ascontiguousarray(self.desc[lo:hi])        # reads the memory-mapped range
            f = ctx.features(rows)
            f.set_keypoints(np.asarray(self.kp[lo:hi]))          # the range's keypoints beside its rows (match_localize)
            self._resident[key] = (f, lo)
        return self._resident[key]

    def release(self):
        for f, _ in self._resident.values():
            f.free()
        self._resident = {}

    def as_roi(self, ctx, main_map):
        """The whole resident map in the shape the reference's callers hand to find_matches / find_keypoints_transform
        / solve_IK (the `roi` dataclass, match_finder.py:16-23): pose_from_roi then runs unchanged against the database
        instead of a freshly extracted crop (position_finder.py:292-294)."""
        from .cvtypes import DeviceDescriptors, KeyPoints
        from .match_finder import roi
        f_map, _ = self.to_device(ctx)
        return roi(img=main_map.img, kp=KeyPoints(np.asarray(self.kp)), dp=DeviceDescriptors(f_map),
                   pixel_size=main_map.pixel_size, pixel_x_main_map=0, pixel_y_main_map=0)

    # ---- matching a frame against the resident map -----------------------------------------------------------
    def match(self, ctx, frame_features, ratio=0.7):
        """kNN (k=2) + Lowe ratio of a frame's descriptor set against the resident map (single GPU).
        -> (query rows, map rows, distances) of the good matches; map rows index `kp` / `xy`."""
        import torch
        f_map, lo = self.to_device(ctx)
        nq = len(frame_features)
        out4 = torch.empty((max(nq, 1), 4), dtype=torch.int32, device="cuda:%d" % ctx.device)
        ctx.knn2_dev(frame_features, f_map, out4.data_ptr(), lo)
        return ctx.ratio_dev(out4.data_ptr(), nq, ratio)

    def locate(self, ctx, frame_features, frame_xy=None, ratio=0.7, thr=0.1, seed=0, min_good=10):
        """The frame's homography into FULL-MAP pixel coordinates (find_matches + find_keypoints_transform against
        the whole map instead of a crop): -> (H or None, number of good matches)."""
        if frame_xy is None:
            # the frame's set comes from sift_extract* and carries its keypoints: the whole tail in one library call
            # (same survivors, same H as the calls below)
            f_map, lo = self.to_device(ctx)
            gq, gt, _, H, _ = ctx.match_localize(frame_features, f_map, ratio, min_good, thr, seed)
            return H, len(gq)
        gq, gt, _ = self.match(ctx, frame_features, ratio)
        if len(gq) <= min_good:                                   # position_finder.py:316
            return None, len(gq)
        H, _ = ctx.homography(np.asarray(frame_xy)[gq], np.asarray(self.xy)[gt], thr, seed)
        return H, len(gq)
