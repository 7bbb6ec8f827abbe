"""Map descriptor database (image_matching_amd/mapdb.py; reference: the in-memory tile cache of
position_finder.py:248-272 and match_finder.py:86-128). CPU: tiling, merge of a sharded build, the on-disk
format. GPU: build == direct extraction, sharded build == single build, a frame located against the resident map."""
import json
import os

import numpy as np
import pytest

from image_matching_amd import synth
from image_matching_amd.mapdb import MapDB


def _part(rank, world, grid, counts, seed):
    rng = np.random.default_rng(seed)
    mine = [t for t in range(len(grid)) if t % world == rank]
    n = sum(counts[t] for t in mine)
    tiles = np.zeros((len(grid), 6), np.int32)
    tiles[:, :4] = np.asarray(grid, np.int32)
    for t in mine:
        tiles[t, 5] = counts[t]
    desc = rng.integers(0, 256, size=(n, 128), dtype=np.uint8)
    kp = rng.random((n, 7), dtype=np.float32)
    at = 0
    for t in mine:                      # tag every row with its tile so that the merged order can be checked
        kp[at:at + counts[t], 6] = t
        at += counts[t]
    meta = {"format": 1, "height": 300, "width": 500, "tile": 200, "margin": 16, "pixel_size": 0.5, "corners": None,
            "sift": {}, "rank": rank, "world": world}
    return MapDB(desc, kp, tiles, meta)


def test_tile_grid_covers_the_map_once():
    grid = MapDB.tile_grid(300, 500, 200)
    assert len(grid) == 6 and grid[0] == (0, 0, 200, 200) and grid[-1] == (400, 200, 500, 300)
    cover = np.zeros((300, 500), np.int32)
    for x0, y0, x1, y1 in grid:
        cover[y0:y1, x0:x1] += 1
    assert (cover == 1).all()


def test_merge_orders_rows_tile_by_tile():
    grid = MapDB.tile_grid(300, 500, 200)
    counts = [5, 0, 7, 3, 11, 2]
    parts = [_part(r, 3, grid, counts, 10 + r) for r in (2, 0, 1)]          # any order
    db = MapDB.merge(parts)
    assert len(db) == sum(counts) and db.meta["world"] == 1
    assert db.tiles[:, 4].tolist() == [0, 5, 5, 12, 15, 26] and db.tiles[:, 5].tolist() == [5, 5, 12, 15, 26, 28]
    for t, (r0, r1) in enumerate(db.tiles[:, 4:6]):
        assert (db.kp[r0:r1, 6] == t).all()
    p1 = [p for p in parts if p.meta["rank"] == 1][0]                       # rank 1 holds tiles 1 and 4
    assert np.array_equal(db.desc[15:26], p1.desc[0:11])
    with pytest.raises(ValueError):
        MapDB.merge(parts[:2])


def test_save_load_round_trip(tmp_path):
    grid = MapDB.tile_grid(300, 500, 200)
    db = MapDB.merge([_part(0, 1, grid, [4, 4, 4, 4, 4, 4], 3)])
    path = str(tmp_path / "map.db")
    db.save(path)
    assert sorted(os.listdir(path)) == ["desc.npy", "kp.npy", "meta.json", "tiles.npy"]
    back = MapDB.load(path)
    assert isinstance(back.desc, np.memmap) and back.desc.dtype == np.uint8
    assert np.array_equal(back.desc, db.desc) and np.array_equal(back.kp, db.kp) and np.array_equal(back.tiles, db.tiles)
    assert back.meta == db.meta
    meta = json.load(open(os.path.join(path, "meta.json")))
    meta["format"] = 99
    json.dump(meta, open(os.path.join(path, "meta.json"), "w"))
    with pytest.raises(ValueError):
        MapDB.load(path)


@pytest.mark.gpu
def test_build_matches_direct_extraction_and_sharded_build(ctx, tmp_path):
    m = synth.synth_map(2048, seed=31)
    # one tile that covers the map: exactly the direct extraction
    f = ctx.sift_extract(m)
    kp, desc = f.download()
    f.free()
    whole = MapDB.build(ctx, m, tile=4096, margin=0, pixel_size=0.5)
    assert np.array_equal(whole.kp, kp) and np.array_equal(whole.desc, desc.astype(np.uint8))
    # 2 x 2 tiles with margins: every keypoint inside its tile's core, row ranges contiguous
    db = MapDB.build(ctx, m, tile=1024, margin=128, pixel_size=0.5)
    assert len(db.tiles) == 4 and db.tiles[0, 4] == 0 and db.tiles[-1, 5] == len(db)
    assert (db.tiles[1:, 4] == db.tiles[:-1, 5]).all()
    for x0, y0, x1, y1, r0, r1 in db.tiles:
        x, y = db.kp[r0:r1, 0], db.kp[r0:r1, 1]
        assert ((x >= x0) & (x < x1) & (y >= y0) & (y < y1)).all()
    # (not the direct extraction's count: CLAHE's 8 x 8 grid is laid over each crop, as in the reference's per-ROI
    # find_kp_dp, so smaller crops are equalised more locally and yield more keypoints)
    assert 0.8 * len(kp) < len(db) < 1.5 * len(kp)
    # the same build sharded over two "ranks" (tiles 0,2 / 1,3), merged: identical to the single build
    parts = [MapDB.build(ctx, m, tile=1024, margin=128, pixel_size=0.5, rank=r, world=2) for r in (0, 1)]
    merged = MapDB.merge(parts)
    assert np.array_equal(merged.desc, db.desc) and np.array_equal(merged.kp, db.kp) and np.array_equal(merged.tiles, db.tiles)
    # disk -> HBM -> a frame located in full-map pixels within 1 px of the planted homography
    path = str(tmp_path / "m.db")
    db.save(path)
    back = MapDB.load(path)
    frame, Htrue = synth.synth_frame(m, 5, size=(480, 640), scale_range=(0.9, 1.1))
    ff = ctx.sift_extract(frame)
    fxy = ff.download(descriptors=False)[0][:, :2]
    H, n_good = back.locate(ctx, ff, fxy)
    assert H is not None and n_good > 50
    corners = np.array([[0, 0], [0, 479], [639, 479], [639, 0]], dtype=np.float64)

    def proj(Hm, p):
        q = np.c_[p, np.ones(len(p))] @ np.asarray(Hm).T
        return q[:, :2] / q[:, 2:]
    assert np.abs(proj(H, corners) - proj(Htrue, corners)).max() <= 1.0
    # without the coordinates: the frame's set carries them, the resident set got the database's keypoints when it was
    # loaded (imatch_features_set_keypoints) -- one library call (imatch_match_localize), the same H bit for bit
    H1, n1 = back.locate(ctx, ff)
    assert n1 == n_good and np.array_equal(H1, H)
    # resident set: a second call does not upload again, and a sharded load splits the rows contiguously
    assert back.to_device(ctx)[0] is back.to_device(ctx)[0]
    f0, lo0 = back.to_device(ctx, 0, 2)
    f1, lo1 = back.to_device(ctx, 1, 2)
    assert lo0 == 0 and lo1 == len(f0) and len(f0) + len(f1) == len(back)
    back.release()
    ff.free()
