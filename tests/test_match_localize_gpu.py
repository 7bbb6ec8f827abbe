"""imatch_match_localize -- kNN, ratio test, coordinate lookup, Hartley normalisation and hypothesis scoring of one frame
without a host round trip in between -- against the per-frame calls it stands for (imatch_knn2_dev + imatch_ratio_dev +
imatch_homography), bit for bit, and against the CPU oracle's chain on the same frame."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import image_matching_amd as im
from image_matching_amd import synth
import oracle

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _separately(ctx, f, f_map, ratio, min_good, thr, seed):
    out4 = torch.empty((max(len(f), 1), 4), dtype=torch.int32, device="cuda:0")
    ctx.knn2_dev(f, f_map, out4.data_ptr())
    gq, gt, gd = ctx.ratio_dev(out4.data_ptr(), len(f), ratio)
    H = mask = None
    if len(gq) > min_good:
        kp = f.download(descriptors=False)[0]
        mkp = f_map.download(descriptors=False)[0]
        H, mask = ctx.homography(kp[gq, :2], mkp[gt, :2], thr, seed=seed)
    return gq, gt, gd, H, mask


@pytest.fixture(scope="module")
def scene():
    ctx = im.Context(0)
    map_img = synth.synth_map(2048, seed=77)
    f_map = ctx.sift_extract(map_img, im.SiftParams())
    yield ctx, map_img, f_map
    f_map.free()


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[3] is None) == (b[3] is None)
    if a[3] is not None:
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


@pytest.mark.parametrize("frame_seed", [0, 1, 2, 3])
def test_one_call_equals_the_separate_calls(scene, frame_seed):
    ctx, map_img, f_map = scene
    frame = synth.synth_frame(map_img, frame_seed, size=(480, 640))[0]
    f = ctx.sift_extract(frame, im.SiftParams())
    before = ctx.match_localize_mismatches()
    got = ctx.match_localize(f, f_map, 0.7, 10, 0.1, seed=frame_seed)
    want = _separately(ctx, f, f_map, 0.7, 10, 0.1, frame_seed)
    assert len(got[0]) > 10 and got[3] is not None
    _same(got, want)
    # the device's Hartley transforms were the host's: the call did not go through imatch_homography
    assert ctx.match_localize_mismatches() == before
    f.free()


def test_against_the_oracle_chain(scene):
    ctx, map_img, f_map = scene
    frame = synth.synth_frame(map_img, 5, size=(480, 640))[0]
    f = ctx.sift_extract(frame, im.SiftParams())
    gq, gt, gd, H, mask = ctx.match_localize(f, f_map, 0.7, 10, 0.1, seed=9)
    kp, desc = f.download()
    mkp, mdesc = f_map.download()
    idx, dist = oracle.knn2(desc.astype(np.uint8), mdesc.astype(np.uint8))
    oq, ot, od = oracle.ratio(idx, dist, 0.7)
    assert np.array_equal(gq, oq) and np.array_equal(gt, ot) and np.array_equal(gd, od)
    Ho, mo = oracle.homography(kp[oq, :2], mkp[ot, :2], 0.1, seed=9)
    assert np.array_equal(H, Ho) and np.array_equal(mask, mo)
    f.free()


def test_too_few_matches_and_unrelated_frames(scene):
    ctx, map_img, f_map = scene
    other = synth.synth_map(640, seed=4242)[:480, :640]      # a frame of another place: hardly a match survives
    f = ctx.sift_extract(np.ascontiguousarray(other), im.SiftParams())
    got = ctx.match_localize(f, f_map, 0.7, 10, 0.1, seed=1)
    want = _separately(ctx, f, f_map, 0.7, 10, 0.1, 1)
    _same(got, want)
    # min_good above the survivors: lists as ever, no homography
    frame = synth.synth_frame(map_img, 1, size=(480, 640))[0]
    f2 = ctx.sift_extract(frame, im.SiftParams())
    gq, gt, gd, H, mask = ctx.match_localize(f2, f_map, 0.7, 100000, 0.1, seed=1)
    assert len(gq) > 10 and H is None and mask is None
    _same((gq, gt, gd, None, None), _separately(ctx, f2, f_map, 0.7, 100000, 0.1, 1))
    # a flat frame: no keypoints at all
    f3 = ctx.sift_extract(np.full((480, 640), 90, np.uint8), im.SiftParams())
    assert len(f3) == 0
    gq, gt, gd, H, mask = ctx.match_localize(f3, f_map, 0.7, 10, 0.1)
    assert len(gq) == 0 and H is None
    for x in (f, f2, f3):
        x.free()


def test_sets_without_coordinates_are_refused(scene):
    ctx, map_img, f_map = scene
    rows = np.random.default_rng(0).integers(0, 256, (300, 128)).astype(np.uint8)
    f = ctx.features(rows)
    with pytest.raises(im.ImatchError, match="imatch_sift_extract"):
        ctx.match_localize(f, f_map)
    f.free()


_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r + "/tests")
import image_matching_amd as im
from image_matching_amd import synth
from test_match_localize_gpu import _separately, _same
ctx = im.Context(0)
map_img = synth.synth_map(2048, seed=77)
f_map = ctx.sift_extract(map_img, im.SiftParams())
frame = synth.synth_frame(map_img, 2, size=(480, 640))[0]
f = ctx.sift_extract(frame, im.SiftParams())
got = ctx.match_localize(f, f_map, 0.7, 10, 0.1, seed=3)
want = _separately(ctx, f, f_map, 0.7, 10, 0.1, 3)
_same(got, want)
print("OK", len(got[0]), ctx.match_localize_mismatches())
"""


@pytest.mark.parametrize("env,expect_mismatch", [({"IMATCH_TAIL_CAP": "16"}, 0), ({"IMATCH_TAIL_MISMATCH": "1"}, 1)])
def test_fallbacks_give_the_same_answer(env, expect_mismatch):
    """More survivors than the block holds (forced: a capacity of 16), and device-side transforms that are not the host's
    (forced): both take the per-frame calls inside the library and must give their answer."""
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, ROOT)], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    tag, n, mism = r.stdout.strip().splitlines()[-1].split()
    assert tag == "OK" and int(n) > 16 and int(mism) == expect_mismatch
