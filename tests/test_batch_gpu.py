"""GPU parity of the batched entry points: every image of a batch must get exactly what the
single-image entry points give it (keypoints, descriptors, good matches, homography)."""
import numpy as np
import pytest
import torch

import image_matching_amd as im
from image_matching_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene():
    m = synth.synth_map(1024, seed=7)
    frames = [synth.synth_frame(m, i, size=(240, 320))[0] for i in range(5)]
    return m, frames


@pytest.mark.parametrize("clahe", [True, False])
def test_extract_batch_equals_single(ctx, scene, clahe):
    _, frames = scene
    dev = torch.device("cuda", 0)
    stack = torch.from_numpy(np.stack(frames)).to(dev)
    params = im.SiftParams(clahe=clahe)
    f, offs = ctx.sift_extract_batch_dev(stack.data_ptr(), len(frames), 240 * 320, 240, 320, 320, params)
    kp, desc = f.download()
    rows = np.array([0, len(f) - 1, len(f) // 2, 0], dtype=np.int64)
    assert np.array_equal(f.keypoints_xy(rows), kp[rows, :2])          # the gather findHomography's caller makes (MF:231-232)
    with pytest.raises(Exception):
        f.keypoints_xy([len(f)])
    assert offs[0] == 0 and offs[-1] == len(f) and np.all(np.diff(offs) > 50)
    for b, fr in enumerate(frames):
        one = ctx.sift_extract(fr, params)
        kp1, desc1 = one.download()
        assert np.array_equal(kp[offs[b]:offs[b + 1]], kp1), b
        assert np.array_equal(desc[offs[b]:offs[b + 1]], desc1), b
        one.free()
    f.free()


def test_batch_of_one_and_ragged_pitch(ctx, scene):
    _, frames = scene
    dev = torch.device("cuda", 0)
    # images 100 KB apart, rows 384 bytes apart
    buf = torch.zeros((3, 102400), dtype=torch.uint8, device=dev)
    for b in range(3):
        pad = np.zeros((240, 384), np.uint8)
        pad[:, :320] = frames[b]
        buf[b, :240 * 384] = torch.from_numpy(pad.reshape(-1)).to(dev)
    torch.cuda.synchronize()   # torch's fills ran on its stream; the context's (non-blocking) stream reads the buffer next
    f, offs = ctx.sift_extract_batch_dev(buf.data_ptr(), 3, 102400, 240, 320, 384)
    kp, desc = f.download()
    for b in range(3):
        one = ctx.sift_extract(frames[b])
        kp1, desc1 = one.download()
        assert np.array_equal(kp[offs[b]:offs[b + 1]], kp1) and np.array_equal(desc[offs[b]:offs[b + 1]], desc1)
    f1, o1 = ctx.sift_extract_batch_dev(buf.data_ptr(), 1, 102400, 240, 320, 384)
    assert o1.tolist() == [0, offs[1]]


def test_full_batch_at_the_bench_shape_and_odd_sizes(ctx, scene):
    """64 frames of 640 x 480 in one call (bench.py's step), one of them flat (an empty segment in the middle), and a batch
    of 7 odd-sized images: every image as the single-image call gives it."""
    m, _ = scene
    dev = torch.device("cuda", 0)
    base = [synth.synth_frame(m, 40 + i, size=(480, 640), scale_range=(0.5, 0.6))[0] for i in range(6)]
    frames = [base[i % 6] for i in range(64)]
    frames[29] = np.full((480, 640), 93, np.uint8)
    frames[63] = np.ascontiguousarray(base[2][::-1])
    stack = torch.from_numpy(np.stack(frames)).to(dev)
    f, offs = ctx.sift_extract_batch_dev(stack.data_ptr(), 64, 480 * 640, 480, 640, 640)
    kp, desc = f.download()
    assert offs[29] == offs[30] and offs[-1] == len(f)
    singles = {}
    for b in (0, 1, 5, 28, 29, 30, 31, 32, 62, 63):
        key = b % 6 if b not in (29, 63) else b
        if key not in singles:
            one = ctx.sift_extract(frames[b])
            singles[key] = one.download()
            one.free()
        kp1, desc1 = singles[key]
        assert np.array_equal(kp[offs[b]:offs[b + 1]], kp1) and np.array_equal(desc[offs[b]:offs[b + 1]], desc1), b
    f.free()
    odd = [np.ascontiguousarray(m[10 * i:10 * i + 243, 7 * i:7 * i + 325]) for i in range(7)]
    stack = torch.from_numpy(np.stack(odd)).to(dev)
    f, offs = ctx.sift_extract_batch_dev(stack.data_ptr(), 7, 243 * 325, 243, 325, 325, im.SiftParams(clahe=False))
    kp, desc = f.download()
    for b in range(7):
        one = ctx.sift_extract(odd[b], im.SiftParams(clahe=False))
        kp1, desc1 = one.download()
        assert np.array_equal(kp[offs[b]:offs[b + 1]], kp1) and np.array_equal(desc[offs[b]:offs[b + 1]], desc1), b
        one.free()
    f.free()


def test_largest_batch_256_frames_at_the_bench_shape(ctx, scene):
    """IMATCH_MAX_BATCH = 256 frames of 640 x 480 in one call (bench.py's step at configs[1]): a sample of the images, the
    first and the last included, as the single-image call gives them."""
    m, _ = scene
    dev = torch.device("cuda", 0)
    base = [synth.synth_frame(m, 60 + i, size=(480, 640), scale_range=(0.5, 0.6))[0] for i in range(5)]
    frames = [base[i % 5] for i in range(256)]
    frames[200] = np.ascontiguousarray(base[1][::-1])
    stack = torch.from_numpy(np.stack(frames)).to(dev)
    f, offs = ctx.sift_extract_batch_dev(stack.data_ptr(), 256, 480 * 640, 480, 640, 640)
    kp, desc = f.download()
    assert len(offs) == 257 and offs[-1] == len(f)
    singles = {}
    for b in (0, 1, 127, 128, 199, 200, 201, 254, 255):
        key = b % 5 if b != 200 else b
        if key not in singles:
            one = ctx.sift_extract(frames[b])
            singles[key] = one.download()
            one.free()
        kp1, desc1 = singles[key]
        assert np.array_equal(kp[offs[b]:offs[b + 1]], kp1) and np.array_equal(desc[offs[b]:offs[b + 1]], desc1), b
    f.free()


def test_batch_of_128_images_through_the_whole_chain_and_257_refused(ctx, scene):
    """128 images (half of IMATCH_MAX_BATCH) through extraction, one kNN launch, the batched ratio test and
    the batched homography: a sample of the images as the single-image chain gives them; one image more is refused."""
    m, frames5 = scene
    dev = torch.device("cuda", 0)
    f_map = ctx.sift_extract(m)
    map_xy = f_map.download(descriptors=False)[0][:, :2]
    frames = [frames5[i % 5] for i in range(128)]
    frames[77] = np.ascontiguousarray(frames5[3][:, ::-1])
    stack = torch.from_numpy(np.stack(frames)).to(dev)
    f, offs = ctx.sift_extract_batch_dev(stack.data_ptr(), 128, 240 * 320, 240, 320, 320)
    kp, desc = f.download()
    assert len(offs) == 129 and offs[-1] == len(f)
    out4 = torch.empty((len(f), 4), dtype=torch.int32, device=dev)
    ctx.knn2_dev(f, f_map, out4.data_ptr())
    gq, gt, gd, goff = ctx.ratio_batch_dev(out4.data_ptr(), offs, 0.7)
    src = np.concatenate([kp[offs[b] + gq[goff[b]:goff[b + 1]], :2] for b in range(128)])
    H, mask, found = ctx.homography_batch(src, map_xy[gt], goff, 0.1, seeds=np.arange(128))
    for b in (0, 64, 77, 127):
        one = ctx.sift_extract(frames[b])
        kp1, desc1 = one.download()
        assert np.array_equal(kp[offs[b]:offs[b + 1]], kp1) and np.array_equal(desc[offs[b]:offs[b + 1]], desc1), b
        idx, dist = ctx.knn2(one, f_map)
        o4 = out4[offs[b]:offs[b + 1]].cpu().numpy()
        assert np.array_equal(o4[:, [1, 3]], idx)
        o1 = torch.empty((len(one), 4), dtype=torch.int32, device=dev)
        ctx.knn2_dev(one, f_map, o1.data_ptr())
        q1, t1, d1 = ctx.ratio_dev(o1.data_ptr(), len(one), 0.7)
        assert np.array_equal(gq[goff[b]:goff[b + 1]], q1) and np.array_equal(gt[goff[b]:goff[b + 1]], t1)
        if len(q1) >= 4:
            H1, _ = ctx.homography(kp1[q1, :2], map_xy[t1], 0.1, seed=b)
            assert (H1 is None) == (not found[b])
            if H1 is not None:
                assert np.array_equal(H[b], H1)
        one.free()
    f.free()
    big = torch.zeros((257, 64 * 64), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    with pytest.raises(Exception):
        ctx.sift_extract_batch_dev(big.data_ptr(), 257, 64 * 64, 64, 64, 64)
    f_map.free()


def test_match_chain_batch_equals_single(ctx, scene):
    m, frames = scene
    dev = torch.device("cuda", 0)
    f_map = ctx.sift_extract(m)
    map_xy = f_map.download(descriptors=False)[0][:, :2]
    stack = torch.from_numpy(np.stack(frames)).to(dev)
    f, offs = ctx.sift_extract_batch_dev(stack.data_ptr(), len(frames), 240 * 320, 240, 320, 320)
    kp = f.download(descriptors=False)[0]
    out4 = torch.empty((len(f), 4), dtype=torch.int32, device=dev)
    ctx.knn2_dev(f, f_map, out4.data_ptr())
    gq, gt, gd, goff = ctx.ratio_batch_dev(out4.data_ptr(), offs, 0.7)
    src = np.concatenate([kp[offs[b] + gq[goff[b]:goff[b + 1]], :2] for b in range(len(frames))])
    H, mask, found = ctx.homography_batch(src, map_xy[gt], goff, 0.1, seeds=np.arange(len(frames)))
    for b, fr in enumerate(frames):
        one = ctx.sift_extract(fr)
        o1 = torch.empty((len(one), 4), dtype=torch.int32, device=dev)
        ctx.knn2_dev(one, f_map, o1.data_ptr())
        ctx.synchronize()
        assert torch.equal(o1, out4[offs[b]:offs[b + 1]])
        q1, t1, d1 = ctx.ratio_dev(o1.data_ptr(), len(one), 0.7)
        assert np.array_equal(q1, gq[goff[b]:goff[b + 1]]) and np.array_equal(t1, gt[goff[b]:goff[b + 1]])
        assert np.array_equal(d1, gd[goff[b]:goff[b + 1]])
        kp1 = one.download(descriptors=False)[0]
        H1, m1 = ctx.homography(kp1[q1, :2], map_xy[t1], 0.1, seed=b)
        assert (H1 is not None) == bool(found[b])
        if H1 is not None:
            assert np.array_equal(H1, H[b]) and np.array_equal(m1.ravel(), mask[goff[b]:goff[b + 1]])
    assert found.sum() >= 3


def test_batch_limits(ctx, scene):
    _, frames = scene
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(frames[0]).to(dev)
    with pytest.raises(im.ImatchError):
        ctx.sift_extract_batch_dev(t.data_ptr(), 257, 240 * 320, 240, 320, 320)
    with pytest.raises(im.ImatchError):
        ctx.sift_extract_batch_dev(t.data_ptr(), 2, 100, 240, 320, 320)
    # empty segments
    gq, gt, gd, goff = ctx.ratio_batch_dev(0, [0, 0, 0], 0.7)
    assert goff.tolist() == [0, 0, 0]
    H, mask, found = ctx.homography_batch(np.zeros((3, 2)), np.zeros((3, 2)), [0, 3], 0.1)
    assert not found.any()
