"""GPU parity of the AKAZE path (csrc/akaze.hip) with the CPU oracle ("Spec A", oracle/akaze_oracle.c), through the
C-ABI. Bars: the scale space, the derivative planes and the detector response are bit-exact (same f32 operation order,
correctly rounded division and square root on both sides); keypoint position/size/response/level identical;
orientation within 0.01 degree and descriptors within L2 1e-4 for >= 99 % of keypoints (atan2f / expf / sincosf differ
in the last place between libm and the device library, which can move a sample across an angle slice)."""
import os

import numpy as np
import pytest

import oracle
from image_matching_amd import AkazeParams, ImatchError, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compare(ctx, img, planes=True, threshold=0.0005, n_octaves=4, n_layers=5):
    want = oracle.akaze(img, threshold, n_octaves, n_layers, keep_levels=planes)
    kp, desc = ctx.akaze_extract(img, AkazeParams(threshold, n_octaves, n_layers, clahe=False))
    if planes:
        assert ctx.akaze_levels() == len(want.levels)
        assert ctx.akaze_debug_layer(5, 0)[0, 0] == np.float32(want.kcontrast)
        for i, lv in enumerate(want.levels):
            for kind, name in enumerate(("Lt", "Lsmooth", "Lx", "Ly", "Ldet")):
                got = ctx.akaze_debug_layer(kind, i)
                assert np.array_equal(got, lv[name]), (i, name, float(np.abs(got - lv[name]).max()))
    assert len(kp) == len(want.kp)
    assert np.array_equal(kp[:, [0, 1, 2, 4, 5, 6]], want.kp[:, [0, 1, 2, 4, 5, 6]])
    if len(kp):
        da = np.abs((kp[:, 3] - want.kp[:, 3] + 180.0) % 360.0 - 180.0)
        dd = np.linalg.norm(desc - want.desc, axis=1)
        same = da < 0.01
        assert same.mean() >= 0.99, (same.mean(), np.sort(da)[-5:])
        assert (dd[same] < 1e-4).mean() >= 0.99, np.sort(dd[same])[-5:]
        assert np.allclose(np.linalg.norm(desc, axis=1), 1.0, atol=1e-5)
    return kp, desc


def test_frame_640x480_all_planes(ctx):
    img = np.ascontiguousarray(synth.synth_map(1024, seed=3)[100:580, 200:840])
    kp, _ = _compare(ctx, img)
    assert len(kp) > 100


def test_other_sizes(ctx):
    m = synth.synth_map(1024, seed=9)
    for h, w in ((240, 320), (88, 200), (360, 648), (321, 448), (479, 641), (250, 333), (1024, 1024)):   # odd sizes: general INTER_AREA
        _compare(ctx, np.ascontiguousarray(m[:h, :w]), planes=(h < 500))


def test_awkward_sizes(ctx):
    """Either side of the single-workgroup level's 19 200-pixel limit, thin strips, odd sizes in both axes; on the small ones
    a noise crop as well (many keypoints per level)."""
    m = synth.synth_map(1536, seed=31)
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, (700, 700), dtype=np.uint8)
    total = 0
    for h, w in [(120, 160), (120, 161), (121, 160), (40, 600), (600, 40), (277, 139), (139, 277), (64, 64), (65, 65), (320, 240),
                 (553, 555), (97, 131), (200, 96), (768, 1024), (1100, 900)]:
        y0, x0 = int(rng.integers(0, 1536 - h)), int(rng.integers(0, 1536 - w))
        _compare(ctx, np.ascontiguousarray(m[y0:y0 + h, x0:x0 + w]), planes=(h * w < 400000))
        if h * w < 50000:
            kp, _ = _compare(ctx, np.ascontiguousarray(noise[:h, :w]), planes=True, threshold=1e-5)
            total += len(kp)
    assert total > 200


def test_other_parameters(ctx):
    """Other thresholds, octave and sublevel counts than the reference's call (sigma_size reaches 5 with 3 sublevels... the
    FED cycles and borders change with them)."""
    img = np.ascontiguousarray(synth.synth_map(1024, seed=12)[:384, :512])
    _compare(ctx, img, threshold=0.001, n_octaves=2, n_layers=3)
    _compare(ctx, img, threshold=0.0002, n_octaves=3, n_layers=4)
    _compare(ctx, img, threshold=0.0005, n_octaves=1, n_layers=1)


def test_noise_image_overflows_and_regrows_the_candidate_list(ctx):
    """White noise has far more 3 x 3 maxima than the list sized from the image holds: the scan is repeated into a larger
    list, the result is the oracle's."""
    img = np.random.default_rng(5).integers(0, 256, (480, 640), dtype=np.uint8)
    kp, _ = _compare(ctx, img, planes=False, threshold=1e-6)
    assert len(kp) > 480 * 640 // 16 // 4


def test_flat_and_tiny_images(ctx):
    kp, desc = ctx.akaze_extract(np.full((120, 160), 77, np.uint8), AkazeParams(clahe=False))
    assert kp.shape == (0, 7) and desc.shape == (0, 64)
    _compare(ctx, np.random.default_rng(1).integers(0, 256, (48, 64), dtype=np.uint8))   # one octave, no room inside the border
    with pytest.raises(ImatchError):
        ctx.akaze_extract(np.zeros((64, 64), np.uint8), AkazeParams(n_octaves=0))


def test_clahe_first_as_find_kp_dp(ctx):
    """find_kp_dp applies CLAHE(30, (8, 8)) before the detector and returns the equalised image (MF:200-203,221)."""
    img = np.ascontiguousarray(synth.synth_map(512, seed=4)[:240, :320])
    kp, desc, out = ctx.akaze_extract(img, AkazeParams(), image=True)
    eq = oracle.clahe(img, 30.0, (8, 8))
    assert np.array_equal(out, eq)
    want = oracle.akaze(eq)
    assert np.array_equal(kp[:, [0, 1, 2, 4, 5, 6]], want.kp[:, [0, 1, 2, 4, 5, 6]])


def test_device_image_and_strided_rows(ctx):
    """imatch_akaze_extract_dev: the frame hand-off from a torch tensor, rows further apart than the width."""
    import torch
    img = np.ascontiguousarray(synth.synth_map(512, seed=8)[:240, :320])
    want_kp, want_desc = ctx.akaze_extract(img, AkazeParams())
    wide = torch.zeros((240, 384), dtype=torch.uint8, device="cuda:0")
    wide[:, :320] = torch.from_numpy(img).to("cuda:0")
    torch.cuda.synchronize()
    for clahe in (True, False):
        kp, desc = ctx.akaze_extract_dev(wide.data_ptr(), 240, 320, 384, AkazeParams(clahe=clahe))
        ref_kp, ref_desc = (want_kp, want_desc) if clahe else ctx.akaze_extract(img, AkazeParams(clahe=False))
        assert np.array_equal(kp, ref_kp) and np.array_equal(desc, ref_desc)


def test_whole_map_8192(ctx):
    """BASELINE.json's 8192 x 8192 map through the AKAZE detector in one call (calculate_db's extraction with
    detector="akaze"): ~1.6e5 keypoints, the same as the oracle's to the bars above."""
    kp, _ = _compare(ctx, synth.synth_map(8192, seed=21), planes=False)
    assert len(kp) > 100000


def test_golden_fixture(ctx):
    g = np.load(os.path.join(ROOT, "tests", "golden", "akaze_small.npz"))
    kp, desc = ctx.akaze_extract(g["img"], AkazeParams(clahe=False))
    assert np.array_equal(kp[:, [0, 1, 2, 4, 5, 6]], g["kp"][:, [0, 1, 2, 4, 5, 6]])
    assert np.array_equal(ctx.akaze_debug_layer(0, 4), g["Lt_4"]) and np.array_equal(ctx.akaze_debug_layer(4, 6), g["Ldet_6"])
    assert np.median(np.linalg.norm(desc - g["desc"], axis=1)) < 1e-5
