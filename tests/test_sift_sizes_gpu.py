"""GPU parity of CLAHE and the SIFT path on awkward image sizes (tile, wave and octave boundaries +-1, very wide and very
tall strips): Gaussian and DoG planes bit-exact against the oracle, keypoints to test_sift_gpu.py's bars."""
import numpy as np
import pytest
import oracle
import image_matching_amd as im
from image_matching_amd import synth
from test_sift_gpu import _match_keypoints
pytestmark = pytest.mark.gpu

SHAPES = [(255, 257), (256, 256), (257, 255), (511, 513), (129, 1023), (1023, 129), (33, 700), (700, 33), (65, 63),
          (321, 479), (600, 799), (18, 300), (1025, 1025), (127, 128), (385, 640)]


def test_awkward_sizes(ctx):
    m = synth.synth_map(1536, seed=77)
    rng = np.random.default_rng(3)
    for h, w in SHAPES:
        y0, x0 = int(rng.integers(0, 1536 - h)), int(rng.integers(0, 1536 - w))
        img = np.ascontiguousarray(m[y0:y0 + h, x0:x0 + w])
        want = oracle.sift(img, keep_pyramid=True)
        f = ctx.sift_extract(img, im.SiftParams(clahe=False))
        assert ctx.sift_debug_octaves() == len(want.gauss), (h, w)
        for o in range(len(want.gauss)):
            for i in range(6):
                assert np.array_equal(ctx.sift_debug_layer(0, o, i), want.gauss[o][i]), (h, w, "gauss", o, i)
            for i in range(5):
                assert np.array_equal(ctx.sift_debug_layer(1, o, i), want.dog[o][i]), (h, w, "dog", o, i)
        kp, desc = f.download()
        if len(want.kp) > 20:
            idx, dist = _match_keypoints(kp, want.kp)
            found = idx >= 0
            l2 = np.linalg.norm(desc[idx[found]] - want.desc[found], axis=1)
            print(h, w, len(kp), len(want.kp), "matched %.4f exact %.4f descL2 p99 %.2f" % (found.mean(), (dist[found] == 0).mean(), np.percentile(l2, 99)), flush=True)
            assert found.mean() >= 0.99 and abs(len(kp) - len(want.kp)) <= 0.01 * len(want.kp) + 2, (h, w)
        else:
            print(h, w, len(kp), len(want.kp), flush=True)
            assert abs(len(kp) - len(want.kp)) <= 2
        assert np.array_equal(ctx.clahe(img), oracle.clahe(img)), (h, w, "clahe")
        f.free()
