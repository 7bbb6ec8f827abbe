"""Golden fixture of the AKAZE path, made by the CPU oracle (oracle/akaze_oracle.c) in this container.
PARITY UNPINNED against OpenCV (absent here); the fixture pins the oracle and the HIP path to this recorded state.
Usage: python tests/golden/make_akaze_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from image_matching_amd import synth  # noqa: E402


def main():
    img = np.ascontiguousarray(synth.synth_map(512, seed=11)[40:280, 60:380])    # 240 x 320: two octaves
    r = oracle.akaze(img, keep_levels=True)
    np.savez_compressed(os.path.join(HERE, "akaze_small.npz"), img=img, kp=r.kp, desc=r.desc,
                        kcontrast=np.float32(r.kcontrast), Lt_4=r.levels[4]["Lt"], Ldet_6=r.levels[6]["Ldet"])
    print(len(r.kp), r.kcontrast, len(r.levels))


main()
