#!/bin/bash
# One frame against the 236 k-row map: the launcher's knobs, one at a time (kernel time by HIP events, tools/knn_shapes.py).
set -o pipefail
R=$(cd "$(dirname "$0")/../.." && pwd)
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 $R/tools/knn_shapes.py 3022x235864 2>&1 | grep "Q=" || exit 1; }
run A=1
run IMATCH_KNN_FORM=8
run IMATCH_KNN_FORM=4 IMATCH_KNN_CHUNKS=32
run IMATCH_KNN_FORM=4 IMATCH_KNN_CHUNKS=48
run IMATCH_KNN_FORM=4 IMATCH_KNN_CHUNKS=96
run IMATCH_KNN_FORM=4 IMATCH_KNN_CHUNKS=128
run IMATCH_KNN_FORM=8 IMATCH_KNN_CHUNKS=64
run IMATCH_KNN_FORM=8 IMATCH_KNN_CHUNKS=96
run IMATCH_KNN_FORM=8 IMATCH_KNN_CHUNKS=192
run IMATCH_KNN_YOUNG=2
run IMATCH_KNN_YOUNG=4
run IMATCH_KNN_YOUNG=16
run IMATCH_KNN_ROT=0
