#!/bin/bash
# development: env-knob sweep of the int8 kNN launch over one frame's shapes
set -o pipefail
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/knn_i8_sweep_frame.txt
: > $out
S="2951x235864 2951x58717 1500x58717 5000x235864 2951x1000000"
run() { echo "== $1" >> $out; env $1 timeout -k 10 120 python3 $root/tools/knn_shapes.py $S >> $out 2>&1 || exit 1; }
run "X=0"
run "IMATCH_KNN_FORM=4"
run "IMATCH_KNN_FORM=8"
run "IMATCH_KNN_FORM=4 IMATCH_KNN_YOUNG=4"
run "IMATCH_KNN_FORM=4 IMATCH_KNN_YOUNG=2"
run "IMATCH_KNN_FORM=8 IMATCH_KNN_YOUNG=4"
