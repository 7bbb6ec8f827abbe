#!/bin/bash
# development: env-knob sweep of the int8 kNN launch over the shapes that matter
set -o pipefail
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/knn_i8_sweep.txt
: > $out
S="754853x235864 188136x235864 131072x2500000"
run() { echo "== $1" >> $out; env $1 timeout -k 10 120 python3 $root/tools/knn_shapes.py $2 >> $out 2>&1 || exit 1; }
run "X=0" "$S 2951x235864 2951x58717 4096x5000000"
run "IMATCH_KNN_TN=256" "$S"
run "IMATCH_KNN_CHUNKS=8" "$S"
run "IMATCH_KNN_CHUNKS=32" "$S"
run "IMATCH_KNN_CHUNKS=64" "$S"
run "IMATCH_KNN_YOUNG=4" "$S"
