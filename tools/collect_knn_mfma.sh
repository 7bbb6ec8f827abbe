#!/bin/bash
# MFMA-utilisation evidence for the kNN kernel (north_star: "rocprof counters evidence ... MFMA utilisation for matching"):
# one SQ pass and one GRBM pass of tools/knn_bench.py at the bench's launch shape and at one configs[4] shard.
# Usage (on the GPU box): tools/collect_knn_mfma.sh <tag>  ->  gpurun_out/<tag>/
set -o pipefail
tag=${1:-knn_mfma}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for shape in "188136 235864" "131072 2500000"; do
  set -- $shape
  n=$1x$2
  timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
      -d "$out/sq_$n" -- python3 "$root/tools/knn_bench.py" $1 $2 3 > "$out/sq_$n.log" 2>&1 || exit 1
  python3 "$root/tools/pmc_sum.py" "$out"/sq_$n/*/*_results.db "$out/sq_$n.md" > /dev/null || exit 1
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d "$out/grbm_$n" -- python3 "$root/tools/knn_bench.py" $1 $2 3 > "$out/grbm_$n.log" 2>&1 || exit 1
  python3 "$root/tools/pmc_sum.py" "$out"/grbm_$n/*/*_results.db "$out/grbm_$n.md" > /dev/null || exit 1
  rm -rf "$out/sq_$n" "$out/grbm_$n"
done
echo done >&2
