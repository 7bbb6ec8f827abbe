#!/bin/bash
# Runs on the GPU box (gpurun): the round's bench line, rocprofv3 kernel-trace summaries and PMC passes.
# Usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>/...
set -o pipefail
tag=${1:-r1}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
step() { echo "== $*" >&2; }

step bench
timeout -k 10 600 python3 -u "$root/bench.py" > "$out/bench.json" 2> "$out/bench.err" || exit 1

step kernel-trace bench
timeout -k 10 400 rocprofv3 --kernel-trace -d "$out/kt_bench" -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline \
    > "$out/kt_bench.json" 2> "$out/kt_bench.err" || exit 1
python3 "$root/tools/rocpd_stats.py" "$out"/kt_bench/*/*_results.db "$out/bench_kernel_stats.body.md" > /dev/null || exit 1
# header: the tracer's cost on the dominant kernel, from the two JSON lines
python3 - "$out" <<'PY' || exit 1
import json, sys
out = sys.argv[1]
ms = lambda f: json.loads(open(out + "/" + f).read().strip().splitlines()[-1])["roofline"]["launch_ms"]
head = ("rocprofv3 --kernel-trace of `bench.py --steps 5 --warmup 2 --no-cpu-baseline` (tools/collect_profiles.sh). Compare the per-shape kNN rows\n"
        "at the end with roofline.launch_ms of the same run (%.2f ms) and of the un-profiled bench beside it (%.2f ms): the difference is the\n"
        "tracer's cost on this kernel.\n\n" % (ms("kt_bench.json"), ms("bench.json")))
open(out + "/bench_kernel_stats.md", "w").write(head + open(out + "/bench_kernel_stats.body.md").read())
PY
rm -f "$out/bench_kernel_stats.body.md"

step kernel-trace map extraction
timeout -k 10 400 rocprofv3 --kernel-trace -d "$out/kt_map" -- python3 "$root/tools/extract_profile.py" 8192 3 \
    > "$out/kt_map.log" 2>&1 || exit 1
python3 "$root/tools/rocpd_stats.py" "$out"/kt_map/*/*_results.db "$out/map_extract_kernel_stats.md" > /dev/null || exit 1
python3 "$root/tools/rocpd_timeline.py" "$out"/kt_map/*/*_results.db 64 "$out/map_extract_timeline.md" > /dev/null || exit 1

step pmc FETCH_SIZE
export IMATCH_BENCH_SERIES=0     # the PMC passes need the headline workload only
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline \
    > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.err" || exit 1
python3 "$root/tools/pmc_sum.py" "$out"/pmc_fetch/*/*_results.db "$out/pmc_fetch.md" > /dev/null || exit 1

step pmc WRITE_SIZE
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline \
    > "$out/pmc_write.json" 2> "$out/pmc_write.err" || exit 1
python3 "$root/tools/pmc_sum.py" "$out"/pmc_write/*/*_results.db "$out/pmc_write.md" > /dev/null || exit 1

python3 "$root/tools/pmc_merge.py" "$out/pmc_fetch.md" "$out/pmc_write.md" "$out/bench.json" "$out/pmc.md" "$out/pmc_knn.json" || exit 1
unset IMATCH_BENCH_SERIES

step pmc map extraction
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_mapf" -- python3 "$root/tools/extract_profile.py" 8192 2 > "$out/pmc_mapf.log" 2>&1 || exit 1
python3 "$root/tools/pmc_sum.py" "$out"/pmc_mapf/*/*_results.db "$out/pmc_map_fetch.md" > /dev/null || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_mapw" -- python3 "$root/tools/extract_profile.py" 8192 2 > "$out/pmc_mapw.log" 2>&1 || exit 1
python3 "$root/tools/pmc_sum.py" "$out"/pmc_mapw/*/*_results.db "$out/pmc_map_write.md" > /dev/null || exit 1
python3 "$root/tools/pmc_pyramid.py" "$out/pmc_map_fetch.md" "$out/pmc_map_write.md" 8192 2 "$out/pmc_pyramid.json" "$out/pmc_pyramid.md" || exit 1
rm -rf "$out"/pmc_mapf "$out"/pmc_mapw

step replay
timeout -k 10 400 python3 "$root/tools/replay_c3.py" 240 8192 > "$out/replay_c3.json" 2> "$out/replay_c3.err" || exit 1

# the .db files are large; keep the summaries
rm -rf "$out"/kt_bench "$out"/kt_map "$out"/pmc_fetch "$out"/pmc_write
echo done >&2
