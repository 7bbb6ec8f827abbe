# extra evidence for the round: kNN shapes, frame phases + kernel timeline, launch time against idle gaps, MFMA counters
set -o pipefail
root=$(cd "$(dirname "$0")/../.." && pwd)
out=$root/gpurun_out/r2c_extra
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $root/tools/knn_shapes.py 188136x235864 131072x2500000 94068x1000000 4096x5000000 2951x1000000 5000x235864 2951x235864 1500x235864 5000x58717 2951x58717 1500x58717 800x58717 > $out/knn_shapes.txt 2>&1 || exit 1
timeout -k 10 200 python3 $root/tools/frame_profile.py 4096 100 > $out/frame_profile_4096.txt 2>&1 || exit 1
timeout -k 10 200 python3 $root/tools/frame_profile.py 8192 100 > $out/frame_profile_8192.txt 2>&1 || exit 1
timeout -k 10 200 python3 $root/tools/dbg/knn_cold.py > $out/knn_launch_vs_idle.txt 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace -d $out/tl -- python3 $root/tools/frame_profile.py 4096 12 > $out/tl.log 2>&1 || exit 1
f=$(find $out/tl -name "*.db" | head -1)
python3 $root/tools/rocpd_timeline.py $f 46 $out/frame_timeline.md > /dev/null || exit 1
rm -rf $out/tl
bash $root/tools/collect_knn_mfma.sh r2c_extra/mfma || exit 1
echo done
