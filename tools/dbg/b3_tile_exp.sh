set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${B3_VARIANTS:-base b3s}; do
  if [ $v = base ]; then unset IMATCH_LIB; else export IMATCH_LIB=$R/ab/lib_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$v -- python3 $R/tools/dbg/frame_stages.py 8192 20 > $R/gpurun_out/kt_$v.log 2>&1 || exit 1
  python3 $R/tools/rocpd_timeline.py $R/gpurun_out/kt_$v/*/*_results.db 40 $R/gpurun_out/b3_$v.md > /dev/null || exit 1
  rm -rf $R/gpurun_out/kt_$v
  FS_GC=freeze FS_NOPROF=1 FS_ONE_CALL=1 timeout -k 10 120 python3 $R/tools/dbg/frame_stages.py 8192 500 > $R/gpurun_out/b3w_$v.txt 2>&1 || exit 1
done
