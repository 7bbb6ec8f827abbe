cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc_desc -- python3 $GRAFT_REPO_ROOT/tools/batch_profile.py 64 2048 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_desc.log 2>&1 || exit 1
python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py $GRAFT_REPO_ROOT/gpurun_out/pmc_desc/*/*_results.db $GRAFT_REPO_ROOT/gpurun_out/pmc_desc.md > /dev/null || exit 1
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_desc
grep "describe_kernel<4>" $GRAFT_REPO_ROOT/gpurun_out/pmc_desc.md
