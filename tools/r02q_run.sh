cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/mwm_one.py 217 64 3
out=gpurun_out/pmc_mwm_one; mkdir -p $out
pass=1
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p$pass -- python3 tools/mwm_one.py 217 64 1 > $out/p$pass.log 2>&1; pass=$((pass+1)); }
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_ACTIVE_INST_FLAT SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU
python3 tools/pmc_summary.py $out sq_mwm_kernel
