#!/bin/bash
# counters of the scoring kernel inside alignment step 2 (96 x 5000 probe): bash tools/pmc_a5000.sh TAG
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_${tag}_a5000; mkdir -p $out
pass=1
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p$pass -- python3 tools/a5000_full.py 96 5000 > $out/p$pass.log 2>&1; pass=$((pass+1)); }
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run FETCH_SIZE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $out sq_score_kernel > $out/score.txt
python3 tools/pmc_summary.py $out sq_scan6_kernel > $out/scan.txt
cat $out/score.txt
