#!/bin/bash
# Collects PMC counters for the scan kernel on the S1000 probe (separate --pmc passes, kernel-trace only).
# usage (on the GPU box): bash tools/pmc_scan.sh OUTDIR [NSEQ] [N]
out=${1:-gpurun_out/pmc}; nseq=${2:-256}; n=${3:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p$pass -- python tools/s1000_probe.py $nseq $n 1 > $out/p$pass.log 2>&1; pass=$((pass+1)); }
pass=1
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run FETCH_SIZE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVE_CYCLES
python tools/pmc_summary.py $out
