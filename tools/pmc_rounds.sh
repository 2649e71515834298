#!/bin/bash
# PMC counters of sq_rounds_kernel on the S1000 x 1024 fold: bash tools/pmc_rounds.sh OUTDIR [NSEQ] [N]   (on the GPU box)
out=${1:-gpurun_out/pmc_rounds}; nseq=${2:-1024}; n=${3:-1000}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
pass=1
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p$pass -- python3 tools/s1000_probe.py $nseq $n 1 --noprof > $out/p$pass.log 2>&1; pass=$((pass+1)); }
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run FETCH_SIZE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_WAVE32_INSTS
run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
python3 tools/pmc_summary.py $out sq_rounds_kernel
