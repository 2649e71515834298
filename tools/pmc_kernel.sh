#!/bin/bash
# PMC counters of one kernel on the S1000 probe: bash tools/pmc_kernel.sh KERNEL OUTDIR [NSEQ] [N] [extra probe args]
kern=$1; out=${2:-gpurun_out/pmc_k}; nseq=${3:-512}; n=${4:-1000}; shift 4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
pass=1
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p$pass -- python3 tools/s1000_probe.py $nseq $n 1 $EXTRA > $out/p$pass.log 2>&1; pass=$((pass+1)); }
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run FETCH_SIZE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $out $kern
