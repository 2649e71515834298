#!/bin/bash
# PMC counters of one matching kernel on SRtest150: bash tools/pmc_algo.sh E|H|N KERNEL OUTDIR
algo=$1; kern=$2; out=${3:-gpurun_out/pmc_algo}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
pass=1
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p$pass -- python3 tools/algo_probe.py $algo 3 > $out/p$pass.log 2>&1; pass=$((pass+1)); }
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run FETCH_SIZE
python3 tools/pmc_summary.py $out $kern
