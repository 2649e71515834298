#!/bin/bash
# All PMC evidence of a round in one go (on the MI355X box): bash tools/pmc_all.sh TAG
#   S1000 x 1024 probe -> counters of sq_rounds_kernel (the persistent round kernel: one launch per fold)
#   SRtest150 Edmonds probe -> counters of the blossom kernel (sq_mwm_single_kernel for a batch alone);  256 x S1000 fill probe -> counters of sq_fill_kernel
# writes profiles/TAG_{rounds,mwm,fill}_pmc.txt and profiles/traffic.json (stamped with the kernel sources' hash).
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
s=gpurun_out/pmc_${tag}_s1000; e=gpurun_out/pmc_${tag}_mwm; f=gpurun_out/pmc_${tag}_fill
mkdir -p $s $e $f profiles
pass=1
run() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/p$pass -- python3 $probe > $out/p$pass.log 2>&1; pass=$((pass+1)); }
out=$s; probe="tools/s1000_probe.py 1024 1000 1"
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run FETCH_SIZE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
out=$e; probe="tools/algo_probe.py E 3"; pass=1
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run FETCH_SIZE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
out=$f; probe="tools/fill_probe.py"; pass=1
run SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run FETCH_SIZE
run WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $f sq_fill_kernel > profiles/${tag}_fill_pmc.txt
python3 tools/pmc_summary.py $s sq_rounds_kernel > profiles/${tag}_rounds_pmc.txt
python3 tools/pmc_summary.py $e sq_mwm > profiles/${tag}_mwm_pmc.txt    # (sq_mwm_single_kernel: one batch alone; sq_mwm_kernel: batches in flight -- the same code per graph)
python3 tools/make_traffic.py $tag sq_rounds_kernel=$s sq_mwm_kernel:sq_mwm=$e sq_fill_kernel=$f > /dev/null
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_*_pmc.txt profiles/traffic.json gpurun_out/profiles_$tag/
cat profiles/traffic.json
