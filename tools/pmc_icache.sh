#!/bin/bash
# instruction-cache counters of the round kernels: bash tools/pmc_icache.sh  (on the GPU box) -> gpurun_out/pmc_icache/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pmc_icache; mkdir -p $o
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $o/s1000/p1 -- python3 tools/s1000_probe.py 1024 1000 1 --noprof > $o/s1000.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $o/bench/p1 -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline > $o/bench.log 2>&1
python3 tools/pmc_summary.py $o/s1000 sq_rounds_kernel | grep total
python3 tools/pmc_summary.py $o/bench sq_pool_round_kernel | grep total
python3 tools/pmc_summary.py $o/bench sq_mwm_kernel | grep total
python3 tools/pmc_summary.py $o/bench sq_lsap_kernel | grep total
python3 tools/pmc_summary.py $o/bench sq_tail_rank_kernel | grep total
