#!/bin/bash
# wave cycles of the matching jobs' finish kernel over three headline steps: bash tools/finish_pmc.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/finish_pmc; mkdir -p $o
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $o/p1 -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline > /dev/null 2>&1
python3 tools/pmc_bench_agg.py $o/p1 | grep -E "^kernel|sq_algo_finish|sq_algo_edges|sq_algo_sizes|sq_pool_round_kernel|sq_state_scan|sq_scan6|sq_score_kernel|sq_bits_masks"
rm -rf $o
