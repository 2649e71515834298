#!/bin/bash
# Vector instructions of the pool round kernel's phases on the headline step, by running a phase twice (the phases are idempotent)
# and counting SQ_INSTS_VALU: bash tools/pr_dup.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pr_dup; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for v in ${PR_DUP_SET:-BASE SETUP STATE SCAN BPS SCORE CHOOSE}; do
  if [ $v = BASE ]; then d=""; else d="-DSQ_PR_DUP_$v"; fi
  SQ_DEFS="$d" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $o/$v/p1 -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline > /dev/null 2>&1
  echo "$v: $(python3 tools/pmc_bench_agg.py $o/$v/p1 | grep sq_pool_round_kernel)"
  rm -rf $o/$v
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
