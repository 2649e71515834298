#!/bin/bash
# A/B of the pool round kernel's touch-ahead loads on the headline step: bash tools/pr_touch_ab.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/pr_touch; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for v in ${PR_AB_SET:-OFF ON OFF ON}; do
  if [ $v = ON ]; then d="${PR_AB_ON:-}"; else d="${PR_AB_OFF:--DSQ_PR_NO_TOUCH}"; fi
  SQ_DEFS="$d" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
  echo "$v: $(python3 bench.py --steps 6 --warmup 2 --no-cpu --no-stream --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
for v in ${PR_AB_PMC:-OFF ON}; do
  if [ $v = ON ]; then d="${PR_AB_ON:-}"; else d="${PR_AB_OFF:--DSQ_PR_NO_TOUCH}"; fi
  SQ_DEFS="$d" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $o/$v/p1 -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline > /dev/null 2>&1
  echo "$v: $(python3 tools/pmc_bench_agg.py $o/$v/p1 | grep sq_pool_round_kernel)"
  rm -rf $o/$v
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
