#!/bin/bash
# A/B of two prebuilt libraries (tools/_libA.so, tools/_libB.so) on the headline step, alternating on one box
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for r in 1 2 3 4 5; do for v in A B; do
  cp tools/_lib$v.so squarna_amd/libsquarna_hip.so
  echo "$v: $(python3 bench.py --steps 12 --warmup 3 --no-cpu --no-stream --no-roofline --no-alignment 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"], "single_batch", d["single_batch"]["ms_per_fold"], d["single_batch"]["best_ms"])')"
done; done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
