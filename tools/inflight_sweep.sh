#!/bin/bash
# the headline step over batches in flight x sets per batch: bash tools/inflight_sweep.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
for kr in ${SWEEP:-"8 12" "12 12" "16 12" "8 16" "8 24" "12 8" "16 8" "6 16" "4 24"}; do
  set -- $kr
  echo "inflight $1 x sets $2: $(python3 bench.py --steps 6 --warmup 2 --inflight $1 --replicas $2 --no-cpu --no-stream --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
done
