#!/bin/bash
# A/B of an environment switch on the headline step, alternating on one box: bash tools/env_ab.sh "SQ_PR_NO_CLASS=1" "SQ_PR_NO_CLASS=0" [rounds] [steps]
cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${3:-4}); do for v in "$1" "$2"; do
  echo "$v: $(env $v python3 bench.py --steps ${4:-12} --warmup 3 --no-cpu --no-stream --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
