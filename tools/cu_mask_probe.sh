#!/bin/bash
# the headline step on fewer compute units (ROC_GLOBAL_CU_MASK): does the step scale with the chip?  bash tools/cu_mask_probe.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $2 python3 bench.py --steps 6 --warmup 2 --no-cpu --no-stream --no-roofline 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"; }
run "all 256 CUs" "SQ_X=0"
run "192 CUs" "ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffffffffffffffffffff"
run "128 CUs" "ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff"
run "64 CUs" "ROC_GLOBAL_CU_MASK=0xffffffffffffffff"
run "all 256 CUs" "SQ_X=0"
