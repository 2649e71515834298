#!/bin/bash
# headline step under single env knobs, the default in between (one box)
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 python3 bench.py --steps 12 --warmup 3 --no-cpu --no-stream --no-roofline --no-alignment $2 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"; }
run A=1
for k in "SQ_POOL_ROUND_NSURV=32" "SQ_POOL_ROUND_NSURV=96" "SQ_MWM_BIN_WAVES=2" "SQ_MWM_BIN_WAVES=8" "SQ_MWM_BIN_BYTES=32768" "SQ_MWM_BIN_BYTES=65536" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=24" "SQ_HOST_THREADS=8" "SQ_HOST_THREADS=32"; do run "$k"; run A=1; done
run A=1 "--inflight 6"; run A=1 "--inflight 10"; run A=1 "--replicas 8"; run A=1 "--replicas 16"; run A=1
