#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag6; mkdir -p $o
for t in 256 512 1024; do echo "== SQ_ROUNDS_THREADS=$t S1000x128"; SQ_ROUNDS_THREADS=$t python tools/s1000_probe.py 128 1000 6 --noprof 2>&1 | grep "fold ms"; done > $o/threads.txt
for t in 512 1024; do echo "== SQ_ROUNDS_THREADS=$t S2000x125"; SQ_ROUNDS_THREADS=$t python tools/s1000_probe.py 125 2000 6 --noprof --shape 2>&1 | grep "fold ms"; done >> $o/threads.txt
for t in 128 256 512; do echo "== SQ_SCORE_THREADS=$t pools_long 500"; SQ_NO_POOL_ROUND=1 SQ_SCORE_THREADS=$t python tools/pools_long_probe.py 500 500 500nobpp 1 2>&1 | grep "^fused"; done >> $o/threads.txt
cat $o/threads.txt
