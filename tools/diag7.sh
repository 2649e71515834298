#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag7; mkdir -p $o
{
for t in 64 128 192 256; do echo "== SQ_SCORE_POOL_THREADS=$t 500nobpp 500x500"; SQ_SCORE_POOL_THREADS=$t python tools/pools_long_probe.py 500 500 500nobpp 2 2>&1 | grep "^fused"; done
for t in 64 128 256 512; do echo "== SQ_SCORE_POOL_THREADS=$t 1000nobpp 1000x128"; SQ_SCORE_POOL_THREADS=$t python tools/pools_long_probe.py 1000 128 1000nobpp 1 2>&1 | grep "^fused"; done
for t in 128 256; do echo "== SQ_SCORE_POOL_THREADS=$t nobpp 300x1000"; SQ_SCORE_POOL_THREADS=$t python tools/pools_long_probe.py 300 1000 nobpp 1 2>&1 | grep "^fused"; done
echo "== S2000x125 / S1000x128 / alignment with the new thread rule"
python tools/s1000_probe.py 125 2000 6 --noprof --shape 2>&1 | grep "fold ms"
python tools/s1000_probe.py 128 1000 6 --noprof 2>&1 | grep "fold ms"
python tools/a5000_phases.py 512 5000 2>&1 | grep "steps 1-3\|step 2"
} > $o/sweep.txt 2>&1
cat $o/sweep.txt
