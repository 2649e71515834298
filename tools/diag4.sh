#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag4; mkdir -p $o
python tools/pools_long_probe.py 500 2000 500nobpp 2 > $o/pools_long.txt 2>&1
SQ_NO_POOL_ROUND=1 python tools/fuzz_parity.py 400 nobpp > $o/fuzz_launched.txt 2>&1
FUZZ_NMIN=260 FUZZ_NMAX=340 python tools/fuzz_parity.py 48 500nobpp > $o/fuzz_500.txt 2>&1
SQ_NO_POOL_ROUND=1 SQ_NO_ROUNDS=1 FUZZ_POOLLIM=1 python tools/fuzz_parity.py 300 fastest > $o/fuzz_chain.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/gputest.txt
cat $o/pools_long.txt $o/gputest.txt; tail -n 1 $o/fuzz_launched.txt $o/fuzz_500.txt $o/fuzz_chain.txt
