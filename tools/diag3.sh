#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag3; mkdir -p $o
python tools/pools_long_probe.py 500 2000 500nobpp 2 > $o/pools_long.txt 2>&1
python tools/a5000_phases.py 512 5000 2>&1 | grep -v "^\[" > $o/a5000.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $o/gputest.txt
SQ_NO_POOL_ROUND=1 python tools/fuzz_parity.py 400 nobpp > $o/fuzz_launched.txt 2>&1
FUZZ_NMIN=260 FUZZ_NMAX=340 python tools/fuzz_parity.py 48 500nobpp > $o/fuzz_500.txt 2>&1
cat $o/pools_long.txt $o/a5000.txt $o/gputest.txt; tail -3 $o/fuzz_launched.txt $o/fuzz_500.txt
