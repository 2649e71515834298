#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag5; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_SCORE_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
SQ_NO_POOL_ROUND=1 python tools/pools_long_probe.py 500 256 500nobpp 0 > $o/score_prof.txt 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
grep -c "score block" $o/score_prof.txt
grep "score block" $o/score_prof.txt | awk '{for(i=1;i<=NF;i++){if($i=="setup")a+=$(i+1); if($i=="phaseA")b+=$(i+1); if($i=="phaseB")c+=$(i+1); if($i=="scored")d+=$(i+1); if($i==":492")e+=$(i+1)}} END {print "sum setup", a, "phaseA", b, "phaseB", c, "scored(first wave)", d, "passed492(first wave)", e}'
grep "score block" $o/score_prof.txt | shuf -n 25 --random-source=<(yes)
