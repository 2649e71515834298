#!/bin/bash
# step 2 of config 5 under the round kernel's in-kernel timers (on the GPU box): bash tools/r06_a5000_prof.sh [env assignments]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_a5000; mkdir -p $out
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_ROUNDS_PROF ${XDEFS}" python -c "from squarna_amd.build import build_library; build_library(force=True)"
env "$@" python3 tools/a5000_full.py 512 5000 > $out/prof.log 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
grep -c "rounds block" $out/prof.log; grep "rounds block" $out/prof.log | tail -8; grep alignment $out/prof.log
