#!/bin/bash
# the blossom kernel's scan passes split into classification / application / events (cycle counters, -DSQ_MWM_PROF2) for the SEGMAX values given
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for s in "$@"; do
  SQ_DEFS="-DSQ_MWM_PROF2 -DSQ_MWM_SEGMAX=$s" python -c "from squarna_amd.build import build_library; build_library(force=True)"
  echo "SEGMAX $s: $(python tools/mwm_one.py 217 1 2 2>&1 | grep mwm2 | tail -1)"
done
SQ_DEFS="-DSQ_MWM_PROF" python -c "from squarna_amd.build import build_library; build_library(force=True)"
python tools/mwm_one.py 217 1 2 2>&1 | grep "^mwm" | tail -1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
