#!/bin/bash
# the blossom kernel on the critical SRtest150 graph with 2 / 4 / 6 / 8 vertices per scan pass (SQ_MWM_SEGMAX): bash tools/r06_mwm_segmax.sh
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for s in "$@"; do
  SQ_DEFS="-DSQ_MWM_SEGMAX=$s" python -c "from squarna_amd.build import build_library; build_library(force=True)"
  echo "SEGMAX $s: $(python tools/mwm_one.py 217 1 4 2>&1 | tail -2 | tr '\n' ' ')"
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
