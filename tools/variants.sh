#!/bin/bash
# A/B of compile-time kernel variants on the GPU box: bash tools/variants.sh "DEFS1" "DEFS2" ...
cd $GRAFT_REPO_ROOT
for defs in "$@"; do
  echo "=== SQ_DEFS='$defs'"
  SQ_DEFS="$defs" python squarna_amd/build.py > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  if [ -n "$VCMD" ]; then bash -c "$VCMD"; else python tools/s1000_probe.py ${NSEQ:-512} ${NLEN:-1000} 2 $EXTRA 2>&1 | tail -${TAILN:-1}; fi
done
