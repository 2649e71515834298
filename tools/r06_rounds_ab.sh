#!/bin/bash
# the round kernel's shapes (full batches and an eighth of each) under the settings given as arguments ("" = defaults), on one box:
#   bash tools/r06_rounds_ab.sh "" "SQ_COOP_MIN=0" "SQ_COOP_NUM=2 SQ_COOP_FIRST=2"
cd $GRAFT_REPO_ROOT
for a in "1000 1024 0" "1000 128 0" "2000 1000 1" "2000 125 1" "300 10000 0"; do
  for e in "$@"; do
    echo "== $a  ${e:-defaults}: $(env $e SQ_NO_LAUNCHED=1 python tools/rounds_probe.py $a 7 2>&1 | grep '^rounds' | sed 's/.*(min/(min/')"
  done
done
for e in "$@"; do echo "== A5000 ${e:-defaults}: $(env $e python tools/a5000_full.py 2>&1 | tail -1)"; done
