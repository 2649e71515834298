#!/bin/bash
# pools_long on kept lists (1,000 records of 500 nt): the blossom kernel's bins beside the pools' rounds -- LDS per bin, graphs fully in LDS or hot part only
cd $GRAFT_REPO_ROOT
bash tools/r06_kept_ab.sh 500 10 > /dev/null 2>&1
for e in "A=1" "SQ_MWM_BIN_BYTES=98304" "SQ_MWM_BIN_BYTES=65536" "SQ_MWM_BIN_BYTES=65536 SQ_MWM_ALL_CAP=1" "SQ_MWM_BIN_BYTES=40960 SQ_MWM_ALL_CAP=1" "SQ_MWM_BIN_BYTES=98304 SQ_MWM_BIN_WAVES=2" "SQ_MWM_DUMP=1"; do
  echo "== $e"; env $e python /tmp/kab.py 500 ${1:-1000} 2>&1 | grep "fold ms\|mwm plan" | tail -4
done
