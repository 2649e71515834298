#!/bin/bash
# the list-form pool round kernel under register budgets of 2 / 3 / 4 waves per SIMD: loop time of 500 records of 500 nt
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for d in "$@"; do
  SQ_DEFS="$d" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null
  echo "== $d"; python /tmp/kab.py 500 1000 2>&1 | grep "fold ms\|sha" | tail -4
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
