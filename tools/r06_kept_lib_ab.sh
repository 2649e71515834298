#!/bin/bash
# A/B of two prebuilt libraries (tools/_libA.so, tools/_libB.so) on pools under the list form (COUNT records of N nt), alternating on one box
cd $GRAFT_REPO_ROOT
bash tools/r06_kept_ab.sh 500 10 > /dev/null 2>&1
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for r in 1 2 3; do for v in A B; do
  cp tools/_lib$v.so squarna_amd/libsquarna_hip.so
  echo "$v: $(python /tmp/kab.py ${1:-500} ${2:-1000} ${3:-500nobpp} 2>&1 | grep "fold ms\|sha" | tail -3 | tr '\n' ' ')"
done; done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
