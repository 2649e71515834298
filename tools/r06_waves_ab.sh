#!/bin/bash
# the round kernel compiled for 2 / 3 waves per SIMD (256 / 168 VGPRs: fewer or no spills) against the default 4 (128 VGPRs), full batches at 256 threads per block
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for w in 4 3 2; do
  SQ_DEFS="-DSQ_ROUNDS_WAVES=$w" python -c "from squarna_amd.build import build_library; build_library(force=True)"
  for a in "1000 1024 0" "2000 1000 1" "300 10000 0"; do
    echo "waves $w == $a: $(SQ_NO_LAUNCHED=1 python tools/rounds_probe.py $a 7 2>&1 | grep '^rounds' | sed 's/.*(min/(min/')"
  done
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
