#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag9; mkdir -p $o
run() { for n in "1000 1024 0" "2000 1000 1" "1000 128 0" "2000 125 1" "300 10000 0"; do SQ_NO_LAUNCHED=1 python tools/rounds_probe.py $n 7 2>&1 | grep "^rounds"; done; }
{ echo "== base"; run
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_ABL_NOHSTEMS python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
echo "== no h_stems store"; run
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
} > $o/abl.txt 2>&1
cat $o/abl.txt
