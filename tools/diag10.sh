#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag10; mkdir -p $o
run() { for n in "1000 1024 0" "2000 1000 1" "1000 128 0" "2000 125 1"; do SQ_NO_LAUNCHED=1 python tools/rounds_probe.py $n 9 2>&1 | grep "^rounds" | sed 's/fold ms.*(min/(min/'; done; }
{ echo "== wave walk"; run; run
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_NO_WAVE_WALK python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
echo "== lane walk only"; run; run
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
} > $o/ab.txt 2>&1
cat $o/ab.txt
