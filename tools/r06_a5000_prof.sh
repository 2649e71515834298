#!/bin/bash
# step 2 of config 5 under the round kernel's in-kernel timers + a kernel trace of the whole alignment (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_a5000; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/a5000_full.py 512 5000 > $out/trace.log 2>&1
cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_ROUNDS_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)"
python3 tools/a5000_full.py 512 5000 > $out/prof.log 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
grep -c "rounds block" $out/prof.log; grep "rounds block" $out/prof.log | head -12; head -12 $out/kernel_stats.csv | cut -c1-200
