#!/bin/bash
# diagnostics of round 5 (scratch): alignment fold phases, S1000 x 128 shard, pools_long kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag1; mkdir -p $o
SQ_TIMING=1 python tools/a5000_phases.py 512 5000 > $o/a5000_timing.txt 2>&1
SQ_TIMING=1 python tools/s1000_probe.py 128 1000 4 --noprof > $o/s1000x128_timing.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $o/tr128 -- python3 tools/s1000_probe.py 128 1000 3 --noprof > $o/tr128.log 2>&1
python tools/trace_all.py $o/tr128 > $o/s1000x128_trace.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/pl -- python3 tools/pools_long_probe.py 500 500 500nobpp 1 > $o/pl.log 2>&1
cp $(ls $o/pl/*/*kernel_stats.csv | head -1) $o/pools_long_kernel_stats.csv
rm -rf $o/tr128 $o/pl
tail -40 $o/s1000x128_timing.txt; head -30 $o/pools_long_kernel_stats.csv
