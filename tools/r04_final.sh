#!/bin/bash
# The round's measurement pass (on the MI355X box): bash tools/r04_final.sh  -> gpurun_out/r04/ (copied into profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r04; mkdir -p $o
# (the counters first: bench.py reads profiles/traffic.json, stamped with the hash of the kernel sources)
bash tools/pmc_all.sh r04 > $o/pmc_all.log 2>&1
cp profiles/r04_*_pmc.txt profiles/traffic.json $o/
python bench.py --steps 20 --warmup 3 > $o/bench.out 2> $o/bench.err; tail -1 $o/bench.out > $o/r04_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $o/stats.log 2>&1
cp $(ls $o/stats/*/*kernel_stats.csv | head -1) $o/r04_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/s1000 -- python3 tools/s1000_probe.py 1024 1000 5 --noprof > $o/s1000.log 2>&1
cp $(ls $o/s1000/*/*kernel_stats.csv | head -1) $o/r04_s1000_kernel_stats.csv
{ for n in "1000 1024 0" "300 10000 0" "2000 1000 1"; do python tools/rounds_probe.py $n 5 2>&1 | grep "^rounds\|^launched\|identical"; done; } > $o/r04_rounds_probe.txt
python tools/a5000_full.py 512 5000 2>&1 | grep "alignment, steps" > $o/r04_a5000_full.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $o/pmc_bench -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline > $o/pmc_bench.log 2>&1
python3 tools/pmc_bench_agg.py $o/pmc_bench > $o/r04_bench_wave_cycles.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/r04_gputest.txt
rocprofv3 --kernel-trace --output-format csv -d $o/tr1 -- python3 tools/single_fold.py 6 > $o/single_fold.log 2>&1
{ grep "^fold" $o/single_fold.log; python tools/trace_all.py $o/tr1 | grep -v "sq_state_kernel\|sq_scan6\|sq_score_kernel\|sq_pool_"; } > $o/r04_single_fold_trace.txt
python tools/stream_pipe.py 8 12 10 2>&1 | grep "^step" > $o/r04_stream_pipe.txt
bash tools/mwm_prof.sh 2>&1 | grep "^mwm\|^record" > $o/r04_mwm_phases.txt
rm -rf $o/stats $o/s1000 $o/pmc_bench $o/tr1
cat $o/r04_gputest.txt; cat $o/r04_rounds_probe.txt $o/r04_a5000_full.txt; head -12 $o/r04_bench_wave_cycles.txt
