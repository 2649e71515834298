#!/bin/bash
# The round's measurement pass (on the MI355X box): bash tools/r05_final.sh [TAG]  -> gpurun_out/TAG/ (copied into profiles/)
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; mkdir -p $o
# (the counters first: bench.py reads profiles/traffic.json, stamped with the hash of the kernel sources)
bash tools/pmc_all.sh $tag > $o/pmc_all.log 2>&1
cp profiles/${tag}_*_pmc.txt profiles/traffic.json $o/
python bench.py --steps 20 --warmup 3 > $o/bench.out 2> $o/bench.err; tail -1 $o/bench.out > $o/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $o/stats.log 2>&1
cp $(ls $o/stats/*/*kernel_stats.csv | head -1) $o/${tag}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/s1000 -- python3 tools/s1000_probe.py 1024 1000 5 --noprof > $o/s1000.log 2>&1
cp $(ls $o/s1000/*/*kernel_stats.csv | head -1) $o/${tag}_s1000_kernel_stats.csv
{ for n in "1000 1024 0" "300 10000 0" "2000 1000 1"; do python tools/rounds_probe.py $n 5 2>&1 | grep "^rounds\|^launched\|identical"; done; } > $o/${tag}_rounds_probe.txt
# the headline step under the counters, kernel by kernel (two passes: wave cycles / instruction mix, LDS + waits); the pool round kernel's own summary
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o/pmc_bench/p1 -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline > $o/pmc_bench1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_bench/p2 -- python3 bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline > $o/pmc_bench2.log 2>&1
python3 tools/pmc_bench_agg.py $o/pmc_bench/p1 > $o/${tag}_bench_wave_cycles.txt
python3 tools/pmc_summary.py $o/pmc_bench sq_pool_round_kernel > $o/${tag}_pool_round_pmc.txt
python tools/a5000_phases.py 512 5000 2>&1 | grep -v "^\[" > $o/${tag}_a5000_phases.txt
python tools/pools_long_probe.py 500 2000 500nobpp 2 2>&1 | grep "^fused\|^launched\|identical" > $o/${tag}_pools_long.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/${tag}_gputest.txt
rocprofv3 --kernel-trace --output-format csv -d $o/tr1 -- python3 tools/single_fold.py 6 > $o/single_fold.log 2>&1
{ grep "^fold" $o/single_fold.log; python tools/trace_all.py $o/tr1 | grep -v "sq_state_kernel\|sq_scan6\|sq_score_kernel\|sq_pool_"; } > $o/${tag}_single_fold_trace.txt
python tools/stream_pipe.py 8 12 10 2>&1 | grep "^step" > $o/${tag}_stream_pipe.txt
bash tools/mwm_prof.sh 2>&1 | grep "^mwm\|^record" > $o/${tag}_mwm_phases.txt
rm -rf $o/stats $o/s1000 $o/pmc_bench $o/tr1
cat $o/${tag}_gputest.txt; cat $o/${tag}_rounds_probe.txt $o/${tag}_a5000_phases.txt $o/${tag}_pools_long.txt; head -12 $o/${tag}_bench_wave_cycles.txt; cat $o/${tag}_pool_round_pmc.txt
