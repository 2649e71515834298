#!/bin/bash
# The round's measurement pass (on the MI355X box): bash tools/r05_final.sh [TAG]  -> gpurun_out/TAG/ (copied into profiles/)
tag=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; mkdir -p $o
# (the counters first: bench.py reads profiles/traffic.json, stamped with the hash of the kernel sources)
bash tools/pmc_all.sh $tag > $o/pmc_all.log 2>&1
# the headline step under the counters, kernel by kernel (separate passes: wave cycles / instruction mix, LDS + waits, bytes read, bytes written): the pool round kernel's own summary; traffic.json then carries all four kernels
pb="bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o/pmc_bench/p1 -- python3 $pb > $o/pmc_bench1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_bench/p2 -- python3 $pb > $o/pmc_bench2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_bench/p3 -- python3 $pb > $o/pmc_bench3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $o/pmc_bench/p4 -- python3 $pb > $o/pmc_bench4.log 2>&1
python3 tools/pmc_bench_agg.py $o/pmc_bench/p1 > $o/${tag}_bench_wave_cycles.txt
python3 tools/pmc_summary.py $o/pmc_bench sq_pool_round_kernel > profiles/${tag}_pool_round_pmc.txt
python3 tools/make_traffic.py $tag sq_rounds_kernel=gpurun_out/pmc_${tag}_s1000 sq_mwm_kernel:sq_mwm=gpurun_out/pmc_${tag}_mwm sq_fill_kernel=gpurun_out/pmc_${tag}_fill sq_pool_round_kernel=$o/pmc_bench > /dev/null
cp profiles/${tag}_*_pmc.txt profiles/traffic.json $o/
python bench.py --steps 20 --warmup 3 > $o/bench.out 2> $o/bench.err; tail -1 $o/bench.out > $o/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $o/stats.log 2>&1
cp $(ls $o/stats/*/*kernel_stats.csv | head -1) $o/${tag}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/s1000 -- python3 tools/s1000_probe.py 1024 1000 5 --noprof > $o/s1000.log 2>&1
cp $(ls $o/s1000/*/*kernel_stats.csv | head -1) $o/${tag}_s1000_kernel_stats.csv
{ for n in "1000 1024 0" "300 10000 0" "2000 1000 1"; do python tools/rounds_probe.py $n 5 2>&1 | grep "^rounds\|^launched\|identical"; done; } > $o/${tag}_rounds_probe.txt
{ for n in "1000 128 0" "2000 125 1"; do python tools/rounds_probe.py $n 7 2>&1 | grep "^rounds\|^launched\|identical"; done; } > $o/${tag}_shard_probe.txt
# launch shapes: threads per structure of the round kernel on a shard, threads per structure of the pooled score kernel
{ for t in 256 512 1024; do echo "SQ_ROUNDS_THREADS=$t S1000 x 128:"; SQ_ROUNDS_THREADS=$t python tools/s1000_probe.py 128 1000 6 --noprof 2>&1 | grep "fold ms"; done
  for t in 512 1024; do echo "SQ_ROUNDS_THREADS=$t S2000 x 125 (+ SHAPE):"; SQ_ROUNDS_THREADS=$t python tools/s1000_probe.py 125 2000 6 --noprof --shape 2>&1 | grep "fold ms"; done
  for t in 64 128 256 512; do echo "SQ_SCORE_POOL_THREADS=$t 500nobpp 500 nt x 500:"; SQ_SCORE_POOL_THREADS=$t python tools/pools_long_probe.py 500 500 500nobpp 1 2>&1 | grep "^fused"; done; } > $o/${tag}_launch_shapes.txt 2>&1
# the greedy loop of one batch alone (no E / H / N job): rounds enqueued ahead of the host / one by one, and its kernel trace
{ for a in 0 3; do echo "SQ_POOL_AHEAD=$a:"; SQ_POOL_AHEAD=$a python tools/greedy_fold.py greedynobpp 8 2>&1 | grep "^fold" | tail -5; done; } > $o/${tag}_greedy_fold.txt
rocprofv3 --kernel-trace --output-format csv -d $o/trg -- python3 tools/greedy_fold.py greedynobpp 4 > $o/greedy_trace.log 2>&1
python tools/trace_all.py $o/trg | tail -48 >> $o/${tag}_greedy_fold.txt
python tools/a5000_phases.py 512 5000 2>&1 | grep -v "^\[" > $o/${tag}_a5000_phases.txt
python tools/pools_long_probe.py 500 2000 500nobpp 2 2>&1 | grep "^fused\|^launched\|identical" > $o/${tag}_pools_long.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/${tag}_gputest.txt
rocprofv3 --kernel-trace --output-format csv -d $o/tr1 -- python3 tools/single_fold.py 6 > $o/single_fold.log 2>&1
{ grep "^fold" $o/single_fold.log; python tools/trace_all.py $o/tr1 | grep -v "sq_state_kernel\|sq_scan6\|sq_score_kernel\|sq_pool_"; } > $o/${tag}_single_fold_trace.txt
python tools/stream_pipe.py 8 12 10 2>&1 | grep "^step" > $o/${tag}_stream_pipe.txt
bash tools/mwm_prof.sh 2>&1 | grep "^mwm\|^record" > $o/${tag}_mwm_phases.txt
rm -rf $o/stats $o/s1000 $o/pmc_bench $o/tr1 $o/trg
cat $o/${tag}_gputest.txt; cat $o/${tag}_rounds_probe.txt $o/${tag}_shard_probe.txt $o/${tag}_launch_shapes.txt; head -12 $o/${tag}_greedy_fold.txt; cat $o/${tag}_a5000_phases.txt $o/${tag}_pools_long.txt; head -12 $o/${tag}_bench_wave_cycles.txt; cat profiles/${tag}_pool_round_pmc.txt
