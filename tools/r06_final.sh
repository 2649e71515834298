#!/bin/bash
# The round's measurement pass (on the MI355X box): bash tools/r06_final.sh [TAG]  -> gpurun_out/TAG/ (copied into profiles/)
# Counters first (bench.py reads profiles/traffic.json, stamped with the hash of the kernel sources), then the bench line, the
# kernel statistics of the same command, the probes of DESIGN.md's tables and the GPU tests.
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; mkdir -p $o
bash tools/pmc_all.sh $tag > $o/pmc_all.log 2>&1
# config 5 (512 x 5000 alignment, all three steps) under the counters: the 4,700-nt launch of the round kernel, the scatter, the select
a=gpurun_out/pmc_${tag}_a5000; mkdir -p $a
pa() { rocprofv3 --kernel-trace --pmc "${@:2}" --output-format csv -d $a/p$1 -- python3 tools/a5000_full.py 512 5000 > $a/p$1.log 2>&1; }
pa 1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
pa 2 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pa 3 FETCH_SIZE
pa 4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
{ for k in sq_rounds_kernel sq_scatter_all_kernel sq_colselect_kernel; do echo "== $k"; python3 tools/pmc_summary.py $a $k; done; } > profiles/${tag}_a5000_pmc.txt
# the headline step under the counters, kernel by kernel (separate passes)
pb="bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline --no-alignment"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $o/pmc_bench/p1 -- python3 $pb > $o/pmc_bench1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $o/pmc_bench/p2 -- python3 $pb > $o/pmc_bench2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $o/pmc_bench/p3 -- python3 $pb > $o/pmc_bench3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $o/pmc_bench/p4 -- python3 $pb > $o/pmc_bench4.log 2>&1
python3 tools/pmc_bench_agg.py $o/pmc_bench/p1 > $o/${tag}_bench_wave_cycles.txt
python3 tools/pmc_summary.py $o/pmc_bench sq_pool_round_kernel > profiles/${tag}_pool_round_pmc.txt
bash tools/r06_pools_long_pmc.sh > $o/pools_long_pmc.log 2>&1      # (the list form of the pool round kernel, 500 records of 500 nt: gpurun_out/pmc_r06_pl)
python3 tools/pmc_summary.py gpurun_out/pmc_r06_pl sq_pool_round_root_kernel > profiles/${tag}_pool_round_root_pmc.txt
python3 tools/make_traffic.py $tag sq_pool_round_root_kernel=gpurun_out/pmc_r06_pl sq_rounds_kernel=gpurun_out/pmc_${tag}_s1000 sq_mwm_kernel:sq_mwm=gpurun_out/pmc_${tag}_mwm sq_fill_kernel=gpurun_out/pmc_${tag}_fill sq_pool_round_kernel=$o/pmc_bench \
    a5000_rounds:sq_rounds_kernel=$a a5000_scatter:sq_scatter_all_kernel=$a a5000_colselect:sq_colselect_kernel=$a > /dev/null
cp profiles/${tag}_*_pmc.txt profiles/traffic.json $o/
python bench.py --steps 20 --warmup 3 > $o/bench.out 2> $o/bench.err; tail -1 $o/bench.out > $o/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu > $o/stats.log 2>&1
cp $(ls $o/stats/*/*kernel_stats.csv | head -1) $o/${tag}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/s1000 -- python3 tools/s1000_probe.py 1024 1000 5 --noprof > $o/s1000.log 2>&1
cp $(ls $o/s1000/*/*kernel_stats.csv | head -1) $o/${tag}_s1000_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $o/a5000 -- python3 tools/a5000_full.py 512 5000 > $o/a5000.log 2>&1
cp $(ls $o/a5000/*/*kernel_stats.csv | head -1) $o/${tag}_a5000_kernel_stats.csv
{ for n in "1000 1024 0" "300 10000 0" "2000 1000 1"; do python tools/rounds_probe.py $n 5 2>&1 | grep "^rounds\|^launched\|identical"; done; } > $o/${tag}_rounds_probe.txt
{ for n in "1000 128 0" "2000 125 1"; do python tools/rounds_probe.py $n 7 2>&1 | grep "^rounds\|^launched\|identical"; done; } > $o/${tag}_shard_probe.txt
python tools/a5000_phases.py 512 5000 2>&1 | grep -v "^\[" > $o/${tag}_a5000_phases.txt
{ bash tools/r06_a5000_prof.sh X=1 | grep "rounds block" | head -12; } > $o/${tag}_a5000_round_timers.txt 2>&1
python tools/pools_long_probe.py 500 2000 500nobpp 2 2>&1 | grep "^lists\|^launched\|identical" > $o/${tag}_pools_long.txt
# the list form of the pool round kernel: A/B against the launched rounds at five shapes, its counters, a fold's kernels and rounds
{ for sh in "500 1000 500nobpp" "300 2000 nobpp" "400 1000 alt" "800 400 greedynobpp" "1000 300 1000nobpp"; do echo "#### $sh"; bash tools/r06_kept_ab.sh $sh 2>&1 | grep "^==\|fold ms\|sha" | awk '/^==/{c=0} {c++; if (c==1 || c>=5) print}'; done
  echo "#### counters, 500 records of 500 nt (rocprofv3 --pmc, separate passes)"; grep -A3 "== sq_pool_round_root_kernel" $o/pools_long_pmc.log | cut -c1-600
  echo "#### kernels of 1,000 records of 500 nt (rocprofv3 --kernel-trace --stats, three calls) and the rounds of the last fold"; CNT=1000 bash tools/r06_kept_trace.sh 2>&1 | head -64; } > $o/${tag}_pools_long_kept.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $o/${tag}_gputest.txt
rocprofv3 --kernel-trace --output-format csv -d $o/tr1 -- python3 tools/single_fold.py 6 > $o/single_fold.log 2>&1
{ grep "^fold" $o/single_fold.log; python tools/trace_all.py $o/tr1 | grep -v "sq_state_kernel\|sq_scan6\|sq_score_kernel\|sq_pool_"; } > $o/${tag}_single_fold_trace.txt
python tools/stream_pipe.py 8 12 10 2>&1 | grep "^step" > $o/${tag}_stream_pipe.txt
bash tools/mwm_prof.sh 2>&1 | grep "^mwm\|^record" > $o/${tag}_mwm_phases.txt
rm -rf $o/stats $o/s1000 $o/a5000 $o/pmc_bench $o/tr1
cat $o/${tag}_gputest.txt; cat $o/${tag}_rounds_probe.txt $o/${tag}_shard_probe.txt; cat $o/${tag}_a5000_phases.txt $o/${tag}_pools_long.txt; head -12 $o/${tag}_bench_wave_cycles.txt; cat profiles/${tag}_a5000_pmc.txt | cut -c1-400
