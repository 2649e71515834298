#!/bin/bash
# Round-3 measurement pass (on the MI355X box): the bench line, the rocprofv3 kernel stats of the same command, the bench
# under a 4-CPU affinity mask, the strong-scaling legs at world size 1, a sweep of batches in flight x sets per batch, the
# CPU time per host phase of one fold, the GPU test log.  Everything lands in gpurun_out/final/ (copy into profiles/).
out=gpurun_out/final; mkdir -p $out
python bench.py --steps 20 --warmup 3 2> $out/bench.err | tail -1 > $out/r03_bench.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu --no-stream > /tmp/bstats.log 2>&1 )
cp $(find /tmp/bstats -name "*kernel_stats.csv" | head -1) $out/r03_bench_kernel_stats.csv
{ echo "== bench.py --no-cpu --no-roofline --no-stream, all CPUs of the quota"; python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline --no-stream 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'seq/s', d['ms_per_step'], 'ms per step; host', d['host'])";
  echo "== the same under taskset -c 0-3 (4 CPUs)"; taskset -c 0-3 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline --no-stream 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'seq/s', d['ms_per_step'], 'ms per step; host', d['host'])";
  echo "== the same under taskset -c 0-1 (2 CPUs)"; taskset -c 0-1 python bench.py --steps 20 --warmup 3 --no-cpu --no-roofline --no-stream 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'seq/s', d['ms_per_step'], 'ms per step; host', d['host'])"; } > $out/r03_cpus.txt 2>&1
{ for w in S300 S1000 S2000; do python bench.py --workload $w --steps 10 --warmup 2 2>/dev/null | tail -1; done; } > $out/r03_sharded_world1.txt
{ echo "== batches in flight (K) x SRtest150 sets per batch (R): seq/s, ms per step, busy CPUs";
  for kr in "1 1" "1 6" "1 24" "2 12" "4 6" "8 3" "8 6" "8 12" "8 24" "12 12" "16 3"; do set -- $kr; python bench.py --steps 10 --warmup 2 --inflight $1 --replicas $2 --no-cpu --no-roofline --no-stream 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$1 R=$2', d['value'], d['ms_per_step'], d['host']['busy_cpus'])"; done;
  echo "== CPU time per host phase of one fold (SQ_CPUACC=1, one 219-record batch alone; 'caller' and 'end' are mostly the spin-wait on the blossom kernel)"; SQ_CPUACC=1 python tools/concurrent_probe.py 1 6 2>&1 | grep -E "cpu ms|^K=|CPU" | tail -3;
  echo "== the same with 8 batches in flight (relaxed waits: sleeping instead of spinning)"; python tools/concurrent_probe.py 8 10 2>&1 | grep -E "^K=|CPU" | tail -2;
  echo "== one big batch: per-kernel ms (HIP events; kernels of different streams overlap)"; for r in 1 6 24; do python tools/big_batch_probe.py $r 2>&1 | grep -E "^R=|^\{.bits" | tail -2; done; } > $out/r03_concurrency.txt 2>&1
python -m pytest tests -m gpu -q 2>&1 | grep -E " passed| failed|error" > $out/r03_gputest.txt
{ python tools/predict_probe.py S300 3; python tools/predict_probe.py S1000 3; } 2>&1 | grep Predict > $out/r03_predict.txt
cat $out/r03_cpus.txt $out/r03_gputest.txt $out/r03_predict.txt
