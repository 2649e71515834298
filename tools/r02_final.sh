#!/bin/bash
# Round-end measurement pass (on the MI355X box): bench line, rocprof kernel stats of the same command, strong-scaling
# legs at world size 1, the K sweep, the GPU test log.  Everything lands in gpurun_out/final/ (copy into profiles/).
out=gpurun_out/final; mkdir -p $out
python bench.py --steps 20 --warmup 3 2> $out/bench.err | tail -1 > $out/r02_bench.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu > /tmp/bstats.log 2>&1 )
cp $(find /tmp/bstats -name "*kernel_stats.csv" | head -1) $out/r02_bench_kernel_stats.csv
{ for w in S300 S1000 S2000; do python bench.py --workload $w --steps 10 --warmup 2 2>/dev/null | tail -1; done; } > $out/r02_sharded_world1.txt
{ echo "== one 219-record set per batch"; for k in 1 2 4 8 12; do python tools/concurrent_probe.py $k 20 2>&1 | grep -E "^K=|CPU"; done;
  echo "== two sets per batch (PROBE_REPLICAS=2)"; for k in 1 2 4 8 12; do PROBE_REPLICAS=2 python tools/concurrent_probe.py $k 20 2>&1 | grep -E "^K=|CPU"; done;
  echo "== three sets per batch, lock step and free-running (sq_fold_concurrent_n: no barrier between the steps)"; for k in 4 8 12; do PROBE_REPLICAS=3 PROBE_MAX_STRUCTS=12288 python tools/concurrent_probe.py $k 20 --free 2>&1 | grep -E "^K=|CPU"; done;
  echo "== host-driven rounds (SQ_NO_POOL=1), one set per batch"; for k in 1 8; do SQ_NO_POOL=1 python tools/concurrent_probe.py $k 20 2>&1 | grep -E "^K=|CPU"; done;
  echo "== CPU time per host phase of one fold (SQ_CPUACC=1, one batch alone)"; SQ_CPUACC=1 python tools/concurrent_probe.py 1 4 2>&1 | grep "cpu ms" | tail -1; } > $out/r02_concurrency.txt
python -m pytest tests -m gpu -q 2>&1 | grep -E " passed| failed|error" > $out/r02_gputest.txt
ls -la $out
{ python tools/predict_probe.py S300 3; python tools/predict_probe.py S1000 3; } 2>&1 | grep Predict > $out/r02_predict.txt
cat $out/r02_predict.txt
bash tools/pool_scale_probe.sh > $out/r02_predict_pools.txt 2>&1
python tools/scale_soak.py 2>&1 | grep -vE "amdgpu.ids" > $out/r02_scale_soak.txt
python tools/scale_soak.py ali_5000 very_long_chain very_long_pool 2>&1 | grep -vE "amdgpu.ids|warmup" >> $out/r02_scale_soak.txt
python tools/a5000_full.py 512 5000 2>&1 | grep alignment > $out/r02_a5000_full.txt
cat $out/r02_a5000_full.txt
