python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "runalgo or text_matches or concurrent or sync_path or two_lanes" 2>&1 | grep -E "passed|failed|rror" | tail -2
for k in 1 2 4 5 6 8 12; do python tools/concurrent_probe.py $k 30 2>&1 | tail -1; done
for wl in S300 S1000; do for sb in 4 8; do
python bench.py --workload $wl --steps 5 --warmup 2 --sub-batches $sb 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl sub=$sb', d['value'], 'seq/s', d['ms_per_step'], 'ms')"
done; done
