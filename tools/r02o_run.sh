for wl in S300 S1000 S2000; do for sb in 1 2 4 8; do
python bench.py --workload $wl --steps 5 --warmup 2 --sub-batches $sb 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl sub=$sb', d['value'], 'seq/s', d['ms_per_step'], 'ms')"
done; done
python tools/s1000_probe.py 1024 1000 3 2>&1 | tail -2
