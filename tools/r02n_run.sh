python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py --steps 20 --warmup 3 > gpurun_out/r02n_bench.json 2> gpurun_out/r02n_bench.err
tail -3 gpurun_out/r02n_bench.err
cat gpurun_out/r02n_bench.json
python bench.py --workload S300 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -1
python bench.py --workload S2000 --steps 3 --warmup 1 --no-cpu 2>&1 | tail -1
