python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/r02_gputest.log; cat gpurun_out/r02_gputest.log
bash tools/pmc_all.sh r02 > gpurun_out/r02_pmc_all.log 2>&1; tail -50 gpurun_out/r02_pmc_all.log
cp profiles/traffic.json gpurun_out/profiles_r02/ 2>/dev/null
python bench.py --steps 20 --warmup 3 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -2 gpurun_out/r02_bench.err; cat gpurun_out/r02_bench.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_bench_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/r02_bench_prof.json 2> gpurun_out/r02_bench_prof.err
ls gpurun_out/r02_bench_stats/*/ | head
