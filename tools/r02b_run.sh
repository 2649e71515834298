set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02b_gputest.log
cat gpurun_out/r02b_gputest.log
for k in 1 2 4 8; do python tools/concurrent_probe.py $k 10; done > gpurun_out/r02b_concurrent.log 2>&1
cat gpurun_out/r02b_concurrent.log
SQ_TIMING=1 python tools/concurrent_probe.py 4 2 > gpurun_out/r02b_concurrent_timing.log 2>&1
python tools/s1000_probe.py 1024 1000 3 > gpurun_out/r02b_s1000.log 2>&1
python tools/s1000_probe.py 1024 1000 6 --noprof >> gpurun_out/r02b_s1000.log 2>&1
python tools/s1000_probe.py 1000 2000 2 --shape >> gpurun_out/r02b_s1000.log 2>&1
cat gpurun_out/r02b_s1000.log
bash tools/pmc_kernel.sh sq_score_kernel gpurun_out/pmc_score_r02b 1024 1000 > gpurun_out/r02b_score_pmc.txt 2>&1
cat gpurun_out/r02b_score_pmc.txt
# blossom kernel with its built-in phase timers (profiling build, only on this box)
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_MWM_PROF" python -m squarna_amd.build > /dev/null 2>&1
python tools/algo_probe.py E 2 > gpurun_out/r02b_mwm_prof.log 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
tail -40 gpurun_out/r02b_mwm_prof.log
