python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02c_gputest.log
cat gpurun_out/r02c_gputest.log
for k in 1 2 4 8 12; do python tools/concurrent_probe.py $k 10; done > gpurun_out/r02c_concurrent.log 2>&1
cat gpurun_out/r02c_concurrent.log
SQ_MWM_CLASSES=1 python tools/concurrent_probe.py 1 10 2>&1 | tail -1
SQ_MWM_CLASSES=1 python tools/concurrent_probe.py 4 10 2>&1 | tail -1
SQ_TIMING=1 python tools/concurrent_probe.py 4 2 > gpurun_out/r02c_concurrent_timing.log 2>&1
python tools/algo_probe.py E 4 2>&1 | tail -3
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_MWM_PROF" python -m squarna_amd.build > /dev/null 2>&1
python tools/algo_probe.py E 2 > gpurun_out/r02c_mwm_prof.log 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
tail -6 gpurun_out/r02c_mwm_prof.log
