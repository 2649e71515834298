python -m pytest tests/test_hip_parity.py tests/test_hip_parity2.py -m gpu -x -q -k "runalgo or text_matches or dropins or bpp or override" 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/mwm_one.py 217 4 3 2>&1 | tail -2
python tools/mwm_one.py 218 4 2 2>&1 | tail -1
for k in 1 4; do python tools/concurrent_probe.py $k 12 2>&1 | tail -1; done
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_MWM_PROF2" python -m squarna_amd.build > /tmp/build.log 2>&1 || tail -20 /tmp/build.log
python tools/mwm_one.py 217 1 2 2>&1 | tail -2
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
