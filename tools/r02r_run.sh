cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_MWM_PROF2" python -m squarna_amd.build > /tmp/build.log 2>&1 || tail -20 /tmp/build.log
python tools/mwm_one.py 217 4 2 2>&1 | tail -12
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
