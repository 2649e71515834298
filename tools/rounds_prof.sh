# the round kernel's per-phase timers (S1000 x 1024) and the three BASELINE shapes: bash tools/rounds_prof.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
o=gpurun_out/s2; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_ROUNDS_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)"
python tools/rounds_probe.py 1000 1024 0 1 2>&1 | grep "^rounds block" | head -12 > $o/rounds_prof.txt
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
{ for n in "1000 1024 0" "300 10000 0" "2000 1000 1"; do python tools/rounds_probe.py $n 5 2>&1 | grep "^rounds\|^launched\|identical"; done; } > $o/rounds_probe.txt
cat $o/rounds_prof.txt $o/rounds_probe.txt
