# the round kernel's per-phase timers on the shapes given: bash tools/rounds_prof2.sh "1000 128 0" "2000 125 1" ...  (on the GPU box)
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_ROUNDS_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)"
for a in "$@"; do echo "== $a"; SQ_NO_LAUNCHED=1 python tools/rounds_probe.py $a 1 2>&1 | grep "^rounds block.*n=" | head -${PROF_LINES:-4}; done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
