cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for w in 2 3; do
SQ_DEFS=-DSQ_ROUNDS_WAVES=$w python -c "from squarna_amd.build import build_library; build_library(force=True)"
echo "== waves_per_eu $w"
for a in "1000 128" "2000 125 1" "1000 512" "1000 1024" "300 1250"; do SQ_NO_LAUNCHED=1 timeout 300 python tools/rounds_probe.py $a 2>&1 | tail -3 | grep -v "launched\|identical"; done
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
echo "== default (4)"
for a in "1000 128" "2000 125 1" "1000 512" "1000 1024" "300 1250"; do SQ_NO_LAUNCHED=1 timeout 300 python tools/rounds_probe.py $a 2>&1 | tail -3 | grep -v "launched\|identical"; done
