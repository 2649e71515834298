cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
for c in 1 3 4; do
SQ_DEFS=-DSQ_ROUNDS_CHUNK=$c python -c "from squarna_amd.build import build_library; build_library(force=True)"
echo "CHUNK $c"; for n in "1000 1024 0" "300 10000 0" "2000 1000 1"; do python tools/rounds_probe.py $n 5 2>&1 | grep "^rounds"; done
done
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
