# the ranking kernel's phases on SRtest150 (one batch alone): bash tools/tail_prof.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
o=gpurun_out/s2; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_TAIL_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)"
python tools/single_fold.py 2 2>&1 | grep "^tail s=" | tail -12 > $o/tail_prof.txt
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
SQ_TIMING=1 python3 tools/single_fold.py 4 > $o/single_timing_all.txt 2>&1
cat $o/tail_prof.txt
