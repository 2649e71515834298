#!/bin/bash
# diagnostics of round 5 (scratch): alignment sub-batch sizes, step-1 host profile, score kernel phases on pools_long
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag2; mkdir -p $o
for gb in 32 72 140; do echo "== SQ_DENSE_GB=$gb"; SQ_DENSE_GB=$gb python tools/a5000_phases.py 512 5000 2>&1 | grep -v "^\["; done > $o/a5000_dense.txt 2>&1
python - > $o/step1_profile.txt 2>&1 <<'P'
import cProfile, pstats, io, os, sys, random, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import scale_soak as S
from squarna_amd import Predict
rng = random.Random(5000)
with tempfile.NamedTemporaryFile("w", suffix=".afa", delete=False) as f:
    f.write(S.msa(rng, 512, 5000)); path = f.name
Predict(inputfile=path, alignment=True, step3="1", write_to=io.StringIO())
pr = cProfile.Profile(); pr.enable()
Predict(inputfile=path, alignment=True, step3="1", write_to=io.StringIO())
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue())
P
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_SCORE_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
SQ_NO_POOL_ROUND=1 python tools/pools_long_probe.py 500 128 500nobpp 0 > $o/score_prof.txt 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
cat $o/a5000_dense.txt; head -60 $o/step1_profile.txt; grep "score block" $o/score_prof.txt | awk '{n++; a+=$(NF-6); b+=$(NF-4); c+=$(NF-2)} END {print n, "lines; sum setup", a, "phaseA", b, "phaseB", c}'; grep "score block" $o/score_prof.txt | sort -t= -k3 -n | tail -5
