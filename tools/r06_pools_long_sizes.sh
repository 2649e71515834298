#!/bin/bash
# pools_long: the generations' sizes round by round (SQ_TIMING + SQ_POOL_DEBUG) of one sub-batch of 500-nt records
cd $GRAFT_REPO_ROOT
cat > /tmp/pl_sz.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config("500nobpp"))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), 500)), None, None, None, psets, None) for _ in range(int(sys.argv[1]))]
eng = HipEngine()
eng.fold_records_packed(recs, poollim=1000)
PY
SQ_TIMING=1 SQ_POOL_DEBUG=1 python /tmp/pl_sz.py 500 2> /tmp/pl_sz.err
grep -c "^\[pool\] round" /tmp/pl_sz.err
grep "^\[pool\] round" /tmp/pl_sz.err | awk '{gsub(":","",$3); s+=$5; if ($5>m) m=$5; print $3, $5, $7, $9} END {print "sum S", s, "max", m}' | tail -90
grep -v "^\[pool\] round" /tmp/pl_sz.err | head -40
