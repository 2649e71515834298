#!/bin/bash
# pools_long: the launched score kernel's own timers (SQ_DEFS=-DSQ_SCORE_PROF: every 509th block prints its phases), summed up
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_SCORE_PROF" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null
python /dev/stdin > /tmp/sp.out 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config("500nobpp"))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), 500)), None, None, None, psets, None) for _ in range(500)]
HipEngine().fold_records_packed(recs, poollim=1000)
PY
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
python - <<'PY'
import re, collections
rows = []
for ln in open("/tmp/sp.out"):
    m = re.match(r"score block (\d+): n=(\d+) ncand=(\d+) nstrand=(\d+) threads (\d+) \| first wave: passed :492 (\d+), scored (\d+) in (\d+) groups \| us: setup ([\d.]+) phaseA ([\d.]+) phaseB ([\d.]+) total ([\d.]+)", ln)
    if m: rows.append([float(x) for x in m.groups()])
print(len(rows), "blocks sampled")
import numpy as np
a = np.array(rows)
names = "block n ncand nstrand threads passed492 scored groups setup phaseA phaseB total".split()
for k, nm in enumerate(names):
    if k < 2: continue
    print("%-10s mean %9.1f  p10 %9.1f  p50 %9.1f  p90 %9.1f  max %9.1f" % (nm, a[:, k].mean(), *np.percentile(a[:, k], [10, 50, 90]), a[:, k].max()))
# by nstrand buckets
for lo, hi in ((0, 8), (8, 24), (24, 48), (48, 80), (80, 999)):
    s = a[(a[:, 3] >= lo) & (a[:, 3] < hi)]
    if len(s): print("nstrand %3d-%3d: %5d blocks, ncand %7.0f passed %6.0f scored %6.0f | setup %6.1f A %6.1f B %6.1f total %6.1f us" % (lo, hi, len(s), s[:, 2].mean(), s[:, 5].mean(), s[:, 6].mean(), s[:, 8].mean(), s[:, 9].mean(), s[:, 10].mean(), s[:, 11].mean()))
PY
tail -3 /tmp/sp.out
