#!/bin/bash
# pools_long on the one-wave round kernel over root lists (SQ_POOL_ROOT=1): its own phase timers (SQ_DEFS=-DSQ_PR_PROF), and the fold's time beside the launched form's
cd $GRAFT_REPO_ROOT
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
cat > /tmp/pr.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config(os.environ.get("CFG", "500nobpp")))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), int(os.environ.get("NNT", "500")))), None, None, None, psets, None) for _ in range(int(os.environ.get("CNT", "500")))]
eng = HipEngine()
for _ in range(int(sys.argv[1])):
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.fold_records_packed(recs, poollim=1000); torch.cuda.synchronize()
    print("fold ms %.1f" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
PY
SQ_DEFS="-DSQ_PR_PROF -DSQ_PR_ROOT_WAVES=${PRW:-3}" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null
python /tmp/pr.py 1 > /tmp/pr.out 2>&1
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
python - <<'PY'
import re, numpy as np
rows = []
for ln in open("/tmp/pr.out"):
    m = re.match(r"pool round list (\d+) kept (\d+) walked (\d+) cutruns (\d+) walkserves (\d+) us: walks ([\d.]+) cuts ([\d.]+) alloc ([\d.]+) \| s=(\d+) n=(\d+) nstrand=(\d+) ns=(\d+) nin=(\d+) \| us: bps ([\d.]+) entry ([\d.]+) ext ([\d.]+) extend ([\d.]+) setup ([\d.]+) state ([\d.]+) scan ([\d.]+) score ([\d.]+) choose ([\d.]+) total ([\d.]+)", ln)
    if m: rows.append([float(x) for x in m.groups()])
a = np.array(rows); print(len(a), "structures sampled")
names = "list kept walked cutruns walkserves t_walks t_cuts t_alloc s n nstrand ns nin bps entry ext extend setup state scan score choose total".split()
for k in list(range(8)) + list(range(10, len(names))):
    print("%-8s mean %8.1f p50 %8.1f p90 %8.1f" % (names[k], a[:, k].mean(), *np.percentile(a[:, k], [50, 90])))
PY
