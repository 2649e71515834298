#!/bin/bash
# pools_long: kept lists (default) against the launched round kernels (SQ_NO_POOL_KEPT=1) and the root-list kernel (SQ_NO_POOL_KEPT=1 SQ_POOL_ROOT=1):
# fold time of 500 records of 500 nt, packed records compared
cd $GRAFT_REPO_ROOT
cat > /tmp/kab.py <<'PY'
import os, sys, time, hashlib
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
n, cnt = int(sys.argv[1]), int(sys.argv[2])
names, psets = ParseConfig(builtin_config(sys.argv[3] if len(sys.argv) > 3 else "500nobpp"))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for _ in range(cnt)]
eng = HipEngine()
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = eng.fold_records_packed(recs, poollim=1000); torch.cuda.synchronize()
    print("fold ms %.1f driver %d peak %d" % ((time.perf_counter() - t0) * 1e3, eng.last_fold_driver, eng.last_fold_peak))
h = hashlib.sha256()
for o in out: h.update(o)
print("sha", h.hexdigest()[:16])
PY
for m in "kept:" "kept_copied:SQ_NO_DETACH=1" "launched:SQ_NO_POOL_KEPT=1"; do
  echo "== ${m%%:*}"; env SQ_TIMING=1 ${m#*:} python /tmp/kab.py ${1:-500} ${2:-500} ${3:-500nobpp} 2>&1 | grep "fold ms\|sha\|rounds=\|kept lists\|total\|E/H/N: begin" | tail -8
done
