#!/bin/bash
# pools_long (500 nt x 1,000, 500nobpp, poollim 1000): per-kernel time of one engine pass (rocprofv3 --kernel-trace --stats)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/pl_trace
cat > /tmp/pl_one.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config("500nobpp"))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), 500)), None, None, None, psets, None) for _ in range(1000)]
eng = HipEngine()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = eng.fold_records_packed(recs, poollim=1000); torch.cuda.synchronize()
    print("ms %.1f driver %d peak %d" % ((time.perf_counter() - t0) * 1e3, eng.last_fold_driver, eng.last_fold_peak), flush=True)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pl_trace -o pl -- python3 /tmp/pl_one.py > gpurun_out/pl_trace/log.txt 2>&1; grep "^ms" gpurun_out/pl_trace/log.txt
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/pl_trace/**/pl_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %6s total_ms %9.2f avg_us %9.1f pct %s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
