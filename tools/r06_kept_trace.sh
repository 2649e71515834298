#!/bin/bash
# pools_long on kept lists: per-kernel time of one engine pass over 500 records of 500 nt (rocprofv3 --kernel-trace --stats), and the rounds' timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
[ -n "$1" ] && { cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so; SQ_DEFS="$1" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null; }
rm -rf gpurun_out/kept_trace; mkdir -p gpurun_out/kept_trace
cat > /tmp/kt.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config("500nobpp"))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), 500)), None, None, None, psets, None) for _ in range(int(os.environ.get("CNT", "500")))]
eng = HipEngine()
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = eng.fold_records_packed(recs, poollim=1000); torch.cuda.synchronize()
    print("ms %.1f driver %d peak %d" % ((time.perf_counter() - t0) * 1e3, eng.last_fold_driver, eng.last_fold_peak), flush=True)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kept_trace -o kt -- python3 /tmp/kt.py > gpurun_out/kept_trace/log.txt 2>&1; grep "^ms" gpurun_out/kept_trace/log.txt
[ -n "$1" ] && cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kept_trace/**/kt_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-40s calls %6s total_ms %9.2f avg_us %9.1f" % (r["Name"][:40], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
t = glob.glob("gpurun_out/kept_trace/**/kt_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r["Start_Timestamp"]))
# the last fold: rounds = runs of sq_pool_round_root_kernel between scan kernels
b = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sq_fold_begin")][-1]
seg = rows[b:]
t0 = int(seg[0]["Start_Timestamp"])
rk = [r for r in seg if r["Kernel_Name"].startswith("sq_pool_round_root")]
print("last sub-batch: %d round-kernel launches, busy %.1f ms, first start %.1f ms, last end %.1f ms" % (len(rk), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rk) / 1e6, (int(rk[0]["Start_Timestamp"]) - t0) / 1e6, (int(rk[-1]["End_Timestamp"]) - t0) / 1e6))
for r in rk[::3]:
    g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
    print("   start %8.2f ms dur %8.1f us grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, g // 64))
PY
