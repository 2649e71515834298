"""Pools on sequences beyond 256 nt: fold_records with the launched round kernels (default) and with the one-wave round kernel
over per-job root lists (SQ_POOL_ROOT=1), packed records compared.  usage: pools_root_probe.py [N] [COUNT] [CONFIG] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
cfg = sys.argv[3] if len(sys.argv) > 3 else "500nobpp"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
names, psets = ParseConfig(builtin_config(cfg))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for _ in range(count)]
packs = {}
for mode in ("launched", "root"):
    if mode == "root":
        os.environ["SQ_POOL_ROOT"] = "1"
    else:
        os.environ.pop("SQ_POOL_ROOT", None)
    eng = HipEngine()
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = eng.fold_records_packed(recs, poollim=1000); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    packs[mode] = [o if isinstance(o, bytes) else bytes(o[0]) for o in out]
    print("%s %s N=%d x %d: fold_records ms %s (min %.1f) driver %d peak %d" % (mode, cfg, n, count, " ".join("%.1f" % t for t in ts), min(ts), eng.last_fold_driver, eng.last_fold_peak), flush=True)
same = packs["launched"] == packs["root"]
print("identical:", same)
if not same:
    bad = [k for k in range(count) if packs["launched"][k] != packs["root"][k]]
    print("differ:", len(bad), "first", bad[:5])
