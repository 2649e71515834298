"""pools_long: COUNT random-ACGU sequences of N nt under a config with branching pools (500nobpp: the reference's own
default for N >= 500), poollim 1000, through the engine (sub-batches sized to the device pools' slots) -- fold times with the
list form of the pool round kernel (every structure reads the list its parent left: the default from round 6) and with the
launched round kernels (SQ_NO_POOL_KEPT), packed records compared.
usage: pools_long_probe.py [N] [COUNT] [CONFIG] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cfg = sys.argv[3] if len(sys.argv) > 3 else "500nobpp"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
names, psets = ParseConfig(builtin_config(cfg))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for _ in range(count)]
packs = {}
for mode in ("lists", "launched"):
    if mode == "launched":
        os.environ["SQ_NO_POOL_KEPT"] = "1"
    else:
        os.environ.pop("SQ_NO_POOL_KEPT", None)
    eng = HipEngine()
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = eng.fold_records_packed(recs, poollim=1000); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    packs[mode] = [bytes(o) for o in out]
    print("%s %s N=%d x %d: fold_records ms %s (min %.1f) driver %d peak %d" % (mode, cfg, n, count, " ".join("%.1f" % t for t in ts), min(ts), eng.last_fold_driver, eng.last_fold_peak), flush=True)
print("identical:", packs["lists"] == packs["launched"])
