#!/usr/bin/env python3
"""Low-complexity sequences of 257-700 nt (GC repeats, long G / C blocks, AU repeats) under pools: thousands of runs tie for the best
finalscore -- the list form against the launched rounds (SQ_NO_POOL_KEPT), packed records compared, fold paths and drivers shown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
cfg = sys.argv[1] if len(sys.argv) > 1 else "500nobpp"
poollim = int(sys.argv[2]) if len(sys.argv) > 2 else 100
names, psets = ParseConfig(builtin_config(cfg))
rng = np.random.default_rng(7)
seqs = ["GC" * 150, "G" * 140 + "AAAA" + "C" * 140, "GGGCCC" * 60, "AU" * 200, "GGGGAAAACCCC" * 30, "GCAU" * 100,
        "".join(rng.choice(list("GC"), 400)), "".join(rng.choice(list("GCU"), 500)), "G" * 300 + "C" * 300, "GU" * 180]
recs = [(s, None, None, None, psets, None) for s in seqs]
out = {}
for mode in ("lists", "launched"):
    if mode == "launched":
        os.environ["SQ_NO_POOL_KEPT"] = "1"
    else:
        os.environ.pop("SQ_NO_POOL_KEPT", None)
    eng = HipEngine()
    t0 = time.perf_counter()
    res = eng.fold_records_packed(recs, poollim=poollim)
    out[mode] = [bytes(o) for o in res]
    print("%s: %.1f s driver %d peak %d capacity retries %d" % (mode, time.perf_counter() - t0, eng.last_fold_driver, eng.last_fold_peak, getattr(eng, "capacity_retries", 0)), flush=True)
print("identical:", out["lists"] == out["launched"], [len(x) for x in out["lists"]])
