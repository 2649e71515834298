#!/usr/bin/env python3
"""HipEngine.fold_records end to end (Prepared + Batch set-up + fold + results) on a synthetic workload, for
SQ_ENGINE_LANES = 1, 2, 4, 8.  usage: engine_lanes_probe.py S300|S1000 [REPS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
wl = sys.argv[1] if len(sys.argv) > 1 else "S300"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names, psets = ParseConfig(builtin_config("fastest"))
items = bench.synthetic(wl)
recs = [(s, None, None, None, psets, None) for s, line in items]
for lanes in (1, 2, 4, 8):
    os.environ["SQ_ENGINE_LANES"] = str(lanes)
    ts = []
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = HipEngine().fold_records(recs, poollim=1)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s lanes=%d: fold_records %.1f ms (best of %d; all %s)" % (wl, lanes, min(ts), reps, " ".join("%.0f" % t for t in ts)), flush=True)
