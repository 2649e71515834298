#!/usr/bin/env python3
"""HipEngine.fold_records (prepare + fold + result extraction) on random sequences: engine-level sequences/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
nseq, n = int(sys.argv[1]), int(sys.argv[2])
names, psets = ParseConfig(builtin_config("fastest"))
rng = np.random.default_rng(1000)
recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for _ in range(nseq)]
eng = HipEngine()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.fold_records(recs, poollim=1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("threads=%s: %d x %d in %.1f ms -> %.0f seq/s" % (os.environ.get("SQ_ENGINE_THREADS", "2"), nseq, n, dt * 1e3, nseq / dt), flush=True)
