#!/usr/bin/env python3
"""cProfile of pools_long (COUNT records of N nt, 500nobpp, poollim 1000), second engine pass: where the host's time goes
between the sub-batches.  usage: r06_pools_long_cprofile.py [N] [COUNT]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
count = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
names, psets = ParseConfig(builtin_config("500nobpp"))
rng = np.random.default_rng(500)
recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for _ in range(count)]
eng = HipEngine()
out = eng.fold_records_packed(recs, poollim=1000)
print("packed bytes", sum(len(o[0]) if not isinstance(o, bytes) else len(o) for o in out))
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
out = eng.fold_records_packed(recs, poollim=1000)
pr.disable()
torch.cuda.synchronize(); print("ms %.1f" % ((time.perf_counter() - t0) * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
