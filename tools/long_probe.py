#!/usr/bin/env python3
"""Folds NSEQ random sequences of length N under a preset (default 1000nobpp: greedy + N + E + H) and prints timings."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
nseq, n = int(sys.argv[1]), int(sys.argv[2]); cfg = sys.argv[3] if len(sys.argv) > 3 else "1000nobpp"
names, psets = ParseConfig(builtin_config(cfg))
rng = np.random.default_rng(n)
seqs = ["".join(rng.choice(list("ACGU"), n)) for _ in range(nseq)]
with Batch([Prepared(s) for s in seqs], [psets] * nseq, fp32=False) as b:
    b.profile(True)
    for rep in range(2):
        b.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.fold(poollim=1000)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%s %d x %d: fold %.1f ms | " % (cfg, nseq, n, dt * 1e3) + "  ".join("%s %.1f" % (nm, b.profile_get(k)[0]) for k, nm in enumerate(
            ["bits", "state", "scan", "score", "edmonds", "hungarian", "nussinov"])), flush=True)
