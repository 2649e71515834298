#!/usr/bin/env python3
"""K batches folded concurrently (sq_fold_concurrent) versus one batch of the same total size.
usage: twolane_probe.py NSEQ N [K ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared, fold_concurrently
nseq, n = int(sys.argv[1]), int(sys.argv[2])
ks = [int(x) for x in sys.argv[3:]] or [2]
names, psets = ParseConfig(builtin_config("fastest"))
rng = np.random.default_rng(1000)
seqs = ["".join(rng.choice(list("ACGU"), n)) for _ in range(nseq)]
prep = [Prepared(s) for s in seqs]
one = Batch(prep, [psets] * nseq, max_structs=nseq, fp32=False)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    one.fold(poollim=1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("one batch of %d: %.2f ms" % (nseq, (t1 - t0) * 1e3), flush=True)
one.close()
for K in ks:
    parts = [Batch(prep[k::K], [psets] * len(prep[k::K]), max_structs=nseq, fp32=False) for k in range(K)]
    for rep in range(3):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        fold_concurrently(parts, poollim=1)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%d batches of %d concurrently: %.2f ms" % (K, nseq // K, (t2 - t1) * 1e3), flush=True)
    for b in parts:
        b.close()
