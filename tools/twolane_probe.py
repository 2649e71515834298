#!/usr/bin/env python3
"""Experiment: two half-batches folded concurrently from two host threads (sq_fold releases the GIL) versus one
batch of the same total size -- an upper bound for what overlapping host bookkeeping with kernels can give."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
nseq, n = int(sys.argv[1]), int(sys.argv[2])
names, psets = ParseConfig(builtin_config("fastest"))
rng = np.random.default_rng(1000)
seqs = ["".join(rng.choice(list("ACGU"), n)) for _ in range(nseq)]
prep = [Prepared(s) for s in seqs]
one = Batch(prep, [psets] * nseq, max_structs=nseq, fp32=False)
halves = [Batch(prep[k::2], [psets] * len(prep[k::2]), max_structs=nseq, fp32=False) for k in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    one.fold(poollim=1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    th = [threading.Thread(target=lambda b=b: b.fold(poollim=1)) for b in halves]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("one batch %.2f ms; two half-batches in two threads %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
