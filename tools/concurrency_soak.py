#!/usr/bin/env python3
"""Thread-safety soak: K batches with DIFFERENT random records folded concurrently (sq_fold_concurrent) many times; every
batch's results must equal what the same batch gives when folded alone.  usage: concurrency_soak.py [K=6] [REPS=15] [CONFIG=nobpp] [POOLLIM=1000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util
import torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared, fold_concurrently
from squarna_amd.dbn import ProcessReacts

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
cfg = sys.argv[3] if len(sys.argv) > 3 else "nobpp"
pl = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
names, psets = ParseConfig(builtin_config(cfg))
batches, alone = [], []
for k in range(K):
    recs = fz.make(120 + 37 * k, 1000 + k)
    prepared = [Prepared(s, r, x, None) for s, r, x in recs]
    with torch.cuda.stream(torch.cuda.Stream()):
        b = Batch(prepared, [psets] * len(prepared), fp32=False)
    b.fold(poollim=pl)
    alone.append([repr(b.result(q)) for q in range(b.nseq)])
    batches.append(b)
bad = 0
for r in range(reps):
    fold_concurrently(batches, poollim=pl)
    for k, b in enumerate(batches):
        got = [repr(b.result(q)) for q in range(b.nseq)]
        if got != alone[k]:
            bad += 1
            print("MISMATCH rep %d batch %d" % (r, k), flush=True)
print("%d batches x %d concurrent folds (config %s, poollim %d): %d mismatches" % (K, reps, cfg, pl, bad))
for b in batches:
    b.close()
sys.exit(1 if bad else 0)
