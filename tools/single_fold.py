#!/usr/bin/env python3
"""ONE SRtest150 batch (219 records, c=nobpp, poollim 1000) folded REPS times: the single_batch leg alone (for kernel traces).
usage: single_fold.py [REPS=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
names, psets = ParseConfig(builtin_config("nobpp"))
recs = load_srtest150()
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
with Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=4096) as b:
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.fold(poollim=1000)
        torch.cuda.synchronize()
        print("fold %d: %.3f ms" % (r, (time.perf_counter() - t0) * 1e3), flush=True)
