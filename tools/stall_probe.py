#!/usr/bin/env python3
"""Where do the occasional 40 ms go?  fold / torch.cuda.synchronize timed apart, 12 folds of one SRtest150 batch."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
names, psets = ParseConfig(builtin_config("nobpp"))
recs = load_srtest150()
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
mode = sys.argv[1] if len(sys.argv) > 1 else "sync"
if mode == "nogc": gc.disable()
if mode == "freeze": gc.collect(); gc.freeze()
with Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=4096) as b:
    for r in range(12):
        t0 = time.perf_counter()
        b.fold(poollim=1000)
        t1 = time.perf_counter()
        if mode == "sync": torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("fold %d: fold %.3f ms  sync %.3f ms" % (r, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
