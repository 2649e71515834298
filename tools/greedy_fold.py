#!/usr/bin/env python3
"""ONE SRtest150 batch (219 records) under a config without E / H / N jobs folded REPS times: the greedy loop of the device
pools alone (for kernel traces).  usage: greedy_fold.py [CONFIG=greedynobpp] [REPS=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
cfg = sys.argv[1] if len(sys.argv) > 1 else "greedynobpp"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
names, psets = ParseConfig(builtin_config(cfg))
recs = load_srtest150()
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
with Batch(prepared, [psets] * len(prepared), fp32=False) as b:
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.fold(poollim=1000)
        torch.cuda.synchronize()
        print("fold %d: %.3f ms  paths %d" % (r, (time.perf_counter() - t0) * 1e3, b.fold_paths), flush=True)
