#!/usr/bin/env python3
"""SRtest150 replicated REP times in ONE batch (c=nobpp): fold time per repetition; SQ_TIMING=1 shows the phases."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
rep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
recs = load_srtest150()
names, psets = ParseConfig(builtin_config("nobpp"))
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs] * rep
with Batch(prepared, [psets] * len(prepared), fp32=False) as b:
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.fold(poollim=1000)
        torch.cuda.synchronize()
        print("x%d: %d records, fold %.2f ms" % (rep, len(prepared), (time.perf_counter() - t0) * 1e3), flush=True)
