#!/usr/bin/env python3
"""Runs one matching algorithm (E / H / N) alone on SRtest150 under c=nobpp a few times (for rocprofv3:
the kernel's time without the concurrent greedy rounds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
from squarna_amd.inputs import ParseDefaultInput

algo = sys.argv[1] if len(sys.argv) > 1 else "E"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
path = os.path.join(os.path.dirname(__file__), "squarna_amd", "data", "datasets", "SRtest150.fas")
recs = list(ParseDefaultInput(path, "qf"))
names, psets = ParseConfig(builtin_config("nobpp"))
ps = [p for p in psets if algo in p["algorithms"]][:1]
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
with Batch(prepared, [ps] * len(prepared), fp32=False) as b:
    for r in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = b.run_algo(list(range(len(prepared))), algo)
        torch.cuda.synchronize()
        print("%s: %.2f ms, %d stems" % (algo, (time.perf_counter() - t0) * 1e3, sum(len(o) for o in out)), flush=True)
