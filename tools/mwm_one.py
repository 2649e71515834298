#!/usr/bin/env python3
"""The blossom kernel on ONE SRtest150 graph replicated COUNT times (every wave does the same work: PMC totals / COUNT =
per-graph instruction counts).  usage: mwm_one.py [RECORD=217] [COUNT=64] [REPS=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
rec = int(sys.argv[1]) if len(sys.argv) > 1 else 217
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
recs = load_srtest150()
names, psets = ParseConfig(builtin_config("nobpp"))
ps = [p for p in psets if "E" in p["algorithms"]][:1]
_, seq, reacts, restr, ref = recs[rec]
prepared = [Prepared(seq, reacts, restr, ref) for _ in range(count)]
with Batch(prepared, [ps] * count, fp32=False) as b:
    for r in range(reps):
        b.profile(True); b.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = b.run_algo(list(range(count)), "E")
        torch.cuda.synchronize()
        ms = b.profile_get(4)[0]
        c = b.mwm_counters()
        print("record %d x %d: wall %.2f ms, kernel %.3f ms, passes %d events %d -> %.0f cycles/pass at 2.4 GHz" % (
            rec, count, (time.perf_counter() - t0) * 1e3, ms, c["max_passes"], c["max_events"], ms * 2.4e6 / max(c["max_passes"], 1)), flush=True)
