#!/usr/bin/env python3
"""The blossom kernel on long random sequences (a build with SQ_DEFS=-DSQ_MWM_PROF prints its phase timers per graph).
usage: mwm_long_probe.py N [COUNT] [CONFIG]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
n = int(sys.argv[1]); cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = sys.argv[3] if len(sys.argv) > 3 else "edmondsnobpp"
names, psets = ParseConfig(builtin_config(cfg))
rng = np.random.default_rng(n)
recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for _ in range(cnt)]
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    HipEngine().fold_records(recs)
    torch.cuda.synchronize()
    print("%s N=%d x %d: %.1f ms" % (cfg, n, cnt, (time.time() - t0) * 1e3), flush=True)
