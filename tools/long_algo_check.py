#!/usr/bin/env python3
"""E/H/N on long sequences (the 500nobpp / 1000nobpp presets): GPU fold vs the CPU oracle, with timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import sqrn_oracle as O
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
cfg = sys.argv[1]; n = int(sys.argv[2]); cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
names, psets = ParseConfig(builtin_config(cfg))
rng = np.random.default_rng(n)
for k in range(cnt):
    seq = "".join(rng.choice(list("ACGU"), n))
    t0 = time.time(); g = HipEngine().fold_records([(seq, None, None, None, psets, None)])[0]; tg = time.time() - t0
    t0 = time.time(); e = O.SQRNdbnseq(seq, None, None, None, psets); to = time.time() - t0
    ok = g[0] == e[0] and [x[0] for x in g[1]] == [x[0] for x in e[1]] and [list(x[2]) for x in g[1]] == [list(x[2]) for x in e[1]]
    print("%s N=%d: gpu %.2f s, oracle %.1f s, %s (%d structures)" % (cfg, n, tg, to, "identical" if ok else "MISMATCH", len(g[1])), flush=True)
