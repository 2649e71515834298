#!/usr/bin/env python3
"""Small driver for profiling: folds NSEQ random sequences of length N (c=fastest pl=1) once or
twice so rocprofv3 sees the scan kernel in its HBM-streaming regime."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
noprof = "--noprof" in sys.argv      # kernel timers off: the fold is free to split its rounds over two lanes
shape = "--shape" in sys.argv        # S2000: reactivity line drawn per position from "_+#" with p = (0.5, 0.3, 0.2)
names, psets = ParseConfig(builtin_config("fastest"))
rng = np.random.default_rng(1000)
seqs = ["".join(rng.choice(list("ACGU"), n)) for _ in range(nseq)]
reacts = ["".join(rng.choice(list("_+#"), n, p=[0.5, 0.3, 0.2])) for _ in range(nseq)] if shape else [None] * nseq
from squarna_amd.dbn import ProcessReacts, ReactDict
prepared = [Prepared(s, ProcessReacts([ReactDict[c] for c in r], M=1.8, B=-0.6) if r else None) for s, r in zip(seqs, reacts)]
with Batch(prepared, [psets] * nseq, max_structs=nseq if noprof else min(nseq, 4096), fp32=False) as b:
    b.profile(not noprof)
    for r in range(reps):
        b.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.fold(poollim=1)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if noprof:
            walls = globals().setdefault("walls", []); walls.append(dt * 1e3)
            if r == reps - 1:
                print("fold ms: min %.2f median %.2f  (all: %s)" % (min(walls), sorted(walls)[len(walls) // 2], " ".join("%.1f" % w for w in walls)))
            continue
        k = 7 if b.fold_paths & 4 else 2
        ms, launches, by = b.profile_get(k)
        print("fold %.2f ms; %s %.3f ms over %d launches, %.1f GB/s algorithmic" % (dt * 1e3, "rounds" if k == 7 else "scan", ms, launches, by / ms / 1e6))
        print("   " + "  ".join("%s %.3f ms/%d" % (nm, *b.profile_get(q)[:2]) for q, nm in ((0, "fill"), (1, "state"), (2, "scan"), (3, "score"), (7, "rounds"))))
