"""Device tail against the host tail on the GPU box: SQ_TAIL_CHECK=1 makes every fold run both and compare the packed
records byte for byte (sq_host.hip).  Usage: SQ_TAIL_CHECK=1 python tools/tail_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SQ_TAIL_CHECK", "1")
import numpy as np
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
from squarna_amd.inputs import ParseDefaultInput

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
recs = list(ParseDefaultInput(os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtest150.fas"), "qf"))
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
for cfg, pl in (("nobpp", 1000), ("nobpp", 1), ("fastest", 1), ("alt", 1000), ("greedynobpp", 3)):
    names, psets = ParseConfig(builtin_config(cfg))
    print("==", cfg, "poollim", pl, flush=True)
    with Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=8192) as b:
        b.fold(poollim=pl)
        print("driver", b.fold_driver, flush=True)
        b.limit_results(3)
        b.fold(poollim=pl)
rng = np.random.default_rng(5)
from squarna_amd.dbn import ProcessReacts, ReactDict
names, psets = ParseConfig(builtin_config("fastest"))
for n, cnt, shape in ((300, 64, False), (1000, 16, False), (2000, 8, True)):
    pp = []
    for _ in range(cnt):
        seq = "".join(rng.choice(list("ACGU"), n))
        line = "".join(rng.choice(list("_+#"), n, p=[0.5, 0.3, 0.2])) if shape else None
        pp.append(Prepared(seq, ProcessReacts([ReactDict[c] for c in line], M=1.8, B=-0.6) if line else None))
    print("== synthetic", n, flush=True)
    with Batch(pp, [psets] * cnt, fp32=False, max_structs=cnt) as b:
        b.fold(poollim=1)
        print("driver", b.fold_driver, flush=True)
