#!/usr/bin/env python3
"""Flake hunt: the same batch folded REPS times through every driver; each result must equal the first (a wrong result
that depends on timing -- a result block read before it had arrived, found this round -- shows up here, not in a seeded
parity run).  usage: repeat_soak.py [REPS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SCEN = [
    # name, config, records, nmin, nmax, fold options, environment
    ("chained rounds", "fastest", 96, 30, 300, dict(poollim=1), {}),
    ("device pools + E/H/N", "nobpp", 48, 30, 200, dict(poollim=1000), {}),
    ("device pools, narrow", "alt", 64, 30, 200, dict(poollim=5), {}),
    ("host loop + E/H/N", "nobpp", 48, 30, 200, dict(poollim=1000), {"SQ_NO_POOL": "1"}),
    ("host loop, width 1", "fastest", 96, 30, 300, dict(poollim=1), {"SQ_NO_CHAIN": "1"}),
    ("Edmonds only", "edmondsnobpp", 48, 40, 220, dict(poollim=1000), {}),
    ("Hungarian + Nussinov", "hungariannobpp", 48, 40, 220, dict(algos=frozenset("HN")), {}),
]
bad = 0
for name, cfg, cnt, nmin, nmax, kw, env in SCEN:
    os.environ["FUZZ_NMIN"], os.environ["FUZZ_NMAX"] = str(nmin), str(nmax)
    names, psets = ParseConfig(builtin_config(cfg))
    recs = [(s, r, x, None, psets, None) for s, r, x in F.make(cnt, 4242)]
    os.environ.update(env)
    try:
        t0 = time.time()
        first = HipEngine().fold_records(recs, **kw)
        diff = 0
        for rep in range(reps):
            again = HipEngine().fold_records(recs, **kw)
            if [a[:2] for a in again] != [f[:2] for f in first]:
                diff += 1
        print("%-24s %d records x %d folds: %d differ from the first (%.1f s)" % (name, cnt, reps, diff, time.time() - t0), flush=True)
        bad += diff
    finally:
        for k in env:
            del os.environ[k]
print("flaky folds:", bad)
sys.exit(1 if bad else 0)
