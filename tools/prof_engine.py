import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
nseq, n = int(sys.argv[1]), int(sys.argv[2])
names, psets = ParseConfig(builtin_config("fastest"))
rng = np.random.default_rng(1000)
recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for _ in range(nseq)]
eng = HipEngine()
eng.fold_records(recs, poollim=1)
cProfile.run("eng.fold_records(recs, poollim=1)", "/tmp/pe.out")
pstats.Stats("/tmp/pe.out").sort_stats("tottime").print_stats(18)
