#!/usr/bin/env python3
"""cProfile of BASELINE config 5 (512 x 5000, all three steps), second call: where the host's time goes.  usage: r06_a5000_cprofile.py [NSEQ] [NCOL]"""
import cProfile, io, os, pstats, random, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import scale_soak as S
from squarna_amd import Predict
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
with tempfile.NamedTemporaryFile("w", suffix=".afa", delete=False) as f:
    f.write(S.msa(random.Random(5000), nseq, ncol))
    path = f.name
Predict(inputfile=path, alignment=True, step3="u", write_to=io.StringIO())
pr = cProfile.Profile()
pr.enable()
Predict(inputfile=path, alignment=True, step3="u", write_to=io.StringIO())
pr.disable()
os.unlink(path)
pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
