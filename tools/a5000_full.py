#!/usr/bin/env python3
"""BASELINE config 5 end to end: Predict(alignment=True) on a synthetic NSEQ x NCOL alignment, all three steps.
usage: a5000_full.py [NSEQ] [NCOL]"""
import hashlib, io, os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import scale_soak as S
from squarna_amd import Predict
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = random.Random(5000)
with tempfile.NamedTemporaryFile("w", suffix=".afa", delete=False) as f:
    f.write(S.msa(rng, nseq, ncol))
    path = f.name
for rep in range(2):
    buf = io.StringIO()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Predict(inputfile=path, alignment=True, step3="u", write_to=buf)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d x %d alignment, steps 1-3: %.2f s  (%d chars, sha256 %s)  peak device memory %.1f GB" % (
        nseq, ncol, dt, len(buf.getvalue()), hashlib.sha256(buf.getvalue().encode()).hexdigest()[:16],
        torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
os.unlink(path)
