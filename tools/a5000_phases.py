#!/usr/bin/env python3
"""Where BASELINE config 5 (512 x 5000 alignment, all three steps) spends its time: wall-clock per phase, from timing wrappers
around the alignment's functions (no change to the product code).  usage: a5000_phases.py [NSEQ] [NCOL]"""
import collections, functools, hashlib, io, os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import scale_soak as S
import squarna_amd
from squarna_amd import Predict, align, engine, api
T = collections.OrderedDict()


def timed(mod, name, label):
    f = getattr(mod, name)

    @functools.wraps(f)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            torch.cuda.synchronize(); T[label] = T.get(label, 0.0) + time.perf_counter() - t0
    setattr(mod, name, w)


timed(align, "SQRNdbnali", "step 1: SQRNdbnali (stems of every row + column matrix + first-fit)")
timed(align, "MatrixToDBNs", "  of which MatrixToDBNs (Python first-fit over the selected cells)")

timed(engine.HipEngine, "fold_records", "step 2: fold of every row weighted by the matrix (HipEngine.fold_records)")
timed(align, "Consensus", "Consensus (Python, steps 2 and 3)")
timed(align, "RunSQRNdbnali", "RunSQRNdbnali (all of the alignment mode)")
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rng = random.Random(5000)
with tempfile.NamedTemporaryFile("w", suffix=".afa", delete=False) as f:
    f.write(S.msa(rng, nseq, ncol))
    path = f.name
for rep in range(2):
    T.clear()
    buf = io.StringIO()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Predict(inputfile=path, alignment=True, step3="u", write_to=buf)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d x %d alignment, steps 1-3: %.2f s  (sha256 %s)" % (nseq, ncol, dt, hashlib.sha256(buf.getvalue().encode()).hexdigest()[:16]), flush=True)
    for k, v in T.items():
        print("   %7.3f s  %s" % (v, k))
    print("   %7.3f s  everything else (parsing the alignment, Prepared records, text)" % (dt - T.get("RunSQRNdbnali (all of the alignment mode)", 0.0)))
os.unlink(path)
