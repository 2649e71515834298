#!/usr/bin/env python3
"""Randomised alignment-mode soak: Predict(alignment=True) text from the HIP engine vs the same host layer on the
CPU oracle engine, on random small MSAs (mutated copies of a random ancestor with gaps), all step-3 modes."""
import io, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from squarna_amd import Predict, engine as E
from tests.oracle_engine import OracleEngine

nmsa = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for t in range(nmsa):
    nseq, ncol = int(rng.integers(3, 16)), int(rng.integers(30, 140))
    anc = rng.choice(list("ACGU"), ncol)
    rows = []
    for _ in range(nseq):
        row = anc.copy()
        m = rng.random(ncol) < 0.15
        row[m] = rng.choice(list("ACGU"), int(m.sum()))
        row[rng.random(ncol) < 0.08] = "-"
        rows.append("".join(row))
    with tempfile.NamedTemporaryFile("w", suffix=".afa", delete=False) as f:
        for k, r in enumerate(rows):
            f.write(">s%d\n%s\n" % (k, r))
        path = f.name
    kw = dict(inputfile=path, alignment=True, step3="ui12"[t % 4], verbose=bool(t % 3 == 0))
    a, b = io.StringIO(), io.StringIO()
    Predict(write_to=a, **kw)
    with E.use_engine(OracleEngine()):
        Predict(write_to=b, **kw)
    os.unlink(path)
    if a.getvalue() != b.getvalue():
        bad += 1
        print("MISMATCH msa %d (%d x %d, step3=%s)" % (t, nseq, ncol, kw["step3"]), flush=True)
print("%d alignments, %d mismatches" % (nmsa, bad))
sys.exit(1 if bad else 0)
