#!/usr/bin/env python3
"""SURVEY 8d's check, run in the BUILD CONTAINER only (the reference does not travel): the oracle in the reference's algorithmic
form (oracle/sqrn_pyform.py) against the imported reference itself -- same structures, and its speed within 15 % of the
reference's on a stated sample of SRtest150 (single process, mp=False).  The result is recorded in BASELINE.md.
usage: python tests/golden/check_reference_form.py [STRIDE=9]"""
import os, sys, time
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sqrn_oracle as O, sqrn_pyform as P
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.inputs import ParseDefaultInput
REF = "/root/reference/src/SQUARNA"
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 9
names, psets = ParseConfig(builtin_config("nobpp"))
recs = list(ParseDefaultInput(os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtest150.fas"), "qf"))[::stride]
sys.path.insert(0, REF)
import SQRNdbnseq as R, SQUARNA as RS                      # noqa: E402  (the imported reference)
rnames, rpsets = RS.ParseConfig(os.path.join(REF, "nobpp.conf"))


def run(fn):
    t0 = time.perf_counter()
    out = [fn(r) for r in recs]
    return time.perf_counter() - t0, out


best = {}
for rep in range(3):
    tc, a = run(lambda r: O.SQRNdbnseq(r[1], r[2], r[3], r[4], psets))
    P.install(True)
    try:
        tp, b = run(lambda r: O.SQRNdbnseq(r[1], r[2], r[3], r[4], psets))
    finally:
        P.install(False)
    tr, c = run(lambda r: R.SQRNdbnseq(r[1], r[2], r[3], r[4], rpsets, mp=False))
    for k, v in (("port", tc), ("form", tp), ("reference", tr)):
        best[k] = min(best.get(k, 1e9), v)
    assert all(repr(x) == repr(y) for x, y in zip(a, b)), "the two forms of the oracle differ"
    assert all(x[0] == z[0] and [p[0] for p in x[1]] == [p[0] for p in z[1]] for x, z in zip(b, c)), "oracle != reference"
ratio = best["form"] / best["reference"]
print("%d SRtest150 records (every %dth), c=nobpp, one process, best of 3: C port %.2f s | oracle in the reference's form %.2f s | imported "
      "reference (mp=False) %.2f s | form / reference %.3f (%s the +-15 %% of SURVEY 8d) | port is %.1f x the reference per core" % (
          len(recs), stride, best["port"], best["form"], best["reference"], ratio, "within" if 0.85 <= ratio <= 1.15 else "OUTSIDE",
          best["reference"] / best["port"]))
