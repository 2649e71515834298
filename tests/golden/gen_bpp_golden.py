#!/usr/bin/env python3
"""Golden vectors for the bpp != 0 paramsets (SQRNdbnseq.py:341-364).

ViennaRNA is absent from this image, and the reference only needs *a module named RNA*.  This script
installs tests/fake_rna.py (a deterministic stand-in, see its header) as ``RNA`` and runs the REAL
reference on top of it: BPMatrix with bpp_power != 0, whole SQRNdbnseq folds under def.conf (12
paramsets, 7 of them with bpp, all four algorithms), and Predict() with the DEFAULT configuration
(def.conf + priority paramsets bppN,bppH1,bppH2, SQUARNA.py:683-703).  What the fixtures pin is the
application of the probabilities and everything downstream of it; the probabilities stay unpinned.

Runs only in the build container (/root/reference); outputs are data:
  tests/golden/bpp.json              BPMatrix cells, SQRNdbnseq tuples, the ViennaRNA call sequences
  tests/golden/text/*_def.txt        Predict texts (+ entries in digests.json)

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_bpp_golden.py
"""
import hashlib
import io
import json
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/SQUARNA"
sys.path.insert(0, REF)

from tests import fake_rna          # noqa: E402
fake_rna.install()                  # BEFORE the reference runs: its `import RNA` finds the stand-in

import numpy as np                  # noqa: E402
import SQRNdbnseq as R              # noqa: E402  (the reference)
import SQUARNA as RC                # noqa: E402
from gen_golden import jsonable, sparse, rnd_seq, rnd_restraints, add_bp_restraints   # noqa: E402


def gen():
    rng = random.Random(2024)
    W = {"GC": 3.25, "AU": 1.25, "GU": -1.25}
    out = dict(bpmatrix=[], fold=[], calls=[])
    # ---- BPMatrix with a probability term: multiply (p > 0), add (p < 0), both zero branches (len % 11 == 3 / 7)
    for n, power, kind in [(24, 0.5, ""), (24, -1.0, ""), (31, 0.5, "r"), (31, -1.0, "r"), (40, 2.0, "x"),
                           (25, 0.5, ""), (25, -1.0, "r"), (29, 0.5, ""), (29, -1.0, ""), (36, -0.5, "s"),
                           (33, 1.0, "rx"), (38, -1.0, "x"), (35, -0.5, "rx")]:   # (added term + restraints: cells with bool == 0 get it too)
        seq = rnd_seq(rng, n)
        if "s" in kind:
            seq = seq[:n // 2] + "&" + seq[n // 2 + 1:]
        restr = rnd_restraints(rng, n, 0.15) if "x" in kind else "." * n
        reacts = None
        if "r" in kind:
            reacts = R.ProcessReacts([rng.choice([0.0, 0.5, 1.0, -999, 0.3, 0.8]) for _ in range(n)], M=1.8, B=-0.6)
        rbps, rxs, rl, rr = R.ParseRestraints(restr)
        b, s = R.BPMatrix(seq, W, rxs, rl, rr, False, reacts, bpp_power=power, M=1.8, B=-0.6)
        out["bpmatrix"].append(dict(seq=seq, weights=W, restraints=restr, reacts=reacts, bpp_power=power,
                                    bool=[[i, j] for i, j, _ in sparse(b)], score=sparse(s)))
        out["calls"].append(dict(seq=seq, reacts=reacts, calls=jsonable(list(fake_rna.CALLS))))
    # ---- whole folds under def.conf
    names, psets = RC.ParseConfig(os.path.join(REF, "def.conf"))
    prio = {i for i, nm in enumerate(names) if nm in ("bppN", "bppH1", "bppH2")}
    cases = []
    for n in (16, 25, 29, 40, 58, 77, 90):                  # 25 % 11 == 3 (rescale retry), 29 % 11 == 7 (stays zero)
        cases.append((rnd_seq(rng, n), None, None))
    seq = rnd_seq(rng, 60)
    line = "".join(rng.choice("_+#") for _ in range(60))
    cases.append((seq, R.ProcessReacts([R.ReactDict[c] for c in line], M=1.8, B=-0.6), None))
    seq = rnd_seq(rng, 48)
    cases.append((seq, None, add_bp_restraints(rng, seq, rnd_restraints(rng, 48, 0.05), 2)))
    seq = rnd_seq(rng, 52)
    cases.append((seq[:26] + "&" + seq[27:], None, None))
    for k, (seq, reacts, restr) in enumerate(cases):
        for kw in (dict(), dict(priority=prio, rankby=(2, 0, 1), poollim=100)):
            res = R.SQRNdbnseq(seq, reacts, restr, None, psets, mp=False, **kw)
            out["fold"].append(dict(tag="bpp[%d]" % k, seq=seq, reacts=reacts, restraints=restr, config="def",
                                    kw=jsonable(kw), out=jsonable(res)))
    with open(os.path.join(HERE, "bpp.json"), "w") as f:
        json.dump(jsonable(out), f, separators=(",", ":"))
    print("bpp.json", os.path.getsize(os.path.join(HERE, "bpp.json")), "bytes")
    # ---- Predict with the default configuration (config not given: def.conf, autoconfig, default priority)
    with open(os.path.join(HERE, "digests.json")) as f:
        digests = json.load(f)
    jobs = [("s16_def", dict(inputseq="ACGUACGUACUCGACG")),
            ("seq_input_def", dict(inputfile=os.path.join(REF, "examples/seq_input.fas"))),
            ("shape_input_def_rb", dict(inputfile=os.path.join(REF, "examples/shape_input.fas"), rankby="s", hardrest=True)),
            ("SRtest150_def", dict(inputfile=os.path.join(REF, "datasets/SRtest150.fas"), inputformat="qf"))]
    for tag, kw in jobs:
        buf = io.StringIO()
        RC.Predict(write_to=buf, byseq=True, threads=8, **kw)
        txt = buf.getvalue()
        with open(os.path.join(HERE, "text", tag + ".txt"), "w") as f:
            f.write(txt)
        args = {k: (os.path.relpath(v, REF) if k == "inputfile" else v) for k, v in kw.items()}
        digests[tag] = dict(args=args, lines=txt.count("\n"), sha256=hashlib.sha256(txt.encode()).hexdigest(),
                            fake_rna=True)
        print(tag, digests[tag]["lines"], digests[tag]["sha256"][:16], flush=True)
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(digests, f, separators=(",", ":"))


if __name__ == "__main__":
    gen()
