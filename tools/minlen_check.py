import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import sqrn_oracle as O
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config("greedynobpp"))
rng = np.random.default_rng(9)
bad = 0
for minlen, minbp in ((1, 0.0), (1, 3.0), (2, 0.0), (7, 0.0), (33, 0.0), (40, 0.0)):
    ps = [dict(psets[0], minlen=minlen, minbpscore=minbp, bpweights={"GC": 3.25, "AU": 1.25, "GU": -1.25})]
    recs = []
    for n in (30, 95, 140):
        if minlen >= 33:       # plant a long helix so that long runs exist
            half = "".join(rng.choice(list("ACGU"), 60))
            comp = half[::-1].translate(str.maketrans("ACGU", "UGCA"))
            seq = half + "GAAA" + comp + "".join(rng.choice(list("ACGU"), n))
        else:
            seq = "".join(rng.choice(list("ACGU"), n))
        recs.append((seq, None, None, None, ps, None))
    got = HipEngine().fold_records(recs)
    for r, g in zip(recs, got):
        e = O.SQRNdbnseq(r[0], None, None, None, ps)
        ok = g[0] == e[0] and [x[0] for x in g[1]] == [x[0] for x in e[1]]
        if not ok:
            bad += 1
            print("MISMATCH minlen", minlen, minbp, len(r[0]), g[0], e[0])
print("minlen sweep:", "ok" if not bad else "%d mismatches" % bad)
