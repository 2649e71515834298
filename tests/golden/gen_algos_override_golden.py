#!/usr/bin/env python3
"""Golden folds with the `algos=` override (SQRNdbnseq.py:1046,1065-1066,1094-1100): every paramset then runs
the SAME user-chosen set of algorithms, several non-greedy ones per paramset.

The reference iterates a Python *set* of one-letter strings at :1094, whose order depends on
PYTHONHASHSEED; the order decides which stemset is seen first (:1201-1212) and so can only matter for
structures that tie on every ranking key.  This script therefore re-runs itself under several hash
seeds and only keeps a case when the reference's return value is the same under all of them (all
cases below are); the product and the oracle use the fixed order E, H, N.

Output: tests/golden/fold_algos.json.   Usage: python tests/golden/gen_algos_override_golden.py
"""
import json
import os
import random
import subprocess
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/SQUARNA"
SEEDS = ("0", "1", "4", "5", "11")


def cases():
    rng = random.Random(5)
    out = []
    for n in (30, 50, 80, 120, 150):
        for t in range(3):
            seq = "".join(rng.choice("ACGU") for _ in range(n))
            reacts = None
            if t == 1:
                reacts = "".join(rng.choice("_+#") for _ in range(n))
            for algos in ("EHN", "EHNG", "HN", "EG"):
                out.append((seq, reacts, algos))
    return out


def worker():
    sys.path.insert(0, REF)
    sys.path.insert(0, HERE)
    import SQRNdbnseq as R
    import SQUARNA as RC
    from gen_golden import jsonable
    names, psets = RC.ParseConfig(os.path.join(REF, "nobpp.conf"))
    res = []
    for seq, reacts, algos in cases():
        r = R.SQRNdbnseq(seq, reacts, None, None, psets, mp=False, algos=set(algos), poollim=100)
        res.append(dict(seq=seq, reacts=reacts, algos=algos, config="nobpp", kw=dict(poollim=100), out=jsonable(r)))
    json.dump(res, sys.stdout, separators=(",", ":"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker()
        sys.exit(0)
    runs = []
    for seed in SEEDS:
        env = dict(os.environ, PYTHONHASHSEED=seed, PYTHONDONTWRITEBYTECODE="1")
        runs.append(json.loads(subprocess.check_output([sys.executable, __file__, "worker"], env=env)))
    keep = [c for k, c in enumerate(runs[0]) if all(r[k] == c for r in runs[1:])]
    print("%d of %d cases are independent of the set order under hash seeds %s" % (len(keep), len(runs[0]), ",".join(SEEDS)))
    with open(os.path.join(HERE, "fold_algos.json"), "w") as f:
        json.dump(keep, f, separators=(",", ":"))
    print("fold_algos.json", os.path.getsize(os.path.join(HERE, "fold_algos.json")), "bytes")
