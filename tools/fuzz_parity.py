#!/usr/bin/env python3
"""Randomised parity soak: NSEQ random sequences (with occasional reactivities / restraints) folded under a config
by the CPU oracle (worker processes, before the GPU is touched) and by the HIP engine; every field of the
SQRNdbnseq tuple is compared.  usage: fuzz_parity.py NSEQ CONFIG [SEED]   (FUZZ_POOLLIM=1: the device-chained rounds)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def make(nseq, seed):
    rng = np.random.default_rng(seed)
    recs = []
    for k in range(nseq):
        n = int(rng.integers(int(os.environ.get("FUZZ_NMIN", "12")), int(os.environ.get("FUZZ_NMAX", "170"))))
        seq = "".join(rng.choice(list("ACGU"), n, p=[0.22, 0.28, 0.28, 0.22]))
        reacts = None
        if k % 5 == 0:
            reacts = [float(x) for x in np.round(rng.random(n), 3)]
        if k % 10 == 0:                                  # few distinct values (encoded reactivities): the table path
            reacts = [float(x) for x in rng.choice([0.0, 0.1, 0.35, 0.5, 0.8, 1.0], n)]
        restr = None
        if k % 7 == 0:
            r = ["."] * n
            for p in rng.choice(n, size=max(1, n // 15), replace=False):
                r[int(p)] = "_/\\+"[int(rng.integers(0, 4))]
            restr = "".join(r)
        if k % 11 == 0 and n > 40:                       # a restraint helix (paired brackets) somewhere
            r = list(restr) if restr else ["."] * n
            a = int(rng.integers(0, n // 2 - 8)); ln = int(rng.integers(2, 6)); b = int(rng.integers(n // 2 + 6, n - 1))
            for t in range(ln):
                if a + t < b - t - 3:
                    r[a + t], r[b - t] = "(", ")"
            restr = "".join(r)
        if k % 13 == 0 and n > 30:                       # two chains
            p = int(rng.integers(10, n - 10))
            seq = seq[:p] + "&" + seq[p + 1:]
            if restr:
                restr = restr[:p] + "." + restr[p + 1:]
                if restr.count("(") != restr.count(")"):
                    restr = restr.replace("(", ".").replace(")", ".")
        if k % 17 == 0:                                  # gaps, lower case, T, unknown letters
            seq = seq.replace("U", "T", 2).lower()[: n // 2] + seq[n // 2:]
            q = int(rng.integers(0, n))
            seq = seq[:q] + "-" + seq[q + 1:]
            q2 = int(rng.integers(0, n))
            if seq[q2] not in "&-":
                seq = seq[:q2] + "N" + seq[q2 + 1:]
            if restr:
                restr = restr[:q] + "." + restr[q + 1:]
                if restr.count("(") != restr.count(")"):
                    restr = restr.replace("(", ".").replace(")", ".")
        recs.append((seq, reacts, restr))
    return recs


def _init(cfg):
    global O, PS
    sys.path.insert(0, ROOT)
    from oracle import sqrn_oracle as O_
    from squarna_amd.config import ParseConfig, builtin_config
    O = O_
    PS = ParseConfig(builtin_config(cfg))[1]
    O.lib()


def _one(rec):
    seq, reacts, restr = rec
    r = O.SQRNdbnseq(seq, reacts, restr, None, PS, poollim=int(os.environ.get("FUZZ_POOLLIM", "1000")))
    return r[0], [(d, tuple(s), list(p)) for d, s, p in r[1]]


def main():
    nseq = int(sys.argv[1]); cfg = sys.argv[2]; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    recs = make(nseq, seed)
    if os.environ.get("FUZZ_SET") == "pools_long":         # bench.py's pools_long records (random 500-nt sequences) instead
        import bench
        recs = [(s, None, None) for s in bench.pools_long_sequences(nseq)]
    import multiprocessing as mp
    t0 = time.time()
    with mp.get_context("spawn").Pool(min(os.cpu_count() or 1, 64), initializer=_init, initargs=(cfg,)) as pool:
        exp = pool.map(_one, recs, chunksize=4)
    print("oracle: %.1f s" % (time.time() - t0), flush=True)
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import HipEngine
    psets = ParseConfig(builtin_config(cfg))[1]
    got = HipEngine().fold_records([(s, r, x, None, psets, None) for s, r, x in recs],
                                   poollim=int(os.environ.get("FUZZ_POOLLIM", "1000")))
    bad = 0
    for k, (g, e) in enumerate(zip(got, exp)):
        ok = g[0] == e[0] and len(g[1]) == len(e[1]) and all(
            a[0] == b[0] and list(a[2]) == list(b[2]) and all(abs(x - y) <= 1e-5 for x, y in zip(a[1], b[1]))
            for a, b in zip(g[1], e[1]))
        if not ok:
            bad += 1
            print("MISMATCH record %d n=%d reacts=%s restr=%s\n  seq %s\n  got %s\n  exp %s" % (
                k, len(recs[k][0]), recs[k][1] is not None, recs[k][2], recs[k][0], g[0], e[0]), flush=True)
    print("%d records (config %s, poollim %s), %d mismatches" % (nseq, cfg, os.environ.get("FUZZ_POOLLIM", "1000"), bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
