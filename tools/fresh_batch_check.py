"""Freshly created batches of the same records (two without reactivities, one with explicit halves) folded once each: the
packed records must be equal byte for byte, pads included -- catches results that depend on what a recycled pinned buffer or
workspace held.  usage: fresh_batch_check.py [TRIALS=20] [FIRST=0]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
names, psets = ParseConfig(builtin_config("nobpp"))
def packed(b):
    buf, off = b.pack_all()
    return [buf[off[k]:off[k + 1]].tobytes() for k in range(b.nseq)]
bad = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    if len(sys.argv) > 2 and trial < int(sys.argv[2]): continue
    rng = random.Random(77 + trial)
    seqs = ["".join(rng.choice("ACGU") for _ in range(rng.randrange(30, 140))) for _ in range(24)]
    plain = [Prepared(s, None) for s in seqs]
    explicit = [Prepared(s, [0.5] * len(s)) for s in seqs]
    with Batch(plain, [psets] * len(seqs), fp32=False) as a, Batch(explicit, [psets] * len(seqs), fp32=False) as b, Batch(plain, [psets] * len(seqs), fp32=False) as a2:
        a.fold(poollim=1000); b.fold(poollim=1000); a2.fold(poollim=1000)
        pa, pb, pa2 = packed(a), packed(b), packed(a2)
        for k in range(len(seqs)):
            if pa[k] != pb[k] or pa[k] != pa2[k]:
                bad += 1
                ha, hb, h2 = (np.frombuffer(x[:32], np.int64) for x in (pa[k], pb[k], pa2[k]))
                print("trial", trial, "seq", k, "hdr a", ha, "b", hb, "a2", h2, "len", len(pa[k]), len(pb[k]), len(pa2[k]), "a==a2", pa[k] == pa2[k], "b==a2", pb[k] == pa2[k])
                x, y = np.frombuffer(pa[k], np.uint8), np.frombuffer(pa2[k], np.uint8)
                d = np.flatnonzero(x != y)
                ns, n = int(ha[0]), int(ha[1])
                sec = lambda o: "hdr" if o < 32 else "metrics" if o < 160 else "scores[%d].%d" % ((o - 160) // 24, ((o - 160) % 24) // 8) if o < 160 + 24 * ns else "masks[%d]" % ((o - 160 - 24 * ns) // 8) if o < 160 + 32 * ns else "levels[row %d pos %d]" % ((o - 160 - 32 * ns) // (2 * n), ((o - 160 - 32 * ns) % (2 * n)) // 2)
                print("   %d bytes differ; first at %s, last at %s; sections: %s" % (len(d), sec(int(d[0])), sec(int(d[-1])), sorted({sec(int(o)).split('[')[0] for o in d})))
                sa = np.frombuffer(pa[k][160:160 + 24 * ns], np.float64).reshape(ns, 3); s2 = np.frombuffer(pa2[k][160:160 + 24 * ns], np.float64).reshape(ns, 3)
                rows = np.flatnonzero((sa != s2).any(1))
                print("   score rows differing:", rows[:10], "a:", sa[rows[:3]].tolist(), "a2:", s2[rows[:3]].tolist())
                a2.fold(poollim=1000); print("   a2 refolded == a:", packed(a2)[k] == pa[k])
                a.fold(poollim=1000); b.fold(poollim=1000)
                ra, rb = packed(a)[k], packed(b)[k]
                print("   refold: a same as before", ra == pa[k], " b same as before", rb == pb[k], " a==b now", ra == rb)
print("bad", bad)
