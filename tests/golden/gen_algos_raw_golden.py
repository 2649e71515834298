#!/usr/bin/env python3
"""Raw return values of the reference's SQRNalgos.Edmonds / Hungarian / Nussinov (SQRNalgos.py:44-135) on the stem
lists of tests/golden/algos.json: the pairs exactly as returned -- for Edmonds including the (u, v) orientation
networkx gives every pair and the sorted order -- for the drop-in functions squarna_amd.Edmonds / Hungarian /
Nussinov.  Output: tests/golden/algos_raw.json (index-aligned with algos.json).
Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_algos_raw_golden.py"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src/SQUARNA")
import SQRNalgos as A          # noqa: E402  (the reference)
import SQRNdbnseq as R         # noqa: E402

with open(os.path.join(HERE, "algos.json")) as f:
    cases = json.load(f)
out = []
for c in cases:
    stems = [[[(i + k, j - k) for k in range(ln)], ln, sc] for i, j, ln, sc in c["stems"]]
    n = len(c["seq"])
    if c["algo"] == "E":
        raw = A.Edmonds(stems)
    elif c["algo"] == "H":
        raw = A.Hungarian(c["seq"], stems, n, R.SEPS)
    else:
        raw = A.Nussinov(c["seq"], stems, n, R.SEPS)
    out.append(dict(name=c["name"], algo=c["algo"], raw=[[int(v), int(w)] for v, w in raw]))
flipped = sum(1 for o in out if o["algo"] == "E" for v, w in o["raw"] if v > w)
print("cases", len(out), "Edmonds pairs returned as (larger, smaller):", flipped)
with open(os.path.join(HERE, "algos_raw.json"), "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("algos_raw.json", os.path.getsize(os.path.join(HERE, "algos_raw.json")), "bytes")
