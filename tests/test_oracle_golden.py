"""Pins the CPU oracle (oracle/) against vectors generated from the reference.

Every fixture under tests/golden/*.json was produced by tests/golden/gen_golden.py
importing febos/SQUARNA v3.2.2 in the build container.  Float comparisons are
exact (==): the oracle restates the same fp64 operations in the same order.
"""
import json
import os

import numpy as np
import pytest

from oracle import sqrn_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unnan(x):
    if isinstance(x, list):
        return [unnan(v) for v in x]
    return float("nan") if x == "nan" else x


def same(a, b):
    """Deep equality treating nan == nan and tuples == lists."""
    if isinstance(a, (list, tuple)) and isinstance(b, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, float) and isinstance(b, float) and np.isnan(a) and np.isnan(b):
        return True
    return a == b


def test_bpmatrix_golden():
    for c in load("bpmatrix.json"):
        rbps, rxs, rl, rr = O.ParseRestraints(c["restraints"])
        b, s = O.BPMatrix(c["seq"], c["weights"], rxs, rl, rr, c["interchainonly"], c["reacts"])
        got_b = [[int(i), int(j)] for i, j in zip(*np.nonzero(b))]
        assert got_b == c["bool"], c["seq"]
        got_s = [[int(i), int(j), float(s[i, j])] for i, j in zip(*np.nonzero(s))]
        assert got_s == c["score"], c["seq"]


def test_annotate_golden():
    for c in load("annotate.json"):
        rbps, rxs, rl, rr = O.ParseRestraints(c["restraints"])
        b, s = O.BPMatrix(c["seq"], c["weights"], rxs, rl, rr, False, c["reacts"])
        for rnd in c["rounds"]:
            rstems = [tuple(x) for x in rnd["rstems"]]
            used = {bp for st in rstems for bp in O.stem_bps(st)}
            rest = set(rbps) - used
            got = O.AnnotateStems(b, s, rest, rstems, c["minlen"], c["minscore"])
            assert [list(x) for x in got] == rnd["stems"], (c["seq"], rnd["rstems"])


def _prep(seq, reacts, restraints):
    seq = seq.upper().replace("T", "U")
    restraints = restraints or "." * len(seq)
    reacts = reacts or [0.5] * len(seq)
    shortseq, shortrest = O.UnAlign(seq, restraints)
    shortreacts = [reacts[i] for i in range(len(seq)) if seq[i] not in O.GAPS]
    return shortseq, shortrest, shortreacts


def test_optimalstems_trace_golden(conf_cache={}):
    from squarna_amd.config import ParseConfig, builtin_config
    ncalls = 0
    for tr in load("optimal.json"):
        names, psets = ParseConfig(builtin_config(tr["config"]))
        gsets = [p for p in psets if "G" in p["algorithms"]]
        shortseq, shortrest, shortreacts = _prep(tr["seq"], tr["reacts"], tr["restraints"])
        rbps, rxs, rl, rr = O.ParseRestraints(shortrest)
        mats = {}
        for call in tr["calls"]:
            ps = gsets[call["g"]]
            if call["g"] not in mats:
                mats[call["g"]] = O.BPMatrix(shortseq, ps["bpweights"], rxs, rl, rr,
                                             tr["kw"].get("interchainonly", False), shortreacts)
            b, s = mats[call["g"]]
            got = O.OptimalStems(shortseq, [tuple(x) for x in call["rstems"]], b, s, shortreacts, rbps,
                                 call["subopt"], ps["minlen"], ps["minbpscore"],
                                 ps["minbpscore"] * ps["minfinscorefactor"], ps["bracketweight"],
                                 ps["distcoef"], ps["orderpenalty"], ps["loopbonus"])
            assert [list(x) for x in got] == call["out"], (tr["tag"], tr["config"], call["rstems"])
            ncalls += 1
    assert ncalls > 1000


def test_fold_golden():
    from squarna_amd.config import ParseConfig, builtin_config
    for c in load("fold.json"):
        names, psets = ParseConfig(builtin_config(c["config"]))
        kw = dict(c["kw"])
        if "rankby" in kw:
            kw["rankby"] = tuple(kw["rankby"])
        out = O.SQRNdbnseq(c["seq"], c["reacts"], c["restraints"], c["reference"], psets, **kw)
        exp = unnan(c["out"])
        got = [out[0], [[d, list(sc), list(ps)] for d, sc, ps in out[1]], list(out[2]), list(out[3])]
        assert same(got, exp), (c["tag"], c["config"])


def test_algos_golden():
    from squarna_amd.config import ParseConfig, builtin_config
    names, psets = ParseConfig(builtin_config("nobpp"))
    ps = dict(zip(names, psets))
    for c in load("algos.json"):
        p = ps[c["paramset"]]
        rbps, rxs, rl, rr = O.ParseRestraints(c["restraints"])
        b, s = O.BPMatrix(c["seq"], p["bpweights"], rxs, rl, rr, False, c["reacts"])
        stems = O.AnnotateStems(b, s, rbps, [], p["minlen"], p["minbpscore"])
        assert [list(x) for x in stems] == c["stems"]
        ll = 3 - int(len(c["seq"]) > 500)
        got = O.RunAlgo(c["seq"], b, s, rbps, p["minlen"], p["minbpscore"], algo=c["algo"], levellimit=ll)
        assert [list(x[:4]) for x in got] == c["stemset"], (c["name"], c["algo"])
        if c["algo"] == "N":
            assert [list(x) for x in O.Nussinov(c["seq"], stems, len(c["seq"]))] == c["pairs"]


def test_appendix_b_known_answer():
    """GGGAAAACCC under alt.conf."""
    w = {"GC": 3.25, "AU": 1.25, "GU": -1.25}
    b, s = O.BPMatrix("GGGAAAACCC", w, set(), set(), set())
    cells = {(i, j) for i in range(3) for j in (7, 8, 9)}
    assert {(int(i), int(j)) for i, j in zip(*np.nonzero(b))} == cells
    assert all(s[i, j] == 3.25 for i, j in cells)
    stems = O.AnnotateStems(b, s, set(), [], 2, 0)
    assert stems == [(0, 8, 2, 6.5), (0, 9, 3, 9.75), (1, 9, 2, 6.5)]


# ---- bpp != 0 paramsets (dbnseq:341-364): fixtures from the REAL reference running on the ViennaRNA stand-in
# ---- tests/fake_rna.py (tests/golden/gen_bpp_golden.py).  Pins the application of the probabilities, the
# ---- zero-probability retry / skip, and everything downstream; the probabilities themselves stay unpinned.
def test_bpp_bpmatrix_golden(fake_rna):
    g = load("bpp.json")
    for c, calls in zip(g["bpmatrix"], g["calls"]):
        rbps, rxs, rl, rr = O.ParseRestraints(c["restraints"])
        b, s = O.BPMatrix(c["seq"], c["weights"], rxs, rl, rr, False, c["reacts"], bpp_power=c["bpp_power"])
        assert [[int(i), int(j)] for i, j in zip(*np.nonzero(b))] == c["bool"], c["seq"]
        got_s = [[int(i), int(j), float(s[i, j])] for i, j in zip(*np.nonzero(s))]
        assert got_s == c["score"], (c["seq"], c["bpp_power"])
        # the oracle made the same calls into `RNA`, with the same arguments, as the reference did
        mine = json.loads(json.dumps(list(fake_rna.CALLS)))
        assert mine == calls["calls"], (c["seq"], mine, calls["calls"])
    lens = {len(c["seq"]) % 11 for c in g["bpmatrix"]}
    assert {3, 7} <= lens                     # both zero branches (:355-364) are in the fixture


def test_bpp_fold_golden(fake_rna):
    from squarna_amd.config import ParseConfig, builtin_config
    names, psets = ParseConfig(builtin_config("def"))
    assert len(psets) == 12 and sum(1 for p in psets if p["bpp"]) == 7
    n = 0
    for c in load("bpp.json")["fold"]:
        kw = dict(c["kw"])
        if "rankby" in kw:
            kw["rankby"] = tuple(kw["rankby"])
        out = O.SQRNdbnseq(c["seq"], c["reacts"], c["restraints"], None, psets, **kw)
        got = [out[0], [[d, list(sc), list(ps)] for d, sc, ps in out[1]], list(out[2]), list(out[3])]
        assert same(got, unnan(c["out"])), (c["tag"], c["kw"])
        n += 1
    assert n >= 20


def test_algos_override_golden():
    """`algos=` override: several non-greedy algorithms per paramset (dbnseq:1065-1066,1094-1100), reference
    outputs that do not depend on its set iteration order (tests/golden/gen_algos_override_golden.py)."""
    from squarna_amd.config import ParseConfig, builtin_config
    names, psets = ParseConfig(builtin_config("nobpp"))
    cases = load("fold_algos.json")
    assert len(cases) >= 40 and {c["algos"] for c in cases} >= {"EHN", "EHNG", "HN", "EG"}
    for c in cases:
        out = O.SQRNdbnseq(c["seq"], c["reacts"], None, None, psets, algos=set(c["algos"]), **c["kw"])
        got = [out[0], [[d, list(sc), list(ps)] for d, sc, ps in out[1]], list(out[2]), list(out[3])]
        assert same(got, unnan(c["out"])), (c["seq"], c["algos"])


def test_reference_form_of_the_oracle_equals_the_c_form():
    """oracle/sqrn_pyform.py -- the oracle's hot loops as interpreted per-cell Python, the form cpu_baseline.reference_form is
    timed in -- gives the tuples of the C form: short SRtest150 records under nobpp (G, N, E, H paramsets) and the example
    records with restraints, reactivities and separators."""
    from oracle import sqrn_oracle as O, sqrn_pyform as P
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.inputs import ParseDefaultInput
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names, psets = ParseConfig(builtin_config("nobpp"))
    recs = sorted(ParseDefaultInput(os.path.join(root, "squarna_amd", "data", "datasets", "SRtest150.fas"), "qf"), key=lambda r: len(r[1]))[:60:6]
    recs += [r for r in ParseDefaultInput(os.path.join(root, "squarna_amd", "data", "examples", "seq_input.fas"), "q") if len(r[1]) <= 80]
    for name, seq, reacts, restr, ref in recs:
        want = O.SQRNdbnseq(seq, reacts, restr, ref, psets)
        P.install(True)
        try:
            got = O.SQRNdbnseq(seq, reacts, restr, ref, psets)
        finally:
            P.install(False)
        assert repr(got) == repr(want), name
