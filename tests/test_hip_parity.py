"""GPU parity: the HIP path (through the C ABI) against the reference's golden vectors
and against the CPU oracle on seeded random inputs.

Bars (north_star): stem indices bit-identical; float stem scores within 1e-5
(in practice exact: decisions are taken in fp64 in the reference's operation order).
"""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def conf(name):
    from squarna_amd.config import ParseConfig, builtin_config
    return ParseConfig(builtin_config(name))


def mk_pset(weights, minlen=2, minbpscore=4.5, **kw):
    ps = dict(bpweights=weights, bpp=0, algorithms={"G"}, suboptmax=1.0, suboptmin=1.0, suboptsteps=1.0,
              minlen=minlen, minbpscore=minbpscore, minfinscorefactor=1.0, bracketweight=-2.0,
              distcoef=0.09, orderpenalty=1.0, loopbonus=0.125, maxstemnum=1e6)
    ps.update(kw)
    return ps


def prep(seq, reacts=None, restraints=None, dbn=None):
    from squarna_amd.engine import Prepared
    return Prepared(seq, reacts, restraints, dbn)


def close_stems(got, exp, what=""):
    assert len(got) == len(exp), (what, got, exp)
    for g, e in zip(got, exp):
        assert list(g[:3]) == list(e[:3]), (what, g, e)
        for a, b in zip(g[3:], e[3:]):
            assert abs(a - b) <= TOL * max(1.0, abs(b)), (what, g, e)


def test_bpmatrix_golden():
    from squarna_amd.engine import Batch
    cases = load("bpmatrix.json")
    # interchainonly is a batch-level switch: two batches
    for ico in (False, True):
        sel = [c for c in cases if bool(c["interchainonly"]) == ico]
        preps = [prep(c["seq"], c["reacts"], c["restraints"]) for c in sel]
        psets = [[mk_pset(c["weights"])] for c in sel]
        with Batch(preps, psets, interchainonly=ico) as b:
            b.fill()
            for k, c in enumerate(sel):
                bm, sm = b.bpmatrix(k)
                got_b = [[int(i), int(j)] for i, j in zip(*np.nonzero(bm))]
                assert got_b == c["bool"], c["seq"]
                exp = np.zeros_like(sm)
                for i, j, v in c["score"]:
                    exp[i, j] = v
                assert np.allclose(sm, exp, rtol=1e-12, atol=0), c["seq"]
                nz = [[int(i), int(j)] for i, j in zip(*np.nonzero(sm))]
                assert nz == [[i, j] for i, j, _ in c["score"]]


def test_annotate_golden():
    from squarna_amd.engine import Batch
    cases = load("annotate.json")
    preps = [prep(c["seq"], c["reacts"], c["restraints"]) for c in cases]
    psets = [[mk_pset(c["weights"], c["minlen"], c["minscore"])] for c in cases]
    with Batch(preps, psets) as b:
        sj, ss, exp = [], [], []
        for k, c in enumerate(cases):
            for rnd in c["rounds"]:
                sj.append(k)
                ss.append([tuple(x) for x in rnd["rstems"]])
                exp.append(rnd["stems"])
        got = b.optimal(sj, ss, mode=1)
        for g, e, k in zip(got, exp, sj):
            close_stems([x[:4] for x in g], e, cases[k]["seq"])


def test_optimalstems_trace_golden():
    from squarna_amd.engine import Batch
    ncalls = 0
    for tr in load("optimal.json"):
        names, psets = conf(tr["config"])
        gsets = [p for p in psets if "G" in p["algorithms"]]
        p = prep(tr["seq"], tr["reacts"], tr["restraints"])
        with Batch([p], [gsets], interchainonly=tr["kw"].get("interchainonly", False)) as b:
            sj = [c["g"] for c in tr["calls"]]
            ss = [[tuple(x) for x in c["rstems"]] for c in tr["calls"]]
            so = [c["subopt"] for c in tr["calls"]]
            got = b.optimal(sj, ss, subopt=so, mode=0)
            for g, c in zip(got, tr["calls"]):
                close_stems(g, c["out"], (tr["tag"], tr["config"], c["rstems"]))
                ncalls += 1
    assert ncalls > 1000


def _same_fold(got, exp, tag):
    assert got[0] == exp[0], (tag, "consensus", got[0], exp[0])
    assert len(got[1]) == len(exp[1]), (tag, len(got[1]), len(exp[1]))
    for g, e in zip(got[1], exp[1]):
        assert g[0] == e[0], (tag, g, e)
        assert all(abs(a - b) <= TOL for a, b in zip(g[1], e[1])), (tag, g, e)
        assert list(g[2]) == list(e[2]), (tag, g, e)
    for g, e in zip(list(got[2]) + list(got[3]), list(exp[2]) + list(exp[3])):
        if e == "nan":
            assert g != g, tag
        else:
            assert abs(g - e) <= TOL, (tag, got[2], got[3], exp[2], exp[3])


def test_fold_golden_all_configs():
    from squarna_amd.engine import HipEngine
    eng = HipEngine()
    n = 0
    for c in load("fold.json"):
        names, psets = conf(c["config"])
        kw = dict(c["kw"])
        if "rankby" in kw:
            kw["rankby"] = tuple(kw["rankby"])
        out = eng.fold_records([(c["seq"], c["reacts"], c["restraints"], c["reference"], psets, None)], **kw)[0]
        _same_fold(out, c["out"], (c["tag"], c["config"]))
        n += 1
    assert n > 50


def _rand_case(rng, n):
    seq = "".join(rng.choice("ACGU") for _ in range(n))
    kind = rng.randrange(4)
    reacts = restr = None
    if kind == 1:
        from squarna_amd.dbn import ProcessReacts, ReactDict
        reacts = ProcessReacts([ReactDict[rng.choice("_+#")] for _ in range(n)], M=1.8, B=-0.6)
    if kind == 2:
        restr = "".join(rng.choice("..........._/\\") for _ in range(n))
    if kind == 3:
        from squarna_amd.dbn import ProcessReacts
        reacts = ProcessReacts([rng.random() * 1.4 - 0.2 for _ in range(n)], M=1.8, B=-0.6)
    return seq, reacts, restr


@pytest.mark.parametrize("cfg,sizes", [("greedynobpp", (40, 77, 150)), ("alt", (33, 90)),
                                       ("fastest", (300, 513, 700))])
def test_fold_vs_oracle_random(cfg, sizes):
    """Seeded random inputs, sizes beyond the goldens, checked against the CPU oracle."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf(cfg)
    rng = random.Random({"greedynobpp": 101, "alt": 202, "fastest": 303}[cfg])
    recs = []
    for n in sizes:
        for _ in range(3):
            seq, reacts, restr = _rand_case(rng, n)
            recs.append((seq, reacts, restr, None, psets, None))
    got = HipEngine().fold_records(recs, poollim=100, rankby=(2, 0, 1))
    for r, g in zip(recs, got):
        exp = O.SQRNdbnseq(r[0], r[1], r[2], None, psets, poollim=100, rankby=(2, 0, 1))
        exp = [exp[0], [[d, list(s), list(p)] for d, s, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(g, exp, (cfg, len(r[0])))


@pytest.mark.parametrize("cfg,count", [("fastest", 700), ("greedynobpp", 300)])
def test_big_batch_two_lanes_vs_oracle(cfg, count):
    """>= 512 greedy jobs in one batch: sq_fold splits its rounds over two lanes (two host threads, two streams) and
    ranks finished sequences while the others still fold; every record must still equal the CPU oracle.  Mixed
    lengths, reactivities, restraint lines, a second chain and unknown letters keep the two lanes uneven."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf(cfg)
    rng = random.Random(4242 + count)
    recs = []
    for k in range(count):
        seq, reacts, restr = _rand_case(rng, rng.randrange(12, 90))
        if k % 9 == 0 and len(seq) > 30:
            p = rng.randrange(10, len(seq) - 10)
            seq = seq[:p] + "&" + seq[p + 1:]
        if k % 13 == 0:
            p = rng.randrange(len(seq))
            seq = seq[:p] + "N" + seq[p + 1:]
        recs.append((seq, reacts, restr, None, psets, None))
    got = HipEngine().fold_records(recs, poollim=50, rankby=(2, 0, 1))
    assert len(got) == count
    for r, g in zip(recs, got):
        exp = O.SQRNdbnseq(r[0], r[1], r[2], None, psets, poollim=50, rankby=(2, 0, 1))
        exp = [exp[0], [[d, list(s), list(p)] for d, s, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(g, exp, (cfg, r[0]))


def test_edge_cases():
    """Empty-ish and degenerate inputs the reference handles (N < 5 has no diagonals)."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf("greedynobpp")
    recs = [(s, None, None, None, psets, None) for s in
            ("A", "GC", "GGCC", "GGGCC", "AAAAAAAAAA", "GGGG;CCCC", "GC&GC", "GGGGAAAACCCC", "N" * 12,
             "GGGGG-AAAA--CCCCC")]
    got = HipEngine().fold_records(recs)
    for r, g in zip(recs, got):
        exp = O.SQRNdbnseq(r[0], None, None, None, psets)
        exp = [exp[0], [[d, list(s), list(p)] for d, s, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(g, exp, r[0])


# ---- end-to-end text: Predict() on the HIP engine == the reference's own output, byte for byte
GPU_TEXT = ["s16_nobpp", "seq_input_nobpp", "SRtest150_nobpp", "s16_fastest", "shape_input_fastest", "shape_input_alt_rf26", "seq_input_entropy",
            "seq_input_ico", "seq_input_greedynobpp_rf10", "seq_input_evalonly", "SRtest150_fastest",
            "SRtest150_fastest_pl1", "SRtest150_alt", "SRtest150_greedynobpp", "ali_input_a", "ali_input_a_verbose",
            "ali_input_a_s3i", "ali_input_a_s31", "ali_input_a_entropy", "demo_afa_a"]


@pytest.mark.parametrize("tag", GPU_TEXT)
def test_predict_text_matches_reference_on_gpu(tag):
    import hashlib
    import io
    from squarna_amd import Predict
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        dig = json.load(f)[tag]
    kw = dict(dig["args"])
    if "inputfile" in kw:
        kw["inputfile"] = os.path.join(os.path.dirname(GOLDEN), "..", "squarna_amd", "data", kw["inputfile"])
    buf = io.StringIO()
    Predict(write_to=buf, **kw)
    txt = buf.getvalue()
    with open(os.path.join(GOLDEN, "text", tag + ".txt")) as f:
        exp = f.read()
    if txt != exp:
        tl, el = txt.split("\n"), exp.split("\n")
        bad = [(k, a, b) for k, (a, b) in enumerate(zip(tl, el)) if a != b][:3]
        raise AssertionError("text differs (%d vs %d lines): %r" % (len(tl), len(el), bad))
    assert hashlib.sha256(txt.encode()).hexdigest() == dig["sha256"]


def test_api_shims_match_oracle():
    """BPMatrix / AnnotateStems / OptimalStems with the reference's signatures."""
    import squarna_amd as S
    from oracle import sqrn_oracle as O
    w = {"GC": 3.25, "AU": 1.25, "GU": -1.25}
    seq = "GGGAAAUCCCGCGAAAGCGUUUACGC"
    b, s = S.BPMatrix(seq, w, {3}, {20}, set())
    ob, os_ = O.BPMatrix(seq, w, {3}, {20}, set())
    assert (b == ob).all() and np.allclose(s, os_, rtol=1e-12, atol=0)
    stems = S.AnnotateStems(ob, os_, set(), [], 2, 0)
    exp = O.AnnotateStems(ob, os_, set(), [], 2, 0)
    assert [(st[0][0][0], st[0][0][1], st[1], st[2]) for st in stems] == exp
    got = S.OptimalStems(seq, [], ob, os_, [0.5] * len(seq), set(), 0.8, 2, 4.5, 4.5, -2.0, 0.09, 1.0, 0.125)
    exp = O.OptimalStems(seq, [], ob, os_, [0.5] * len(seq), set(), 0.8, 2, 4.5, 4.5, -2.0, 0.09, 1.0, 0.125)
    assert [(st[0][0][0], st[0][0][1], st[1]) for st in got] == [e[:3] for e in exp]
    assert all(abs(st[3] - e[4]) <= TOL for st, e in zip(got, exp))


def test_runalgo_golden():
    """a-8 / a-9 / Nussinov: RunAlgo stemsets for E, H, N against the reference (scipy / networkx)."""
    from squarna_amd.engine import Batch
    names, psets = conf("nobpp")
    ps = dict(zip(names, psets))
    cases = load("algos.json")
    for algo in "EHN":
        sel = [c for c in cases if c["algo"] == algo]
        preps = [prep(c["seq"], c["reacts"], c["restraints"]) for c in sel]
        with Batch(preps, [[ps[c["paramset"]]] for c in sel]) as b:
            got = b.run_algo(list(range(len(sel))), algo)
        for g, c in zip(got, sel):
            close_stems([x[:4] for x in g], c["stemset"], (c["name"], algo))


def test_runalgo_vs_oracle_random():
    """E / H / N on seeded random sequences (ties everywhere) against the oracle's scipy/networkx calls."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import Batch
    names, psets = conf("nobpp")
    ps = dict(zip(names, psets))
    rng = random.Random(77)
    seqs = ["".join(rng.choice("ACGU") for _ in range(n)) for n in (20, 35, 50, 64, 80, 100, 120, 150, 200, 260)]
    seqs += ["GGGGAAAACCCC" * 6, "GCGCGCGCAAAAGCGCGCGC" * 4, "ACGU" * 30]
    for algo, pname in (("E", "defE"), ("H", "defH"), ("N", "defN")):
        p = ps[pname]
        with Batch([prep(s) for s in seqs], [[p]] * len(seqs)) as b:
            got = b.run_algo(list(range(len(seqs))), algo)
        for s, g in zip(seqs, got):
            bm, sm = O.BPMatrix(s, p["bpweights"], set(), set(), set(), False, [0.5] * len(s))
            exp = O.RunAlgo(s, bm, sm, [], p["minlen"], p["minbpscore"], algo=algo, levellimit=3 - int(len(s) > 500))
            close_stems([x[:4] for x in g], [e[:4] for e in exp], (algo, len(s)))


# ---- BASELINE.json sizes: synthetic S300 / S1000 / S2000, against the oracle on a
#      sample, plus size-independent properties on the whole batch
def _synthetic(n, count, seed, reacts=False):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        seq = "".join(rng.choice(list("ACGU"), n))
        rc = None
        if reacts:                                   # C4: reactivity line drawn from "_+#" with p = (0.5, 0.3, 0.2)
            from squarna_amd.dbn import ProcessReacts, ReactDict
            line = "".join(rng.choice(list("_+#"), n, p=[0.5, 0.3, 0.2]))
            rc = ProcessReacts([ReactDict[c] for c in line], M=1.8, B=-0.6)
        out.append((seq, rc))
    return out


@pytest.mark.parametrize("n,count,seed,reacts,sample", [(300, 64, 300, False, 16), (1000, 24, 1000, False, 8),
                                                        (2000, 12, 2000, True, 10)])
def test_baseline_sizes_vs_oracle_and_properties(n, count, seed, reacts, sample):
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf("fastest")
    data = _synthetic(n, count, seed, reacts)
    recs = [(s, rc, None, None, psets, None) for s, rc in data]
    got = HipEngine().fold_records(recs, poollim=1)
    # (1) a sample against the CPU oracle: structures identical, scores within 1e-5
    for k in range(sample):
        s, rc = data[k]
        exp = O.SQRNdbnseq(s, rc, None, None, psets, poollim=1)
        exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(got[k], exp, ("S%d" % n, k))
    # (2) batch independence + determinism: folding a record alone gives the same answer as in the batch
    for k in (0, count - 1):
        alone = HipEngine().fold_records([recs[k]], poollim=1)[0]
        assert alone[0] == got[k][0] and alone[1] == got[k][1], ("alone vs batch", n, k)
    again = HipEngine().fold_records(recs[:4], poollim=1)
    assert [g[:2] for g in again] == [g[:2] for g in got[:4]]
    # (3) every predicted structure is a valid matching of allowed pairs (GC/AU/GU, i < j - 3)
    from squarna_amd.dbn import DBNToPairs
    ok = {"GC", "CG", "AU", "UA", "GU", "UG"}
    for (s, rc), g in zip(data, got):
        for dbn, sc, ps_ in g[1]:
            assert len(dbn) == n
            pairs = DBNToPairs(dbn)
            used = [p for bp in pairs for p in bp]
            assert len(used) == len(set(used))
            assert all(s[i] + s[j] in ok and j - i >= 4 for i, j in pairs)


# ---- alignment step 1 on the device: column matrix == the reference's sequential accumulation, bit for bit
def _random_msa(rng, nseq, ncol, mut=0.15, gap=0.10):
    anc = rng.choice(list("ACGU"), ncol)
    rows = []
    for _ in range(nseq):
        row = anc.copy()
        m = rng.random(ncol) < mut
        row[m] = rng.choice(list("ACGU"), int(m.sum()))
        row[rng.random(ncol) < gap] = "-"
        rows.append("".join(row))
    return rows


@pytest.mark.parametrize("with_reacts", [False, True])
def test_align_matrix_matches_sequential_accumulation(with_reacts):
    from squarna_amd.engine import HipEngine
    from tests.oracle_engine import OracleEngine
    rng = np.random.default_rng(77 + with_reacts)
    rows = _random_msa(rng, 24, 260)
    w = {"GC": 3.25, "AU": 2.0, "GU": -1.0}
    recs = []
    for r in rows:
        reacts = [float(x) for x in rng.random(len(r))] if with_reacts else None    # non-dyadic scores: order matters
        recs.append((r, reacts, "." * len(r)))
    L = len(rows[0])
    exp = np.zeros((L, L))
    for (seq, _, _), (short, stems) in zip(recs, OracleEngine().yield_stems(recs, w, 2, 4.5)):
        cols = [c for c, ch in enumerate(seq) if ch not in "-.~"]
        for i, j, ln, sc in stems:                                  # SQRNdbnali.py:233-237
            for k in range(ln):
                v, ww = cols[i + k], cols[j - k]
                exp[v, ww] += sc
                exp[ww, v] += sc
    eng = HipEngine()
    got = eng.stem_matrix(recs, w, 2, 4.5)
    g = got.cpu().numpy()
    if with_reacts:
        # cell scores carry the documented sqrt-vs-pow(x, 0.5) 1-ulp deviation; same order of additions
        assert np.array_equal(g != 0, exp != 0)
        assert np.allclose(g, exp, rtol=1e-13, atol=0), float(np.max(np.abs(g - exp) / np.maximum(np.abs(exp), 1e-300)))
        exp = g
    else:
        assert np.array_equal(g, exp)
    thr = 4.5 * 6
    idx, val = eng.matrix_cells(got, thr)
    flat = exp.flatten()
    e_idx = np.array([q for q in np.flatnonzero(flat >= thr) if q % L - q // L >= 4], np.int64)
    assert np.array_equal(idx, e_idx) and np.array_equal(val, flat[e_idx])


# ---- bpp != 0 paramsets: the dense probability term applied by the fill (dbnseq:341-364).  ViennaRNA is not
# ---- available, so the probabilities come from a synthetic source shared by the oracle and the product: what is
# ---- checked is the application (add for bpp < 0, multiply for bpp > 0) and everything downstream of it.
def _fake_bpp(seq, reacts, M, B):
    n = len(seq)
    rng = np.random.default_rng(n * 7919 + sum(map(ord, seq)))
    m = np.triu(rng.random((n, n)) ** 3, 1)
    return m


@pytest.mark.parametrize("power", [0.5, -1.0])
def test_bpp_term_matches_oracle(power):
    from squarna_amd import engine as E
    from oracle import sqrn_oracle as O
    names, psets = conf("nobpp")
    psets = [dict(ps, bpp=power) for ps in psets]           # greedy, Nussinov, Edmonds and Hungarian paramsets, all with the term
    rng = np.random.default_rng(5)
    recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for n in (40, 77, 120, 33)]
    old = E.set_bpp_provider(_fake_bpp)
    O.BPP_SOURCE = _fake_bpp
    try:
        got = E.HipEngine().fold_records(recs)
        for r, g in zip(recs, got):
            exp = O.SQRNdbnseq(r[0], None, None, None, psets)
            exp = [exp[0], [[d, list(s), list(p)] for d, s, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
            _same_fold(g, exp, r[0])
    finally:
        E.set_bpp_provider(old)
        O.BPP_SOURCE = None


def test_bpp_without_vienna_raises_clearly():
    from squarna_amd import engine as E
    names, psets = conf("nobpp")
    psets = [dict(psets[0], bpp=0.5)]
    try:
        import RNA  # noqa: F401
        pytest.skip("ViennaRNA is installed here")
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match="ViennaRNA"):
        E.HipEngine().fold_records([("GGGAAACCC", None, None, None, psets, None)])


# ---- error behaviour of the C ABI (status code + sq_last_error text, surfaced as RuntimeError by the binding)
def test_c_abi_rejects_bad_arguments():
    import torch
    from squarna_amd.engine import Batch
    names, psets = conf("greedynobpp")
    p = prep("GGGAAAUCCCGCGAAAGCGUUUACGC", None, None)
    with Batch([p], [psets[:1]], fp32=False) as b:
        with pytest.raises(RuntimeError, match="NO_FP32"):
            b.fill()                                                # no fp32 matrices in this workspace
        with pytest.raises(RuntimeError, match="bad stem"):
            b.optimal([0], [[(5, 3, 2)]])                           # i > j
        with pytest.raises(RuntimeError, match="bad job"):
            b.run_algo([7], "E")
        m = torch.zeros((30, 30), dtype=torch.float64, device="cuda")
        n = len(p.shortseq)
        with pytest.raises(RuntimeError, match="gap map"):
            b.align_accumulate([0], [list(range(n - 1))], m)        # wrong length
        with pytest.raises(RuntimeError, match="gap map"):
            b.align_accumulate([0], [list(range(n))[::-1]], m)      # not increasing
        b.align_accumulate([0], [list(range(2, n + 2))], m)        # a valid one still works afterwards
        torch.cuda.synchronize()
        assert float(m.sum().item()) > 0 and bool(torch.equal(m, m.T))
    with pytest.raises(RuntimeError, match="bpp_term"):
        Batch([p], [[dict(psets[0], bpp=0.5)]], fp32=False)         # bpp paramset without its probability term


def test_matching_sync_path_equals_side_streams(monkeypatch):
    """E/H/N staged on side streams (default) and the synchronous chunked path give the same folds."""
    from squarna_amd.engine import HipEngine
    names, psets = conf("nobpp")
    rng = np.random.default_rng(11)
    recs = [("".join(rng.choice(list("ACGU"), n)), None, None, None, psets, None) for n in (35, 90, 141, 60, 12)]
    a = HipEngine().fold_records(recs)
    monkeypatch.setenv("SQ_ALGO_SYNC", "1")
    b = HipEngine().fold_records(recs)
    assert a == b


def test_annotate_round_larger_than_pinned_output():
    """More than 2^18 stems in one AnnotateStems round: the tail of the list is fetched from device memory."""
    from squarna_amd.engine import HipEngine
    from tests.oracle_engine import OracleEngine
    rng = np.random.default_rng(3)
    w = {"GC": 3.25, "AU": 2.0, "GU": -1.0}
    recs = [("".join(rng.choice(list("ACGU"), 1200)), None, "." * 1200) for _ in range(8)]
    got = HipEngine().yield_stems(recs, w, 2, 0.0)
    exp = OracleEngine().yield_stems(recs, w, 2, 0.0)
    total = 0
    for (gs, gst), (es, est) in zip(got, exp):
        assert gs == es
        g = [(int(a), int(b), int(c), float(d)) for a, b, c, d in zip(gst["i"], gst["j"], gst["len"], gst["bpscore"])]
        assert g == [tuple(x) for x in est]
        total += len(g)
    assert total > (1 << 18)


@pytest.mark.parametrize("minlen,minbp", [(1, 0.0), (1, 3.0), (7, 0.0), (33, 0.0), (40, 0.0)])
def test_minlen_extremes_match_oracle(minlen, minbp):
    """Run extraction at the edges of the 32-row word logic: single-cell stems and stems longer than a word."""
    from squarna_amd.engine import HipEngine
    from oracle import sqrn_oracle as O
    names, psets = conf("greedynobpp")
    ps = [dict(psets[0], minlen=minlen, minbpscore=minbp, bpweights={"GC": 3.25, "AU": 1.25, "GU": -1.25})]
    rng = np.random.default_rng(9 + minlen)
    recs = []
    for n in (30, 95, 140):
        seq = "".join(rng.choice(list("ACGU"), n))
        if minlen >= 33:                                            # plant a 60-bp helix so that long runs exist
            half = "".join(rng.choice(list("ACGU"), 60))
            seq = half + "GAAA" + half[::-1].translate(str.maketrans("ACGU", "UGCA")) + seq
        recs.append((seq, None, None, None, ps, None))
    got = HipEngine().fold_records(recs)
    for r, g in zip(recs, got):
        e = O.SQRNdbnseq(r[0], None, None, None, ps)
        e = [e[0], [[d, list(s), list(p)] for d, s, p in e[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(g, e, (minlen, len(r[0])))


def test_concurrent_batches_equal_sequential_folds():
    """sq_fold_concurrent: several batches folded at the same time give the results of folding them one by one."""
    from squarna_amd.engine import Batch, fold_concurrently
    names, psets = conf("nobpp")
    rng = np.random.default_rng(21)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in rng.integers(20, 150, 24)]
    groups = [seqs[0::3], seqs[1::3], seqs[2::3]]
    exp = []
    for g in groups:
        with Batch([prep(s, None, None) for s in g], [psets] * len(g), fp32=False) as b:
            b.fold(poollim=1000)
            exp.append([b.result(k) for k in range(len(g))])
    batches = [Batch([prep(s, None, None) for s in g], [psets] * len(g), fp32=False) for g in groups]
    try:
        fold_concurrently(batches, poollim=1000)
        got = [[b.result(k) for k in range(len(g))] for b, g in zip(batches, groups)]
    finally:
        for b in batches:
            b.close()
    assert repr(got) == repr(exp)
