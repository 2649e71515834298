"""GPU parity, round 2 additions (all through the C ABI of libsquarna_hip.so):

* the graph-level drop-ins Edmonds / Hungarian / Nussinov / RunAlgo against the reference's raw return values;
* bpp != 0 paramsets against fixtures the REAL reference produced on the ViennaRNA stand-in tests/fake_rna.py;
* the `algos=` override (several non-greedy algorithms per paramset) against reference folds;
* alignment step 1 at BASELINE config 5's size (5000 columns) against the CPU oracle, and the full 512 x 5000
  matrix through size-independent properties;
* PredictSharded under the "nccl" backend (RCCL) at world size 1 with the HIP engine, no device argument.
"""
import hashlib
import io
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
DATA = os.path.join(os.path.dirname(HERE), "squarna_amd", "data")
TOL = 1e-5


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def conf(name):
    from squarna_amd.config import ParseConfig, builtin_config
    return ParseConfig(builtin_config(name))


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


def _stems(c):
    return [[[(i + k, j - k) for k in range(ln)], ln, sc] for i, j, ln, sc in c["stems"]]


# ---- a-8 / a-9 drop-ins -----------------------------------------------------------------------------------
def test_matching_dropins_return_what_the_reference_returns():
    """squarna_amd.Edmonds / Hungarian / Nussinov (sq_mwm / sq_lsap / sq_nussinov) == SQRNalgos.Edmonds / Hungarian /
    Nussinov on the same stem lists: identical lists, for Edmonds including every pair's (u, v) orientation."""
    import squarna_amd as S
    from squarna_amd.dbn import SEPS
    cases, raw = load("algos.json"), load("algos_raw.json")
    assert len(cases) == len(raw) > 100
    flipped = 0
    for c, r in zip(cases, raw):
        assert (c["name"], c["algo"]) == (r["name"], r["algo"])
        stems, n = _stems(c), len(c["seq"])
        if c["algo"] == "E":
            got = S.Edmonds(stems)
            flipped += sum(1 for v, w in got if v > w)
        elif c["algo"] == "H":
            got = S.Hungarian(c["seq"], stems, n, SEPS)
        else:
            got = S.Nussinov(c["seq"], stems, n, SEPS)
        assert [list(p) for p in got] == r["raw"], (c["name"], c["algo"])
    assert flipped > 100                        # the orientation really is exercised


def test_matching_dropins_edge_cases():
    import squarna_amd as S
    from squarna_amd.dbn import SEPS
    assert S.Edmonds([]) == [] and S.Hungarian("ACGU", [], 4, SEPS) == [] and S.Nussinov("ACGU", [], 4, SEPS) == []
    # a repeated edge takes the last weight and keeps its first position (networkx add_weighted_edges_from);
    # labels need not be dense
    import networkx as nx
    edges = [(10, 70, 2.0), (70, 30, 3.0), (30, 90, 2.0), (10, 70, 4.0), (90, 10, 1.0)]
    G = nx.Graph()
    G.add_weighted_edges_from(edges)
    exp = sorted(nx.max_weight_matching(G))
    stems = [[[(u, v)], 1, w] for u, v, w in edges]
    assert S.Edmonds(stems, power=1.0) == exp
    with pytest.raises(RuntimeError, match="non-negative"):
        S.Edmonds([[[(-1, 3)], 1, 2.0]])
    with pytest.raises(RuntimeError, match="outside the matrix"):
        S.Hungarian("ACGU", [[[(0, 9)], 1, 2.0]], 4, SEPS)


def test_runalgo_dropin_matches_reference():
    """RunAlgo with the reference's signature (caller matrices in, stemset out) == the reference's stemsets."""
    import squarna_amd as S
    from squarna_amd.dbn import DBNToPairs
    names, psets = conf("nobpp")
    ps = dict(zip(names, psets))
    cases = load("algos.json")
    n = 0
    for c in cases[::3] + cases[1::9]:
        p = ps[c["paramset"]]
        restr = c["restraints"]
        rxs = {i for i, ch in enumerate(restr) if ch in "_+"}
        rl = {i for i, ch in enumerate(restr) if ch == "/"}
        rr = {i for i, ch in enumerate(restr) if ch == "\\"}
        bm, sm = S.BPMatrix(c["seq"], p["bpweights"], rxs, rl, rr, False, c["reacts"])
        got = S.RunAlgo(c["seq"], bm, sm, set(DBNToPairs(restr)), [], p["minlen"], p["minbpscore"], algo=c["algo"],
                        levellimit=3 - int(len(c["seq"]) > 500))
        flat = [[st[0][0][0], st[0][0][1], st[1], st[2]] for st in got]
        assert len(flat) == len(c["stemset"]), (c["name"], c["algo"])
        for g, e in zip(flat, c["stemset"]):
            assert g[:3] == e[:3] and abs(g[3] - e[3]) <= TOL, (c["name"], c["algo"], g, e)
        assert all(st[2] == st[3] and st[4] == '' and st[0] == [(st[0][0][0] + k, st[0][0][1] - k) for k in range(st[1])]
                   for st in got)
        n += 1
    assert n > 40


# ---- f-4: bpp != 0 paramsets, pinned by the real reference on the ViennaRNA stand-in --------------------------
def test_bpp_bpmatrix_golden_gpu(fake_rna):
    import squarna_amd as S
    g = load("bpp.json")
    for c, calls in zip(g["bpmatrix"], g["calls"]):
        restr = c["restraints"]
        rxs = {i for i, ch in enumerate(restr) if ch in "_+"}
        rl = {i for i, ch in enumerate(restr) if ch == "/"}
        rr = {i for i, ch in enumerate(restr) if ch == "\\"}
        bm, sm = S.BPMatrix(c["seq"], c["weights"], rxs, rl, rr, False, c["reacts"], bpp_power=c["bpp_power"])
        assert [[int(i), int(j)] for i, j in zip(*np.nonzero(bm))] == c["bool"], c["seq"]
        exp = np.zeros_like(sm)
        for i, j, v in c["score"]:
            exp[i, j] = v
        assert np.array_equal(sm != 0, exp != 0), (c["seq"], c["bpp_power"])
        assert np.allclose(sm, exp, rtol=1e-12, atol=0), (c["seq"], c["bpp_power"])
        # the product made the same calls into `RNA`, with the same arguments, as the reference did
        mine = json.loads(json.dumps(list(fake_rna.CALLS)))
        assert mine == calls["calls"], (c["seq"], mine, calls["calls"])


def test_bpp_fold_golden_gpu(fake_rna):
    """Whole folds under def.conf (12 paramsets, 7 with probabilities, G/N/H/E) == the reference's tuples."""
    from squarna_amd.engine import HipEngine
    names, psets = conf("def")
    cases = load("bpp.json")["fold"]
    recs, kws = [], []
    for c in cases:
        kw = dict(c["kw"])
        if "rankby" in kw:
            kw["rankby"] = tuple(kw["rankby"])
        if "priority" in kw:
            kw["priority"] = set(kw["priority"])
        recs.append((c["seq"], c["reacts"], c["restraints"], None, psets, None))
        kws.append(kw)
    eng = HipEngine()
    groups = {}
    for k, c in enumerate(cases):                            # each option set of the fixture as ONE batch
        groups.setdefault(json.dumps(c["kw"], sort_keys=True), []).append(k)
    assert len(groups) == 2
    for idx in groups.values():
        out = eng.fold_records([recs[k] for k in idx], **kws[idx[0]])
        for k, o in zip(idx, out):
            _same_fold(o, cases[k]["out"], (cases[k]["tag"], cases[k]["kw"]))
    assert len(cases) >= 20


@pytest.mark.parametrize("tag", ["s16_def", "seq_input_def", "shape_input_def_rb", "SRtest150_def"])
def test_predict_text_default_config_with_bpp_gpu(tag, fake_rna):
    """Predict() with the DEFAULT configuration (def.conf, default priority paramsets): byte-identical to the text the
    reference printed on the same ViennaRNA stand-in."""
    from squarna_amd import Predict
    dig = load("digests.json")[tag]
    assert dig.get("fake_rna")
    kw = dict(dig["args"])
    if "inputfile" in kw:
        kw["inputfile"] = os.path.join(DATA, kw["inputfile"])
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


# ---- `algos=` override: E, H and N stemsets of one paramset ----------------------------------------------------
def test_algos_override_golden_gpu():
    from squarna_amd.engine import HipEngine
    names, psets = conf("nobpp")
    cases = load("fold_algos.json")
    eng = HipEngine()
    for algos in sorted({c["algos"] for c in cases}):
        sel = [c for c in cases if c["algos"] == algos]
        out = eng.fold_records([(c["seq"], c["reacts"], None, None, psets, None) for c in sel], algos=set(algos),
                               **sel[0]["kw"])
        for c, o in zip(sel, out):
            _same_fold(o, c["out"], (algos, c["seq"]))
    assert len(cases) >= 40


# ---- alignment step 1 at BASELINE config 5's size (512 sequences x 5000 columns) -------------------------------
def _msa(nseq, ncol, seed=5000, mut=0.15, gap=0.10):
    """one random ancestor, per-site mutation 0.15, per-site gap 0.10."""
    rng = np.random.default_rng(seed)
    anc = rng.choice(list("ACGU"), ncol)
    rows = []
    for _ in range(nseq):
        row = anc.copy()
        m = rng.random(ncol) < mut
        row[m] = rng.choice(list("ACGU"), int(m.sum()))
        row[rng.random(ncol) < gap] = "-"
        rows.append("".join(row))
    return rows


def _oracle_matrix(recs, w, minlen, minbp):
    """The reference's parent-side loop (SQRNdbnali.py:233-237) over the oracle's YieldStems: sequences, stems and
    cells in order; np.add.at applies the additions one by one, so every cell sees the reference's fp64 order."""
    from tests.oracle_engine import OracleEngine
    L = len(recs[0][0])
    exp = np.zeros((L, L))
    for rec in recs:                                              # one sequence at a time: bounded memory
        (short, stems), = OracleEngine().yield_stems([rec], w, minlen, minbp)
        if not stems:
            continue
        cols = np.array([c for c, ch in enumerate(rec[0]) if ch not in "-.~"], np.int64)
        st = np.array([s[:3] for s in stems], np.int64)
        sc = np.array([s[3] for s in stems], np.float64)
        si, sj, sl = st[:, 0], st[:, 1], st[:, 2]
        k = np.arange(int(sl.sum())) - np.repeat(np.cumsum(sl) - sl, sl)
        v, ww, val = cols[np.repeat(si, sl) + k], cols[np.repeat(sj, sl) - k], np.repeat(sc, sl)
        np.add.at(exp, (v, ww), val)
        np.add.at(exp, (ww, v), val)
    return exp


@pytest.mark.parametrize("with_reacts,nseq", [(False, 24), (True, 8)])
def test_align_config5_columns_vs_oracle(with_reacts, nseq):
    """5000-column alignment (config 5's width, ali.conf weights): the device column matrix == the sequential
    accumulation of the CPU oracle's stems, bit for bit without reactivities (the one-launch atomic path) and within
    1e-13 with float reactivities (the ordered one-launch-per-sequence path; only the documented sqrt-vs-pow ulp)."""
    from squarna_amd.engine import HipEngine
    names, psets = conf("ali")
    ps = psets[0]
    rows = _msa(nseq, 5000, seed=5000 + with_reacts)
    rng = np.random.default_rng(9)
    recs = []
    for r in rows:
        reacts = [float(x) for x in rng.random(len(r))] if with_reacts else None
        recs.append((r, reacts, "." * len(r)))
    exp = _oracle_matrix(recs, ps["bpweights"], ps["minlen"], ps["minbpscore"])
    eng = HipEngine()
    got_t = eng.stem_matrix(recs, ps["bpweights"], ps["minlen"], ps["minbpscore"])
    got = got_t.cpu().numpy()
    assert got.shape == (5000, 5000)
    if with_reacts:
        assert np.array_equal(got != 0, exp != 0)
        assert np.allclose(got, exp, rtol=1e-13, atol=0)
        exp = got
    else:
        assert np.array_equal(got, exp)
    thr = ps["minbpscore"] * nseq
    idx, val = eng.matrix_cells(got_t, thr)
    flat = exp.reshape(-1)
    hit = np.flatnonzero(flat >= thr)
    hit = hit[(hit % 5000) - (hit // 5000) >= 4]
    assert len(hit) > 1000
    assert np.array_equal(idx, hit) and np.array_equal(val, flat[hit])


def test_align_config5_full_size_properties():
    """512 x 5000 (BASELINE config 5) through properties: symmetric; equal to the sum of the matrices of its
    32-sequence chunks (every sum is exact for ali.conf's dyadic weights, so any grouping gives the same bits);
    the device threshold selection == numpy's on the same matrix; first rows == an independent 16-sequence run."""
    import torch
    from squarna_amd.engine import HipEngine
    names, psets = conf("ali")
    ps = psets[0]
    rows = _msa(512, 5000)
    recs = [(r, None, "." * 5000) for r in rows]
    eng = HipEngine()
    args = (ps["bpweights"], ps["minlen"], ps["minbpscore"])
    full = eng.stem_matrix(recs, *args)
    assert tuple(full.shape) == (5000, 5000) and bool(torch.equal(full, full.T))
    acc = torch.zeros_like(full)
    for lo in range(0, 512, 32):
        acc += eng.stem_matrix(recs[lo:lo + 32], *args)
    assert bool(torch.equal(acc, full))
    assert float(full.sum().item()) > 0
    thr = ps["minbpscore"] * 512
    idx, val = eng.matrix_cells(full, thr)
    flat = full.cpu().numpy().reshape(-1)
    hit = np.flatnonzero(flat >= thr)
    hit = hit[(hit % 5000) - (hit // 5000) >= 4]
    assert len(hit) > 10000
    assert np.array_equal(idx, hit) and np.array_equal(val, flat[hit])


# ---- multi-GPU driver on the product engine: "nccl" == RCCL, world size 1 (one GPU on the test box) ----------------
def test_predict_sharded_under_nccl_world1(tmp_path):
    """PredictSharded as INTEGRATION.md documents it (no device argument) under the RCCL backend with the HIP engine:
    the single-sequence gather (all_gather of packed text) and the alignment path (all_reduce of the device matrix,
    all_gather_object of the step-2 folds) give the reference's bytes."""
    import torch
    import torch.distributed as dist
    from squarna_amd.parallel import PredictSharded
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method="file://" + str(tmp_path / "rdzv"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        dig = load("digests.json")
        for tag in ("SRtest150_fastest", "seq_input_nobpp", "ali_input_a", "ali_input_a_entropy"):
            kw = dict(dig[tag]["args"])
            if "inputfile" in kw:
                kw["inputfile"] = os.path.join(DATA, kw["inputfile"])
            buf = io.StringIO()
            PredictSharded(write_to=buf, **kw)
            with open(os.path.join(GOLDEN, "text", tag + ".txt")) as f:
                assert buf.getvalue() == f.read(), tag
        # synonyms resolve as in Predict (later aliases win): i= over inputfile=, c= over configfile=
        buf = io.StringIO()
        PredictSharded(write_to=buf, inputfile="/nonexistent", i="datasets/SRtest150.fas", inputformat="qf",
                       configfile="nobpp", c="fastest", pl=1)
        with open(os.path.join(GOLDEN, "text", "SRtest150_fastest_pl1.txt")) as f:
            assert buf.getvalue() == f.read()
    finally:
        dist.destroy_process_group()


# ---- device-chained rounds (poollim == 1, sq_chain.hip): against the oracle and against the host-driven loop --------
def _chain_records(count, seed, nmin, nmax):
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(HERE), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    old = os.environ.get("FUZZ_NMIN"), os.environ.get("FUZZ_NMAX")
    os.environ["FUZZ_NMIN"], os.environ["FUZZ_NMAX"] = str(nmin), str(nmax)
    try:
        return fz.make(count, seed)
    finally:
        for k, v in zip(("FUZZ_NMIN", "FUZZ_NMAX"), old):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _fold_both_drivers(recs, **kw):
    from squarna_amd.engine import HipEngine
    assert "SQ_NO_CHAIN" not in os.environ
    chained = HipEngine().fold_records(recs, **kw)
    os.environ["SQ_NO_CHAIN"] = "1"
    try:
        hosted = HipEngine().fold_records(recs, **kw)
    finally:
        del os.environ["SQ_NO_CHAIN"]
    return chained, hosted


@pytest.mark.parametrize("config,count,nmin,nmax,sample", [("fastest", 400, 5, 420, 60), ("nobpp", 160, 12, 260, 30),
                                                           ("greedynobpp", 120, 12, 200, 24)])
def test_chained_rounds_match_host_loop_and_oracle(config, count, nmin, nmax, sample):
    """poollim=1: the rounds chained on the device (stem choice, strand insertion, pseudoknot levels, retirement) give
    exactly what the host-driven loop gives (same kernels, so every score bit-for-bit), and what the oracle gives."""
    from oracle import sqrn_oracle as O
    names, psets = conf(config)
    raw = _chain_records(count, 4242, nmin, nmax)
    recs = [(s, r, x, None, psets, None) for s, r, x in raw]
    chained, hosted = _fold_both_drivers(recs, poollim=1)
    for k, (a, b) in enumerate(zip(chained, hosted)):
        assert a[0] == b[0] and a[1] == b[1], (config, k, raw[k][0], a[:2], b[:2])
    for k in range(sample):
        s, r, x = raw[k]
        exp = O.SQRNdbnseq(s, r, x, None, psets, poollim=1)
        exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(chained[k], exp, (config, "chained", k))


def test_chained_rounds_maxstemnum_and_long_pseudoknotted():
    """Retirement by stem count (maxstemnum 0 / 1 / 3, SQRNdbnseq.py:1168-1174) and long random sequences whose
    structures stack several pseudoknot levels (the device-side level rule), chained vs host loop vs oracle."""
    from oracle import sqrn_oracle as O
    names, psets = conf("fastest")
    rng = np.random.default_rng(77)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in rng.integers(60, 400, 48)]
    for msn in (0, 1, 3):
        ps = [dict(psets[0], maxstemnum=msn)]
        recs = [(s, None, None, None, ps, None) for s in seqs]
        chained, hosted = _fold_both_drivers(recs, poollim=1)
        assert [c[:2] for c in chained] == [h[:2] for h in hosted], msn
        for k in range(10):
            exp = O.SQRNdbnseq(seqs[k], None, None, None, ps, poollim=1)
            exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
            _same_fold(chained[k], exp, ("maxstemnum", msn, k))
    # long, GC-rich: many crossing stems
    longs = ["".join(rng.choice(list("ACGU"), int(n), p=[0.15, 0.35, 0.35, 0.15])) for n in (900, 1300, 1700)]
    ps = [dict(psets[0], orderpenalty=0.0, minlen=3, minbpscore=6)]          # pseudoknots are free: deep level stacks
    recs = [(s, None, None, None, ps, None) for s in longs]
    chained, hosted = _fold_both_drivers(recs, poollim=1)
    assert [c[:2] for c in chained] == [h[:2] for h in hosted]
    deep = max(sum(ch in c[1][0][0] for ch in "[{<A") for c in chained)
    assert deep >= 3, "expected structures with at least four bracket levels"
    exp = O.SQRNdbnseq(longs[0], None, None, None, ps, poollim=1)
    exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
    _same_fold(chained[0], exp, ("long", 0))


# ---- device pools (poollim > 1, sq_pool.hip): against the host-driven loop and the oracle ----------------------------
def _fold_pool_and_host(recs, **kw):
    from squarna_amd.engine import HipEngine
    assert "SQ_NO_POOL" not in os.environ
    pooled = HipEngine().fold_records(recs, **kw)
    os.environ["SQ_NO_POOL"] = "1"
    try:
        hosted = HipEngine().fold_records(recs, **kw)
    finally:
        del os.environ["SQ_NO_POOL"]
    return pooled, hosted


@pytest.mark.parametrize("config,count,nmin,nmax,poollim,sample", [("nobpp", 200, 12, 220, 1000, 30), ("greedynobpp", 200, 12, 260, 3, 30),
                                                                   ("greedynobpp", 120, 30, 300, 40, 16), ("fastest", 150, 5, 300, 7, 20)])
def test_device_pools_match_host_loop_and_oracle(config, count, nmin, nmax, poollim, sample):
    """poollim > 1: ChooseStems' conflict filter, the children's slots, cursize / cursubopt, the stopper and the order of
    finstemsets booked on the device give exactly what the host-driven loop gives (same kernels: every score bit for
    bit, every structure in the same rank) and what the oracle gives."""
    from oracle import sqrn_oracle as O
    names, psets = conf(config)
    raw = _chain_records(count, 777 + poollim, nmin, nmax)
    recs = [(s, r, x, None, psets, None) for s, r, x in raw]
    pooled, hosted = _fold_pool_and_host(recs, poollim=poollim)
    for k, (a, b) in enumerate(zip(pooled, hosted)):
        assert a[0] == b[0] and a[1] == b[1], (config, poollim, k, raw[k][0], a[:2], b[:2])
    for k in range(sample):
        s, r, x = raw[k]
        exp = O.SQRNdbnseq(s, r, x, None, psets, poollim=poollim)
        exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(pooled[k], exp, (config, "pooled", poollim, k))


def test_device_pools_overflow_falls_back_and_maxstemnum():
    """A fold whose pools outgrow the device slots is repeated by the host loop (same results); maxstemnum 0 / 2 / 3
    exercises the 'full' children (SQRNdbnseq.py:1123-1129: moved to finstemsets at the start of the next round)."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf("greedynobpp")
    raw = _chain_records(60, 99, 40, 200)
    recs = [(s, r, x, None, psets, None) for s, r, x in raw]
    normal = HipEngine().fold_records(recs, poollim=1000)
    os.environ["SQ_POOL_SLOTS"] = str(2 * len(recs) + 3)            # room for the first generation only
    try:
        eng = HipEngine()
        cramped = eng.fold_records(recs, poollim=1000)
        assert eng.last_fold_driver == 3                      # device pools gave up, the host loop repeated the fold
    finally:
        del os.environ["SQ_POOL_SLOTS"]
    assert [c[:2] for c in cramped] == [n[:2] for n in normal]
    for msn in (0, 2, 3):
        ps = [dict(p, maxstemnum=msn) for p in psets]
        recs2 = [(s, None, None, None, ps, None) for s, r, x in raw[:24]]
        pooled, hosted = _fold_pool_and_host(recs2, poollim=1000)
        assert [c[:2] for c in pooled] == [h[:2] for h in hosted], msn
        for k in range(6):
            exp = O.SQRNdbnseq(raw[k][0], None, None, None, ps, poollim=1000)
            exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
            _same_fold(pooled[k], exp, ("pool maxstemnum", msn, k))


def test_round_output_overflow_splits_the_round():
    """A round that emits more stems than the round output holds (AnnotateStems passes of E / H / N over thousands of
    records; wide pools) is repeated in smaller chunks: same results as with room for everything, for both loop drivers."""
    from squarna_amd.engine import HipEngine
    names, psets = conf("nobpp")
    raw = _chain_records(80, 4242, 30, 180)
    recs = [(s, r, x, None, psets, None) for s, r, x in raw]
    normal = HipEngine().fold_records(recs, poollim=20)
    os.environ["SQ_OUT_CAP"] = "3000"
    try:
        cramped = HipEngine().fold_records(recs, poollim=20)
        os.environ["SQ_NO_POOL"] = "1"
        cramped_host = HipEngine().fold_records(recs, poollim=20)
    finally:
        del os.environ["SQ_OUT_CAP"]
        os.environ.pop("SQ_NO_POOL", None)
    assert [c[:2] for c in cramped] == [n[:2] for n in normal]
    assert [c[:2] for c in cramped_host] == [n[:2] for n in normal]


@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_device_pools_in_chunks(chunk):
    """Generations larger than the candidate arena go through state .. choose in chunks that reuse the arena
    (SQ_POOL_CHUNK forces small ones): same structures, scores and ranks as one chunk and as the host loop."""
    from squarna_amd.engine import HipEngine
    names, psets = conf("nobpp")
    raw = _chain_records(40, 1717, 20, 200)
    recs = [(s, r, x, None, psets, None) for s, r, x in raw]
    pooled, hosted = _fold_pool_and_host(recs, poollim=30)
    os.environ["SQ_POOL_CHUNK"] = str(chunk)
    try:
        eng = HipEngine()
        chunked = eng.fold_records(recs, poollim=30)
        assert eng.last_fold_driver == 2                      # (not a fallback to the host loop)
    finally:
        del os.environ["SQ_POOL_CHUNK"]
    assert [c[:2] for c in chunked] == [p[:2] for p in pooled]
    assert [c[:2] for c in chunked] == [h[:2] for h in hosted]


def test_result_limit_shows_the_top_structures_only():
    """sq_result_limit / fold_records(keep=k): the first k structures of every record, consensus and metrics unchanged."""
    from squarna_amd.engine import HipEngine
    names, psets = conf("nobpp")
    raw = _chain_records(30, 99, 20, 160)
    recs = [(s, r, x, None, psets, None) for s, r, x in raw]
    full = HipEngine().fold_records(recs, poollim=50)
    top = HipEngine().fold_records(recs, poollim=50, keep=3)
    assert any(len(f[1]) > 3 for f in full)
    for f, t in zip(full, top):
        assert t[0] == f[0] and t[1] == f[1][:3] and len(t[1]) == min(3, len(f[1]))


@pytest.mark.parametrize("config,poollim", [("fastest", 1), ("greedynobpp", 6)])
def test_many_restraint_pairs(config, poollim):
    """More restraint base pairs than one byte counts (alignment mode restrains the second iteration of step 1 by the
    structure of the first, SQRNdbnali.py:359-362: hundreds of pairs on long alignments), most of them on ONE anti-diagonal (a long hairpin: i + j constant) plus helices
    elsewhere, some complementary and some not (SQRNdbnseq.py:438-443: the restraint cell stays pairable only where the
    bases pair), also with hardrest: the fold equals the oracle's."""
    import numpy as np
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf(config)
    rng = np.random.default_rng(5)
    comp = {"A": "U", "U": "A", "G": "C", "C": "G"}
    recs = []
    for n, step in ((1300, 2), (700, 1), (900, 3)):
        seq = list(rng.choice(list("ACGU"), n))
        restr = ["."] * n
        npairs = 0
        for i in range(0, n // 2 - 4, step):                   # pairs (i, n-1-i): one anti-diagonal
            if rng.random() < 0.85:
                restr[i], restr[n - 1 - i] = "(", ")"
                npairs += 1
                if rng.random() < 0.7:
                    seq[n - 1 - i] = comp[seq[i]]
        assert npairs > 254 or step == 3
        recs.append(("".join(seq), None, "".join(restr)))
    for hardrest in (False, True):
        got = HipEngine().fold_records([(s, r, x, None, psets, None) for s, r, x in recs], poollim=poollim, hardrest=hardrest)
        for k, (s, r, x) in enumerate(recs):
            exp = O.SQRNdbnseq(s, r, x, None, psets, poollim=poollim, hardrest=hardrest)
            exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
            _same_fold(got[k], exp, (config, "restraints", hardrest, k))


def test_shared_device_stem_matrix_equals_per_record_matrices():
    """Alignment step 2 (SQRNdbnseq.py:1031-1034,1084-1085): one L x L matrix on the device + column maps
    (mul_matrix_dev, gathered by sq_gather_mul_kernel) gives what one host matrix per record gives, and the oracle."""
    import numpy as np
    import torch
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf("ali")
    rng = np.random.default_rng(21)
    L = 150
    anc = rng.choice(list("ACGU"), L)
    rows = []
    for _ in range(12):
        row = anc.copy()
        m = rng.random(L) < 0.15
        row[m] = rng.choice(list("ACGU"), int(m.sum()))
        row[rng.random(L) < 0.1] = "-"
        rows.append("".join(row))
    sm = rng.random((L, L)) * (rng.random((L, L)) < 0.3)
    sm = (sm + sm.T) / np.max(sm + sm.T) * 5
    on_host = HipEngine().fold_records([(r, None, None, None, psets, sm) for r in rows], poollim=30)
    dev = torch.from_numpy(sm).to("cuda").contiguous()
    on_dev = HipEngine().fold_records([(r, None, None, None, psets, dev) for r in rows], poollim=30)
    assert [d[:2] for d in on_dev] == [h[:2] for h in on_host]
    for k in range(4):
        exp = O.SQRNdbnseq(rows[k], None, None, None, psets, poollim=30, stemmatrix=sm)
        exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(on_dev[k], exp, ("shared stem matrix", k))


def test_streamed_edmonds_results_are_stable_over_many_folds():
    """Edmonds results are collected job by job while the kernel is still running (per-job flags in pinned memory).  Before
    the publishers read host memory back ahead of the flag store (sq_hostflag.h), about one result block in 10^5 was read
    while it was still arriving: 3 folds in 1,000 of a batch like this one gave a different structure for some record.
    400 folds of the same 32 records, every one equal to the first, the first equal to the oracle."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import HipEngine
    names, psets = conf("hungariannobpp")
    raw = _chain_records(32, 20611, 60, 170)
    recs = [(s, None, None, None, psets, None) for s, r, x in raw]
    kw = dict(algos=frozenset("E"), toplim=1, poollim=2, rankby=(1, 2, 0))
    first = HipEngine().fold_records(recs, **kw)
    for k in range(0, 32, 4):
        exp = O.SQRNdbnseq(raw[k][0], None, None, None, psets, **kw)
        exp = [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]
        _same_fold(first[k], exp, ("streamed E", k))
    for rep in range(400):
        again = HipEngine().fold_records(recs, **kw)
        assert [a[:2] for a in again] == [f[:2] for f in first], rep


def test_predict_sharded_world2_with_the_hip_engine():
    """World size 2 with the PRODUCT engine: two ranks under gloo, both folding on cuda:0 (a one-GPU box cannot run RCCL
    at world size 2): the shard / gather of the single-sequence mode and the all_reduce + sharded step 2 of alignment
    mode give the golden texts (tools/world2_hip_check.py)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29743", os.path.join(root, "tools", "world2_hip_check.py")]
    r = subprocess.run(cmd, env=env, cwd=root, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.count("identical to the golden text") == 5, r.stdout[-2000:]
