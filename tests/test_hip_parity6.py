"""GPU parity, round 6: default-on paths with subtle preconditions against their plain forms and the oracle -- bit words
formed on the fly inside the persistent round kernel (SQ_NO_FLY_BITS), the wait deferred behind the device tail
(SQ_NO_DEFER_WAIT), jobs whose dense matrix the fill must form (bpp terms, multiplier matrices) on the fly path; the
alignment's rows weighted through the gap map from ONE shared matrix; the array form of the alignment's row preparation.

Every check goes through the C ABI (libsquarna_hip.so); the oracle (oracle/) is the checker.
"""
import os

import numpy as np
import pytest

from tests.test_hip_parity import TOL, conf, _same_fold  # noqa: F401
from tests.test_hip_parity2 import _chain_records
from tests.test_hip_parity4 import _packed

pytestmark = pytest.mark.gpu


def _oracle_fold(O, s, r, x, psets, **kw):
    exp = O.SQRNdbnseq(s, r, x, None, psets, **kw)
    return [exp[0], [[d, list(sc), list(p)] for d, sc, p in exp[1]], ["nan"] * 6, ["nan"] * 7]


@pytest.mark.parametrize("count,nmin,nmax", [(28, 400, 1000), (120, 12, 320)])
def test_fly_bits_and_deferred_wait_equal_their_plain_forms(count, nmin, nmax, monkeypatch):
    """poollim = 1 on 400-1,000-nt records -- and, since the kernel forms its words at every length, on 12-320-nt ones -- with
    reactivities, restraints (pairs included), separators and gaps: the fold whose
    round kernel forms the bit words itself and whose host waits once, behind the device tail (the defaults), against the
    fold that writes the bit matrices (SQ_NO_FLY_BITS) and the one that waits for the chain first (SQ_NO_DEFER_WAIT): packed
    records byte for byte, and the oracle's structures for a sample."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import Batch, Prepared
    names, psets = conf("fastest")
    raw = _chain_records(count, 6001, nmin, nmax)
    prepared = [Prepared(s, r, x) for s, r, x in raw]
    n = len(prepared)
    with Batch(prepared, [psets] * n, max_structs=4 * n, fp32=False) as b:
        b.fold(poollim=1)
        assert b.fold_driver == 1 and (b.fold_paths & 4), (b.fold_driver, b.fold_paths)
        want, res = _packed(b, n), b.results_all()
        for env in ("SQ_NO_FLY_BITS", "SQ_NO_DEFER_WAIT"):
            monkeypatch.setenv(env, "1")
            b.fold(poollim=1)
            assert b.fold_driver == 1 and (b.fold_paths & 4), (env, b.fold_driver, b.fold_paths)
            assert _packed(b, n) == want, env
            monkeypatch.delenv(env)
    for k in (0, 5, 11):
        s, r, x = raw[k]
        _same_fold(res[k][0], _oracle_fold(O, s, r, x, psets, poollim=1), ("fly bits", k))


def _fake_bpp(seq, reacts, M, B):
    n = len(seq)
    rng = np.random.default_rng(n * 7919 + sum(map(ord, seq)))
    return np.triu(rng.random((n, n)) ** 3, 1)


@pytest.mark.parametrize("kind", ["bpp_mul", "bpp_add", "stemmatrix"])
def test_dense_matrix_jobs_on_the_chain_path_equal_the_oracle(kind, monkeypatch):
    """G-only jobs whose cells live in the dense fp64 arena -- a bpp term (multiplied: bpp > 0, added: bpp < 0) or a caller's
    stem matrix -- at poollim = 1 on sequences from 400 nt on, where the round kernel would form its bit words on the fly: the
    fill that forms score x term / score + term must still run (it was skipped, the kernel scored the raw
    term).  Against the oracle and against SQ_NO_FLY_BITS."""
    from squarna_amd import engine as E
    from oracle import sqrn_oracle as O
    names, psets = conf("fastest")
    rng = np.random.default_rng(77)
    seqs = ["".join(rng.choice(list("ACGU"), int(n))) for n in (410, 523, 600)]
    mats = [None] * len(seqs)
    if kind == "stemmatrix":
        mats = []
        for s in seqs:
            m = np.triu(rng.integers(0, 6, (len(s), len(s))).astype(np.float64), 1)
            mats.append(m + m.T)
    else:
        psets = [dict(ps, bpp=0.5 if kind == "bpp_mul" else -1.0) for ps in psets]
    recs = [(s, None, None, None, psets, m) for s, m in zip(seqs, mats)]
    old = E.set_bpp_provider(_fake_bpp)
    O.BPP_SOURCE = _fake_bpp
    try:
        got = E.HipEngine().fold_records(recs, poollim=1)
        monkeypatch.setenv("SQ_NO_FLY_BITS", "1")
        plain = E.HipEngine().fold_records(recs, poollim=1)
        monkeypatch.delenv("SQ_NO_FLY_BITS")
        for r, g, p in zip(recs, got, plain):
            exp = O.SQRNdbnseq(r[0], None, None, None, psets, poollim=1, stemmatrix=r[5])
            exp = [exp[0], [[d, list(s), list(q)] for d, s, q in exp[1]], ["nan"] * 6, ["nan"] * 7]
            _same_fold(g, exp, (kind, len(r[0])))
            _same_fold(p, exp, (kind, "no fly", len(r[0])))
    finally:
        E.set_bpp_provider(old)
        O.BPP_SOURCE = None


@pytest.mark.parametrize("config,count,nmin,nmax", [("fastest", 60, 200, 900), ("alt", 40, 60, 300)])
def test_walk_slices_early_end_and_wave_continuation_equal_the_plain_walk(config, count, nmin, nmax, monkeypatch):
    """ScoreStems' walk of the persistent round kernel in its three forms -- the defaults (slices, an early end at the order
    factor's bound, the wave's continuation on structures of 192 strands and more), the continuation forced on every structure
    (SQ_WAVE_WALK_MIN=1, also with 64 lanes' worth of walks handed over: SQ_WAVE_WALK_LANES=64) and no early end at all
    (SQ_NO_EARLY_WALK) -- give the same packed records on records with reactivities, restraints and separators, pseudoknots
    free of charge (orderpenalty 0: every order factor equal) and expensive (orderpenalty 2), and equal the launched rounds."""
    from squarna_amd.engine import Batch, Prepared
    names, psets = conf(config)
    raw = _chain_records(count, 6116, nmin, nmax)
    prepared = [Prepared(s, r, x) for s, r, x in raw]
    for op in (None, 0.0, 2.0, -0.5):                             # (-0.5: the order factors GROW with the levels -- no early end)
        ps = psets if op is None else [dict(p, orderpenalty=op) for p in psets]
        with Batch(prepared, [ps] * count, max_structs=4 * count, fp32=False) as b:
            b.fold(poollim=1)
            assert b.fold_driver == 1 and (b.fold_paths & 4), (b.fold_driver, b.fold_paths)
            want = _packed(b, count)
            for env in ({"SQ_WAVE_WALK_MIN": "1"}, {"SQ_WAVE_WALK_MIN": "1", "SQ_WAVE_WALK_LANES": "64"}, {"SQ_NO_EARLY_WALK": "1"},
                        {"SQ_NO_ROUNDS": "1"}):
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                b.fold(poollim=1)
                assert _packed(b, count) == want, (config, op, env)
                for k in env:
                    monkeypatch.delenv(k)


def _msa_text(nseq, ncol, seed):
    import random
    rng = random.Random(seed)
    anc = [rng.choice("ACGU") for _ in range(ncol)]
    for _ in range(max(2, ncol // 40)):
        a, ln = rng.randint(0, ncol // 2 - 12), rng.randint(4, 8)
        b = rng.randint(ncol // 2 + 8, ncol - 1)
        for t in range(ln):
            if a + t < b - t - 4:
                anc[b - t] = {"A": "U", "U": "A", "G": "C", "C": "G"}[anc[a + t]]
    rows = []
    for k in range(nseq):
        row = [rng.choice("ACGU") if rng.random() < 0.12 else ch for ch in anc]
        row = ["-" if rng.random() < 0.06 else ch for ch in row]
        rows.append(">s%d\n%s" % (k, "".join(row)))
    return "\n".join(rows) + "\n"


@pytest.mark.parametrize("algos", [None, "GEHN"])
def test_alignment_rows_read_the_shared_matrix_like_their_gathered_slices(algos, tmp_path, monkeypatch):
    """Alignment mode on a 40 x 700 alignment, all three steps: the rows of step 2 reading their weights from the ONE shared
    diagonal-major matrix through the gap map (default) against per-row slices gathered from it (SQ_MUL_GATHER=1, the
    round-3 form), and step 1's rows prepared at once as array code against one record per row (SQ_NO_PACKED_ROWS=1): the same
    text, verbose output included.  algos = GEHN: the rows' E / H / N jobs take RunAlgo's host-driven filters, which gather
    one slice per job on demand."""
    import io
    from squarna_amd import Predict
    path = tmp_path / "ali.afa"
    path.write_text(_msa_text(40, 700, 77))
    kw = dict(inputfile=str(path), alignment=True, step3="u", verbose=True)
    if algos:
        kw["algorithms"] = algos

    def run():
        buf = io.StringIO()
        Predict(write_to=buf, **kw)
        return buf.getvalue()
    want = run()
    assert "Step-3(u)" in want and want.count("\n") > 40
    for env, val in (("SQ_MUL_GATHER", "1"), ("SQ_NO_PACKED_ROWS", "1"), ("SQ_ROUNDS_TLDS", "12")):
        # (SQ_ROUNDS_TLDS=12: the round kernel's LDS lists hold twelve stems -- every row that takes more hands its job to the
        # device pools, as a row does that outgrows the lists sized for two blocks per CU)
        monkeypatch.setenv(env, val)
        assert run() == want, env
        monkeypatch.delenv(env)


def _pool_records(count, seed, nmin, nmax):
    """Random-ACGU records of nmin..nmax nt, a few of them with reactivities / restraints / separators (tools/fuzz_parity.py)."""
    raw = _chain_records(count, seed, nmin, nmax)
    return raw


@pytest.mark.parametrize("config,count,nmin,nmax", [("500nobpp", 24, 300, 620), ("alt", 12, 260, 400)])
def test_pools_on_kept_lists_equal_the_launched_rounds_and_the_oracle(config, count, nmin, nmax, monkeypatch):
    """Pools wider than one on sequences of 257-1,024 nt: every structure reads the list its parent left -- runs, bpscores, the
    finalscores no strand of its own stem comes near -- instead of scanning and scoring anew (sq_fold_paths bit 7).  Against the
    launched round kernels (SQ_NO_POOL_KEPT), the root lists alone (SQ_NO_POOL_KEPT + SQ_POOL_ROOT), a page pool that runs
    dry after a few pages (SQ_KEPT_GB: the children of a structure without a list start from the root list), rounds cut into
    launches of 300 structures (SQ_POOL_CHUNK) -- packed records byte for byte -- and the oracle for a sample."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import Batch, Prepared
    names, psets = conf(config)
    raw = _pool_records(count, 6262, nmin, nmax)
    prepared = [Prepared(s, r, x) for s, r, x in raw]
    n = len(prepared)

    def fold(paths_bit7, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            with Batch(prepared, [psets] * n, max_structs=6000 * n, fp32=False, pool_lists=True) as b:
                b.fold(poollim=100)
                assert b.fold_driver == 2, (env, b.fold_driver)
                assert bool(b.fold_paths & 128) == paths_bit7, (env, b.fold_paths)
                return _packed(b, n), b.results_all()
        finally:
            for k in env:
                monkeypatch.delenv(k)
    want, res = fold(True)
    assert fold(False, SQ_NO_POOL_KEPT="1")[0] == want
    assert fold(False, SQ_NO_POOL_KEPT="1", SQ_POOL_ROOT="1")[0] == want
    assert fold(True, SQ_KEPT_GB="0.02")[0] == want
    assert fold(True, SQ_POOL_CHUNK="300")[0] == want
    for k in (0, 7, n - 1):
        s, r, x = raw[k]
        _same_fold(res[k][0], _oracle_fold(O, s, r, x, psets, poollim=100), (config, "kept lists", k))


def test_packed_results_handed_over_equal_their_copies():
    """sq_result_detach: the records' pinned buffer becomes the caller's -- read-only memoryviews whose bytes are those of
    sq_result_view / sq_result_pack_all --, the batch has no results until it folds again, and it folds again into a new buffer
    while the views of the first fold still stand."""
    import gc
    from squarna_amd.engine import Batch, Prepared
    names, psets = conf("nobpp")
    raw = _chain_records(40, 6363, 30, 200)
    prepared = [Prepared(s, r, x) for s, r, x in raw]
    n = len(prepared)
    with Batch(prepared, [psets] * n, max_structs=64 * n, fp32=False) as b:
        b.fold(poollim=100)
        want = _packed(b, n)
        views = b.detach_packed()
        assert views is not None and [bytes(v) for v in views] == want
        assert all(v.readonly for v in views)
        assert b.detach_packed() is None                              # (nothing left to hand over)
        b.fold(poollim=100)                                           # a new buffer; the first fold's views are untouched
        assert _packed(b, n) == want and [bytes(v) for v in views] == want
        again = b.detach_packed()
        assert [bytes(v) for v in again] == want
    del views, again
    gc.collect()                                                      # (both buffers go back to the library's pinned cache)


@pytest.mark.parametrize("lanes", ["1", "2"])
def test_sub_batches_of_equal_weight_equal_one_batch(lanes, monkeypatch):
    """HipEngine.fold_records_packed with fewer slots than the input's pools want: the records go through sub-batches of equal
    weight (what the first one's pools reached sizes the rest; a second call starts from what the first one learned), one after
    the other or two at a time from threads (SQ_ENGINE_SUBLANES=2) -- the packed records are those of ONE batch, in input order."""
    from squarna_amd import engine as E
    names, psets = conf("nobpp")
    raw = _chain_records(90, 6464, 60, 330)
    recs = [(s, r, x, None, psets, None) for s, r, x in raw]
    want = [bytes(o) for o in E.HipEngine().fold_records_packed(recs, poollim=50)]
    real_cap = E.pool_slot_cap
    monkeypatch.setattr(E, "pool_slot_cap", lambda maxn, want=None: min(real_cap(maxn), 9000))
    monkeypatch.setenv("SQ_ENGINE_SUBLANES", lanes)
    eng = E.HipEngine()
    for call in range(2):
        got = [bytes(o) for o in eng.fold_records_packed(recs, poollim=50)]
        assert got == want, (lanes, call)
    assert eng._pool_scale                                           # (the second call was sized by the first one's pools)
