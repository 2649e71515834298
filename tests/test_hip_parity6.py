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


def test_fly_bits_and_deferred_wait_equal_their_plain_forms(monkeypatch):
    """poollim = 1 on 400-1,000-nt records with reactivities, restraints (pairs included), separators and gaps: the fold whose
    round kernel forms the bit words itself and whose host waits once, behind the device tail (the defaults), against the
    fold that writes the bit matrices (SQ_NO_FLY_BITS) and the one that waits for the chain first (SQ_NO_DEFER_WAIT): packed
    records byte for byte, and the oracle's structures for a sample."""
    from oracle import sqrn_oracle as O
    from squarna_amd.engine import Batch, Prepared
    names, psets = conf("fastest")
    raw = _chain_records(28, 6001, 400, 1000)
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
