"""Drop-in function surface of the reference's ``SQRNdbnseq`` module, routed to the HIP engine.

Same names, argument meaning and return shapes as ``src/SQUARNA/SQRNdbnseq.py``:
``BPMatrix`` (:258), ``AnnotateStems`` (:427), ``OptimalStems`` (:792), ``SQRNdbnseq`` (:973),
``RunSQRNdbnseq`` (:1289).  All arithmetic of the hot path runs on the GPU through
libsquarna_hip.so (see engine.py); this module only converts Python records and prints.
"""
import sys

import numpy as np

from .dbn import (GAPS, SEPS, ReactDict, ProcessReacts, EncodedReactivities, DBNToPairs, UnAlign,
                  ReAlign, ParseRestraints, PairsToStems, BRACKETS)
from . import engine as _engine


# ------------------------------------------------------------------ per-call shims (unit parity)
def _pset(weights, minlen=2.0, minbpscore=0.0, **kw):
    ps = dict(bpweights=weights, bpp=0, algorithms={"G"}, suboptmax=1.0, suboptmin=1.0, suboptsteps=1.0,
              minlen=minlen, minbpscore=minbpscore, minfinscorefactor=1.0, bracketweight=-2.0,
              distcoef=0.09, orderpenalty=1.0, loopbonus=0.125, maxstemnum=1e6)
    ps.update(kw)
    return ps


def _restraint_line(n, rxs=(), rlefts=(), rrights=(), rbps=()):
    line = ['.'] * n
    for i in rxs:
        line[i] = '_'
    for i in rlefts:
        line[i] = '/'
    for i in rrights:
        line[i] = '\\'
    for k, (v, w) in enumerate(sorted(rbps)):
        if k >= len(BRACKETS):
            raise ValueError("too many restraint base pairs")
        line[v], line[w] = BRACKETS[k][0], BRACKETS[k][1]
    return ''.join(line)


def BPMatrix(seq, weights, rxs, rlefts, rrights, interchainonly=False, reacts=None, bpp_power=0,
             M=1.8, B=-0.6):
    """(bpboolmatrix, bpscorematrix), dense N x N float64 -- SQRNdbnseq.py:258-367.
    Computed on the GPU in fp64 (sq_bpmatrix_read).  bpp_power != 0 takes ViennaRNA's base-pair
    probabilities from the host (engine.vienna_bpp, i.e. `import RNA`); the term (bppm/max)**|p| is uploaded
    with the batch and the fill kernel applies it on the device: scoremat *= term (p > 0) or += term (p < 0)
    (SQRNdbnseq.py:350-364)."""
    n = len(seq)
    prep = _engine.Prepared(seq, list(reacts) if reacts is not None else None,
                            _restraint_line(n, rxs, rlefts, rrights))
    prep.shortseq = seq                     # BPMatrix takes the sequence as it is (already gap-free)
    ps = _pset(weights, bpp=float(bpp_power))
    term = _engine.bpp_terms([prep], [[ps]], M, B) if bpp_power else None
    if term is None or term[0] is None:     # no probabilities (bpp == 0, or max(bppm) == 0: the matrix stays as it is)
        ps = _pset(weights)
        with _engine.Batch([prep], [[ps]], interchainonly=interchainonly, fp32=False) as b:
            return b.bpmatrix(0)
    with _engine.Batch([prep], [[ps]], interchainonly=interchainonly, bpp=term) as b:
        b.fill()                            # forms the weighted matrix in the dense arena
        return b.bpmatrix(0)


def _stems_ijl(rstems):
    return [(st[0][0][0], st[0][0][1], st[1]) for st in rstems]


def _stem_record(i, j, ln, *rest):
    return [[(i + k, j - k) for k in range(ln)], ln] + list(rest)


def AnnotateStems(bpboolmatrix, bpscorematrix, rbps, rstems, minlen, minscore, diff=0, span=-1):
    """All maximal runs of allowed pairs along the anti-diagonals with len >= minlen and
    score >= minscore, in the reference's order -- SQRNdbnseq.py:427-495."""
    if diff != 0 or span != -1:
        raise NotImplementedError("diff/span are never passed by the reference (SQRNdbnseq.py:427-428)")
    n = bpboolmatrix.shape[0]
    prep = _engine.Prepared('N' * n, None, _restraint_line(n, rbps=rbps))
    ps = _pset({}, minlen=minlen, minbpscore=minscore)
    with _engine.Batch([prep], [[ps]], ext=[(np.asarray(bpboolmatrix, float), np.asarray(bpscorematrix, float))]) as b:
        out = b.optimal([0], [_stems_ijl(rstems)], mode=1)[0]
    return [_stem_record(i, j, ln, sc) for i, j, ln, sc, _ in out]


def OptimalStems(seq, rstems, bpboolmatrix, bpscorematrix, reacts, rbps=set(), subopt=1.0, minlen=2,
                 minbpscore=6, minfinscore=0, bracketweight=1.0, distcoef=0.1, orderpenalty=0.0,
                 loopbonus=0.0):
    """The top stems for one greedy round -- SQRNdbnseq.py:792-833."""
    n = len(seq)
    prep = _engine.Prepared('N' * n, None, _restraint_line(n, rbps=rbps))
    prep.shortseq = seq
    ps = _pset({}, minlen=minlen, minbpscore=minbpscore, bracketweight=bracketweight, distcoef=distcoef,
               orderpenalty=orderpenalty, loopbonus=loopbonus,
               minfinscorefactor=(minfinscore / minbpscore) if minbpscore else 0.0)
    if minbpscore == 0 and minfinscore != 0:
        raise ValueError("minfinscore with minbpscore == 0 is not representable")
    with _engine.Batch([prep], [[ps]], ext=[(np.asarray(bpboolmatrix, float), np.asarray(bpscorematrix, float))]) as b:
        out = b.optimal([0], [_stems_ijl(rstems)], subopt=[subopt], mode=0)[0]
    return [_stem_record(i, j, ln, bps, fin, '') for i, j, ln, bps, fin in out]


# ------------------------------------------------------------------ a-8 / a-9: RunAlgo and the matching functions
def _device_workspace(nbytes):
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("squarna_amd needs an AMD GPU (MI355X / gfx950): torch.cuda is not available and there "
                           "is no CPU fallback")
    ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
    base = ws.data_ptr()
    return ws, (base + 255) // 256 * 256, torch.cuda.current_stream().cuda_stream


def _cells_of(stems, matrix, N, value):
    """(v, w, value) triples of the reference's two input forms: stems (cells of every stem, `value(stem score)`)
    or a matrix (upper cells > 0, `value(cell)`)."""
    if matrix is None:
        return [(v, w, value(st[2])) for st in stems for v, w in st[0]]
    matrix = np.asarray(matrix)
    n = matrix.shape[0] if N is None else N
    return [(v, w, value(matrix[v, w])) for v in range(n - 1) for w in range(v + 1, n) if matrix[v, w] > 0]


def Edmonds(stems, power=1.7, matrix=None):
    """Maximum-weight matching of the stem cells -- SQRNalgos.py:96-110.  networkx.max_weight_matching restated
    step by step on the GPU (sq_mwm): same pairs, same (u, v) orientation, same sorted order as the reference
    returns, including which of several optimal matchings is found."""
    from . import _lib
    import ctypes as C
    L = _lib.load()
    edges = _cells_of(stems, matrix, None, lambda x: x ** power)          # weights through the host libm, as the reference
    m = len(edges)
    if m == 0:
        return []
    eu = np.array([e[0] for e in edges], np.int32)
    ev = np.array([e[1] for e in edges], np.int32)
    ew = np.array([e[2] for e in edges], np.float64)
    off = np.array([0, m], np.int64)
    nbytes = C.c_size_t(0)
    _lib.check(L.sq_mwm_workspace_bytes(1, off.ctypes.data, eu.ctypes.data, ev.ctypes.data, C.byref(nbytes)))
    ws, ptr, stream = _device_workspace(nbytes.value)
    pairs = np.zeros(2 * m + 2, np.int32)
    poff = np.zeros(2, np.int64)
    _lib.check(L.sq_mwm(1, off.ctypes.data, eu.ctypes.data, ev.ctypes.data, ew.ctypes.data, pairs.ctypes.data, m + 1,
                        poff.ctypes.data, C.c_void_p(ptr), C.c_size_t(nbytes.value), C.c_void_p(stream)))
    return [(int(pairs[2 * k]), int(pairs[2 * k + 1])) for k in range(int(poff[1]))]


def Hungarian(seq, stems, N, seps, minloop=3, power=1.7, matrix=None):
    """Linear-sum-assignment matching of the stem cells -- SQRNalgos.py:113-135.  scipy's
    linear_sum_assignment restated step by step on the GPU (sq_lsap); the mutual-pair filter (:130-133) is
    O(N) host work."""
    from . import _lib
    import ctypes as C
    L = _lib.load()
    if matrix is None:
        cells = _cells_of(stems, None, N, lambda x: x ** power)           # mat[v,w] = mat[w,v] = -(score ** power)
    else:                                                                  # mat = -(matrix ** power)
        mat = np.asarray(matrix, dtype=float)
        if not np.array_equal(mat, mat.T) or np.any(np.diag(mat) != 0):
            raise NotImplementedError("Hungarian(matrix=...): sq_lsap takes symmetric matrices with a zero diagonal")
        cells = [(v, w, mat[v, w] ** power) for v in range(N - 1) for w in range(v + 1, N) if mat[v, w] != 0]
    if N == 0:
        return []
    cv = np.array([c[0] for c in cells], np.int32)
    cw = np.array([c[1] for c in cells], np.int32)
    wt = np.array([c[2] for c in cells], np.float64)
    n = np.array([N], np.int32)
    off = np.array([0, len(cells)], np.int64)
    nbytes = C.c_size_t(0)
    _lib.check(L.sq_lsap_workspace_bytes(1, n.ctypes.data, off.ctypes.data, C.byref(nbytes)))
    ws, ptr, stream = _device_workspace(nbytes.value)
    sol = np.zeros(N, np.int32)
    _lib.check(L.sq_lsap(1, n.ctypes.data, off.ctypes.data, cv.ctypes.data, cw.ctypes.data, wt.ctypes.data,
                         sol.ctypes.data, C.c_void_p(ptr), C.c_size_t(nbytes.value), C.c_void_p(stream)))
    nonzero = {(int(v), int(w)) for v, w, x in cells if -x != 0}
    nonzero |= {(w, v) for v, w in nonzero}
    out = []
    for k in range(N):                                                     # :130-133
        j = int(sol[k])
        if j < 0 or not (k < j - minloop or (k < j and any(ch in seps for ch in seq[k + 1:j]))):
            continue
        if int(sol[j]) == k and (k, j) in nonzero:
            out.append((k, j))
    return out


def Nussinov(seq, stems, N, seps, minloop=3, matrix=None):
    """Nussinov DP over the stem cells + BackTrack -- SQRNalgos.py:44-93, on the GPU (sq_nussinov)."""
    from . import _lib
    from .dbn import encode_seq
    import ctypes as C
    if minloop != 3 or set(seps) != SEPS:
        raise NotImplementedError("the device kernel is built for minloop = 3 and the separators ';' '&' "
                                  "(the only values the reference passes, SQRNdbnseq.py:565)")
    L = _lib.load()
    cells = _cells_of(stems, matrix, N, lambda x: x)
    if N == 0:
        return []
    cv = np.array([c[0] for c in cells], np.int32)
    cw = np.array([c[1] for c in cells], np.int32)
    sc = np.array([c[2] for c in cells], np.float64)
    n = np.array([N], np.int32)
    off = np.array([0, len(cells)], np.int64)
    codes = np.frombuffer(encode_seq(seq), np.uint8)
    assert len(codes) == N, "sequence length differs from N"
    nbytes = C.c_size_t(0)
    _lib.check(L.sq_nussinov_workspace_bytes(1, n.ctypes.data, off.ctypes.data, C.byref(nbytes)))
    ws, ptr, stream = _device_workspace(nbytes.value)
    pairs = np.zeros(2 * (N + 4), np.int32)
    poff = np.zeros(2, np.int64)
    _lib.check(L.sq_nussinov(1, n.ctypes.data, codes.ctypes.data, off.ctypes.data, cv.ctypes.data, cw.ctypes.data,
                             sc.ctypes.data, pairs.ctypes.data, N + 4, poff.ctypes.data, C.c_void_p(ptr),
                             C.c_size_t(nbytes.value), C.c_void_p(stream)))
    return [(int(pairs[2 * k]), int(pairs[2 * k + 1])) for k in range(int(poff[1]))]


def RunAlgo(seq, bpboolmatrix, bpscorematrix, restbps, rstems, minlen, minscore, algo="E", levellimit=3):
    """Single-sequence prediction by Edmonds / Hungarian / Nussinov over the stems of the given matrices, with the
    reference's stem filters -- SQRNdbnseq.py:548-595.  One call of sq_run_algos on a one-job batch that carries
    the caller's matrices: AnnotateStems, the matching kernel and the filters all run inside the library."""
    if rstems:
        raise NotImplementedError("RunAlgo with pre-selected stems: the reference always passes [] (SQRNdbnseq.py:1097)")
    if algo not in ("E", "H", "N"):
        return []
    n = len(seq)
    prep = _engine.Prepared('N' * n, None, _restraint_line(n, rbps=restbps))
    prep.shortseq = seq
    ps = _pset({}, minlen=minlen, minbpscore=minscore, algorithms={algo})
    with _engine.Batch([prep], [[ps]], ext=[(np.asarray(bpboolmatrix, float), np.asarray(bpscorematrix, float))],
                       fp32=False) as b:
        out = b.run_algo([0], algo, levellimit=levellimit)[0]
    return [_stem_record(i, j, ln, sc, sc, '') for i, j, ln, sc, _ in out]


# ------------------------------------------------------------------ host-side scores of a given structure
_BPSC = {"GU": -0.5, "UG": -0.5, "AU": 1.5, "UA": 1.5, "GC": 4.0, "CG": 4.0}


def ScoreStruct(seq, stemset, reacts):
    """(total, structure, reactivity) scores of a stem set -- SQRNdbnseq.py:861-899.
    Host-side: used for the evaluation of the *reference* structure when printing."""
    thescore = 0
    paired = set()
    for stem in stemset:
        bpsum = 0
        for v, w in stem[0]:
            bpsum += _BPSC.get(seq[v] + seq[w], 0.0)
            paired.add(v)
            paired.add(w)
        if bpsum > 0:
            thescore += bpsum ** 1.7
    sepnum = sum(1 for c in seq if c in SEPS)
    reactscore = 1 - sum(reacts[i] if i in paired else 1 - reacts[i]
                         for i in range(len(seq)) if seq[i] not in SEPS) / (len(seq) - sepnum)
    return round(thescore * reactscore, 3), round(thescore, 3), round(reactscore, 3)


def ReferenceScores(seq, ref, reacts):
    """SQRNdbnseq.py:958-970."""
    if not reacts:
        reacts = [0.5 for _ in range(len(seq))]
    reacts = [reacts[i] for i in range(len(seq)) if seq[i] not in GAPS]
    seq, ref = UnAlign(seq, ref)
    return ScoreStruct(seq, PairsToStems(sorted(DBNToPairs(ref))), reacts)


# ------------------------------------------------------------------ SQRNdbnseq / RunSQRNdbnseq
def SQRNdbnseq(seq, reacts=None, restraints=None, dbn=None, paramsets=[], conslim=1, toplim=5,
               hardrest=False, rankbydiff=False, rankby=(0, 2, 1), interchainonly=False, threads=1,
               mp=True, stemmatrix=None, poollim=1000, entropy=False, algos=set(), levellimit=None,
               priority=set(), M=1.8, B=-0.6):
    """Predict alternative secondary structures of one sequence -- SQRNdbnseq.py:973-1286.
    Returns (consensus, [(dbn, (total, struct, react), [paramset ids]), ...], [6 metrics], [7 metrics]).
    `threads`/`mp` are accepted for compatibility; the work runs on the GPU."""
    assert set(rankby) == {0, 1, 2} and len(rankby) == 3, "Invalid ranking indices"
    eng = _engine.get_engine()
    rec = (seq, reacts, restraints, dbn, paramsets, stemmatrix)
    if entropy:
        return eng.entropy(rec, interchainonly=interchainonly)
    return eng.fold_records([rec], conslim=conslim, toplim=toplim, hardrest=hardrest,
                            rankbydiff=rankbydiff, rankby=rankby, interchainonly=interchainonly,
                            poollim=poollim, algos=algos, levellimit=levellimit, priority=priority, M=M, B=B)[0]


def resolve_priority(priority, paramsetnames, rfam=None):
    """Names -> indices (SQRNdbnseq.py:1303-1310)."""
    if rfam and priority == {'bppN', 'bppH1', 'bppH2'}:
        priority = None
    if priority:
        return {i for i in range(len(paramsetnames)) if paramsetnames[i] in priority}
    return set()


def RunSQRNdbnseq(name, sequence, reactivities, restraints, reference, paramsetnames, paramsets,
                  threads, rankbydiff, rankby, hardrest, interchainonly, toplim, outplim, conslim,
                  reactformat, evalonly, poollim=1000, mp=True, sink=sys.stdout, stemmatrix=None,
                  entropy=False, algos={'G', }, levellimit=None, priority=None, rfam=None, M=1.8,
                  B=-0.6, _prediction=None, _ref_scores=None):
    """Print one record's block in the reference's format -- SQRNdbnseq.py:1289-1408.
    `_prediction` lets a batched caller (Predict) pass the result it already has."""
    # (the block is assembled as one string and written once: ten thousand records x eight print() calls were a third
    # of Predict's time on short sequences)
    out = [name, '\n']
    tab = '\t'.join
    priority = resolve_priority(priority, paramsetnames, rfam)
    if entropy:
        ent = SQRNdbnseq(sequence, reactivities, restraints, reference, paramsets, conslim, toplim,
                         hardrest, rankbydiff, rankby, interchainonly, threads, mp, stemmatrix, poollim,
                         entropy=True, algos=algos, M=M, B=B)
        out += [tab([sequence, "entropy:", str(ent)]), '\n']
    else:
        out += [sequence, '\n']
    seps = lambda line: ''.join(line[i] if sequence[i] not in SEPS else sequence[i] for i in range(len(sequence)))
    if reactivities:
        out += [str(EncodedReactivities(sequence, reactivities, reactformat)), "\treactivities\n"]
    if restraints:
        out += [seps(restraints), '\t', "restraints" + ("(" + rfam + ")" if rfam else ""), '\n']
    if reference:
        # (a batched caller passes the scores the C tail computed with the fold; the values are the same)
        if _ref_scores is not None:
            refsc = list(_ref_scores)
            if refsc[1] == 0:
                refsc[1] = 0                # ScoreStruct keeps the int 0 of a structure without a scoring stem (:871)
        else:
            refsc = ReferenceScores(sequence, reference, reactivities)
        out += [tab([seps(reference), "reference"] + [str(x) for x in refsc]), '\n']
    out += ['_' * len(sequence), '\n']
    if evalonly:
        sink.write(''.join(out))
        return None, None, None, None
    prediction = _prediction
    if prediction is None:
        prediction = SQRNdbnseq(sequence, reactivities, restraints, reference, paramsets, conslim, toplim,
                                hardrest, rankbydiff, rankby, interchainonly, threads, mp, stemmatrix,
                                poollim, algos=algos, levellimit=levellimit, priority=priority, M=M, B=B)
    consensus, predicted_structures, consensus_metrics, topN_metrics = prediction
    g4 = rfam and restraints and '+' in restraints          # SQRNdbnseq.py:1361-1363,1388-1390
    plus = lambda s: ''.join(ch if restraints[i] != '+' else '+' for i, ch in enumerate(s)) if g4 else s
    consensus = plus(consensus)
    if reference:
        out += [tab([consensus, "top-{}_consensus".format(conslim),
                     "TP={},FP={},FN={},FS={},PR={},RC={}".format(*consensus_metrics)]), '\n']
    else:
        out += [consensus, '\t', "top-{}_consensus".format(conslim), '\n']
    out += ['=' * len(sequence), '\n']
    for i, (struct, scores, psinds) in enumerate(predicted_structures[:outplim]):
        total, structscore, reactscore = scores
        if structscore == 0:
            structscore = 0                 # the reference keeps the int 0 of an empty structure (:871)
        fields = [plus(struct), "#{}".format(i + 1), str(total), str(structscore), str(reactscore),
                  ','.join(paramsetnames[p] for p in psinds)]
        if reference and i + 1 == topN_metrics[-1]:
            fields.append("TP={},FP={},FN={},FS={},PR={},RC={},RK={}".format(*topN_metrics))
        out += [tab(fields), '\n']
    sink.write(''.join(out))
    return consensus, predicted_structures, consensus_metrics, topN_metrics
