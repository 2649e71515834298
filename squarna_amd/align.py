"""Alignment mode: host-side mirror of ``SQRNdbnali.py`` on top of the HIP engine.

Step 1 folds nothing: for every sequence of the alignment the GPU produces the stems
(BPMatrix fill + anti-diagonal scan, the same kernels as the single-sequence path) and the host
adds their scores into the L x L column matrix through the gap map, exactly in the reference's
order (SQRNdbnali.py:233-237) -- the accumulation is fp64 and order-sensitive.  Step 2 folds every
sequence with the normalised column matrix as a weight (``bpscorematrix *= shortsmat``,
SQRNdbnseq.py:1084-1085) in ONE GPU batch.  Steps and text follow SQRNdbnali.py:332-458.
"""
import io
import sys

import numpy as np

from .dbn import GAPS, SEPS, DBNToPairs, PairsToDBN, EncodedReactivities
from .core import RunSQRNdbnseq
from . import engine as _engine


def ReAlignDict(shortseq, longseq):
    """Unaligned index -> aligned column (SQRNdbnali.py:20-37)."""
    out, k = {}, 0
    for col, ch in enumerate(longseq):
        if k >= len(shortseq):
            break
        if ch not in GAPS:
            out[k] = col
            k += 1
    return out


def MatrixToDBNs(mat, score, depth, verbose=False, sink=sys.stdout, cells=None):
    """Greedy assembly of conflict-free structures from the column matrix
    (SQRNdbnali.py:121-192): cells >= score*depth in decreasing order (ties: flat index order),
    span >= 4, first structure that has both columns free.  cells: optional pre-selected
    (flat indices ascending, values) of the upper cells with span >= 4 (the device path)."""
    N = mat.shape[0]
    thr = score * depth
    if cells is None:
        flat = mat.flatten()
        idx = np.flatnonzero(flat >= thr)
        vals = flat[idx]
    else:
        idx, vals = cells
    if not verbose:
        # only the first structure is used by the caller (:242), and a cell joins it iff both of its columns
        # are still free THERE -- whatever the later structures hold.  One sort (value descending, flat index ascending: what
        # the reference's stable sort over the index-ordered cells gives) and the sequential pass in the library
        import ctypes
        from . import _lib
        idx = np.ascontiguousarray(idx, np.int64)
        sidx = np.ascontiguousarray(idx[np.lexsort((idx, -np.asarray(vals)))])
        out = np.empty(2 * (N // 2 + 1), np.int32)
        n = int(_lib.load().sq_align_first_fit(ctypes.c_void_p(sidx.ctypes.data), ctypes.c_int64(len(sidx)), int(N), 4,
                                     ctypes.c_void_p(out.ctypes.data), ctypes.c_int64(len(out) // 2)))
        assert 0 <= n <= len(out) // 2
        return [PairsToDBN(out[:2 * n].reshape(-1, 2).tolist(), N)]
    order = np.argsort(-vals, kind='stable')                      # stable: equal values keep index order
    res = [[[], set()]]
    print(">Conserved base pairs (one by one)", file=sink)
    for k in order:
        bp = (int(idx[k] // N), int(idx[k] % N))
        if not bp[1] - bp[0] >= 4:
            continue
        for struct in res:
            if bp[0] not in struct[1] and bp[1] not in struct[1]:
                struct[0].append(bp)
                struct[1].add(bp[0])
                struct[1].add(bp[1])
                break
        else:
            res.append([[bp], set(bp)])
        print(PairsToDBN([bp], N), round(float(vals[k]), 3), sep='\t', file=sink)
    dbns = [PairsToDBN(struct[0], N) for struct in res]
    print(">Conserved base pairs (assembled)", file=sink)
    for dbn in dbns:
        print(dbn, file=sink)
    return dbns


def Metrics(ref, pred):
    """SQRNdbnali.py:195-208."""
    if not ref:
        return [np.nan] * 6
    rb, pb = set(DBNToPairs(ref)), set(DBNToPairs(pred))
    TP, FP, FN = len(pb & rb), len(pb - rb), len(rb - pb)
    PRC = (round(TP / (TP + FP), 3)) if (TP + FP) else 1
    RCL = (round(TP / (TP + FN), 3)) if (TP + FN) else 1
    FSC = (round(2 * TP / (2 * TP + FP + FN), 3)) if (2 * TP + FP + FN) else 1
    return [TP, FP, FN, FSC, PRC, RCL]


def SQRNdbnali(objs, defrests=None, defreacts=None, defref=None, bpweights={}, interchainonly=False,
               minlen=2, minbpscore=0, threads=1, verbose=False, sink=sys.stdout, M=1.8, B=-0.6):
    """Step-1 iteration: (first assembled dbn, L x L stem matrix) -- SQRNdbnali.py:211-242."""
    L = len(objs[0][1])
    recs = [(obj[1].upper().replace("T", "U"), obj[2], defrests if defrests else obj[3]) for obj in objs]
    eng = _engine.get_engine()
    if hasattr(eng, "stem_matrix"):
        # device path: the L x L matrix is accumulated and thresholded on the GPU; only the surviving cells come back
        stemmatrix = eng.stem_matrix(recs, bpweights, minlen, minbpscore, interchainonly)
        reduce_hook = getattr(eng, "reduce_matrix", None)
        if reduce_hook is not None:
            stemmatrix = reduce_hook(stemmatrix)                   # multi-GPU: all_reduce(sum) of the partial matrices
        cells = eng.matrix_cells(stemmatrix, minbpscore * len(objs), sort=verbose)   # (the non-verbose form sorts once, by value)
        pred = MatrixToDBNs(stemmatrix, minbpscore, len(objs), verbose, sink=sink, cells=cells)
        return pred[0], stemmatrix
    stemmatrix = np.zeros((L, L))
    allstems = eng.yield_stems(recs, bpweights, minlen, minbpscore, interchainonly)
    for (seq, _, _), (shortseq, stems) in zip(recs, allstems):     # reference order: sequences, stems, cells
        cols = np.array([c for c, ch in enumerate(seq) if ch not in GAPS], np.int64)     # ReAlignDict (:20-37)
        if isinstance(stems, np.ndarray):
            si, sj, sl, sc = stems["i"], stems["j"], stems["len"], stems["bpscore"]
        else:
            si, sj, sl, sc = (np.array([st[q] for st in stems]) for q in range(4))
        if not len(si):
            continue
        # cells of every stem, stems kept in emission order: np.add.at applies them sequentially, so
        # each column cell receives its fp64 additions in exactly the reference's order (:233-237)
        k = np.arange(int(sl.sum())) - np.repeat(np.cumsum(sl) - sl, sl)
        v = cols[np.repeat(si, sl) + k]
        w = cols[np.repeat(sj, sl) - k]
        val = np.repeat(sc, sl)
        np.add.at(stemmatrix, (v, w), val)
        np.add.at(stemmatrix, (w, v), val)
    reduce_hook = getattr(eng, "reduce_matrix", None)
    if reduce_hook is not None:
        stemmatrix = reduce_hook(stemmatrix)
    pred = MatrixToDBNs(stemmatrix, minbpscore, len(objs), verbose, sink=sink)
    return pred[0], stemmatrix


def _consensus_bulk(structs, freqlimit):
    """Consensus for many long lines (config 5: 512 x 5000): DBNToPairs of all lines in ONE library call (sq_dbn_pairs),
    the counts with numpy.  The reference's order is kept exactly: a stable sort by descending count over the dict's
    insertion order (SQRNdbnali.py:285) == count descending, first occurrence ascending.  Lines beyond ASCII (more than 30
    pseudoknot levels: Cyrillic bracket letters) go the same way, their letters recoded to single bytes (dbn.bracket_bytes)."""
    import ctypes
    from . import _lib
    N = len(structs[0])
    from .dbn import bracket_bytes
    text = "".join(structs)
    raw = bracket_bytes(text)                                        # (one byte per character: the Cyrillic bracket letters recoded)
    L = _lib.load()
    off = np.zeros(len(structs) + 1, np.int64)
    np.cumsum([len(x) for x in structs], out=off[1:])
    poff = np.zeros(len(structs) + 1, np.int64)
    rp = np.zeros(max(len(text), 2), np.int32)                      # (a line of n characters has at most n / 2 pairs)
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
    _lib.check(L.sq_dbn_pairs(raw, ptr(off), len(structs), ptr(rp), len(rp) // 2, ptr(poff)))
    pairs = rp[:2 * int(poff[-1])].reshape(-1, 2).astype(np.int64)
    width = max(max(len(x) for x in structs), 1)
    keys = pairs[:, 0] * width + pairs[:, 1]
    uniq, first, counts = np.unique(keys, return_index=True, return_counts=True)
    order = np.lexsort((first, -counts))
    lim = freqlimit * len(structs)
    seen = bytearray(width + 1)
    res = []
    for k, c in zip(uniq[order].tolist(), counts[order].tolist()):
        if c < lim:
            break                                                    # (counts only fall from here on)
        v, w = divmod(k, width)
        if not seen[v] and not seen[w]:
            seen[v] = seen[w] = 1
            res.append((v, w))
    return PairsToDBN(res, N)


def Consensus(structs, freqlimit=0.0, verbose=False, sink=sys.stdout):
    """Most frequent non-conflicting base pairs (SQRNdbnali.py:271-304)."""
    if not verbose and structs and len(structs) * len(structs[0]) >= 4096:
        fast = _consensus_bulk(structs, freqlimit)
        if fast is not None:
            return fast
    bps = {}
    freqlimit *= len(structs)
    for struct in structs:
        for bp in DBNToPairs(struct):
            bps[bp] = bps.get(bp, 0) + 1
    resbps, seen = [], set()
    if verbose:
        print(">Step 2, Populated base pairs", file=sink)
    for bp in sorted(bps.keys(), key=lambda x: bps[x], reverse=True):
        if verbose:
            print(PairsToDBN([bp], len(structs[0])), bps[bp], file=sink)
        if bps[bp] >= freqlimit and bp[0] not in seen and bp[1] not in seen:
            seen.add(bp[0])
            seen.add(bp[1])
            resbps.append(bp)
    return PairsToDBN(list(set(resbps)), len(structs[0]))


def ReactScore(reacts, seq, dbn):
    """SQRNdbnali.py:307-329."""
    if not reacts:
        return 0.5
    paired = {p for bp in DBNToPairs(dbn) for p in bp}
    sepnum = sum(1 for c in seq if c in SEPS)
    return 1 - sum(reacts[i] if i in paired else 1 - reacts[i]
                   for i in range(len(seq)) if seq[i] not in SEPS) / (len(seq) - sepnum)


def RunSQRNdbnali(objs, defreacts, defrests, defref, levellimit, freqlimit, verbose, step3, paramsetnames,
                  paramsets, threads, rankbydiff, rankby, hardrest, interchainonly, toplim, outplim, conslim,
                  reactformat, poollim, entropy=False, algos={'G', }, sink=sys.stdout, M=1.8, B=-0.6):
    """Alignment-based prediction and its text block -- SQRNdbnali.py:332-458."""
    N = len(objs[0][1])
    bpweights = paramsets[0]['bpweights']
    minlen = paramsets[0]['minlen']
    minbpscore = paramsets[0]['minbpscore']
    if verbose:
        print(">Step 1, Iteration 1", file=sink)
    pred_dbn, smat = SQRNdbnali(objs, defrests, defreacts, defref, bpweights, interchainonly, minlen, minbpscore,
                                threads, verbose, sink=sink, M=M, B=B)
    if verbose:
        print(">Step 1, Iteration 2", file=sink)
    pred_dbn = SQRNdbnali(objs, pred_dbn, defreacts, defref, bpweights, interchainonly, minlen, minbpscore,
                          threads, verbose, sink=sink, M=M, B=B)[0]
    step1dbn = PairsToDBN(DBNToPairs(pred_dbn), N, levellimit=levellimit)
    if step3 != '1':                                                 # (only step 2 reads it)
        if not isinstance(smat, np.ndarray) and not entropy:
            # device path: the normalised matrix stays on the GPU (the same two IEEE operations per cell as numpy's
            # smat / max * 5); step 2 gathers every sequence's rows and columns from it there (Batch(mul_shared=...))
            smat = (smat / smat.max() * 5).contiguous()              # :371
        else:
            if not isinstance(smat, np.ndarray):
                smat = smat.cpu().numpy()                            # (the entropy pass folds record by record on host matrices)
            smat = smat / np.max(smat) * 5                           # :371
    if verbose:
        print(">Step 1, Result", file=sink)
        print(step1dbn, file=sink)
    structs = []
    if step3 != '1':
        if verbose:
            print(">Step 2, Individuals", file=sink)
        # one GPU batch for all sequences (the reference uses Pool.imap over sequences, :382-390)
        recs = [(obj[1], obj[2], obj[3], obj[4], paramsets, smat) for obj in objs]
        eng = _engine.get_engine()
        if entropy:
            preds = [None] * len(objs)
        else:
            preds = eng.fold_records(recs, conslim=conslim, toplim=toplim, hardrest=hardrest, rankbydiff=rankbydiff,
                                     rankby=rankby, interchainonly=interchainonly, poollim=poollim, algos=algos,
                                     levellimit=None, priority=set(), M=M, B=B)
        for obj, pred in zip(objs, preds):
            name, seq, reacts, rests, ref = obj
            buf = io.StringIO()
            cons = RunSQRNdbnseq(name, seq, reacts, rests, ref, paramsetnames, paramsets, threads, rankbydiff,
                                 rankby, hardrest, interchainonly, toplim, outplim, conslim, reactformat, False,
                                 poollim, mp=False, sink=buf, stemmatrix=smat, entropy=entropy, algos=algos,
                                 M=M, B=B, _prediction=pred)[0]
            if verbose:
                print(buf.getvalue(), end='', file=sink)
            structs.append(cons)
        step2dbn = Consensus(structs, freqlimit, verbose, sink=sink)
        if verbose:
            print(">Step 2, Consensus", file=sink)
            for lim in range(0, 101, 5):
                print(Consensus(structs, lim / 100), str(lim) + '%', sep='\t', file=sink)
    else:
        step2dbn = '.' * N
    step2dbn = PairsToDBN(DBNToPairs(step2dbn), N, levellimit=levellimit)
    if verbose:
        print("=" * N, file=sink)
    first = objs[0][1]
    seps = lambda line: ''.join(line[i] if first[i] not in SEPS else first[i] for i in range(N))
    if defreacts:
        print(EncodedReactivities(first, defreacts, reactformat), "reactivities", sep='\t', file=sink)
    if defrests:
        print(seps(defrests), "restraints", sep='\t', file=sink)
    if defref:
        print(seps(defref), "reference", sep='\t', file=sink)
    if defreacts or defref or defrests:
        print("_" * N, file=sink)
    fmt = "TP={},FP={},FN={},FS={},PR={},RC={}"
    print(step1dbn, "Step-1" + ('\t' + str(round(ReactScore(defreacts, first, step1dbn), 2))) * bool(defreacts),
          fmt.format(*Metrics(defref, step1dbn)) * bool(defref), sep='\t', file=sink)
    print(step2dbn,
          "Step-2" + "(skipped)" * (step3 == '1') +
          ('\t' + str(round(ReactScore(defreacts, first, step2dbn), 2))) * bool(defreacts) * (step3 != '1'),
          fmt.format(*Metrics(defref, step2dbn)) * bool(defref) * (step3 != '1'), sep='\t', file=sink)
    if step3 == '1':
        step3dbn = step1dbn
    elif step3 == '2':
        step3dbn = step2dbn
    elif step3 == 'i':
        step3dbn = PairsToDBN(sorted(set(DBNToPairs(step1dbn)) & set(DBNToPairs(step2dbn))), N)
    else:
        step1pairs = DBNToPairs(step1dbn)
        seen_pos = set(pos for bp in step1pairs for pos in bp)
        for v, w in DBNToPairs(step2dbn):
            if v not in seen_pos and w not in seen_pos:
                step1pairs.append((v, w))
        step3dbn = PairsToDBN(sorted(step1pairs), N)
    print(step3dbn, "Step-3({})".format(step3) +
          ('\t' + str(round(ReactScore(defreacts, first, step3dbn), 2))) * bool(defreacts),
          fmt.format(*Metrics(defref, step3dbn)) * bool(defref), sep='\t', file=sink)
