"""oracle/sqrn_pyform.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The hot loops of the CPU oracle once more, in the REFERENCE'S ALGORITHMIC FORM: interpreted per-cell Python loops over
NumPy arrays with scalar indexing, one full re-scan of the matrix per AnnotateStems call, a per-position walk per
candidate in ScoreStems, a pure-Python O(N^3) Nussinov ("kept in the reference's algorithmic form ... so that
its speed is representative").  oracle/sqrn_oracle.c states the same algorithms in C and is ~11 x faster per core; this
module exists so that `cpu_baseline` can ALSO be quoted in the form the reference itself runs in
(bench.py: cpu_baseline.reference_form), and so that the ratio between the two forms is measured, not assumed.

`install()` swaps these functions into oracle.sqrn_oracle (BPMatrix, AnnotateStems, OptimalStems, greedy, Nussinov,
pair_levels); everything above them -- RunAlgo's filters, Edmonds / Hungarian through networkx / scipy, the ranking tail --
is sqrn_oracle's own Python in both forms.  Results are identical to the C form (tests/test_oracle_golden.py).
Reference = febos/SQUARNA v3.2.2, src/SQUARNA/SQRNdbnseq.py =: dbnseq, SQRNalgos.py =: algos.
"""
import numpy as np

SEPS = {';', '&'}               # dbnseq:14


def BPMatrix(seq, weights, rxs, rlefts, rrights, interchainonly=False, reacts=None, bpp_power=0, M=1.8, B=-0.6):
    """dbnseq:258-367 (bpp_power == 0): two double loops over the upper cells, set lookups and numpy scalar stores per cell."""
    if bpp_power:
        raise NotImplementedError("reference form: bpp != 0 needs ViennaRNA (parity unpinned)")
    n = len(seq)
    bps = {}
    for key, val in weights.items():                                   # :282-284 both orientations
        bps[key] = val
        bps[key[::-1]] = val
    chains = {}
    if interchainonly:                                                 # :264-271
        cur = 0
        for i, ch in enumerate(seq):
            if ch in SEPS:
                cur += 1
            else:
                chains[i] = cur
    boolmat = np.zeros((n, n))
    scoremat = np.zeros((n, n))
    defaultreacts = reacts is None or set(reacts) == {0.5}             # :273
    for i in range(n - 1):
        inc4 = 4                                                       # :294-297
        for chk in (1, 2):
            if i + chk < n and seq[i + chk] in SEPS:
                inc4 = chk + 1
        for j in range(i + inc4, n):                                   # :299-304
            ok = (seq[i] + seq[j]) in bps
            if interchainonly and ok:
                ok = chains.get(i, 0) != chains.get(j, 0)
            if ok:
                ok = i not in rxs and j not in rxs and j not in rlefts and i not in rrights
            boolmat[i, j] = 1.0 if ok else 0.0
    for i in range(n - 1):                                             # :318-338 the second pass over the same cells
        inc4 = 4
        for chk in (1, 2):
            if i + chk < n and seq[i + chk] in SEPS:
                inc4 = chk + 1
        for j in range(i + inc4, n):
            w = bps.get(seq[i] + seq[j], 0)                            # :308-311 unknown pairs weigh 0
            if defaultreacts:
                rf = 1.0
            else:
                rf = ((1 - (reacts[i] + reacts[j]) / 2) * 2) ** 0.5     # :333
            if w <= 0:
                rf = 1 / max(rf, 0.01)                                 # :335-336
            scoremat[i, j] = w * boolmat[i, j] * rf                    # :338
    return boolmat, scoremat


def _runs_of_diagonal(cells):
    """dbnseq:379-402: the maximal runs of active cells of one diagonal (slices of its list of cell records)."""
    runs, first = [], -1
    for k, rec in enumerate(cells):
        if rec[0] and first < 0:
            first = k
        if not rec[0] and first >= 0:
            runs.append(cells[first:k])
            first = -1
    if first >= 0:
        runs.append(cells[first:])
    return runs


def _stems_of_runs(runs):
    """dbnseq:405-418 with diff = 0 (the only value any caller passes): every run is one stem -- its pairs outer -> inner, its
    length, its cells' scores summed left to right from int 0."""
    stems = []
    for run in runs:
        bps = [rec[2] for rec in run]
        total = sum(rec[1] for rec in run)
        stems.append((bps[0][0], bps[0][1], len(bps), float(total)))
    return stems


def AnnotateStems(boolmat, scoremat, rbps, rstems, minlen, minscore):
    """dbnseq:427-495 (+ 379-424): copy the matrix, zero the rows and columns of every paired position and of the restraint
    pairs (keeping their own cell), then walk EVERY anti-diagonal cell by cell, outside in, cutting it into maximal runs;
    a run's score is summed left to right from int 0 (:416).  -> [(i, j, len, bpscore)] in emission order."""
    n = boolmat.shape[0]
    m = boolmat.copy()                                                 # :431
    for v, w in rbps:                                                  # :438-443
        m[v, :] = 0; m[:, v] = 0; m[w, :] = 0; m[:, w] = 0
        m[v, w] = boolmat[v, w]
    for st in rstems:                                                  # :446-451
        for b in range(st[2]):
            v, w = st[0] + b, st[1] - b
            m[v, :] = 0; m[:, v] = 0; m[w, :] = 0; m[:, w] = 0
    out = []
    starts = [(0, x) for x in range(4, n)] + [(y, n - 1) for y in range(1, n - 4)]    # :456-457
    for x, y in starts:
        # the diagonal as a list of cell records, outside in (:481-489): activity, score, position -- a record per cell
        cells = []
        i, j = x, y
        while i <= j - 1:
            cells.append([m[i, j], scoremat[i, j], (i, j)])
            i += 1
            j -= 1
        for st in _stems_of_runs(_runs_of_diagonal(cells)):            # (a call per step and diagonal, as the reference's helpers)
            if st[2] >= minlen and st[3] >= minscore:                  # :491-493
                out.append(st)
    return out


def _crosses(i, j, k, l):                                              # dbnseq:114-116
    return (i < k < j < l) or (k < i < l < j)


def pair_levels(pairs):
    """dbnseq:119-150: crossing counts, order by (count, i), first fit into groups, groups ranked by size (stable)."""
    n = len(pairs)
    if n == 0:
        return [], 0
    cc = [sum(1 for b in range(n) if a != b and _crosses(pairs[a][0], pairs[a][1], pairs[b][0], pairs[b][1])) for a in range(n)]
    order = sorted(range(n), key=lambda a: (cc[a], pairs[a][0]))        # :125 (stable; the input is sorted)
    groups = []
    for p in order:                                                    # :130-136
        for g in groups:
            if not any(_crosses(pairs[p][0], pairs[p][1], pairs[q][0], pairs[q][1]) for q in g):
                g.append(p)
                break
        else:
            groups.append([p])
    groups.sort(key=len, reverse=True)                                 # :139
    level = [0] * n
    for lv, g in enumerate(groups):
        for p in g:
            level[p] = lv + 1
    return level, len(groups)


_GOODLOOPS = {(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0), (2, 2), (1, 2), (2, 1), (3, 1), (1, 3),
              (2, 3), (3, 2), (3, 3), (3, 4), (4, 3), (4, 4), (4, 2), (2, 4)}      # dbnseq:615-622


def _score_stems(seq, stems, rstems, minfinscore, bracketweight, distcoef, orderpenalty, loopbonus):
    """dbnseq:607-751: per candidate a walk over every position between its innermost pair."""
    n = len(seq)
    partner = [-1] * n                                                 # :625
    pairs = []
    for st in rstems:                                                  # :631-635
        for b in range(st[2]):
            v, w = st[0] + b, st[1] - b
            partner[v], partner[w] = w, v
            pairs.append((v, w))
    pairs.sort()
    level, _ = pair_levels(pairs)                                      # :638
    levelof = {}
    for (v, w), lv in zip(pairs, level):
        levelof[v] = levelof[w] = lv
    res = []
    for i0, j0, ln, bpscore in stems:                                  # :641
        sa, sb = i0 + ln - 1, j0 - ln + 1                              # :655
        dots = brackets = 0
        levelset = set()
        nblock = 0
        be0 = be1 = 0
        inblockend = -1
        between = False
        for pos in range(sa + 1, sb):                                  # :665-689
            pr = partner[pos]
            if pr == -1:
                if pos > inblockend:
                    dots += 1
                if seq[pos] in SEPS:
                    between = True
            elif pr < sa or pr > sb:
                if pos > inblockend:
                    brackets += 1
                    levelset.add(levelof[pos])
            elif pos < pr and pr > inblockend:
                inblockend = pr
                if nblock == 0:
                    be0, be1 = pos, pr
                nblock += 1
        goodloop, diff1 = 0, 0                                         # :692-698
        if nblock == 1 and (be0 - sa - 1, sb - be1 - 1) in _GOODLOOPS:
            goodloop, diff1 = 1, abs((be0 - sa - 1) - (sb - be1 - 1))
        goodloopout, diff2 = 0, 0                                      # :700-711
        vv, ww = i0 - 1, j0 + 1
        while vv >= 0 and i0 - vv - 1 < 5 and partner[vv] == -1:
            vv -= 1
        while ww < n and ww - j0 - 1 < 5 and partner[ww] == -1:
            ww += 1
        if ww < n and partner[vv] == ww and partner[ww] == vv and (i0 - vv - 1, ww - j0 - 1) in _GOODLOOPS:   # (:708: vv == -1 reads the last position)
            goodloopout, diff2 = 1, abs((i0 - vv - 1) - (ww - j0 - 1))
        loopfactor = 1 + loopbonus * goodloop * (2 - diff1 / 2.0) + loopbonus * goodloopout * (2 - diff2 / 2.0)   # :715
        gnra = sb - sa - 1 == 4 and seq[sa + 1] == 'G' and seq[sa + 3] in 'GA' and seq[sa + 4] == 'A'           # :598-604
        tetrafactor = 1 + 0.25 * (1 if gnra else 0)
        idealdist = 4 if inblockend == -1 else 2                       # :721
        stemdist = dots + bracketweight * brackets                     # :723
        sdf = 1.0 if between else (1 / (1 + abs(stemdist - idealdist))) ** distcoef     # :726
        of = (1.0 / (1 + len(levelset))) ** orderpenalty               # :729
        fin = bpscore * sdf * of * loopfactor * 1 * tetrafactor        # :732
        if not goodloop and not goodloopout and ln < 3:                # :744-745
            fin = -1
        if fin >= minfinscore:                                         # :751
            res.append((i0, j0, ln, bpscore, float(fin)))
    return res


def _shares_base(a, b):
    sa = set(range(a[0], a[0] + a[2])) | set(range(a[1] - a[2] + 1, a[1] + 1))
    return any(p in sa for p in range(b[0], b[0] + b[2])) or any(p in sa for p in range(b[1] - b[2] + 1, b[1] + 1))


def _choose_stems(allstems, subopt):
    """dbnseq:754-789: stable sort by finalscore, descending; the first, then every stem within the range that shares a
    base with all stems taken so far."""
    if not allstems:
        return []
    ranked = sorted(allstems, key=lambda s: s[4], reverse=True)         # :758
    res = [ranked[0]]
    rng = subopt * ranked[0][4]                                        # :769
    for st in ranked[1:]:
        if st[4] < rng:                                                # :778
            break
        if all(_shares_base(st, r) for r in res):
            res.append(st)
    return res


def OptimalStems(seq, rstems, boolmat, scoremat, reacts, rbps=(), subopt=1.0, minlen=2, minbpscore=6, minfinscore=0,
                 bracketweight=1.0, distcoef=0.1, orderpenalty=0.0, loopbonus=0.0):
    """dbnseq:792-833 -> [(i, j, len, bpscore, finalscore)]."""
    have = {(st[0] + b, st[1] - b) for st in rstems for b in range(st[2])}
    rest = [bp for bp in sorted(rbps) if tuple(bp) not in have]        # :801
    cands = AnnotateStems(boolmat, scoremat, rest, rstems, minlen, minbpscore)
    scored = _score_stems(seq, cands, rstems, minfinscore, bracketweight, distcoef, orderpenalty, loopbonus)
    return _choose_stems(scored, subopt)


def greedy(seq, boolmat, scoremat, rbps, ps, poollim):
    """dbnseq:1102-1199 (mp=False) -> (finished structures in the order the reference appends them, R)."""
    cursubopt = ps["suboptmin"]                                        # :1069
    inc = (ps["suboptmax"] - ps["suboptmin"]) / ps["suboptsteps"]      # :1071
    minfin = ps["minbpscore"] * ps["minfinscorefactor"]
    cur, fin, cursize, calls = [[]], [], 1, 0
    while cur:
        if len(cur) > cursize:                                         # :1162-1165
            cursize = len(cur)
            if cursubopt < ps["suboptmax"]:
                cursubopt += inc
        nxt = []
        for stems in cur:
            if len(stems) == ps["maxstemnum"]:                         # :1168-1174
                fin.append(stems)
                continue
            new = OptimalStems(seq, stems, boolmat, scoremat, None, rbps, cursubopt, ps["minlen"], ps["minbpscore"], minfin,
                               ps["bracketweight"], ps["distcoef"], ps["orderpenalty"], ps["loopbonus"])      # :1182
            calls += 1
            if new:                                                    # :1190-1193
                stopper = 1 if cursize >= poollim else len(new)
                for st in new[:stopper]:
                    nxt.append(stems + [st])
            else:
                fin.append(stems)                                      # :1196
        cur = nxt
    return fin, calls


def Nussinov(seq, stems, N):
    """algos:44-93 + BackTrack :6-41 (matrix=None form): O(N^3) min-plus by diagonals, first best k wins, the pair is kept
    when it is at least as good as leaving j unpaired."""
    minloop = 3
    if N <= 0:
        return []
    scores = {}
    for st in stems:
        for b in range(st[2]):
            scores[(st[0] + b, st[1] - b)] = -st[3]
    D = np.zeros((N, N))
    K = {}
    for h in range(1, N):
        for i in range(N - h):
            j = i + h
            bestk, bestscorek = -1, 10 ** 9
            for k in range(i, j - 1):
                if (k, j) in scores:
                    scorek = D[i, k - 1] + D[k + 1, j - 1] + scores[(k, j)]      # (:74: k == i reads D[i, -1])
                    if scorek < bestscorek:
                        bestk, bestscorek = k, scorek
            if bestscorek <= D[i, j - 1]:
                K[(i, j)] = bestk
                D[i, j] = bestscorek
            else:
                D[i, j] = D[i, j - 1]
    pairs, queue = [], {(0, N - 1)}

    def sep(a, b):
        return any(0 <= x < N and seq[x] in SEPS for x in range(a, b))
    while queue:
        nq = set()
        for i, j in queue:
            if i < 0 or j < 0 or i >= N or j >= N:
                continue
            if (i, j) in K:
                k = K[(i, j)]
                if (k - 1) - i > minloop or ((k - 1) - i > 0 and sep(i + 1, k - 1)):
                    nq.add((i, k - 1))
                if (j - 1) - (k + 1) > minloop or ((j - 1) - (k + 1) > 0 and sep(k + 2, j - 1)):
                    nq.add((k + 1, j - 1))
                pairs.append((k, j))
            elif (j - 1) - i > minloop or ((j - 1) - i > 0 and sep(i + 1, j - 1)):
                nq.add((i, j - 1))
        queue = nq
    return sorted(pairs)


_SAVED = {}


def install(on=True):
    """Swap the per-cell Python forms into oracle.sqrn_oracle (on=False: the C forms back)."""
    from oracle import sqrn_oracle as O
    names = ("BPMatrix", "AnnotateStems", "OptimalStems", "greedy", "Nussinov", "pair_levels")
    if on and not _SAVED:
        for nm in names:
            _SAVED[nm] = getattr(O, nm)
        O.BPMatrix, O.AnnotateStems, O.OptimalStems, O.greedy, O.Nussinov = BPMatrix, AnnotateStems, OptimalStems, greedy, Nussinov

        def _levels(pairs):                                          # (sqrn_oracle's form: ({pair: level}, number of groups))
            lev, ng = pair_levels([tuple(p) for p in pairs])
            return {tuple(p): int(l) for p, l in zip(pairs, lev)}, ng
        O.pair_levels = _levels
    elif not on and _SAVED:
        for nm in names:
            setattr(O, nm, _SAVED[nm])
        _SAVED.clear()
