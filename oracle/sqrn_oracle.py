"""oracle/sqrn_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python front of the CPU oracle: thin ctypes bindings to ``liboracle.so``
(oracle/sqrn_oracle.c, the fp64 restatement of the reference's hot loops) plus
a restatement of the host-side tail of ``SQRNdbnseq`` (dedupe, ScoreStruct,
RankStructs, PairsToDBN, consensus, metrics).  Every function cites the
reference lines it follows (reference = febos/SQUARNA v3.2.2,
``src/SQUARNA/SQRNdbnseq.py`` =: dbnseq, ``SQRNalgos.py`` =: algos).

Edmonds / Hungarian call the same third-party libraries the reference calls
(networkx 3.4.2 ``max_weight_matching``, scipy 1.15.3
``linear_sum_assignment`` -- both are part of this image, here and on the GPU
box), exactly as algos:96-135 does.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Parity pinned by tests/test_oracle_golden.py.
bpp != 0 (ViennaRNA) is not restated: parity unpinned, rejected here.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GAPS = {'-', '.', '~'}          # dbnseq:12
SEPS = {';', '&'}               # dbnseq:14

# dbnseq:17-30 (reactivity alphabet -- data)
ReactDict = {"_": 0.00, "+": 0.50, "#": 1.00, "?": -999}
for _d in range(10):                      # "0" -> 0.05 ... "9" -> 0.95 (decimal literals)
    ReactDict[str(_d)] = float("0.%d5" % _d)
for _k, _c in enumerate("abcdefghijklmnopqrstuvwxyz"):   # "a" -> 0.00, step 0.04, "z" -> 1.00
    ReactDict[_c] = float("%.2f" % (0.04 * _k))

# dbnseq:108-112 (bracket alphabet -- data)
BRACKETS = ['()', '[]', '{}', '<>'] + [c + c.lower() for c in "ABCDEFGHIJKLMNOPQRSTUVWXYZ"] + \
           [c + c.lower() for c in "БГДЁЖЙЛПФЦЧШЩЬЫЪЭЮЯ"]


class Params(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("minlen", "minbpscore", "minfinscore", "bracketweight", "distcoef",
                 "orderpenalty", "loopbonus", "suboptmin", "suboptmax", "suboptsteps",
                 "maxstemnum")]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "sqrn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_greedy_calls.restype = C.c_long
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ---------------------------------------------------------------- helpers
def ProcessReacts(reacts, missing_threshold=-10, middle=0.5, reverse=False, M=1.8, B=1.6):
    """dbnseq:32-59."""
    neutral = np.exp(-B / M) - 1
    if reverse:
        neutral, middle = middle, neutral
    if not reacts:
        return []
    out = []
    for x in reacts:
        if x <= missing_threshold or np.isnan(x):
            x = neutral
        else:
            x = min(max(0, x), 1)
        if x <= neutral:
            out.append((middle / neutral) * x)
        else:
            out.append(middle + ((x - neutral) / (1 - neutral)) * (1 - middle))
    return out


_CLOSE = {b[1]: b[0] for b in BRACKETS}


def DBNToPairs(dbn):
    """dbnseq:172-207."""
    stacks, pairs = {}, set()
    for i, ch in enumerate(dbn):
        if ch in _CLOSE.values():
            stacks.setdefault(ch, []).append(i)
        elif ch in _CLOSE:
            st = stacks.get(_CLOSE[ch])
            if st:
                pairs.add((st.pop(), i))
    return sorted(pairs)


def pair_levels(pairs):
    """dbnseq:104-150: {bp: level} for normalised, sorted, unique pairs."""
    if not pairs:
        return {}, 0
    arr = np.array(pairs, dtype=np.int32).reshape(-1)
    lev = np.zeros(len(pairs), dtype=np.int32)
    ng = lib().orc_pair_levels(_p(arr, C.c_int), len(pairs), _p(lev, C.c_int))
    return {tuple(p): int(l) for p, l in zip(pairs, lev)}, ng


def PairsToDBN(newpairs, length=0, returnlevels=False, levellimit=-1):
    """dbnseq:104-163."""
    pairs = sorted(set((min(v, w), max(v, w)) for v, w in newpairs))
    levels, ng = pair_levels(pairs)
    if returnlevels:
        return levels
    dbn = ['.'] * length
    glyphs = BRACKETS + ['..'] * max(0, ng - len(BRACKETS))
    for (v, w), lv in levels.items():
        if levellimit >= 0 and lv > levellimit:
            continue
        dbn[v], dbn[w] = glyphs[lv - 1][0], glyphs[lv - 1][1]
    return ''.join(dbn)


def UnAlign(seq, dbn):
    """dbnseq:236-255."""
    clean = list(dbn)
    for v, w in DBNToPairs(dbn):
        if seq[v] in GAPS or seq[w] in GAPS:
            clean[v] = clean[w] = '.'
    keep = [i for i in range(len(seq)) if seq[i] not in GAPS]
    return ''.join(seq[i] for i in keep), ''.join(clean[i] for i in keep)


def ReAlign(shortdbn, longseq, seqmode=False):
    """dbnseq:210-233."""
    assert len(shortdbn) + sum(longseq.count(g) for g in GAPS) == len(longseq)
    it = iter(shortdbn)
    return ''.join(('-' if seqmode else '.') if ch in GAPS else next(it) for ch in longseq)


def ParseRestraints(restraints):
    """dbnseq:370-376."""
    rbps = DBNToPairs(restraints)
    rxs = {i for i, c in enumerate(restraints) if c in '_+'}
    rl = {i for i, c in enumerate(restraints) if c == '/'}
    rr = {i for i, c in enumerate(restraints) if c == '\\'}
    return rbps, rxs, rl, rr


def PairsToStems(sp):
    """dbnseq:498-517 -> list of (i, j, len)."""
    out = []
    for k, (v, w) in enumerate(sp):
        if k and sp[k - 1][0] + 1 == v and sp[k - 1][1] == w + 1:
            out[-1][2] += 1
        else:
            out.append([v, w, 1])
    return [tuple(s) for s in out]


def stem_bps(st):
    i, j, ln = st[0], st[1], st[2]
    return [(i + k, j - k) for k in range(ln)]


# ---------------------------------------------------------------- core bindings
#: source of base-pair probabilities for bpp != 0 paramsets: fn(seq, reacts, M, B) -> N x N array or None.
#: ViennaRNA is absent here, so this stays None unless a test installs a synthetic source: the APPLICATION of
#: the probabilities (dbnseq:350-364) is restated below, the probabilities themselves are parity-unpinned.
BPP_SOURCE = None


def ViennaBPP(seq, reacts, M=1.8, B=-0.6):
    """dbnseq:342-364: the calls the reference makes into ViennaRNA's Python module `RNA` (third party, absent from
    this image), restated as a BPP_SOURCE: fold_compound on the sequence with separators / non-ASCII letters
    replaced by N (:343-344), SHAPE pseudo-energies unless the reactivities are the default (:345-347), pf + bpp
    (:348-349), and ONE retry after exp_params_rescale(mfe) when every probability is zero (:355-359).
    Returns the N x N matrix (row/column 0 of ViennaRNA's 1-based table dropped) or None when it is still all zero.
    Pinned against the real reference running on tests/fake_rna.py (tests/golden/gen_bpp_golden.py)."""
    import RNA
    fc = RNA.fold_compound(''.join([ch if ch not in SEPS and ord(ch) <= 127 else 'N' for ch in seq]))
    if not (reacts is None or set(reacts) == {0.5}):                    # dbnseq:273
        fc.sc_add_SHAPE_deigan(ProcessReacts(list(reacts), reverse=True, M=M, B=B), m=M, b=B)
    fc.pf()
    bppm = np.array(fc.bpp())[1:, 1:]
    if np.max(bppm) > 0:
        return bppm
    (ss, mfe) = fc.mfe()
    fc.exp_params_rescale(mfe)
    fc.pf()
    bppm = np.array(fc.bpp())[1:, 1:]
    return bppm if np.max(bppm) > 0 else None


def BPMatrix(seq, weights, rxs, rlefts, rrights, interchainonly=False, reacts=None,
             bpp_power=0, M=1.8, B=-0.6):
    """dbnseq:258-367 (bpp_power != 0 needs BPP_SOURCE)."""
    if bpp_power and BPP_SOURCE is None:
        raise NotImplementedError("oracle: bpp != 0 needs ViennaRNA (parity unpinned)")
    n = len(seq)
    flags = np.zeros(max(n, 1), dtype=np.uint8)
    for i in rxs:
        flags[i] |= 1
    for i in rlefts:
        flags[i] |= 2
    for i in rrights:
        flags[i] |= 4
    keys = ''.join(weights.keys()).encode('latin-1')
    vals = np.array(list(weights.values()), dtype=np.float64)
    b = np.zeros((n, n)); s = np.zeros((n, n))
    rc = None if reacts is None else np.array(reacts, dtype=np.float64)
    lib().orc_bpmatrix(seq.encode('latin-1', 'replace'), n, keys, _p(vals, C.c_double), len(weights),
                       _p(flags, C.c_uint8), int(bool(interchainonly)),
                       None if rc is None else _p(rc, C.c_double),
                       _p(b, C.c_double), _p(s, C.c_double))
    if bpp_power:                                                      # dbnseq:350-364 (the rescale retry is the source's job)
        bppm = BPP_SOURCE(seq, reacts, M, B)
        if bppm is not None and np.max(bppm) > 0:
            if bpp_power < 0:
                s += (bppm / np.max(bppm)) ** (-bpp_power)
            else:
                s *= (bppm / np.max(bppm)) ** bpp_power
    return b, s


def _flat_pairs(rbps):
    a = np.array(sorted(rbps), dtype=np.int32).reshape(-1) if rbps else np.zeros(2, np.int32)
    return a, len(rbps)


def _flat_stems(rstems):
    a = np.array([list(s[:3]) for s in rstems], dtype=np.int32).reshape(-1) if rstems else np.zeros(3, np.int32)
    return a, len(rstems)


def AnnotateStems(boolmat, scoremat, rbps, rstems, minlen, minscore):
    """dbnseq:427-495 -> [(i, j, len, bpscore)] in emission order."""
    n = boolmat.shape[0]
    pr, npr = _flat_pairs(rbps)
    st, nst = _flat_stems(rstems)
    cap = max(16, n * n // 2 + 16)
    ijl = np.zeros(3 * cap, np.int32); sc = np.zeros(cap)
    cnt = lib().orc_annotate(_p(np.ascontiguousarray(boolmat), C.c_double),
                             _p(np.ascontiguousarray(scoremat), C.c_double), n,
                             _p(pr, C.c_int), npr, _p(st, C.c_int), nst,
                             C.c_double(minlen), C.c_double(minscore),
                             _p(ijl, C.c_int), _p(sc, C.c_double), cap)
    return [(int(ijl[3 * k]), int(ijl[3 * k + 1]), int(ijl[3 * k + 2]), float(sc[k])) for k in range(cnt)]


def _params(ps, minfinscore=None):
    p = Params()
    p.minlen = ps["minlen"]; p.minbpscore = ps["minbpscore"]
    p.minfinscore = ps["minbpscore"] * ps["minfinscorefactor"] if minfinscore is None else minfinscore
    p.bracketweight = ps["bracketweight"]; p.distcoef = ps["distcoef"]
    p.orderpenalty = ps["orderpenalty"]; p.loopbonus = ps["loopbonus"]
    p.suboptmin = ps["suboptmin"]; p.suboptmax = ps["suboptmax"]; p.suboptsteps = ps["suboptsteps"]
    p.maxstemnum = ps["maxstemnum"]
    return p


def OptimalStems(seq, rstems, boolmat, scoremat, reacts, rbps=(), subopt=1.0, minlen=2,
                 minbpscore=6, minfinscore=0, bracketweight=1.0, distcoef=0.1,
                 orderpenalty=0.0, loopbonus=0.0):
    """dbnseq:792-833 -> [(i, j, len, bpscore, finalscore)]."""
    n = len(seq)
    p = Params()
    p.minlen, p.minbpscore, p.minfinscore = minlen, minbpscore, minfinscore
    p.bracketweight, p.distcoef, p.orderpenalty, p.loopbonus = bracketweight, distcoef, orderpenalty, loopbonus
    pr, npr = _flat_pairs(rbps)
    st, nst = _flat_stems(rstems)
    cap = 4096
    ijl = np.zeros(3 * cap, np.int32); bps = np.zeros(cap); fin = np.zeros(cap)
    cnt = lib().orc_optimal(seq.encode('latin-1', 'replace'), n,
                            _p(np.ascontiguousarray(boolmat), C.c_double),
                            _p(np.ascontiguousarray(scoremat), C.c_double),
                            _p(pr, C.c_int), npr, _p(st, C.c_int), nst, C.c_double(subopt), C.byref(p),
                            _p(ijl, C.c_int), _p(bps, C.c_double), _p(fin, C.c_double), cap)
    assert cnt <= cap
    return [(int(ijl[3 * k]), int(ijl[3 * k + 1]), int(ijl[3 * k + 2]), float(bps[k]), float(fin[k]))
            for k in range(cnt)]


def greedy(seq, boolmat, scoremat, rbps, paramset, poollim):
    """Greedy pool loop, dbnseq:1102-1199 -> (finished structures, R)."""
    n = len(seq)
    p = _params(paramset)
    pr, npr = _flat_pairs(rbps)
    L = lib()
    nfin = L.orc_greedy(seq.encode('latin-1', 'replace'), n,
                        _p(np.ascontiguousarray(boolmat), C.c_double),
                        _p(np.ascontiguousarray(scoremat), C.c_double),
                        _p(pr, C.c_int), npr, C.byref(p), int(poollim))
    out = []
    for k in range(nfin):
        m = L.orc_greedy_nstems(k)
        ijl = np.zeros(3 * max(m, 1), np.int32); bps = np.zeros(max(m, 1)); fin = np.zeros(max(m, 1))
        L.orc_greedy_get(k, _p(ijl, C.c_int), _p(bps, C.c_double), _p(fin, C.c_double))
        out.append([(int(ijl[3 * t]), int(ijl[3 * t + 1]), int(ijl[3 * t + 2]), float(bps[t]), float(fin[t]))
                    for t in range(m)])
    return out, int(L.orc_greedy_calls())


# ---------------------------------------------------------------- E / H / N
def Edmonds(stems, power=1.7):
    """algos:96-110 (same third-party call as the reference)."""
    import networkx as nx
    edges = [(v, w, st[3] ** power) for st in stems for v, w in stem_bps(st)]
    G = nx.Graph()
    G.add_weighted_edges_from(edges)
    return sorted(nx.max_weight_matching(G))


def Hungarian(seq, stems, N, minloop=3, power=1.7):
    """algos:113-135 (same third-party call as the reference)."""
    from scipy.optimize import linear_sum_assignment
    mat = np.zeros((N, N))
    for st in stems:
        for v, w in stem_bps(st):
            mat[v, w] = mat[w, v] = -(st[3] ** power)
    ri, ci = linear_sum_assignment(mat)
    sol = {int(i): int(j) for i, j in zip(ri, ci)}
    return [(k, sol[k]) for k in sol
            if (k < sol[k] - minloop or k < sol[k] and any(ch in SEPS for ch in seq[k + 1:sol[k]]))
            and sol[k] in sol and sol[sol[k]] == k and mat[k, sol[k]] != 0]


def Nussinov(seq, stems, N):
    """algos:44-93 via the C restatement."""
    ijl = np.array([list(s[:3]) for s in stems], dtype=np.int32).reshape(-1) if stems else np.zeros(3, np.int32)
    sc = np.array([s[3] for s in stems], dtype=np.float64) if stems else np.zeros(1)
    cap = N + 4
    out = np.zeros(2 * cap, np.int32)
    cnt = lib().orc_nussinov(seq.encode('latin-1', 'replace'), N, _p(ijl, C.c_int), _p(sc, C.c_double),
                             len(stems), _p(out, C.c_int), cap)
    return sorted((int(out[2 * k]), int(out[2 * k + 1])) for k in range(cnt))


def RunAlgo(seq, boolmat, scoremat, restbps, minlen, minscore, algo="E", levellimit=3):
    """dbnseq:548-595 -> [(i, j, len, score, score)]."""
    stems = AnnotateStems(boolmat, scoremat, restbps, [], minlen, minscore)
    N = boolmat.shape[0]
    pairs = []
    if algo == "E":
        pairs = Edmonds(stems)
    elif algo == "N":
        pairs = Nussinov(seq, stems, N)
    elif algo == "H":
        pairs = Hungarian(seq, stems, N)

    def keep(stemlist, levels=None):
        out = []
        for st in stemlist:
            if levels is not None and levels[(st[0], st[1])] > 1 and st[2] < 4:
                continue
            score = 0
            for v, w in stem_bps(st):
                score = score + scoremat[v, w]
            if score >= minscore and st[2] >= minlen:
                out.append((st[0], st[1], st[2], float(score), float(score)))
        return out

    stemset = keep(PairsToStems(sorted((min(v, w), max(v, w)) for v, w in pairs)))
    pairs = [bp for st in stemset for bp in stem_bps(st)]
    pairs = DBNToPairs(PairsToDBN(pairs, N, levellimit=levellimit))
    levels = PairsToDBN(pairs, N, returnlevels=True)
    return keep(PairsToStems(sorted(pairs)), levels)


# ---------------------------------------------------------------- tail
_BPSC = {"GU": -0.5, "UG": -0.5, "AU": 1.5, "UA": 1.5, "GC": 4.0, "CG": 4.0}


def ScoreStruct(seq, stemset, reacts):
    """dbnseq:861-899."""
    thescore = 0
    paired = set()
    for st in stemset:
        bpsum = 0
        for v, w in stem_bps(st):
            bpsum += _BPSC.get(seq[v] + seq[w], 0.0)
            paired.add(v); paired.add(w)
        if bpsum > 0:
            thescore += bpsum ** 1.7
    sepnum = sum(1 for c in seq if c in SEPS)
    reactscore = 1 - sum(reacts[i] if i in paired else 1 - reacts[i]
                         for i in range(len(seq)) if seq[i] not in SEPS) / (len(seq) - sepnum)
    return round(thescore * reactscore, 3), round(thescore, 3), round(reactscore, 3)


def RankStructs(stemsets, rankbydiff=False, rankby=(0, 2, 1), priority=frozenset()):
    """dbnseq:902-955; entries are [stems, scores, psids]."""
    key = lambda x: [x[1][rb] for rb in rankby]
    fin = sorted(stemsets, key=key, reverse=True)
    fin = [s for s in fin if priority & set(s[2])] + [s for s in fin if not (priority & set(s[2]))]
    if not rankbydiff or len(fin) < 3:
        return fin
    bpsets = {id(s): {bp for st in s[0] for bp in stem_bps(st)} for s in fin}
    allbps = set().union(*bpsets.values())
    seen = set(bpsets[id(fin[0])])
    cur = 1
    while seen != allbps and cur < len(fin) - 1:
        fin = fin[:cur] + sorted(fin[cur:], key=lambda x: (len(bpsets[id(x)] - seen), key(x)), reverse=True)
        seen |= bpsets[id(fin[cur])]
        cur += 1
    return fin[:cur] + sorted(fin[cur:], key=key, reverse=True)


def SQRNdbnseq(seq, reacts=None, restraints=None, dbn=None, paramsets=(), conslim=1, toplim=5,
               hardrest=False, rankbydiff=False, rankby=(0, 2, 1), interchainonly=False,
               threads=1, mp=False, stemmatrix=None, poollim=1000, entropy=False, algos=frozenset(),
               levellimit=None, priority=frozenset(), M=1.8, B=-0.6, _stats=None):
    """dbnseq:973-1286; same return tuple as the reference."""
    assert set(rankby) == {0, 1, 2} and len(rankby) == 3, "Invalid ranking indices"
    seq = seq.upper().replace("T", "U")
    if not restraints:
        restraints = '.' * len(seq)
    assert len(seq) == len(restraints), "Invalid restraints given"
    if not reacts:
        reacts = [0.5] * len(seq)
    assert len(reacts) == len(seq), "Invalid reactivities given"
    if isinstance(reacts, str):
        reacts = ProcessReacts([ReactDict[ch] for ch in reacts])      # dbnseq:1020 (B = 1.6)
    shortseq, shortrest = UnAlign(seq, restraints)
    shortreacts = [reacts[i] for i in range(len(seq)) if seq[i] not in GAPS]
    if dbn:
        assert len(seq) == len(dbn)
        shortseq, shortdbn = UnAlign(seq, dbn)
    if stemmatrix is not None:
        gap = [i for i in range(len(seq)) if seq[i] in GAPS]
        shortsmat = np.delete(np.delete(stemmatrix, gap, 0), gap, 1)
    rbps, rxs, rl, rr = ParseRestraints(shortrest)
    N = len(shortseq)
    if levellimit is None:
        levellimit = 3 - int(N > 500)
    fixedalgos = set(algos)
    fins, seen = [], {}
    R = 0
    for psi, ps in enumerate(paramsets):
        cur_algos = fixedalgos if fixedalgos else ps['algorithms']
        boolmat, scoremat = BPMatrix(shortseq, ps["bpweights"], rxs, rl, rr, interchainonly,
                                     reacts=shortreacts, bpp_power=ps["bpp"], M=M, B=B)
        if stemmatrix is not None:
            scoremat = scoremat * shortsmat
        if entropy:
            return Entropy(boolmat, scoremat, rbps, ps["minlen"], ps["minbpscore"])
        finstemsets = []
        for algo in sorted(cur_algos):           # reference iterates a set (order unspecified)
            if algo == "G":
                continue
            finstemsets.append(RunAlgo(shortseq, boolmat, scoremat, rbps, ps["minlen"],
                                       ps["minbpscore"], algo=algo, levellimit=levellimit))
            R += 1
        if "G" in cur_algos:
            g, calls = greedy(shortseq, boolmat, scoremat, rbps, ps, poollim)
            finstemsets.extend(g)
            R += calls
        for fs in finstemsets:
            key = tuple(sorted(bp for st in fs for bp in stem_bps(st)))
            if key not in seen:
                fins.append([fs, ScoreStruct(shortseq, fs, shortreacts), psi])
                seen[key] = {psi}
            else:
                seen[key].add(psi)
    if _stats is not None:
        _stats["R"] = R; _stats["N"] = N
    for f in fins:
        key = tuple(sorted(bp for st in f[0] for bp in stem_bps(st)))
        f[2] = sorted(seen[key])
    fins = RankStructs(fins, rankbydiff, rankby, priority=set(priority))
    forced = {(v, w) for v, w in rbps
              if shortseq[v] + shortseq[w] in paramsets[-1]["bpweights"] or
              shortseq[w] + shortseq[v] in paramsets[-1]["bpweights"]} if hardrest else set()
    dbns = [PairsToDBN({bp for st in f[0] for bp in stem_bps(st)} | forced, N) for f in fins]
    consbps = set()
    top = fins[:conslim]
    if top:
        consbps = {bp for st in top[0][0] for bp in stem_bps(st)}
        for f in top[1:]:
            consbps &= {bp for st in f[0] for bp in stem_bps(st)}
    consbps |= forced
    dbns = [ReAlign(x, seq) for x in dbns]
    cons = ReAlign(PairsToDBN(consbps, N), seq)
    sepfix = lambda s: ''.join(s[i] if seq[i] not in SEPS else seq[i] for i in range(len(seq)))
    dbns = [sepfix(x) for x in dbns]
    cons = sepfix(cons)
    preds = [(dbns[k], fins[k][1], fins[k][2]) for k in range(len(dbns))]
    if not dbn:
        return cons, preds, [np.nan] * 6, [np.nan] * 7

    def prf(pred, known):
        tp, fp, fn = len(pred & known), len(pred - known), len(known - pred)
        prc = round(tp / (tp + fp), 3) if tp + fp else 1
        rcl = round(tp / (tp + fn), 3) if tp + fn else 1
        fsc = round(2 * tp / (2 * tp + fp + fn), 3) if 2 * tp + fp + fn else 1
        return tp, fp, fn, fsc, prc, rcl

    known = set(DBNToPairs(shortdbn))
    consresult = list(prf(consbps, known))
    best, result = -1, []
    for rank, f in enumerate(fins):
        m = prf({bp for st in f[0] for bp in stem_bps(st)} | forced, known)
        if m[3] > best:
            best, result = m[3], list(m) + [rank + 1]
        if rank + 1 >= toplim:
            break
    return cons, preds, consresult, result


def Entropy(boolmat, scoremat, restbps, minlen, minscore):
    """dbnseq:520-545."""
    stems = AnnotateStems(boolmat, scoremat, restbps, [], minlen, minscore)
    N = boolmat.shape[0]
    sm = np.zeros((N, N))
    for st in stems:
        for v, w in stem_bps(st):
            sm[v, w] = sm[w, v] = st[3]
    ent = 0
    for i in range(N):
        row = sm[i, :]
        if row.sum():
            probs = [p for p in row / row.sum() if p]
            ent += sum(-(probs * np.log2(probs)))
    return str(round(ent / N, 3))
