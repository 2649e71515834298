"""Pure helpers of the host layer (strings <-> pairs, reactivity pre-processing).

Host-side mirror of the L0 helpers of the reference (SQRNdbnseq.py:12-255,370-376);
they are O(N) string work that stays on the host by design.
"""
import math

GAPS = {'-', '.', '~'}          # SQRNdbnseq.py:12
SEPS = {';', '&'}               # SQRNdbnseq.py:14

#: reactivity alphabet (SQRNdbnseq.py:17-30): 3-level, 10-level and 26-level encodings
ReactDict = {"_": 0.00, "+": 0.50, "#": 1.00, "?": -999}
ReactDict.update({str(d): float("0.%d5" % d) for d in range(10)})
ReactDict.update({c: float("%.2f" % (0.04 * k)) for k, c in enumerate("abcdefghijklmnopqrstuvwxyz")})

#: bracket alphabet for pseudoknot levels (SQRNdbnseq.py:108-112)
BRACKETS = ['()', '[]', '{}', '<>'] + [c + c.lower() for c in "ABCDEFGHIJKLMNOPQRSTUVWXYZ"] + \
           [c + c.lower() for c in "БГДЁЖЙЛПФЦЧШЩЬЫЪЭЮЯ"]
_OPEN = {b[0]: b[1] for b in BRACKETS}
_CLOSE = {b[1]: b[0] for b in BRACKETS}


def ProcessReacts(reacts, missing_threshold=-10, middle=0.5, reverse=False, M=1.8, B=1.6):
    """Normalise raw reactivities to [0,1] around the neutral point (SQRNdbnseq.py:32-59)."""
    neutral = math.exp(-B / M) - 1
    if reverse:
        neutral, middle = middle, neutral
    if not reacts:
        return []
    out = []
    for x in reacts:
        if x <= missing_threshold or x != x:        # missing value or NaN
            x = neutral
        else:
            x = min(max(0, x), 1)
        if x <= neutral:
            out.append((middle / neutral) * x)
        else:
            out.append(middle + ((x - neutral) / (1 - neutral)) * (1 - middle))
    return out


def EncodedReactivities(seq, reacts, reactformat):
    """Floats -> one character per position (SQRNdbnseq.py:82-101)."""
    clipped = [x if 0 <= x <= 1 else 0 if x < 0 else 1 for x in reacts]
    if reactformat == 3:
        line = ["_+##"[int(x * 3)] for x in clipped]
    elif reactformat == 10:
        line = ['01234567899'[int(x * 10)] for x in clipped]
    else:
        line = ['abcdefghijklmnopqrstuvwxyz'[int(x * 25 + 0.5)] for x in clipped]
    return ''.join(seq[i] if seq[i] in SEPS else line[i] for i in range(len(seq)))


def DBNToPairs(dbn):
    """Dot-bracket string -> sorted list of pairs; unmatched closers are ignored
    (SQRNdbnseq.py:172-207)."""
    stacks, pairs = {}, set()
    for i, ch in enumerate(dbn):
        if ch in _OPEN:
            stacks.setdefault(ch, []).append(i)
        elif ch in _CLOSE:
            st = stacks.get(_CLOSE[ch])
            if st:
                pairs.add((st.pop(), i))
    return sorted(pairs)


_BRACKET_LUT = None


def bracket_bytes(text):
    """Dot-bracket text as one byte per character for the library's sq_dbn_pairs: ASCII as it is, the alphabet's Cyrillic
    letters (levels 31-49) recoded to 0x80 + k (opening) / 0xA0 + k (closing); any other character beyond ASCII is no bracket
    (a dot).  A table lookup over the code points (str.translate with a dict costs ~0.5 us per character beyond ASCII)."""
    import numpy as np
    global _BRACKET_LUT
    if text.isascii():
        return text.encode("ascii")
    if _BRACKET_LUT is None:
        lut = np.full(0x500, ord('.'), np.uint8)
        lut[:0x80] = np.arange(0x80, dtype=np.uint8)
        for k, b in enumerate(BRACKETS[30:]):
            lut[ord(b[0])], lut[ord(b[1])] = 0x80 + k, 0xA0 + k
        _BRACKET_LUT = lut
    cp = np.frombuffer(text.encode("utf-32-le"), np.uint32)
    return _BRACKET_LUT[np.minimum(cp, 0x4FF)].tobytes() if (cp < 0x500).all() else \
        np.where(cp < 0x500, _BRACKET_LUT[np.minimum(cp, 0x4FF)], ord('.')).astype(np.uint8).tobytes()


def levels_to_dbn(levels):
    """Signed per-position levels (+L open, -L close, 0 dot) -> dot-bracket string.
    Levels beyond the alphabet print as dots (SQRNdbnseq.py:142-143)."""
    nb = len(BRACKETS)
    out = []
    for v in levels:
        if v == 0:
            out.append('.')
        elif v > 0:
            out.append(BRACKETS[v - 1][0] if v <= nb else '.')
        else:
            out.append(BRACKETS[-v - 1][1] if -v <= nb else '.')
    return ''.join(out)


_GAP_LUT = None
_DROP_GAPS = {ord(g): None for g in GAPS}


def gap_mask(seq):
    """Boolean numpy array: True where seq has a gap character (vectorised; non-latin-1 characters are letters)."""
    import numpy as np
    global _GAP_LUT
    if _GAP_LUT is None:
        _GAP_LUT = np.zeros(256, bool)
        for g in GAPS:
            _GAP_LUT[ord(g)] = True
    return _GAP_LUT[np.frombuffer(seq.encode('latin-1', 'replace'), np.uint8)]


_ALN_PAIRS = (None, None, None)     # the line DBNToPairs was last asked for by UnAlign, and its pairs as two arrays


def _aligned_pairs(dbn):
    """DBNToPairs(dbn) as (v, w) arrays; the last line is remembered -- the rows of an alignment share one restraint line.
    (ONE tuple, replaced by one assignment: threads that prepare batches side by side each see a whole entry.)"""
    import numpy as np
    global _ALN_PAIRS
    line, v, w = _ALN_PAIRS
    if line != dbn:
        pairs = DBNToPairs(dbn)
        v = np.array([p[0] for p in pairs], np.int64)
        w = np.array([p[1] for p in pairs], np.int64)
        _ALN_PAIRS = (dbn, v, w)
    return v, w


def UnAlign(seq, dbn, want_pairs=False):
    """Drop gap columns; pairs touching a gap become dots first (SQRNdbnseq.py:236-255).  want_pairs: a third value, the
    pairs of the returned line (== DBNToPairs of it: taking matched pairs and unmatched brackets out of a line leaves the
    matching of the others as it was) or None when they were not formed."""
    import numpy as np
    if '-' not in seq and '.' not in seq and '~' not in seq:
        return (seq, dbn, None) if want_pairs else (seq, dbn)
    gaps = gap_mask(seq)
    keep = np.flatnonzero(~gaps)
    shortseq = seq.translate(_DROP_GAPS)
    if dbn.count('.') == len(dbn):                       # no brackets at all: nothing to clean
        return (shortseq, '.' * len(keep), []) if want_pairs else (shortseq, '.' * len(keep))
    try:
        arr, codec = np.frombuffer(dbn.encode('latin-1'), np.uint8).copy(), 'latin-1'
    except UnicodeEncodeError:
        # bracket letters beyond latin-1 (the alphabet's Cyrillic pairs, from 31 pseudoknot levels on): the same array code over
        # code points (until round 6 a per-character loop + two DBNToPairs per row: 1 s of config 5's step 1)
        arr, codec = np.frombuffer(dbn.encode('utf-32-le'), np.uint32).copy(), 'utf-32-le'
    v, w = _aligned_pairs(dbn)
    bad = gaps[v] | gaps[w]
    arr[v[bad]] = 46
    arr[w[bad]] = 46
    short = arr[keep].tobytes().decode(codec)
    if not want_pairs:
        return shortseq, short
    rank = np.cumsum(~gaps) - 1
    ok = ~bad
    return shortseq, short, list(zip(rank[v[ok]].tolist(), rank[w[ok]].tolist()))


def ReAlign(shortdbn, longseq, seqmode=False):
    """Re-insert the gap columns of longseq into shortdbn (SQRNdbnseq.py:210-233)."""
    assert len(shortdbn) + sum(longseq.count(g) for g in GAPS) == len(longseq), \
        "Cannot ReAlign dbn string - wrong number of gaps:\n{}\n{}".format(longseq, shortdbn)
    it = iter(shortdbn)
    return ''.join(('-' if seqmode else '.') if ch in GAPS else next(it) for ch in longseq)


def ParseRestraints(restraints, rbps=None):
    """Restraint line -> (bps, unpaired, no-left, no-right) (SQRNdbnseq.py:370-376).  rbps: DBNToPairs(restraints) when the
    caller has it already (UnAlign(..., want_pairs=True))."""
    if restraints.count('.') == len(restraints):         # the common case: no restraints
        return [], set(), set(), set()
    if rbps is None:
        rbps = DBNToPairs(restraints)
    rxs = {i for i, c in enumerate(restraints) if c in '_+'} if ('_' in restraints or '+' in restraints) else set()
    rlefts = {i for i, c in enumerate(restraints) if c == '/'} if '/' in restraints else set()
    rrights = {i for i, c in enumerate(restraints) if c == '\\'} if '\\' in restraints else set()
    return rbps, rxs, rlefts, rrights


def PairsToStems(sorted_pairs):
    """Group consecutive stacked pairs into [[bps], len] records (SQRNdbnseq.py:498-517)."""
    stems = []
    for k, (v, w) in enumerate(sorted_pairs):
        if k and sorted_pairs[k - 1][0] + 1 == v and sorted_pairs[k - 1][1] == w + 1:
            stems[-1][0].append((v, w))
            stems[-1][1] += 1
        else:
            stems.append([[(v, w)], 1])
    return stems


_CODE_LUT = None


def encode_seq(seq):
    """Letter codes of include/squarna_hip.h: 'A'..'Z' -> 0..25, ';' -> 26, '&' -> 27, other -> 28."""
    import numpy as np
    global _CODE_LUT
    if _CODE_LUT is None:
        _CODE_LUT = np.full(256, 28, np.uint8)
        _CODE_LUT[65:91] = np.arange(26, dtype=np.uint8)
        _CODE_LUT[ord(';')] = 26
        _CODE_LUT[ord('&')] = 27
    # characters outside latin-1 become '?' -> 28 ("other"), as in the per-character rule
    return _CODE_LUT[np.frombuffer(seq.encode('latin-1', 'replace'), np.uint8)].tobytes()


def PairsToDBN(newpairs, length=0, returnlevels=False, levellimit=-1):
    """Pairs -> dot-bracket string with pseudoknot levels (SQRNdbnseq.py:104-163): pairs sorted by
    (crossing count, i) are first-fitted into conflict-free groups, the largest group gets '()'.
    Host-side helper of the alignment layer (the fold path computes levels in C++).  The crossing
    relation is evaluated as one boolean matrix, so alignment-sized pair lists stay cheap."""
    import numpy as np
    pairs = sorted(set((min(v, w), max(v, w)) for v, w in newpairs))
    P = len(pairs)
    groups = []                                                    # lists of pair indices
    if P:
        a = np.array([p[0] for p in pairs], np.int64)
        b = np.array([p[1] for p in pairs], np.int64)
        # X[p, q]: p = (i, j), q = (k, l) cross  <=>  i < k < j < l  or  k < i < l < j   (:114-116)
        X = ((a[:, None] < a[None, :]) & (a[None, :] < b[:, None]) & (b[:, None] < b[None, :]))
        X |= X.T
        count = X.sum(axis=1)
        order = np.lexsort((a, count))                             # :125 sort by (crossings, i), stable
        cap = 8
        member = np.zeros((cap, P), bool)                          # member[g, q]: pair q sits in group g
        for p in order:                                            # :130-136 first fit
            p = int(p)
            ng = len(groups)
            if ng and count[p]:
                conflict = (member[:ng] & X[p]).any(axis=1)
                g = ng if conflict.all() else int(np.argmin(conflict))
            else:
                g = 0
            if g == ng:
                if ng == cap:
                    member = np.vstack([member, np.zeros((cap, P), bool)])
                    cap *= 2
                groups.append([])
            groups[g].append(p)
            member[g, p] = True
    groups.sort(key=len, reverse=True)                             # :139 (stable)
    if returnlevels:
        return {pairs[p]: lev + 1 for lev, group in enumerate(groups) for p in group}
    if levellimit >= 0:
        groups = groups[:levellimit]
    glyphs = BRACKETS + ['..'] * max(0, len(groups) - len(BRACKETS))
    dbn = ['.'] * length
    for k, group in enumerate(groups):
        for p in group:
            v, w = pairs[p]
            dbn[v], dbn[w] = glyphs[k][0], glyphs[k][1]
    return ''.join(dbn)
