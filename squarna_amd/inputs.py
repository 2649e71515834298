"""Input parsers of the host layer: SQUARNA's fasta-like default format, FASTA, Stockholm
and Clustal.  Host-side mirror of SQUARNA.py:80-390; every parser yields
(name, sequence, reactivities, restraints, reference) tuples.
"""
import sys

from .dbn import GAPS, ReactDict, ProcessReacts


def _decode_reactivities(text, length, M, B):
    """A reactivities line is either one symbol per base or whitespace-separated floats
    (SQUARNA.py:145-149)."""
    if len(text) != length:
        return ProcessReacts(list(map(float, text.split())), M=M, B=B)
    return ProcessReacts([ReactDict[ch] for ch in text], M=M, B=B)


def _check_record(name, sequence, reacts, restr, ref, M, B):
    """The checks and the reactivity decoding at the end of a record (SQUARNA.py:140-176)."""
    n = len(sequence)
    try:
        if reacts:
            reacts = _decode_reactivities(reacts, n, M, B)
        assert not reacts or len(reacts) == n
    except Exception:
        raise ValueError('Inappropriate reactivities line for entry "{}":\n {}'.format(name[1:], reacts))
    assert not restr or len(restr) == n, \
        'Inappropriate restraints line for entry "{}":\n {}'.format(name[1:], restr)
    assert not ref or len(ref) == n, \
        'Inappropriate reference line for entry "{}":\n {}'.format(name[1:], ref)
    return name, sequence, reacts, restr, ref


def ParseDefaultInput(inputname, inputformat, returndefaults=False, ignore=False, M=1.8, B=-0.6):
    """Default format: '>' name line, then one line per letter of `inputformat`
    (q seQuence, t reacTivities, r Restraints, f reFerence, x skipped).  Lines before
    the first '>' are defaults applied to every record of matching length
    (SQUARNA.py:80-203)."""
    q_ind = inputformat.index('q')
    t_ind, r_ind, f_ind = (inputformat.find(c) for c in 'trf')
    warned = {"reactivities": False, "restraints": False, "reference": False}
    defaults = {"t": None, "r": None, "f": None}

    def fallback(kind, current, default, n, fits):
        """Use the default line when the record has none and the lengths agree."""
        if current or not default:
            return current
        if fits(default, n):
            return default
        if not warned[kind]:
            warned[kind] = True
            msg = "WARNING: some sequences differ in length from the default {} line".format(kind)
            if not ignore:
                raise ValueError(msg + " [Switch on the iw/ignore parameter to proceed anyway]")
            print(msg, file=sys.stderr)
        return current

    nfields = len(inputformat)
    fits_reacts = lambda d, n: len(d) == n or len(d.split()) == n      # noqa: E731
    fits_line = lambda d, n: len(d) == n                                # noqa: E731

    def record(name, lines):
        if len(lines) < nfields:
            lines = lines + [None] * (nfields - len(lines))
        sequence = lines[q_ind].split()[0]            # trailing comments allowed (SQUARNA.py:98-104)
        reacts = lines[t_ind] if t_ind > 0 else None
        restr = lines[r_ind].split()[0] if r_ind > 0 and lines[r_ind] else None
        ref = lines[f_ind].split()[0] if f_ind > 0 and lines[f_ind] else None
        if defaults["t"] or defaults["r"] or defaults["f"]:     # (no default line: nothing to fall back to)
            n = len(sequence)
            reacts = fallback("reactivities", reacts, defaults["t"], n, fits_reacts)
            restr = fallback("restraints", restr, defaults["r"], n, fits_line)
            ref = fallback("reference", ref, defaults["f"], n, fits_line)
        return _check_record(name, sequence, reacts, restr, ref, M, B)

    name, lines = None, []
    with open(inputname) as fh:
        for raw in fh:
            if not raw.startswith('>'):
                lines.append(raw.strip())
                continue
            if name:
                yield record(name, lines)
            else:                                      # lines before the first entry = defaults
                d = lines + [None] * (len(inputformat) - 1 - len(lines))
                d.insert(q_ind, None)
                defaults["t"] = d[t_ind] if t_ind > 0 else None
                defaults["r"] = d[r_ind] if r_ind > 0 else None
                defaults["f"] = d[f_ind] if f_ind > 0 else None
                if returndefaults:
                    yield (defaults["t"], defaults["r"], defaults["f"])
                    return
            name, lines = raw.strip(), []
    if name:
        yield record(name, lines)


def ScanDefaultInput(inputname, inputformat):
    """The records of a default-format file as COLUMNS: (text, rows), rows[k] = offsets and lengths into text of record k's
    name line, sequence, reactivities line, restraints, reference (length -1: None) -- the library's record scan
    (sq_parse_default, host code; SQUARNA.py:80-203).  None when the file needs ParseDefaultInput's loop: default lines
    in front of the first record, bytes outside ASCII, carriage returns, a record without its sequence line.  Not on
    Predict()'s path yet: building ten thousand tuples costs Python what the scan saves; it is the first half of an
    ingestion that hands the batch its records as columns.  `rows_to_records` gives the tuples."""
    import ctypes as C
    import numpy as np
    from . import _lib
    L = _lib.load()
    q_ind = inputformat.index('q')
    t_ind, r_ind, f_ind = (inputformat.find(c) for c in 'trf')
    with open(inputname, "rb") as fh:
        data = fh.read()
    cap = data.count(b"\n>") + 1
    out = np.empty((cap, 10), np.int64)
    L.sq_parse_default.restype = C.c_int64
    n = L.sq_parse_default(C.c_char_p(data), C.c_int64(len(data)), C.c_int32(len(inputformat)), C.c_int32(q_ind), C.c_int32(t_ind),
                           C.c_int32(r_ind), C.c_int32(f_ind), C.c_void_p(out.ctypes.data), C.c_int64(cap))
    if n < 0:
        return None
    return data.decode("ascii"), out[:n]


def rows_to_records(text, rows, M=1.8, B=-0.6):
    """ParseDefaultInput's tuples from ScanDefaultInput's columns: the same values, the same checks in the same order."""
    for no, nl, so, sl, to, tl, ro, rl, fo, fl in rows.tolist():
        yield _check_record(text[no:no + nl], text[so:so + sl], text[to:to + tl] if tl >= 0 else None,
                            text[ro:ro + rl] if rl >= 0 else None, text[fo:fo + fl] if fl >= 0 else None, M, B)


_NOT_ACGUT = {ord(c): None for c in "ACGUTacgut"}


def GuessFormat(inp):
    """default / fasta / stockholm / clustal, plus "single entry?" (SQUARNA.py:206-236)."""
    with open(inp) as fh:
        first = fh.readline()
        if first.startswith('#') and "STOCKHOLM" in first:
            return "stockholm", 0
        if first.startswith("CLUSTAL"):
            return "clustal", 0
        entries = 1 if first.startswith(">") else 0
        seqlines = 0
        for line in fh:
            if line.startswith(">"):
                entries += 1
                continue
            # (letters of ACGUT in either case: one translate instead of an upper() and five count() per line)
            if len(line) - len(line.translate(_NOT_ACGUT)) > len(line) / 2:
                seqlines += 1
            if seqlines > 1000:
                break
        if seqlines > entries and entries > 0:
            return "fasta", (entries == 1)
    return "default", (entries == 1)


def ParseFasta(inp, returndefaults=False):
    """SQUARNA.py:239-256."""
    if returndefaults:
        yield (None, None, None)
        return
    name, seq = None, ''
    with open(inp) as fh:
        for line in fh:
            if line.startswith('>'):
                if name:
                    yield (name, seq, None, None, None)
                name, seq = line.strip(), ''
            elif line.strip():
                seq += line.strip()
    yield (name, seq, None, None, None)


def ReadStockholm(stkfile):
    """Headers, sequence names/dict, #=GC names/dict (SQUARNA.py:259-315)."""
    seqnames, seqdict, gcnames, gcdict, headers = [], {}, [], {}, []
    try:
        fh = open(stkfile)
        lines = fh.readlines()
    except UnicodeDecodeError:
        fh = open(stkfile, encoding="iso8859-15")
        lines = fh.readlines()
    fh.close()
    for line in lines:
        if line.startswith('#=GC '):
            parts = line.strip().split()
            key, val = ' '.join(parts[1:-1]), parts[-1]
            if key not in gcdict:
                gcnames.append(key)
                gcdict[key] = val
            else:
                gcdict[key] += val
        elif line.startswith('#'):
            headers.append(line)
        elif line.startswith('//') or not line.strip():
            continue
        else:
            parts = line.strip().split()
            key, val = ' '.join(parts[:-1]), parts[-1]
            if key not in seqdict:
                seqnames.append(key)
                seqdict[key] = val
            else:
                seqdict[key] += val
    headers = [h for h in headers if not h.startswith("#=GF SQ")] + \
              [h for h in headers if h.startswith("#=GF SQ")]
    return headers, seqnames, seqdict, gcnames, gcdict


def ParseStockholm(inp, returndefaults=False):
    """SS_cons is the default reference (SQUARNA.py:318-327)."""
    headers, seqnames, seqdict, gcnames, gcdict = ReadStockholm(inp)
    sscons = gcdict["SS_cons"] if "SS_cons" in gcnames else None
    if returndefaults:
        return None, None, sscons
    return [('>' + n, seqdict[n], None, None, sscons) for n in seqnames], len(seqnames) == 1


def ParseClustal(inp, returndefaults=False):
    """SQUARNA.py:330-347."""
    if returndefaults:
        return None, None, None
    seqs, names = {}, []
    with open(inp) as fh:
        for line in fh:
            if line.strip() and not line.startswith("CLUSTAL") and not line.startswith(' '):
                name, chunk = line.strip().split()
                if name not in seqs:
                    names.append(name)
                    seqs[name] = ''
                seqs[name] += chunk
    return [('>' + n, seqs[n], None, None, None) for n in names], len(names) == 1


def ParseSeq(inputseq, returndefaults, inputrestr):
    """SQUARNA.py:350-354."""
    if returndefaults:
        return None, None, None
    return [('>inputseq', inputseq, None, inputrestr, None)]


def ParseInput(inputseq, inputname, inputformat, returndefaults=False, fmt="unknown", ignore=False,
               inputrestr=None, M=1.8, B=-0.6):
    """Parser selector (SQUARNA.py:357-390)."""
    if inputseq:
        return ParseSeq(inputseq, returndefaults, inputrestr), fmt, True
    single = None
    if fmt == "unknown":
        fmt, single = GuessFormat(inputname)
        if fmt != "default":
            print("Non-default input file format is recognized: {}".format(fmt.upper()))
    if fmt == "default":
        if returndefaults:
            return next(ParseDefaultInput(inputname, inputformat, returndefaults, M=M, B=B)), fmt
        return ParseDefaultInput(inputname, inputformat, returndefaults, ignore=ignore, M=M, B=B), fmt, single
    if fmt == "fasta":
        if returndefaults:
            return next(ParseFasta(inputname, returndefaults)), fmt
        return ParseFasta(inputname, returndefaults), fmt, single
    if fmt == "stockholm":
        if returndefaults:
            return ParseStockholm(inputname, returndefaults), fmt
        parsed, single = ParseStockholm(inputname, returndefaults)
        return parsed, fmt, single
    if fmt == "clustal":
        if returndefaults:
            return ParseClustal(inputname, returndefaults), fmt
        parsed, single = ParseClustal(inputname, returndefaults)
        return parsed, fmt, single
