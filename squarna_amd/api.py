"""``Predict`` and ``Main``: the drop-in API/CLI surface of SQUARNA (SQUARNA.py:416-1257).

Same keyword arguments, synonyms, validation messages, config lookup and output text as the
reference.  Single-sequence predictions are batched across the input records and folded on
the GPU (the reference parallelises the same loop over CPU processes, SQUARNA.py:887-935);
blocks are printed in input order; alignment mode (``a``) runs through ``squarna_amd.align``.
``bpp != 0`` paramsets take their base-pair probabilities from ViennaRNA on the host (``import RNA``,
as the reference does).  Out of scope of this build: Rfam/G4/RBP restraint discovery.
"""
import io
import os
import sys

from .config import ParseConfig, DATA_DIR
from .dbn import GAPS
from .inputs import ParseInput
from .core import RunSQRNdbnseq, resolve_priority
from . import engine as _engine

#: records folded per GPU batch (bounded by the sum of N^2 as well)
BATCH_RECORDS = 16384                 # records folded in one GPU batch (bigger batches: fewer, fuller kernel launches)
BATCH_CELLS = 2 * 1024 * 1024 * 1024   # ... bounded by sum of N^2 x paramsets


def Predict(inputfile=None, fileformat="unknown", inputseq=None, configfile=None, inputformat="qtrf",
            maxstemnum=None, threads=os.cpu_count(), byseq=False, algorithms='', entropy=False,
            rankby="r", evalonly=False, hardrest=False, interchainonly=False, toplim=5, outplim=None,
            conslim=1, poollim=1000, reactformat=3, alignment=False, levellimit=None, freqlimit=0.35,
            verbose=False, step3="u", ignorewarn=False, HOME_DIR=None, write_to=None, priority=None,
            rfam=False, g4=False, M=1.8, B=-0.6, rbp=False,
            i=None, ff=None, c=None, config=None, s=None, seq=None, a=None, ali=None, algo=None,
            algorithm=None, rb=None, fl=None, freqlim=None, ll=None, levlim=None, tl=None, ol=None,
            cl=None, pl=None, pr=None, s3=None, msn=None, rf=None, eo=None, hr=None, ico=None, iw=None,
            ignore=None, t=None, bs=None, v=None, inputrestr=None, _select=None, _on_block=None,
            _lengths_only=False):
    """Print SQUARNA predictions for the given input (see SQUARNA.py:431-600 for the
    meaning of every parameter; short synonyms are accepted exactly as there)."""
    # synonyms, later ones win as in SQUARNA.py:602-664
    def pick(cur, *alts):
        for alt in alts:
            if alt is not None:
                cur = alt
        return cur
    inputfile = pick(inputfile, i); fileformat = pick(fileformat, ff)
    configfile = pick(configfile, config, c); inputseq = pick(inputseq, seq, s)
    alignment = pick(alignment, ali, a); algorithms = pick(algorithms, algorithm, algo)
    rankby = pick(rankby, rb); freqlimit = pick(freqlimit, freqlim, fl)
    levellimit = pick(levellimit, levlim, ll); toplim = pick(toplim, tl); outplim = pick(outplim, ol)
    conslim = pick(conslim, cl); poollim = pick(poollim, pl); priority = pick(priority, pr)
    step3 = pick(step3, s3); maxstemnum = pick(maxstemnum, msn); reactformat = pick(reactformat, rf)
    evalonly = pick(evalonly, eo); hardrest = pick(hardrest, hr); interchainonly = pick(interchainonly, ico)
    ignorewarn = pick(ignorewarn, ignore, iw); threads = pick(threads, t); byseq = pick(byseq, bs)
    verbose = pick(verbose, v)

    if HOME_DIR is None:
        HOME_DIR = DATA_DIR
    if write_to is None:
        write_to = sys.stdout
    if inputfile != None and not os.path.exists(inputfile) and os.path.exists(os.path.join(HOME_DIR, inputfile)):
        inputfile = os.path.join(HOME_DIR, inputfile)

    # ---- validation (SQUARNA.py:677-808), same messages
    assert os.path.exists(str(inputfile)) or inputseq, "Input file does not exist."
    assert fileformat in {'unknown', 'fasta', 'default', 'stockholm', 'clustal'}, \
        "Wrong fileformat, choose one of these: default,fasta,stockholm,clustal"
    if configfile is None:
        configfileset = False
        configfile = os.path.join(HOME_DIR, "def.conf")
        configfile500 = os.path.join(HOME_DIR, "500.conf")
        configfile1000 = os.path.join(HOME_DIR, "1000.conf")
        priority = set('bppN,bppH1,bppH2'.split(',')) if priority is None else {x for x in priority.split(',') if x}
    else:
        configfileset = True
        if not os.path.exists(configfile):
            if os.path.exists(os.path.join(HOME_DIR, configfile + ".conf")):
                configfile = os.path.join(HOME_DIR, configfile + ".conf")
            elif os.path.exists(os.path.join(HOME_DIR, configfile)):
                configfile = os.path.join(HOME_DIR, configfile)
        assert os.path.exists(configfile), "Config file does not exist."
        priority = set() if priority is None else {x for x in priority.split(',') if x}
    assert ''.join(sorted(inputformat.replace('x', ''))) in {"q", "fq", "qr", "qt", "qrt", "fqr", "fqt", "fqrt"}, \
        'Inappropriate inputformat value (subset of "fqrtx" with "q" being mandatory): {}'.format(inputformat)

    def as_int(value, what, check, label):
        try:
            value = int(float(value))
            assert check(value)
            return value
        except Exception:
            raise ValueError("Inappropriate {} value ({}): {}".format(what, label, value))

    maxstemnumset = maxstemnum is not None
    maxstemnum = as_int(maxstemnum, "maxstemnum", lambda x: x >= 0, "non-negative integer") if maxstemnumset else 10 ** 6
    try:
        threads = min(max(1, int(float(threads))), os.cpu_count())
    except Exception:
        raise ValueError("Inappropriate threads value (integer): {}".format(threads))
    try:
        M = float(M)
    except Exception:
        raise ValueError("Inappropriate M value (float): {}".format(M))
    try:
        B = float(B)
    except Exception:
        raise ValueError("Inappropriate B value (float): {}".format(B))
    try:
        algos = set(algorithms.upper())
        assert algos <= {'E', 'G', 'H', 'N'}
    except Exception:
        raise ValueError('Inappropriate algorithm value (should be subset of "eghn"): {}'.format(algorithms))
    assert rankby in {"r", "s", "rs", "dr", "ds", "drs"}, \
        'Inappropriate rankby value (r/s/rs/dr/ds/drs): {}'.format(rankby)
    outplimset = outplim is not None
    if outplimset:
        outplim = as_int(outplim, "outplim", lambda x: x > 0, "positive integer")
    toplim = as_int(toplim, "toplim", lambda x: x > 0, "positive integer")
    if not outplimset:
        outplim = toplim
    conslim = as_int(conslim, "conslim", lambda x: x > 0, "positive integer")
    poollim = as_int(poollim, "poollim", lambda x: x > 0, "positive integer")
    assert int(float(reactformat)) in {3, 10, 26}, "Inappropriate reactformat value (3/10/26): {}".format(reactformat)
    reactformat = int(float(reactformat))
    if levellimit is not None:
        try:
            levellimit = int(float(levellimit))
        except Exception:
            raise ValueError("Inappropriate levellimit value (integer): {}".format(levellimit))
    try:
        freqlimit = float(freqlimit)
        assert 0 <= freqlimit <= 1
    except Exception:
        raise ValueError("Inappropriate freqlimit value (float between 0.0 and 1.0): {}".format(freqlimit))
    try:
        step3 = step3.lower()
        assert step3 in {'u', 'i', '1', '2'}
    except Exception:
        raise ValueError("Inappropriate freqlimit value (float between 0.0 and 1.0): {}".format(step3))

    rankbydiff = "d" in rankby                                   # SQUARNA.py:810-820
    if "r" in rankby and "s" in rankby:
        rankby = (0, 2, 1)
    elif "r" in rankby:
        rankby = (2, 0, 1)
    elif "s" in rankby:
        rankby = (1, 2, 0)

    if alignment and not configfileset:                          # SQUARNA.py:822-824
        configfile = os.path.join(HOME_DIR, "ali.conf")
    if rfam or g4 or rbp:
        raise NotImplementedError("Rfam / G4 / RBP restraint discovery (SQRNrfam.py) is out of scope of this build")

    paramsetnames, paramsets = ParseConfig(configfile)
    if not configfileset:
        paramsetnames500, paramsets500 = ParseConfig(configfile500)
        paramsetnames1000, paramsets1000 = ParseConfig(configfile1000)
    if maxstemnumset:
        sets = [paramsets] + ([paramsets500, paramsets1000] if not configfileset else [])
        for group in sets:
            for ps in group:
                ps['maxstemnum'] = maxstemnum

    inputs, fmt, single_input = ParseInput(inputseq, inputfile, inputformat, fmt=fileformat,
                                           ignore=ignorewarn, inputrestr=inputrestr, M=M, B=B)
    if _lengths_only:                                            # PredictSharded: the cost model's record lengths
        return [len(rec[1]) for rec in inputs]
    if alignment:                                                # SQUARNA.py:938-991
        from .align import RunSQRNdbnali
        from .dbn import ReactDict, ProcessReacts
        defR, defS, defF = ParseInput(inputseq, inputfile, inputformat, returndefaults=True, fmt=fmt,
                                      ignore=ignorewarn, M=M, B=B)[0]
        objs = [obj for obj in inputs]
        N = len(objs[0][1])
        assert all(len(obj[1]) == N for obj in objs), 'The sequences are not aligned'
        try:
            if defR:
                if len(defR) != N:
                    defR = ProcessReacts(list(map(float, defR.split())), M=M, B=B)
                else:
                    defR = ProcessReacts([ReactDict[ch] for ch in defR], M=M, B=B)
            assert not defR or len(defR) == N
        except Exception:
            raise ValueError('Inappropriate default reactivities line:\n {}'.format(defR))
        assert not defS or len(defS) == N, 'Inappropriate default restraints line:\n {}'.format(defS)
        assert not defF or len(defF) == N, 'Inappropriate default reference line:\n {}'.format(defF)
        if levellimit is None:
            levellimit = 3 - int(N > 500)
        RunSQRNdbnali(objs, defR, defS, defF, levellimit, freqlimit, verbose, step3, paramsetnames, paramsets,
                      threads, rankbydiff, rankby, hardrest, interchainonly, toplim, outplim, conslim, reactformat,
                      poollim, entropy=entropy, algos=algos, sink=write_to, M=M, B=B)
        return

    def config_for(seq):                                        # autoconfig, SQUARNA.py:868-878
        if configfileset:
            return paramsetnames, paramsets
        if len(seq) >= 1000:
            return paramsetnames1000, paramsets1000
        if len(seq) >= 500:
            return paramsetnames500, paramsets500
        return paramsetnames, paramsets

    eng = _engine.get_engine()
    common = dict(conslim=conslim, toplim=toplim, hardrest=hardrest, rankbydiff=rankbydiff, rankby=rankby,
                  interchainonly=interchainonly, poollim=poollim, algos=algos, levellimit=levellimit, M=M, B=B,
                  keep=max(int(outplim), 1))                     # (only the printed structures are fetched)

    def flush(batch):
        """Fold a batch of records on the GPU, then print every block in input order."""
        preds = [None] * len(batch)
        refsc = [None] * len(batch)
        texts = [None] * len(batch)
        if not evalonly and not entropy:
            # records with different priority index sets cannot share one fold call
            groups = {}
            for k, rec in enumerate(batch):
                names = rec[5]
                groups.setdefault(tuple(sorted(resolve_priority(priority, names))), []).append(k)
            # The library writes the output blocks itself (sq_write_blocks) when the engine can: the records then carry their
            # block fields (name, encoded reactivity line, which list of paramset names they print)
            blocks = getattr(eng, "writes_blocks", False)
            psnames, psname_idx = [], {}
            for prio, idx in groups.items():
                if blocks:
                    from .dbn import EncodedReactivities
                    recs = []
                    for k in idx:
                        name, seq, reacts, restrs, ref, names = batch[k][:6]
                        key = id(names)
                        if key not in psname_idx:
                            psname_idx[key] = len(psnames)
                            psnames.append(list(names))
                        rline = str(EncodedReactivities(seq, reacts, reactformat)) if reacts else None
                        recs.append((seq, reacts, restrs, ref, batch[k][6], None, name, rline, psname_idx[key]))
                    res = eng.fold_records(recs, priority=set(prio), _blocks=dict(psnames=psnames, conslim=conslim, outplim=outplim),
                                           **common)
                    if len(groups) == 1 and not _on_block and hasattr(res, "blocks") and hasattr(res.blocks, "text"):
                        write_to.write(res.blocks.text)            # every block of the batch, in input order: one write
                        return
                    for k, (kind, val) in zip(idx, res):
                        if kind == "text":
                            texts[k] = val
                        else:
                            preds[k], refsc[k] = val
                    continue
                res = eng.fold_records([(batch[k][1], batch[k][2], batch[k][3], batch[k][4], batch[k][6], None)
                                        for k in idx], priority=set(prio), **common)
                got_ref = getattr(eng, "last_ref_scores", None)
                for q, (k, r) in enumerate(zip(idx, res)):
                    preds[k] = r
                    if got_ref is not None and len(got_ref) == len(idx):
                        refsc[k] = got_ref[q]
        for k, (name, seq, reacts, restrs, ref, names, psets, index) in enumerate(batch):
            if texts[k] is not None:                               # the library's block: byte for byte RunSQRNdbnseq's
                if _on_block:
                    _on_block(index, texts[k])
                else:
                    write_to.write(texts[k])
                continue
            sink = io.StringIO() if _on_block else write_to
            RunSQRNdbnseq(name, seq, reacts, restrs, ref, names, psets, threads, rankbydiff, rankby,
                          hardrest, interchainonly, toplim, outplim, conslim, reactformat, evalonly, poollim,
                          mp=False, sink=sink, entropy=entropy, algos=algos, levellimit=levellimit,
                          priority=priority, rfam=False, M=M, B=B, _prediction=preds[k], _ref_scores=refsc[k])
            if _on_block:
                _on_block(index, sink.getvalue())

    # (thousands of small result containers: the cyclic collector only costs time here, see HipEngine.fold_records)
    import gc
    gc_was = gc.isenabled()
    gc.disable()
    try:
        _predict_records(inputs, _select, config_for, flush)
    finally:
        if gc_was:
            gc.enable()


def _predict_records(inputs, _select, config_for, flush):
    """The record loop of Predict's single-sequence mode: batches of records folded on the GPU, blocks printed in
    input order (the reference's ordered Pool.imap, SQUARNA.py:887-935)."""
    batch, cells = [], 0
    for index, (name, seq, reacts, restrs, ref) in enumerate(inputs):
        if _select is not None and index not in _select:
            continue
        names, psets = config_for(seq)
        batch.append((name, seq, reacts, restrs, ref, names, psets, index))
        cells += len(seq) * len(seq) * len(psets)
        if len(batch) >= BATCH_RECORDS or cells >= BATCH_CELLS:
            flush(batch)
            batch, cells = [], 0
    if batch:
        flush(batch)


def Main():
    """Command line: ``key=value`` tokens, bare flags and ``-key value`` forms (SQUARNA.py:994-1257)."""
    def usage():
        print("\nUsage:\n\nSQUARNA i=inputfile [OPTIONS]\n\nSQUARNA s=ACGUGUCAC [OPTIONS]\n")
        print("For further details read the help message:\n\nSQUARNA --help\n")
        sys.exit(1)

    args = sys.argv[1:]
    if not args:
        usage()
    if any(h in args for h in ("--help", "-help", "help", "--h", "-h", "h", "--H", "-H", "H")):
        readme = os.path.join(DATA_DIR, "README.md")
        print(open(readme).read() if os.path.exists(readme) else Predict.__doc__)
        sys.exit(0)

    valued = {"algo", "algorithm", "algos", "algorithms", "b", "c", "config", "i", "input", "if", "inputformat",
              "rb", "rankby", "ff", "fileformat", "fl", "freqlim", "ll", "levlim", "tl", "toplim", "ol",
              "outplim", "cl", "conslim", "pl", "poollim", "pr", "priority", "s3", "step3", "m", "msn",
              "maxstemnum", "rf", "reactformat", "s", "seq", "sequence", "t", "threads"}
    flags = {"a", "ali", "alignment", "bs", "byseq", "ent", "entropy", "eo", "evalonly", "g4", "hr", "hardrest",
             "iw", "ignore", "ico", "interchainonly", "rbp", "rfam", "v", "verbose"}
    norm, k = [], 0
    while k < len(args):                                         # "-key value" -> "key=value"
        tok = args[k]
        bare = tok.lstrip('-').lower()
        if tok.startswith('-') and tok.count('-') <= 2 and bare in valued and len(tok) - len(tok.lstrip('-')) in (1, 2):
            norm.append(tok.lstrip('-') + '=' + args[k + 1])
            k += 1
        elif tok.startswith('-') and bare in flags and len(tok) - len(tok.lstrip('-')) in (1, 2):
            norm.append(tok.lstrip('-'))
        else:
            norm.append(tok)
        k += 1

    kw = dict(poollim=100)                                       # CLI default differs from the API (SQUARNA.py:1047)
    keymap = {"algo": "algorithms", "algos": "algorithms", "algorithm": "algorithms", "algorithms": "algorithms",
              "s": "inputseq", "seq": "inputseq", "sequence": "inputseq", "i": "inputfile", "input": "inputfile",
              "ff": "fileformat", "fileformat": "fileformat", "c": "configfile", "config": "configfile",
              "if": "inputformat", "inputformat": "inputformat", "msn": "maxstemnum", "maxstemnum": "maxstemnum",
              "t": "threads", "threads": "threads", "rb": "rankby", "rankby": "rankby", "tl": "toplim",
              "toplim": "toplim", "ol": "outplim", "outplim": "outplim", "cl": "conslim", "conslim": "conslim",
              "pl": "poollim", "poollim": "poollim", "pr": "priority", "priority": "priority",
              "rf": "reactformat", "reactformat": "reactformat", "ll": "levellimit", "levlim": "levellimit",
              "levellim": "levellimit", "levlimit": "levellimit", "levellimit": "levellimit", "fl": "freqlimit",
              "freqlim": "freqlimit", "freqlimit": "freqlimit", "frequencylim": "freqlimit",
              "frequencylimit": "freqlimit", "s3": "step3", "step3": "step3", "m": "M", "b": "B"}
    flagmap = {"bs": "byseq", "byseq": "byseq", "eo": "evalonly", "evalonly": "evalonly", "hr": "hardrest",
               "hardrest": "hardrest", "ico": "interchainonly", "interchainonly": "interchainonly",
               "a": "alignment", "ali": "alignment", "alignment": "alignment", "v": "verbose",
               "verbose": "verbose", "iw": "ignorewarn", "ignore": "ignorewarn", "ent": "entropy",
               "entropy": "entropy", "rbp": "rbp", "rfam": "rfam", "g4": "g4"}
    lowered = {"fileformat", "inputformat"}
    for arg in norm:
        key, eq, val = arg.partition('=')
        lk = key.lower()
        if eq and lk in keymap:
            dest = keymap[lk]
            if dest == "algorithms" and not val:
                continue
            if dest in lowered:
                val = val.lower()
            if dest == "rankby":
                val = ''.join(sorted(val.lower()))
            kw[dest] = val
        elif not eq and lk in flagmap:
            kw[flagmap[lk]] = True
        elif len(norm) == 1:                                     # a lone token: file or sequence
            if os.path.exists(arg):
                kw["inputfile"] = arg
            elif sum(arg.lower().count(x) for x in (GAPS | set("acgut"))) > len(arg) / 2:
                kw["inputseq"] = arg
            else:
                kw["inputfile"] = arg
        else:
            print("Unrecognized option: {}".format(arg))
    print(kw.get("inputfile"))                                   # SQUARNA.py:1248
    Predict(**kw)


if __name__ == "__main__":
    Main()
