#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference (febos/SQUARNA v3.2.2).

Runs only in the build container, where /root/reference exists.  Nothing from
the reference (source, bytecode) is written into the repo: the outputs are
*data* -- inputs and the values the reference returned for them:

  tests/golden/bpmatrix.json   BPMatrix          (SQRNdbnseq.py:258-367)
  tests/golden/annotate.json   AnnotateStems     (SQRNdbnseq.py:427-495)
  tests/golden/optimal.json    OptimalStems call trace of whole folds
                               (SQRNdbnseq.py:792-833, 1102-1199)
  tests/golden/fold.json       SQRNdbnseq return tuples (SQRNdbnseq.py:973-1286)
  tests/golden/algos.json      RunAlgo E/H/N stemsets (SQRNdbnseq.py:548-595)
  tests/golden/text/*.txt      Predict(..., write_to=buf) full outputs
  tests/golden/digests.json    sha256 of those texts

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import os, sys, io, json, hashlib, math, random

sys.dont_write_bytecode = True
REF = "/root/reference/src/SQUARNA"
sys.path.insert(0, REF)
import numpy as np
import SQRNdbnseq as R          # noqa: E402  (the reference)
import SQUARNA as RC            # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(os.path.dirname(os.path.dirname(HERE)), "squarna_amd", "data")


def conf(name):
    names, psets = RC.ParseConfig(os.path.join(REF, name + ".conf"))
    return names, psets


def jsonable(x):
    if isinstance(x, (set, frozenset)):
        return sorted(jsonable(v) for v in x)
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, dict):
        return {k: jsonable(v) for k, v in x.items()}
    if isinstance(x, (np.floating,)):
        x = float(x)
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, float) and math.isnan(x):
        return "nan"
    return x


def sparse(mat):
    out = []
    n = mat.shape[0]
    for i in range(n):
        for j in range(n):
            if mat[i, j] != 0:
                out.append([i, j, float(mat[i, j])])
    return out


def rnd_seq(rng, n, alphabet="ACGU"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def rnd_restraints(rng, n, p=0.12):
    s = []
    for _ in range(n):
        if rng.random() < p:
            s.append(rng.choice("_+/\\"))
        else:
            s.append(".")
    return "".join(s)


def add_bp_restraints(rng, seq, restr, k=2):
    restr = list(restr)
    n = len(seq)
    pairs = {"GC", "CG", "AU", "UA", "GU", "UG"}
    tries = 0
    placed = 0
    brs = ["()", "[]", "{}"]
    while placed < k and tries < 200:
        tries += 1
        i = rng.randrange(0, n - 5)
        j = rng.randrange(i + 4, n)
        if restr[i] != "." or restr[j] != "." or seq[i] + seq[j] not in pairs:
            continue
        b = brs[placed % 3]
        restr[i], restr[j] = b[0], b[1]
        placed += 1
    return "".join(restr)


def gen_bpmatrix(rng):
    W = [{"GC": 3.25, "AU": 1.25, "GU": -1.25}, {"GC": 2.0, "AU": 1.0, "GU": 1.0},
         {"GC": 3.75, "AU": 1.75, "GU": 0.5}]
    cases = []

    def run(seq, w, restr=None, ico=False, reacts=None):
        restr = restr or "." * len(seq)
        rbps, rxs, rl, rr = R.ParseRestraints(restr)
        b, s = R.BPMatrix(seq, w, rxs, rl, rr, ico, reacts)
        cases.append(dict(seq=seq, weights=w, restraints=restr, interchainonly=ico,
                          reacts=reacts, bool=[[i, j] for i, j, _ in sparse(b)],
                          score=sparse(s)))

    run("GGGAAAACCC", W[0])
    run("ACGUACGUACUCGACG", W[0])
    for n in (5, 9, 17, 33, 48):
        run(rnd_seq(rng, n), W[rng.randrange(3)])
    # separators at every offset (inc4 rule), non-ACGU letters
    base = "GGGGCCCCAAAUUUGGGCCC"
    for k in range(1, len(base) - 1, 3):
        s = base[:k] + ";" + base[k:]
        run(s, W[0])
        run(s, W[1], ico=True)
    run("GGGNNCCCXAAUUU&GGCCC", W[2])
    run("GG;;CC&&GGAACC", W[1])
    # restraint symbols
    for _ in range(6):
        s = rnd_seq(rng, 30)
        run(s, W[rng.randrange(3)], restr=rnd_restraints(rng, 30, 0.2))
    # reactivities: encoded and float, incl. missing, >1, <0 (through ProcessReacts)
    for _ in range(6):
        n = 28
        s = rnd_seq(rng, n)
        raw = [rng.choice([0.0, 0.5, 1.0, -999, 0.3, 1.7, -0.2, float("nan")]) for _ in range(n)]
        reacts = R.ProcessReacts(raw, M=1.8, B=-0.6)
        run(s, W[rng.randrange(3)], reacts=reacts, restr=rnd_restraints(rng, n, 0.1))
    enc = "_+#0123456789abcxyz?"
    for _ in range(4):
        n = 26
        s = rnd_seq(rng, n)
        line = "".join(rng.choice(enc) for _ in range(n))
        reacts = R.ProcessReacts([R.ReactDict[c] for c in line], M=1.8, B=-0.6)
        run(s, W[0], reacts=reacts)
    run("GGGGAAAACCCC", W[0], reacts=[0.5] * 12)      # default-reacts shortcut
    return cases


def stems_out(stems):
    return [[s[0][0][0], s[0][0][1], s[1], s[2]] + list(s[3:4]) for s in stems]


def gen_annotate(rng):
    cases = []
    P = [dict(w={"GC": 3.25, "AU": 1.25, "GU": -1.25}, minlen=2, minscore=4.5),
         dict(w={"GC": 2.0, "AU": 1.0, "GU": 1.0}, minlen=2, minscore=3),
         dict(w={"GC": 3.25, "AU": 1.25, "GU": -1.25}, minlen=4, minscore=7),
         dict(w={"GC": 3.25, "AU": 1.25, "GU": -1.25}, minlen=2, minscore=0)]
    for t in range(40):
        n = rng.choice([10, 16, 23, 31, 40, 57, 64, 90])
        seq = rnd_seq(rng, n)
        if t % 5 == 1:
            k = rng.randrange(2, n - 2)
            seq = seq[:k] + ";" + seq[k + 1:]
        restr = "." * n
        if t % 3 == 1:
            restr = rnd_restraints(rng, n, 0.08)
        if t % 4 == 2:
            restr = add_bp_restraints(rng, seq, restr, 2)
        reacts = None
        if t % 6 == 3:
            reacts = R.ProcessReacts([rng.choice([0.0, 0.5, 1.0, -999, 0.25]) for _ in range(n)],
                                     M=1.8, B=-0.6)
        p = P[t % 4]
        rbps, rxs, rl, rr = R.ParseRestraints(restr)
        b, s = R.BPMatrix(seq, p["w"], rxs, rl, rr, False, reacts)
        rstems = []
        rounds = []
        for _ in range(3):
            restbps = set(rbps) - {bp for st in rstems for bp in st[0]}
            stems = R.AnnotateStems(b, s, restbps, rstems, p["minlen"], p["minscore"])
            rounds.append(dict(rstems=[[st[0][0][0], st[0][0][1], st[1]] for st in rstems],
                               stems=stems_out(stems)))
            if not stems:
                break
            pick = stems[rng.randrange(len(stems))]
            rstems = rstems + [pick]
        cases.append(dict(seq=seq, weights=p["w"], restraints=restr, reacts=reacts,
                          minlen=p["minlen"], minscore=p["minscore"], rounds=rounds))
    return cases


def trace_fold(seq, reacts, restraints, ref, names, psets, **kw):
    """Run SQRNdbnseq (mp=False) while recording every OptimalStems call."""
    calls = []
    orig = R.OptimalStems
    state = dict(mat=None, psi=-1)

    def spy(seq_, rstems, bm, sm, rc, rbps=set(), subopt=1.0, minlen=2, minbpscore=6,
            minfinscore=0, bracketweight=1.0, distcoef=0.1, orderpenalty=0.0, loopbonus=0.0):
        res = orig(seq_, rstems, bm, sm, rc, rbps, subopt, minlen, minbpscore, minfinscore,
                   bracketweight, distcoef, orderpenalty, loopbonus)
        if state["mat"] is not sm:
            state["mat"] = sm
            state["psi"] += 1
        calls.append(dict(g=state["psi"],
                          rstems=[[st[0][0][0], st[0][0][1], st[1]] for st in rstems],
                          subopt=subopt,
                          out=[[st[0][0][0], st[0][0][1], st[1], st[2], st[3]] for st in res]))
        return res

    R.OptimalStems = spy
    try:
        out = R.SQRNdbnseq(seq, reacts, restraints, ref, psets, mp=False, **kw)
    finally:
        R.OptimalStems = orig
    return out, calls


def read_default_records(path):
    return list(RC.ParseDefaultInput(path, "qtrf", M=1.8, B=-0.6))


def gen_fold(rng):
    folds, traces = [], []
    cfgs = {c: conf(c) for c in ("fastest", "alt", "greedynobpp", "nobpp", "ali")}

    def add(tag, seq, reacts, restr, ref, cfg, trace=False, **kw):
        names, psets = cfgs[cfg]
        if trace:
            out, calls = trace_fold(seq, reacts, restr, ref, names, psets, **kw)
            traces.append(dict(tag=tag, seq=seq, reacts=reacts, restraints=restr,
                               config=cfg, kw=jsonable(kw), calls=jsonable(calls)))
        else:
            out = R.SQRNdbnseq(seq, reacts, restr, ref, psets, mp=False, **kw)
        folds.append(dict(tag=tag, seq=seq, reacts=reacts, restraints=restr, reference=ref,
                          config=cfg, kw=jsonable(kw), out=jsonable(out)))

    add("appendixB", "GGGAAAACCC", None, None, None, "alt", trace=True)
    add("s16-nobpp", "ACGUACGUACUCGACG", None, None, None, "nobpp")
    add("s16-greedy", "ACGUACGUACUCGACG", None, None, None, "greedynobpp", trace=True)
    add("s16-fastest", "ACGUACGUACUCGACG", None, None, None, "fastest", trace=True)
    recs = read_default_records(os.path.join(REF, "examples", "seq_input.fas"))
    for k, (name, seq, reacts, restr, ref) in enumerate(recs):
        for cfg in ("greedynobpp", "alt", "fastest"):
            add("seq_input[%d]" % k, seq, reacts, restr, ref, cfg,
                trace=(cfg != "alt" or len(seq) < 60), poollim=100)
        add("seq_input[%d]-hr" % k, seq, reacts, restr, ref, "alt", hardrest=True,
            rankby=(2, 0, 1), conslim=2)
    recs = read_default_records(os.path.join(REF, "examples", "shape_input.fas"))
    for k, (name, seq, reacts, restr, ref) in enumerate(recs):
        for cfg in ("greedynobpp", "fastest", "ali"):
            add("shape_input[%d]" % k, seq, reacts, restr, ref, cfg, trace=True)
    # random sequences, a few with reactivities / restraints / separators
    for t in range(24):
        n = rng.choice([30, 30, 60, 100, 100, 150])
        seq = rnd_seq(rng, n)
        reacts = restr = None
        if t % 4 == 1:
            line = "".join(rng.choice("_+#") for _ in range(n))
            reacts = R.ProcessReacts([R.ReactDict[c] for c in line], M=1.8, B=-0.6)
        if t % 4 == 2:
            restr = add_bp_restraints(rng, seq, rnd_restraints(rng, n, 0.05), 2)
        if t % 6 == 3:
            k = rng.randrange(5, n - 5)
            seq = seq[:k] + "&" + seq[k + 1:]
        cfg = ("greedynobpp", "alt", "fastest")[t % 3]
        add("rand[%d]" % t, seq, reacts, restr, None, cfg, trace=(n <= 100), poollim=100,
            rankby=(2, 0, 1))
    for t in range(3):
        seq = rnd_seq(rng, 300)
        add("rand300[%d]" % t, seq, None, None, None, "fastest", trace=True, poollim=1)
    return folds, traces


def gen_algos(rng):
    """RunAlgo E/H/N on SRtest150 subset + examples; uniqueness flag for E."""
    import networkx as nx
    import SQRNalgos as A
    names, psets = conf("nobpp")
    ps = {n: p for n, p in zip(names, psets)}
    out = []
    recs = list(RC.ParseDefaultInput(os.path.join(REF, "datasets", "SRtest150.fas"), "qf"))
    picks = recs[::6] + read_default_records(os.path.join(REF, "examples", "seq_input.fas"))[:8]
    for name, seq, reacts, restr, ref in picks:
        seq_u = seq.upper().replace("T", "U")
        restr_u = restr if restr else "." * len(seq_u)
        shortseq, shortrest = R.UnAlign(seq_u, restr_u)
        if reacts:
            rc = [reacts[i] for i in range(len(seq_u)) if seq_u[i] not in R.GAPS]
        else:
            rc = [0.5] * len(shortseq)
        rbps, rxs, rl, rr = R.ParseRestraints(shortrest)
        for pname, algo in (("defE", "E"), ("defH", "H"), ("defN", "N")):
            p = ps[pname]
            b, s = R.BPMatrix(shortseq, p["bpweights"], rxs, rl, rr, False, reacts=rc)
            stems = R.AnnotateStems(b, s, rbps, [], p["minlen"], p["minbpscore"])
            if algo == "E":
                pairs = A.Edmonds(stems)
            elif algo == "H":
                pairs = A.Hungarian(shortseq, stems, len(shortseq), R.SEPS)
            else:
                pairs = A.Nussinov(shortseq, stems, len(shortseq), R.SEPS)
            ll = 3 - int(len(shortseq) > 500)
            stemset = R.RunAlgo(shortseq, b, s, rbps, [], p["minlen"], p["minbpscore"],
                                algo=algo, levellimit=ll)
            rec = dict(name=name, seq=shortseq, restraints=shortrest, reacts=rc, paramset=pname,
                       algo=algo, stems=stems_out(stems),
                       pairs=sorted([min(v, w), max(v, w)] for v, w in pairs),
                       stemset=[[st[0][0][0], st[0][0][1], st[1], st[2]] for st in stemset])
            if algo == "E":
                edges = {}
                for st in stems:
                    for v, w in st[0]:
                        edges[(v, w)] = st[2] ** 1.7
                tot = sum(edges[(min(v, w), max(v, w))] for v, w in pairs)
                uniq = True
                for v, w in pairs:
                    G = nx.Graph()
                    G.add_weighted_edges_from([(a, b_, wt) for (a, b_), wt in edges.items()
                                               if (a, b_) != (min(v, w), max(v, w))])
                    alt = nx.max_weight_matching(G)
                    t2 = sum(edges[(min(a, b_), max(a, b_))] for a, b_ in alt)
                    if abs(t2 - tot) <= 1e-9 * max(1.0, abs(tot)):
                        uniq = False
                        break
                rec["total_weight"] = tot
                rec["unique"] = uniq
            if algo == "H":
                rec["total_weight"] = sum((st[2] ** 1.7) for st in stems for v, w in st[0]
                                          if [v, w] in rec["pairs"])
            out.append(jsonable(rec))
    return out


def gen_text():
    os.makedirs(os.path.join(HERE, "text"), exist_ok=True)
    digests = {}
    jobs = [
        ("SRtest150_fastest", dict(inputfile=os.path.join(REF, "datasets/SRtest150.fas"), inputformat="qf", configfile="fastest")),
        ("SRtest150_fastest_pl1", dict(inputfile=os.path.join(REF, "datasets/SRtest150.fas"), inputformat="qf", configfile="fastest", poollim=1)),
        ("SRtest150_alt", dict(inputfile=os.path.join(REF, "datasets/SRtest150.fas"), inputformat="qf", configfile="alt")),
        ("SRtest150_greedynobpp", dict(inputfile=os.path.join(REF, "datasets/SRtest150.fas"), inputformat="qf", configfile="greedynobpp")),
        ("SRtest150_nobpp", dict(inputfile=os.path.join(REF, "datasets/SRtest150.fas"), inputformat="qf", configfile="nobpp")),
        ("s16_nobpp", dict(inputseq="ACGUACGUACUCGACG", configfile="nobpp")),
        ("s16_fastest", dict(inputseq="ACGUACGUACUCGACG", configfile="fastest")),
        ("seq_input_nobpp", dict(inputfile=os.path.join(REF, "examples/seq_input.fas"), configfile="nobpp")),
        ("seq_input_greedynobpp_rf10", dict(inputfile=os.path.join(REF, "examples/seq_input.fas"), configfile="greedynobpp", reactformat=10, rankby="rs", hardrest=True, toplim=3, outplim=7, conslim=2)),
        ("shape_input_fastest", dict(inputfile=os.path.join(REF, "examples/shape_input.fas"), configfile="fastest")),
        ("shape_input_alt_rf26", dict(inputfile=os.path.join(REF, "examples/shape_input.fas"), configfile="alt", reactformat=26, rankby="drs")),
        ("seq_input_evalonly", dict(inputfile=os.path.join(REF, "examples/seq_input.fas"), configfile="fastest", evalonly=True)),
        ("seq_input_entropy", dict(inputfile=os.path.join(REF, "examples/seq_input.fas"), configfile="alt", entropy=True)),
        ("seq_input_ico", dict(inputfile=os.path.join(REF, "examples/seq_input.fas"), configfile="alt", interchainonly=True, algorithms="g")),
        ("ali_input_a", dict(inputfile=os.path.join(REF, "examples/ali_input.afa"), alignment=True)),
        ("ali_input_a_verbose", dict(inputfile=os.path.join(REF, "examples/ali_input.afa"), alignment=True, verbose=True)),
        ("ali_input_a_s3i", dict(inputfile=os.path.join(REF, "examples/ali_input.afa"), alignment=True, step3="i", levellimit=1)),
        ("ali_input_a_s31", dict(inputfile=os.path.join(REF, "examples/ali_input.afa"), alignment=True, step3="1", freqlimit=0.5)),
        ("ali_input_a_entropy", dict(inputfile=os.path.join(REF, "examples/ali_input.afa"), alignment=True, verbose=True, entropy=True)),
        ("demo_afa_a", dict(inputfile=os.path.join(REF, "examples/demo.afa"), alignment=True, step3="2", reactformat=10)),
    ]
    only = os.environ.get("GOLDEN_ONLY")
    if only:
        jobs = [j for j in jobs if j[0].startswith(only)]
        with open(os.path.join(HERE, "digests.json")) as f:
            digests = json.load(f)
    for tag, kw in jobs:
        buf = io.StringIO()
        RC.Predict(write_to=buf, byseq=True, threads=8, **kw)
        txt = buf.getvalue()
        with open(os.path.join(HERE, "text", tag + ".txt"), "w") as f:
            f.write(txt)
        args = {k: (os.path.relpath(v, REF) if k == "inputfile" else v) for k, v in kw.items()}
        digests[tag] = dict(args=args, lines=txt.count("\n"),
                            sha256=hashlib.sha256(txt.encode()).hexdigest())
        print(tag, digests[tag]["lines"], digests[tag]["sha256"][:16], flush=True)
    return digests


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(name, os.path.getsize(os.path.join(HERE, name)), "bytes", flush=True)


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"bpmatrix", "annotate", "fold", "algos", "text"}
    if "bpmatrix" in which:
        dump("bpmatrix.json", gen_bpmatrix(random.Random(11)))
    if "annotate" in which:
        dump("annotate.json", gen_annotate(random.Random(12)))
    if "fold" in which:
        folds, traces = gen_fold(random.Random(13))
        dump("fold.json", folds)
        dump("optimal.json", traces)
    if "algos" in which:
        dump("algos.json", gen_algos(random.Random(14)))
    if "text" in which:
        dump("digests.json", gen_text())
