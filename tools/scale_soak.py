#!/usr/bin/env python3
"""Predict() on inputs the unit tests do not reach by size: many records, long records, mixed lengths, wide pools, with
restraints and reactivities, and alignments of hundreds of sequences or thousands of columns.  Each case runs with the device drivers and again with the host-driven loop
(SQ_NO_POOL / SQ_NO_CHAIN) and the printed texts must be equal.  usage: scale_soak.py [case ...]"""
import hashlib, io, os, random, sys, tempfile, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from squarna_amd import Predict


def rnd_seq(rng, n):
    return "".join(rng.choice("ACGU") for _ in range(n))


def records(rng, count, nmin, nmax, extras):
    out = []
    for k in range(count):
        n = rng.randint(nmin, nmax)
        s = rnd_seq(rng, n)
        lines = [">r%d" % k, s]
        if extras and k % 3 == 1:                                   # reactivities
            lines.append(" ".join("%.2f" % rng.random() for _ in range(n)))
        if extras and k % 3 == 2:                                   # restraints: a few unpaired positions (after an empty reactivities line)
            lines += ["", "".join("_" if rng.random() < 0.05 else "." for _ in range(n))]
        out.append("\n".join(lines))
    return "\n".join(out) + "\n"


CASES = {
    # name: (count, nmin, nmax, extras, config, poollim)
    "many_short_wide": (10000, 20, 120, False, "nobpp", 1000),
    "s300_wide": (2000, 300, 300, False, "nobpp", 1000),
    "s1000_wide": (96, 1000, 1000, False, "nobpp", 100),
    "mixed_extras": (600, 5, 900, True, "nobpp", 50),
    "long_chain": (48, 1500, 2500, True, "fastest", 1),
    "alt_mixed": (400, 10, 400, True, "alt", 100),
    "huge_count_chain": (60000, 30, 150, False, "fastest", 1),     # several Predict batches (BATCH_RECORDS)
    "huge_count_pools": (40000, 30, 120, True, "nobpp", 100),
    "very_long_chain": (6, 4500, 6000, False, "fastest", 1),       # > 1024 stems per structure: the level scratch in dynamic LDS
    "very_long_pool": (4, 4200, 5000, False, "greedynobpp", 8),
    "giant_chain": (2, 20000, 31000, False, "fastest", 1),         # near the 32,000-nt limit of the 16-bit positions
    "giant_pool": (1, 12000, 12000, False, "greedynobpp", 4),
}


def msa(rng, nseq, ncol):
    """mutated copies of a random ancestor with gaps (alignment mode)"""
    anc = [rng.choice("ACGU") for _ in range(ncol)]
    # a few planted helices so that the consensus is not empty
    for _ in range(ncol // 40):
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


ALIGN_CASES = {
    # name: (sequences, columns, step3)
    "ali_wide": (300, 400, "u"),
    "ali_long": (40, 1500, "i"),
    "ali_many": (1500, 120, "1"),
    "ali_5000": (48, 5000, "u"),          # BASELINE config 5's width through all three steps (not in the default list: ~1 min)
}


def run_align(name):
    nseq, ncol, step3 = ALIGN_CASES[name]
    rng = random.Random(zlib.crc32(name.encode()))
    with tempfile.NamedTemporaryFile("w", suffix=".afa", delete=False) as f:
        f.write(msa(rng, nseq, ncol))
        path = f.name
    shas = []
    for env in ({}, {"SQ_NO_POOL": "1", "SQ_NO_CHAIN": "1"}):
        os.environ.update(env)
        try:
            buf = io.StringIO()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Predict(inputfile=path, alignment=True, step3=step3, write_to=buf)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        finally:
            for k in env:
                del os.environ[k]
        shas.append(hashlib.sha256(buf.getvalue().encode()).hexdigest())
        print("%-16s %-12s %8.1f ms  %d chars  %s" % (name, "host loop" if env else "device", dt * 1e3, len(buf.getvalue()), shas[-1][:16]), flush=True)
    os.unlink(path)
    assert shas[0] == shas[1], name
    return True


def run(name):
    if name in ALIGN_CASES:
        return run_align(name)
    count, nmin, nmax, extras, config, poollim = CASES[name]
    rng = random.Random(zlib.crc32(name.encode()))
    text = records(rng, count, nmin, nmax, extras)
    with tempfile.NamedTemporaryFile("w", suffix=".fas", delete=False) as f:
        f.write(text)
        path = f.name
    shas = []
    for env in ({}, {"SQ_NO_POOL": "1", "SQ_NO_CHAIN": "1"}):
        os.environ.update(env)
        try:
            buf = io.StringIO()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Predict(inputfile=path, inputformat="qtr" if extras else "q", configfile=config, poollim=poollim, write_to=buf)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        finally:
            for k in env:
                del os.environ[k]
        shas.append(hashlib.sha256(buf.getvalue().encode()).hexdigest())
        print("%-16s %-12s %8.1f ms  %d chars  %s" % (name, "host loop" if env else "device", dt * 1e3, len(buf.getvalue()), shas[-1][:16]), flush=True)
    os.unlink(path)
    assert shas[0] == shas[1], name
    return True


if __name__ == "__main__":
    names = sys.argv[1:] or ([c for c in CASES if not c.startswith("giant")] + [a for a in ALIGN_CASES if a != "ali_5000"])
    CASES["warmup"] = (64, 50, 300, True, "nobpp", 50)
    run("warmup")
    for n in names:
        run(n)
    print("all equal")
