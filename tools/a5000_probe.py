#!/usr/bin/env python3
"""A5000 (BASELINE config 5): NSEQ x NCOL synthetic MSA = one random ancestor, per-site mutation
0.15, per-site gap 0.10, seed 5000; ali.conf; step 1 only (s3=1).  Times the two step-1 iterations on the GPU
and, with --cpu K, the CPU oracle's YieldStems on K sequences (extrapolated linearly)."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def make_msa(nseq, ncol, seed=5000, mut=0.15, gap=0.10):
    rng = np.random.default_rng(seed)
    anc = rng.choice(list("ACGU"), ncol)
    rows = []
    for _ in range(nseq):
        row = anc.copy()
        m = rng.random(ncol) < mut
        row[m] = rng.choice(list("ACGU"), int(m.sum()))
        row[rng.random(ncol) < gap] = "-"
        rows.append("".join(row))
    return rows


def main():
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    ncol = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    ncpu = int(sys.argv[sys.argv.index("--cpu") + 1]) if "--cpu" in sys.argv else 0
    rows = make_msa(nseq, ncol)
    from squarna_amd.config import ParseConfig, builtin_config
    names, psets = ParseConfig(builtin_config("ali"))
    ps = psets[0]
    recs = [(r, None, "." * ncol) for r in rows]
    if ncpu:
        from tests.oracle_engine import OracleEngine
        t0 = time.perf_counter()
        OracleEngine().yield_stems(recs[:ncpu], ps["bpweights"], ps["minlen"], ps["minbpscore"])
        dt = time.perf_counter() - t0
        print("cpu oracle (1 core): %d sequences %.2f s -> %.3f s/sequence, %.1f s per iteration of %d"
              % (ncpu, dt, dt / ncpu, dt / ncpu * nseq, nseq), flush=True)
    import torch
    from squarna_amd.engine import HipEngine
    from squarna_amd.align import MatrixToDBNs
    eng = HipEngine()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = eng.stem_matrix(recs, ps["bpweights"], ps["minlen"], ps["minbpscore"])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        cells = eng.matrix_cells(m, ps["minbpscore"] * nseq)
        t2 = time.perf_counter()
        dbn = MatrixToDBNs(m, ps["minbpscore"], nseq, cells=cells)[0]
        t3 = time.perf_counter()
        cellsN2 = sum((len(r) - r.count("-")) ** 2 for r in rows)
        print("step-1 iteration: stem matrix %.1f ms (%.0f seq/s, %.2f Gcell/s), select %d cells %.1f ms, assemble %.1f ms; pairs %d"
              % ((t1 - t0) * 1e3, nseq / (t1 - t0), cellsN2 / 2 / (t1 - t0) / 1e9, len(cells[0]), (t2 - t1) * 1e3,
                 (t3 - t2) * 1e3, sum(1 for c in dbn if c == "(")), flush=True)
    print("matrix checksum %.6f max %.3f" % (float(m.sum().item()), float(m.max().item())))


if __name__ == "__main__":
    main()
