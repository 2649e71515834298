#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X folding core.

metric   : sequences/sec on SRtest150 (219 records, 8-150 nt, if=qf, c=nobpp: all five algorithms), whole job
           (bit-matrix fill + greedy stem loop + Edmonds / Hungarian / Nussinov + ranking tail), inputs resident in HBM.
step     : one fold of `--inflight` (default 8) independent batches per GPU, in flight at the same time (sq_fold_concurrent:
           one host thread and one set of streams per batch), each holding `--replicas` (default 12) copies of the 219-record
           set: 21,024 records per step.  One batch of 219 records alone is a single 5.8 ms latency chain (the blossom
           kernel of its largest graph, one wave) that leaves the chip > 99 % idle; batches in flight are the steady state
           of a server that streams input files, and the copies per batch amortise the fixed cost of the ~200 kernel
           launches of a fold (measured: 8 x 6 copies 429 k seq/s, 8 x 12 466 k, 8 x 24 463 k).  The latency of ONE
           219-record batch is reported beside it (`single_batch`), and `one_pass` / `stream` / `end_to_end` time the
           same records arriving from the host.  N > 1: weak scaling -- every rank folds its own batches, independent
           sequences, no data-path collective.
roofline : `roofline` = the dominant kernel of the S1000 leg (1,024 random-ACGU sequences, N = 1000, c=fastest,
           pl=1): sq_rounds_kernel, the persistent round kernel -- ONE launch per fold.  achieved = SURVEY 8d's algorithmic
           bytes of the launch (2 N^2 per AnnotateStems evaluation x the evaluations, asserted equal to sq_result_evals) /
           its launch time from HIP events on the library's stream (live); `traffic` = HBM bytes per launch from the rocprofv3
           PMC pass recorded in profiles/traffic.json (withheld when the kernels' sources have changed since).  `rooflines`
           = the other kernels that dominate a leg (blossom kernel of the headline step with its cycles per scan pass, the
           fp32 fill), each with the resource that binds it.
cpu_baseline: the CPU oracle (oracle/, a C port of the reference algorithm + the reference's own scipy / networkx
           calls) on the same workload, one process per host core; S1000 / S2000 on a stated subsample.  Its workers are
           spawned before the GPU is initialised and wait; the leg itself runs LAST (ten seconds of all-core load in front
           of the GPU legs made the latency-bound ones slower).
--workload S300|S1000|S2000: strong-scaling mode (SURVEY 8d workloads sharded over the ranks with lpt_partition, one
           RCCL all_gather of the packed results per step).  With --gpus N > 1 the default run adds a short sharded
           S300 leg as a secondary field.

Launch: python bench.py [--gpus N --steps K --warmup W]; for N > 1 via torch.distributed.run.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SIMD = 1024              # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4            # shader clock the blossom kernel runs at (measured 2.39-2.41 GHz with s_memtime, profiles/)


def load_srtest150():
    from squarna_amd.inputs import ParseDefaultInput
    path = os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtest150.fas")
    return list(ParseDefaultInput(path, "qf"))


def synthetic(workload):
    """S300 = 10,000 x 300 (seed 300), S1000 = 1,024 x 1000 (seed 1000), S2000 = 1,000 x 2000 (seed 2000)
    with a reactivity line drawn per position from "_+#" with p = (0.5, 0.3, 0.2); i.i.d. uniform ACGU."""
    import numpy as np
    n, count, seed, shape = {"S300": (300, 10000, 300, False), "S1000": (1000, 1024, 1000, False),
                             "S2000": (2000, 1000, 2000, True)}[workload]
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        seq = "".join(rng.choice(list("ACGU"), n))
        line = "".join(rng.choice(list("_+#"), n, p=[0.5, 0.3, 0.2])) if shape else None
        out.append((seq, line))
    return out


def pools_long_sequences(count=2000, n=500, seed=500):
    """The pools_long set: `count` i.i.d. uniform ACGU sequences of n nt (seed 500)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return ["".join(rng.choice(list("ACGU"), n)) for _ in range(count)]


def a5000_msa(nseq=512, ncol=5000, seed=5000):
    """BASELINE config 5's alignment: mutated copies (per-site substitution 0.12, per-site gap 0.06) of one random ancestor with
    ncol / 40 planted helices between its halves; the rows of the text are '>s<k>' + sequence.  (The generator of tools/
    scale_soak.py's ali cases and of every A5000 figure since round 2: sha256 a08a1b36a4af2978 of the three steps' text.)"""
    import random
    rng = random.Random(seed)
    anc = [rng.choice("ACGU") for _ in range(ncol)]
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


def prepare_synthetic(items):
    from squarna_amd.engine import Prepared
    from squarna_amd.dbn import ProcessReacts, ReactDict
    return [Prepared(s, ProcessReacts([ReactDict[c] for c in line], M=1.8, B=-0.6) if line else None) for s, line in items]


# ---------------------------------------------------------------- cpu_baseline (the oracle; checker, never the product)
def _oracle_init(cfg):
    global _O, _PSETS, _PSETS_BY
    sys.path.insert(0, ROOT)
    from oracle import sqrn_oracle as O
    from squarna_amd.config import ParseConfig, builtin_config
    _O = O
    _PSETS_BY = {c: ParseConfig(builtin_config(c))[1] for c in {cfg, "fastest", "500nobpp", "ali"}}
    _PSETS = _PSETS_BY[cfg]
    O.lib()


def _oracle_align_row(seq):
    """cpu_baseline worker task, BASELINE config 5: one row of the alignment through step 1's per-row work (SQRNdbnali.py:60-108,
    233-237) with the CPU oracle -- UnAlign, BPMatrix, AnnotateStems, the stems' cells added into an L x L matrix through the gap map."""
    import numpy as np
    O = _O
    ps = _PSETS_BY["ali"][0]
    t0 = time.perf_counter()
    seq = seq.upper().replace("T", "U")
    shortseq, shortrest = O.UnAlign(seq, "." * len(seq))
    rbps, rxs, rl, rr = O.ParseRestraints(shortrest)
    bm, sm = O.BPMatrix(shortseq, ps["bpweights"], rxs, rl, rr, False, None)
    stems = O.AnnotateStems(bm, sm, rbps, [], ps["minlen"], ps["minbpscore"])
    t1 = time.perf_counter()
    cols = np.array([c for c, ch in enumerate(seq) if ch not in O.GAPS], np.int64)
    mat = np.zeros((len(seq), len(seq)))
    for i, j, ln, sc in stems:                             # (the reference's own loop over the cells, :233-237)
        for t in range(ln):
            v, w = cols[i + t], cols[j - t]
            mat[v, w] += sc
            mat[w, v] += sc
    return time.perf_counter() - t0, time.perf_counter() - t1


def _oracle_one(rec):
    """cpu_baseline worker task: fold one record with the CPU oracle (a 7th field names another built-in config)."""
    name, seq, reacts, restr, ref, poollim = rec[:6]
    psets = _PSETS_BY[rec[6]] if len(rec) > 6 else _PSETS
    t0 = time.perf_counter()
    _O.SQRNdbnseq(seq, reacts, restr, ref, psets, poollim=poollim)
    return time.perf_counter() - t0


def _oracle_one_refform(rec):
    """cpu_baseline worker task: (seconds in the reference's algorithmic form -- oracle/sqrn_pyform.py: per-cell Python loops --,
    seconds in the C port) for one record."""
    from oracle import sqrn_pyform as P
    name, seq, reacts, restr, ref, poollim = rec[:6]
    t0 = time.perf_counter()
    _O.SQRNdbnseq(seq, reacts, restr, ref, _PSETS, poollim=poollim)
    t1 = time.perf_counter()
    P.install(True)
    try:
        _O.SQRNdbnseq(seq, reacts, restr, ref, _PSETS, poollim=poollim)
    finally:
        P.install(False)
    return time.perf_counter() - t1, t1 - t0


def effective_cpus():
    """CPUs this process may really use: hardware threads, affinity mask and the cgroup CPU quota (a container with
    cpu.max = "1600000 100000" shows 256 hardware threads and gets 16 CPUs worth of time)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return n


def cpu_workers_start(recs, cfg):
    """The cpu_baseline leg's worker processes: spawned BEFORE this process touches the GPU (a process that has initialised
    the GPU must not start others), warmed, and left idle until cpu_baseline() runs at the END of the bench -- ten seconds of
    all-core load in front of the GPU legs left the host slower for a while (one 219-record fold alone: 5.5 ms without the
    leg in front, 6.4 ms behind it).  The workers run their numerical libraries single-threaded: 256 processes x a BLAS /
    OpenMP pool each would measure oversubscription, not the algorithm."""
    import multiprocessing as mp
    cores = effective_cpus()
    saved = {}
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        saved[var] = os.environ.get(var)
        os.environ[var] = "1"                              # (inherited by the spawned workers; restored below)
    pool = mp.get_context("spawn").Pool(cores, initializer=_oracle_init, initargs=(cfg,))
    pool.map(_oracle_one, [r + (1000,) for r in recs[:cores]])   # warm the workers (imports, dlopen)
    for var, val in saved.items():
        if val is None:
            os.environ.pop(var, None)
        else:
            os.environ[var] = val
    return pool, cores


def cpu_baseline(workers, recs, cfg, target_s=10.0):
    """Times the oracle on the GPU box's host cores (one process per CPU the job may use -- cpu_workers_start --, records
    handed out dynamically, longest first) for about target_s seconds of wall time."""
    pool, cores = workers
    recs = [r + (1000,) for r in recs]
    _oracle_init(cfg)
    # one thread alone over the SAME record mix the pool folds (the whole set once; the pool repeats it longest first), so
    # that per_thread_seq_per_s / one_thread_alone_seq_per_s is what sharing the host costs a worker and nothing else
    per_pass = sum(_oracle_one(r) for r in recs)
    tasks = sorted(recs, key=lambda r: -len(r[1])) * 400   # more than any host finishes in target_s: cut off by the clock
    done, busy = 0, 0.0
    # a sliding window of 3 x cores tasks in flight (the same pool takes the S1000 / S2000 samples afterwards: nothing may
    # be left queued behind the clock)
    from collections import deque
    it, pend = iter(tasks), deque()
    for _ in range(3 * cores):
        pend.append(pool.apply_async(_oracle_one, (next(it),)))
    t0 = time.perf_counter()
    while True:
        dt_one = pend.popleft().get()
        done += 1
        busy += dt_one
        wall = time.perf_counter() - t0
        if wall >= target_s and done >= len(recs):
            break
        pend.append(pool.apply_async(_oracle_one, (next(it),)))
    for r in pend:                                         # (a few dozen short folds)
        r.get()
    out = dict(value=round(done / wall, 1), unit="seq/s", cores=cores, kind="port",
               per_thread_seq_per_s=round(done / max(busy, 1e-9), 2), one_thread_alone_seq_per_s=round(len(recs) / per_pass, 2),
               per_thread_note="both over whole passes of the record set (the pool's last pass may be cut by the clock: longest first, "
                               "so its figure is a lower bound by at most one pass in %d)" % max(1, done // len(recs)),
               sample="SRtest150 records (longest first, repeated) for %.1f s of wall time: %d folds, c=%s, C oracle "
                      "(oracle/sqrn_oracle.c + Python tail, scipy / networkx for H / E), one single-threaded process per CPU the "
                      "job may use (cores = min(hardware threads %d, affinity, cgroup quota)), summed worker time %.1fs" % (
                          wall, done, cfg, os.cpu_count() or 1, busy))
    # the CPU path "in the reference's algorithmic form" -- interpreted per-cell loops over NumPy arrays, a full re-scan
    # per AnnotateStems call (oracle/sqrn_pyform.py; within 15 % of the imported reference) -- on every fourth record
    sample = recs[::4]
    t0 = time.perf_counter()
    both = list(pool.imap_unordered(_oracle_one_refform, sample, chunksize=1))
    wall_rf = time.perf_counter() - t0
    s_ref, s_port = sum(b[0] for b in both), sum(b[1] for b in both)
    out["reference_form"] = dict(
        value=round(cores * len(sample) / s_ref, 1), unit="seq/s", cores=cores, kind="port in the reference's algorithmic form",
        per_thread_seq_per_s=round(len(sample) / s_ref, 2), port_per_thread_seq_per_s_same_records=round(len(sample) / s_port, 2),
        ratio_to_the_port=round(s_port / s_ref, 4),
        sample="every fourth SRtest150 record (%d records), c=%s, one per process (wall %.2fs): the oracle with its hot loops as interpreted "
               "per-cell Python (oracle/sqrn_pyform.py: BPMatrix, AnnotateStems, ScoreStems, ChooseStems, the pool loop, Nussinov, the level rule; "
               "Edmonds / Hungarian through networkx / scipy as in the reference) -- identical results; value = cores x records / summed seconds "
               "(linear extrapolation to all cores busy); the imported reference itself runs within 15 %% of this form (BASELINE.md: measured in "
               "the build container, the reference does not travel)" % (len(sample), cfg, wall_rf))
    # the synthetic BASELINE sizes on a stated subsample (time a subsample, extrapolate linearly)
    others = {}
    import numpy as np
    from squarna_amd.dbn import ProcessReacts, ReactDict
    for wl, take in (("S1000", 64), ("S2000", 32)):
        items = synthetic(wl)[:min(take, cores)]
        tasks = [("", s, ProcessReacts([ReactDict[c] for c in line], M=1.8, B=-0.6) if line else None, None, None, 1, "fastest")
                 for s, line in items]
        t0 = time.perf_counter()
        per = list(pool.imap_unordered(_oracle_one, tasks, chunksize=1))
        wall = time.perf_counter() - t0
        mean_s = float(np.mean(per))
        others[wl] = dict(value=round(cores / mean_s, 2), unit="seq/s", cores=cores, kind="port",
                          seconds_per_sequence_one_core=round(mean_s, 4),
                          sample="first %d sequences of %s, c=fastest pl=1, one per process (wall %.2fs); value = cores / "
                                 "mean seconds per sequence (linear extrapolation to all cores busy)" % (len(tasks), wl, wall))
    # pools_long (branching pools on 500-nt sequences, the reference's own default from 500 nt on): a few sequences
    tasks = [("", sq, None, None, None, 1000, "500nobpp") for sq in pools_long_sequences(min(8, cores))]
    t0 = time.perf_counter()
    per = list(pool.imap_unordered(_oracle_one, tasks, chunksize=1))
    wall = time.perf_counter() - t0
    others["pools_long"] = dict(value=round(cores / float(np.mean(per)), 2), unit="seq/s", cores=cores, kind="port",
                                seconds_per_sequence_one_core=round(float(np.mean(per)), 4),
                                sample="first %d sequences of the pools_long set, c=500nobpp poollim=1000, one per process (wall %.2fs); "
                                       "value = cores / mean seconds per sequence" % (len(tasks), wall))
    # BASELINE config 5 (A5000, SURVEY 8d: step 1 of the 512 x 5000 alignment): the per-row work of a few rows, one per process
    rows = [ln for ln in a5000_msa().split("\n") if ln and not ln.startswith(">")][:min(8, cores)]
    t0 = time.perf_counter()
    per = list(pool.imap_unordered(_oracle_align_row, rows, chunksize=1))
    wall = time.perf_counter() - t0
    row_s, acc_s = float(np.mean([p[0] for p in per])), float(np.mean([p[1] for p in per]))
    others["A5000"] = dict(value=round(2 * 512 * row_s / cores, 2), unit="s (step 1: both iterations of 512 rows)", cores=cores, kind="port",
                           higher_is_better=False, seconds_per_row_one_core=round(row_s, 3), of_which_python_accumulate_loop=round(acc_s, 3),
                           sample="first %d rows of the 512 x 5000 alignment (seed 5000), ali.conf, one per process (wall %.2fs): UnAlign + BPMatrix + "
                                  "AnnotateStems (C oracle) + the stems' cells added into the L x L matrix by the reference's own Python loop; value = "
                                  "2 iterations x 512 rows x mean seconds per row / cores (linear extrapolation to all cores busy; MatrixToDBNs' sort "
                                  "of L^2 cells in a Python dict, SQRNdbnali.py:131-133, is not in it)" % (len(rows), wall))
    out["other_workloads"] = others
    pool.terminate()
    return out


def mean_fs(results):
    fs_c = [r[2][3] for r in results]
    fs_b = [r[3][3] for r in results]
    return sum(fs_c) / len(fs_c), sum(fs_b) / len(fs_b)


# ---------------------------------------------------------------- PMC summaries (profiles/traffic.json)
def kernels_hash():
    h = hashlib.sha256()
    for f in ("sq_kernels.hip", "sq_cells.h", "sq_context.h", "sq_context.hip", "sq_match.hip", "sq_blossom.h", "sq_rounds.hip",
              "sq_rounds.h", "sq_scan.h", "sq_score.h", "sq_extend.h", "sq_pool_round.hip", "sq_pool_round.h", "sq_cellrun.h",
              "sq_device.h", "sq_gather.hip"):
        with open(os.path.join(ROOT, "squarna_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def host_threads(inflight, local_world):
    """Worker threads the library gives a batch (sq_pool in sq_host.hip): 4 x the rank's CPUs / batches in flight, 8..32,
    unless SQ_HOST_THREADS says otherwise."""
    if os.environ.get("SQ_HOST_THREADS"):
        return int(os.environ["SQ_HOST_THREADS"])
    cores = max(1, effective_cpus() // max(1, local_world))
    return min(max(4 * cores // max(1, inflight), 8), 32)


def load_pmc():
    """profiles/traffic.json: per kernel the counters of a rocprofv3 --pmc pass (tools/make_traffic.py), valid only for
    the kernel sources they were measured on."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return {}, "profiles/traffic.json missing"
    with open(path) as f:
        d = json.load(f)
    if d.get("kernels_sha16") != kernels_hash():
        return {}, "profiles/traffic.json was measured on other kernel sources (%s != %s): counters withheld" % (
            d.get("kernels_sha16"), kernels_hash())
    return d, None


# ---------------------------------------------------------------- S1000 roofline leg
def roofline_leg(nseq, n, pmc, pmc_note, seed=1000):
    """S1000: nseq random ACGU sequences of length n, c=fastest pl=1.  The leg's dominant kernel is the persistent round
    kernel (sq_rounds.hip: ONE launch runs every AnnotateStems / ScoreStems / ChooseStems round of every structure), so
    `roofline` is SURVEY 8d's figure for it: algorithmic bytes = 2 N^2 per AnnotateStems evaluation x the evaluations the
    launch performs, over the launch time measured live with HIP events on the library's stream.  The kernel keeps each
    structure's runs between rounds instead of re-reading the matrix, so the algorithmic rate exceeds the HBM peak (8d:
    "an implementation that avoids re-reading may exceed 100 %"); `traffic` is what the kernel really moves (PMC)."""
    import torch
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import Batch
    names, psets = ParseConfig(builtin_config("fastest"))
    items = synthetic("S1000")[:nseq] if n == 1000 else None
    prepared = prepare_synthetic(items)
    with Batch(prepared, [psets] * nseq, max_structs=nseq, fp32=False) as b:
        b.fold(poollim=1)                      # warm-up (also page-in)
        assert b.fold_paths & 4, "the S1000 leg must run the persistent round kernel"
        b.profile(True)
        reps = 5
        b.profile_reset()
        torch.cuda.synchronize()
        for _ in range(reps):
            b.fold(poollim=1)
        torch.cuda.synchronize()
        ms, launches, alg_bytes = b.profile_get(7)
        fms, flaunches, fbytes = b.profile_get(0)
        evals = sum(b.evals(k) for k in range(nseq))
        b.profile(False)
        walls = []
        for _ in range(5):                                  # whole-fold wall time, timers off
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.fold(poollim=1)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
    wall = min(walls)
    assert launches == reps, launches
    avg_ms = ms / launches
    per_launch = alg_bytes / launches
    # 2 N^2 bytes per AnnotateStems evaluation and nothing else -- booked per LIVE structure and round
    assert abs(per_launch - evals * 2.0 * n * n) <= 1e-9 * max(per_launch, 1.0), (per_launch, evals, n)
    alg_gbs = per_launch / (avg_ms * 1e-3) / 1e9
    k = pmc.get("sq_rounds_kernel") or {}
    traffic = (k.get("fetch_bytes_per_launch", 0) + k.get("write_bytes_per_launch", 0)) if k else None
    # The roof that binds the kernel is vector instruction issue (a wave64 VALU instruction holds its SIMD for 4 cycles):
    # achieved = VALU issue cycles per second = SQ_INSTS_VALU of ONE launch (PMC, same kernel sources) x 4 / the launch time
    # measured live; peak = every SIMD issuing every cycle.  Always <= 1.  SURVEY 8d's algorithmic figure -- which exceeds the
    # HBM peak because the kernel never re-reads the matrix -- is reported beside it, as is the HBM traffic it really has.
    valu = k.get("sq_insts_valu_per_launch")
    peak_issue = CLOCK_GHZ * N_SIMD                                   # G issue cycles / s
    issue_gcs = (valu * 4.0 / (avg_ms * 1e-3) / 1e9) if valu else None
    first = dict(
        bound="valu_issue", kernel="sq_rounds_kernel", unit="G VALU issue cycles/s", peak=round(peak_issue, 1),
        achieved=round(issue_gcs, 1) if issue_gcs else None, frac=round(issue_gcs / peak_issue, 4) if issue_gcs else None,
        traffic=traffic,
        how="the kernel is bound by vector instruction issue and the latency of dependent LDS / L2 loads, not by bandwidth: "
            "achieved = SQ_INSTS_VALU of one launch (rocprofv3 PMC, %s) x 4 cycles / the launch time measured live (HIP events on "
            "the library's stream, %d launches); peak = %.1f GHz x %d SIMDs; frac = achieved / peak (<= 1).  `traffic` = HBM bytes "
            "the launch really moves (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE), `hbm` = that over the launch time against the "
            "%d GB/s peak.  `algorithmic` = SURVEY 8d's figure: 2 N^2 bytes per AnnotateStems evaluation x the evaluations of "
            "the launch (the fp32 upper triangle the reference re-scans every round) over the same time -- above the HBM peak "
            "because the kernel keeps each structure's runs between rounds and cuts them against the chosen stem instead of "
            "re-reading the matrix (8d: 'an implementation that avoids re-reading may exceed 100 %%'); it is an "
            "algorithm-avoidance ratio, not a bandwidth" % (k.get("source", pmc_note or "no PMC file"), reps, CLOCK_GHZ, N_SIMD, HBM_PEAK_GBS),
        binding_resource="VALU issue + dependent LDS / L2 loads; not HBM",
        hbm=dict(traffic_bytes_per_launch=traffic,
                 achieved_GBs=round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                 frac_of_peak=round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None, peak_GBs=HBM_PEAK_GBS),
        hbm_frac_of_real_traffic=round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
        avg_launch_ms=round(avg_ms, 4), launches=int(launches), issue=issue_share(k, avg_ms),
        wave_cycles_waiting=k.get("wait_share"), lds_bank_conflict_share=k.get("lds_conflict_share"),
        algorithmic=dict(bytes_per_launch=round(per_launch), GBs=round(alg_gbs, 1), x_of_hbm_peak=round(alg_gbs / HBM_PEAK_GBS, 3)),
        workload="S1000: %d random-ACGU seqs N=%d seed %d c=fastest pl=1" % (nseq, n, seed), evals_R=int(evals),
        alg_bytes_equals_evals_R_x_2N2=True,
        whole_fold=dict(ms=round(wall * 1e3, 2), seq_per_s=round(nseq / wall, 1),
                        alg_GBs=round(per_launch / wall / 1e9, 1), x_of_hbm_peak=round(per_launch / wall / 1e9 / HBM_PEAK_GBS, 3),
                        how="one sq_fold call, best of 5, profiling off; SURVEY 8d's bytes(N, R) = R 2N^2 over WALL time (the 4 N^2 "
                            "fp32 fill bytes are not counted: this fold writes no matrix at all -- the kernel's only scan forms the "
                            "words of the bit matrix from letter masks in LDS)"),
        kernel_ms=dict(bits=round(fms / reps, 3), rounds=round(avg_ms, 3)),
        dominant_of="S1000 leg: %.0f %% of its kernel time" % (100.0 * ms / max(ms + fms, 1e-9)))
    return first, None


def issue_share(k, launch_ms):
    """Share of the chip's vector-issue slots a kernel's VALU instructions take: a wave64 VALU instruction occupies its
    SIMD for 4 cycles, so wave-instructions (PMC SQ_INSTS_VALU) x 4 / (launch time x clock x SIMDs).  The roof that
    binds the bit-diagonal scan and the fill (they are instruction streams, not byte streams)."""
    v = (k or {}).get("sq_insts_valu_per_launch")
    if not v or not launch_ms:
        return None
    return dict(valu_wave_insts_per_launch=int(v), salu_wave_insts_per_launch=k.get("sq_insts_salu_per_launch"),
                valu_issue_frac=round(v * 4.0 / (launch_ms * 1e-3 * CLOCK_GHZ * 1e9 * N_SIMD), 3),
                how="SQ_INSTS_VALU x 4 cycles / (launch time x %.1f GHz x %d SIMDs)" % (CLOCK_GHZ, N_SIMD))


def fill_leg(nseq=256, n=1000, pmc=None):
    """The API op sq_bpmatrix_fill (a-1 as north_star words it: coalesced HBM writes of the N x N fp32 score matrix) on
    `nseq` S1000 sequences: 4 N^2 bytes written per job (+ the N^2/8 bit matrix), HIP events around the launches."""
    import torch
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import Batch
    names, psets = ParseConfig(builtin_config("fastest"))
    prepared = prepare_synthetic(synthetic("S1000")[:nseq])
    with Batch(prepared, [psets] * nseq, fp32=True) as b:
        b.fill()
        torch.cuda.synchronize()
        b.profile(True)
        b.profile_reset()
        for _ in range(5):
            b.fill()
        torch.cuda.synchronize()
        ms, launches, by = b.profile_get(0)
        b.profile(False)
    per_ms = ms / 5
    gbs = by / 5 / (per_ms * 1e-3) / 1e9 if per_ms > 0 else 0.0
    km = (pmc or {}).get("sq_fill_kernel") or {}
    traffic = (km.get("fetch_bytes_per_launch", 0) + km.get("write_bytes_per_launch", 0)) if km else None
    return dict(kernel="sq_fill_kernel", leg="sq_bpmatrix_fill on %d S1000 sequences" % nseq, bound="hbm",
                unit="GB/s", peak=HBM_PEAK_GBS, achieved=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4),
                traffic=traffic, algorithmic_bytes=round(by / 5), ms_per_fill=round(per_ms, 4), issue=issue_share(km, per_ms),
                pmc=km.get("source"),
                how="achieved = 4 N^2 bytes per job (the fp32 matrix, written once) / time of the fill's launches (HIP events, "
                    "mean of 5); traffic = FETCH_SIZE x 2 + WRITE_SIZE of the same launch (profiles/traffic.json); the fold "
                    "path does not use this op (it writes N^2/8 bytes of bit matrix instead)")


# ---------------------------------------------------------------- the drop-in API end to end
def predict_leg(config, reps=5):
    """Predict() on SRtest150: file in, text out (parse + prepare + upload + fold + format), host buffers on both sides --
    the PCIe / Python-inclusive rate of the drop-in entry point.  Never `value`."""
    import io
    from squarna_amd import Predict
    path = os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtest150.fas")
    ts, chars = [], 0
    for r in range(reps + 1):
        buf = io.StringIO()
        t0 = time.perf_counter()
        Predict(inputfile=path, inputformat="qf", configfile=config, write_to=buf)
        if r:
            ts.append((time.perf_counter() - t0) * 1e3)
        chars = len(buf.getvalue())
    ts.sort()
    return dict(what="squarna_amd.Predict(inputfile=SRtest150.fas, if=qf, c=%s) into a text buffer, median of %d calls" % (config, reps),
                ms_per_call=round(ts[len(ts) // 2], 2), seq_per_s=round(219 / ts[len(ts) // 2] * 1e3, 1), text_chars=chars)


# ---------------------------------------------------------------- streaming: new inputs every step, set-up and read-out timed
def stream_leg(config, K, R, steps, warmup, device):
    """The headline workload as a stream of NEW requests: every step builds K batches of 219 x R records it has not seen
    in the previous step (a window sliding over SRtest150 + SRtrain150: 485 records), and the timed region covers
    Batch() -- host arrays + sq_batch_create, i.e. the upload of the inputs --, the fold of the K batches in flight and
    sq_result_pack_all of every batch.  Record parsing / ProcessReacts (Prepared) is done once, outside.
    Returns (stream dict, one_pass dict): one_pass is K = 1, R = 1 -- one pass over 219 records, create + fold + pack."""
    import torch
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import Batch, Prepared, fold_concurrently
    from squarna_amd.inputs import ParseDefaultInput
    names, psets = ParseConfig(builtin_config(config))
    recs = load_srtest150()
    recs += list(ParseDefaultInput(os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtrain150.fas"), "qf"))
    allp = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
    streams = [torch.cuda.Stream(device) for _ in range(K)]
    # what a server does once it is up: the objects alive now (torch, numpy, the prepared records) leave the collector's
    # generations -- a full collection over them takes 40-60 ms and a step that builds thousands of records triggers one
    # every few steps (tools/stall_probe.py: the same 40 ms inside one fold in ten; none after gc.freeze())
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()                                              # (re-enabled when the leg is over; the steps make no reference cycles.
                                                              # With the collector on, the lists of a few steps add up to one full
                                                              # collection every eighth step or so: 100-140 ms in which both the
                                                              # folding and the building thread stand still -- tools/stream_pipe.py)

    def one_step(t, k, r):
        batches = []
        try:
            for q in range(k):
                start = ((t * k + q) * 97) % len(allp)
                sel = [allp[(start + i) % len(allp)] for i in range(219 * r)]
                with torch.cuda.stream(streams[q]):
                    batches.append(Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096 * r))
            if k == 1:
                batches[0].fold(poollim=1000)
            else:
                fold_concurrently(batches, poollim=1000)
            return sum(int(b.pack_all()[1][-1]) for b in batches)
        finally:
            for b in batches:
                b.close()

    def timed(k, r, nsteps, nwarm):
        for t in range(nwarm):
            one_step(t, k, r)
        torch.cuda.synchronize()
        per, packed = [], 0
        t00 = time.perf_counter()
        for t in range(nsteps):
            t0 = time.perf_counter()
            packed = one_step(nwarm + t, k, r)
            per.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        return time.perf_counter() - t00, sorted(per), packed

    # one_pass FIRST: the leg's later parts leave the process in another state (eight request threads, worker pools and pinned
    # buffers sized for batches of 12 sets) -- round 5's driver line had it LAST and read 4.11 ms where the same code alone
    # reads 3.2-3.3 on the same box (tools/one_pass_probe.py over seven trees from round 4's end to round 6: no commit moves it)
    dt1, per1, packed1 = timed(1, 1, 30, 3)
    dt, per, packed = timed(K, R, steps, warmup)
    stream = dict(what="SRtest150 + SRtrain150 (485 records) as a stream: every step takes the next windows of 219 x %d records for "
                       "%d batches; timed per step: Batch() (host arrays + sq_batch_create = upload) + fold of the batches in flight "
                       "+ sq_result_pack_all; c=%s poollim=1000; %d warm-up steps (the pinned-buffer cache meets every size of the windows "
                       "after five), gc.freeze() + gc.disable() for the leg" % (R, K, config, warmup),
                  seq_per_s=round(219 * R * K * steps / dt, 1), ms_per_step=round(dt / steps * 1e3, 3),
                  median_ms_per_step=round(per[len(per) // 2], 3), max_ms_per_step=round(per[-1], 3), steps=steps, packed_bytes_per_step=packed)
    # the same stream as a server runs it: while the batches of one step fold (a library call: no interpreter lock), a
    # second thread builds the batches of the next one; everything else as above (new records every step, upload and
    # read-out inside the timed region)
    try:
        import threading

        streams2 = [torch.cuda.Stream(device) for _ in range(K)]   # (the next step's uploads must not queue behind this step's kernels)

        def build(t):
            out = []
            for q in range(K):
                start = ((t * K + q) * 97) % len(allp)
                sel = [allp[(start + i) % len(allp)] for i in range(219 * R)]
                with torch.cuda.stream((streams, streams2)[t & 1][q]):
                    out.append(Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096 * R))
            return out

        pipe_ms = []

        pipe_parts = []

        def run_pipelined(nsteps, base):
            nxt = build(base)
            packed_ = 0
            for t in range(nsteps):
                ts0 = time.perf_counter()
                cur, box = nxt, {}
                th = threading.Thread(target=lambda: box.setdefault("b", build(base + t + 1))) if t + 1 < nsteps else None
                if th:
                    th.start()
                try:
                    fold_concurrently(cur, poollim=1000)
                    ts1 = time.perf_counter()
                    packed_ = sum(int(b.pack_all()[1][-1]) for b in cur)
                    ts2 = time.perf_counter()
                finally:
                    for b in cur:
                        b.close()
                    ts3 = time.perf_counter()
                    if th:
                        th.join()
                nxt = box.get("b")
                ts4 = time.perf_counter()
                pipe_ms.append((ts4 - ts0) * 1e3)
                pipe_parts.append(dict(fold=round((ts1 - ts0) * 1e3, 1), pack=round((ts2 - ts1) * 1e3, 1), close=round((ts3 - ts2) * 1e3, 1),
                                       wait_for_build=round((ts4 - ts3) * 1e3, 1)))
            return packed_
        run_pipelined(4, 1000)
        torch.cuda.synchronize()
        del pipe_ms[:]
        del pipe_parts[:]
        t0 = time.perf_counter()
        packed2 = run_pipelined(steps, 2000)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        stream["pipelined"] = dict(seq_per_s=round(219 * R * K * steps / dt2, 1), ms_per_step=round(dt2 / steps * 1e3, 3), packed_bytes_per_step=packed2,
                                   per_step_ms=[round(x, 1) for x in pipe_ms],
                                   median_ms_per_step=round(sorted(pipe_ms)[len(pipe_ms) // 2], 3),
                                   slowest_step=pipe_parts[max(range(len(pipe_ms)), key=lambda q: pipe_ms[q])] if pipe_ms else None,
                                   how="the next step's Batch() calls on a second host thread while this step folds; the first build is inside the time")
    except Exception as e:                                    # (never take the sequential figure down)
        stream["pipelined"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # The same stream through a ROLLING window: K request slots, each a host thread of its own that creates its next batch,
    # folds it, reads it out and destroys it -- no barrier between the slots, so the batches drift apart as the resident
    # batches of the headline do (a step's barrier costs the ramp-up and the matching kernels' tail of EVERY step: a fold of
    # fresh batches 26 ms against 18.8 resident).  New records every batch, upload and read-out inside the time.
    try:
        import threading
        rstreams = [torch.cuda.Stream(device) for _ in range(K)]
        errs = []

        def slot(q, nb, base, out):
            try:
                for t in range(nb):
                    start = (((base + t) * K + q) * 97) % len(allp)
                    sel = [allp[(start + i) % len(allp)] for i in range(219 * R)]
                    with torch.cuda.stream(rstreams[q]):
                        b = Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096 * R)
                    try:
                        b.set_inflight(K)
                        b.fold(poollim=1000)
                        out[q] = int(b.pack_all()[1][-1])
                    finally:
                        b.close()
            except Exception as e:                            # (reported by the main thread)
                errs.append(e)

        def rolling(nb, base):
            out = [0] * K
            th = [threading.Thread(target=slot, args=(q, nb, base, out)) for q in range(K)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            if errs:
                raise errs[0]
            return sum(out)
        rolling(3, 3000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        packed3 = rolling(steps, 4000)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
        stream["rolling"] = dict(seq_per_s=round(219 * R * K * steps / dt3, 1), ms_per_batch_round=round(dt3 / steps * 1e3, 3), packed_bytes_last_round=packed3,
                                 how="%d request slots, each a host thread that runs Batch() -> sq_fold -> sq_result_view -> close for %d new "
                                     "batches of 219 x %d records one after the other, no barrier between the slots (sq_batch_set_inflight(%d)); "
                                     "the interpreter lock serialises the slots' Python parts" % (K, steps, R, K))
    except Exception as e:
        stream["rolling"] = {"error": "%s: %s" % (type(e).__name__, e)}
    dt1_last, per1_last, _ = timed(1, 1, 10, 3)
    gc.enable()
    one = dict(what="ONE pass over 219 records (a different window every call): Batch() + sq_fold + sq_result_pack_all, nothing "
                    "else in flight, median of 30, measured before the leg's batches of 8 x 12 sets",
               ms_after_the_stream_legs=round(per1_last[len(per1_last) // 2], 3),
               ms=round(per1[len(per1) // 2], 3), best_ms=round(per1[0], 3),
               seq_per_s=round(219 / per1[len(per1) // 2] * 1e3, 1), packed_bytes=packed1)
    return stream, one


# ---------------------------------------------------------------- strong scaling: a synthetic workload sharded over the ranks
SUB_BATCHES = {"S300": 4, "S1000": 2, "S2000": 4}     # concurrent batches per rank that measured best at world size 1


def sharded_leg(workload, steps, warmup, rank, world, device, sub_batches=0):
    """The SURVEY 8d workload sharded with lpt_partition (cost N^2), every rank folds its shard (inputs resident), then
    ONE RCCL all_gather of the packed results (sq_result_pack_all) to every rank; rank 0 holds all records.  Returns
    the rank-0 dict (None elsewhere)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import Batch
    from squarna_amd.parallel import lpt_partition, gather_bytes
    names, psets = ParseConfig(builtin_config("fastest"))
    items = synthetic(workload)
    parts = lpt_partition([float(len(s)) ** 2 for s, _ in items], world)
    mine = parts[rank]
    prepared = prepare_synthetic([items[k] for k in mine])
    # the rank's shard as `sub_batches` batches folded concurrently (sq_fold_concurrent): the host bookkeeping of one
    # overlaps the kernels of the others; contiguous slices, so the concatenated packs keep the shard's order
    from squarna_amd.engine import fold_concurrently
    if sub_batches <= 0:
        sub_batches = SUB_BATCHES.get(workload, 4)
    nb = max(1, min(sub_batches, len(prepared) // 64 or 1))
    cuts = [len(prepared) * q // nb for q in range(nb + 1)]
    batches = []
    for q in range(nb):
        with torch.cuda.stream(torch.cuda.Stream(device)):
            batches.append(Batch(prepared[cuts[q]:cuts[q + 1]], [psets] * (cuts[q + 1] - cuts[q]),
                                 max_structs=max(cuts[q + 1] - cuts[q], 1), fp32=False))
    torch.cuda.synchronize()

    def pack_shard():
        bufs, offs, base = [], [np.zeros(1, np.int64)], 0
        for b in batches:
            buf, off = b.pack_all()
            bufs.append(buf)
            offs.append(off[1:] + base)
            base += int(off[-1])
        return np.concatenate(bufs), np.concatenate(offs)

    def step():
        if nb == 1:
            batches[0].fold(poollim=1)
        else:
            fold_concurrently(batches, poollim=1)
        buf, off = pack_shard()
        if world == 1:
            return [(buf, off)]
        head = np.concatenate([np.array([len(mine)], np.int64), np.array(mine, np.int64), off]).view(np.uint8)
        # the product's own result gather (squarna_amd.parallel.gather_bytes, also the end of PredictSharded): exact
        # sizes, to rank 0 only; the packed records go as their own segment (a single batch: straight from the library's
        # pinned result buffer)
        return gather_bytes([head, buf], 0, device)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(3):                                        # set-up, not warm-up: page-in of the workspaces and pinned
        step()                                                # buffers, worker pools, allocator arenas
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    fence()
    dt_mine = dt = time.perf_counter() - t0
    per_rank_ms = [round(dt / steps * 1e3, 3)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        every = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [round(float(x.item()) / steps * 1e3, 3) for x in every]
        dt = max(float(x.item()) for x in every)
    # verification on rank 0: every record arrived once, and a sample of other ranks' records equals a local fold
    ok, checked = True, 0
    if rank == 0 and world > 1:
        seen = {}
        for r in range(world):
            raw = np.ascontiguousarray(last[r])
            cnt = int(raw[:8].view(np.int64)[0])
            idx = raw[8:8 + 8 * cnt].view(np.int64)
            off = raw[8 + 8 * cnt:8 + 8 * cnt + 8 * (cnt + 1)].view(np.int64)
            payload = raw[8 + 8 * cnt + 8 * (cnt + 1):]
            for q in range(cnt):
                seen[int(idx[q])] = payload[off[q]:off[q + 1]].tobytes()
        ok = sorted(seen) == list(range(len(items)))
        sample = [k for k in range(0, len(items), max(1, len(items) // 48)) if k not in set(mine)][:48]
        if sample:
            with Batch(prepare_synthetic([items[k] for k in sample]), [psets] * len(sample), fp32=False) as bb:
                bb.fold(poollim=1)
                buf, off = bb.pack_all()
                for q, k in enumerate(sample):
                    # (the `evals` word and everything else is per record: bytes must be identical)
                    ok = ok and seen[k] == buf[off[q]:off[q + 1]].tobytes()
                    checked += 1
    evals = sum(b.evals(k) for b in batches for k in range(b.nseq))
    for b in batches:
        b.close()
    if rank != 0:
        return None
    return dict(workload="%s: %d seqs, c=fastest pl=1 (greedy rounds chained on the device), sharded by lpt_partition (N^2) over "
                         "%d rank(s); step = fold of the resident shard (as %d concurrent batches) + sq_result_pack_all + the "
                         "result gather to rank 0 (parallel.gather_bytes: exact sizes, RCCL send / recv)" % (workload, len(items), world, nb),
                seq_per_s=round(len(items) * steps / dt, 1), ms_per_step=round(dt / steps * 1e3, 3), steps=steps,
                n_ranks_seen=world if world == 1 else dist.get_world_size(), ms_per_step_by_rank=per_rank_ms,
                records_rank0=len(mine), evals_R_rank0=int(evals),
                gathered_records_complete=bool(ok), records_checked_against_local_fold=checked)


# ---------------------------------------------------------------- strong-scaling proxy on ONE GPU
def scaling_proxy_leg(workloads=("S300", "S1000", "S2000"), shards=(2, 4, 8), reps=5):
    """What a rank of an N-GPU run folds, measured at world size 1: rank 0's lpt_partition shard of the SURVEY 8d workload
    folded alone (resident inputs, fold + sq_result_pack_all, one batch), next to the whole workload.  The shards are
    independent (no data-path collective; the packed results of a rank are KBs), so T(shard) bounds the step of an N-GPU
    run from below and predicted_efficiency = T(full) / (N x T(shard)) is what strong scaling can reach at most."""
    import torch
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import Batch
    from squarna_amd.parallel import lpt_partition
    names, psets = ParseConfig(builtin_config("fastest"))
    out = {}
    for wl in workloads:
        items = synthetic(wl)
        cost = [float(len(s)) ** 2 for s, _ in items]

        def time_of(idx):
            prepared = prepare_synthetic([items[k] for k in idx])
            with Batch(prepared, [psets] * len(prepared), max_structs=max(len(prepared), 1), fp32=False) as b:
                for _ in range(2):
                    b.fold(poollim=1)
                    b.pack_all()
                ts = []
                for _ in range(reps):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    b.fold(poollim=1)
                    b.pack_all()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2] * 1e3
        full = time_of(list(range(len(items))))
        row = {"records": len(items), "ms_full": round(full, 3)}
        for n in shards:
            mine = lpt_partition(cost, n)[0]
            t = time_of(mine)
            row["shard_of_%d" % n] = {"records_rank0": len(mine), "ms_shard": round(t, 3),
                                      "predicted_efficiency": round(full / (n * t), 3)}
        out[wl] = row
    out["how"] = ("one GPU: ms_full = fold + pack of the whole workload as one resident batch (median of %d); ms_shard = the same for rank "
                  "0's lpt_partition (N^2) shard of an N-rank run; predicted_efficiency = ms_full / (N x ms_shard) -- an upper bound of "
                  "strong scaling (the gather of the packed records and rank imbalance come on top)" % reps)
    return out


# ---------------------------------------------------------------- SHAPE data under nobpp
def shape_leg(recs, config, K, R, steps, device):
    """SRtest150 with a reactivity line per record (drawn per position from "_+#", p = 0.5 / 0.3 / 0.2: raw 0.0 / 0.5 / 1.0
    through ProcessReacts, the S2000 recipe of SURVEY 8d), c=nobpp: the reference's main use case with probing data.  Every
    E / H job then needs stemscore ** 1.7 from the host libm (bulk, sq_algos_dev.h); RunAlgo itself stays on the device."""
    import numpy as np
    import torch
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.dbn import ProcessReacts, ReactDict
    from squarna_amd.engine import Batch, Prepared, fold_concurrently
    names, psets = ParseConfig(builtin_config(config))
    rng = np.random.default_rng(150)
    prepared = []
    for _, seq, reacts, restr, ref in recs:
        line = rng.choice(list("_+#"), len(seq), p=[0.5, 0.3, 0.2])
        prepared.append(Prepared(seq, ProcessReacts([ReactDict[c] for c in line], M=1.8, B=-0.6), restr, ref))
    n = len(prepared)
    lat = []
    for q in range(8):                                        # one pass: Batch() + fold + pack, nothing else in flight
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with Batch(prepared, [psets] * n, fp32=False) as b:
            b.fold(poollim=1000)
            b.pack_all()
            paths = b.fold_paths
        torch.cuda.synchronize()
        if q >= 2:
            lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    batches = []
    for _ in range(K):
        with torch.cuda.stream(torch.cuda.Stream(device)):
            batches.append(Batch(prepared * R, [psets] * (n * R), fp32=False, max_structs=4096 * R))
    torch.cuda.synchronize()
    try:
        fold_concurrently(batches, reps=2, poollim=1000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fold_concurrently(batches, reps=steps, poollim=1000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        paths_many = min(b.fold_paths for b in batches)
    finally:
        for b in batches:
            b.close()
    return dict(what="SRtest150 + a '_+#' reactivity line per record, c=%s poollim=1000" % config,
                one_pass_ms=round(lat[len(lat) // 2], 3), one_pass_seq_per_s=round(n / lat[len(lat) // 2] * 1e3, 1),
                resident=dict(batches_in_flight=K, sets_per_batch=R, steps=steps, ms_per_step=round(dt / steps * 1e3, 3),
                              seq_per_s=round(n * R * K * steps / dt, 1)),
                runalgo_on_device=bool(paths & 2) and bool(paths_many & 2), tail_on_device=bool(paths & 1))


# ---------------------------------------------------------------- branching pools on long sequences
def pools_long_leg(count=2000, n=500, reps=3):
    """`count` random-ACGU sequences of `n` nt under 500nobpp -- the reference's own configuration from 500 nt on
    (SQUARNA.py:875-878 -> 500.conf; two greedy paramsets with suboptimality ranges 0.9-0.95, i.e. pools that branch, + E / H /
    N) -- at the default pool limit of 1000 (SQUARNA.py:421), through the engine (HipEngine.fold_records_packed: sub-batches
    sized to the device pools' slots; host lists in, packed records out, handed over without a copy).  From round 6 a round of
    the pools of 257-1,024 nt structures is ONE launch of the list form of the round kernel (sq_pool_round.hip: every
    structure reads the list its parent left -- runs, bpscores, the finalscores the new stem does not touch -- and scores what
    changed); until then it was six launches in twenty chunks + a host round trip, every structure scanned and scored anew
    (SQ_NO_POOL_KEPT=1: that form)."""
    import torch
    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import HipEngine
    names, psets = ParseConfig(builtin_config("500nobpp"))
    recs = [(sq, None, None, None, psets, None) for sq in pools_long_sequences(count, n)]
    eng = HipEngine()
    ts = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.fold_records_packed(recs, poollim=1000)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = min(ts[1:])
    return dict(what="%d random-ACGU sequences of %d nt (seed 500), c=500nobpp poollim=1000, HipEngine.fold_records_packed (host lists in, "
                     "packed records out), best of %d calls after one warm-up" % (count, n, reps),
                seconds=round(best, 3), seq_per_s=round(count / best, 1), packed_bytes=int(sum(len(o) for o in out)),
                driver=int(eng.last_fold_driver), peak_structures_first_sub_batch=int(eng.last_fold_peak))


def alignment_leg(pmc, nseq=512, ncol=5000, reps=3):
    """BASELINE config 5 (SQRNdbnali.py:60-108,211-242,332-458): the 512 x 5000 synthetic alignment under ali.conf
    through squarna_amd.Predict(alignment=True) -- step 1 alone (s3=1: SURVEY's A5000) and all three steps (step3=u), text digests --
    and, from a replay of the two device phases with the library's HIP-event timers on the batch's stream, the kernels that
    carry them: step 1's scan / filter / scatter / select, step 2's ONE launch of the persistent round kernel over 512
    structures of ~4,700 nt."""
    import hashlib, io, tempfile
    import numpy as np
    import torch
    import ctypes as C
    from squarna_amd import Predict, align, _lib
    from squarna_amd import engine as E
    from squarna_amd.config import ParseConfig, builtin_config
    text = a5000_msa(nseq, ncol)
    with tempfile.NamedTemporaryFile("w", suffix=".afa", delete=False) as f:
        f.write(text)
        path = f.name
    try:
        def run(step3):
            buf = io.StringIO()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            Predict(inputfile=path, alignment=True, step3=step3, write_to=buf)
            torch.cuda.synchronize()
            return time.perf_counter() - t0, hashlib.sha256(buf.getvalue().encode()).hexdigest()[:16]
        run("u")                                                    # warm-up (allocator, pinned buffers)
        full = sorted(run("u") for _ in range(reps))
        one = sorted(run("1") for _ in range(reps))
        # phases of one more call (wall clock around the alignment's own functions)
        T = {}
        saved = []

        def timed(mod, name, key):
            fn = getattr(mod, name)

            def w(*a, **k):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    torch.cuda.synchronize(); T[key] = T.get(key, 0.0) + time.perf_counter() - t0
            saved.append((mod, name, fn))
            setattr(mod, name, w)
        timed(align, "SQRNdbnali", "step1_s"); timed(align, "MatrixToDBNs", "step1_first_fit_s")
        timed(E.HipEngine, "fold_records", "step2_fold_s"); timed(align, "Consensus", "consensus_s")
        try:
            run("u")
        finally:
            for mod, name, fn in saved:
                setattr(mod, name, fn)
    finally:
        os.unlink(path)
    # ---- the device phases once more, under the library's per-kernel timers ----
    seqs = [ln.upper().replace("T", "U") for ln in text.split("\n") if ln and not ln.startswith(">")]
    names, psets = ParseConfig(builtin_config("ali"))
    ps0 = psets[0]
    eng = E.HipEngine()
    dev = torch.device("cuda", torch.cuda.current_device())
    ps1 = dict(bpweights=ps0["bpweights"], bpp=0, algorithms={"G"}, suboptmax=1.0, suboptmin=1.0, suboptsteps=1.0, minlen=ps0["minlen"],
               minbpscore=ps0["minbpscore"], minfinscorefactor=1.0, bracketweight=-2.0, distcoef=0.09, orderpenalty=1.0, loopbonus=0.125, maxstemnum=1e6)
    matrix = torch.zeros((ncol, ncol), dtype=torch.float64, device=dev)
    pk = E.PackedRows(seqs, None)
    n2 = float((pk.lengths.astype(np.float64) ** 2).sum())
    with E.Batch(pk, [[ps1]] * nseq, cand_per_nt=64, fp32=False) as b:
        b.profile(True); b.profile_reset()
        b.align_accumulate_packed(pk, matrix)
        k1 = {nm: b.profile_get(k) for k, nm in ((0, "bits"), (1, "state"), (2, "scan"), (3, "filter"), (8, "scatter"))}
    # cells the scatter adds: from the stems of the first 8 rows (exact for those, extrapolated)
    st8 = eng.yield_stems([(s, None, None) for s in seqs[:8]], ps0["bpweights"], ps0["minlen"], ps0["minbpscore"])
    cells_row = float(np.mean([int(st["len"].sum()) for _, st in st8]))
    stream = torch.cuda.current_stream(dev)
    idx = torch.empty(1 << 20, dtype=torch.int64, device=dev); val = torch.empty(1 << 20, dtype=torch.float64, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sel_ms = []
    for _ in range(4):                                              # (launched on torch's current stream: its events see the kernel)
        e0.record(stream)
        _lib.check(_lib.load().sq_colmatrix_select(C.c_void_p(matrix.data_ptr()), ncol, float(ps0["minbpscore"] * nseq), 4, C.c_void_p(idx.data_ptr()),
                                                   C.c_void_p(val.data_ptr()), 1 << 20, C.c_void_p(cnt.data_ptr()), C.c_void_p(stream.cuda_stream)))
        e1.record(stream); e1.synchronize()
        sel_ms.append(e0.elapsed_time(e1))
    sel = min(sel_ms[1:])
    smat = (matrix / matrix.max() * 5).contiguous()
    recs = [(s, None, None, None, psets, smat) for s in seqs]
    opts = dict(conslim=1, toplim=5, hardrest=False, rankbydiff=False, rankby=(0, 2, 1), interchainonly=False, poollim=1000, algos=set(),
                levellimit=None, priority=set())
    b2, fo = eng._make_batch(recs, None, opts)
    try:
        b2.fold(**fo)                                               # warm-up
        b2.profile(True); b2.profile_reset()
        b2.fold(**fo)
        rms, rl, rbytes = b2.profile_get(7)
        evals = sum(b2.evals(k) for k in range(nseq))
        paths = b2.fold_paths
    finally:
        b2.close()
    kr, ks, kc = pmc.get("a5000_rounds") or {}, pmc.get("a5000_scatter") or {}, pmc.get("a5000_colselect") or {}

    def hbm(k, ms):
        tb = (k.get("fetch_bytes_per_launch", 0) + k.get("write_bytes_per_launch", 0)) if k else None
        return dict(traffic=tb, achieved_GBs=round(tb / (ms * 1e-3) / 1e9, 1) if tb and ms else None,
                    frac=round(tb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tb and ms else None, pmc=k.get("source") if k else None)
    sc_ms, sc_l = k1["scatter"][0], max(k1["scatter"][1], 1)
    sc_alg = 16.0 * cells_row * nseq                                  # a read-modify-write of one double per cell of every kept stem
    sel_alg = 8.0 * ncol * (ncol - 4) / 2.0                           # the upper cells with span >= 4, read once
    rooflines = [
        dict(kernel="sq_scatter_all_kernel", leg="A5000 step 1, iteration 1 (512 rows into the 5000 x 5000 fp64 column matrix, hardware fp64 atomics)",
             bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS, avg_launch_ms=round(sc_ms / sc_l, 3), launches=int(sc_l),
             algorithmic_bytes=round(sc_alg), achieved=round(sc_alg / (sc_ms * 1e-3) / 1e9, 1), frac=round(sc_alg / (sc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             algorithmic_how="16 bytes (read + write of a double) x the cells of every kept stem: %.0f cells per row (exact for the first 8 rows, "
                             "extrapolated) x %d rows; the addends are exact (dyadic weights), so the order of the atomics cannot change a bit" % (cells_row, nseq),
             hbm=hbm(ks, sc_ms / sc_l)),
        dict(kernel="sq_colselect_kernel", leg="A5000 step 1 (MatrixToDBNs' candidates: the upper cells >= minbpscore x rows, span >= 4)",
             bound="hbm", unit="GB/s", peak=HBM_PEAK_GBS, avg_launch_ms=round(sel, 4), algorithmic_bytes=round(sel_alg),
             achieved=round(sel_alg / (sel * 1e-3) / 1e9, 1), frac=round(sel_alg / (sel * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             how="HIP events on torch's current stream (the stream the kernel is launched on), best of 3 after a warm-up", hbm=hbm(kc, sel)),
        dict(kernel="sq_rounds_kernel", leg="A5000 step 2: ONE launch, 512 structures of ~4,700 nt (1,024-thread blocks, one per CU), weights read "
                                            "through the gap map from the shared diagonal-major matrix",
             bound="latency + hbm: a round streams the structure's run list (16 + 16 bytes per run) and scores the runs whose bound reaches the bar",
             avg_launch_ms=round(rms / max(rl, 1), 2), launches=int(rl), evals_R=int(evals),
             algorithmic=dict(bytes_per_launch=round(rbytes / max(rl, 1)), GBs=round(rbytes / max(rl, 1) / (rms / max(rl, 1) * 1e-3) / 1e9, 1),
                              how="SURVEY 8d: 2 N^2 bytes per AnnotateStems evaluation x the launch's evaluations"),
             valu_issue_frac=(issue_share(kr, rms / max(rl, 1)) or {}).get("valu_issue_frac"), wave_cycles_waiting=kr.get("wait_share"),
             lds_bank_conflict_share=kr.get("lds_conflict_share"), hbm=hbm(kr, rms / max(rl, 1)), fold_paths=int(paths)),
    ]
    return dict(
        what="BASELINE config 5: %d x %d synthetic alignment (seed 5000: one ancestor, substitutions 0.12, gaps 0.06, %d planted helices), ali.conf, "
             "squarna_amd.Predict(inputfile, alignment=True) into a text buffer; median of %d calls after one warm-up" % (nseq, ncol, ncol // 40, reps),
        step1_only=dict(seconds=round(one[len(one) // 2][0], 3), sha256_16=one[0][1], how="step3='1' (SURVEY 8d's A5000: both iterations of step 1, steps 2-3 skipped)"),
        all_steps=dict(seconds=round(full[len(full) // 2][0], 3), sha256_16=full[0][1], how="step3='u'"),
        phases={k: round(v, 3) for k, v in T.items()},
        device_ms=dict(step1_iteration1={nm: round(v[0], 3) for nm, v in k1.items()}, colselect=round(sel, 4), step2_rounds_kernel=round(rms / max(rl, 1), 2)),
        cells_N2_sum=n2, rooflines=rooflines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="nobpp")
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent SRtest150 batches in flight per GPU (0 = 8)")
    ap.add_argument("--replicas", type=int, default=12,
                    help="copies of the 219-record SRtest150 set per batch (kernels and host rounds are shared by the copies)")
    ap.add_argument("--workload", default="srtest150", choices=["srtest150", "S300", "S1000", "S2000"])
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="strong-scaling mode: concurrent batches per rank (0 = the workload's measured best: %s)" % SUB_BATCHES)
    ap.add_argument("--regions", type=int, default=3, help="timed regions of `steps` steps each; the median one is reported")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-stream", action="store_true", help="skip the stream / one_pass legs")
    ap.add_argument("--no-roofline", action="store_true", help="skip the S1000 roofline leg")
    ap.add_argument("--no-alignment", action="store_true", help="skip the alignment leg (BASELINE config 5)")
    ap.add_argument("--roofline-seqs", type=int, default=1024)    # S1000 = 1,024 sequences
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # (squarna_amd sets the same default at import; see its __init__)

    recs = load_srtest150()
    cpu = cpu_workers = None
    if rank == 0 and world == 1 and not args.no_cpu and args.workload == "srtest150":
        cpu_workers = cpu_workers_start(recs, args.config)   # before any GPU initialisation; the leg itself runs last

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    if args.workload != "srtest150":                          # strong-scaling mode
        res = sharded_leg(args.workload, args.steps, args.warmup, rank, world, device, args.sub_batches)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({
                "metric": "sequences/sec (%s synthetic, single-sequence mode, sharded)" % args.workload,
                "value": res["seq_per_s"], "unit": "seq/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic i.i.d. uniform ACGU (SURVEY 8d seeds)",
                "config": {"workload": res["workload"]}, "sharded": res}))
        return

    from squarna_amd.config import ParseConfig, builtin_config
    from squarna_amd.engine import Batch, Prepared, fold_concurrently
    names, psets = ParseConfig(builtin_config(args.config))
    prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    # (the fold is device-resident: a batch in flight costs its host thread a few launches and waits per round, so the
    # number of batches in flight does not follow the CPU count any more -- 2 to 4 CPUs are busy with 8 batches)
    K = args.inflight if args.inflight > 0 else 8
    R = max(1, args.replicas)
    nset = len(prepared)                                      # 219 records
    batches = []
    for _ in range(K):                                        # inputs resident in HBM; one stream per batch
        with torch.cuda.stream(torch.cuda.Stream(device)):
            batches.append(Batch(prepared * R, [psets] * (nset * R), fp32=False, max_structs=4096 * R))
    torch.cuda.synchronize()

    def run(nsteps):
        """nsteps steps: every batch folded nsteps times.  The batches run free (sq_fold_concurrent_n: one host thread per
        batch folds its batch back to back, no barrier between the steps), as resident batches serving a stream would."""
        if nsteps <= 0:
            return
        if K == 1:
            for _ in range(nsteps):
                batches[0].fold(poollim=1000)
        else:
            fold_concurrently(batches, reps=nsteps, poollim=1000)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    # Three timed regions of exactly `steps` steps each, every one bracketed by barrier + synchronize and reduced with MAX
    # over the ranks; the line reports the MEDIAN region (the folds are chains of short kernels: host jitter shows).
    regions = []
    for _ in range(max(1, args.regions)):
        fence()
        t0 = time.perf_counter()
        cpu0 = time.process_time()
        run(args.steps)
        fence()
        dt = time.perf_counter() - t0
        host_cpu = time.process_time() - cpu0                 # CPU time of this rank's process (all its threads)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        regions.append((dt, host_cpu))
    region_ms = [round(r[0] / args.steps * 1e3, 3) for r in regions]
    dt, host_cpu = sorted(regions)[len(regions) // 2]

    # ---- secondary, OUTSIDE the timed region ----------------------------------------------------------------------
    results = [batches[0].result(k) for k in range(nset)]
    fs_c, fs_b = mean_fs(results)
    evals = sum(batches[0].evals(k) for k in range(nset))
    same = all(repr(b.result(k + r * nset)) == repr(results[k]) for b in batches for r in range(R) for k in range(r, nset, 7))
    for b in batches:
        b.close()
    # one batch alone, set up as a caller with a single batch would (its own worker pool, nothing else in flight)
    b0 = Batch(prepared, [psets] * len(prepared), fp32=False)
    for _ in range(3):
        b0.fold(poollim=1000)
    lat = []
    for _ in range(10):                                       # one batch alone: the latency of a single fold
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        b0.fold(poollim=1000)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t1) * 1e3)
    lat.sort()
    # per-kernel time of one fold (HIP events on the streams the kernels run on): 3 extra folds of one batch; the
    # matching kernels run on side streams, concurrently with the greedy rounds
    b0.profile(True)
    b0.profile_reset()
    for _ in range(3):
        b0.fold(poollim=1000)
    torch.cuda.synchronize()
    kernel_ms = {nm: round(b0.profile_get(k)[0] / 3, 3) for k, nm in enumerate(
        ["bits", "state", "scan", "score_select", "edmonds", "hungarian", "nussinov"])}
    mwm = b0.mwm_counters()
    b0.profile(False)

    same = same and all(repr(b0.result(k)) == repr(results[k]) for k in range(0, len(prepared), 5))
    b0.close()
    # the same batch under greedynobpp (no Edmonds / Hungarian / Nussinov job beside the pools: the fold IS the greedy loop
    # of the device pools, a chain of rounds) -- with the rounds enqueued ahead of the host and one by one
    greedy_alone = None
    if not args.no_stream:
        from squarna_amd.config import ParseConfig as _PC, builtin_config as _bc
        gpsets = _PC(_bc("greedynobpp"))[1]
        greedy_alone = {}
        for tag, env in (("ms_per_fold", None), ("ms_per_fold_rounds_one_by_one", "0")):
            if env is not None:
                os.environ["SQ_POOL_AHEAD"] = env
            try:
                with Batch(prepared, [gpsets] * len(prepared), fp32=False) as bg:
                    for _ in range(3):
                        bg.fold(poollim=1000)
                    gl = []
                    for _ in range(10):
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        bg.fold(poollim=1000)
                        torch.cuda.synchronize()
                        gl.append((time.perf_counter() - t1) * 1e3)
                    greedy_alone[tag] = round(sorted(gl)[len(gl) // 2], 3)
                    if env is None:
                        greedy_alone["rounds_enqueued_ahead"] = bool(bg.fold_paths & 32)
            finally:
                os.environ.pop("SQ_POOL_AHEAD", None)
        greedy_alone["how"] = "ONE 219-record batch alone, c=greedynobpp poollim=1000, median of 10 folds"

    pmc, pmc_note = load_pmc()
    rooflines = []
    if mwm["max_passes"]:
        e_ms = kernel_ms["edmonds"]
        passes = mwm["max_passes"]                            # (per graph: the same in each of the 3 profiled folds)
        km = pmc.get("sq_mwm_kernel") or {}
        obj = dict(kernel="sq_mwm_kernel", leg="SRtest150 (the headline step)",
                   bound="latency: one wave per graph walks a chain of dependent LDS reads (no bandwidth or FLOP roof applies)",
                   share_of_fold_kernel_time=round(e_ms / max(sum(kernel_ms.values()), 1e-9), 3),
                   avg_launch_ms=e_ms,
                   critical_graph=dict(vertices=mwm["n"], edges=mwm["m"], scan_passes=passes, events=mwm["max_events"]),
                   cycles_per_scan_pass=round(e_ms * 1e-3 * CLOCK_GHZ * 1e9 / max(passes, 1)),
                   cycles_how="whole kernel time (its slowest wave = the critical graph) x %.1f GHz / that graph's scan passes "
                              "(a pass scans up to four queue vertices: their neighbour lists share the 64 lanes); includes its "
                              "events (16 %% of the time), dual steps (9 %%) and stage set-up (5 %%): profiles/r04_mwm_phases.txt" % CLOCK_GHZ,
                   floor_cycles_per_scan_pass=1500,
                   floor_how="4 dependent LDS round trips per pass (the lists' slots -> neighbour -> its blossom's label -> "
                             "the best-edge minimum read back; ~130 cycles each for a lone wave; the queue entries of the next "
                             "pass are fetched alongside) + ~200 instructions (segment set-up, classification, the ordered "
                             "best-edge stores) at the ~4.7 cycles / instruction one wave alone issues at (PMC: "
                             "SQ_ACTIVE_INST_ANY / instructions); the pass count itself is bounded below by events + substages")
        if km:
            tb = km.get("fetch_bytes_per_launch", 0) + km.get("write_bytes_per_launch", 0)
            obj.update(hbm=dict(traffic=tb, achieved_GBs=round(tb / (e_ms * 1e-3) / 1e9, 3),
                                frac=round(tb / (e_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6)),
                       wave_cycles_waiting=km.get("wait_share"), pmc=km.get("source"))
        rooflines.append(obj)

    kp = pmc.get("sq_pool_round_kernel") or {}
    if kp:
        # the headline step's largest kernel by time and wave cycles (one wave takes a structure of a pool through a round); its
        # launches are hundreds per step and overlap, so the entry is the PMC pass's own per-launch figures, not a live timing
        us = kp.get("avg_launch_us_under_pmc") or 0.0
        valu = kp.get("sq_insts_valu_per_launch")
        tb = kp.get("fetch_bytes_per_launch", 0) + kp.get("write_bytes_per_launch", 0)
        rooflines.append(dict(
            kernel="sq_pool_round_kernel", leg="SRtest150 (the headline step), serialised under rocprofv3 --pmc",
            bound="latency: one wave per structure runs the round's phases one after the other (state, scan, ScoreStems, ChooseStems); "
                  "no bandwidth or FLOP roof applies",
            avg_launch_us_under_pmc=us, launches_in_probe=kp.get("launches_in_probe"),
            valu_issue_frac=round(valu * 4.0 / (us * 1e-6 * CLOCK_GHZ * 1e9 * N_SIMD), 4) if valu and us else None,
            wave_cycles_waiting=kp.get("wait_share"), lds_bank_conflict_share=kp.get("lds_conflict_share"),
            waves_per_launch=kp.get("sq_waves_per_launch"),
            hbm=dict(traffic=tb, achieved_GBs=round(tb / (us * 1e-6) / 1e9, 2) if us else None,
                     frac=round(tb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if us else None),
            pmc=kp.get("source")))

    roof = None
    if rank == 0 and not args.no_roofline:
        roof, other_obj = roofline_leg(args.roofline_seqs, 1000, pmc, pmc_note)   # the leg's dominant kernel first
        if other_obj:
            rooflines.append(other_obj)
        try:
            rooflines.append(fill_leg(pmc=pmc))
        except Exception as e:                                # (a secondary leg never takes the headline down)
            rooflines.append({"kernel": "sq_fill_kernel", "error": "%s: %s" % (type(e).__name__, e)})

    end_to_end = None
    if rank == 0 and world == 1:
        try:
            end_to_end = predict_leg(args.config)
        except Exception as e:                                # (a secondary leg never takes the headline down)
            end_to_end = {"error": "%s: %s" % (type(e).__name__, e)}

    shape = proxy = None
    if rank == 0 and world == 1 and not args.no_stream:
        torch.cuda.empty_cache()
        try:
            shape = shape_leg(recs, args.config, K, R, max(3, args.steps // 4), device)
        except Exception as e:                                # (a secondary leg never takes the headline down)
            shape = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not args.no_roofline:
        try:
            proxy = scaling_proxy_leg()
        except Exception as e:                                # (a secondary leg never takes the headline down)
            proxy = {"error": "%s: %s" % (type(e).__name__, e)}

    alignment = None
    if rank == 0 and world == 1 and not args.no_alignment:
        torch.cuda.empty_cache()
        try:
            alignment = alignment_leg(pmc)
            rooflines.extend(alignment.pop("rooflines"))
        except Exception as e:                                # (a secondary leg never takes the headline down)
            alignment = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    pools_long = None
    if rank == 0 and world == 1 and not args.no_stream:
        torch.cuda.empty_cache()
        try:
            pools_long = pools_long_leg()
        except Exception as e:                                # (a secondary leg never takes the headline down)
            pools_long = {"error": "%s: %s" % (type(e).__name__, e)}

    stream = one_pass = None
    if rank == 0 and world == 1 and not args.no_stream:
        torch.cuda.empty_cache()                              # (every leg starts from a clean allocator: the blocks the legs before it left
        try:                                                  # cached have other sizes, and a leg that allocates per step then pays for them)
            from squarna_amd import _lib
            _lib.load().sq_host_cache_trim()                  # (the same for the library's idle pinned buffers: the S2000 legs leave
                                                              # hundreds of MB each, evicted one hipHostFree at a time -- a 465 ms step)
            stream, one_pass = stream_leg(args.config, K, R, max(5, args.steps // 2), 6, device)
        except Exception as e:                                # (a secondary leg never takes the headline down)
            stream = {"error": "%s: %s" % (type(e).__name__, e)}
        import gc
        gc.enable()                                           # (the leg switches the collector off for its steps)

    # the CPU baseline LAST (its workers have been waiting since before the GPU was initialised): all cores busy for ten
    # seconds in front of the GPU legs made the latency-bound ones slower
    if cpu_workers is not None:
        cpu = cpu_baseline(cpu_workers, recs, args.config)
        if one_pass and "seq_per_s" in one_pass:
            one_pass["vs_cpu_baseline"] = round(one_pass["seq_per_s"] / cpu["value"], 1)
        if stream and "seq_per_s" in stream:
            stream["vs_cpu_baseline"] = round(stream["seq_per_s"] / cpu["value"], 1)
        if end_to_end and "seq_per_s" in end_to_end:
            end_to_end["vs_cpu_baseline"] = round(end_to_end["seq_per_s"] / cpu["value"], 1)

    sharded = None
    if world > 1:
        # strong scaling next to the weak-scaling value: the three SURVEY 8d workloads sharded over the ranks (short legs)
        sharded = {}
        for wl, st_, wu_ in (("S300", 5, 2), ("S1000", 5, 2), ("S2000", 3, 1)):
            try:
                sharded[wl] = sharded_leg(wl, st_, wu_, rank, world, device)
            except Exception as e:                            # (never let a secondary leg take the headline down)
                sharded[wl] = {"error": "%s: %s" % (type(e).__name__, e)} if rank == 0 else None
        nranks_seen = dist.get_world_size()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass
    if rank != 0:
        return
    per_step = nset * R * K
    line = {
        "metric": "sequences/sec (SRtest150, single-sequence mode)",
        "value": round(per_step * world * args.steps / dt, 1),
        "unit": "seq/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "timed_regions_ms_per_step": region_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "dtype_note": "scores and every decision in fp64, in the reference's operation order; the scan itself works on 1-bit cell activity",
        "data": "SRtest150.fas shipped with the reference (219 records, 8-150 nt, reference dbn per record)",
        "config": {"workload": "SRtest150 if=qf c=%s poollim=1000; %d independent batches in flight per GPU (sq_fold_concurrent, "
                               "one stream set each), each holding the 219-record set %d time(s); a step folds all of them; the "
                               "batches run free inside the timed region (every batch is folded `steps` times back to back, "
                               "no barrier between the steps: sq_fold_concurrent_n)" % (args.config, K, R),
                   "batches_in_flight": K, "sets_per_batch": R, "host_cpus": effective_cpus(),
                   "host_threads_per_batch": host_threads(K, local_world), "seqs_per_gpu_per_step": per_step,
                   "paramsets": names, "evals_R_per_step": int(evals) * K * R, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
        "single_batch": {"ms_per_fold": round(lat[len(lat) // 2], 3), "best_ms": round(lat[0], 3),
                         "seq_per_s": round(len(prepared) / lat[len(lat) // 2] * 1e3, 1),
                         "how": "ONE 219-record batch alone (a fresh batch, nothing else in flight), median / best of 10 folds"},
        "single_batch_greedynobpp": greedy_alone,
        "host": {"cpu_ms_per_step": round(host_cpu / args.steps * 1e3, 1), "busy_cpus": round(host_cpu / dt, 1),
                 "cpu_quota": effective_cpus(),
                 "note": "rank 0's process CPU time inside the timed region (all threads): launches and waits of the fold threads; "
                         "RunAlgo's filters, the edge lists and the ranking tails run on the device"},
        "kernel_ms_per_fold": kernel_ms,
        "f1": {"mean_FS_consensus": round(fs_c, 4), "mean_FS_best_of_top5": round(fs_b, 4),
               "batches_in_flight_agree": bool(same)},
        "roofline": roof,
        "rooflines": rooflines,
        "pmc_note": pmc_note,
        "cpu_baseline": cpu,
        "vs_cpu_baseline": {"value": round(per_step * world * args.steps / dt / cpu["value"], 1),
                            "single_batch": round(len(prepared) / lat[len(lat) // 2] * 1e3 / cpu["value"], 1),
                            "vs_reference_form": round(per_step * world * args.steps / dt / cpu["reference_form"]["value"], 1) if cpu.get("reference_form") else None,
                            "note": "ratios to cpu_baseline.value (the C-port oracle on this host's %d CPUs) and to cpu_baseline.reference_form.value "
                                    "(the same oracle in the reference's algorithmic form: interpreted per-cell loops)" % cpu["cores"]} if cpu else None,
        "one_pass": one_pass,
        "stream": stream,
        "end_to_end": end_to_end,
        "shape_nobpp": shape,
        "pools_long": dict(pools_long, vs_cpu_baseline=round(pools_long["seq_per_s"] / cpu["other_workloads"]["pools_long"]["value"], 1))
                      if pools_long and cpu and "seq_per_s" in pools_long and "pools_long" in cpu.get("other_workloads", {}) else pools_long,
        "alignment": dict(alignment, vs_cpu_baseline=dict(
                              step1_only=round(cpu["other_workloads"]["A5000"]["value"] / alignment["step1_only"]["seconds"], 1),
                              note="cpu_baseline.other_workloads.A5000 (the oracle's step 1 on this host's cores, extrapolated from a few rows) / step1_only.seconds"))
                     if alignment and cpu and "step1_only" in alignment and "A5000" in cpu.get("other_workloads", {}) else alignment,
        "strong_scaling_proxy": proxy,
        "sharded": sharded,
        "n_ranks_seen": world if world == 1 else nranks_seen,
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
