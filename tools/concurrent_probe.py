#!/usr/bin/env python3
"""K SRtest150 batches (c=nobpp) in flight at once through sq_fold_concurrent, each on its own stream:
ms per step (one step = K folds) and sequences/s.  usage: concurrent_probe.py K [REPS] [--same-stream] [--free]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared, fold_concurrently

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
same = "--same-stream" in sys.argv
recs = load_srtest150()
names, psets = ParseConfig(builtin_config(os.environ.get("PROBE_CONFIG", "nobpp")))
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
prepared = prepared * int(os.environ.get("PROBE_REPLICAS", "1"))      # the set several times over in ONE batch
batches, streams = [], []
for k in range(K):
    st = torch.cuda.current_stream() if same else torch.cuda.Stream()
    streams.append(st)
    with torch.cuda.stream(st):
        batches.append(Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=int(os.environ.get("PROBE_MAX_STRUCTS", "4096"))))
torch.cuda.synchronize()
walls = []
cpu0 = None
for r in range(reps + 2):
    if r == 2:
        cpu0 = (time.process_time(), time.perf_counter())
        if os.environ.get("SAMPLER_MANUAL"):            # tools/prof/libsampler.so preloaded: sample the steady state only
            import ctypes
            ctypes.CDLL(os.environ["LD_PRELOAD"].split(":")[0]).sampler_start()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if K == 1:
        batches[0].fold(poollim=1000)
    else:
        fold_concurrently(batches, poollim=1000)
    torch.cuda.synchronize()
    walls.append((time.perf_counter() - t0) * 1e3)
walls = walls[2:]
best, med = min(walls), sorted(walls)[len(walls) // 2]
print("K=%d: step ms min %.2f median %.2f -> %.0f seq/s (median)  all: %s" % (
    K, best, med, K * len(prepared) / med * 1e3, " ".join("%.1f" % w for w in walls)), flush=True)
if "--free" in sys.argv and K > 1:
    # the same work without a barrier between the steps: every batch folded `reps` times back to back (sq_fold_concurrent_n)
    torch.cuda.synchronize(); c0 = time.process_time(); t0 = time.perf_counter()
    fold_concurrently(batches, reps=reps, poollim=1000)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("K=%d free-running: %.2f ms per step-equivalent -> %.0f seq/s; CPU %.1f busy" % (
        K, dt / reps * 1e3, K * len(prepared) * reps / dt, (time.process_time() - c0) / dt), flush=True)
cpu = time.process_time() - cpu0[0]; wall = time.perf_counter() - cpu0[1]
print("      CPU: %.1f ms of process CPU time per step = %.1f busy CPUs on average" % (cpu / reps * 1e3, cpu / wall), flush=True)
for b in batches:
    b.close()
