"""Where does a slow Batch.close() of the pipelined stream go?  The bench's stream leg replayed with close() split into
sq_batch_destroy and the release of the workspace tensor.  usage: close_probe.py"""
import os, sys, time, threading, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared, fold_concurrently
from squarna_amd.inputs import ParseDefaultInput
K, R = 8, 12
names, psets = ParseConfig(builtin_config("nobpp"))
recs = load_srtest150() + list(ParseDefaultInput(os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtrain150.fas"), "qf"))
allp = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
streams = [[torch.cuda.Stream() for _ in range(K)] for _ in range(2)]
gc.collect(); gc.freeze(); gc.disable()
def build(t, box):
    out = []
    for q in range(K):
        start = ((t * K + q) * 97) % len(allp)
        sel = [allp[(start + i) % len(allp)] for i in range(219 * R)]
        with torch.cuda.stream(streams[t & 1][q]):
            out.append(Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096 * R))
    box["b"] = out
def run(nsteps, base):
    box = {}; build(base, box); nxt = box["b"]
    for t in range(nsteps):
        t0 = time.perf_counter()
        cur, box = nxt, {}
        th = threading.Thread(target=build, args=(base + t + 1, box)); th.start()
        fold_concurrently(cur, poollim=1000)
        n = sum(int(b.pack_all()[1][-1]) for b in cur)
        t1 = time.perf_counter()
        td = tw = 0.0
        for b in cur:
            a = time.perf_counter(); b.L.sq_batch_destroy(b.h); b.h = None
            c = time.perf_counter(); b.workspace = None
            d = time.perf_counter(); td += c - a; tw += d - c
        t2 = time.perf_counter()
        th.join(); nxt = box["b"]
        print("base %d step %d: fold+pack %.1f ms  destroy %.1f  workspace release %.1f  join %.1f  mem reserved %.1f GB" % (
            base, t, (t1 - t0) * 1e3, td * 1e3, tw * 1e3, (time.perf_counter() - t2) * 1e3, torch.cuda.memory_reserved() / 2**30), flush=True)
    for b in nxt: b.close()
run(4, 1000)
run(10, 2000)
