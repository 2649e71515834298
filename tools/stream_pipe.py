"""The pipelined stream leg step by step: the next step's Batch() calls on a second thread while this step folds.
usage: stream_pipe.py [K=8] [R=12] [STEPS=10]"""
import os, sys, time, threading, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared, fold_concurrently
from squarna_amd.inputs import ParseDefaultInput
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
names, psets = ParseConfig(builtin_config("nobpp"))
recs = load_srtest150() + list(ParseDefaultInput(os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtrain150.fas"), "qf"))
allp = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
streams = [[torch.cuda.Stream() for _ in range(K)] for _ in range(2)]
gc.collect(); gc.freeze()
if os.environ.get('PIPE_GC_OFF', '1') == '1': gc.disable()     # (PIPE_GC_OFF=0: the collector stays on -- a full collection every eighth step or so)
def build(t, box):
    t0 = time.perf_counter()
    out = []
    for q in range(K):
        start = ((t * K + q) * 97) % len(allp)
        sel = [allp[(start + i) % len(allp)] for i in range(219 * R)]
        with torch.cuda.stream(streams[t & 1][q]):
            out.append(Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096 * R))
    box["b"] = out; box["ms"] = (time.perf_counter() - t0) * 1e3
box = {}
build(0, box)
nxt = box["b"]
for t in range(steps):
    t0 = time.perf_counter()
    cur, box = nxt, {}
    th = threading.Thread(target=build, args=(t + 1, box))
    th.start()
    fold_concurrently(cur, poollim=1000)
    t1 = time.perf_counter()
    n = sum(int(b.pack_all()[1][-1]) for b in cur)
    t2 = time.perf_counter()
    for b in cur: b.close()
    t3 = time.perf_counter()
    th.join()
    t4 = time.perf_counter()
    nxt = box["b"]
    print("step %d: fold %.1f ms  pack %.1f  close %.1f  join-wait %.1f  (build thread %.1f ms)  total %.1f" % (t, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, box["ms"], (t4 - t0) * 1e3), flush=True)
for b in nxt: b.close()
