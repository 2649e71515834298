"""The stream leg's step split into its phases: K x Batch() / fold of the K batches / pack / close.  usage: stream_phases.py [K=8] [R=12] [STEPS=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared, fold_concurrently
from squarna_amd.inputs import ParseDefaultInput
K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
names, psets = ParseConfig(builtin_config("nobpp"))
recs = load_srtest150() + list(ParseDefaultInput(os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtrain150.fas"), "qf"))
allp = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
streams = [torch.cuda.Stream() for _ in range(K)]
for t in range(steps):
    t0 = time.perf_counter()
    batches = []
    for q in range(K):
        start = ((t * K + q) * 97) % len(allp)
        sel = [allp[(start + i) % len(allp)] for i in range(219 * R)]
        with torch.cuda.stream(streams[q]):
            batches.append(Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096 * R))
    t1 = time.perf_counter()
    fold_concurrently(batches, poollim=1000)
    t2 = time.perf_counter()
    n = sum(int(b.pack_all()[1][-1]) for b in batches)
    t3 = time.perf_counter()
    for b in batches: b.close()
    t4 = time.perf_counter()
    print("step %d: build %.1f ms  fold %.1f ms  pack %.1f ms  close %.1f ms" % (t, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
