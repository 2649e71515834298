#!/usr/bin/env python3
"""`one_pass` of bench.py on its own (ONE pass over 219 records, a different window every call: Batch() + sq_fold + pack_all,
nothing else in flight), with the three parts timed -- runs against any tree that holds a squarna_amd package (the bisect
of round 6: python tools/one_pass_probe.py [TREE] [STEPS])."""
import os, sys, time
tree = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
sys.path.insert(0, tree)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
from squarna_amd.inputs import ParseDefaultInput
import gc
names, psets = ParseConfig(builtin_config("nobpp"))
data = os.path.join(tree, "squarna_amd", "data", "datasets")
recs = list(ParseDefaultInput(os.path.join(data, "SRtest150.fas"), "qf")) + list(ParseDefaultInput(os.path.join(data, "SRtrain150.fas"), "qf"))
allp = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
gc.collect(); gc.freeze(); gc.disable()
rows = []
for t in range(3 + steps):
    start = (t * 97) % len(allp)
    sel = [allp[(start + i) % len(allp)] for i in range(219)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096)
    t1 = time.perf_counter()
    b.fold(poollim=1000)
    t2 = time.perf_counter()
    n = int(b.pack_all()[1][-1])
    t3 = time.perf_counter()
    b.close()
    t4 = time.perf_counter()
    if t >= 3:
        rows.append(((t4 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
rows.sort()
med = rows[len(rows) // 2]
col = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
print("one_pass %s: median %.3f ms best %.3f | medians: Batch() %.3f fold %.3f pack %.3f close %.3f" % (
    os.path.basename(tree), med[0], rows[0][0], col(1), col(2), col(3), col(4)), flush=True)
