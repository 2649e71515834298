import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
names, psets = ParseConfig(builtin_config("nobpp"))
recs = load_srtest150()
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs] * 12
# warm the caches
for _ in range(3):
    with Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=4096 * 12) as b:
        b.fold(poollim=1000); b.pack_all()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=4096 * 12)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ts = []
    for k in range(4):
        if k < 2: sys.stderr.write("---- fold %d of a fresh batch\n" % k)
        os.environ["SQ_TIMING"] = "1" if k < 2 and rep == 2 else ""
        if not os.environ["SQ_TIMING"]: del os.environ["SQ_TIMING"]
        torch.cuda.synchronize(); ta = time.perf_counter(); b.fold(poollim=1000); torch.cuda.synchronize(); ts.append((time.perf_counter() - ta) * 1e3)
    b.close()
    print("create %.2f ms; folds %s" % ((t1 - t0) * 1e3, " ".join("%.2f" % t for t in ts)), flush=True)
