"""One batch of R copies of SRtest150 (c=nobpp, poollim=1000) folded alone a few times; prints fold times.
usage: pool_probe.py [R] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
names, psets = ParseConfig(builtin_config("nobpp"))
recs = load_srtest150()
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs] * R
with Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=4096 * R) as b:
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); b.fold(poollim=1000); torch.cuda.synchronize()
        print("fold %.2f ms driver %d paths %d" % ((time.perf_counter() - t0) * 1e3, b.fold_driver, b.fold_paths), flush=True)
