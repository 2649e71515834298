"""cProfile of the stream leg's per-step host work: Batch() of 219 x R SRtest150-like records + fold + pack.  usage: stream_profile.py [R=12]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
names, psets = ParseConfig(builtin_config("nobpp"))
allp = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in load_srtest150()]
def step(t):
    sel = [allp[(t * 97 + i) % len(allp)] for i in range(219 * R)]
    b = Batch(sel, [psets] * len(sel), fp32=False, max_structs=4096 * R)
    b.fold(poollim=1000); n = int(b.pack_all()[1][-1]); b.close(); return n
for t in range(3): step(t)
t0 = time.perf_counter(); step(5); print("one step ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable()
for t in range(4): step(10 + t)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
