"""Where one pass over SRtest150 (Batch() + fold + pack_all) spends its time.  usage: onepass_probe.py [profile]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared

recs = load_srtest150()
names, psets = ParseConfig(builtin_config("nobpp"))
t0 = time.perf_counter()
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
print("Prepared x %d: %.2f ms" % (len(prepared), (time.perf_counter() - t0) * 1e3))
for rep in range(6):
    t0 = time.perf_counter()
    b = Batch(prepared, [psets] * len(prepared), fp32=False)
    t1 = time.perf_counter()
    b.fold(poollim=1000)
    t2 = time.perf_counter()
    buf, off = b.pack_all()
    t3 = time.perf_counter()
    b.close()
    t4 = time.perf_counter()
    print("Batch %.2f  fold %.2f  pack %.2f  close %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
if len(sys.argv) > 1:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        b = Batch(prepared, [psets] * len(prepared), fp32=False)
        b.fold(poollim=1000)
        b.pack_all()
        b.close()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
