"""Where Batch() spends its time on S300 x 10,000: Python array building vs sq_batch_create.  usage: batch_create_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
import bench
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd import engine as E
names, psets = ParseConfig(builtin_config("fastest"))
items = bench.synthetic("S300")
prepared = [E.Prepared(s, None) for s, _ in items]
L = E._lib.load()
orig = L.sq_batch_create
acc = [0.0]
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); acc[0] += time.perf_counter() - t0; return r
L.sq_batch_create = Wrap()
for rep in range(4):
    acc[0] = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = E.Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=len(prepared))
    t1 = time.perf_counter()
    b.close()
    print("Batch() %.2f ms of which sq_batch_create %.2f ms; close %.2f ms" % ((t1 - t0) * 1e3, acc[0] * 1e3, (time.perf_counter() - t1) * 1e3))
