#!/usr/bin/env python3
"""Where a sharded step spends its time: fold vs sq_result_pack_all, for 1..K concurrent sub-batches.
usage: shard_probe.py S300|S1000|S2000 [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, fold_concurrently

w = sys.argv[1] if len(sys.argv) > 1 else "S300"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
names, psets = ParseConfig(builtin_config("fastest"))
prepared = bench.prepare_synthetic(bench.synthetic(w))
dev = torch.device("cuda:0")
for nb in (1, 2, 4):
    cuts = [len(prepared) * q // nb for q in range(nb + 1)]
    batches = []
    for q in range(nb):
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            batches.append(Batch(prepared[cuts[q]:cuts[q + 1]], [psets] * (cuts[q + 1] - cuts[q]),
                                 max_structs=max(cuts[q + 1] - cuts[q], 1), fp32=False))
    torch.cuda.synchronize()
    tf, tp = [], []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        if nb == 1:
            batches[0].fold(poollim=1)
        else:
            fold_concurrently(batches, poollim=1)
        t1 = time.perf_counter()
        for b in batches:
            b.pack_all()
        t2 = time.perf_counter()
        if r:
            tf.append((t1 - t0) * 1e3); tp.append((t2 - t1) * 1e3)
    print("%s nb=%d fold %s | pack %s" % (w, nb, " ".join("%.2f" % x for x in tf), " ".join("%.2f" % x for x in tp)))
    for b in batches:
        b.close()
