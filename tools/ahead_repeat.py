#!/usr/bin/env python3
"""One SRtest150 batch alone folded again and again with the rounds of the device pools enqueued ahead of the host: every fold's
packed records must equal the first one's (the host follows a ring of published headers: a wait that returns early shows up
here).  usage: ahead_repeat.py [REPS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in load_srtest150()]
for cfg in ("greedynobpp", "alt", "nobpp"):
    names, psets = ParseConfig(builtin_config(cfg))
    with Batch(prepared, [psets] * len(prepared), fp32=False) as b:
        b.fold(poollim=1000); buf, off = b.pack_all(); want = bytes(buf[:off[-1]]); bad = 0
        for r in range(reps):
            b.fold(poollim=1000); buf, off = b.pack_all(); bad += bytes(buf[:off[-1]]) != want
        print(cfg, "%d folds," % reps, bad, "differ, paths", b.fold_paths)
