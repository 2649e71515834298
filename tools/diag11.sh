#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/diag11; mkdir -p $o
python -m pytest tests/test_hip_parity4.py -x -q -k "ahead or pool_round or overflow or optimistic" 2>&1 | tail -15 > $o/t1.txt
cat $o/t1.txt
python - > $o/greedy.txt 2>&1 <<'P'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
recs = load_srtest150()
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
for cfg in ("greedynobpp", "alt", "nobpp"):
    names, psets = ParseConfig(builtin_config(cfg))
    for env in ("0", "1", "2", "3", "4", "6"):
        os.environ["SQ_POOL_AHEAD"] = env
        with Batch(prepared, [psets] * len(prepared), fp32=False) as b:
            for _ in range(3): b.fold(poollim=1000)
            ts = []
            for _ in range(12):
                torch.cuda.synchronize(); t0 = time.perf_counter(); b.fold(poollim=1000); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            print("%s SQ_POOL_AHEAD=%s: median %.3f best %.3f ms  paths %d driver %d peak %d" % (cfg, env, ts[len(ts) // 2], ts[0], b.fold_paths, b.fold_driver, b.fold_peak_structs), flush=True)
P
cat $o/greedy.txt | grep -v amdgpu
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
