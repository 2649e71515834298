#!/usr/bin/env python3
"""Times Predict() on SRtest150 (if=qf) for a config, a few repetitions; SQ_TIMING=1 shows the phases."""
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from squarna_amd import Predict

conf = sys.argv[1] if len(sys.argv) > 1 else "nobpp"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
path = os.path.join(os.path.dirname(__file__), "..", "squarna_amd", "data", "datasets", "SRtest150.fas")
for r in range(reps):
    buf = io.StringIO()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Predict(inputfile=path, inputformat="qf", configfile=conf, write_to=buf)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.1f ms  (%d chars)" % (conf, dt * 1e3, len(buf.getvalue())), flush=True)
