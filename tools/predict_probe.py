#!/usr/bin/env python3
"""Predict() end to end on a synthetic FASTA-like file (parse + prepare + upload + fold + format + print to a buffer).
usage: predict_probe.py S300|S1000 [REPS] [CONFIG [POOLLIM]]   (default: fastest, poollim 1)"""
import io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from squarna_amd import Predict
wl = sys.argv[1] if len(sys.argv) > 1 else "S300"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
conf = sys.argv[3] if len(sys.argv) > 3 else "fastest"
kw = {"poollim": int(sys.argv[4])} if len(sys.argv) > 4 else ({"poollim": 1} if len(sys.argv) <= 3 else {})
items = bench.synthetic(wl)
if os.environ.get("PROBE_N"):
    items = items[:int(os.environ["PROBE_N"])]
with tempfile.NamedTemporaryFile("w", suffix=".fas", delete=False) as f:
    for k, (s, line) in enumerate(items):
        f.write(">s%d\n%s\n" % (k, s))
    path = f.name
for r in range(reps):
    buf = io.StringIO()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Predict(inputfile=path, inputformat="q", configfile=conf, write_to=buf, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s Predict: %.1f ms -> %.0f seq/s (%d chars)" % (wl, dt * 1e3, len(items) / dt, len(buf.getvalue())), flush=True)
if os.environ.get("PROBE_SHA"):
    import hashlib
    print("sha256 of the output:", hashlib.sha256(buf.getvalue().encode()).hexdigest(), flush=True)
os.unlink(path)
