"""Times one fold of a BASELINE-shaped batch (c=fastest pl=1) with the persistent round kernel and with the launched rounds.
usage: rounds_probe.py N COUNT [reacts] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
from tests.test_hip_parity import _synthetic
n, count = int(sys.argv[1]), int(sys.argv[2])
reacts = len(sys.argv) > 3 and sys.argv[3] == "1"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
names, psets = ParseConfig(builtin_config("fastest"))
data = _synthetic(n, count, n, reacts)
prepared = [Prepared(s, rc) for s, rc in data]
packs = {}
for mode in (("rounds",) if os.environ.get("SQ_NO_LAUNCHED") else ("rounds", "launched")):
    if mode == "launched":
        os.environ["SQ_NO_ROUNDS"] = "1"
    else:
        os.environ.pop("SQ_NO_ROUNDS", None)
    with Batch(prepared, [psets] * count, max_structs=count, fp32=False) as b:
        b.fold(poollim=1)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); b.fold(poollim=1); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        buf, off = b.pack_all()
        packs[mode] = bytes(buf[:off[-1]])
        ev = sum(b.evals(k) for k in range(count))
        print("%s N=%d x %d: fold ms %s  (min %.3f) evals %d paths %d" % (mode, n, count, " ".join("%.3f" % t for t in ts), min(ts), ev, b.fold_paths), flush=True)
print("identical:", packs["rounds"] == packs.get("launched", packs["rounds"]))
