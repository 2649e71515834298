"""One big SRtest150 batch (R copies of the 219-record set, c=nobpp, poollim 1000): fold wall time and per-kernel time
(HIP events inside the library).  usage: big_batch_probe.py R"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import load_srtest150
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24
recs = load_srtest150()
names, psets = ParseConfig(builtin_config(os.environ.get("PROBE_CONFIG", "nobpp")))
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs] * R
b = Batch(prepared, [psets] * len(prepared), fp32=False, max_structs=4096 * R)
for _ in range(3):
    b.fold(poollim=1000)
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); b.fold(poollim=1000); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("R=%d: %d records, fold ms %s -> %.0f seq/s" % (R, len(prepared), " ".join("%.2f" % t for t in ts), len(prepared) / min(ts) * 1e3))
b.profile(True); b.profile_reset()
b.fold(poollim=1000)
torch.cuda.synchronize()
print({nm: round(b.profile_get(k)[0], 3) for k, nm in enumerate(["bits", "state", "scan", "score_select", "edmonds", "hungarian", "nussinov"])})
print(b.mwm_counters())
