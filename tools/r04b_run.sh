python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
recs = bench.load_srtest150()
names, psets = ParseConfig(builtin_config("nobpp"))
prepared = [Prepared(seq, reacts, restr, ref) for _, seq, reacts, restr, ref in recs]
with Batch(prepared, [psets] * len(prepared), fp32=False) as b0:
    b0.fold(poollim=1000)
    b0.profile(True); b0.profile_reset()
    for _ in range(3): b0.fold(poollim=1000)
    torch.cuda.synchronize()
    print({nm: round(b0.profile_get(k)[0] / 3, 3) for k, nm in enumerate(["bits", "state", "scan", "score_select", "edmonds", "hungarian", "nussinov"])})
PY
for k in 1 8; do python tools/concurrent_probe.py $k 20 2>&1 | grep -E "^K=" | cut -c1-100; done
