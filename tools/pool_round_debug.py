"""Debug: folds fuzz records with the pool round kernel and with the launched kernels, printing the pool sizes per round."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SQ_TIMING"] = "1"; os.environ["SQ_POOL_DEBUG"] = "1"
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import Batch, Prepared
from tests.test_hip_parity2 import _chain_records
cfg, count, nmin, nmax, pl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
names, psets = ParseConfig(builtin_config(cfg))
raw = _chain_records(count, 900 + pl, nmin, nmax)
prepared = [Prepared(s, r, x) for s, r, x in raw]
for mode in ("SQ_POOL_ROUND_ALWAYS", "SQ_NO_POOL_ROUND"):
    os.environ[mode] = "1"
    print("=====", mode, flush=True)
    with Batch(prepared, [psets] * count, max_structs=8192, fp32=False) as b:
        b.fold(poollim=pl)
        print("driver", b.fold_driver, flush=True)
    del os.environ[mode]
