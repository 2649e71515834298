import cProfile, pstats, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine
names, psets = ParseConfig(builtin_config("fastest"))
items = bench.synthetic("S300")
recs = [(s, None, None, None, psets, None) for s, line in items]
HipEngine().fold_records(recs, poollim=1)
pr = cProfile.Profile(); pr.enable()
HipEngine().fold_records(recs, poollim=1)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4000])
