"""cProfile of Predict() on S300 x 10,000 (c=fastest, pl=1).  usage: predict_profile_s300.py"""
import cProfile, io, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from squarna_amd import Predict
items = bench.synthetic("S300")
with tempfile.NamedTemporaryFile("w", suffix=".fas", delete=False) as f:
    for k, (s, line) in enumerate(items):
        f.write(">s%d\n%s\n" % (k, s))
    path = f.name
for _ in range(2):
    Predict(inputfile=path, inputformat="q", configfile="fastest", poollim=1, write_to=io.StringIO())
t0 = time.perf_counter(); Predict(inputfile=path, inputformat="q", configfile="fastest", poollim=1, write_to=io.StringIO()); print("ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable()
Predict(inputfile=path, inputformat="q", configfile="fastest", poollim=1, write_to=io.StringIO())
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
os.unlink(path)
