"""cProfile of Predict() on SRtest150 (c=nobpp) and on S300 x N (c=fastest, pl=1).  usage: predict_profile.py [N]"""
import cProfile, io, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from squarna_amd import Predict
path = os.path.join(ROOT, "squarna_amd", "data", "datasets", "SRtest150.fas")
for _ in range(3):
    Predict(inputfile=path, inputformat="qf", configfile="nobpp", write_to=io.StringIO())
ts = []
for _ in range(5):
    t0 = time.perf_counter(); Predict(inputfile=path, inputformat="qf", configfile="nobpp", write_to=io.StringIO()); ts.append((time.perf_counter() - t0) * 1e3)
print("SRtest150 Predict ms:", " ".join("%.2f" % t for t in ts))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    Predict(inputfile=path, inputformat="qf", configfile="nobpp", write_to=io.StringIO())
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
