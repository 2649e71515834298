python tools/srtest_probe.py nobpp 6 2>&1 | tail -4
python - <<'PY'
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.getcwd())
from squarna_amd import Predict
kw = dict(inputfile="datasets/SRtest150.fas", inputformat="qf", configfile="nobpp")
for _ in range(3):
    Predict(write_to=io.StringIO(), **kw)
pr = cProfile.Profile(); pr.enable()
Predict(write_to=io.StringIO(), **kw)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
PY
