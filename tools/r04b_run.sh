python -m pytest tests -m gpu -x -q 2>&1 | grep -E " passed| failed|rror|assert" | tail -4
python tools/predict_probe.py S300 3 2>&1 | grep Predict
python tools/predict_probe.py S1000 3 2>&1 | grep Predict
cat > /tmp/pd.py <<'PY'
import io, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from squarna_amd import Predict
path = os.path.join("squarna_amd", "data", "datasets", "SRtest150.fas")
for r in range(4):
    buf = io.StringIO(); t0 = time.perf_counter()
    Predict(inputfile=path, inputformat="qf", configfile="nobpp", write_to=buf)
    print("SRtest150 Predict nobpp: %.1f ms (%d chars)" % ((time.perf_counter() - t0) * 1e3, len(buf.getvalue())))
PY
python /tmp/pd.py 2>&1 | grep Predict
