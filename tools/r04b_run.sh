python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for per in 4 8 16 32 64; do
SQ_FILL_PER=$per python - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
r = bench.fill_leg(); print(os.environ["SQ_FILL_PER"], r["achieved"], r["frac"], r["ms_per_fill"])
PY
done
