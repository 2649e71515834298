#!/bin/bash
# the matching jobs' finish kernel: phase timers inside the headline step (-DSQ_FIN_PROF): bash tools/fin_prof_crowded.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
o=gpurun_out/s2; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_FIN_PROF" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
python bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline 2>&1 | grep "^finish algo" > $o/fin_prof_crowded.txt
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
python - <<'PY'
import re, collections
rows = [l for l in open("gpurun_out/s2/fin_prof_crowded.txt")]
keys = ["load", "pairs", "stems", "filter", "levels", "count", "log", "total"]
for algo in ("0", "1", "2", "3"):
    acc = collections.Counter(); n = 0
    for l in rows:
        if ("algo=%s " % algo) not in l: continue
        m = {k: float(v) for k, v in re.findall(r"(load|pairs|stems|filter|levels|count|log|total) ([0-9.]+)", l)}
        if len(m) == 8:
            for k in keys: acc[k] += m[k]
            n += 1
    if n: print("algo %s: %d jobs sampled; mean us:" % (algo, n), " ".join("%s %.1f" % (k, acc[k] / n) for k in keys))
PY
