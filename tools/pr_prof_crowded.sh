#!/bin/bash
# the pool round kernel's phase timers inside the HEADLINE step (8 batches x 12 sets in flight): bash tools/pr_prof_crowded.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
o=gpurun_out/s2; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS="-DSQ_PR_PROF ${PR_PROF_DEFS:-}" python -c "from squarna_amd.build import build_library; build_library(force=True)" > /dev/null 2>&1
python bench.py --steps 2 --warmup 1 --regions 1 --no-cpu --no-stream --no-roofline 2>&1 | grep "^pool round" > $o/pr_prof_crowded.txt
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
python - <<'PY'
import re, collections
rows = [l for l in open("gpurun_out/s2/pr_prof_crowded.txt")]
keys = ["bps", "entry", "ext", "extend", "setup", "state", "scan", "score", "choose", "total"]
for sel, name in [(lambda l: "nstrand=0 " in l, "round 0"), (lambda l: "nstrand=0 " not in l, "children")]:
    acc = collections.Counter(); n = 0
    for l in rows:
        if not sel(l): continue
        m = {k: float(v) for k, v in re.findall(r"(bps|entry|ext|extend|setup|state|scan|score|choose|total) ([0-9.]+)", l)}
        if len(m) == 10:
            for k in keys: acc[k] += m[k]
            n += 1
    print("%s: %d structure-rounds sampled; mean us:" % (name, n), " ".join("%s %.1f" % (k, acc[k] / max(n, 1)) for k in keys))
PY
