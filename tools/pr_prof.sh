# the pool round kernel's phases per structure (one batch of 12 SRtest150 sets): bash tools/pr_prof.sh  (on the GPU box)
cd $GRAFT_REPO_ROOT
o=gpurun_out/s2; mkdir -p $o
cp squarna_amd/libsquarna_hip.so /tmp/lib_keep.so
SQ_DEFS=-DSQ_PR_PROF python -c "from squarna_amd.build import build_library; build_library(force=True)"
python tools/pool_probe.py 12 2 2>&1 | grep "^pool round" > $o/pr_prof_all.txt
cp /tmp/lib_keep.so squarna_amd/libsquarna_hip.so
python - <<'PY'
import re, collections
rows = [l for l in open("gpurun_out/s2/pr_prof_all.txt")]
keys = ["bps", "entry", "ext", "extend", "setup", "state", "scan", "score", "choose", "total"]
acc = collections.Counter(); n = 0
for l in rows:
    m = {k: float(v) for k, v in re.findall(r"(bps|entry|ext|extend|setup|state|scan|score|choose|total) ([0-9.]+)", l)}
    if len(m) == 10:
        for k in keys: acc[k] += m[k]
        n += 1
print("%d structure-rounds sampled; mean us:" % n, " ".join("%s %.1f" % (k, acc[k] / max(n, 1)) for k in keys))
PY
head -5 $o/pr_prof_all.txt
