#!/bin/bash
# first-round duration of sq_score_kernel under compile-time variants (e.g. "-DSQ_SCORE_WAVES=4", "-DSQ_SCORE_CHUNK=2"):
# bash tools/score_ablate.sh "DEFS1" "DEFS2" ...   (rocprofv3 kernel trace of tools/s1000_probe.py per variant)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for defs in "$@"; do
  SQ_DEFS="$defs" timeout 300 python squarna_amd/build.py > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; continue; }
  rm -rf /tmp/abl; (cd /tmp && timeout 120 rocprofv3 --kernel-trace -d /tmp/abl -o a -- python3 $GRAFT_REPO_ROOT/tools/s1000_probe.py ${NSEQ:-1024} ${NLEN:-1000} 1 > /tmp/abl.log 2>&1)
  python3 - "$defs" <<'PY'
import sqlite3, sys, glob
f = glob.glob('/tmp/abl/**/*.db', recursive=True)
if not f: print(sys.argv[1], "no db"); sys.exit()
cur = sqlite3.connect(f[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(cur.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
sc = [r[1] / 1e3 for r in rows if r[0].startswith('sq_score_kernel')]
print("%-40s score launches %d first %.1f us second %.1f us total %.1f us" % (sys.argv[1] or "(baseline)", len(sc), sc[0], sc[1] if len(sc) > 1 else 0, sum(sc)))
PY
done
