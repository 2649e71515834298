cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in "1024 1000" "128 1000"; do
rm -rf gpurun_out/tr; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 tools/s1000_probe.py $a 3 --noprof > gpurun_out/tr.log 2>&1
echo "== $a"; tail -2 gpurun_out/tr.log
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/tr/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last fold: find last sq_rounds_kernel and print kernels around it
idx=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("sq_rounds_kernel")][-1]
# find start: the sq_fold_begin_kernel before it
j=idx
while j>0 and not rows[j]["Kernel_Name"].startswith("sq_fold_begin"): j-=1
j=max(0,j-3)
t0=int(rows[j]["Start_Timestamp"])
for r in rows[j:]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print("%8.1f %8.1f  %7.1f us  %s" % (s,e,e-s,r["Kernel_Name"][:50]))
PY
done
