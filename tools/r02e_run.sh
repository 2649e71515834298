cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02e_trace_c6 -- python3 tools/concurrent_probe.py 1 3 > gpurun_out/r02e_c6.log 2>&1
SQ_MWM_CLASSES=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r02e_trace_c1 -- python3 tools/concurrent_probe.py 1 3 > gpurun_out/r02e_c1.log 2>&1
python3 - <<'PY'
import csv, glob
for tag in ("c6", "c1"):
    f = glob.glob("gpurun_out/r02e_trace_%s/*/*kernel_trace.csv" % tag)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last fold: find the last sq_bits_masks_kernel (start of a fold)
    starts = [k for k, r in enumerate(rows) if r["Kernel_Name"].startswith("sq_bits_masks")]
    k0 = starts[-1]
    t0 = int(rows[k0]["Start_Timestamp"])
    print("==", tag)
    for r in rows[k0:]:
        nm = r["Kernel_Name"].split("(")[0]
        if nm.startswith(("sq_mwm", "sq_lsap", "sq_nussinov", "sq_flag", "sq_bits")) or True:
            s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
            if nm.startswith(("sq_mwm", "sq_lsap", "sq_nussinov", "sq_flag")) or e - s > 150:
                print("%-22s start %8.1f us  dur %8.1f us  grid %s lds %s queue %s" % (nm, s, e - s, r.get("Grid_Size_X", r.get("Grid_Size")), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")), r.get("Queue_Id")))
PY
