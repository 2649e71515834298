#!/usr/bin/env python3
"""Every kernel of the LAST fold in a rocprofv3 kernel trace (one batch): start, duration, gap to the previous end on the same queue.
usage: trace_all.py TRACE_DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k0 = [k for k, r in enumerate(rows) if r["Kernel_Name"].startswith(("sq_fold_begin", "sq_bits_masks"))]
k0 = [k for k in k0 if rows[k]["Kernel_Name"].startswith("sq_fold_begin")][-1] if any(rows[k]["Kernel_Name"].startswith("sq_fold_begin") for k in k0) else k0[-1]
t0 = int(rows[k0]["Start_Timestamp"])
for r in rows[k0:]:
    nm = r["Kernel_Name"].split("(")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-26s start %8.1f  end %8.1f  dur %7.1f us  wg %6d x %4s  queue %s" % (nm, s, e, e - s, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Workgroup_Size_X"], r.get("Queue_Id")))
