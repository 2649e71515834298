#!/usr/bin/env python3
"""Timeline of the long kernels of the LAST fold in a rocprofv3 kernel trace (one batch): start, duration, grid, LDS, queue."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k0 = [k for k, r in enumerate(rows) if r["Kernel_Name"].startswith("sq_bits_masks")][-1]
t0 = int(rows[k0]["Start_Timestamp"])
for r in rows[k0:]:
    nm = r["Kernel_Name"].split("(")[0]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if nm.startswith(("sq_mwm", "sq_lsap", "sq_nussinov", "sq_flag")) or e - s > 100:
        print("%-20s start %8.1f us  dur %8.1f us  wg %6d  lds %6s  queue %s" % (nm, s, e - s, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r.get("LDS_Block_Size", "?"), r.get("Queue_Id")))
print("last kernel ends at %.1f us" % ((max(int(r["End_Timestamp"]) for r in rows[k0:]) - t0) / 1e3))
