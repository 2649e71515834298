#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs written by tools/pmc_scan.sh (per kernel, first and total)."""
import csv, glob, sys, collections
out = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "sq_scan6_kernel"
for f in sorted(glob.glob(out + "/p*/*/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    by = collections.OrderedDict()
    for r in rows:
        if r["Kernel_Name"].startswith(kern):
            by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
            by[r["Dispatch_Id"]]["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = list(by)
    if not ids:
        continue
    tot = collections.Counter()
    for v in by.values():
        for c, x in v.items():
            tot[c] += x
    print(f.split("/")[-3], "first:", {k: round(v, 1) for k, v in by[ids[0]].items()})
    print(f.split("/")[-3], "total:", {k: round(v, 1) for k, v in tot.items()}, "dispatches", len(ids))
