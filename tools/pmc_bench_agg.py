#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc pass over bench.py by kernel name: wave-cycles held, share waiting, instructions.
usage: pmc_bench_agg.py DIR"""
import csv, glob, sys, collections
tot = collections.defaultdict(collections.Counter)
nd = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"].split("(")[0]
        tot[nm][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); nd[nm] += 1
            tot[nm]["_dur_us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
allw = sum(v["SQ_WAVE_CYCLES"] for v in tot.values()) or 1.0
print("%-28s %7s %12s %7s %7s %9s %9s %9s %9s" % ("kernel", "calls", "wave-Mcyc", "share", "wait", "VALU-M", "SALU-M", "LDS-M", "dur-ms"))
for nm, v in sorted(tot.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:24]:
    wc = v["SQ_WAVE_CYCLES"]
    print("%-28s %7d %12.1f %7.3f %7.2f %9.1f %9.1f %9.1f %9.1f" % (nm, nd[nm], wc / 1e6, wc / allw, v["SQ_WAIT_ANY"] / wc if wc else 0,
          v["SQ_INSTS_VALU"] / 1e6, v["SQ_INSTS_SALU"] / 1e6, v["SQ_INSTS_LDS"] / 1e6, v["_dur_us"] / 1e3))
