#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace of the LAST step of concurrent_probe.py: per kernel name count / total / mean,
the union of busy time, and the step's extent.  usage: trace_summary.py TRACE_DIR"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
starts = [k for k, r in enumerate(rows) if r["Kernel_Name"].startswith("sq_bits_masks")]
k0 = starts[-K]                         # the last step = the last K folds
sel = rows[k0:]
t0 = int(sel[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in sel)
by = collections.OrderedDict()
for r in sel:
    nm = r["Kernel_Name"].split("(")[0]
    d = by.setdefault(nm, [0, 0.0])
    d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
short = [(s, e) for (s, e), r in zip(sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel if not r["Kernel_Name"].startswith(("sq_mwm", "sq_lsap", "sq_nussinov"))), sel)]
b2, cs, ce = 0, short[0][0], short[0][1]
for s, e in short[1:]:
    if s > ce: b2 += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
b2 += ce - cs
print("step extent %.2f ms, union of kernel time %.2f ms, union without mwm/lsap/nussinov %.2f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, b2 / 1e6, len(sel)))
for nm, (c, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("  %-24s %5d launches  total %9.1f us  mean %8.1f us" % (nm, c, us, us / c))
