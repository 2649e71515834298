#!/usr/bin/env python3
"""Per-queue timeline of a rocprofv3 --kernel-trace run of the headline step: for every HSA queue the kernels it ran, the time
it was busy, idle between kernels, and the kernels that took most of it.  usage: crowd_timeline.py DIR"""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
print("kernels %d, span %.1f ms, columns: queue / stream" % (len(rows), (t1 - t0) / 1e6))
byq = collections.defaultdict(list)
for r in rows: byq[(r[3], r[4])].append(r)
print("%-14s %7s %9s %9s %9s  top kernels (ms)" % ("queue/stream", "kernels", "busy ms", "idle ms", "span ms"))
for q, v in sorted(byq.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1]))[:40]:
    busy = 0; end = v[0][0]
    for a, b, *_ in v:                      # union of intervals (kernels of one queue may overlap)
        if b > end: busy += b - max(a, end); end = b
    span = v[-1][1] - v[0][0]
    top = collections.Counter()
    for a, b, nm, *_ in v: top[nm] += b - a
    print("%-14s %7d %9.1f %9.1f %9.1f  %s" % ("%s/%s" % q, len(v), busy / 1e6, (span - busy) / 1e6, span / 1e6,
          " ".join("%s %.1f" % (k.replace("sq_", "").replace("_kernel", ""), t / 1e6) for k, t in top.most_common(4))))
