#!/usr/bin/env python3
"""profiles/traffic.json from rocprofv3 --pmc passes (tools/pmc_kernel.sh / tools/pmc_algo.sh output directories):
per kernel the HBM bytes per launch (FETCH_SIZE is reported in KB and counts 64 B per 128-B request on gfx950: x 2,
MI355X_MICROARCH.md section HBM; WRITE_SIZE in KB, exact), the share of wave cycles spent waiting and the share of
LDS-busy cycles that are bank-conflict cycles, stamped with a hash of the kernel sources they were measured on
(bench.py withholds them when the sources have changed).
usage: make_traffic.py TAG KERNEL[:NAME_PREFIX]=PMCDIR [...]      (TAG names the committed summaries, e.g. r02)"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernels_hash

tag = sys.argv[1]
out = {"kernels_sha16": kernels_hash(), "round": tag,
       "note": "FETCH_SIZE x 2 (gfx950) and WRITE_SIZE from separate --pmc passes; per launch = total over the dispatches of "
               "the probe / their number"}
for arg in sys.argv[2:]:
    kern, d = arg.split("=")
    kern, _, prefix = kern.partition(":")                      # KEY:PREFIX=DIR -- kernels whose names start with PREFIX, filed under KEY
    prefix = prefix or kern
    tot, ndisp, dur = collections.Counter(), {}, {}
    for f in sorted(glob.glob(d + "/p*/*/*counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith(prefix):
                continue
            c = r["Counter_Name"]
            tot[c] += float(r["Counter_Value"])
            if (c, r["Dispatch_Id"]) not in seen:
                seen.add((c, r["Dispatch_Id"]))
                ndisp[c] = ndisp.get(c, 0) + 1
                dur[c] = dur.get(c, 0.0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not tot:
        print("no rows for", kern, "in", d)
        continue
    summary = "a5000" if kern.startswith("a5000_") else kern.replace("sq_", "").replace("_kernel", "")   # (the alignment's three kernels share one summary file)
    e = {"source": "profiles/%s_%s_pmc.txt (rocprofv3 --pmc, %s)" % (tag, summary, os.path.basename(d.rstrip("/")))}
    if "FETCH_SIZE" in tot:
        e["fetch_bytes_per_launch"] = round(tot["FETCH_SIZE"] * 1024 * 2 / ndisp["FETCH_SIZE"])
    if "WRITE_SIZE" in tot:
        e["write_bytes_per_launch"] = round(tot["WRITE_SIZE"] * 1024 / ndisp["WRITE_SIZE"])
    if tot.get("SQ_WAVE_CYCLES"):
        e["wait_share"] = round(tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], 3)
    if tot.get("SQ_ACTIVE_INST_LDS"):
        e["lds_conflict_share"] = round(tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_ACTIVE_INST_LDS"], 3)
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAVES"):
        if c in tot:
            e[c.lower() + "_per_launch"] = round(tot[c] / ndisp[c])
    c0 = next(iter(ndisp))
    e["launches_in_probe"] = ndisp[c0]
    e["avg_launch_us_under_pmc"] = round(dur[c0] / ndisp[c0], 1)
    out[kern] = e
with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
