#!/usr/bin/env python3
"""How full is the GPU during the timed steps of a rocprofv3 kernel trace?  For the folds number F0 .. F1 of the trace (a fold
starts with sq_bits_masks_kernel; bench.py --inflight K: step s is folds K s .. K s + K - 1): the
time-weighted distribution of (a) kernels in flight and (b) waves demanded by the kernels in flight (grid / 64, each
kernel capped at the 5,120 waves the chip holds at 5 waves per SIMD), plus per kernel name the share of the
wave-time demanded.  usage: trace_load.py TRACE_DIR F0 F1"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
F0, F1 = int(sys.argv[2]), int(sys.argv[3])
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("sq_bits_masks")]
ev = []
lo = starts[F0]; T1 = starts[F1] if F1 < len(starts) else max(int(r["End_Timestamp"]) for r in rows)
by = collections.Counter()
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= lo or s >= T1: continue
    s = max(s, lo); e = min(e, T1)
    g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    w = min(max(1, g // 64), 5120)
    ev.append((s, 1, w)); ev.append((e, -1, -w))
    by[r["Kernel_Name"].split("(")[0]] += w * (e - s)
ev.sort()
nk = nw = 0; last = lo
hk = collections.Counter(); hw = collections.Counter()
for t, dk, dw in ev:
    dt = t - last
    if dt > 0:
        hk[min(nk, 64) // 4 * 4] += dt
        b = 0 if nw == 0 else (1 if nw < 256 else 2 if nw < 1024 else 3 if nw < 2560 else 4 if nw < 5120 else 5)
        hw[b] += dt
    nk += dk; nw += dw; last = t
tot = T1 - lo
print("window %.1f ms" % (tot / 1e6))
print("kernels in flight (time share): " + "  ".join("%d-%d: %.2f" % (k, k + 3, v / tot) for k, v in sorted(hk.items())))
names = ["idle", "<256 waves", "256-1023", "1024-2559", "2560-5119", ">=5120"]
print("waves demanded  (time share): " + "  ".join("%s: %.2f" % (names[k], v / tot) for k, v in sorted(hw.items())))
tw = sum(by.values())
print("wave-time demanded by kernel: " + "  ".join("%s %.2f" % (k, v / tw) for k, v in by.most_common(10)))
print("mean waves demanded: %.0f" % (tw / tot))
