#!/usr/bin/env python3
"""Attributes the samples of tools/prof/libsampler.so to functions: self time by the innermost frame, and inclusive
time by the first frame inside libsquarna_hip.so.  usage: symbolize.py sampler.out [path/to/libsquarna_hip.so]"""
import bisect, collections, os, subprocess, sys
path = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "..", "squarna_amd", "libsquarna_hip.so")
maps, samples = [], []
for line in open(path):
    if line.startswith("map "):
        f = line.split()
        lo, hi = (int(x, 16) for x in f[1].split("-"))
        off = int(f[3], 16)
        name = f[6] if len(f) > 6 else "?"
        maps.append((lo, hi, off, name))
    elif line.startswith("s "):
        samples.append([int(x, 16) for x in line.split()[1:]])
maps.sort()
los = [m[0] for m in maps]
def where(a):
    k = bisect.bisect_right(los, a) - 1
    if k >= 0 and maps[k][0] <= a < maps[k][1]:
        return maps[k][3], a - maps[k][0] + maps[k][2]
    return "?", a
# symbols of the library
syms = []
out = subprocess.run(["nm", "-C", "--defined-only", "-n", lib], capture_output=True, text=True).stdout
for l in out.splitlines():
    p = l.split(None, 2)
    if len(p) == 3 and p[1] in "tTwW":
        syms.append((int(p[0], 16), p[2]))
saddr = [s[0] for s in syms]
_symcache = {}
def sym_prepare(offs):
    """llvm-symbolizer: with a -gline-tables-only build the innermost INLINED function and its line are named"""
    tool = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
    todo = sorted(set(offs))
    if not todo or not os.path.exists(tool):
        return
    inp = "\n".join(hex(o) for o in todo) + "\n"
    out = subprocess.run([tool, "--obj=" + lib, "-C", "-f", "-i"], input=inp, capture_output=True, text=True).stdout
    blocks = [b for b in out.split("\n\n") if b.strip()]
    for o, blk in zip(todo, blocks):
        ls = blk.strip().splitlines()
        if len(ls) >= 2 and ls[0] != "??":
            inner = ls[0][:70]; outer = ls[-2][:70] if len(ls) >= 4 else ""
            loc = os.path.basename(ls[1]).rsplit(":", 1)[0]
            _symcache[o] = "%s @%s%s" % (inner, loc, ("  <in " + outer + ">") if outer and outer != inner else "")
def sym(off):
    if off in _symcache:
        return _symcache[off]
    k = bisect.bisect_right(saddr, off) - 1
    return syms[k][1][:110] if k >= 0 else hex(off)
_other = {}
def sym_other(mod, off):
    """dynamic symbols of another module (libc, libstdc++, libamdhip64 ...)"""
    if mod not in _other:
        tab = []
        if os.path.exists(mod):
            o = subprocess.run(["nm", "-D", "-C", "--defined-only", "-n", mod], capture_output=True, text=True).stdout
            for l in o.splitlines():
                q = l.split(None, 2)
                if len(q) == 3 and q[1] in "tTwWiI":
                    tab.append((int(q[0], 16), q[2]))
        _other[mod] = (tab, [t[0] for t in tab])
    tab, ad = _other[mod]
    k = bisect.bisect_right(ad, off) - 1
    return "[%s] %s" % (os.path.basename(mod), tab[k][1][:80] if k >= 0 else hex(off))
self_t, incl_t, mods = collections.Counter(), collections.Counter(), collections.Counter()
sym_prepare([where(a)[1] for s in samples for a in s if "libsquarna_hip" in where(a)[0]])
for s in samples:
    if not s:
        continue
    m, off = where(s[0])
    base = os.path.basename(m)
    mods[base] += 1
    self_t[sym(off) if "libsquarna_hip" in m else sym_other(m, off)] += 1
    seen = set()
    for a in s:
        m2, off2 = where(a)
        if "libsquarna_hip" in m2:
            f = sym(off2)
            if f not in seen:
                incl_t[f] += 1; seen.add(f)
n = len(samples)
print("%d samples   (%s)" % (n, open(path).readline().strip()))
site = collections.Counter()
for smp in samples:
    for a in smp:
        m2, off2 = where(a)
        if "libsquarna_hip" in m2:
            site[sym(off2)] += 1
            break
print("-- first frame inside libsquarna_hip.so (call site of the sampled event)")
for k, v in site.most_common(45): print("  %6.2f%%  %s" % (100.0 * v / n, k))
print("-- by module (innermost frame)")
for k, v in mods.most_common(12): print("  %6.2f%%  %s" % (100.0 * v / n, k))
print("-- self time")
for k, v in self_t.most_common(40): print("  %6.2f%%  %s" % (100.0 * v / n, k))
print("-- inclusive (functions of libsquarna_hip.so anywhere on the stack)")
for k, v in incl_t.most_common(45): print("  %6.2f%%  %s" % (100.0 * v / n, k))

# call paths at function granularity (symbol table only: robust against inlining)
def fsym(off):
    k = bisect.bisect_right(saddr, off) - 1
    return syms[k][1].split("(")[0][:60] if k >= 0 else hex(off)
paths = collections.Counter()
for smp in samples:
    names = []
    for a in smp:
        m2, off2 = where(a)
        if "libsquarna_hip" in m2:
            f = fsym(off2)
            if not names or names[-1] != f:
                names.append(f)
    paths[" < ".join(names[:5])] += 1
print("-- call paths inside libsquarna_hip.so (innermost first)")
for k, v in paths.most_common(40): print("  %6.2f%%  %s" % (100.0 * v / n, k))
