"""Builds libsquarna_hip.so (gfx950 only) in-tree with hipcc.  No JIT cache: the built
.so travels with the tree (it is git-ignored, not gpurun-ignored)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsquarna_hip.so")
SOURCES = ["sq_kernels.hip", "sq_host.hip", "sq_batch.hip", "sq_round_host.hip", "sq_fold.hip", "sq_results.hip", "sq_tail.cpp", "sq_match.hip", "sq_algos.hip", "sq_graph.hip", "sq_chain.hip", "sq_pool.hip", "sq_gather.hip", "sq_tail_dev.hip", "sq_algos_dev.hip", "sq_text.cpp", "sq_parse.cpp", "sq_context.hip", "sq_rounds.hip", "sq_pool_round.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-x", "hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not os.path.isdir(os.path.join(CSRC, f))] + \
           [os.path.join(os.path.dirname(HERE), "include", "squarna_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the C-ABI shared library."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libsquarna_hip.so cannot be built")
    objs = []
    bdir = os.path.join(CSRC, "build")
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(bdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + os.environ.get("SQ_DEFS", "").split() + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
