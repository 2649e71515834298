import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import numpy as np, torch
from a5000_probe import make_msa
from squarna_amd.config import ParseConfig, builtin_config
from squarna_amd.engine import HipEngine, Prepared, Batch
from squarna_amd.dbn import gap_mask
rows = make_msa(512, 5000)
names, psets = ParseConfig(builtin_config("ali")); ps0 = psets[0]
ps = dict(bpweights=ps0["bpweights"], bpp=0, algorithms={"G"}, suboptmax=1.0, suboptmin=1.0, suboptsteps=1.0, minlen=ps0["minlen"], minbpscore=ps0["minbpscore"], minfinscorefactor=1.0, bracketweight=-2.0, distcoef=0.09, orderpenalty=1.0, loopbonus=0.125, maxstemnum=1e6)
eng = HipEngine()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    matrix = torch.zeros((5000, 5000), dtype=torch.float64, device="cuda")
    prepared, cols = [], []
    for seq in rows:
        p = Prepared(seq, None, "." * 5000, None); p.shortreacts = [0.5] * len(p.shortseq); p.plain_reacts = True
        prepared.append(p); cols.append(np.flatnonzero(~gap_mask(seq)).astype(np.int32))
    t1 = time.perf_counter()
    b = Batch(prepared, [[ps]] * 512, max_structs=eng.max_structs, cand_per_nt=max(eng.cand_per_nt, 64), fp32=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    b.align_accumulate(list(range(512)), cols, matrix)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    b.close(); t4 = time.perf_counter()
    print("prepared %.1f  batch %.1f  accumulate %.1f  close %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
