#!/usr/bin/env python3
"""World size 2 with the PRODUCT engine on ONE GPU: two ranks under gloo, both folding on cuda:0 (a one-GPU box cannot run
RCCL at world size 2).  PredictSharded's shard / gather / reduce logic with HIP batches on both ranks; rank 0 compares with
the golden texts.  Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1
--master-port 29741 tools/world2_hip_check.py"""
import io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from squarna_amd.parallel import PredictSharded
torch.cuda.set_device(0)
dist.init_process_group("gloo")
GOLDEN = os.path.join(ROOT, "tests", "golden")
dig = json.load(open(os.path.join(GOLDEN, "digests.json")))
bad = 0
for tag in ("SRtest150_fastest", "seq_input_nobpp", "ali_input_a", "ali_input_a_verbose", "demo_afa_a"):
    kw = dict(dig[tag]["args"])
    if "inputfile" in kw:
        kw["inputfile"] = os.path.join(ROOT, "squarna_amd", "data", kw["inputfile"])
    buf = io.StringIO()
    PredictSharded(write_to=buf, **kw)
    if dist.get_rank() == 0:
        ok = buf.getvalue() == open(os.path.join(GOLDEN, "text", tag + ".txt")).read()
        print("%-24s world 2, HIP engine on both ranks: %s" % (tag, "identical to the golden text" if ok else "DIFFERENT"), flush=True)
        bad += not ok
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
