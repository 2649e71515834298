import io, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from squarna_amd.parallel import PredictSharded
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for kw, gold in ((dict(inputfile="datasets/SRtest150.fas", inputformat="qf", configfile="fastest"), "SRtest150_fastest"),
                 (dict(inputfile=os.path.join(root, "squarna_amd/data/examples/ali_input.afa"), alignment=True), None)):
    buf = io.StringIO()
    try:
        PredictSharded(write_to=buf, device=torch.device("cuda", 0), **kw)
    except Exception as e:
        print("ERR", type(e).__name__, e); continue
    if gold:
        exp = open(os.path.join(root, "tests/golden/text", gold + ".txt")).read()
        print(gold, "identical" if buf.getvalue() == exp else "DIFFERENT", len(buf.getvalue()))
    else:
        print(buf.getvalue()[-300:])
dist.destroy_process_group()
