"""Multi-process path on CPU (gloo, world_size 2): sharding + the single result gather.
The compute engine is the test-only OracleEngine; what is checked is that the sharded
output is byte-identical to the reference text."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_lpt_partition_balances():
    from squarna_amd.parallel import lpt_partition
    costs = [n * n for n in (150, 8, 9, 100, 100, 30, 60, 61, 149, 12)]
    parts = lpt_partition(costs, 4)
    assert sorted(k for p in parts for k in p) == list(range(len(costs)))
    loads = [sum(costs[k] for k in p) for p in parts]
    assert max(loads) <= 1.35 * (sum(costs) / 4)
    assert lpt_partition(costs, 1) == [list(range(len(costs)))]


WORKER = textwrap.dedent("""
    import io, os, sys
    sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    from squarna_amd import engine as E
    from squarna_amd.parallel import PredictSharded
    from tests.oracle_engine import OracleEngine
    dist.init_process_group("gloo")
    buf = io.StringIO()
    with E.use_engine(OracleEngine()):
        PredictSharded(write_to=buf, inputfile="datasets/SRtest150.fas", inputformat="qf", configfile="fastest")
    if dist.get_rank() == 0:
        open(%(out)r, "w").write(buf.getvalue())
    dist.barrier()
    dist.destroy_process_group()
""")


def test_sharded_predict_world2_matches_reference(tmp_path):
    out = str(tmp_path / "sharded.txt")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    with open(os.path.join(GOLDEN, "text", "SRtest150_fastest.txt")) as f:
        assert open(out).read() == f.read()


ALI_WORKER = textwrap.dedent("""
    import io, os, sys
    sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    from squarna_amd import engine as E
    from squarna_amd.parallel import PredictSharded
    from tests.oracle_engine import OracleEngine
    dist.init_process_group("gloo")
    with E.use_engine(OracleEngine()):
        for tag, kw in %(jobs)r:
            buf = io.StringIO()
            PredictSharded(write_to=buf, **kw)
            if dist.get_rank() == 0:
                open(os.path.join(%(out)r, tag + ".txt"), "w").write(buf.getvalue())
    dist.barrier()
    dist.destroy_process_group()
""")


def test_sharded_alignment_world2_matches_reference(tmp_path):
    """Alignment mode: sequences sharded over 2 ranks, all_reduce(sum) of the partial stem matrices
    (ali.conf weights are dyadic, so the reduced matrix equals the sequential sum exactly)."""
    import json
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        dig = json.load(f)
    jobs = []
    for tag in ("ali_input_a", "ali_input_a_verbose", "ali_input_a_entropy", "demo_afa_a"):
        kw = dict(dig[tag]["args"])
        kw["inputfile"] = os.path.join(ROOT, "squarna_amd", "data", kw["inputfile"])
        jobs.append((tag, kw))
    script = tmp_path / "worker.py"
    script.write_text(ALI_WORKER % dict(root=ROOT, out=str(tmp_path), jobs=jobs))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29733", str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    for tag, _ in jobs:
        with open(os.path.join(GOLDEN, "text", tag + ".txt")) as f:
            assert open(str(tmp_path / (tag + ".txt"))).read() == f.read(), tag
