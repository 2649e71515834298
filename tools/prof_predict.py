import io, os, sys, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from squarna_amd import Predict
path = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "squarna_amd/data/datasets/SRtest150.fas")
def run():
    buf = io.StringIO()
    Predict(inputfile=path, inputformat="qf", configfile="nobpp", write_to=buf)
run(); run()
cProfile.run("run()", "/tmp/p.out")
pstats.Stats("/tmp/p.out").sort_stats("cumtime").print_stats(28)
