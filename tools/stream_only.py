"""bench.py's stream leg alone (sequential / pipelined / rolling).  usage: stream_only.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
stream, one = bench.stream_leg("nobpp", 8, 12, steps, 6, dev)
for k in ("seq_per_s", "ms_per_step"):
    print("sequential", k, stream[k])
print("pipelined", {k: stream["pipelined"].get(k) for k in ("seq_per_s", "ms_per_step", "slowest_step", "error")})
print("rolling", stream.get("rolling"))
print("one_pass", {k: one[k] for k in ("ms", "best_ms", "seq_per_s")})
