"""Strong-scaling proxy of bench.py alone: python tools/shard_probe2.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.set_device(0)
from bench import scaling_proxy_leg
print(json.dumps(scaling_proxy_leg(), indent=0))
