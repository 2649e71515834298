"""Multi-GPU driver: independent input records are sharded over the ranks of one node
(one process per GPU, torch.distributed; backend "nccl" == RCCL over xGMI on MI355X), every
rank folds its own shard, and ONE collective at the end gathers the printed blocks to rank 0,
which emits them in input order (the reference's ordered ``Pool.imap``, SQUARNA.py:887-935).
There is no exchange inside the data path.
"""
import io
import os
import sys

import numpy as np


def lpt_partition(costs, world):
    """Longest-processing-time-first: deal records (heaviest first) to the least loaded rank.
    Returns a list of index lists, one per rank."""
    order = sorted(range(len(costs)), key=lambda k: (-costs[k], k))
    load = [0.0] * world
    parts = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda q: (load[q], q))
        parts[r].append(k)
        load[r] += costs[k]
    return [sorted(p) for p in parts]


def gather_blocks(blocks, total, device=None, group=None):
    """blocks: {record index: text}.  Returns the full ordered list on rank 0, None elsewhere.
    One all_gather of sizes + one all_gather of a packed uint8 payload (RCCL on GPU ranks)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    idx = sorted(blocks)
    payload = [blocks[k].encode() for k in idx]
    head = np.array([len(idx)] + [v for k, b in zip(idx, payload) for v in (k, len(b))], dtype=np.int64)
    body = np.frombuffer(head.tobytes() + b''.join(payload), dtype=np.uint8)
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([body.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    cap = int(max(int(s.item()) for s in sizes))
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine[:body.size] = torch.from_numpy(body.copy()).to(dev)
    parts = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    if rank != 0:
        return None
    out = [None] * total
    for r in range(world):
        raw = parts[r][:int(sizes[r].item())].cpu().numpy().tobytes()
        n = int(np.frombuffer(raw[:8], dtype=np.int64)[0])
        meta = np.frombuffer(raw[8:8 + 16 * n], dtype=np.int64).reshape(n, 2)
        off = 8 + 16 * n
        for k, ln in meta:
            out[int(k)] = raw[off:off + int(ln)].decode()
            off += int(ln)
    assert all(b is not None for b in out), "a record was not folded by any rank"
    return out


def PredictSharded(write_to=None, device=None, **kwargs):
    """`Predict` across the ranks of an initialised torch.distributed group.
    Same keyword arguments as `Predict`; rank 0 writes the complete output, in input order,
    byte-identical to the single-process output."""
    import torch.distributed as dist
    from .api import Predict
    from .inputs import ParseInput
    assert dist.is_initialized(), "initialise torch.distributed first (torchrun)"
    world, rank = dist.get_world_size(), dist.get_rank()
    # parse once to get the record lengths (cheap, O(input size)); cost model: N^2 per record
    probe = dict(kwargs)
    inputfile = probe.get("inputfile", probe.get("i"))
    inputseq = probe.get("inputseq", probe.get("s", probe.get("seq")))
    if inputfile is not None and not os.path.exists(inputfile):
        from .config import DATA_DIR
        cand = os.path.join(probe.get("HOME_DIR") or DATA_DIR, inputfile)
        if os.path.exists(cand):
            inputfile = cand
    recs, _, _ = ParseInput(inputseq, inputfile, probe.get("inputformat", "qtrf"),
                            fmt=probe.get("fileformat", probe.get("ff", "unknown")),
                            ignore=bool(probe.get("ignorewarn", probe.get("iw", False))),
                            inputrestr=probe.get("inputrestr"),
                            M=float(probe.get("M", 1.8)), B=float(probe.get("B", -0.6)))
    lens = [len(r[1]) for r in recs]
    parts = lpt_partition([float(n) * n for n in lens], world)
    blocks = {}
    Predict(write_to=io.StringIO(), _select=set(parts[rank]), _on_block=lambda k, t: blocks.__setitem__(k, t),
            **kwargs)
    out = gather_blocks(blocks, len(lens), device=device)
    if rank == 0:
        sink = write_to if write_to is not None else sys.stdout
        for b in out:
            sink.write(b)
    return out
