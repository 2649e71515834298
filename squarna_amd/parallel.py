"""Multi-GPU driver: independent input records are sharded over the ranks of one node
(one process per GPU, torch.distributed; backend "nccl" == RCCL over xGMI on MI355X), every
rank folds its own shard, and ONE collective at the end gathers the printed blocks to rank 0,
which emits them in input order (the reference's ordered ``Pool.imap``, SQUARNA.py:887-935).
There is no exchange inside the data path.
"""
import io
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


def _collective_device(device=None, group=None):
    """Where the tensors of a collective must live: the caller's choice, else the current GPU under the
    "nccl" backend (RCCL has no CPU tensors), else the CPU (gloo)."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return torch.device(device)
    if "nccl" in str(dist.get_backend(group)).lower():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


MAX_SEGMENTS = 4


def gather_bytes(body, dst=0, device=None, group=None):
    """The single result gather of the multi-GPU path: every rank's packed payload travels to rank `dst` ONLY,
    at its exact size -- one all_gather of the sizes, then one batch of point-to-point transfers (RCCL send / recv over
    xGMI on GPU ranks, i.e. a gather without padding; gloo on CPU).  `body`: a uint8 array, or up to MAX_SEGMENTS of them
    (e.g. an index header + the library's pinned result buffer): the segments are sent as they are -- the pinned buffer
    goes host -> device in ONE DMA, no pageable copy in front -- and arrive concatenated.  Returns the list of payloads
    (uint8 numpy arrays, by rank) on `dst`, None elsewhere.  PredictSharded and bench.py's sharded legs both end here."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    segs = [body] if isinstance(body, np.ndarray) or not isinstance(body, (list, tuple)) else list(body)
    segs = [np.ascontiguousarray(x, dtype=np.uint8) for x in segs]
    assert 1 <= len(segs) <= MAX_SEGMENTS
    if world == 1:
        return [segs[0] if len(segs) == 1 else np.concatenate(segs)]
    dev = _collective_device(device, group)
    size = torch.tensor([x.size for x in segs] + [-1] * (MAX_SEGMENTS - len(segs)), dtype=torch.int64, device=dev)
    sizes = [torch.zeros(MAX_SEGMENTS, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [[int(v) for v in t.tolist() if v >= 0] for t in sizes]
    ops, bufs, keep = [], [None] * world, []
    if rank == dst:
        for r in range(world):
            if r != dst:
                bufs[r] = [torch.empty(n, dtype=torch.uint8, device=dev) if n else None for n in sizes[r]]
                ops += [dist.P2POp(dist.irecv, t, _global_rank(r, group), group) for t in bufs[r] if t is not None]
    else:
        for x in segs:
            if x.size:
                if not x.flags.writeable:
                    x = x.copy()                                # (torch wants a writable array; it never writes a send buffer)
                t = torch.from_numpy(x)
                t = t.to(dev, non_blocking=True) if dev.type == "cuda" else t
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, _global_rank(dst, group), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        if r == dst:
            out.append(segs[0] if len(segs) == 1 else np.concatenate(segs))
            continue
        n = sum(sizes[r])
        if dev.type == "cuda":                                  # device -> pinned host, one buffer per sender
            host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            at = 0
            for t in bufs[r]:
                if t is not None:
                    host[at:at + t.numel()].copy_(t, non_blocking=True)
                    at += t.numel()
            torch.cuda.synchronize(dev)
            out.append(host.numpy())
        else:
            parts = [t.numpy() for t in bufs[r] if t is not None]
            out.append(np.concatenate(parts) if parts else np.zeros(0, np.uint8))
    return out


def _global_rank(r, group):
    import torch.distributed as dist
    return r if group is None else dist.get_global_rank(group, r)


def allgather_bytes(body, device=None, group=None):
    """Every rank's payload to EVERY rank (alignment step 2: all ranks go on with the consensus): sizes first, then one
    all_gather of the payloads padded to the largest.  Returns the list of payloads by rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    body = np.ascontiguousarray(body, dtype=np.uint8)
    if world == 1:
        return [body]
    dev = _collective_device(device, group)
    size = torch.tensor([body.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size, group=group)
    sizes = [int(t.item()) for t in sizes]
    cap = max(max(sizes), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if body.size:
        mine[:body.size] = torch.from_numpy(body).to(dev)
    parts = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return [parts[r][:sizes[r]].cpu().numpy() for r in range(world)]


def pack_indexed(items):
    """{record index: bytes} -> one uint8 payload: int64 count, then (index, length) pairs, then the blobs."""
    idx = sorted(items)
    head = np.array([len(idx)] + [v for k in idx for v in (k, len(items[k]))], dtype=np.int64)
    return np.frombuffer(bytearray(head.tobytes() + b''.join(items[k] for k in idx)), dtype=np.uint8)   # (writable: torch.from_numpy)


def unpack_indexed(raw, out):
    """Inverse of pack_indexed: stores every blob of the payload at out[index]."""
    raw = raw.tobytes() if not isinstance(raw, (bytes, bytearray)) else raw
    if not raw:
        return
    n = int(np.frombuffer(raw[:8], dtype=np.int64)[0])
    meta = np.frombuffer(raw[8:8 + 16 * n], dtype=np.int64).reshape(n, 2)
    off = 8 + 16 * n
    for k, ln in meta:
        out[int(k)] = raw[off:off + int(ln)]
        off += int(ln)


def gather_blocks(blocks, total, device=None, group=None):
    """blocks: {record index: text}.  Returns the full ordered list on rank 0, None elsewhere (gather_bytes: the text
    blocks go to rank 0 only, unpadded)."""
    got = gather_bytes(pack_indexed({k: t.encode() for k, t in blocks.items()}), 0, device, group)
    if got is None:
        return None
    out = [None] * total
    for raw in got:
        unpack_indexed(raw, out)
    assert all(b is not None for b in out), "a record was not folded by any rank"
    return [b.decode() for b in out]


class ShardedAlignEngine:
    """Alignment mode over the ranks of a group: every rank owns a CONTIGUOUS block of the alignment's
    sequences (balanced on N^2), accumulates the partial L x L stem matrix of its block, and ONE
    all_reduce(sum) per step-1 iteration (RCCL over xGMI on GPU ranks) makes the full matrix available to
    every rank; MatrixToDBNs then runs replicated.  Step-2 folds are sharded the same way and exchanged with
    one all_gather.  Within a block the per-cell summation order is the reference's; across blocks the
    all_reduce adds the partial sums in ring order, so cells of non-dyadic scores can differ from the
    sequential sum in the last bits (dyadic bpweights without reactivities -- ali.conf -- are exact)."""

    def __init__(self, base, group=None):
        import torch.distributed as dist
        self.base, self.group = base, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if hasattr(base, "stem_matrix"):
            self.stem_matrix = self._stem_matrix
            self.matrix_cells = base.matrix_cells

    def _block(self, lens):
        cost = np.cumsum([0.0] + [float(n) * n for n in lens])
        cut = [int(np.searchsorted(cost, cost[-1] * r / self.world, side="left")) for r in range(self.world + 1)]
        cut[0], cut[-1] = 0, len(lens)
        return cut[self.rank], max(cut[self.rank], cut[self.rank + 1])

    @staticmethod
    def _ungapped(seq):
        return sum(1 for ch in seq if ch not in "-.~")

    def _stem_matrix(self, records, bpweights, minlen, minbpscore, interchainonly=False):
        import torch
        lo, hi = self._block([self._ungapped(r[0]) for r in records])
        if hi > lo:
            return self.base.stem_matrix(records[lo:hi], bpweights, minlen, minbpscore, interchainonly)
        n = len(records[0][0])
        return torch.zeros((n, n), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))

    def yield_stems(self, records, bpweights, minlen, minbpscore, interchainonly=False):
        lo, hi = self._block([self._ungapped(r[0]) for r in records])
        mine = self.base.yield_stems(records[lo:hi], bpweights, minlen, minbpscore, interchainonly) if hi > lo else []
        blank = [("", [])] * len(records)                        # sequences of other ranks add nothing here
        return blank[:lo] + list(mine) + blank[hi:]

    def reduce_matrix(self, matrix):
        import torch
        import torch.distributed as dist
        if isinstance(matrix, np.ndarray):                     # (host matrices: engines without stem_matrix)
            t = torch.from_numpy(np.ascontiguousarray(matrix)).to(_collective_device(None, self.group))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()
        dist.all_reduce(matrix, op=dist.ReduceOp.SUM, group=self.group)
        return matrix

    def entropy(self, record, interchainonly=False):
        return self.base.entropy(record, interchainonly=interchainonly)

    def fold_records(self, recs, **kw):
        """Step 2: every rank folds its block; the results travel as the library's packed records (sq_result_pack_all
        layout), one all_gather of uint8 tensors -- every rank needs them, the consensus is computed replicated."""
        lo, hi = self._block([self._ungapped(r[0]) for r in recs])
        out = [None] * len(recs)
        if hasattr(self.base, "fold_records_packed"):
            from .engine import Prepared, unpack_result
            blobs = self.base.fold_records_packed(recs[lo:hi], **kw) if hi > lo else []
            for raw in allgather_bytes(pack_indexed({lo + q: bl for q, bl in enumerate(blobs)}), group=self.group):
                unpack_indexed(raw, out)
            return [unpack_result(Prepared(r[0], r[1], r[2], r[3]), bl)[0] for r, bl in zip(recs, out)]
        # engines without packed results (the tests' CPU oracle engine): pickled tuples, same exchange
        import pickle
        mine = self.base.fold_records(recs[lo:hi], **kw) if hi > lo else []
        for raw in allgather_bytes(pack_indexed({lo + q: pickle.dumps(x) for q, x in enumerate(mine)}), group=self.group):
            unpack_indexed(raw, out)
        return [pickle.loads(x) for x in out]


def PredictSharded(write_to=None, device=None, **kwargs):
    """`Predict` across the ranks of an initialised torch.distributed group.
    Same keyword arguments (and synonyms) as `Predict`; rank 0 writes the complete output in input order.
    Single-sequence mode: byte-identical to the single-process output (records are independent).
    Alignment mode: byte-identical when every partial sum is exact (dyadic bpweights -- all shipped configs --
    and no reactivity factors); otherwise the all_reduce adds the per-rank partial matrices in a different
    fp64 order than the reference's sequential loop (SQRNdbnali.py:233-237), so cells may differ in the last bits.
    `device`: where the collective's tensors live; default = the current GPU under "nccl" (RCCL), CPU under gloo."""
    import torch.distributed as dist
    from .api import Predict
    assert dist.is_initialized(), "initialise torch.distributed first (torchrun)"
    world, rank = dist.get_world_size(), dist.get_rank()
    if any(kwargs.get(k) for k in ("alignment", "ali", "a")):
        # alignment mode: sequences of the MSA are sharded, one all_reduce per step-1 iteration
        from . import engine as _engine
        buf = io.StringIO()
        with _engine.use_engine(ShardedAlignEngine(_engine.get_engine())):
            Predict(write_to=buf, **kwargs)
        if rank == 0:
            (write_to if write_to is not None else sys.stdout).write(buf.getvalue())
            return [buf.getvalue()]
        return None
    # record lengths from Predict's own parsing (same synonyms, same HOME_DIR lookup, same warnings policy), so the
    # shard every rank computes is the shard Predict folds; cost model: N^2 per record
    lens = Predict(write_to=io.StringIO(), _lengths_only=True, **kwargs)
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
