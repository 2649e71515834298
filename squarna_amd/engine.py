"""HIP engine: batches sequences, owns the device workspace (a torch uint8 tensor) and
drives libsquarna_hip.so through its C ABI.  PyTorch is plumbing here (device memory,
stream, torch.distributed); all arithmetic runs in the hand-written gfx950 kernels.

There is NO CPU fallback: without the built library or without a GPU the engine raises.
Tests may install another engine with :func:`use_engine` (tests/ only use that to check
the host-side text layer on CPU against the oracle).
"""
import ctypes as C
import os
import struct
import contextlib

import weakref
import numpy as np

from . import _lib
from .dbn import (gap_mask, GAPS, SEPS, ReactDict, ProcessReacts, DBNToPairs, UnAlign, ReAlign,
                  ParseRestraints, levels_to_dbn, encode_seq, BRACKETS)


_STEM_DT = np.dtype([("i", "<i4"), ("j", "<i4"), ("len", "<i4"), ("reserved", "<i4"),
                     ("bpscore", "<f8"), ("finscore", "<f8")])


class Prepared:
    """One input record after the host pre-processing of SQRNdbnseq.py:1001-1037."""
    __slots__ = ("seq", "shortseq", "shortrest", "shortreacts", "shortdbn", "rbps", "rxs",
                 "rlefts", "rrights", "gapidx", "sepidx", "plain_reacts", "refpairs")

    _NONE = ([], frozenset())

    def __init__(self, seq, reacts=None, restraints=None, dbn=None):
        seq = seq.upper().replace("T", "U")                          # :1004
        if not reacts and not restraints and not dbn and seq.isalpha():
            # a plain record (letters only: no gap column, no separator; nothing but the sequence given): every field is
            # what the general path below would compute, without its per-record string work -- most records of a big
            # input are like this
            self.seq = self.shortseq = seq
            self.shortrest = None
            self.shortreacts = None                                  # (all 0.5: plain_reacts says so)
            self.plain_reacts = True
            self.gapidx = self.sepidx = self.rbps = self._NONE[0]
            self.rxs = self.rlefts = self.rrights = self._NONE[1]
            self.shortdbn = self.refpairs = None
            return
        if dbn and not reacts and not restraints and seq.isalpha():
            # the same record with a known structure beside it (a benchmark set): what the general path below computes
            # for it, without its string passes (no gap column: UnAlign returns its arguments; no restraint: four empties)
            assert len(seq) == len(dbn)
            n = len(seq)
            self.seq = self.shortseq = seq
            self.shortrest = '.' * n
            self.shortreacts = [0.5] * n
            self.plain_reacts = True
            self.gapidx, self.sepidx, self.rbps = [], [], []
            self.rxs, self.rlefts, self.rrights = set(), set(), set()
            self.shortdbn = dbn
            self.refpairs = None
            return
        if not restraints:
            restraints = '.' * len(seq)                              # :1007-1008
        assert len(seq) == len(restraints), "Invalid restraints given"
        self.plain_reacts = not reacts                               # all 0.5: the batch fills them in one go
        if not reacts:
            reacts = [0.5] * len(seq)                                # :1013-1014
        assert len(reacts) == len(seq), "Invalid reactivities given"
        if type(reacts) == str:                                      # :1019-1020 (default B = 1.6)
            reacts = ProcessReacts([ReactDict[ch] for ch in reacts])
        self.seq = seq
        self.shortseq, self.shortrest, rbps = UnAlign(seq, restraints, want_pairs=True)     # :1023
        if '-' in seq or '.' in seq or '~' in seq:
            gaps = gap_mask(seq)
            self.gapidx = np.flatnonzero(gaps).tolist()
        else:
            gaps, self.gapidx = None, []
        self.sepidx = [i for i, ch in enumerate(seq) if ch in SEPS] if (';' in seq or '&' in seq) else []
        if self.plain_reacts:
            self.shortreacts = [0.5] * len(self.shortseq)
        elif not self.gapidx:
            self.shortreacts = list(reacts)
        else:
            self.shortreacts = np.asarray(reacts, dtype=np.float64)[~gaps].tolist()
        self.shortdbn = None
        self.refpairs = None                                         # pairs of the known structure (Batch._fold_args), formed once
        if dbn:
            assert len(seq) == len(dbn)
            self.shortseq, self.shortdbn = UnAlign(seq, dbn)         # :1026-1028
        self.rbps, self.rxs, self.rlefts, self.rrights = ParseRestraints(self.shortrest, rbps)   # :1037


class PackedRows:
    """The rows of ONE alignment after the host pre-processing of SQRNdbnseq.py:1001-1037 / SQRNdbnali.py:60-86, for all rows at
    once as array code: the alignment is one uint8[rows, columns] array, and letter codes, gap maps, restraint flags and the
    restraint pairs that survive each row's gaps (UnAlign, SQRNdbnseq.py:236-255) come out of it with a handful of numpy
    calls -- what a list of per-row Prepared records holds, in the layout Batch uploads (config 5: 2 x 512 rows of 5,000
    columns were 1.3 s of per-row string work, most of it a character loop over a restraint line whose bracket letters
    leave latin-1 beyond 30 pseudoknot levels).  Rows without reactivities, one restraint line shared by all rows (or none).
    cols: the alignment column of every position, row after row (ReAlignDict, SQRNdbnali.py:20-37)."""
    __slots__ = ("nseq", "seq_off", "codes", "flags", "reacts", "rbp_off", "rbps", "cols", "lengths")

    def __init__(self, seqs, restraint_line=None):
        from .dbn import _CODE_LUT, encode_seq
        R, Lc = len(seqs), len(seqs[0])
        if _CODE_LUT is None:
            encode_seq("A")                                          # (builds the table)
        from .dbn import _CODE_LUT as LUT
        A = np.frombuffer("".join(seqs).encode("latin-1", "replace"), np.uint8).reshape(R, Lc)
        gap_mask("-")                                                # (builds the gap table)
        from .dbn import _GAP_LUT
        keep = ~_GAP_LUT[A]
        self.nseq = R
        self.lengths = keep.sum(axis=1)
        self.seq_off = np.zeros(R + 1, np.int32)
        np.cumsum(self.lengths, out=self.seq_off[1:])
        ltot = int(self.seq_off[-1])
        self.codes = LUT[A[keep]] if ltot else np.zeros(1, np.uint8)
        self.cols = np.ascontiguousarray(np.nonzero(keep)[1], np.int32) if ltot else np.zeros(1, np.int32)
        self.reacts = None
        self.flags = np.zeros(max(ltot, 1), np.uint8)
        self.rbp_off = np.zeros(R + 1, np.int32)
        self.rbps = np.zeros(2, np.int32)
        if restraint_line and restraint_line.count(".") != len(restraint_line):
            assert len(restraint_line) == Lc, "Invalid restraints given"
            cp = np.frombuffer(restraint_line.encode("utf-32-le"), np.uint32)        # code points: any bracket alphabet
            fl = np.zeros(Lc, np.uint8)
            fl[(cp == ord("_")) | (cp == ord("+"))] |= 1             # SQRNdbnseq.py:370-376: unpaired
            fl[cp == ord("/")] |= 2                                  # no pair to the left
            fl[cp == ord("\\")] |= 4                                 # no pair to the right
            if fl.any() and ltot:
                self.flags = np.ascontiguousarray(np.broadcast_to(fl, (R, Lc))[keep])
            pairs = DBNToPairs(restraint_line)                       # once: every row shares the line
            if pairs:
                v = np.fromiter((p[0] for p in pairs), np.int64, len(pairs))
                w = np.fromiter((p[1] for p in pairs), np.int64, len(pairs))
                ok = keep[:, v] & keep[:, w]                         # a pair that touches a gap of the row is dropped (:243-249)
                rank = np.cumsum(keep, axis=1, dtype=np.int32) - 1   # column -> position of the row
                rr, pp = np.nonzero(ok)                              # row-major: every row's pairs in the line's (sorted) order
                rb = np.empty((len(rr), 2), np.int32)
                rb[:, 0] = rank[rr, v[pp]]
                rb[:, 1] = rank[rr, w[pp]]
                np.cumsum(ok.sum(axis=1), out=self.rbp_off[1:])
                if len(rr):
                    self.rbps = rb.reshape(-1)


_HDR = struct.Struct("<4q")
_MET = struct.Struct("<16d")
_MASK_IDS = [[q for q in range(4) if (m >> q) & 1] for m in range(16)]

#: code points of the bracket characters by signed level (+L opening, -L closing, 0 dot; levels beyond the
#: alphabet print as dots, SQRNdbnseq.py:142-143), indexed by level + _NBR + 1
_NBR = len(BRACKETS)
_LEVEL_CP = np.full(2 * _NBR + 3, ord('.'), np.uint32)
for _l in range(1, _NBR + 1):
    _LEVEL_CP[_NBR + 1 + _l] = ord(BRACKETS[_l - 1][0])
    _LEVEL_CP[_NBR + 1 - _l] = ord(BRACKETS[_l - 1][1])


class _BlockRun:
    """fold_records(..., _blocks=...) result of one batch whose blocks the library wrote completely."""
    __slots__ = ("blocks",)

    def __init__(self, blocks):
        self.blocks = blocks

    def __len__(self):
        return len(self.blocks)

    def __iter__(self):
        return (("text", t) for t in self.blocks)


class _Blocks:
    """Output blocks of all records of a batch as ONE string + offsets (a list of per-record slices on demand): a caller
    that prints them in order writes the string once."""
    __slots__ = ("text", "off")

    def __init__(self, text, off):
        self.text, self.off = text, off

    def __len__(self):
        return len(self.off) - 1

    def __iter__(self):
        o, t = self.off.tolist(), self.text
        return (t[o[k]:o[k + 1]] for k in range(len(o) - 1))


def _pset_struct(ps):
    out = _lib.ParamSet()
    for key, val in ps["bpweights"].items():                         # SQRNdbnseq.py:282-284
        a, b = encode_seq(key)
        if a > 25 or b > 25:
            raise ValueError("bpweights keys must be two letters: %r" % key)
        out.bpweight[a * 32 + b] = val
        out.inbps[a * 32 + b] = 1
        out.bpweight[b * 32 + a] = val
        out.inbps[b * 32 + a] = 1
    out.bpp = float(ps.get("bpp", 0))
    for k in ("suboptmax", "suboptmin", "suboptsteps", "minlen", "minbpscore", "minfinscorefactor",
              "bracketweight", "distcoef", "orderpenalty", "loopbonus", "maxstemnum"):
        setattr(out, k, float(ps[k]))
    out.algorithms = sum(_lib.ALGO_BITS[a] for a in ps["algorithms"])
    return out


def _ptr(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


class Batch:
    """A device-resident batch of fold jobs (one per (record, paramset))."""

    def __init__(self, prepared, psets_per_record, interchainonly=False, ext=None, mul=None,
                 max_structs=0, cand_per_nt=0, device=None, fp32=True, bpp=None, mul_shared=None, pool_lists=False):
        """fp32=False leaves the fp32 score matrices out of the workspace (4 N^2 bytes per job): everything
        but fill() works -- folding only needs the 1-bit-per-cell matrices.
        pool_lists=True: the batch will be folded with pools wider than one; with sequences of 257-1,024 nt the workspace
        then holds the pages of the lists a pool's structures hand to their children (SQ_BATCH_POOL_LISTS).
        mul_shared = (M, cols, maxabs): ONE L x L fp64 torch tensor on the GPU that weights every job of every record
        (alignment step 2), cols[k] = the alignment columns of record k's gap-free positions, maxabs >= max |M|."""
        import torch
        L = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("squarna_amd needs an AMD GPU (MI355X / gfx950): torch.cuda is not "
                               "available and there is no CPU fallback")
        self.torch = torch
        self.L = L
        if isinstance(prepared, PackedRows):
            # an alignment's rows as arrays (no per-row records: the batch computes, it has no results to decode)
            pk, prepared = prepared, None
            nseq, ltot = pk.nseq, int(pk.seq_off[-1])
            self.prepared = None
            self.seq_off, self.codes, self.flags, self.reacts, self.rbp_off, self.rbps = pk.seq_off, pk.codes, pk.flags, pk.reacts, pk.rbp_off, pk.rbps
        else:
            nseq, ltot = self._host_arrays(prepared)
        self._pool_lists = bool(pool_lists)
        self._finish_init(nseq, ltot, psets_per_record, interchainonly, ext, mul, max_structs, cand_per_nt, device, fp32, bpp, mul_shared)

    def _host_arrays(self, prepared):
        """The per-position arrays of the batch from a list of Prepared records; (nseq, total positions)."""
        self.prepared = prepared
        nseq = len(prepared)
        self.seq_off = np.zeros(nseq + 1, np.int32)
        np.cumsum(np.fromiter((len(p.shortseq) for p in prepared), np.int64, nseq), out=self.seq_off[1:])
        ltot = int(self.seq_off[-1])
        self.codes = np.frombuffer(encode_seq(''.join(p.shortseq for p in prepared)), np.uint8).copy() \
            if ltot else np.zeros(1, np.uint8)
        self.flags = np.zeros(max(ltot, 1), np.uint8)
        # no record with reactivities: the library takes NULL for "0.5 everywhere" (8 bytes per position neither filled,
        # scanned nor uploaded)
        self.reacts = None if all(p.plain_reacts for p in prepared) else np.full(max(ltot, 1), 0.5, np.float64)
        self.rbp_off = np.zeros(nseq + 1, np.int32)
        rbps = []
        # (most records of a big input are plain: only the ones with restraints or reactivities take the loop)
        for k, p in enumerate(prepared):
            if p.plain_reacts and not (p.rbps or p.rxs or p.rlefts or p.rrights):
                continue
            o = int(self.seq_off[k])
            for i in p.rxs:
                self.flags[o + i] |= 1
            for i in p.rlefts:
                self.flags[o + i] |= 2
            for i in p.rrights:
                self.flags[o + i] |= 4
            if not p.plain_reacts:
                self.reacts[o:o + len(p.shortseq)] = p.shortreacts
            if p.rbps:
                rbps.extend(p.rbps)
                self.rbp_off[k + 1] = len(p.rbps)
        np.cumsum(self.rbp_off, out=self.rbp_off)
        self.rbps = np.array(rbps, np.int32).reshape(-1) if rbps else np.zeros(2, np.int32)
        return nseq, ltot

    def _finish_init(self, nseq, ltot, psets_per_record, interchainonly, ext, mul, max_structs, cand_per_nt, device, fp32, bpp, mul_shared):
        torch, L = self.torch, self.L
        # unique paramsets by identity
        uniq, self.psets_py = {}, []
        first = psets_per_record[0] if nseq else []
        if nseq and all(pl is first for pl in psets_per_record):
            # one configuration for every record (the usual case): the job lists are a repeat / tile
            idx = []
            for ps in first:
                if id(ps) not in uniq:
                    uniq[id(ps)] = len(self.psets_py)
                    self.psets_py.append(ps)
                idx.append(uniq[id(ps)])
            npl = len(first)
            self.job_seq = np.repeat(np.arange(nseq, dtype=np.int32), npl)
            self.job_pset = np.tile(np.array(idx, np.int32), nseq)
            self.seq_jobs = None                                     # (k -> range(k * npl, (k + 1) * npl), formed on demand)
            self._npl = npl
        else:
            job_seq, job_pset = [], []
            self.seq_jobs = []
            for k, plist in enumerate(psets_per_record):
                mine = []
                for ps in plist:
                    if id(ps) not in uniq:
                        uniq[id(ps)] = len(self.psets_py)
                        self.psets_py.append(ps)
                    mine.append(len(job_seq))
                    job_seq.append(k)
                    job_pset.append(uniq[id(ps)])
                self.seq_jobs.append(mine)
            self.job_seq = np.array(job_seq, np.int32)
            self.job_pset = np.array(job_pset, np.int32)
        self.psets_c = (_lib.ParamSet * len(self.psets_py))(*[_pset_struct(p) for p in self.psets_py])
        njobs = len(self.job_seq)
        d = _lib.BatchDesc()
        d.nseq = nseq
        d.seq_off = _ptr(self.seq_off, C.POINTER(C.c_int32))
        d.codes = _ptr(self.codes, C.POINTER(C.c_uint8))
        d.flags = _ptr(self.flags, C.POINTER(C.c_uint8))
        d.reacts = _ptr(self.reacts, C.POINTER(C.c_double)) if self.reacts is not None else None
        d.rbp_off = _ptr(self.rbp_off, C.POINTER(C.c_int32))
        d.rbps = _ptr(self.rbps, C.POINTER(C.c_int32))
        d.npset = len(self.psets_py)
        d.psets = self.psets_c
        d.njobs = njobs
        d.job_seq = _ptr(self.job_seq, C.POINTER(C.c_int32))
        d.job_pset = _ptr(self.job_pset, C.POINTER(C.c_int32))
        self._keep = []

        def ptr_array(mats):
            arr = (C.c_void_p * njobs)()
            for j, m in enumerate(mats):
                if m is not None:
                    m = np.ascontiguousarray(m, dtype=np.float64)
                    self._keep.append(m)
                    arr[j] = m.ctypes.data
            return arr

        if ext is not None:
            self._eb = ptr_array([e[0] if e is not None else None for e in ext])
            self._es = ptr_array([e[1] if e is not None else None for e in ext])
            d.ext_bool = C.cast(self._eb, C.POINTER(C.c_void_p))
            d.ext_score = C.cast(self._es, C.POINTER(C.c_void_p))
        if mul is not None:
            self._mul = ptr_array(mul)
            d.mul_score = C.cast(self._mul, C.POINTER(C.c_void_p))
        if mul_shared is not None:
            M, cols, maxabs = mul_shared
            assert M.is_cuda and M.dtype == torch.float64 and M.dim() == 2 and M.shape[0] == M.shape[1] and M.is_contiguous()
            self._mul_M = M
            self._mul_cols = np.ascontiguousarray(np.concatenate([np.asarray(c, np.int32) for c in cols])
                                                  if ltot else np.zeros(1, np.int32), dtype=np.int32)
            assert len(self._mul_cols) == max(ltot, 1)
            self._mul_flag = np.ones(max(njobs, 1), np.uint8)
            d.mul_matrix_dev = C.c_void_p(M.data_ptr())
            d.mul_L = int(M.shape[0])
            d.mul_cols = _ptr(self._mul_cols, C.POINTER(C.c_int32))
            d.mul_shared = _ptr(self._mul_flag, C.POINTER(C.c_uint8))
            d.mul_maxabs = float(maxabs)
        if bpp is not None:                                          # per job: (bppm/max)**|bpp| or None (SQRNdbnseq.py:350-364)
            self._bpp = ptr_array(bpp)
            d.bpp_term = C.cast(self._bpp, C.POINTER(C.c_void_p))
        d.interchainonly = int(bool(interchainonly))
        d.max_structs = int(max_structs)
        d.cand_per_nt = int(cand_per_nt)
        d.batch_flags = (0 if fp32 else _lib.BATCH_NO_FP32) | (_lib.BATCH_POOL_LISTS if self._pool_lists else 0)
        self.desc = d
        nbytes = C.c_size_t(0)
        _lib.check(L.sq_batch_workspace_bytes(C.byref(d), C.byref(nbytes)))
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        # (sizes in coarse steps: batches of a stream differ by a few records, and torch's caching allocator only hands a cached
        # block back for a request it nearly fits -- every new size was a hipMalloc, the occasional one with a device-wide
        # free of cached blocks in front: 40-190 ms steps in the stream leg)
        # (sixteen size classes per octave from 256 MB on: the 6 GB workspaces of a stream's batches -- 12 SRtest150 sets each --
        # differ by a few per cent, which in 64 MB steps was a new size every other step: torch's reserved memory grew from 95
        # to 173 GB over fourteen steps of the pipelined stream, and a step paid hundreds of ms for the device-wide free)
        want = nbytes.value + 256
        step = (1 << (want.bit_length() - 5)) if want >= (256 << 20) else (8 << 20) if want >= (16 << 20) else (1 << 20)
        self.workspace = torch.empty((want + step - 1) // step * step, dtype=torch.uint8, device=self.device)
        base = self.workspace.data_ptr()
        aligned = (base + 255) // 256 * 256
        self.stream = torch.cuda.current_stream(self.device)
        h = C.c_void_p()
        _lib.check(L.sq_batch_create(C.byref(h), C.byref(d), C.c_void_p(aligned),
                                     C.c_size_t(nbytes.value), C.c_void_p(self.stream.cuda_stream)))
        self.h = h
        self._refs = None
        self.njobs = njobs
        self.nseq = nseq

    # -- lifecycle
    def close(self):
        if getattr(self, "h", None):
            self.L.sq_batch_destroy(self.h)
            self.h = None
            self.workspace = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- a-1
    def fill(self):
        _lib.check(self.L.sq_bpmatrix_fill(self.h))

    def bpmatrix(self, job):
        n = int(self.seq_off[self.job_seq[job] + 1] - self.seq_off[self.job_seq[job]])
        b = np.zeros((n, n)); s = np.zeros((n, n))
        _lib.check(self.L.sq_bpmatrix_read(self.h, job, _ptr(b), _ptr(s)))
        return b, s

    # -- a-2..a-6
    def optimal(self, struct_job, struct_stems, subopt=None, mode=0, out_cap=None, as_array=False):
        """struct_stems: list (per structure) of (i, j, len) tuples -> list of lists of
        (i, j, len, bpscore, finalscore)."""
        ns = len(struct_job)
        sj = np.array(struct_job, np.int32)
        off = np.zeros(ns + 1, np.int32)
        flat = []
        for k, st in enumerate(struct_stems):
            flat.extend(st)
            off[k + 1] = len(flat)
        stems = (_lib.Stem * max(len(flat), 1))()
        for k, t in enumerate(flat):
            stems[k].i, stems[k].j, stems[k].len = int(t[0]), int(t[1]), int(t[2])
        so = np.array(subopt if subopt is not None else [1.0] * ns, np.float64)
        if out_cap is None:
            out_cap = 1 << 16 if mode == 0 else 1 << 20
        out = (_lib.Stem * out_cap)()
        out_off = np.zeros(ns + 1, np.int32)
        _lib.check(self.L.sq_optimal_stems(self.h, ns, _ptr(sj), _ptr(off), stems, _ptr(so), mode,
                                           out, out_cap, _ptr(out_off)))
        if as_array:                                                   # structured views, no per-stem objects
            arr = np.frombuffer(out, dtype=_STEM_DT, count=int(out_off[ns])).copy()
            return [arr[out_off[k]:out_off[k + 1]] for k in range(ns)]
        res = []
        for k in range(ns):
            res.append([(out[q].i, out[q].j, out[q].len, out[q].bpscore, out[q].finscore)
                        for q in range(out_off[k], out_off[k + 1])])
        return res

    # -- alignment step 1
    def align_accumulate(self, jobs, cols_per_job, matrix):
        """Adds the stem scores of the listed jobs, in order, into the device L x L fp64 tensor `matrix`
        through the gap maps cols_per_job[k] (unaligned index -> column); SQRNdbnali.py:233-237."""
        L = int(matrix.shape[0])
        assert matrix.dtype == self.torch.float64 and matrix.is_contiguous() and tuple(matrix.shape) == (L, L)
        ja = np.array(jobs, np.int32)
        off = np.zeros(len(jobs) + 1, np.int32)
        for k, c in enumerate(cols_per_job):
            off[k + 1] = off[k] + len(c)
        cols = np.concatenate([np.asarray(c, np.int32) for c in cols_per_job]) if len(jobs) else np.zeros(1, np.int32)
        cols = np.ascontiguousarray(cols, np.int32)
        _lib.check(self.L.sq_align_accumulate(self.h, len(jobs), _ptr(ja), _ptr(off), _ptr(cols), L,
                                              C.c_void_p(matrix.data_ptr())))

    def align_accumulate_packed(self, pk, matrix):
        """align_accumulate for every row of a PackedRows batch, its gap maps as they are (no per-row lists)."""
        L = int(matrix.shape[0])
        assert matrix.dtype == self.torch.float64 and matrix.is_contiguous() and tuple(matrix.shape) == (L, L)
        ja = np.arange(pk.nseq, dtype=np.int32)
        _lib.check(self.L.sq_align_accumulate(self.h, pk.nseq, _ptr(ja), _ptr(pk.seq_off), _ptr(pk.cols), L,
                                              C.c_void_p(matrix.data_ptr())))

    # -- a-8 / a-9 / Nussinov
    def run_algo(self, jobs, algo, levellimit=None, out_cap=1 << 16):
        """RunAlgo (SQRNdbnseq.py:548-595) for the listed jobs under 'E', 'H' or 'N':
        list (per job) of (i, j, len, score, score)."""
        nj = len(jobs)
        ja = np.array(jobs, np.int32)
        out = (_lib.Stem * out_cap)()
        off = np.zeros(nj + 1, np.int32)
        _lib.check(self.L.sq_run_algos(self.h, nj, _ptr(ja), _lib.ALGO_BITS[algo],
                                       -1 if levellimit is None else int(levellimit), out, out_cap, _ptr(off)))
        return [[(out[q].i, out[q].j, out[q].len, out[q].bpscore, out[q].finscore)
                 for q in range(off[k], off[k + 1])] for k in range(nj)]

    def set_inflight(self, n):
        """Tell the batch that n batches are folded at the same time from threads of the caller (sq_batch_set_inflight)."""
        _lib.check(self.L.sq_batch_set_inflight(self.h, int(n)))

    # -- a-7 + a-10
    def fold(self, **opts):
        """priority: per record, set of local paramset indices (or one set for all)."""
        o, ref_off, rp, has = self._fold_args(**opts)
        _lib.check(self.L.sq_fold(self.h, C.byref(o), _ptr(ref_off), _ptr(rp), _ptr(has)))

    def limit_results(self, k):
        """The result getters show only the first k structures of every record (sq_result_limit; 0 = all)."""
        _lib.check(self.L.sq_result_limit(self.h, int(k or 0)))

    @property
    def fold_peak_structs(self):
        """Most structures any round of the last fold held at once (sq_fold_peak_structs; 0: host-driven loop)."""
        return int(self.L.sq_fold_peak_structs(self.h))

    @property
    def fold_driver(self):
        """Driver of the last fold's greedy pool loop (sq_fold_driver): 0 host loop, 1 chained rounds, 2 device pools,
        3 device pools repeated by the host loop."""
        return int(self.L.sq_fold_driver(self.h))

    @property
    def fold_paths(self):
        """Bit 0: the last fold's ranking tail ran on the device, bit 1: RunAlgo's edge lists and filters did (sq_fold_paths)."""
        return int(self.L.sq_fold_paths(self.h))

    def _fold_args(self, poollim=1000, conslim=1, toplim=5, hardrest=False, rankbydiff=False,
                   rankby=(0, 2, 1), levellimit=None, algos=frozenset(), priority=None):
        o = _lib.FoldOpts()
        o.poollim, o.conslim, o.toplim = int(poollim), int(conslim), int(toplim)
        o.hardrest, o.rankbydiff = int(bool(hardrest)), int(bool(rankbydiff))
        for t in range(3):
            o.rankby[t] = int(rankby[t])
        o.levellimit = -1 if levellimit is None else int(levellimit)
        o.algos = sum(_lib.ALGO_BITS[a] for a in algos)
        mask = 0
        for p in (priority or ()):
            mask |= 1 << int(p)
        o.priority_mask = mask
        if self._refs is None:                       # reference pairs are static per batch
            ref_off = np.zeros(self.nseq + 1, np.int32)
            has = np.zeros(max(self.nseq, 1), np.uint8)
            dbns = [p.shortdbn or "" for p in self.prepared]
            text = "".join(dbns)
            if text.isascii():
                # DBNToPairs (SQRNdbnseq.py:172-207) for all known structures in one library call (sq_dbn_pairs)
                off = np.zeros(self.nseq + 1, np.int64)
                np.cumsum([len(x) for x in dbns], out=off[1:])
                poff = np.zeros(self.nseq + 1, np.int64)
                rp = np.zeros(max(len(text), 2), np.int32)                  # (a line of n characters has at most n / 2 pairs)
                _lib.check(self.L.sq_dbn_pairs(text.encode("ascii"), _ptr(off), self.nseq, _ptr(rp), len(rp) // 2, _ptr(poff)))
                ref_off[:] = poff
                has[:self.nseq] = [1 if x else 0 for x in dbns]
                rp = rp[:max(2 * int(poff[-1]), 2)]
            else:                                                        # (bracket letters beyond ASCII: the Python form)
                refs = []
                for k, p in enumerate(self.prepared):
                    if p.shortdbn:
                        has[k] = 1
                        if p.refpairs is None:
                            p.refpairs = DBNToPairs(p.shortdbn)
                        refs.extend(p.refpairs)
                    ref_off[k + 1] = len(refs)
                rp = np.array(refs, np.int32).reshape(-1) if refs else np.zeros(2, np.int32)
            self._refs = (ref_off, rp, has)
        ref_off, rp, has = self._refs
        return o, ref_off, rp, has

    def result(self, k, with_ref=False):
        """SQRNdbnseq return tuple of record k (SQRNdbnseq.py:1285-1286); with_ref: (tuple, reference scores or None)."""
        L = self.L
        nbytes = L.sq_result_pack_size(self.h, k)
        buf = bytearray(nbytes)
        cbuf = (C.c_char * nbytes).from_buffer(buf)
        _lib.check(L.sq_result_pack(self.h, k, cbuf, nbytes))
        out = self._unpack(k, buf, 0)
        return out if with_ref else out[0]

    def results_all(self):
        """[(SQRNdbnseq tuple, reference scores or None)] for every record, from ONE sq_result_pack_all call: the whole
        dot-bracket rows come as ASCII text from one sq_result_dbn_all call, headers / scores / masks through numpy views; a
        record then costs a few slices (records with gap columns, separators or > 30 pseudoknot levels take the per-record
        path)."""
        buf, off = self.pack_all()
        raw = buf.tobytes()
        # the dot-bracket rows of every record as ASCII, formed by the library in one call
        tbytes = int(self.L.sq_result_dbn_all_size(self.h))
        tbuf = np.zeros(max(tbytes, 8), np.uint8)
        toff = np.zeros(self.nseq + 1, np.int64)
        deep = np.zeros(max(self.nseq, 1), np.uint8)
        _lib.check(self.L.sq_result_dbn_all(self.h, _ptr(tbuf), tbytes, _ptr(toff), _ptr(deep)))
        text_all = tbuf[:tbytes].tobytes().decode('latin-1')
        toffl, deepl = toff.tolist(), deep.tolist()
        # headers, metrics, scores and masks of all records through numpy views (the records start 8-byte aligned)
        q = np.frombuffer(raw, '<i8', len(raw) // 8)
        qu = np.frombuffer(raw, '<u8', len(raw) // 8)               # paramset masks: bit 63 may be set (64 paramsets)
        d = np.frombuffer(raw, '<f8', len(raw) // 8)
        b8 = (off[:-1] // 8).astype(np.int64)
        ns_a, n_a, ref_a = q[b8].tolist(), q[b8 + 1].tolist(), q[b8 + 2].tolist()
        met_a = d[b8[:, None] + (4 + np.arange(16))].tolist()
        b8l = b8.tolist()
        nan6, nan7 = [np.nan] * 6, [np.nan] * 7
        out = []
        for k in range(self.nseq):
            p = self.prepared[k]
            if p.gapidx or p.sepidx or deepl[k]:
                out.append(self._unpack(k, raw, int(off[k])))
                continue
            ns, n, sb = ns_a[k], n_a[k], b8l[k] + 20
            sc = d[sb:sb + 3 * ns].tolist()
            mk = qu[sb + 3 * ns:sb + 4 * ns].tolist()
            t0 = toffl[k]                                          # the record's rows in text_all
            preds = [(text_all[t0 + (t + 1) * n:t0 + (t + 2) * n], tuple(sc[3 * t:3 * t + 3]),
                      list(_MASK_IDS[mk[t]]) if mk[t] < 16 else [b for b in range(64) if (mk[t] >> b) & 1]) for t in range(ns)]
            if ref_a[k]:
                met = met_a[k]
                out.append(((text_all[t0:t0 + n], preds, _metrics(met[:6]), _metrics(met[6:12]) + [int(met[12])]), tuple(met[13:16])))
            else:
                out.append(((text_all[t0:t0 + n], preds, list(nan6), list(nan7)), None))
        return out

    def _unpack(self, k, buf, base):
        return unpack_result(self.prepared[k], buf, base)

    def write_blocks(self, names, seqs, reactlines, restrs, refs, nameset, psnames, conslim, outplim):
        """The output blocks of RunSQRNdbnseq (SQRNdbnseq.py:1301-1406) for every record, formed by the library from the
        packed results of the last fold (sq_write_blocks): list of str, None for a record the library leaves to the
        caller (bracket levels beyond ASCII).  None when the batch's results are not packed (host tail) or some input
        line is not ASCII: the caller formats from results_all()."""
        if not (self.fold_paths & 1):
            return None
        fields = []
        for col in (names, seqs, reactlines, restrs, refs):
            if all(x is None or x == "" for x in col):
                fields.append(None)
                continue
            text = "\n".join(x or "" for x in col)
            if not text.isascii() or "\0" in text:                    # (the library reads NUL-terminated ASCII lines)
                return None
            fields.append(text.encode("ascii"))
        if fields[0] is None or fields[1] is None:
            return None
        d = _lib.BlockDesc()
        d.nrec = self.nseq
        d.names, d.seqs, d.reacts, d.restr, d.refs = fields
        ns = np.ascontiguousarray(nameset, np.int32)
        d.nameset = _ptr(ns, C.POINTER(C.c_int32))
        if not all(nm.isascii() and "\0" not in nm and "\n" not in nm for x in psnames for nm in x):
            return None                                               # (paramset names the library cannot carry: the caller formats)
        pn = [("\n".join(x)).encode("ascii") for x in psnames]
        arr = (C.c_char_p * len(pn))(*pn)
        d.psnames = arr
        d.nsets, d.conslim, d.outplim = len(pn), int(conslim), int(outplim)
        off = np.zeros(self.nseq + 1, np.int64)
        skipped = np.zeros(max(self.nseq, 1), np.uint8)
        cap = int(self.L.sq_result_dbn_all_size(self.h)) + sum(len(f) for f in fields if f) * 2 + 200 * self.nseq * (2 + int(outplim)) + 4096
        for _ in range(2):
            buf = np.empty(cap, np.uint8)                              # (no zero fill: the library writes what it reports)
            n = int(self.L.sq_write_blocks(self.h, C.byref(d), _ptr(buf), cap, _ptr(off), _ptr(skipped)))
            if n >= 0:
                break
            if n > -16:
                _lib.check(int(n))
            cap = -n
        text = str(memoryview(buf)[:n], "ascii")
        if not skipped[:self.nseq].any():
            return _Blocks(text, off)
        o = off.tolist()
        sk = skipped.tolist()
        return [None if sk[k] else text[o[k]:o[k + 1]] for k in range(self.nseq)]

    def pack_all(self, copy=False):
        """(uint8 array, int64 offsets[nseq + 1]): the packed results of every record (sq_result_view, else sq_result_pack_all)
        -- the payload of the multi-GPU result gather.  The array is a view of a buffer the batch reuses (copy=True: of the
        batch's own pack buffer, never of the library's pinned one)."""
        # the records where the device tail wrote them (the library's pinned buffer): no copy.  The views are valid until the
        # batch folds again or closes -- every caller below turns them into bytes / tuples before that
        pb, po, nb = C.c_void_p(), C.c_void_p(), C.c_int64()
        rc = self.L.sq_result_view(self.h, C.byref(pb), C.byref(po), C.byref(nb))
        if rc == 0 and not copy:
            n = max(int(nb.value), 0)
            buf = np.ctypeslib.as_array((C.c_uint8 * max(n, 1)).from_address(pb.value))[:n]
            off = np.ctypeslib.as_array((C.c_int64 * (self.nseq + 1)).from_address(po.value))
            return buf, off
        if rc < 0:
            _lib.check(rc)
        nbytes = int(self.L.sq_result_pack_all_size(self.h))
        # the batch keeps its pack buffer (fresh pages for tens of MB per call cost more than the packing itself); the
        # returned view is valid until the next pack_all of this batch
        buf = getattr(self, "_packbuf", None)
        if buf is None or buf.size < max(nbytes, 8):
            buf = self._packbuf = np.empty(max(nbytes, 8) + (max(nbytes, 8) >> 3), np.uint8)
        off = np.zeros(self.nseq + 1, np.int64)
        _lib.check(self.L.sq_result_pack_all(self.h, _ptr(buf), nbytes, _ptr(off)))
        return buf[:nbytes], off

    def detach_packed(self):
        """The packed results of every record as read-only memoryviews of the library's pinned buffer, which leaves the batch
        with them (sq_result_detach): no copy; the buffer goes back to the library when the last view is dropped.  None when
        the records are not in that form (the host tail ran): pack_all then."""
        pb, po, nb = C.c_void_p(), C.c_void_p(), C.c_int64()
        if self.L.sq_result_view(self.h, C.byref(pb), C.byref(po), C.byref(nb)) != 0:
            return None
        off = np.ctypeslib.as_array((C.c_int64 * (self.nseq + 1)).from_address(po.value)).tolist()
        if self.L.sq_result_detach(self.h, C.byref(pb), C.byref(nb)) != 0:
            return None
        arr = (C.c_uint8 * max(int(nb.value), 1)).from_address(pb.value)
        arr._owner = _PinnedOwner(self.L, pb.value)
        mv = memoryview(arr).toreadonly()
        return [mv[off[k]:off[k + 1]] for k in range(self.nseq)]

    def evals(self, k):
        return int(self.L.sq_result_evals(self.h, k))

    # -- measurement
    def profile(self, on=True):
        self.L.sq_profile_enable(self.h, int(on))

    def profile_reset(self):
        self.L.sq_profile_reset(self.h)

    def mwm_counters(self):
        """Blossom kernel work since the last profile_reset: dict(graphs, passes, and the critical graph's
        max_passes, max_events, n, m) -- sq_profile_counters."""
        out = (C.c_int64 * 6)()
        _lib.check(self.L.sq_profile_counters(self.h, 4, out))
        return dict(zip(("graphs", "passes", "max_passes", "max_events", "n", "m"), [int(x) for x in out]))

    def profile_get(self, kernel):
        ms, n, by = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(self.L.sq_profile_get(self.h, kernel, C.byref(ms), C.byref(n), C.byref(by)))
        return ms.value, n.value, by.value


class _PinnedOwner:
    """Returns a detached pinned buffer to the library when the last view of it is gone (Batch.detach_packed)."""

    def __init__(self, L, ptr):
        self.L, self.ptr = L, ptr

    def __del__(self):
        try:
            self.L.sq_buffer_release(C.c_void_p(self.ptr))
        except Exception:                                            # (interpreter shutdown)
            pass


def unpack_result(p, buf, base=0):
    """(SQRNdbnseq return tuple, reference scores or None) of ONE packed result record (sq_result_pack layout, see
    include/squarna_hip.h) of the prepared record `p`: any rank can decode a record another rank folded."""
    ns, n, has_ref, evals = _HDR.unpack_from(buf, base)
    met = _MET.unpack_from(buf, base + 32)
    o = base + 160
    scores = struct.unpack_from("<%dd" % (3 * ns), buf, o); o += 24 * ns
    masks = struct.unpack_from("<%dQ" % ns, buf, o); o += 8 * ns
    seq = p.seq
    if True:
        lev = np.frombuffer(buf, np.int16, (ns + 1) * n, o).reshape(ns + 1, n)
        # levels -> bracket characters for all rows at once (code-point table), gap columns and separators
        # re-inserted with array assignments (SQRNdbnseq.py:1239-1246)
        cp = _LEVEL_CP[np.clip(lev, -_NBR - 1, _NBR + 1) + (_NBR + 1)]                     # (ns+1, n) uint32
        if p.gapidx or p.sepidx:
            full = np.full((ns + 1, len(seq)), ord('.'), np.uint32)
            keep = np.ones(len(seq), bool)
            keep[p.gapidx] = False
            full[:, keep] = cp
            for i in p.sepidx:
                full[:, i] = ord(seq[i])
            cp = full
        width = cp.shape[1]
        text = cp.tobytes().decode('utf-32-le')
    cons = text[:width]
    preds = []
    for t in range(ns):
        m = masks[t]
        preds.append((text[(t + 1) * width:(t + 2) * width], scores[3 * t:3 * t + 3],
                      list(_MASK_IDS[m]) if m < 16 else [q for q in range(64) if (m >> q) & 1]))
    if has_ref:
        consres = _metrics(met[:6])
        res = _metrics(met[6:12]) + [int(met[12])]
        return (cons, preds, consres, res), tuple(met[13:16])
    return (cons, preds, [np.nan] * 6, [np.nan] * 7), None



def _metrics(m):
    """[TP, FP, FN, FS, PR, RC] with the reference's int/float types: a ratio whose
    denominator is empty is the int 1, everything else a rounded float
    (SQRNdbnseq.py:1256-1258,1273-1275)."""
    tp, fp, fn = int(m[0]), int(m[1]), int(m[2])
    fs = float(m[3]) if 2 * tp + fp + fn else 1
    pr = float(m[4]) if tp + fp else 1
    rc = float(m[5]) if tp + fn else 1
    return [tp, fp, fn, fs, pr, rc]


def fold_concurrently(batches, reps=1, **opts):
    """Fold several batches at the same time (sq_fold_concurrent: one host thread per batch inside the library):
    while one batch's host code books a round, the kernels of the others keep the GPU busy.  Batches are
    independent, so the results are the ones of folding them one after the other.  reps > 1: every batch is folded
    that many times back to back without a barrier between the repetitions (sq_fold_concurrent_n)."""
    args = [b._fold_args(**opts) for b in batches]
    n = len(batches)
    hs = (C.c_void_p * n)(*[b.h for b in batches])
    offs = (C.c_void_p * n)(*[a[1].ctypes.data for a in args])
    rps = (C.c_void_p * n)(*[a[2].ctypes.data for a in args])
    has = (C.c_void_p * n)(*[a[3].ctypes.data for a in args])
    if reps > 1:
        _lib.check(batches[0].L.sq_fold_concurrent_n(hs, n, C.byref(args[0][0]), offs, rps, has, int(reps)))
    else:
        _lib.check(batches[0].L.sq_fold_concurrent(hs, n, C.byref(args[0][0]), offs, rps, has))


def vienna_bpp(shortseq, reacts, M=1.8, B=-0.6):
    """Base-pair probability matrix of one sequence exactly as the reference obtains it (SQRNdbnseq.py:342-364):
    ViennaRNA's partition function (with SHAPE pseudo-energies when reactivities are given), rescaled once when all
    probabilities vanish.  Host-side third-party code, outside the accelerated path; None when max(bppm) == 0."""
    try:
        import RNA
    except ImportError:
        raise RuntimeError("this configuration has bpp != 0 paramsets, which need ViennaRNA's Python module `RNA` "
                           "on the host (SQRNdbnseq.py:341-364); it is not installed. Use a config without bpp "
                           "(e.g. c=nobpp) or install ViennaRNA.") from None
    fc = RNA.fold_compound(''.join(ch if ch not in SEPS and ord(ch) <= 127 else 'N' for ch in shortseq))
    if reacts is not None and set(reacts) != {0.5}:
        fc.sc_add_SHAPE_deigan(ProcessReacts(list(reacts), reverse=True, M=M, B=B), m=M, b=B)
    fc.pf()
    bppm = np.array(fc.bpp())[1:, 1:]
    if np.max(bppm) > 0:
        return bppm
    (ss, mfe) = fc.mfe()
    fc.exp_params_rescale(mfe)
    fc.pf()
    bppm = np.array(fc.bpp())[1:, 1:]
    return bppm if np.max(bppm) > 0 else None


_bpp_provider = vienna_bpp


def set_bpp_provider(fn):
    """Replace the source of base-pair probabilities (fn(shortseq, reacts, M, B) -> N x N array or None)."""
    global _bpp_provider
    old, _bpp_provider = _bpp_provider, (fn or vienna_bpp)
    return old


def bpp_terms(prepared, psets, M=1.8, B=-0.6):
    """Per job (record-major, paramset-minor) the dense term the fill applies for bpp != 0 paramsets:
    (bppm / max(bppm)) ** |bpp|  (SQRNdbnseq.py:350-354), or None.  Returns None when no paramset needs one."""
    if not any(ps.get("bpp", 0) for pl in psets for ps in pl):
        return None
    out = []
    for p, pl in zip(prepared, psets):
        bppm = None
        if any(ps.get("bpp", 0) for ps in pl):
            bppm = _bpp_provider(p.shortseq, p.shortreacts if p.shortreacts is not None else [0.5] * len(p.shortseq), M, B)    # once per sequence
            if bppm is not None:
                bppm = np.asarray(bppm, dtype=np.float64)
        for ps in pl:
            power = ps.get("bpp", 0)
            if power and bppm is not None:
                out.append(np.ascontiguousarray((bppm / np.max(bppm)) ** abs(power)))
            else:
                out.append(None)
    return out


def _free_device_bytes():
    """Free device memory as a new workspace sees it: what the driver reports plus what torch's caching allocator holds without
    using it (the workspaces of earlier batches: a call that sized its batches by the driver's figure alone got smaller ones
    than the call before it, whose workspace it could have had back)."""
    import torch
    return int(torch.cuda.mem_get_info()[0]) + max(0, int(torch.cuda.memory_reserved()) - int(torch.cuda.memory_allocated()))


def _kept_bytes_per_slot(maxn):
    """Bytes per structure slot of the lists a pool's structures hand to their children (SQ_BATCH_POOL_LISTS, sequences of
    257-1,024 nt): SQ_KEPT_PPS pages of 6 KB per generation (default 3 at 500 nt, growing with the square of the length), a row
    of 256 page numbers, a count."""
    if not 256 < maxn <= 1024 or "SQ_NO_POOL_KEPT" in os.environ:
        return 0
    pps = float(os.environ["SQ_KEPT_PPS"]) if "SQ_KEPT_PPS" in os.environ else max(1.0, 3.0 * (maxn / 500.0) ** 2)
    return int(2 * (pps * 6144 + 1028))


def pool_slot_cap(maxn, want=None):
    """Most structure slots a batch of sequences up to maxn nt should get: a slot of the device pools (sq_pool.hip) costs
    ~56 bytes per nucleotide; all slots stay within a sixth of the free device memory (at most 2 Mi).
    want: the slots the caller is about to ask for -- when the driver's figure alone grants them, the allocator's idle blocks are
    not counted (torch.cuda.memory_reserved() walks the allocator's statistics: 0.25 ms of a 5-ms Predict() on SRtest150)."""
    import torch
    per_slot = 8 * (maxn + 34) + 72 * (maxn // 2 + 1) + 2600 + _kept_bytes_per_slot(maxn)
    if not torch.cuda.is_available():
        return int(max(4096, min((16 << 30) // 6 // per_slot, 2 << 20)))
    cap = int(max(4096, min(int(torch.cuda.mem_get_info()[0]) // 6 // per_slot, 2 << 20)))
    if want is not None and want <= cap:
        return cap
    return int(max(4096, min(_free_device_bytes() // 6 // per_slot, 2 << 20)))


def pool_slots_wanted(ngreedy, poollim, n=None):
    """Structure slots for `ngreedy` greedy jobs of an n-nt sequence under pools wider than 1: the device pools hold a
    whole generation of every job's pool.  A pool overshoots poollim before the stopper (SQRNdbnseq.py:1147) holds it (it
    grows by a factor of 1.5 to 3.5 per round), and a short sequence never fills it: measured on random sequences the
    generations peak at ~1.75e-5 n^3 structures per job (6 at 20-120 nt, 485 at 300 nt) until poollim bounds them (130
    at 1000 nt under poollim 100).  Twice that, and at least 16."""
    p = min(int(poollim), 1024)
    per_job = min(3 * p, p + 512)
    if n is not None:
        per_job = min(per_job, max(16, int(4e-5 * float(n) ** 3)))
    return int(ngreedy) * per_job


def _shared_weights(records):
    """The records are an alignment's rows weighted by ONE device matrix (alignment step 2)."""
    sm0 = records[0][5] if records and len(records[0]) > 5 else None
    return sm0 is not None and hasattr(sm0, "is_cuda") and sm0.is_cuda and all(len(r) > 5 and r[5] is sm0 for r in records)


def pool_slots_wanted_many(lengths, psets_per_record, poollim, rarely_branch=False):
    """pool_slots_wanted for every record of a batch (numpy array): the greedy-job count per distinct paramset list is
    counted once (the records of an input usually share one list), the per-length part is vectorised.
    rarely_branch: the rows of an alignment under paramsets whose range factor is 1.0 -- their pools branch only at exact
    ties that share a base (SQRNdbnseq.py:769-789), the weights of a stem matrix make those rare, and the library folds such
    jobs as chains first (sq_fold.hip): sixteen slots per job (a fold that outgrows them is repeated by the host loop)."""
    if rarely_branch and all(ps["suboptmin"] == 1.0 and ps["suboptmax"] == 1.0 for pl in {id(p): p for p in psets_per_record}.values()
                             for ps in pl if "G" in ps["algorithms"]):
        return np.array([16 * sum(1 for ps in pl if "G" in ps["algorithms"]) for pl in psets_per_record], np.int64)
    ng_of, ng = {}, np.empty(len(psets_per_record), np.int64)
    for k, pl in enumerate(psets_per_record):
        v = ng_of.get(id(pl))
        if v is None:
            v = ng_of[id(pl)] = sum(1 for ps in pl if "G" in ps["algorithms"])
        ng[k] = v
    p = min(int(poollim), 1024)
    n = np.asarray(lengths, np.float64)
    per_job = np.minimum(min(3 * p, p + 512), np.maximum(16, (4e-5 * n ** 3).astype(np.int64)))
    return ng * per_job


class HipEngine:
    """Default engine: everything on the GPU through libsquarna_hip.so."""
    name = "hip"
    writes_blocks = True          # fold_records(..., _blocks=cfg): the library forms Predict's output blocks

    def __init__(self, max_structs=0, cand_per_nt=0):
        self.max_structs = max_structs
        self.cand_per_nt = cand_per_nt
        #: per record of the last fold_records call: ScoreStruct of its known structure (C tail) or None
        self.last_ref_scores = None

    def fold_records(self, records, **opts):
        """records: list of (seq, reacts, restraints, dbn, paramsets, stemmatrix);
        returns the list of SQRNdbnseq return tuples, in order."""
        # Building tens of thousands of small containers (Prepared records, result tuples) with the cyclic collector on
        # costs ~10 us per record in generation scans of objects that hold no cycles: 220 of 340 ms for 10,000 records.
        import gc
        was = gc.isenabled()
        gc.disable()
        try:
            return self._fold_records(records, **opts)
        finally:
            if was:
                gc.enable()

    def fold_records_packed(self, records, **opts):
        """Like fold_records, but every record's result stays in the library's packed form (bytes, sq_result_pack layout):
        the payload ranks exchange (parallel.py).  `unpack_result(Prepared(...), blob)` decodes one on any rank."""
        return self.fold_records(records, _packed=True, **opts)

    def _fold_records(self, records, **opts):
        """`keep`: only the first `keep` structures of every record are fetched (Predict prints outplim of them)."""
        poollim = opts.get("poollim", 1000)
        if not self.max_structs and poollim > 1 and len(records) > 1:
            # wide pools: as many records per batch as the device pools have slots for (a fold that outgrows them is
            # repeated by the library's host loop -- correct, but several times slower)
            lens = [len(r[0]) for r in records]
            per_rec = pool_slots_wanted_many(lens, [r[4] for r in records], poollim, rarely_branch=_shared_weights(records))
            cap = pool_slot_cap(max(lens), want=int(per_rec.sum()))
            if int(per_rec.sum()) > cap:
                return self._fold_in_sub_batches(records, per_rec.tolist(), cap, opts)
        out, refs = self._fold_groups([records], [None], opts)
        self.last_ref_scores = refs[0]
        return out[0]

    def _fold_in_sub_batches(self, records, per_rec, cap, opts):
        """Consecutive sub-batches sized to the device-pool slots.  What the pools of the first one really reached scales
        the estimate for the rest (a fold weighted by an alignment's stem matrix keeps one or two structures per job).
        (Measured: folding the sub-batches two at a time on streams of their own gains nothing -- 512 x 5000-column
        alignment 10.5 s either way, 3000 x 300 nt with pools of a thousand 2.1 s: these folds are bound by the scoring
        kernel, not by gaps between rounds -- and costs the second batch's memory.)"""
        # dense per-job matrices (alignment step 2: N x N fp64 + fp32 per job) bound a sub-batch as well: 32 GB of them
        # (allocating and touching 100 GB per batch costs more than the larger rounds save)
        # (not the rows of an alignment weighted by ONE device matrix: the kernels read it through the gap map, no slice exists --
        # unless SQ_MUL_GATHER=1 asks for the round-3 form)
        direct = _shared_weights(records) and "SQ_MUL_GATHER" not in os.environ
        dense = [12.0 * len(r[0]) ** 2 * len(r[4]) if len(r) > 5 and r[5] is not None and not direct else 0.0 for r in records]
        dense_cap = float(os.environ.get("SQ_DENSE_GB", "32")) * 1e9
        if sum(dense) > dense_cap:
            # sub-batches of equal weight (a last one of a few records would run its rounds on a mostly empty chip)
            dense_cap = sum(dense) / np.ceil(sum(dense) / dense_cap) + max(dense)

        def next_group(lo, scale, cap):
            hi, g, gb = lo, 0.0, 0.0
            while hi < len(records) and (hi == lo or (g + max(16.0, per_rec[hi] * scale) <= cap and gb + dense[hi] <= dense_cap)):
                g += max(16.0, per_rec[hi] * scale)
                gb += dense[hi]
                hi += 1
            return hi, int(g)

        # SQ_ENGINE_SUBLANES=2: two sub-batches at a time, each from a thread of its own (a batch's fold releases the GIL).
        # Measured again in round 6 (1,000 records of 500 nt, pools of a thousand on kept lists): 566 ms against 398 one after
        # the other -- each lane's batches get half of the slots, so there are twice as many, and every one of them waits ~100 ms
        # for the 500-vertex Edmonds graphs of its records however few they are.  Off by default.
        # Not the rows weighted by ONE device matrix: they share that tensor on the caller's stream.
        lanes = max(1, int(os.environ.get("SQ_ENGINE_SUBLANES", "1")))
        if direct or sum(dense) > 0 or opts.get("_blocks"):
            lanes = 1
        n = len(records)
        out, refs = [None] * n, [None] * n
        # (the scale the pools of an earlier call reached under the same paramsets, pool limit and lengths: the estimate is for the
        # widest pools a configuration can have -- 500nobpp at 500 nt reaches an eighth of it -- and a first sub-batch sized by it
        # was a quarter of the records, with the full wait for its Edmonds graphs)
        memo = self.__dict__.setdefault("_pool_scale", {})
        mkey = (tuple(tuple(sorted((k, str(v)) for k, v in ps.items())) for ps in records[0][4]),       # (the paramsets by content: an id() is reused)
                int(opts.get("poollim", 1000)), max(len(r[0]) for r in records) // 64)
        state = {"lo": 0, "scale": memo.get(mkey, 1.0), "first": True, "err": None, "driver": 0, "peak": 0}
        cap_lane = cap // lanes if lanes > 1 else cap
        per_arr = np.asarray(per_rec, np.float64)
        import threading
        lock = threading.Lock()

        def take():
            with lock:
                if state["err"] is not None or state["lo"] >= n:
                    return None
                lo = state["lo"]
                # (sub-batches of equal weight: what is left goes into as few batches as the slots allow, each with the same share --
                # a full one and a remainder of a fifth left the remainder its own wait for the Edmonds graphs on a mostly empty chip)
                left = float(np.maximum(16.0, per_arr[lo:] * state["scale"]).sum())
                cap_now = cap_lane
                if left > cap_lane:
                    cap_now = min(cap_lane, left / np.ceil(left / cap_lane) + float(per_arr[lo:].max()) * state["scale"] + 16.0)
                hi, g = next_group(lo, state["scale"], cap_now)
                state["lo"] = hi
                return lo, hi, g

        def work(k):
            import torch
            # (a lane keeps its stream for the engine's lifetime: torch's caching allocator hands a freed workspace only to the
            # stream it was allocated on -- a new stream per call allocated tens of GB anew every time)
            if lanes > 1:
                streams = self.__dict__.setdefault("_lane_streams", {})
                if k not in streams:
                    streams[k] = torch.cuda.Stream()
            ctx = torch.cuda.stream(streams[k]) if lanes > 1 else contextlib.nullcontext()
            try:
                with ctx:
                    while True:
                        job = take()
                        if job is None:
                            return
                        a, b, g = job
                        info = {}
                        o, r = self._fold_groups([records[a:b]], [g], opts, info=info, inflight=lanes)
                        out[a:b] = o[0]
                        refs[a:b] = r[0]
                        with lock:
                            state["driver"] = max(state["driver"], info["driver"])
                            state["peak"] = max(state["peak"], info["peak"])
                            if state["first"] and info["peak"] > 0 and info["driver"] == 2:
                                state["scale"] = min(state["scale"], max(2.0 * info["peak"] / max(sum(per_rec[a:b]), 1), 1e-4))
                                state["first"] = False
                            elif info["driver"] == 3:
                                state["scale"] = min(1.0, state["scale"] * 4)
                            memo[mkey] = state["scale"]
            except BaseException as e:                                   # (re-raised on the caller's thread)
                with lock:
                    if state["err"] is None:
                        state["err"] = e

        if lanes == 1:
            work(0)
        else:
            ths = [threading.Thread(target=work, args=(k,)) for k in range(lanes)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        if state["err"] is not None:
            raise state["err"]
        self.last_fold_driver, self.last_fold_peak = state["driver"], state["peak"]
        self.last_ref_scores = refs
        return out

    def _make_batch(self, records, slots_hint, opts, grow=(1, 1)):
        """(Batch, fold options) for these records.  opts: fold_records' keyword arguments (not modified).  grow: factors on
        the candidate capacity per nucleotide and on the structure slots (a fold that outgrew them is repeated)."""
        opts = dict(opts)
        opts.pop("_packed", None)
        opts.pop("_blocks", None)
        interchainonly = opts.pop("interchainonly", False)
        keep = opts.pop("keep", None)
        M, B = opts.pop("M", 1.8), opts.pop("B", -0.6)
        prepared = [Prepared(r[0], r[1], r[2], r[3]) for r in records]
        psets = [r[4] for r in records]
        bpp = bpp_terms(prepared, psets, M, B)
        mul = None
        mul_shared = None
        sm0 = records[0][5] if len(records[0]) > 5 else None
        if sm0 is not None and hasattr(sm0, "is_cuda") and sm0.is_cuda and all(len(r) > 5 and r[5] is sm0 for r in records):
            # alignment step 2 with the stem matrix still on the GPU: no per-record copies (Batch(mul_shared=...))
            # max |M| of the matrix, remembered through a WEAK reference: the engine is process-wide and must not keep an
            # L x L device matrix (200 MB at L = 5000) alive after the alignment that owns it has ended
            ref, val = getattr(self, "_sm_maxabs", (None, None))
            if ref is None or ref() is not sm0:
                val = float(sm0.abs().max().item())
                self._sm_maxabs = (weakref.ref(sm0), val)
            mul_shared = (sm0, [np.flatnonzero(~gap_mask(r[0])).astype(np.int32) for r in records], val)
        elif any(len(r) > 5 and r[5] is not None for r in records):
            mul = []
            for r, p in zip(records, prepared):
                sm = r[5] if len(r) > 5 else None
                if sm is not None and hasattr(sm, "is_cuda"):
                    sm = sm.cpu().numpy()
                if sm is not None:                                   # :1031-1034
                    sm = np.delete(np.delete(np.asarray(sm, dtype=np.float64), p.gapidx, 0), p.gapidx, 1)
                mul.extend([sm] * len(r[4]))
        # structure slots of the batch: the device-side pools / chained rounds hold every structure of a round at once, so
        # the default grows with the number of (sequence, paramset) jobs (a fold that still outgrows it is repeated by
        # the library's host loop)
        njobs = sum(len(pl) for pl in psets)
        max_structs = self.max_structs if self.max_structs else max(4096, min(4 * njobs, 262144))
        if not self.max_structs and opts.get("poollim", 1000) > 1:
            want = slots_hint if slots_hint else int(pool_slots_wanted_many(
                [len(p.shortseq) for p in prepared], psets, opts.get("poollim", 1000), rarely_branch=_shared_weights(records)).sum())
            max_structs = max(max_structs, min(want, pool_slot_cap(max(len(p.shortseq) for p in prepared), want=want)))
        cand = self.cand_per_nt
        if grow[0] > 1:
            # (the library sizes a structure's candidate records by max(cand_per_nt x N, its estimate for random sequences --
            # ~0.19 N^2 runs at minlen 1): the factor applies to the larger of the two
            nmax = max(len(p.shortseq) for p in prepared)
            runs = max(0.375 ** (max(1.0, float(np.ceil(ps["minlen"]))) - 1.0) for pl in psets for ps in pl)
            cand = int(max(cand, 32, 0.117 * 1.6 * nmax * runs) * grow[0]) + 1
        b = Batch(prepared, psets, interchainonly=interchainonly, mul=mul, fp32=False, bpp=bpp,
                  max_structs=max_structs * grow[1], cand_per_nt=cand, mul_shared=mul_shared,
                  pool_lists=opts.get("poollim", 1000) > 1 and mul_shared is None)
        b.limit_results(keep)
        return b, opts

    def _fits_device(self, records, slots_hint, opts, grow):
        """Whether a batch of these records with its capacities grown by `grow` still fits the free device memory (half of
        it: the fold's scratch and the caller's tensors live there too)."""
        import torch
        n = max((len(r[0]) for r in records), default=1)
        njobs = sum(len(r[4]) for r in records)
        structs = (self.max_structs if self.max_structs else max(4096, min(4 * njobs, 262144))) * grow[1]
        per_slot = 8 * (n + 34) + 72 * (n // 2 + 1) + 2600 + (_kept_bytes_per_slot(n) if opts.get("poollim", 1000) > 1 else 0)   # (pool_slot_cap's figure)
        cand = max(self.cand_per_nt, 32) * grow[0] * n * 32.0              # candidate records of a structure, 32 bytes each
        free = _free_device_bytes()
        return structs * per_slot + min(structs, 4 * njobs) * cand <= free // 2

    def _fold_groups(self, groups, hints, opts, info=None, inflight=1):
        """Folds every group of records as one batch, all of them at the same time; ([results], [reference scores]) per
        group.  One group with SQ_ENGINE_LANES=2 and a big input: cut into two concurrent batches (for one-shot calls
        the second batch's set-up costs more than the overlap saves, so that is opt-in)."""
        lanes = int(os.environ.get("SQ_ENGINE_LANES", "1"))
        back = None
        if len(groups) == 1 and lanes >= 2 and len(groups[0]) >= 256 and hints[0] is None:
            recs = groups[0]
            cost = [float(len(r[0])) ** 2 * len(r[4]) for r in recs]
            if sum(cost) >= 1e8 and not any(len(r) > 5 and r[5] is not None and hasattr(r[5], "is_cuda") for r in recs):
                from .parallel import lpt_partition
                back = [p for p in lpt_partition(cost, 2) if p]
                groups, hints = [[recs[k] for k in idx] for idx in back], [None] * len(back)
        batches = []
        try:
            fold_opts = None
            import torch
            if len(groups) > 1:
                torch.cuda.current_stream().synchronize()            # (inputs made on this stream, e.g. the shared stem matrix)
            # The reference has no capacities (SQRNdbnseq.py:427-495 builds Python lists): a fold that outgrows what its batch
            # was created with -- the candidate records of a structure are sized for random sequences, GC-only or minlen = 1
            # inputs hold several times as many runs -- is repeated with a larger batch; the caller never sees the error.
            grow = [1, 1]
            for attempt in range(8):
                for q, (recs, hint) in enumerate(zip(groups, hints)):
                    # concurrent batches on streams of their own: kernels of one fill the gaps of the other
                    ctx = torch.cuda.stream(torch.cuda.Stream()) if q > 0 else contextlib.nullcontext()
                    with ctx:
                        b, fold_opts = self._make_batch(recs, hint, opts, tuple(grow))
                    batches.append(b)
                try:
                    if len(batches) == 1:
                        if inflight > 1:
                            batches[0].set_inflight(inflight)            # (sub-batches folded side by side from the engine's threads)
                        batches[0].fold(**fold_opts)
                    else:
                        fold_concurrently(batches, **fold_opts)
                    break
                except _lib.CapacityError as e:
                    # which capacity, from the library's own code (sq_last_capacity): candidate records and a round's output
                    # records grow with cand_per_nt, the log of final structures with max_structs; a fixed limit is raised.
                    # The growth stops where the batch would no longer fit the device: the ORIGINAL error is raised then, not
                    # an allocation failure
                    which = {_lib.CAP_CANDIDATES: 0, _lib.CAP_OUTPUT: 0, _lib.CAP_STRUCTS: 1}.get(e.kind)
                    if which is None or attempt == 7:
                        raise
                    grow[which] *= 4
                    if not all(self._fits_device(recs, hint, opts, tuple(grow)) for recs, hint in zip(groups, hints)):
                        raise
                    self.capacity_retries = getattr(self, "capacity_retries", 0) + 1
                    for b in batches:
                        b.close()
                    batches = []
            if info is not None:
                info["driver"], info["peak"] = max(b.fold_driver for b in batches), batches[0].fold_peak_structs
            else:
                self.last_fold_driver = max(b.fold_driver for b in batches)
                self.last_fold_peak = batches[0].fold_peak_structs
            if opts.get("_packed"):
                res = []
                for b in batches:
                    views = b.detach_packed() if "SQ_NO_DETACH" not in os.environ else None
                    if views is not None:                            # (no copy: the records stay where the device wrote them)
                        res.append([(v, None) for v in views])
                        continue
                    buf, off = b.pack_all()
                    res.append([(buf[off[k]:off[k + 1]].tobytes(), None) for k in range(b.nseq)])
            elif opts.get("_blocks"):
                # Predict's printing path: the library writes the blocks; a record it leaves out (or a batch whose tail ran
                # on the host) comes back as its result tuple and the caller formats it.  Records carry their block fields
                # behind the fold's: (..., name, encoded reactivity line, index of their paramset-name list)
                cfg = opts["_blocks"]
                res = []
                for b, recs in zip(batches, groups):
                    texts = b.write_blocks([r[6] for r in recs], [r[0] for r in recs], [r[7] for r in recs],
                                           [r[2] for r in recs], [r[3] for r in recs], [r[8] for r in recs],
                                           cfg["psnames"], cfg["conslim"], cfg["outplim"])
                    if isinstance(texts, _Blocks):
                        res.append(_BlockRun(texts))
                    elif texts is None or any(t is None for t in texts):
                        full = b.results_all()
                        res.append([(("text", texts[k]) if texts and texts[k] is not None else ("pred", full[k]), None) for k in range(b.nseq)])
                    else:
                        res.append([(("text", t), None) for t in texts])
            else:
                res = [b.results_all() for b in batches]
        finally:
            for b in batches:
                b.close()
        if any(isinstance(both, _BlockRun) for both in res):
            # (whole-text results stay whole when there is one batch; several batches / lanes: per-record entries)
            if len(res) == 1 and back is None:
                return [res[0]], [[None] * len(res[0].blocks)]
            res = [[(("text", t), None) for t in both.blocks] if isinstance(both, _BlockRun) else both for both in res]
        outs, refs = [[r[0] for r in both] for both in res], [[r[1] for r in both] for both in res]
        if back is not None:                                         # undo the two-lane cut
            n = sum(len(idx) for idx in back)
            o, rf = [None] * n, [None] * n
            for idx, oo, rr in zip(back, outs, refs):
                for k, x, y in zip(idx, oo, rr):
                    o[k], rf[k] = x, y
            return [o], [rf]
        return outs, refs

    def yield_stems(self, records, bpweights, minlen, minbpscore, interchainonly=False):
        """Alignment step 1 (SQRNdbnali.py:60-108): for every (seq, reacts, restraints) the stems of
        the gap-free sequence in emission order.  Returns [(shortseq, [(i, j, len, score), ...])]."""
        ps = dict(bpweights=bpweights, bpp=0, algorithms={"G"}, suboptmax=1.0, suboptmin=1.0, suboptsteps=1.0,
                  minlen=minlen, minbpscore=minbpscore, minfinscorefactor=1.0, bracketweight=-2.0, distcoef=0.09,
                  orderpenalty=1.0, loopbonus=0.125, maxstemnum=1e6)
        prepared = []
        for seq, reacts, restraints in records:
            p = Prepared(seq, reacts if reacts else None, restraints, None)
            if not reacts:
                p.shortreacts = [0.5] * len(p.shortseq)                # YieldStems passes reacts=None (:83)
                p.plain_reacts = True
            prepared.append(p)
        out, cap = [], 1 << 21
        est = [int(0.25 * len(p.shortseq) ** 2 * 0.375 ** (max(minlen, 1) - 1)) + 256 for p in prepared]
        lo = 0
        while lo < len(prepared):                                      # chunks sized to the stem buffer
            hi, tot = lo, 0
            while hi < len(prepared) and (hi == lo or tot + est[hi] <= cap):
                tot += est[hi]
                hi += 1
            chunk = prepared[lo:hi]
            with Batch(chunk, [[ps]] * len(chunk), interchainonly=interchainonly, max_structs=self.max_structs,
                       cand_per_nt=max(self.cand_per_nt, 64), fp32=False) as b:
                res = b.optimal(list(range(len(chunk))), [[] for _ in chunk], mode=1, out_cap=max(cap, tot),
                                as_array=True)
            out.extend((p.shortseq, st) for p, st in zip(chunk, res))
            lo = hi
        return out

    def stem_matrix(self, records, bpweights, minlen, minbpscore, interchainonly=False):
        """Alignment step 1 on the device (SQRNdbnali.py:211-242 without MatrixToDBNs): the L x L fp64 column
        matrix of stem scores over all (seq, reacts, restraints) records, as a torch tensor on the GPU.
        Sequences are applied in order, one scatter launch each, so every cell is summed in the reference's
        order; nothing but the gap maps crosses PCIe."""
        import torch
        ps = dict(bpweights=bpweights, bpp=0, algorithms={"G"}, suboptmax=1.0, suboptmin=1.0, suboptsteps=1.0,
                  minlen=minlen, minbpscore=minbpscore, minfinscorefactor=1.0, bracketweight=-2.0, distcoef=0.09,
                  orderpenalty=1.0, loopbonus=0.125, maxstemnum=1e6)
        Lcols = len(records[0][0])
        dev = torch.device("cuda", torch.cuda.current_device())
        matrix = torch.zeros((Lcols, Lcols), dtype=torch.float64, device=dev)
        r0 = records[0][2]
        if (not any(r[1] for r in records) and all((r[2] or None) == (r0 or None) for r in records)
                and all(len(r[0]) == Lcols for r in records) and "SQ_NO_PACKED_ROWS" not in os.environ):
            # the usual alignment: no per-row reactivities, one restraint line for every row (iteration 2) or none -- all rows
            # prepared at once as array code (PackedRows), chunks of rows sized like the per-row form below
            lo = 0
            while lo < len(records):
                hi, cells = lo, 0
                while hi < len(records) and (hi == lo or cells + Lcols ** 2 <= 16e9):
                    cells += Lcols ** 2
                    hi += 1
                pk = PackedRows([r[0] for r in records[lo:hi]], r0 or None)
                with Batch(pk, [[ps]] * (hi - lo), interchainonly=interchainonly, max_structs=self.max_structs,
                           cand_per_nt=max(self.cand_per_nt, 64), fp32=False) as b:
                    b.align_accumulate_packed(pk, matrix)
                    torch.cuda.synchronize(dev)
                lo = hi
            return matrix
        prepared, cols = [], []
        for seq, reacts, restraints in records:
            p = Prepared(seq, reacts if reacts else None, restraints, None)
            if not reacts:
                p.shortreacts = [0.5] * len(p.shortseq)                # YieldStems passes reacts=None (:83)
                p.plain_reacts = True
            prepared.append(p)
            cols.append(np.flatnonzero(~gap_mask(seq)).astype(np.int32))   # ReAlignDict (:20-37)
        # chunks of sequences sized to ~24 GB of bit matrices + candidates
        lo = 0
        while lo < len(prepared):
            hi, cells = lo, 0
            while hi < len(prepared) and (hi == lo or cells + len(prepared[hi].shortseq) ** 2 <= 16e9):
                cells += len(prepared[hi].shortseq) ** 2
                hi += 1
            chunk = prepared[lo:hi]
            with Batch(chunk, [[ps]] * len(chunk), interchainonly=interchainonly, max_structs=self.max_structs,
                       cand_per_nt=max(self.cand_per_nt, 64), fp32=False) as b:
                b.align_accumulate(list(range(len(chunk))), cols[lo:hi], matrix)
                torch.cuda.synchronize(dev)
            lo = hi
        return matrix

    def matrix_cells(self, matrix, threshold, minspan=4, sort=True):
        """(flat indices, values) of the upper cells >= threshold with span >= minspan of a device matrix,
        sorted by flat index unless sort=False (MatrixToDBNs' candidates, SQRNdbnali.py:127-148)."""
        import torch
        Lcols = int(matrix.shape[0])
        cap = 1 << 16
        stream = torch.cuda.current_stream(matrix.device)
        while True:                                                # result buffers are torch tensors (caller-owned)
            idx = torch.empty(cap, dtype=torch.int64, device=matrix.device)
            val = torch.empty(cap, dtype=torch.float64, device=matrix.device)
            cnt = torch.zeros(1, dtype=torch.int64, device=matrix.device)
            _lib.check(_lib.load().sq_colmatrix_select(C.c_void_p(matrix.data_ptr()), Lcols, float(threshold), int(minspan),
                                                       C.c_void_p(idx.data_ptr()), C.c_void_p(val.data_ptr()), cap,
                                                       C.c_void_p(cnt.data_ptr()), C.c_void_p(stream.cuda_stream)))
            n = int(cnt.item())
            if n <= cap:
                break
            cap = n
        idx, val = idx[:n].cpu().numpy(), val[:n].cpu().numpy()
        if not sort:
            return idx, val
        order = np.argsort(idx, kind="stable")
        return idx[order], val[order]

    def entropy(self, record, interchainonly=False):
        """Mean row entropy of the stem matrix under the FIRST paramset, as a string
        (SQRNdbnseq.py:520-545, 1087-1089); the stems come from the GPU scan (mode 1)."""
        seq, reacts, restraints, dbn, paramsets = record[:5]
        p = Prepared(seq, reacts, restraints, dbn)
        ps = paramsets[0]
        mul = None
        sm = record[5] if len(record) > 5 else None
        if sm is not None:                                           # alignment step 2: bpscorematrix * shortsmat
            mul = [np.delete(np.delete(np.asarray(sm, dtype=np.float64), p.gapidx, 0), p.gapidx, 1)]   # (:1031-1034,1084-1085)
        with Batch([p], [[ps]], interchainonly=interchainonly, max_structs=self.max_structs, mul=mul,
                   cand_per_nt=max(self.cand_per_nt, 64)) as b:
            stems = b.optimal([0], [[]], mode=1)[0]
        n = len(p.shortseq)
        sm = np.zeros((n, n))
        for i, j, ln, sc, _ in stems:
            for k in range(ln):
                sm[i + k, j - k] = sc
                sm[j - k, i + k] = sc
        ent = 0
        for i in range(n):
            row = sm[i, :]
            if row.sum():
                probs = [q for q in row / row.sum() if q]
                ent += sum(-(probs * np.log2(probs)))
        return str(round(ent / n, 3))


_engine = None


def get_engine():
    global _engine
    if _engine is None:
        _engine = HipEngine()
    return _engine


@contextlib.contextmanager
def use_engine(engine):
    """Temporarily install another engine (tests only)."""
    global _engine
    old = _engine
    _engine = engine
    try:
        yield engine
    finally:
        _engine = old
