"""ctypes bindings of libsquarna_hip.so (include/squarna_hip.h).

The HIP library is the ONLY compute path of this package: loading fails loudly
when the shared object is missing, and every entry point fails loudly when no
MI355X is visible.  There is no CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsquarna_hip.so")

ALPHABET = 32
ALGO_BITS = {"G": 1, "N": 2, "H": 4, "E": 8}

#: every symbol include/squarna_hip.h declares
SYMBOLS = ["sq_version", "sq_last_error", "sq_last_capacity", "sq_batch_workspace_bytes", "sq_batch_create",
           "sq_batch_destroy", "sq_bpmatrix_fill", "sq_bpmatrix_read", "sq_optimal_stems",
           "sq_fold", "sq_result_nstruct", "sq_result_consensus", "sq_result_struct",
           "sq_result_metrics", "sq_result_evals", "sq_result_pack_size", "sq_result_pack",
           "sq_result_pack_all_size", "sq_result_pack_all", "sq_result_view", "sq_result_detach", "sq_buffer_release", "sq_batch_set_inflight", "sq_result_dbn_all_size", "sq_result_dbn_all",
           "sq_profile_enable", "sq_profile_get", "sq_profile_reset", "sq_profile_counters", "sq_run_algos",
           "sq_align_accumulate", "sq_colmatrix_select", "sq_fold_concurrent", "sq_fold_concurrent_n", "sq_fold_driver", "sq_fold_paths", "sq_fold_peak_structs", "sq_result_limit",
           "sq_mwm_workspace_bytes", "sq_mwm", "sq_lsap_workspace_bytes", "sq_lsap",
           "sq_nussinov_workspace_bytes", "sq_nussinov", "sq_dbn_pairs", "sq_write_blocks", "sq_parse_default",
           "sq_host_cache_trim", "sq_align_first_fit"]

BATCH_NO_FP32 = 1
BATCH_POOL_LISTS = 2


class ParamSet(C.Structure):
    _fields_ = [("bpweight", C.c_double * (ALPHABET * ALPHABET)),
                ("inbps", C.c_uint8 * (ALPHABET * ALPHABET)),
                ("bpp", C.c_double),
                ("suboptmax", C.c_double), ("suboptmin", C.c_double), ("suboptsteps", C.c_double),
                ("minlen", C.c_double), ("minbpscore", C.c_double), ("minfinscorefactor", C.c_double),
                ("bracketweight", C.c_double), ("distcoef", C.c_double), ("orderpenalty", C.c_double),
                ("loopbonus", C.c_double), ("maxstemnum", C.c_double),
                ("algorithms", C.c_uint32), ("reserved", C.c_uint32)]


class BatchDesc(C.Structure):
    _fields_ = [("nseq", C.c_int32),
                ("seq_off", C.POINTER(C.c_int32)),
                ("codes", C.POINTER(C.c_uint8)),
                ("flags", C.POINTER(C.c_uint8)),
                ("reacts", C.POINTER(C.c_double)),
                ("rbp_off", C.POINTER(C.c_int32)),
                ("rbps", C.POINTER(C.c_int32)),
                ("npset", C.c_int32),
                ("psets", C.POINTER(ParamSet)),
                ("njobs", C.c_int32),
                ("job_seq", C.POINTER(C.c_int32)),
                ("job_pset", C.POINTER(C.c_int32)),
                ("ext_bool", C.POINTER(C.c_void_p)),
                ("ext_score", C.POINTER(C.c_void_p)),
                ("mul_score", C.POINTER(C.c_void_p)),
                ("bpp_term", C.POINTER(C.c_void_p)),
                ("interchainonly", C.c_int32),
                ("max_structs", C.c_int32),
                ("cand_per_nt", C.c_int32),
                ("batch_flags", C.c_int32),
                ("mul_matrix_dev", C.c_void_p),
                ("mul_L", C.c_int32),
                ("mul_cols", C.POINTER(C.c_int32)),
                ("mul_shared", C.POINTER(C.c_uint8)),
                ("mul_maxabs", C.c_double)]


class Stem(C.Structure):
    _fields_ = [("i", C.c_int32), ("j", C.c_int32), ("len", C.c_int32), ("reserved", C.c_int32),
                ("bpscore", C.c_double), ("finscore", C.c_double)]


class BlockDesc(C.Structure):
    _fields_ = [("nrec", C.c_int32), ("names", C.c_char_p), ("seqs", C.c_char_p), ("reacts", C.c_char_p),
                ("restr", C.c_char_p), ("refs", C.c_char_p), ("nameset", C.POINTER(C.c_int32)),
                ("psnames", C.POINTER(C.c_char_p)), ("nsets", C.c_int32), ("conslim", C.c_int32), ("outplim", C.c_int32)]


class FoldOpts(C.Structure):
    _fields_ = [("poollim", C.c_int32), ("conslim", C.c_int32), ("toplim", C.c_int32),
                ("hardrest", C.c_int32), ("rankbydiff", C.c_int32), ("rankby", C.c_int32 * 3),
                ("levellimit", C.c_int32), ("algos", C.c_uint32), ("priority_mask", C.c_uint64)]


_lib = None


def load():
    """Load the C-ABI library (building nothing: see squarna_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsquarna_hip.so is missing (%s). Build it with `python -m squarna_amd.build` "
            "(hipcc --offload-arch=gfx950). squarna_amd has no CPU fallback." % LIB_PATH)
    # torch first: the library's HIP calls must resolve to the ONE HIP runtime of the process, the one torch brings
    # (loaded the other way round, the library pulls in the system's libamdhip64 and torch then its own copy: the second
    # runtime finds no device -- "hipHostMalloc: no ROCm-capable device is detected" in build() + smoke() in one process)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.sq_last_error.restype = C.c_char_p
    L.sq_batch_destroy.restype = None
    L.sq_host_cache_trim.restype = C.c_longlong
    L.sq_align_first_fit.restype = C.c_int64
    L.sq_host_cache_trim.argtypes = []
    L.sq_result_evals.restype = C.c_int64
    L.sq_result_pack_size.restype = C.c_int64
    L.sq_result_pack_all_size.restype = C.c_int64
    L.sq_result_pack_all_size.argtypes = [C.c_void_p]
    L.sq_result_pack_all.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.sq_batch_set_inflight.argtypes = [C.c_void_p, C.c_int32]
    L.sq_result_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.sq_result_detach.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.sq_buffer_release.argtypes = [C.c_void_p]
    L.sq_buffer_release.restype = None
    L.sq_result_dbn_all_size.restype = C.c_int64
    L.sq_result_dbn_all_size.argtypes = [C.c_void_p]
    L.sq_result_dbn_all.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.sq_batch_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(BatchDesc), C.c_void_p, C.c_size_t, C.c_void_p]
    L.sq_batch_workspace_bytes.argtypes = [C.POINTER(BatchDesc), C.POINTER(C.c_size_t)]
    L.sq_batch_destroy.argtypes = [C.c_void_p]
    L.sq_bpmatrix_fill.argtypes = [C.c_void_p]
    L.sq_bpmatrix_read.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.sq_optimal_stems.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.sq_fold.argtypes = [C.c_void_p, C.POINTER(FoldOpts), C.c_void_p, C.c_void_p, C.c_void_p]
    L.sq_result_nstruct.argtypes = [C.c_void_p, C.c_int32]
    L.sq_result_consensus.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.sq_result_struct.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sq_result_metrics.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.sq_result_evals.argtypes = [C.c_void_p, C.c_int32]
    L.sq_result_pack_size.argtypes = [C.c_void_p, C.c_int32]
    L.sq_result_pack.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.sq_profile_enable.argtypes = [C.c_void_p, C.c_int32]
    L.sq_profile_get.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_double)]
    L.sq_profile_reset.argtypes = [C.c_void_p]
    L.sq_profile_counters.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.sq_run_algos.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                               C.c_void_p]
    L.sq_fold_concurrent.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FoldOpts), C.c_void_p, C.c_void_p, C.c_void_p]
    L.sq_result_limit.argtypes = [C.c_void_p, C.c_int32]
    L.sq_fold_peak_structs.argtypes = [C.c_void_p]
    L.sq_fold_peak_structs.restype = C.c_int64
    L.sq_fold_driver.argtypes = [C.c_void_p]
    L.sq_fold_paths.argtypes = [C.c_void_p]
    L.sq_dbn_pairs.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    L.sq_write_blocks.argtypes = [C.c_void_p, C.POINTER(BlockDesc), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.sq_write_blocks.restype = C.c_int64
    L.sq_fold_paths.restype = C.c_int32
    L.sq_fold_driver.restype = C.c_int32
    L.sq_fold_concurrent_n.argtypes = [C.c_void_p, C.c_int32, C.POINTER(FoldOpts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.sq_align_accumulate.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.sq_colmatrix_select.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p]
    L.sq_mwm_workspace_bytes.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    L.sq_mwm.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                         C.c_void_p, C.c_size_t, C.c_void_p]
    L.sq_lsap_workspace_bytes.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    L.sq_lsap.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_void_p, C.c_size_t, C.c_void_p]
    L.sq_nussinov_workspace_bytes.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    L.sq_nussinov.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    _lib = L
    return L


CAP_CANDIDATES, CAP_STRUCTS, CAP_OUTPUT, CAP_FIXED = 1, 2, 3, 4      # sq_last_capacity (include/squarna_hip.h)


class CapacityError(RuntimeError):
    """A capacity the batch was created with (candidate records per structure, the log of final structures) did not hold the
    fold: status -3 of the C ABI.  `kind` is sq_last_capacity()'s answer -- which capacity --, so that the engine can repeat
    the fold with a larger batch (engine.HipEngine._fold_groups) without reading the message."""

    def __init__(self, msg, kind=0):
        super().__init__(msg)
        self.kind = kind


def check(rc):
    if rc != 0:
        L = load()
        msg = "libsquarna_hip: %s (code %d)" % (L.sq_last_error().decode(), rc)
        if rc == -3:
            raise CapacityError(msg, int(L.sq_last_capacity()))
        raise RuntimeError(msg)
