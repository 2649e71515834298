/*
 * squarna_hip.h -- C ABI of libsquarna_hip.so, the MI355X (gfx950) folding core
 * that replaces the single-sequence hot path of febos/SQUARNA v3.2.2.
 *
 * The reference is pure Python and has no FFI of its own; the
 * boundary a maintainer would bind is the Python function surface of
 * src/SQUARNA/SQRNdbnseq.py.  Each entry point below names the reference
 * function(s) it replaces (file:line).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions across the boundary.
 *   - Every function returns 0 on success, <0 for an invalid argument / capacity
 *     problem, >0 for a HIP error code; sq_last_error() gives the message.
 *   - The caller owns all DEVICE memory: a single workspace (e.g. a torch uint8
 *     tensor) whose size sq_batch_workspace_bytes() reports.  The library never
 *     allocates or frees device memory; it keeps small pinned host staging
 *     buffers per batch.
 *   - All device work is enqueued on the caller's stream (hipStream_t passed as
 *     void*); entry points that return host results synchronise that stream.
 *   - Sequences are passed pre-encoded by the host layer (upper-cased, T->U,
 *     gap-free: SQRNdbnseq.py:1004,1023): code 0..25 = 'A'..'Z', 26 = ';',
 *     27 = '&', 28 = any other symbol.
 */
#ifndef SQUARNA_HIP_H
#define SQUARNA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQ_API __attribute__((visibility("default")))

#define SQ_ALPHABET 32
#define SQ_CODE_SEP1 26   /* ';' */
#define SQ_CODE_SEP2 27   /* '&' */
#define SQ_CODE_OTHER 28

/* position flags (ParseRestraints, SQRNdbnseq.py:370-376) */
#define SQ_FLAG_RX 1      /* '_' or '+': forced unpaired           */
#define SQ_FLAG_RLEFT 2   /* '/' : may not be the 3' partner (j)    */
#define SQ_FLAG_RRIGHT 4  /* '\': may not be the 5' partner (i)     */

/* algorithms bitmask (config key "algorithms", SQUARNA.py:62-63) */
#define SQ_ALGO_G 1
#define SQ_ALGO_N 2
#define SQ_ALGO_H 4
#define SQ_ALGO_E 8

/* One parameter set of a .conf file (SQUARNA.py:15-77; read at SQRNdbnseq.py:1050-1062). */
typedef struct sq_paramset {
    double bpweight[SQ_ALPHABET * SQ_ALPHABET]; /* weight of pair (a,b), both orientations filled (SQRNdbnseq.py:282-284) */
    uint8_t inbps[SQ_ALPHABET * SQ_ALPHABET];   /* 1 iff the pair is a key of bpweights (either orientation)          */
    double bpp;                /* != 0: the batch must carry the job's probability term in sq_batch_desc.bpp_term
                                  (computed by the caller from ViennaRNA, SQRNdbnseq.py:341-364); the fill applies it */
    double suboptmax, suboptmin, suboptsteps;
    double minlen, minbpscore, minfinscorefactor;
    double bracketweight, distcoef, orderpenalty, loopbonus;
    double maxstemnum;
    uint32_t algorithms;       /* SQ_ALGO_* */
    uint32_t reserved;
} sq_paramset;

/* Host-side description of a batch: sequences, per-position inputs, jobs.
 * A job = one (sequence, paramset) pair = one pass of SQRNdbnseq.py:1048-1199. */
typedef struct sq_batch_desc {
    int32_t nseq;
    const int32_t *seq_off;     /* [nseq+1] offsets into the per-position arrays                */
    const uint8_t *codes;       /* letter codes                                                  */
    const uint8_t *flags;       /* SQ_FLAG_*                                                     */
    const double *reacts;       /* per-position reactivities after ProcessReacts (SQRNdbnseq.py:32-59); NULL: 0.5 everywhere
                                 * (SQRNdbnseq.py:1013-1014: no record of the batch came with reactivities) */
    const int32_t *rbp_off;     /* [nseq+1] offsets (in pairs) into rbps                         */
    const int32_t *rbps;        /* restraint base pairs (v,w), v<w, gap-free coordinates; any number, a position in at most one */
    int32_t npset;
    const sq_paramset *psets;
    int32_t njobs;
    const int32_t *job_seq;     /* [njobs] */
    const int32_t *job_pset;    /* [njobs] */
    /* optional dense fp64 inputs, one N x N row-major matrix per job (NULL entries allowed, array may be NULL): */
    const double *const *ext_bool;   /* AnnotateStems/OptimalStems shims: caller-supplied bpboolmatrix   */
    const double *const *ext_score;  /* ... and bpscorematrix (SQRNdbnseq.py:427-428,792-797)            */
    const double *const *mul_score;  /* alignment step-2 weighting, bpscorematrix *= shortsmat (SQRNdbnseq.py:1084-1085) */
    const double *const *bpp_term;   /* paramsets with bpp != 0: (bppm / max(bppm)) ** |bpp| computed by the caller from
                                        ViennaRNA's base-pair probabilities; the fill applies scoremat *= term (bpp > 0)
                                        or scoremat += term (bpp < 0) (SQRNdbnseq.py:341-364).  A job takes either
                                        mul_score or bpp_term, not both.  NULL entry with bpp != 0: max(bppm) was 0,
                                        the matrix stays as it is (:350,360).                                        */
    int32_t interchainonly;     /* SQRNdbnseq.py:264-271,301 */
    int32_t max_structs;        /* structures evaluated per round chunk (0 = default 4096); also the structure
                                   slots of the device-side pools / chained rounds: a fold whose pools outgrow
                                   them is repeated by the library's host loop (same results, slower)           */
    int32_t cand_per_nt;        /* candidate capacity per structure = cand_per_nt * N (0 = 32)  */
    int32_t batch_flags;        /* SQ_BATCH_* */
    /* Alignment step 2 without one dense matrix per sequence (SQRNdbnseq.py:1031-1034 deletes the gap rows and columns
     * of the same L x L stem matrix for every sequence, :1084-1085 multiplies): ONE L x L row-major fp64 matrix in
     * DEVICE memory (the step-1 matrix never has to leave the GPU) and, per sequence, the alignment column of each of
     * its gap-free positions.  A job with mul_shared[j] != 0 takes bpscorematrix[a][b] *= M[cols[a]][cols[b]]: at
     * sq_batch_create the library copies M into the workspace in a diagonal-major layout (on the batch's stream: M must
     * be complete there; the caller's matrix is not read afterwards) and the kernels read a cell's weight from that copy
     * through `cols` where they need it -- no per-sequence N x N slice exists (SQ_MUL_GATHER=1 in the environment restores
     * the gathered slices of earlier versions: 8 N^2 bytes per job).  Such a job has no mul_score / bpp_term / caller
     * matrices.  All NULL / 0: not used. */
    const double *mul_matrix_dev;   /* device pointer */
    int32_t mul_L;
    const int32_t *mul_cols;        /* [seq_off[nseq]] host: column of every position (same offsets as codes)   */
    const uint8_t *mul_shared;      /* [njobs] host                                                              */
    double mul_maxabs;              /* max |M| (an upper bound is enough: it only widens the scan's fp32 margin) */
} sq_batch_desc;

/* No fp32 score matrices in the workspace (4 N^2 bytes per job saved): everything except
 * sq_bpmatrix_fill works, the fold path only needs the 1-bit-per-cell diagonal matrices. */
#define SQ_BATCH_NO_FP32 1
/* The batch will be folded with pools wider than one (poollim > 1) and holds sequences of 257-1,024 nt: the workspace
 * reserves pages for the lists every structure of such a pool hands to its children -- its runs with their bpscores and
 * the finalscores it knew (SQRNdbnseq.py:427-495 and :640-751 are then evaluated for what the child's new stem changed,
 * not for the whole structure; the results are the same).  Without the flag (or with SQ_NO_POOL_KEPT=1 in the environment)
 * such pools run the launched round kernels.  Pages per structure slot: SQ_KEPT_PPS (default 3 of 6 KB per generation at 500 nt, growing with the square of the length),
 * at most SQ_KEPT_GB gigabytes (default 48) in all. */
#define SQ_BATCH_POOL_LISTS 2

typedef struct sq_batch sq_batch;   /* opaque */

/* One stem as the reference's [bps, len, bpscore, finalscore] record
 * (SQRNdbnseq.py:417,747): bps = (i+k, j-k), k = 0..len-1. */
typedef struct sq_stem {
    int32_t i, j, len, reserved;
    double bpscore;
    double finscore;
} sq_stem;

/* Options of the SQRNdbnseq tail (SQRNdbnseq.py:973-980). */
typedef struct sq_fold_opts {
    int32_t poollim;        /* SQRNdbnseq.py:1147,1191 (1: rounds chained on the device; > 1: pools booked on the device) */
    int32_t conslim;        /* :1236 */
    int32_t toplim;         /* :1282 */
    int32_t hardrest;       /* :1226-1228 */
    int32_t rankbydiff;     /* :917-955 */
    int32_t rankby[3];      /* :907 */
    int32_t levellimit;     /* <0: default 3 - (N > 500)  (:1043-1044) */
    uint32_t algos;         /* SQ_ALGO_* override, 0 = use each paramset's own (:1065-1066) */
    uint64_t priority_mask; /* bit p set: paramset p is prioritised (:912-913) */
} sq_fold_opts;

/* ---- library -------------------------------------------------------------- */
SQ_API int sq_version(void);
SQ_API const char *sq_last_error(void);
/* Status -3 means "a capacity the batch was created with did not hold the fold" (the reference has no capacities: its lists are
 * Python lists, SQRNdbnseq.py:427-495).  sq_last_capacity() says which one the calling thread's last -3 was about, so that a
 * caller can repeat the fold with a larger batch without reading the message:
 *   SQ_CAP_CANDIDATES  candidate records per structure (sq_batch_desc.cand_per_nt)
 *   SQ_CAP_STRUCTS     structure slots / the log of final structures (sq_batch_desc.max_structs)
 *   SQ_CAP_OUTPUT      the output records of a round (sized from the candidate arena: grows with cand_per_nt)
 *   SQ_CAP_FIXED       a limit no descriptor field moves (64 pseudoknot levels, a chain's stem list, the blossom arrays)
 * 0 when the last error was not a capacity. */
#define SQ_CAP_CANDIDATES 1
#define SQ_CAP_STRUCTS 2
#define SQ_CAP_OUTPUT 3
#define SQ_CAP_FIXED 4
SQ_API int sq_last_capacity(void);
/* The library keeps idle pinned host buffers of destroyed batches for the next ones (hipHostMalloc / hipHostFree cost
 * milliseconds; SQ_PINNED_CACHE_MB bounds the cache).  sq_host_cache_trim() gives every idle buffer back to the driver
 * -- what torch.cuda.empty_cache() is for device memory: a process that changes its workload (a bench between its legs,
 * a server between tenants) calls it so that the next batches do not pay for evicting buffers of sizes nobody asks for
 * again.  Returns the number of bytes released.  No reference counterpart (the reference has no device). */
SQ_API long long sq_host_cache_trim(void);

/* ---- batch lifecycle ------------------------------------------------------- */
/* Bytes of device workspace sq_batch_create() needs for this description. */
SQ_API int sq_batch_workspace_bytes(const sq_batch_desc *desc, size_t *bytes);
/* Uploads the O(N) inputs; the N x N matrices are produced on device by sq_bpmatrix_fill. */
SQ_API int sq_batch_create(sq_batch **out, const sq_batch_desc *desc, void *dev_workspace, size_t workspace_bytes,
                    void *hip_stream);
SQ_API void sq_batch_destroy(sq_batch *b);

/* ---- a-1  BPMatrix (SQRNdbnseq.py:258-367) -------------------------------------
 * Fills the fp32 scan matrix of every job (row pitch ld = 32*ceil((N-1)/32)+1 floats,
 * quiet-NaN sentinel where bpboolmatrix == 0).  Asynchronous on the stream. */
SQ_API int sq_bpmatrix_fill(sq_batch *b);
/* Returns job's (bpboolmatrix, bpscorematrix) exactly as the reference would:
 * dense N x N fp64, computed on device in fp64.  Synchronises. */
SQ_API int sq_bpmatrix_read(sq_batch *b, int32_t job, double *boolmat, double *scoremat);

/* ---- a-2..a-6  AnnotateStems / ScoreStems / ChooseStems / OptimalStems ------------
 * (SQRNdbnseq.py:427-495, 607-751, 754-789, 792-833)
 * Evaluates one greedy round for `nstruct` partial structures at once.
 * struct s belongs to job struct_job[s]; its already-selected stems are
 * stems[stem_off[s] .. stem_off[s+1]) (only i, j, len are read).
 * mode 0 (OptimalStems): out receives the stems ChooseStems would return for
 *        subopt[s], in its order, with bpscore and finalscore.
 * mode 1 (AnnotateStems): out receives every stem with len >= minlen and
 *        bpscore >= minbpscore in the reference's emission order
 *        (anti-diagonal i+j ascending, then i ascending); finscore = 0.
 * out_off[nstruct+1] delimits each structure's slice of out[0..out_cap).
 * Synchronises the stream. */
SQ_API int sq_optimal_stems(sq_batch *b, int32_t nstruct, const int32_t *struct_job, const int32_t *stem_off,
                     const sq_stem *stems, const double *subopt, int32_t mode,
                     sq_stem *out, int32_t out_cap, int32_t *out_off);

/* ---- a-8 / a-9 (+ Nussinov)  RunAlgo with Hungarian / Edmonds / Nussinov -------------------
 * (SQRNdbnseq.py:548-595, SQRNalgos.py:44-135).  For each listed job: AnnotateStems with no
 * selected stems, then the matching / DP on the device -- scipy's linear_sum_assignment and
 * networkx's max_weight_matching restated step by step, so ties resolve as in the reference --
 * then RunAlgo's stem filters.  algo is ONE of SQ_ALGO_E / SQ_ALGO_H / SQ_ALGO_N;
 * levellimit < 0 selects the default 3 - (N > 500).  out gets each job's stemset
 * (bpscore = finscore = raw stem score).  Synchronises the stream. */
SQ_API int sq_run_algos(sq_batch *b, int32_t njob, const int32_t *job_ids, int32_t algo, int32_t levellimit,
                        sq_stem *out, int32_t out_cap, int32_t *out_off);

/* ---- a-8 / a-9 at graph level: the reference's SQRNalgos.Hungarian / Edmonds / Nussinov called on their own -----
 * (a maintainer patches `Edmonds(stems)` etc. one for one).  No batch: the inputs are plain host
 * arrays, the results come back in host arrays, `dev_workspace` is caller-owned DEVICE scratch of at least
 * sq_*_workspace_bytes() bytes (256-byte aligned), work is enqueued on hip_stream and the call synchronises it.
 *
 * sq_mwm -- Edmonds (SQRNalgos.py:96-110): networkx.max_weight_matching (3.4.2, maxcardinality=False) of `ngraph`
 * graphs given as weighted edge lists; graph g owns edges [edge_off[g], edge_off[g+1]) of (eu, ev, ew).  Vertex
 * labels are arbitrary non-negative ints (sequence positions in the reference).  As in networkx: nodes are ordered
 * by first appearance, a repeated edge keeps its first position and takes the last weight.  pairs[2k], pairs[2k+1]:
 * graph g's matched pairs are k in [pair_off[g], pair_off[g+1]), each oriented (u, v) as networkx returns it and
 * sorted as Edmonds() sorts them (:109); ties between optimal matchings resolve as in networkx (step-exact). */
SQ_API int sq_mwm_workspace_bytes(int32_t ngraph, const int64_t *edge_off, const int32_t *eu, const int32_t *ev,
                                  size_t *bytes);
SQ_API int sq_mwm(int32_t ngraph, const int64_t *edge_off, const int32_t *eu, const int32_t *ev, const double *ew,
                  int32_t *pairs, int64_t pair_cap, int64_t *pair_off, void *dev_workspace, size_t workspace_bytes,
                  void *hip_stream);
/* sq_lsap -- the assignment step of Hungarian (SQRNalgos.py:119-127): scipy.optimize.linear_sum_assignment(mat)
 * (1.15.3) for `nprob` symmetric n[g] x n[g] matrices that are zero except mat[cv, cw] = mat[cw, cv] = -weight for
 * the listed cells (a repeated cell takes the last weight).  col4row: concatenated, n[g] ints per problem --
 * col4row[r] is the column assigned to row r, the `col_ind` scipy returns (ties resolve as in scipy). */
SQ_API int sq_lsap_workspace_bytes(int32_t nprob, const int32_t *n, const int64_t *cell_off, size_t *bytes);
SQ_API int sq_lsap(int32_t nprob, const int32_t *n, const int64_t *cell_off, const int32_t *cv, const int32_t *cw,
                   const double *weight, int32_t *col4row, void *dev_workspace, size_t workspace_bytes,
                   void *hip_stream);
/* sq_nussinov -- Nussinov + BackTrack (SQRNalgos.py:6-93, minloop = 3) for `nprob` sequences: SCORES[(cv, cw)] =
 * -score for the listed cells (cv < cw); codes: the sequences' letter codes, concatenated (n[g] each; only the
 * separators SQ_CODE_SEP1/2 matter).  pairs / pair_off as for sq_mwm, each problem's pairs sorted (:41). */
SQ_API int sq_nussinov_workspace_bytes(int32_t nprob, const int32_t *n, const int64_t *cell_off, size_t *bytes);
SQ_API int sq_nussinov(int32_t nprob, const int32_t *n, const uint8_t *codes, const int64_t *cell_off,
                       const int32_t *cv, const int32_t *cw, const double *score, int32_t *pairs, int64_t pair_cap,
                       int64_t *pair_off, void *dev_workspace, size_t workspace_bytes, void *hip_stream);

/* ---- a-7 + a-10  greedy pool loop and the ranking tail of SQRNdbnseq ---------------
 * (SQRNdbnseq.py:1048-1286).  Folds every sequence of the batch under its jobs and
 * keeps the per-sequence results inside the batch for the getters below.
 * refs: optional per-sequence known structure as pairs (NULL = no metrics). */
SQ_API int sq_fold(sq_batch *b, const sq_fold_opts *opts, const int32_t *ref_off, const int32_t *ref_pairs,
            const uint8_t *has_ref);

/* The same for several independent batches at once, one host thread per batch: while one batch's host code books a
 * round, the kernels of the others keep the GPU busy (S1000: two batches of 512 fold in 9 ms instead of 12.4 ms as
 * one batch of 1,024).  Give the batches different streams if their kernels should overlap on the GPU as well.
 * ref_off / ref_pairs / has_ref: per batch, as for sq_fold (each entry, or the arrays themselves, may be NULL).
 * Returns the first non-zero status (its text is available from sq_last_error on the calling thread). */
SQ_API int sq_fold_concurrent(sq_batch *const *batches, int32_t nbatch, const sq_fold_opts *opts,
                              const int32_t *const *ref_off, const int32_t *const *ref_pairs, const uint8_t *const *has_ref);
/* The steady-state form: every batch is folded `reps` times back to back by its own host thread, WITHOUT a barrier
 * between the repetitions -- the batches drift apart and overlap each other's host-heavy and GPU-heavy phases (a
 * server that re-uses resident batches for a stream of identical requests; bench.py's timed region is one such call).
 * After the call every batch holds the results of its last fold. */
/* A caller that folds several batches at once from threads of its OWN (a server with a rolling window of requests: each slot
 * creates, folds, reads out and destroys its batches independently, no barrier between them) tells every batch how many are
 * in flight -- what sq_fold_concurrent does for its batches: the host threads then wait for the device with pauses instead of
 * spinning on every CPU, and the worker pools and side streams are sized for a shared chip.  n < 1 counts as 1. */
SQ_API int sq_batch_set_inflight(sq_batch *b, int32_t n);
SQ_API int sq_fold_concurrent_n(sq_batch *const *batches, int32_t nbatch, const sq_fold_opts *opts,
                                const int32_t *const *ref_off, const int32_t *const *ref_pairs, const uint8_t *const *has_ref,
                                int32_t reps);

/* Which driver ran the greedy pool loop of the batch's last fold: 0 the host loop over round kernels, 1 rounds chained on
 * the device (poollim == 1), 2 device pools, 3 device pools that outgrew a capacity and were repeated by the host loop.
 * All give identical results; the number is for tests and tuning (max_structs). */
SQ_API int32_t sq_fold_driver(const sq_batch *b);
/* Which parts of the batch's last fold ran on the device beyond the kernels of a round: bit 0 -- the ranking tail
 * (SQRNdbnseq.py:1201-1286: dedupe, ScoreStruct, RankStructs, bracket rows, consensus, metrics; else the host tail took it:
 * rankbydiff, forced hardrest pairs, conslim > 1, > 4096 final structures of one sequence), bit 1 -- RunAlgo's edge lists and
 * stem filters (SQRNdbnseq.py:548-595; else host-built: reactivity factors / non-dyadic weights on E / H paramsets,
 * sequences above 4096 nt), bit 2 -- the rounds of width-1 pools (driver 1) as ONE launch of the persistent round kernel
 * (SQRNdbnseq.py:1102-1199: a block per structure loops over its own rounds; else one chain of launches per round:
 * jobs with dense matrices, sequences above 8192 nt, SQ_NO_ROUNDS), bit 3 -- a round of the device pools (driver 2) as ONE
 * kernel per structure list + the scan kernel (sequences up to 256 nt with batches in flight or >= 4096 jobs; else state /
 * scan / score / choose / extend kernels), bit 4 -- jobs of wider pools that almost never branch (range factor 1.0, cells from
 * a dense fp64 matrix: the alignment's rows, bpp terms) ran as chains on the persistent round kernel first; the ones that met a
 * tie were folded by the device pools, bit 5 -- the rounds of the device pools were enqueued ahead of the host (a batch alone:
 * every launch covers all structure slots and follows the generation sizes the device publishes; SQ_POOL_AHEAD), bit 6 -- the
 * device pools of sequences of 257-1,024 nt ran the one-wave round kernel over per-job root lists (the runs of the empty
 * structure with their bpscores, checked against every structure's partner array) instead of the launched scan and score
 * kernels (SQ_POOL_ROOT, or bit 7), bit 7 -- ... and every structure read the list its PARENT left (SQ_BATCH_POOL_LISTS: runs,
 * bpscores and the finalscores no strand of the child's stem comes near) in the root list's place, scoring only what the new
 * stem changed, one launch per round.  Identical results either way; for tests and tuning. */
SQ_API int32_t sq_fold_paths(const sq_batch *b);
/* Most structures any round of the batch's last fold evaluated at once (device pools: the largest generation; 0 when the
 * host-driven loop ran).  A host that folds a stream of similar batches sizes max_structs from it. */
SQ_API int64_t sq_fold_peak_structs(const sq_batch *b);

/* Result getters (valid after sq_fold until the next sq_fold / destroy). */
/* k > 0: the getters below (single and bulk) show only the first k structures of every sequence, in rank order -- a
 * caller that prints the top few (RunSQRNdbnseq's outplim, SQRNdbnseq.py:1289-1410) need not fetch a pool of a thousand.
 * k = 0 (default): all of them, as SQRNdbnseq returns them.  Consensus and metrics do not depend on it.  A limit set
 * BEFORE sq_fold also spares the fold the bracket strings of the structures beyond it (most of the ranking tail's time
 * for pools of a thousand); those structures are then not kept, and raising the limit afterwards does not bring them back. */
SQ_API int sq_result_limit(sq_batch *b, int32_t k);
SQ_API int32_t sq_result_nstruct(const sq_batch *b, int32_t seq);
/* levels: per position, 0 = unpaired, +L = opening bracket of level L, -L = closing. */
SQ_API int sq_result_consensus(const sq_batch *b, int32_t seq, int16_t *levels);
SQ_API int sq_result_struct(const sq_batch *b, int32_t seq, int32_t k, int16_t *levels, double scores[3],
                     uint64_t *pset_mask);
SQ_API int sq_result_metrics(const sq_batch *b, int32_t seq, double cons[6], double best[7]);
/* Bulk form of the getters: one buffer per sequence, little-endian, 8-byte aligned sections:
 *   int64  nstruct, n, has_ref, evals
 *   double cons_metrics[6], best_metrics[7], ref_scores[3]   (ref_scores: ScoreStruct of the known structure,
 *                                                              ReferenceScores SQRNdbnseq.py:958-970; NaN without one)
 *   double scores[nstruct][3]
 *   uint64 pset_mask[nstruct]
 *   int16  levels[1 + nstruct][n]      (row 0 = consensus)
 * sq_result_pack_size returns the bytes needed. */
SQ_API int64_t sq_result_pack_size(const sq_batch *b, int32_t seq);
SQ_API int sq_result_pack(const sq_batch *b, int32_t seq, void *buf, int64_t cap);
/* Every sequence of the batch in one call (the payload of the multi-GPU result gather): record s occupies
 * [off[s], off[s+1]) of buf, off has nseq + 1 entries, records start 8-byte aligned. */
SQ_API int64_t sq_result_pack_all_size(const sq_batch *b);
SQ_API int sq_result_pack_all(const sq_batch *b, void *buf, int64_t cap, int64_t *off);
/* The same records WITHOUT a copy, where the device tail wrote them: *buf = the library's pinned host buffer, *off = its
 * nseq + 1 offsets (both owned by the batch, valid until its next sq_fold or sq_batch_destroy; read-only).  Returns 0 and the
 * pointers, or 1 when the last fold's records are not in that form (the host tail ran, or sq_result_limit shows fewer
 * structures than were packed): then sq_result_pack_all forms them.  Replaces the copy a caller of SQRNdbnseq never had to
 * make (the reference returns its tuples by reference, SQRNdbnseq.py:1285-1286). */
SQ_API int sq_result_view(const sq_batch *b, const void **buf, const int64_t **off, int64_t *nbytes);
/* ... and the buffer itself handed over: after sq_result_view returned 0, sq_result_detach makes the records' pinned buffer
 * the CALLER's -- the batch forgets it (and its results: the next fold takes a new buffer), its destruction leaves it alone --
 * until sq_buffer_release returns it to the library's pinned cache.  For a caller that keeps the records beyond the batch
 * (the results of a Python call outlive the batch that made them: 544 MB per thousand 500-nt records under pools of a
 * thousand were copied record by record, a quarter of the call).  Copy the offsets first: they stay the batch's. */
SQ_API int sq_result_detach(sq_batch *b, void **buf, int64_t *nbytes);
SQ_API void sq_buffer_release(void *buf);
/* Dot-bracket rows of every record as ASCII text: record s occupies [off[s], off[s+1]) of buf with its consensus row
 * and then its nstruct structure rows, n characters each (gap-free coordinates; gap columns and separators are
 * re-inserted by the caller, SQRNdbnseq.py:1239-1246).  Levels 1..30 print as ( [ { < A..Z and ) ] } > a..z (:107-112);
 * deep[s] = 1 when record s uses a deeper level (the reference continues with Cyrillic letters): use the level form
 * (sq_result_pack) for those. */
SQ_API int64_t sq_result_dbn_all_size(const sq_batch *b);
SQ_API int sq_result_dbn_all(const sq_batch *b, char *buf, int64_t cap, int64_t *off, uint8_t *deep);
/* ---- text of the drop-in layer (per-record work of RunSQRNdbnseq, in C++ because a batch has thousands of records) ----
 * sq_dbn_pairs: DBNToPairs (SQRNdbnseq.py:172-207) for nrec dot-bracket lines: line r = text[off[r], off[r+1]); brackets
 * ( ) [ ] { } < > A..Z / a..z, unmatched closers ignored; line r's pairs, sorted, at pairs[2 pair_off[r] .. 2 pair_off[r+1]). */
SQ_API int sq_dbn_pairs(const char *text, const int64_t *off, int32_t nrec, int32_t *pairs, int64_t pair_cap, int64_t *pair_off);
/* sq_write_blocks: the output block of RunSQRNdbnseq (SQRNdbnseq.py:1301-1406: name, sequence, optional reactivities /
 * restraints / reference lines, the consensus line, up to outplim structure lines with scores, paramset names and metrics)
 * for every record of a batch whose last fold left packed results (sq_fold_paths bit 0).  The per-record input lines come
 * '\n'-joined, one line per record in batch order (an empty line: the record has none; a NULL field: no record has one);
 * seqs are the INPUT sequences (gap columns and separators included: they are re-inserted into every bracket row,
 * :1239-1246); reacts the already encoded reactivity lines (EncodedReactivities, :82-101).  Record r's block is
 * buf[off[r], off[r+1]); skipped[r] = 1: the record uses bracket levels beyond the ASCII alphabet, its block is empty and
 * left to the caller.  Returns the bytes written, or -(bytes needed + 16) when cap is too small (nothing written). */
typedef struct sq_block_desc {
    int32_t nrec;
    const char *names, *seqs, *reacts, *restr, *refs;
    const int32_t *nameset;         /* [nrec] which list of paramset names a record prints (NULL: list 0)      */
    const char *const *psnames;     /* [nsets] the names of a configuration's paramsets, '\n'-joined            */
    int32_t nsets, conslim, outplim;
} sq_block_desc;
SQ_API int64_t sq_write_blocks(const sq_batch *b, const sq_block_desc *d, char *buf, int64_t cap, int64_t *off, uint8_t *skipped);

/* sq_parse_default -- the record scan of the reference's default input format (SQUARNA.py:80-203, ParseDefaultInput) over
 * the bytes of a file: a '>' line names a record, the lines behind it are its fields in the order of `inputformat`
 * (nfields letters; q_ind / t_ind / r_ind / f_ind: the positions of q, t, r, f in it, -1 when absent; as in the reference
 * a t / r / f at position 0 is not read).  out[rec][10] = offset, length of: the stripped name line, the sequence token,
 * the stripped reactivities line, the restraints token, the reference token (length -1: None; restraints / reference
 * also for an empty line).  Returns the number of records, or -1 when the file needs the general parser: default lines
 * in front of the first record, a byte outside ASCII, NUL or '\r', a record without a sequence token, more than cap
 * records.  Host code; values are not validated here (lengths, reactivities): the caller's checks stay the reference's. */
/* sq_align_first_fit -- the first structure of MatrixToDBNs (SQRNdbnali.py:121-192, the one its caller keeps, :242): the cells
 * flat[k] = v * N + w, given in decreasing order of value (ties: flat index ascending), are taken one by one; a cell of span
 * >= minspan whose columns are both free joins.  pairs: up to cap (v, w) pairs in the order taken; returns their number. */
SQ_API int64_t sq_align_first_fit(const int64_t *flat, int64_t n, int32_t N, int32_t minspan, int32_t *pairs, int64_t cap);
SQ_API int64_t sq_parse_default(const char *text, int64_t len, int32_t nfields, int32_t q_ind, int32_t t_ind, int32_t r_ind,
                                int32_t f_ind, int64_t *out, int64_t cap);

/* R = number of AnnotateStems evaluations the reference algorithm performs for this sequence. */
SQ_API int64_t sq_result_evals(const sq_batch *b, int32_t seq);

/* ---- alignment step 1 (SQRNdbnali.py:60-108, 211-242) ------------------------------------
 * For each listed job, in order: AnnotateStems with no selected stems, then
 *   matrix[cols[v], cols[w]] += stem score;  matrix[cols[w], cols[v]] += stem score
 * for every base pair (v, w) of every stem -- the parent-side loop of SQRNdbnali (:233-237).
 * cols[col_off[k] + p] is the alignment column of position p of job_ids[k] (ReAlignDict, :20-37).
 * d_matrix: caller-owned DEVICE memory, L x L fp64 row-major, accumulated into: zero it first, then pass
 * the same matrix to as many calls as the alignment needs.  Both additions of a pair are the same number
 * in the same order, so the library accumulates the upper triangle and copies it onto the lower one at the
 * end of every call (the lower triangle of the input is overwritten).
 * A cell receives at most one addition per sequence and sequences are applied in list order, so
 * every cell sees the reference's fp64 summation order (when every sum is exact anyway -- pair weights
 * that are multiples of 2^-10, no reactivities -- the sequences of a chunk are added in one launch).
 * The matrix is complete when the call returns. */
SQ_API int sq_align_accumulate(sq_batch *b, int32_t njob, const int32_t *job_ids, const int32_t *col_off,
                               const int32_t *cols, int32_t L, double *d_matrix);
/* Cells (v, w) of a device L x L fp64 matrix with w - v >= minspan and value >= threshold
 * (MatrixToDBNs' candidates, SQRNdbnali.py:127-148).  All buffers are caller-owned DEVICE memory:
 * d_idx[cap] (flat indices v*L+w), d_val[cap], d_count[1] (zeroed here; may end up > cap, then only
 * the first cap hits were stored).  Unordered.  Asynchronous on hip_stream. */
SQ_API int sq_colmatrix_select(const double *d_matrix, int32_t L, double threshold, int32_t minspan,
                               int64_t *d_idx, double *d_val, int64_t cap, uint64_t *d_count, void *hip_stream);

/* ---- measurement ------------------------------------------------------------
 * Kernel ids: 0 fill, 1 state, 2 stem_scan, 3 stem_score (+ select), 4 Edmonds, 5 Hungarian,
 * 6 Nussinov.  When enabled, every launch is bracketed by hipEvents on the stream it runs on
 * (the matching kernels run on the batch's side streams). */
SQ_API int sq_profile_enable(sq_batch *b, int32_t on);
SQ_API int sq_profile_get(sq_batch *b, int32_t kernel, double *total_ms, int64_t *launches, double *alg_bytes);
SQ_API int sq_profile_reset(sq_batch *b);
/* Work counters of the blossom kernel (kernel 4) since the last reset, always on: out[0] graphs matched, out[1] their
 * scan passes (one pass = one chunk of <= 64 neighbours of a popped S-vertex), and for the graph with the most
 * passes -- the kernel's critical path, one wave per graph -- out[2] its passes, out[3] its lane-0 events, out[4] its
 * vertices, out[5] its edges. */
SQ_API int sq_profile_counters(sq_batch *b, int32_t kernel, int64_t out[6]);

#ifdef __cplusplus
}
#endif
#endif /* SQUARNA_HIP_H */
