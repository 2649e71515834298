// sq_tail_dev.h -- the ranking tail of SQRNdbnseq (SQRNdbnseq.py:1201-1286: dedupe of the final structures across
// paramsets, ScoreStruct, RankStructs, PairsToDBN of the structures that are shown, consensus, metrics) on the device.
//
// Input: the device log of FINAL structures -- one SqPoolFin record per structure (job, place in the job's finstemsets
// order, stems) written by the greedy drivers (sq_chain.hip / sq_pool.hip) and by the E / H / N collectors.  Output: every
// sequence's result record in the C ABI's packed layout (sq_result_pack, include/squarna_hip.h) plus its dot-bracket rows
// as ASCII text, written by the kernels straight into pinned host memory: after a fold the host holds the bytes
// sq_result_pack_all / sq_result_dbn_all hand out, and no per-sequence host code has run.
//
// What the device path does not cover falls back to the host tail (sq_tail.cpp) for the whole batch, with identical
// results: rankbydiff, hardrest with forced pairs, conslim > 1, more than SQ_TAIL_MAXM final structures for one sequence,
// stem lists that are not disjoint stacks, scores beyond the exact range of sq_round3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sq_device.h"

#define SQ_TAIL_MAXM 4096          // final structures of one sequence the O(M^2) dedupe / rank passes take
#define SQ_FIN_SRC_LOG 0u          // SqPoolFin::pad: stems in the log's SqPoolStem array
#define SQ_FIN_SRC_CHAIN 1u        // ... in the chain's SqChainStem array (stem_off = the job's toff)
#define SQ_FIN_KIND_E 0u           // SqPoolFin::round_kind of the E / H / N stemsets (they precede the greedy structures
#define SQ_FIN_KIND_H 1u           // of their job, :1094-1100, in this order)
#define SQ_FIN_KIND_N 2u
#define SQ_FIN_KIND_G0 4u          // greedy structures: SQ_FIN_KIND_G0 + 2 * round + kind (sq_pool.hip)

struct SqTailSeq {                 // per sequence, between the kernels
    uint32_t first, count;         // its entries: ord[first, first + count)
    uint32_t D, nshow;             // distinct structures, structures shown
    uint32_t nprf;                 // ranks whose metrics are computed (min(D, toplim) with a reference)
    uint32_t pad;
    long long rec_bytes, txt_bytes;
    long long rec_off, txt_off;
    long long evals;
};

struct SqTailIO {
    // the log
    const SqPoolFin *fin; const SqPoolStem *fin_stems; const uint32_t *nfin_ptr;
    const SqChainStem *chain_stems;
    uint32_t fin_cap, fin_stem_cap;
    // per job
    uint32_t *job_cnt, *job_start, *job_fill;     // [njobs + 1]
    const long long *job_evals;                   // [njobs] AnnotateStems evaluations of the greedy part
    int32_t njobs, nseq;
    const int32_t *seq_job0;                      // [nseq + 1] first job of every sequence (jobs of a sequence are contiguous)
    // per entry scratch
    uint32_t *ord, *ord2;                         // entries grouped by job, each job's in finstemsets order
    SqPoolStem *cstems;                           // canonical stems (maximal stacks, ascending i): [fin_stem_cap + chain_T]
    uint32_t *cs_n; unsigned long long *hash;
    uint32_t *rep;                                // first entry (position in the sequence's list) with the same base pairs
    unsigned long long *mask;                     // paramsets that produced the structure (bit = job's index in its sequence)
    double *scores;                               // [3] per entry
    uint32_t *dlist, *rlist;                      // distinct entries in list order / in rank order (positions)
    // per sequence
    SqTailSeq *seqs;
    // options
    int32_t rankby[3]; int32_t toplim, result_limit, conslim;   // conslim: 0 or 1 (more: host tail)
    unsigned long long priority_mask;
    // reference structures
    const int16_t *refp;                          // per position: partner in the known structure, -1 none (NULL: no references)
    const int32_t *ref_n;                         // per sequence: -1 no reference, else its number of distinct pairs
    // tables
    const double *pow17h; int32_t pow17h_len;     // pow(k / 2, 1.7), host libm (ScoreStruct, :884)
    // outputs (pinned host memory, written by sq_tail_pack_kernel)
    char *rec_buf; char *txt_buf; uint8_t *deep;
    long long *h_totals;                          // pinned: [0] record bytes, [1] text bytes, [2] fallback flag, [3] entries, [4] most structures shown for one sequence, [5] a record did not fit the buffers
    uint32_t *fallback;                           // device: [0] some sequence needs the host tail, [1] a record did not fit the result buffers
    int32_t tmax;                                 // most stems of any structure (sizes the level scratch)
};

extern "C" {
__global__ void sq_tail_count_kernel(SqTailIO t);
__global__ void sq_tail_scan_kernel(SqTailIO t);
__global__ void sq_tail_scatter_kernel(SqTailIO t);
__global__ void sq_tail_rank_kernel(SqDevCtx c, SqTailIO t, int bitwords, int keycap, int refp_lds);
__global__ void sq_tail_offsets_kernel(SqTailIO t, volatile uint32_t *h_seq, uint32_t seq);
__global__ void sq_tail_pack_kernel(SqDevCtx c, SqTailIO t, int rowcap, long long rec_cap, long long txt_cap);
__global__ void sq_fold_begin_kernel(uint32_t *fin_ctr, long long *job_evals, uint32_t *job_cnt, int njobs);
__global__ void sq_fin_keep_algos_kernel(SqPoolFin *fin, uint32_t *fin_ctr, uint32_t fin_cap, long long *job_evals, uint32_t *job_cnt, int njobs);
__global__ void sq_tail_done_kernel(SqTailIO t, long long *h_rec_off, long long *h_txt_off, volatile uint32_t *h_seq, uint32_t seq);
__global__ void sq_fin_append_kernel(const SqPoolFin *src, const SqPoolStem *src_stems, int n, SqPoolFin *fin, SqPoolStem *stems,
                                     uint32_t *ctr, uint32_t fin_cap, uint32_t stem_cap);
}
