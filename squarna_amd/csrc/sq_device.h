// sq_device.h -- kernel argument bundles (passed by value) and kernel prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include "sq_internal.h"
#include "sq_context.h"

struct SqDevCtx {
    const SqJob *jobs;
    const SqPsetDev *psets;
    const uint8_t *codes;    // per position
    const uint8_t *flags;
    const uint8_t *inc4;     // minimal j - i for a pair starting at i (SQRNdbnseq.py:294-297)
    const int16_t *chain;    // chain ordinal (interchainonly, :264-271)
    const uint8_t *e0c;      // restraint mask code: 0 free, k+1 = k-th restraint bp (v,w) of the sequence (:438-443)
    const double *reacts;
    const uint8_t *ridx;     // per position: index of its reactivity among the sequence's distinct values (SqJob::react_levels)
    float *mat32;            // fp32 scan-matrix arena
    double *mat64;           // dense fp64 arena (external / weighted matrices only)
    const double *sdftab;    // pow tables
    uint32_t *bits;          // diagonal bit matrices (activity of the unmasked BPMatrix, 1 bit per cell)
    const uint32_t *rbpk;    // restraint base pairs, v | (w << 16), per sequence (SqJob::rb_off, nrb)
    const double *rftab;     // reactfactor tables, 256 doubles each (SqJob::rf_idx)
    const double *powtab;    // stemscore ** 1.7 tables (SqPsetDev::pow_off)
    // alignment step 2 (SQRNdbnseq.py:1031-1034,1084-1085): the shared stem matrix, DIAGONAL-major over the alignment's columns
    // (sq_diag_index(mulL, v, w)), and the column of every position; jobs with SqJob::mulsh read their weights from it
    const double *mulM; const int32_t *mulcols; int32_t mulL;
};

struct SqState {             // per-structure-slot arrays, `stride` elements per slot
    int16_t *P, *U, *SU;     // partner, prefix #unpaired, prefix #unpaired separators
    uint8_t *E8;             // scan mask code per position: 0 free, 255 masked, k+1 restraint bp k
    int32_t stride;
    uint32_t *FB;            // per slot: free-position bit words, forward [0, fbstride/2) and reversed + padded
    int32_t fbstride;        // words per slot (even)
};
#define SQ_GPAD 128          // bit offset of the reversed free-position array (window starts never go negative)

// Round I/O without copies or stream waits: the round's structures and strands are read by the state
// kernel straight from pinned host memory (and mirrored into device memory for the later kernels), the
// selected stems are written straight into pinned host memory, and a one-thread kernel at the end of the
// round publishes the counters and a sequence number the host spins on.
#ifndef SQ_SCORE_CHUNK
#define SQ_SCORE_CHUNK 4          // candidates per thread and chunk of sq_score_kernel's two-phase loop
#endif

struct SqRoundIO {
    const SqStruct *h_structs;   // pinned, host-written
    const SqStrand *h_strands;
    SqStruct *d_structs;         // device mirrors
    SqStrand *d_strands;
    SqOut *h_out;                // pinned: records [0, h_cap)
    SqOut *d_out;                // device: records [h_cap, out_cap)
    uint32_t h_cap, out_cap;
    SqCounters *h_ctr;           // pinned
    volatile uint32_t *h_seq;    // pinned: id of the last finished round
};

struct SqScanArgs {
    SqCand *cands;
    uint32_t *cand_cnt;      // per slot
    unsigned long long *best; // per slot: order-preserving image of the round's best finalscore (0: none yet)
    uint32_t *ok_cnt;        // per slot: candidates that passed the exact thresholds (length of the SqOk list)
    SqCounters *ctr;
};

// device-chained rounds: where sq_chain_kernel keeps and reports the structures
struct SqChainIO {
    SqChain *chain;               // per structure
    SqChainStem *stems;           // device: stems of every structure (slice per structure)
    SqStrand *strands;            // device: two strand buffers per structure
    int16_t *sidx;                // stem index of every strand (same layout as strands)
    SqStemOut *h_stems;           // pinned: the chosen stems in selection order (same slices as `stems`)
    unsigned long long *h_fin;    // pinned: finished structures, job | nstems << 32 | (ended by maxstemnum) << 63
    uint32_t *d_nfin;             // device: entries of h_fin
    volatile uint32_t *h_nfin;    // pinned copy, published by sq_chain_done_kernel before the round's sequence number
    // the batch's device log of final structures (input of the device tail, sq_tail_dev.hip): a retired structure gets a
    // record that points at its slice of `stems`, and its job's evaluation count
    SqPoolFin *fin; uint32_t *fin_ctr; uint32_t fin_cap;
    long long *job_evals;
};

// Kept lists (sequences of 257-1,024 nt): every structure of the pools leaves the list of its runs -- key, length, exact bpscore and,
// where ScoreStems ran, the finalscore -- for its children, which cut it against the two strands of their own stem instead of
// scanning, and keep every finalscore no strand of that stem comes near (the rule of sq_rounds.hip's chains: a finalscore reads
// the strands inside the run's span and within six positions of it, and the NUMBER of levels among them).  The lists live in
// pages of SQ_KEPT_PG entries -- [keys][length | SQ_RX_FIN | SQ_RX_LVL][bpscores][finalscores] -- taken from a pool per
// generation (two generations: the parents' pages are read while the children's are written; sq_pool_scan_kernel empties the
// pool of the generation after next); a structure's page numbers stand in its row of `tab`.  A structure that finds the pool
// empty (or needs more than SQ_KEPT_TAB pages: 65,536 runs) leaves no list: its children start from the job's root list.
#define SQ_KEPT_PG 256
#define SQ_KEPT_TAB 256
#define SQ_KEPT_NOLIST 0xFFFFFFFFu
#define SQ_KEPT_PAGE_BYTES (SQ_KEPT_PG * 24)
struct SqKept {
    char *pages;            // [2][npages] pages
    uint32_t *ctr;          // [2] pages taken, per generation (+ [2]: the most pages a generation took, [3]: structures that left no list)
    uint32_t *cnt;          // [2][smax] entries of a structure's list (SQ_KEPT_NOLIST: none)
    uint32_t *tab;          // [2][smax][SQ_KEPT_TAB] its pages
    uint32_t npages;        // pages per generation
    int32_t on;
};
struct SqKeptPage { uint32_t *key, *lf; double *bps, *fin; };
__device__ __forceinline__ SqKeptPage sq_kept_page(const SqKept &K, int gen, uint32_t id)
{
    char *b = K.pages + ((size_t)gen * K.npages + id) * (size_t)SQ_KEPT_PAGE_BYTES;
    return SqKeptPage{reinterpret_cast<uint32_t *>(b), reinterpret_cast<uint32_t *>(b + 4 * SQ_KEPT_PG), reinterpret_cast<double *>(b + 8 * SQ_KEPT_PG),
                      reinterpret_cast<double *>(b + 16 * SQ_KEPT_PG)};
}

// device pools: two generations of structure slots (parents / children), slot c of generation p at (p * smax + c)
struct SqPoolIO {
    SqStruct *structs;            // [2][smax]
    SqChain *recs;                // [2][smax]   (toff, tcap = pt, nstems, anycross, maxstems)
    SqChainStem *stems;           // [2][smax][pt]
    SqStrand *strands;            // [2][smax][2 pt]
    int16_t *sidx;                // [2][smax][2 pt]
    int32_t smax, pt, cmax;       // smax: slots per generation in the arrays (stride)
    int32_t slots;                // slots usable this fold (<= smax)
    int32_t chunk;                // structures whose candidates fit the arena at once: slot s uses region s % chunk, and the
                                  // round's kernels up to sq_pool_choose_kernel run over slots [k chunk, (k + 1) chunk) in turn
    int32_t poollim;
    long long maxcap;             // candidate records per slot
    SqPoolJob *jobs; int32_t njobs;
    const int32_t *jobrec_of;     // batch job index -> record in jobs[]
    int32_t *nchild;              // [smax] children of every structure of the round (0: final)
    int32_t *child_off;           // [smax + 1] exclusive scan of nchild
    uint8_t *finalflag;           // [smax] 1: the structure is final and still has to be logged
    SqPoolPick *chosen;           // [2][smax][cmax]: the launched kernels use the first half; sq_pool_round_kernel the half of its generation
    int32_t *parent_of;           // [smax] next generation: parent (position in this round's list, bits 0-25) of every child and
                                  // the index of its pick among the parent's chosen stems (bits 26-31)  (sq_pool_scan_kernel)
    SqPoolHdr *hdr;
    // the batch's device log of final structures ([0] entries, [1] stems, [2] overflow in fin_ctr): read by the device
    // tail (sq_tail_dev.hip); the host-driven tail copies it out
    SqPoolFin *fin; uint32_t fin_cap;
    SqPoolStem *fin_stems; uint32_t fin_stem_cap;
    uint32_t *fin_ctr;
    long long *job_evals;                           // [batch jobs] evaluations of every greedy job (sq_pool_publish_kernel)
    SqPoolHdr *h_hdr;                               // pinned copies, published by the scan kernel: a ring of SQ_POOL_HDR_RING records, the
                                                    // round with sequence number q writes record q % SQ_POOL_HDR_RING (rounds may be
                                                    // enqueued ahead of the host: it reads the record of the round it waited for)
    SqPoolJob *h_jobs;                              // pinned copy of the job records (sq_pool_publish_kernel)
    uint32_t *kept_ctr;                             // SqKept::ctr of a fold on kept lists (else nullptr): sq_pool_scan_kernel empties the next generation's pool
};

#include "sq_hostflag.h"

// Wave-wide minima / maxima with DPP row shifts and row broadcasts (the gfx9 scan sequence row_shr 1, 2, 4, 8, row_bcast 15
// and 31 leaves the reduction of all 64 lanes in lane 63): VALU only.  A butterfly of __shfl_xor costs six trips over the
// LDS crossbar per 32-bit word (sq_blossom.h has the same functions for its own use and the measurements).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double sq_dpp_f64(double ident, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sq_wave_min_f64(double v)      // every lane gets the minimum (no NaNs)
{
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double o;
    o = sq_dpp_f64<0x111, 0xf>(inf, v); v = __builtin_fmin(o, v);
    o = sq_dpp_f64<0x112, 0xf>(inf, v); v = __builtin_fmin(o, v);
    o = sq_dpp_f64<0x114, 0xf>(inf, v); v = __builtin_fmin(o, v);
    o = sq_dpp_f64<0x118, 0xf>(inf, v); v = __builtin_fmin(o, v);
    o = sq_dpp_f64<0x142, 0xa>(inf, v); v = __builtin_fmin(o, v);
    o = sq_dpp_f64<0x143, 0xc>(inf, v); v = __builtin_fmin(o, v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sq_wave_max_f64(double v)      // every lane gets the maximum (no NaNs; the sign of a zero as fmax leaves it)
{
    const double ninf = __longlong_as_double((long long)0xFFF0000000000000ull);
    double o;
    o = sq_dpp_f64<0x111, 0xf>(ninf, v); v = __builtin_fmax(o, v);
    o = sq_dpp_f64<0x112, 0xf>(ninf, v); v = __builtin_fmax(o, v);
    o = sq_dpp_f64<0x114, 0xf>(ninf, v); v = __builtin_fmax(o, v);
    o = sq_dpp_f64<0x118, 0xf>(ninf, v); v = __builtin_fmax(o, v);
    o = sq_dpp_f64<0x142, 0xa>(ninf, v); v = __builtin_fmax(o, v);
    o = sq_dpp_f64<0x143, 0xc>(ninf, v); v = __builtin_fmax(o, v);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int sq_wave_min_i32(int v)
{
    const int big = 0x7fffffff;
    int o;
    o = __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false); v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false); v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false); v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false); v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false); v = o < v ? o : v;
    o = __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false); v = o < v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}

// inclusive prefix sum over the lanes (the gfx9 scan sequence of DPP row shifts and row broadcasts: VALU only); lane 63 holds the total
// A record every lane reads at the same address, through the scalar cache (s_load into SGPRs: one request per wave, batched
// dwordx4/x8/x16) wherever it stands in the kernel: the compiler only does that by itself for loads it can prove no store of
// the kernel precedes.  For data the kernel itself never writes.
template <class T> __device__ __forceinline__ T sq_kload(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "sq_kload: whole words");
    typedef const __attribute__((address_space(4))) uint32_t *kptr;
    const kptr q = (kptr)p;
    union { T v; uint32_t w[sizeof(T) / 4]; } u;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) u.w[i] = q[i];
    return u.v;
}

__device__ __forceinline__ int sq_wave_scan_add_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// LDS operations of one wave execute in program order; this keeps the compiler from moving them across the point (the
// barrier of code in which ONE wave hands data to its own lanes through LDS)
__device__ __forceinline__ void sq_wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// One entry + nst stems of the device log of final structures: both counters (fin_ctr[0] entries, fin_ctr[1] stems: adjacent,
// 8-byte aligned) in ONE 64-bit atomic -- two returning atomics in a row were two trips to L2 for every final structure, on
// counters every wave of the batch's kernels shares.  (The entry counter cannot carry into the stems': the log has < 2^32 entries.)
__device__ __forceinline__ void sq_log_reserve(uint32_t *fin_ctr, uint32_t nst, uint32_t &idx, uint32_t &so)
{
    const unsigned long long r = atomicAdd(reinterpret_cast<unsigned long long *>(fin_ctr), ((unsigned long long)nst << 32) | 1ull);
    idx = (uint32_t)r; so = (uint32_t)(r >> 32);
}

// order-preserving map double -> uint64 (never 0 for a real number), so a per-structure maximum is one atomicMax
__device__ __forceinline__ unsigned long long sq_ord(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double sq_unord(unsigned long long o)
{
    return __longlong_as_double((long long)((o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o));
}

// the two regions of a structure's candidate slice (see sq_internal.h)
__host__ __device__ inline SqKey *sq_keys(const SqScanArgs &a, const SqStruct &st) { return reinterpret_cast<SqKey *>(a.cands + st.cand_off); }
__host__ __device__ inline SqOk *sq_oks(const SqScanArgs &a, const SqStruct &st, int cand_cap)
{
    return reinterpret_cast<SqOk *>(reinterpret_cast<char *>(a.cands + st.cand_off) + (size_t)cand_cap * sizeof(SqKey));
}

extern "C" {
__global__ void sq_fill_kernel(SqDevCtx c, int only_ext, int mul_done);
__global__ void sq_bits_direct_kernel(SqDevCtx c);
__global__ void sq_bits_masks_kernel(SqDevCtx c, int max_letters);
__global__ void sq_dense64_kernel(SqDevCtx c, int job, double *boolmat, double *scoremat);
__global__ void sq_import_kernel(SqDevCtx c);
__global__ void sq_state_kernel(SqDevCtx c, SqRoundIO io, SqState st, SqScanArgs a, int lds_n, int chained);
__global__ void sq_chain_kernel(SqDevCtx c, SqStruct *structs, SqScanArgs a, SqChainIO cio, int tmax);
__global__ void sq_chain_init_kernel(const SqStruct *h_structs, const SqChain *h_chain, SqStruct *d_structs, SqChainIO cio,
                                     SqScanArgs a, int S, int first);
__global__ void sq_chain_done_kernel(SqRoundIO io, SqScanArgs a, SqChainIO cio, uint32_t seq);
__global__ void sq_done_kernel(SqRoundIO io, SqScanArgs a, uint32_t seq);
__global__ void sq_pool_init_kernel(const SqStruct *h_structs, const SqChain *h_recs, const SqPoolJob *h_jobs, const int32_t *h_jobrec,
                                    int32_t *d_jobrec, int nbatchjobs, SqPoolIO pio, SqScanArgs a, int S0);
__global__ void sq_pool_choose_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, SqPoolIO pio, int nsurv);
__global__ void sq_pool_scan_kernel(SqPoolIO pio, SqScanArgs a, SqRoundIO io, int parity, uint32_t seq);
__global__ void sq_pool_extend_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, int parity);
__global__ void sq_pool_publish_kernel(SqPoolIO pio, SqScanArgs a, SqRoundIO io, uint32_t seq);
__global__ void sq_mirror_kernel(double *matrix, int L);
__global__ void sq_scatter_all_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, const int32_t *cols,
                                      const int32_t *col_start, int L, double *matrix);
__global__ void sq_scatter_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, int sidx, const int32_t *cols,
                                  int L, double *matrix);
__global__ void sq_colselect_kernel(const double *matrix, int L, double thr, int minspan, long long *idx_out,
                                    double *val_out, long long cap, unsigned long long *count);
__global__ void sq_scan6_kernel(SqDevCtx c, const SqStruct *structs, SqState stt, SqScanArgs a);
__global__ void sq_state_scan_kernel(SqDevCtx c, SqRoundIO io, SqState stt, SqScanArgs a, int lds_n, int chained);
__global__ void sq_bits_kernel(SqDevCtx c, int only_ext);
__global__ void sq_score_kernel(SqDevCtx c, const SqStruct *structs, const SqStrand *strands, SqState stt,
                                SqScanArgs a, SqRoundIO io, int lds_n, int lds_n_reacts, int lds_n_state, int surv_off, int cell_off,
                                int str_off, int str_cap, SqCtxTab ct, int bound);
__global__ void sq_bps_kernel(SqDevCtx c, const SqStruct *structs, const SqStrand *strands, SqState stt,
                              SqScanArgs a, SqRoundIO io, int mode, int lds_n, int lds_n_reacts, int surv_off, int cell_off);
__global__ void sq_select_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, SqRoundIO io);
}

// sq_extend.h: LDS of one structure's level scratch (chain / pool-extend kernels), lists of up to T stems
#define SQ_CHAIN_TMAX 11264
__host__ __device__ static inline size_t sq_extend_lds_bytes(int T) { const size_t t = ((size_t)T + 7) & ~(size_t)7; return t * 14 + 64 * 4 + 64; }

// sq_gather.hip: the N x N weighting slices of the jobs in job_list from ONE shared L x L device matrix through the
// per-position alignment columns (alignment step 2, SQRNdbnseq.py:1031-1034,1084-1085)
void sq_launch_gather_mul(const SqDevCtx &c, const int32_t *job_list, int njl, int maxn, hipStream_t st, double *dst_one = nullptr);
// the caller's row-major L x L matrix -> the diagonal-major copy the kernels read (SqDevCtx::mulM)
void sq_launch_mul_diag(const double *M, int L, double *dst, hipStream_t st);
