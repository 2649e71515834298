// sq_tail_dev.hip -- the ranking tail of SQRNdbnseq on the device (see sq_tail_dev.h): SQRNdbnseq.py:1201-1286 with
// ScoreStruct (:861-899), RankStructs (:902-913; rankbydiff stays on the host), PairsToDBN (:104-163) of the structures
// that are shown, the top-1 consensus (:1236) and the metrics against a known structure (:1249-1285).
//
// Kernels, in stream order:
//   sq_tail_count / scan / scatter   group the log's entries by job (counting sort on the job index)
//   sq_tail_rank_kernel              one block per sequence: each job's entries into finstemsets order, canonical stems
//                                    (maximal stacks, ascending) + hash per entry, first occurrence of every base-pair
//                                    set (dedupe, :1201-1220), ScoreStruct of the distinct ones, the stable ranking,
//                                    metrics of the top ranks, the sequence's record size
//   sq_tail_offsets_kernel           one block: record / text offsets of all sequences, totals to the host
//   sq_tail_pack_kernel              one wave per (sequence, slice of its shown structures): pseudoknot levels at stem
//                                    level (sq_stem_levels_wave), the int16 level rows and the ASCII rows, header, scores,
//                                    masks, metrics -- written straight into pinned host memory in the C ABI's layout
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/squarna_hip.h"
#include "sq_tail_dev.h"
#include "sq_extend.h"

// ---- Python's round(x, 3) (SQRNdbnseq.py:891-893,1256-1258): the decimal nearest to the EXACT binary value, ties to
// even, then the double nearest to that decimal.  x = m 2^e exactly, so 1000 x = (1000 m) / 2^-e is an integer
// quotient and remainder in 64-bit arithmetic; k / 1000.0 is one correctly rounded division of exact integers, i.e. what
// strtod returns for the decimal's text.  Exact for |x| < 2^53 / 1000; beyond, *inexact is set (the host tail takes over).
__device__ __forceinline__ double sq_round3(double x, uint32_t *inexact)
{
    if (!(x == x) || fabs(x) == INFINITY) return x;
    const double ax = fabs(x);
    if (ax >= 4503599627370496.0) return x;                            // >= 2^52: an integer
    if (ax >= 9.0e12) { *inexact = 1; return x; }
    const unsigned long long bits = (unsigned long long)__double_as_longlong(ax);
    const int ex = (int)((bits >> 52) & 0x7FFull);
    unsigned long long m = bits & 0xFFFFFFFFFFFFFull;
    int e;
    if (ex == 0) e = -1074; else { m |= 1ull << 52; e = ex - 1075; }
    unsigned long long k;
    if (e >= 0) k = (m << e) * 1000ull;                                // (ax < 9e12: no overflow)
    else {
        const int E = -e;
        const unsigned long long p = m * 1000ull;                      // < 2^63
        if (E >= 64) k = 0ull;                                         // 1000 x < 1/2
        else {
            const unsigned long long q = p >> E, r = p & ((1ull << E) - 1ull), half = 1ull << (E - 1);
            k = q + ((r > half || (r == half && (q & 1ull))) ? 1ull : 0ull);
        }
    }
    const double res = (double)k / 1000.0;
    return x < 0 ? -res : res;
}

__device__ __forceinline__ unsigned long long sq_mix_stem(int i, int j, int len)
{
    unsigned long long x = ((unsigned long long)(uint32_t)i << 40) ^ ((unsigned long long)(uint32_t)j << 20) ^ (unsigned long long)(uint32_t)len;
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// stem q of a log entry, whichever array holds it
__device__ __forceinline__ SqPoolStem sq_fin_stem(const SqTailIO &t, const SqPoolFin &F, int q)
{
    if (F.pad == SQ_FIN_SRC_CHAIN) { const SqChainStem x = t.chain_stems[F.stem_off + q]; return SqPoolStem{(int16_t)x.i, (int16_t)x.j, (int16_t)x.len, 0}; }
    return t.fin_stems[F.stem_off + q];
}
__device__ __forceinline__ SqPoolStem *sq_fin_canon(const SqTailIO &t, const SqPoolFin &F)
{
    return t.cstems + (F.pad == SQ_FIN_SRC_CHAIN ? (size_t)t.fin_stem_cap + F.stem_off : (size_t)F.stem_off);
}

// ---- grouping by job ---------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void sq_tail_count_kernel(SqTailIO t)
{
    const uint32_t nfin = min(*t.nfin_ptr, t.fin_cap);
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nfin; e += gridDim.x * 256) atomicAdd(&t.job_cnt[t.fin[e].job], 1u);
}

extern "C" __global__ __launch_bounds__(1024) void sq_tail_scan_kernel(SqTailIO t)
{
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x, n = t.njobs;
    const int ipt = (n + 1023) / 1024;
    const int lo = min(tid * ipt, n), hi = min(lo + ipt, n);
    uint32_t sum = 0;
    for (int q = lo; q < hi; q++) sum += t.job_cnt[q];
    s_part[tid] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (int q = lo; q < hi; q++) { t.job_start[q] = run; t.job_fill[q] = run; run += t.job_cnt[q]; }
    if (tid == 1023) t.job_start[n] = s_part[1023];
    __syncthreads();
    for (int s = tid; s < t.nseq; s += 1024) {
        const uint32_t a = t.job_start[t.seq_job0[s]], b = t.job_start[t.seq_job0[s + 1]];
        t.seqs[s].first = a; t.seqs[s].count = b - a;
    }
    if (tid == 0) {
        // the log overflowed (entries, or stems: fin_ctr[1] counts them, fin_ctr[2] is the writers' flag -- an entry whose
        // stems found no room is a tombstone without stems): nothing below may be trusted, the host reports it
        if (*t.nfin_ptr > t.fin_cap || t.nfin_ptr[1] > t.fin_stem_cap || t.nfin_ptr[2] != 0) { *t.fallback = 1; t.h_totals[6] = 1; }
        t.h_totals[3] = (long long)min(*t.nfin_ptr, t.fin_cap);
    }
}

extern "C" __global__ __launch_bounds__(256) void sq_tail_scatter_kernel(SqTailIO t)
{
    const uint32_t nfin = min(*t.nfin_ptr, t.fin_cap);
    for (uint32_t e = blockIdx.x * 256 + threadIdx.x; e < nfin; e += gridDim.x * 256) {
        const uint32_t p = atomicAdd(&t.job_fill[t.fin[e].job], 1u);
        t.ord[p] = e;
    }
}

// ---- ScoreStruct (:861-899) of one stem list by one wave -----------------------------------------------------------
// stems: get(q) for q < T, in the list's own order (the sum over stems runs in that order).  s_bits: the wave's LDS
// bitmap of paired positions (only touched when the reactivities are not all 0.5).  Every lane returns the same values.
// codes: the sequence's letter codes (the ranking kernel stages them in LDS once per sequence: a stem's cells were two
// dependent byte loads from global memory each); nsep: its separators (counted once per sequence).
template <class Get>
__device__ __forceinline__ void sq_score_struct_wave(const SqDevCtx &c, const SqTailIO &t, const SqJob &jb, Get get, int T, uint32_t *s_bits,
                                                     int lane, double out[3], uint32_t *fallback, const uint8_t *codes, int nsep)
{
    const int n = jb.n;
    const bool marks = !jb.default_reacts;
    if (marks) {
        for (int w = lane; w < (n + 31) / 32; w += 64) s_bits[w] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    double thescore = 0;
    for (int q0 = 0; q0 < T; q0 += 64) {
        const int q = q0 + lane;
        double f = 0;
        if (q < T) {
            const SqPoolStem s = get(q);
            double bpsum = 0;
            for (int k = 0; k < s.len; k++) {
                const int v = s.i + k, w = s.j - k;
                const int a = codes[v], b = codes[w];
                const int A = 0, C = 2, G = 6, U = 20;
                double bp = 0.0;
                if ((a == G && b == U) || (a == U && b == G)) bp = -0.5;
                else if ((a == A && b == U) || (a == U && b == A)) bp = 1.5;
                else if ((a == G && b == C) || (a == C && b == G)) bp = 4.0;
                bpsum += bp;                                           // (multiples of 1/2: exact in any order)
                if (marks) { atomicOr(&s_bits[v >> 5], 1u << (v & 31)); atomicOr(&s_bits[w >> 5], 1u << (w & 31)); }
            }
            if (bpsum > 0) {                                           // :884  bpsum ** 1.7 through the host libm's table
                const int idx = (int)(bpsum * 2.0);
                if (idx < t.pow17h_len) f = t.pow17h[idx]; else *fallback = 1;
            }
        }
        const int cnt = min(64, T - q0);
        const int flo = __double2loint(f), fhi = __double2hiint(f);
        for (int u = 0; u < cnt; u++)                                  // the reference's order of additions (u is uniform: v_readlane)
            thescore += __hiloint2double(__builtin_amdgcn_readlane(fhi, u), __builtin_amdgcn_readlane(flo, u));
    }
    double reactscore;
    if (!marks) {
        const int sep = nsep;
        // every term is exactly 0.5: the sum is 0.5 (n - sep) whatever the order
        reactscore = 1 - (0.5 * (double)(n - sep)) / (double)(n - sep);
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const double *reacts = c.reacts + jb.pos_off;
        int sep = 0;
        double acc = 0;                                                // :894-896, position by position (every lane the same)
        for (int i = 0; i < n; i++) {
            const int cd = codes[i];
            if (cd == SQ_CODE_SEP1 || cd == SQ_CODE_SEP2) { sep++; continue; }
            const double r = reacts[i];
            acc += ((s_bits[i >> 5] >> (i & 31)) & 1u) ? r : 1 - r;
        }
        reactscore = 1 - acc / (double)(n - sep);
    }
    uint32_t inexact = 0;
    out[0] = sq_round3(thescore * reactscore, &inexact);
    out[1] = sq_round3(thescore, &inexact);
    out[2] = sq_round3(reactscore, &inexact);
    if (inexact) *fallback = 1;
}

// TP / FP / FN / FS / PR / RC of a set of stems against the known structure (:1252-1258)
__device__ __forceinline__ void sq_prf_wave(const int16_t *refp, int known_n, const SqPoolStem *cs, int cn, int lane, double m[6], uint32_t *fallback)
{
    int tp = 0, np = 0;
    for (int q = lane; q < cn; q += 64) {
        const SqPoolStem s = cs[q];
        np += s.len;
        for (int k = 0; k < s.len; k++) tp += refp[s.i + k] == (int16_t)(s.j - k) ? 1 : 0;
    }
    tp = sq_wave_sum32(tp); np = sq_wave_sum32(np);
    const int fp = np - tp, fn = known_n - tp;
    uint32_t inexact = 0;
    m[0] = tp; m[1] = fp; m[2] = fn;
    m[3] = (2 * tp + fp + fn) ? sq_round3(2.0 * tp / (double)(2 * tp + fp + fn), &inexact) : 1.0;
    m[4] = (tp + fp) ? sq_round3((double)tp / (double)(tp + fp), &inexact) : 1.0;
    m[5] = (tp + fn) ? sq_round3((double)tp / (double)(tp + fn), &inexact) : 1.0;
    if (inexact) *fallback = 1;
}

#define SQ_TAIL_THREADS 256            // a block's threads on a crowded chip; a batch alone with several jobs per sequence: SQ_TAIL_THREADS_WIDE
#define SQ_TAIL_THREADS_WIDE 1024
#define SQ_TAIL_BITWORDS 1024          // LDS bitmap words per wave: sequences up to 32768 nt
#define SQ_TAIL_PIECES 12              // wave-sized jobs of the ranking kernel's last phase that are dealt to the waves (2 + toplim)

extern "C" __global__ __launch_bounds__(SQ_TAIL_THREADS_WIDE) void sq_tail_rank_kernel(SqDevCtx c, SqTailIO t, int bitwords, int keycap, int refp_lds)
{
    // (one bitmap of the sequence's positions per wave, in the block's DYNAMIC LDS sized for the batch's longest sequence:
    // a static array for 32,768 nt cost every block 16 KB -- 20 bytes do for 150 nt)
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits_dyn[];   // [waves of the block][bitwords], then keycap x (3 rank keys, priority)
    __shared__ uint32_t s_wsum[SQ_TAIL_THREADS_WIDE / 64];
    __shared__ uint32_t s_run;
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nthr = blockDim.x, nwv = nthr >> 6;                  // (the entry-per-wave passes b and e are as fast as the block has waves)
    double *const s_key = reinterpret_cast<double *>(s_bits_dyn + ((nwv * (uint32_t)bitwords + 1u) & ~1u));   // [keycap][3], in rankby order
    uint8_t *const s_pri = reinterpret_cast<uint8_t *>(s_key + 3 * (size_t)keycap);                          // [keycap]
    uint8_t *const s_codes = s_pri + (((size_t)keycap + 7) & ~(size_t)7);                                    // [32 x bitwords] the sequence's letter codes
    int16_t *const s_refp = reinterpret_cast<int16_t *>(s_codes + 32 * (size_t)bitwords);                   // [32 x bitwords] partners in the known structure (refp_lds)
    __shared__ int s_nsep;
    SqTailSeq &S = t.seqs[s];
    const uint32_t first = S.first, M = S.count;
    const int j0 = t.seq_job0[s], j1 = t.seq_job0[s + 1];
    const SqJob jb = c.jobs[j0];                                        // (every job of the sequence: same n, letters, reactivities)
    const int n = jb.n;
    auto bail = [&]() {                                                 // the host tail takes the batch
        if (tid == 0) { *t.fallback = 1; S.D = 0; S.nshow = 0; S.nprf = 0; S.rec_bytes = 0; S.txt_bytes = 0; S.evals = 0; }
    };
    if (M > SQ_TAIL_MAXM || n > 32 * bitwords) { bail(); return; }
    if (tid == 0) s_nsep = 0;
    __syncthreads();
    {
        int sep = 0;
        const bool stage_ref = refp_lds && t.ref_n && t.ref_n[s] >= 0;
        for (int p = tid; p < n; p += nthr) {
            const uint8_t cd = c.codes[jb.pos_off + p];
            s_codes[p] = cd;
            sep += (cd == SQ_CODE_SEP1 || cd == SQ_CODE_SEP2) ? 1 : 0;
            if (stage_ref) s_refp[p] = t.refp[jb.pos_off + p];      // (the metrics and the known structure's stems read it bp by bp)
        }
        if (sep) atomicAdd(&s_nsep, sep);
    }
    __syncthreads();
#ifdef SQ_TAIL_PROF
    long long _tp[8];
#endif
#ifdef SQ_TAIL_PROF
    if (tid == 0) _tp[0] = wall_clock64();
#endif
    // ---- a. every job's entries into finstemsets order: (kind, pos) ascending ----
    for (int j = j0; j < j1; j++) {
        const uint32_t lo = t.job_start[j], m = t.job_start[j + 1] - lo;
        if (m < 2) continue;
        for (uint32_t x = tid; x < m; x += nthr) {
            const uint32_t e = t.ord[lo + x];
            const unsigned long long kx = ((unsigned long long)t.fin[e].round_kind << 32) | (uint32_t)t.fin[e].pos;
            uint32_t r = 0;
            for (uint32_t y = 0; y < m; y++) {
                const uint32_t f = t.ord[lo + y];
                const unsigned long long ky = ((unsigned long long)t.fin[f].round_kind << 32) | (uint32_t)t.fin[f].pos;
                r += (ky < kx || (ky == kx && f < e)) ? 1u : 0u;
            }
            t.ord2[lo + r] = e;
        }
        __syncthreads();
        for (uint32_t x = tid; x < m; x += nthr) t.ord[lo + x] = t.ord2[lo + x];
        __syncthreads();
    }
    __syncthreads();
#ifdef SQ_TAIL_PROF
    if (tid == 0) _tp[1] = wall_clock64();
#endif
    // ---- b. canonical stems (maximal stacks in ascending order of i) + hash, one wave per entry ----
    // The hash only groups candidates for pass c (which compares stem by stem): sum over the canonical stems of
    // mix(stem) x (odd constant + 2 x position), the same from either form below.
    // Structures of up to 64 stems (all but pathological ones): a stem per lane, in registers -- rank among the stems with
    // len > 0 by a uniform loop of readlanes, the sorted list by a shuffle, stacks joined by comparing neighbours (a merged
    // stack ends where its last stem ends: the serial rule's comparison with the merged predecessor is the comparison with
    // the stem before), lengths from the segment's last stem.  (Until round 4 lane 0 walked the sorted list in global
    // memory: a chain of dependent loads per stem, 100 of the kernel's 290 us on SRtest150.)
    for (uint32_t x = wave; x < M; x += nwv) {
        const uint32_t e = t.ord[first + x];
        const SqPoolFin F = t.fin[e];
        const int T = F.nstems;
        SqPoolStem *cs = sq_fin_canon(t, F);
        if (T <= 64) {
            const SqPoolStem mine = lane < T ? sq_fin_stem(t, F, lane) : SqPoolStem{0, 0, 0, 0};
            const bool valid = lane < T && mine.len > 0;
            const unsigned long long vm = __ballot(valid);
            const int V = (int)__popcll(vm);
            const int mi = mine.i;
            int r = 0;
            for (int p = 0; p < T; p++) {
                if (!((vm >> p) & 1ull)) continue;                        // (uniform)
                const int ip = __builtin_amdgcn_readlane(mi, p);
                r += (ip < mi || (ip == mi && p < lane)) ? 1 : 0;
            }
            int src = 0;
            for (int p = 0; p < T; p++) {
                if (!((vm >> p) & 1ull)) continue;
                const int rp = __builtin_amdgcn_readlane(r, p);
                if (rp == lane) src = p;
            }
            const int packed_ij = ((int)(uint16_t)mine.i) | ((int)(uint16_t)mine.j << 16);
            const int sij = __shfl(packed_ij, src, 64), slen = __shfl((int)mine.len, src, 64);
            const int si = (int)(int16_t)(sij & 0xFFFF), sj = (int)(int16_t)((uint32_t)sij >> 16);
            const int pi = __shfl_up(si, 1, 64), pj = __shfl_up(sj, 1, 64), pl = __shfl_up(slen, 1, 64);
            const bool in = lane < V;
            const bool joins = in && lane > 0 && si == pi + pl && sj == pj - pl;
            const bool bad = in && lane > 0 && si < pi + pl;             // 5' strands overlap: not disjoint stacks
            const unsigned long long heads = __ballot(in && !joins);
            const int seg = (int)__popcll(heads & ((2ull << lane) - 1ull)) - 1;
            const unsigned long long above = lane < 63 ? heads >> (lane + 1) : 0ull;
            const int last = above ? lane + (int)__ffsll((long long)above) - 1 : V - 1;   // the segment's last stem
            const int endi = __shfl(si + slen, last < 0 ? 0 : last, 64);
            const bool head = in && !joins;
            unsigned long long contrib = 0ull;
            if (head) {
                const int total = endi - si;
                cs[seg] = SqPoolStem{(int16_t)si, (int16_t)sj, (int16_t)total, 0};
                contrib = sq_mix_stem(si, sj, total) * (0xD6E8FEB86659FD93ull + 2ull * (unsigned long long)seg);
            }
            for (int off = 32; off > 0; off >>= 1) contrib += (unsigned long long)__shfl_xor((long long)contrib, off, 64);
            if (lane == 0) { t.cs_n[first + x] = (uint32_t)__popcll(heads); t.hash[first + x] = contrib; t.mask[first + x] = 0ull; }
            if (__ballot(bad) != 0ull && lane == 0) *t.fallback = 1;
            continue;
        }
        for (int q = lane; q < T; q += 64) {                            // rank by i (stems of a structure start at distinct positions)
            const SqPoolStem sq = sq_fin_stem(t, F, q);
            int r = 0;
            for (int p = 0; p < T; p++) { const SqPoolStem sp = sq_fin_stem(t, F, p); r += (sp.i < sq.i || (sp.i == sq.i && p < q)) ? 1 : 0; }
            cs[r] = sq;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            int cn = 0; bool bad = false;
            for (int q = 0; q < T; q++) {
                const SqPoolStem sq = cs[q];
                if (sq.len <= 0) continue;
                if (cn) {
                    SqPoolStem &p = cs[cn - 1];
                    if (sq.i < p.i + p.len) { bad = true; break; }      // 5' strands overlap: not disjoint stacks
                    if (sq.i == p.i + p.len && sq.j == p.j - p.len) { p.len = (int16_t)(p.len + sq.len); continue; }
                }
                cs[cn++] = sq;
            }
            unsigned long long h = 0ull;
            for (int q = 0; q < cn; q++) h += sq_mix_stem(cs[q].i, cs[q].j, cs[q].len) * (0xD6E8FEB86659FD93ull + 2ull * (unsigned long long)q);
            t.cs_n[first + x] = (uint32_t)cn; t.hash[first + x] = h; t.mask[first + x] = 0ull;
            if (bad) *t.fallback = 1;
        }
    }
    __threadfence_block();
    __syncthreads();
#ifdef SQ_TAIL_PROF
    if (tid == 0) _tp[2] = wall_clock64();
#endif
    // ---- c. first occurrence of every base-pair set (:1203-1207) ----
    for (uint32_t x = tid; x < M; x += nthr) {
        const unsigned long long hx = t.hash[first + x];
        const uint32_t cn = t.cs_n[first + x];
        const SqPoolStem *cx = sq_fin_canon(t, t.fin[t.ord[first + x]]);
        uint32_t rep = x;
        for (uint32_t y = 0; y < x; y++) {
            if (t.hash[first + y] != hx || t.cs_n[first + y] != cn) continue;
            const SqPoolStem *cy = sq_fin_canon(t, t.fin[t.ord[first + y]]);
            bool same = true;
            for (uint32_t q = 0; q < cn && same; q++) same = cx[q].i == cy[q].i && cx[q].j == cy[q].j && cx[q].len == cy[q].len;
            if (same) { rep = y; break; }
        }
        t.rep[first + x] = rep;
    }
    __threadfence_block();
    __syncthreads();
#ifdef SQ_TAIL_PROF
    if (tid == 0) _tp[3] = wall_clock64();
#endif
    // ---- d. producers of every distinct structure (:1207-1220) and the distinct ones in list order ----
    if (tid == 0) s_run = 0;
    __syncthreads();
    for (uint32_t x0 = 0; x0 < M; x0 += nthr) {
        const uint32_t x = x0 + tid;
        bool dist = false;
        if (x < M) {
            const uint32_t rep = t.rep[first + x];
            atomicOr(&t.mask[first + rep], 1ull << (uint32_t)(t.fin[t.ord[first + x]].job - j0));
            dist = rep == x;
        }
        const unsigned long long bal = __ballot(dist);
        if (lane == 0) s_wsum[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_run;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        if (dist) t.dlist[first + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = x;
        __syncthreads();
        if (tid == 0) { uint32_t a = 0; for (uint32_t w = 0; w < nwv; w++) a += s_wsum[w]; s_run += a; }
        __syncthreads();
    }
    const uint32_t D = s_run;
    __threadfence_block();
    __syncthreads();
#ifdef SQ_TAIL_PROF
    if (tid == 0) _tp[4] = wall_clock64();
#endif
    // ---- e. ScoreStruct of the distinct structures: the FIRST producer's stem list, in its own order (:1208) ----
    for (uint32_t k = wave; k < D; k += nwv) {
        const uint32_t x = t.dlist[first + k];
        const SqPoolFin F = t.fin[t.ord[first + x]];
        double sc[3];
        sq_score_struct_wave(c, t, jb, [&](int q) { return sq_fin_stem(t, F, q); }, F.nstems, s_bits_dyn + (size_t)wave * bitwords, lane, sc, t.fallback, s_codes, s_nsep);
        if (lane == 0) { t.scores[3 * (size_t)(first + x)] = sc[0]; t.scores[3 * (size_t)(first + x) + 1] = sc[1]; t.scores[3 * (size_t)(first + x) + 2] = sc[2]; }
    }
    __threadfence_block();
    __syncthreads();
#ifdef SQ_TAIL_PROF
    if (tid == 0) _tp[5] = wall_clock64();
#endif
    // ---- f. RankStructs (:907-913): stable descending sort on the rankby keys, prioritised paramsets first ----
    if (D <= (uint32_t)keycap) {
        // the keys of the distinct structures side by side in LDS: the D x D comparisons then read broadcast LDS words
        // instead of chasing dlist -> scores through global memory (82 -> 10 us for 185 structures)
        for (uint32_t k = tid; k < D; k += nthr) {
            const uint32_t x = t.dlist[first + k];
            const double *sx = t.scores + 3 * (size_t)(first + x);
            for (int q = 0; q < 3; q++) s_key[3 * k + q] = sx[t.rankby[q]];
            s_pri[k] = (t.mask[first + x] & t.priority_mask) != 0ull ? 1 : 0;
        }
        __syncthreads();
        for (uint32_t k = tid; k < D; k += nthr) {
            const double b0 = s_key[3 * k], b1 = s_key[3 * k + 1], b2 = s_key[3 * k + 2];
            const bool px = s_pri[k] != 0;
            uint32_t r = 0;
            for (uint32_t l = 0; l < D; l++) {
                const double a0 = s_key[3 * l], a1 = s_key[3 * l + 1], a2 = s_key[3 * l + 2];
                const bool py = s_pri[l] != 0;
                bool before;
                if (t.priority_mask && px != py) before = py;           // :912-913 (a stable partition)
                else {
                    int cmp = a0 > b0 ? 1 : (a0 < b0 ? -1 : 0);          // > 0: l sorts before k
                    if (cmp == 0) cmp = a1 > b1 ? 1 : (a1 < b1 ? -1 : 0);
                    if (cmp == 0) cmp = a2 > b2 ? 1 : (a2 < b2 ? -1 : 0);
                    before = cmp > 0 || (cmp == 0 && l < k);
                }
                r += (before && l != k) ? 1u : 0u;
            }
            t.rlist[first + r] = t.dlist[first + k];
        }
    } else
    for (uint32_t k = tid; k < D; k += nthr) {
        const uint32_t x = t.dlist[first + k];
        const double *sx = t.scores + 3 * (size_t)(first + x);
        const bool px = (t.mask[first + x] & t.priority_mask) != 0ull;
        uint32_t r = 0;
        for (uint32_t l = 0; l < D; l++) {
            if (l == k) continue;
            const uint32_t y = t.dlist[first + l];
            const double *sy = t.scores + 3 * (size_t)(first + y);
            const bool py = (t.mask[first + y] & t.priority_mask) != 0ull;
            bool before;
            if (t.priority_mask && px != py) before = py;               // :912-913 (a stable partition)
            else {
                int cmp = 0;                                            // > 0: y sorts before x
                for (int q = 0; q < 3 && cmp == 0; q++) {
                    const double a = sy[t.rankby[q]], b = sx[t.rankby[q]];
                    cmp = a > b ? 1 : (a < b ? -1 : 0);
                }
                before = cmp > 0 || (cmp == 0 && l < k);
            }
            r += before ? 1u : 0u;
        }
        t.rlist[first + r] = x;
    }
    __threadfence_block();
    __syncthreads();
#ifdef SQ_TAIL_PROF
    if (tid == 0) _tp[6] = wall_clock64();
#endif
    // ---- g. what is shown, metrics of the top ranks, sizes ----
    const uint32_t nshow = t.result_limit > 0 ? min(D, (uint32_t)t.result_limit) : D;
    const int known_n = t.ref_n ? t.ref_n[s] : -1;
    const bool has_ref = known_n >= 0;
    const uint32_t nprf = has_ref ? min(D, (uint32_t)max(t.toplim, 1)) : 0u;
    // The pieces of this phase -- ScoreStruct of the known structure, the consensus' metrics, the metrics of the top ranks -- are
    // independent wave-sized jobs: they are dealt to the block's waves (until round 5 the first wave did them one after the
    // other while the others were through: 46 of the kernel's 200 us for SRtest150's largest record) and lane 0 collects them.
    __shared__ double s_piece[SQ_TAIL_PIECES][6];                       // [0]: ref scores (3), [1]: consensus, [2 + r]: rank r
    const int16_t *refp = has_ref ? (refp_lds ? s_refp : t.refp + jb.pos_off) : nullptr;
    const bool spread = has_ref && nprf + 2 <= SQ_TAIL_PIECES;
    auto ref_scores = [&](double (&ref_sc)[3]) {
        // ReferenceScores (:958-970): ScoreStruct of PairsToStems(sorted pairs of the known structure); the stems are
        // read off the partner array on the fly: a pair (i, p), i < p, starts a stem unless (i - 1, p + 1) is a pair
        // (the starts are listed in the LDS of the rank keys, which pass f is done with, when they fit)
        int16_t *const s_start = reinterpret_cast<int16_t *>(s_key);
        const int startcap = 12 * keycap;
        int nst = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            bool start = false;
            if (i < n) { const int p = refp[i]; start = p > i && !(i > 0 && refp[i - 1] == p + 1); }
            const unsigned long long bal = __ballot(start);
            if (start) { const int idx = nst + (int)__popcll(bal & ((1ull << lane) - 1ull)); if (idx < startcap) s_start[idx] = (int16_t)i; }
            nst += __popcll(bal);
        }
        const bool listed = nst <= startcap;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        auto ref_stem = [&](int q) {                                // q-th stem start in ascending order of i
            int seen = 0;
            SqPoolStem out{0, 0, 0, 0};
            if (listed) {
                const int i = s_start[q], p = refp[i];
                int len = 1;
                while (i + len < n && refp[i + len] == p - len && p - len > i + len) len++;
                return SqPoolStem{(int16_t)i, (int16_t)p, (int16_t)len, 0};
            }
            for (int i = 0; i < n; i++) {
                const int p = refp[i];
                if (!(p > i) || (i > 0 && refp[i - 1] == p + 1)) continue;
                if (seen++ != q) continue;
                int len = 1;
                while (i + len < n && refp[i + len] == p - len && p - len > i + len) len++;
                out = SqPoolStem{(int16_t)i, (int16_t)p, (int16_t)len, 0};
                break;
            }
            return out;
        };
        sq_score_struct_wave(c, t, jb, ref_stem, nst, s_bits_dyn + (size_t)wave * bitwords, lane, ref_sc, t.fallback, s_codes, s_nsep);
    };
    auto cons_metrics = [&](double (&m)[6]) {
        // consensus = the top-ranked structure (conslim == 1, :845-858,1236) or nothing (conslim == 0, or no structure)
        if (D > 0 && t.conslim == 1) {
            const uint32_t x0 = t.rlist[first];
            sq_prf_wave(refp, known_n, sq_fin_canon(t, t.fin[t.ord[first + x0]]), (int)t.cs_n[first + x0], lane, m, t.fallback);
        } else sq_prf_wave(refp, known_n, nullptr, 0, lane, m, t.fallback);
    };
    auto rank_metrics = [&](uint32_t r, double (&m)[6]) {
        const uint32_t x = t.rlist[first + r];
        sq_prf_wave(refp, known_n, sq_fin_canon(t, t.fin[t.ord[first + x]]), (int)t.cs_n[first + x], lane, m, t.fallback);
    };
    if (spread) {
        for (uint32_t pc = wave; pc < nprf + 2; pc += nwv) {
            double m[6] = {0, 0, 0, 0, 0, 0};
            if (pc == 0) { double sc[3]; ref_scores(sc); m[0] = sc[0]; m[1] = sc[1]; m[2] = sc[2]; }
            else if (pc == 1) cons_metrics(m);
            else rank_metrics(pc - 2, m);
            if (lane == 0) for (int q = 0; q < 6; q++) s_piece[pc][q] = m[q];
        }
        __syncthreads();
    }
    if (wave == 0) {
        double cons_m[6], best_m[7], ref_sc[3];
        for (int q = 0; q < 6; q++) cons_m[q] = NAN;
        for (int q = 0; q < 7; q++) best_m[q] = NAN;
        for (int q = 0; q < 3; q++) ref_sc[q] = NAN;
        if (has_ref) {
            double best = -1;                                           // :1262-1283
            if (spread) {
                for (int q = 0; q < 6; q++) cons_m[q] = s_piece[1][q];
                for (uint32_t r = 0; r < nprf; r++)
                    if (s_piece[2 + r][3] > best) { best = s_piece[2 + r][3]; for (int q = 0; q < 6; q++) best_m[q] = s_piece[2 + r][q]; best_m[6] = (double)(r + 1); }
                for (int q = 0; q < 3; q++) ref_sc[q] = s_piece[0][q];
            } else {
                cons_metrics(cons_m);
                for (uint32_t r = 0; r < nprf; r++) {
                    double m[6];
                    rank_metrics(r, m);
                    if (m[3] > best) { best = m[3]; for (int q = 0; q < 6; q++) best_m[q] = m[q]; best_m[6] = (double)(r + 1); }
                }
                ref_scores(ref_sc);
            }
        }
        if (lane == 0) {
            double *met = t.scores + 3 * (size_t)t.fin_cap + 16 * (size_t)s;   // per-sequence metrics behind the entry scores
            for (int q = 0; q < 6; q++) met[q] = cons_m[q];
            for (int q = 0; q < 7; q++) met[6 + q] = best_m[q];
            for (int q = 0; q < 3; q++) met[13 + q] = ref_sc[q];
            long long ev = 0;
            for (int j = j0; j < j1; j++) ev += t.job_evals[j];
            for (uint32_t x = 0; x < M; x++) ev += t.fin[t.ord[first + x]].round_kind < SQ_FIN_KIND_G0 ? 1 : 0;   // one AnnotateStems pass per E / H / N set
            S.D = D; S.nshow = nshow; S.nprf = nprf;
            S.rec_bytes = (32 + 128 + 32 * (long long)nshow + 2 * (1 + (long long)nshow) * n + 7) & ~7ll;
            S.txt_bytes = (1 + (long long)nshow) * n;
            S.evals = ev;
        }
    }
#ifdef SQ_TAIL_PROF
    __syncthreads();
    if (tid == 0) {
        _tp[7] = wall_clock64();
        if (_tp[7] - _tp[0] > 15000)
            printf("tail s=%d n=%d M=%u D=%u jobs=%d | us: sort %.0f canon %.0f first %.0f producers %.0f score %.0f rank %.0f show %.0f\n", s, n, M, D, j1 - j0,
                   (_tp[1] - _tp[0]) * 0.01, (_tp[2] - _tp[1]) * 0.01, (_tp[3] - _tp[2]) * 0.01, (_tp[4] - _tp[3]) * 0.01, (_tp[5] - _tp[4]) * 0.01,
                   (_tp[6] - _tp[5]) * 0.01, (_tp[7] - _tp[6]) * 0.01);
    }
#endif
}

extern "C" __global__ __launch_bounds__(1024) void sq_tail_offsets_kernel(SqTailIO t, volatile uint32_t *h_seq, uint32_t seq)
{
    __shared__ long long s_a[1024], s_b[1024];
    __shared__ uint32_t s_maxu;
    const int tid = threadIdx.x, n = t.nseq;
    const int ipt = (n + 1023) / 1024;
    const int lo = min(tid * ipt, n), hi = min(lo + ipt, n);
    long long a = 0, b = 0;
    uint32_t mu = 0;
    for (int q = lo; q < hi; q++) { a += t.seqs[q].rec_bytes; b += t.seqs[q].txt_bytes; mu = max(mu, t.seqs[q].nshow); }
    s_a[tid] = a; s_b[tid] = b;
    if (tid == 0) s_maxu = 0;
    __syncthreads();
    atomicMax(&s_maxu, mu);
    for (int d = 1; d < 1024; d <<= 1) {
        const long long va = tid >= d ? s_a[tid - d] : 0, vb = tid >= d ? s_b[tid - d] : 0;
        __syncthreads();
        s_a[tid] += va; s_b[tid] += vb;
        __syncthreads();
    }
    long long ra = s_a[tid] - a, rb = s_b[tid] - b;
    for (int q = lo; q < hi; q++) { t.seqs[q].rec_off = ra; t.seqs[q].txt_off = rb; ra += t.seqs[q].rec_bytes; rb += t.seqs[q].txt_bytes; }
    __syncthreads();
    if (tid == 0) {
        t.h_totals[0] = s_a[1023]; t.h_totals[1] = s_b[1023]; t.h_totals[2] = (long long)*t.fallback; t.h_totals[4] = (long long)s_maxu;
        sq_host_write_flush(t.h_totals);
        *h_seq = seq;
    }
}

// ---- levels + the packed record --------------------------------------------------------------------------------------
// grid (nseq, Y): block (s, y) is one wave and forms the rows of structures y, y + Y, ... of sequence s; y == 0 also writes
// the header, the scores / masks, the metrics and the consensus row.
extern "C" __global__ __launch_bounds__(64) void sq_tail_pack_kernel(SqDevCtx c, SqTailIO t, int rowcap, long long rec_cap, long long txt_cap)
{
    extern __shared__ __attribute__((aligned(16))) char sq_tail_dyn[];   // [rowcap int16 row][sq_extend_lds_bytes(tmax)]
    int16_t *row = reinterpret_cast<int16_t *>(sq_tail_dyn);
    SqExtendLds L = sq_extend_lds(sq_tail_dyn + (((size_t)rowcap * 2 + 15) & ~(size_t)15), t.tmax);
    const int s = blockIdx.x, y = blockIdx.y, Y = gridDim.y, lane = threadIdx.x;
    const SqTailSeq S = t.seqs[s];
    const SqJob jb = c.jobs[t.seq_job0[s]];
    const int n = jb.n;
    const uint32_t first = S.first, ns = S.nshow;
    // (launched before the host has seen the sizes whenever buffers of an earlier fold exist: a record that does not fit
    // them is not written, the host grows the buffers and repeats the launch)
    if (S.rec_off + S.rec_bytes > rec_cap || S.txt_off + S.txt_bytes > txt_cap) { if (threadIdx.x == 0) t.fallback[1] = 1; return; }
    char *rec = t.rec_buf + S.rec_off;
    char *txt = t.txt_buf + S.txt_off;
    int16_t *lev0 = reinterpret_cast<int16_t *>(rec + 160 + 32 * (size_t)ns);
    bool deep = false;
    auto put_row = [&](size_t r) {                                      // the LDS row -> level row r and text row r of the record
        int16_t *dst = lev0 + r * (size_t)n;
        char *tx = txt + r * (size_t)n;
        for (int i = lane; i < n; i += 64) {
            const int v = row[i];
            dst[i] = (int16_t)v;
            char ch = '.';
            if (v > 0) { if (v <= 4) ch = "([{<"[v - 1]; else if (v <= 30) ch = (char)('A' + v - 5); else deep = true; }
            else if (v < 0) { if (v >= -4) ch = ")]}>"[-v - 1]; else if (v >= -30) ch = (char)('a' - v - 5); else deep = true; }
            tx[i] = ch;
        }
    };
    for (uint32_t r = y; r < ns; r += Y) {
        const uint32_t x = t.rlist[first + r];
        const SqPoolStem *cs = sq_fin_canon(t, t.fin[t.ord[first + x]]);
        const int T = (int)t.cs_n[first + x];
        for (int i = lane; i < n; i += 64) row[i] = 0;
        bool anyc = false;
        for (int q = lane; q < T; q += 64) { const SqPoolStem st = cs[q]; L.i[q] = st.i; L.j[q] = st.j; L.len[q] = st.len; }
        __syncthreads();
        for (int q = lane; q < T; q += 64) {                            // crossing weights (:121-124)
            const int qi = L.i[q], qj = L.j[q];
            int cc = 0;
            for (int p = 0; p < T; p++) if (sq_chain_cross(qi, qj, L.i[p], L.j[p])) cc += L.len[p];
            L.cc[q] = cc;
            anyc |= cc != 0;
        }
        const bool cross = __ballot(anyc) != 0ull;
        __syncthreads();
        if (cross) sq_stem_levels_wave(L, T, lane, t.fallback);          // (more than 64 levels: the host tail reports it)
        for (int q = lane; q < T; q += 64) {
            const int lv = cross ? L.lvl[q] : 1;
            const int si = L.i[q], sj = L.j[q], sl = L.len[q];
            for (int k = 0; k < sl; k++) { row[si + k] = (int16_t)lv; row[sj - k] = (int16_t)-lv; }
        }
        __syncthreads();
        put_row((size_t)r + 1);
        if (r == 0 && t.conslim == 1) put_row(0);                       // the consensus is the top structure (:1236)
        __syncthreads();
    }
    if (y == 0) {
        if (ns == 0 || t.conslim != 1) {                         // no structure (or conslim == 0): an empty consensus
            for (int i = lane; i < n; i += 64) row[i] = 0;
            __syncthreads();
            put_row(0);
        }
        const double *met = t.scores + 3 * (size_t)t.fin_cap + 16 * (size_t)s;
        if (lane == 0) {
            long long *hdr = reinterpret_cast<long long *>(rec);
            hdr[0] = (long long)ns; hdr[1] = n; hdr[2] = (t.ref_n && t.ref_n[s] >= 0) ? 1 : 0; hdr[3] = S.evals;
        }
        if (lane < 16) reinterpret_cast<double *>(rec + 32)[lane] = met[lane];
        {
            // the record is padded to a multiple of 8 bytes: the pad is part of what sq_result_pack hands out and of what ranks
            // exchange -- zeros, not what the (recycled) pinned buffer held
            const long long used = 160 + 32 * (long long)ns + 2 * (1 + (long long)ns) * n;
            if (lane < (int)(S.rec_bytes - used)) rec[used + lane] = 0;
        }
        double *sc = reinterpret_cast<double *>(rec + 160);
        unsigned long long *mk = reinterpret_cast<unsigned long long *>(rec + 160 + 24 * (size_t)ns);
        for (uint32_t r = lane; r < ns; r += 64) {
            const uint32_t x = t.rlist[first + r];
            for (int q = 0; q < 3; q++) sc[3 * (size_t)r + q] = t.scores[3 * (size_t)(first + x) + q];
            mk[r] = t.mask[first + x];
        }
    }
    if (__ballot(deep) != 0ull && lane == 0) t.deep[s] = 1;
}

// end of the tail (one block): every record's offsets for the getters, the fallback flag once more (the pack kernel may
// have raised it), then the word the host polls
extern "C" __global__ __launch_bounds__(1024) void sq_tail_done_kernel(SqTailIO t, long long *h_rec_off, long long *h_txt_off, volatile uint32_t *h_seq, uint32_t seq)
{
    for (int s = threadIdx.x; s < t.nseq; s += 1024) { h_rec_off[s] = t.seqs[s].rec_off; h_txt_off[s] = t.seqs[s].txt_off; }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        t.h_totals[2] = (long long)*t.fallback;
        t.h_totals[5] = (long long)t.fallback[1];         // some record did not fit the result buffers
        t.fallback[1] = 0;
        sq_host_write_flush(t.h_totals);
        *h_seq = seq;
    }
}

// start of a fold: the log of final structures, the per-job evaluation counts and the tail's per-job counters start empty
extern "C" __global__ __launch_bounds__(256) void sq_fold_begin_kernel(uint32_t *fin_ctr, long long *job_evals, uint32_t *job_cnt, int njobs)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < 16) fin_ctr[q] = 0;
    if (q < njobs) job_evals[q] = 0;
    if (q <= njobs) job_cnt[q] = 0;
}

// The device pools gave up (a capacity) and the host loop repeats the greedy part: the structures the aborted pools -- and the
// chains in front of them -- had logged leave the log, the stemsets of E / H / N (appended by sq_algo_finish_kernel on the side
// streams, which the caller has waited for) stay.  One block; the kept entries move to the front in order, their stems stay
// where they are.  Evaluation counts and the tail's per-job counters start over.
extern "C" __global__ __launch_bounds__(1024) void sq_fin_keep_algos_kernel(SqPoolFin *fin, uint32_t *fin_ctr, uint32_t fin_cap, long long *job_evals,
                                                                           uint32_t *job_cnt, int njobs)
{
    __shared__ uint32_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t n = min(fin_ctr[0], fin_cap);
    uint32_t base = 0;
    for (uint32_t q0 = 0; q0 < n; q0 += 1024) {
        const uint32_t q = q0 + tid;
        SqPoolFin F;
        const bool keep = q < n && (F = fin[q], F.round_kind < SQ_FIN_KIND_G0);
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_w[wv] = (uint32_t)__popcll(m);
        __syncthreads();                                     // (every entry of the tile is read before any moves)
        uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wv; w++) at += s_w[w];
        if (keep) fin[at] = F;
        for (int w = 0; w < 16; w++) base += s_w[w];
        __syncthreads();
    }
    for (int q = tid; q < njobs; q += 1024) job_evals[q] = 0;
    for (int q = tid; q <= njobs; q += 1024) job_cnt[q] = 0;
    if (tid == 0) { fin_ctr[0] = base; fin_ctr[2] = 0; }     // (fin_ctr[1]: the stems of the dropped entries stay allocated)
}

// appends host-built final structures (the E / H / N stemsets while their filters run on the host; the empty structure of
// a job whose maxstemnum is 0) to the device log: src / src_stems in pinned host memory, stem_off relative to src_stems
extern "C" __global__ __launch_bounds__(256) void sq_fin_append_kernel(const SqPoolFin *src, const SqPoolStem *src_stems, int n, SqPoolFin *fin,
                                                                      SqPoolStem *stems, uint32_t *ctr, uint32_t fin_cap, uint32_t stem_cap)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    SqPoolFin F = src[q];
    const uint32_t idx = atomicAdd(&ctr[0], 1u), so = atomicAdd(&ctr[1], (uint32_t)F.nstems);
    if (idx >= fin_cap || so + (uint32_t)F.nstems > stem_cap) {
        ctr[2] = 1;
        if (idx < fin_cap) { F.nstems = 0; F.stem_off = 0; F.pad = SQ_FIN_SRC_LOG; fin[idx] = F; }   // (no slot of the log stays unwritten)
        return;
    }
    for (int k = 0; k < F.nstems; k++) stems[so + k] = src_stems[F.stem_off + k];
    F.stem_off = so; F.pad = SQ_FIN_SRC_LOG;
    fin[idx] = F;
}

// =====================================================================================================================
// host side
// =====================================================================================================================
#include <algorithm>
#include <cstring>
#include <vector>
#include "sq_host.h"

bool sq_tail_device_wanted(const sq_batch *b, const sq_fold_opts &o)
{
    const bool off = b->sw.no_device_tail;                               // (per fold: tests compare both tails in one process)
    if (off || !b->tail.seq_job0 || b->tail.njobs <= 0) return false;
    if (o.rankbydiff) return false;                                     // :917-955 stays on the host
    if (o.conslim != 0 && o.conslim != 1) return false;                 // consensus of several structures: host
    for (int q = 0; q < 3; q++) if (o.rankby[q] < 0 || o.rankby[q] > 2) return false;
    if (o.hardrest) {                                                   // forced restraint pairs (:1226-1228): host
        for (int s = 0; s < b->nseq; s++) if (b->rbp_off[s + 1] > b->rbp_off[s]) return false;
    }
    return true;
}

static int tail_wait(sq_batch *b, SqLane &ln, uint32_t seq, const char *what)
{
    return sq_wait_word(b, ln.h_seq, seq, b->stream, what);
}

// The known structures of the batch: partner per position + number of distinct pairs per sequence (:1249-1251), uploaded on
// the batch's stream.  Nothing of it depends on the fold, so the fold calls it while the matching kernels still run (the
// host is waiting anyway) and sq_tail_device finds b->tail_refs_state set: 1 no known structure, 2 uploaded; a return value
// other than 0 leaves the state 0 and sq_tail_device repeats the call (1: the host tail's case, 2: an error).
int sq_tail_refs(sq_batch *b, const int32_t *ref_off, const int32_t *ref_pairs, const uint8_t *has_ref)
{
    hipStream_t st = b->stream;
    b->tail_refs_state = 0;
    bool any_ref = false;
    if (has_ref) for (int s = 0; s < b->nseq && !any_ref; s++) any_ref = has_ref[s] != 0;
    if (any_ref) {
        const size_t need = 2 * (size_t)b->ltot + 4 * (size_t)b->nseq + 64;
        if (b->h_ref_cap < need) {
            hipStreamSynchronize(st);
            sq_pinned_put(b->h_ref); b->h_ref = nullptr; b->h_ref_cap = 0;
            void *p = nullptr;
            if (sq_pinned_get(&p, need + need / 2)) return 2;
            b->h_ref = (char *)p; b->h_ref_cap = need + need / 2;
        }
        int16_t *refp = (int16_t *)b->h_ref;
        int32_t *refn = (int32_t *)(b->h_ref + ((2 * (size_t)b->ltot + 15) & ~(size_t)15));
        memset(refp, 0xFF, 2 * (size_t)b->ltot);
        for (int s = 0; s < b->nseq; s++) {
            refn[s] = -1;
            if (!has_ref[s]) continue;
            const int off = b->seq_off[s], n = b->seq_off[s + 1] - off;
            int cnt = 0;
            for (int k = ref_off[s]; k < ref_off[s + 1]; k++) {
                int v = ref_pairs[2 * k], w = ref_pairs[2 * k + 1];
                if (v > w) std::swap(v, w);
                if (v < 0 || w >= n || v == w) return 1;                // (not a structure of this sequence: host tail)
                if (refp[off + v] == (int16_t)w && refp[off + w] == (int16_t)v) continue;   // a repeated pair
                if (refp[off + v] != -1 || refp[off + w] != -1) return 1;   // a position in two pairs: the host tail's sets
                refp[off + v] = (int16_t)w; refp[off + w] = (int16_t)v; cnt++;
            }
            refn[s] = cnt;
        }
        if (sq_check(hipMemcpyAsync(b->d_refp, refp, 2 * (size_t)b->ltot, hipMemcpyHostToDevice, st), "upload of the known structures") ||
            sq_check(hipMemcpyAsync(b->d_refn, refn, 4 * (size_t)b->nseq, hipMemcpyHostToDevice, st), "upload of the known structures")) return 2;
    }
    b->tail_refs_state = any_ref ? 2 : 1;
    return 0;
}

int sq_tail_device(sq_batch *b, const sq_fold_opts &o, const int32_t *ref_off, const int32_t *ref_pairs, const uint8_t *has_ref)
{
    hipStream_t st = b->stream;
    SqLane &ln = b->lane_full;
    SqTailIO t = b->tail;
    if (!b->h_tail_totals) {
        void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
        if (sq_pinned_get(&p0, 64) || sq_pinned_get(&p1, 8 * ((size_t)b->nseq + 1)) || sq_pinned_get(&p2, 8 * ((size_t)b->nseq + 1)) ||
            sq_pinned_get(&p3, (size_t)b->nseq + 64)) return 2;
        b->h_tail_totals = (long long *)p0; b->h_rec_off = (long long *)p1; b->h_txt_off = (long long *)p2; b->h_deep = (uint8_t *)p3;
    }
    memset(b->h_tail_totals, 0, 64);
    t.h_totals = b->h_tail_totals;
    for (int q = 0; q < 3; q++) t.rankby[q] = o.rankby[q];
    t.toplim = o.toplim; t.result_limit = b->result_limit; t.conslim = o.conslim; t.priority_mask = o.priority_mask;
    t.tmax = std::max(b->chain_tmax, 1);
    // known structures (sq_tail_refs; normally done already, while the fold waited for the matching kernels)
    if (b->tail_refs_state == 0) { const int rr = sq_tail_refs(b, ref_off, ref_pairs, has_ref); if (rr) return rr; }
    t.refp = b->tail_refs_state == 2 ? b->d_refp : nullptr; t.ref_n = b->tail_refs_state == 2 ? b->d_refn : nullptr;
    b->tail_refs_state = 0;
    const unsigned nb = 256;
    memset(b->h_deep, 0, (size_t)b->nseq);                   // (pinned host memory; the previous fold's kernels are long done)
    hipLaunchKernelGGL(sq_tail_count_kernel, dim3(nb), dim3(256), 0, st, t);
    hipLaunchKernelGGL(sq_tail_scan_kernel, dim3(1), dim3(1024), 0, st, t);
    hipLaunchKernelGGL(sq_tail_scatter_kernel, dim3(nb), dim3(256), 0, st, t);
    {
        const int bitwords = std::min(SQ_TAIL_BITWORDS, (b->maxn + 31) / 32 + 1);
        // a batch alone whose sequences have several jobs (tens to hundreds of final structures each): sixteen waves per
        // sequence instead of four -- the per-entry passes of the slowest sequence are what the fold's tail waits for
        const bool wide = !(b->inflight > 1 || b->njobs >= 4096) && b->njobs > b->nseq;
        const int thr = wide ? SQ_TAIL_THREADS_WIDE : SQ_TAIL_THREADS;
        const int keycap = wide ? 1024 : 256;                    // rank keys staged in LDS (25 bytes each; more structures: global path)
        const int refp_lds = bitwords <= 128 ? 1 : 0;                // the known structure's partner array in LDS too (sequences up to 4,096 nt)
        const size_t lds = ((((size_t)(thr / 64) * bitwords + 1) & ~(size_t)1) * 4) + (size_t)keycap * 24 + (((size_t)keycap + 7) & ~(size_t)7) + (size_t)32 * bitwords + 16 +
                           (refp_lds ? (size_t)64 * bitwords : 0);   // + the letter codes (+ the partners)
        if (lds > 160 * 1024) return 1;                          // (the host tail takes such a batch)
        if (lds > 64 * 1024) sq_max_dynamic_lds((const void *)sq_tail_rank_kernel, 160 * 1024);
        hipLaunchKernelGGL(sq_tail_rank_kernel, dim3(b->nseq), dim3(thr), lds, st, b->ctx, t, bitwords, keycap, refp_lds);
    }
    uint32_t seq = ++*ln.round_seq;
    hipLaunchKernelGGL(sq_tail_offsets_kernel, dim3(1), dim3(1024), 0, st, t, ln.h_seq, seq);
    if (sq_check(hipGetLastError(), "device tail launch")) return 2;
    const int rowcap = (b->maxn + 8) & ~7;
    const size_t dyn = (((size_t)rowcap * 2 + 15) & ~(size_t)15) + sq_extend_lds_bytes(t.tmax);
    if (dyn > 160 * 1024) return 1;
    if (dyn > 64 * 1024) sq_max_dynamic_lds((const void *)sq_tail_pack_kernel, 160 * 1024);
    // The records go straight into pinned buffers.  With buffers of an earlier fold at hand the pack kernel is launched
    // right behind the offsets -- no host round trip in between --, with the shape of that fold; a record that does not fit is
    // reported and the launch repeated with larger buffers.  The first fold of a batch waits for the sizes.
    auto launch_pack = [&](int maxshow) -> uint32_t {
        const int Y = std::max(1, std::min({maxshow, 64, std::max(1, 8192 / std::max(b->nseq, 1))}));
        t.rec_buf = b->h_rec; t.txt_buf = b->h_txt; t.deep = b->h_deep;
        hipLaunchKernelGGL(sq_tail_pack_kernel, dim3(b->nseq, Y), dim3(64), dyn, st, b->ctx, t, rowcap, (long long)b->h_rec_cap, (long long)b->h_txt_cap);
        const uint32_t sq2 = ++*ln.round_seq;
        hipLaunchKernelGGL(sq_tail_done_kernel, dim3(1), dim3(1024), 0, st, t, b->h_rec_off, b->h_txt_off, ln.h_seq, sq2);
        return sq2;
    };
    auto grow = [&](size_t rec_bytes, size_t txt_bytes) -> int {
        if (b->h_rec_cap < rec_bytes + 64) {
            sq_pinned_put(b->h_rec); b->h_rec = nullptr; b->h_rec_cap = 0;
            void *p = nullptr;
            const size_t cap = rec_bytes + rec_bytes / 4 + 4096;
            if (sq_pinned_get(&p, cap)) return 2;
            b->h_rec = (char *)p; b->h_rec_cap = cap;
        }
        if (b->h_txt_cap < txt_bytes + 64) {
            sq_pinned_put(b->h_txt); b->h_txt = nullptr; b->h_txt_cap = 0;
            void *p = nullptr;
            const size_t cap = txt_bytes + txt_bytes / 4 + 4096;
            if (sq_pinned_get(&p, cap)) return 2;
            b->h_txt = (char *)p; b->h_txt_cap = cap;
        }
        return 0;
    };
    int r = 0;
    bool packed = false;
    if (b->h_rec_cap && b->h_txt_cap) {
        seq = launch_pack(b->tail_maxshow > 0 ? b->tail_maxshow : 8);
        if (sq_check(hipGetLastError(), "device tail launch")) return 2;
        r = tail_wait(b, ln, seq, "device tail (records)");
        if (r) return r;
        if (b->h_tail_totals[6]) { sq_set_capacity_error(SQ_CAP_STRUCTS, "the log of final structures overflowed (raise max_structs)"); return -3; }
        if (b->h_tail_totals[2]) return 1;                              // some sequence needs the host tail
        packed = !b->h_tail_totals[5];
    } else {
        r = tail_wait(b, ln, seq, "device tail (ranking)");
        if (r) return r;
        if (b->h_tail_totals[6]) { sq_set_capacity_error(SQ_CAP_STRUCTS, "the log of final structures overflowed (raise max_structs)"); return -3; }
        if (b->h_tail_totals[2]) return 1;
    }
    const size_t rec_bytes = (size_t)b->h_tail_totals[0], txt_bytes = (size_t)b->h_tail_totals[1];
    b->tail_maxshow = (int)std::max<long long>(b->h_tail_totals[4], 1);
    if (!packed) {
        r = grow(rec_bytes, txt_bytes);
        if (r) return r;
        seq = launch_pack(b->tail_maxshow);
        if (sq_check(hipGetLastError(), "device tail launch")) return 2;
        r = tail_wait(b, ln, seq, "device tail (records)");
        if (r) return r;
        if (b->h_tail_totals[2] || b->h_tail_totals[5]) return 1;
    }
    b->h_rec_off[b->nseq] = (long long)rec_bytes; b->h_txt_off[b->nseq] = (long long)txt_bytes;
    b->packed_ok = true; b->packed_limit = b->result_limit;
    return 0;
}
