// sq_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the folding core.
//
//   sq_fill_kernel    a-1  BPMatrix           SQRNdbnseq.py:258-338   (HBM-write bound)
//   sq_dense64_kernel a-1  exact fp64 (bool, score) for the API shim / external matrices
//   sq_import_kernel       caller matrices -> fp32 scan matrix
//   sq_state_kernel        partner / mask / prefix arrays of a partial structure (:446-451, :625-635)
//   sq_scan_kernel    a-2  AnnotateStems      SQRNdbnseq.py:427-495   (HBM-read bound, the hot kernel)
//   sq_score_kernel   a-4..a-6 ScoreStems + ChooseStems range filter  SQRNdbnseq.py:607-789
//
// Layout decisions: the scan matrix is fp32 row-major with row pitch
// ld == 1 (mod 32).  Cell (i, s-i) of anti-diagonal s then sits at float offset
// i*(ld-1) + s, so a lane that owns the four diagonals s..s+3 (s % 4 == 0) walks
// them with 16-byte aligned dwordx4 loads, a wave (256 diagonals, s0 % 256 == 0)
// reads whole 128-byte lines, and the running stem (len, sum) lives in registers:
// no LDS transposes, no cross-lane traffic, every HBM byte is read exactly once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sq_internal.h"
#include "sq_device.h"

// ------------------------------------------------------------------------------------
// cell predicates / values (fp64, same operation order as the reference)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bool sq_cell_bool(const SqDevCtx &c, const SqJob &jb, const SqPsetDev *ps, int i, int j)
{
    const uint8_t *codes = c.codes + jb.pos_off;
    const uint8_t *flags = c.flags + jb.pos_off;
    if (j < i + (int)c.inc4[jb.pos_off + i]) return false;            // :294-299 (also j <= i)
    const int a = codes[i], b = codes[j];
    if (!ps->inbps[a * 32 + b]) return false;                         // :300
    const int fi = flags[i], fj = flags[j];
    if ((fi | fj) & 1) return false;                                  // :302 rxs
    if (fj & 2) return false;                                         // :303 rlefts
    if (fi & 4) return false;                                         // :304 rrights
    if (jb.interchainonly && c.chain[jb.pos_off + i] == c.chain[jb.pos_off + j]) return false;   // :301
    return true;
}

// value of scoremat[i,j] for a cell whose bool is 1 (:329-338)
__device__ __forceinline__ double sq_cell_score(const SqDevCtx &c, const SqJob &jb, const SqPsetDev *ps, int i, int j)
{
    const uint8_t *codes = c.codes + jb.pos_off;
    const double w = ps->w[codes[i] * 32 + codes[j]];
    double rf = 1.0;
    if (!jb.default_reacts) {
        const double *r = c.reacts + jb.pos_off;
        rf = sqrt((1.0 - (r[i] + r[j]) / 2.0) * 2.0);                 // x**0.5 (see DESIGN.md on pow vs sqrt)
    }
    if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);                   // :335-336
    return w * rf;
}

// exact cell value used for every decision (fp64): dense matrix when the job has one
__device__ __forceinline__ double sq_cell_exact(const SqDevCtx &c, const SqJob &jb, const SqPsetDev *ps, int i, int j)
{
    if (jb.mat64_off >= 0) return c.mat64[jb.mat64_off + (int64_t)i * jb.n + j];
    return sq_cell_score(c, jb, ps, i, j);
}

// ------------------------------------------------------------------------------------
// a-1  fill: one thread = 4 consecutive floats of the padded N x ld matrix (16-byte stores)
// ------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void sq_fill_kernel(SqDevCtx c)
{
    const SqJob jb = c.jobs[blockIdx.y];
    if (jb.has_ext == 1) return;                       // imported from caller matrices instead
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n, ld = jb.ld;
    const int64_t total4 = ((int64_t)n * ld + 3) >> 2;
    float *mat = c.mat32 + jb.mat_off;
    double *m64 = jb.mat64_off >= 0 ? c.mat64 + jb.mat64_off : nullptr;   // has_ext == 2: holds the multiplier
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (int64_t)gridDim.x * 256) {
        const int64_t idx = q << 2;
        int i = (int)(idx / ld);
        int j = (int)(idx - (int64_t)i * ld);
        float out[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t bits = SQ_SENT_BITS;
            if (i < n && j < n && j > i && sq_cell_bool(c, jb, ps, i, j)) {
                double v = sq_cell_score(c, jb, ps, i, j);
                if (m64) {                              // :1084-1085 bpscorematrix * shortsmat
                    v = v * m64[(int64_t)i * n + j];
                    m64[(int64_t)i * n + j] = v;
                }
                bits = __float_as_uint((float)v);
                if (bits == SQ_SENT_BITS) bits = 0x7FC00001u;   // a genuine NaN value stays "present"
            } else if (m64 && i < n && j < n) {
                m64[(int64_t)i * n + j] = 0.0;
            }
            out[k] = __uint_as_float(bits);
            if (++j == ld) { j = 0; i++; }
        }
        *reinterpret_cast<float4 *>(mat + idx) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// exact dense (bool, score) of one job, fp64 N x N, for sq_bpmatrix_read
extern "C" __global__ __launch_bounds__(256) void sq_dense64_kernel(SqDevCtx c, int job, double *boolmat, double *scoremat)
{
    const SqJob jb = c.jobs[job];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n;
    const int64_t total = (int64_t)n * n;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int i = (int)(q / n), j = (int)(q - (int64_t)i * n);
        double b = 0.0, s = 0.0;
        if (j > i && sq_cell_bool(c, jb, ps, i, j)) {
            b = 1.0;
            s = sq_cell_score(c, jb, ps, i, j);
        }
        boolmat[q] = b;
        scoremat[q] = s;
    }
}

// caller-supplied (bool, score) fp64 matrices -> fp32 scan matrix.  mat64 arena of an
// ext job holds [score N*N][bool N*N].
extern "C" __global__ __launch_bounds__(256) void sq_import_kernel(SqDevCtx c)
{
    const SqJob jb = c.jobs[blockIdx.y];
    if (jb.has_ext != 1) return;
    const int n = jb.n, ld = jb.ld;
    const double *sc = c.mat64 + jb.mat64_off;
    const double *bl = sc + (int64_t)n * n;
    float *mat = c.mat32 + jb.mat_off;
    const int64_t total = (int64_t)n * ld;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int i = (int)(q / ld), j = (int)(q - (int64_t)i * ld);
        uint32_t bits = SQ_SENT_BITS;
        if (j < n && j > i && bl[(int64_t)i * n + j] != 0.0) {
            bits = __float_as_uint((float)sc[(int64_t)i * n + j]);
            if (bits == SQ_SENT_BITS) bits = 0x7FC00001u;
        }
        mat[q] = __uint_as_float(bits);
    }
}

// ------------------------------------------------------------------------------------
// per-structure state: partner P, mask code E, prefix counts U (unpaired), SU (unpaired separators)
// ------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void sq_state_kernel(SqDevCtx c, const SqStruct *structs,
                                                                  const SqStrand *strands, SqState st)
{
    const SqStruct s = structs[blockIdx.x];
    const SqJob jb = c.jobs[s.job];
    const int n = jb.n;
    int16_t *P = st.P + (int64_t)s.slot * st.stride;
    int16_t *E = st.E + (int64_t)s.slot * st.stride;
    int16_t *U = st.U + (int64_t)s.slot * st.stride;
    int16_t *SU = st.SU + (int64_t)s.slot * st.stride;
    const int16_t *e0 = c.e0 + jb.pos_off;
    const uint8_t *codes = c.codes + jb.pos_off;
    const int tid = threadIdx.x;
    for (int p = tid; p < n; p += 256) { P[p] = -1; E[p] = e0[p]; }
    __syncthreads();
    const SqStrand *sd = strands + s.strand_off;
    for (int k = tid; k < s.nstrand; k += 256) {
        const SqStrand x = sd[k];
        for (int t = 0; t < x.len; t++) {
            const int pos = x.start + t;
            P[pos] = (int16_t)(x.pstart - t);           // :634-635
            E[pos] = (int16_t)(-2 - pos);               // :446-451 row+column of a paired base are masked
        }
    }
    __syncthreads();
    // block-wide exclusive prefix sums over positions
    __shared__ int part[256], partS[256];
    const int ipt = (n + 255) >> 8;
    const int lo = tid * ipt, hi = min(n, lo + ipt);
    int cu = 0, cs = 0;
    for (int p = lo; p < hi; p++) {
        const bool un = P[p] == -1;
        cu += un;
        cs += un && (codes[p] == 26 || codes[p] == 27);
    }
    part[tid] = cu; partS[tid] = cs;
    __syncthreads();
    if (tid == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < 256; k++) { int t = part[k]; part[k] = a; a += t; t = partS[k]; partS[k] = b; b += t; }
    }
    __syncthreads();
    cu = part[tid]; cs = partS[tid];
    for (int p = lo; p < hi; p++) {
        U[p] = (int16_t)cu; SU[p] = (int16_t)cs;
        const bool un = P[p] == -1;
        cu += un;
        cs += un && (codes[p] == 26 || codes[p] == 27);
    }
    if (hi == n && lo <= n) { U[n] = (int16_t)cu; SU[n] = (int16_t)cs; }
}

// ------------------------------------------------------------------------------------
// a-2  stem scan.  block = 4 waves; wave = 256 anti-diagonals (4 per lane) x SEG rows.
// A run is owned by the wave whose row segment contains its first cell; that wave
// keeps reading past its segment until the run ends (runs are short), and skips a
// run that was already open in the row above its segment.
// ------------------------------------------------------------------------------------
#define SQ_SEG 64
#define SQ_UNR 8

#define SQ_STAGE 96   // candidates staged in LDS per wave before one aggregated global append

struct SqStage {        // per-wave LDS staging of emitted candidates (no returning global atomics in the row loop)
    uint32_t key[SQ_STAGE];
    uint32_t len[SQ_STAGE];
    float sum[SQ_STAGE];
    uint32_t count;
};

__device__ __forceinline__ void sq_emit_global(const SqScanArgs &a, const SqStruct &st, int cap, uint32_t key,
                                               uint32_t len, float sum)
{
    const uint32_t slot = atomicAdd(a.cand_cnt + st.slot, 1u);
    if (slot >= (uint32_t)cap) { a.ctr->cand_ovf = 1; return; }
    SqCand cd;
    cd.key = key; cd.len = len; cd.sum32 = sum; cd.flags = 0; cd.bps = 0; cd.fin = 0;
    a.cands[st.cand_off + slot] = cd;
}

__device__ __forceinline__ void sq_emit(SqStage *sg, const SqScanArgs &a, const SqStruct &st, int cap, int s, int rend,
                                        int len, float sum, float asum, double minlen, double minscore)
{
    if ((double)len < minlen) return;                                   // :492
    // fp32 prefilter with a rigorous rounding margin; the exact fp64 test is in sq_score_kernel
    const double ub = (double)sum + (double)asum * (double)(len + 2) * 1.1920928955078125e-07;
    if (!(ub >= minscore)) return;
    const uint32_t key = ((uint32_t)s << 16) | (uint32_t)(rend - len);
    const uint32_t slot = atomicAdd(&sg->count, 1u);                    // LDS atomic
    if (slot < SQ_STAGE) { sg->key[slot] = key; sg->len[slot] = (uint32_t)len; sg->sum[slot] = sum; }
    else sq_emit_global(a, st, cap, key, (uint32_t)len, sum);          // staging full: rare direct append
}

// whole wave: append the staged candidates with ONE global atomic
__device__ __forceinline__ void sq_flush(SqStage *sg, const SqScanArgs &a, const SqStruct &st, int cap, int lane)
{
    uint32_t n = sg->count;
    if (n > SQ_STAGE) n = SQ_STAGE;
    if (n == 0) return;
    uint32_t b0 = 0;
    if (lane == 0) b0 = atomicAdd(a.cand_cnt + st.slot, n);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
    for (uint32_t k = lane; k < n; k += 64) {
        const uint32_t slot = base + k;
        if (slot >= (uint32_t)cap) { a.ctr->cand_ovf = 1; continue; }
        SqCand cd;
        cd.key = sg->key[k]; cd.len = sg->len[k]; cd.sum32 = sg->sum[k]; cd.flags = 0; cd.bps = 0; cd.fin = 0;
        a.cands[st.cand_off + slot] = cd;
    }
}

extern "C" __global__ __launch_bounds__(256) void sq_scan_kernel(SqDevCtx c, const SqStruct *structs, SqState stt, SqScanArgs a)
{
    extern __shared__ int16_t e_lds[];
    __shared__ SqStage s_stage[4];
    const SqStruct st = structs[blockIdx.x];
    const SqJob jb = c.jobs[st.job];
    const int n = jb.n, ld = jb.ld;
    if (n < 5) return;                                                  // :456-457 no diagonals
    const int nband = (2 * n - 5 + 255) >> 8;
    const int nseg = ((n >> 1) + 130 + SQ_SEG - 1) / SQ_SEG;
    const int nsg = (nseg + 3) >> 2;
    const int tile = blockIdx.y;
    if (tile >= nband * nsg) return;
    const int band = tile / nsg, sg = tile - band * nsg;
    const int s0 = band << 8;
    const int smin = max(s0, 4), smax = min(s0 + 255, 2 * n - 6);       // :456-457 s in [4, 2N-6]
    if (smin > smax) return;
    const int rmin = max(0, smin - (n - 1)), rmax = (smax - 1) >> 1;    // :486 i <= j-1
    const int rblk = rmin + sg * 4 * SQ_SEG;
    if (rblk > rmax) return;

    const int16_t *eg = stt.E + (int64_t)st.slot * stt.stride;
    for (int p = threadIdx.x; p < n; p += 256) e_lds[p] = eg[p];
    if (threadIdx.x < 4) s_stage[threadIdx.x].count = 0;
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    SqStage *stg = &s_stage[wave];
    const int rbeg = rblk + wave * SQ_SEG;
    if (rbeg > rmax) return;
    const int rend = min(rbeg + SQ_SEG, rmax + 1);

    const SqPsetDev *ps = c.psets + jb.pset;
    const double minlen = ps->minlen, minscore = ps->minbpscore;
    const int cap = jb.cand_cap;

    int lo[4], hi[4], len[4];
    float sum[4], asum[4];
    bool skip[4];
    const int sl = s0 + 4 * lane;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = sl + k;
        const bool ok = s >= 4 && s <= 2 * n - 6;
        lo[k] = ok ? max(0, s - (n - 1)) : 0x3fffffff;
        hi[k] = ok ? (s - 1) >> 1 : -1;
        len[k] = 0; sum[k] = 0.f; asum[k] = 0.f; skip[k] = false;
    }
    // float offset of cell (r, sl - r) is r*(ld-1) + sl
    const float *base = c.mat32 + jb.mat_off + sl;
    const int64_t pitch = ld - 1;

    auto cell_active = [&](int r, int k, int er, float v) -> bool {
        const bool valid = (r >= lo[k]) & (r <= hi[k]);
        const int j = valid ? sl + k - r : 0;
        const int ej = e_lds[j];
        return valid & (ej == er) & (__float_as_uint(v) != SQ_SENT_BITS);   // :438-451 mask + bool
    };

    // row above the segment: a run that is already open there belongs to another wave
    if (rbeg > rmin) {
        const int r = rbeg - 1;
        const int er = __builtin_amdgcn_readfirstlane((int)e_lds[r]);
        if (er > -2) {
            const float4 v = *reinterpret_cast<const float4 *>(base + (int64_t)r * pitch);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) skip[k] = cell_active(r, k, er, vv[k]);
        }
    }

    auto do_row = [&](int r, int er, const float4 &v, bool tail) {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool act = (er > -2) && cell_active(r, k, er, vv[k]);
            if (act) {
                if (tail && len[k] == 0) skip[k] = true;             // starts in the next wave's segment
                if (!skip[k]) { len[k]++; sum[k] += vv[k]; asum[k] += fabsf(vv[k]); }
            } else {
                if (len[k] > 0) {
                    sq_emit(stg, a, st, cap, sl + k, r, len[k], sum[k], asum[k], minlen, minscore);
                    len[k] = 0; sum[k] = 0.f; asum[k] = 0.f;
                }
                skip[k] = false;
            }
        }
    };

    int r = rbeg;
    for (; r + SQ_UNR <= rend; r += SQ_UNR) {
        float4 v[SQ_UNR];
        int er[SQ_UNR];
#pragma unroll
        for (int u = 0; u < SQ_UNR; u++) {
            er[u] = __builtin_amdgcn_readfirstlane((int)e_lds[r + u]);
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (er[u] > -2)                                            // masked row: nothing to read
                v[u] = *reinterpret_cast<const float4 *>(base + (int64_t)(r + u) * pitch);
        }
#pragma unroll
        for (int u = 0; u < SQ_UNR; u++) do_row(r + u, er[u], v[u], false);
    }
    for (; r < rend; r++) {
        const int er = __builtin_amdgcn_readfirstlane((int)e_lds[r]);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (er > -2) v = *reinterpret_cast<const float4 *>(base + (int64_t)r * pitch);
        do_row(r, er, v, false);
    }
    // finish the runs that started in this segment (also flushes at r == rmax + 1)
    while (r <= rmax + 1 && __ballot((len[0] | len[1] | len[2] | len[3]) > 0) != 0ull) {
        int er = -2;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r <= rmax) {
            er = __builtin_amdgcn_readfirstlane((int)e_lds[r]);
            if (er > -2) v = *reinterpret_cast<const float4 *>(base + (int64_t)r * pitch);
        }
        do_row(r, er, v, true);
        r++;
    }
    __builtin_amdgcn_wave_barrier();
    sq_flush(stg, a, st, cap, lane);
}

// ------------------------------------------------------------------------------------
// a-4..a-6  exact rescoring + ScoreStems closed form + range filter (one block per structure)
// ------------------------------------------------------------------------------------
#define SQ_LDS_STRANDS 1024

__device__ __forceinline__ bool sq_goodloop(int x, int y)             // :615-622
{
    // rows x = 0..4, bit y set when (x, y) is a "good" internal loop
    const unsigned tab[5] = {0x07u /*0:{0,1,2}*/, 0x0Fu /*1:{0,1,2,3}*/, 0x1Fu /*2:{0..4}*/, 0x1Eu /*3:{1,2,3,4}*/,
                             0x1Cu /*4:{2,3,4}*/};
    if ((unsigned)x > 4u || (unsigned)y > 4u) return false;
    return (tab[x] >> y) & 1u;
}

extern "C" __global__ __launch_bounds__(256) void sq_score_kernel(SqDevCtx c, const SqStruct *structs,
                                                                  const SqStrand *strands, SqState stt, SqScanArgs a,
                                                                  SqOut *out, uint32_t out_cap, int mode)
{
    __shared__ SqStrand s_str[SQ_LDS_STRANDS];
    __shared__ double r_fin[4];
    __shared__ uint32_t r_key[4];
    __shared__ int r_any[4];
    const SqStruct st = structs[blockIdx.x];
    const SqJob jb = c.jobs[st.job];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n;
    const int tid = threadIdx.x;
    uint32_t ncand = a.cand_cnt[st.slot];
    if (ncand > (uint32_t)jb.cand_cap) ncand = jb.cand_cap;
    const SqStrand *S = strands + st.strand_off;
    if (st.nstrand <= SQ_LDS_STRANDS) {
        for (int k = tid; k < st.nstrand; k += 256) s_str[k] = S[k];
        S = s_str;
    }
    __syncthreads();
    const int16_t *P = stt.P + (int64_t)st.slot * stt.stride;
    const int16_t *U = stt.U + (int64_t)st.slot * stt.stride;
    const int16_t *SU = stt.SU + (int64_t)st.slot * stt.stride;
    const uint8_t *codes = c.codes + jb.pos_off;
    SqCand *cands = a.cands + st.cand_off;
    const double minbps = ps->minbpscore, minfin = ps->minfinscore;

    double best = 0.0; uint32_t bestkey = 0xFFFFFFFFu; int any = 0;

    for (uint32_t q = tid; q < ncand; q += 256) {
        SqCand cd = cands[q];
        const int s = (int)(cd.key >> 16), i0 = (int)(cd.key & 0xFFFFu), L = (int)cd.len, j0 = s - i0;
        // exact bpscore: sum(...) left to right starting from int 0  (:416)
        double bps = 0.0;
        for (int t = 0; t < L; t++) {
            const double v = sq_cell_exact(c, jb, ps, i0 + t, j0 - t);
            bps = bps + v;
        }
        bool ok = bps >= minbps;                                        // :492
        double fin = 0.0;
        if (ok && mode == 0) {
            const int sa = i0 + L - 1, sb = j0 - L + 1;                 // :655 innermost bp
            int inblockend = -1, nrec = 0, be0 = 0, be1 = 0, covered = 0, brackets = 0;
            uint64_t levelset = 0;
            int lo = 0, hi = st.nstrand;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (S[mid].start <= sa) lo = mid + 1; else hi = mid; }
            for (int k = lo; k < st.nstrand; k++) {                     // closed form of the walk :665-689
                const SqStrand x = S[k];
                if (x.start >= sb) break;
                const int pfirst = x.pstart, plast = x.pstart - (x.len - 1);
                bool wing;
                if (x.left) {
                    wing = pfirst > sb;
                    if (!wing && pfirst > inblockend) {                 // :687-689 sub-ECR face
                        if (nrec == 0) { be0 = x.start; be1 = pfirst; }
                        nrec++;
                        const int from = x.start > inblockend ? x.start : inblockend + 1;
                        covered += U[pfirst + 1] - U[from];
                        inblockend = pfirst;
                    }
                } else {
                    wing = plast < sa;
                }
                if (wing && x.start > inblockend) {                     // :679-684
                    brackets += x.len;
                    if (x.level > SQ_MAXLEVELS) a.ctr->level_ovf = 1;
                    else levelset |= 1ull << (x.level - 1);
                }
            }
            const int dots = (U[sb] - U[sa + 1]) - covered;             // :670-673
            const bool between = (SU[sb] - SU[sa + 1]) > 0;             // :675-676
            bool goodloop = false; int diff1 = 0;                       // :692-698
            if (nrec == 1 && sq_goodloop(be0 - sa - 1, sb - be1 - 1)) {
                goodloop = true;
                diff1 = abs((be0 - sa - 1) - (sb - be1 - 1));
            }
            bool goodloopout = false; int diff2 = 0;                    // :700-711
            {
                int vv = i0 - 1, ww = j0 + 1;
                while (vv >= 0 && i0 - vv - 1 < 5 && P[vv] == -1) vv--;
                while (ww < n && ww - j0 - 1 < 5 && P[ww] == -1) ww++;
                if (vv >= 0 && ww < n && P[vv] == ww && sq_goodloop(i0 - vv - 1, ww - j0 - 1)) {
                    goodloopout = true;
                    diff2 = abs((i0 - vv - 1) - (ww - j0 - 1));
                }
            }
            const double lb = ps->loopbonus;
            const double loopfactor = (1.0 + (lb * (goodloop ? 1.0 : 0.0)) * (2.0 - diff1 / 2.0))
                                      + (lb * (goodloopout ? 1.0 : 0.0)) * (2.0 - diff2 / 2.0);   // :715
            bool gnra = false;                                          // :598-604,718
            if (sb - sa - 1 == 4 && codes[sa + 1] == 6 && (codes[sa + 3] == 6 || codes[sa + 3] == 0) && codes[sa + 4] == 0)
                gnra = true;
            const double tetra = gnra ? 1.25 : 1.0;
            const double ideal = nrec == 0 ? 4.0 : 2.0;                 // :721
            const double stemdist = (double)dots + ps->bracketweight * (double)brackets;   // :723
            const double dd = fabs(stemdist - ideal);
            double sdf = 1.0;                                           // :726
            if (!between) {
                const int di = (int)dd;
                if (ps->bw_integral && di < ps->sdf_len) sdf = c.sdftab[ps->sdf_off + di];
                else sdf = pow(1.0 / (1.0 + dd), ps->distcoef);
            }
            const double of = ps->oftab[__popcll(levelset)];            // :728-729
            fin = bps * sdf * of * loopfactor * tetra;                  // :732 (reactfactor == 1)
            if (!goodloop && !goodloopout && L < 3) fin = -1.0;         // :744-745
            ok = fin >= minfin;                                         // :751
        }
        cd.bps = bps; cd.fin = fin; cd.flags = ok ? 1u : 0u;
        cands[q].bps = bps; cands[q].fin = fin; cands[q].flags = cd.flags;
        if (ok) {
            if (mode == 1) {
                const uint32_t o = atomicAdd(&a.ctr->nout, 1u);
                if (o < out_cap) { SqOut r = {(int32_t)blockIdx.x, cd.key, L, 0, bps, 0.0}; out[o] = r; }
                else a.ctr->out_ovf = 1;
            } else if (!any || fin > best || (fin == best && cd.key < bestkey)) {
                any = 1; best = fin; bestkey = cd.key;                   // :758 stable sort: ties keep emission order
            }
        }
    }
    if (mode == 1) return;

    // block argmax (finalscore desc, key asc)
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const uint32_t ok_ = __shfl_xor(bestkey, off);
        const int oa = __shfl_xor(any, off);
        if (oa && (!any || ob > best || (ob == best && ok_ < bestkey))) { any = 1; best = ob; bestkey = ok_; }
    }
    if ((tid & 63) == 0) { r_fin[tid >> 6] = best; r_key[tid >> 6] = bestkey; r_any[tid >> 6] = any; }
    __syncthreads();
    any = 0; best = 0.0; bestkey = 0xFFFFFFFFu;
    for (int w = 0; w < 4; w++)
        if (r_any[w] && (!any || r_fin[w] > best || (r_fin[w] == best && r_key[w] < bestkey))) {
            any = 1; best = r_fin[w]; bestkey = r_key[w];
        }
    if (!any) return;
    const double range = st.subopt * best;                              // :769
    for (uint32_t q = tid; q < ncand; q += 256) {
        const SqCand cd = cands[q];
        if (cd.flags && !(cd.fin < range)) {                            // :778
            const uint32_t o = atomicAdd(&a.ctr->nout, 1u);
            if (o < out_cap) { SqOut r = {(int32_t)blockIdx.x, cd.key, (int32_t)cd.len, 0, cd.bps, cd.fin}; out[o] = r; }
            else a.ctr->out_ovf = 1;
        }
    }
}
