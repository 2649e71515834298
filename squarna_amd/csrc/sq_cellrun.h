// sq_cellrun.h -- exact cell values and bpscores of runs from ONE small LDS table (the form sq_score_kernel introduced), for
// the kernels that keep a job's per-position data in LDS over several phases (sq_rounds.hip, sq_pool_round.hip).
//
// Every position carries a combined index ci = class x R + level (class: rank of its letter among the letters the paramset
// pairs, one extra class for all others; level: index of its reactivity among the sequence's <= 16 distinct values, R = 1
// without reactivity factors), and cell[ci_i x cstride + ci_j] = w x reactfactor -- the very expression of sq_cell_score
// (SQRNdbnseq.py:329-338).  A cell then costs two byte reads and one table read.  Arbitrary float reactivities: weights
// from the table (R = 1), the factor per cell from the inputs.
#pragma once
#include <hip/hip_runtime.h>
#include "sq_device.h"
#include "sq_cells.h"

struct SqCellTmp { uint32_t lmask; uint8_t cls[32]; double rv[16]; };     // scratch of the set-up (static LDS of the caller)

struct SqCellEnv {
    const uint8_t *ci;        // [n] combined index per position (LDS)
    const double *cell;       // the table (LDS)
    int cstride;
    bool cell_tab;            // the table holds the final cell value (default reactivities, or reactivity levels in the table)
    int KR;
    uint32_t zero4;           // four bytes of a combined index of the class of letters that pair with nothing: its cells are 0.0
};

// All threads of the block: class indices l_ci[n], letter codes l_code[n] (may be nullptr) and the table s_cell (room for the
// batch's cell_entries doubles).  Contains block barriers; the arrays are ready when it returns.
// (PRE, one wave, n <= 256: the letter mask, the wave's letter codes and -- for jobs without reactivity factors whose table has at
// most 64 entries -- its cell of the table were asked for by the caller as soon as it knew the job: sq_cell_preload)
struct SqCellPre { uint32_t lmask; uint32_t code4; double cell; bool tab; };
__device__ __forceinline__ SqCellPre sq_cell_preload(const SqDevCtx &c, const SqPsetDev *ps, int64_t pos_off, int n, bool plain, int lane)
{
    SqCellPre P;
    uint32_t c4 = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) c4 |= (uint32_t)c.codes[pos_off + min(lane + 64 * t, n - 1)] << (8 * t);   // (no branch: the four go out together)
    P.code4 = c4;
    P.lmask = sq_kload(&ps->lmask);
    const int K = __popc(P.lmask) + 1;
    P.tab = plain && K * (K | 1) <= 64;
    P.cell = ps->celltab[lane];                       // (32 x 33 entries: any lane's is there; used when P.tab)
    return P;
}

template <bool PRE = false>
__device__ __forceinline__ SqCellEnv sq_cell_setup(const SqDevCtx &c, const SqJob &jb, const SqPsetDev *ps, SqCellTmp &T, uint8_t *l_ci,
                                                   uint8_t *l_code, double *s_cell, int tid, int nthr, const SqCellPre *pre = nullptr)
{
    const int n = jb.n;
    // (the letter mask comes with the paramset -- until late round 4 every block derived it from the 1,024 bytes of the
    // pairing table first: one more trip to L2 and two barriers per structure and round of the pools' round kernel)
    const uint32_t lmask = PRE ? pre->lmask : ps->lmask;
    const int K = __popc(lmask) + 1;
    const bool react_tab = !jb.default_reacts && jb.react_levels > 0 && K * jb.react_levels <= 32;   // (the host sizes the table by the same rule)
    const int R = react_tab ? jb.react_levels : 1;
    const int KR = K * R, cstride = KR | 1;
    if (react_tab) {
        for (int p = tid; p < n; p += nthr) T.rv[c.ridx[jb.pos_off + p]] = c.reacts[jb.pos_off + p];   // (all writers of a level store the same value)
        __syncthreads();
    }
    for (int p = tid; p < n; p += nthr) {
        const uint8_t code = PRE ? (uint8_t)(pre->code4 >> (8 * (p >> 6))) : c.codes[jb.pos_off + p];
        const uint32_t cb = code & 31u;
        const int cl = (lmask >> cb) & 1u ? __popc(lmask & ((1u << cb) - 1u)) : K - 1;
        if (l_code) l_code[p] = code;
        l_ci[p] = (uint8_t)(react_tab ? cl * R + c.ridx[jb.pos_off + p] : cl);
    }
    if (!react_tab) {
        // (no reactivity factor in the table: it is the paramset's own, built by the host -- SqPsetDev::celltab, same layout)
        if (PRE && pre->tab) { if (tid < K * cstride) s_cell[tid] = pre->cell; }
        else for (int e = tid; e < K * cstride; e += nthr) s_cell[e] = ps->celltab[e];
    } else
    for (int e = tid; e < KR * KR; e += nthr) {
        const int ci = e / KR, cj = e - ci * KR;
        const int ca = ci / R, cb = cj / R;
        // letter code of a class: the ca-th set bit of lmask (class K-1: any letter without pairs, weight 0 with everything)
        int la, lb;
        {
            uint32_t m = lmask; for (int t = 0; t < ca && m; t++) m &= m - 1;
            la = ca < K - 1 ? __ffs((int)m) - 1 : -1;
            m = lmask; for (int t = 0; t < cb && m; t++) m &= m - 1;
            lb = cb < K - 1 ? __ffs((int)m) - 1 : -1;
        }
        const double w = (la >= 0 && lb >= 0) ? ps->w[la * 32 + lb] : 0.0;
        double v = w;                                                   // default reactivities: w * 1 (and 1/1)
        if (react_tab) {
            double rf = jb.rf_idx >= 0 ? c.rftab[(int64_t)jb.rf_idx * 256 + (ci - ca * R) * 16 + (cj - cb * R)]
                                       : sqrt((1.0 - (T.rv[ci - ca * R] + T.rv[cj - cb * R]) / 2.0) * 2.0);
            if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);
            v = w * rf;
        }
        s_cell[ci * cstride + cj] = v;
    }
    __syncthreads();
    SqCellEnv e;
    e.ci = l_ci; e.cell = s_cell; e.cstride = cstride; e.cell_tab = jb.default_reacts || react_tab; e.KR = KR;
    e.zero4 = (uint32_t)((K - 1) * R) * 0x01010101u;
    return e;
}

__device__ __forceinline__ double sq_cellrun_exact(const SqCellEnv &e, const SqDevCtx &c, const SqJob &jb, int i, int j)
{
    const double w = e.cell[e.ci[i] * e.cstride + e.ci[j]];
    if (e.cell_tab) return w;
    double rf;
    if (jb.rf_idx >= 0) rf = sq_reactfactor(c, jb, i, j);                  // (levels that do not fit the cell table)
    else {
        const double ri = c.reacts[jb.pos_off + i], rj = c.reacts[jb.pos_off + j];   // same expression as sq_cell_score
        rf = sqrt((1.0 - (ri + rj) / 2.0) * 2.0);
    }
    if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);
    return w * rf;
}

// Four consecutive cells (i + k, j - k), k < nv <= 4, when the table holds the final cell value (j >= 3): the combined
// indices of the four i positions are four consecutive bytes of ci and those of the j positions the four bytes ending at
// j -- two aligned 32-bit LDS reads and a byte align each instead of eight byte gathers.
__device__ __forceinline__ void sq_cellrun_cells4(const SqCellEnv &e, int i, int j, int nv, double (&v)[4])
{
    const uint32_t *ciw = reinterpret_cast<const uint32_t *>(e.ci);
    const int q = j - 3;
    const uint32_t a0 = ciw[i >> 2], a1 = ciw[(i >> 2) + 1], b0 = ciw[q >> 2], b1 = ciw[(q >> 2) + 1];
    uint32_t xi = __builtin_amdgcn_alignbyte(a1, a0, (uint32_t)(i & 3));   // bytes i .. i + 3
    uint32_t xj = __builtin_amdgcn_alignbyte(b1, b0, (uint32_t)(q & 3));   // bytes j - 3 .. j
    if (nv < 4) {                                                           // cells past the run's end: the class without pairs -- they read 0.0
        const uint32_t ki = (1u << (8 * nv)) - 1u, kj = ~((1u << (8 * (4 - nv))) - 1u);
        xi = (xi & ki) | (e.zero4 & ~ki);
        xj = (xj & kj) | (e.zero4 & ~kj);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = e.cell[((xi >> (8 * k)) & 255u) * e.cstride + ((xj >> (8 * (3 - k))) & 255u)];
}

// bpscore of the run (i0 + t, j0 - t), t < L: sum(...) left to right from int 0 (SQRNdbnseq.py:416).  pos: the same sum over
// the cells' positive parts -- no piece of the run can ever score more (fp addition is monotone)
// (POS = false: the caller has no use for the positive parts -- the pools' round kernel sums runs of the CURRENT structure, whose
// pieces it never meets again)
template <bool POS = true>
__device__ __forceinline__ double sq_cellrun_bps(const SqCellEnv &e, const SqDevCtx &c, const SqJob &jb, int i0, int j0, int L, double &pos)
{
    double acc = 0.0, accp = 0.0;
    if (jb.mat64_off >= 0) {
        // jobs with a dense fp64 matrix (caller matrices, bpp terms, the alignment's weighted rows): the cells themselves.  In
        // the diagonal-major layout (sq_cells.h) the cells of a run are consecutive doubles
        const double *m = c.mat64 + jb.mat64_off;
        const int64_t at = sq_m64_index(jb, i0, j0), step = jb.mat64_diag ? 1 : (int64_t)jb.n - 1;
        for (int t = 0; t < L; t += 4) {
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = t + k < L ? m[at + (int64_t)(t + k) * step] : 0.0;
#pragma unroll
            for (int k = 0; k < 4; k++) { acc = acc + v[k]; if (POS) accp = accp + (v[k] > 0.0 ? v[k] : 0.0); }
        }
        pos = accp;
        return acc;
    }
    if (jb.mulsh) {
        // the alignment's rows: cell = score x the shared matrix's cell of the two columns (sq_cells.h), the same product the
        // gather kernel used to store per job.  The four weights of a step are asked for together
        const int32_t *cl = c.mulcols + jb.pos_off;
        const int64_t ml = c.mulL;
        for (int t = 0; t < L; t += 4) {
            double w[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int tt = t + k < L ? t + k : L - 1; w[k] = c.mulM[sq_diag_index(ml, cl[i0 + tt], cl[j0 - tt])]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { const int tt = t + k < L ? t + k : L - 1; v[k] = sq_cellrun_exact(e, c, jb, i0 + tt, j0 - tt) * w[k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double x = t + k < L ? v[k] : 0.0;
                acc = acc + x;
                if (POS) accp = accp + (x > 0.0 ? x : 0.0);
            }
        }
        pos = accp;
        return acc;
    }
    for (int t = 0; t < L; t += 4) {
        double v[4];
        if (e.cell_tab && j0 - t >= 3) {
            // (the cells past the run's end come out of the table as +0.0 -- the class of letters without pairs --: adding them
            // leaves the sums what they are, no select per cell)
            sq_cellrun_cells4(e, i0 + t, j0 - t, min(4, L - t), v);
#pragma unroll
            for (int k = 0; k < 4; k++) { acc = acc + v[k]; if (POS) accp = accp + (v[k] > 0.0 ? v[k] : 0.0); }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tt = t + k < L ? t + k : L - 1;
            v[k] = sq_cellrun_exact(e, c, jb, i0 + tt, j0 - tt);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const double x = t + k < L ? v[k] : 0.0;
            acc = acc + x;
            if (POS) accp = accp + (x > 0.0 ? x : 0.0);
        }
    }
    pos = accp;
    return acc;
}

// An upper bound of a run's finalscore that is tighter than the paramset's ub_of x ub_lf x 1.25 (sq_internal.h): the
// reference's product bpscore x distance factor (<= 1) x order factor x loop factor x tetraloop factor (:732) with the
// order factor at its maximum, the tetraloop factor EXACT (:598-604,718) and each of the two loop bonuses (:692-715) only
// where it can apply at all -- an internal loop needs a paired position within five of either end inside the span, a loop
// outside one within five on either side.  Same multiplications, same order, rounding is monotone; a margin of 2^-30 on
// top.  U: prefix counts of unpaired positions; code: letter codes; n: sequence length.  +inf: no bound.
__device__ __forceinline__ double sq_run_upper(double bps, int i0, int j0, int L, const int16_t *U, const uint8_t *code, int n,
                                               double ub_of, double ub_lf, double lb)
{
    if (!(bps >= 0) || !(ub_lf < INFINITY)) return INFINITY;
    const int sa = i0 + L - 1, sb = j0 - L + 1, gap = sb - sa - 1;
    double lf = 1.0;
    if (lb >= 0) {
        const int g5 = gap < 5 ? gap : 5;
        const bool glp = ((int)U[min(sa + 6, sb)] - (int)U[sa + 1]) < g5 && ((int)U[sb] - (int)U[max(sb - 5, sa + 1)]) < g5;
        const bool glop = ((int)U[i0] - (int)U[max(i0 - 5, 0)]) < min(5, i0) && ((int)U[min(j0 + 6, n)] - (int)U[j0 + 1]) < min(5, n - 1 - j0);
        lf = (1.0 + (glp ? lb * 2.0 : 0.0)) + (glop ? lb * 2.0 : 0.0);
    }
    const bool gnra = gap == 4 && code[sa + 1] == 6 && (code[sa + 3] == 6 || code[sa + 3] == 0) && code[sa + 4] == 0;
    return (((bps * ub_of) * lf) * (gnra ? 1.25 : 1.0)) * (1.0 + 0x1p-30);
}
