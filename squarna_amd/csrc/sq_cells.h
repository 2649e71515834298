// sq_cells.h -- scoremat / boolmat cells on the device: the expressions of BPMatrix (SQRNdbnseq.py:258-367), shared by the
// kernels that take decisions on them (fill, scoring, RunAlgo's filters, the alignment's weighting slices).
#pragma once
#include <hip/hip_runtime.h>
#include "sq_device.h"

// Index of cell (i, j) of a job's dense fp64 matrix.  Row-major N x N, or -- jobs weighted by the alignment's shared stem
// matrix (mat64_diag) -- DIAGONAL-major: the cells of anti-diagonal s = i + j are contiguous, ordered by i.  The cells of a
// stem (i + k, j - k) then lie next to each other: the scoring kernel sums them with consecutive reads instead of reads
// N - 1 doubles apart (alignment step 2 on 4,500-nt sequences: every cell a cache line of its own).
__host__ __device__ __forceinline__ int64_t sq_diag_index(int64_t n, int i, int j)
{
    const int64_t s = (int64_t)i + j;
    if (s < n) return s * (s + 1) / 2 + i;                            // diagonals 0 .. s - 1 hold 1 + 2 + .. + s cells
    const int64_t m = 2 * n - 1 - s;                                  // cells on diagonal s (and on every later one: m, m - 1, .. 1)
    return n * n - m * (m + 1) / 2 + (i - (s - (n - 1)));
}
__host__ __device__ __forceinline__ int64_t sq_m64_index(const SqJob &jb, int i, int j)
{
    const int64_t n = jb.n;
    if (!jb.mat64_diag) return (int64_t)i * n + j;
    return sq_diag_index(n, i, j);
}

// The weight of cell (i, j) of a job whose rows and columns are positions of an alignment's sequence (SqJob::mulsh): the shared
// stem matrix's cell of the two columns (SQRNdbnseq.py:1031-1034: shortsmat = stemmatrix without the gap rows and columns).
// The matrix is diagonal-major over the columns: the cells of a stem -- (i + k, j - k) -- are neighbours wherever the
// sequence has no gap inside the stem, exactly as in the per-job slices the round-3 form materialised.
__device__ __forceinline__ double sq_mulsh_weight(const SqDevCtx &c, const SqJob &jb, int i, int j)
{
    const int32_t *cl = c.mulcols + jb.pos_off;
    return c.mulM[sq_diag_index(c.mulL, cl[i], cl[j])];
}

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

// reactfactor ((1 - (r_i + r_j) / 2) * 2) ** 0.5 of a cell (:333).  Sequences whose reactivities take <= 16 distinct values
// (every encoded input does) read it from the table the host built with its libm pow -- CPython's `**` -- so those
// factors are the reference's bit for bit; arbitrary float reactivities take IEEE sqrt, which differs from pow(x, 0.5)
// by one ulp for ~0.08 % of x.
__device__ __forceinline__ double sq_reactfactor(const SqDevCtx &c, const SqJob &jb, int i, int j)
{
    if (jb.rf_idx >= 0) {
        const uint8_t *lv = c.ridx + jb.pos_off;
        return c.rftab[(int64_t)jb.rf_idx * 256 + lv[i] * 16 + lv[j]];
    }
    const double *r = c.reacts + jb.pos_off;
    return sqrt((1.0 - (r[i] + r[j]) / 2.0) * 2.0);
}

// value of scoremat[i,j] for a cell whose bool is 1 (:329-338)
__device__ __forceinline__ double sq_cell_score(const SqDevCtx &c, const SqJob &jb, const SqPsetDev *ps, int i, int j)
{
    const uint8_t *codes = c.codes + jb.pos_off;
    const double w = ps->w[codes[i] * 32 + codes[j]];
    if (jb.default_reacts) return w;                                  // reactfactor 1 (and 1/1 for w <= 0): w * 1.0
    double rf = sq_reactfactor(c, jb, i, j);
    if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);                   // :335-336
    return w * rf;
}

// exact cell value used for every decision (fp64): dense matrix when the job has one
__device__ __forceinline__ double sq_cell_exact(const SqDevCtx &c, const SqJob &jb, const SqPsetDev *ps, int i, int j)
{
    if (jb.mat64_off >= 0) return c.mat64[jb.mat64_off + sq_m64_index(jb, i, j)];
    if (jb.mulsh) return sq_cell_score(c, jb, ps, i, j) * sq_mulsh_weight(c, jb, i, j);      // :1084-1085 (a cell whose bool is 1)
    return sq_cell_score(c, jb, ps, i, j);
}

