// sq_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the folding core.
//
//   sq_bits_masks_kernel / sq_bits_direct_kernel   a-1  BPMatrix as a diagonal bit matrix (1 bit per cell): all the
//                         fold path keeps per cell                                          SQRNdbnseq.py:258-304
//   sq_fill_kernel        a-1  the fp32 score matrix (API op sq_bpmatrix_fill; jobs with caller / multiplier matrices)
//   sq_dense64_kernel     a-1  exact fp64 (bool, score) for the API shim / external matrices
//   sq_import_kernel           caller matrices -> fp32 matrix + bits
//   sq_state_kernel            partner / mask / prefix arrays + free-position bit words of a partial structure (:446-451, :625-635)
//   sq_scan6_kernel       a-2  AnnotateStems: bit-diagonal scan, one lane per anti-diagonal, 32 rows per step (:427-495)
//   sq_score_kernel       a-4..a-6 exact fp64 bpscore filter + ScoreStems closed form; sq_select_kernel: ChooseStems range (:607-789)
//   sq_bps_kernel              the bpscore filter alone (AnnotateStems output, alignment survivor lists)
//   sq_scatter_* / sq_mirror_kernel / sq_colselect_kernel   alignment step 1 (SQRNdbnali.py:211-242)
//
// Layout decisions: DESIGN.md section 3.  The fp32 matrix (row pitch ld == 1 mod 32, quiet-NaN where bool == 0) only
// serves the API op and caller-supplied matrices; every decision of the fold is taken in fp64 by the scoring kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sq_internal.h"
#include "sq_device.h"

#include "sq_cells.h"
#include "sq_context.h"

// ------------------------------------------------------------------------------------
// a-1  fill: one thread = 4 consecutive floats of the padded N x ld matrix (16-byte stores)
// ------------------------------------------------------------------------------------
// Jobs without a dense fp64 term and with n <= SQ_FILL_LDS_N take the fast path: the per-position inputs (letter code +
// restraint flags in one byte, minimal span, chain, reactivity) and the pair-weight table are staged in LDS once per
// block, a thread then walks the flat padded matrix with a grid stride (row / column advanced incrementally: no 64-bit
// division per store) and a cell costs two LDS byte reads, one table read and a handful of compares before the
// 16-byte store.  The generic path (every input from global memory, per cell) serves jobs with a bpp term / multiplier
// matrix and very long sequences.
#define SQ_FILL_LDS_N 4096
extern "C" __global__ __launch_bounds__(256) void sq_fill_kernel(SqDevCtx c, int only_ext, int mul_done)
{
    const SqJob jb = c.jobs[blockIdx.y];
    if (jb.has_ext == 1) return;                       // imported from caller matrices instead
    if (only_ext && jb.has_ext == 0) return;           // the fold path of such jobs only needs the bit matrix
    if (only_ext && (jb.mat64_diag || jb.mulsh)) return;   // weighted by the alignment's shared matrix: read through the gap map (sq_cells.h), or gathered (sq_gather.hip)
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n, ld = jb.ld;
    float *mat = c.mat32 + jb.mat_off;
    extern __shared__ __attribute__((aligned(16))) char fill_dyn[];
    if (jb.mat64_off < 0 && !jb.mulsh && n <= SQ_FILL_LDS_N) {
        __shared__ double s_w[32 * 33];                // pair weights, row stride 33: the letters' rows start on different banks
        __shared__ float s_wf[32 * 33];                // (float)weight, or the sentinel where the pair is not in bps
        const int np = (n + 15) & ~15;
        uint8_t *l_attr = reinterpret_cast<uint8_t *>(fill_dyn);            // code | flags << 5
        uint8_t *l_inc4 = l_attr + np;
        int16_t *l_chain = reinterpret_cast<int16_t *>(l_inc4 + np);
        double *l_react = reinterpret_cast<double *>(l_inc4 + np + 2 * (size_t)np);
        const int tid = threadIdx.x;
        const bool ico = jb.interchainonly != 0, defr = jb.default_reacts != 0;
        for (int p = tid; p < n; p += 256) {
            l_attr[p] = (uint8_t)((c.codes[jb.pos_off + p] & 31) | ((c.flags[jb.pos_off + p] & 7) << 5));
            l_inc4[p] = c.inc4[jb.pos_off + p];
            if (ico) l_chain[p] = c.chain[jb.pos_off + p];
            if (!defr) l_react[p] = c.reacts[jb.pos_off + p];
        }
        for (int e = tid; e < 1024; e += 256) {
            const double w = ps->w[e];
            uint32_t fb = SQ_SENT_BITS;
            if (ps->inbps[e]) { fb = __float_as_uint((float)w); if (fb == SQ_SENT_BITS) fb = 0x7FC00001u; }
            s_w[(e >> 5) * 33 + (e & 31)] = w;
            s_wf[(e >> 5) * 33 + (e & 31)] = __uint_as_float(fb);
        }
        __syncthreads();
        const uint32_t total4 = (uint32_t)(((int64_t)n * ld + 3) >> 2), stride4 = gridDim.x * 256u;
        uint32_t q = blockIdx.x * 256u + (uint32_t)tid;
        if (q >= total4) return;
        int i = (int)((q << 2) / (uint32_t)ld), j = (int)((q << 2) - (uint32_t)i * (uint32_t)ld);
        const int sdi = (int)((stride4 << 2) / (uint32_t)ld), sdj = (int)((stride4 << 2) - (uint32_t)sdi * (uint32_t)ld);
        for (; q < total4; q += stride4) {
            uint32_t out[4] = {SQ_SENT_BITS, SQ_SENT_BITS, SQ_SENT_BITS, SQ_SENT_BITS};
            // a 16-byte store whose cells all lie on or below the diagonal (or in the padding rows) is pure sentinel
            const bool same_row = j + 3 < ld;
            if (same_row && i < n && j + 3 > i && defr && !ico) {
                // the four cells share the row: its attributes once, the four column attributes as one 32-bit window of
                // l_attr (two aligned reads + a byte align), a cell = table read + three compares (the kernel is bound
                // by its VALU stream, not by HBM: profiles/r02_fill_pmc.txt)
                const int ai = l_attr[i];
                const int ci = ai & 31;
                if (!((ai >> 5) & 5)) {                                  // :302-304 row side: no '_' / '/' restraint at i
                    const int thr = i + (int)l_inc4[i];                  // :294-300 smallest allowed j (> i)
                    const uint32_t *aw = reinterpret_cast<const uint32_t *>(l_attr);
                    const uint32_t x = __builtin_amdgcn_alignbyte(aw[(j >> 2) + 1], aw[j >> 2], (uint32_t)(j & 3));
                    const float *row = s_wf + ci * 33;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t aj = (x >> (8 * k)) & 255u;
                        const int jj = j + k;
                        const bool ok = jj >= thr && jj < n && !((aj >> 5) & 3u);   // column side: no '_' / '\\' restraint at j
                        out[k] = ok ? __float_as_uint(row[aj & 31u]) : SQ_SENT_BITS;
                    }
                }
            } else if (!(same_row && (j + 3 <= i || i >= n))) {
                int ii = i, jj = j;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (ii < n && jj < n && jj > ii) {
                        const int ai = l_attr[ii], aj = l_attr[jj];
                        const int ci = ai & 31, cj = aj & 31, fi = ai >> 5, fj = aj >> 5;
                        uint32_t bits = __float_as_uint(s_wf[ci * 33 + cj]);                         // :294-300 inbps
                        bool ok = jj >= ii + (int)l_inc4[ii] && !((fi | fj) & 1) && !(fj & 2) && !(fi & 4);   // :300, :302-304
                        if (ok && ico) ok = l_chain[ii] != l_chain[jj];                              // :301
                        if (ok && !defr && bits != SQ_SENT_BITS) {
                            const double w = s_w[ci * 33 + cj];                                      // same expressions as sq_cell_score
                            double rf = jb.rf_idx >= 0 ? sq_reactfactor(c, jb, ii, jj)
                                                       : sqrt((1.0 - (l_react[ii] + l_react[jj]) / 2.0) * 2.0);
                            if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);
                            bits = __float_as_uint((float)(w * rf));
                            if (bits == SQ_SENT_BITS) bits = 0x7FC00001u;   // a genuine NaN value stays "present"
                        }
                        out[k] = ok ? bits : SQ_SENT_BITS;
                    }
                    if (++jj == ld) { jj = 0; ii++; }
                }
            }
            *reinterpret_cast<uint4 *>(mat + ((size_t)q << 2)) = make_uint4(out[0], out[1], out[2], out[3]);
            i += sdi; j += sdj;
            if (j >= ld) { j -= ld; i++; }
        }
        return;
    }
    const int64_t total4 = ((int64_t)n * ld + 3) >> 2;
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
                if (jb.mulsh) v = v * sq_mulsh_weight(c, jb, i, j);       // :1084-1085 through the gap map
                if (m64) {                              // :1084-1085 bpscorematrix * shortsmat
                    if (mul_done) v = m64[sq_m64_index(jb, i, j)];          // the arena already holds the product
                    else {                                               // :352-354 bpp term, :1084-1085 stem matrix
                        const int64_t at = sq_m64_index(jb, i, j);
                        v = jb.ext_add ? v + m64[at] : v * m64[at];
                        m64[at] = v;
                    }
                }
                bits = __float_as_uint((float)v);
                if (bits == SQ_SENT_BITS) bits = 0x7FC00001u;   // a genuine NaN value stays "present"
            } else if (m64 && i < n && j < n && !jb.ext_add) {
                m64[sq_m64_index(jb, i, j)] = 0.0;          // (an ADDED term stays where bool == 0: scoremat += term covers every cell, :352)
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
            if (jb.mulsh) s = s * sq_mulsh_weight(c, jb, i, j);           // :1084-1085
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

// Diagonal bit matrix of a job: bit b of word (w, s) <-> bpboolmatrix[i, s - i] with i = 32 w + b, for the
// cells AnnotateStems visits (4 <= s <= 2N-6, i < j; :456-457, :486).  Word-row major (pitch bpitch >= 2N),
// so the 64 lanes of a scan wave -- 64 consecutive diagonals -- read 64 consecutive words.  Built once
// per job from the filled fp32 matrix (one more pass over N^2/2 cells); every later AnnotateStems
// evaluation reads 1 bit per cell instead of 4 bytes.
extern "C" __global__ __launch_bounds__(256) void sq_bits_kernel(SqDevCtx c, int only_ext)
{
    const SqJob jb = c.jobs[blockIdx.y];
    if (only_ext && jb.has_ext != 1) return;
    const int n = jb.n, ld = jb.ld, bp = jb.bpitch;
    const float *mat = c.mat32 + jb.mat_off;
    uint32_t *bits = c.bits + jb.bits_off;
    const int64_t total = (int64_t)jb.nw * bp;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int w = (int)(q / bp), s = (int)(q - (int64_t)w * bp);
        uint32_t word = 0;
        if (s >= 4 && s <= 2 * n - 6) {
#pragma unroll 8
            for (int b = 0; b < 32; b++) {
                const int i = 32 * w + b, j = s - i;
                if (j > i && j < n && __float_as_uint(mat[(int64_t)i * ld + j]) != SQ_SENT_BITS) word |= 1u << b;
            }
        }
        bits[q] = word;
    }
}

// The same bit matrix straight from the O(N) inputs (bpboolmatrix never depends on scores, :286-304):
// the fold path needs no fp32 matrix at all -- candidates are re-scored exactly from the inputs -- so
// sq_fold builds only this (N^2/16 bytes written per job instead of 4 N^2).  One block = one word-row w
// (rows 32w..32w+31) x 256 diagonals; row attributes are LDS broadcasts, column attributes consecutive bytes.
extern "C" __global__ __launch_bounds__(256) void sq_bits_direct_kernel(SqDevCtx c)
{
    // per row: the set of partner letters it may pair with (bit = letter code; empty when the row itself is
    // excluded by its restraint flags) and the smallest allowed j; per column: its letter code (31 = excluded)
    __shared__ uint32_t s_rmask[32];
    __shared__ int s_rjmin[32];
    __shared__ uint8_t s_ccode[256 + 32];
    __shared__ int16_t s_rch[32], s_cch[256 + 32];
    __shared__ uint32_t s_pm[32];                        // partner mask of every letter under this paramset
    const SqJob jb = c.jobs[blockIdx.y];
    if (jb.has_ext == 1 || !jb.bits_owner) return;      // bool comes from the caller's matrix (sq_bits_kernel) / from an earlier job of the sequence
    const int n = jb.n, bp = jb.bpitch;
    const int ntile = (bp + 255) >> 8;
    const SqPsetDev *ps = c.psets + jb.pset;
    uint32_t *bits = c.bits + jb.bits_off;
    const int tid = threadIdx.x;
    if (tid < 32) s_pm[tid] = ps->pmask[tid];           // :300 (codes 0..28; 31 never set) -- from the host: one load instead of 29 byte loads behind branches
    for (int blk = blockIdx.x; blk < jb.nw * ntile; blk += gridDim.x) {
        const int w = blk / ntile, s0 = (blk - w * ntile) << 8;
        const int s = s0 + tid;
        const int i0 = 32 * w;
        // the tile has cells only if some i in [i0, i0+31], j = s - i with i < j < n exists
        const bool live = s0 + 255 >= 2 * i0 + 1 && s0 <= i0 + 31 + n - 1 && s0 <= 2 * n - 6 && s0 + 255 >= 4;
        __syncthreads();
        if (live) {
            const int jbase = s0 - i0 - 31;              // column of (b = 31, tid = 0)
            for (int t = tid; t < 32 + 256 + 32; t += 256) {
                const bool isrow = t < 32;
                const int p = isrow ? i0 + t : jbase + (t - 32);
                const bool in = p >= 0 && p < n;
                const uint32_t code = in ? c.codes[jb.pos_off + p] : 31u, fl = in ? c.flags[jb.pos_off + p] : 1u;
                const int16_t ch = in ? c.chain[jb.pos_off + p] : (int16_t)0;
                if (isrow) {
                    s_rmask[t] = (in && !(fl & 1u) && !(fl & 4u)) ? s_pm[code & 31u] : 0u;         // :302, :304 (row side)
                    s_rjmin[t] = p + (in ? (int)c.inc4[jb.pos_off + p] : 0);                       // :294-299
                    s_rch[t] = ch;
                } else {
                    s_ccode[t - 32] = (uint8_t)((in && !(fl & 1u) && !(fl & 2u)) ? code : 31u);  // :302, :303 (column side)
                    s_cch[t - 32] = ch;
                }
            }
        }
        __syncthreads();
        if (s >= bp) continue;
        uint32_t word = 0;
        if (live && s >= 4 && s <= 2 * n - 6) {
#pragma unroll 8
            for (int b = 0; b < 32; b++) {
                const int cj = tid + 31 - b;               // column s - (i0 + b) relative to jbase
                const int j = s - i0 - b;
                bool ok = ((s_rmask[b] >> s_ccode[cj]) & 1u) && j >= s_rjmin[b];                   // :294-304
                if (jb.interchainonly) ok = ok && s_rch[b] != s_cch[cj];                             // :301
                if (ok) word |= 1u << b;
            }
        }
        bits[(int64_t)w * bp + s] = word;
    }
}

// The same bit matrix from letter masks: for the letters x present in the sequence, R_x = the rows of a word-row with
// letter x (and no row-side restraint flag), M_x = the columns whose letter may pair with x (and no column-side
// flag) as a bit array over j.  A word (w, s) is then OR_x R_x & reverse(M_x[t-31 .. t]), t = s - 32w: a few word
// operations instead of 32 cell tests; only the words next to the main diagonal check the minimal loop length bit
// by bit (:294-299).  grid = (parts, jobs): a block rebuilds the O(N) masks of its job and writes every `parts`-th
// word-row.  Not for interchainonly batches (the chain test does not factor): they use sq_bits_direct_kernel.
extern "C" __global__ __launch_bounds__(256) void sq_bits_masks_kernel(SqDevCtx c, int max_letters)
{
    extern __shared__ __attribute__((aligned(16))) char s_bm[];
    __shared__ uint32_t s_pm[32];
    __shared__ uint32_t s_present;
    __shared__ uint8_t s_letter[32];
    const SqJob jb = c.jobs[blockIdx.y];
    if (jb.has_ext == 1 || !jb.bits_owner) return;      // bool comes from the caller's matrix (sq_bits_kernel) / from an earlier job of the sequence
    const int n = jb.n, bp = jb.bpitch, nw = jb.nw, mw = nw + 3;
    const SqPsetDev *ps = c.psets + jb.pset;
    uint32_t *bits = c.bits + jb.bits_off;
    const int tid = threadIdx.x;
    const int npad = (n + 3) & ~3;
    uint8_t *s_ccode = reinterpret_cast<uint8_t *>(s_bm);               // [npad] column letter (31 = excluded)
    uint8_t *s_rcode = s_ccode + npad;                                  // [npad] row letter (31 = excluded)
    uint8_t *s_inc = s_rcode + npad;                                    // [npad] minimal j - i
    uint32_t *s_M = reinterpret_cast<uint32_t *>(s_inc + npad);         // [max_letters][mw], one zero word in front
    uint32_t *s_R = s_M + max_letters * mw;                             // [nw][max_letters]
    if (tid < 32) s_pm[tid] = ps->pmask[tid];           // :300 (codes 0..28; 31 never set) -- from the host: one load instead of 29 byte loads behind branches
    if (tid == 0) s_present = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (int p = tid; p < n; p += 256) {
        const uint32_t code = c.codes[jb.pos_off + p], fl = c.flags[jb.pos_off + p];
        s_ccode[p] = (uint8_t)((!(fl & 1u) && !(fl & 2u)) ? code : 31u);            // :302, :303 (column side)
        const bool rowok = !(fl & 1u) && !(fl & 4u);                                  // :302, :304 (row side)
        s_rcode[p] = (uint8_t)(rowok ? code : 31u);
        s_inc[p] = c.inc4[jb.pos_off + p];
        if (rowok && code < 29u && s_pm[code]) mine |= 1u << code;    // (letters that pair with nothing have no cells)
    }
    if (mine) atomicOr(&s_present, mine);
    __syncthreads();
    const uint32_t present = s_present;
    const int nlet = __popc(present);                                    // <= max_letters (host: letters of the batch)
    if (tid < 32) {
        uint32_t m = present; int k = 0;
        while (m) { const int x = __ffs((int)m) - 1; m &= m - 1; if (k == tid) s_letter[tid] = (uint8_t)x; k++; }
    }
    __syncthreads();
    for (int e = tid; e < nlet * mw; e += 256) {                         // column masks
        const int k = e / mw, q = e - k * mw;
        const uint32_t pm = s_pm[s_letter[k]];
        uint32_t word = 0;
        const int j0 = (q - 1) * 32;
        if (q >= 1 && j0 < n)
            for (int bb = 0; bb < 32 && j0 + bb < n; bb++) word |= ((pm >> s_ccode[j0 + bb]) & 1u) << bb;
        s_M[e] = word;
    }
    for (int e = tid; e < nw * nlet; e += 256) {                         // row masks
        const int w = e / nlet, k = e - w * nlet;
        const uint32_t x = s_letter[k];
        uint32_t word = 0;
        for (int bb = 0; bb < 32 && 32 * w + bb < n; bb++) word |= (uint32_t)(s_rcode[32 * w + bb] == x) << bb;
        s_R[w * max_letters + k] = word;
    }
    __syncthreads();
    for (int w = blockIdx.x; w < nw; w += gridDim.x) {
        const int i0 = 32 * w;
        const uint32_t *R = s_R + w * max_letters;
        for (int s = tid; s < bp; s += 256) {
            uint32_t word = 0;
            const int t = s - i0;                                        // column of bit 0
            if (s >= 4 && s <= 2 * n - 6 && t >= 0 && t - 31 < n) {
                const int q = t + 1;                                     // bit index of column t-31 behind the zero word
                for (int k = 0; k < nlet; k++) {
                    const uint32_t *M = s_M + k * mw + (q >> 5);
                    const uint64_t two = ((uint64_t)M[1] << 32) | M[0];
                    word |= R[k] & __brev((uint32_t)(two >> (q & 31)));
                }
                const int d = s - 2 * i0;                                // j - i of bit b is d - 2b
                if (word && d < 4 + 62) {
                    uint32_t keep = 0;
                    for (int bb = 0; bb < 32 && i0 + bb < n; bb++)
                        if (d - 2 * bb >= (int)s_inc[i0 + bb]) keep |= 1u << bb;            // :294-299
                    word &= keep;
                }
            }
            bits[(int64_t)w * bp + s] = word;
        }
    }
}

// ------------------------------------------------------------------------------------
// per-structure state: partner P, mask code E, prefix counts U (unpaired), SU (unpaired separators)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void sq_put_out(const SqRoundIO &io, SqScanArgs &a, const SqOut &r)
{
    const uint32_t o = atomicAdd(&a.ctr->nout, 1u);
    if (o < io.h_cap) io.h_out[o] = r;                   // straight into pinned host memory
    else if (o < io.out_cap) io.d_out[o] = r;
    else a.ctr->out_ovf = 1;
}

extern "C" __global__ void sq_done_kernel(SqRoundIO io, SqScanArgs a, uint32_t seq)
{
    *io.h_ctr = *a.ctr;
    sq_host_write_flush(io.h_ctr);                       // (the output records in pinned memory: earlier kernels)
    *io.h_seq = seq;
}

// The arrays are assembled in LDS (7 bytes per position) and written out once, coalesced; sequences that do not
// fit (n > lds_n) are assembled in place in global memory by the same code.
template <bool LDS>
__device__ __forceinline__ void sq_state_build(const SqDevCtx &c, const SqStruct &s, const SqJob &jb, const SqStrand *sd,
                                               const SqState &st, int16_t *P, uint8_t *E, int16_t *U, int16_t *SU)
{
    __shared__ int wave_u[4], wave_s[4];
    const int n = jb.n, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nthr = blockDim.x, nwv = nthr >> 6;                     // 256 threads, or 64 for short sequences (one wave does it)
    const uint8_t *e0 = c.e0c + jb.pos_off;
    const uint8_t *codes = c.codes + jb.pos_off;
    for (int p = tid; p < n; p += nthr) { P[p] = -1; E[p] = e0[p]; }
    __syncthreads();
    for (int k = tid; k < s.nstrand; k += nthr) {
        const SqStrand x = sd[k];
        for (int t = 0; t < x.len; t++) {
            const int pos = x.start + t;
            P[pos] = (int16_t)(x.pstart - t);           // :634-635
            E[pos] = 255;                               // :446-451 row+column of a paired base are masked
        }
    }
    __syncthreads();
    // exclusive prefix counts of unpaired positions (U) and unpaired separators (SU): blockDim positions per step,
    // ballots inside a wave, the four wave totals through LDS, a running base across steps
    int base_u = 0, base_s = 0;
    for (int p0 = 0; p0 < n; p0 += nthr) {
        const int p = p0 + tid;
        const bool un = p < n && P[p] == -1;
        const bool us = un && (codes[p] == 26 || codes[p] == 27);
        const unsigned long long mu = __ballot(un), ms = __ballot(us);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (lane == 0) { wave_u[wv] = __popcll(mu); wave_s[wv] = __popcll(ms); }
        __syncthreads();
        int pu = base_u + __popcll(mu & below), pS = base_s + __popcll(ms & below);
        for (int q = 0; q < wv; q++) { pu += wave_u[q]; pS += wave_s[q]; }
        if (p < n) { U[p] = (int16_t)pu; SU[p] = (int16_t)pS; }
        for (int q = 0; q < nwv; q++) { base_u += wave_u[q]; base_s += wave_s[q]; }
        __syncthreads();
    }
    if (tid == 0) { U[n] = (int16_t)base_u; SU[n] = (int16_t)base_s; }
    // free-position bit words for the bit-diagonal scan: F bit p = (E[p] == 0); G bit k + SQ_GPAD = F[n-1-k].
    // One ballot = two words.
    const int fbh = st.fbstride >> 1;
    uint32_t *FBs = st.FB + (int64_t)s.slot * st.fbstride;
    for (int m2 = wv; 2 * m2 < fbh; m2 += nwv) {          // m2: pair of words (2 m2, 2 m2 + 1) of either array
        const int pf = 64 * m2 + lane;                  // forward array: position
        const unsigned long long bf = __ballot(pf < n && E[pf] == 0);
        const int pr = n - 1 - (64 * m2 + lane - SQ_GPAD);   // reversed array: bit 64 m2 + lane <-> position n-1-(bit - pad)
        const unsigned long long br = __ballot(pr >= 0 && pr < n && E[pr] == 0);
        if (lane == 0) {
            if (2 * m2 < fbh) { FBs[2 * m2] = (uint32_t)bf; FBs[fbh + 2 * m2] = (uint32_t)br; }
            if (2 * m2 + 1 < fbh) { FBs[2 * m2 + 1] = (uint32_t)(bf >> 32); FBs[fbh + 2 * m2 + 1] = (uint32_t)(br >> 32); }
        }
    }
    if (LDS) {                                          // one coalesced write of everything
        __syncthreads();
        int16_t *gP = st.P + (int64_t)s.slot * st.stride, *gU = st.U + (int64_t)s.slot * st.stride;
        int16_t *gSU = st.SU + (int64_t)s.slot * st.stride;
        uint8_t *gE = st.E8 + (int64_t)s.slot * st.stride * 2;
        for (int p = tid; p <= n; p += nthr) {
            if (p < n) { gP[p] = P[p]; gE[p] = E[p]; }
            gU[p] = U[p]; gSU[p] = SU[p];
        }
    }
}

// chained: the structures and strands already live in device memory (sq_chain_kernel maintains them; io.h_* point at
// the same arrays), finished structures carry nstrand < 0 and the counters accumulate over the whole chain.
// the state of one structure (block = structure); returns false for a structure that is final (chained rounds)
__device__ __forceinline__ bool sq_state_body(const SqDevCtx &c, const SqRoundIO &io, const SqState &st, SqScanArgs &a, int lds_n,
                                              int chained, SqStruct &s)
{
    extern __shared__ __attribute__((aligned(16))) char st_dyn[];
    s = io.h_structs[blockIdx.x];                         // pinned host memory: one read per structure per round
    if (threadIdx.x == 0) {
        if (!chained) io.d_structs[blockIdx.x] = s;
        a.cand_cnt[s.slot] = 0; a.best[s.slot] = 0ull; a.ok_cnt[s.slot] = 0;
        if (blockIdx.x == 0 && !chained) { a.ctr->nout = 0; a.ctr->cand_ovf = 0; a.ctr->out_ovf = 0; a.ctr->level_ovf = 0; }
    }
    if (s.nstrand < 0) return false;                      // (chained) the structure is final
    if (!chained)
        for (int k = threadIdx.x; k < s.nstrand; k += blockDim.x) io.d_strands[s.strand_off + k] = io.h_strands[s.strand_off + k];
    __syncthreads();
    const SqJob jb = c.jobs[s.job];
    const SqStrand *sd = io.d_strands + s.strand_off;
    const int n = jb.n;
    if (n <= lds_n) {
        const int np = (lds_n + 8) & ~7;                  // arrays of n + 1 entries, 8-byte aligned sections
        int16_t *P = reinterpret_cast<int16_t *>(st_dyn), *U = P + np, *SU = U + np;
        uint8_t *E = reinterpret_cast<uint8_t *>(SU + np);
        sq_state_build<true>(c, s, jb, sd, st, P, E, U, SU);
    } else {
        sq_state_build<false>(c, s, jb, sd, st, st.P + (int64_t)s.slot * st.stride, st.E8 + (int64_t)s.slot * st.stride * 2,
                              st.U + (int64_t)s.slot * st.stride, st.SU + (int64_t)s.slot * st.stride);
    }
    return true;
}

extern "C" __global__ __launch_bounds__(256) void sq_state_kernel(SqDevCtx c, SqRoundIO io, SqState st, SqScanArgs a, int lds_n,
                                                                 int chained)
{
    SqStruct s;
    sq_state_body(c, io, st, a, lds_n, chained, s);
}

// ------------------------------------------------------------------------------------
// candidate staging of the scan: (key, len) records collected in LDS, appended to the structure's key array with
// one global atomic per flush
// ------------------------------------------------------------------------------------
#ifndef SQ5_STAGE
#define SQ5_STAGE 256
#endif
__device__ __forceinline__ void sq_emit_global(const SqScanArgs &a, const SqStruct &st, int cap, uint32_t key,
                                               uint32_t len, float sum)
{
    const uint32_t slot = atomicAdd(a.cand_cnt + st.slot, 1u);
    if (slot >= (uint32_t)cap) { a.ctr->cand_ovf = 1; return; }
    (void)sum;
    sq_keys(a, st)[slot] = SqKey{key, len};
}

// ------------------------------------------------------------------------------------
// a-2  stem scan, bit-diagonal form (sq_scan.h): one wave = 64 anti-diagonals of one structure, the runs staged in LDS
// ------------------------------------------------------------------------------------
struct SqScan6Lds {
    uint2 stage[SQ5_STAGE];
};

#include "sq_scan.h"
// the scan of one structure's diagonal groups gy0, gy0 + gystep, .. (one wave = the whole block): sq_scan.h, with the runs
// staged in LDS and appended to the structure's key array
struct SqScan6Sink {
    SqScan6Lds &L; const SqScanArgs &a; const SqStruct &st; int cap;
    uint32_t n;                               // runs staged since the last flush (a register: the wave is the whole block)
    // sole: this wave is the structure's only writer (one block walks all its diagonal groups): the places in the key array are
    // counted in a register (`base`) and the count is stored once at the end -- a returning atomic per flush was a trip to L2
    // per diagonal group of a wave that lives 15-20 us
    bool sole; uint32_t base;
    __device__ __forceinline__ uint32_t take(uint32_t cnt, int lane)
    {
        if (sole) { const uint32_t b0 = base; base += cnt; return b0; }
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(a.cand_cnt + st.slot, cnt);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
    }
    __device__ __forceinline__ void finish(int lane) { if (sole && lane == 0) a.cand_cnt[st.slot] = base; }
    // the places of a word-row's runs: in the staging buffer (flushed first when they do not fit), or -- more runs than the
    // buffer holds in one word-row -- straight in the structure's key array (index | 0x80000000)
    __device__ __forceinline__ uint32_t reserve(uint32_t total, int lane)
    {
        if (n + total > SQ5_STAGE) flush(lane);
        if (total > SQ5_STAGE) return 0x80000000u | take(total, lane);
        const uint32_t b0 = n; n += total; return b0;
    }
    __device__ __forceinline__ void put(uint32_t at, uint32_t key, uint32_t len)
    {
        if (at & 0x80000000u) {
            const uint32_t slot = at & 0x7FFFFFFFu;
            if (slot >= (uint32_t)cap) a.ctr->cand_ovf = 1; else sq_keys(a, st)[slot] = SqKey{key, len};
        } else L.stage[at] = make_uint2(key, len);
    }
    __device__ __forceinline__ void flush(int lane)
    {
        __syncthreads();
        if (n) {
            const uint32_t base0 = take(n, lane);
            for (uint32_t k = lane; k < n; k += 64) {
                const uint32_t slot = base0 + k;
                if (slot >= (uint32_t)cap) { a.ctr->cand_ovf = 1; continue; }
                sq_keys(a, st)[slot] = SqKey{L.stage[k].x, L.stage[k].y};
            }
        }
        n = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void poll(int lane) { if (n > SQ5_STAGE / 2) flush(lane); }
    __device__ __forceinline__ void drain(int lane) { flush(lane); }
};
__device__ __forceinline__ void sq_scan6_body(const SqDevCtx &c, const SqStruct &st, const SqState &stt, SqScanArgs &a, int gy0, int gystep)
{
    __shared__ __attribute__((aligned(16))) SqScan6Lds L;
    extern __shared__ uint32_t sq6_fg[];                                // F words then G words of the structure
    if (st.nstrand < 0) return;                                         // (chained rounds) the structure is final
    const SqJob jb = c.jobs[st.job];
    const int n = jb.n;
    if (n < 5) return;                                                  // :456-457 no diagonals
    const int lane0 = threadIdx.x;
    const int fbh = stt.fbstride >> 1;
    {
        const uint32_t *FBg0 = stt.FB + (int64_t)st.slot * stt.fbstride;
        for (int m = lane0; m < 2 * fbh; m += 64) sq6_fg[m] = FBg0[m];
        __syncthreads();
    }
    // one block = the diagonal groups blockIdx.y, blockIdx.y + gridDim.y, ..: a launch for short sequences gives a structure
    // ONE wave that walks all its groups (five for 100 nt) instead of one wave per group -- each of those spent most of its
    // few microseconds on the set-up above, and with batches in flight wave slots are what the chip runs out of
    SqScan6Sink sink{L, a, st, jb.cand_cap, 0u, gystep == 1, 0u};       // (cand_cnt of the slot: zeroed by the state kernel in front)
    sq_scan6_groups(c, jb, sq6_fg, sq6_fg + fbh, fbh, stt.E8 + (int64_t)st.slot * stt.stride * 2, gy0, gystep, lane0, sink, SqBitsGlobal{c.bits + jb.bits_off, jb.bpitch});
    sink.finish(lane0);
}

extern "C" __global__ __launch_bounds__(64) void sq_scan6_kernel(SqDevCtx c, const SqStruct *structs, SqState stt, SqScanArgs a)
{
    const SqStruct st = structs[blockIdx.x];
    sq_scan6_body(c, st, stt, a, (int)blockIdx.y, (int)gridDim.y);
}

// Short sequences on a crowded chip: the state and the scan of a structure by ONE wave in ONE launch (the two kernels had
// the same grid there: a wave per structure each).  The state goes to global memory as before -- the scoring kernel reads
// it --, the scan part reads back what it needs (free-position words, mask codes) behind a fence.
extern "C" __global__ __launch_bounds__(64) void sq_state_scan_kernel(SqDevCtx c, SqRoundIO io, SqState stt, SqScanArgs a, int lds_n, int chained)
{
    SqStruct s;
    if (!sq_state_body(c, io, stt, a, lds_n, chained, s)) return;
    // (the scan part reads what this block wrote: a fence at workgroup scope orders the block's own global accesses -- a
    // device-scope fence writes the L2 back and cost more than the whole kernel: 580 k -> 320 k sequences/s)
    __threadfence_block();
    __syncthreads();
    sq_scan6_body(c, s, stt, a, 0, 1);
}


// ------------------------------------------------------------------------------------
// a-4..a-6  exact rescoring + ScoreStems closed form + range filter (one block per structure)
// ------------------------------------------------------------------------------------
#define SQ_LDS_STRANDS 1024

#include "sq_score.h"

#ifndef SQ_DIAG_TOGETHER
#define SQ_DIAG_TOGETHER 2        // candidates of a chunk whose first cells are read together (alignment step 2)
#endif
#ifndef SQ_SCORE_WAVES
#define SQ_SCORE_WAVES 5                             // 96 VGPRs: five waves per SIMD instead of four at 97
#endif
template <bool FULL>
__device__ __forceinline__ void sq_score_body(const SqDevCtx &c, const SqStruct *structs, const SqStrand *strands,
                                              const SqState &stt, SqScanArgs &a, const SqRoundIO &io, int mode, int lds_n,
                                              int lds_n_reacts, int lds_n_state, int surv_off, int cell_off, int str_off = 0, int str_cap = 0,
                                              const SqCtxTab ct = SqCtxTab{}, int bound = 1)
{
    // (the structure's strands and their skip pointers live in the block's DYNAMIC LDS, sized by the host for the longest
    // strand list a structure of the launch can have: a static array for 1,024 strands cost every block 10 KB, and LDS a
    // scoring block holds is LDS its neighbours on the CU -- other scoring blocks, the blossom blocks of the side stream --
    // cannot get; 150-nt sequences need 780 bytes)
    // Cell values from ONE table: every position carries a combined index ci = class * R + level (class: rank of its
    // letter among the letters the paramset pairs, one extra class for all others; level: index of its reactivity
    // among the sequence's <= 16 distinct values, R = 1 without reactivity factors), and
    //     s_cell[ci_i * cstride + ci_j] = w * reactfactor   (the very expression of sq_cell_score, built once per block).
    // A cell then costs two byte reads and one table read.  The 32 x 32 weight table this replaces had a row stride
    // of 256 bytes = the whole bank span, so the pairs of the four letters sat on four bank pairs (4-way conflicts,
    // 72 % of the LDS-busy cycles of the kernel, profiles/r01k_score_pmc_*); the compact table's odd stride spreads
    // the <= (K R)^2 live entries over all banks.  Arbitrary float reactivities: weights from the table (R = 1), the
    // factor per cell as before.  The table sits in the block's dynamic LDS at cell_off, sized by the host for the
    // largest K R of the batch (25 entries for ACGU without reactivity levels; a static 32 x 33 array cost 8 KB per block).
    __shared__ uint8_t s_cls[32];
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];   // letter codes [n] (+ reactivities [n] when they fit)
    double *const s_cell = reinterpret_cast<double *>(s_dyn + cell_off);
    SqStrand *const s_str = reinterpret_cast<SqStrand *>(s_dyn + str_off);
    uint16_t *const s_skip = reinterpret_cast<uint16_t *>(s_str + str_cap);   // 5' strand k closes a block: next strand that can matter after it
    const SqStruct st = structs[blockIdx.x];
#ifdef SQ_SCORE_PROF
    const long long _p0 = wall_clock64(); long long _pa = 0, _pb = 0, _ps = 0;
    int _nsc = 0, _ngr = 0, _n492 = 0;       // ScoreStems calls of the first wave, its groups, candidates of the first wave that passed :492
#endif
    const SqJob jb = c.jobs[st.job];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    uint32_t ncand = a.cand_cnt[st.slot];
    if (ncand > (uint32_t)jb.cand_cap) ncand = jb.cand_cap;
    if ((uint32_t)blockIdx.y * (uint32_t)nthr >= ncand) return;         // this part has no candidates
    const SqStrand *S = strands + st.strand_off;
    const bool lds_strands = FULL && st.nstrand <= str_cap;
    if (lds_strands) {
        for (int k = tid; k < st.nstrand; k += nthr) s_str[k] = S[k];
        S = s_str;
    }
    // structures without crossing stems: the strand sweep in closed form from the round's context tables (sq_context.h)
    const bool use_ctx = FULL && ct.rec != nullptr && st.nstrand > 0 && ct.ok[st.slot] != 0;
    const SqCtxRec *const ctx_rec = ct.rec + (size_t)st.slot * ct.cap;
    const int16_t *const ctx_depth = ct.depth + (size_t)st.slot * ct.cap;
    const uint16_t *const ctx_rmq = ct.rmq + (size_t)st.slot * ct.cap * ct.levels;
    __syncthreads();
#ifdef SQ_CTX_CHECK
    if (lds_strands) {
#else
    if (lds_strands && !use_ctx) {
#endif
        // Once the sweep of ScoreStems has registered the block [start, partner] of a 5' strand, the strands that
        // start inside it are inert unless they are 5' strands whose partner lies beyond the block's end (they extend
        // it or are wings); skip[k] = the first strand behind k that starts outside the block or is such a strand.
        for (int k = tid; k < st.nstrand; k += nthr) {
            const SqStrand x = s_str[k];
            int q = k + 1;
            if (x.left) {
                const int pf = x.pstart;
                while (q < st.nstrand) {
                    const SqStrand y = s_str[q];
                    if (y.start > pf || (y.left && y.pstart > pf)) break;
                    q++;
                }
            }
            s_skip[k] = (uint16_t)q;
        }
    }
    const int16_t *P = stt.P + (int64_t)st.slot * stt.stride;
    const int16_t *U = stt.U + (int64_t)st.slot * stt.stride;
    const int16_t *SU = stt.SU + (int64_t)st.slot * stt.stride;
    // the exact re-scoring touches codes / weights / reactivities once per cell: keep them in LDS
    double *l_reacts = reinterpret_cast<double *>(s_dyn + ((n + 15) & ~15));
    // ScoreStems walks the partner array and reads the prefix counts with dependent loads: LDS copies when they
    // fit (3 x int16 per position, after the codes and reactivities of the launch's longest sequence)
    if (mode == 0 && lds_n_state >= n) {
        int16_t *lP = reinterpret_cast<int16_t *>(s_dyn + ((lds_n + 15) & ~15) + (size_t)8 * lds_n_reacts);
        int16_t *lU = lP + ((lds_n_state + 8) & ~7), *lSU = lU + ((lds_n_state + 8) & ~7);
        // 32-bit copies (the arrays start 64-byte aligned: stride is a multiple of 32 positions)
        const uint32_t *gP = reinterpret_cast<const uint32_t *>(P), *gU = reinterpret_cast<const uint32_t *>(U);
        const uint32_t *gS = reinterpret_cast<const uint32_t *>(SU);
        uint32_t *wP = reinterpret_cast<uint32_t *>(lP), *wU = reinterpret_cast<uint32_t *>(lU), *wS = reinterpret_cast<uint32_t *>(lSU);
        const int nw2 = (n + 2) >> 1;                       // covers entries 0..n
        for (int q = tid; q < nw2; q += nthr) { wP[q] = gP[q]; wU[q] = gU[q]; wS[q] = gS[q]; }
        P = lP; U = lU; SU = lSU;
    }
    const bool lds_cells = jb.mat64_off < 0 && !jb.mulsh && lds_n >= n;
    const bool any_reacts = lds_cells && !jb.default_reacts;
    // classes of the letters: K pairing letters + one class for everything else.  lmask: bit a set iff letter a has a
    // pair in the paramset (row a of inbps is not all zero) -- the first 32 threads test a row each, one ballot
    __shared__ uint32_t s_lmask;
    if (lds_cells && tid < 64) {
        uint32_t any8 = 0;
        if (tid < 32) {
            const uint32_t *ib = reinterpret_cast<const uint32_t *>(ps->inbps) + tid * 8;
#pragma unroll
            for (int q = 0; q < 8; q++) any8 |= ib[q];
        }
        const unsigned long long bal = __ballot(any8 != 0);
        if (tid == 0) s_lmask = (uint32_t)bal;
    }
    __syncthreads();
    const uint32_t lmask = lds_cells ? s_lmask : 0u;
    const int K = __popc(lmask) + 1;
    // few distinct reactivity values (encoded input): level index per position, reactfactors folded into the table
    const bool react_tab = any_reacts && jb.react_levels > 0 && K * jb.react_levels <= 32;   // (the host sizes LDS by the same rule)
    const bool lds_reacts = any_reacts && !react_tab && lds_n_reacts >= n;                    // per-cell factors from LDS copies
    const int R = react_tab ? jb.react_levels : 1;
    const int KR = K * R, cstride = KR | 1;
    const bool cell_tab = lds_cells && (jb.default_reacts || react_tab);     // the table holds the final cell value
    __shared__ double s_rv[16];
    uint8_t *l_ci = reinterpret_cast<uint8_t *>(s_dyn);                       // combined index per position
    if (lds_cells) {
        if (tid < 32) s_cls[tid] = (lmask >> tid) & 1u ? (uint8_t)__popc(lmask & ((1u << tid) - 1u)) : (uint8_t)(K - 1);
        if (react_tab)
            for (int p = tid; p < n; p += nthr) s_rv[c.ridx[jb.pos_off + p]] = c.reacts[jb.pos_off + p];   // (all writers of a level store the same value)
        else if (lds_reacts) for (int p = tid; p < n; p += nthr) l_reacts[p] = c.reacts[jb.pos_off + p];
    }
    __syncthreads();
    if (lds_cells) {
        for (int p = tid; p < n; p += nthr) {
            const int cl = s_cls[c.codes[jb.pos_off + p] & 31];
            l_ci[p] = (uint8_t)(react_tab ? cl * R + c.ridx[jb.pos_off + p] : cl);
        }
        if (!react_tab) {
            // (no reactivity factor in the table: it is the paramset's own, built by the host -- SqPsetDev::celltab, same layout)
            for (int e = tid; e < K * cstride; e += nthr) s_cell[e] = ps->celltab[e];
        } else
        for (int e = tid; e < KR * KR; e += nthr) {
            const int ci = e / KR, cj = e - ci * KR;
            const int ca = ci / R, cb = cj / R;
            // letter code of a class: the ca-th set bit of lmask (class K-1: any letter without pairs, weight 0 with everything)
            int la = 31, lb = 31;
            {
                uint32_t m = lmask; for (int t = 0; t < ca && m; t++) m &= m - 1;
                la = ca < K - 1 ? __ffs((int)m) - 1 : -1;
                m = lmask; for (int t = 0; t < cb && m; t++) m &= m - 1;
                lb = cb < K - 1 ? __ffs((int)m) - 1 : -1;
            }
            const double w = (la >= 0 && lb >= 0) ? ps->w[la * 32 + lb] : 0.0;
            double v = w;                                                   // default reactivities: w * 1 (and 1/1)
            if (react_tab) {
                // (react_tab implies react_levels > 0, i.e. the host built the sequence's pow table)
                double rf = jb.rf_idx >= 0 ? c.rftab[(int64_t)jb.rf_idx * 256 + (ci - ca * R) * 16 + (cj - cb * R)]
                                           : sqrt((1.0 - (s_rv[ci - ca * R] + s_rv[cj - cb * R]) / 2.0) * 2.0);
                if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);
                v = w * rf;
            }
            s_cell[ci * cstride + cj] = v;
        }
    }
    __syncthreads();
    const uint8_t *codes = c.codes + jb.pos_off;
    const SqKey *keys = sq_keys(a, st);
    SqOk *oks = sq_oks(a, st, jb.cand_cap);
    // the survivor list can take as many records as remain in the slice after the key array
    const uint32_t ok_cap = (uint32_t)(((size_t)jb.cand_cap * (sizeof(SqCand) - sizeof(SqKey))) / sizeof(SqOk));
    const double minbps = ps->minbpscore, minfin = ps->minfinscore;
    // the paramset's scalars once, ahead of the candidate loops: the compiler cannot hoist these loads itself (the loops
    // store survivors to global memory), and inside ScoreStems each of them is another dependent memory access on the chain
    const double ps_lb = ps->loopbonus, ps_bw = ps->bracketweight, ps_dc = ps->distcoef;
    const int ps_bwint = ps->bw_integral, ps_sdflen = ps->sdf_len;
    const double *const ps_sdf = c.sdftab + ps->sdf_off;
    const double *const ps_of = ps->oftab;
    auto cell_exact = [&](int i, int j) -> double {
        if (!lds_cells) return sq_cell_exact(c, jb, ps, i, j);
        const double w = s_cell[l_ci[i] * cstride + l_ci[j]];             // cell_tab: the cell itself
        if (cell_tab) return w;
        double rf;
        if (jb.rf_idx >= 0) rf = sq_reactfactor(c, jb, i, j);                  // (levels that do not fit the cell table)
        else {
            const double ri = lds_reacts ? l_reacts[i] : c.reacts[jb.pos_off + i];   // same expression as sq_cell_score
            const double rj = lds_reacts ? l_reacts[j] : c.reacts[jb.pos_off + j];
            rf = sqrt((1.0 - (ri + rj) / 2.0) * 2.0);
        }
        if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);
        return w * rf;
    };

    // Four consecutive cells (i + k, j - k), k < nv <= 4, of a stem when the table holds the final cell value: the combined
    // indices of the four i positions are four consecutive bytes of l_ci and those of the j positions the four bytes
    // ending at j -- two aligned 32-bit LDS reads and a byte align each instead of eight byte gathers (the gathers at
    // data-dependent addresses are what the kernel's LDS bank conflicts come from, and its first phase is bound by them).
    const uint32_t *l_ciw = reinterpret_cast<const uint32_t *>(l_ci);
    const uint32_t zero4 = (uint32_t)((K - 1) * R) * 0x01010101u;      // (its cells are 0.0: adding them leaves a sum what it is)
    auto cells4 = [&](int i, int j, int nv, double (&v)[4]) {
        const int q = j - 3;                                        // lowest j position of the window (>= 0 here)
        const uint32_t a0 = l_ciw[i >> 2], a1 = l_ciw[(i >> 2) + 1], b0 = l_ciw[q >> 2], b1 = l_ciw[(q >> 2) + 1];
        uint32_t xi = __builtin_amdgcn_alignbyte(a1, a0, (uint32_t)(i & 3));   // bytes i .. i + 3
        uint32_t xj = __builtin_amdgcn_alignbyte(b1, b0, (uint32_t)(q & 3));   // bytes j - 3 .. j
        if (nv < 4) {                                               // cells past the stem's end: the class without pairs -- they read +0.0
            const uint32_t ki = (1u << (8 * nv)) - 1u, kj = ~((1u << (8 * (4 - nv))) - 1u);
            xi = (xi & ki) | (zero4 & ~ki);
            xj = (xj & kj) | (zero4 & ~kj);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = s_cell[((xi >> (8 * k)) & 255u) * cstride + ((xj >> (8 * (3 - k))) & 255u)];
    };

    double best = 0.0; int any = 0;

    const uint32_t qstep = gridDim.y * nthr;
    // exact bpscore of the SQ_SCORE_CHUNK candidates of a thread: sum(...) left to right starting from int 0 (:416),
    // four cells per step: the lookups of one step (letter codes, then the weight) are independent and overlap, the
    // additions of each candidate keep the reference's order; padding cells (past a candidate's end) add +0.0, which
    // leaves its sum unchanged.  The filter-only kernel advances its candidates TOGETHER (16 lookups in flight; 3.1 ->
    // 1.9 ms per alignment chunk); under ScoreStems' register budget that spills, so there they go one by one.
    auto chunk_bps = [&](const SqKey (&cd)[SQ_SCORE_CHUNK], double (&bps)[SQ_SCORE_CHUNK]) {
        if (FULL && jb.mat64_diag) {
            // alignment step 2: the cells of a stem are consecutive doubles of the job's diagonal-major product matrix
            // (sq_cells.h) -- one index per candidate, and the first four cells of TWO candidates in flight together
            // (the reads come from HBM / L2: their latency, not their number, is what this phase waits for)
            const double *const m64 = c.mat64 + jb.mat64_off;
#pragma unroll
            for (int u = 0; u < SQ_SCORE_CHUNK; u += SQ_DIAG_TOGETHER) {
                const double *mp[SQ_DIAG_TOGETHER]; int Ls[SQ_DIAG_TOGETHER]; double v[SQ_DIAG_TOGETHER][4];
#pragma unroll
                for (int h = 0; h < SQ_DIAG_TOGETHER; h++) {
                    const int s = (int)(cd[u + h].key >> 16), i0 = (int)(cd[u + h].key & 0xFFFFu);
                    Ls[h] = (int)cd[u + h].len;
                    mp[h] = m64 + (Ls[h] > 0 ? sq_m64_index(jb, i0, s - i0) : 0);
#pragma unroll
                    for (int k = 0; k < 4; k++) v[h][k] = mp[h][Ls[h] > 0 ? min(k, Ls[h] - 1) : 0];
                }
#pragma unroll
                for (int h = 0; h < SQ_DIAG_TOGETHER; h++) {
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < 4; k++) acc = acc + (k < Ls[h] ? v[h][k] : 0.0);
                    for (int t = 4; t < Ls[h]; t++) acc = acc + mp[h][t];
                    bps[u + h] = acc;
                }
            }
            return;
        }
        if (FULL) {
#pragma unroll
            for (int u = 0; u < SQ_SCORE_CHUNK; u++) {
                const int s = (int)(cd[u].key >> 16), i0 = (int)(cd[u].key & 0xFFFFu), j0 = s - i0, L = (int)cd[u].len;
                double acc = 0.0;
                for (int t = 0; t < L; t += 4) {
                    double v[4];
                    if (cell_tab && j0 - t >= 3) {
                        cells4(i0 + t, j0 - t, min(4, L - t), v);
#pragma unroll
                        for (int k = 0; k < 4; k++) acc = acc + v[k];           // (cells past the end are +0.0)
                        continue;
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int tt = t + k < L ? t + k : L - 1;
                        v[k] = cell_exact(i0 + tt, j0 - tt);
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) acc = acc + (t + k < L ? v[k] : 0.0);
                }
                bps[u] = acc;
            }
            return;
        }
        int lmax = 0;
#pragma unroll
        for (int u = 0; u < SQ_SCORE_CHUNK; u++) { bps[u] = 0.0; lmax = max(lmax, (int)cd[u].len); }
        for (int t = 0; t < lmax; t += 4) {
            double v[SQ_SCORE_CHUNK][4];
#pragma unroll
            for (int u = 0; u < SQ_SCORE_CHUNK; u++) {
                const int s = (int)(cd[u].key >> 16), i0 = (int)(cd[u].key & 0xFFFFu), j0 = s - i0, L = (int)cd[u].len;
                if (cell_tab && t < L && j0 - t >= 3) cells4(i0 + t, j0 - t, min(4, L - t), v[u]);
                else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int tt = max(min(t + k, L - 1), 0);
                        v[u][k] = cell_exact(i0 + tt, j0 - tt);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < SQ_SCORE_CHUNK; u++) {
                const int L = (int)cd[u].len;
#pragma unroll
                for (int k = 0; k < 4; k++) bps[u] = bps[u] + (t + k < L ? v[u][k] : 0.0);
            }
        }
    };
    if (!FULL) {
        // bpscore filter only (:492): OptimalStems output (mode 1) or the alignment's survivor list (mode 2).  The
        // survivors of a chunk of SQ_SCORE_CHUNK x blockDim candidates are gathered in LDS and written out with ONE
        // global atomic per block and chunk: all blocks of a structure append to the same counter, and one atomic
        // per wave made that counter the bottleneck of long sequences (A5000: 3.2 -> 0.6 ms per 47 sequences).
        double *s_bps = reinterpret_cast<double *>(s_dyn + surv_off);
        uint32_t *s_key = reinterpret_cast<uint32_t *>(s_bps + SQ_SCORE_CHUNK * nthr);
        uint16_t *s_len = reinterpret_cast<uint16_t *>(s_key + SQ_SCORE_CHUNK * nthr);
        __shared__ uint32_t s_n, s_base;
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (uint32_t q0 = blockIdx.y * nthr; q0 < ncand; q0 += SQ_SCORE_CHUNK * qstep) {
            SqKey cd[SQ_SCORE_CHUNK];
#pragma unroll
            for (int u = 0; u < SQ_SCORE_CHUNK; u++) {
                const uint32_t q = q0 + (uint32_t)u * qstep + tid;
                cd[u] = q < ncand ? keys[q] : SqKey{0u, 0u};            // (len 0: never appended)
            }
            double bpsv[SQ_SCORE_CHUNK];
            chunk_bps(cd, bpsv);
#pragma unroll
            for (int u = 0; u < SQ_SCORE_CHUNK; u++) {
                const int L = (int)cd[u].len;
                const double bps = bpsv[u];
                const bool ok = L > 0 && bps >= minbps;
                const unsigned long long okm = __ballot(ok);
                if (okm) {
                    uint32_t base = 0;
                    const int leader = __ffsll((long long)okm) - 1;
                    if ((tid & 63) == leader) base = atomicAdd(&s_n, (uint32_t)__popcll(okm));
                    base = (uint32_t)__shfl((int)base, leader);
                    if (ok) {
                        const uint32_t pos = base + (uint32_t)__popcll(okm & ((1ull << (tid & 63)) - 1ull));
                        s_key[pos] = cd[u].key; s_len[pos] = (uint16_t)L; s_bps[pos] = bps;
                    }
                }
            }
            __syncthreads();
            const uint32_t ns = s_n;
            if (tid == 0 && ns) s_base = mode == 1 ? atomicAdd(&a.ctr->nout, ns) : atomicAdd(a.ok_cnt + st.slot, ns);
            __syncthreads();
            const uint32_t base = s_base;
            for (uint32_t k = tid; k < ns; k += nthr) {
                const uint32_t pos = base + k;
                if (mode == 1) {
                    const SqOut r = {(int32_t)blockIdx.x, s_key[k], (int32_t)s_len[k], 0, s_bps[k], 0.0};
                    if (pos < io.h_cap) io.h_out[pos] = r;              // straight into pinned host memory
                    else if (pos < io.out_cap) io.d_out[pos] = r;
                    else a.ctr->out_ovf = 1;
                } else if (pos < ok_cap) oks[pos] = SqOk{s_key[k], (uint32_t)s_len[k], s_bps[k], 0.0};
                else a.ctr->cand_ovf = 1;
            }
            __syncthreads();
            if (tid == 0) s_n = 0;
            __syncthreads();
        }
        return;
    }

    // mode 0.  Two phases per chunk of SQ_SCORE_CHUNK x blockDim candidates: (A) every thread computes the bpscore of its
    // candidates (cheap: LDS only) and the ones that pass :492 are appended to a list in LDS; (B) ScoreStems runs on
    // FULL groups of blockDim survivors (its chain of dependent loads is what the kernel waits for, so lanes idling
    // on rejected candidates were the cost); the remainder (< blockDim) is carried into the next chunk.
    double *s_bps = reinterpret_cast<double *>(s_dyn + surv_off);
    uint32_t *s_key = reinterpret_cast<uint32_t *>(s_bps + (SQ_SCORE_CHUNK + 1) * nthr);
    uint16_t *s_len = reinterpret_cast<uint16_t *>(s_key + (SQ_SCORE_CHUNK + 1) * nthr);
    __shared__ uint32_t s_nsurv, s_okn;
    // Branch and bound.  Every reader of the survivor list keeps only finalscores >= subopt x best (:769-778; width-1
    // chains and the stopper :1147 only the best itself), and a candidate's finalscore cannot exceed
    //     ub(bpscore) = ((bpscore * max orderfactor) * max loopfactor) * 1.25        (sq_internal.h; bpscore >= 0)
    // -- the same multiplications in the same order with every factor at its maximum, and rounding is monotone.  So a
    // candidate whose bound lies below subopt x (the best finalscore seen so far) -- or below minfinscore -- cannot be
    // among them whatever ScoreStems returns, and its strand sweep, the expensive part of this kernel on long
    // sequences, is skipped.  s_best: the block's running best, exchanged with the structure's other blocks through
    // a.best once per chunk.  The results are those of the exhaustive evaluation bit for bit.
    __shared__ unsigned long long s_best;
    const double ub_of = ps->ub_of, ub_lf = bound ? ps->ub_lf : INFINITY;
    auto upper = [&](double bps) -> double { return bps >= 0 ? (((bps * ub_of) * ub_lf) * 1.25) * (1.0 + 0x1p-30) : INFINITY; };
    const double st_subopt = st.subopt;
    const bool solo = gridDim.y == 1;                  // this block scores every candidate of its structure
    const SqStemsEnv env = {S, s_skip, st.nstrand, lds_strands, P, U, SU, codes, n, use_ctx, ctx_rec, ctx_depth, ctx_rmq, ct.cap,
                            ps_lb, ps_bw, ps_dc, ps_bwint, ps_sdflen, ps_sdf, ps_of, a.ctr};
    if (tid == 0) { s_nsurv = 0; s_okn = 0; s_best = 0ull; }
    __syncthreads();
#ifdef SQ_SCORE_PROF
    _ps = wall_clock64() - _p0;
#endif
    for (uint32_t q0 = blockIdx.y * nthr; q0 < ncand; q0 += SQ_SCORE_CHUNK * qstep) {
#ifdef SQ_SCORE_PROF
        const long long _t0 = wall_clock64();
#endif
        SqKey cd[SQ_SCORE_CHUNK];
#pragma unroll
        for (int u = 0; u < SQ_SCORE_CHUNK; u++) {
            const uint32_t q = q0 + (uint32_t)u * qstep + tid;
            cd[u] = q < ncand ? keys[q] : SqKey{0u, 0u};                // (len 0: bps 0, never appended)
        }
        double bpsv[SQ_SCORE_CHUNK];
        chunk_bps(cd, bpsv);
        // what a finalscore must reach to matter (read once per chunk; s_best only grows)
        double need = minfin;
        {
            const unsigned long long sb = s_best;
            if (sb) { const double r = st_subopt * sq_unord(sb); need = r > need ? r : need; }
        }
#pragma unroll
        for (int u = 0; u < SQ_SCORE_CHUNK; u++) {
            const int L = (int)cd[u].len;
            const double bps = bpsv[u];
#ifdef SQ_SCORE_PROF
            _n492 += __popcll(__ballot(L > 0 && bps >= minbps));
#endif
            bool ok = L > 0 && bps >= minbps && !(upper(bps) < need);         // :492, and the bound
            const unsigned long long okm = __ballot(ok);
            if (okm) {
                uint32_t base = 0;
                const int leader = __ffsll((long long)okm) - 1;
                if ((tid & 63) == leader) base = atomicAdd(&s_nsurv, (uint32_t)__popcll(okm));
                base = (uint32_t)__shfl((int)base, leader);
                if (ok) {
                    const uint32_t pos = base + (uint32_t)__popcll(okm & ((1ull << (tid & 63)) - 1ull));
                    s_key[pos] = cd[u].key; s_len[pos] = (uint16_t)L; s_bps[pos] = bps;
                }
            }
        }
        __syncthreads();
#ifdef SQ_SCORE_PROF
        const long long _t1 = wall_clock64(); _pa += _t1 - _t0;
#endif
        const uint32_t ns = s_nsurv;
        const bool last = q0 + SQ_SCORE_CHUNK * qstep >= ncand;
        uint32_t done = 0;
        while (done + (uint32_t)nthr <= ns || (last && done < ns)) {
            const uint32_t idx = done + tid;
            done += nthr;
            const bool have = idx < ns;
            const uint32_t key = have ? s_key[idx] : 0u;
            const int L = have ? (int)s_len[idx] : 0;
            const double bps = have ? s_bps[idx] : 0.0;
            const int s = (int)(key >> 16), i0 = (int)(key & 0xFFFFu), j0 = s - i0;
            bool ok = have;
            double fin = 0.0;
            {
                const unsigned long long sbst = s_best;                     // (the groups before this one may have raised it)
                if (sbst && upper(bps) < st_subopt * sq_unord(sbst)) ok = false;
            }
#ifdef SQ_SCORE_PROF
            _nsc += __popcll(__ballot(ok)); _ngr++;
#endif
            if (ok) {
                fin = sq_stem_finalscore(env, i0, j0, L, bps);
                ok = fin >= minfin;                                         // :751
            }
            // survivors are appended to the structure's SqOk list: one atomic per wave, lanes ranked by ballot
            const unsigned long long okm = __ballot(ok);
            if (okm) {
                uint32_t base = 0;
                const int leader = __ffsll((long long)okm) - 1;
                // (one block per structure: the block owns the list, its place counter lives in LDS -- a returning global
                // atomic per wave and pass of ScoreStems was a memory round trip on the critical path of every block)
                if ((tid & 63) == leader) base = solo ? atomicAdd(&s_okn, (uint32_t)__popcll(okm)) : atomicAdd(a.ok_cnt + st.slot, (uint32_t)__popcll(okm));
                base = (uint32_t)__shfl((int)base, leader);
                if (ok) {
                    const uint32_t pos = base + (uint32_t)__popcll(okm & ((1ull << (tid & 63)) - 1ull));
                    if (pos < ok_cap) oks[pos] = SqOk{key, (uint32_t)L, bps, fin};
                    else a.ctr->cand_ovf = 1;
                    if (!any || fin > best) { any = 1; best = fin; }        // :769 only the best VALUE matters for the range
                }
                // the wave's best finalscore of this group into the block's running best
                double wb = ok ? fin : -INFINITY;
                for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(wb, off); wb = o > wb ? o : wb; }
                if ((tid & 63) == leader) atomicMax(&s_best, sq_ord(wb));
            }
        }
        __syncthreads();                                                    // every thread has read the list
        // no group was due: the list stays as it is (and the block's best has not moved: nothing to exchange) -- four
        // barriers and a global round trip less per chunk; long sequences run dozens of such chunks per launch
        if (done == 0) continue;
        if (!solo && tid == 0) {                                            // exchange with the structure's other blocks
            const unsigned long long mine = s_best;
            const unsigned long long seen = mine ? atomicMax(a.best + st.slot, mine) : atomicMax(a.best + st.slot, 0ull);
            if (seen > mine) s_best = seen;
        }
        const uint32_t rem = ns > done ? ns - done : 0u;                    // < blockDim: carried to the next chunk
        uint32_t ck = 0; uint16_t cl = 0; double cb = 0.0;
        if ((uint32_t)tid < rem) { ck = s_key[done + tid]; cl = s_len[done + tid]; cb = s_bps[done + tid]; }
        __syncthreads();
        if ((uint32_t)tid < rem) { s_key[tid] = ck; s_len[tid] = cl; s_bps[tid] = cb; }
        if (tid == 0) s_nsurv = rem;
        __syncthreads();
#ifdef SQ_SCORE_PROF
        _pb += wall_clock64() - _t1;
#endif
    }
#ifdef SQ_SCORE_PROF
    if (tid == 0 && (blockIdx.x % 509) == 0 && blockIdx.y == 0)
        printf("score block %d: n=%d ncand=%u nstrand=%d threads %d | first wave: passed :492 %d, scored %d in %d groups | us: setup %.1f phaseA %.1f phaseB %.1f total %.1f\n", (int)blockIdx.x, n, ncand, st.nstrand, nthr,
               _n492, _nsc, _ngr, _ps * 0.01, _pa * 0.01, _pb * 0.01, (wall_clock64() - _p0) * 0.01);
#endif

    if (solo) {                                         // the length of the survivor list (zeroed by the state kernel)
        __syncthreads();
        if (tid == 0) a.ok_cnt[st.slot] = s_okn < ok_cap ? s_okn : ok_cap;
    }
    // wave maximum, then one atomicMax per wave on the structure's slot
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oa = __shfl_xor(any, off);
        if (oa && (!any || ob > best)) { any = 1; best = ob; }
    }
    if ((tid & 63) == 0 && any) atomicMax(a.best + st.slot, sq_ord(best));
}

// mode 0 (the greedy rounds): bpscore filter + ScoreStems, two phases (see the body)
extern "C" __global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(SQ_SCORE_WAVES))) void sq_score_kernel(SqDevCtx c, const SqStruct *structs,
                                                                  const SqStrand *strands, SqState stt, SqScanArgs a,
                                                                  SqRoundIO io, int lds_n, int lds_n_reacts, int lds_n_state, int surv_off,
                                                                  int cell_off, int str_off, int str_cap, SqCtxTab ct, int bound)
{
    sq_score_body<true>(c, structs, strands, stt, a, io, 0, lds_n, lds_n_reacts, lds_n_state, surv_off, cell_off, str_off, str_cap, ct, bound);
}

// modes 1 / 2 (OptimalStems output, alignment survivor list): the bpscore filter alone, as its own kernel so that its
// loop is not compiled under the register budget of ScoreStems
extern "C" __global__ __launch_bounds__(1024) void sq_bps_kernel(SqDevCtx c, const SqStruct *structs, const SqStrand *strands,
                                                                SqState stt, SqScanArgs a, SqRoundIO io, int mode, int lds_n,
                                                                int lds_n_reacts, int surv_off, int cell_off)
{
    sq_score_body<false>(c, structs, strands, stt, a, io, mode, lds_n, lds_n_reacts, 0, surv_off, cell_off);
}

// ChooseStems range filter (:769-778): candidates within subopt * best of the structure's best finalscore
extern "C" __global__ __launch_bounds__(256) void sq_select_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a,
                                                                 SqRoundIO io)
{
    const SqStruct st = structs[blockIdx.x];
    const unsigned long long ob = a.best[st.slot];
    if (ob == 0ull) return;
    const SqJob jb = c.jobs[st.job];
    const uint32_t nok = a.ok_cnt[st.slot];
    const SqOk *oks = sq_oks(a, st, jb.cand_cap);
    const double range = st.subopt * sq_unord(ob);                      // :769
    for (uint32_t q = blockIdx.y * 256 + threadIdx.x; q < nok; q += gridDim.y * 256) {
        const SqOk cd = oks[q];
        if (!(cd.fin < range)) {                                        // :778
            const SqOut r = {(int32_t)blockIdx.x, cd.key, (int32_t)cd.len, 0, cd.bps, cd.fin};
            sq_put_out(io, a, r);
        }
    }
}

// ------------------------------------------------------------------------------------
// alignment step 1 (SQRNdbnali.py:233-237): the stems of ONE sequence (structure sidx of the round, already
// re-scored by sq_score_kernel in mode 2) added into the L x L column matrix through the gap map.  The cells of
// one sequence's stems are distinct, so plain read-modify-writes are race-free; sequences are separate launches
// in stream order, which is the reference's summation order per cell.
// ------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void sq_scatter_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a,
                                                                   int sidx, const int32_t *cols, int L, double *matrix)
{
    const SqStruct st = structs[sidx];
    const SqJob jb = c.jobs[st.job];
    const uint32_t nok = a.ok_cnt[st.slot];
    const SqOk *oks = sq_oks(a, st, jb.cand_cap);
    for (uint32_t q = blockIdx.x * 256 + threadIdx.x; q < nok; q += gridDim.x * 256) {
        const SqOk cd = oks[q];
        const int s = (int)(cd.key >> 16), i0 = (int)(cd.key & 0xFFFFu), j0 = s - i0;
        for (int t = 0; t < (int)cd.len; t++) {
            const int64_t v = cols[i0 + t], w = cols[j0 - t];
            matrix[v * L + w] += cd.bps;                   // v < w: the lower triangle is mirrored at the end
        }
    }
}

// The same accumulation for ALL sequences of a chunk in one launch (grid.y = structure), with hardware fp64 atomic
// adds.  Only used when every addend and every partial sum is exactly representable (weights are multiples of
// 2^-k, no reactivity factors: sq_align_accumulate checks), so the order of the additions cannot change a bit.
extern "C" __global__ __launch_bounds__(256) void sq_scatter_all_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a,
                                                                       const int32_t *cols, const int32_t *col_start, int L,
                                                                       double *matrix)
{
    const SqStruct st = structs[blockIdx.x];
    const SqJob jb = c.jobs[st.job];
    const uint32_t nok = a.ok_cnt[st.slot];
    const SqOk *oks = sq_oks(a, st, jb.cand_cap);
    const int32_t *mycols = cols + col_start[blockIdx.x];
    for (uint32_t q = blockIdx.y * 256 + threadIdx.x; q < nok; q += gridDim.y * 256) {
        const SqOk cd = oks[q];
        const int s = (int)(cd.key >> 16), i0 = (int)(cd.key & 0xFFFFu), j0 = s - i0;
        for (int t = 0; t < (int)cd.len; t++) {
            const int64_t v = mycols[i0 + t], w = mycols[j0 - t];
            unsafeAtomicAdd(&matrix[v * L + w], cd.bps);   // v < w
        }
    }
}

// lower triangle := transpose of the upper one (32 x 32 tiles through LDS, both sides coalesced)
extern "C" __global__ __launch_bounds__(256) void sq_mirror_kernel(double *matrix, int L)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;               // tile (by, bx) of the upper triangle: bx >= by
    if (bx < by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
    for (int r = ty; r < 32; r += 8) {
        const int v = by * 32 + r, w = bx * 32 + tx;
        tile[r][tx] = (v < L && w < L) ? matrix[(int64_t)v * L + w] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int w = bx * 32 + r, v = by * 32 + tx;         // writes row w, columns v
        if (w < L && v < L && v < w) matrix[(int64_t)w * L + v] = tile[tx][r];
    }
}

extern "C" __global__ __launch_bounds__(256) void sq_colselect_kernel(const double *matrix, int L, double thr, int minspan,
                                                                     long long *idx_out, double *val_out, long long cap,
                                                                     unsigned long long *count)
{
    // A block takes rows blockIdx.x, + gridDim.x, ...; the cells it selects are staged in LDS and written out a thousand at a time
    // behind ONE atomic on the counter.  (Iteration 1 of a conserved alignment selects hundreds of thousands of cells: an atomic
    // per cell -- and, tried first in round 6, per wave -- on one address was the kernel's whole 1.8 ms for 100 MB of reads.)
    __shared__ long long s_idx[1024];
    __shared__ double s_val[1024];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) s_n = 0u;
    __syncthreads();
    auto flush = [&]() {                                    // (block-uniform call, between barriers)
        const uint32_t n = s_n;
        if (tid == 0) s_base = atomicAdd(count, (unsigned long long)n);
        __syncthreads();
        const unsigned long long base = s_base;
        for (uint32_t k = tid; k < n; k += 256) if ((long long)(base + k) < cap) { idx_out[base + k] = s_idx[k]; val_out[base + k] = s_val[k]; }
        __syncthreads();
        if (tid == 0) s_n = 0u;
        __syncthreads();
    };
    for (int v = blockIdx.x; v < L; v += gridDim.x) {
        const double *row = matrix + (int64_t)v * L;
        for (int wb = max(v + minspan, 0); wb < L; wb += 256) {                                                 // w - v >= minspan (:147)
            const int w = wb + tid;
            const double x = w < L ? row[w] : 0.0;
            const bool hit = w < L && x >= thr;
            const unsigned long long m = __ballot(hit);
            if (m != 0ull) {
                uint32_t b0 = 0u;
                if (lane == 0) b0 = atomicAdd(&s_n, (uint32_t)__popcll(m));
                b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
                if (hit) { const uint32_t at = b0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); s_idx[at] = (int64_t)v * L + w; s_val[at] = x; }
            }
            __syncthreads();
            if (s_n > 768u) flush();                         // (room for the next 256)
        }
    }
    if (s_n > 0u) flush();
}
