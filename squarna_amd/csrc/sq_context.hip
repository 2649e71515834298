// sq_context.hip -- the tables behind ScoreStems' closed-form strand sweep (sq_context.h): one block per structure of the
// round, built from the structure's sorted strand list in LDS and written to the table slice of the structure's state slot.
#include <hip/hip_runtime.h>
#include "sq_context.h"

#define SQ_CTX_THREADS 256
#define SQ_CTX_PER 5                      // gaps per thread: cap <= 1,280
#define SQ_CTX_MAXCAP 1025                // 1,024 strands (the scoring kernel's LDS strand list) + 1

// exclusive prefix sum over the block (256 threads = 4 waves); total: the sum over all threads
__device__ __forceinline__ int sq_ctx_scan(int v, int *wave_tot, int &total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wv; q++) base += wave_tot[q];
    total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return base + x - v;
}

extern "C" __global__ __launch_bounds__(SQ_CTX_THREADS) void sq_context_kernel(const SqStruct *structs, const SqStrand *strands, SqCtxTab t)
{
    __shared__ SqStrand s_str[SQ_CTX_MAXCAP];
    __shared__ uint16_t s_pi[SQ_CTX_MAXCAP], s_pl[SQ_CTX_MAXCAP + 1];
    __shared__ int16_t s_dep[SQ_CTX_MAXCAP];
    __shared__ uint16_t s_nr[SQ_CTX_MAXCAP], s_nl[SQ_CTX_MAXCAP], s_fr[SQ_CTX_MAXCAP], s_fl[SQ_CTX_MAXCAP];
    __shared__ uint16_t s_v[6][SQ_CTX_MAXCAP];            // rb rc rw lb lc lw
    __shared__ uint16_t s_m[2][SQ_CTX_MAXCAP];
    __shared__ int s_wave[4];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const SqStruct st = structs[blockIdx.x];
    if (st.nstrand < 0) return;                            // final since an earlier round: nothing reads its tables
    const int ns = st.nstrand, cap = t.cap;
    uint8_t *okp = t.ok + st.slot;
    if (ns + 1 > cap || ns + 1 > SQ_CTX_MAXCAP) { if (tid == 0) *okp = 0; return; }
    const SqStrand *S = strands + st.strand_off;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    bool bad = false;
    for (int k = tid; k < ns; k += SQ_CTX_THREADS) { const SqStrand x = S[k]; s_str[k] = x; bad |= x.level != 1; }
    // ---- prefix sums over the strands: lengths (s_pl), depth before each strand (s_dep) ----
    const int chunk = (ns + SQ_CTX_THREADS - 1) / SQ_CTX_THREADS;
    const int k0 = min(tid * chunk, ns), k1 = min(k0 + chunk, ns);
    __syncthreads();
    int sl = 0, sd = 0;
    for (int k = k0; k < k1; k++) { sl += s_str[k].len; sd += s_str[k].left ? 1 : -1; }
    int tot_l, tot_d;
    int bl = sq_ctx_scan(sl, s_wave, tot_l);
    int bd = sq_ctx_scan(sd, s_wave, tot_d);
    for (int k = k0; k < k1; k++) {
        s_pl[k] = (uint16_t)bl; s_dep[k] = (int16_t)bd;
        bad |= bd < 0;
        bl += s_str[k].len; bd += s_str[k].left ? 1 : -1;
    }
    if (tid == 0) { s_pl[ns] = (uint16_t)tot_l; s_dep[ns] = (int16_t)tot_d; }
    bad |= tot_d != 0;
    // ---- partner strand of every strand: the strand that starts at the partner of this strand's last position ----
    for (int k = tid; k < ns; k += SQ_CTX_THREADS) {
        const SqStrand x = s_str[k];
        const int want = x.pstart - (x.len - 1);
        int lo = 0, hi = ns;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_str[mid].start < want) lo = mid + 1; else hi = mid; }
        if (lo >= ns || s_str[lo].start != want || s_str[lo].left == x.left || (x.left ? lo <= k : lo >= k)) { bad = true; lo = k; }
        s_pi[k] = (uint16_t)lo;
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (s_bad) { if (tid == 0) *okp = 0; return; }         // crossing stems (or a list this closed form does not cover): the walk
    // ---- chains ----
    for (int g = tid; g <= ns; g += SQ_CTX_THREADS) {
        uint16_t nr = (uint16_t)ns, rb = 0, rc = 0, rw = 0, fr = (uint16_t)ns;
        if (g < ns) {
            const SqStrand x = s_str[g];
            if (x.left) {
                const int p = s_pi[g];
                nr = (uint16_t)(p + 1); rb = 1; fr = (uint16_t)g;
                rc = (uint16_t)((x.pstart - x.start + 1) - ((int)s_pl[p + 1] - (int)s_pl[g]));
            } else { nr = (uint16_t)(g + 1); rw = (uint16_t)x.len; fr = (uint16_t)(g + 1); }
        }
        uint16_t nl = 0, lb = 0, lc = 0, lw = 0, fl = 0;
        if (g > 0) {
            const SqStrand x = s_str[g - 1];
            if (!x.left) {
                const int o = s_pi[g - 1];
                const SqStrand y = s_str[o];
                nl = (uint16_t)o; lb = 1; fl = (uint16_t)g;
                lc = (uint16_t)((y.pstart - y.start + 1) - ((int)s_pl[g] - (int)s_pl[o]));
            } else { nl = (uint16_t)(g - 1); lw = (uint16_t)x.len; fl = (uint16_t)(g - 1); }
        }
        s_nr[g] = nr; s_nl[g] = nl; s_fr[g] = fr; s_fl[g] = fl;
        s_v[0][g] = rb; s_v[1][g] = rc; s_v[2][g] = rw; s_v[3][g] = lb; s_v[4][g] = lc; s_v[5][g] = lw;
    }
    __syncthreads();
    // pointer jumping: after r rounds every gap holds the sums over the next 2^r links of its chain
    for (int span = 1; span <= ns; span <<= 1) {
        uint16_t nr[SQ_CTX_PER], nl[SQ_CTX_PER], fr[SQ_CTX_PER], fl[SQ_CTX_PER], v[SQ_CTX_PER][6];
#pragma unroll
        for (int q = 0; q < SQ_CTX_PER; q++) {
            const int g = tid + q * SQ_CTX_THREADS;
            if (g <= ns) {
                const int a = s_nr[g], b = s_nl[g];
                nr[q] = s_nr[a]; nl[q] = s_nl[b];
                fr[q] = s_fr[s_fr[g]]; fl[q] = s_fl[s_fl[g]];
#pragma unroll
                for (int c = 0; c < 3; c++) { v[q][c] = (uint16_t)(s_v[c][g] + s_v[c][a]); v[q][3 + c] = (uint16_t)(s_v[3 + c][g] + s_v[3 + c][b]); }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SQ_CTX_PER; q++) {
            const int g = tid + q * SQ_CTX_THREADS;
            if (g <= ns) {
                s_nr[g] = nr[q]; s_nl[g] = nl[q]; s_fr[g] = fr[q]; s_fl[g] = fl[q];
#pragma unroll
                for (int c = 0; c < 6; c++) s_v[c][g] = v[q][c];
            }
        }
        __syncthreads();
    }
    // ---- out: records, depths, the sparse table of rightmost minima ----
    SqCtxRec *rec = t.rec + (size_t)st.slot * cap;
    int16_t *dep = t.depth + (size_t)st.slot * cap;
    uint16_t *rmq = t.rmq + (size_t)st.slot * cap * t.levels;
    for (int g = tid; g <= ns; g += SQ_CTX_THREADS) {
        SqCtxRec r;
        r.rb = s_v[0][g]; r.rc = s_v[1][g]; r.rw = s_v[2][g]; r.lb = s_v[3][g]; r.lc = s_v[4][g]; r.lw = s_v[5][g];
        r.fr = s_fr[g];                                    // first opener at or behind g (ns: none)
        const int q = s_fl[g];                             // gap behind the last closer before g (0: none)
        r.fl = q > 0 ? s_pi[q - 1] : (uint16_t)0;
        rec[g] = r; dep[g] = s_dep[g];
        s_m[0][g] = (uint16_t)g;
    }
    __syncthreads();
    for (int j = 1; j <= t.levels; j++) {
        const int half = 1 << (j - 1);
        const uint16_t *src = s_m[(j - 1) & 1];
        uint16_t *dst = s_m[j & 1];
        for (int g = tid; g <= ns; g += SQ_CTX_THREADS) {
            int c = src[g];
            if (g + half <= ns) { const int c2 = src[g + half]; if (s_dep[c2] <= s_dep[c]) c = c2; }
            dst[g] = (uint16_t)c;
            rmq[(size_t)(j - 1) * cap + g] = (uint16_t)c;
        }
        __syncthreads();
    }
    if (tid == 0) *okp = 1;
}

size_t sq_context_bytes_per_gap(int cap, int *levels)
{
    int lv = 0;
    while ((2 << lv) <= cap) lv++;                         // floor(log2(cap))
    if (lv < 1) lv = 1;
    if (levels) *levels = lv;
    return sizeof(SqCtxRec) + 2 + 2 * (size_t)lv;
}

void sq_launch_context(const SqStruct *structs, const SqStrand *strands, const SqCtxTab &t, int S, hipStream_t st)
{
    hipLaunchKernelGGL(sq_context_kernel, dim3(S), dim3(SQ_CTX_THREADS), 0, st, structs, strands, t);
}
