// sq_algos_dev.hip -- RunAlgo (SQRNdbnseq.py:548-595) around the matching kernels, on the device.
//
// The host-driven form (sq_algos.hip) brings the AnnotateStems pass of every E / H / N job to the host, builds the edge
// lists there, reads the matchings back and runs the reference's stem filters in C++: the largest host phase of a fold.
// Here the stems never leave the device:
//   sq_algo_sizes_kernel    per job: number of edges (= cells of its stems) and, for Edmonds, of graph vertices -- the
//                           one thing the host needs (it lays out scratch and plans the blossom kernel's LDS bins)
//   sq_algo_edges_kernel    per job: the stems in the reference's emission order (key ascending), their cells as the
//                           edge list the matching kernels read -- weights stemscore ** 1.7 from the host-libm table of the
//                           paramset (SQRNalgos.py:101,122), Nussinov: the score itself (:49) --, for Edmonds the
//                           vertex numbering by first appearance (networkx's node order, :98-109)
//   sq_algo_finish_kernel   per job, behind the matching kernel: matched pairs (Edmonds: mates; Hungarian: mutual
//                           assignments of stem cells, :130-133; Nussinov: BackTrack's list) -> PairsToStems -> the score /
//                           length filter with exactly re-summed cells -> pseudoknot level limit -> short pseudoknotted
//                           stems dropped (:570-595) -> the job's stemset appended to the device log of final structures
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/squarna_hip.h"
#include "sq_device.h"
#include "sq_extend.h"
#include "sq_match.h"
#include "sq_tail_dev.h"
#include "sq_algos_dev.h"
#include "sq_cells.h"

__device__ __forceinline__ double sq_algo_cell(const SqDevCtx &c, const SqJob &jb, const SqPsetDev *ps, int i, int j) { return sq_cell_exact(c, jb, ps, i, j); }

// ---- sizes ----------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void sq_algo_sizes_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, SqAlgoSize *sizes, SqAlgoRaw raw)
{
    __shared__ uint32_t s_seen[SQ_ALGO_MAXN / 32];
    __shared__ int s_edges, s_nv, s_rawbase;
    const SqStruct st = structs[blockIdx.x];
    const SqJob jb = c.jobs[st.job];
    const int tid = threadIdx.x, nthr = (int)blockDim.x;   // (256 threads for a batch alone, one wave per job on a crowded chip)
    const uint32_t nok = a.ok_cnt[st.slot];
    const SqOk *oks = sq_oks(a, st, jb.cand_cap);
    for (int w = tid; w < (jb.n + 31) / 32; w += nthr) s_seen[w] = 0u;
    if (tid == 0) {
        s_edges = 0; s_nv = 0;
        int rb = -1;
        if (raw.need && raw.need[blockIdx.x]) {                          // the job's stem scores go to the host (see SqAlgoRaw)
            const uint32_t at = atomicAdd(raw.ctr, nok);
            rb = at + nok <= raw.cap ? (int)at : -2;
        }
        s_rawbase = rb;
    }
    __syncthreads();
    const int rawbase = s_rawbase;
    int e = 0;
    for (uint32_t q = tid; q < nok; q += nthr) {
        const SqOk cd = oks[q];
        if (rawbase >= 0) raw.vals[rawbase + q] = cd.bps;
        const int s = (int)(cd.key >> 16), i0 = (int)(cd.key & 0xFFFFu), j0 = s - i0;
        e += (int)cd.len;
        for (int t = 0; t < (int)cd.len; t++) {
            atomicOr(&s_seen[(i0 + t) >> 5], 1u << ((i0 + t) & 31));
            atomicOr(&s_seen[(j0 - t) >> 5], 1u << ((j0 - t) & 31));
        }
    }
    atomicAdd(&s_edges, e);
    __syncthreads();
    int nv = 0;
    for (int w = tid; w < (jb.n + 31) / 32; w += nthr) nv += __popc(s_seen[w]);
    atomicAdd(&s_nv, nv);
    __syncthreads();
    if (tid == 0) sizes[blockIdx.x] = SqAlgoSize{s_edges, s_nv, (int32_t)nok, rawbase};
}

// ---- edge lists -----------------------------------------------------------------------------------------------------
// scratch of a structure: the SqKey part of its candidate slice (dead once the bpscore filter has run):
// [nok x uint32 sorted survivor index][nok x uint32 first edge of that stem]
//
// nokcap > 0: the launch's jobs keep their stem lists in LDS (sq_algo_edges_lds): keys, lengths, the order and the edge
// offsets -- the reference's emission order by a bucket sort over the anti-diagonals (key = diagonal << 16 | row: a bucket
// holds the few stems of one diagonal), O(stems).  Until round 4 every thread ranked its stem against ALL keys in global
// memory and every later pass chased survivor index -> record through L2: 85-110 us for SRtest150's 657 jobs, a
// quarter of it left.  nokcap == 0 (lists too long for LDS): that form.
// exclusive prefix sum of v[0 .. cnt) in place by the block (256 threads); returns the total to every thread
__device__ __forceinline__ uint32_t sq_block_excl_scan(uint32_t *v, int cnt, uint32_t *s_wsum, uint32_t *s_run, int tid)
{
    const int lane = tid & 63, wave = tid >> 6, nthr = (int)blockDim.x, nwv = nthr >> 6;
    if (tid == 0) *s_run = 0;
    __syncthreads();
    for (int x0 = 0; x0 < cnt; x0 += nthr) {
        const int x = x0 + tid;
        const uint32_t mine = x < cnt ? v[x] : 0u;
        uint32_t inc = mine;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(inc, d, 64); if (lane >= d) inc += y; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t before = *s_run;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        if (x < cnt) v[x] = before + inc - mine;
        __syncthreads();
        if (tid == 0) { uint32_t a = 0; for (int w = 0; w < nwv; w++) a += s_wsum[w]; *s_run += a; }
        __syncthreads();
    }
    return *s_run;
}

extern "C" __global__ __launch_bounds__(256) void sq_algo_edges_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, const SqAlgoJob *jobs, int maxn_lds,
                                                                      SqAlgoStatPtrs zs, int nokcap)
{
    if (blockIdx.x == 0 && threadIdx.x < 3 && zs.p[threadIdx.x]) {      // (the finish kernels of the items count into these)
        SqAlgoStat z; memset(&z, 0, sizeof(z)); *zs.p[threadIdx.x] = z;
    }
    // (dynamic LDS sized for the batch's longest sequence -- 6 bytes per position; static arrays for 4,096 nt cost every block 24 KB)
    extern __shared__ __attribute__((aligned(16))) char sq_edges_dyn[];
    int32_t *const s_first = reinterpret_cast<int32_t *>(sq_edges_dyn);        // Edmonds: first edge slot a position appears in
    int16_t *const s_id = reinterpret_cast<int16_t *>(s_first + maxn_lds);    // position -> vertex id
    __shared__ uint32_t s_wsum[4], s_run;
    const SqStruct st = structs[blockIdx.x];
    const SqJob jb = c.jobs[st.job];
    const SqAlgoJob aj = jobs[blockIdx.x];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = jb.n, nthr = (int)blockDim.x, nwv = nthr >> 6;   // (256 threads, or one wave on a crowded chip)
    const uint32_t nok = a.ok_cnt[st.slot];
    const SqOk *oks = sq_oks(a, st, jb.cand_cap);
    uint32_t *sidx = reinterpret_cast<uint32_t *>(sq_keys(a, st));
    uint32_t *eoff = sidx + nok;
    const bool fast = nokcap > 0 && nok <= (uint32_t)nokcap;
    // LDS lists of the fast form, behind the per-position arrays: [diagonal starts 2 maxn + 2][diagonal fills 2 maxn + 2]
    // [keys][edge offsets][lengths][sorted order][bucket order]
    uint32_t *const s_dstart = reinterpret_cast<uint32_t *>(sq_edges_dyn + (((size_t)6 * maxn_lds + 15) & ~(size_t)15));
    uint32_t *const s_dfill = s_dstart + 2 * maxn_lds + 2;
    uint32_t *const s_key = s_dfill + 2 * maxn_lds + 2;
    uint32_t *const s_eoff = s_key + nokcap;
    uint32_t *const s_tmp = s_eoff + nokcap;
    uint32_t *const s_sidx = s_tmp + nokcap;
    uint32_t *const s_bord = s_sidx + nokcap;
    if (fast) {
        const int nd = 2 * n + 1;                                       // diagonals s = i + j < 2 n
        for (int d = tid; d <= nd; d += nthr) { s_dstart[d] = 0u; s_dfill[d] = 0u; }
        __syncthreads();
        for (uint32_t x = tid; x < nok; x += nthr) {
            const SqOk cd = oks[x];
            s_key[x] = cd.key; s_tmp[x] = cd.len;                       // (lengths in emission order follow below)
            atomicAdd(&s_dstart[cd.key >> 16], 1u);
        }
        __syncthreads();
        sq_block_excl_scan(s_dstart, nd, s_wsum, &s_run, tid);
        // the stems of a diagonal in any order, then every stem's place among them: key ascending
        for (uint32_t x = tid; x < nok; x += nthr) {
            const uint32_t d = s_key[x] >> 16;
            s_bord[s_dstart[d] + atomicAdd(&s_dfill[d], 1u)] = x;      // (the stems of a diagonal, in any order)
        }
        __syncthreads();
        for (uint32_t p = tid; p < nok; p += nthr) {
            const uint32_t x = s_bord[p], k = s_key[x], d = k >> 16;
            const uint32_t lo = s_dstart[d], hi = lo + s_dfill[d];
            uint32_t r = lo;
            for (uint32_t q = lo; q < hi; q++) r += s_key[s_bord[q]] < k ? 1u : 0u;
            s_sidx[r] = x; sidx[r] = x;
        }
        __syncthreads();
        for (uint32_t x = tid; x < nok; x += nthr) s_eoff[x] = s_tmp[s_sidx[x]];   // lengths in emission order
        __syncthreads();
        sq_block_excl_scan(s_eoff, (int)nok, s_wsum, &s_run, tid);
        for (uint32_t x = tid; x < nok; x += nthr) eoff[x] = s_eoff[x];
    } else {
    // the reference's emission order: anti-diagonal ascending, then row ascending == key ascending (keys are distinct).  Lists
    // too long for LDS (a 500-nt sequence under Nussinov's threshold: 8,000 stems) take the same bucket sort over the
    // anti-diagonals with the bucket order in global memory (the edge offsets' array, which is written last): until round 6
    // every stem was ranked against ALL keys -- O(stems^2) reads of global memory, 15-30 ms for the 1,500-3,000 jobs of a
    // batch of 500-nt records, in front of the greedy loop on the batch's stream
    {
        const int nd = 2 * n + 1;
        uint32_t *const bord = eoff;
        for (int d = tid; d <= nd; d += nthr) { s_dstart[d] = 0u; s_dfill[d] = 0u; }
        __syncthreads();
        for (uint32_t x = tid; x < nok; x += nthr) atomicAdd(&s_dstart[oks[x].key >> 16], 1u);
        __syncthreads();
        sq_block_excl_scan(s_dstart, nd, s_wsum, &s_run, tid);
        for (uint32_t x = tid; x < nok; x += nthr) {
            const uint32_t d = oks[x].key >> 16;
            bord[s_dstart[d] + atomicAdd(&s_dfill[d], 1u)] = x;        // (the stems of a diagonal, in any order)
        }
        __threadfence_block();
        __syncthreads();
        for (uint32_t p = tid; p < nok; p += nthr) {
            const uint32_t x = bord[p], k = oks[x].key, d = k >> 16;
            const uint32_t lo = s_dstart[d], hi = lo + s_dfill[d];
            uint32_t r = lo;
            for (uint32_t q = lo; q < hi; q++) r += oks[bord[q]].key < k ? 1u : 0u;
            sidx[r] = x;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) s_run = 0;
    __threadfence_block();
    __syncthreads();
    // first edge of every stem: exclusive prefix sum of the lengths in that order
    for (uint32_t x0 = 0; x0 < nok; x0 += nthr) {
        const uint32_t x = x0 + tid;
        const uint32_t len = x < nok ? oks[sidx[x]].len : 0u;
        uint32_t inc = len;                                             // inclusive scan inside the wave
        for (int d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inc, d, 64); if (lane >= d) inc += v; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        uint32_t before = s_run;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        if (x < nok) eoff[x] = before + inc - len;
        __syncthreads();
        if (tid == 0) { uint32_t a2 = 0; for (int w = 0; w < nwv; w++) a2 += s_wsum[w]; s_run += a2; }
        __syncthreads();
    }
    }
    __threadfence_block();
    __syncthreads();
    // (the fast form reads order, key, length and offset from LDS; a stem's length: the next offset minus its own)
    auto stem_of = [&](uint32_t x, uint32_t &key, int &len, int &e0, uint32_t &src) {
        if (fast) {
            src = s_sidx[x]; key = s_key[src]; e0 = (int)s_eoff[x]; len = (int)s_tmp[src];
        } else {
            src = sidx[x];
            const SqOk cd = oks[src];
            key = cd.key; len = (int)cd.len; e0 = (int)eoff[x];
        }
    };
    if (aj.algo == SQ_ALGO_E) {
        // networkx numbers the nodes in order of first appearance in the edge list (v before w of every edge)
        for (int p = tid; p < n; p += nthr) s_first[p] = 0x7FFFFFFF;
        __syncthreads();
        for (uint32_t x = tid; x < nok; x += nthr) {
            uint32_t key, src; int len, e0;
            stem_of(x, key, len, e0, src);
            const int s = (int)(key >> 16), i0 = (int)(key & 0xFFFFu), j0 = s - i0;
            for (int t = 0; t < len; t++) { atomicMin(&s_first[i0 + t], 2 * (e0 + t)); atomicMin(&s_first[j0 - t], 2 * (e0 + t) + 1); }
        }
        __syncthreads();
        for (int p = tid; p < n; p += nthr) {
            const int f = s_first[p];
            int id = -1;
            if (f != 0x7FFFFFFF) { id = 0; for (int q = 0; q < n; q++) id += s_first[q] < f ? 1 : 0; aj.vid2pos[id] = p; }
            s_id[p] = (int16_t)id;
        }
        __syncthreads();
    }
    for (uint32_t x = tid; x < nok; x += nthr) {
        uint32_t key, src; int len, e0;
        stem_of(x, key, len, e0, src);
        const int s = (int)(key >> 16), i0 = (int)(key & 0xFFFFu), j0 = s - i0;
        double wt;
        if (aj.algo == SQ_ALGO_N) wt = oks[src].bps;                    // Nussinov: the score itself (SQRNalgos.py:49)
        else if (aj.raw) wt = aj.raw[src];                              // :101,122: the host libm's power -- the job's list,
        else wt = c.powtab[ps->pow_off + (int)(oks[src].bps * ps->pow_scale)];   // or the paramset's table (scores k 2^-q exactly)
        SqMatchEdge *e = aj.edges + e0;
        for (int t = 0; t < len; t++) {
            const int v = i0 + t, w = j0 - t;
            e[t] = aj.algo == SQ_ALGO_E ? SqMatchEdge{s_id[v], s_id[w], wt} : SqMatchEdge{v, w, wt};
        }
    }
}

// ---- RunAlgo's filters behind the matching ---------------------------------------------------------------------------
// one wave per job.  LDS: pairs (p, q) [n / 2 + 1 each], the stems of the matching and the level scratch
extern "C" __global__ __launch_bounds__(64) void sq_algo_finish_kernel(SqDevCtx c, const SqAlgoJob *jobs, const SqMatchJob *mj, const int32_t *out,
                                                                      const int32_t *cnt, int levellimit_opt, SqPoolFin *fin, SqPoolStem *fin_stems,
                                                                      uint32_t *fin_ctr, uint32_t fin_cap, uint32_t fin_stem_cap, SqAlgoStat *stats,
                                                                      int tcap)
{
    extern __shared__ __attribute__((aligned(16))) char sq_fin_dyn[];
    const int q = blockIdx.x, lane = threadIdx.x;
#ifdef SQ_FIN_PROF
    long long _ft[10]; int _fk = 0;
#define FINPROF() do { _ft[_fk++] = wall_clock64(); } while (0)
#else
#define FINPROF() do {} while (0)
#endif
    FINPROF();
    const SqAlgoJob aj = jobs[q];
    const SqMatchJob m = mj[q];
    const SqJob jb = c.jobs[aj.job];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n, half = n / 2 + 2;
    { int x_ = n + (int)m.n + aj.algo; asm volatile("" :: "v"(x_)); } FINPROF();
    int16_t *pp = reinterpret_cast<int16_t *>(sq_fin_dyn), *pq = pp + half, *sp = pq + half, *sq2 = sp + half;   // pairs, sorted pairs
    double *bps = reinterpret_cast<double *>(sq_fin_dyn + ((8 * (size_t)half + 15) & ~(size_t)15));               // per stem
    int16_t *keep = reinterpret_cast<int16_t *>(bps + half);                                                      // kept stem indices
    SqExtendLds L = sq_extend_lds(reinterpret_cast<char *>(keep + half + ((half & 1) ? 1 : 0) + 4), tcap);
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); };
    // ---- the matched pairs ----
    int np = 0;
    if (aj.algo == SQ_ALGO_E) {
        const int32_t *mate = out + m.out_off;
        const int nv = m.n;
        if (nv > 0 && mate[0] == -2) { if (lane == 0) stats->bad = 1; return; }       // blossom capacity exceeded
        // (the measurement -- the graph with the most scan passes, their sum -- is the publish kernel's: three atomics per job on
        // ONE record every wave of the launch shares stood in front of this wave's loads on the in-order memory counter)
        for (int v0 = 0; v0 < nv; v0 += 64) {
            const int v = v0 + lane;
            const int mt = v < nv ? mate[v] : -1;
            const bool is = v < nv && mt > v;
            const unsigned long long bal = __ballot(is);
            if (is) {
                const int k = np + __popcll(bal & ((1ull << lane) - 1ull));
                int a0 = aj.vid2pos[v], b0 = aj.vid2pos[mt];
                if (a0 > b0) { const int t = a0; a0 = b0; b0 = t; }
                pp[k] = (int16_t)a0; pq[k] = (int16_t)b0;
            }
            np += __popcll(bal);
        }
    } else if (aj.algo == SQ_ALGO_N) {
        const int32_t *pr = out + 2 * m.out_off;
        np = cnt[q];
        for (int k = lane; k < np; k += 64) {
            int a0 = pr[2 * k], b0 = pr[2 * k + 1];
            if (a0 > b0) { const int t = a0; a0 = b0; b0 = t; }
            pp[k] = (int16_t)a0; pq[k] = (int16_t)b0;
        }
    } else {
        // Hungarian (SQRNalgos.py:130-133): k < sol[k] with more than 3 positions (or a separator) between, assigned
        // mutually, on a cell of the matrix that is not zero == a cell of a stem whose score ** 1.7 is not zero
        const int32_t *sol = out + m.out_off;
        const uint8_t *codes = c.codes + jb.pos_off;
        const SqMatchEdge *ed = aj.edges;
        const int ne = m.nedges;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int kk = k0 + lane;
            bool is = false;
            int sk = -1;
            if (kk < n) {
                sk = sol[kk];
                is = sk >= 0 && kk < sk;
                if (is) {
                    bool far = kk < sk - 3;
                    if (!far) for (int x = kk + 1; x < sk; x++) if (codes[x] == SQ_CODE_SEP1 || codes[x] == SQ_CODE_SEP2) { far = true; break; }
                    is = far && sol[sk] == kk;
                }
                if (is) {
                    // the edge list is sorted by (v + w, v): binary search for the cell (kk, sk)
                    int lo = 0, hi = ne;
                    const int ks = kk + sk;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const int es = ed[mid].v + ed[mid].w;
                        if (es < ks || (es == ks && ed[mid].v < kk)) lo = mid + 1; else hi = mid;
                    }
                    is = lo < ne && ed[lo].v == kk && ed[lo].w == sk && !(-ed[lo].weight == 0);
                }
            }
            const unsigned long long bal = __ballot(is);
            if (is) { const int k = np + __popcll(bal & ((1ull << lane) - 1ull)); pp[k] = (int16_t)kk; pq[k] = (int16_t)sk; }
            np += __popcll(bal);
        }
    }
    wsync();
    // ---- sorted(pairs) (:570), PairsToStems (:498-517) ----
    FINPROF();
    for (int k = lane; k < np; k += 64) {
        const int a0 = pp[k], b0 = pq[k];
        int r = 0;
        for (int y = 0; y < np; y++) { const int a1 = pp[y], b1 = pq[y]; r += (a1 < a0 || (a1 == a0 && b1 < b0)) ? 1 : 0; }
        sp[r] = (int16_t)a0; sq2[r] = (int16_t)b0;
    }
    wsync();
    // stems into L.i / L.j / L.len: a pair starts a stem unless it stacks onto its predecessor in the sorted list
    int T = 0;
    for (int k0 = 0; k0 < np; k0 += 64) {
        const int k = k0 + lane;
        const bool start = k < np && !(k > 0 && sp[k - 1] + 1 == sp[k] && sq2[k - 1] == sq2[k] + 1);
        const unsigned long long bal = __ballot(start);
        if (start) {
            const int t = T + __popcll(bal & ((1ull << lane) - 1ull));
            int len = 1;
            while (k + len < np && sp[k + len - 1] + 1 == sp[k + len] && sq2[k + len - 1] == sq2[k + len] + 1) len++;
            pp[t] = sp[k]; pq[t] = sq2[k]; keep[t] = (int16_t)len;    // (pp / pq are free again: stems i, j; keep: len for now)
        }
        T += __popcll(bal);
    }
    wsync();
    // ---- first filter (:571-579): raw score re-summed from the matrix cells, left to right from 0 ----
    FINPROF();
    const double minbps = ps->minbpscore, minlen = ps->minlen;
    int K = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        bool ok = false;
        double s = 0;
        int si = 0, sj = 0, sl = 0;
        if (t < T) {
            si = pp[t]; sj = pq[t]; sl = keep[t];
            for (int k = 0; k < sl; k++) s = s + sq_algo_cell(c, jb, ps, si + k, sj - k);
            ok = s >= minbps && (double)sl >= minlen;
        }
        const unsigned long long bal = __ballot(ok);
        wsync();                                                         // (every lane has read its keep[t] before the slots are reused)
        if (ok) {
            const int k = K + __popcll(bal & ((1ull << lane) - 1ull));
            if (k < tcap) { L.i[k] = (int16_t)si; L.j[k] = (int16_t)sj; L.len[k] = (int16_t)sl; bps[k] = s; }
        }
        K += __popcll(bal);
    }
    if (K > tcap) { if (lane == 0) stats->bad = 2; return; }             // (cannot happen: disjoint stems of >= minlen pairs)
    wsync();
    // ---- level limit (:581): DBNToPairs(PairsToDBN(pairs, N, levellimit)) drops the levels above the limit ----
    FINPROF();
    const int levellimit = levellimit_opt >= 0 ? levellimit_opt : 3 - (n > 500 ? 1 : 0);   // :1043-1044
    auto levels = [&](int cntT) {                                       // L.lvl of the first cntT stems in L
        bool anyc = false;
        for (int t = lane; t < cntT; t += 64) {
            const int qi = L.i[t], qj = L.j[t];
            int cc = 0;
            for (int p = 0; p < cntT; p++) if (sq_chain_cross(qi, qj, L.i[p], L.j[p])) cc += L.len[p];
            L.cc[t] = cc;
            anyc |= cc != 0;
        }
        const bool cross = __ballot(anyc) != 0ull;
        __syncthreads();
        if (cross) sq_stem_levels_wave(L, cntT, lane, &stats->level_ovf);
        else { for (int t = lane; t < cntT; t += 64) L.lvl[t] = 1; __syncthreads(); }
    };
    levels(K);
    int K2 = 0;
    for (int t0 = 0; t0 < K; t0 += 64) {                                // compaction in place (destination index <= source index)
        const int t = t0 + lane;
        const bool ok = t < K && (levellimit < 0 || L.lvl[t] <= levellimit) && L.lvl[t] <= 49;   // 49 bracket types exist
        const unsigned long long bal = __ballot(ok);
        const int si = t < K ? L.i[t] : 0, sj = t < K ? L.j[t] : 0, sl = t < K ? L.len[t] : 0;
        const double s = t < K ? bps[t] : 0;
        __syncthreads();
        if (ok) { const int k = K2 + __popcll(bal & ((1ull << lane) - 1ull)); L.i[k] = (int16_t)si; L.j[k] = (int16_t)sj; L.len[k] = (int16_t)sl; bps[k] = s; }
        K2 += __popcll(bal);
        __syncthreads();
    }
    levels(K2);                                                          // :582 the levels of what is left
    // ---- second filter (:586-594): short pseudoknotted stems, then the same thresholds (same sums) ----
    FINPROF();
    int nout = 0;
    for (int t = lane; t < K2; t += 64) nout += (L.lvl[t] > 1 && L.len[t] < 4) ? 0 : 1;
    nout = sq_wave_sum32(nout);
    uint32_t idx = 0, so = 0;
    if (lane == 0) sq_log_reserve(fin_ctr, (uint32_t)nout, idx, so);
    idx = (uint32_t)__shfl((int)idx, 0, 64); so = (uint32_t)__shfl((int)so, 0, 64);
    FINPROF();
    if (idx >= fin_cap || so + (uint32_t)nout > fin_stem_cap) {
        if (lane == 0) {                                                 // (no slot of the log stays unwritten; the tail reports the flag)
            fin_ctr[2] = 1;
            if (idx < fin_cap) fin[idx] = SqPoolFin{aj.job, aj.algo == SQ_ALGO_E ? SQ_FIN_KIND_E : aj.algo == SQ_ALGO_H ? SQ_FIN_KIND_H : SQ_FIN_KIND_N, 0, 0, 0u, SQ_FIN_SRC_LOG};
        }
        return;
    }
    int w = 0;
    for (int t0 = 0; t0 < K2; t0 += 64) {                               // the stems stay in sorted order (ascending i)
        const int t = t0 + lane;
        const bool ok = t < K2 && !(L.lvl[t] > 1 && L.len[t] < 4);
        const unsigned long long bal = __ballot(ok);
        if (ok) fin_stems[so + w + __popcll(bal & ((1ull << lane) - 1ull))] = SqPoolStem{L.i[t], L.j[t], L.len[t], 0};
        w += __popcll(bal);
    }
    if (lane == 0) fin[idx] = SqPoolFin{aj.job, aj.algo == SQ_ALGO_E ? SQ_FIN_KIND_E : aj.algo == SQ_ALGO_H ? SQ_FIN_KIND_H : SQ_FIN_KIND_N, 0, nout, so, SQ_FIN_SRC_LOG};
#ifdef SQ_FIN_PROF
    FINPROF();
    if (lane == 0 && (q % 499) == 0) printf("finish algo=%d n=%d np=%d | us: load %.1f pairs %.1f stems %.1f filter %.1f levels %.1f count %.1f log %.1f total %.1f\n", aj.algo, n, np,
        (_ft[1] - _ft[0]) * 0.01, (_ft[2] - _ft[1]) * 0.01, (_ft[3] - _ft[2]) * 0.01, (_ft[4] - _ft[3]) * 0.01, (_ft[5] - _ft[4]) * 0.01, (_ft[6] - _ft[5]) * 0.01, (_ft[7] - _ft[6]) * 0.01, (_ft[7] - _ft[0]) * 0.01);
#endif
}

// behind the finish kernel: the launch's statistics to the host, then the completion word
extern "C" __global__ __launch_bounds__(64) void sq_algo_publish_kernel(SqAlgoStat *stats, SqAlgoStat *h_stats, const SqMatchJob *mj, const int32_t *out,
                                                                        int is_edmonds, uint32_t *flag, uint32_t value, int nj)
{
    // Edmonds: the graph with the most scan passes (the blossom kernel's critical path), the passes of all graphs and their
    // number -- read from the kernel's result blocks by ONE wave, eight jobs per lane in flight (the finish kernel's waves
    // used to count them with three atomics each on this one record)
    const int lane = threadIdx.x;
    unsigned long long mx = 0ull, sum = 0ull, cnt = 0ull;
    if (is_edmonds)
        for (int q0 = lane; q0 < nj; q0 += 64 * 8) {
            int nn[8], oo[8], first[8], pass[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int q = q0 + 64 * k; const bool ok = q < nj; nn[k] = ok ? mj[q].n : 0; oo[k] = ok ? mj[q].out_off : 0; }
#pragma unroll
            for (int k = 0; k < 8; k++) { first[k] = nn[k] > 0 ? out[oo[k]] : -2; pass[k] = nn[k] > 0 ? out[oo[k] + 2 * nn[k]] : 0; }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (nn[k] <= 0 || first[k] == -2) continue;              // (no graph / capacity exceeded: the finish kernel's rule)
                const unsigned long long p = (unsigned long long)(uint32_t)pass[k];
                const unsigned long long key = (p << 32) | (uint32_t)(q0 + 64 * k);
                mx = key > mx ? key : mx; sum += p; cnt += 1ull;
            }
        }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long omx = __shfl_xor(mx, d, 64);
        mx = omx > mx ? omx : mx; sum += __shfl_xor(sum, d, 64); cnt += __shfl_xor(cnt, d, 64);
    }
    if (lane != 0) return;
    SqAlgoStat s = *stats;
    if (is_edmonds) { s.max_pass_job = mx; s.passes = sum; s.graphs = cnt; }
    if (is_edmonds && s.graphs) {
        const SqMatchJob m = mj[(uint32_t)s.max_pass_job];
        s.max_events = out[m.out_off + 2 * m.n + 1]; s.max_n = m.n; s.max_m = m.nedges;
    }
    *h_stats = s;
    sq_host_write_flush(h_stats);
    *flag = value;
}
