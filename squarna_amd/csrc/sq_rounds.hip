// sq_rounds.hip -- a-7 for width-1 pools (poollim == 1, SQRNdbnseq.py:1102-1199) as ONE launch per fold: a persistent
// block per structure that loops over its own rounds.
//
// With a pool limit of one a structure has exactly one child per round -- parent + the first stem of ChooseStems
// (:754-789) -- and no structure ever looks at another.  The launched form (sq_chain.hip) still ran every round as
// six launches over ALL structures: state, context, scan, score, chain, done; every round as slow as its slowest
// structure, every kernel re-reading what the previous one wrote.  Here a block owns its structure from the empty one
// to the final one:
//
//   * the structure's state (partner array, prefix counts of unpaired positions, sorted strands with their levels)
//     lives in LDS for the whole fold and is updated in place when a stem is chosen;
//   * AnnotateStems (:427-495) runs as a bit-diagonal scan (sq_scan.h) ONCE, for the empty structure.  Choosing a
//     stem only ever masks rows and columns (:446-451), so the maximal runs of a later round are exactly the pieces the
//     newly paired positions leave of the previous round's runs: the block keeps its list of runs (key, length, exact
//     bpscore) in its slice of the candidate arena and cuts it against the two strands of the new stem -- O(runs) per
//     round instead of O(N^2 / 32) words, no bit matrix read after the first round.  (A live restraint pair keeps its cell
//     while both ends are unpaired, :438-443; pairing an end masks it like any other cell: the same monotone rule.)
//     The bpscore of a piece is summed anew from its cells, outer -> inner from int 0 (:416), from the LDS cell table;
//   * ScoreStems (:607-751) on the runs that pass :492, behind the same branch and bound as sq_score_kernel; the best
//     finalscore with the smallest emission key among equals is ChooseStems' first element (:758 stable sort);
//   * the extension (crossing weights, pseudoknot levels, sorted strand list: sq_extend.h) by the block's first wave,
//     retirement (no stem left :1192-1193, maxstemnum :1168-1174) with the same records sq_chain_kernel writes.
//
// Results are those of the launched rounds bit for bit (tests fold both ways: SQ_NO_ROUNDS); what the kernel does not
// take (jobs with dense matrices, sequences beyond SQ_ROUNDS_MAXN) keeps the launched form.
#include <hip/hip_runtime.h>
#include "sq_device.h"

#define SQ_EXTEND_SYNC() sq_wave_lds_fence()      // (the extension runs on the first wave of a wider block)
#include "sq_extend.h"
#include "sq_tail_dev.h"
#include "sq_cells.h"
#include "sq_cellrun.h"
#include "sq_score.h"
#include "sq_scan.h"
#include "sq_rounds.h"

#ifndef SQ_ROUNDS_WAVES
#define SQ_ROUNDS_WAVES 4          // waves per SIMD the register budget allows (128 VGPRs): four 256-thread blocks per CU
#endif

// the first round's scan: every wave stages its runs in its own LDS buffer and appends them to the block's list
struct SqRoundsSink {
    uint2 *stage; uint32_t *cnt;             // this wave's staging buffer and fill count (LDS)
    uint32_t *nlist;                         // the block's list length (LDS)
    SqRun *list; uint32_t cap; SqCounters *ctr;
    __device__ __forceinline__ void put(uint32_t pos, uint32_t key, uint32_t len)
    {
        if (pos < cap) list[pos] = SqRun{key, len, __longlong_as_double(0x7FF8000000000000ll)};
        else ctr->cand_ovf = 1;
    }
    __device__ __forceinline__ void emit(uint32_t key, uint32_t len)
    {
        const uint32_t slot = atomicAdd(cnt, 1u);
        if (slot < SQ_ROUNDS_STAGE) stage[slot] = make_uint2(key, len);
        else put(atomicAdd(nlist, 1u), key, len);
    }
    __device__ __forceinline__ void flush(int lane)
    {
        sq_wave_lds_fence();
        uint32_t n = *cnt;
        if (n > SQ_ROUNDS_STAGE) n = SQ_ROUNDS_STAGE;
        if (n) {
            uint32_t b0 = 0;
            if (lane == 0) b0 = atomicAdd(nlist, n);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
            for (uint32_t k = lane; k < n; k += 64) put(base + k, stage[k].x, stage[k].y);
        }
        sq_wave_lds_fence();
        if (lane == 0) *cnt = 0;
        sq_wave_lds_fence();
    }
    __device__ __forceinline__ void poll(int lane) { sq_wave_lds_fence(); if (*cnt > SQ_ROUNDS_STAGE / 2) flush(lane); }
    __device__ __forceinline__ void drain(int lane) { flush(lane); }
};

// The two strands of a new stem k = (i0, j0, len) into the sorted strand list S[0 .. nstrand) (+ the stem index of each
// strand, X), IN PLACE, by one wave: every strand moves up by the number of new strands that start before it (0, 1 or 2), the
// chunks of 64 taken from the top so that nothing is overwritten before it is read; levels from L.lvl when stems cross.
__device__ __forceinline__ void sq_rounds_insert_strands(SqExtendLds &L, bool anycross, int k, SqStrand *S, int16_t *X, int nstrand,
                                                         int i0, int j0, int len, int lane)
{
    const int ls = i0, rs = j0 - len + 1;                               // starts of the 5' and the 3' strand (ls < rs)
    int below_l = 0, below_r = 0;
    for (int q0 = 0; q0 < nstrand; q0 += 64) {
        const int q = q0 + lane;
        const int st = q < nstrand ? S[q].start : 0x7fff;
        below_l += __popcll(__ballot(st < ls)); below_r += __popcll(__ballot(st < rs));
    }
    for (int q0 = ((nstrand + 63) & ~63) - 64; q0 >= 0; q0 -= 64) {
        const int q = q0 + lane;
        const bool valid = q < nstrand;
        SqStrand x = valid ? S[q] : SqStrand{0, 0, 0, 0, 0};
        const int sx = valid ? X[q] : 0;
        sq_wave_lds_fence();                                            // (the whole chunk is read before any of it moves)
        if (valid) {
            if (anycross) x.level = L.lvl[sx];
            const int at = q + (x.start < ls ? 0 : 1) + (x.start < rs ? 0 : 1);
            S[at] = x; X[at] = (int16_t)sx;
        }
        sq_wave_lds_fence();
    }
    if (lane == 0) {
        const uint8_t lv = anycross ? L.lvl[k] : (uint8_t)1;
        S[below_l] = SqStrand{(int16_t)ls, (int16_t)len, (int16_t)j0, lv, 1};
        S[below_r + 1] = SqStrand{(int16_t)rs, (int16_t)len, (int16_t)(i0 + len - 1), lv, 0};
        X[below_l] = (int16_t)k; X[below_r + 1] = (int16_t)k;
    }
}

extern "C" __global__ __launch_bounds__(SQ_ROUNDS_THREADS) __attribute__((amdgpu_waves_per_eu(SQ_ROUNDS_WAVES))) void sq_rounds_kernel(SqDevCtx c, SqStruct *structs, SqScanArgs a, SqChainIO cio,
                                                                                 SqRoundsArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) char rd_dyn[];
    __shared__ int s_wave_u[SQ_ROUNDS_THREADS / 64], s_wave_s[SQ_ROUNDS_THREADS / 64];
    __shared__ uint32_t s_nlist, s_ndead, s_nsurv;
    __shared__ SqCellTmp s_ctmp;
    __shared__ unsigned long long s_best;
    __shared__ double s_wfin[SQ_ROUNDS_THREADS / 64], s_wbps[SQ_ROUNDS_THREADS / 64], s_wsec[SQ_ROUNDS_THREADS / 64];
    __shared__ uint32_t s_wkey[SQ_ROUNDS_THREADS / 64], s_wlen[SQ_ROUNDS_THREADS / 64];
    __shared__ int s_wany[SQ_ROUNDS_THREADS / 64];
    __shared__ int s_cross;
#ifdef SQ_ROUNDS_PROF
    __shared__ uint32_t s_clean;
#endif

    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    const int b = blockIdx.x;
#ifdef SQ_ROUNDS_PROF
    // in-kernel timers (100 MHz wall clock): set-up, scan, cut, scoring phase A / B, pick, extension + state
    long long _pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long _t = wall_clock64(); const long long _t00 = _t;
    long long _cnt[4] = {0, 0, 0, 0};       // runs cut, runs scored (phase A), survivors (phase B), rounds
#define RPROF(k) do { const long long _n = wall_clock64(); _pt[k] += _n - _t; _t = _n; } while (0)
#define RPROF_OUT() do { if (tid == 0 && (b % 97) == 0) printf("rounds block %d clean %u n=%d rounds %lld | us: setup %.1f scan %.1f cut %.1f A %.1f B %.1f pick %.1f ext %.1f total %.1f | cut %lld scoredA %lld survB %lld\n", \
        b, s_clean, n, _cnt[3], _pt[0] * 0.01, _pt[1] * 0.01, _pt[2] * 0.01, _pt[3] * 0.01, _pt[4] * 0.01, _pt[5] * 0.01, _pt[6] * 0.01, (wall_clock64() - _t00) * 0.01, _cnt[0], _cnt[1], _cnt[2]); } while (0)
#else
#define RPROF(k) do {} while (0)
#define RPROF_OUT() do {} while (0)
#endif
    const SqStruct st = structs[b];
    if (st.nstrand < 0) return;
    const SqChain ch = cio.chain[b];
    const SqJob jb = c.jobs[st.job];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n;
    const SqRoundsLds Lo = sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, nthr);
    int16_t *const P = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_P);
    int16_t *const U = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_U);
    int16_t *const SU = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_SU);
    uint8_t *const E = reinterpret_cast<uint8_t *>(rd_dyn + Lo.off_E);
    uint8_t *const l_ci = reinterpret_cast<uint8_t *>(rd_dyn + Lo.off_ci);
    uint8_t *const l_code = reinterpret_cast<uint8_t *>(rd_dyn + Lo.off_code);
    uint32_t *const FG = reinterpret_cast<uint32_t *>(rd_dyn + Lo.off_fg);
    double *const s_cell = reinterpret_cast<double *>(rd_dyn + Lo.off_cell);
    SqStrand *const strbuf = reinterpret_cast<SqStrand *>(rd_dyn + Lo.off_str);
    int16_t *const sidxbuf = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_sidx);
    uint16_t *const s_skip = reinterpret_cast<uint16_t *>(rd_dyn + Lo.off_skip);
    char *const uni = rd_dyn + Lo.off_union;
    const int str_cap = ra.str_cap;

    auto retire = [&](int nstems, int by_count) {           // the records sq_chain_kernel writes (sq_chain.hip)
        if (tid == 0) {
            structs[b].nstrand = -1;
            const uint32_t idx = atomicAdd(cio.d_nfin, 1u);
            cio.h_fin[idx] = (unsigned long long)(uint32_t)st.job | ((unsigned long long)(uint32_t)nstems << 32) |
                             ((unsigned long long)(by_count ? 1 : 0) << 63);
            const uint32_t li = atomicAdd(&cio.fin_ctr[0], 1u);
            if (li < cio.fin_cap) cio.fin[li] = SqPoolFin{st.job, SQ_FIN_KIND_G0, 0, nstems, (uint32_t)ch.toff, SQ_FIN_SRC_CHAIN};
            else cio.fin_ctr[2] = 1;
            cio.job_evals[st.job] = (long long)nstems + (by_count ? 0 : 1);
        }
    };

    // ---- once per fold: letter classes, the cell table (sq_score_kernel builds them per round), the empty structure ----
    if (tid == 0) s_nlist = 0;
#ifdef SQ_ROUNDS_PROF
    if (tid == 0) s_clean = 0;
#endif
    const SqCellEnv cenv = sq_cell_setup(c, jb, ps, s_ctmp, l_ci, l_code, s_cell, tid, nthr);
    {
        const uint8_t *e0 = c.e0c + jb.pos_off;
        for (int p = tid; p < n; p += nthr) { P[p] = -1; E[p] = e0[p]; }
    }
    __syncthreads();

    // exclusive prefix counts of unpaired positions (U) and unpaired separators (SU) from P (sq_state_build's scan)
    auto prefix_counts = [&]() {
        int base_u = 0, base_s = 0;
        for (int p0 = 0; p0 < n; p0 += nthr) {
            const int p = p0 + tid;
            const bool un = p < n && P[p] == -1;
            const bool us = un && (l_code[p] == 26 || l_code[p] == 27);
            const unsigned long long mu = __ballot(un), ms = __ballot(us);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (lane == 0) { s_wave_u[wv] = __popcll(mu); s_wave_s[wv] = __popcll(ms); }
            __syncthreads();
            int pu = base_u + __popcll(mu & below), pS = base_s + __popcll(ms & below);
            for (int q = 0; q < wv; q++) { pu += s_wave_u[q]; pS += s_wave_s[q]; }
            if (p < n) { U[p] = (int16_t)pu; SU[p] = (int16_t)pS; }
            for (int q = 0; q < nwv; q++) { base_u += s_wave_u[q]; base_s += s_wave_s[q]; }
            __syncthreads();
        }
        if (tid == 0) { U[n] = (int16_t)base_u; SU[n] = (int16_t)base_s; }
        __syncthreads();
    };
    prefix_counts();
    RPROF(0);

    const int cap = jb.cand_cap;
    SqRun *const listA = reinterpret_cast<SqRun *>(a.cands + st.cand_off), *const listB = listA + cap;
    const int minlen = max(1, (int)ceil(ps->minlen));

    // ---- the first round's AnnotateStems: bit-diagonal scan of the empty structure into listB ----
    if (n >= 5) {                                                   // :456-457 (shorter sequences have no diagonals)
        const int fbh = Lo.fbh;
        for (int m2 = wv; 2 * m2 < fbh; m2 += nwv) {                // free-position words, forward and reversed (sq_state_build)
            const int pf = 64 * m2 + lane;
            const unsigned long long bf = __ballot(pf < n && E[pf] == 0);
            const int pr = n - 1 - (64 * m2 + lane - SQ_GPAD);
            const unsigned long long br = __ballot(pr >= 0 && pr < n && E[pr] == 0);
            if (lane == 0) {
                FG[2 * m2] = (uint32_t)bf; FG[fbh + 2 * m2] = (uint32_t)br;
                if (2 * m2 + 1 < fbh) { FG[2 * m2 + 1] = (uint32_t)(bf >> 32); FG[fbh + 2 * m2 + 1] = (uint32_t)(br >> 32); }
            }
        }
        uint32_t *const wcnt = reinterpret_cast<uint32_t *>(uni) + 4 * wv;
        uint2 *const wstage = reinterpret_cast<uint2 *>(uni + 16 * nwv) + (size_t)wv * SQ_ROUNDS_STAGE;
        if (lane == 0) *wcnt = 0;
        __syncthreads();
        SqRoundsSink sink{wstage, wcnt, &s_nlist, listB, (uint32_t)cap, a.ctr};
        sq_scan6_groups(c, jb, FG, FG + fbh, fbh, E, wv, nwv, lane, sink);
    }
    __syncthreads();
    uint32_t ncur = s_nlist < (uint32_t)cap ? s_nlist : (uint32_t)cap;
    __syncthreads();
    RPROF(1);

    // bpscore of a run and the sum of its cells' positive parts (sq_cellrun.h): a run whose positive parts miss :492 is
    // dead for good
    auto run_bps = [&](int i0, int j0, int L, double &pos) -> double { return sq_cellrun_bps(cenv, c, jb, i0, j0, L, pos); };

    const double minbps = ps->minbpscore, minfin = ps->minfinscore;
    const double ps_lb = ps->loopbonus, ps_bw = ps->bracketweight, ps_dc = ps->distcoef;
    const int ps_bwint = ps->bw_integral, ps_sdflen = ps->sdf_len;
    const double *const ps_sdf = c.sdftab + ps->sdf_off;
    const double *const ps_of = ps->oftab;
    const double ub_of = ps->ub_of, ub_lf = ra.bound ? ps->ub_lf : INFINITY;
    // a width-1 pool only ever uses ChooseStems' FIRST element (the highest finalscore, the smallest key among equals): a run
    // whose bound is below the best finalscore seen so far can neither be it nor tie with it -- the bar is the best itself, not
    // the suboptimality range below it (which the pools' round kernel needs: its children come from the whole range)
    const double st_subopt = 1.0;

    // ---- the first round's list: exact bpscores, dead runs dropped, ordered by descending bpscore (buckets of 0.5) so that
    // the scoring pass meets the strong candidates first and its bound prunes the rest; later rounds keep the order ----
    {
        uint32_t *const hist = reinterpret_cast<uint32_t *>(uni);             // [256] counts, then fill pointers
        uint32_t *const start = hist + 256;                                  // [256]
        for (int k = tid; k < 512; k += nthr) hist[k] = 0;
        __syncthreads();
        auto bucket = [&](double bps) -> int { const double x = bps * 2.0; return 255 - (x >= 255.0 ? 255 : (x > 0.0 ? (int)x : 0)); };
        for (uint32_t q = tid; q < ncur; q += nthr) {
            const SqRun r = listB[q];
            const int i = (int)(r.key & 0xFFFFu), j = (int)(r.key >> 16) - i;
            double pos;
            const double bps = run_bps(i, j, (int)r.len, pos);
            if (pos < minbps) listB[q].len = 0;
            else { listB[q].bps = bps; atomicAdd(&hist[bucket(bps)], 1u); }
        }
        __threadfence_block();
        __syncthreads();
        if (wv == 0) {                                                       // exclusive prefix over the 256 buckets
            uint32_t h[4], tot = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { h[k] = hist[4 * lane + k]; tot += h[k]; }
            uint32_t inc = tot;
            for (int off = 1; off < 64; off <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, off); if (lane >= off) inc += y; }
            uint32_t base = inc - tot;
#pragma unroll
            for (int k = 0; k < 4; k++) { start[4 * lane + k] = base; base += h[k]; }
            if (lane == 63) s_nlist = inc;
        }
        __syncthreads();
        for (int k = tid; k < 256; k += nthr) hist[k] = 0;
        __syncthreads();
        for (uint32_t q = tid; q < ncur; q += nthr) {
            const SqRun r = listB[q];
            if (r.len) {
                const int bk = bucket(r.bps);
                listA[start[bk] + atomicAdd(&hist[bk], 1u)] = r;
            }
        }
        if (tid == 0) s_ndead = 0;
        __threadfence_block();
        __syncthreads();
    }
    RPROF(2);

    uint32_t *const s_ring = reinterpret_cast<uint32_t *>(uni);    // list indices of the runs that wait for ScoreStems
    const uint32_t smask = (uint32_t)Lo.surv_cap - 1u;

    // the structure's stems with their crossing weights (:121-124) stay in LDS between rounds; the level rule's scratch
    // shares the survivor ring's region
    SqExtendLds XL;
    XL.cc = reinterpret_cast<int32_t *>(rd_dyn + Lo.off_stems);
    XL.i = reinterpret_cast<int16_t *>(XL.cc + Lo.t8); XL.j = XL.i + Lo.t8; XL.len = XL.j + Lo.t8;
    XL.gsize = reinterpret_cast<int32_t *>(XL.len + Lo.t8);      // (groups and their sizes stay too: a stem that crosses nothing joins group 0)
    XL.grp = reinterpret_cast<uint8_t *>(XL.gsize + 64);
    XL.ord = reinterpret_cast<int16_t *>(uni);
    XL.lvl = reinterpret_cast<uint8_t *>(XL.ord + Lo.t8); XL.rank = XL.lvl + Lo.t8;

    // the bound on a run's finalscore (sq_cellrun.h: exact tetraloop factor, loop bonuses only where they can apply)
    auto upper_of = [&](double bps, int i0, int j0, int L) -> double { return sq_run_upper(bps, i0, j0, L, U, l_code, n, ub_of, ub_lf, ps_lb); };

    int nstems = 0, nstrand = 0, ngroups = 0;       // (ngroups: level groups in use, first wave only)
    bool anycross = false;
    int za0 = 1, za1 = 0, zb0 = 1, zb1 = 0;                         // the two strands of the stem chosen last
    SqRun *list = listA, *other = listB;
    SqChainStem *const gst = cio.stems + ch.toff;

    for (int round = 0;; round++) {
#ifdef SQ_ROUNDS_PROF
        _cnt[3]++;
#endif
        // ---- dead entries out (order kept) when they are a third of the list or the list runs out of room ----
        uint32_t nl = s_nlist < (uint32_t)cap ? s_nlist : (uint32_t)cap;
        {
            const uint32_t nd = s_ndead;
            __syncthreads();
            if (nd * 3 > nl || nl + ((nl - nd) >> 1) + 64 > (uint32_t)cap) {
                uint32_t base = 0;
                for (uint32_t q0 = 0; q0 < nl; q0 += nthr) {
                    const uint32_t q = q0 + tid;
                    const SqRun r = q < nl ? list[q] : SqRun{0u, 0u, 0.0};
                    const unsigned long long m = __ballot(r.len != 0);
                    if (lane == 0) s_wave_u[wv] = __popcll(m);
                    __syncthreads();
                    uint32_t off = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    for (int w = 0; w < wv; w++) off += (uint32_t)s_wave_u[w];
                    if (r.len) other[off] = r;
                    for (int w = 0; w < nwv; w++) base += (uint32_t)s_wave_u[w];
                    __syncthreads();
                }
                SqRun *t = list; list = other; other = t;
                nl = base;
                if (tid == 0) { s_nlist = base; s_ndead = 0; }
                __threadfence_block();
                __syncthreads();
            }
        }
        RPROF(2);

        // ---- one pass over the list: the runs cut against the new stem's strands, in place (the first piece that is
        // still alive takes the run's entry, further pieces go to the end of the list and are met later in this pass),
        // then ScoreStems on the runs that pass :492; the best finalscore, the smallest key among equals.  The runs that
        // pass wait in a ring in LDS until a full group of them is due (the chain of dependent loads of ScoreStems is what
        // the pass waits for: idle lanes are its cost); the next chunk's entries are on their way meanwhile ----
        if (tid == 0) { s_nsurv = 0; s_best = 0ull; }
        const SqStrand *const S = strbuf;
        const SqStemsEnv env = {S, s_skip, nstrand, true, P, U, SU, l_code, n, false, nullptr, nullptr, nullptr, 0,
                                ps_lb, ps_bw, ps_dc, ps_bwint, ps_sdflen, ps_sdf, ps_of, a.ctr};
        __syncthreads();
        double bfin = 0.0, bbps = 0.0; uint32_t bkey = 0, blen = 0; int bany = 0;
        double sfin = -INFINITY;                                    // the best finalscore of the OTHER runs (ties of a pool that may branch)
        uint32_t head = 0;                                          // entries of the ring taken so far (s_nsurv: entries put)
        SqRun rr[SQ_ROUNDS_CHUNK];
#pragma unroll
        for (int u = 0; u < SQ_ROUNDS_CHUNK; u++) {
            const uint32_t q = (uint32_t)u * nthr + tid;
            rr[u] = q < nl ? list[q] : SqRun{0u, 0u, 0.0};
        }
        for (uint32_t q0 = 0; q0 < nl; q0 += SQ_ROUNDS_CHUNK * nthr) {
            double need = minfin;
            {
                const unsigned long long sb = s_best;
                if (sb) { const double r = st_subopt * sq_unord(sb); need = r > need ? r : need; }
            }
#pragma unroll
            for (int u = 0; u < SQ_ROUNDS_CHUNK; u++) {
                SqRun r = rr[u];
                int L = (int)r.len;
                if (L > 0 && round > 0) {
                    const int i = (int)(r.key & 0xFFFFu), s = (int)(r.key >> 16), j = s - i;
                    // cell t of the run: row i + t, column j - t; masked when either lies on a strand [za0, za1] or [zb0, zb1]
                    const int lo0 = za0 - i, hi0 = za1 - i, lo1 = zb0 - i, hi1 = zb1 - i, lo2 = j - za1, hi2 = j - za0, lo3 = j - zb1, hi3 = j - zb0;
                    if ((lo0 < L && hi0 >= 0) || (lo1 < L && hi1 >= 0) || (lo2 < L && hi2 >= 0) || (lo3 < L && hi3 >= 0)) {
                        const uint32_t q = q0 + (uint32_t)u * nthr + tid;
                        bool first = true;
                        int t0 = 0;
                        while (t0 < L) {
                            int pb = t0;
#pragma unroll
                            for (int rep = 0; rep < 4; rep++) {
                                if (pb >= lo0 && pb <= hi0) pb = hi0 + 1;
                                if (pb >= lo1 && pb <= hi1) pb = hi1 + 1;
                                if (pb >= lo2 && pb <= hi2) pb = hi2 + 1;
                                if (pb >= lo3 && pb <= hi3) pb = hi3 + 1;
                            }
                            if (pb >= L) break;
                            int pe = L;
                            if (lo0 > pb && lo0 < pe) pe = lo0;
                            if (lo1 > pb && lo1 < pe) pe = lo1;
                            if (lo2 > pb && lo2 < pe) pe = lo2;
                            if (lo3 > pb && lo3 < pe) pe = lo3;
                            t0 = pe;
                            const int plen = pe - pb;
                            if (plen < minlen) continue;
                            double pos;
                            const double bps = run_bps(i + pb, j - pb, plen, pos);
                            if (pos < minbps) continue;
                            const SqRun piece = {((uint32_t)s << 16) | (uint32_t)(i + pb), (uint32_t)plen, bps};
                            if (first) { first = false; r = piece; }
                            else {
                                const uint32_t at = atomicAdd(&s_nlist, 1u);
                                if (at < (uint32_t)cap) list[at] = piece;
                                else a.ctr->cand_ovf = 1;
                            }
                        }
                        if (first) { r.len = 0; atomicAdd(&s_ndead, 1u); }
                        list[q] = r;
                        L = (int)r.len;
                    }
                }
                bool ok = L > 0 && r.bps >= minbps;                             // :492
                // and the bound -- first with every factor at its maximum (ps->ub_lf; the same products in the same order: never
                // below the exact bound), which needs nothing but the run's bpscore: the list is ordered by bpscore, so once a
                // strong run has set `need` most of the list ends here without touching the prefix counts in LDS
                if (ok && ub_lf < INFINITY && r.bps >= 0) ok = !(((((r.bps * ub_of) * ub_lf) * 1.25) * (1.0 + 0x1p-30)) < need);
                if (ok) {
                    const int i0 = (int)(r.key & 0xFFFFu), j0 = (int)(r.key >> 16) - i0;
                    ok = !(upper_of(r.bps, i0, j0, L) < need);
                }
                const unsigned long long okm = __ballot(ok);
                if (okm) {
                    uint32_t base = 0;
                    const int leader = __ffsll((long long)okm) - 1;
                    if (lane == leader) base = atomicAdd(&s_nsurv, (uint32_t)__popcll(okm));
                    base = (uint32_t)__shfl((int)base, leader);
                    if (ok) {
                        const uint32_t pos = (base + (uint32_t)__popcll(okm & ((1ull << lane) - 1ull))) & smask;
                        s_ring[pos] = q0 + (uint32_t)u * nthr + tid;
                    }
                }
            }
            __threadfence_block();
            __syncthreads();
            RPROF(3);
            nl = s_nlist < (uint32_t)cap ? s_nlist : (uint32_t)cap;             // (pieces appended by this chunk are met later)
#pragma unroll
            for (int u = 0; u < SQ_ROUNDS_CHUNK; u++) {                         // the next chunk's entries: in flight during ScoreStems
                const uint32_t q = q0 + (uint32_t)(SQ_ROUNDS_CHUNK + u) * nthr + tid;
                rr[u] = q < nl ? list[q] : SqRun{0u, 0u, 0.0};
            }
            const uint32_t tailp = s_nsurv;
            const bool last = q0 + SQ_ROUNDS_CHUNK * nthr >= nl;
            while (tailp - head >= (uint32_t)nthr || (last && head != tailp)) {
                const uint32_t idx = head + tid;
                const bool have = idx - head < tailp - head;
#ifdef SQ_ROUNDS_PROF
                if (tid == 0) _cnt[2] += min(tailp - head, (uint32_t)nthr);
#endif
                head += min(tailp - head, (uint32_t)nthr);
                const SqRun rb = have ? list[s_ring[idx & smask]] : SqRun{0u, 0u, 0.0};
                const uint32_t key = rb.key;
                const int L = (int)rb.len;
                const double bps = rb.bps;
                const int s = (int)(key >> 16), i0 = (int)(key & 0xFFFFu), j0 = s - i0;
                bool ok = have;
                if (ok) {
                    const unsigned long long sbst = s_best;
                    if (sbst && upper_of(bps, i0, j0, L) < st_subopt * sq_unord(sbst)) ok = false;
                }
                double fin = 0.0;
#ifdef SQ_ROUNDS_PROF
                {   // (what keeping finalscores between rounds would save: evaluations the last stem cannot have changed)
                    const bool clean = ok && round > 0 && s_cross == 0 && (za1 < i0 - 6 || za0 > j0 + 6) && (zb1 < i0 - 6 || zb0 > j0 + 6);
                    const unsigned long long cm = __ballot(clean);
                    if (tid == 0) _cnt[1] += 0;
                    if (lane == 0) atomicAdd(&s_clean, (uint32_t)__popcll(cm));
                }
#endif
                if (ok) {
                    fin = sq_stem_finalscore(env, i0, j0, L, bps);
                    ok = fin >= minfin;                                     // :751
                }
                if (ok) {
                    if (!bany || fin > bfin || (fin == bfin && key < bkey)) { if (bany && bfin > sfin) sfin = bfin; bany = 1; bfin = fin; bkey = key; blen = (uint32_t)L; bbps = bps; }
                    else if (fin > sfin) sfin = fin;
                }
                if (__ballot(ok) != 0ull) {
                    const double wb = sq_wave_max_f64(ok ? fin : -INFINITY);
                    if (lane == 0) atomicMax(&s_best, sq_ord(wb));
                }
            }
            __syncthreads();                                                    // (the ring's taken entries may be overwritten now)
            RPROF(4);
        }
#ifdef SQ_ROUNDS_PROF
        _cnt[0] += nl;
#endif
        // ---- ChooseStems' first element over the block ----
        {
            // the wave's best: the highest finalscore, the smallest key among equals (one lane: keys are distinct); the best of
            // all the OTHER runs beside it (DPP reductions: the butterfly of six values they replace was 54 trips over the LDS
            // crossbar per round)
            const double wmax = sq_wave_max_f64(bany ? bfin : -INFINITY);
            const bool cand = bany && bfin == wmax;
            const int kmin = sq_wave_min_i32(cand ? (int)bkey : 0x7fffffff);       // (keys < 2^30: SQ_ROUNDS_MAXN)
            const bool win = cand && (int)bkey == kmin;
            const unsigned long long wm = __ballot(win);
            const double sec = sq_wave_max_f64(bany && !win && bfin > sfin ? bfin : sfin);
            if (wm) {
                const int wl = __ffsll((long long)wm) - 1;
                blen = (uint32_t)__builtin_amdgcn_readlane((int)blen, wl);
                bbps = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bbps), wl), __builtin_amdgcn_readlane(__double2loint(bbps), wl));
                bany = 1; bfin = wmax; bkey = (uint32_t)kmin;
            } else bany = 0;
            sfin = sec;
        }
        if (lane == 0) { s_wany[wv] = bany; s_wfin[wv] = bfin; s_wbps[wv] = bbps; s_wkey[wv] = bkey; s_wlen[wv] = blen; s_wsec[wv] = sfin; }
        __syncthreads();
        bany = 0; sfin = -INFINITY;
        for (int q = 0; q < nwv; q++) {
            if (s_wsec[q] > sfin) sfin = s_wsec[q];
            if (s_wany[q] && (!bany || s_wfin[q] > bfin || (s_wfin[q] == bfin && s_wkey[q] < bkey))) {
                if (bany && bfin > sfin) sfin = bfin;
                bany = 1; bfin = s_wfin[q]; bkey = s_wkey[q]; blen = s_wlen[q]; bbps = s_wbps[q];
            } else if (s_wany[q] && s_wfin[q] > sfin) sfin = s_wfin[q];
        }
        RPROF(5);
        if (!bany) { retire(nstems, 0); RPROF_OUT(); return; }      // :1192-1193 no new stem: the structure is final
        if (ra.ties && sfin == bfin) {
            // a pool that may branch: ChooseStems returns every run that reaches the best finalscore and shares a base with the
            // ones taken (:769-789, range factor 1.0) -- two runs at the top mean the pool MAY grow here: the structure stops,
            // unfinished, and the device pools fold its job (rare: these jobs weigh their cells with a dense fp64 matrix)
            if (tid == 0) {
                structs[b].nstrand = -1;
                const uint32_t idx = atomicAdd(cio.d_nfin, 1u);
                cio.h_fin[idx] = (unsigned long long)(uint32_t)st.job | (1ull << 62);
            }
            return;
        }
        const int i0 = (int)(bkey & 0xFFFFu), j0 = (int)(bkey >> 16) - i0, len = (int)blen;
        const int k = nstems;
        if (k >= ch.tcap) { if (tid == 0) a.ctr->out_ovf = 1; retire(k, 0); return; }
        __syncthreads();                                            // (every thread has read the wave bests)
        // ---- the child.  First wave: crossing weights, levels when stems cross, the sorted strand list (sq_extend.h);
        // the other waves: partner array and prefix counts -- positions p and above lose the new pairs below p, and
        // separators never pair, so SU stays ----
        const int zb = j0 - len + 1;
        if (wv == 0) {
            int mycc = 0, mycross = 0;
            for (int q = lane; q < k; q += 64)
                if (sq_chain_cross(XL.i[q], XL.j[q], i0, j0)) { XL.cc[q] += len; mycc += XL.len[q]; mycross = 1; }
            const int newcc = sq_wave_sum32(mycc);
            const bool ac = anycross || __ballot(mycross) != 0ull;
            if (lane == 0) {
                XL.i[k] = (int16_t)i0; XL.j[k] = (int16_t)j0; XL.len[k] = (int16_t)len; XL.cc[k] = newcc;
                gst[k] = SqChainStem{i0, j0, len, newcc};           // (the tail reads the stems there; the weights of the others stay in LDS)
                cio.h_stems[ch.toff + k] = SqStemOut{i0, j0, len, 0, bbps, bfin};
                s_cross = ac ? 1 : 0;
            }
            sq_wave_lds_fence();
            if (ac) {
                // levels: the full rule when the new stem crosses something (crossing weights changed: the order of the first
                // fit may have); a stem without crossings joins group 0 and only the groups' ranking is taken anew
                if (anycross && __ballot(mycross) == 0ull && ngroups > 0) sq_stem_levels_join(XL, k + 1, ngroups, len, lane);
                else ngroups = sq_stem_levels_wave(XL, k + 1, lane, &a.ctr->level_ovf);
            }
            sq_rounds_insert_strands(XL, ac, k, strbuf, sidxbuf, nstrand, i0, j0, len, lane);
        }
        if (nwv == 1 || wv > 0) {
            const int w0 = nwv == 1 ? 0 : wv - 1, wn = nwv == 1 ? 1 : nwv - 1;
            for (int t = w0 * 64 + lane; t < len; t += wn * 64) {   // :634-635
                P[i0 + t] = (int16_t)(j0 - t);
                P[j0 - t] = (int16_t)(i0 + t);
            }
            for (int p = i0 + 1 + w0 * 64 + lane; p <= n; p += wn * 64)
                U[p] = (int16_t)((int)U[p] - (min(p - i0, len) + min(max(p - zb, 0), len)));
        }
        __syncthreads();
        anycross = s_cross != 0;
        nstems = k + 1; nstrand += 2;
        if ((double)nstems == ch.maxstems) { retire(nstems, 1); return; }   // :1168-1174 (checked before the next evaluation)
        {
            // skip pointers over the blocks ScoreStems' sweep registers (sq_score_kernel)
            const SqStrand *const S2 = strbuf;
            const int16_t *const X2 = sidxbuf;
            for (int q = tid; q < nstrand; q += nthr) {
                const SqStrand x = S2[q];
                int z = q + 1;
                if (x.left) {
                    const int pf = x.pstart;
                    if (XL.cc[X2[q]] == 0) {
                        // a stem that crosses nothing: no 5' strand inside its block reaches beyond it, so the pointer is the
                        // first strand that starts behind the partner -- a binary search instead of a walk over the block
                        int lo = q + 1, hi = nstrand;
                        while (lo < hi) { const int mid = (lo + hi) >> 1; if (S2[mid].start > pf) hi = mid; else lo = mid + 1; }
                        z = lo;
                    } else {
                        while (z < nstrand) {
                            const SqStrand y = S2[z];
                            if (y.start > pf || (y.left && y.pstart > pf)) break;
                            z++;
                        }
                    }
                }
                s_skip[q] = (uint16_t)z;
            }
        }
        za0 = i0; za1 = i0 + len - 1; zb0 = zb; zb1 = j0;
        __syncthreads();
        RPROF(6);
    }
}
