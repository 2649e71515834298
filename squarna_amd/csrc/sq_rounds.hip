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
// take (sequences beyond SQ_ROUNDS_MAXN, lists that outgrow the LDS) keeps the launched form.
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

// the first round's scan: a wave reserves the places of a word-row's runs in the block's list with ONE LDS atomic and the lanes
// write their runs there (sq_scan.h; until round 5 every run went through a per-wave staging buffer behind an atomic of its own)
struct SqRoundsSink {
    uint32_t *nlist;                         // the block's list length (LDS)
    SqRun *list; uint32_t cap; SqCounters *ctr;
    __device__ __forceinline__ uint32_t reserve(uint32_t total, int lane)
    {
        uint32_t b0 = 0;
        if (lane == 0) b0 = atomicAdd(nlist, total);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
    }
    __device__ __forceinline__ void put(uint32_t pos, uint32_t key, uint32_t len)
    {
        if (pos < cap) list[pos] = SqRun{key, len, __longlong_as_double(0x7FF8000000000000ll)};
        else ctr->cand_ovf = 1;
    }
    __device__ __forceinline__ void poll(int) {}
    __device__ __forceinline__ void drain(int) {}
};

// The two strands of a new stem k = (i0, j0, len) into the sorted strand list S[0 .. nstrand) (+ the stem index of each
// strand, X, and its skip pointer, K), IN PLACE, by one wave: every strand moves up by the number of new strands that start
// before it (0, 1 or 2), the chunks of 64 taken from the top so that nothing is overwritten before it is read; levels from
// L.lvl when stems cross.
//
// Skip pointers (ScoreStems' walk jumps over a block it has registered): K[q] of a 5' strand q with partner end pf is the first
// strand behind q that starts beyond pf or is a 5' strand whose partner lies beyond pf; q + 1 for a 3' strand.  Until round 6 the
// first wave took them anew every round -- a lane per strand walking its block, as long as the longest block: most of the 340 us
// the extension took per round on structures of 700 strands.  They are MAINTAINED now: the old strands keep their starts and
// partners, so an old pointer still names the first old strand that ends the block; it moves up with that strand, and only a
// new strand that lands inside the block and ends it -- the 5' strand when the new stem's partner end lies beyond pf, the 3'
// strand when it starts beyond pf -- comes in front of it.  The new 5' strand's own pointer: a walk that jumps along the
// others' pointers (a strand that does not end its block ends nothing inside its own either).
__device__ __forceinline__ void sq_rounds_insert_strands(SqExtendLds &L, bool anycross, int k, SqStrand *S, int16_t *X, uint16_t *K, int nstrand,
                                                         int i0, int j0, int len, int lane)
{
    const int ls = i0, rs = j0 - len + 1;                               // starts of the 5' and the 3' strand (ls < rs)
    int below_l = 0, below_r = 0;
    for (int q0 = 0; q0 < nstrand; q0 += 64) {
        const int q = q0 + lane;
        const int st = q < nstrand ? S[q].start : 0x7fff;
        below_l += __popcll(__ballot(st < ls)); below_r += __popcll(__ballot(st < rs));
    }
    const int ia = below_l, ib = below_r + 1;                           // where the new strands go
    for (int q0 = ((nstrand + 63) & ~63) - 64; q0 >= 0; q0 -= 64) {
        const int q = q0 + lane;
        const bool valid = q < nstrand;
        SqStrand x = valid ? S[q] : SqStrand{0, 0, 0, 0, 0};
        const int sx = valid ? X[q] : 0;
        const int zo = valid ? (int)K[q] : 0;
        sq_wave_lds_fence();                                            // (the whole chunk is read before any of it moves)
        if (valid) {
            if (anycross) x.level = L.lvl[sx];
            const int at = q + (q >= below_l ? 1 : 0) + (q >= below_r ? 1 : 0);
            int z = at + 1;
            if (x.left) {
                const int pf = x.pstart;
                z = zo + (zo >= below_l ? 1 : 0) + (zo >= below_r ? 1 : 0);     // (old strands [0, below) start in front of the new one)
                if (ib > at && ib < z && rs > pf) z = ib;
                if (ia > at && ia < z && j0 > pf) z = ia;
            }
            S[at] = x; X[at] = (int16_t)sx; K[at] = (uint16_t)z;
        }
        sq_wave_lds_fence();
    }
    if (lane == 0) {
        const uint8_t lv = anycross ? L.lvl[k] : (uint8_t)1;
        S[ia] = SqStrand{(int16_t)ls, (int16_t)len, (int16_t)j0, lv, 1};
        S[ib] = SqStrand{(int16_t)rs, (int16_t)len, (int16_t)(i0 + len - 1), lv, 0};
        X[ia] = (int16_t)k; X[ib] = (int16_t)k;
        K[ib] = (uint16_t)(ib + 1);
    }
    sq_wave_lds_fence();
    // the new 5' strand's pointer
    {
        const int ns2 = nstrand + 2;
        int z = ia + 1;
        while (z < ns2) {
            const SqStrand y = S[z];
            if (y.start > j0 || (y.left && y.pstart > j0)) break;
            z = y.left ? (int)K[z] : z + 1;
        }
        if (lane == 0) K[ia] = (uint16_t)z;
    }
}

extern "C" __global__ __launch_bounds__(SQ_ROUNDS_THREADS) __attribute__((amdgpu_waves_per_eu(SQ_ROUNDS_WAVES))) void sq_rounds_kernel(SqDevCtx c, SqStruct *structs, SqScanArgs a, SqChainIO cio,
                                                                                 SqRoundsArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) char rd_dyn[];
    __shared__ int s_wave_u[SQ_ROUNDS_THREADS / 64], s_wave_s[SQ_ROUNDS_THREADS / 64];
    __shared__ uint32_t s_nlist, s_ndead;
    __shared__ int s_regroup, s_ready;
    __shared__ uint32_t s_unit, s_present;
    __shared__ SqCellTmp s_ctmp;
    __shared__ unsigned long long s_best;
    __shared__ double s_wfin[SQ_ROUNDS_THREADS / 64], s_wbps[SQ_ROUNDS_THREADS / 64], s_wsec[SQ_ROUNDS_THREADS / 64];
    __shared__ uint32_t s_wkey[SQ_ROUNDS_THREADS / 64], s_wlen[SQ_ROUNDS_THREADS / 64];
    __shared__ int s_wany[SQ_ROUNDS_THREADS / 64];
    __shared__ int s_cross, s_ofmono;

    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    const int b = blockIdx.x;
#ifdef SQ_ROUNDS_PROF
    // in-kernel timers of the block's first wave (100 MHz wall clock): set-up, scan, ordering + compaction, the pass's stream / cut /
    // bound / score steps, pick, extension + state
    long long _pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long _t = wall_clock64(); const long long _t00 = _t;
    long long _cnt[4] = {0, 0, 0, 0};       // entries streamed, finalscores taken from the list, runs scored (first wave), rounds
    long long _xl[4] = {0, 0, 0, 0};        // the full level rule: order, first fit, ranks; crossing stems placed
    long long _xt[4] = {0, 0, 0, 0};        // the extension: levels, full rule / joins; strands + skip pointers; [3] rounds with the full rule
    long long _sp[6] = {0, 0, 0, 0, 0, 0};  // the score step: [0] wait for the structure, [1] entry loads, [3] the rest; [4] steps
#define SPROF(k) do { const long long _n = wall_clock64(); _sp[k] += _n - _t2; _t2 = _n; } while (0)
#define RPROF(k) do { const long long _n = wall_clock64(); _pt[k] += _n - _t; _t = _n; } while (0)
#define RPROF_OUT() do { if ((tid == 0 || tid == 64) && (b % 97) == 0) printf("rounds block %d wave %d extension us: levels %.1f (%lld rounds with the full rule: order %.1f first fit %.1f of %lld stems, ranks %.1f) strands + skip pointers %.1f\n", b, wv, _xt[0] * 0.01, _xt[3], _xl[0] * 0.01, _xl[1] * 0.01, _xl[3], _xl[2] * 0.01, _xt[1] * 0.01); if ((tid == 0 || tid == 64) && (b % 97) == 0) printf("rounds block %d wave %d score steps %lld | us: wait for the structure %.1f entry loads %.1f ScoreStems + stores + pick %.1f\n", b, wv, _sp[4], _sp[0] * 0.01, _sp[1] * 0.01, _sp[3] * 0.01); if ((tid == 0 || tid == 64) && (b % 97) == 0) printf("rounds block %d wave %d n=%d rounds %lld | us: setup %.1f scan %.1f order %.1f stream %.1f cut %.1f bound %.1f score %.1f pick %.1f ext %.1f between %.1f total %.1f | wave 0: streamed %lld kept %lld scored %lld\n", \
        b, wv, n, _cnt[3], _pt[0] * 0.01, _pt[1] * 0.01, _pt[2] * 0.01, _pt[3] * 0.01, _pt[4] * 0.01, _pt[5] * 0.01, _pt[6] * 0.01, _pt[7] * 0.01, _pt[8] * 0.01, _pt[9] * 0.01, (wall_clock64() - _t00) * 0.01, _cnt[0], _cnt[1], _cnt[2]); } while (0)
#else
#define RPROF(k) do {} while (0)
#define RPROF_OUT() do {} while (0)
#define SPROF(k) do {} while (0)
#endif
    const SqStruct st = structs[b];
    if (st.nstrand < 0) return;
    const SqChain ch = cio.chain[b];
    const SqJob jb = c.jobs[st.job];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n;
    const SqRoundsLds Lo = sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, nthr, ra.su);
    int16_t *const P = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_P);
    int16_t *const U = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_U);
    int16_t *const SU = ra.su ? reinterpret_cast<int16_t *>(rd_dyn + Lo.off_SU) : nullptr;   // (no separator in the launch: nothing lies between chains)
    uint8_t *const E = reinterpret_cast<uint8_t *>(rd_dyn + Lo.off_E);
    uint8_t *const l_ci = reinterpret_cast<uint8_t *>(rd_dyn + Lo.off_ci);
    uint8_t *const l_code = reinterpret_cast<uint8_t *>(rd_dyn + Lo.off_code);
    uint32_t *const FG = reinterpret_cast<uint32_t *>(rd_dyn + Lo.off_fg);
    double *const s_cell = reinterpret_cast<double *>(rd_dyn + Lo.off_cell);
    SqStrand *const strbuf = reinterpret_cast<SqStrand *>(rd_dyn + Lo.off_str);
    int16_t *const sidxbuf = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_sidx);
    uint16_t *const s_skip = reinterpret_cast<uint16_t *>(rd_dyn + Lo.off_skip);
    char *const uni = rd_dyn + Lo.off_union;

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
            if (p < n) { U[p] = (int16_t)pu; if (SU) SU[p] = (int16_t)pS; }
            for (int q = 0; q < nwv; q++) { base_u += s_wave_u[q]; base_s += s_wave_s[q]; }
            __syncthreads();
        }
        if (tid == 0) { U[n] = (int16_t)base_u; if (SU) SU[n] = (int16_t)base_s; }
        __syncthreads();
    };
    prefix_counts();
    RPROF(0);

    // The structure's slice of the candidate arena (cand_cap 32-byte units): the list of the rounds as two arrays of `cap`
    // 16-byte records -- the runs (LA, lower half) and their kept bounds / finalscores (LB, upper half; valid as LA's flags say).
    // The first round's scan stages its raw runs in the upper half.
    const int cap = jb.cand_cap;
    SqRunA *const LA = reinterpret_cast<SqRunA *>(a.cands + st.cand_off);
    SqRunB *const LB = reinterpret_cast<SqRunB *>(LA + cap);
    SqRun *const raw = reinterpret_cast<SqRun *>(LA + cap);
    const int minlen = max(1, (int)ceil(ps->minlen));

    // ---- the first round's AnnotateStems: bit-diagonal scan of the empty structure into the staging half ----
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
        __syncthreads();
        SqRoundsSink sink{&s_nlist, raw, (uint32_t)cap, a.ctr};
        if (ra.fly > 0) {
            // the letter masks of sq_bits_masks_kernel, in the LDS the strands and stems of the later rounds will take (the
            // structure is empty now)
            char *const mreg = rd_dyn + Lo.off_str;
            const int npad = (n + 3) & ~3, nw = (n + 31) >> 5, maxl = ra.fly;
            uint8_t *const m_ccode = reinterpret_cast<uint8_t *>(mreg), *const m_rcode = m_ccode + npad, *const m_inc = m_rcode + npad;
            uint32_t *const m_M = reinterpret_cast<uint32_t *>(m_inc + npad), *const m_R = m_M + maxl * fbh;
            uint32_t mine = 0;
            if (tid == 0) s_present = 0u;
            __syncthreads();
            for (int p = tid; p < n; p += nthr) {
                const uint32_t code = c.codes[jb.pos_off + p], fl = c.flags[jb.pos_off + p];
                m_ccode[p] = (uint8_t)((!(fl & 1u) && !(fl & 2u)) ? code : 31u);          // :302, :303 (column side)
                const bool rowok = !(fl & 1u) && !(fl & 4u);                                // :302, :304 (row side)
                m_rcode[p] = (uint8_t)(rowok ? code : 31u);
                m_inc[p] = c.inc4[jb.pos_off + p];
                if (rowok && code < 29u && ps->pmask[code]) mine |= 1u << code;          // (letters that pair with nothing have no cells)
            }
            if (mine) atomicOr(&s_present, mine);
            __syncthreads();
            const uint32_t present = s_present;
            const int nlet = __popc(present);                                              // <= maxl (host: letters of the batch)
            auto letter = [&](int k) -> uint32_t { uint32_t m = present; for (int t = 0; t < k; t++) m &= m - 1; return (uint32_t)(__ffs((int)m) - 1); };
            for (int e = tid; e < nlet * fbh; e += nthr) {                                  // column masks over the reversed positions (G's layout)
                const int k = e / fbh, q = e - k * fbh;
                const uint32_t pm = ps->pmask[letter(k)];
                uint32_t word = 0;
                for (int bb = 0; bb < 32; bb++) {
                    const int pos = n - 1 - (32 * q + bb - SQ_GPAD);
                    if (pos >= 0 && pos < n) word |= ((pm >> m_ccode[pos]) & 1u) << bb;
                }
                m_M[e] = word;
            }
            for (int e = tid; e < nw * nlet; e += nthr) {                                   // row masks
                const int w = e / nlet, k = e - w * nlet;
                const uint32_t x = letter(k);
                uint32_t word = 0;
                for (int bb = 0; bb < 32 && 32 * w + bb < n; bb++) word |= (uint32_t)(m_rcode[32 * w + bb] == x) << bb;
                m_R[w * maxl + k] = word;
            }
            __syncthreads();
            SqBitsFly fly;
            fly.Mr = m_M; fly.R = m_R; fly.inc = m_inc; fly.fbh = fbh; fly.nlet = nlet; fly.maxl = maxl; fly.n = n;
            sq_scan6_groups(c, jb, FG, FG + fbh, fbh, E, wv, nwv, lane, sink, fly);
        } else
            sq_scan6_groups(c, jb, FG, FG + fbh, fbh, E, wv, nwv, lane, sink, SqBitsGlobal{c.bits + jb.bits_off, jb.bpitch});
    }
    __syncthreads();
    const uint32_t ncur = s_nlist < (uint32_t)cap ? s_nlist : (uint32_t)cap;
    __syncthreads();
    RPROF(1);

    // bpscore of a run and the sum of its cells' positive parts (sq_cellrun.h): a run whose positive parts miss :492 is
    // dead for good
    auto run_bps = [&](int i0, int j0, int L, double &pos) -> double { return sq_cellrun_bps(cenv, c, jb, i0, j0, L, pos); };

    const double minbps = ps->minbpscore, minfin = ps->minfinscore;
    const double ps_lb = ps->loopbonus, ps_bw = ps->bracketweight, ps_dc = ps->distcoef;
    const int ps_bwint = ps->bw_integral, ps_sdflen = ps->sdf_len;
    const double *const ps_sdf = c.sdftab + ps->sdf_off;
    // (the two tables ScoreStems reads last, at the end of its chain of dependent loads: in LDS)
    double *const l_sdf = reinterpret_cast<double *>(rd_dyn + Lo.off_tab), *const l_of = l_sdf + SQ_ROUNDS_SDF_LDS;
    const int l_sdflen = ps_sdflen < SQ_ROUNDS_SDF_LDS ? ps_sdflen : SQ_ROUNDS_SDF_LDS;
    for (int k = tid; k < l_sdflen; k += nthr) l_sdf[k] = ps_sdf[k];
    const double ub_of = ps->ub_of, ub_lf = ra.bound ? ps->ub_lf : INFINITY;
    // (l_ofr[p] >= of[p] / of_max, rounded up: what the order factor's share of a run's bound shrinks to once p levels are known)
    double *const l_ofr = l_of + SQ_MAXLEVELS + 2;
    for (int k = tid; k <= SQ_MAXLEVELS; k += nthr) { const double o = ps->oftab[k]; l_of[k] = o; l_ofr[k] = (o / ub_of) * (1.0 + 0x1p-40); }
    if (wv == 0) {
        // the order factors do not grow with the number of levels (orderpenalty >= 0): the early end of ScoreStems' walk relies on it
        const bool up = lane < SQ_MAXLEVELS && ps->oftab[lane + 1] > ps->oftab[lane];
        const unsigned long long bad = __ballot(up);
        if (lane == 0) s_ofmono = (bad == 0ull && ub_of > 0.0 && ub_lf < INFINITY && !ra.no_early) ? 1 : 0;
    }
    const double *const ps_of = l_of;
    // (a width-1 pool only ever uses ChooseStems' FIRST element -- the highest finalscore, the smallest key among equals: a run
    // whose bound is below the best finalscore seen so far can neither be it nor tie with it.  The bar is the best itself, not
    // the suboptimality range below it, which the pools' round kernel needs: its children come from the whole range)

    // ---- the first round's list: exact bpscores, dead runs dropped, ordered by descending bpscore (buckets of 0.5) so that
    // the pass meets the strong candidates first and its bound prunes the rest; later rounds keep the order roughly ----
    {
        uint32_t *const hist = reinterpret_cast<uint32_t *>(uni);             // [256] counts, then fill pointers
        uint32_t *const start = hist + 256;                                  // [256]
        for (int k = tid; k < 512; k += nthr) hist[k] = 0;
        __syncthreads();
        auto bucket = [&](double bps) -> int { const double x = bps * 2.0; return 255 - (x >= 255.0 ? 255 : (x > 0.0 ? (int)x : 0)); };
        for (uint32_t q = tid; q < ncur; q += nthr) {
            const SqRun r = raw[q];
            const int i = (int)(r.key & 0xFFFFu), j = (int)(r.key >> 16) - i;
            double pos;
            const double bps = run_bps(i, j, (int)r.len, pos);
            if (pos < minbps) raw[q].len = 0;
            else { raw[q].bps = bps; atomicAdd(&hist[bucket(bps)], 1u); }
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
            const SqRun r = raw[q];
            if (r.len) {
                const int bk = bucket(r.bps);
                LA[start[bk] + atomicAdd(&hist[bk], 1u)] = SqRunA{r.key, r.len, r.bps};
            }
        }
        if (tid == 0) { s_ndead = 0; s_best = 0ull; s_regroup = 0; s_ready = 0; s_unit = 0u; }
        __threadfence_block();
        __syncthreads();
    }
    RPROF(2);

    // the structure's stems with their crossing weights (:121-124) stay in LDS between rounds
    SqExtendLds XL;
    XL.cc = reinterpret_cast<int32_t *>(rd_dyn + Lo.off_stems);
    XL.i = reinterpret_cast<int16_t *>(XL.cc + Lo.t8); XL.j = XL.i + Lo.t8; XL.len = XL.j + Lo.t8;
    XL.gsize = reinterpret_cast<int32_t *>(XL.len + Lo.t8);      // (groups and their sizes stay too: a stem that crosses nothing joins group 0)
    XL.grp = reinterpret_cast<uint8_t *>(XL.gsize + 64);
    XL.ord = reinterpret_cast<int16_t *>(rd_dyn + Lo.off_lvl);
    XL.lvl = reinterpret_cast<uint8_t *>(XL.ord + Lo.t8); XL.rank = XL.lvl + Lo.t8;

    // the bound on a run's finalscore (sq_cellrun.h: exact tetraloop factor, loop bonuses only where they can apply)
    auto upper_of = [&](double bps, int i0, int j0, int L) -> double { return sq_run_upper(bps, i0, j0, L, U, l_code, n, ub_of, ub_lf, ps_lb); };

    int nstems = 0, nstrand = 0, ngroups = 0;       // (ngroups: level groups in use, first wave only)
    bool anycross = false;
    int za0 = 0x7000, za1 = -0x7000, zb0 = 0x7000, zb1 = -0x7000;   // the two strands of the stem chosen last (none yet: empty intervals)
    SqChainStem *const gst = cio.stems + ch.toff;
    // this wave's work queues of the list pass (list indices; the counts live in registers: nobody else touches them)
    uint32_t *const qcut = reinterpret_cast<uint32_t *>(uni) + (size_t)wv * SQ_RQ_WORDS, *const qcand = qcut + SQ_RQ_CAP, *const qsurv = qcand + SQ_RQ_CAP;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t nl0 = s_nlist;                                          // entries of the list when the pass starts (the same for every wave)
    int roundno = 0;
    bool compact = false;                                           // dead entries leave the list before this round's pass

    for (;;) {
#ifdef SQ_ROUNDS_PROF
        _cnt[3]++;
#endif
        // ---- dead entries out, in place, order kept, when they are a third of the list or the list runs out of room ----
        if (compact) {
            __syncthreads();                                           // (the first wave is through with the structure: every wave takes part)
            uint32_t base = 0;
            for (uint32_t q0 = 0; q0 < nl0; q0 += nthr) {
                const uint32_t q = q0 + tid;
                SqRunA ra_ = {0u, 0u, 0.0}; SqRunB rb_ = {0.0, 0.0};
                if (q < nl0) { ra_ = LA[q]; rb_ = LB[q]; }
                const bool live = (ra_.lf & SQ_RX_LEN) != 0;
                const unsigned long long m = __ballot(live);
                if (lane == 0) s_wave_u[wv] = __popcll(m);
                __syncthreads();                                       // (the chunk is read: its entries and the ones below may be overwritten)
                uint32_t off = base + (uint32_t)__popcll(m & lt_mask);
                for (int w = 0; w < wv; w++) off += (uint32_t)s_wave_u[w];
                if (live && off != q) { LA[off] = ra_; LB[off] = rb_; }
                for (int w = 0; w < nwv; w++) base += (uint32_t)s_wave_u[w];
                __threadfence_block();
                __syncthreads();
            }
            nl0 = base;
            if (tid == 0) { s_nlist = base; s_ndead = 0; }
            __syncthreads();
        }
        RPROF(2);

        // ---- one pass over the list.  The waves take the entries in units of 64 (two at a time, from a counter in LDS) and
        // sort them into three work queues of their own; a queue is served when it holds a full wave of work, so the rare steps
        // run with all lanes busy (until round 5 they ran inside the stream, a handful of lanes at a time, and were most of a
        // round):
        //   * cut      the run meets a strand of the new stem: its pieces, outer -> inner; the first live piece takes the
        //              entry, the others go to the end of the list (and are handled here, by the lane that made them);
        //   * bound    no valid bound: sq_run_upper, kept in the entry;
        //   * score    the bound reaches the best finalscore so far: ScoreStems, kept in the entry.
        // An entry whose finalscore is still valid costs a compare.  No block barrier inside the pass: the waves only share the
        // unit counter, the best finalscore so far (an atomic maximum in LDS) and the end of the list.  The stream is
        // straight-line code: every test is evaluated for every entry as integer arithmetic and combined without branches.
        // The first wave joins late: it is still putting the last stem into the structure (levels, strands, skip pointers), which
        // only ScoreStems reads -- the score step waits for s_ready.
        const SqStrand *const S = strbuf;
        const SqStemsEnv env = {S, s_skip, nstrand, true, P, U, SU, l_code, n, false, nullptr, nullptr, nullptr, 0,
                                ps_lb, ps_bw, ps_dc, ps_bwint, ps_sdflen, ps_sdf, ps_of, a.ctr, l_sdf, l_sdflen};
        const int rgmask = s_regroup != 0 ? 0 : -1;                // (0: the last stem changed the level groups -- no finalscore is kept)
        double bfin = -INFINITY, bbps = 0.0; uint32_t bkey = 0xFFFFFFFFu, blen = 0;   // (bfin == -inf: none yet; real finalscores are finite)
        double sfin = -INFINITY;                                    // the best finalscore of the OTHER runs (ties of a pool that may branch)
        auto take = [&](bool valid, double fin, uint32_t key, uint32_t L, double bps) {   // branch-free
            const bool better = valid & ((fin > bfin) | ((fin == bfin) & (key < bkey)));
            const double demoted = better ? bfin : (valid ? fin : -INFINITY);
            sfin = demoted > sfin ? demoted : sfin;
            bfin = better ? fin : bfin; bkey = better ? key : bkey; blen = better ? L : blen; bbps = better ? bps : bbps;
        };
        auto raise = [&](bool mine, double fin) {                   // the block's best so far (wave-uniform call)
            if (__ballot(mine) != 0ull) {
                const double wb = sq_wave_max_f64(mine ? fin : -INFINITY);
                if (lane == 0) atomicMax(&s_best, sq_ord(wb));
            }
        };
        auto bar = [&]() -> double {                                // what a run has to reach: :751's threshold, the best so far
            const unsigned long long sb = __atomic_load_n(&s_best, __ATOMIC_RELAXED);
            const double r = sb ? sq_unord(sb) : minfin;
            return r > minfin ? r : minfin;
        };
        uint32_t nX = 0, nC = 0, nS = 0;                            // entries of the three queues
        auto push = [&](uint32_t *qb, uint32_t &cnt, bool p, uint32_t v) {   // (wave-uniform call)
            const unsigned long long m = __ballot(p);
            if (p) qb[cnt + (uint32_t)__popcll(m & lt_mask)] = v;
            cnt += (uint32_t)__popcll(m);
        };
        auto push2 = [&](uint32_t *qb, uint32_t &cnt, bool p0, uint32_t v0, bool p1, uint32_t v1) {   // two units at once; most find nothing
            const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
            if ((m0 | m1) != 0ull) {
                const uint32_t c0 = (uint32_t)__popcll(m0);
                if (p0) qb[cnt + (uint32_t)__popcll(m0 & lt_mask)] = v0;
                if (p1) qb[cnt + c0 + (uint32_t)__popcll(m1 & lt_mask)] = v1;
                cnt += c0 + (uint32_t)__popcll(m1);
            }
        };
        const uint32_t nunits = (nl0 + 63u) >> 6;
        auto grab = [&]() -> uint32_t {                             // the next two units nobody has taken yet
            uint32_t v = 0u;
            if (lane == 0) v = atomicAdd(&s_unit, 2u);
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
        };
        uint32_t g = grab();
        // the entries of the units in hand, on their way
        SqRunA ca0 = {0u, 0u, 0.0}, ca1 = {0u, 0u, 0.0}; SqRunB cb0 = {0.0, 0.0}, cb1 = {0.0, 0.0};
        auto fetch = [&](uint32_t gg) {
            const uint32_t q0 = gg * 64 + lane, q1 = q0 + 64;
            ca0.lf = 0u; ca1.lf = 0u;
            if (q0 < nl0) { ca0 = LA[q0]; cb0 = LB[q0]; }
            if (q1 < nl0) { ca1 = LA[q1]; cb1 = LB[q1]; }
        };
        if (g < nunits) fetch(g);
#ifdef SQ_ROUNDS_PROF
        { RPROF(9); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long _n = wall_clock64(); _pt[8] += _n - _t; _t = _n; }   // (the first fetch's latency, booked as ext)
#endif
        bool cact = false, cfirst = false; uint32_t cq = 0; int ci = 0, cs = 0, cL = 0, ct0 = 0;   // the run this lane is cutting
        bool struct_ready = nwv == 1 || wv == 0;                    // (the first wave made the structure itself)
        // what the stream finds out about one entry: cut / needs a bound / needs ScoreStems / holds a finalscore that counts
        struct Cls { bool cut, cand, surv, mine; };
        auto classify = [&](const SqRunA &r, const SqRunB &rb, uint32_t q, double need) -> Cls {
            const int L = (int)(r.lf & SQ_RX_LEN);
            const int i = (int)(r.key & 0xFFFFu), j = (int)(r.key >> 16) - i, sa = i + L - 1, sb = j - L + 1;
            // strand [z0, z1] meets [lo, hi]  <=>  (hi - z0 | z1 - lo) >= 0: the sign bit of the AND over several tests is clear when
            // one of them holds (positions are small: nothing overflows; "no stem yet" is a pair of empty intervals)
#define SQ_OV(z0, z1, lo, hi) (((hi) - (z0)) | ((z1) - (lo)))
            // rows [i, sa] or columns [sb, j] meet a strand of the new stem: the run is cut.  A strand within six positions of
            // the run's span: the finalscore is void; within six of its four ends: the bound too
            const int nocut = SQ_OV(za0, za1, i, sa) & SQ_OV(zb0, zb1, i, sa) & SQ_OV(za0, za1, sb, j) & SQ_OV(zb0, zb1, sb, j);
            const int nofd = SQ_OV(za0, za1, i - 6, j + 6) & SQ_OV(zb0, zb1, i - 6, j + 6) & (rgmask | ((r.lf & SQ_RX_LVL) ? 0 : -1));
            const int noud = SQ_OV(za0, za1, i - 6, sa + 6) & SQ_OV(zb0, zb1, i - 6, sa + 6) & SQ_OV(za0, za1, sb - 6, j + 6) & SQ_OV(zb0, zb1, sb - 6, j + 6);
#undef SQ_OV
            const uint32_t fdm = ~(uint32_t)(nofd >> 31), udm = ~(uint32_t)(noud >> 31);   // all ones: void
            const bool live = L > 0;                                            // (lanes beyond the list hold length 0)
            Cls c;
            c.cut = live & (nocut >= 0);
            const uint32_t lf = r.lf & ~((fdm & (SQ_RX_FIN | SQ_RX_FB)) | (udm & (SQ_RX_UB | SQ_RX_FIN | SQ_RX_FB)));
            const bool stay = live & !c.cut;
            if (stay & (lf != r.lf)) LA[q].lf = lf;
            const bool p492 = stay & (r.bps >= minbps);                         // :492
            c.mine = p492 & ((lf & SQ_RX_FIN) != 0u) & (rb.fin >= minfin);      // :751
            c.surv = p492 & ((lf & (SQ_RX_FIN | SQ_RX_UB)) == SQ_RX_UB) & !(rb.ub < need) & !(((lf & SQ_RX_FB) != 0u) & (rb.fin < need));
            c.cand = p492 & ((lf & (SQ_RX_FIN | SQ_RX_UB)) == 0u);
            return c;
        };
        int cpb = 0, cplen = 0;                                     // the piece of that run which is handled next
        // the next piece of at least minlen cells of the run in hand, from cell ct0 on -- cell t of the run: row i + t, column
        // j - t; masked when either lies on a strand [za0, za1] or [zb0, zb1].  None left: the run is through (and dead, if no
        // piece of it stayed in the list)
        auto find_piece = [&]() {
            const int i = ci, j = cs - ci, L = cL;
            const int lo0 = za0 - i, hi0 = za1 - i, lo1 = zb0 - i, hi1 = zb1 - i, lo2 = j - za1, hi2 = j - za0, lo3 = j - zb1, hi3 = j - zb0;
            int pb = -1, plen = 0, t0 = ct0;
            while (t0 < L) {
                int b0 = t0;
#pragma unroll
                for (int rep = 0; rep < 4; rep++) {
                    if (b0 >= lo0 && b0 <= hi0) b0 = hi0 + 1;
                    if (b0 >= lo1 && b0 <= hi1) b0 = hi1 + 1;
                    if (b0 >= lo2 && b0 <= hi2) b0 = hi2 + 1;
                    if (b0 >= lo3 && b0 <= hi3) b0 = hi3 + 1;
                }
                if (b0 >= L) { t0 = L; break; }
                int pe = L;
                if (lo0 > b0 && lo0 < pe) pe = lo0;
                if (lo1 > b0 && lo1 < pe) pe = lo1;
                if (lo2 > b0 && lo2 < pe) pe = lo2;
                if (lo3 > b0 && lo3 < pe) pe = lo3;
                t0 = pe;
                if (pe - b0 >= minlen) { pb = b0; plen = pe - b0; break; }
            }
            ct0 = t0; cpb = pb; cplen = plen;
            if (pb < 0) {
                if (cfirst) { LA[cq].lf = 0u; atomicAdd(&s_ndead, 1u); }
                cact = false;
            }
        };
        RPROF(9);
        for (;;) {
            // ---- the stream: two units of 64 entries per step until a queue holds a wave of work or the list ends ----
            while (g < nunits && (nX | nC | nS) < 64u) {
                const double need = bar();
                const uint32_t q0 = g * 64 + lane, q1 = q0 + 64;
                const SqRunA r0 = ca0, r1 = ca1; const SqRunB rb0 = cb0, rb1 = cb1;
                g = grab();
                if (g < nunits) fetch(g);
                const Cls c0 = classify(r0, rb0, q0, need), c1 = classify(r1, rb1, q1, need);
                bool d0 = c0.cand, d1 = c1.cand;
                if (__ballot(d0 | d1) != 0ull) {
                    // (no bound yet: first with every factor at its maximum, which needs nothing but the bpscore)
                    const double k0 = (((r0.bps * ub_of) * ub_lf) * 1.25) * (1.0 + 0x1p-30), k1 = (((r1.bps * ub_of) * ub_lf) * 1.25) * (1.0 + 0x1p-30);
                    d0 = d0 & !((ub_lf < INFINITY) & (r0.bps >= 0) & (k0 < need));
                    d1 = d1 & !((ub_lf < INFINITY) & (r1.bps >= 0) & (k1 < need));
                }
                if (__ballot(c0.mine | c1.mine) != 0ull) {
                    take(c0.mine, rb0.fin, r0.key, r0.lf & SQ_RX_LEN, r0.bps);
                    take(c1.mine, rb1.fin, r1.key, r1.lf & SQ_RX_LEN, r1.bps);
                    const double f0 = c0.mine ? rb0.fin : -INFINITY, f1 = c1.mine ? rb1.fin : -INFINITY;
                    raise(c0.mine | c1.mine, f0 > f1 ? f0 : f1);
                }
#ifdef SQ_ROUNDS_PROF
                { const int km = __popcll(__ballot(c0.mine)) + __popcll(__ballot(c1.mine)); _cnt[0] += 128; _cnt[1] += km; }   // (every lane counts: the ballots are the wave's)
#endif
                push2(qcut, nX, c0.cut, q0, c1.cut, q1);
                push2(qcand, nC, d0, q0, d1, q1);
                push2(qsurv, nS, c0.surv, q0, c1.surv, q1);
            }
            sq_wave_lds_fence();
            RPROF(3);
            const bool end = g >= nunits;
            const unsigned long long cm = __ballot(cact);
            const bool cutbusy = cm != 0ull || nX > 0;
            if (nS >= 64 || (end && !cutbusy && nC == 0 && nS > 0)) {
                // ---- score: ScoreStems for a wave of runs whose bound reaches the bar ----
#ifdef SQ_ROUNDS_PROF
                long long _t2 = wall_clock64(); _sp[4]++;
#endif
                if (!struct_ready) {                                            // (the first wave is still putting the last stem in)
                    while (__atomic_load_n(&s_ready, __ATOMIC_RELAXED) < roundno) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    struct_ready = true;
                }
                SPROF(0);
                const double need = bar();
                const uint32_t m = nS < 64 ? nS : 64; nS -= m;
                const bool have = (uint32_t)lane < m;
                const uint32_t q = have ? qsurv[nS + lane] : 0u;
                uint32_t key = 0, lf = 0; double bps = 0.0, ub = 0.0;
                if (have) { const SqRunA r = LA[q]; key = r.key; lf = r.lf; bps = r.bps; ub = LB[q].ub; }
#ifdef SQ_ROUNDS_PROF
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                SPROF(1);
#endif
                bool ok = have && !(ub < need);                                 // (the bar may have risen since the run was queued)
#ifdef SQ_ROUNDS_PROF
                _cnt[2] += __popcll(__ballot(ok));
#endif
                double fin = 0.0;
                const int L = (int)(lf & SQ_RX_LEN);
                const int i0 = (int)(key & 0xFFFFu), j0 = (int)(key >> 16) - i0;
                // ScoreStems' walk (sq_score.h), in parts.  Side by side -- a lane per run -- a slice of strands at a time, while many
                // lanes are walking: such a step lasts as long as its longest walk (263 strands on an alignment's row of 700, where the
                // average walk visits 73), so the walk ENDS EARLY where the order factor alone puts the run below the bar (seven
                // walks in eight of such a row, after a quarter of their strands: the bound stays with the entry, SQ_RX_FB), and
                // when only a few lanes are still walking on a structure of hundreds of strands the whole wave takes what is left
                // of each of them, a strand per lane (sq_walk_wave: ~1 us per 64 strands where a lone lane pays two dependent LDS
                // reads per strand).  (A wave per run for EVERY run was measured twice and lost: the steps are full.)
                const int sa = i0 + L - 1, sb = j0 - L + 1;
                const bool early = s_ofmono != 0;
                const bool wavewalk = nstrand >= ra.wave_min;
                SqWalkPart st = {0, -1, 0, 0, 0, 0, 0, 0u, 0u};
                int status = SQ_WALK_DONE;
                double bnd = 0.0;
                if (ok) sq_walk_begin(env, sa, st);
                for (bool walking = ok;;) {
                    if (walking) {
                        status = sq_walk_lane(env, sa, sb, st, wavewalk ? 16 : 0x7fffffff, early, ub, l_ofr, bar(), &bnd);
                        walking = status == SQ_WALK_MORE;
                    }
                    unsigned long long wm = __ballot(walking);
                    if (wm == 0ull) break;
                    if (__popcll(wm) <= ra.wave_lanes) {
                        while (wm != 0ull) {
                            const int r = __ffsll((long long)wm) - 1;
                            wm &= wm - 1ull;
#define SQ_RL(x) __builtin_amdgcn_readlane((int)(x), r)
                            SqWalkPart w = {SQ_RL(st.k), SQ_RL(st.inblockend), SQ_RL(st.nrec), SQ_RL(st.be0), SQ_RL(st.be1), SQ_RL(st.covered), SQ_RL(st.brackets),
                                            (uint32_t)SQ_RL(st.lv0), (uint32_t)SQ_RL(st.lv1)};
                            const double rub = __hiloint2double(SQ_RL(__double2hiint(ub)), SQ_RL(__double2loint(ub)));
                            double b2 = 0.0;
                            const int s2 = sq_walk_wave(env, SQ_RL(sa), SQ_RL(sb), w, early, rub, l_ofr, bar(), &b2, lane);
#undef SQ_RL
                            if (lane == r) { st = w; status = s2; bnd = b2; }
                        }
                        break;
                    }
                }
                if (ok) {
                    if (status == SQ_WALK_OUT) {
                        LB[q].fin = bnd; LA[q].lf = (lf & ~SQ_RX_FIN) | SQ_RX_FB | SQ_RX_LVL;
                        ok = false;
                    } else {
                        const SqWalk w = {st.nrec, st.be0, st.be1, st.covered, st.brackets, (uint64_t)st.lv0 | ((uint64_t)st.lv1 << 32)};
                        fin = sq_stem_finalscore_of(env, i0, j0, L, bps, w);
                        // (a finalscore that met no bracket strand -- no strand inside the span whose partner lies outside it -- does not
                        // read the levels at all: it outlives the rounds that renumber them, SQ_RX_LVL says which do not)
                        LB[q].fin = fin; LA[q].lf = (lf & ~(SQ_RX_LVL | SQ_RX_FB)) | SQ_RX_FIN | (st.brackets == 0 ? 0u : SQ_RX_LVL);
                        ok = fin >= minfin;                                     // :751
                    }
                }
                take(ok, fin, key, (uint32_t)L, bps);
                raise(ok, fin);
                SPROF(3);
                RPROF(6);
                continue;
            }
            if (nC >= 64 || (end && !cutbusy && nC > 0)) {
                // ---- bound: sq_run_upper for a wave of runs that have none ----
                const double need = bar();
                const uint32_t m = nC < 64 ? nC : 64; nC -= m;
                const bool have = (uint32_t)lane < m;
                const uint32_t q = have ? qcand[nC + lane] : 0u;
                bool ps2 = false;
                if (have) {
                    const SqRunA r = LA[q];
                    const int L = (int)(r.lf & SQ_RX_LEN), i0 = (int)(r.key & 0xFFFFu), j0 = (int)(r.key >> 16) - i0;
                    const double ub = upper_of(r.bps, i0, j0, L);
                    LB[q].ub = ub; LA[q].lf = r.lf | SQ_RX_UB;
                    ps2 = !(ub < need);
                }
                push(qsurv, nS, ps2, q);
                sq_wave_lds_fence();
                RPROF(5);
                continue;
            }
            if (cm != 0ull || nX >= 64 || (end && nX > 0)) {
                // ---- cut: every lane one run, one piece per step; a run leaves with its last piece ----
                const double need = bar();
                if (cm == 0ull) {
                    const uint32_t m = nX < 64 ? nX : 64; nX -= m;
                    if ((uint32_t)lane < m) {
                        cq = qcut[nX + lane];
                        const SqRunA r = LA[cq];
                        ci = (int)(r.key & 0xFFFFu); cs = (int)(r.key >> 16); cL = (int)(r.lf & SQ_RX_LEN); ct0 = 0; cfirst = true; cact = true;
                        find_piece();
                    }
                }
                bool ps2 = false; uint32_t at = 0;
                if (cact) {
                    const int i = ci, j = cs - ci, pb = cpb, plen = cplen;
                    double pos;
                    const double bps = run_bps(i + pb, j - pb, plen, pos);
                    if (!(pos < minbps)) {
                        at = cq;
                        bool room = true;
                        if (!cfirst) {
                            at = atomicAdd(&s_nlist, 1u);
                            if (at >= (uint32_t)cap) { a.ctr->cand_ovf = 1; room = false; }
                        }
                        if (room) {
                            cfirst = false;
                            uint32_t lf = (uint32_t)plen;
                            if (bps >= minbps) {                                // :492 (a piece below it stays for ITS pieces)
                                const double ub = upper_of(bps, i + pb, j - pb, plen);
                                lf |= SQ_RX_UB;
                                LB[at].ub = ub;
                                ps2 = !(ub < need);
                            }
                            LA[at] = SqRunA{((uint32_t)cs << 16) | (uint32_t)(i + pb), lf, bps};
                        }
                    }
                    find_piece();
                }
                push(qsurv, nS, ps2, at);
                sq_wave_lds_fence();
                RPROF(4);
                continue;
            }
            if (end) break;
        }
        int bany = bfin > -INFINITY ? 1 : 0;
        __threadfence_block();
        // ---- ChooseStems' first element over the block ----
        {
            // the wave's best: the highest finalscore, the smallest key among equals (one lane: keys are distinct); the best of
            // all the OTHER runs beside it (DPP reductions: the butterfly of six values they replace was 54 trips over the LDS
            // crossbar per round)
            const double wmax = sq_wave_max_f64(bfin);
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
        RPROF(7);
        if (!bany) { retire(nstems, 0); RPROF_OUT(); return; }      // :1192-1193 no new stem: the structure is final
        if (ra.ties && sfin == bfin) {
            // a pool that may branch: ChooseStems returns, beside its first element, every run that reaches the best finalscore
            // (range factor 1.0) AND shares a base with all the runs taken before it (:769-789).  With the first element alone in
            // the list that is: some other run at the best finalscore shares a base with the winner -- then the pool grows here,
            // the structure stops, unfinished, and the device pools fold its job.  Tied runs that touch no base of the winner
            // change nothing (the next rounds meet them again).  Every run at the best finalscore was scored or kept this round
            // (its bound cannot lie below the bar), so its entry holds the finalscore
            const int wi = (int)(bkey & 0xFFFFu), wj = (int)(bkey >> 16) - wi, wl = (int)blen;
            const uint32_t nl = s_nlist < (uint32_t)cap ? s_nlist : (uint32_t)cap;
            int found = 0;
            for (uint32_t q = tid; q < nl; q += nthr) {
                const SqRunA r = LA[q];
                const int L = (int)(r.lf & SQ_RX_LEN);
                if (L == 0 || !(r.lf & SQ_RX_FIN) || r.key == bkey) continue;
                if (LB[q].fin != bfin) continue;
                const int i = (int)(r.key & 0xFFFFu), j = (int)(r.key >> 16) - i;
                // strands [i, i + L) and (j - L, j] against the winner's
                const bool hit = (i <= wi + wl - 1 && i + L - 1 >= wi) || (i <= wj && i + L - 1 >= wj - wl + 1) ||
                                 (j - L + 1 <= wi + wl - 1 && j >= wi) || (j - L + 1 <= wj && j >= wj - wl + 1);
                if (hit) found = 1;
            }
            if (__syncthreads_or(found)) {
                if (tid == 0) {
                    structs[b].nstrand = -1;
                    const uint32_t idx = atomicAdd(cio.d_nfin, 1u);
                    cio.h_fin[idx] = (unsigned long long)(uint32_t)st.job | (1ull << 62);
                }
                return;
            }
        }
        const int i0 = (int)(bkey & 0xFFFFu), j0 = (int)(bkey >> 16) - i0, len = (int)blen;
        const int k = nstems;
        if (k >= ch.tcap) { if (tid == 0) a.ctr->out_ovf = 1; retire(k, 0); return; }
        if (k >= ra.tmax) {
            // the block's LDS lists were sized for fewer stems than the structure's bound (two blocks per CU: sq_fold.hip): the
            // structure stops, unfinished, like one that meets a tie -- the device pools fold its job
            if (tid == 0) {
                if (ra.ties) {
                    structs[b].nstrand = -1;
                    const uint32_t idx = atomicAdd(cio.d_nfin, 1u);
                    cio.h_fin[idx] = (unsigned long long)(uint32_t)st.job | (1ull << 62);
                } else a.ctr->out_ovf = 1;
            }
            if (!ra.ties) retire(k, 0);
            return;
        }
        // ---- the child.  Before the barrier: the first wave books the stem and its crossing weights (:121-124) -- which say
        // whether the levels will be taken anew --, the other waves update partner array and prefix counts: positions p and
        // above lose the new pairs below p, and separators never pair, so SU stays ----
        const int zb = j0 - len + 1;
        int mycross = 0; bool ac = anycross;
        if (wv == 0) {
            int mycc = 0;
            for (int q = lane; q < k; q += 64)
                if (sq_chain_cross(XL.i[q], XL.j[q], i0, j0)) { XL.cc[q] += len; mycc += XL.len[q]; mycross = 1; }
            const int newcc = sq_wave_sum32(mycc);
            mycross = __ballot(mycross) != 0ull ? 1 : 0;
            ac = anycross || mycross;
            if (lane == 0) {
                XL.i[k] = (int16_t)i0; XL.j[k] = (int16_t)j0; XL.len[k] = (int16_t)len; XL.cc[k] = newcc;
                gst[k] = SqChainStem{i0, j0, len, newcc};           // (the tail reads the stems there; the weights of the others stay in LDS)
                cio.h_stems[ch.toff + k] = SqStemOut{i0, j0, len, 0, bbps, bfin};   // (pinned; measured: the posted write costs a round nothing)
                s_cross = ac ? 1 : 0;
                s_best = 0ull;                                      // (the next pass starts without a bar and with all units to take)
                s_unit = 0u;
                // levels: the full rule when the new stem crosses something (crossing weights changed: the order of the first fit
                // may have); a stem without crossings joins group 0 and only the groups' ranking is taken anew -- the groups stay
                // what they were, so the NUMBER of levels among any set of strands does, and that number is all ScoreStems takes
                // from the levels (:728-729): the finalscores kept in the list stay valid.  After the full rule none does
                s_regroup = ac && !(anycross && !mycross && ngroups > 0) ? 1 : 0;
            }
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
        const bool was_cross = anycross;
        anycross = s_cross != 0;
        nl0 = s_nlist < (uint32_t)cap ? s_nlist : (uint32_t)cap;     // (with the pieces this round appended; nothing is appended outside the pass)
        { const uint32_t nd = s_ndead; compact = nd * 3 > nl0 || nl0 + ((nl0 - nd) >> 1) + 64 > (uint32_t)cap; }
        nstems = k + 1;
        if ((double)nstems == ch.maxstems) { retire(nstems, 1); return; }   // :1168-1174 (checked before the next evaluation)
        roundno++;
        za0 = i0; za1 = i0 + len - 1; zb0 = zb; zb1 = j0;
        RPROF(7);
        // ---- the rest of the child by the first wave alone, while the others are in the next pass: levels when stems cross, the
        // sorted strand list (sq_extend.h), the skip pointers; s_ready tells the score steps ----
        if (wv == 0) {
#ifdef SQ_ROUNDS_PROF
            long long _x0 = wall_clock64();
#endif
            if (ac) {
                if (was_cross && !mycross && ngroups > 0) sq_stem_levels_join(XL, k + 1, ngroups, len, lane);
                else {
#ifdef SQ_ROUNDS_PROF
                    ngroups = sq_stem_levels_wave(XL, k + 1, lane, &a.ctr->level_ovf, _xl);
                    _xt[3]++;
#else
                    ngroups = sq_stem_levels_wave(XL, k + 1, lane, &a.ctr->level_ovf);
#endif
                }
            }
#ifdef SQ_ROUNDS_PROF
            { const long long _n = wall_clock64(); _xt[0] += _n - _x0; _x0 = _n; }
#endif
            sq_rounds_insert_strands(XL, ac, k, strbuf, sidxbuf, s_skip, nstrand, i0, j0, len, lane);   // (+ the skip pointers over the blocks ScoreStems' sweep registers)
            sq_wave_lds_fence();
#ifdef SQ_ROUNDS_PROF
            { const long long _n = wall_clock64(); _xt[1] += _n - _x0; }
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __atomic_store_n(&s_ready, roundno, __ATOMIC_RELAXED);
        }
        nstrand += 2;
        RPROF(8);
    }
}
