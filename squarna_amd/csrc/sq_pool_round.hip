// sq_pool_round.hip -- one round of the device pools (sq_pool.hip) for SHORT structures as ONE kernel.
//
// A round of the pools was state -> scan -> score -> choose (-> pool scan -> extend): four launches in which a
// structure of a 60-nt sequence is four blocks that each spend half of their few microseconds on set-up, hand their
// results to the next one through global memory (partner / prefix arrays, free-position words, candidate keys,
// survivor records, the round's best) and hold wave slots while they wait for it.  With batches in flight the chip is
// short of exactly those slots.  Here ONE wave takes a structure through the whole of
// OptimalStems (SQRNdbnseq.py:792-833) and ChooseStems (:754-789):
//
//   state       partner array, prefix counts, free-position words from the structure's sorted strands -- in LDS;
//   scan        AnnotateStems as the bit-diagonal scan (sq_scan.h); the runs are staged in LDS and, 64 at a time, given
//               their exact bpscore at once (the candidate list never exists); those that pass :492 wait in LDS
//               (beyond its room: in the structure's slice of the candidate arena);
//   score       ScoreStems on them behind the bound of sq_cellrun.h, finalscores beside them in LDS;
//   choose      the stopper's single pick (:1147), or the range :769-778, the reference's stable descending order and
//               the conflict filter :779-789 -- sq_pool_choose_kernel's steps on LDS data.
//
//   extend      comes FIRST: a child builds itself -- parent + its pick through sq_extend_structure (crossing weights, levels,
//               sorted strands), strands straight into LDS -- and leaves its lists in its slot for its own children; the
//               launched form's sq_pool_extend_kernel (one wave per parent, a chain of a dozen dependent global loads per
//               child: 95 % of its wave cycles waiting, a fifth of all wave cycles of a crowded step) is gone.  A child that
//               is full (:1123-1129) and a structure that finds no stem (:1155-1156) log themselves as final.
//
// Between two rounds only sq_pool_scan_kernel runs (children's slots and parents, the jobs' pool state).  Results are those
// of the launched kernels bit for bit (tests fold both ways: SQ_NO_POOL_ROUND).
#include <hip/hip_runtime.h>
#include "sq_device.h"
#include "sq_extend.h"
#include "sq_tail_dev.h"
#include "sq_cells.h"
#include "sq_cellrun.h"
#include "sq_score.h"
#ifndef SQ_PR_AHEAD
#define SQ_PR_AHEAD 2              // word-rows of the bit matrix whose loads the scan issues together (sq_scan.h: SQ6_AHEAD)
#endif
#define SQ6_AHEAD SQ_PR_AHEAD
#include "sq_scan.h"
#include "sq_pool_round.h"
#include "sq_rounds.h"             // (SqRun: a run with its exact bpscore -- the root lists' record)

#define SQ_POOL_CMAX 64         // stems ChooseStems may return for one structure (== lanes of the conflict test)

__device__ __forceinline__ bool sq_pr_shares_base(int ai, int aj, int al, int bi, int bj, int bl)   // :783-786
{
    const int as0 = ai, as1 = ai + al - 1, at0 = aj - al + 1, at1 = aj;
    const int bs0 = bi, bs1 = bi + bl - 1, bt0 = bj - bl + 1, bt1 = bj;
    return (as0 <= bs1 && bs0 <= as1) || (as0 <= bt1 && bt0 <= as1) || (at0 <= bs1 && bs0 <= at1) || (at0 <= bt1 && bt0 <= at1);
}

// the survivors of :492: the first `cap` in LDS, the rest in the structure's slice of the candidate arena
struct SqPrSurv {
    double *bps, *fin; uint32_t *key; uint16_t *len, *place; int cap;
    SqOk *spill; uint32_t spill_cap; SqCounters *ctr;
    uint32_t *pool_ovf;                               // (list form) SqPoolHdr::ovf: more runs within range than a structure's region holds
    __device__ __forceinline__ void put(uint32_t at, uint32_t k, int L, double b)
    {
        if (at < (uint32_t)cap) { key[at] = k; len[at] = (uint16_t)L; bps[at] = b; }
        else if (at - cap < spill_cap) spill[at - cap] = SqOk{k, (uint32_t)L, b, 0.0};
        else ctr->cand_ovf = 1;
    }
    // (kept lists: the finalscore the parent left -- NaN: none -- and the run's place in the structure's own list; a spilled
    // survivor keeps the place in the upper half of its length word)
    __device__ __forceinline__ void put_kept(uint32_t at, uint32_t k, int L, double b, double f, uint32_t pl)
    {
        if (at < (uint32_t)cap) { key[at] = k; len[at] = (uint16_t)L; bps[at] = b; fin[at] = f; place[at] = (uint16_t)pl; }
        else if (at - cap < spill_cap) spill[at - cap] = SqOk{k, (uint32_t)L | (pl << 16), b, f};
        else *pool_ovf = 1;         // (ChooseStems' own room is smaller still: like its overflow, the host's loop repeats the fold)
    }
    __device__ __forceinline__ uint32_t get_place(uint32_t at) const { return at < (uint32_t)cap ? (uint32_t)place[at] : spill[at - cap].len >> 16; }
    __device__ __forceinline__ void get(uint32_t at, uint32_t &k, int &L, double &b) const
    {
        if (at < (uint32_t)cap) { k = key[at]; L = len[at]; b = bps[at]; }
        else { const SqOk o = spill[at - cap]; k = o.key; L = (int)(o.len & 0xFFFFu); b = o.bps; }
    }
    __device__ __forceinline__ void set_fin(uint32_t at, double f) { if (at < (uint32_t)cap) fin[at] = f; else spill[at - cap].fin = f; }
    __device__ __forceinline__ double get_fin(uint32_t at) const { return at < (uint32_t)cap ? fin[at] : spill[at - cap].fin; }
};

// the scan's sink: runs staged in LDS; whenever 64 of them wait they get their bpscore and the ones that pass :492 join
// the survivors (the wave is the whole block: its barrier orders the LDS traffic).  The fill counts live in registers: the
// wave reserves the places of a word-row's runs at once (sq_scan.h)
#ifdef SQ_PR_PROF
__shared__ long long sq_pr_prof_bps;                  // (instrumentation: time inside the sink's flushes)
#endif
struct SqPrSink {
    uint2 *stage;                                     // staging buffer of SQ_PR_STAGE entries (LDS)
    uint2 *over; uint32_t over_cap;                   // runs beyond the buffer: the structure's key array in the arena
    const SqCellEnv &cenv; const SqDevCtx &c; const SqJob &jb; SqPrSurv &sv; uint32_t &ns;
    double minbps; SqCounters *ctr;
    uint32_t n, nover;                                // runs staged since the last flush (beyond SQ_PR_STAGE: in `over`), runs in `over` before them
    __device__ __forceinline__ uint32_t reserve(uint32_t total, int) { const uint32_t b0 = n; n += total; return b0; }
    __device__ __forceinline__ void put(uint32_t at, uint32_t key, uint32_t len)
    {
        if (at < SQ_PR_STAGE) stage[at] = make_uint2(key, len);
        else {
            const uint32_t g = nover + (at - SQ_PR_STAGE);
            if (g < over_cap) over[g] = make_uint2(key, len); else ctr->cand_ovf = 1;
        }
    }
    __device__ __forceinline__ void score64(const uint2 *src, uint32_t m, int lane)      // m <= 64 runs
    {
        const bool have = (uint32_t)lane < m;
        const uint2 kl = have ? src[lane] : make_uint2(0u, 0u);
        double bps = 0.0, pos = 0.0;
        if (have) {
            const int i = (int)(kl.x & 0xFFFFu), j = (int)(kl.x >> 16) - i;
            bps = sq_cellrun_bps<false>(cenv, c, jb, i, j, (int)kl.y, pos);
#ifdef SQ_PR_DUP_BPS
            { int i2 = i; asm volatile("" : "+v"(i2)); const double b2 = sq_cellrun_bps<false>(cenv, c, jb, i2, j, (int)kl.y, pos); if (b2 != bps) ctr->cand_ovf = 1; }
#endif
        }
        const bool ok = have && bps >= minbps;                                            // :492
        const unsigned long long m2 = __ballot(ok);
        if (ok) sv.put(ns + (uint32_t)__popcll(m2 & ((1ull << lane) - 1ull)), kl.x, (int)kl.y, bps);
        ns += (uint32_t)__popcll(m2);
    }
    __device__ __forceinline__ void flush(int lane)
    {
#ifdef SQ_PR_PROF
        const long long t0_ = wall_clock64();
#endif
        __syncthreads();
        const uint32_t m = n < SQ_PR_STAGE ? n : SQ_PR_STAGE;
        for (uint32_t b0 = 0; b0 < m; b0 += 64) score64(stage + b0, min(m - b0, 64u), lane);
        if (n > SQ_PR_STAGE) { const uint32_t more = n - SQ_PR_STAGE; nover = nover + more < over_cap ? nover + more : over_cap; }
        n = 0;
        __syncthreads();
#ifdef SQ_PR_PROF
        if (lane == 0) sq_pr_prof_bps += wall_clock64() - t0_;
#endif
    }
    __device__ __forceinline__ void poll(int lane) { if (n >= 64u) flush(lane); }
    __device__ __forceinline__ void drain(int lane) { flush(lane); }
};

struct SqPrNullSink {       // (instrumentation: -DSQ_PR_DUP_SCAN runs the scan a second time into nothing, for instruction counts)
    uint32_t n;
    __device__ __forceinline__ uint32_t reserve(uint32_t total, int) { const uint32_t b0 = n; n += total; return b0; }
    __device__ __forceinline__ void put(uint32_t at, uint32_t key, uint32_t len) { n ^= (at ^ key ^ len) & 0u; }
    __device__ __forceinline__ void poll(int) {}
    __device__ __forceinline__ void drain(int) {}
};
struct SqEnt4 { uint32_t key[4], lf[4]; double bps[4], fin[4]; };      // a page of the list form's source: four entries per lane
#ifndef SQ_PR_WAVES
#define SQ_PR_WAVES 4              // waves per SIMD the register budget is set for (4: 128 VGPRs)
#endif
// ROOT: AnnotateStems as a pass over the job's root list (SqPoolRoundArgs::root; a kernel of its own -- sq_pool_round_root_kernel --
// so that the scanning form's code stays what it was: it is bound by vector-instruction issue)
template <bool ROOT>
__device__ __forceinline__ void sq_pool_round_body(const SqDevCtx &c, const SqScanArgs &a, const SqPoolIO &pio, const SqPoolRoundArgs &ra)
{
    extern __shared__ __attribute__((aligned(16))) char pr_dyn[];
    __shared__ SqCellTmp s_ctmp;
    __shared__ int s_ri[SQ_POOL_CMAX], s_rj[SQ_POOL_CMAX], s_rl[SQ_POOL_CMAX];
    const int lane = threadIdx.x;
    const int s = ra.lo + (int)blockIdx.x;                  // the structure's position in the round's list == its slot
    // (rounds enqueued ahead of the host are launched with every slot as their grid: the generation's size is the scan kernel's
    // word -- zero once the pools have run empty or a capacity was exceeded)
    // (the header and the child's word are read together: the first link of the entry's chain of dependent loads)
    const SqPoolHdr hdr0 = sq_kload(pio.hdr);
    const uint32_t pw0 = (uint32_t)sq_kload(pio.parent_of + s);
    asm volatile("" :: "s"(pw0));                           // (asked for HERE, not below the branch)
    if (ra.ahead && s >= (int)(ra.parity ? hdr0.S[1] : hdr0.S[0])) return;
#ifdef SQ_PR_PROF
    long long _pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long _t = wall_clock64(); const long long _t00 = _t; int _ninh = 0, _nwalk = 0; long long _xw = 0, _xc = 0, _xa = 0; int _ncutrun = 0, _nwserve = 0;
    if (threadIdx.x == 0) sq_pr_prof_bps = 0;
#define PRPROF(k) do { const long long _n = wall_clock64(); _pt[k] += _n - _t; _t = _n; } while (0)
#ifndef SQ_PR_PROF_SLOW
#define SQ_PR_PROF_SLOW 1000000      /* us: structures slower than this are printed too */
#endif
#define PRPROF_OUT(ns_, nin_) do { if (lane == 0 && ((s % 997) == 0 || (wall_clock64() - _t00) > 100ll * SQ_PR_PROF_SLOW)) printf("pool round list %u kept %d walked %d cutruns %d walkserves %d us: walks %.1f cuts %.1f alloc %.1f | s=%d n=%d nstrand=%d ns=%u nin=%d | us: bps %.1f entry %.1f ext %.1f extend %.1f setup %.1f state %.1f scan %.1f score %.1f choose %.1f total %.1f\n", \
        mycnt, _ninh, _nwalk, _ncutrun, _nwserve, _xw * 0.01, _xc * 0.01, _xa * 0.01, s, n, nstrand, (unsigned)(ns_), (int)(nin_), sq_pr_prof_bps * 0.01, _pt[6] * 0.01, _pt[7] * 0.01, _pt[0] * 0.01, _pt[1] * 0.01, _pt[2] * 0.01, _pt[3] * 0.01, _pt[4] * 0.01, _pt[5] * 0.01, (wall_clock64() - _t00) * 0.01); } while (0)
#else
#define PRPROF(k) do {} while (0)
#define PRPROF_OUT(ns_, nin_) do {} while (0)
#endif
    const size_t cur = (size_t)ra.parity * pio.smax, prv = (size_t)(ra.parity ^ 1) * pio.smax;
    const uint32_t round = hdr0.round;                      // (sq_pool_scan_kernel advances it behind this kernel)
    // the second link: the parent's records and the pick (round 0: the structure's own records), asked for before the LDS is carved
    const int p = round ? (int)(pw0 & 0x3FFFFFFu) : 0, k = round ? (int)(pw0 >> 26) : 0;   // (sq_pool_scan_kernel: parent | pick index << 26)
    const size_t prow = round ? prv + (size_t)p : cur + (size_t)s;
    const SqStruct pst = sq_kload(pio.structs + prow);
    const SqChain prec = sq_kload(pio.recs + prow);
    const SqPoolPick pk = sq_kload(pio.chosen + (round ? prow * pio.cmax + k : 0));
    const SqPoolRoundLds Lo = sq_pool_round_lds(ra.lds_n, ra.str_cap, ra.cell_entries, ra.surv_cap, ra.tmax);
    int16_t *const P = reinterpret_cast<int16_t *>(pr_dyn + Lo.off_P);
    int16_t *const U = reinterpret_cast<int16_t *>(pr_dyn + Lo.off_U);
    int16_t *const SU = reinterpret_cast<int16_t *>(pr_dyn + Lo.off_SU);
    uint8_t *const E = reinterpret_cast<uint8_t *>(pr_dyn + Lo.off_E);
    uint8_t *const l_ci = reinterpret_cast<uint8_t *>(pr_dyn + Lo.off_ci);
    uint8_t *const l_code = reinterpret_cast<uint8_t *>(pr_dyn + Lo.off_code);
    uint32_t *const FG = reinterpret_cast<uint32_t *>(pr_dyn + Lo.off_fg);
    SqStrand *const s_str = reinterpret_cast<SqStrand *>(pr_dyn + Lo.off_str);
    int16_t *const s_sidx = reinterpret_cast<int16_t *>(pr_dyn + Lo.off_sidx);
    uint16_t *const s_skip = reinterpret_cast<uint16_t *>(pr_dyn + Lo.off_skip);
    uint2 *const s_stage = reinterpret_cast<uint2 *>(pr_dyn + Lo.off_stage);
    double *const s_cell = reinterpret_cast<double *>(pr_dyn + Lo.off_cell);
    // the structure's stems (for the log of final structures); the extension's other arrays borrow the room of the staging
    // buffer and the survivors, which nobody uses yet
    SqExtendLds XL;
    XL.i = reinterpret_cast<int16_t *>(pr_dyn + Lo.off_stems); XL.j = XL.i + Lo.t8; XL.len = XL.j + Lo.t8;
    XL.cc = reinterpret_cast<int32_t *>(pr_dyn + Lo.off_stage);
    XL.gsize = XL.cc + Lo.t8;
    XL.ord = reinterpret_cast<int16_t *>(XL.gsize + 64);
    XL.grp = reinterpret_cast<uint8_t *>(XL.ord + Lo.t8); XL.lvl = XL.grp + Lo.t8; XL.rank = XL.lvl + Lo.t8;

    int job, nstems, nstrand; double maxstems;
    auto log_final = [&](uint32_t round_kind, int nst) {    // (sq_pool_extend_kernel's record; the stems from LDS)
        uint32_t idx = 0, so = 0;
        if (lane == 0) sq_log_reserve(pio.fin_ctr, (uint32_t)nst, idx, so);
        idx = (uint32_t)__shfl((int)idx, 0, 64); so = (uint32_t)__shfl((int)so, 0, 64);
        if (idx >= pio.fin_cap || so + (uint32_t)nst > pio.fin_stem_cap) {
            if (lane == 0) {                                 // (the host repeats the fold with its own loop and an empty log)
                pio.hdr->ovf = 1; pio.fin_ctr[2] = 1;
                if (idx < pio.fin_cap) pio.fin[idx] = SqPoolFin{job, SQ_FIN_KIND_G0 + round_kind, s, 0, 0u, SQ_FIN_SRC_LOG};
            }
            return;
        }
        for (int q = lane; q < nst; q += 64) pio.fin_stems[so + q] = SqPoolStem{XL.i[q], XL.j[q], XL.len[q], 0};
        if (lane == 0) pio.fin[idx] = SqPoolFin{job, SQ_FIN_KIND_G0 + round_kind, s, nst, so, SQ_FIN_SRC_LOG};
    };
    // (the loads below are ordered by what they depend on, not by who uses them: the kernel's entry is a chain of dependent
    // trips to L2 -- header, parent, the parent's records and pick, its stems and strands, the job -- in a wave that lives ~30 us;
    // everything a link of the chain names is asked for as soon as that link has arrived)
    static_assert(SQ_PR_MAXN <= 256, "the entry asks for a lane's letters and masks as four bytes");
    SqPoolJob *J; double cursub; int cursize; SqCellPre cpre; uint32_t e4 = 0;
    // (kept lists: the strands of the child's own stem -- none: intervals nothing meets -- and whether that stem crosses another,
    // i.e. the levels were taken anew, sq_rounds.hip)
    int za0 = 0x7FFF, za1 = -0x7FFF, zb0 = 0x7FFF, zb1 = -0x7FFF; bool regroup = false;
    if (round == 0) {                                       // the empty structure of a job (sq_pool_init_kernel)
        job = pst.job; nstems = 0; nstrand = 0; maxstems = prec.maxstems;
        J = pio.jobs + sq_kload(pio.jobrec_of + job); cursub = sq_kload(&J->cursubopt); cursize = sq_kload(&J->cursize);
        if (!ROOT) {                                         // (n <= SQ_PR_MAXN = 4 x 64: a lane's letters and masks are four bytes each)
            const SqJob *const jp = c.jobs + job;
            const int jn = sq_kload(&jp->n); const int64_t jpos = sq_kload(&jp->pos_off);
            cpre = sq_cell_preload(c, c.psets + sq_kload(&jp->pset), jpos, jn, sq_kload(&jp->default_reacts) || sq_kload(&jp->react_levels) == 0, lane);
#pragma unroll
            for (int t = 0; t < 4; t++) e4 |= (uint32_t)c.e0c[jpos + min(lane + 64 * t, jn - 1)] << (8 * t);   // (no branch: the four go out together)
        }
    } else {
        // ---- the child builds itself: parent p (previous generation) + its k-th pick ----
        job = pst.job; maxstems = prec.maxstems;
        if (prec.nstems >= pio.pt) { if (lane == 0) { pio.hdr->ovf = 1; pio.nchild[s] = 0; } return; }
        const SqExtendPre pre = sq_extend_preload(pio.stems + prec.toff, prec.nstems, pio.strands + pst.strand_off, pio.sidx + pst.strand_off,
                                                  pst.nstrand, lane);
        J = pio.jobs + sq_kload(pio.jobrec_of + job); cursub = sq_kload(&J->cursubopt); cursize = sq_kload(&J->cursize);
        if (!ROOT) {                                         // (n <= SQ_PR_MAXN = 4 x 64: a lane's letters and masks are four bytes each)
            const SqJob *const jp = c.jobs + job;
            const int jn = sq_kload(&jp->n); const int64_t jpos = sq_kload(&jp->pos_off);
            cpre = sq_cell_preload(c, c.psets + sq_kload(&jp->pset), jpos, jn, sq_kload(&jp->default_reacts) || sq_kload(&jp->react_levels) == 0, lane);
#pragma unroll
            for (int t = 0; t < 4; t++) e4 |= (uint32_t)c.e0c[jpos + min(lane + 64 * t, jn - 1)] << (8 * t);   // (no branch: the four go out together)
        }
        const int i0 = (int)(pk.key & 0xFFFFu), j0 = (int)(pk.key >> 16) - i0, len = (int)pk.len;
        const int toff = (int)((cur + (size_t)s) * (size_t)pio.pt);
        SqChainStem *const cst = pio.stems + toff;
#ifdef SQ_PR_PROF
        { int i0_ = i0 + len + (int)maxstems; asm volatile("" :: "v"(i0_)); } PRPROF(6);
#endif
        const bool anyc = sq_extend_structure<true>(XL, a, pio.stems + prec.toff, prec.nstems, prec.anycross != 0, pio.strands + pst.strand_off,
                                                    pio.sidx + pst.strand_off, pst.nstrand, i0, j0, len, cst, s_str, s_sidx, lane, &pre);
        __syncthreads();
        PRPROF(7);
        if (ROOT) { za0 = i0; za1 = i0 + len - 1; zb0 = j0 - len + 1; zb1 = j0; regroup = XL.cc[prec.nstems] != 0; }
        nstems = prec.nstems + 1; nstrand = pst.nstrand + 2;
        const bool full = (double)nstems == maxstems;
        if (lane == 0) {
            SqStruct cs;
            cs.job = job; cs.strand_off = 2 * toff; cs.nstrand = full ? -1 : nstrand; cs.slot = s;
            cs.subopt = cursub;
            cs.cand_off = (int64_t)(s % pio.chunk) * pio.maxcap;
            pio.structs[cur + s] = cs;
            SqChain cr;
            cr.toff = toff; cr.tcap = pio.pt; cr.nstems = nstems; cr.anycross = anyc ? 1 : 0; cr.maxstems = maxstems;
            pio.recs[cur + s] = cr;
        }
        if (full) {                                          // :1123-1129: moved to finstemsets at the start of this round
            log_final(2u * round, nstems);
            if (lane == 0) { pio.nchild[s] = 0; pio.finalflag[s] = 0; }
            return;
        }
        for (int q = lane; q < nstrand; q += 64) {           // the lists its own children will start from
            pio.strands[2 * (size_t)toff + q] = s_str[q];
            pio.sidx[2 * (size_t)toff + q] = s_sidx[q];
        }
        __syncthreads();
    }
    PRPROF(0);
    if (lane == 0) atomicAdd((unsigned long long *)&J->evals, 1ull);
    const SqJob jb = c.jobs[job];                           // (vector loads, issued together; through the scalar cache: more instructions, no faster)
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n;
    SqStruct st;                                            // what the phases below read of a structure record
    st.job = job; st.slot = s; st.nstrand = nstrand; st.subopt = cursub; st.cand_off = (int64_t)(s % pio.chunk) * pio.maxcap; st.strand_off = 0;
#ifdef SQ_PR_DUP_SETUP
    { const SqCellEnv dup_ = sq_cell_setup<!ROOT>(c, jb, ps, s_ctmp, l_ci, l_code, s_cell, lane, 64, &cpre); (void)dup_; __syncthreads(); }
#endif
    const SqCellEnv cenv = sq_cell_setup<!ROOT>(c, jb, ps, s_ctmp, l_ci, l_code, s_cell, lane, 64, &cpre);
    PRPROF(1);
    bool from_parent = false;
    uint32_t R = 0, ptab[4] = {0u, 0u, 0u, 0u};
    const SqRun *root = nullptr;
    // kept lists (sq_pool_round.h): this structure's own list -- entries so far, pages taken (lane k holds the k-th page's number)
    const SqKept &K = ra.kept;
    const int gen = ra.parity;
    const size_t krow = cur + (size_t)s;
    bool mylist = ROOT && K.on != 0;
    uint32_t mycnt = 0, mypages = 0, mytab[4] = {0u, 0u, 0u, 0u};     // (page k: lane k & 63 of word k >> 6)
    // page number `page` of a table (every lane calls, each with a page of its own)
    auto tab_get = [&](const uint32_t (&t)[4], uint32_t page, uint32_t npg) -> uint32_t {
        const int l = (int)(page & 63u);
        uint32_t v = (uint32_t)__shfl((int)t[0], l, 64);
        if (npg > 64u) {                                                 // (wave-uniform: few lists are that long)
            const uint32_t b = (uint32_t)__shfl((int)t[1], l, 64), c2 = (uint32_t)__shfl((int)t[2], l, 64), d = (uint32_t)__shfl((int)t[3], l, 64);
            const uint32_t h = page >> 6;
            v = h == 0u ? v : h == 1u ? b : h == 2u ? c2 : d;
        }
        return v;
    };
    const unsigned long long below = (1ull << lane) - 1ull;
    const double qnan = __longlong_as_double(0x7FF8000000000000ll);
    // (list form) the source -- the list the PARENT left, or without one the job's root list -- and its first page, asked for HERE:
    // the three dependent trips to memory (count, page numbers, page) overlap with the building of the structure's state
    if (ROOT && K.on && round) {
        const uint32_t pc = sq_kload(K.cnt + prv + (size_t)p);
        if (pc != SQ_KEPT_NOLIST) {
            from_parent = true; R = pc;
            const uint32_t *const prow_tab = K.tab + (prv + (size_t)p) * SQ_KEPT_TAB;
            const uint32_t npg = (pc + SQ_KEPT_PG - 1u) / SQ_KEPT_PG;
#pragma unroll
            for (int h = 0; h < 4; h++) if ((uint32_t)(64 * h) < npg) ptab[h] = prow_tab[64 * h + lane];
        }
    }
    if (ROOT) {
        root = reinterpret_cast<const SqRun *>(a.cands + ra.root_off + (int64_t)pio.jobrec_of[job] * ra.root_units);
        if (!from_parent) R = a.cand_cnt[pio.jobrec_of[job]];
    }
    // the source comes in by the page: four entries per lane -- sixteen loads -- at once
    // (an entry per lane and step, the next step's on their way, left every step waiting for a trip to memory)
    auto load_page = [&](uint32_t pk) -> SqEnt4 {
        SqEnt4 e;
#pragma unroll
        for (int t = 0; t < 4; t++) { e.key[t] = 0u; e.lf[t] = 0u; e.bps[t] = 0.0; e.fin[t] = 0.0; }
        if (from_parent) {
            const uint32_t ph = pk >> 6;                             // (wave-uniform)
            const uint32_t pw = ph == 0u ? ptab[0] : ph == 1u ? ptab[1] : ph == 2u ? ptab[2] : ptab[3];
            const uint32_t pid = (uint32_t)__builtin_amdgcn_readlane((int)pw, (int)(pk & 63u));
            const SqKeptPage pg = sq_kept_page(K, gen ^ 1, pid);
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t o = 64u * t + (uint32_t)lane;
                if (pk * SQ_KEPT_PG + o < R) { e.key[t] = pg.key[o]; e.lf[t] = pg.lf[o]; e.bps[t] = pg.bps[o]; e.fin[t] = pg.fin[o]; }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t q = pk * SQ_KEPT_PG + 64u * t + (uint32_t)lane;
                if (q < R) { const SqRun r = root[q]; e.key[t] = r.key; e.lf[t] = r.len; e.bps[t] = r.bps; }
            }
        }
        return e;
    };
    SqEnt4 cur4;
#pragma unroll
    for (int t = 0; t < 4; t++) { cur4.key[t] = 0u; cur4.lf[t] = 0u; cur4.bps[t] = 0.0; cur4.fin[t] = 0.0; }
    if (ROOT && n >= 5) cur4 = load_page(0u);

    // ---- the structure's state (sq_state_build): partner array, mask codes, prefix counts, free-position words ----
#ifdef SQ_PR_DUP_STATE
    for (int dup_ = 0; dup_ < 2; dup_++)
#endif
    {
        for (int p = lane; p < n; p += 64) { P[p] = -1; E[p] = ROOT ? c.e0c[jb.pos_off + p] : (uint8_t)(e4 >> (8 * (p >> 6))); }   // (c.e0c: asked for at the entry)
        __syncthreads();
        for (int k = lane; k < st.nstrand; k += 64) {
            const SqStrand x = s_str[k];
            for (int t = 0; t < x.len; t++) {
                const int pos = x.start + t;
                P[pos] = (int16_t)(x.pstart - t);           // :634-635
                E[pos] = 255;                               // :446-451 row + column of a paired base are masked
            }
        }
        __syncthreads();
        int base_u = 0, base_s = 0;
        for (int p0 = 0; p0 < n; p0 += 64) {
            const int p = p0 + lane;
            const bool un = p < n && P[p] == -1;
            const bool us = un && (l_code[p] == 26 || l_code[p] == 27);
            const unsigned long long mu = __ballot(un), ms = __ballot(us);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (p < n) { U[p] = (int16_t)(base_u + __popcll(mu & below)); SU[p] = (int16_t)(base_s + __popcll(ms & below)); }
            base_u += __popcll(mu); base_s += __popcll(ms);
        }
        if (lane == 0) { U[n] = (int16_t)base_u; SU[n] = (int16_t)base_s; }
        if (ROOT) {
            // the root-list pass: one bit per position, set while it is unpaired (forward order; two zero words behind the last)
            const int nwb = ((n + 31) >> 5) + 2;
            for (int m2 = 0; 2 * m2 < nwb; m2++) {
                const int pf = 64 * m2 + lane;
                const unsigned long long bf = __ballot(pf < n && P[pf] == -1);
                if (lane == 0) { FG[2 * m2] = (uint32_t)bf; FG[2 * m2 + 1] = (uint32_t)(bf >> 32); }
            }
        }
        const int fbh = ROOT ? 0 : Lo.fbh;                   // (free-position words: the scan's)
        for (int m2 = 0; 2 * m2 < fbh; m2++) {
            const int pf = 64 * m2 + lane;
            const unsigned long long bf = __ballot(pf < n && E[pf] == 0);
            const int pr = n - 1 - (64 * m2 + lane - SQ_GPAD);
            const unsigned long long br = __ballot(pr >= 0 && pr < n && E[pr] == 0);
            if (lane == 0) {
                FG[2 * m2] = (uint32_t)bf; FG[fbh + 2 * m2] = (uint32_t)br;
                if (2 * m2 + 1 < fbh) { FG[2 * m2 + 1] = (uint32_t)(bf >> 32); FG[fbh + 2 * m2 + 1] = (uint32_t)(br >> 32); }
            }
        }
        // skip pointers over the blocks ScoreStems' sweep registers (sq_score_kernel)
        for (int k = lane; k < st.nstrand; k += 64) {
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
        __syncthreads();
    }

    PRPROF(2);
    // ---- AnnotateStems + :492 ----
    SqPrSurv sv;
    sv.bps = reinterpret_cast<double *>(pr_dyn + Lo.off_surv); sv.fin = sv.bps + ra.surv_cap;
    sv.key = reinterpret_cast<uint32_t *>(sv.fin + ra.surv_cap); sv.len = reinterpret_cast<uint16_t *>(sv.key + ra.surv_cap);
    sv.place = sv.len + ra.surv_cap;
    sv.cap = ra.surv_cap; sv.ctr = a.ctr; sv.pool_ovf = &pio.hdr->ovf;
    // the structure's slice of the arena (cand_cap 32-byte units): runs the staging buffer could not take, then spilled survivors
    uint2 *const over = reinterpret_cast<uint2 *>(a.cands + st.cand_off);
    sv.spill = sq_oks(a, st, jb.cand_cap);
    sv.spill_cap = (uint32_t)(((size_t)jb.cand_cap * (sizeof(SqCand) - sizeof(SqKey))) / sizeof(SqOk));
    if (ROOT && ra.kept.on) {          // (no candidates in the arena: the whole region -- SqPoolIO::maxcap units -- takes survivors)
        sv.spill = reinterpret_cast<SqOk *>(a.cands + st.cand_off);
        sv.spill_cap = (uint32_t)(((size_t)pio.maxcap * sizeof(SqCand)) / sizeof(SqOk));
    }
    const double minbps = ps->minbpscore, minfin = ps->minfinscore;
    uint32_t ns = 0;
    double best = 0.0; bool anybest = false;                           // the best finalscore (wave-uniform)
    if (ROOT && n >= 5) {
        // ---- AnnotateStems, :492 and ScoreStems as ONE pass over a list.  Choosing stems only ever masks rows and columns
        // (:446-451), so the maximal runs of this structure are the pieces its paired positions leave of its parent's runs -- and
        // of the empty structure's (sq_rounds.hip keeps a chain's list on the same rule).  The source is the list the PARENT left
        // (kept lists, sq_device.h: its live runs with their exact bpscores and the finalscores it knew) or, without one, the
        // job's root list.  A run whose rows and columns are all unpaired -- four reads of the prefix counts -- stands as it is; a
        // cut run is walked cell by cell against the partner array and its pieces of minlen cells and more are summed anew.  A
        // finalscore of the parent's stands while no strand of the new stem comes within six positions of the run's span and the
        // levels were not taken anew under a finalscore that read them; a run without one goes through ScoreStems only when its
        // bound (sq_run_upper) reaches the range under the best finalscore so far.  The rare steps -- cut runs, walks -- wait in
        // LDS until a whole wave of them is there.  What the rest of the round reads (ChooseStems) is the handful of runs
        // whose finalscore is within range of the best one met before them: only those join the survivors.
        const int minlen = max(1, (int)ceil(ps->minlen));
        const double ps_lb = ps->loopbonus;
        const double ub_of = ps->ub_of, ub_lf = ra.bound ? ps->ub_lf : INFINITY;
        const SqStemsEnv env = {s_str, s_skip, st.nstrand, true, P, U, SU, l_code, n, false, nullptr, nullptr, nullptr, 0,
                                ps_lb, ps->bracketweight, ps->distcoef, ps->bw_integral, ps->sdf_len, c.sdftab + ps->sdf_off, ps->oftab, a.ctr};
        const double subopt = st.subopt;
        // the survivors' room: the runs within range (SqPrSurv, a shorter list), behind them the walks' queue
        struct WQ { uint32_t key, lp; double bps; };                              // lp: length | place in this structure's list << 16
        sv.cap = ra.surv_cap - 88;
        sv.bps = reinterpret_cast<double *>(pr_dyn + Lo.off_surv); sv.fin = sv.bps + sv.cap;
        sv.key = reinterpret_cast<uint32_t *>(sv.fin + sv.cap); sv.len = reinterpret_cast<uint16_t *>(sv.key + sv.cap);
        sv.place = sv.len + sv.cap;
        WQ *const wq = reinterpret_cast<WQ *>(pr_dyn + Lo.off_surv + (((size_t)24 * sv.cap + 15) & ~(size_t)15));   // 128 entries
        uint32_t nwq = 0;
        // the pages this structure's list will need, in one go (it is at most as long as its source, pieces of cut runs aside: those
        // take further pages one by one) -- a returning atomic per page was a trip to memory every fourth step
        if (mylist && R > 0u) {
            const uint32_t need = min((R + SQ_KEPT_PG - 1u) / SQ_KEPT_PG, (uint32_t)SQ_KEPT_TAB);
            uint32_t id0 = 0u;
#ifdef SQ_PR_PROF
            const long long _a0 = wall_clock64();
#endif
            if (lane == 0) id0 = atomicAdd(K.ctr + gen, need);
            id0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)id0);
#ifdef SQ_PR_PROF
            _xa += wall_clock64() - _a0;
#endif
            if (id0 + need > K.npages) mylist = false;
            else {
#pragma unroll
                for (int h = 0; h < 4; h++)
                    if ((uint32_t)(64 * h + lane) < need) { mytab[h] = id0 + (uint32_t)(64 * h + lane); K.tab[krow * SQ_KEPT_TAB + 64 * h + lane] = mytab[h]; }
                mypages = need;
            }
        }
        // a run joins this structure's list (every lane calls; returns the run's place)
        auto append = [&](bool add, uint32_t key, uint32_t lf, double bps, double fin) -> uint32_t {
            const unsigned long long m = __ballot(add);
            if (m == 0ull || !mylist) return 0u;
            const uint32_t tot = (uint32_t)__popcll(m), pos = mycnt + (uint32_t)__popcll(m & below);
            while (mypages * SQ_KEPT_PG < mycnt + tot) {
                uint32_t id = 0u;
                if (lane == 0) id = atomicAdd(K.ctr + gen, 1u);
                id = (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
                if (id >= K.npages || mypages >= SQ_KEPT_TAB) { mylist = false; break; }
#pragma unroll
                for (int h = 0; h < 4; h++) if ((int)mypages == 64 * h + lane) mytab[h] = id;
                if (lane == 0) K.tab[krow * SQ_KEPT_TAB + mypages] = id;
                mypages++;
            }
            if (!mylist) return 0u;
            const uint32_t pid = tab_get(mytab, pos / SQ_KEPT_PG, mypages);
            if (add) {
                const SqKeptPage pg = sq_kept_page(K, gen, pid);
                const uint32_t o = pos % SQ_KEPT_PG;
                pg.key[o] = key; pg.lf[o] = lf; pg.bps[o] = bps; pg.fin[o] = fin;
            }
            mycnt += tot;
            return pos;
        };
        // an exact finalscore (before :751): it raises the bar, and joins the survivors when it is within range of the best one
        // met so far -- a run that is not never will be (the best only grows; every lane calls)
        auto consider = [&](bool has, double f, uint32_t key, int L, double bps, uint32_t place) {
            const bool cand = has && f >= minfin;                                 // :751
            const double wb = sq_wave_max_f64(cand ? f : -INFINITY);
            if (!(wb > -INFINITY)) return;
            const bool ins = cand && !(anybest && f < subopt * best);
            const unsigned long long m = __ballot(ins);
            if (ins) sv.put_kept(ns + (uint32_t)__popcll(m & below), key, L, bps, f, place);
            ns += (uint32_t)__popcll(m);
            if (!anybest || wb > best) { anybest = true; best = wb; }
        };
        // a run that passed :492 without a finalscore: behind its bound it waits for ScoreStems (every lane calls)
        auto enqueue = [&](bool has, uint32_t key, int L, double bps, uint32_t place) {
            bool want = has;
            if (want) {
                const int i0 = (int)(key & 0xFFFFu), j0 = (int)(key >> 16) - i0;
                const double ub = sq_run_upper(bps, i0, j0, L, U, l_code, n, ub_of, ub_lf, ps_lb);
                if (ub < minfin || (anybest && ub < subopt * best)) want = false;    // no reader of the list can use it
            }
            const unsigned long long m = __ballot(want);
            if (want) wq[nwq + (uint32_t)__popcll(m & below)] = WQ{key, (uint32_t)L | (place << 16), bps};
            nwq += (uint32_t)__popcll(m);
        };
        auto serve_walks = [&](bool all) {
#ifdef SQ_PR_PROF
            const long long _w0 = wall_clock64();
#endif
            while (nwq >= 64u || (all && nwq > 0u)) {
#ifdef SQ_PR_PROF
                _nwserve++;
#endif
                __syncthreads();
                const uint32_t take = nwq < 64u ? nwq : 64u;
                const bool mine = (uint32_t)lane < take;
                const WQ e = mine ? wq[nwq - take + (uint32_t)lane] : WQ{0u, 0u, 0.0};
                nwq -= take;
                __syncthreads();
                const int L = (int)(e.lp & 0xFFFFu), i0 = (int)(e.key & 0xFFFFu), j0 = (int)(e.key >> 16) - i0;
                const uint32_t place = e.lp >> 16;
                const uint32_t wpid = tab_get(mytab, place / SQ_KEPT_PG, mypages);
                bool ok = mine;
                if (ok) {                                                        // (the bar has risen since the run joined the queue)
                    const double ub = sq_run_upper(e.bps, i0, j0, L, U, l_code, n, ub_of, ub_lf, ps_lb);
                    if (ub < minfin || (anybest && ub < subopt * best)) ok = false;
                }
#ifdef SQ_PR_PROF
                _nwalk += __popcll(__ballot(ok));
#endif
                double f = 0.0;
                if (ok) {
                    const SqWalk w = sq_stem_walk(env, i0, j0, L);
                    f = sq_stem_finalscore_of(env, i0, j0, L, e.bps, w);
                    if (mylist) {                                                // (for this structure's children)
                        const SqKeptPage pg = sq_kept_page(K, gen, wpid);
                        const uint32_t o = place % SQ_KEPT_PG;
                        pg.fin[o] = f; pg.lf[o] = (uint32_t)L | SQ_RX_FIN | (w.brackets != 0 ? SQ_RX_LVL : 0u);
                    }
                }
                consider(ok, f, e.key, L, e.bps, place);
            }
#ifdef SQ_PR_PROF
            _xw += wall_clock64() - _w0;
#endif
        };
        // the cut runs (key, length), 64 at a time: every lane walks ONE run cell by cell against the partner array; its pieces of
        // minlen cells and more are summed anew, one piece per lane and step of the wave (most runs leave one or two).  Taken
        // one by one where the stream met them, a step of the wave served a single lane
        uint2 *const cutq = s_stage;                                              // (SQ_PR_STAGE >= 128 entries: at most 63 wait when 64 more arrive)
        uint32_t ncq = 0;
        auto serve_cuts = [&](bool all) {
#ifdef SQ_PR_PROF
            const long long _c0 = wall_clock64(), _w1 = _xw;
#endif
            while (ncq >= 64u || (all && ncq > 0u)) {
                __syncthreads();
                const uint32_t take = ncq < 64u ? ncq : 64u;
                bool cut = (uint32_t)lane < take;
                const uint2 e = cut ? cutq[ncq - take + (uint32_t)lane] : make_uint2(0u, 0u);
                ncq -= take;
                __syncthreads();
                const int L = (int)e.y, i = (int)(e.x & 0xFFFFu), j = (int)(e.x >> 16) - i;
                int t0 = 0;
                // (runs of up to 32 cells -- nearly all -- as a word of live cells: the rows' window of the unpaired bits AND the
                // bit-reversed window of the columns'; the pieces then come out of the word without another read)
                uint32_t alive = 0u;
                const bool word = L <= 32;
                if (cut && word) {
                    const uint32_t rw = __builtin_amdgcn_alignbit(FG[(i >> 5) + 1], FG[i >> 5], (uint32_t)(i & 31));        // bit t: row i + t
                    const int q = j - 31;                                                                                       // bit k of the window: column q + k
                    const uint32_t cw = q >= 0 ? __builtin_amdgcn_alignbit(FG[(q >> 5) + 1], FG[q >> 5], (uint32_t)(q & 31)) : FG[0] << (uint32_t)(-q);
                    alive = rw & __brev(cw) & (L == 32 ? 0xFFFFFFFFu : ((1u << L) - 1u));                                       // bit t: column j - t
                }
                while (__ballot(cut) != 0ull) {
                    int pb = -1, pl = 0;
                    if (cut && word) {
                        while (alive) {
                            const int b0 = __ffs((int)alive) - 1;
                            const uint32_t up = ~(alive >> b0);
                            const int len = up ? __ffs((int)up) - 1 : 32 - b0;
                            alive &= len + b0 >= 32 ? 0u : ~((1u << (len + b0)) - 1u);
                            if (len >= minlen) { pb = b0; pl = len; break; }
                        }
                        if (pb < 0) cut = false;
                    } else if (cut) {
                        int t = t0;
                        while (t < L) {
                            while (t < L && !(P[i + t] == -1 && P[j - t] == -1)) t++;
                            const int b0 = t;
                            while (t < L && P[i + t] == -1 && P[j - t] == -1) t++;
                            if (t - b0 >= minlen) { pb = b0; pl = t - b0; break; }
                        }
                        t0 = t;
                        if (pb < 0) cut = false;
                    }
                    double bps = 0.0, pos = 0.0;
                    if (pb >= 0) bps = sq_cellrun_bps(cenv, c, jb, i + pb, j - pb, pl, pos);
                    const uint32_t pkey = ((uint32_t)(i + j) << 16) | (uint32_t)(i + pb);
                    // (a piece whose positive parts miss :492 has no piece of its own that passes: it leaves the list)
                    const uint32_t pplace = append(pb >= 0 && !(pos < minbps), pkey, (uint32_t)pl, bps, qnan);
                    enqueue(pb >= 0 && bps >= minbps, pkey, pl, bps, pplace);       // :492
                    if (nwq >= 64u) serve_walks(false);
                }
            }
#ifdef SQ_PR_PROF
            _xc += (wall_clock64() - _c0) - (_xw - _w1);
#endif
        };
#define SQ_OV(z0, z1, lo, hi) (((hi) - (z0)) | ((z1) - (lo)))
        for (uint32_t b0 = 0; b0 < R; b0 += SQ_KEPT_PG) {                      // (the first page: asked for before the structure's state was built)
            for (uint32_t q0 = b0; q0 < b0 + SQ_KEPT_PG && q0 < R; q0 += 64) {
                const uint32_t rkey = cur4.key[0], rlf = cur4.lf[0]; const double rbps = cur4.bps[0], rfin = cur4.fin[0];
#pragma unroll
                for (int t = 0; t < 3; t++) { cur4.key[t] = cur4.key[t + 1]; cur4.lf[t] = cur4.lf[t + 1]; cur4.bps[t] = cur4.bps[t + 1]; cur4.fin[t] = cur4.fin[t + 1]; }
                // (ONE page in registers: the next one is asked for when this one's last group has been taken out -- a second page
                // under way cost 24 registers, and at 168 the kernel runs three waves per SIMD: 332 -> 305 ms per 1,000 records)
                if (q0 + 64 >= b0 + SQ_KEPT_PG && b0 + SQ_KEPT_PG < R) cur4 = load_page(b0 / SQ_KEPT_PG + 1u);
                const bool have = q0 + (uint32_t)lane < R;
                const int L = (int)(rlf & SQ_RX_LEN), i = (int)(rkey & 0xFFFFu), j = (int)(rkey >> 16) - i;
                const bool whole = have && ((int)U[i + L] - (int)U[i]) == L && ((int)U[j + 1] - (int)U[j - L + 1]) == L;
                const int nofd = SQ_OV(za0, za1, i - 6, j + 6) & SQ_OV(zb0, zb1, i - 6, j + 6);       // (< 0: neither strand meets [i - 6, j + 6])
                const bool finok = from_parent && (rlf & SQ_RX_FIN) != 0u && nofd < 0 && !(regroup && (rlf & SQ_RX_LVL) != 0u);
                const uint32_t wplace = append(whole, rkey, finok ? rlf : (uint32_t)L, rbps, finok ? rfin : qnan);
                const bool p492 = whole && rbps >= minbps;                           // :492
#ifdef SQ_PR_PROF
                _ninh += __popcll(__ballot(p492 && finok));
#endif
                consider(p492 && finok, rfin, rkey, L, rbps, wplace);
                enqueue(p492 && !finok, rkey, L, rbps, wplace);
                // the cut runs wait in LDS until 64 of them are there (a run of exactly minlen cells that lost one is gone)
                const bool cutrun = have && !whole && L > minlen;
                const unsigned long long m = __ballot(cutrun);
#ifdef SQ_PR_PROF
                _ncutrun += __popcll(m);
#endif
                if (cutrun) cutq[ncq + (uint32_t)__popcll(m & below)] = make_uint2(rkey, (uint32_t)L);
                ncq += (uint32_t)__popcll(m);
                if (ncq >= 64u) serve_cuts(false);
                if (nwq >= 64u) serve_walks(false);
            }
        }
        serve_cuts(true);
        serve_walks(true);
#undef SQ_OV
        if (K.on && lane == 0) {
            K.cnt[krow] = mylist ? mycnt : SQ_KEPT_NOLIST;
            if (!mylist) atomicAdd(K.ctr + 3, 1u);
        }
    } else if (n >= 5) {                                                // :456-457 (shorter sequences have no diagonals)
#ifdef SQ_PR_DUP_SCAN
        { SqPrNullSink nul_{0u}; sq_scan6_groups(c, jb, FG, FG + Lo.fbh, Lo.fbh, E, 0, 1, lane, nul_, SqBitsGlobal{c.bits + jb.bits_off, jb.bpitch});
          if (nul_.n == 0xFFFFFFFFu) a.ctr->cand_ovf = 1; }
#endif
        SqPrSink sink{s_stage, over, (uint32_t)jb.cand_cap, cenv, c, jb, sv, ns, minbps, a.ctr, 0u, 0u};
        sq_scan6_groups(c, jb, FG, FG + Lo.fbh, Lo.fbh, E, 0, 1, lane, sink, SqBitsGlobal{c.bits + jb.bits_off, jb.bpitch});
        __threadfence_block();
        __syncthreads();
        const uint32_t no = sink.nover;
        for (uint32_t b0 = 0; b0 < no; b0 += 64) sink.score64(over + b0, min(no - b0, 64u), lane);
    }
    __threadfence_block();
    __syncthreads();

    PRPROF(3);
    // ---- ScoreStems on the survivors, finalscores beside them; the best one ----
    const double ps_lb = ps->loopbonus;
    const double ub_of = ps->ub_of, ub_lf = ra.bound ? ps->ub_lf : INFINITY;
    const SqStemsEnv env = {s_str, s_skip, st.nstrand, true, P, U, SU, l_code, n, false, nullptr, nullptr, nullptr, 0,
                            ps_lb, ps->bracketweight, ps->distcoef, ps->bw_integral, ps->sdf_len, c.sdftab + ps->sdf_off, ps->oftab, a.ctr};
    const double subopt = st.subopt;
#ifdef SQ_PR_DUP_SCORE
    for (int dup_ = 0; dup_ < 2; dup_++) { best = 0.0; anybest = false;
#endif
    if (!ROOT)                                                          // (the list form scored while it streamed)
    for (uint32_t g = 0; g < ns; g += 64) {
        const uint32_t idx = g + lane;
        const bool have = idx < ns;
        uint32_t key = 0; int L = 0; double bps = 0.0;
        if (have) sv.get(idx, key, L, bps);
        const int i0 = (int)(key & 0xFFFFu), j0 = (int)(key >> 16) - i0;
        bool ok = have;
        if (ok) {
            const double ub = sq_run_upper(bps, i0, j0, L, U, l_code, n, ub_of, ub_lf, ps_lb);
            if (ub < minfin || (anybest && ub < subopt * best)) ok = false;      // no reader of the list can use it
        }
        double fin = -INFINITY;
        if (ok) {
            fin = sq_stem_finalscore(env, i0, j0, L, bps);
            if (!(fin >= minfin)) fin = -INFINITY;                               // :751
        }
        if (have) sv.set_fin(idx, fin);
        const double wb = sq_wave_max_f64(fin);
        if (wb > -INFINITY && (!anybest || wb > best)) { anybest = true; best = wb; }
    }
#ifdef SQ_PR_DUP_SCORE
    __syncthreads(); }
#endif
    __threadfence_block();
    __syncthreads();
    PRPROF(4);
    if (!anybest) {                                         // no stem passed the thresholds: the structure is final (:1155-1156)
        log_final(2u * round + 1u, nstems);
        if (lane == 0) { pio.nchild[s] = 0; pio.finalflag[s] = 0; }
        return;
    }

    // ---- ChooseStems (sq_pool_choose_kernel's steps) ----
    SqPoolPick *out = pio.chosen + (cur + (size_t)s) * pio.cmax;
    const bool one = cursize >= pio.poollim;             // :1147 stopper
    if (one) {
        // only ChooseStems' first element is used: the highest finalscore, the smallest emission key among equals
        unsigned long long pick = ~0ull;
        for (uint32_t q = lane; q < ns; q += 64)
            if (sv.get_fin(q) == best) {
                uint32_t key; int L; double bps;
                sv.get(q, key, L, bps);
                const unsigned long long v = ((unsigned long long)key << 32) | q; pick = v < pick ? v : pick;
            }
        pick = sq_wave_min64(pick);
        if (lane == 0) {
            uint32_t key; int L; double bps;
            sv.get((uint32_t)pick, key, L, bps);
            out[0] = SqPoolPick{key, (uint32_t)L, bps, best};
            pio.nchild[s] = 1; pio.finalflag[s] = 0;
        }
        return;
    }
    // the candidates within range (:769-778), gathered over the arrays the earlier phases are done with
    const double range = cursub * best;
    double *const c_fin = reinterpret_cast<double *>(pr_dyn);
    uint32_t *const c_key = reinterpret_cast<uint32_t *>(c_fin + Lo.choose_cap);
    uint16_t *const c_q = reinterpret_cast<uint16_t *>(c_key + Lo.choose_cap), *const c_ord = c_q + Lo.choose_cap;
    int nin = 0, nres = 0;
    bool over_c = false;
#ifdef SQ_PR_DUP_CHOOSE
    for (int dup_ = 0; dup_ < 2; dup_++) { nin = 0; nres = 0; over_c = false; __syncthreads();
#endif
    for (uint32_t g = 0; g < ns; g += 64) {
        const uint32_t q = g + lane;
        const double f = q < ns ? sv.get_fin(q) : -INFINITY;
        const bool keep = !(f < range);                     // (-inf: rejected or pruned)
        uint32_t key = 0; int L = 0; double bps = 0.0;
        if (keep) sv.get(q, key, L, bps);
        const unsigned long long m = __ballot(keep);
        const int pos = nin + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < Lo.choose_cap) { c_q[pos] = (uint16_t)q; c_fin[pos] = f; c_key[pos] = key; }
        nin += __popcll(m);
    }
    if (nin > Lo.choose_cap || ns > 65535u) {
        if (lane == 0) { pio.hdr->ovf = 1; pio.nchild[s] = 0; pio.finalflag[s] = 0; }
        return;
    }
    __syncthreads();
    // the reference's stable descending sort: finalscore descending, emission order (key) ascending
    for (int x = lane; x < nin; x += 64) {
        const double fx = c_fin[x]; const uint32_t kx = c_key[x];
        int r = 0;
        for (int y = 0; y < nin; y++) { const double fy = c_fin[y]; r += (fy > fx || (fy == fx && c_key[y] < kx)) ? 1 : 0; }
        c_ord[r] = (uint16_t)x;
    }
    __syncthreads();
    // the conflict filter (:779-789): a candidate joins when it shares a base with every stem taken so far
    for (int t = 0; t < nin; t++) {
        const int x = c_ord[t];
        uint32_t key; int cl; double bps;
        sv.get(c_q[x], key, cl, bps);
        const int ci = (int)(key & 0xFFFFu), cj = (int)(key >> 16) - ci;
        const bool mine = lane < nres ? sq_pr_shares_base(ci, cj, cl, s_ri[lane], s_rj[lane], s_rl[lane]) : true;
        if (__ballot(!mine) != 0ull) continue;
        if (nres == SQ_POOL_CMAX || nres == pio.cmax) { over_c = true; break; }
        if (lane == 0) {
            s_ri[nres] = ci; s_rj[nres] = cj; s_rl[nres] = cl;
            out[nres] = SqPoolPick{key, (uint32_t)cl, bps, c_fin[x]};
        }
        nres++;
        __syncthreads();
    }
#ifdef SQ_PR_DUP_CHOOSE
    }
#endif
    PRPROF(5);
    PRPROF_OUT(ns, nin);
    if (over_c) { if (lane == 0) pio.hdr->ovf = 1; nres = 0; }
    else if (nres == 0) log_final(2u * round + 1u, nstems);
    if (lane == 0) { pio.nchild[s] = nres; pio.finalflag[s] = 0; }
}


extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SQ_PR_WAVES, SQ_PR_WAVES))) void sq_pool_round_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra)
{
    sq_pool_round_body<false>(c, a, pio, ra);
}
#ifndef SQ_PR_ROOT_WAVES
#define SQ_PR_ROOT_WAVES 3         // (the list form: 168 VGPRs, 16 spilled; at 128 it spilled 190 of them and every step of its stream waited for scratch -- loop of 500 x 500 nt 153 -> 125 ms; 2 waves, no spill: 8 % slower than 3)
#endif
extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SQ_PR_ROOT_WAVES, SQ_PR_ROOT_WAVES))) void sq_pool_round_root_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra)
{
    sq_pool_round_body<true>(c, a, pio, ra);
}

// The root lists: per job of the pools the runs of its EMPTY structure -- AnnotateStems of round 0 (bit-diagonal scan) -- with
// their exact bpscores, the ones whose positive cell parts cannot reach :492 dropped (no piece of them ever passes, sq_cellrun.h).
// One wave per job, once per fold; every structure of the job's pool then reads the list instead of scanning (SqPoolRoundArgs::root).
#define SQ_PR_ROOT_STAGE 1088          // runs the root kernel stages in LDS (a word-row of a wave emits at most 64 x 16)
struct SqPrRootSink {
    uint2 *stage;                                     // SQ_PR_ROOT_STAGE entries (LDS)
    SqRun *root; uint32_t cap;
    const SqCellEnv &cenv; const SqDevCtx &c; const SqJob &jb; double minbps; SqCounters *ctr;
    uint32_t n, cnt;                                  // runs staged, runs written
    __device__ __forceinline__ uint32_t reserve(uint32_t total, int lane)
    {
        if (n + total > SQ_PR_ROOT_STAGE) flush(lane);
        const uint32_t b0 = n; n += total; return b0;
    }
    __device__ __forceinline__ void put(uint32_t at, uint32_t key, uint32_t len) { if (at < SQ_PR_ROOT_STAGE) stage[at] = make_uint2(key, len); else ctr->cand_ovf = 1; }
    __device__ __forceinline__ void flush(int lane)
    {
        __syncthreads();
        for (uint32_t b0 = 0; b0 < n; b0 += 64) {
            const bool have = b0 + (uint32_t)lane < n;
            const uint2 kl = have ? stage[b0 + lane] : make_uint2(0u, 0u);
            double bps = 0.0, pos = 0.0;
            if (have) { const int i = (int)(kl.x & 0xFFFFu), j = (int)(kl.x >> 16) - i; bps = sq_cellrun_bps(cenv, c, jb, i, j, (int)kl.y, pos); }
            const bool keep = have && !(pos < minbps);             // (a run whose positive parts miss :492 has no piece that passes)
            const unsigned long long m = __ballot(keep);
            const uint32_t at = cnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (keep) { if (at < cap) root[at] = SqRun{kl.x, kl.y, bps}; else ctr->cand_ovf = 1; }
            cnt += (uint32_t)__popcll(m);
        }
        n = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void poll(int lane) { if (n >= 256u) flush(lane); }
    __device__ __forceinline__ void drain(int lane) { flush(lane); }
};

extern "C" __global__ __launch_bounds__(64) void sq_pool_root_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra)
{
    extern __shared__ __attribute__((aligned(16))) char pr_dyn[];
    __shared__ SqCellTmp s_ctmp;
    const int lane = threadIdx.x;
    const int sx = ra.lo + (int)blockIdx.x;                 // the job's record == the slot of its empty structure in generation 0
    const int job = pio.structs[sx].job;
    const SqJob jb = c.jobs[job];
    const SqPsetDev *ps = c.psets + jb.pset;
    const int n = jb.n;
    // LDS: mask codes, class indices, free-position words, the cell table, the staging buffer
    const int np = (ra.lds_n + 8) & ~7, fbh = ((ra.lds_n + 2 + 31) >> 5) + 8;
    uint8_t *const E = reinterpret_cast<uint8_t *>(pr_dyn);
    uint8_t *const l_ci = E + np;
    uint32_t *const FG = reinterpret_cast<uint32_t *>(l_ci + np);
    double *const s_cell = reinterpret_cast<double *>(reinterpret_cast<char *>(FG) + ((8 * fbh + 15) & ~15));
    uint2 *const stage = reinterpret_cast<uint2 *>(s_cell + ra.cell_entries);
    const SqCellEnv cenv = sq_cell_setup(c, jb, ps, s_ctmp, l_ci, nullptr, s_cell, lane, 64);
    {
        const uint8_t *e0 = c.e0c + jb.pos_off;
        for (int p = lane; p < n; p += 64) E[p] = e0[p];
        __syncthreads();
        for (int m2 = 0; 2 * m2 < fbh; m2++) {              // free-position words of the empty structure, forward and reversed
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
    }
    SqRun *const root = reinterpret_cast<SqRun *>(a.cands + ra.root_off + (int64_t)sx * ra.root_units);
    // the runs first go, as the scan finds them, to the empty structure's own region of the arena (nobody needs it before round
    // 0); from there into the root list ordered by descending bpscore (buckets of 0.5, as sq_rounds.hip orders a chain's list):
    // a structure's round then meets the strong runs first, its best finalscore early, and the bound prunes the rest of its
    // survivors (in scan order most of a 500-nt structure's 1,300 survivors went through ScoreStems)
    SqRun *const tmp = reinterpret_cast<SqRun *>(a.cands + pio.structs[sx].cand_off);
    const uint32_t cap = (uint32_t)min((long long)(2 * ra.root_units), (long long)pio.maxcap * 2);
    SqPrRootSink sink{stage, tmp, cap, cenv, c, jb, ps->minbpscore, a.ctr, 0u, 0u};
    if (n >= 5) sq_scan6_groups(c, jb, FG, FG + fbh, fbh, E, 0, 1, lane, sink, SqBitsGlobal{c.bits + jb.bits_off, jb.bpitch});
    const uint32_t R = sink.cnt < cap ? sink.cnt : cap;
    __threadfence_block();
    __syncthreads();
    uint32_t *const hist = reinterpret_cast<uint32_t *>(stage);          // [256] counts, then [256] fill pointers (the staging buffer is free)
    for (int k = lane; k < 512; k += 64) hist[k] = 0;
    __syncthreads();
    auto bucket = [&](double bps) -> int { const double x = bps * 2.0; return 255 - (x >= 255.0 ? 255 : (x > 0.0 ? (int)x : 0)); };
    for (uint32_t q = lane; q < R; q += 64) atomicAdd(&hist[bucket(tmp[q].bps)], 1u);
    __syncthreads();
    {
        uint32_t h[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { h[k] = hist[4 * lane + k]; tot += h[k]; }
        const uint32_t incl = (uint32_t)sq_wave_scan_add_i32((int)tot);
        uint32_t base = incl - tot;
#pragma unroll
        for (int k = 0; k < 4; k++) { hist[256 + 4 * lane + k] = base; base += h[k]; }
    }
    __syncthreads();
    for (uint32_t q = lane; q < R; q += 64) {
        const SqRun r = tmp[q];
        root[atomicAdd(&hist[256 + bucket(r.bps)], 1u)] = r;
    }
    if (lane == 0) a.cand_cnt[sx] = R;
}
