// sq_score.h -- ScoreStems (SQRNdbnseq.py:607-751) for ONE candidate stem, as a closed form over the structure's sorted
// strands: shared by the scoring kernel of the launched rounds (sq_kernels.hip) and the persistent round kernel
// (sq_rounds.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "sq_internal.h"
#include "sq_device.h"
#include "sq_context.h"

__device__ __forceinline__ bool sq_goodloop(int x, int y)             // :615-622
{
    // rows x = 0..4, bit y set when (x, y) is a "good" internal loop
    const unsigned tab[5] = {0x07u /*0:{0,1,2}*/, 0x0Fu /*1:{0,1,2,3}*/, 0x1Fu /*2:{0..4}*/, 0x1Eu /*3:{1,2,3,4}*/,
                             0x1Cu /*4:{2,3,4}*/};
    if ((unsigned)x > 4u || (unsigned)y > 4u) return false;
    return (tab[x] >> y) & 1u;
}

// grid = (structures, parts): the candidates of a structure are dealt to `parts` blocks; the round's best
// finalscore of the structure is combined with atomicMax, the range filter (:769-778) runs in sq_select_kernel.
// ScoreStems for one candidate stem (SQRNdbnseq.py:607-751): what the scoring kernels know about the structure ...
struct SqStemsEnv {
    const SqStrand *S; const uint16_t *skip; int nstrand; bool have_skip;   // sorted strands (+ skip pointers over registered blocks)
    const int16_t *P, *U, *SU; const uint8_t *codes; int n;                 // partner array, prefix counts, letter codes
    bool use_ctx; const SqCtxRec *ctx_rec; const int16_t *ctx_depth; const uint16_t *ctx_rmq; int ctx_cap;   // sq_context.h
    double lb, bw, dc; int bwint, sdflen; const double *sdf, *of;           // the paramset's scalars and tables
    SqCounters *ctr;
    const double *sdf_l; int sdf_llen;                                      // the first entries of sdf once more, in LDS (nullptr / 0: none)
};
// What the walk over the positions between the innermost pair finds (:665-689): the sub-ECR faces it registers (their count, the
// first one's ends), the positions they cover, the bracket positions outside them and the set of their levels.
struct SqWalk { int nrec, be0, be1, covered, brackets; uint64_t levelset; };

// The walk by ONE lane: a closed form over the sorted strands, skip pointers over the registered blocks.
__device__ __forceinline__ SqWalk sq_stem_walk(const SqStemsEnv &e, int i0, int j0, int L)
{
    const int sa = i0 + L - 1, sb = j0 - L + 1;                 // :655 innermost bp
    int inblockend = -1, nrec = 0, be0 = 0, be1 = 0, covered = 0, brackets = 0;
    uint64_t levelset = 0;
    int lo = 0, hi = e.nstrand;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (e.S[mid].start <= sa) lo = mid + 1; else hi = mid; }
    SqCtxOut cx = {0, 0, 0, 0, 0};
    if (e.use_ctx) {
        int lo2 = lo; hi = e.nstrand;                          // first strand that starts at or behind sb
        while (lo2 < hi) { const int mid = (lo2 + hi) >> 1; if (e.S[mid].start < sb) lo2 = mid + 1; else hi = mid; }
        if (lo2 > lo) sq_ctx_query(e.ctx_rec, e.ctx_depth, e.ctx_rmq, e.ctx_cap, e.S, lo, lo2, cx);
#ifndef SQ_CTX_CHECK
        nrec = cx.nrec; be0 = cx.be0; be1 = cx.be1; covered = cx.covered; brackets = cx.brackets;
        levelset = brackets > 0 ? 1ull : 0ull;                  // (every strand on level 1)
        lo = e.nstrand;                                        // the walk below has nothing left to do
#endif
    }
    for (int k = lo; k < e.nstrand;) {                         // closed form of the walk :665-689
        const SqStrand x = e.S[k];
        if (x.start >= sb) break;
        int nk = k + 1;
        const int pfirst = x.pstart, plast = x.pstart - (x.len - 1);
        bool wing;
        if (x.left) {
            wing = pfirst > sb;
            if (!wing && pfirst > inblockend) {                 // :687-689 sub-ECR face
                if (nrec == 0) { be0 = x.start; be1 = pfirst; }
                nrec++;
                const int from = x.start > inblockend ? x.start : inblockend + 1;
                covered += e.U[pfirst + 1] - e.U[from];
                inblockend = pfirst;
                if (e.have_skip) nk = e.skip[k];                // nothing inside the block can matter
            }
        } else {
            wing = plast < sa;
        }
        if (wing && x.start > inblockend) {                     // :679-684
            brackets += x.len;
            if (x.level > SQ_MAXLEVELS) e.ctr->level_ovf = 1;
            else levelset |= 1ull << (x.level - 1);
        }
        k = nk;
    }
#ifdef SQ_CTX_CHECK
    if (e.use_ctx && (cx.nrec != nrec || cx.covered != covered || cx.brackets != brackets ||
                    (nrec == 1 && (cx.be0 != be0 || cx.be1 != be1)) || (levelset != (brackets > 0 ? 1ull : 0ull))))
        printf("CTX MISMATCH struct %d cand (%d,%d,%d) nstrand %d: walk nrec %d cov %d br %d be %d %d | ctx nrec %d cov %d br %d be %d %d\n",
               (int)blockIdx.x, i0, j0, L, e.nstrand, nrec, covered, brackets, be0, be1, cx.nrec, cx.covered, cx.brackets, cx.be0, cx.be1);
#endif
    return SqWalk{nrec, be0, be1, covered, brackets, levelset};
}

// ---- the walk in parts (sq_rounds.hip): a lane takes it up to a budget of strands and may end it early; what is left of the
// few walks that outlast the others of their wave is taken by the whole wave, a strand per lane.
//
// Ending early: the bracket strands met so far show p levels, the rest of the walk can only add levels, and the paramset's
// order factors do not grow with the level count (the caller checks that once), so
//     finalscore <= bound x of[p] / of_max        (bound: sq_run_upper, which holds the order factor at its maximum)
// -- ofr[p] holds that ratio, rounded up -- and a run whose product misses the bar can neither be the round's best nor tie with it.
struct SqWalkPart {
    int k, inblockend, nrec, be0, be1, covered, brackets;
    uint32_t lv0, lv1;                       // the level set, two words
};
#define SQ_WALK_DONE 0
#define SQ_WALK_MORE 1                       // the budget is spent: the state says where the walk stands
#define SQ_WALK_OUT 2                        // ended early: *bnd holds the bound that missed the bar

__device__ __forceinline__ void sq_walk_begin(const SqStemsEnv &e, int sa, SqWalkPart &st)
{
    int lo = 0, hi = e.nstrand;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (e.S[mid].start <= sa) lo = mid + 1; else hi = mid; }
    st.k = lo; st.inblockend = -1; st.nrec = 0; st.be0 = 0; st.be1 = 0; st.covered = 0; st.brackets = 0; st.lv0 = 0u; st.lv1 = 0u;
}

// up to `budget` strands of the walk of the span (sa, sb) by ONE lane (skip pointers over the blocks it registers)
__device__ __forceinline__ int sq_walk_lane(const SqStemsEnv &e, int sa, int sb, SqWalkPart &st, int budget, bool early, double ub,
                                            const double *ofr, double need, double *bnd)
{
    int k = st.k, inblockend = st.inblockend;
    for (; k < e.nstrand && budget > 0; budget--) {
        const SqStrand x = e.S[k];
        if (x.start >= sb) { k = e.nstrand; break; }
        int nk = k + 1;
        const int pfirst = x.pstart, plast = x.pstart - (x.len - 1);
        bool wing;
        if (x.left) {
            wing = pfirst > sb;
            if (!wing && pfirst > inblockend) {                 // :687-689 sub-ECR face
                if (st.nrec == 0) { st.be0 = x.start; st.be1 = pfirst; }
                st.nrec++;
                const int from = x.start > inblockend ? x.start : inblockend + 1;
                st.covered += e.U[pfirst + 1] - e.U[from];
                inblockend = pfirst;
                nk = e.skip[k];                                 // nothing inside the block can matter
            }
        } else wing = plast < sa;
        k = nk;
        if (wing && x.start > inblockend) {                     // :679-684
            st.brackets += x.len;
            const uint32_t lv = x.level;
            if (lv > SQ_MAXLEVELS) e.ctr->level_ovf = 1;
            else {
                const uint32_t o0 = st.lv0, o1 = st.lv1;
                if (lv > 32u) st.lv1 |= 1u << (lv - 33u); else st.lv0 |= 1u << (lv - 1u);
                if (early && (st.lv0 != o0 || st.lv1 != o1)) {
                    const double b = ub * ofr[__popc(st.lv0) + __popc(st.lv1)];
                    if (b < need) { *bnd = b; st.k = k; st.inblockend = inblockend; return SQ_WALK_OUT; }
                }
            }
        }
    }
    st.k = k; st.inblockend = inblockend;
    return k >= e.nstrand ? SQ_WALK_DONE : SQ_WALK_MORE;
}

__device__ __forceinline__ int sq_dpp_scan_max_i32(int v)        // inclusive prefix maximum over the lanes (values >= -1)
{
    int o;
    o = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;
    o = __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;
    return v;
}
__device__ __forceinline__ uint32_t sq_dpp_reduce_or_u32(uint32_t v)   // every lane gets the OR over the lanes
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// The rest of ONE walk by the whole wave (every lane passes the same span and state; every lane gets the result): a strand per
// lane, 64 at a time.  The walk's inblockend is a running maximum -- of the partner ends of the 5' strands whose partner lies
// inside the span -- so what a strand sees of the strands before it is an exclusive prefix maximum (a DPP scan): a face is
// registered where a partner end exceeds it (partner ends are distinct), a bracket strand counts where its start does.  No
// skip pointers: the strands inside a registered block fall out by the same two tests, as in the walk without them.
__device__ __forceinline__ int sq_walk_wave(const SqStemsEnv &e, int sa, int sb, SqWalkPart &st, bool early, double ub, const double *ofr,
                                            double need, double *bnd, int lane)
{
    int carry = st.inblockend, nrec = st.nrec, be0 = st.be0, be1 = st.be1, covered = 0, brackets = 0;
    uint32_t lv0 = 0u, lv1 = 0u;
    bool lovf = false;
    const int ns = e.nstrand;
    for (int k0 = st.k; k0 < ns; k0 += 64) {
        const int k = k0 + lane;
        const SqStrand x = e.S[k < ns ? k : ns - 1];
        const bool in = k < ns && x.start < sb;                              // (sorted by start: the lanes behind the span's end are a suffix)
        const int pfirst = x.pstart, plast = x.pstart - (x.len - 1);
        const bool nwl = in && x.left && pfirst <= sb;                       // a 5' strand whose partner lies inside the span
        const bool wing = in && (x.left ? pfirst > sb : plast < sa);        // :679-684 a strand whose partner lies outside it
        const int inc = sq_dpp_scan_max_i32(nwl ? pfirst : -1);
        int excl = __builtin_amdgcn_update_dpp(-1, inc, 0x138, 0xf, 0xf, false);   // wave_shr:1 -- the lanes before this one
        excl = excl > carry ? excl : carry;
        const bool reg = nwl && pfirst > excl;                               // :687-689 sub-ECR face
        if (reg) covered += (int)e.U[pfirst + 1] - (int)e.U[x.start > excl ? x.start : excl + 1];
        if (wing && x.start > excl) {
            brackets += x.len;
            if (x.level > SQ_MAXLEVELS) lovf = true;
            else if (x.level > 32) lv1 |= 1u << (x.level - 33);
            else lv0 |= 1u << (x.level - 1);
        }
        const unsigned long long rm = __ballot(reg);
        if (nrec == 0 && rm != 0ull) {
            const int f = __ffsll((long long)rm) - 1;
            be0 = __builtin_amdgcn_readlane((int)x.start, f); be1 = __builtin_amdgcn_readlane(pfirst, f);
        }
        nrec += __popcll(rm);
        const int top = __builtin_amdgcn_readlane(inc, 63);
        carry = top > carry ? top : carry;
        const bool last = __ballot(!in) != 0ull;
        if (early || last) {
            const uint32_t a0 = sq_dpp_reduce_or_u32(lv0) | st.lv0, a1 = (__ballot(lv1 != 0u) != 0ull ? sq_dpp_reduce_or_u32(lv1) : 0u) | st.lv1;
            if (early) {
                const double b = ub * ofr[__popc(a0) + __popc(a1)];
                if (b < need) { *bnd = b; return SQ_WALK_OUT; }
            }
            if (last) { st.lv0 = a0; st.lv1 = a1; break; }
        }
    }
    if (__ballot(lovf) != 0ull && lane == 0) e.ctr->level_ovf = 1;
    st.lv0 |= sq_dpp_reduce_or_u32(lv0); st.lv1 |= (__ballot(lv1 != 0u) != 0ull ? sq_dpp_reduce_or_u32(lv1) : 0u);
    st.k = ns; st.inblockend = carry; st.nrec = nrec; st.be0 = be0; st.be1 = be1;
    st.covered += __builtin_amdgcn_readlane(sq_wave_scan_add_i32(covered), 63);
    st.brackets += __builtin_amdgcn_readlane(sq_wave_scan_add_i32(brackets), 63);
    return SQ_WALK_DONE;
}

// ... and the finalscore of the stem (i0, j0, L) with bpscore bps from what its walk found (the caller applies :751's threshold)
__device__ __forceinline__ double sq_stem_finalscore_of(const SqStemsEnv &e, int i0, int j0, int L, double bps, const SqWalk &w)
{
    double fin = 0.0;
    const int sa = i0 + L - 1, sb = j0 - L + 1;                 // :655 innermost bp
    const int nrec = w.nrec, be0 = w.be0, be1 = w.be1, covered = w.covered, brackets = w.brackets;
    const uint64_t levelset = w.levelset;
    const int dots = (e.U[sb] - e.U[sa + 1]) - covered;             // :670-673
    const bool between = e.SU != nullptr && (e.SU[sb] - e.SU[sa + 1]) > 0;   // :675-676 (nullptr: no separator anywhere)
    bool goodloop = false; int diff1 = 0;                       // :692-698
    if (nrec == 1 && sq_goodloop(be0 - sa - 1, sb - be1 - 1)) {
        goodloop = true;
        diff1 = abs((be0 - sa - 1) - (sb - be1 - 1));
    }
    bool goodloopout = false; int diff2 = 0;                    // :700-711
    {
        // the two outward walks over <= 5 unpaired positions (:702-707), from the prefix counts: the k
        // positions next to the stem are all unpaired iff the count over them is k -- ten independent
        // reads instead of two chains of dependent ones
        const int ui = e.U[i0], uj = e.U[j0 + 1];
        int cl = 0, cr = 0;
#pragma unroll
        for (int k = 1; k <= 5; k++) {
            const int a1 = i0 - k, b1 = j0 + 1 + k;
            cl += (a1 >= 0 && ui - e.U[a1 >= 0 ? a1 : 0] == k) ? 1 : 0;
            cr += (b1 <= e.n && e.U[b1 <= e.n ? b1 : e.n] - uj == k) ? 1 : 0;
        }
        const int vv = i0 - 1 - cl, ww = j0 + 1 + cr;
        if (vv >= 0 && ww < e.n && e.P[vv] == ww && sq_goodloop(cl, cr)) {
            goodloopout = true;
            diff2 = abs(cl - cr);
        }
    }
    const double lb = e.lb;
    const double loopfactor = (1.0 + (lb * (goodloop ? 1.0 : 0.0)) * (2.0 - diff1 / 2.0))
                              + (lb * (goodloopout ? 1.0 : 0.0)) * (2.0 - diff2 / 2.0);   // :715
    bool gnra = false;                                          // :598-604,718
    if (sb - sa - 1 == 4 && e.codes[sa + 1] == 6 && (e.codes[sa + 3] == 6 || e.codes[sa + 3] == 0) && e.codes[sa + 4] == 0)
        gnra = true;
    const double tetra = gnra ? 1.25 : 1.0;
    const double ideal = nrec == 0 ? 4.0 : 2.0;                 // :721
    const double stemdist = (double)dots + e.bw * (double)brackets;   // :723
    const double dd = fabs(stemdist - ideal);
    double sdf = 1.0;                                           // :726
    if (!between) {
        const int di = (int)dd;
        if (e.bwint && di < e.sdflen) sdf = di < e.sdf_llen ? e.sdf_l[di] : e.sdf[di];
        else sdf = pow(1.0 / (1.0 + dd), e.dc);
    }
    const double of = e.of[__popcll(levelset)];            // :728-729
    fin = bps * sdf * of * loopfactor * tetra;                  // :732 (reactfactor == 1)
    if (!goodloop && !goodloopout && L < 3) fin = -1.0;         // :744-745
    return fin;
}

__device__ __forceinline__ double sq_stem_finalscore(const SqStemsEnv &e, int i0, int j0, int L, double bps)
{
    return sq_stem_finalscore_of(e, i0, j0, L, bps, sq_stem_walk(e, i0, j0, L));
}
