// sq_extend.h -- one structure + one stem on the device: the stem-level crossing weights, the pseudoknot levels of all
// strands (PairsToDBN's rule at stem level, SQRNdbnseq.py:104-150: order by (crossing weight, start), first fit into
// groups, groups ranked by size) and the sorted strand list with the two new strands.  One wave per structure; shared by
// sq_chain_kernel (width-1 pools: the child replaces its parent) and sq_pool_extend_kernel (device pools: children are
// written to new slots).
#pragma once
#include <hip/hip_runtime.h>
#include "sq_device.h"

// The wave's own barrier between its LDS writes and reads.  Kernels whose block IS that wave use the block barrier;
// a kernel that runs the extension on one wave of a wider block (sq_rounds.hip) defines a wave-level fence instead.
#ifndef SQ_EXTEND_SYNC
#define SQ_EXTEND_SYNC() __syncthreads()
#endif

// stems per structure the level scratch can hold: 14 bytes of LDS each, a block's dynamic LDS is sized for the launch's
// longest list (SQ_CHAIN_TMAX in sq_device.h = what 160 KB hold; structures that could grow longer run the host loop)

__device__ __forceinline__ bool sq_chain_cross(int ai, int aj, int bi, int bj)     // SQRNdbnseq.py:114-116
{
    return (ai < bi && bi < aj && aj < bj) || (bi < ai && ai < bj && bj < aj);
}

__device__ __forceinline__ unsigned long long sq_wave_or64(unsigned long long v)
{
    for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ unsigned long long sq_wave_min64(unsigned long long v)
{
    for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int sq_wave_sum32(int v)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

#define SQ_LEVELS_REG 8            // stems per lane the first fit of the levels keeps in registers (lists of up to 512 stems)
__device__ __forceinline__ uint32_t sq_extend_or_u32(uint32_t v)       // every lane gets the OR over the lanes (DPP: VALU only)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

struct SqExtendLds {
    int16_t *i, *j, *len, *ord;
    int32_t *cc;
    uint8_t *grp, *lvl;
    int32_t *gsize;              // [64]
    uint8_t *rank;               // [64]
};
// carves the arrays for lists of up to T stems out of the block's dynamic LDS (sq_extend_lds_bytes(T) bytes at base)
__device__ __forceinline__ SqExtendLds sq_extend_lds(char *base, int T)
{
    const int t = (T + 7) & ~7;
    SqExtendLds L;
    L.cc = reinterpret_cast<int32_t *>(base);
    L.gsize = L.cc + t;
    L.i = reinterpret_cast<int16_t *>(L.gsize + 64);
    L.j = L.i + t; L.len = L.j + t; L.ord = L.len + t;
    L.grp = reinterpret_cast<uint8_t *>(L.ord + t);
    L.lvl = L.grp + t;
    L.rank = L.lvl + t;
    return L;
}

// PairsToDBN's level rule at stem level (:119-150) for the T stems in L.i / L.j / L.len with their crossing weights L.cc:
// stems that cross nothing all land in group 0; the others are ordered by (weight, start), first-fitted into groups
// and the groups ranked by size.  Result: L.lvl[q] = 1-based level of stem q.  One wave; the caller's block is that wave
// (every SQ_EXTEND_SYNC() below is the wave's own barrier).  level_ovf: set when more than SQ_MAXLEVELS groups appear.
__device__ __forceinline__ int sq_stem_levels_wave(SqExtendLds &L, int T, int lane, uint32_t *level_ovf, long long *prof = nullptr)
{
    long long pt0 = prof ? wall_clock64() : 0;
#define SQ_LVPROF(k) do { if (prof) { const long long n_ = wall_clock64(); prof[k] += n_ - pt0; pt0 = n_; } } while (0)
    // stems that cross nothing sort first (weight 0) and all land in group 0
    int g0 = 0, has0 = 0;
    for (int q = lane; q < T; q += 64) {
        const bool free_ = L.cc[q] == 0;
        L.grp[q] = free_ ? 0 : 255;
        if (free_) { g0 += L.len[q]; has0 = 1; }
    }
    g0 = sq_wave_sum32(g0);
    int ngroups = __ballot(has0) != 0ull ? 1 : 0;
    if (lane == 0) L.gsize[0] = g0;
    // order of the crossing stems: (weight, start) ascending (:125); starts are distinct
    int nx = 0;
    if (T <= 64 * SQ_LEVELS_REG) {
        // (lists of up to 512 stems: ONE sweep over the stems with every lane's keys -- weight << 15 | start, weights stay below
        // 2^16: they are sums of stem lengths -- in registers and the loads of a step unconditional, so that they pipeline; until
        // round 6 a sweep per 64 stems whose second load hung behind a branch: ~90 us of a 360-stem structure's round)
        uint32_t key[SQ_LEVELS_REG]; int rk[SQ_LEVELS_REG];
#pragma unroll
        for (int c = 0; c < SQ_LEVELS_REG; c++) {
            const int q = 64 * c + lane;
            const bool x = q < T && L.cc[q] > 0;
            key[c] = x ? ((uint32_t)L.cc[q] << 15) | (uint32_t)(uint16_t)L.i[q] : 0u;     // (0: not a crossing stem)
            rk[c] = 0;
            nx += __popcll(__ballot(x));
        }
#pragma unroll 4
        for (int p = 0; p < T; p++) {
            const uint32_t cp = (uint32_t)L.cc[p], ip = (uint32_t)(uint16_t)L.i[p];
            const uint32_t kp = cp ? (cp << 15) | ip : 0xFFFFFFFFu;                          // (a stem without crossings is below nobody)
#pragma unroll
            for (int c = 0; c < SQ_LEVELS_REG; c++) rk[c] += kp < key[c] ? 1 : 0;
        }
#pragma unroll
        for (int c = 0; c < SQ_LEVELS_REG; c++) if (key[c]) L.ord[rk[c]] = (int16_t)(64 * c + lane);
    } else
    for (int q0 = 0; q0 < T; q0 += 64) {
        const int q = q0 + lane;
        const bool x = q < T && L.cc[q] > 0;
        if (x) {
            const int cq = L.cc[q], iq = L.i[q];
            int r = 0;
            for (int p = 0; p < T; p++) {
                const int cp = L.cc[p];
                r += (cp > 0 && (cp < cq || (cp == cq && L.i[p] < iq))) ? 1 : 0;
            }
            L.ord[r] = (int16_t)q;
        }
        nx += __popcll(__ballot(x));
    }
    SQ_EXTEND_SYNC();
    SQ_LVPROF(0);
    // first fit (:130-136): a stem joins the first group none of whose members it crosses
    if (T <= 64 * SQ_LEVELS_REG) {
        // Lists of up to 512 stems: every lane keeps its stems' ends and groups in REGISTERS for the whole loop (until round 6 each
        // of the nx steps read all T stems' groups and ends from LDS, reduced the blocked groups with twelve ds_bpermute and went
        // through two barriers for lane 0's update: 335 us per round on structures of 360 stems, most of a round's extension --
        // and the other waves of the round kernel wait for it before they score).  A step: the next stem's ends (asked for one
        // step ahead), the crossing tests on registers, the blocked groups by DPP OR reductions, the owner's register update.
        int pij[SQ_LEVELS_REG], g[SQ_LEVELS_REG];
#pragma unroll
        for (int c = 0; c < SQ_LEVELS_REG; c++) {
            const int q = 64 * c + lane;
            const bool v = q < T;
            pij[c] = v ? ((int)(uint16_t)L.i[q] | ((int)(uint16_t)L.j[q] << 16)) : 0;
            g[c] = v ? (int)L.grp[q] : 255;
        }
        int mygs = lane == 0 ? g0 : 0;                           // lane k: the size of group k
        int pn = nx > 0 ? L.ord[0] : 0;
        int ni = L.i[pn], nj = L.j[pn], nl = L.len[pn];
        for (int t = 0; t < nx; t++) {
            const int p = pn, pi = ni, pj = nj, plen = nl;
            if (t + 1 < nx) { pn = L.ord[t + 1]; ni = L.i[pn]; nj = L.j[pn]; nl = L.len[pn]; }
            uint32_t b0 = 0u, b1 = 0u;
#pragma unroll
            for (int c = 0; c < SQ_LEVELS_REG; c++) {
                if (64 * c >= T) break;                                  // (wave-uniform)
                const int qi = pij[c] & 0xFFFF, qj = (int)((uint32_t)pij[c] >> 16);
                const bool x = g[c] != 255 && sq_chain_cross(pi, pj, qi, qj);
                if (x) { if (g[c] < 32) b0 |= 1u << g[c]; else b1 |= 1u << (g[c] - 32); }
            }
            b0 = sq_extend_or_u32(b0);
            b1 = ngroups > 32 ? sq_extend_or_u32(b1) : 0u;
            const unsigned long long blocked = (unsigned long long)b0 | ((unsigned long long)b1 << 32);
            int placed = blocked == ~0ull ? 64 : __ffsll((long long)~blocked) - 1;
            if (placed > ngroups) placed = ngroups;
            if (placed >= SQ_MAXLEVELS) { if (lane == 0) *level_ovf = 1; placed = SQ_MAXLEVELS - 1; }   // (reported as an error)
            else if (placed == ngroups) ngroups++;
#pragma unroll
            for (int c = 0; c < SQ_LEVELS_REG; c++) if (c == (p >> 6) && lane == (p & 63)) g[c] = placed;
            if (lane == (p & 63)) L.grp[p] = (uint8_t)placed;
            if (lane == placed) mygs += plen;
        }
        if (lane < 64) L.gsize[lane] = lane < ngroups ? mygs : 0;
        SQ_EXTEND_SYNC();
    } else
    for (int t = 0; t < nx; t++) {
        const int p = L.ord[t];
        const int pi = L.i[p], pj = L.j[p];
        unsigned long long blocked = 0ull;
        for (int q = lane; q < T; q += 64) {
            const int g = L.grp[q];
            if (g != 255 && sq_chain_cross(pi, pj, L.i[q], L.j[q])) blocked |= 1ull << g;
        }
        blocked = sq_wave_or64(blocked);
        int placed = blocked == ~0ull ? 64 : __ffsll((long long)~blocked) - 1;
        if (placed > ngroups) placed = ngroups;
        if (placed >= SQ_MAXLEVELS) { if (lane == 0) *level_ovf = 1; placed = SQ_MAXLEVELS - 1; }   // (reported as an error)
        else if (placed == ngroups) { ngroups++; if (lane == 0) L.gsize[placed] = 0; }
        SQ_EXTEND_SYNC();
        if (lane == 0) { L.grp[p] = (uint8_t)placed; L.gsize[placed] += L.len[p]; }
        SQ_EXTEND_SYNC();
    }
    SQ_LVPROF(1);
    if (prof) prof[3] += nx;
    // groups ranked by size, descending, stable (:139); level = rank + 1
    if (lane < ngroups) {
        const int gs = L.gsize[lane];
        int r = 0;
        for (int h = 0; h < ngroups; h++) { const int hs = L.gsize[h]; r += (hs > gs || (hs == gs && h < lane)) ? 1 : 0; }
        L.rank[lane] = (uint8_t)(r + 1);
    }
    SQ_EXTEND_SYNC();
    for (int q = lane; q < T; q += 64) L.lvl[q] = L.rank[L.grp[q]];
    SQ_EXTEND_SYNC();
    SQ_LVPROF(2);
#undef SQ_LVPROF
    return ngroups;                                   // (groups in use: L.grp / L.gsize stay valid for sq_stem_levels_join)
}

// The same levels after ONE more stem (index T - 1, length len) that crosses nothing, given the groups of the T - 1 stems
// before it (L.grp, L.gsize, ngroups >= 1 as sq_stem_levels_wave left them): a stem without crossings never blocks a group
// and sorts in front of the crossing ones (:125), so the first fit of every other stem is what it was; the new stem joins
// group 0, whose size grows, and only the ranking of the groups by size (:139) is taken anew.
__device__ __forceinline__ void sq_stem_levels_join(SqExtendLds &L, int T, int ngroups, int len, int lane)
{
    if (lane == 0) { L.grp[T - 1] = 0; L.gsize[0] += len; }
    SQ_EXTEND_SYNC();
    if (lane < ngroups) {
        const int gs = L.gsize[lane];
        int r = 0;
        for (int h = 0; h < ngroups; h++) { const int hs = L.gsize[h]; r += (hs > gs || (hs == gs && h < lane)) ? 1 : 0; }
        L.rank[lane] = (uint8_t)(r + 1);
    }
    SQ_EXTEND_SYNC();
    for (int q = lane; q < T; q += 64) L.lvl[q] = L.rank[L.grp[q]];
    SQ_EXTEND_SYNC();
}

// The sorted strand list of a child: the parent's nstrand strands psrc[] (+ the stem index of each, pssrc[]) with the two
// strands of the new stem k = (i0, j0, len) inserted, levels from L.lvl when stems cross (else level 1 everywhere).
// cdst[] / csdst[] take nstrand + 2 entries and must NOT be the parent's arrays.
// (PRE: the parent's first stems and strands were asked for before the wave had anything else to do with them -- sq_extend_preload;
// the kernels that find their parent through a chain of dependent loads issue everything the chain's end names at once)
struct SqExtendPre { SqChainStem st0; SqStrand x[2]; int16_t sx[2]; };
__device__ __forceinline__ SqExtendPre sq_extend_preload(const SqChainStem *pst, int k, const SqStrand *psrc, const int16_t *pssrc, int nstrand, int lane)
{
    SqExtendPre P;
    // (no branches: clamped indices, so that the loads go out together -- entries past the lists' ends are never used)
    P.st0 = pst[min(lane, max(k - 1, 0))];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int q = min(64 * t + lane, max(nstrand - 1, 0));
        P.x[t] = psrc[q];
        P.sx[t] = pssrc[q];
    }
    return P;
}

template <bool PRE = false>
__device__ __forceinline__ void sq_extend_strands(SqExtendLds &L, bool anycross, int k, const SqStrand *psrc, const int16_t *pssrc, int nstrand,
                                                  int i0, int j0, int len, SqStrand *cdst, int16_t *csdst, int lane, const SqExtendPre *pre = nullptr)
{
    const int ls = i0, rs = j0 - len + 1;                               // starts of the 5' and the 3' strand (ls < rs)
    int below_l = 0, below_r = 0;
#pragma unroll 2
    for (int q0 = 0; q0 < nstrand; q0 += 64) {
        const int q = q0 + lane;
        const bool valid = q < nstrand;
        SqStrand x; int sx;
        if (PRE && q0 < 128) { x = pre->x[q0 >> 6]; sx = pre->sx[q0 >> 6]; }
        else { x = valid ? psrc[q] : SqStrand{0, 0, 0, 0, 0}; sx = valid ? pssrc[q] : 0; }
        const bool bl = valid && x.start < ls, br = valid && x.start < rs;
        if (valid) {
            if (anycross) x.level = L.lvl[sx];
            const int at = q + (bl ? 0 : 1) + (br ? 0 : 1);
            cdst[at] = x; csdst[at] = (int16_t)sx;
        }
        below_l += __popcll(__ballot(bl)); below_r += __popcll(__ballot(br));
    }
    if (lane == 0) {
        const uint8_t lv = anycross ? L.lvl[k] : (uint8_t)1;
        cdst[below_l] = SqStrand{(int16_t)ls, (int16_t)len, (int16_t)j0, lv, 1};
        cdst[below_r + 1] = SqStrand{(int16_t)rs, (int16_t)len, (int16_t)(i0 + len - 1), lv, 0};
        csdst[below_l] = (int16_t)k; csdst[below_r + 1] = (int16_t)k;
    }
}

// parent: k stems pst[] (with their crossing weights), nstrand sorted strands psrc[] + the stem index of each (pssrc[]).
// child: cst[0..k] (may be the parent's array: the weights are updated in place), cdst[] / csdst[] (nstrand + 2 entries;
// must NOT be the parent's).  (i0, j0, len): the new stem.  Returns whether some pair of the child's stems crosses.
template <bool PRE = false>
__device__ __forceinline__ bool sq_extend_structure(SqExtendLds &L, const SqScanArgs &a, const SqChainStem *pst, int k, bool parent_anycross,
                                                    const SqStrand *psrc, const int16_t *pssrc, int nstrand, int i0, int j0, int len,
                                                    SqChainStem *cst, SqStrand *cdst, int16_t *csdst, int lane, const SqExtendPre *pre = nullptr)
{
    // ---- crossing weights (:121-124), kept per stem between rounds ----
    int mycc = 0, mycross = 0;
    for (int q = lane; q < k; q += 64) {
        const SqChainStem x = PRE && q < 64 ? pre->st0 : pst[q];
        int cc = x.cc;
        if (sq_chain_cross(x.i, x.j, i0, j0)) { cc += len; mycc += x.len; mycross = 1; }
        cst[q] = SqChainStem{x.i, x.j, x.len, cc};
        L.i[q] = (int16_t)x.i; L.j[q] = (int16_t)x.j; L.len[q] = (int16_t)x.len; L.cc[q] = cc;
    }
    const int newcc = sq_wave_sum32(mycc);
    const bool anycross = parent_anycross || __ballot(mycross) != 0ull;
    if (lane == 0) {
        L.i[k] = (int16_t)i0; L.j[k] = (int16_t)j0; L.len[k] = (int16_t)len; L.cc[k] = newcc;
        cst[k] = SqChainStem{i0, j0, len, newcc};
    }
    SQ_EXTEND_SYNC();
    const int T = k + 1;
    // ---- levels (only when stems cross; otherwise every strand stays on level 1) ----
    if (anycross) sq_stem_levels_wave(L, T, lane, &a.ctr->level_ovf);
    // ---- strands: the sorted list with the two new strands ----
    sq_extend_strands<PRE>(L, anycross, k, psrc, pssrc, nstrand, i0, j0, len, cdst, csdst, lane, pre);
    return anycross;
}
