// sq_scan.h -- a-2  AnnotateStems as a bit-diagonal scan (SQRNdbnseq.py:427-495), shared by sq_scan6_kernel (launched
// rounds) and the persistent round kernel (sq_rounds.hip: its first round).
//
// AnnotateStems only needs to know WHERE the unmasked cells are -- every candidate's score is recomputed exactly in
// fp64 -- so the scan reads the job's diagonal bit matrix (sq_bits_*_kernel) instead of a score matrix:
//   active(s, i) = base(s, i)  &  free[i]  &  free[s - i]       (+ the live restraint pairs)
// One lane = one anti-diagonal, one loop step = 32 rows:
//   base word   one coalesced 4-byte load per lane (64 consecutive diagonals = 256 B per wave);
//   row word    wave-uniform LDS read of the free-position bit array F;
//   column word a 32-bit window of the REVERSED array G (bit k <-> position n-1-k) starting at
//               n-1-s+32w: it advances by exactly one word per step, so each step reads one new LDS
//               word and funnel-shifts it against the previous one (v_alignbit);
// and the runs of the 32 rows come out of the word with bit tricks (below).  Per structure the scan touches
// N^2/16 bytes of (L2-resident) bits instead of 2 N^2 bytes of HBM.
#pragma once
#include <hip/hip_runtime.h>
#include "sq_internal.h"
#include "sq_device.h"

#ifndef SQ6_AHEAD
#define SQ6_AHEAD 2
#endif

// One wave scans the diagonal groups gy0, gy0 + gystep, .. of a structure of job jb.  F / G: the structure's free-position
// bit words (forward / reversed + SQ_GPAD, fbh words each, in LDS); eg: its mask codes per position (0 free, 1 end of a
// live restraint pair, 255 paired).  Sink: where the runs of at least minlen cells go --
//   reserve(n, lane)  all lanes (wave-uniform n): places for n runs, returns the first one's index;
//   put(at, key, len) from any lane, into a reserved place, key = (s << 16) | first row;
//   poll(lane)        after every word-row, all lanes (room to flush a staging buffer);
//   drain(lane)       after every group of 64 diagonals, all lanes.
// Bits: where the words of the job's diagonal bit matrix come from -- word(w, s) = bit b <-> cell (32w + b, s - 32w - b) of the
// unmasked BPMatrix, zero outside the matrix.  SqBitsGlobal reads the matrix the bit kernels wrote; SqBitsFly (below) forms
// the word from letter masks in LDS, so that a fold that scans every job ONCE never writes the matrix at all.
struct SqBitsGlobal {
    const uint32_t *bp; int bpitch;
    __device__ __forceinline__ void start(int) {}
    __device__ __forceinline__ uint32_t next(int w, int s, int, int) { return bp[(int64_t)w * bpitch + s]; }
};
// The letter-mask form of sq_bits_masks_kernel (sq_kernels.hip), word by word: for the letters x present in the sequence,
// R[w][x] = the rows of word-row w with letter x (and no row-side restraint flag), Mr[x] = the columns whose letter may pair
// with x (and no column-side flag) as a bit array over the REVERSED positions, laid out like the scan's own G array (bit
// k + SQ_GPAD <-> position n - 1 - k, fbh words, zeros outside the sequence).  The columns of word (w, s) -- s - 32w - b, b = 0 .. 31
// -- are then the very 32-bit window of Mr[x] the scan takes of G: it advances by one word per word-row, so a step costs one
// LDS word and one funnel shift per letter (the first form looked each window up anew: 40 instructions per word).  Only
// the words next to the main diagonal check the minimal loop length bit by bit (:294-299).  Stateful: start() at the first
// word-row of a diagonal group, then next() for every word-row in turn.
#define SQ_FLY_MAXL 8               // letters of a batch the masks are kept for (more: the bit kernel's matrices)
struct SqBitsFly {
    const uint32_t *Mr, *R; const uint8_t *inc; int fbh, nlet, maxl, n;
    uint32_t lo[SQ_FLY_MAXL];
    __device__ __forceinline__ void start(int gidx)
    {
#pragma unroll
        for (int k = 0; k < SQ_FLY_MAXL; k++) lo[k] = k < nlet ? Mr[k * fbh + gidx] : 0u;
    }
    __device__ __forceinline__ uint32_t next(int w, int s, int gi, int gsh)
    {
        uint32_t word = 0;
        const uint32_t *Rw = R + w * maxl;
#pragma unroll
        for (int k = 0; k < SQ_FLY_MAXL; k++)
            if (k < nlet) {
                const uint32_t hi = Mr[k * fbh + gi];
                word |= Rw[k] & __builtin_amdgcn_alignbit(hi, lo[k], (uint32_t)gsh);
                lo[k] = hi;
            }
        const int i0 = 32 * w, t = s - i0;                                   // column of bit 0
        if (!(s >= 4 && s <= 2 * n - 6 && t >= 0 && t - 31 < n)) word = 0;      // (outside the matrix: the window is not this diagonal's)
        const int d = s - 2 * i0;                                            // j - i of bit b is d - 2b
        if (word && d < 4 + 62) {
            uint32_t keep = 0;
            for (int bb = 0; bb < 32 && i0 + bb < n; bb++)
                if (d - 2 * bb >= (int)inc[i0 + bb]) keep |= 1u << bb;          // :294-299
            word &= keep;
        }
        return word;
    }
};
// LDS of the masks: row letters, minimal j - i (a byte per position each), Mr, R
__host__ __device__ inline size_t sq_bits_fly_bytes(int n, int nletters)
{
    const size_t npad = ((size_t)n + 3) & ~(size_t)3, nw = ((size_t)n + 31) / 32, fbh = (((size_t)n + 2 + 31) >> 5) + 8;
    return 3 * npad + 4 * (size_t)nletters * fbh + 4 * nw * (size_t)nletters + 16;
}

template <class Sink, class Bits>
__device__ __forceinline__ void sq_scan6_groups(const SqDevCtx &c, const SqJob &jb, const uint32_t *F, const uint32_t *G, int fbh,
                                                const uint8_t *eg, int gy0, int gystep, int lane, Sink &sink, Bits bits)
{
    const int n = jb.n;
    const SqPsetDev *ps = c.psets + jb.pset;
    const int minlen = max(1, (int)ceil(ps->minlen));
    const int ngroups = ((2 * n - 5 + 63) >> 6) + 1;
    for (int gy = gy0; gy < ngroups; gy += gystep) {
        const int s0 = gy << 6;
        const int smin = max(s0, 4), smax = min(s0 + 63, 2 * n - 6);        // :456-457 s in [4, 2N-6]
        if (smin > smax) continue;
        const int rmin = max(0, smin - (n - 1)), rmax = (smax - 1) >> 1;    // :486 i <= j-1
        const int wlo = rmin >> 5, whi = rmax >> 5;
        const int s = s0 + lane;
        // window of the reversed array for (s, wlo): first bit n-1-s+32 wlo (+pad); lanes outside the valid
        // diagonals have zero base words, their window only has to stay inside the array
        const int q0 = min(max(n - 1 - s + 32 * wlo + SQ_GPAD, 0), 32 * (fbh - (whi - wlo) - 3));
        const int gidx = q0 >> 5, gsh = q0 & 31;
        // Restraint base pairs (:438-443: the cell of a restraint pair stays pairable while both ends are free).  The
        // sequence's list is sorted by (i + j, i), so the pairs of this lane's diagonal are one run of it, in row order: two
        // binary searches here, then a pointer that only moves forward as the rows go by.  Any number of pairs.
        const uint32_t *rlist = c.rbpk + jb.rb_off;
        int rp = 0, rend = 0;
        if (jb.nrb) {
            int lo = 0, hi = jb.nrb;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint32_t pk = rlist[mid];
                if ((int)(pk & 0xFFFFu) + (int)(pk >> 16) < s) lo = mid + 1; else hi = mid;
            }
            rp = lo; hi = jb.nrb;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const uint32_t pk = rlist[mid];
                if ((int)(pk & 0xFFFFu) + (int)(pk >> 16) <= s) lo = mid + 1; else hi = mid;
            }
            rend = lo;
        }

        int carry = 0;
        uint32_t glo = G[gidx];
        bits.start(gidx);
        // window of minlen ones by doubling: Y &= Y >> ysh_q, five fixed steps (shift 0 once the window is complete)
        int ysh0, ysh1, ysh2, ysh3, ysh4;
        {
            const int want = minlen < 32 ? minlen : 32;
            int have = 1;
            ysh0 = min(have, want - have); have += ysh0;
            ysh1 = min(have, want - have); have += ysh1;
            ysh2 = min(have, want - have); have += ysh2;
            ysh3 = min(have, want - have); have += ysh3;
            ysh4 = min(have, want - have);
        }
        // SQ6_AHEAD word-rows per trip: their (independent) loads are issued together, so a wave waits for HBM / L2 once
        // per group instead of once per word (a single word of look-ahead did not survive the compiler's wait counts)
        for (int w0 = wlo; w0 <= whi; w0 += SQ6_AHEAD) {
            uint32_t bw[SQ6_AHEAD];
#pragma unroll
            for (int k = 0; k < SQ6_AHEAD; k++) bw[k] = w0 + k <= whi ? bits.next(w0 + k, s, gidx + 1 + (w0 + k - wlo), gsh) : 0u;
#pragma unroll
            for (int k = 0; k < SQ6_AHEAD; k++) {
                const int w = w0 + k;
                if (w > whi) break;
                const uint32_t base = bw[k];
                const uint32_t ghi = G[gidx + 1 + (w - wlo)];
                const uint32_t gw = __builtin_amdgcn_alignbit(ghi, glo, gsh);   // columns s-32w-b, b = 0..31
                glo = ghi;
                uint32_t A = base & F[w] & gw;                              // :438-451 free row and column
                while (rp < rend) {                                         // (no lane enters without restraint pairs on its diagonal)
                    const uint32_t pk = rlist[rp];
                    const int v = (int)(pk & 0xFFFFu);
                    if ((v >> 5) > w) break;
                    if ((v >> 5) == w && eg[v] == 1 && eg[pk >> 16] == 1) A |= base & (1u << (v & 31));   // :438-443 restraint bp stays pairable
                    rp++;
                }
                // maximal runs of the word (bit b = row 32w + b; `carry` rows of an open run precede row 32w).  Every lane first
                // finds what it has to emit -- the run that continues the carried one, the starts of the runs strictly inside --,
                // then the WAVE reserves their places at once: the lanes' counts as a prefix sum (DPP), one request to the sink
                // per word-row.  (Until round 5 every run took its place with an LDS atomic of its own on ONE counter: some
                // 180 same-address atomics per word-row of a wave at 1,000 nt, served one after the other, each waited for by
                // its lane -- most of the scan's time.)
                bool e0 = false; int l0 = 0, r0 = 0; uint32_t starts = 0u, rest = 0u;
                if (A == 0xFFFFFFFFu) carry += 32;
                else {
                    const int lead = __ffs((int)~A) - 1;                    // the run that continues the carried one (maybe empty)
                    l0 = carry + lead; r0 = 32 * w - carry; e0 = l0 >= minlen;
                    const int trail = __clz((int)~A);                       // the run still open at row 32w + 31
                    carry = trail;
                    // runs strictly inside: starts with a full window of minlen ones above them
                    rest = A & ~((1u << lead) - 1u);
                    if (trail) rest &= 0xFFFFFFFFu >> trail;
                    uint32_t Y = rest;
                    Y &= Y >> ysh0; Y &= Y >> ysh1; Y &= Y >> ysh2; Y &= Y >> ysh3; Y &= Y >> ysh4;
                    starts = rest & ~(rest << 1) & Y;
                }
                const int cnt = (e0 ? 1 : 0) + __popc(starts);
                if (__ballot(cnt > 0) != 0ull) {
                    const int incl = sq_wave_scan_add_i32(cnt);
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, 63);
                    uint32_t at = sink.reserve(total, lane) + (uint32_t)(incl - cnt);
                    if (e0) sink.put(at++, ((uint32_t)s << 16) | (uint32_t)r0, (uint32_t)l0);
                    while (starts) {
                        const int p = __ffs((int)starts) - 1;
                        starts &= starts - 1;
                        const int len = __ffs((int)~(rest >> p)) - 1;
                        sink.put(at++, ((uint32_t)s << 16) | (uint32_t)(32 * w + p), (uint32_t)len);
                    }
                }
                sink.poll(lane);
            }
        }
        {
            const bool e1 = carry >= minlen;                                   // the run still open behind the last word-row
            const unsigned long long m1 = __ballot(e1);
            if (m1) {
                const uint32_t at = sink.reserve((uint32_t)__popcll(m1), lane) + (uint32_t)__popcll(m1 & ((1ull << lane) - 1ull));
                if (e1) sink.put(at, ((uint32_t)s << 16) | (uint32_t)(32 * (whi + 1) - carry), (uint32_t)carry);
            }
        }
        sink.drain(lane);
    }
}
