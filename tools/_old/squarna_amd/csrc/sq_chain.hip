// sq_chain.hip -- a-7 on the device for width-1 pools (poollim == 1, SQRNdbnseq.py:1102-1199).
//
// With a pool limit of one every structure has exactly one child per round: parent + the first stem of ChooseStems
// (:754-789: stable descending sort by finalscore == highest finalscore, earliest emission among equals).  There is no
// host decision in that loop, so the rounds are chained on the stream: after the scoring kernel of a round,
// sq_chain_kernel (one wave per structure)
//   * picks that stem from the structure's list of threshold survivors,
//   * appends it to the structure's stem list (device copy for the crossing tests, pinned copy for the host's tail),
//   * inserts its two strands into the sorted strand list (written into the structure's other strand buffer),
//   * recomputes the pseudoknot levels of all strands when stems cross (the stem-level restatement of PairsToDBN's
//     level rule, :104-150, the same steps as sq_stem_levels on the host: crossing weights, order by (weight, start),
//     first fit into groups, groups ranked by size),
//   * or retires the structure (no survivor, :1192-1193; maxstemnum reached, :1168-1174) and reports it.
// The next round's state kernel reads the structures and strands where this kernel left them.  The host only enqueues
// rounds ahead of the device and watches the list of finished structures (sq_chain_fold in sq_host.hip).
#include <hip/hip_runtime.h>
#include "sq_device.h"

#define SQ_CHAIN_TMAX 1024      // stems per structure the level scratch holds (longer chains run the host loop)

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

extern "C" __global__ __launch_bounds__(64) void sq_chain_kernel(SqDevCtx c, SqStruct *structs, SqScanArgs a, SqChainIO cio)
{
    __shared__ int16_t s_i[SQ_CHAIN_TMAX], s_j[SQ_CHAIN_TMAX], s_len[SQ_CHAIN_TMAX], s_ord[SQ_CHAIN_TMAX];
    __shared__ int32_t s_cc[SQ_CHAIN_TMAX];
    __shared__ uint8_t s_grp[SQ_CHAIN_TMAX], s_lvl[SQ_CHAIN_TMAX];
    __shared__ int32_t s_gsize[64];
    __shared__ uint8_t s_rank[64];
    const int b = blockIdx.x, lane = threadIdx.x;
    const SqStruct st = structs[b];
    if (st.nstrand < 0) return;                                         // final since an earlier round
    const SqChain ch = cio.chain[b];
    const SqJob jb = c.jobs[st.job];
    auto retire = [&](int nstems, int by_count) {
        if (lane == 0) {
            structs[b].nstrand = -1;
            const uint32_t idx = atomicAdd(cio.d_nfin, 1u);
            cio.h_fin[idx] = (unsigned long long)(uint32_t)st.job | ((unsigned long long)(uint32_t)nstems << 32) |
                             ((unsigned long long)(by_count ? 1 : 0) << 63);
        }
    };
    const unsigned long long ob = a.best[st.slot];
    if (ob == 0ull) { retire(ch.nstems, 0); return; }                   // no stem passed the thresholds: the structure is final
    // ---- ChooseStems' first element: the highest finalscore, the smallest emission key among equals ----
    const double bestfin = sq_unord(ob);
    const uint32_t nok = a.ok_cnt[st.slot];
    const SqOk *oks = sq_oks(a, st, jb.cand_cap);
    unsigned long long pick = ~0ull;
    for (uint32_t q = lane; q < nok; q += 64) {
        const SqOk cd = oks[q];
        if (cd.fin == bestfin) { const unsigned long long v = ((unsigned long long)cd.key << 32) | q; pick = v < pick ? v : pick; }
    }
    pick = sq_wave_min64(pick);
    if (pick == ~0ull) { retire(ch.nstems, 0); return; }                // (not reachable: best comes from this list)
    const SqOk cd = oks[(uint32_t)pick];
    const int i0 = (int)(cd.key & 0xFFFFu), j0 = (int)(cd.key >> 16) - i0, len = (int)cd.len;
    const int k = ch.nstems;
    if (k >= ch.tcap) { if (lane == 0) a.ctr->out_ovf = 1; retire(k, 0); return; }
    SqChainStem *gst = cio.stems + ch.toff;
    // ---- crossing weights (:121-124), kept per stem between rounds ----
    int mycc = 0, mycross = 0;
    for (int q = lane; q < k; q += 64) {
        const SqChainStem x = gst[q];
        int cc = x.cc;
        if (sq_chain_cross(x.i, x.j, i0, j0)) { cc += len; gst[q].cc = cc; mycc += x.len; mycross = 1; }
        s_i[q] = (int16_t)x.i; s_j[q] = (int16_t)x.j; s_len[q] = (int16_t)x.len; s_cc[q] = cc;
    }
    const int newcc = sq_wave_sum32(mycc);
    const bool anycross = ch.anycross || __ballot(mycross) != 0ull;
    if (lane == 0) {
        s_i[k] = (int16_t)i0; s_j[k] = (int16_t)j0; s_len[k] = (int16_t)len; s_cc[k] = newcc;
        gst[k] = SqChainStem{i0, j0, len, newcc};
        cio.h_stems[ch.toff + k] = SqStemOut{i0, j0, len, 0, cd.bps, cd.fin};
    }
    __syncthreads();
    const int T = k + 1;
    // ---- levels (only when stems cross; otherwise every strand stays on level 1) ----
    if (anycross) {
        // stems that cross nothing sort first (weight 0) and all land in group 0
        int g0 = 0, has0 = 0;
        for (int q = lane; q < T; q += 64) {
            const bool free_ = s_cc[q] == 0;
            s_grp[q] = free_ ? 0 : 255;
            if (free_) { g0 += s_len[q]; has0 = 1; }
        }
        g0 = sq_wave_sum32(g0);
        int ngroups = __ballot(has0) != 0ull ? 1 : 0;
        if (lane == 0) s_gsize[0] = g0;
        // order of the crossing stems: (weight, start) ascending (:125); starts are distinct
        int nx = 0;
        for (int q0 = 0; q0 < T; q0 += 64) {
            const int q = q0 + lane;
            const bool x = q < T && s_cc[q] > 0;
            if (x) {
                const int cq = s_cc[q], iq = s_i[q];
                int r = 0;
                for (int p = 0; p < T; p++) {
                    const int cp = s_cc[p];
                    r += (cp > 0 && (cp < cq || (cp == cq && s_i[p] < iq))) ? 1 : 0;
                }
                s_ord[r] = (int16_t)q;
            }
            nx += __popcll(__ballot(x));
        }
        __syncthreads();
        // first fit (:130-136): a stem joins the first group none of whose members it crosses
        for (int t = 0; t < nx; t++) {
            const int p = s_ord[t];
            const int pi = s_i[p], pj = s_j[p];
            unsigned long long blocked = 0ull;
            for (int q = lane; q < T; q += 64) {
                const int g = s_grp[q];
                if (g != 255 && sq_chain_cross(pi, pj, s_i[q], s_j[q])) blocked |= 1ull << g;
            }
            blocked = sq_wave_or64(blocked);
            int placed = blocked == ~0ull ? 64 : __ffsll((long long)~blocked) - 1;
            if (placed > ngroups) placed = ngroups;
            if (placed >= SQ_MAXLEVELS) { if (lane == 0) a.ctr->level_ovf = 1; placed = SQ_MAXLEVELS - 1; }   // (reported as an error)
            else if (placed == ngroups) { ngroups++; if (lane == 0) s_gsize[placed] = 0; }
            __syncthreads();
            if (lane == 0) { s_grp[p] = (uint8_t)placed; s_gsize[placed] += s_len[p]; }
            __syncthreads();
        }
        // groups ranked by size, descending, stable (:139); level = rank + 1
        if (lane < ngroups) {
            const int gs = s_gsize[lane];
            int r = 0;
            for (int h = 0; h < ngroups; h++) { const int hs = s_gsize[h]; r += (hs > gs || (hs == gs && h < lane)) ? 1 : 0; }
            s_rank[lane] = (uint8_t)(r + 1);
        }
        __syncthreads();
        for (int q = lane; q < T; q += 64) s_lvl[q] = s_rank[s_grp[q]];
        __syncthreads();
    }
    // ---- strands: the sorted list with the two new strands, into the structure's other buffer ----
    const int base = 4 * ch.toff;
    const int nxt = st.strand_off == base ? base + 2 * ch.tcap : base;
    const SqStrand *src = cio.strands + st.strand_off;
    const int16_t *ssrc = cio.sidx + st.strand_off;
    SqStrand *dst = cio.strands + nxt;
    int16_t *sdst = cio.sidx + nxt;
    const int ls = i0, rs = j0 - len + 1;                               // starts of the 5' and the 3' strand (ls < rs)
    int below_l = 0, below_r = 0;
    for (int q0 = 0; q0 < st.nstrand; q0 += 64) {
        const int q = q0 + lane;
        const bool valid = q < st.nstrand;
        SqStrand x = valid ? src[q] : SqStrand{0, 0, 0, 0, 0};
        const int sx = valid ? ssrc[q] : 0;
        const bool bl = valid && x.start < ls, br = valid && x.start < rs;
        if (valid) {
            if (anycross) x.level = s_lvl[sx];
            const int at = q + (bl ? 0 : 1) + (br ? 0 : 1);
            dst[at] = x; sdst[at] = (int16_t)sx;
        }
        below_l += __popcll(__ballot(bl)); below_r += __popcll(__ballot(br));
    }
    if (lane == 0) {
        const uint8_t lv = anycross ? s_lvl[k] : (uint8_t)1;
        dst[below_l] = SqStrand{(int16_t)ls, (int16_t)len, (int16_t)j0, lv, 1};
        dst[below_r + 1] = SqStrand{(int16_t)rs, (int16_t)len, (int16_t)(i0 + len - 1), lv, 0};
        sdst[below_l] = (int16_t)k; sdst[below_r + 1] = (int16_t)k;
        structs[b].strand_off = nxt;
        structs[b].nstrand = st.nstrand + 2;
        cio.chain[b].nstems = T;
        cio.chain[b].anycross = anycross ? 1 : 0;
    }
    if ((double)T == ch.maxstems) retire(T, 1);                         // :1168-1174 (checked before the next evaluation)
}

// start of a chain: the structure and chain records from pinned host memory (read in place: no copy engine, no
// stream wait), counters cleared
extern "C" __global__ __launch_bounds__(256) void sq_chain_init_kernel(const SqStruct *h_structs, const SqChain *h_chain, SqStruct *d_structs,
                                                                      SqChainIO cio, SqScanArgs a, int S, int first)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < S) { d_structs[q] = h_structs[q]; cio.chain[q] = h_chain[q]; }
    if (q == 0) {
        a.ctr->nout = 0; a.ctr->cand_ovf = 0; a.ctr->out_ovf = 0; a.ctr->level_ovf = 0;
        if (first) *cio.d_nfin = 0;
    }
}

// end of a chained round: counters and the number of finished structures, then the sequence number the host watches
extern "C" __global__ void sq_chain_done_kernel(SqRoundIO io, SqScanArgs a, SqChainIO cio, uint32_t seq)
{
    *io.h_ctr = *a.ctr;
    *cio.h_nfin = *cio.d_nfin;
    __threadfence_system();
    *io.h_seq = seq;
}
