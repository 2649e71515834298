// sq_pool.hip -- a-5 / a-7 on the device for pools of any width (SQRNdbnseq.py:754-789, 1102-1199).
//
// The host-driven loop books every round on the CPU: ChooseStems' ordered conflict filter over the round's output
// records, one child structure per chosen stem (copies of the parent's stem and strand lists, strands re-sorted,
// pseudoknot levels recomputed), the pool's size / subopt bookkeeping.  With several batches in flight that work -- not
// the GPU -- bounded the throughput.  Here the structures of all pools live in device slots, two
// generations of them (parents, children), and a round is
//     state -> scan -> score -> sq_pool_choose_kernel -> sq_pool_scan_kernel -> sq_pool_extend_kernel
//   choose   one wave per structure: the survivors within subopt x best (:769-778), ordered like the reference's stable
//            descending sort (finalscore descending, emission key ascending), then the conflict filter (:779-789: a
//            candidate joins when it shares a base with EVERY stem chosen so far) and the stopper (:1147);
//   scan     one block: exclusive scan of the children counts = the children's slots (the next round's list keeps the
//            reference's order: parents in order, each one's children in ChooseStems' order), per job the next list
//            range, cursize / cursubopt (:1116-1120), the next round's size published to the host;
//   extend   one wave per parent: every child = parent + stem through sq_extend_structure (crossing weights, levels,
//            strands) into its slot of the other generation; structures without a new stem (:1155-1156) and children that
//            reach maxstemnum (:1123-1129) go to the pinned log of final structures with their place in finstemsets'
//            order (round, kind, position).
// The host launches the rounds with the exact grid (it reads the next round's size the scan kernel publishes, while
// the extend kernel is still running) and only sorts the log at the end.  A generation may hold more structures than
// the candidate arena has room for: state .. choose then run over the list in chunks that reuse the arena (what a
// structure leaves behind for scan / extend -- its chosen stems -- sits in per-slot arrays).  Any capacity overflow (slots, chosen stems
// per structure, survivors per structure, log) sets a flag and the host repeats the fold with its own loop.
#include <hip/hip_runtime.h>
#include "sq_device.h"
#include "sq_extend.h"
#include "sq_tail_dev.h"

#define SQ_POOL_NSURV 1024      // survivors within the range, per structure, the choose kernel sorts in LDS
#define SQ_POOL_CMAX 64         // stems ChooseStems may return for one structure (== lanes of the conflict test)

__device__ __forceinline__ bool sq_pool_shares_base(int ai, int aj, int al, int bi, int bj, int bl)   // :783-786
{
    const int as0 = ai, as1 = ai + al - 1, at0 = aj - al + 1, at1 = aj;
    const int bs0 = bi, bs1 = bi + bl - 1, bt0 = bj - bl + 1, bt1 = bj;
    return (as0 <= bs1 && bs0 <= as1) || (as0 <= bt1 && bt0 <= as1) || (at0 <= bs1 && bs0 <= at1) || (at0 <= bt1 && bt0 <= at1);
}

// start of a fold: the first generation (one empty structure per job), the job records and the header from pinned memory
extern "C" __global__ __launch_bounds__(256) void sq_pool_init_kernel(const SqStruct *h_structs, const SqChain *h_recs,
                                                                     const SqPoolJob *h_jobs, const int32_t *h_jobrec, int32_t *d_jobrec,
                                                                     int nbatchjobs, SqPoolIO pio, SqScanArgs a, int S0)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < S0) { pio.structs[q] = h_structs[q]; pio.recs[q] = h_recs[q]; }
    if (q < pio.njobs) pio.jobs[q] = h_jobs[q];
    if (q < nbatchjobs) d_jobrec[q] = h_jobrec[q];
    if (q == 0) {
        SqPoolHdr h;
        h.S[0] = (uint32_t)S0; h.S[1] = 0; h.round = 0; h.nfin = 0; h.nfin_stems = 0; h.ovf = 0; h.active_jobs = (uint32_t)pio.njobs; h.peak = (uint32_t)S0;
        *pio.hdr = h;
        a.ctr->nout = 0; a.ctr->cand_ovf = 0; a.ctr->out_ovf = 0; a.ctr->level_ovf = 0;
    }
}

extern "C" __global__ __launch_bounds__(64) void sq_pool_choose_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, SqPoolIO pio, int nsurv)
{
    // the survivors within the range, sorted in LDS: room for `nsurv` of them (18 bytes each) in the block's dynamic LDS --
    // 1,024 for long sequences, fewer for short ones (a block's LDS is LDS its neighbours on the CU cannot get); a
    // structure with more survivors in range raises the overflow flag and the host repeats the fold with its own loop
    extern __shared__ __attribute__((aligned(16))) char sq_choose_dyn[];
    double *const s_fin = reinterpret_cast<double *>(sq_choose_dyn);
    uint32_t *const s_q = reinterpret_cast<uint32_t *>(s_fin + nsurv), *const s_key = s_q + nsurv;
    uint16_t *const s_ord = reinterpret_cast<uint16_t *>(s_key + nsurv);
    __shared__ int s_ri[SQ_POOL_CMAX], s_rj[SQ_POOL_CMAX], s_rl[SQ_POOL_CMAX];
    const int lane = threadIdx.x;
    const SqStruct st = structs[blockIdx.x];                // (structs: the chunk of the round's list this launch covers)
    const int s = st.slot;                                  // == the structure's position in the round's list
    if (st.nstrand < 0) {                                   // a child that was full (:1123-1129): logged when it was made
        if (lane == 0) { pio.nchild[s] = 0; pio.finalflag[s] = 0; }
        return;
    }
    SqPoolJob *J = pio.jobs + pio.jobrec_of[st.job];
    if (lane == 0) atomicAdd((unsigned long long *)&J->evals, 1ull);
    const unsigned long long ob = a.best[st.slot];
    if (ob == 0ull) {                                       // no stem passed the thresholds: the structure is final (:1155)
        if (lane == 0) { pio.nchild[s] = 0; pio.finalflag[s] = 1; }
        return;
    }
    const SqJob jb = c.jobs[st.job];
    const uint32_t nok = a.ok_cnt[st.slot];
    const SqOk *oks = sq_oks(a, st, jb.cand_cap);
    SqPoolPick *out = pio.chosen + (size_t)s * pio.cmax;
    const bool one = J->cursize >= pio.poollim;             // :1147 stopper
    if (one) {
        // only ChooseStems' first element is used: the highest finalscore, the smallest emission key among equals
        const double bestfin = sq_unord(ob);
        unsigned long long pick = ~0ull;
        for (uint32_t q = lane; q < nok; q += 64) {
            const SqOk cd = oks[q];
            if (cd.fin == bestfin) { const unsigned long long v = ((unsigned long long)cd.key << 32) | q; pick = v < pick ? v : pick; }
        }
        pick = sq_wave_min64(pick);
        if (lane == 0) {
            if (pick == ~0ull) { pio.nchild[s] = 0; pio.finalflag[s] = 1; }
            else {
                const SqOk cd = oks[(uint32_t)pick];
                out[0] = SqPoolPick{cd.key, cd.len, cd.bps, cd.fin};
                pio.nchild[s] = 1; pio.finalflag[s] = 0;
            }
        }
        return;
    }
    // ---- the candidates within range (:769-778) ----
    const double range = J->cursubopt * sq_unord(ob);
    int n = 0;
    for (uint32_t q0 = 0; q0 < nok; q0 += 64) {
        const uint32_t q = q0 + lane;
        const bool valid = q < nok;
        SqOk cd;
        if (valid) cd = oks[q];
        const bool keep = valid && !(cd.fin < range);
        const unsigned long long m = __ballot(keep);
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < nsurv) { s_q[pos] = q; s_fin[pos] = cd.fin; s_key[pos] = cd.key; }
        n += __popcll(m);
    }
    if (n > nsurv) {
        if (lane == 0) { pio.hdr->ovf = 1; pio.nchild[s] = 0; pio.finalflag[s] = 0; }
        return;
    }
    __syncthreads();
    // ---- the reference's stable descending sort: finalscore descending, emission order (key) ascending ----
    for (int x = lane; x < n; x += 64) {
        const double fx = s_fin[x]; const uint32_t kx = s_key[x];
        int r = 0;
        for (int y = 0; y < n; y++) { const double fy = s_fin[y]; r += (fy > fx || (fy == fx && s_key[y] < kx)) ? 1 : 0; }
        s_ord[r] = (uint16_t)x;
    }
    __syncthreads();
    // ---- the conflict filter (:779-789): a candidate joins when it shares a base with every stem taken so far ----
    int nres = 0;
    bool over = false;
    for (int t = 0; t < n; t++) {
        const SqOk cd = oks[s_q[s_ord[t]]];
        const int ci = (int)(cd.key & 0xFFFFu), cj = (int)(cd.key >> 16) - ci, cl = (int)cd.len;
        const bool mine = lane < nres ? sq_pool_shares_base(ci, cj, cl, s_ri[lane], s_rj[lane], s_rl[lane]) : true;
        if (__ballot(!mine) != 0ull) continue;
        if (nres == SQ_POOL_CMAX || nres == pio.cmax) { over = true; break; }
        if (lane == 0) {
            s_ri[nres] = ci; s_rj[nres] = cj; s_rl[nres] = cl;
            out[nres] = SqPoolPick{cd.key, cd.len, cd.bps, cd.fin};
        }
        nres++;
        __syncthreads();
    }
    if (lane == 0) {
        if (over) { pio.hdr->ovf = 1; nres = 0; }
        pio.nchild[s] = nres; pio.finalflag[s] = (nres == 0 && !over) ? 1 : 0;
    }
}

// one block: the children's slots, the jobs' next ranges and pool state, the next round's size for the host
// (1,024 threads for a batch alone; 256 with batches in flight: a block of sixteen waves waits for a CU with four free slots
// on every SIMD, and a crowded chip -- one-wave matching blocks that stay for milliseconds -- rarely has one: 184 us per
// launch on average in the crowded trace of round 4, 5 us alone)
extern "C" __global__ __launch_bounds__(1024) void sq_pool_scan_kernel(SqPoolIO pio, SqScanArgs a, SqRoundIO io, int parity, uint32_t seq)
{
    __shared__ int s_part[1024];
    __shared__ int s_active;
    const int tid = threadIdx.x;
    SqPoolHdr *H = pio.hdr;
    const int S = (int)H->S[parity];
    const int nthr = (int)blockDim.x;
    const int ipt = (S + nthr - 1) / nthr;
    const int lo = min(tid * ipt, S), hi = min(lo + ipt, S);
    int sum = 0;
    for (int q = lo; q < hi; q++) sum += pio.nchild[q];
    s_part[tid] = sum;
    if (tid == 0) s_active = 0;
    __syncthreads();
    for (int d = 1; d < nthr; d <<= 1) {                     // inclusive scan of the per-thread sums
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;
    for (int q = lo; q < hi; q++) {
        const int nc = pio.nchild[q];
        pio.child_off[q] = run;
        // (sq_pool_round_kernel: a child builds itself at the start of its round, from its parent and its pick)
        // (parent in bits 0-25, the pick's index in ChooseStems' list in bits 26-31 -- cmax is 64 --: one dependent load less at
        // the child's entry than the parent's child_off would take)
        for (int k = 0; k < nc; k++) if (run + k < pio.slots) pio.parent_of[run + k] = (int32_t)((uint32_t)q | ((uint32_t)k << 26));
        run += nc;
    }
    const int total = s_part[nthr - 1];
    if (tid == 0) pio.child_off[S] = total;
    __syncthreads();
    const bool fits = total <= pio.slots && S <= (1 << 26);
    for (int j = tid; j < pio.njobs; j += nthr) {
        SqPoolJob J = pio.jobs[j];
        if (J.count == 0) continue;                          // the job's pool ran empty in an earlier round
        const int nf = pio.child_off[J.first], nl = pio.child_off[J.first + J.count];
        J.first = nf; J.count = nl - nf;
        if (J.count > J.cursize) {                           // :1116-1120, evaluated at the start of the next round
            J.cursize = J.count;
            if (J.cursubopt < J.suboptmax) J.cursubopt += J.suboptinc;
        }
        pio.jobs[j] = J;
        if (J.count > 0) atomicAdd(&s_active, 1);
    }
    __syncthreads();
    if (tid == 0) {
        if (!fits) H->ovf = 1;
        H->S[parity ^ 1] = fits ? (uint32_t)total : 0u;
        H->round += 1;
        H->active_jobs = (uint32_t)s_active;
        if ((uint32_t)total > H->peak && fits) H->peak = (uint32_t)total;
        if (pio.kept_ctr) {
            // kept lists: the next generation writes the pool its grandparents' lists came from -- nobody reads those any more
            if (pio.kept_ctr[parity] > pio.kept_ctr[2]) pio.kept_ctr[2] = pio.kept_ctr[parity];
            pio.kept_ctr[parity ^ 1] = 0u;
        }
        *io.h_ctr = *a.ctr;
        pio.h_hdr[seq % SQ_POOL_HDR_RING] = *H;
        sq_host_write_flush(io.h_ctr);
        *io.h_seq = seq;
    }
}

extern "C" __global__ __launch_bounds__(64) void sq_pool_extend_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, int parity)
{
    extern __shared__ __attribute__((aligned(16))) char sq_pool_dyn[];       // sq_extend_lds_bytes(pio.pt)
    SqExtendLds L = sq_extend_lds(sq_pool_dyn, pio.pt);
    const int s = blockIdx.x, lane = threadIdx.x;
    const size_t cur = (size_t)parity * pio.smax, nxt = (size_t)(parity ^ 1) * pio.smax;
    const SqStruct st = pio.structs[cur + s];
    const SqChain rec = pio.recs[cur + s];
    const int nch = pio.nchild[s];
    // (the scan kernel of this round has already advanced hdr.round: the round just evaluated is round - 1)
    const uint32_t round = pio.hdr->round - 1u;
    auto log_final = [&](uint32_t round_kind, int pos, const SqChainStem *stems, int nst) {
        uint32_t idx = 0, so = 0;
        if (lane == 0) sq_log_reserve(pio.fin_ctr, (uint32_t)nst, idx, so);
        idx = (uint32_t)__shfl((int)idx, 0, 64); so = (uint32_t)__shfl((int)so, 0, 64);
        if (idx >= pio.fin_cap || so + (uint32_t)nst > pio.fin_stem_cap) {
            if (lane == 0) {                                 // (the host repeats the fold with its own loop and an empty log)
                pio.hdr->ovf = 1; pio.fin_ctr[2] = 1;
                if (idx < pio.fin_cap) pio.fin[idx] = SqPoolFin{st.job, SQ_FIN_KIND_G0 + round_kind, pos, 0, 0u, SQ_FIN_SRC_LOG};
            }
            return;
        }
        for (int q = lane; q < nst; q += 64) { const SqChainStem x = stems[q]; pio.fin_stems[so + q] = SqPoolStem{(int16_t)x.i, (int16_t)x.j, (int16_t)x.len, 0}; }
        if (lane == 0) pio.fin[idx] = SqPoolFin{st.job, SQ_FIN_KIND_G0 + round_kind, pos, nst, so, SQ_FIN_SRC_LOG};
    };
    if (st.nstrand < 0) return;                              // full child of the previous round: logged then
    const SqChainStem *pst = pio.stems + rec.toff;
    // grid (parents, Y): block (s, y) makes the children y, y + Y, ... of parent s -- the children of a parent are
    // independent (each in its own slot), and one wave per parent serialised them (the longest kernel of a crowded round)
    if (nch == 0) {
        if (blockIdx.y == 0 && pio.finalflag[s]) log_final(2u * round + 1u, s, pst, rec.nstems);
        return;
    }
    if ((int)blockIdx.y >= nch) return;
    if (rec.nstems >= pio.pt) { if (lane == 0) pio.hdr->ovf = 1; return; }
    const SqStrand *psrc = pio.strands + st.strand_off;
    const int16_t *pssrc = pio.sidx + st.strand_off;
    const int c0 = pio.child_off[s];
    const SqPoolPick *picks = pio.chosen + (size_t)s * pio.cmax;
    for (int k = blockIdx.y; k < nch; k += gridDim.y) {
        const int cslot = c0 + k;
        if (cslot >= pio.slots) { if (lane == 0) pio.hdr->ovf = 1; return; }
        const SqPoolPick pk = picks[k];
        const int i0 = (int)(pk.key & 0xFFFFu), j0 = (int)(pk.key >> 16) - i0, len = (int)pk.len;
        const int toff = (int)((nxt + (size_t)cslot) * (size_t)pio.pt);
        SqChainStem *cst = pio.stems + toff;
        const bool anyc = sq_extend_structure(L, a, pst, rec.nstems, rec.anycross != 0, psrc, pssrc, st.nstrand, i0, j0, len,
                                              cst, pio.strands + 2 * (size_t)toff, pio.sidx + 2 * (size_t)toff, lane);
        __syncthreads();
        const int T = rec.nstems + 1;
        const bool full = (double)T == rec.maxstems;
        if (lane == 0) {
            SqStruct cs;
            cs.job = st.job; cs.strand_off = 2 * toff; cs.nstrand = full ? -1 : st.nstrand + 2; cs.slot = cslot;
            // the range factor the NEXT round's choose kernel will apply (the scan kernel has already advanced it, :1116-1120):
            // the scoring kernel bounds its candidates with it
            cs.subopt = pio.jobs[pio.jobrec_of[st.job]].cursubopt;
            cs.cand_off = (int64_t)(cslot % pio.chunk) * pio.maxcap;
            pio.structs[nxt + cslot] = cs;
            SqChain cr;
            cr.toff = toff; cr.tcap = pio.pt; cr.nstems = T; cr.anycross = anyc ? 1 : 0; cr.maxstems = rec.maxstems;
            pio.recs[nxt + cslot] = cr;
        }
        if (full) {                                          // :1123-1129: moved to finstemsets at the start of the next round
            __threadfence_block();
            __syncthreads();
            log_final(2u * (round + 1u), cslot, cst, T);
        }
        __syncthreads();
    }
}

// end of a fold: the final header and the job records (evaluation counts) for the host
extern "C" __global__ __launch_bounds__(256) void sq_pool_publish_kernel(SqPoolIO pio, SqScanArgs a, SqRoundIO io, uint32_t seq)
{
    for (int j = threadIdx.x; j < pio.njobs; j += 256) { const SqPoolJob J = pio.jobs[j]; pio.h_jobs[j] = J; pio.job_evals[J.job] = J.evals; }
    __syncthreads();
    if (threadIdx.x == 0) {
        pio.hdr->nfin = pio.fin_ctr[0]; pio.hdr->nfin_stems = pio.fin_ctr[1];
        *io.h_ctr = *a.ctr;
        pio.h_hdr[seq % SQ_POOL_HDR_RING] = *pio.hdr;
        sq_host_write_flush(io.h_ctr);                   // (the log of final structures: the extend kernels)
        *io.h_seq = seq;
    }
}
