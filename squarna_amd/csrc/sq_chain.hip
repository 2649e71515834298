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
#include "sq_extend.h"
#include "sq_tail_dev.h"

extern "C" __global__ __launch_bounds__(64) void sq_chain_kernel(SqDevCtx c, SqStruct *structs, SqScanArgs a, SqChainIO cio, int tmax)
{
    extern __shared__ __attribute__((aligned(16))) char sq_chain_dyn[];      // sq_extend_lds_bytes(tmax)
    SqExtendLds L = sq_extend_lds(sq_chain_dyn, tmax);
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
            // the device log (sq_tail_dev.hip): the structure's stems stay where they are, in the job's slice of cio.stems
            const uint32_t li = atomicAdd(&cio.fin_ctr[0], 1u);
            if (li < cio.fin_cap) cio.fin[li] = SqPoolFin{st.job, SQ_FIN_KIND_G0, 0, nstems, (uint32_t)ch.toff, SQ_FIN_SRC_CHAIN};
            else cio.fin_ctr[2] = 1;
            cio.job_evals[st.job] = (long long)nstems + (by_count ? 0 : 1);   // one evaluation per round it took part in
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
    if (lane == 0) cio.h_stems[ch.toff + k] = SqStemOut{i0, j0, len, 0, cd.bps, cd.fin};
    // the structure's other strand buffer takes the new list
    const int base = 4 * ch.toff;
    const int nxt = st.strand_off == base ? base + 2 * ch.tcap : base;
    const bool anycross = sq_extend_structure(L, a, gst, k, ch.anycross != 0, cio.strands + st.strand_off, cio.sidx + st.strand_off,
                                              st.nstrand, i0, j0, len, gst, cio.strands + nxt, cio.sidx + nxt, lane);
    const int T = k + 1;
    if (lane == 0) {
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
    sq_host_write_flush(cio.h_nfin);                     // (the stems and the list of finished structures: earlier kernels)
    *io.h_seq = seq;
}
