// sq_batch.hip -- the workspace layout of a batch, sq_batch_workspace_bytes, sq_batch_create, sq_batch_destroy (C ABI: include/squarna_hip.h).
#include "sq_host_int.h"


// row pitch of the scan matrix: ld == 1 (mod 32) (aligned anti-diagonal walk, see sq_kernels.hip) and
// (ld - 1) / 32 odd, so that the byte stride between rows is an odd multiple of 128 B and consecutive rows
// of a wave rotate through all memory channels instead of camping on a power-of-two stride
// diagonal bit matrix of a job (sq_bits_kernel): nw word-rows of bpitch words
static inline int32_t bits_nw(int n) { return (n + 31) / 32; }
static inline int32_t bits_pitch(int n) { return (int32_t)align_up((size_t)2 * n, 64) + 64; }
static inline int32_t ld_of(int n)
{
    int k = (std::max(n, 1) - 1 + 31) / 32;
    if (!getenv("SQ_LD_POW2") && (k & 1) == 0) k++;
    return 32 * k + 1;
}

// most stems one structure of a job can hold: they are disjoint and have at least ceil(minlen) (>= 1) base pairs
// every reactivity of a sequence 0.5 (SQRNdbnseq.py:273: the record came without reactivities)?  Block-wise without a branch
// per element, so that the compiler vectorises the comparison: most records of a big input are like this
static inline bool all_half(const double *r, int n)
{
    int i = 0;
    for (; i + 32 <= n; i += 32) {
        bool ok = true;
        for (int k = 0; k < 32; k++) ok &= r[i + k] == 0.5;
        if (!ok) return false;
    }
    for (; i < n; i++) if (r[i] != 0.5) return false;
    return true;
}


// fp32 score matrices are planned unless the caller opts out
static inline bool want_fp32(const sq_batch_desc *d)
{
    return !(d->batch_flags & SQ_BATCH_NO_FP32);
}

// stemscore ** 1.7 table of a paramset (SqPsetDev::pow_off): entries needed for sequences up to maxn nt, 0 when the
// paramset does not qualify (weights not multiples of 2^-10, no E / H algorithm, minbpscore <= 0, table beyond 4 Mi entries)
static int64_t pow17_entries(const sq_paramset &ps, int maxn, double &scale)
{
    scale = 1.0;
    if (!(ps.algorithms & (SQ_ALGO_E | SQ_ALGO_H)) || !(ps.minbpscore > 0)) return 0;
    int q = 0;
    double maxw = 0;
    for (int k = 0; k < 32 * 32; k++) {
        if (!ps.inbps[k]) continue;
        const double w = ps.bpweight[k];
        if (!(std::fabs(w) <= 1024.0)) return 0;
        while (q <= 10 && w * std::ldexp(1.0, q) != std::floor(w * std::ldexp(1.0, q))) q++;
        if (q > 10) return 0;
        maxw = std::max(maxw, w);
    }
    if (!(maxw > 0)) return 0;
    scale = std::ldexp(1.0, q);
    const double entries = std::floor((double)(maxn / 2 + 1) * maxw * scale) + 2;
    return entries <= (double)((int64_t)4 << 20) ? (int64_t)entries : 0;
}

namespace {
struct Layout {
    size_t off_codes, off_flags, off_inc4, off_chain, off_e0, off_reacts, off_ridx, off_jobs, off_psets, off_sdf, off_rftab, off_powtab;
    int64_t n_rftab, pow_entries;
    size_t off_mat32, off_mat64, off_structs, off_strands, off_state, off_cnt, off_ctr, off_cands, off_out;
    size_t off_bits, off_rbpk, off_fb;
    size_t off_ctx_rec = 0, off_ctx_depth = 0, off_ctx_rmq = 0, off_ctx_ok = 0;   // ScoreStems context tables (sq_context.h)
    int ctx_cap = 0, ctx_levels = 0;
    size_t off_crec, off_cstems, off_cstrands, off_csidx, off_cnfin;   // device-chained rounds (sq_chain.hip)
    size_t off_pstructs, off_precs, off_pstems, off_pstrands, off_psidx, off_pjobs, off_pjobrec, off_pnchild, off_pchoff,
           off_pflag, off_pchosen, off_pparent, off_phdr;                            // device pools (sq_pool.hip)
    size_t off_kctr, off_kcnt, off_ktab, off_kpages; uint32_t kept_pages;               // kept lists of the pools (sq_device.h: SqKept; 0 pages: none)
    // device log of final structures + scratch of the device tail (sq_tail_dev.hip)
    size_t off_fin, off_fin_stems, off_fin_ctr, off_jobevals, off_t_jobs, off_t_seqjob0, off_t_ord, off_t_cstems, off_t_csn, off_t_hash,
           off_t_rep, off_t_mask, off_t_scores, off_t_dlist, off_t_rlist, off_t_seqs, off_t_refp, off_t_refn, off_t_pow;
    uint32_t fin_cap, fin_stem_cap; int32_t pow_len;
    size_t off_mulcols;              // alignment columns of every position (shared L x L weighting matrix), else unused
    size_t off_mulM;                 // the shared weighting matrix, diagonal-major (sq_gather.hip)
    bool mul_direct;                 // jobs weighted by the shared matrix read it through the gap map; false (SQ_MUL_GATHER=1): per-job slices
    size_t off_algo, algo_bytes;     // scratch of the Hungarian / Nussinov kernels (Edmonds borrows the end of the candidate arena)
    int32_t pool_pt;                 // stems per slot (0: no device pools for this batch)
    int64_t chain_T;                 // summed stem capacity of all jobs
    size_t total;
    int64_t ltot, sdf_len, mat32_floats, mat64_doubles, cand_records, bits_words;
    int32_t maxn, stride, max_structs, strand_cap, cpn, fbstride;
    uint32_t out_cap;
};

int plan(const sq_batch_desc *d, Layout &L)
{
    if (!d || d->nseq <= 0 || d->njobs <= 0 || d->npset <= 0) { sq_set_error("empty batch"); return -1; }
    L.ltot = d->seq_off[d->nseq];
    L.maxn = 0;
    for (int s = 0; s < d->nseq; s++) L.maxn = std::max(L.maxn, d->seq_off[s + 1] - d->seq_off[s]);
    if (L.maxn > 32000) { sq_set_error("sequence longer than 32000 nt"); return -1; }
    L.max_structs = d->max_structs > 0 ? d->max_structs : 4096;
    L.cpn = d->cand_per_nt > 0 ? d->cand_per_nt : 32;
    L.mat32_floats = 0; L.mat64_doubles = 0; L.bits_words = 0;
    L.mul_direct = d->mul_matrix_dev != nullptr && getenv("SQ_MUL_GATHER") == nullptr;
    int64_t sum_cap = 0;
    for (int j = 0; j < d->njobs; j++) {
        const int s = d->job_seq[j];
        if (s < 0 || s >= d->nseq || d->job_pset[j] < 0 || d->job_pset[j] >= d->npset) { sq_set_error("bad job"); return -1; }
        const int64_t n = d->seq_off[s + 1] - d->seq_off[s];
        const bool ext_any = (d->ext_score && d->ext_score[j]) || (d->mul_score && d->mul_score[j]) ||
                             (d->bpp_term && d->bpp_term[j]) || (d->mul_shared && d->mul_shared[j]);
        // (jobs weighted by the shared stem matrix need no fp32 matrix: their product is formed by the gather kernel)
        const bool shared_only = d->mul_shared && d->mul_shared[j] && !(d->ext_score && d->ext_score[j]);
        if (want_fp32(d) || (ext_any && !shared_only)) L.mat32_floats += (int64_t)align_up((size_t)(n * ld_of((int)n)), 64);
        L.bits_words += (int64_t)bits_nw((int)n) * bits_pitch((int)n);
        const bool ext = d->ext_score && d->ext_score[j];
        const bool mul = (d->mul_score && d->mul_score[j]) || (d->bpp_term && d->bpp_term[j]) ||
                         (d->mul_shared && d->mul_shared[j] && !L.mul_direct);
        if (ext) L.mat64_doubles += 2 * n * n;
        else if (mul) L.mat64_doubles += n * n;
    }
    L.mat32_floats += 1024 + (int64_t)160 * ld_of(L.maxn);     // reads of rows past a short segment stay inside the arena
    L.sdf_len = 0;
    for (int p = 0; p < d->npset; p++) {
        const double bw = d->psets[p].bracketweight;
        if (bw == std::floor(bw) && std::fabs(bw) <= 64) L.sdf_len += (int64_t)std::max(1.0, std::fabs(bw)) * L.maxn + 16;
    }
    L.stride = (int32_t)align_up((size_t)L.maxn + 2, 32);
    L.fbstride = 2 * (L.stride / 32 + 8);
    L.strand_cap = (int32_t)std::min<int64_t>((int64_t)L.max_structs * 64 + L.maxn, 1 << 24);
    int64_t maxcap = (int64_t)L.cpn * L.maxn + 256;
    {
        std::vector<double> runs(d->npset);                 // share of the cells that start a maximal run of >= minlen (per paramset)
        for (int p = 0; p < d->npset; p++) runs[p] = std::pow(0.375, std::max(1.0, std::ceil(d->psets[p].minlen)) - 1.0);
        for (int j = 0; j < d->njobs; j++) {
            const int sq = d->job_seq[j];
            const double nn = d->seq_off[sq + 1] - d->seq_off[sq];
            const int64_t cj = (int64_t)(0.117 * nn * nn * runs[d->job_pset[j]] * 1.6 + 256);
            maxcap = std::max<int64_t>(maxcap, cj);
            sum_cap += std::max<int64_t>(cj, (int64_t)L.cpn * (int64_t)nn);
        }
    }
    L.cand_records = std::min<int64_t>((int64_t)L.max_structs * maxcap, (int64_t)160 << 20);   // (5 GiB of 32-byte records at most)
    // ... unless ONE structure per job needs more: the rows of a long alignment (512 x 4,700 nt: 50 MB of run records each) fold
    // as chains of ONE launch when the arena holds them all -- five launches of a fifth of the chip's blocks otherwise (40 GiB at most)
    if (d->mul_matrix_dev && L.mul_direct) L.cand_records = std::max<int64_t>(L.cand_records, std::min<int64_t>(sum_cap, (int64_t)1280 << 20));
    L.cand_records = std::max<int64_t>(L.cand_records, maxcap);
    // the dense fp64 read-back (sq_bpmatrix_read) borrows the candidate arena
    L.cand_records = std::max<int64_t>(L.cand_records, (int64_t)(2 * (int64_t)L.maxn * L.maxn * 8 / sizeof(SqCand)) + 16);
    L.out_cap = (uint32_t)std::min<int64_t>(L.cand_records, (int64_t)4 << 20);
    if (const char *e = getenv("SQ_OUT_CAP")) L.out_cap = (uint32_t)std::min<int64_t>(L.out_cap, std::max(64, atoi(e)));   // (tests: rounds split on output overflow)
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
    L.off_codes = take(L.ltot); L.off_flags = take(L.ltot); L.off_inc4 = take(L.ltot);
    L.off_chain = take(L.ltot * 2); L.off_e0 = take(L.ltot * 2); L.off_reacts = take(L.ltot * 8); L.off_ridx = take(L.ltot);
    L.off_jobs = take(sizeof(SqJob) * d->njobs); L.off_psets = take(sizeof(SqPsetDev) * d->npset);
    L.off_sdf = take(8 * (size_t)std::max<int64_t>(L.sdf_len, 1));
    // one 16 x 16 reactfactor table per sequence whose reactivities are not all 0.5 (sq_batch_create fills the ones
    // whose reactivities take <= 16 values)
    L.n_rftab = 0;
    for (int s = 0; s < d->nseq; s++)
        if (d->reacts && !all_half(d->reacts + d->seq_off[s], d->seq_off[s + 1] - d->seq_off[s])) L.n_rftab++;
    L.off_rftab = take(8 * 256 * (size_t)std::max<int64_t>(L.n_rftab, 1));
    L.pow_entries = 0;
    for (int p = 0; p < d->npset; p++) { double sc; L.pow_entries += pow17_entries(d->psets[p], L.maxn, sc); }
    L.off_powtab = take(8 * (size_t)std::max<int64_t>(L.pow_entries, 1));
    L.off_mat32 = take(4 * (size_t)L.mat32_floats);
    L.off_mat64 = take(8 * (size_t)std::max<int64_t>(L.mat64_doubles, 1));
    L.off_structs = take(sizeof(SqStruct) * L.max_structs);
    L.off_strands = take(sizeof(SqStrand) * (size_t)L.strand_cap);
    L.off_state = take((size_t)4 * 2 * L.stride * L.max_structs);
    L.off_cnt = take(16 * (size_t)align_up((size_t)L.max_structs, 2));   // cand_cnt (u32), best (u64), ok_cnt (u32) per slot
    L.off_ctr = take(2 * 64);                               // one SqCounters per fold lane (64 bytes apart)
    L.off_cands = take(sizeof(SqCand) * (size_t)L.cand_records);
    L.off_out = take(sizeof(SqOut) * (size_t)L.out_cap);
    L.off_bits = take(4 * (size_t)std::max<int64_t>(L.bits_words, 1));
    L.off_rbpk = take(4 * (size_t)std::max<int>(d->rbp_off[d->nseq], 1));
    L.off_fb = take(4 * (size_t)L.fbstride * L.max_structs);
    {
        // ScoreStems' closed-form strand sweep (sq_context.h): tables for every structure of a launch, for batches with
        // sequences long enough that the walk over the strands is what the scoring kernel waits for
        // (measured, whole fold with / without the tables: 10,000 x 300 nt 4.5 / 4.1 ms -- the context kernel costs more than the
        // short walks it replaces --, 1,024 x 1000 nt one fold alone 4.6 / 4.8 (scoring kernel 2.18 / 2.48), two sub-batches side by
        // side 4.45 / 4.2, 1,000 x 2000 nt 27.0 / 29.6: from 800 nt on)
        const int ctx_min_n = getenv("SQ_CTX_MIN_N") ? atoi(getenv("SQ_CTX_MIN_N")) : 800;
        int pt_max = 1;
        for (int j = 0; j < d->njobs; j++)
            pt_max = std::max(pt_max, chain_tcap(d->seq_off[d->job_seq[j] + 1] - d->seq_off[d->job_seq[j]], d->psets[d->job_pset[j]].minlen));
        const int cap = std::min(1024, 2 * pt_max + 2) + 1;
        int lv = 0;
        const size_t per_gap = sq_context_bytes_per_gap(cap, &lv);
        const size_t total = per_gap * (size_t)cap * (size_t)L.max_structs;
        if (ctx_min_n >= 0 && L.maxn >= ctx_min_n && total <= ((size_t)2 << 30)) {
            L.ctx_cap = cap; L.ctx_levels = lv;
            L.off_ctx_rec = take(sizeof(SqCtxRec) * (size_t)cap * L.max_structs);
            L.off_ctx_depth = take(2 * (size_t)cap * L.max_structs);
            L.off_ctx_rmq = take(2 * (size_t)lv * cap * L.max_structs);
            L.off_ctx_ok = take((size_t)L.max_structs);
        }
    }
    // chained rounds: per job, room for the most stems a structure can hold (disjoint stems of >= minlen pairs)
    L.chain_T = 0;
    for (int j = 0; j < d->njobs; j++) L.chain_T += chain_tcap(d->seq_off[d->job_seq[j] + 1] - d->seq_off[d->job_seq[j]], d->psets[d->job_pset[j]].minlen);
    L.off_crec = take(sizeof(SqChain) * (size_t)d->njobs);
    L.off_cstems = take(sizeof(SqChainStem) * (size_t)L.chain_T);
    L.off_cstrands = take(sizeof(SqStrand) * 4 * (size_t)L.chain_T);
    L.off_csidx = take(sizeof(int16_t) * 4 * (size_t)L.chain_T);
    L.off_cnfin = take(64);
    // device pools: two generations of max_structs slots, each with room for the most stems any job's structure can hold
    L.pool_pt = 0;
    for (int j = 0; j < d->njobs; j++)
        L.pool_pt = std::max(L.pool_pt, chain_tcap(d->seq_off[d->job_seq[j] + 1] - d->seq_off[d->job_seq[j]], d->psets[d->job_pset[j]].minlen));
    // (such batches keep the host-driven loop: lists longer than the level scratch holds; slot offsets beyond 31 bits)
    if (L.pool_pt > SQ_CHAIN_TMAX || 8 * (int64_t)L.max_structs * L.pool_pt >= ((int64_t)1 << 31)) L.pool_pt = 0;
    {
        const size_t sm = (size_t)L.max_structs, pt = (size_t)L.pool_pt;
        const size_t on = pt ? 1 : 0;
        L.off_pstructs = take(on * 2 * sm * sizeof(SqStruct)); L.off_precs = take(on * 2 * sm * sizeof(SqChain));
        L.off_pstems = take(on * 2 * sm * pt * sizeof(SqChainStem)); L.off_pstrands = take(on * 2 * sm * 2 * pt * sizeof(SqStrand));
        L.off_psidx = take(on * 2 * sm * 2 * pt * sizeof(int16_t));
        L.off_pjobs = take(on * (size_t)d->njobs * sizeof(SqPoolJob)); L.off_pjobrec = take(on * (size_t)d->njobs * 4);
        L.off_pnchild = take(on * sm * 4); L.off_pchoff = take(on * (sm + 1) * 4); L.off_pflag = take(on * sm);
        L.off_pchosen = take(on * 2 * sm * 64 * sizeof(SqPoolPick)); L.off_pparent = take(on * sm * 4); L.off_phdr = take(64);
        // kept lists (SQ_BATCH_POOL_LISTS, sequences beyond the scanning round kernel's 256 nt): the counters, a row of page
        // numbers per slot and generation, the pages
        L.kept_pages = 0; L.off_kctr = L.off_kcnt = L.off_ktab = L.off_kpages = 0;
        if (on && (d->batch_flags & SQ_BATCH_POOL_LISTS) && L.maxn > 256 && L.maxn <= 1024 && !getenv("SQ_NO_POOL_KEPT")) {
            // (pages per slot and generation: measured at 500 nt, 4.0 pages per structure of the largest generation, which fills
            // half of the slots; a list grows with the square of the length)
            const double pps = getenv("SQ_KEPT_PPS") ? std::max(0.25, atof(getenv("SQ_KEPT_PPS"))) : std::max(1.0, 3.0 * ((double)L.maxn / 500.0) * ((double)L.maxn / 500.0));
            const double gb = getenv("SQ_KEPT_GB") ? std::max(0.01, atof(getenv("SQ_KEPT_GB"))) : 48.0;
            const double np = std::min((double)sm * pps, gb * 1073741824.0 / 2.0 / (double)SQ_KEPT_PAGE_BYTES);
            L.kept_pages = (uint32_t)std::max(64.0, std::min(np, 4.0e9));
            L.off_kctr = take(256); L.off_kcnt = take(2 * sm * 4); L.off_ktab = take(2 * sm * SQ_KEPT_TAB * 4);
            L.off_kpages = take(2 * (size_t)L.kept_pages * SQ_KEPT_PAGE_BYTES);
        }
    }
    {
        // the log of final structures: every structure of every pool ends there once -- measured: 1.4 x the largest generation.
        // Two entries per structure slot (65,536 at least, 4 Mi at most) + one per job (chained rounds, E / H / N stemsets),
        // with a third of the most stems a structure can hold each (8 .. 128) + every job's stem capacity once
        int pt_any = 1;
        for (int j = 0; j < d->njobs; j++)
            pt_any = std::max(pt_any, chain_tcap(d->seq_off[d->job_seq[j] + 1] - d->seq_off[d->job_seq[j]], d->psets[d->job_pset[j]].minlen));
        const int64_t want = std::min<int64_t>(std::max<int64_t>(65536, 2 * (int64_t)L.max_structs), (int64_t)4 << 20);
        L.fin_cap = (uint32_t)(want + 2 * (int64_t)d->njobs);
        L.fin_stem_cap = (uint32_t)std::min<int64_t>(std::min<int64_t>(want * std::min(std::max(pt_any / 3, 8), 128), (int64_t)48 << 20) + 2 * L.chain_T,
                                                     (int64_t)0x7FFFFFF0);
        if (const char *e = getenv("SQ_FIN_STEM_CAP")) L.fin_stem_cap = (uint32_t)std::min<int64_t>(L.fin_stem_cap, std::max(16, atoi(e)));   // (tests: the log's stem room runs out)
        L.pow_len = 4 * L.maxn + 16;
        const size_t fc = L.fin_cap;
        L.off_fin = take(sizeof(SqPoolFin) * fc); L.off_fin_stems = take(sizeof(SqPoolStem) * (size_t)L.fin_stem_cap);
        L.off_fin_ctr = take(64); L.off_jobevals = take(8 * (size_t)d->njobs);
        L.off_t_jobs = take(3 * 4 * ((size_t)d->njobs + 1)); L.off_t_seqjob0 = take(4 * ((size_t)d->nseq + 1));
        L.off_t_ord = take(2 * 4 * fc); L.off_t_cstems = take(sizeof(SqPoolStem) * ((size_t)L.fin_stem_cap + (size_t)L.chain_T));
        L.off_t_csn = take(4 * fc); L.off_t_hash = take(8 * fc); L.off_t_rep = take(4 * fc); L.off_t_mask = take(8 * fc);
        L.off_t_scores = take(8 * (3 * fc + 16 * (size_t)d->nseq)); L.off_t_dlist = take(4 * fc); L.off_t_rlist = take(4 * fc);
        L.off_t_seqs = take(sizeof(SqTailSeq) * (size_t)d->nseq);
        L.off_t_refp = take(2 * (size_t)L.ltot); L.off_t_refn = take(4 * (size_t)d->nseq);
        L.off_t_pow = take(8 * (size_t)L.pow_len);
    }
    L.off_mulcols = take(d->mul_matrix_dev ? 4 * (size_t)L.ltot : 0);
    L.off_mulM = take(d->mul_matrix_dev ? 8 * (size_t)d->mul_L * (size_t)d->mul_L : 0);
    // Hungarian and Nussinov: their scratch (n x n tables) is known from the lengths, so they get room of their own and
    // always run beside the greedy rounds (16 GB at most; what does not fit borrows from the candidate arena like Edmonds)
    {
        size_t need = 0;
        for (int j = 0; j < d->njobs; j++) {
            const uint32_t al = d->psets[d->job_pset[j]].algorithms;
            const size_t n = (size_t)(d->seq_off[d->job_seq[j] + 1] - d->seq_off[d->job_seq[j]]);
            const size_t edges = 4 * n * n + 8192;                 // (positive cells: ~0.19 n^2 of 16 bytes; job / result records)
            if (al & SQ_ALGO_H) need += align_up(sq_lsap_scratch_bytes((int)n), 256) + edges;
            if (al & SQ_ALGO_N) need += align_up(sq_nussinov_scratch_bytes((int)n), 256) + edges;
        }
        L.algo_bytes = std::min<size_t>(need ? need + 65536 : 0, (size_t)16 << 30);
        L.off_algo = take(L.algo_bytes);
    }
    L.total = o;
    return 0;
}
}  // namespace

extern "C" int sq_batch_workspace_bytes(const sq_batch_desc *desc, size_t *bytes)
{
    Layout L;
    int r = plan(desc, L);
    if (r) return r;
    *bytes = L.total;
    return 0;
}

extern "C" int sq_batch_create(sq_batch **out, const sq_batch_desc *d, void *ws, size_t ws_bytes, void *hip_stream)
{
#ifdef SQ_CREATE_PROF
    // (phase timers of this function: SQ_DEFS=-DSQ_CREATE_PROF python -m squarna_amd.build; one line per call on stderr)
    std::vector<std::pair<const char *, double>> _cp; _cp.emplace_back("start", now_s());
#endif
    Layout L;
    int r = plan(d, L);
    if (r) return r;
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("plan", now_s());
#endif
    if (!ws || ws_bytes < L.total) { sq_set_error("workspace too small"); return -2; }
    if (((uintptr_t)ws & 255) != 0) { sq_set_error("workspace must be 256-byte aligned"); return -2; }
    for (int j = 0; j < d->njobs; j++) {
        const bool term = d->bpp_term && d->bpp_term[j];
        if (d->psets[d->job_pset[j]].bpp != 0 && !d->bpp_term) {
            sq_set_error("bpp != 0 paramsets need bpp_term: (bppm/max)^|bpp| from ViennaRNA's base-pair probabilities (SQRNdbnseq.py:341-364)");
            return -4;
        }
        if (term && d->psets[d->job_pset[j]].bpp == 0) { sq_set_error("bpp_term given for a paramset with bpp == 0"); return -1; }
        const bool shared = d->mul_shared && d->mul_shared[j];
        if (shared && (!d->mul_matrix_dev || !d->mul_cols || d->mul_L <= 0)) { sq_set_error("mul_shared without mul_matrix_dev / mul_cols / mul_L"); return -1; }
        if (shared && (term || (d->mul_score && d->mul_score[j]) || (d->ext_score && d->ext_score[j]))) {
            sq_set_error("a job takes either the shared weighting matrix or its own matrices, not both"); return -4;
        }
        if (term && ((d->mul_score && d->mul_score[j]) || (d->ext_score && d->ext_score[j]))) {
            sq_set_error("a job takes either bpp_term or mul_score / caller matrices, not both"); return -4;
        }
    }
    sq_batch *b = new sq_batch();
    sq_read_fold_switches(b->sw);                           // (every sq_fold refreshes them; the per-call ops read these)
    b->stream = (hipStream_t)hip_stream;
    if (hipGetDevice(&b->device) != hipSuccess) b->device = -1;      // the caller's current device: every thread the library spawns adopts it
    b->nseq = d->nseq; b->npset = d->npset; b->njobs = d->njobs; b->maxn = L.maxn; b->ltot = L.ltot;
    b->seq_off.assign(d->seq_off, d->seq_off + d->nseq + 1);
    b->codes.assign(d->codes, d->codes + L.ltot);
    b->seq_has_sep.assign((size_t)d->nseq, 0);
    for (int s2 = 0; s2 < d->nseq; s2++)
        for (int64_t q = d->seq_off[s2]; q < d->seq_off[s2 + 1]; q++)
            if (d->codes[q] == SQ_CODE_SEP1 || d->codes[q] == SQ_CODE_SEP2) { b->seq_has_sep[(size_t)s2] = 1; break; }
    b->flags.assign(d->flags, d->flags + L.ltot);
    b->reacts_null = d->reacts == nullptr;
    if (d->reacts) b->reacts.assign(d->reacts, d->reacts + L.ltot);   // (NULL: 0.5 everywhere -- sq_host_reacts forms the array if a host path asks)
    b->rbp_off.assign(d->rbp_off, d->rbp_off + d->nseq + 1);
    b->rbps.assign(d->rbps, d->rbps + 2 * (size_t)d->rbp_off[d->nseq]);
    b->job_seq.assign(d->job_seq, d->job_seq + d->njobs);
    b->job_pset.assign(d->job_pset, d->job_pset + d->njobs);
    b->psets.assign(d->psets, d->psets + d->npset);
    b->interchainonly = d->interchainonly;
    b->max_structs = L.max_structs; b->cand_per_nt = L.cpn;
    b->cand_records = L.cand_records; b->out_cap = L.out_cap; b->strand_cap = L.strand_cap;
    b->mat32_bytes = 4 * (size_t)L.mat32_floats;
    b->has_fp32 = want_fp32(d);

#ifdef SQ_CREATE_PROF
    _cp.emplace_back("copies", now_s());
#endif
    char *base = (char *)ws;
    // ---- per-position derived arrays (host, O(N)) ----
    std::vector<uint8_t> inc4(L.ltot);
    std::vector<int16_t> chain(L.ltot, 0);
    std::vector<uint8_t> e0(L.ltot, 0);
    {
        uint32_t seen = 0;
        for (uint8_t cd : b->codes) if (cd < 29) seen |= 1u << cd;
        b->nletters = __builtin_popcount(seen);
    }
    for (int s = 0; s < d->nseq; s++) {
        const int off = d->seq_off[s], n = d->seq_off[s + 1] - off;
        auto sep = [&](int p) { return b->codes[off + p] == SQ_CODE_SEP1 || b->codes[off + p] == SQ_CODE_SEP2; };
        {
            // one chain (no separator in the sequence -- nearly every record): minimum span 4 everywhere, chain 0; only the
            // restraint pairs below are left to do
            bool anysep = false;
            const uint8_t *cd = b->codes.data() + off;
            for (int i = 0; i < n; i++) anysep |= (cd[i] == SQ_CODE_SEP1) | (cd[i] == SQ_CODE_SEP2);
            if (!anysep && d->rbp_off[s + 1] == d->rbp_off[s]) {
                std::fill(inc4.begin() + off, inc4.begin() + off + n, (uint8_t)4);
                continue;
            }
        }
        int curr = 0;
        for (int i = 0; i < n; i++) {
            int v = 4;                                   // SQRNdbnseq.py:294-297
            for (int chk = 1; chk <= 2; chk++)
                if (i + chk < n && sep(i + chk)) v = chk + 1;
            inc4[off + i] = (uint8_t)v;
            if (sep(i)) curr++;                          // :264-271
            else chain[off + i] = (int16_t)curr;
        }
        for (int k = d->rbp_off[s]; k < d->rbp_off[s + 1]; k++) {
            const int v = d->rbps[2 * k], w = d->rbps[2 * k + 1];
            if (v < 0 || w >= n || v >= w) { delete b; sq_set_error("bad restraint pair"); return -1; }
            if (e0[off + v] || e0[off + w]) { delete b; sq_set_error("a position in two restraint base pairs"); return -1; }
            e0[off + v] = 1; e0[off + w] = 1;                  // 1: end of a restraint pair (0: free, 255: masked by the structure)
        }
    }
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("positions", now_s());
#endif
    // ---- paramsets with host-libm pow tables ----
    std::vector<SqPsetDev> pd(d->npset);
    std::vector<double> sdf, powtab;
    for (int p = 0; p < d->npset; p++) {
        const sq_paramset &ps = d->psets[p];
        SqPsetDev &x = pd[p];
        memset(&x, 0, sizeof x);
        memcpy(x.w, ps.bpweight, sizeof x.w);
        memcpy(x.inbps, ps.inbps, sizeof x.inbps);
        for (int a = 0; a < 32; a++)
            for (int q = 0; q < 32; q++) if (ps.inbps[a * 32 + q]) { x.lmask |= 1u << a; break; }
        for (int a = 0; a < 32; a++)
            for (int q = 0; q < 29; q++) if (ps.inbps[a * 32 + q]) x.pmask[a] |= 1u << q;
        {
            int letters[32], K = 0;
            for (int a = 0; a < 32; a++) if ((x.lmask >> a) & 1u) letters[K++] = a;
            K += 1;                                                     // (the class of the letters that pair with nothing)
            const int cstride = K | 1;
            for (int ci = 0; ci < K; ci++)
                for (int cj = 0; cj < K; cj++)
                    x.celltab[ci * cstride + cj] = (ci < K - 1 && cj < K - 1) ? x.w[letters[ci] * 32 + letters[cj]] : 0.0;
        }
        x.minlen = ps.minlen; x.minbpscore = ps.minbpscore;
        x.minfinscore = ps.minbpscore * ps.minfinscorefactor;          // SQRNdbnseq.py:1073
        x.bracketweight = ps.bracketweight; x.distcoef = ps.distcoef;
        x.orderpenalty = ps.orderpenalty; x.loopbonus = ps.loopbonus;
        for (int k = 0; k <= SQ_MAXLEVELS; k++) x.oftab[k] = pow(1.0 / (1 + k), ps.orderpenalty);   // :729
        {
            // maxima of the finalscore's factors (sq_internal.h): orderfactor over the table; loopfactor :715 with both
            // loops good and equal sides (loopbonus >= 0; a negative bonus only lowers it below 1); the distance factor
            // (1 / (1 + d)) ** distcoef is <= 1 for distcoef >= 0
            double of = x.oftab[0];
            for (int k = 1; k <= SQ_MAXLEVELS; k++) of = x.oftab[k] > of ? x.oftab[k] : of;
            const double lb = ps.loopbonus;
            x.ub_of = of;
            x.ub_lf = lb >= 0 ? (1.0 + lb * 2.0) + lb * 2.0 : 1.0;
            if (!(ps.distcoef >= 0) || !(of >= 0) || !std::isfinite(of) || !std::isfinite(lb)) x.ub_lf = INFINITY;
        }
        {
            bool dy = true;
            for (int q = 0; q < 32 * 32 && dy; q++) {
                const double w = x.w[q] * 1024.0;
                dy = std::fabs(x.w[q]) <= 1024.0 && w == std::floor(w);
            }
            b->pset_dyadic.push_back(dy ? 1 : 0);
            int kletters = 0;                                            // letters with at least one pair (+ 1 class for the rest)
            for (int a = 0; a < 32; a++) {
                bool any = false;
                for (int q = 0; q < 32; q++) any |= ps.inbps[a * 32 + q] != 0;
                kletters += any ? 1 : 0;
            }
            b->pset_classes.push_back(kletters + 1);
        }
        {
            double sc = 1.0;
            const int64_t ne = pow17_entries(ps, L.maxn, sc);
            x.pow_off = (int32_t)powtab.size(); x.pow_len = (int32_t)ne; x.pow_scale = sc;
            for (int64_t k = 0; k < ne; k++) powtab.push_back(pow((double)k / sc, 1.7));          // SQRNalgos.py:101,122
        }
        const double bw = ps.bracketweight;
        x.bw_integral = (bw == std::floor(bw) && std::fabs(bw) <= 64) ? 1 : 0;
        x.sdf_off = (int32_t)sdf.size(); x.sdf_len = 0;
        if (x.bw_integral) {
            x.sdf_len = (int32_t)(std::max(1.0, std::fabs(bw)) * L.maxn + 16);
            for (int k = 0; k < x.sdf_len; k++) sdf.push_back(pow(1.0 / (1.0 + (double)k), ps.distcoef));   // :726
        }
    }
    // reactivity levels: encoded reactivities (3 / 10 / 26 symbols) take few distinct values per sequence; with <= 16
    // of them the reactfactor of a cell is a table lookup instead of an fp64 sqrt (and division) per cell and round
    std::vector<uint8_t> ridx(L.ltot, 0);
    std::vector<int32_t> seq_levels(d->nseq, 0), seq_rf(d->nseq, -1);
    std::vector<double> rftab;
    std::vector<uint8_t> seq_def(d->nseq, 0);                // every reactivity of the sequence 0.5 (:273)
    for (int s = 0; s < d->nseq; s++) {
        const int off = d->seq_off[s], n = d->seq_off[s + 1] - off;
        if (!d->reacts || all_half(d->reacts + off, n)) {    // one level (index 0 everywhere: ridx is zeroed), no factor table
            seq_def[s] = 1; seq_levels[s] = n > 0 ? 1 : 0;
            continue;
        }
        double vals[16]; int nv = 0; bool fits = true;
        for (int i = 0; i < n && fits; i++) {
            const double r = d->reacts[off + i];
            int q = 0;
            while (q < nv && !(vals[q] == r)) q++;
            if (q == nv) { if (nv == 16 || r != r) { fits = false; break; } vals[nv++] = r; }
            ridx[off + i] = (uint8_t)q;
        }
        seq_levels[s] = fits ? nv : 0;
        // (1 - (r_a + r_b) / 2) * 2) ** 0.5 for every pair of the sequence's levels through the host's libm pow, which is
        // what CPython's `**` calls (SQRNdbnseq.py:333): the device reads these instead of taking a sqrt
        seq_rf[s] = -1;
        if (fits && nv > 0 && (int64_t)(rftab.size() / 256) < L.n_rftab) {
            seq_rf[s] = (int32_t)(rftab.size() / 256);
            rftab.resize(rftab.size() + 256, 0.0);
            double *T = rftab.data() + (size_t)seq_rf[s] * 256;
            for (int a = 0; a < nv; a++)
                for (int c2 = 0; c2 < nv; c2++) T[a * 16 + c2] = pow((1.0 - (vals[a] + vals[c2]) / 2.0) * 2.0, 0.5);
        }
    }
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("paramsets", now_s());
#endif
    // ---- jobs ----
    b->jobs.resize(d->njobs);
    int64_t m32 = 0, m64 = 0, mbits = 0;
    std::vector<uint32_t> rbpk((size_t)d->rbp_off[d->nseq]);
    for (size_t k = 0; k < rbpk.size(); k++) rbpk[k] = (uint32_t)d->rbps[2 * k] | ((uint32_t)d->rbps[2 * k + 1] << 16);
    for (int sq = 0; sq < d->nseq; sq++)                       // per sequence by (i + j, i): a diagonal's pairs are one run (sq_scan6_kernel)
        std::sort(rbpk.begin() + d->rbp_off[sq], rbpk.begin() + d->rbp_off[sq + 1], [](uint32_t x, uint32_t y) {
            const uint32_t sx = (x & 0xFFFFu) + (x >> 16), sy = (y & 0xFFFFu) + (y >> 16);
            return sx != sy ? sx < sy : (x & 0xFFFFu) < (y & 0xFFFFu);
        });
    std::vector<double> pset_maxabs(2 * (size_t)d->npset, 0.0);   // largest |cell| a paramset can produce: plain / with reactivity factors
    std::vector<double> pset_runs(d->npset);                      // share of the cells that start a maximal run of >= minlen
    for (int p = 0; p < d->npset; p++) {
        const sq_paramset &ps = d->psets[p];
        pset_runs[p] = std::pow(0.375, std::max(1.0, std::ceil(ps.minlen)) - 1.0);
        for (int q = 0; q < 32 * 32; q++) {
            if (!ps.inbps[q]) continue;
            const double w = ps.bpweight[q];
            pset_maxabs[2 * p] = std::max(pset_maxabs[2 * p], std::fabs(w));
            pset_maxabs[2 * p + 1] = std::max(pset_maxabs[2 * p + 1], std::fabs(w) * (w > 0 ? 1.4142135623730951 : 100.0));   // SQRNdbnseq.py:333-336
        }
    }
    // paramsets that pair the same letters give a sequence the same boolean matrix (SQRNdbnseq.py:300-304: letters,
    // restraints and the minimal loop decide a cell): one bit matrix per (sequence, letter set)
    std::vector<int> pset_sig(d->npset);
    for (int p = 0; p < d->npset; p++) {
        pset_sig[p] = p;
        for (int q = 0; q < p; q++) {
            bool same = true;
            for (int e = 0; e < 32 * 32 && same; e++) same = (d->psets[p].inbps[e] != 0) == (d->psets[q].inbps[e] != 0);
            if (same) { pset_sig[p] = pset_sig[q]; break; }
        }
    }
    // (not with fp32 score matrices: sq_bits_kernel then derives every job's bits from ITS matrix)
    const bool no_share = getenv("SQ_NO_SHARED_BITS") != nullptr || b->has_fp32;
    std::vector<int64_t> shared_bits((size_t)d->nseq * (size_t)std::max(d->npset, 1), -1);
    for (int j = 0; j < d->njobs; j++) {
        SqJob &J = b->jobs[j];
        const int s = d->job_seq[j];
        J.n = d->seq_off[s + 1] - d->seq_off[s];
        J.ld = ld_of(J.n); J.seq = s; J.pset = d->job_pset[j];
        J.pos_off = d->seq_off[s];
        J.mat64_off = -1; J.has_ext = 0;
        J.nw = bits_nw(J.n); J.bpitch = bits_pitch(J.n);
        J.rb_off = d->rbp_off[s]; J.nrb = d->rbp_off[s + 1] - d->rbp_off[s];
        const bool ext = d->ext_score && d->ext_score[j];
        {
            int64_t &slot = shared_bits[(size_t)s * (size_t)d->npset + (size_t)pset_sig[d->job_pset[j]]];
            if (!ext && !no_share && slot >= 0) { J.bits_off = slot; J.bits_owner = 0; }      // (a caller's boolean matrix: the job's own)
            else {
                J.bits_off = mbits; J.bits_owner = 1; mbits += (int64_t)J.nw * J.bpitch;
                if (!ext && !no_share) slot = J.bits_off;
            }
        }
        const bool term = d->bpp_term && d->bpp_term[j];
        const bool shared = d->mul_shared && d->mul_shared[j];
        J.mulsh = (shared && !ext && L.mul_direct) ? 1 : 0;
        const bool mul = (d->mul_score && d->mul_score[j]) || term || (shared && !J.mulsh);
        J.ext_add = term && d->psets[J.pset].bpp < 0 ? 1 : 0;
        if (ext) { J.mat64_off = m64; J.has_ext = 1; m64 += 2 * (int64_t)J.n * J.n; }
        else if (mul) { J.mat64_off = m64; J.has_ext = 2; m64 += (int64_t)J.n * J.n; }
        J.mat_off = -1;
        J.mat64_diag = (shared && !ext && !J.mulsh) ? 1 : 0;   // SQ_MUL_GATHER=1: the gather kernel writes score x weight, diagonal-major (sq_cells.h)
        if (b->has_fp32 || (J.has_ext && !J.mat64_diag)) { J.mat_off = m32; m32 += (int64_t)align_up((size_t)J.n * J.ld, 64); }
        const bool def = seq_def[s] != 0;                 // SQRNdbnseq.py:273
        J.default_reacts = def ? 1 : 0;
        J.react_levels = def ? 0 : seq_levels[s];
        J.rf_idx = def ? -1 : seq_rf[s];
        J.interchainonly = d->interchainonly;
        {
            const double est = 0.117 * (double)J.n * J.n * pset_runs[J.pset] * 1.6 + 256;   // maximal runs with len >= minlen
            J.cand_cap = (int32_t)std::max<int64_t>((int64_t)L.cpn * J.n, (int64_t)est);
        }
        // bound of |cell| for the scan's fp32 prefilter margin
        double mx = 0;
        const size_t nn = (size_t)J.n * J.n;
        if (ext) {
            for (size_t q = 0; q < nn; q++) if (d->ext_bool[j] && d->ext_bool[j][q] != 0) mx = std::max(mx, std::fabs(d->ext_score[j][q]));
        } else {
            mx = pset_maxabs[2 * J.pset + (def ? 0 : 1)];
            if (shared) mx *= std::fabs(d->mul_maxabs);
            else if (mul) {
                const double *tm = term ? d->bpp_term[j] : d->mul_score[j];
                double mm = 0;
                for (size_t q = 0; q < nn; q++) mm = std::max(mm, std::fabs(tm[q]));
                mx = J.ext_add ? mx + mm : mx * mm;
            }
        }
        J.maxabs = (float)(mx * 1.0000002);
    }
    // cell table of the scoring kernels (dynamic LDS): K R x (K R | 1) doubles for the largest K R of the batch, K = the
    // paramset's letter classes, R = the sequence's reactivity levels when K R <= 32 (else 1: factors per cell)
    b->cell_entries = 32;
    for (const SqJob &J : b->jobs) {
        const int K = b->pset_classes[J.pset];
        const int R = (!J.default_reacts && J.react_levels > 0 && K * J.react_levels <= 32) ? J.react_levels : 1;
        const int KR = K * R;
        b->cell_entries = std::max(b->cell_entries, KR * (KR | 1));
    }
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("jobs", now_s());
#endif
    // ---- device carve + uploads ----
    b->ctx.codes = (uint8_t *)(base + L.off_codes); b->ctx.flags = (uint8_t *)(base + L.off_flags);
    b->ctx.inc4 = (uint8_t *)(base + L.off_inc4); b->ctx.chain = (int16_t *)(base + L.off_chain);
    b->ctx.e0c = (uint8_t *)(base + L.off_e0); b->ctx.reacts = (double *)(base + L.off_reacts); b->ctx.ridx = (uint8_t *)(base + L.off_ridx);
    b->ctx.jobs = (SqJob *)(base + L.off_jobs); b->ctx.psets = (SqPsetDev *)(base + L.off_psets);
    b->ctx.sdftab = (double *)(base + L.off_sdf); b->ctx.rftab = (double *)(base + L.off_rftab);
    b->ctx.powtab = (double *)(base + L.off_powtab);
    b->ctx.mat32 = (float *)(base + L.off_mat32); b->ctx.mat64 = (double *)(base + L.off_mat64);
    b->d_structs = (SqStruct *)(base + L.off_structs); b->d_strands = (SqStrand *)(base + L.off_strands);
    int16_t *stbase = (int16_t *)(base + L.off_state);
    const size_t plane = (size_t)L.stride * L.max_structs;
    b->state.P = stbase; b->state.E8 = (uint8_t *)(stbase + plane); b->state.U = stbase + 2 * plane; b->state.SU = stbase + 3 * plane;
    b->state.stride = L.stride;
    b->state.FB = (uint32_t *)(base + L.off_fb); b->state.fbstride = L.fbstride;
    b->ctxtab = SqCtxTab{};
    if (L.ctx_cap) {
        b->ctxtab.rec = (SqCtxRec *)(base + L.off_ctx_rec); b->ctxtab.depth = (int16_t *)(base + L.off_ctx_depth);
        b->ctxtab.rmq = (uint16_t *)(base + L.off_ctx_rmq); b->ctxtab.ok = (uint8_t *)(base + L.off_ctx_ok);
        b->ctxtab.cap = L.ctx_cap; b->ctxtab.levels = L.ctx_levels;
    }
    b->ctx.bits = (uint32_t *)(base + L.off_bits); b->ctx.rbpk = (uint32_t *)(base + L.off_rbpk);
    b->scan.cand_cnt = (uint32_t *)(base + L.off_cnt); b->scan.ctr = (SqCounters *)(base + L.off_ctr);
    b->scan.best = (unsigned long long *)(base + L.off_cnt + 4 * align_up((size_t)L.max_structs, 2));
    b->scan.ok_cnt = (uint32_t *)(base + L.off_cnt + 12 * align_up((size_t)L.max_structs, 2));
    b->scan.cands = (SqCand *)(base + L.off_cands);
    b->d_out = (SqOut *)(base + L.off_out);
    b->chain.chain = (SqChain *)(base + L.off_crec); b->chain.stems = (SqChainStem *)(base + L.off_cstems);
    b->chain.strands = (SqStrand *)(base + L.off_cstrands); b->chain.sidx = (int16_t *)(base + L.off_csidx);
    b->chain.d_nfin = (uint32_t *)(base + L.off_cnfin);
    b->chain_T = L.chain_T;
    {   // the device log of final structures and the device tail's arrays
        b->d_fin = (SqPoolFin *)(base + L.off_fin); b->d_fin_stems = (SqPoolStem *)(base + L.off_fin_stems);
        b->d_fin_ctr = (uint32_t *)(base + L.off_fin_ctr); b->d_job_evals = (long long *)(base + L.off_jobevals);
        b->fin_cap = L.fin_cap; b->fin_stem_cap = L.fin_stem_cap;
        b->d_refp = (int16_t *)(base + L.off_t_refp); b->d_refn = (int32_t *)(base + L.off_t_refn);
        b->chain.fin = b->d_fin; b->chain.fin_ctr = b->d_fin_ctr; b->chain.fin_cap = L.fin_cap; b->chain.job_evals = b->d_job_evals;
        SqTailIO &T = b->tail;
        T.fin = b->d_fin; T.fin_stems = b->d_fin_stems; T.nfin_ptr = b->d_fin_ctr; T.chain_stems = b->chain.stems;
        T.fin_cap = L.fin_cap; T.fin_stem_cap = L.fin_stem_cap;
        uint32_t *tj = (uint32_t *)(base + L.off_t_jobs);
        T.job_cnt = tj; T.job_start = tj + (d->njobs + 1); T.job_fill = tj + 2 * ((size_t)d->njobs + 1);
        T.job_evals = b->d_job_evals; T.njobs = d->njobs; T.nseq = d->nseq;
        T.seq_job0 = (int32_t *)(base + L.off_t_seqjob0);
        T.ord = (uint32_t *)(base + L.off_t_ord); T.ord2 = T.ord + L.fin_cap;
        T.cstems = (SqPoolStem *)(base + L.off_t_cstems); T.cs_n = (uint32_t *)(base + L.off_t_csn);
        T.hash = (unsigned long long *)(base + L.off_t_hash); T.rep = (uint32_t *)(base + L.off_t_rep);
        T.mask = (unsigned long long *)(base + L.off_t_mask); T.scores = (double *)(base + L.off_t_scores);
        T.dlist = (uint32_t *)(base + L.off_t_dlist); T.rlist = (uint32_t *)(base + L.off_t_rlist);
        T.seqs = (SqTailSeq *)(base + L.off_t_seqs);
        T.pow17h = (double *)(base + L.off_t_pow); T.pow17h_len = L.pow_len;
        T.fallback = b->d_fin_ctr + 3;
    }
    b->chain_tmax = 1;
    for (int j = 0; j < d->njobs; j++)
        b->chain_tmax = std::max(b->chain_tmax, chain_tcap(d->seq_off[d->job_seq[j] + 1] - d->seq_off[d->job_seq[j]], d->psets[d->job_pset[j]].minlen));
    b->algo_scratch = L.algo_bytes ? base + L.off_algo : nullptr; b->algo_bytes = L.algo_bytes; b->algo_used = 0;
    if (L.pool_pt) {
        SqPoolIO &P = b->pool_io;
        P.structs = (SqStruct *)(base + L.off_pstructs); P.recs = (SqChain *)(base + L.off_precs);
        P.stems = (SqChainStem *)(base + L.off_pstems); P.strands = (SqStrand *)(base + L.off_pstrands);
        P.sidx = (int16_t *)(base + L.off_psidx);
        P.smax = L.max_structs; P.pt = L.pool_pt; P.cmax = 64;
        P.jobs = (SqPoolJob *)(base + L.off_pjobs); P.jobrec_of = (int32_t *)(base + L.off_pjobrec);
        P.nchild = (int32_t *)(base + L.off_pnchild); P.child_off = (int32_t *)(base + L.off_pchoff);
        P.finalflag = (uint8_t *)(base + L.off_pflag); P.chosen = (SqPoolPick *)(base + L.off_pchosen); P.parent_of = (int32_t *)(base + L.off_pparent);
        P.hdr = (SqPoolHdr *)(base + L.off_phdr);
        P.fin = b->d_fin; P.fin_stems = b->d_fin_stems; P.fin_cap = L.fin_cap; P.fin_stem_cap = L.fin_stem_cap;
        P.fin_ctr = b->d_fin_ctr; P.job_evals = b->d_job_evals;
        P.kept_ctr = nullptr;
        b->kept = SqKept{nullptr, nullptr, nullptr, nullptr, 0u, 0};
        if (L.kept_pages) b->kept = SqKept{base + L.off_kpages, (uint32_t *)(base + L.off_kctr), (uint32_t *)(base + L.off_kcnt), (uint32_t *)(base + L.off_ktab), L.kept_pages, 1};
    }

    hipStream_t st = b->stream;
    // Uploads go through a pinned staging buffer of the library.  A copy straight from pageable memory makes the runtime
    // register the caller's pages with the driver; when the allocator later returns such pages to the kernel (munmap /
    // heap trim) the driver evicts the process's queues for tens of milliseconds -- measured as 20-35 ms stalls in the
    // third fold after a batch was created.  Buffers larger than the staging area go in slices.
    struct Stager {
        hipStream_t st; char *buf = nullptr; size_t cap = 0, cur = 0; int rc = 0;
        ~Stager() { if (buf) { hipStreamSynchronize(st); sq_pinned_put(buf); } }
        int put(void *dst, const void *src, size_t bytes)
        {
            const char *s = (const char *)src; char *d = (char *)dst;
            while (bytes) {
                if (cur == cap) { rc = sq_check(hipStreamSynchronize(st), "upload"); if (rc) return rc; cur = 0; }
                const size_t take = std::min(bytes, cap - cur);
                memcpy(buf + cur, s, take);
                rc = sq_check(hipMemcpyAsync(d, buf + cur, take, hipMemcpyHostToDevice, st), "upload");
                if (rc) return rc;
                cur += (take + 255) & ~(size_t)255; if (cur > cap) cur = cap;
                s += take; d += take; bytes -= take;
            }
            return 0;
        }
    } stager;
    stager.st = st;
    {
        size_t want = (size_t)L.ltot * 16 + 8 * rftab.size() + 8 * powtab.size() + 8 * (size_t)L.pow_len + 4 * ((size_t)d->nseq + 1) + sizeof(SqJob) * d->njobs + sizeof(SqPsetDev) * d->npset + 8 * sdf.size() + 4 * rbpk.size() + 16384;
        for (int j = 0; j < d->njobs; j++)
            if (b->jobs[j].has_ext && !(d->mul_shared && d->mul_shared[j])) want += (size_t)b->jobs[j].n * b->jobs[j].n * 8 * (b->jobs[j].has_ext == 1 ? 2 : 1);
        stager.cap = std::min<size_t>(std::max<size_t>(want, (size_t)1 << 20), (size_t)64 << 20) & ~(size_t)255;
        void *pb = nullptr;
        if (sq_pinned_get(&pb, stager.cap)) { delete b; return 2; }
        stager.buf = (char *)pb;
    }
#define UP(dst, src, bytes) do { int _r = stager.put((void *)(dst), (src), (bytes)); if (_r) { hipStreamSynchronize(st); delete b; return _r; } } while (0)
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("carve", now_s());
#endif
    UP(b->ctx.codes, b->codes.data(), L.ltot); UP(b->ctx.flags, b->flags.data(), L.ltot);
    UP(b->ctx.inc4, inc4.data(), L.ltot); UP(b->ctx.chain, chain.data(), L.ltot * 2);
    UP(b->ctx.e0c, e0.data(), L.ltot);
    if (d->reacts) UP(b->ctx.reacts, b->reacts.data(), L.ltot * 8);
    else hipLaunchKernelGGL(sq_fill_f64_kernel, dim3(256), dim3(256), 0, st, const_cast<double *>(b->ctx.reacts), (long long)L.ltot, 0.5);
    UP(b->ctx.ridx, ridx.data(), L.ltot);
    b->ridx = ridx;
    UP(b->ctx.jobs, b->jobs.data(), sizeof(SqJob) * d->njobs);
    UP(b->ctx.psets, pd.data(), sizeof(SqPsetDev) * d->npset);
    if (!sdf.empty()) UP(b->ctx.sdftab, sdf.data(), 8 * sdf.size());
    if (!rftab.empty()) UP(b->ctx.rftab, rftab.data(), 8 * rftab.size());
    if (!powtab.empty()) UP(b->ctx.powtab, powtab.data(), 8 * powtab.size());
    b->psets_dev = pd;                                         // (host copy: which paramsets have a power table)
    b->rftab.swap(rftab);                                      // (host copy: RunAlgo's stem filters re-sum cells, sq_algos.hip)
    if (!rbpk.empty()) UP(b->ctx.rbpk, rbpk.data(), 4 * rbpk.size());
    {
        // device tail: the first job of every sequence -- it needs each sequence's jobs contiguous, in sequence order, at
        // most 64 of them (the paramset mask); any other job list keeps the host tail -- and pow(k / 2, 1.7) from the
        // host's libm for ScoreStruct's stem terms (:884: sums of 4 / 1.5 / -0.5 per pair are multiples of 1/2)
        std::vector<int32_t> sj0((size_t)d->nseq + 1, 0);
        bool grouped = true;
        int j = 0;
        for (int sq = 0; sq < d->nseq; sq++) {
            sj0[sq] = j;
            while (j < d->njobs && d->job_seq[j] == sq) j++;
            if (j == sj0[sq] || j - sj0[sq] > 64) grouped = false;
        }
        sj0[d->nseq] = j;
        if (j != d->njobs) grouped = false;
        if (grouped) UP(b->tail.seq_job0, sj0.data(), 4 * sj0.size());
        else b->tail.seq_job0 = nullptr;
        std::vector<double> pw((size_t)L.pow_len);
        for (int k = 0; k < L.pow_len; k++) pw[k] = pow(0.5 * (double)k, 1.7);
        UP(b->tail.pow17h, pw.data(), 8 * pw.size());
    }
    for (int j = 0; j < d->njobs; j++) {
        const SqJob &J = b->jobs[j];
        const size_t nn = (size_t)J.n * J.n * 8;
        if (J.has_ext == 1) {
            if (!d->ext_bool || !d->ext_bool[j]) { hipStreamSynchronize(st); delete b; sq_set_error("ext_score without ext_bool"); return -1; }
            UP(b->ctx.mat64 + J.mat64_off, d->ext_score[j], nn);
            UP(b->ctx.mat64 + J.mat64_off + (int64_t)J.n * J.n, d->ext_bool[j], nn);
        } else if (J.has_ext == 2 && !(d->mul_shared && d->mul_shared[j])) {
            UP(b->ctx.mat64 + J.mat64_off, (d->bpp_term && d->bpp_term[j]) ? d->bpp_term[j] : d->mul_score[j], nn);
        }
    }
    if (d->mul_matrix_dev && d->mul_shared) {
        // jobs weighted by the shared L x L matrix: the kernels read it through the column maps (sq_mulsh_weight) from a
        // diagonal-major copy; SQ_MUL_GATHER=1: their N x N slices are gathered from it here
        int32_t *d_cols = (int32_t *)(base + L.off_mulcols);
        b->ctx.mulM = (double *)(base + L.off_mulM); b->ctx.mulcols = d_cols; b->ctx.mulL = d->mul_L;
        sq_launch_mul_diag((const double *)d->mul_matrix_dev, d->mul_L, (double *)(base + L.off_mulM), st);
        for (int64_t q = 0; q < L.ltot; q++)
            if (d->mul_cols[q] < 0 || d->mul_cols[q] >= d->mul_L) { hipStreamSynchronize(st); delete b; sq_set_error("mul_cols out of range"); return -1; }
        UP(d_cols, d->mul_cols, 4 * (size_t)L.ltot);
        std::vector<int32_t> jl;
        for (int j = 0; j < d->njobs; j++) if (d->mul_shared[j] && !b->jobs[j].mulsh) jl.push_back(j);
        if (!jl.empty()) {
            // (the job list travels in the candidate arena's first bytes: nothing else uses it before the first fold)
            int32_t *d_jl = (int32_t *)(base + L.off_cands);
            UP(d_jl, jl.data(), 4 * jl.size());
            sq_launch_gather_mul(b->ctx, d_jl, (int)jl.size(), L.maxn, st);
            if (sq_check(hipGetLastError(), "sq_gather_mul_kernel")) { hipStreamSynchronize(st); delete b; return 2; }
        }
    }
#undef UP
    // pinned staging
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("uploads", now_s());
#endif
    if (sq_pinned_get((void **)&b->h_structs, sizeof(SqStruct) * L.max_structs) ||
        sq_pinned_get((void **)&b->h_strands, sizeof(SqStrand) * (size_t)L.strand_cap) ||
        sq_pinned_get((void **)&b->h_ctr, sizeof(SqCounters)) ||
        sq_pinned_get((void **)&b->h_seq, 64)) { delete b; return 2; }
    *b->h_seq = 0; b->round_seq = 0;
    b->h_out_cap = (uint32_t)std::min<uint64_t>(1u << 18, L.out_cap);
    if (sq_pinned_get((void **)&b->h_out, sizeof(SqOut) * (size_t)b->h_out_cap) ||
        sq_pinned_get((void **)&b->h_ctr2, sizeof(SqCounters)) ||
        sq_pinned_get((void **)&b->h_seq2, 64)) { delete b; return 2; }
    // (cached buffers come back with their old contents: the completion words must not look like a finished round)
    memset(b->h_ctr, 0, sizeof(SqCounters)); memset(b->h_ctr2, 0, sizeof(SqCounters));
    memset(b->h_seq, 0, 64); memset(b->h_seq2, 0, 64);
    *b->h_seq2 = 0;
    {   // the lane that spans all round buffers, and its two halves
        SqLane &F = b->lane_full;
        F.h_structs = b->h_structs; F.h_strands = b->h_strands; F.h_out = b->h_out; F.h_ctr = b->h_ctr; F.h_seq = b->h_seq;
        F.d_structs = b->d_structs; F.d_strands = b->d_strands; F.d_out = b->d_out; F.d_ctr = b->scan.ctr;
        F.h_out_cap = b->h_out_cap; F.out_cap = b->out_cap; F.slot0 = 0; F.max_structs = b->max_structs;
        F.strand_cap = b->strand_cap; F.cand0 = 0; F.cand_records = b->cand_records;
        F.round_seq = &b->round_seq;
        for (int k = 0; k < 2; k++) {
            SqLane &H = b->lane_half[k];
            const int ms0 = b->max_structs / 2, sc0 = b->strand_cap / 2;
            const uint32_t ho0 = b->h_out_cap / 2, oc0 = b->out_cap / 2;
            H.slot0 = k ? ms0 : 0; H.max_structs = k ? b->max_structs - ms0 : ms0;
            H.h_structs = b->h_structs + H.slot0; H.d_structs = b->d_structs + H.slot0;
            H.strand_cap = k ? b->strand_cap - sc0 : sc0;
            H.h_strands = b->h_strands + (k ? sc0 : 0); H.d_strands = b->d_strands + (k ? sc0 : 0);
            H.h_out_cap = k ? b->h_out_cap - ho0 : ho0; H.out_cap = k ? b->out_cap - oc0 : oc0;
            H.h_out = b->h_out + (k ? ho0 : 0); H.d_out = b->d_out + (k ? oc0 : 0);
            H.h_ctr = k ? b->h_ctr2 : b->h_ctr; H.h_seq = k ? b->h_seq2 : b->h_seq;
            H.round_seq = k ? &b->round_seq2 : &b->round_seq;   // (one counter per completion word)
            H.d_ctr = (SqCounters *)((char *)b->scan.ctr + (k ? 64 : 0));
        }
    }
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("pinned+kernels", now_s());
#endif
    int rr = sq_check(hipStreamSynchronize(st), "sync after upload");   // host vectors above go out of scope
    if (rr) { delete b; return rr; }
#ifdef SQ_CREATE_PROF
    _cp.emplace_back("sync", now_s());
    { std::string line = "[sq_batch_create ms]"; for (size_t k = 1; k < _cp.size(); k++) { char t[64]; snprintf(t, sizeof t, " %s %.2f", _cp[k].first, (_cp[k].second - _cp[k - 1].second) * 1e3); line += t; } fprintf(stderr, "%s\n", line.c_str()); }
#endif
    b->results.resize(d->nseq);
    *out = b;
    return 0;
}

extern "C" void sq_batch_destroy(sq_batch *b)
{
    if (!b) return;
    static const bool trace = getenv("SQ_PINNED_TRACE") != nullptr;     // (with the cache's trace: a destroy that takes long, by phase)
    const double td0 = trace ? now_s() : 0;
    hipStreamSynchronize(b->stream);
    for (int k = 0; k < 4; k++) if (b->side[k]) { hipStreamSynchronize(b->side[k]); sq_stream_put(b->device, b->side[k]); }
    if (b->lane_stream) { hipStreamSynchronize(b->lane_stream); sq_stream_put(b->device, b->lane_stream); }
    const double td1 = trace ? now_s() : 0;
    sq_event_put(b->device, b->class_ev);
    sq_event_put(b->device, b->edges_ev);
    sq_pinned_put(b->h_structs); sq_pinned_put(b->h_strands); sq_pinned_put(b->h_ctr); sq_pinned_put(b->h_seq);
    sq_pinned_put(b->h_ctr2); sq_pinned_put(b->h_seq2); sq_pinned_put(b->h_out);
    for (int k = 0; k < 4; k++) sq_pinned_put(b->stage_buf[k]);
    sq_pinned_put(b->chain.h_stems); sq_pinned_put(b->chain.h_fin); sq_pinned_put((void *)b->chain.h_nfin);
    sq_pinned_put(b->h_chain);
    sq_pinned_put(b->pool_io.h_hdr); sq_pinned_put(b->pool_io.h_jobs);
    sq_pinned_put(b->h_tail_totals); sq_pinned_put(b->h_rec_off); sq_pinned_put(b->h_txt_off); sq_pinned_put(b->h_deep);
    sq_pinned_put(b->h_rec); sq_pinned_put(b->h_txt); sq_pinned_put(b->h_app); sq_pinned_put(b->h_ref);
    sq_pinned_put(b->h_pool_recs); sq_pinned_put(b->h_pool_jobs); sq_pinned_put(b->h_pool_jobrec);
    const double td2 = trace ? now_s() : 0;
    sq_pool_put(b->pool);
    sq_event_put(b->device, b->lane_ev);
    for (auto &p : b->prof) {
        for (auto &e : p.pending) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
        for (auto &e : p.pool) hipEventDestroy(e);
    }
    const double td3 = trace ? now_s() : 0;
    delete b;
    if (trace && now_s() - td0 > 0.01)
        fprintf(stderr, "[sq_batch_destroy] %.1f ms: stream syncs %.1f, pinned buffers %.1f, pool + events %.1f, delete %.1f\n", (now_s() - td0) * 1e3,
                (td1 - td0) * 1e3, (td2 - td1) * 1e3, (td3 - td2) * 1e3, (now_s() - td3) * 1e3);
}

