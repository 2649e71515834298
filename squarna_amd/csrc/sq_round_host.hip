// sq_round_host.hip -- the launches of one round, the host-driven round driver, the per-call C ABI of a-1 .. a-6 (sq_bpmatrix_fill / _read, sq_optimal_stems) and alignment step 1.
#include "sq_host_int.h"

// ---- a-1 -----------------------------------------------------------------------------------
// full = 1: fp32 score matrices of every job (the API op).  full = 0: only what the fold path reads -- the
// bit matrices, computed straight from the O(N) inputs; jobs with caller / multiplier matrices still go
// through the fp32 fill (it imports the bool matrix and forms score * multiplier in the dense arena).
int sq_fill_impl(sq_batch *b, int full)
{
    int64_t maxq = 0, maxw = 0; double bytes = 0; bool any_ext1 = false, any_ext = false;
    for (const SqJob &J : b->jobs) {
        maxq = std::max<int64_t>(maxq, ((int64_t)J.n * J.ld + 3) / 4);
        maxw = std::max<int64_t>(maxw, (int64_t)J.nw * ((J.bpitch + 255) / 256));
        if (full || J.has_ext) bytes += 4.0 * J.n * J.n;                // algorithmic: one fp32 N x N write
        any_ext1 |= J.has_ext == 1; any_ext |= J.has_ext != 0;
    }
    for (int j0 = 0; j0 < b->njobs; j0 += 32768) {
        const int nj = std::min(32768, b->njobs - j0);
        SqDevCtx c = b->ctx; c.jobs = b->ctx.jobs + j0;
        dim3 grid((unsigned)std::min<int64_t>(std::max<int64_t>((maxq + 255) / 256, 1), 1024), (unsigned)nj);
        if (full || any_ext) {
            ProfScope ps(b, 0, j0 == 0 ? bytes : 0);
            // the fill's fast path stages the O(N) inputs in LDS (12 bytes per position) when the longest sequence fits;
            // its blocks are fewer and fatter than the generic path's so that the staging is amortised
            const bool lds_inputs = b->maxn <= 4096;
            // 16-byte stores per thread: as many as leave ~4096 blocks in the launch (the LDS staging of a block is amortised over them)
            static const int fper_env = getenv("SQ_FILL_PER") ? atoi(getenv("SQ_FILL_PER")) : 0;
            const int64_t fper = 256 * (fper_env > 0 ? (int64_t)fper_env
                                                     : std::min<int64_t>(std::max<int64_t>(maxq * nj / (256 * 4096), 4), 64));
            const size_t fdyn = lds_inputs ? (size_t)12 * ((b->maxn + 15) & ~15) + 64 : 0;
            dim3 fgrid(lds_inputs ? (unsigned)std::min<int64_t>(std::max<int64_t>((maxq + fper - 1) / fper, 1), 1024) : grid.x, (unsigned)nj);
            hipLaunchKernelGGL(sq_fill_kernel, fgrid, dim3(256), fdyn, b->stream, c, full ? 0 : 1, b->mul_applied ? 1 : 0);
        }
        if (any_ext1) hipLaunchKernelGGL(sq_import_kernel, grid, dim3(256), 0, b->stream, c);
        if (full && !getenv("SQ_BITS_DIRECT")) hipLaunchKernelGGL(sq_bits_kernel, grid, dim3(256), 0, b->stream, c, 0);
        else {
            if (any_ext1) hipLaunchKernelGGL(sq_bits_kernel, grid, dim3(256), 0, b->stream, c, 1);
            dim3 g2((unsigned)std::min<int64_t>(std::max<int64_t>(maxw, 1), 2048), (unsigned)nj);
            double bbytes = 0;
            if (j0 == 0 && !(full || any_ext)) for (const SqJob &J : b->jobs) bbytes += 4.0 * J.nw * J.bpitch;   // bit words written
            ProfScope ps(b, 0, bbytes);
            // letter-mask formulation unless the chain test is on or the O(N) tables outgrow LDS
            const int nwmax = (b->maxn + 31) / 32;
            const size_t mdyn = 3 * (size_t)((b->maxn + 3) & ~3) + 4 * (size_t)b->nletters * (nwmax + 3) + 4 * (size_t)nwmax * b->nletters + 16;
            static const bool no_masks = getenv("SQ_BITS_NOMASKS") != nullptr;
            if (!b->interchainonly && !no_masks && b->nletters > 0 && mdyn <= 60 * 1024) {
                const int bparts = std::max(1, std::min(nwmax, (2048 + nj - 1) / nj));
                hipLaunchKernelGGL(sq_bits_masks_kernel, dim3(bparts, nj), dim3(256), mdyn, b->stream, c, b->nletters);
            } else
                hipLaunchKernelGGL(sq_bits_direct_kernel, g2, dim3(256), 0, b->stream, c);
        }
    }
    HIPCK(hipGetLastError());
    b->bits_ready = true;
    if (full || any_ext) b->mul_applied = true;
    if (full) b->filled = true;
    return 0;
}

extern "C" int sq_bpmatrix_fill(sq_batch *b)
{
    if (!b->has_fp32) { sq_set_error("batch was created with SQ_BATCH_NO_FP32: no fp32 score matrices to fill"); return -4; }
    return sq_fill_impl(b, 1);
}

int sq_prepare_scan(sq_batch *b)
{
    return b->bits_ready ? 0 : sq_fill_impl(b, 0);
}

extern "C" int sq_bpmatrix_read(sq_batch *b, int32_t job, double *boolmat, double *scoremat)
{
    if (job < 0 || job >= b->njobs) { sq_set_error("bad job index"); return -1; }
    const SqJob &J = b->jobs[job];
    const size_t nn = (size_t)J.n * J.n;
    if (J.has_ext == 1) { sq_set_error("job uses caller matrices"); return -1; }
    if (J.mat64_diag) { sq_set_error("job is weighted by the shared stem matrix: its dense matrix is not kept row-major"); return -4; }
    double *tmp = (double *)b->scan.cands;                  // borrowed: idle between rounds
    hipLaunchKernelGGL(sq_dense64_kernel, dim3((unsigned)std::min<size_t>((nn + 255) / 256 + 1, 2048)), dim3(256), 0,
                       b->stream, b->ctx, job, tmp, tmp + nn);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(boolmat, tmp, nn * 8, hipMemcpyDeviceToHost, b->stream));
    HIPCK(hipMemcpyAsync(scoremat, tmp + nn, nn * 8, hipMemcpyDeviceToHost, b->stream));
    HIPCK(hipStreamSynchronize(b->stream));
    if (J.has_ext == 2 && b->mul_applied) {                      // weighted matrix lives in the dense arena
        HIPCK(hipMemcpy(scoremat, b->ctx.mat64 + J.mat64_off, nn * 8, hipMemcpyDeviceToHost));
    }
    return 0;
}

// ---- stem-level pseudoknot levels (== PairsToDBN(returnlevels) on the stems' bps) -----------
static inline bool stems_cross(const HStem &a, const HStem &b)
{
    return (a.i < b.i && b.i < a.j && a.j < b.j) || (b.i < a.i && a.i < b.j && b.j < a.j);   // SQRNdbnseq.py:114-116
}

void sq_stem_levels(const std::vector<HStem> &stems, std::vector<int> &level)
{
    const int T = (int)stems.size();
    level.assign(T, 1);
    if (T < 2) return;
    // scratch kept per thread: this runs once per new structure per round, allocation-free after warm-up
    static thread_local std::vector<int> cc, order, grp, gsize, gord, rank;
    cc.assign(T, 0);
    bool any = false;
    for (int a = 0; a < T; a++)
        for (int b = a + 1; b < T; b++)
            if (stems_cross(stems[a], stems[b])) { cc[a] += stems[b].len; cc[b] += stems[a].len; any = true; }
    if (!any) return;                                       // one group holds everything
    order.resize(T);
    for (int a = 0; a < T; a++) order[a] = a;
    std::sort(order.begin(), order.end(), [&](int a, int b) {   // :125 key (cross_count, p[0])
        if (cc[a] != cc[b]) return cc[a] < cc[b];
        return stems[a].i < stems[b].i;
    });
    grp.assign(T, -1); gsize.clear();
    for (int t = 0; t < T; t++) {                           // :130-136 first fit
        const int p = order[t];
        int placed = (cc[p] == 0 && !gsize.empty()) ? 0 : -1;   // a stem that crosses nothing fits the first group
        for (int g = 0; g < (int)gsize.size() && placed < 0; g++) {
            bool ok = true;
            for (int u = 0; u < t && ok; u++)
                if (grp[order[u]] == g && stems_cross(stems[p], stems[order[u]])) ok = false;
            if (ok) placed = g;
        }
        if (placed < 0) { placed = (int)gsize.size(); gsize.push_back(0); }
        grp[p] = placed; gsize[placed] += stems[p].len;
    }
    gord.resize(gsize.size());
    for (size_t g = 0; g < gsize.size(); g++) gord[g] = (int)g;
    for (size_t g = 1; g < gord.size(); g++) {              // :139 stable, descending by size (insertion sort: a handful of
        const int x = gord[g];                              // groups, and std::stable_sort would allocate its buffer per call)
        size_t q = g;
        while (q > 0 && gsize[gord[q - 1]] < gsize[x]) { gord[q] = gord[q - 1]; q--; }
        gord[q] = x;
    }
    rank.resize(gsize.size());
    for (size_t r = 0; r < gord.size(); r++) rank[gord[r]] = (int)r;
    for (int a = 0; a < T; a++) level[a] = rank[grp[a]] + 1;
}

static inline void set_levels(HStruct &s, const std::vector<int> &level)
{
    for (size_t k = 0; k < s.stems.size(); k++) {
        const HStem &st = s.stems[k];
        const uint8_t lv = (uint8_t)std::min(level[k], 255);
        for (int half = 0; half < 2; half++) {
            const int16_t start = (int16_t)(half == 0 ? st.i : st.j - st.len + 1);
            auto it = std::lower_bound(s.strands.begin(), s.strands.end(), start,
                                       [](const SqStrand &x, int16_t v) { return x.start < v; });
            it->level = lv;
        }
    }
}

void sq_build_strands(HStruct &s)
{
    s.strands.clear();
    s.anycross = false;
    for (const HStem &st : s.stems) {
        s.strands.push_back(SqStrand{(int16_t)st.i, (int16_t)st.len, (int16_t)st.j, 1, 1});
        s.strands.push_back(SqStrand{(int16_t)(st.j - st.len + 1), (int16_t)st.len, (int16_t)(st.i + st.len - 1), 1, 0});
    }
    std::sort(s.strands.begin(), s.strands.end(), [](const SqStrand &x, const SqStrand &y) { return x.start < y.start; });
    for (size_t a = 0; a < s.stems.size() && !s.anycross; a++)
        for (size_t b = a + 1; b < s.stems.size(); b++)
            if (stems_cross(s.stems[a], s.stems[b])) { s.anycross = true; break; }
    if (s.anycross) {
        std::vector<int> level;
        sq_stem_levels(s.stems, level);
        set_levels(s, level);
    }
}

// take_parent: the parent is dead after this child (its last one): its vectors are moved instead of copied
void sq_extend_struct(const HStruct &parent, const HStem &stem, HStruct &child, bool take_parent)
{
    child.job = parent.job;
    if (take_parent) {
        HStruct &p = const_cast<HStruct &>(parent);
        child.stems = std::move(p.stems);
        child.strands = std::move(p.strands);
    } else {
        child.stems.reserve(parent.stems.size() + 1);
        child.stems = parent.stems;
        child.strands.reserve(parent.strands.size() + 2);
        child.strands = parent.strands;
    }
    child.stems.push_back(stem);
    const SqStrand l{(int16_t)stem.i, (int16_t)stem.len, (int16_t)stem.j, 1, 1};
    const SqStrand r{(int16_t)(stem.j - stem.len + 1), (int16_t)stem.len, (int16_t)(stem.i + stem.len - 1), 1, 0};
    auto cmp = [](const SqStrand &x, const SqStrand &y) { return x.start < y.start; };
    child.strands.insert(std::upper_bound(child.strands.begin(), child.strands.end(), l, cmp), l);
    child.strands.insert(std::upper_bound(child.strands.begin(), child.strands.end(), r, cmp), r);
    child.anycross = parent.anycross;
    if (!child.anycross)
        for (size_t k = 0; k + 1 < child.stems.size(); k++) if (stems_cross(child.stems[k], stem)) { child.anycross = true; break; }
    if (child.anycross) {                                  // levels can change globally: full rule
        static thread_local std::vector<int> level;
        sq_stem_levels(child.stems, level);
        set_levels(child, level);
    }
}

// ---- round driver ---------------------------------------------------------------------------
static inline bool shares_base(const HStem &a, const HStem &b)       // SQRNdbnseq.py:783-786
{
    const int as0 = a.i, as1 = a.i + a.len - 1, at0 = a.j - a.len + 1, at1 = a.j;
    const int bs0 = b.i, bs1 = b.i + b.len - 1, bt0 = b.j - b.len + 1, bt1 = b.j;
    auto ov = [](int x0, int x1, int y0, int y1) { return x0 <= y1 && y0 <= x1; };
    return ov(as0, as1, bs0, bs1) || ov(as0, as1, bt0, bt1) || ov(at0, at1, bs0, bs1) || ov(at0, at1, bt0, bt1);
}


// the kernels of one round over S structures: state arrays, bit-diagonal scan, exact scoring (mode 0: + ScoreStems),
// and for host-driven greedy rounds the range filter that writes the round's output records
void sq_launch_round_kernels(sq_batch *b, hipStream_t st, int S, int maxn, int64_t maxcap, bool need_reacts, double scan_bytes,
                             int mode, const SqRoundIO &io, const SqScanArgs &scan, SqStruct *d_structs, SqStrand *d_strands,
                             bool chained, bool pooled, const SqPoolRoundArgs *pool_round)
{
    const bool crowded = b->inflight > 1 || b->njobs >= 4096;    // (by the batch, not by the launch: a batch's rounds all run one way)
    // short sequences on a crowded chip: state and scan in one launch, one wave per structure (sq_state_scan_kernel)
    static const bool no_fuse = getenv("SQ_NO_STATE_SCAN_FUSE") != nullptr;
    static const int st_short_env = getenv("SQ_STATE_SHORT_THREADS") ? atoi(getenv("SQ_STATE_SHORT_THREADS")) : 64;
    static const int sc_short_env = getenv("SQ_SCAN_SHORT_WAVES") ? atoi(getenv("SQ_SCAN_SHORT_WAVES")) : 1;
    const bool fuse = crowded && maxn <= 200 && maxn >= 5 && !no_fuse && st_short_env == 64 && sc_short_env == 1;
    // the pools' short structures: extension + state + scan + score + choose of a structure by ONE wave in ONE launch
    // (sq_pool_round.hip; pool_fold decides per fold and then launches no extend kernel)
    if (pool_round) {
        const SqPoolRoundLds lo = sq_pool_round_lds(pool_round->lds_n, pool_round->str_cap, pool_round->cell_entries, pool_round->surv_cap, pool_round->tmax);
        ProfScope ps(b, 3, scan_bytes);
        if (pool_round->root) hipLaunchKernelGGL(sq_pool_round_root_kernel, dim3(S), dim3(64), lo.total, st, b->ctx, scan, b->pool_io, *pool_round);
        else hipLaunchKernelGGL(sq_pool_round_kernel, dim3(S), dim3(64), lo.total, st, b->ctx, scan, b->pool_io, *pool_round);
        return;
    }
    if (fuse) {
        ProfScope ps(b, 2, scan_bytes);
        const size_t dyn_state = (size_t)7 * ((maxn + 8) & ~7) + 64, dyn_scan = 4 * (size_t)b->state.fbstride;
        // (the structure records go to the device by ONE copy first: read from the pinned array by every wave, each of the
        // launch's thousands of waves began with a read over PCIe -- 3.0 -> 1.2 G wave cycles per three headline steps,
        // the headline +1.6 %; SQ_NO_STATE_COPY: the old form)
        static const bool copy_first = getenv("SQ_NO_STATE_COPY") == nullptr;
        SqRoundIO io2 = io;
        if (copy_first && !chained && io.h_structs != io.d_structs) {
            hipMemcpyAsync(io.d_structs, io.h_structs, (size_t)S * sizeof(SqStruct), hipMemcpyHostToDevice, st);
            io2.h_structs = io.d_structs;
        }
        hipLaunchKernelGGL(sq_state_scan_kernel, dim3(S), dim3(64), std::max(dyn_state, dyn_scan), st, b->ctx, io2, b->state, scan, maxn, chained ? 1 : 0);
    }
    if (!fuse) {
        ProfScope ps(b, 1, 0);
        // the per-structure arrays are assembled in LDS (7 bytes per position) when the longest sequence fits
        const int st_lds_n = maxn <= 8000 ? maxn : 0;
        const size_t st_dyn = st_lds_n ? (size_t)7 * ((st_lds_n + 8) & ~7) + 64 : 0;
        // (sequences up to 200 nt: one wave builds the arrays in three or four steps; four waves per structure held four
        // times the wave slots for the same few microseconds -- with batches in flight the chip is short of exactly those)
        // "crowded": the chip is (or will be) short of wave slots -- several batches in flight, or a batch of four thousand
        // jobs and more (a 219-record batch alone -- 1,095 jobs -- keeps its rounds a latency chain).  Then a short structure gets ONE wave in the state, scan and scoring kernels; a small batch
        // alone keeps the wide blocks (its greedy rounds are a latency chain: one wave per structure made them 1.5 ms
        // longer per 219-record fold, hidden behind the blossom kernel only when there is one)
        static const int state_short = getenv("SQ_STATE_SHORT_THREADS") ? std::max(64, std::min(256, atoi(getenv("SQ_STATE_SHORT_THREADS")) / 64 * 64)) : 64;
        hipLaunchKernelGGL(sq_state_kernel, dim3(S), dim3(maxn <= 200 && crowded ? state_short : 256), st_dyn, st, b->ctx, io, b->state, scan, st_lds_n, chained ? 1 : 0);
    }
    // mode 0: the context tables of the round's structures (only long-sequence batches carry them)
    const bool ctx_on = mode == 0 && b->ctxtab.rec != nullptr && b->score_ctx;
    if (ctx_on) sq_launch_context(d_structs, d_strands, b->ctxtab, S, st);
    if (maxn >= 5 && !fuse) {
        ProfScope ps(b, 2, scan_bytes);
        // bit-diagonal scan: one wave = 64 anti-diagonals
        // (sequences up to 200 nt: one wave per structure walks all its diagonal groups, see the kernel)
        static const int scan_short = getenv("SQ_SCAN_SHORT_WAVES") ? std::max(1, atoi(getenv("SQ_SCAN_SHORT_WAVES"))) : 1;
        const int scan_groups = (2 * maxn - 5 + 63) / 64 + 1;
        hipLaunchKernelGGL(sq_scan6_kernel, dim3(S, maxn <= 200 && crowded ? std::min(scan_short, scan_groups) : scan_groups), dim3(64), 4 * (size_t)b->state.fbstride, st,
                           b->ctx, d_structs, b->state, scan);
    }
    {
        ProfScope ps(b, 3, 0);
        // dynamic LDS: letter codes of the longest sequence, plus its reactivities when they fit in 32 KiB
        const int lds_n = maxn <= 16384 ? maxn : 0;
        static const int nr_lim = getenv("SQ_SCORE_NR_LIM") ? atoi(getenv("SQ_SCORE_NR_LIM")) : 4096;
        // (only when some job needs them: sequences whose reactivities go through the cell table leave the room to the
        // partner / prefix arrays -- S2000 with encoded SHAPE: 16 KB that pushed those arrays out to global memory)
        const int lds_nr = need_reacts && maxn <= nr_lim ? maxn : 0;
        // partner / prefix arrays (3 x int16) too, while a block stays small enough for four blocks per CU
        // (the reactivity case is bound by fp64 sqrt/div throughput and prefers the occupancy)
        static const size_t state_lim = getenv("SQ_SCORE_STATE_LIM") ? (size_t)atol(getenv("SQ_SCORE_STATE_LIM")) : 24 * 1024;
        const size_t dyn_base = lds_n ? (size_t)((lds_n + 15) & ~15) + (size_t)8 * lds_nr + 16 : 0;
        const int lds_ns = (lds_n && mode == 0 && dyn_base + (size_t)6 * ((maxn + 8) & ~7) <= state_lim) ? maxn : 0;
        size_t dyn = lds_n ? (size_t)((lds_n + 15) & ~15) + (size_t)8 * lds_nr + (size_t)6 * ((lds_ns + 8) & ~7) + 16 : 0;
        // few structures: deal each structure's candidates to several blocks so that the launch still fills the chip
        static const int score_threads = getenv("SQ_SCORE_THREADS") ? atoi(getenv("SQ_SCORE_THREADS")) : 0;
        static const int score_parts = getenv("SQ_SCORE_PARTS") ? atoi(getenv("SQ_SCORE_PARTS")) : 0;
        static const int score_target = getenv("SQ_SCORE_TARGET") ? atoi(getenv("SQ_SCORE_TARGET")) : 512;
        // mode 0 (two-phase loop): ~512 blocks of 512 threads; the one-pass modes want many small blocks in flight
        int parts = std::max(1, std::min({512, ((mode == 0 ? score_target : 4096) + S - 1) / S, (int)(maxcap / 1024)}));
        if (score_parts) parts = score_parts;
        // mode 0: 512 threads per structure suit long sequences (S1000: 4.6 ms against 5.4 ms; S2000: 28 against 36);
        // short ones leave half of such a block idle behind its set-up (n = 300: 10,000 chains 5.9 -> 4.6 ms, pools
        // of a thousand 24 -> 16 ns per structure and round with 256).  SRtest150 (up to ~500 nt) measures the same
        // either way within the run-to-run spread and keeps 512.
        static const int short_thr = getenv("SQ_SCORE_SHORT_THREADS") ? atoi(getenv("SQ_SCORE_SHORT_THREADS")) : 64;
        // (the pools' generations of thousands of structures: most of them late in their fold, with one or two thousand candidates
        // left -- 500nobpp on 500-nt sequences: 382 / 315 / 301 ms per 500 sequences with 512 / 256 / 128 threads)
        const int pooled_thr = getenv("SQ_SCORE_POOL_THREADS") ? std::max(64, std::min(1024, atoi(getenv("SQ_SCORE_POOL_THREADS")) / 64 * 64)) : 128;
        const int thr0 = maxn <= 200 ? (crowded ? short_thr : 128) : (pooled && S >= 2048 ? std::min(pooled_thr, maxn <= 400 ? 256 : 512) : (maxn <= 400 ? 256 : 512));
        // (the one-pass modes on a crowded chip: a structure of a short sequence has ~150 candidates -- one wave, not four)
        const int thr = score_threads ? score_threads : (mode == 0 ? thr0 : (maxn <= 200 && crowded ? 64 : (parts == 1 && S < 2048 ? 512 : 256)));
        // the cell table (K R x (K R | 1) doubles for the batch's largest K R), then
        // mode 0: list of the bpscore survivors of a chunk (5 x threads entries of 8 + 4 + 2 bytes) behind the tables
        const int cell_off = (int)((dyn + 15) & ~(size_t)15);
        dyn = (size_t)cell_off + 8 * (size_t)b->cell_entries;
        const int surv_off = (int)((dyn + 15) & ~(size_t)15);
        dyn = (size_t)surv_off + (size_t)14 * (SQ_SCORE_CHUNK + (mode == 0 ? 1 : 0)) * thr;   // (one-pass modes: no carry-over)
        // mode 0: the structure's strands + skip pointers (10 bytes each) for the longest list a structure of this launch
        // can have -- device-booked rounds: two strands per stem of the batch's longest stem list; host-driven: 1,024
        const int str_cap = mode == 0 ? ((chained || pooled) ? std::min(1024, 2 * std::max(b->chain_tmax, 1) + 2) : 1024) : 0;
        const int str_off = (int)((dyn + 15) & ~(size_t)15);
        if (mode == 0) dyn = (size_t)str_off + (size_t)10 * str_cap + 16;
        if (mode == 0)
            hipLaunchKernelGGL(sq_score_kernel, dim3(S, parts), dim3(thr), dyn, st, b->ctx, d_structs, d_strands, b->state,
                               scan, io, lds_n, lds_nr, lds_ns, surv_off, cell_off, str_off, str_cap, ctx_on ? b->ctxtab : SqCtxTab{}, b->score_bound ? 1 : 0);
        else
            hipLaunchKernelGGL(sq_bps_kernel, dim3(S, parts), dim3(thr), dyn, st, b->ctx, d_structs, d_strands, b->state,
                               scan, io, mode, lds_n, lds_nr, surv_off, cell_off);
        if (mode == 0 && !chained)
            hipLaunchKernelGGL(sq_select_kernel, dim3(S, std::max(1, parts / 2)), dim3(256), 0, st, b->ctx, d_structs, scan, io);
        if (chained && !pooled) {
            SqChainIO cio = b->chain;
            // dynamic LDS: the level scratch for the longest stem list any job of the batch can reach
            const size_t ext_lds = sq_extend_lds_bytes(b->chain_tmax);
            if (ext_lds > 64 * 1024) sq_max_dynamic_lds((const void *)sq_chain_kernel, 160 * 1024);
            hipLaunchKernelGGL(sq_chain_kernel, dim3(S), dim3(64), ext_lds, st, b->ctx, d_structs, scan, cio, b->chain_tmax);
        }
        if (pooled) {
            // survivors within subopt x best the choose kernel sorts in LDS (18 bytes each): 1,024 for long sequences, 384 up to
            // 200 nt (measured on SRtest150 under nobpp / alt / greedynobpp: at most a few dozen are ever in range)
            static const int short_surv = getenv("SQ_POOL_SHORT_NSURV") ? std::max(64, std::min(1024, atoi(getenv("SQ_POOL_SHORT_NSURV")))) : 384;
            const int nsurv = maxn <= 200 ? short_surv : 1024;
            hipLaunchKernelGGL(sq_pool_choose_kernel, dim3(S), dim3(64), (size_t)18 * nsurv + 16, st, b->ctx, d_structs, scan, b->pool_io, nsurv);
        }
    }
}

static int run_chunk(sq_batch *b, SqLane &ln, const std::vector<SView> &structs, size_t lo, size_t hi, int mode,
                     std::vector<std::vector<HStem>> &out, const AlignSink *sink = nullptr)
{
    const int S = (int)(hi - lo);
    long long cpu_t0 = g_cpuacc_on ? CpuScope::now() : 0;
    int nstrand = 0, maxn = 0; int64_t cand_off = ln.cand0, maxcap = 0; double scan_bytes = 0;
    bool need_reacts = false;       // some job computes its reactivity factors per cell (float reactivities, or too many levels for the cell table)
    double tp0 = now_s();
    for (int s = 0; s < S; s++) {
        const SView &hs = structs[lo + s];
        const SqJob &J = b->jobs[hs.job];
        SqStruct &d = ln.h_structs[s];
        d.job = hs.job; d.slot = ln.slot0 + s; d.subopt = hs.subopt; d.cand_off = cand_off;
        cand_off += J.cand_cap; maxcap = std::max<int64_t>(maxcap, J.cand_cap);
        d.strand_off = nstrand; d.nstrand = (int)hs.st->strands.size();
        if (d.nstrand) memcpy(ln.h_strands + nstrand, hs.st->strands.data(), sizeof(SqStrand) * (size_t)d.nstrand);
        nstrand += d.nstrand;
        maxn = std::max(maxn, J.n);
        need_reacts |= !J.default_reacts && !(J.react_levels > 0 && b->pset_classes[J.pset] * J.react_levels <= 32);
        scan_bytes += 2.0 * J.n * J.n;                     // algorithmic: fp32 upper triangle, N^2/2 cells
    }
    hipStream_t st = ln.stream ? ln.stream : b->stream;
    g_t[0] += now_s() - tp0; tp0 = now_s();
    SqRoundIO io;
    io.h_structs = ln.h_structs; io.h_strands = ln.h_strands; io.d_structs = ln.d_structs; io.d_strands = ln.d_strands;
    io.h_out = ln.h_out; io.d_out = ln.d_out; io.h_cap = ln.h_out_cap; io.out_cap = ln.out_cap;
    io.h_ctr = ln.h_ctr; io.h_seq = ln.h_seq;
    SqScanArgs scan = b->scan;                           // this lane's counters
    scan.ctr = ln.d_ctr;
    sq_launch_round_kernels(b, st, S, maxn, maxcap, need_reacts, scan_bytes, mode, io, scan, ln.d_structs, ln.d_strands, false);
    {
        if (mode == 2) {
            // gap maps of the chunk's sequences into the (unused) round output buffer, then one scatter launch per
            // sequence, in list order: stream order == the reference's per-cell summation order (dbnali:233-237)
            int32_t *d_cols = (int32_t *)ln.d_out;
            const int32_t c0 = sink->col_off[lo], c1 = sink->col_off[hi];
            if ((size_t)(c1 - c0 + S) * 4 > (size_t)ln.out_cap * sizeof(SqOut)) { sq_set_error("gap maps do not fit the round buffer"); return -3; }
            HIPCK(hipMemcpyAsync(d_cols, sink->cols + c0, (size_t)(c1 - c0) * 4, hipMemcpyHostToDevice, st));
            // order-free chunk (dyadic weights, no reactivity factors, no caller matrices): every sum is exact, so one
            // launch with atomic adds gives the same bits as the sequential order
            static const bool no_atomic = getenv("SQ_ALIGN_SEQUENTIAL") != nullptr;
            bool order_free = !no_atomic;
            int64_t maxcap = 1;
            for (int k = 0; k < S && order_free; k++) {
                const SqJob &J = b->jobs[structs[lo + k].job];
                order_free = J.default_reacts && J.mat64_off < 0 && !J.mulsh && b->pset_dyadic[J.pset];
                maxcap = std::max<int64_t>(maxcap, J.cand_cap);
            }
            if (order_free) {
                std::vector<int32_t> starts(S);
                for (int k = 0; k < S; k++) starts[k] = sink->col_off[lo + k] - c0;
                int32_t *d_starts = d_cols + (c1 - c0);
                HIPCK(hipMemcpyAsync(d_starts, starts.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
                HIPCK(hipStreamSynchronize(st));             // (starts is a local)
                const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>(maxcap / 4096, 1), 64);
                // algorithmic bytes of the scatter: a read-modify-write of 8 bytes per cell of every kept stem -- booked by the
                // caller from the stems' cells; here the launch is only timed
                ProfScope ps(b, 8, 0);
                // (grid.x = the sequence: blocks that run at the same time hold the same part of different rows' stem lists -- the rows
                // of an alignment have their stems in the same places, so the cells they add to are the same lines of the matrix)
                hipLaunchKernelGGL(sq_scatter_all_kernel, dim3(S, blocks), dim3(256), 0, st, b->ctx, ln.d_structs, scan,
                                   d_cols, d_starts, sink->L, sink->matrix);
            } else {
            ProfScope ps(b, 8, 0);
            for (int k = 0; k < S; k++) {
                const SqJob &J = b->jobs[structs[lo + k].job];
                const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>(J.cand_cap / 1024, 1), 1024);
                hipLaunchKernelGGL(sq_scatter_kernel, dim3(blocks), dim3(256), 0, st, b->ctx, ln.d_structs, scan, k,
                                   d_cols + (sink->col_off[lo + k] - c0), sink->L, sink->matrix);
            }
            }
        }
    }
    const uint32_t seq = ++*ln.round_seq;
    hipLaunchKernelGGL(sq_done_kernel, dim3(1), dim3(1), 0, st, io, scan, seq);
    HIPCK(hipGetLastError());
    if (g_cpuacc_on) { const long long t = CpuScope::now(); g_cpuacc[mode == 1 ? 7 : 5] += t - cpu_t0; cpu_t0 = t; }
    // wait for the round: spin on the sequence number in pinned memory (no driver round trip); a stuck or
    // faulted queue is caught by polling the stream now and then
    { const int wr = sq_wait_word(b, ln.h_seq, seq, st, "round kernels"); if (wr) return wr; }
    if (g_cpuacc_on) g_cpuacc[6] += CpuScope::now() - cpu_t0;
    const SqCounters ctr = *ln.h_ctr;
    if (ctr.cand_ovf) { sq_set_capacity_error(SQ_CAP_CANDIDATES, "candidate capacity exceeded (raise cand_per_nt)"); return -3; }
    if (ctr.out_ovf) { ln.out_ovf_seen = true; sq_set_capacity_error(SQ_CAP_OUTPUT, "round output capacity exceeded (the records of a round: grows with cand_per_nt)"); return -3; }
    if (ctr.level_ovf) { sq_set_error("more than 64 pseudoknot levels"); return -3; }
    const uint32_t nout = ctr.nout;
    const SqOut *ho = ln.h_out;
    if (nout > ln.h_out_cap) {                               // rare: the tail of a huge round sits in device memory
        ln.big_out.resize(nout);
        memcpy(ln.big_out.data(), ln.h_out, sizeof(SqOut) * (size_t)ln.h_out_cap);
        HIPCK(hipMemcpy(ln.big_out.data() + ln.h_out_cap, ln.d_out + ln.h_out_cap,
                        sizeof(SqOut) * (size_t)(nout - ln.h_out_cap), hipMemcpyDeviceToHost));
        ho = ln.big_out.data();
    }
    g_t[1] += now_s() - tp0;
    if (mode == 2) return 0;
    TScope tpost(2);
    CpuScope cpu_post(4);
    // bucket by structure
    std::vector<uint32_t> &cnt = ln.post_cnt, &idx = ln.post_idx, &fillp = ln.post_fill;   // (kept per lane: no allocation per round)
    cnt.assign(S + 1, 0);
    for (uint32_t k = 0; k < nout; k++) cnt[ho[k].st + 1]++;
    for (int s = 0; s < S; s++) cnt[s + 1] += cnt[s];
    idx.resize(nout); fillp.assign(cnt.begin(), cnt.end() - 1);
    for (uint32_t k = 0; k < nout; k++) idx[fillp[ho[k].st]++] = k;
    auto post_one = [&](int s) {
        uint32_t *p0 = idx.data() + cnt[s], *p1 = idx.data() + cnt[s + 1];
        std::vector<HStem> &res = out[lo + s];
        res.clear();
        if (p0 == p1) return;
        auto mk = [&](uint32_t k) {
            const SqOut &o = ho[k];
            const int i0 = (int)(o.key & 0xFFFFu), sdiag = (int)(o.key >> 16);
            return HStem{i0, sdiag - i0, o.len, o.bps, o.fin};
        };
        if (mode == 1) {                                    // emission order: (s, i) ascending
            std::sort(p0, p1, [&](uint32_t x, uint32_t y) { return ho[x].key < ho[y].key; });
            res.reserve((size_t)(p1 - p0));
            for (uint32_t *p = p0; p < p1; p++) res.push_back(mk(*p));
            return;
        }
        // ChooseStems (SQRNdbnseq.py:754-789): stable descending sort == (fin desc, emission key asc)
        std::sort(p0, p1, [&](uint32_t x, uint32_t y) {
            if (ho[x].fin != ho[y].fin) return ho[x].fin > ho[y].fin;
            return ho[x].key < ho[y].key;
        });
        res.push_back(mk(*p0));
        for (uint32_t *p = p0 + 1; p < p1; p++) {           // range filter already applied on device (:778)
            const HStem cand = mk(*p);
            bool all_conf = true;
            for (const HStem &r : res) if (!shares_base(cand, r)) { all_conf = false; break; }
            if (all_conf) res.push_back(cand);
        }
    };
    // structures are independent: big rounds (the AnnotateStems passes of E/H/N) share the sorting among the pool
    if (nout >= 16384) sq_pool(b)->parallel_for(S, post_one);
    else for (int s = 0; s < S; s++) post_one(s);
    return 0;
}

// AnnotateStems(bool, score, rbps, [], minlen, minbpscore) (:553) for a list of jobs, the stems LEFT ON THE DEVICE: structure k
// of the round = jobs[k] with no selected stems, its survivors (SqOk records) in its slice of the candidate arena, and per
// job the sizes the host needs to lay out the matching step (sq_algos_dev.hip).  One round, one wait.  Returns 1 when the
// jobs do not fit one round of the full lane (the caller keeps the host-driven form).
int sq_round_annotate_dev(sq_batch *b, const std::vector<int> &jobs, SqAlgoSize *h_sizes, int64_t *cands_used, const SqAlgoRaw &raw)
{
    { int r = sq_prepare_scan(b); if (r) return r; }
    SqLane &ln = b->lane_full;
    const int S = (int)jobs.size();
    const int64_t avail = b->cand_records - b->cand_reserved;
    if (S > ln.max_structs) return 1;
    int maxn = 0; int64_t cand_off = 0, maxcap = 0; double scan_bytes = 0; bool need_reacts = false;
    for (int s = 0; s < S; s++) {
        const SqJob &J = b->jobs[jobs[s]];
        if (cand_off + J.cand_cap > avail / 2) return 1;    // (the other half of the arena may be lent to the matching kernels)
        SqStruct &d = ln.h_structs[s];
        d.job = jobs[s]; d.slot = s; d.subopt = 1.0; d.cand_off = cand_off; d.strand_off = 0; d.nstrand = 0;
        cand_off += J.cand_cap; maxcap = std::max<int64_t>(maxcap, J.cand_cap);
        maxn = std::max(maxn, J.n);
        need_reacts |= !J.default_reacts && !(J.react_levels > 0 && b->pset_classes[J.pset] * J.react_levels <= 32);
        scan_bytes += 2.0 * J.n * J.n;
    }
    *cands_used = cand_off;
    hipStream_t st = b->stream;
    SqRoundIO io;
    io.h_structs = ln.h_structs; io.h_strands = ln.h_strands; io.d_structs = ln.d_structs; io.d_strands = ln.d_strands;
    io.h_out = ln.h_out; io.d_out = ln.d_out; io.h_cap = ln.h_out_cap; io.out_cap = ln.out_cap;
    io.h_ctr = ln.h_ctr; io.h_seq = ln.h_seq;
    SqScanArgs scan = b->scan;
    scan.ctr = ln.d_ctr;
    sq_launch_round_kernels(b, st, S, maxn, maxcap, need_reacts, scan_bytes, 2, io, scan, ln.d_structs, ln.d_strands, false);
    hipLaunchKernelGGL(sq_algo_sizes_kernel, dim3(S), dim3(b->inflight > 1 || b->njobs >= 4096 ? 64 : 256), 0, st, b->ctx, ln.d_structs, scan, h_sizes, raw);
    const uint32_t seq = ++*ln.round_seq;
    hipLaunchKernelGGL(sq_done_kernel, dim3(1), dim3(1), 0, st, io, scan, seq);
    HIPCK(hipGetLastError());
    { const int wr = sq_wait_word(b, ln.h_seq, seq, st, "AnnotateStems round"); if (wr) return wr; }
    const SqCounters ctr = *ln.h_ctr;
    if (ctr.cand_ovf) { sq_set_capacity_error(SQ_CAP_CANDIDATES, "candidate capacity exceeded (raise cand_per_nt)"); return -3; }
    return 0;
}

int sq_run_round(sq_batch *b, const std::vector<SView> &structs, int mode, std::vector<std::vector<HStem>> &out)
{
    return sq_run_round_impl(b, b->lane_full, structs, mode, out, nullptr);
}
// `ln`: the round buffers to use.  The full lane ends where the arena is lent to matching kernels in flight
// (cand_reserved); the half lanes are set up by sq_fold.
int sq_run_round_impl(sq_batch *b, SqLane &ln, const std::vector<SView> &structs, int mode,
                          std::vector<std::vector<HStem>> &out, const AlignSink *sink)
{
    { int r = sq_prepare_scan(b); if (r) return r; }
    out.resize(structs.size());
    const int64_t avail = &ln == &b->lane_full ? b->cand_records - b->cand_reserved : ln.cand_records;
    size_t lo = 0, limit = (size_t)ln.max_structs;
    while (lo < structs.size()) {
        size_t hi = lo; int64_t cands = 0, strands = 0;
        while (hi < structs.size() && hi - lo < limit) {
            const SqJob &J = b->jobs[structs[hi].job];
            const int64_t ns = (int64_t)structs[hi].st->strands.size();
            if (hi > lo && (cands + J.cand_cap > avail || strands + ns > ln.strand_cap)) break;
            cands += J.cand_cap; strands += ns; hi++;
        }
        if (cands > avail || strands > ln.strand_cap) { sq_set_error("structure does not fit the round buffers"); return -3; }
        ln.out_ovf_seen = false;
        int r = run_chunk(b, ln, structs, lo, hi, mode, out, sink);
        // more stems than the round output holds (AnnotateStems passes of thousands of records): the same structures in
        // smaller chunks.  (Not with a sink: the alignment matrix has already taken part of the chunk.)
        if (r == -3 && ln.out_ovf_seen && !sink && hi - lo > 1) { limit = (hi - lo) / 2; continue; }
        if (r) return r;
        lo = hi;
    }
    return 0;
}

// ---- a-2..a-6 C ABI ---------------------------------------------------------------------------
extern "C" int sq_optimal_stems(sq_batch *b, int32_t nstruct, const int32_t *struct_job, const int32_t *stem_off,
                                const sq_stem *stems, const double *subopt, int32_t mode,
                                sq_stem *out, int32_t out_cap, int32_t *out_off)
{
    SqSlackGuard slack_guard;
    if (!b || nstruct < 0 || (mode != 0 && mode != 1)) { sq_set_error("bad argument"); return -1; }
    std::vector<HStruct> hs(nstruct);
    std::vector<SView> views(nstruct);
    for (int s = 0; s < nstruct; s++) {
        if (struct_job[s] < 0 || struct_job[s] >= b->njobs) { sq_set_error("bad job index"); return -1; }
        hs[s].job = struct_job[s];
        hs[s].subopt = subopt ? subopt[s] : 1.0;
        const int n = b->jobs[hs[s].job].n;
        for (int k = stem_off[s]; k < stem_off[s + 1]; k++) {
            const sq_stem &t = stems[k];
            if (t.len < 1 || t.i < 0 || t.j >= n || t.i + t.len - 1 >= t.j - t.len + 1) { sq_set_error("bad stem"); return -1; }
            hs[s].stems.push_back(HStem{t.i, t.j, t.len, t.bpscore, t.finscore});
        }
        sq_build_strands(hs[s]);
        views[s] = SView{hs[s].job, hs[s].subopt, &hs[s]};
    }
    std::vector<std::vector<HStem>> res;
    int r = sq_run_round(b, views, mode, res);
    if (r) return r;
    int32_t o = 0;
    for (int s = 0; s < nstruct; s++) {
        out_off[s] = o;
        for (const HStem &t : res[s]) {
            if (o >= out_cap) { sq_set_error("out_cap too small"); return -3; }
            out[o++] = sq_stem{t.i, t.j, t.len, 0, t.bps, t.fin};
        }
    }
    out_off[nstruct] = o;
    return 0;
}

// ---- alignment step 1 --------------------------------------------------------------------------------
extern "C" int sq_align_accumulate(sq_batch *b, int32_t njob, const int32_t *job_ids, const int32_t *col_off,
                                   const int32_t *cols, int32_t L, double *d_matrix)
{
    SqSlackGuard slack_guard;
    if (!b || njob < 0 || !job_ids || !col_off || !cols || L <= 0 || !d_matrix) { sq_set_error("bad argument"); return -1; }
    std::vector<HStruct> hs(njob);
    std::vector<SView> views(njob);
    for (int k = 0; k < njob; k++) {
        const int j = job_ids[k];
        if (j < 0 || j >= b->njobs) { sq_set_error("bad job index"); return -1; }
        const int n = b->jobs[j].n;
        if (col_off[k + 1] - col_off[k] != n) { sq_set_error("gap map length differs from the sequence length"); return -1; }
        for (int p = 0; p < n; p++) {
            const int c = cols[col_off[k] + p];
            if (c < 0 || c >= L || (p && c <= cols[col_off[k] + p - 1])) { sq_set_error("gap map is not increasing inside [0, L)"); return -1; }
        }
        hs[k].job = j; views[k] = SView{j, 1.0, &hs[k]};
    }
    AlignSink sink{col_off, cols, L, d_matrix};
    std::vector<std::vector<HStem>> unused;
    const double t0 = now_s();
    for (int k = 0; k < 8; k++) g_t[k] = 0;
    int r = sq_run_round_impl(b, b->lane_full, views, 2, unused, &sink);
    if (!r) {
        const unsigned nt = (unsigned)((L + 31) / 32);
        hipLaunchKernelGGL(sq_mirror_kernel, dim3(nt, nt), dim3(256), 0, b->stream, d_matrix, L);
        r = sq_check(hipStreamSynchronize(b->stream), "sq_mirror_kernel");
    }
    if (b->sw.timing)
        fprintf(stderr, "[sq_align_accumulate] %d sequences: %.3f ms (prep %.3f, gpu+wait %.3f)\n", njob, (now_s() - t0) * 1e3,
                g_t[0] * 1e3, g_t[1] * 1e3);
    return r;
}

extern "C" int sq_colmatrix_select(const double *d_matrix, int32_t L, double threshold, int32_t minspan,
                                   int64_t *d_idx, double *d_val, int64_t cap, uint64_t *d_count, void *hip_stream)
{
    if (!d_matrix || L <= 0 || cap < 0 || !d_count || (cap && (!d_idx || !d_val))) { sq_set_error("bad argument"); return -1; }
    hipStream_t st = (hipStream_t)hip_stream;
    HIPCK(hipMemsetAsync(d_count, 0, 8, st));
    const int64_t total = (int64_t)L * L;
    (void)total;
    hipLaunchKernelGGL(sq_colselect_kernel, dim3((unsigned)std::min<int32_t>(L, 2048)), dim3(256), 0, st,
                       d_matrix, L, threshold, minspan, (long long *)d_idx, d_val, (long long)cap, (unsigned long long *)d_count);
    return sq_check(hipGetLastError(), "sq_colselect_kernel");
}

