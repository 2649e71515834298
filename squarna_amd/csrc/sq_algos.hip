// sq_algos.hip -- host driver of RunAlgo (SQRNdbnseq.py:548-595) for the E / H / N algorithms:
// one AnnotateStems pass on the GPU (scan + exact rescoring), the matching / DP on the GPU
// (sq_match.hip), then the reference's stem filters on the host.
#include <algorithm>
#include <functional>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>
#include "sq_host.h"
#include "sq_match.h"
#include "sq_blossom.h"
#include "sq_algos_dev.h"
#include "sq_cells.h"

#define HIPCK(x) do { int _r = sq_check((x), #x); if (_r) return _r; } while (0)

typedef std::pair<int, int> BP;

#include <chrono>
static inline double sq_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// exact scoremat cell on the host: the same fp64 expression as sq_cell_score (sq_kernels.hip)
static double cell_score_host(const sq_batch *b, const SqJob &J, int i, int j, const double *dense = nullptr)
{
    if (dense) return dense[(size_t)sq_m64_index(J, i, j)];   // jobs with a bpp term / multiplier: the device's exact matrix
    const uint8_t *codes = b->codes.data() + J.pos_off;
    const sq_paramset &ps = b->psets[J.pset];
    const double w = ps.bpweight[codes[i] * 32 + codes[j]];
    double rf = 1.0;
    if (!J.default_reacts) {
        if (J.rf_idx >= 0) {                               // the host-libm table the device reads (sq_reactfactor)
            const uint8_t *lv = b->ridx.data() + J.pos_off;
            rf = b->rftab[(size_t)J.rf_idx * 256 + lv[i] * 16 + lv[j]];
        } else {
            const double *r = sq_host_reacts(b) + J.pos_off;
            rf = sqrt((1.0 - (r[i] + r[j]) / 2.0) * 2.0);
        }
    }
    if (w <= 0) rf = 1.0 / (rf > 0.01 ? rf : 0.01);
    return w * rf;
}

// PairsToStems (SQRNdbnseq.py:498-517) on sorted pairs
static void pairs_to_stems(const std::vector<BP> &sp, std::vector<HStem> &out)
{
    out.clear();
    for (size_t k = 0; k < sp.size(); k++) {
        if (k && sp[k - 1].first + 1 == sp[k].first && sp[k - 1].second == sp[k].second + 1) out.back().len++;
        else out.push_back(HStem{sp[k].first, sp[k].second, 1, 0, 0});
    }
}

// the two filter passes of RunAlgo (:570-595)
// The pairs of a matching form stems (maximal stacks).  All base pairs of a stem cross the same pairs, so PairsToDBN's
// level rule (:104-150: crossing counts, order by (count, start), first fit, groups ranked by size) gives every stem
// one level and can be evaluated per STEM with the stem lengths as weights (sq_stem_levels; DESIGN.md section 5 has the
// argument) -- a dozen stems instead of a hundred pairs, and the rule is quadratic.  Dropping whole stems (score filter,
// level limit) never changes how the remaining pairs stack, so the stems stay the stems.
static void filter_stemset(const sq_batch *b, const SqJob &J, std::vector<BP> pairs, int levellimit,
                           std::vector<HStem> &stemset, const double *dense = nullptr)
{
    const sq_paramset &ps = b->psets[J.pset];
    for (BP &p : pairs) if (p.first > p.second) std::swap(p.first, p.second);
    std::sort(pairs.begin(), pairs.end());
    static thread_local std::vector<HStem> stems, kept, lim;
    static thread_local std::vector<int> lv;
    pairs_to_stems(pairs, stems);
    auto score_of = [&](const HStem &st) {
        double s = 0;                                                    // sum(...) from int 0, left to right
        for (int k = 0; k < st.len; k++) s = s + cell_score_host(b, J, st.i + k, st.j - k, dense);
        return s;
    };
    kept.clear();
    for (HStem &st : stems) {
        st.bps = score_of(st);
        if (st.bps >= ps.minbpscore && (double)st.len >= ps.minlen) kept.push_back(st);
    }
    // DBNToPairs(PairsToDBN(pairs, N, levellimit)) : drop pseudoknot levels above the limit (:581)
    sq_stem_levels(kept, lv);
    lim.clear();
    for (size_t k = 0; k < kept.size(); k++)
        if ((levellimit < 0 || lv[k] <= levellimit) && lv[k] <= 49) lim.push_back(kept[k]);   // 49 bracket types exist
    sq_stem_levels(lim, lv);                                             // :582 levels of what is left
    stemset.clear();
    for (size_t k = 0; k < lim.size(); k++) {
        const HStem &st = lim[k];
        if (lv[k] > 1 && st.len < 4) continue;                           // :589 short pseudoknotted stems
        const double sc = st.bps;                                        // (the same sum as in the first pass)
        if (sc >= ps.minbpscore && (double)st.len >= ps.minlen) stemset.push_back(HStem{st.i, st.j, st.len, sc, sc});
    }
}

// ---- one chunk of matching work: staged (kernel launch, asynchronous) and collected later ------
struct SqAlgoChunk {
    int algo = 0;
    size_t k0 = 0, k1 = 0;                       // jobs[k0, k1) of the caller's list
    std::vector<SqMatchJob> mj;
    std::vector<std::vector<int>> vid2pos;       // Edmonds: graph vertex -> position
    std::vector<uint64_t> seen_hash;             // (SQ_MWM_POSTHOC) hash of every job's result as the collector read it
    SqMatchJob *p_mj = nullptr; SqAlgoJob *p_aj = nullptr; SqAlgoStat *h_stats = nullptr;   // pinned, device-side RunAlgo: the
                                                 // job table in list order, the per-job records, the finish kernel's statistics
    SqMatchJob *p_jobs = nullptr;                // pinned staging: job table and edge list (read by the kernels in place)
    SqMatchEdge *p_edges = nullptr;
    size_t nedges = 0, vids = 0;                 // vids: graph vertices of all Edmonds jobs (device-side RunAlgo: vid2pos)
    size_t outints = 0, scratch = 0, bytes = 0;  // device bytes used from the region's base
    int32_t *d_out = nullptr, *d_cnt = nullptr;  // results: in the pinned staging buffer, written by the kernels in place
    uint32_t *flag = nullptr; uint32_t flag_val = 0;   // pinned completion word published by sq_flag_kernel
    int32_t *bin_head = nullptr;                 // pinned, Edmonds only: first job (row of the sorted table) of every block
    uint32_t *job_flags = nullptr;               // pinned, Edmonds only: per job "mates are in host memory" (== flag_val),
                                                 // indexed by the job's position in the SORTED table (sorted_pos)
    hipStream_t st = nullptr;
    // Edmonds / Hungarian: the kernel's job table is sorted by LDS need (largest first) and launched in size classes,
    // every class with the dynamic LDS of ITS largest job.  A launch-wide LDS size gives each of the 219 SRtest150
    // graphs the 130 KB of the largest one: one block per CU, so two batches in flight already hold every CU's LDS and
    // the 56 KB blocks of the scoring kernel of ALL batches wait for a blossom block to finish (traced on MI355X with 8
    // batches in flight: sq_score_kernel 31 -> 137 us, sq_bps_kernel 33 -> 783 us per launch, profiles/r02j_*).
    // Streams: class 0 of Edmonds (the largest graphs, the critical path) runs on the Edmonds side stream, its other
    // classes go -- largest first -- in front of the Hungarian / Nussinov kernels on THEIR stream: no extra streams
    // (see side_of), and the whole sequence still ends before class 0 does.
    struct Class { int start, count; };
    std::vector<Class> classes;
    std::vector<int> sorted_pos;                 // job q of mj -> row of the sorted table
    std::vector<SqMatchJob> sorted;              // host copy of the sorted table
};

namespace {
// pow(x, 1.7) through the host libm (SQRNalgos.py:101,122), memoised: the stem scores of one sequence take few
// distinct values (sums of a handful of pair weights), and pow dominates the edge build otherwise
struct PowCache {
    uint64_t key[64]; double val[64]; bool used[64];
    PowCache() { for (bool &u : used) u = false; }
    double operator()(double x)
    {
        uint64_t k; memcpy(&k, &x, 8);
        unsigned h = (unsigned)((k * 0x9E3779B97F4A7C15ull) >> 58);
        for (int t = 0; t < 8; t++, h = (h + 1) & 63) {
            if (used[h] && key[h] == k) return val[h];
            if (!used[h]) { used[h] = true; key[h] = k; return val[h] = pow(x, 1.7); }
        }
        return pow(x, 1.7);
    }
};
struct JobBuild { int n = 0; size_t ncell = 0, need = 0, nout = 0; };
}

// Side stream of staging slot `slot` (0 Edmonds, 1 Hungarian, 2 Nussinov).  The two short kernels share a stream: the
// runtime multiplexes all streams of the process onto a few hardware queues (GPU_MAX_HW_QUEUES), streams that share a
// queue serialise, and with several batches in flight every stream less keeps a 6 ms blossom kernel out of some other
// batch's round queue.  Hungarian + Nussinov back to back (1.4 + 1.5 ms) still end long before Edmonds does.
// One batch folded alone keeps Nussinov on a stream of its own (it then ends before the greedy loop does, and the
// collection of the short kernels does not wait behind Edmonds' second size class + Hungarian); sq_fold_concurrent
// with three or more batches in flight asks for the two-stream form.
// (slot 3: the stream of the blossom kernel's smaller size classes -- a stream of its own when the batch is folded alone,
// else in front of the short kernels on theirs.  Until the critical class took 2.2 ms they could sit in front of the
// Hungarian kernel: 1.0 + 1.5 ms then ended after the critical class did)
static inline int side_of(const sq_batch *b, int slot)
{
    static const int forced = getenv("SQ_SIDE_STREAMS") ? atoi(getenv("SQ_SIDE_STREAMS")) : 0;
    const int nside = forced ? forced : b->side_streams;
    if (slot == 3) return nside >= 3 ? 3 : (nside == 2 ? 1 : 0);
    return nside >= 3 ? slot : (nside == 2 ? (slot == 0 ? 0 : 1) : 0);
}

// pinned staging buffer `slot` of the batch, at least `bytes` large (grow-only)
static char *stage_buffer(sq_batch *b, int slot, size_t bytes)
{
    if (b->stage_cap[slot] < bytes) {
        if (b->stage_buf[slot]) { hipStreamSynchronize(slot < 3 && b->side[side_of(b, slot)] ? b->side[side_of(b, slot)] : b->stream); sq_pinned_put(b->stage_buf[slot]); }
        b->stage_buf[slot] = nullptr; b->stage_cap[slot] = 0;
        const size_t cap = bytes + bytes / 2 + 4096;
        if (sq_pinned_get((void **)&b->stage_buf[slot], cap)) return nullptr;
        memset(b->stage_buf[slot], 0, cap);               // (a cached buffer may hold completion flags of an earlier batch)
        b->stage_cap[slot] = cap;
    }
    return b->stage_buf[slot];
}

// Host part: edges and scratch layout of jobs[k0..) of `algo`, as many as fit into region_bytes (ck.k1, ck.bytes).
// The per-job edge lists are built by the worker pool, then packed into pinned staging buffer `slot`.
static int algo_build(sq_batch *b, const std::vector<int> &jobs, const std::vector<std::vector<HStem>> &stems, size_t k0,
                      int algo, size_t region_bytes, int slot, SqAlgoChunk &ck, const SqAlgoSize *dev_sizes = nullptr)
{
    ck = SqAlgoChunk();
    ck.algo = algo; ck.k0 = k0;
    const size_t nj = jobs.size() - k0;
    // pass 1 (pool): sizes only -- cells (= edges) and, for Edmonds, graph vertices (distinct positions)
    const double tb0 = sq_now();
    std::vector<JobBuild> jb(nj);
    auto size_one = [&](int q) {
        const SqJob &J = b->jobs[jobs[k0 + q]];
        JobBuild &B = jb[q];
        size_t ncell = 0;
        if (dev_sizes) ncell = (size_t)dev_sizes[k0 + q].nedges;        // (the stems stayed on the device: sq_algo_sizes_kernel)
        else for (const HStem &s : stems[k0 + q]) ncell += (size_t)s.len;
        B.ncell = ncell;
        if (algo == SQ_ALGO_E) {
            int nv = 0;
            if (dev_sizes) nv = dev_sizes[k0 + q].nv;
            else {
                const std::vector<HStem> &st_ = stems[k0 + q];
                static thread_local std::vector<char> seen;
                seen.assign((size_t)J.n, 0);
                for (const HStem &s : st_)
                    for (int t = 0; t < s.len; t++) {
                        if (!seen[s.i + t]) { seen[s.i + t] = 1; nv++; }
                        if (!seen[s.j - t]) { seen[s.j - t] = 1; nv++; }
                    }
            }
            B.n = nv; B.need = sq_mwm_scratch_bytes(nv, (int)ncell); B.nout = 2 * (size_t)nv + 2;      // mates + first-assignment ranks + (passes, events)
        } else {
            B.n = J.n;
            B.need = algo == SQ_ALGO_H ? sq_lsap_scratch_bytes(J.n) : sq_nussinov_scratch_bytes(J.n);
            B.nout = algo == SQ_ALGO_H ? (size_t)J.n : 2 * ((size_t)J.n + 4);
        }
        B.need = (B.need + 255) & ~(size_t)255;
    };
    // (with the sizes from the device a job costs a few instructions: waking the pool costs more than the loop)
    if (dev_sizes) for (int q = 0; q < (int)nj; q++) size_one(q);
    else sq_pool(b)->parallel_for((int)nj, size_one);
    const double tb1 = sq_now();
    std::vector<SqMatchJob> &mj = ck.mj;
    size_t scratch = 0, outints = 0, nedges = 0, k1 = k0, vids = 0;
    for (; k1 < jobs.size(); k1++) {
        const JobBuild &B = jb[k1 - k0];
        SqMatchJob m;
        m.edge_off = (int64_t)nedges; m.pos_off = b->jobs[jobs[k1]].pos_off;
        m.n = B.n; m.nedges = (int32_t)B.ncell;
        const size_t fixed = (mj.size() + 1) * sizeof(SqMatchJob) + (nedges + B.ncell) * sizeof(SqMatchEdge) +
                             (outints + B.nout + mj.size() + 1) * 4 + 4096 + (dev_sizes ? (vids + (size_t)B.n) * 4 + 1024 + (mj.size() + 1) * sizeof(SqAlgoJob) : 0);
        if (fixed + scratch + B.need > region_bytes) break;
        m.scratch_off = (int64_t)scratch; scratch += B.need;
        m.out_off = (int64_t)(algo == SQ_ALGO_N ? outints / 2 : outints); outints += B.nout;
        nedges += B.ncell;
        if (algo == SQ_ALGO_E) vids += (size_t)B.n;
        mj.push_back(m);
    }
    ck.k1 = k1; ck.outints = outints; ck.scratch = scratch; ck.nedges = nedges; ck.vids = vids;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t rr = o; o = (o + bytes + 255) & ~(size_t)255; return rr; };
    take(mj.size() * sizeof(SqMatchJob)); take(nedges * sizeof(SqMatchEdge) + 16);
    take(outints * 4 + 16); take(mj.size() * 4 + 16);
    if (dev_sizes) { take(vids * 4 + 16); take(sizeof(SqAlgoStat)); take(mj.size() * sizeof(SqAlgoJob) + 16); }   // (+ the finish kernel's job records: Carve::o_aj)
    ck.bytes = o + scratch;
    if (mj.empty()) return 0;
    // pinned staging: [jobs][edges][results][counts][completion word][per-job completion words]
    const size_t jbytes = (mj.size() * sizeof(SqMatchJob) + 255) & ~(size_t)255;
    const size_t ebytes = dev_sizes ? 0 : (nedges * sizeof(SqMatchEdge) + 255) & ~(size_t)255;   // (device-side RunAlgo: edges in the region)
    const size_t obytes = (outints * 4 + 255) & ~(size_t)255, cbytes = (mj.size() * 4 + 255) & ~(size_t)255;
    const size_t fbytes = algo == SQ_ALGO_E ? ((mj.size() * 4 + 255) & ~(size_t)255) : 0;   // job flags; the bin heads take as much again
    const size_t dbytes = dev_sizes ? jbytes + ((mj.size() * sizeof(SqAlgoJob) + 255) & ~(size_t)255) + 256 : 0;
    char *pin = stage_buffer(b, slot, jbytes + ebytes + obytes + cbytes + 256 + 2 * fbytes + dbytes);
    if (!pin) return 2;
    if (dev_sizes) {
        char *dp = pin + jbytes + ebytes + obytes + cbytes + 256 + 2 * fbytes;
        ck.p_mj = (SqMatchJob *)dp; ck.p_aj = (SqAlgoJob *)(dp + jbytes);
        ck.h_stats = (SqAlgoStat *)(dp + jbytes + ((mj.size() * sizeof(SqAlgoJob) + 255) & ~(size_t)255));
        memcpy(ck.p_mj, mj.data(), mj.size() * sizeof(SqMatchJob));
        memset(ck.h_stats, 0, sizeof(SqAlgoStat));
    }
    ck.p_jobs = (SqMatchJob *)pin; ck.p_edges = (SqMatchEdge *)(pin + jbytes);
    ck.d_out = (int32_t *)(pin + jbytes + ebytes); ck.d_cnt = (int32_t *)(pin + jbytes + ebytes + obytes);
    ck.flag = (uint32_t *)(pin + jbytes + ebytes + obytes + cbytes);
    ck.flag_val = ++b->algo_seq;
    *ck.flag = 0;
    if (fbytes) {
        ck.job_flags = (uint32_t *)(pin + jbytes + ebytes + obytes + cbytes + 256); memset(ck.job_flags, 0, mj.size() * 4);
        ck.bin_head = (int32_t *)(pin + jbytes + ebytes + obytes + cbytes + 256 + fbytes);
    }
    {
        // LDS size classes (see SqAlgoChunk::Class): jobs sorted by the dynamic LDS they need, largest first (Nussinov: by the
        // sequence length, which sizes its blocks: 64 threads per 64 positions)
        const size_t nq = mj.size();
        std::vector<size_t> need(nq);
        for (size_t q = 0; q < nq; q++)
            need[q] = algo == SQ_ALGO_E
                          ? SqBlossom::scratch_bytes(mj[q].n, mj[q].nedges, 1) + (((size_t)mj[q].nedges * sizeof(SqMatchEdge) + 15) & ~(size_t)15) + 64
                      : algo == SQ_ALGO_H
                          ? sq_lsap_lds_bytes(mj[q].n, mj[q].nedges)
                          : (size_t)std::max(1, (mj[q].n + 63) / 64);
        std::vector<int> ord(nq);
        for (size_t q = 0; q < nq; q++) ord[q] = (int)q;
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return need[x] > need[y]; });
        ck.sorted.resize(nq); ck.sorted_pos.resize(nq);
        for (size_t r = 0; r < nq; r++) { ck.sorted[r] = mj[ord[r]]; ck.sorted[r].pad = ord[r]; ck.sorted_pos[ord[r]] = (int)r; }   // (pad: the job's index -- the Nussinov kernel files its pair count under it)
        // a new class starts where twice as many blocks would fit a CU (and the current one has a few jobs)
        // Two classes for Edmonds, one for Hungarian: classes that FOLLOW each other on a stream each last as long as
        // their slowest job, so more of them lengthen the chain on the short kernels' stream past the end of Edmonds'
        // class 0 (traced: 4 + 4 classes end at 9.3 ms, one batch alone, instead of 7.2 ms)
        static const int env_classes = getenv("SQ_MWM_CLASSES") ? std::max(1, atoi(getenv("SQ_MWM_CLASSES"))) : 0;
        // (Edmonds: one launch, the graphs packed into multi-wave blocks by LDS need -- sq_mwm_plan)
        // and with one batch alone in two size classes of one-graph blocks, the critical class on its own stream)
        // Hungarian with the chip crowded (batches in flight, or thousands of jobs): every block of a launch gets the LDS of the
        // launch's LARGEST job (62 KB at 150 nt: two one-wave blocks per CU, and LDS the scoring kernels of the other batches
        // do not get) -- three classes give most jobs a quarter of that
        const int env_hclasses = b->sw.lsap_classes;           // (per fold: tests)
        const bool crowded = b->inflight > 1 || nq >= 4096;    // (a batch of 1,314 jobs alone: 7.1 ms without the classes, 8.5 with)
        const int max_classes = algo == SQ_ALGO_H ? (env_hclasses ? env_hclasses : (crowded ? 3 : 1))
                              : algo == SQ_ALGO_N ? (env_hclasses ? env_hclasses : (crowded ? 3 : 1))   // (blocks of 64 / 128 / 192 threads instead of 192 for every job)
                                                  : (env_classes ? env_classes : (b->inflight < 2 ? 2 : 1));
        size_t cur = std::min<size_t>(need[ord[0]], 150 * 1024);
        ck.classes.push_back({0, 0});
        for (size_t r = 0; r < nq; r++) {
            SqAlgoChunk::Class &c = ck.classes.back();
            if (c.count >= 4 && need[ord[r]] * 2 <= cur && (int)ck.classes.size() < max_classes) {
                ck.classes.push_back({(int)r, 1});
                cur = need[ord[r]];
            } else c.count++;
        }
        if (b->sw.mwm_dump && algo == SQ_ALGO_E) {
            fprintf(stderr, "[mwm] %zu graphs, classes:", nq);
            for (auto &c : ck.classes) fprintf(stderr, " [%d +%d: %zu B]", c.start, c.count, need[ord[c.start]]);
            fprintf(stderr, "\n[mwm] (n m bytes):");
            for (size_t r = 0; r < nq; r++) fprintf(stderr, " %d %d %zu;", ck.sorted[r].n, ck.sorted[r].nedges, need[ord[r]]);
            fprintf(stderr, "\n");
        }
        memcpy(ck.p_jobs, ck.sorted.data(), nq * sizeof(SqMatchJob));
    }
    const double tb2 = sq_now();
    if (dev_sizes) {                                      // (the edges are written on the device: sq_algo_edges_kernel)
        if (b->sw.timing) fprintf(stderr, "[sq_algos] layout algo %d: sizes %.3f ms, tables + classes %.3f ms\n", algo, (tb1 - tb0) * 1e3, (tb2 - tb1) * 1e3);
        return 0;
    }
    // pass 2 (pool): the edges, written straight into the pinned buffer
    ck.vid2pos.resize(mj.size());
    if (b->sw.mwm_posthoc) ck.seen_hash.assign(mj.size(), 0);
    sq_pool(b)->parallel_for((int)mj.size(), [&](int q) {
        CpuScope cpu_(2);
        const SqJob &J = b->jobs[jobs[k0 + q]];
        const std::vector<HStem> &st_ = stems[k0 + q];
        SqMatchEdge *e = ck.p_edges + mj[q].edge_off;
        PowCache pow17;
        if (algo == SQ_ALGO_E) {
            static thread_local std::vector<int> pos2id;
            pos2id.assign((size_t)J.n, -1);
            std::vector<int> &ids = ck.vid2pos[q];
            ids.reserve((size_t)mj[q].n);
            for (const HStem &s : st_) {
                const double wt = pow17(s.bps);                      // SQRNalgos.py:101
                for (int t = 0; t < s.len; t++) {
                    const int v = s.i + t, w = s.j - t;
                    if (pos2id[v] < 0) { pos2id[v] = (int)ids.size(); ids.push_back(v); }   // node order = first appearance
                    if (pos2id[w] < 0) { pos2id[w] = (int)ids.size(); ids.push_back(w); }
                    *e++ = SqMatchEdge{pos2id[v], pos2id[w], wt};
                }
            }
        } else {
            for (const HStem &s : st_) {
                const double wt = algo == SQ_ALGO_H ? pow17(s.bps) : s.bps;      // SQRNalgos.py:122 / :49
                for (int t = 0; t < s.len; t++) *e++ = SqMatchEdge{s.i + t, s.j - t, wt};
            }
        }
    });
    if (b->sw.timing) fprintf(stderr, "[sq_algos] build algo %d: sizes %.3f ms, layout+pinned alloc %.3f ms, edges %.3f ms\n", algo,
                                     (tb1 - tb0) * 1e3, (tb2 - tb1) * 1e3, (sq_now() - tb2) * 1e3);
    return 0;
}

// Device part: kernel of a built chunk into [region, region + ck.bytes) on stream st.  The job table and the edges
// are read by the kernels straight from the pinned staging buffer (each kernel reads them once, at its start).
// Nothing is waited for.
static int algo_launch(sq_batch *b, SqAlgoChunk &ck, char *region, hipStream_t st, hipStream_t st2 = nullptr)
{
    ck.st = st;
    const std::vector<SqMatchJob> &mj = ck.mj;
    const int algo = ck.algo;
    if (mj.empty()) return 0;
    // carve: [jobs][edges][out ints][counts][scratch]
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t rr = o; o = (o + bytes + 255) & ~(size_t)255; return rr; };
    take(mj.size() * sizeof(SqMatchJob));                       // (the job table's place in the region; the kernel reads ck.p_jobs)
    const size_t o_edges = take(ck.nedges * sizeof(SqMatchEdge) + 16);
    const size_t o_out = take(ck.outints * 4 + 16), o_cnt = take(mj.size() * 4 + 16), o_scr = take(0);
    const SqMatchJob *d_jobs = ck.p_jobs;
    const SqMatchEdge *d_edges = ck.p_edges;
    (void)o_out; (void)o_cnt;                       // results go to the pinned buffer (algo_build), not to the region
    char *d_scr = region + o_scr;
    const int nj = (int)mj.size();
    hipEvent_t pe0;
    const int pslot = algo == SQ_ALGO_E ? 4 : algo == SQ_ALGO_H ? 5 : 6;
    sq_prof_begin(b, pslot, st, &pe0);
    {
        // `st2` (Edmonds in a fold): the stream of the short kernels takes the classes after the first; st then waits
        // for it, so "st is idle" still means "the chunk is done".  Without st2 the classes follow each other on st.
        hipStream_t other = algo == SQ_ALGO_E ? st2 : nullptr;
        for (size_t cidx = 0; cidx < ck.classes.size(); cidx++) {
            const SqAlgoChunk::Class &c = ck.classes[cidx];
            hipStream_t cs = (cidx > 0 && other) ? other : st;
            const int rl = sq_launch_matching(algo, ck.sorted.data() + c.start, c.count, d_jobs + c.start, d_edges, ck.nedges,
                                              (SqMatchEdge *)(region + o_edges), d_scr, ck.d_out, ck.d_cnt, b->ctx.codes,
                                              ck.job_flags ? ck.job_flags + c.start : nullptr, ck.flag_val, cs,
                                              ck.p_jobs + c.start, ck.bin_head ? ck.bin_head + c.start : nullptr, b->inflight);
            if (rl) return sq_check((hipError_t)rl, "matching kernel launch");
        }
        if (other && ck.classes.size() > 1) {
            if (!b->class_ev) HIPCK(sq_event_get(b->device, &b->class_ev));
            HIPCK(hipEventRecord(b->class_ev, other));
            HIPCK(hipStreamWaitEvent(st, b->class_ev, 0));
        }
    }
    (void)nj;
    sq_prof_end(b, pslot, st, pe0);
    hipLaunchKernelGGL(sq_flag_kernel, dim3(1), dim3(1), 0, st, ck.flag, ck.flag_val);   // "results are in host memory"
    HIPCK(hipGetLastError());
    return 0;
}

// Wait for a staged chunk, read its result and apply the reference's stem filters (:570-595).
// on_job (optional, Edmonds with per-job flags): jobs are collected one by one as the kernel finishes them, smallest
// graphs first, and on_job(k) is called (from a pool worker) as soon as out[k] is final.
static int algo_collect(sq_batch *b, const std::vector<int> &jobs, const std::vector<std::vector<HStem>> &stems,
                        SqAlgoChunk &ck, int levellimit_opt, std::vector<std::vector<HStem>> &out,
                        const std::function<void(size_t)> *on_job = nullptr)
{
    if (ck.mj.empty()) return 0;
    const std::vector<SqMatchJob> &mj = ck.mj;
    const int algo = ck.algo;
    const double tw0 = sq_now();
    const bool streaming = on_job && ck.job_flags;
    if (!streaming) {   // spin on the completion word in pinned memory (no driver round trip, no staged copy)
        const int wr = sq_wait_word(b, ck.flag, ck.flag_val, ck.st, "matching kernel");
        if (wr) return wr;
    }
    const int32_t *h_out_p = ck.d_out, *h_cnt_p = ck.d_cnt;
    const double tw1 = sq_now();
    struct Rep { int algo; double t0, t1; bool on; ~Rep() { if (on) fprintf(stderr, "[sq_algos] algo %d: wait %.3f ms, host filters %.3f ms\n", algo, (t1 - t0) * 1e3, (sq_now() - t1) * 1e3); } } rep{algo, tw0, tw1, b->sw.timing};
    // jobs whose score matrix carries a bpp term / multiplier: RunAlgo's stem filters re-sum cells of THAT matrix
    std::vector<std::vector<double>> dense(mj.size());
    for (size_t q = 0; q < mj.size(); q++) {
        const SqJob &J = b->jobs[jobs[ck.k0 + q]];
        if (J.mat64_off < 0 && !J.mulsh) continue;
        dense[q].resize((size_t)J.n * J.n);
        if (J.mulsh) {
            // an alignment's row (weights read through the gap map from the shared matrix, no slice of its own): its slice for
            // the host's filters, gathered into a buffer of this call
            double *tmp = nullptr; int32_t *d_j = nullptr;
            const int32_t jid = jobs[ck.k0 + q];
            HIPCK(hipMalloc((void **)&tmp, dense[q].size() * 8 + 256));
            d_j = (int32_t *)((char *)tmp + dense[q].size() * 8);
            HIPCK(hipMemcpy(d_j, &jid, 4, hipMemcpyHostToDevice));
            sq_launch_gather_mul(b->ctx, d_j, 1, J.n, b->stream, tmp);
            const hipError_t ge = hipStreamSynchronize(b->stream);
            if (ge == hipSuccess) hipMemcpy(dense[q].data(), tmp, dense[q].size() * 8, hipMemcpyDeviceToHost);
            hipFree(tmp);
            HIPCK(ge);
            continue;
        }
        HIPCK(hipMemcpy(dense[q].data(), b->ctx.mat64 + J.mat64_off, dense[q].size() * 8, hipMemcpyDeviceToHost));
    }
    std::atomic<int> bad{0}, stream_err{0};
    std::vector<int> order(mj.size());
    for (size_t q = 0; q < mj.size(); q++) order[q] = (int)q;
    if (streaming) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return mj[x].nedges < mj[y].nedges; });
    // body of one job: read its result from pinned memory, RunAlgo's filters, optional hook
    auto finish_job = [&](size_t q) {
        CpuScope cpu_(1);
        const size_t k = ck.k0 + q;
        const SqJob &J = b->jobs[jobs[k]];
        const int levellimit = levellimit_opt >= 0 ? levellimit_opt : 3 - (J.n > 500 ? 1 : 0);   // :1043-1044
        std::vector<BP> pairs;
        if (algo == SQ_ALGO_E) {
            const int32_t *mate = h_out_p + mj[q].out_off;
            if (mj[q].n > 0 && mate[0] == -2) { bad = 1; return; }
            {   // measurement: the job with the most scan passes is the kernel's critical path
                const int64_t np = mate[2 * mj[q].n], ne = mate[2 * mj[q].n + 1];
                std::lock_guard<std::mutex> lk(b->mwm_mu);
                b->mwm_stats[0]++; b->mwm_stats[1] += np;
                if (np > b->mwm_stats[2]) { b->mwm_stats[2] = np; b->mwm_stats[3] = ne; b->mwm_stats[4] = mj[q].n; b->mwm_stats[5] = mj[q].nedges; }
            }
            static const bool posthoc = getenv("SQ_MWM_POSTHOC") != nullptr;
            if (posthoc) {                                     // debug: what this read saw, to be compared after the fold
                uint64_t h = 1469598103934665603ull;
                for (int v = 0; v < 2 * mj[q].n + 2; v++) { h ^= (uint32_t)mate[v]; h *= 1099511628211ull; }
                if (ck.seen_hash.size() == mj.size()) ck.seen_hash[q] = h;
            }
            static const bool verify = getenv("SQ_MWM_VERIFY") != nullptr;
            if (verify && mj[q].n > 0) {
                // debug: the same algorithm object run on the host over the job's edges must give the kernel's mates
                const SqMatchEdge *he = ck.p_edges + mj[q].edge_off;
                std::vector<char> scr(SqBlossom::scratch_bytes(mj[q].n, mj[q].nedges) + 64);
                SqBlossom hb;
                hb.init(mj[q].n, mj[q].nedges, he, scr.data());
                hb.run();
                int diff = 0;
                for (int v = 0; v < mj[q].n; v++) diff += hb.mate[v] != mate[v];
                if (diff || hb.error) {
                    const int sp = ck.sorted.empty() ? (int)q : ck.sorted_pos[q];
                    int cls = 0;
                    for (size_t ci = 0; ci < ck.classes.size(); ci++) if (sp >= ck.classes[ci].start) cls = (int)ci;
                    fprintf(stderr, "[mwm verify] job %zu (row %d, class %d of %zu): n %d m %d: %d of the kernel's mates differ from the host run "
                            "(host error %d; kernel passes %d events %d, host passes %d events %d)\n", q, sp, cls, ck.classes.size(), mj[q].n,
                            mj[q].nedges, diff, hb.error, mate[2 * mj[q].n], mate[2 * mj[q].n + 1], hb.stat_pass, hb.stat_event);
                }
            }
            for (int v = 0; v < mj[q].n; v++)
                if (mate[v] > v) pairs.push_back(BP(ck.vid2pos[q][v], ck.vid2pos[q][mate[v]]));
        } else if (algo == SQ_ALGO_N) {
            const int32_t *pp = h_out_p + 2 * mj[q].out_off;
            for (int t = 0; t < h_cnt_p[q]; t++) pairs.push_back(BP(pp[2 * t], pp[2 * t + 1]));
        } else {
            const int32_t *sol = h_out_p + mj[q].out_off;
            const uint8_t *codes = b->codes.data() + J.pos_off;
            // cells with mat[v,w] != 0 (SQRNalgos.py:119-123) as a bit matrix (set, queried, cleared cell by cell: no sort,
            // no O(N^2) clearing); sequences too long for that keep a sorted list
            static thread_local std::vector<int64_t> cells;
            static thread_local std::vector<uint64_t> cellbits;
            const size_t nn = (size_t)J.n * J.n;
            const bool use_bits = nn <= ((size_t)1 << 24);
            if (use_bits && cellbits.size() < (nn + 63) / 64) cellbits.resize((nn + 63) / 64, 0);
            cells.clear();
            PowCache pow17;
            for (const HStem &s : stems[k]) {
                if (-pow17(s.bps) == 0) continue;
                for (int t = 0; t < s.len; t++) cells.push_back((int64_t)(s.i + t) * J.n + (s.j - t));
            }
            if (use_bits) for (int64_t cidx : cells) cellbits[(size_t)cidx >> 6] |= 1ull << (cidx & 63);
            else std::sort(cells.begin(), cells.end());
            for (int kk = 0; kk < J.n; kk++) {                        // SQRNalgos.py:130-133
                const int sk = sol[kk];
                if (sk < 0 || !(kk < sk)) continue;
                bool far = kk < sk - 3;
                if (!far) for (int x = kk + 1; x < sk; x++) if (codes[x] == SQ_CODE_SEP1 || codes[x] == SQ_CODE_SEP2) { far = true; break; }
                if (!far) continue;
                if (sol[sk] != kk) continue;
                const int64_t cidx = (int64_t)kk * J.n + sk;
                if (use_bits ? !((cellbits[(size_t)cidx >> 6] >> (cidx & 63)) & 1ull) : !std::binary_search(cells.begin(), cells.end(), cidx)) continue;
                pairs.push_back(BP(kk, sk));
            }
            if (use_bits) for (int64_t cidx : cells) cellbits[(size_t)cidx >> 6] = 0;
        }
        { CpuScope cpu_f(3); filter_stemset(b, J, pairs, levellimit, out[k], dense[q].empty() ? nullptr : dense[q].data()); }
        if (streaming) { CpuScope cpu_h(5); (*on_job)(k); }
    };
    if (!streaming) {
        sq_pool(b)->parallel_for((int)mj.size(), [&](int qi) { finish_job((size_t)order[qi]); });
    } else {
        // Edmonds job by job.  ONE poller (this thread) sweeps the per-job completion words and hands every group of
        // newly finished jobs to the worker pool -- workers never spin: with several batches in flight, 32 spinning
        // workers per batch would take every host core for the whole length of the blossom kernel.
        std::vector<int> pending(order), ready;
        uint64_t idle = 0;
        while (!pending.empty() && !bad) {
            ready.clear();
            size_t w = 0;
            for (int q : pending) {
                if (*(volatile uint32_t *)(ck.job_flags + ck.sorted_pos[q]) == ck.flag_val) ready.push_back(q);
                else pending[w++] = q;
            }
            pending.resize(w);
            if (!ready.empty()) {
                std::atomic_thread_fence(std::memory_order_acquire);
                idle = 0;
                if (ready.size() <= 2) for (int q : ready) finish_job((size_t)q);
                else sq_pool(b)->parallel_for((int)ready.size(), [&](int t) { finish_job((size_t)ready[t]); }, ready.size() >= 128 ? 1 : 0);
                continue;
            }
            if ((++idle & 0x3FFF) == 0) {
                // a faulted kernel / aborted queue never writes its flags: poll the stream now and then
                const hipError_t qe = hipStreamQuery(ck.st);
                if (qe != hipErrorNotReady && qe != hipSuccess) { stream_err = (int)qe; bad = 3; break; }
                if (qe == hipSuccess || *(volatile uint32_t *)ck.flag == ck.flag_val) {
                    bool missing = false;                    // kernel over: every flag must be there now
                    for (int q : pending) missing |= *(volatile uint32_t *)(ck.job_flags + ck.sorted_pos[q]) != ck.flag_val;
                    if (missing) { bad = 2; break; }
                }
            }
            if (sq_relaxed_waits(b)) sq_wait_step(1 << 20, true);    // (sleeps ~10 us)
            else for (int t = 0; t < 32; t++) sq_wait_step(0, false);
        }
    }
    if (b->sw.mwm_posthoc && algo == SQ_ALGO_E && ck.seen_hash.size() == mj.size()) {
        hipStreamSynchronize(ck.st);
        for (size_t q = 0; q < mj.size(); q++) {
            const int32_t *mate = h_out_p + mj[q].out_off;
            uint64_t h = 1469598103934665603ull;
            for (int v = 0; v < 2 * mj[q].n + 2; v++) { h ^= (uint32_t)mate[v]; h *= 1099511628211ull; }
            if (h != ck.seen_hash[q])
                fprintf(stderr, "[mwm posthoc] job %zu (n %d m %d): the result read when its flag appeared differs from the result after the kernel\n",
                        q, mj[q].n, mj[q].nedges);
        }
    }
    if (bad == 3) return sq_check((hipError_t)stream_err.load(), "matching kernel");
    if (bad == 2) { sq_set_error("matching kernel did not publish a job"); return 2; }
    if (bad) { sq_set_capacity_error(SQ_CAP_FIXED, "blossom capacity exceeded"); return -3; }
    if (!streaming && on_job) for (size_t q = 0; q < mj.size(); q++) (*on_job)(ck.k0 + q);
    return 0;
}

static int algo_annotate(sq_batch *b, const std::vector<int> &jobs, std::vector<std::vector<HStem>> &stems)
{
    // (jobs with caller matrices, has_ext == 1: the stem filters of RunAlgo re-sum the cells of the caller's score
    // matrix, which algo_collect reads back from the dense arena like a weighted matrix)
    // AnnotateStems(bool, score, rbps, [], minlen, minbpscore)  (:553)
    std::vector<HStruct> hs(jobs.size());
    std::vector<SView> views(jobs.size());
    for (size_t k = 0; k < jobs.size(); k++) { hs[k].job = jobs[k]; views[k] = SView{jobs[k], 1.0, &hs[k]}; }
    return sq_run_round(b, views, 1, stems);
}

int sq_run_algo(sq_batch *b, const std::vector<int> &jobs, int algo, int levellimit_opt,
                std::vector<std::vector<HStem>> &out)
{
    out.assign(jobs.size(), {});
    if (jobs.empty()) return 0;
    std::vector<std::vector<HStem>> stems;
    int r = algo_annotate(b, jobs, stems);
    if (r) return r;
    // matching on the device, in chunks that fit the (idle) candidate arena
    const size_t arena = (size_t)(b->cand_records - b->cand_reserved) * sizeof(SqCand);
    size_t k0 = 0;
    while (k0 < jobs.size()) {
        SqAlgoChunk ck;
        r = algo_build(b, jobs, stems, k0, algo, arena, 3, ck);
        if (r) return r;
        if (ck.k1 == k0) { sq_set_error("sequence too long for the matching scratch"); return -3; }
        r = algo_launch(b, ck, (char *)b->scan.cands, b->stream);
        if (r) return r;
        r = algo_collect(b, jobs, stems, ck, levellimit_opt, out);
        if (r) return r;
        k0 = ck.k1;
    }
    return 0;
}

// ---- asynchronous form used by sq_fold: E, H and N run on side streams, in scratch reserved at the end of
// the candidate arena, while the greedy rounds proceed on the batch stream ------------------------------
struct SqAlgoAsync {
    struct Item { int algo; std::vector<int> jobs; std::vector<std::vector<HStem>> stems; SqAlgoChunk ck; bool staged = false; };
    std::vector<Item> items;
    bool dev = false;                             // the whole of RunAlgo runs on the device (sq_algos_dev.hip)
    SqAlgoSize *h_sizes = nullptr;                // pinned: per job of all items, in item order
};
bool sq_algos_on_device(const SqAlgoAsync *pa) { return pa && pa->dev; }

// ---- RunAlgo on the device (sq_algos_dev.hip): one AnnotateStems round for the jobs of all three algorithms with the
// stems left in the arena, the sizes to the host (layout of the scratch, LDS bins of the blossom kernel), the edge lists
// written by a kernel, the matching kernels on the side streams as before, and behind each of them the kernel that applies
// RunAlgo's filters and appends the job's stemset to the device log of final structures.  The host neither sees stems nor
// matchings.  Returns 0: staged, 1: the batch does not qualify (nothing happened that the host-driven form cannot repeat).
static int algos_begin_dev(sq_batch *b, SqAlgoAsync *pa, int levellimit_opt)
{
    const bool off = b->sw.no_device_algos;                             // (per fold: tests compare both forms in one process)
    if (off || pa->items.empty()) return 1;
    std::vector<int> all;
    std::vector<uint8_t> need_raw;
    int maxn = 0, tmax = 1;
    size_t raw_cap = 0;
    for (auto &it : pa->items) {
        for (int j : it.jobs) {
            const SqJob &J = b->jobs[j];
            maxn = std::max(maxn, J.n);
            uint8_t nr = 0;
            if (it.algo != SQ_ALGO_N) {
                // Edmonds / Hungarian weigh an edge with stemscore ** 1.7 from the host libm: on the device that is a table
                // lookup when the score is k 2^-q exactly (dyadic weights, no reactivity factors); the other jobs -- decided
                // per JOB -- leave their stem scores in pinned memory and the host raises them in bulk (SqAlgoRaw)
                if (J.mat64_off >= 0 || J.mulsh) return 1;
                if (!J.default_reacts || b->psets_dev[J.pset].pow_len <= 0) { nr = 1; raw_cap += (size_t)4 * J.n + 64; }
            }
            all.push_back(j);
            need_raw.push_back(nr);
        }
    }
    static const bool no_raw = getenv("SQ_NO_ALGO_RAW") != nullptr;      // (measurement: such batches take the host-driven form)
    if (raw_cap && no_raw) return 1;
    if (maxn > SQ_ALGO_MAXN || maxn < 1) return 1;
    tmax = std::max(b->chain_tmax, 1);
    const size_t fin_lds = sq_algo_finish_lds(maxn, tmax);
    if (fin_lds > 150 * 1024) return 1;
    const int S = (int)all.size();
    // pinned: the sizes of all jobs (slot 3 of the staging buffers is free while a fold runs)
    // (and, behind them, the per-job records the edges kernel reads: one allocation, the buffer must not move in between)
    const size_t o_aj = (sizeof(SqAlgoSize) * (size_t)S + 255) & ~(size_t)255;
    const size_t o_need = (o_aj + sizeof(SqAlgoJob) * (size_t)S + 255) & ~(size_t)255;
    const size_t o_raw = (o_need + (size_t)S + 255) & ~(size_t)255;
    char *pin = stage_buffer(b, 3, o_raw + 8 * raw_cap + 256);
    if (!pin) return 2;
    pa->h_sizes = (SqAlgoSize *)pin;
    SqAlgoRaw raw{nullptr, nullptr, nullptr, 0};
    if (raw_cap) {
        memcpy(pin + o_need, need_raw.data(), (size_t)S);
        raw.need = (const uint8_t *)(pin + o_need); raw.vals = (double *)(pin + o_raw);
        raw.ctr = b->d_fin_ctr + 8;                         // (a word of the fold's counter block: zero since sq_fold_begin_kernel)
        raw.cap = (uint32_t)std::min<size_t>(raw_cap, 0x7FFFFFF0u);
    }
    int64_t cands_used = 0;
    const double tt0 = sq_now();
    int r = sq_round_annotate_dev(b, all, pa->h_sizes, &cands_used, raw);
    if (r) return r;                                       // (1: does not fit one round)
    const double tt1 = sq_now();
    if (raw_cap) {
        // stemscore ** 1.7 with the host's libm, in place (the reference: Python's float ** on every stem, SQRNalgos.py:101,122)
        CpuScope cpu_(2);
        std::vector<std::pair<int, int>> spans;
        size_t total = 0;
        for (int s2 = 0; s2 < S; s2++)
            if (need_raw[s2]) {
                const SqAlgoSize &z = pa->h_sizes[s2];
                if (z.raw_base < 0) return 1;               // the list had no room: the host-driven form takes the batch
                spans.push_back({z.raw_base, z.nok}); total += (size_t)z.nok;
            }
        auto one = [&](int k) { double *v = raw.vals + spans[k].first; for (int q = 0; q < spans[k].second; q++) v[q] = pow(v[q], 1.7); };
        if (total >= 32768) sq_pool(b)->parallel_for((int)spans.size(), one);
        else for (int k = 0; k < (int)spans.size(); k++) one(k);
    }
    // ---- layout of every item: job table, scratch, LDS classes (algo_build without stems) ----
    const int64_t half = b->cand_records / 2;
    size_t base = 0;
    std::vector<char *> regions(pa->items.size(), nullptr);
    int sidx = 0;
    b->algo_used = 0;
    for (size_t q = 0; q < pa->items.size(); q++) {
        auto &it = pa->items[q];
        // algo_build indexes dev_sizes by k0 + q of the item's own list
        const SqAlgoSize *sz = pa->h_sizes + base;
        char *region = nullptr;
        if (it.algo != SQ_ALGO_E && b->algo_bytes > b->algo_used + 65536) {
            const size_t room = b->algo_bytes - b->algo_used;
            r = algo_build(b, it.jobs, it.stems, 0, it.algo, room, sidx, it.ck, sz);
            if (r) return r;
            if (it.ck.k1 == it.jobs.size() && it.ck.bytes + 256 <= room) {
                region = b->algo_scratch + b->algo_used;
                b->algo_used += (it.ck.bytes + 256 + 255) & ~(size_t)255;
            } else it.ck = SqAlgoChunk();
        }
        if (!region) {
            const int64_t free_rec = std::min<int64_t>(half - b->cand_reserved, b->cand_records - b->cand_reserved - cands_used);
            if (free_rec <= 0) { b->cand_reserved = 0; return 1; }
            r = algo_build(b, it.jobs, it.stems, 0, it.algo, (size_t)free_rec * sizeof(SqCand), sidx, it.ck, sz);
            if (r) return r;
            if (it.ck.k1 != it.jobs.size()) { b->cand_reserved = 0; return 1; }      // does not fit as one chunk: host-driven form
            const int64_t used_rec = (int64_t)((it.ck.bytes + 256 + sizeof(SqCand) - 1) / sizeof(SqCand));
            b->cand_reserved += used_rec;
            region = (char *)(b->scan.cands + (b->cand_records - b->cand_reserved));
            region = (char *)(((uintptr_t)region + 255) & ~(uintptr_t)255);
        }
        regions[q] = region;
        base += it.jobs.size();
        sidx++;
    }
    // ---- per-job records of the edges / finish kernels ----
    // region of a chunk: [jobs][edges][out ints][counts][vid2pos][stats][finish records][scratch]
    struct Carve { size_t o_edges, o_out, o_cnt, o_vid, o_stat, o_aj, o_scr; };
    std::vector<Carve> cv(pa->items.size());
    // the edges kernel takes one record per structure of the round: all items' records, contiguous, in round order
    SqAlgoJob *round_aj = (SqAlgoJob *)(pin + o_aj);
    base = 0;
    for (size_t q = 0; q < pa->items.size(); q++) {
        auto &it = pa->items[q];
        const std::vector<SqMatchJob> &mj = it.ck.mj;
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t rr = o; o = (o + bytes + 255) & ~(size_t)255; return rr; };
        take(mj.size() * sizeof(SqMatchJob));
        cv[q].o_edges = take(it.ck.nedges * sizeof(SqMatchEdge) + 16);
        cv[q].o_out = take(it.ck.outints * 4 + 16); cv[q].o_cnt = take(mj.size() * 4 + 16);
        cv[q].o_vid = take(it.ck.vids * 4 + 16); cv[q].o_stat = take(sizeof(SqAlgoStat)); cv[q].o_aj = take(mj.size() * sizeof(SqAlgoJob) + 16); cv[q].o_scr = take(0);
        size_t vid = 0;
        for (size_t k = 0; k < mj.size(); k++) {
            SqAlgoJob aj;
            aj.job = it.jobs[k]; aj.algo = it.algo;
            aj.edges = (SqMatchEdge *)(regions[q] + cv[q].o_edges) + mj[k].edge_off;
            aj.vid2pos = (int32_t *)(regions[q] + cv[q].o_vid) + vid;
            aj.raw = need_raw[base + k] ? raw.vals + pa->h_sizes[base + k].raw_base : nullptr;
            if (it.algo == SQ_ALGO_E) vid += (size_t)mj[k].n;
            it.ck.p_aj[k] = aj;
            round_aj[base + k] = aj;
        }
        base += mj.size();
    }
    const double tt2 = sq_now();
    struct Rep { double t0, t1, t2; bool on; ~Rep() { if (on) fprintf(stderr, "[sq_algos] device RunAlgo: annotate round + wait for the sizes %.3f ms, layout %.3f ms, launches %.3f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, (sq_now() - t2) * 1e3); } } rep_dev{tt0, tt1, tt2, b->sw.timing};
    // ---- launches: edges on the batch stream (it owns the arena), then every item on its side stream ----
    hipStream_t st = b->stream;
    {
        const int maxn_lds = (std::min(b->maxn, SQ_ALGO_MAXN) + 7) & ~7;   // (the device RunAlgo only takes batches up to SQ_ALGO_MAXN nt)
        SqAlgoStatPtrs zs{{nullptr, nullptr, nullptr}};
        for (size_t q = 0; q < pa->items.size() && q < 3; q++) zs.p[q] = (SqAlgoStat *)(regions[q] + cv[q].o_stat);
        // the stem lists in LDS when the longest of the launch fits
        int nokcap = 0;
        for (int s2 = 0; s2 < S; s2++) nokcap = std::max(nokcap, (int)pa->h_sizes[s2].nok);
        nokcap = (nokcap + 63) & ~63;
        // (up to 6,144 stems -- 130 KB, a block per CU: the other form ranks every stem against ALL the keys in global memory,
        // O(stems^2): 15-25 ms for the 1,500 jobs of 500 records of 500 nt, in front of the greedy loop on the batch's stream)
        if (nokcap > 6144 || sq_algo_edges_lds(maxn_lds, nokcap) > 150 * 1024 || b->sw.no_edges_lds) nokcap = 0;
        if (sq_algo_edges_lds(maxn_lds, nokcap) > 48 * 1024) sq_max_dynamic_lds((const void *)sq_algo_edges_kernel, 160 * 1024);
        // (one wave per job on a crowded chip: the work of a job is a few hundred stems behind a handful of trips to L2, and
        // three waves that mostly wait held three wave slots -- sizes + edges were 7 % of a crowded step's wave cycles)
        const bool crowded_k = b->inflight > 1 || b->njobs >= 4096;
        // (the jobs' records to the device first when the scratch has room behind the items' regions: from the pinned table
        // every block of the kernel began with a read over PCIe)
        const SqAlgoJob *edges_aj = round_aj;
        {
            const size_t at = (b->algo_used + 255) & ~(size_t)255, need = (size_t)S * sizeof(SqAlgoJob);
            if (b->algo_scratch && at + need <= b->algo_bytes && !getenv("SQ_NO_STATE_COPY")) {
                HIPCK(hipMemcpyAsync(b->algo_scratch + at, round_aj, need, hipMemcpyHostToDevice, st));
                edges_aj = (const SqAlgoJob *)(b->algo_scratch + at);
            }
        }
        hipLaunchKernelGGL(sq_algo_edges_kernel, dim3(S), dim3(crowded_k ? 64 : 256), sq_algo_edges_lds(maxn_lds, nokcap), st, b->ctx, b->lane_full.d_structs,
                           [&] { SqScanArgs a = b->scan; a.ctr = b->lane_full.d_ctr; return a; }(), edges_aj, maxn_lds, zs, nokcap);
    }
    HIPCK(hipGetLastError());
    if (!b->class_ev) HIPCK(sq_event_get(b->device, &b->class_ev));
    if (!b->edges_ev) HIPCK(sq_event_get(b->device, &b->edges_ev));
    HIPCK(hipEventRecord(b->edges_ev, st));
    if (fin_lds > 64 * 1024) sq_max_dynamic_lds((const void *)sq_algo_finish_kernel, 160 * 1024);
    sidx = 0;
    for (size_t q = 0; q < pa->items.size(); q++) {
        auto &it = pa->items[q];
        SqAlgoChunk &ck = it.ck;
        const int ss = side_of(b, sidx);
        if (!b->side[ss]) { if (sq_check(sq_stream_get(b->device, &b->side[ss]), "hipStreamCreate")) return 2; }
        hipStream_t cs = b->side[ss];
        HIPCK(hipStreamWaitEvent(cs, b->edges_ev, 0));
        ck.st = cs;
        char *region = regions[q];
        const SqMatchEdge *d_edges = (const SqMatchEdge *)(region + cv[q].o_edges);
        int32_t *d_out = (int32_t *)(region + cv[q].o_out), *d_cnt = (int32_t *)(region + cv[q].o_cnt);
        SqAlgoStat *d_stat = (SqAlgoStat *)(region + cv[q].o_stat);
        if (q >= 3) HIPCK(hipMemsetAsync(d_stat, 0, sizeof(SqAlgoStat), cs));   // (the first three: zeroed by the edges kernel)
        const int nj = (int)ck.mj.size();
        // The finish kernel's two records per job go to the device first: from the pinned table every one of its waves began
        // with two reads over PCIe (and the publish kernel's wave walks the table): microseconds in a wave that works for a few.
        SqMatchJob *const dev_mj = (SqMatchJob *)region;
        SqAlgoJob *const dev_aj = (SqAlgoJob *)(region + cv[q].o_aj);
        HIPCK(hipMemcpyAsync(dev_mj, ck.p_mj, (size_t)nj * sizeof(SqMatchJob), hipMemcpyHostToDevice, cs));
        HIPCK(hipMemcpyAsync(dev_aj, ck.p_aj, (size_t)nj * sizeof(SqAlgoJob), hipMemcpyHostToDevice, cs));
        hipEvent_t pe0;
        const int pslot = it.algo == SQ_ALGO_E ? 4 : it.algo == SQ_ALGO_H ? 5 : 6;
        sq_prof_begin(b, pslot, cs, &pe0);
        int rl;
        {
            // the table sorted by LDS need, in size classes.  Edmonds folded alone: its first class (the largest graphs, the
            // critical path) on this stream, the others in front of the short kernels on their stream, joined before the
            // finish kernel
            rl = 0;
            hipStream_t other = nullptr;
            if (it.algo == SQ_ALGO_E && ck.classes.size() > 1 && side_of(b, 3) != ss) {
                const int s2 = side_of(b, 3);
                if (!b->side[s2]) { if (sq_check(sq_stream_get(b->device, &b->side[s2]), "hipStreamCreate")) return 2; }
                other = b->side[s2];
                HIPCK(hipStreamWaitEvent(other, b->edges_ev, 0));
            }
            for (size_t cidx = 0; cidx < ck.classes.size() && !rl; cidx++) {
                const SqAlgoChunk::Class &c2 = ck.classes[cidx];
                rl = sq_launch_matching(it.algo, ck.sorted.data() + c2.start, c2.count, ck.p_jobs + c2.start, d_edges, ck.nedges,
                                        const_cast<SqMatchEdge *>(d_edges), region + cv[q].o_scr, d_out, d_cnt, b->ctx.codes, nullptr, ck.flag_val,
                                        (cidx > 0 && other) ? other : cs,
                                        ck.p_jobs + c2.start, ck.bin_head ? ck.bin_head + c2.start : nullptr, b->inflight);
            }
            if (other && !rl) { HIPCK(hipEventRecord(b->class_ev, other)); HIPCK(hipStreamWaitEvent(cs, b->class_ev, 0)); }
        }
        if (rl) return sq_check((hipError_t)rl, "matching kernel launch");
        sq_prof_end(b, pslot, cs, pe0);
        hipLaunchKernelGGL(sq_algo_finish_kernel, dim3(nj), dim3(64), fin_lds, cs, b->ctx, dev_aj, dev_mj, d_out, d_cnt, levellimit_opt,
                           b->d_fin, b->d_fin_stems, b->d_fin_ctr, b->fin_cap, b->fin_stem_cap, d_stat, tmax);
        hipLaunchKernelGGL(sq_algo_publish_kernel, dim3(1), dim3(64), 0, cs, d_stat, ck.h_stats, dev_mj, d_out, it.algo == SQ_ALGO_E ? 1 : 0,
                           ck.flag, ck.flag_val, nj);
        HIPCK(hipGetLastError());
        it.staged = true;
        sidx++;
    }
    pa->dev = true;
    return 0;
}

// waits for the device-side RunAlgo of a fold (the stemsets are in the device log by then)
static int algos_end_dev(sq_batch *b, SqAlgoAsync *pa)
{
    int r = 0;
    for (size_t q = pa->items.size(); q-- > 0 && !r;) {
        SqAlgoChunk &ck = pa->items[q].ck;
        r = sq_wait_word(b, ck.flag, ck.flag_val, ck.st, "matching kernel");
        if (r) break;
        const SqAlgoStat hs = *ck.h_stats;
        if (hs.bad == 1) { sq_set_capacity_error(SQ_CAP_FIXED, "blossom capacity exceeded"); r = -3; }
        else if (hs.bad) { sq_set_capacity_error(SQ_CAP_FIXED, "stem capacity of RunAlgo's filters exceeded"); r = -3; }
        else if (hs.level_ovf) { sq_set_error("more than 64 pseudoknot levels"); r = -3; }
        if (pa->items[q].algo == SQ_ALGO_E && hs.graphs) {
            std::lock_guard<std::mutex> lk(b->mwm_mu);
            b->mwm_stats[0] += (int64_t)hs.graphs; b->mwm_stats[1] += (int64_t)hs.passes;
            const int64_t np = (int64_t)(hs.max_pass_job >> 32);
            if (np > b->mwm_stats[2]) { b->mwm_stats[2] = np; b->mwm_stats[3] = hs.max_events; b->mwm_stats[4] = hs.max_n; b->mwm_stats[5] = hs.max_m; }
        }
    }
    for (int k = 0; k < 4; k++) if (b->side[k]) hipStreamSynchronize(b->side[k]);
    b->cand_reserved = 0;
    return r;
}

int sq_algos_begin(sq_batch *b, const std::vector<uint32_t> &algos, SqAlgoAsync *&pa, int levellimit_opt, bool want_dev)
{
    pa = new SqAlgoAsync();
    b->algo_used = 0;
    for (int algo : {SQ_ALGO_E, SQ_ALGO_H, SQ_ALGO_N}) {
        SqAlgoAsync::Item it;
        it.algo = algo;
        for (int j = 0; j < b->njobs; j++) if (algos[j] & (uint32_t)algo) it.jobs.push_back(j);
        if (it.jobs.empty()) continue;
        pa->items.push_back(std::move(it));
    }
    if (pa->items.empty()) return 0;
    if (want_dev) {
        const int rd = algos_begin_dev(b, pa, levellimit_opt);
        if (rd == 0) return 0;
        if (rd != 1) return rd;
        for (auto &it : pa->items) { it.ck = SqAlgoChunk(); it.staged = false; }   // the host-driven form, from scratch
        b->algo_used = 0; b->cand_reserved = 0;
    }
    const bool async = !b->sw.algo_sync;
    const int64_t half = b->cand_records / 2;          // at most half of the arena is lent to the matching kernels
    int sidx = 0;
    // Edmonds is the long pole: its AnnotateStems pass and launch go first, alone; the other algorithms share
    // one more pass.  Everything that is staged runs on side streams while the caller proceeds.
    auto stage = [&](SqAlgoAsync::Item &it) -> int {
        const double ts0 = sq_now();
        struct Rep { int algo; double t0; bool on; ~Rep() { if (on) fprintf(stderr, "[sq_algos] stage algo %d: %.3f ms\n", algo, (sq_now() - t0) * 1e3); } } rep{it.algo, ts0, b->sw.timing};
        if (!async || sidx >= 3) return 0;
        char *region = nullptr;
        // Hungarian / Nussinov: the batch's own scratch when the chunk fits it (planned from the lengths)
        if (it.algo != SQ_ALGO_E && b->algo_bytes > b->algo_used + 65536) {
            const size_t room = b->algo_bytes - b->algo_used;
            int rb = algo_build(b, it.jobs, it.stems, 0, it.algo, room, sidx, it.ck);
            if (rb) return rb;
            if (it.ck.k1 == it.jobs.size() && it.ck.bytes + 256 <= room) {
                region = b->algo_scratch + b->algo_used;
                b->algo_used += (it.ck.bytes + 256 + 255) & ~(size_t)255;
            } else it.ck = SqAlgoChunk();
        }
        if (!region) {
            const int64_t free_rec = half - b->cand_reserved;
            if (free_rec <= 0) return 0;
            int rb = algo_build(b, it.jobs, it.stems, 0, it.algo, (size_t)free_rec * sizeof(SqCand), sidx, it.ck);
            if (rb) return rb;
            if (it.ck.k1 != it.jobs.size()) { it.ck = SqAlgoChunk(); return 0; }   // does not fit as one chunk: synchronous later
            const int64_t used_rec = (int64_t)((it.ck.bytes + 256 + sizeof(SqCand) - 1) / sizeof(SqCand));
            b->cand_reserved += used_rec;                   // carved downwards from the end of the arena
            region = (char *)(b->scan.cands + (b->cand_records - b->cand_reserved));
            region = (char *)(((uintptr_t)region + 255) & ~(uintptr_t)255);
        }
        const int ss = side_of(b, sidx);
        if (!b->side[ss]) { if (sq_check(sq_stream_get(b->device, &b->side[ss]), "hipStreamCreate")) return 2; }
        const double tl0 = sq_now();
        hipStream_t st2 = nullptr;
        if (it.algo == SQ_ALGO_E && side_of(b, 3) != ss) {               // the stream of the smaller classes (created here if need be)
            const int s2 = side_of(b, 3);
            if (!b->side[s2]) { if (sq_check(sq_stream_get(b->device, &b->side[s2]), "hipStreamCreate")) return 2; }
            st2 = b->side[s2];
        }
        const int r = algo_launch(b, it.ck, region, b->side[ss], st2);
        if (b->sw.timing) fprintf(stderr, "[sq_algos] launch algo %d: %.3f ms\n", it.algo, (sq_now() - tl0) * 1e3);
        if (r) return r;
        it.staged = true;
        sidx++;
        return 0;
    };
    size_t first_rest = 0;
    const double ta0 = sq_now();
    if (pa->items[0].algo == SQ_ALGO_E) {
        int r = algo_annotate(b, pa->items[0].jobs, pa->items[0].stems);
        if (r) return r;
        if (b->sw.timing) fprintf(stderr, "[sq_algos] annotate E %.3f ms\n", (sq_now() - ta0) * 1e3);
        r = stage(pa->items[0]);
        if (r) return r;
        first_rest = 1;
    }
    std::vector<int> all;
    for (size_t q = first_rest; q < pa->items.size(); q++) all.insert(all.end(), pa->items[q].jobs.begin(), pa->items[q].jobs.end());
    if (!all.empty()) {
        std::vector<std::vector<HStem>> stems;
        int r = algo_annotate(b, all, stems);
        if (r) return r;
        size_t pos = 0;
        for (size_t q = first_rest; q < pa->items.size(); q++) {
            auto &it = pa->items[q];
            it.stems.assign(std::make_move_iterator(stems.begin() + pos), std::make_move_iterator(stems.begin() + pos + it.jobs.size()));
            pos += it.jobs.size();
            r = stage(it);
            if (r) return r;
        }
    }
    if (b->sw.timing) fprintf(stderr, "[sq_algos] annotate + stage %.3f ms\n", (sq_now() - ta0) * 1e3);
    return 0;
}

void sq_algos_abandon(sq_batch *b, SqAlgoAsync *pa)
{
    for (int k = 0; k < 4; k++) if (b->side[k]) hipStreamSynchronize(b->side[k]);
    b->cand_reserved = 0;
    delete pa;
}

int sq_algos_end(sq_batch *b, SqAlgoAsync *pa, int levellimit_opt, std::vector<JobSets> &sets, const SqAlgoEndHooks *hooks)
{
    int r = 0;
    if (pa->dev) { r = algos_end_dev(b, pa); delete pa; return r; }
    // collect staged work first, release the reservation, then run what was not staged
    for (auto &it : pa->items) {
        JobSets js; js.algo = it.algo; js.jobs = it.jobs; js.sets.assign(it.jobs.size(), {});
        sets.push_back(std::move(js));
    }
    bool all_staged = true;
    for (auto &it : pa->items) all_staged &= it.staged;
    const bool stream_e = hooks && all_staged && !pa->items.empty() && pa->items[0].algo == SQ_ALGO_E && pa->items[0].ck.job_flags;
    // last staged first: the short kernels (N, H) are done long before Edmonds, so their host filters run
    // while the blossom kernel is still busy
    for (size_t q = pa->items.size(); q-- > (stream_e ? 1 : 0);) {
        auto &it = pa->items[q];
        if (it.staged && !r) r = algo_collect(b, it.jobs, it.stems, it.ck, levellimit_opt, sets[q].sets);
    }
    if (stream_e && !r) {
        // Edmonds job by job: the caller first takes the H / N stemsets, then every sequence is finished (filters +
        // its ranking tail, in the hook) as soon as its graph is matched -- only the largest graph is waited for
        hooks->after_short(sets);
        auto &it = pa->items[0];
        sets[0].streamed = true;
        const std::function<void(size_t)> cb = [&](size_t k) { hooks->on_e_job(it.jobs[k], sets[0].sets[k]); };
        r = algo_collect(b, it.jobs, it.stems, it.ck, levellimit_opt, sets[0].sets, &cb);
    }
    for (int k = 0; k < 4; k++) if (b->side[k]) hipStreamSynchronize(b->side[k]);
    b->cand_reserved = 0;
    size_t q = 0;
    for (auto &it : pa->items) {
        JobSets &js = sets[q++];
        if (it.staged || r) continue;
        if (it.stems.size() != it.jobs.size()) {          // (sq_algos_begin stopped before this item's AnnotateStems pass)
            r = algo_annotate(b, it.jobs, it.stems);
            if (r) break;
        }
        size_t k0 = 0;
        const size_t arena = (size_t)b->cand_records * sizeof(SqCand);
        while (k0 < it.jobs.size() && !r) {
            SqAlgoChunk ck;
            r = algo_build(b, it.jobs, it.stems, k0, it.algo, arena, 3, ck);
            if (!r && ck.k1 == k0) { sq_set_error("sequence too long for the matching scratch"); r = -3; }
            if (!r) r = algo_launch(b, ck, (char *)b->scan.cands, b->stream);
            if (!r) r = algo_collect(b, it.jobs, it.stems, ck, levellimit_opt, js.sets);
            k0 = ck.k1;
        }
    }
    delete pa;
    return r;
}
extern "C" int sq_run_algos(sq_batch *b, int32_t njob, const int32_t *job_ids, int32_t algo, int32_t levellimit,
                            sq_stem *out, int32_t out_cap, int32_t *out_off)
{
    if (!b || njob < 0 || (algo != SQ_ALGO_E && algo != SQ_ALGO_H && algo != SQ_ALGO_N)) { sq_set_error("bad argument"); return -1; }
    SqSlackGuard slack_guard;
    std::vector<int> jobs(job_ids, job_ids + njob);
    for (int j : jobs) if (j < 0 || j >= b->njobs) { sq_set_error("bad job index"); return -1; }
    std::vector<std::vector<HStem>> res;
    int r = sq_run_algo(b, jobs, algo, levellimit, res);
    if (r) return r;
    int32_t o = 0;
    for (int k = 0; k < njob; k++) {
        out_off[k] = o;
        for (const HStem &t : res[k]) {
            if (o >= out_cap) { sq_set_error("out_cap too small"); return -3; }
            out[o++] = sq_stem{t.i, t.j, t.len, 0, t.bps, t.fin};
        }
    }
    out_off[njob] = o;
    return 0;
}
