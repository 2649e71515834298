// sq_algos.hip -- host driver of RunAlgo (SQRNdbnseq.py:548-595) for the E / H / N algorithms:
// one AnnotateStems pass on the GPU (scan + exact rescoring), the matching / DP on the GPU
// (sq_match.hip), then the reference's stem filters on the host.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>
#include "sq_host.h"
#include "sq_match.h"
#include "sq_blossom.h"

#define HIPCK(x) do { int _r = sq_check((x), #x); if (_r) return _r; } while (0)

typedef std::pair<int, int> BP;

// exact scoremat cell on the host: the same fp64 expression as sq_cell_score (sq_kernels.hip)
static double cell_score_host(const sq_batch *b, const SqJob &J, int i, int j)
{
    const uint8_t *codes = b->codes.data() + J.pos_off;
    const sq_paramset &ps = b->psets[J.pset];
    const double w = ps.bpweight[codes[i] * 32 + codes[j]];
    double rf = 1.0;
    if (!J.default_reacts) {
        const double *r = b->reacts.data() + J.pos_off;
        rf = sqrt((1.0 - (r[i] + r[j]) / 2.0) * 2.0);
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
static void filter_stemset(const sq_batch *b, const SqJob &J, std::vector<BP> pairs, int levellimit,
                           std::vector<HStem> &stemset)
{
    const sq_paramset &ps = b->psets[J.pset];
    for (BP &p : pairs) if (p.first > p.second) std::swap(p.first, p.second);
    std::sort(pairs.begin(), pairs.end());
    std::vector<HStem> stems;
    pairs_to_stems(pairs, stems);
    auto score_of = [&](const HStem &st) {
        double s = 0;                                                    // sum(...) from int 0, left to right
        for (int k = 0; k < st.len; k++) s = s + cell_score_host(b, J, st.i + k, st.j - k);
        return s;
    };
    std::vector<BP> kept;
    for (const HStem &st : stems) {
        const double sc = score_of(st);
        if (sc >= ps.minbpscore && (double)st.len >= ps.minlen)
            for (int k = 0; k < st.len; k++) kept.push_back(BP(st.i + k, st.j - k));
    }
    // DBNToPairs(PairsToDBN(pairs, N, levellimit)) : drop pseudoknot levels above the limit (:581)
    std::sort(kept.begin(), kept.end());
    kept.erase(std::unique(kept.begin(), kept.end()), kept.end());
    std::vector<int> lv;
    sq_pair_levels(kept, lv);
    std::vector<BP> lim;
    for (size_t k = 0; k < kept.size(); k++)
        if ((levellimit < 0 || lv[k] <= levellimit) && lv[k] <= 49) lim.push_back(kept[k]);   // 49 bracket types exist
    sq_pair_levels(lim, lv);                                             // :582 levels of what is left
    pairs_to_stems(lim, stems);
    stemset.clear();
    size_t pos = 0;
    for (const HStem &st : stems) {
        const int level = lv[pos];                                       // level of the stem's first bp
        pos += (size_t)st.len;
        if (level > 1 && st.len < 4) continue;                           // :589 short pseudoknotted stems
        const double sc = score_of(st);
        if (sc >= ps.minbpscore && (double)st.len >= ps.minlen) stemset.push_back(HStem{st.i, st.j, st.len, sc, sc});
    }
}

int sq_run_algo(sq_batch *b, const std::vector<int> &jobs, int algo, int levellimit_opt,
                std::vector<std::vector<HStem>> &out)
{
    out.assign(jobs.size(), {});
    if (jobs.empty()) return 0;
    for (int j : jobs)
        if (b->jobs[j].has_ext) { sq_set_error("E/H/N algorithms need the library's own score matrix"); return -4; }
    // 1. AnnotateStems(bool, score, rbps, [], minlen, minbpscore)  (:553)
    std::vector<HStruct> hs(jobs.size());
    std::vector<SView> views(jobs.size());
    for (size_t k = 0; k < jobs.size(); k++) { hs[k].job = jobs[k]; views[k] = SView{jobs[k], 1.0, &hs[k]}; }
    std::vector<std::vector<HStem>> stems;
    int r = sq_run_round(b, views, 1, stems);
    if (r) return r;
    // 2. matching on the device, in chunks that fit the (idle) candidate arena
    const size_t arena = (size_t)b->cand_records * sizeof(SqCand);
    char *abase = (char *)b->scan.cands;
    size_t k0 = 0;
    while (k0 < jobs.size()) {
        std::vector<SqMatchJob> mj;
        std::vector<SqMatchEdge> me;
        std::vector<std::vector<int>> vid2pos;                           // Edmonds: graph vertex -> position
        size_t scratch = 0, outints = 0, k1 = k0;
        for (; k1 < jobs.size(); k1++) {
            const SqJob &J = b->jobs[jobs[k1]];
            const std::vector<HStem> &st = stems[k1];
            SqMatchJob m;
            m.edge_off = (int64_t)me.size(); m.pos_off = J.pos_off;
            std::vector<int> ids;
            size_t need, nout;
            size_t ncell = 0;
            for (const HStem &s : st) ncell += (size_t)s.len;
            if (algo == SQ_ALGO_E) {
                std::vector<int> pos2id(J.n, -1);
                for (const HStem &s : st) {
                    const double wt = pow(s.bps, 1.7);                   // SQRNalgos.py:101
                    for (int t = 0; t < s.len; t++) {
                        const int v = s.i + t, w = s.j - t;
                        if (pos2id[v] < 0) { pos2id[v] = (int)ids.size(); ids.push_back(v); }   // node order = first appearance
                        if (pos2id[w] < 0) { pos2id[w] = (int)ids.size(); ids.push_back(w); }
                        me.push_back(SqMatchEdge{pos2id[v], pos2id[w], wt});
                    }
                }
                m.n = (int)ids.size(); need = sq_mwm_scratch_bytes(m.n, (int)ncell); nout = (size_t)m.n;
            } else {
                for (const HStem &s : st) {
                    const double wt = algo == SQ_ALGO_H ? pow(s.bps, 1.7) : s.bps;   // SQRNalgos.py:122 / :49
                    for (int t = 0; t < s.len; t++) me.push_back(SqMatchEdge{s.i + t, s.j - t, wt});
                }
                m.n = J.n;
                need = algo == SQ_ALGO_H ? sq_lsap_scratch_bytes(J.n) : sq_nussinov_scratch_bytes(J.n);
                nout = algo == SQ_ALGO_H ? (size_t)J.n : 2 * ((size_t)J.n + 4);
            }
            m.nedges = (int32_t)ncell;
            need = (need + 255) & ~(size_t)255;
            const size_t fixed = (mj.size() + 1) * sizeof(SqMatchJob) + me.size() * sizeof(SqMatchEdge) +
                                 (outints + nout + mj.size() + 1) * 4 + 4096;
            if (!mj.empty() && fixed + scratch + need > arena) { me.resize((size_t)m.edge_off); break; }
            if (fixed + scratch + need > arena) { sq_set_error("sequence too long for the matching scratch"); return -3; }
            m.scratch_off = (int64_t)scratch; scratch += need;
            m.out_off = (int64_t)(algo == SQ_ALGO_N ? outints / 2 : outints); outints += nout;
            mj.push_back(m);
            vid2pos.push_back(std::move(ids));
        }
        // carve: [jobs][edges][out ints][counts][scratch]
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t rr = o; o = (o + bytes + 255) & ~(size_t)255; return rr; };
        const size_t o_jobs = take(mj.size() * sizeof(SqMatchJob)), o_edges = take(me.size() * sizeof(SqMatchEdge) + 16);
        const size_t o_out = take(outints * 4 + 16), o_cnt = take(mj.size() * 4 + 16), o_scr = take(0);
        if (o_scr + scratch > arena) { sq_set_error("matching scratch does not fit"); return -3; }
        SqMatchJob *d_jobs = (SqMatchJob *)(abase + o_jobs);
        SqMatchEdge *d_edges = (SqMatchEdge *)(abase + o_edges);
        int32_t *d_out = (int32_t *)(abase + o_out), *d_cnt = (int32_t *)(abase + o_cnt);
        char *d_scr = abase + o_scr;
        hipStream_t st = b->stream;
        HIPCK(hipMemcpyAsync(d_jobs, mj.data(), mj.size() * sizeof(SqMatchJob), hipMemcpyHostToDevice, st));
        if (!me.empty()) HIPCK(hipMemcpyAsync(d_edges, me.data(), me.size() * sizeof(SqMatchEdge), hipMemcpyHostToDevice, st));
        const int nj = (int)mj.size();
        int maxn = 0, maxm = 0;
        for (const SqMatchJob &m : mj) { maxn = std::max(maxn, m.n); maxm = std::max(maxm, m.nedges); }
        if (algo == SQ_ALGO_H) {
            const int lds = (int)std::min<size_t>((size_t)maxn * 42 + 64, 64 * 1024);
            hipLaunchKernelGGL(sq_lsap_kernel, dim3(nj), dim3(64), lds, st, d_jobs, d_edges, d_scr, d_out, lds);
        }
        else if (algo == SQ_ALGO_N) hipLaunchKernelGGL(sq_nussinov_kernel, dim3(nj), dim3(256), 0, st, d_jobs, d_edges, b->ctx.codes, d_scr, d_out, d_cnt);
        else {
            // dynamic LDS for the blossom state of the largest job (up to 150 KiB of the CU's 160)
            size_t want = SqBlossom::scratch_bytes(maxn, maxm, 1) + (((size_t)maxm * sizeof(SqMatchEdge) + 15) & ~(size_t)15) + 64;
            static bool attr_set = false;
            if (!attr_set) { hipFuncSetAttribute((const void *)sq_mwm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set = true; }
            want = std::min<size_t>(want, 150 * 1024);             // jobs that do not fit run in global memory
            if (getenv("SQ_MWM_NOLDS")) want = 0;
            hipLaunchKernelGGL(sq_mwm_kernel, dim3(nj), dim3(64), want, st, d_jobs, d_edges, d_scr, d_out, (int)want);
        }
        HIPCK(hipGetLastError());
        std::vector<int32_t> h_out(outints + 4), h_cnt(mj.size() + 1);
        HIPCK(hipMemcpyAsync(h_out.data(), d_out, outints * 4, hipMemcpyDeviceToHost, st));
        if (algo == SQ_ALGO_N) HIPCK(hipMemcpyAsync(h_cnt.data(), d_cnt, mj.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        // 3. pairs -> filtered stemset
        for (size_t q = 0; q < mj.size(); q++) {
            const size_t k = k0 + q;
            const SqJob &J = b->jobs[jobs[k]];
            const int levellimit = levellimit_opt >= 0 ? levellimit_opt : 3 - (J.n > 500 ? 1 : 0);   // :1043-1044
            std::vector<BP> pairs;
            if (algo == SQ_ALGO_E) {
                const int32_t *mate = h_out.data() + mj[q].out_off;
                if (mj[q].n > 0 && mate[0] == -2) { sq_set_error("blossom capacity exceeded"); return -3; }
                for (int v = 0; v < mj[q].n; v++)
                    if (mate[v] > v) pairs.push_back(BP(vid2pos[q][v], vid2pos[q][mate[v]]));
            } else if (algo == SQ_ALGO_N) {
                const int32_t *pp = h_out.data() + 2 * mj[q].out_off;
                for (int t = 0; t < h_cnt[q]; t++) pairs.push_back(BP(pp[2 * t], pp[2 * t + 1]));
            } else {
                const int32_t *sol = h_out.data() + mj[q].out_off;
                const uint8_t *codes = b->codes.data() + J.pos_off;
                std::map<BP, double> cells;                               // mat[v,w] (both orientations), SQRNalgos.py:119-123
                for (const HStem &s : stems[k]) {
                    const double wt = -pow(s.bps, 1.7);
                    for (int t = 0; t < s.len; t++) { cells[BP(s.i + t, s.j - t)] = wt; cells[BP(s.j - t, s.i + t)] = wt; }
                }
                for (int kk = 0; kk < J.n; kk++) {                        // SQRNalgos.py:130-133
                    const int sk = sol[kk];
                    if (sk < 0 || !(kk < sk)) continue;
                    bool far = kk < sk - 3;
                    if (!far) for (int x = kk + 1; x < sk; x++) if (codes[x] == SQ_CODE_SEP1 || codes[x] == SQ_CODE_SEP2) { far = true; break; }
                    if (!far) continue;
                    if (sol[sk] != kk) continue;
                    auto it = cells.find(BP(kk, sk));
                    if (it == cells.end() || it->second == 0) continue;
                    pairs.push_back(BP(kk, sk));
                }
            }
            filter_stemset(b, J, pairs, levellimit, out[k]);
        }
        k0 = k1;
    }
    return 0;
}

extern "C" int sq_run_algos(sq_batch *b, int32_t njob, const int32_t *job_ids, int32_t algo, int32_t levellimit,
                            sq_stem *out, int32_t out_cap, int32_t *out_off)
{
    if (!b || njob < 0 || (algo != SQ_ALGO_E && algo != SQ_ALGO_H && algo != SQ_ALGO_N)) { sq_set_error("bad argument"); return -1; }
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
