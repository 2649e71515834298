// sq_graph.hip -- graph-level C entries of the matching step: the drop-ins for the reference's
// SQRNalgos.Edmonds (:96-110), SQRNalgos.Hungarian (:113-135) and SQRNalgos.Nussinov (:44-93) when they are
// called on their own (stems / edge lists in, pairs out) rather than through a fold.  Same kernels as the fold
// path (sq_match.hip): networkx's max_weight_matching and scipy's linear_sum_assignment restated step by step.
// Inputs and results travel through pinned host memory that the kernels read / write in place; the caller's
// device workspace only holds the per-problem scratch of problems that do not fit LDS.
#include <algorithm>
#include <cstring>
#include <unordered_map>
#include <vector>
#include "sq_host.h"
#include "sq_match.h"

#define HIPCK(x) do { int _r = sq_check((x), #x); if (_r) return _r; } while (0)

namespace {
struct Pinned {                          // one pinned block from the process-wide cache, returned on scope exit
    char *p = nullptr;
    ~Pinned() { sq_pinned_put(p); }
    int get(size_t bytes) { return sq_pinned_get((void **)&p, bytes); }
};
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// Graphs as the kernels want them: vertices renumbered in order of first appearance (networkx node order), a repeated
// edge keeps its first position and takes the last weight (Graph.add_weighted_edges_from on an undirected Graph).
struct Graphs {
    std::vector<SqMatchJob> jobs;
    std::vector<SqMatchEdge> edges;
    std::vector<std::vector<int32_t>> labels;   // id -> caller's label
    size_t scratch = 0, outints = 0;
};

int build_graphs(int32_t ngraph, const int64_t *edge_off, const int32_t *eu, const int32_t *ev, const double *ew, Graphs &G)
{
    G.jobs.resize(ngraph); G.labels.resize(ngraph);
    for (int g = 0; g < ngraph; g++) {
        if (edge_off[g + 1] < edge_off[g]) { sq_set_error("edge_off must be non-decreasing"); return -1; }
        std::unordered_map<int32_t, int32_t> id;
        std::unordered_map<uint64_t, size_t> seen;
        std::vector<int32_t> &lab = G.labels[g];
        SqMatchJob &J = G.jobs[g];
        J.edge_off = (int64_t)G.edges.size(); J.pos_off = 0;
        auto vid = [&](int32_t label) {
            auto it = id.find(label);
            if (it != id.end()) return it->second;
            const int32_t k = (int32_t)lab.size();
            id.emplace(label, k); lab.push_back(label);
            return k;
        };
        for (int64_t e = edge_off[g]; e < edge_off[g + 1]; e++) {
            if (eu[e] < 0 || ev[e] < 0) { sq_set_error("vertex labels must be non-negative"); return -1; }
            const int32_t a = vid(eu[e]), b = vid(ev[e]);
            const uint64_t key = ((uint64_t)(uint32_t)std::min(a, b) << 32) | (uint32_t)std::max(a, b);
            auto it = seen.find(key);
            if (it != seen.end()) { G.edges[it->second].weight = ew ? ew[e] : 1.0; continue; }
            seen.emplace(key, G.edges.size());
            G.edges.push_back(SqMatchEdge{a, b, ew ? ew[e] : 1.0});
        }
        J.n = (int32_t)lab.size(); J.nedges = (int32_t)(G.edges.size() - (size_t)J.edge_off);
        J.scratch_off = (int64_t)G.scratch; G.scratch += up256(sq_mwm_scratch_bytes(J.n, J.nedges));
        J.out_off = (int64_t)G.outints; G.outints += 2 * (size_t)J.n + 2;
    }
    return 0;
}

// Cells of symmetric / upper-triangular sparse matrices, one problem per job; duplicates: the last one wins.
struct Cells {
    std::vector<SqMatchJob> jobs;
    std::vector<SqMatchEdge> edges;
    size_t scratch = 0, outints = 0, ncodes = 0;
};
int build_cells(int32_t nprob, const int32_t *n, const int64_t *cell_off, const int32_t *cv, const int32_t *cw,
                const double *val, bool nussinov, Cells &C)
{
    C.jobs.resize(nprob);
    for (int g = 0; g < nprob; g++) {
        if (n[g] < 0 || n[g] > 32000) { sq_set_error("matrix size out of range"); return -1; }
        if (cell_off[g + 1] < cell_off[g]) { sq_set_error("cell_off must be non-decreasing"); return -1; }
        SqMatchJob &J = C.jobs[g];
        J.n = n[g]; J.edge_off = (int64_t)C.edges.size(); J.pos_off = (int64_t)C.ncodes; C.ncodes += (size_t)n[g];
        std::unordered_map<uint64_t, size_t> seen;
        for (int64_t e = cell_off[g]; e < cell_off[g + 1]; e++) {
            int32_t a = cv[e], b = cw[e];
            if (a < 0 || b < 0 || a >= n[g] || b >= n[g] || a == b) { sq_set_error("cell outside the matrix (or on its diagonal)"); return -1; }
            if (nussinov && a > b) { sq_set_error("Nussinov cells must have v < w"); return -1; }
            if (a > b) std::swap(a, b);
            const uint64_t key = ((uint64_t)(uint32_t)a << 32) | (uint32_t)b;
            auto it = seen.find(key);
            if (it != seen.end()) { C.edges[it->second].weight = val[e]; continue; }
            seen.emplace(key, C.edges.size());
            C.edges.push_back(SqMatchEdge{a, b, val[e]});
        }
        J.nedges = (int32_t)(C.edges.size() - (size_t)J.edge_off);
        J.scratch_off = (int64_t)C.scratch;
        C.scratch += up256(nussinov ? sq_nussinov_scratch_bytes(J.n) : sq_lsap_scratch_bytes(J.n));
        J.out_off = (int64_t)(nussinov ? C.outints / 2 : C.outints);
        C.outints += nussinov ? 2 * ((size_t)J.n + 4) : (size_t)J.n;
    }
    return 0;
}

// workspace: [device copy of the edge list][letter codes][scratch]
size_t ws_bytes(size_t nedges, size_t ncodes, size_t scratch) { return up256(nedges * sizeof(SqMatchEdge) + 16) + up256(ncodes + 16) + scratch + 4096; }

// Stage jobs + edges in pinned memory, run the kernel, wait, hand the pinned result area to `read`.
template <class Read>
int run_matching(int algo, const std::vector<SqMatchJob> &jobs, const std::vector<SqMatchEdge> &edges, size_t outints,
                 size_t scratch, const uint8_t *codes, size_t ncodes, void *ws, size_t ws_have, hipStream_t st, Read read)
{
    if (jobs.empty()) return 0;
    const size_t need = ws_bytes(edges.size(), ncodes, scratch);
    if (!ws || ws_have < need) { sq_set_error("workspace too small"); return -2; }
    if (((uintptr_t)ws & 255) != 0) { sq_set_error("workspace must be 256-byte aligned"); return -2; }
    const size_t jb = up256(jobs.size() * sizeof(SqMatchJob)), eb = up256(edges.size() * sizeof(SqMatchEdge) + 16);
    const size_t ob = up256(outints * 4 + 16), cb = up256(jobs.size() * 4 + 16);
    Pinned pin;
    const size_t hb = up256(jobs.size() * 4 + 16);      // blossom: first job of every block (sq_mwm_plan)
    if (pin.get(jb + eb + ob + cb + hb + 256)) return 2;
    SqMatchJob *p_jobs = (SqMatchJob *)pin.p;
    SqMatchEdge *p_edges = (SqMatchEdge *)(pin.p + jb);
    int32_t *p_out = (int32_t *)(pin.p + jb + eb), *p_cnt = (int32_t *)(pin.p + jb + eb + ob);
    memcpy(p_jobs, jobs.data(), jobs.size() * sizeof(SqMatchJob));
    if (!edges.empty()) memcpy(p_edges, edges.data(), edges.size() * sizeof(SqMatchEdge));
    memset(p_out, 0, ob + cb);
    char *base = (char *)ws;
    SqMatchEdge *dev_edges = (SqMatchEdge *)base;
    uint8_t *d_codes = (uint8_t *)(base + up256(edges.size() * sizeof(SqMatchEdge) + 16));
    char *d_scr = (char *)d_codes + up256(ncodes + 16);
    if (codes && ncodes) HIPCK(hipMemcpyAsync(d_codes, codes, ncodes, hipMemcpyHostToDevice, st));
    const int rl = sq_launch_matching(algo, jobs.data(), (int)jobs.size(), p_jobs, p_edges, edges.size(), dev_edges, d_scr,
                                      p_out, p_cnt, d_codes, nullptr, 0, st, p_jobs, (int32_t *)(pin.p + jb + eb + ob + cb), 1);
    if (rl) { hipStreamSynchronize(st); return sq_check((hipError_t)rl, "matching kernel launch"); }
    HIPCK(hipStreamSynchronize(st));                   // results are in host memory (the kernels wrote them in place)
    return read(p_out, p_cnt);
}
}  // namespace

extern "C" int sq_mwm_workspace_bytes(int32_t ngraph, const int64_t *edge_off, const int32_t *eu, const int32_t *ev, size_t *bytes)
{
    if (ngraph < 0 || !edge_off || !bytes || (edge_off[ngraph] > 0 && (!eu || !ev))) { sq_set_error("bad argument"); return -1; }
    Graphs G;
    const int r = build_graphs(ngraph, edge_off, eu, ev, nullptr, G);
    if (r) return r;
    *bytes = ws_bytes(G.edges.size(), 0, G.scratch);
    return 0;
}

extern "C" int sq_mwm(int32_t ngraph, const int64_t *edge_off, const int32_t *eu, const int32_t *ev, const double *ew,
                      int32_t *pairs, int64_t pair_cap, int64_t *pair_off, void *dev_workspace, size_t workspace_bytes,
                      void *hip_stream)
{
    if (ngraph < 0 || !edge_off || !pair_off || (edge_off[ngraph] > 0 && (!eu || !ev || !ew)) || (pair_cap > 0 && !pairs)) {
        sq_set_error("bad argument"); return -1;
    }
    Graphs G;
    int r = build_graphs(ngraph, edge_off, eu, ev, ew, G);
    if (r) return r;
    for (int g = 0; g <= ngraph; g++) pair_off[g] = 0;
    return run_matching(SQ_ALGO_E, G.jobs, G.edges, G.outints, G.scratch, nullptr, 0, dev_workspace, workspace_bytes,
                        (hipStream_t)hip_stream, [&](const int32_t *out, const int32_t *) {
        int64_t np = 0;
        std::vector<std::pair<int32_t, int32_t>> ps;
        for (int g = 0; g < ngraph; g++) {
            const SqMatchJob &J = G.jobs[g];
            const int32_t *mate = out + J.out_off, *mord = mate + J.n;
            if (J.n > 0 && mate[0] == -2) { sq_set_capacity_error(SQ_CAP_FIXED, "blossom capacity exceeded"); return -3; }
            ps.clear();
            // networkx hands every pair out as (u, v) with u the endpoint that entered its `mate` dict first
            // (matching_dict_to_set); Edmonds() then sorts the tuples (SQRNalgos.py:109)
            for (int v = 0; v < J.n; v++)
                if (mate[v] >= 0 && mord[v] < mord[mate[v]]) ps.emplace_back(G.labels[g][v], G.labels[g][mate[v]]);
            std::sort(ps.begin(), ps.end());
            pair_off[g] = np;
            for (auto &p : ps) {
                if (np >= pair_cap) { sq_set_error("pair_cap too small"); return -3; }
                pairs[2 * np] = p.first; pairs[2 * np + 1] = p.second; np++;
            }
        }
        pair_off[ngraph] = np;
        return 0;
    });
}

extern "C" int sq_lsap_workspace_bytes(int32_t nprob, const int32_t *n, const int64_t *cell_off, size_t *bytes)
{
    if (nprob < 0 || !n || !cell_off || !bytes) { sq_set_error("bad argument"); return -1; }
    size_t scratch = 0;
    for (int g = 0; g < nprob; g++) {
        if (n[g] < 0 || n[g] > 32000) { sq_set_error("matrix size out of range"); return -1; }
        scratch += up256(sq_lsap_scratch_bytes(n[g]));
    }
    *bytes = ws_bytes((size_t)std::max<int64_t>(cell_off[nprob], 0), 0, scratch);
    return 0;
}

extern "C" int sq_lsap(int32_t nprob, const int32_t *n, const int64_t *cell_off, const int32_t *cv, const int32_t *cw,
                       const double *weight, int32_t *col4row, void *dev_workspace, size_t workspace_bytes, void *hip_stream)
{
    if (nprob < 0 || !n || !cell_off || !col4row || (cell_off[nprob] > 0 && (!cv || !cw || !weight))) { sq_set_error("bad argument"); return -1; }
    Cells C;
    const int r = build_cells(nprob, n, cell_off, cv, cw, weight, false, C);
    if (r) return r;
    return run_matching(SQ_ALGO_H, C.jobs, C.edges, C.outints, C.scratch, nullptr, 0, dev_workspace, workspace_bytes,
                        (hipStream_t)hip_stream, [&](const int32_t *out, const int32_t *) {
        memcpy(col4row, out, C.outints * 4);
        return 0;
    });
}

extern "C" int sq_nussinov_workspace_bytes(int32_t nprob, const int32_t *n, const int64_t *cell_off, size_t *bytes)
{
    if (nprob < 0 || !n || !cell_off || !bytes) { sq_set_error("bad argument"); return -1; }
    size_t scratch = 0, ncodes = 0;
    for (int g = 0; g < nprob; g++) {
        if (n[g] < 0 || n[g] > 32000) { sq_set_error("matrix size out of range"); return -1; }
        scratch += up256(sq_nussinov_scratch_bytes(n[g])); ncodes += (size_t)n[g];
    }
    *bytes = ws_bytes((size_t)std::max<int64_t>(cell_off[nprob], 0), ncodes, scratch);
    return 0;
}

extern "C" int sq_nussinov(int32_t nprob, const int32_t *n, const uint8_t *codes, const int64_t *cell_off, const int32_t *cv,
                           const int32_t *cw, const double *score, int32_t *pairs, int64_t pair_cap, int64_t *pair_off,
                           void *dev_workspace, size_t workspace_bytes, void *hip_stream)
{
    if (nprob < 0 || !n || !codes || !cell_off || !pair_off || (cell_off[nprob] > 0 && (!cv || !cw || !score)) || (pair_cap > 0 && !pairs)) {
        sq_set_error("bad argument"); return -1;
    }
    Cells C;
    const int r = build_cells(nprob, n, cell_off, cv, cw, score, true, C);
    if (r) return r;
    for (int g = 0; g <= nprob; g++) pair_off[g] = 0;
    return run_matching(SQ_ALGO_N, C.jobs, C.edges, C.outints, C.scratch, codes, C.ncodes, dev_workspace, workspace_bytes,
                        (hipStream_t)hip_stream, [&](const int32_t *out, const int32_t *cnt) {
        int64_t np = 0;
        std::vector<std::pair<int32_t, int32_t>> ps;
        for (int g = 0; g < nprob; g++) {
            const int32_t *pp = out + 2 * C.jobs[g].out_off;
            ps.clear();
            for (int t = 0; t < cnt[g]; t++) ps.emplace_back(pp[2 * t], pp[2 * t + 1]);
            std::sort(ps.begin(), ps.end());                 // BackTrack returns sorted(basepairs) (SQRNalgos.py:41)
            pair_off[g] = np;
            for (auto &p : ps) {
                if (np >= pair_cap) { sq_set_error("pair_cap too small"); return -3; }
                pairs[2 * np] = p.first; pairs[2 * np + 1] = p.second; np++;
            }
        }
        pair_off[nprob] = np;
        return 0;
    });
}
