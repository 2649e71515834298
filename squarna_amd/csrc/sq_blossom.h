// sq_blossom.h -- Edmonds maximum-weight matching for a-9 (SQRNalgos.py:96-110).
//
// The reference calls networkx.max_weight_matching(G) (networkx 3.4.2, pure Python, not
// vendored: Galil's O(n^3) blossom algorithm with dual variables, maxcardinality=False).
// 43 % of the SRtest150 inputs have a non-unique optimum, so this is a
// step-exact restatement: same vertex order (insertion order of the edge list), same
// neighbour order, same LIFO queue, same traversal of blossom leaves, same strict-< tie
// rules and the same fp64 operation order for slacks and deltas.  Python's insertion-ordered
// dicts become arrays + an ordered list of live blossoms; the two recursive generators
// (expandBlossom, augmentBlossom) become explicit stacks; a blossom's childs/edges lists are
// a cyclic doubly linked list through its children (rotation == moving `first`).
//
// Host+device inline code: the product runs it in sq_mwm_kernel (one thread per job); the
// test suite also compiles it on the host to compare with networkx directly.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "sq_match.h"
// (the cycle counters of the profiling builds -- SQ_DEFS=-DSQ_MWM_PROF / -DSQ_MWM_PROF2 -- exist on the device only; the host
// pass of a translation unit that includes this header compiles run() without them)
#if !defined(__HIP_DEVICE_COMPILE__)
#undef SQ_MWM_PROF
#undef SQ_MWM_PROF2
#endif

#ifdef __HIPCC__
#define SQ_HD __host__ __device__
#else
#define SQ_HD
#endif

// cross-lane helpers of the cooperative run(): one lane (host, tests) ...
struct SqCoopSingle {
    SQ_HD int first_true(bool p, int nl) const { return p ? 0 : nl; }
    SQ_HD void min_first(double &, int &, int) const {}
    SQ_HD void min_plain(double &, int = 64) const {}
    SQ_HD void min_i32(int &) const {}
    SQ_HD int excl_scan(int cnt, int &total) const { total = cnt; return 0; }
    SQ_HD int count_true(bool p, int &first) const { first = 0; return p ? 1 : 0; }
    static constexpr int kSegMax = 1;                           // vertices per scan pass
    SQ_HD int readlane(int x, int) const { return x; }
    SQ_HD bool any(bool p) const { return p; }
    SQ_HD void min_pos_f64(double *p, double v) const { if (v < *p) *p = v; }
    SQ_HD void add_i32(int *p, int v) const { *p += v; }
    SQ_HD void min_i32_at(int *p, int v) const { if (v < *p) *p = v; }
};
#ifdef __HIPCC__
// ... or the 64 lanes of a wave.  first_true: lowest lane whose predicate holds (nl if none);
// min_first: lexicographic minimum of (value, index) over the lanes with index < nl, broadcast to all.
struct SqCoopWave {
#ifndef SQ_MWM_SEGMAX
#define SQ_MWM_SEGMAX 4
#endif
    static constexpr int kSegMax = SQ_MWM_SEGMAX;               // vertices per scan pass (their neighbour lists share the 64 lanes)
    __device__ int readlane(int x, int l) const { return __builtin_amdgcn_readlane(x, l); }
    __device__ bool any(bool p) const { return __ballot(p) != 0; }
    __device__ void add_i32(int *p, int v) const { atomicAdd(p, v); }
    __device__ void min_i32_at(int *p, int v) const { atomicMin(p, v); }
    // *p = min(*p, v) for POSITIVE doubles (and +inf): their bit patterns order like unsigned integers
    __device__ void min_pos_f64(double *p, double v) const { atomicMin((unsigned long long *)p, (unsigned long long)__double_as_longlong(v)); }
    __device__ int first_true(bool p, int nl) const
    {
        const unsigned long long b = __ballot(p);
        return b ? (int)__ffsll((long long)b) - 1 : nl;
    }
    __device__ int count_true(bool p, int &first) const        // number of lanes whose predicate holds, and the lowest of them
    {
        const unsigned long long b = __ballot(p);
        first = b ? (int)__ffsll((long long)b) - 1 : 0;
        return (int)__popcll(b);
    }
    // Wave-wide minima with DPP row shifts / row broadcasts (the gfx9 scan sequence: row_shr 1, 2, 4, 8, row_bcast 15
    // and 31 leave the reduction of all 64 lanes in lane 63): VALU only, no LDS crossbar.  The 6-step __shfl_xor
    // reductions these replace cost three ds_bpermute per step for a (double, index) pair; measured with cycle
    // counters on the largest SRtest150 graph, that reduction was 45 % of the kernel's time (it runs on almost every
    // scan pass: a freshly popped vertex is its own blossom with no best edge yet, so every S-neighbour competes).
    template <int CTRL, int ROWMASK>
    static __device__ __forceinline__ double dpp_f64(double ident, double v)
    {
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROWMASK, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    // every lane gets the minimum over lanes [0, nlive) (no NaNs); lanes >= nlive must hold +inf or be ignorable:
    // with nlive <= 16 the first DPP row already holds everything (4 steps), with <= 32 the first two rows (5 steps)
    static __device__ __forceinline__ double wave_min_f64(double v, int nlive = 64)
    {
        const double inf = __longlong_as_double(0x7FF0000000000000ll);
        double o;
        o = dpp_f64<0x111, 0xf>(inf, v); v = __builtin_fmin(o, v);
        o = dpp_f64<0x112, 0xf>(inf, v); v = __builtin_fmin(o, v);
        o = dpp_f64<0x114, 0xf>(inf, v); v = __builtin_fmin(o, v);
        o = dpp_f64<0x118, 0xf>(inf, v); v = __builtin_fmin(o, v);
        int src = 15;
        if (nlive > 16) {
            o = dpp_f64<0x142, 0xa>(inf, v); v = __builtin_fmin(o, v);
            src = 31;
            if (nlive > 32) {
                o = dpp_f64<0x143, 0xc>(inf, v); v = __builtin_fmin(o, v);
                src = 63;
            }
        }
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ int wave_min_i32(int v)
    {
        const int big = 0x7fffffff;
        int o;
        o = __builtin_amdgcn_update_dpp(big, v, 0x111, 0xf, 0xf, false); v = o < v ? o : v;
        o = __builtin_amdgcn_update_dpp(big, v, 0x112, 0xf, 0xf, false); v = o < v ? o : v;
        o = __builtin_amdgcn_update_dpp(big, v, 0x114, 0xf, 0xf, false); v = o < v ? o : v;
        o = __builtin_amdgcn_update_dpp(big, v, 0x118, 0xf, 0xf, false); v = o < v ? o : v;
        o = __builtin_amdgcn_update_dpp(big, v, 0x142, 0xa, 0xf, false); v = o < v ? o : v;
        o = __builtin_amdgcn_update_dpp(big, v, 0x143, 0xc, 0xf, false); v = o < v ? o : v;
        return __builtin_amdgcn_readlane(v, 63);
    }
    // lexicographic minimum of (value, index) over the lanes with index < nl, broadcast to all lanes (index nl: none)
    __device__ void min_first(double &v, int &i, int nl) const
    {
        const bool valid = i < nl;
        const double mv = wave_min_f64(valid ? v : __longlong_as_double(0x7FF0000000000000ll));
        const int mi = wave_min_i32(valid && v == mv ? i : 0x7fffffff);
        v = mv; i = mi == 0x7fffffff ? nl : mi;
    }
    __device__ void min_plain(double &v, int nlive = 64) const { v = wave_min_f64(v, nlive); }
    __device__ void min_i32(int &v) const { v = wave_min_i32(v); }
    __device__ int excl_scan(int cnt, int &total) const       // exclusive prefix sum over the lanes + wave total
    {
        int v = cnt;                                           // inclusive scan: the same DPP sequence, with +
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
        total = __builtin_amdgcn_readlane(v, 63);
        return v - cnt;
    }
};
#endif

#if defined(__HIP_DEVICE_COMPILE__)
extern __shared__ __attribute__((aligned(16))) char sq_mwm_dyn_lds[];   // the kernel's dynamic LDS (sq_mwm_kernel)
#endif

struct SqBlossom {
    // graph
    int n, m;                 // vertices (graph order), undirected edges
    const SqMatchEdge *E;     // E[e] = (v, w, weight), vertex ids in graph order
    int *adj_off, *adj;       // CSR, directed edge codes (2e | dir) in neighbour insertion order
    int *adjv; double *adjw;  // per CSR slot: the neighbour vertex and the edge weight (one LDS level less in the scan)
    // state (index: vertex 0..n-1, blossom n..2n-1)
    int *mate;                // mate vertex or -1 (output)
    int *mate_de;             // directed edge v -> mate[v]
    int *mord; int mord_n;    // mord[v]: rank of v's FIRST mate assignment == its position in networkx's `mate` dict,
                              // which decides the orientation (u, v) of the returned pairs (matching_dict_to_set)
    int8_t *label;            // 0 none, 1 S, 2 T, 5 scanned mark
    int *labeledge, *inblossom, *parent, *base, *bestedge;
    double *dualvar, *bdual;
    double *bslack;           // bslack[x] == slack(bestedge[x]) under the current duals whenever bestedge[x] != -1, +inf when there
                              // is none (the several-vertices-per-pass scan competes for best edges with an LDS minimum on it)
    uint8_t *allow;           // per undirected edge
    int *queue; int qn, qcap;
    // blossoms
    int *sib_next, *sib_prev, *edge_after, *first, *nchild, *nleaf;
    int *mbe_off, *mbe_cnt;   // mybestedges (cnt < 0: None)
    int *pool; int pool_n, pool_cap;
    int *live; int nlive;     // live blossoms in creation order (dict order of blossomparent/blossomdual)
    int *freeb; int nfree;
    // temporaries
    int *tmp_leaves, *tmp_stack, *tmp_path, *tmp_edges, *beto, *beto_keys, *frames;
    int frame_cap;
    int error;

    // tight = 1: capacities that fit LDS for typical stem graphs (overflow sets `error`, the
    // caller then reruns the job with tight = 0 in global memory)
    SQ_HD static int queue_cap(int n, int m, int tight) { return tight ? 4 * n + m + 16 : 8 * n + 2 * m + 16; }
    SQ_HD static int pool_capacity(int n, int m, int tight) { return tight ? 4 * n + m + 256 : n * 32 + 2 * m + 1024; }
    SQ_HD static int frame_ints(int n, int tight) { return tight ? 8 * (n / 2 + 4) : 10 * (2 * n + 2); }
    // The state is carved into three parts so that it can be placed by temperature:
    //   hot   what every scan pass of the queue loop reads or writes (vertex / blossom state + allowedge + queue)
    //   cold  the blossom structure, only touched by lane 0 when a neighbour changes shared state
    //   edge  the adjacency in CSR order (streamed: one coalesced read per pass)
    SQ_HD static int queue_hot_cap(int n) { return 4 * n + 16; }       // S-vertices of one stage (overflow -> error)
    SQ_HD static size_t hot_bytes(int n, int m, int tight)
    {
        const size_t N2 = 2 * (size_t)n + 2;
        const size_t q = tight == 2 ? (size_t)queue_hot_cap(n) : (size_t)queue_cap(n, m, tight);
        return ((size_t)n + N2) * 8                              // dualvar, bslack
               + ((size_t)n + 1 + (size_t)n + 2 * N2 + q) * 4    // adj_off, inblossom, bestedge, labeledge, queue
               + N2 + (size_t)m + 64;                            // label, allow
    }
    SQ_HD static size_t cold_bytes(int n, int m, int tight)
    {
        const size_t N2 = 2 * (size_t)n + 2;
        size_t ints = 3 * (size_t)n                              // mate, mate_de, mord
                      + 2 * N2                                   // parent, base
                      + 8 * N2                                   // sib_next, sib_prev, edge_after, first, nchild, nleaf, mbe_off, mbe_cnt
                      + (size_t)pool_capacity(n, m, tight ? 1 : 0)   // pool
                      + 2 * (size_t)n + 2                        // live, freeb
                      + 2 * N2 + 4 * N2 + 2 * N2                 // tmp_leaves, tmp_stack | tmp_path, tmp_edges (2 each) | beto, beto_keys
                      + (size_t)frame_ints(n, tight ? 1 : 0);    // frames
        return ints * 4 + N2 * 8 + 64;                           // + bdual
    }
    SQ_HD static size_t edge_bytes(int m) { return 2 * (size_t)m * (4 + 4 + 8) + 64; }   // adj, adjv, adjw
    SQ_HD static size_t scratch_bytes(int n, int m, int tight = 0)
    {
        return hot_bytes(n, m, tight) + cold_bytes(n, m, tight) + edge_bytes(m) + 64;
    }

    // hot / cold / edge: where the three parts live; cold == nullptr (edge == nullptr): carved right behind the
    // previous part.  tight: 0 generous capacities, 1 tight, 2 tight with the short queue (hot part alone in LDS).
    SQ_HD void init(int n_, int m_, const SqMatchEdge *edges, char *hot, int tight = 0, bool csr = true,
                    char *cold = nullptr, char *edge = nullptr)
    {
        n = n_; m = m_; E = edges; error = 0;
        const int N2 = 2 * n + 2;
        char *p = hot;
        auto take_d = [&](size_t k) { p = (char *)(((uintptr_t)p + 7) & ~(uintptr_t)7); double *r = (double *)p; p += k * 8; return r; };
        auto take_i = [&](size_t k) { p = (char *)(((uintptr_t)p + 3) & ~(uintptr_t)3); int *r = (int *)p; p += k * 4; return r; };
        // ---- hot
        dualvar = take_d(n); bslack = take_d(N2);
        adj_off = take_i(n + 1); inblossom = take_i(n); bestedge = take_i(N2); labeledge = take_i(N2);
        qcap = tight == 2 ? queue_hot_cap(n) : queue_cap(n, m, tight); queue = take_i(qcap); qn = 0;
        label = (int8_t *)p; p += (N2 + 3) & ~3; allow = (uint8_t *)p; p += ((size_t)m + 3) & ~(size_t)3;   // (cleared as words)
        // ---- cold
        if (cold) p = cold;
        bdual = take_d(N2);
        mate = take_i(n); mate_de = take_i(n); mord = take_i(n); mord_n = 0;
        parent = take_i(N2); base = take_i(N2);
        sib_next = take_i(N2); sib_prev = take_i(N2); edge_after = take_i(N2); first = take_i(N2); nchild = take_i(N2); nleaf = take_i(N2);
        mbe_off = take_i(N2); mbe_cnt = take_i(N2);
        pool_cap = pool_capacity(n, m, tight ? 1 : 0); pool = take_i(pool_cap); pool_n = 0;
        live = take_i(n + 1); freeb = take_i(n + 1);
        tmp_leaves = take_i(N2); tmp_stack = take_i(N2); tmp_path = take_i(2 * (size_t)N2); tmp_edges = take_i(2 * (size_t)N2);
        beto = take_i(N2); beto_keys = take_i(N2); frame_cap = frame_ints(n, tight ? 1 : 0); frames = take_i((size_t)frame_cap);
        // ---- edge
        if (edge) p = edge;
        adjw = take_d(2 * (size_t)m);
        adj = take_i(2 * (size_t)m); adjv = take_i(2 * (size_t)m);
        if (!csr) return;                                       // the caller builds it with build_csr()
        // adjacency in insertion order: edge e = (v, w) appends w to adj[v] and v to adj[w]
        for (int v = 0; v <= n; v++) adj_off[v] = 0;
        for (int e = 0; e < m; e++) { adj_off[E[e].v + 1]++; adj_off[E[e].w + 1]++; }
        for (int v = 0; v < n; v++) adj_off[v + 1] += adj_off[v];
        for (int v = 0; v < n; v++) mate[v] = 0;                 // used as fill cursor
        for (int e = 0; e < m; e++) {
            const int sv = adj_off[E[e].v] + mate[E[e].v]++, sw = adj_off[E[e].w] + mate[E[e].w]++;
            adj[sv] = 2 * e;     adjv[sv] = E[e].w; adjw[sv] = E[e].weight;
            adj[sw] = 2 * e + 1; adjv[sw] = E[e].v; adjw[sw] = E[e].weight;
        }
    }

    // FAST: which parts of the state live in the kernel's dynamic LDS buffer (1: everything incl. the edge list, 2: only
    // the hot part, 0: nothing).  Every array access goes through SQ_LP (hot arrays) / SQ_LQ (cold, edge arrays, E):
    // with the array in LDS the address is formed as <dynamic LDS base> + offset, which lets the compiler prove the
    // address space and emit ds_read / ds_write instead of flat loads through the LDS aperture -- in the scan loop AND
    // in the lane-0 event paths (assignLabel, addBlossom, augmentMatching, ...), which are templates on FAST for that.
    char *origin;             // generic address of the LDS buffer that holds edges + state (FAST runs only)
#if defined(__HIP_DEVICE_COMPILE__)
#define SQ_FAST0 sq_mwm_dyn_lds
#else
#define SQ_FAST0 ((char *)nullptr)
#endif
#define SQ_BINF __builtin_huge_val()
#define SQ_ORDERED() asm volatile("" ::: "memory")
#ifdef __HIPCC__
#define SQ_UNROLL3 _Pragma("unroll 3")
#else
#define SQ_UNROLL3
#endif
#define SQ_LP(p) (FAST ? (decltype(p))(SQ_FAST0 + ((char *)(p) - origin)) : (p))            /* hot arrays */
#define SQ_LQ(p) (FAST == 1 ? (decltype(p))(SQ_FAST0 + ((char *)(p) - origin)) : (p))       /* cold / edge arrays, E */
    template <int FAST = 0> SQ_HD int tail(int de) const { return (de & 1) ? SQ_LQ(E)[de >> 1].w : SQ_LQ(E)[de >> 1].v; }
    template <int FAST = 0> SQ_HD int head(int de) const { return (de & 1) ? SQ_LQ(E)[de >> 1].v : SQ_LQ(E)[de >> 1].w; }
    template <int FAST = 0> SQ_HD double slack(int de) const { return SQ_LP(dualvar)[tail<FAST>(de)] + SQ_LP(dualvar)[head<FAST>(de)] - 2 * SQ_LQ(E)[de >> 1].weight; }
    SQ_HD bool is_blossom(int x) const { return x >= n; }
    template <int FAST = 0> SQ_HD void qpush(int v) { if (qn < qcap) SQ_LP(queue)[qn++] = v; else error = 1; }

    // Blossom.leaves(): stack = [*childs]; pop from the end; sub-blossoms push their childs
    template <int FAST = 0> SQ_HD int leaves(int b, int *out)
    {
        int sn = 0, cnt = 0;
        int c = SQ_LQ(first)[b];
        for (int k = 0; k < SQ_LQ(nchild)[b]; k++) { SQ_LQ(tmp_stack)[sn++] = c; c = SQ_LQ(sib_next)[c]; }
        while (sn) {
            const int t = SQ_LQ(tmp_stack)[--sn];
            if (is_blossom(t)) {
                int cc = SQ_LQ(first)[t];
                for (int k = 0; k < SQ_LQ(nchild)[t]; k++) { SQ_LQ(tmp_stack)[sn++] = cc; cc = SQ_LQ(sib_next)[cc]; }
            } else out[cnt++] = t;
        }
        return cnt;
    }

    template <int FAST = 0> SQ_HD void assignLabel(int w, int t, int de)       // de: labeledge (v, w) or -1
    {
        for (;;) {
            const int b = SQ_LP(inblossom)[w];
            SQ_LP(label)[w] = SQ_LP(label)[b] = (int8_t)t;
            SQ_LP(labeledge)[w] = SQ_LP(labeledge)[b] = de;
            SQ_LP(bestedge)[w] = SQ_LP(bestedge)[b] = -1;
            SQ_LP(bslack)[w] = SQ_LP(bslack)[b] = SQ_BINF;
            if (t == 1) {
                if (is_blossom(b)) {
                    const int c = leaves<FAST>(b, SQ_LQ(tmp_leaves));
                    for (int k = 0; k < c; k++) qpush<FAST>(SQ_LQ(tmp_leaves)[k]);
                } else qpush<FAST>(b);
                return;
            }
            // t == 2: the mate of the base becomes an S-vertex
            const int bs = SQ_LQ(base)[b];
            de = SQ_LQ(mate_de)[bs];                            // (base, SQ_LQ(mate)[base])
            w = SQ_LQ(mate)[bs]; t = 1;
        }
    }

    template <int FAST = 0> SQ_HD int scanBlossom(int v, int w)
    {
        int pn = 0, bs = -1;
        while (v != -1) {
            int b = SQ_LP(inblossom)[v];
            if (SQ_LP(label)[b] & 4) { bs = SQ_LQ(base)[b]; break; }
            SQ_LQ(tmp_path)[pn++] = b;
            SQ_LP(label)[b] = 5;
            if (SQ_LP(labeledge)[b] == -1) v = -1;
            else {
                v = tail<FAST>(SQ_LP(labeledge)[b]);
                b = SQ_LP(inblossom)[v];
                v = tail<FAST>(SQ_LP(labeledge)[b]);
            }
            if (w != -1) { const int t = v; v = w; w = t; }
        }
        for (int k = 0; k < pn; k++) SQ_LP(label)[SQ_LQ(tmp_path)[k]] = 1;
        return bs;
    }

    template <int FAST = 0> SQ_HD int new_blossom() { if (nfree == 0) { error = 2; return n; } return SQ_LQ(freeb)[--nfree]; }

    template <int FAST = 0> SQ_HD void addBlossom(int bs, int de)               // de = (v, w)
    {
        int v = tail<FAST>(de), w = head<FAST>(de);
        const int bb = SQ_LP(inblossom)[bs];
        int bv = SQ_LP(inblossom)[v], bw = SQ_LP(inblossom)[w];
        const int b = new_blossom<FAST>();
        SQ_LQ(base)[b] = bs; SQ_LQ(parent)[b] = -1; SQ_LQ(parent)[bb] = b;
        SQ_LQ(live)[nlive++] = b;
        int *const path = SQ_LQ(tmp_path), *const edgs = SQ_LQ(tmp_edges);          // python lists
        int pn = 0, en = 0;
        edgs[en++] = de;
        while (bv != bb) {
            SQ_LQ(parent)[bv] = b;
            path[pn++] = bv;
            edgs[en++] = SQ_LP(labeledge)[bv];
            v = tail<FAST>(SQ_LP(labeledge)[bv]);
            bv = SQ_LP(inblossom)[v];
        }
        path[pn++] = bb;
        for (int a = 0, z = pn - 1; a < z; a++, z--) { const int t = path[a]; path[a] = path[z]; path[z] = t; }
        for (int a = 0, z = en - 1; a < z; a++, z--) { const int t = edgs[a]; edgs[a] = edgs[z]; edgs[z] = t; }
        while (bw != bb) {
            SQ_LQ(parent)[bw] = b;
            path[pn++] = bw;
            edgs[en++] = SQ_LP(labeledge)[bw] ^ 1;                // (SQ_LP(labeledge)[bw][1], SQ_LP(labeledge)[bw][0])
            w = tail<FAST>(SQ_LP(labeledge)[bw]);
            bw = SQ_LP(inblossom)[w];
        }
        // childs = path, edges = edgs (edges[i] joins childs[i] -> childs[i+1], cyclically)
        SQ_LQ(nchild)[b] = pn; SQ_LQ(first)[b] = path[0];
        for (int k = 0; k < pn; k++) {
            SQ_LQ(sib_next)[path[k]] = path[(k + 1) % pn];
            SQ_LQ(sib_prev)[path[k]] = path[(k + pn - 1) % pn];
            SQ_LQ(edge_after)[path[k]] = edgs[k];
        }
        SQ_LP(label)[b] = 1;
        SQ_LP(labeledge)[b] = SQ_LP(labeledge)[bb];
        SQ_LQ(bdual)[b] = 0;
        {
            const int c = leaves<FAST>(b, SQ_LQ(tmp_leaves));
            SQ_LQ(nleaf)[b] = c;                                   // fixed for the blossom's lifetime
            for (int k = 0; k < c; k++) {
                const int x = SQ_LQ(tmp_leaves)[k];
                if (SQ_LP(label)[SQ_LP(inblossom)[x]] == 2) qpush<FAST>(x);
                SQ_LP(inblossom)[x] = b;
            }
        }
        // bestedgeto: dict bj -> edge (insertion ordered)
        int nk = 0;
        for (int k = 0; k < pn; k++) {
            const int cv = path[k];
            // nblist
            int lstart = pool_n, lcount = 0; bool from_pool = false;
            if (is_blossom(cv) && SQ_LQ(mbe_cnt)[cv] >= 0) {
                lstart = SQ_LQ(mbe_off)[cv]; lcount = SQ_LQ(mbe_cnt)[cv]; from_pool = true;
                SQ_LQ(mbe_cnt)[cv] = -1;
            }
            int nleaf = 1;
            if (!from_pool) {
                if (is_blossom(cv)) nleaf = leaves<FAST>(cv, SQ_LQ(tmp_leaves)); else SQ_LQ(tmp_leaves)[0] = cv;
            }
            int li = 0, ai = 0;                             // iterate nblist lazily
            for (;;) {
                int kde;
                if (from_pool) {
                    if (li >= lcount) break;
                    kde = SQ_LQ(pool)[lstart + li++];
                } else {
                    if (li >= nleaf) break;
                    const int x = SQ_LQ(tmp_leaves)[li];
                    if (ai >= SQ_LP(adj_off)[x + 1] - SQ_LP(adj_off)[x]) { li++; ai = 0; continue; }
                    kde = SQ_LQ(adj)[SQ_LP(adj_off)[x] + ai++];
                }
                int i = tail<FAST>(kde), j = head<FAST>(kde);
                if (SQ_LP(inblossom)[j] == b) { const int t = i; i = j; j = t; }
                const int bj = SQ_LP(inblossom)[j];
                if (bj != b && SQ_LP(label)[bj] == 1) {
                    const int ide = (tail<FAST>(kde) == i) ? kde : (kde ^ 1);   // slack<FAST>(i, j)
                    if (SQ_LQ(beto)[bj] == -1) { SQ_LQ(beto)[bj] = kde; SQ_LQ(beto_keys)[nk++] = bj; }
                    else if (slack<FAST>(ide) < slack<FAST>(SQ_LQ(beto)[bj])) SQ_LQ(beto)[bj] = kde;
                }
            }
            SQ_LP(bestedge)[cv] = -1; SQ_LP(bslack)[cv] = SQ_BINF;
        }
        // b.mybestedges = list(bestedgeto.values())
        SQ_LQ(mbe_off)[b] = pool_n; SQ_LQ(mbe_cnt)[b] = nk;
        int mybest = -1; double mybestslack = SQ_BINF;
        for (int k = 0; k < nk; k++) {
            const int kde = SQ_LQ(beto)[SQ_LQ(beto_keys)[k]];
            SQ_LQ(beto)[SQ_LQ(beto_keys)[k]] = -1;
            if (pool_n < pool_cap) SQ_LQ(pool)[pool_n++] = kde; else error = 3;
            const double ks = slack<FAST>(kde);
            if (mybest == -1 || ks < mybestslack) { mybest = kde; mybestslack = ks; }
        }
        SQ_LP(bestedge)[b] = mybest; SQ_LP(bslack)[b] = mybestslack;
    }

    template <int FAST = 0> SQ_HD void remove_live(int b)
    {
        int k = 0;
        while (k < nlive && SQ_LQ(live)[k] != b) k++;
        for (; k + 1 < nlive; k++) SQ_LQ(live)[k] = SQ_LQ(live)[k + 1];
        nlive--;
        SQ_LQ(freeb)[nfree++] = b;
    }

    // expandBlossom<FAST>(b, endstage) with the recursion of _recurse made explicit
    template <int FAST = 0> SQ_HD void expandBlossom(int b0, bool endstage)
    {
        int *const fr = SQ_LQ(frames);                                  // frame: (b, next child, remaining)
        int sp = 0;
        fr[0] = b0; fr[1] = SQ_LQ(first)[b0]; fr[2] = SQ_LQ(nchild)[b0]; sp = 1;
        while (sp) {
            int *f = fr + 3 * (sp - 1);
            const int b = f[0];
            if (f[2] > 0) {
                const int s = f[1];
                f[1] = SQ_LQ(sib_next)[s]; f[2]--;
                SQ_LQ(parent)[s] = -1;
                if (is_blossom(s)) {
                    if (endstage && SQ_LQ(bdual)[s] == 0) {       // yield s: expand it now, then continue with the next child
                        if (3 * (sp + 1) > frame_cap) { error = 4; return; }
                        int *g = fr + 3 * sp;
                        g[0] = s; g[1] = SQ_LQ(first)[s]; g[2] = SQ_LQ(nchild)[s]; sp++;
                    } else {
                        const int c = leaves<FAST>(s, SQ_LQ(tmp_leaves));
                        for (int k = 0; k < c; k++) SQ_LP(inblossom)[SQ_LQ(tmp_leaves)[k]] = s;
                    }
                } else SQ_LP(inblossom)[s] = s;
                continue;
            }
            if (!endstage && SQ_LP(label)[b] == 2) {
                const int entry = SQ_LP(inblossom)[head<FAST>(SQ_LP(labeledge)[b])];
                int j = 0;
                { int c = SQ_LQ(first)[b]; while (c != entry) { c = SQ_LQ(sib_next)[c]; j++; } }
                int c = entry, jstep;
                if (j & 1) { j -= SQ_LQ(nchild)[b]; jstep = 1; } else jstep = -1;
                int de = SQ_LP(labeledge)[b];                     // (v, w)
                while (j != 0) {
                    int pq;                                // directed (p, q)
                    if (jstep == 1) pq = SQ_LQ(edge_after)[c]; else pq = SQ_LQ(edge_after)[SQ_LQ(sib_prev)[c]] ^ 1;
                    const int w = head<FAST>(de), q = head<FAST>(pq);
                    SQ_LP(label)[w] = 0; SQ_LP(label)[q] = 0;
                    assignLabel<FAST>(w, 2, de);
                    SQ_LP(allow)[pq >> 1] = 1;
                    j += jstep; c = (jstep == 1) ? SQ_LQ(sib_next)[c] : SQ_LQ(sib_prev)[c];
                    if (jstep == 1) de = SQ_LQ(edge_after)[c]; else de = SQ_LQ(edge_after)[SQ_LQ(sib_prev)[c]] ^ 1;
                    SQ_LP(allow)[de >> 1] = 1;
                    j += jstep; c = (jstep == 1) ? SQ_LQ(sib_next)[c] : SQ_LQ(sib_prev)[c];
                }
                const int bw = c;                           // b.childs[0]
                const int w = head<FAST>(de);
                SQ_LP(label)[w] = SQ_LP(label)[bw] = 2;
                SQ_LP(labeledge)[w] = SQ_LP(labeledge)[bw] = de;
                SQ_LP(bestedge)[bw] = -1; SQ_LP(bslack)[bw] = SQ_BINF;
                c = (jstep == 1) ? SQ_LQ(sib_next)[c] : SQ_LQ(sib_prev)[c];
                while (c != entry) {
                    const int bv = c;
                    if (SQ_LP(label)[bv] == 1) { c = (jstep == 1) ? SQ_LQ(sib_next)[c] : SQ_LQ(sib_prev)[c]; continue; }
                    int v;
                    if (is_blossom(bv)) {
                        const int cn = leaves<FAST>(bv, SQ_LQ(tmp_leaves));
                        v = SQ_LQ(tmp_leaves)[cn - 1];
                        for (int k = 0; k < cn; k++) if (SQ_LP(label)[SQ_LQ(tmp_leaves)[k]]) { v = SQ_LQ(tmp_leaves)[k]; break; }
                    } else v = bv;
                    if (SQ_LP(label)[v]) {
                        SQ_LP(label)[v] = 0;
                        SQ_LP(label)[SQ_LQ(mate)[SQ_LQ(base)[bv]]] = 0;
                        assignLabel<FAST>(v, 2, SQ_LP(labeledge)[v]);
                    }
                    c = (jstep == 1) ? SQ_LQ(sib_next)[c] : SQ_LQ(sib_prev)[c];
                }
            }
            SQ_LP(label)[b] = 0; SQ_LP(labeledge)[b] = -1; SQ_LP(bestedge)[b] = -1; SQ_LP(bslack)[b] = SQ_BINF;
            SQ_LQ(parent)[b] = -1; SQ_LQ(base)[b] = -1; SQ_LQ(bdual)[b] = 0; SQ_LQ(mbe_cnt)[b] = -1;
            remove_live<FAST>(b);
            sp--;
        }
    }

    // augmentBlossom<FAST>(b, v) with explicit frames: (b, v, phase, t0, j, jstep, c, de_wx)
    template <int FAST = 0> SQ_HD void augmentBlossom(int b0, int v0)
    {
        int *const fr = SQ_LQ(frames);
        int sp = 1;
        fr[0] = b0; fr[1] = v0; fr[2] = 0;
        while (sp) {
            if (8 * (sp + 1) > frame_cap) { error = 4; return; }
            int *f = fr + 8 * (sp - 1);
            const int b = f[0], v = f[1];
            if (f[2] == 0) {
                int t = v;
                while (SQ_LQ(parent)[t] != b) t = SQ_LQ(parent)[t];
                f[3] = t;
                int j = 0;
                { int c = SQ_LQ(first)[b]; while (c != t) { c = SQ_LQ(sib_next)[c]; j++; } }
                if (j & 1) { f[4] = j - SQ_LQ(nchild)[b]; f[5] = 1; } else { f[4] = j; f[5] = -1; }
                f[6] = t; f[2] = 1;
                if (is_blossom(t)) { int *g = fr + 8 * sp; g[0] = t; g[1] = v; g[2] = 0; sp++; continue; }
            }
            if (f[2] == 1) {                                // top of `while j != 0`
                if (f[4] == 0) {
                    SQ_LQ(first)[b] = f[3];                        // childs = childs[i:] + childs[:i]
                    SQ_LQ(base)[b] = SQ_LQ(base)[SQ_LQ(first)[b]];
                    sp--;
                    continue;
                }
                const int jstep = f[5];
                f[4] += jstep; f[6] = (jstep == 1) ? SQ_LQ(sib_next)[f[6]] : SQ_LQ(sib_prev)[f[6]];
                const int t = f[6];
                f[7] = (jstep == 1) ? SQ_LQ(edge_after)[t] : (SQ_LQ(edge_after)[SQ_LQ(sib_prev)[t]] ^ 1);   // (w, x)
                f[2] = 2;
                if (is_blossom(t)) { int *g = fr + 8 * sp; g[0] = t; g[1] = tail<FAST>(f[7]); g[2] = 0; sp++; continue; }
            }
            if (f[2] == 2) {
                const int jstep = f[5];
                f[4] += jstep; f[6] = (jstep == 1) ? SQ_LQ(sib_next)[f[6]] : SQ_LQ(sib_prev)[f[6]];
                const int t = f[6];
                f[2] = 3;
                if (is_blossom(t)) { int *g = fr + 8 * sp; g[0] = t; g[1] = head<FAST>(f[7]); g[2] = 0; sp++; continue; }
            }
            if (f[2] == 3) {
                const int w = tail<FAST>(f[7]), x = head<FAST>(f[7]);
                SQ_LQ(mate)[w] = x; SQ_LQ(mate_de)[w] = f[7];
                SQ_LQ(mate)[x] = w; SQ_LQ(mate_de)[x] = f[7] ^ 1;
                if (SQ_LQ(mord)[w] < 0) SQ_LQ(mord)[w] = mord_n++;
                if (SQ_LQ(mord)[x] < 0) SQ_LQ(mord)[x] = mord_n++;
                f[2] = 1;
            }
        }
    }

    template <int FAST = 0> SQ_HD void augmentMatching(int de)                     // (v, w)
    {
        for (int side = 0; side < 2; side++) {
            int sj = side == 0 ? de : (de ^ 1);            // (s, j)
            for (;;) {
                const int s = tail<FAST>(sj), j = head<FAST>(sj);
                const int bs = SQ_LP(inblossom)[s];
                if (is_blossom(bs)) augmentBlossom<FAST>(bs, s);
                SQ_LQ(mate)[s] = j; SQ_LQ(mate_de)[s] = sj;
                if (SQ_LQ(mord)[s] < 0) SQ_LQ(mord)[s] = mord_n++;
                if (SQ_LP(labeledge)[bs] == -1) break;
                const int t = tail<FAST>(SQ_LP(labeledge)[bs]);
                const int bt = SQ_LP(inblossom)[t];
                sj = SQ_LP(labeledge)[bt];                         // s, j = SQ_LP(labeledge)[bt]
                const int s2 = tail<FAST>(sj), j2 = head<FAST>(sj);
                if (is_blossom(bt)) augmentBlossom<FAST>(bt, j2);
                SQ_LQ(mate)[j2] = s2; SQ_LQ(mate_de)[j2] = sj ^ 1;
                if (SQ_LQ(mord)[j2] < 0) SQ_LQ(mord)[j2] = mord_n++;
            }
        }
    }

    // cooperative form: every lane of a wave (or the single host thread: lane 0 of 1) calls run() on the
    // SAME object.  The order-dependent work stays on lane 0; the O(n) sweeps of every substage (label
    // clears, the four delta minima, the dual updates) are strided over the lanes.  Minima keep the
    // sequential rule "first strictly smaller wins": per lane the first minimum of its stride, then
    // across lanes the smallest value and, among equals, the smallest iteration index.
    int f_augmented, f_stop, f_break;
    int stat_pass, stat_event;   // scan passes (one chunk of <= 64 neighbours of a popped S-vertex) and lane-0 events of the run
#ifdef SQ_MWM_PROF
    long long pt[8]; long long pc[8];
#define SQ_PT(k, expr) do { const long long _t0 = wall_clock64(); expr; pt[k] += wall_clock64() - _t0; } while (0)
#else
#define SQ_PT(k, expr) do { expr; } while (0)
#endif
    int red_k[3];                     // the dual step's decision for lane 0's action (type, edge, blossom)
    int red_i[64][2];                 // per lane: (blossom, queue position) of the queue refill

    // The same adjacency, built by all lanes.  Counting is order-free: the lanes walk the edge list together and add to
    // the degrees with LDS atomics.  Filling keeps the edge order per vertex: the edges are taken nl at a time in list
    // order; a lane's slot in the list of an end x of its edge is the list's cursor + the number of ends equal to x that
    // the lanes BEFORE it hold (found in as many rounds as the chunk names a vertex: typically one or two), and the
    // cursors move, again with order-free atomics, once the whole chunk is placed.  (Until round 4 every lane walked the WHOLE edge list twice for each of
    // its vertices: 470 us of the critical SRtest150 graph's 3.0 ms, m = 1,320.)  The cursors use bestedge's room, the rounds labeledge's.
    // A self-loop takes two slots of its vertex's list, as it always did (the scan skips them).
    template <int FAST, class Sync, class Coop>
    SQ_HD void build_csr(int lane, int nl, Sync sync, Coop coop)
    {
        const SqMatchEdge *const E_ = SQ_LQ(E);
        int *const adj_off_ = SQ_LP(adj_off), *const adj_ = SQ_LQ(adj), *const adjv_ = SQ_LQ(adjv), *const cur_ = SQ_LP(bestedge), *const tmp_ = SQ_LP(labeledge);
        double *const adjw_ = SQ_LQ(adjw);
        for (int v = lane; v <= n; v += nl) adj_off_[v] = 0;
        sync();
        for (int e = lane; e < m; e += nl) { const SqMatchEdge ed = E_[e]; coop.add_i32(&adj_off_[ed.v + 1], 1); coop.add_i32(&adj_off_[ed.w + 1], 1); }
        sync();
        {
            int run = 0;                                            // adj_off[v + 1] = degrees up to and including v
            for (int v0 = 0; v0 < n; v0 += nl) {
                const int v = v0 + lane;
                const int c = v < n ? adj_off_[v + 1] : 0;
                int total = 0;
                const int ex = coop.excl_scan(c, total);
                if (v < n) { adj_off_[v + 1] = run + ex + c; cur_[v] = run + ex; }
                run += total;
            }
        }
        sync();
        for (int v = lane; v < n; v += nl) tmp_[v] = 0x7fffffff;
        sync();
        for (int e0 = 0; e0 < m; e0 += nl) {
            const int e = e0 + lane;
            const bool ok = e < m;
            const SqMatchEdge ed = E_[ok ? e : 0];
            // ranks among the chunk's ends that name the same vertex, in list order (v before w within an edge): the
            // unplaced end with the smallest id wins a round (an LDS minimum: no order among the lanes assumed)
            int rv = 0, rw = 0;
            bool pv = !ok, pw = !ok;
            for (int r = 0;; r++) {
                if (!pv) coop.min_i32_at(&tmp_[ed.v], 2 * lane);
                if (!pw) coop.min_i32_at(&tmp_[ed.w], 2 * lane + 1);
                sync();
                const bool wv = !pv && tmp_[ed.v] == 2 * lane, ww = !pw && tmp_[ed.w] == 2 * lane + 1;
                sync();
                if (wv) { rv = r; pv = true; tmp_[ed.v] = 0x7fffffff; }
                if (ww) { rw = r; pw = true; tmp_[ed.w] = 0x7fffffff; }
                sync();
                if (!coop.any(!pv || !pw)) break;
            }
            if (ok) {
                const int qv = cur_[ed.v] + rv, qw = cur_[ed.w] + rw;
                adj_[qv] = 2 * e;     adjv_[qv] = ed.w; adjw_[qv] = ed.weight;
                adj_[qw] = 2 * e + 1; adjv_[qw] = ed.v; adjw_[qw] = ed.weight;
            }
            sync();
            if (ok) { coop.add_i32(&cur_[ed.v], 1); coop.add_i32(&cur_[ed.w], 1); }
            sync();
        }
    }

    // FAST: the edges and all state arrays live in ONE LDS buffer whose address the caller passes as `fast0`
    // (and as `origin`, its generic address).  The hot loops then address the arrays as fast0 + offset, which lets
    // the compiler prove the address space and emit ds_read/ds_write instead of flat loads through the LDS aperture.
    // FAST 1: edges, hot, cold and edge parts are one LDS buffer; FAST 2: only the hot part is in LDS (at fast0),
    // everything else in global memory; FAST 0: everything generic.
    template <int FAST, class Sync, class Coop>
    SQ_HD void run(int lane, int nl, Sync sync, Coop coop, char *fast0)
    {
        const int N2 = 2 * n + 2;
#ifdef SQ_MWM_PROF
        const long long _tr0 = wall_clock64();
#endif
        for (int v = lane; v < n; v += nl) { SQ_LQ(mate)[v] = -1; SQ_LQ(mate_de)[v] = -1; SQ_LQ(mord)[v] = -1; SQ_LP(inblossom)[v] = v; }
        if (lane == 0) mord_n = 0;
        for (int x = lane; x < N2; x += nl) {
            SQ_LP(label)[x] = 0; SQ_LP(labeledge)[x] = -1; SQ_LQ(parent)[x] = -1; SQ_LQ(base)[x] = x < n ? x : -1; SQ_LP(bestedge)[x] = -1; SQ_LP(bslack)[x] = SQ_BINF;
            SQ_LQ(bdual)[x] = 0; SQ_LQ(mbe_cnt)[x] = -1; SQ_LQ(mbe_off)[x] = 0; SQ_LQ(beto)[x] = -1; SQ_LQ(nchild)[x] = 0; SQ_LQ(first)[x] = -1;
        }
        if (lane == 0) {
            nlive = 0; nfree = 0;
            for (int b = 2 * n - 1; b >= n; b--) SQ_LQ(freeb)[nfree++] = b;
        }
        if (lane == 0) { stat_pass = 0; stat_event = 0; }
        sync();
        if (n == 0) return;
        if (n >= (1 << 25)) { if (lane == 0) error = 5; sync(); return; }   // (the dual step packs type and entry index into 32 bits)
        {
            double negmax = 0;                                  // max over the edges, shared among the lanes (weights >= 0)
            for (int e = lane; e < m; e += nl) { const SqMatchEdge ed = SQ_LQ(E)[e]; if (ed.v != ed.w && -ed.weight < negmax) negmax = -ed.weight; }
            coop.min_plain(negmax);
            const double maxweight = -negmax;
            for (int v = lane; v < n; v += nl) SQ_LP(dualvar)[v] = maxweight;
        }
        sync();
        // the hot loop works on register copies of the array bases: `this` lives in LDS, and every byte store
        // (label, allowedge) would otherwise force the compiler to reload the pointer members
        const SqMatchEdge *const E_ = SQ_LQ(E);
        const int *const adj_ = SQ_LQ(adj), *const adj_off_ = SQ_LP(adj_off), *const inblossom_ = SQ_LP(inblossom);
        const int *const queue_ = SQ_LP(queue), *const adjv_ = SQ_LQ(adjv);
        const double *const adjw_ = SQ_LQ(adjw);
        int *const labeledge_ = SQ_LP(labeledge), *const bestedge_ = SQ_LP(bestedge);
        int8_t *const label_ = SQ_LP(label);
        uint8_t *const allow_ = SQ_LP(allow);
        double *const dualvar_ = SQ_LP(dualvar), *const bdual_ = SQ_LQ(bdual), *const bslack_ = SQ_LP(bslack);
        const int *const parent_ = SQ_LQ(parent), *const live_ = SQ_LQ(live);
        const int qcap_r = qcap;
#ifdef SQ_MWM_PROF
        long long _tp = wall_clock64();
        const long long _trinit = _tp - _tr0;
        if (lane == 0) for (int k = 0; k < 8; k++) { pt[k] = 0; pc[k] = 0; }
#endif
        int npass = 0, nevent = 0;                              // (wave-uniform registers; published at the end)
#ifdef SQ_MWM_PROF2
        long long p2_cls = 0, p2_app = 0, p2_evt = 0, p2_upd = 0, p2_ini = 0, p2_all = clock64(); int p2_c1 = 0, p2_cn = 0;
#endif
        for (;;) {                                              // stages
#ifdef SQ_MWM_PROF
            if (lane == 0) pc[3]++;
#endif
            // (label and allowedge are cleared as 32-bit words: init() pads both to words)
            for (int x = lane; x < (N2 + 3) / 4; x += nl) reinterpret_cast<uint32_t *>(label_)[x] = 0;
            for (int x = lane; x < N2; x += nl) { labeledge_[x] = -1; bestedge_[x] = -1; bslack_[x] = SQ_BINF; }
            for (int k = lane; k < nlive; k += nl) SQ_LQ(mbe_cnt)[live_[k]] = -1;
            for (int e = lane; e < (m + 3) / 4; e += nl) reinterpret_cast<uint32_t *>(allow_)[e] = 0;
            sync();
            // every free vertex becomes an S-vertex: assignLabel<FAST>(v, 1, None) in vertex order.  Free vertices sit in
            // distinct top-level blossoms, so the label writes are independent; only the queue order is sequential
            // (prefix sum over leaf counts; the leaves of a non-trivial blossom are listed by lane 0).
            {
                if (lane == 0) { pool_n = 0; f_augmented = 0; }
                int qbase = 0;
                const int *const mate_ = SQ_LQ(mate), *const nleaf_ = SQ_LQ(nleaf);
                int *const queue_w = SQ_LP(queue);
                for (int v0 = 0; v0 < n; v0 += nl) {
                    const int v = v0 + lane;
                    const int vc = v < n ? v : 0;
                    const int mt = mate_[vc], b = inblossom_[vc];        // (the loads of one level together)
                    const int lb = label_[b], nlf = nleaf_[b];
                    const bool q = v < n && mt == -1 && lb == 0;
                    const int cnt = q ? (b >= n ? nlf : 1) : 0;
                    int total = 0;
                    const int pos = qbase + coop.excl_scan(cnt, total);
                    const bool fits = pos + cnt <= qcap_r;
                    if (q && !fits) error = 1;
                    if (q && fits) {
                        label_[v] = 1; label_[b] = 1;                    // (labeledge, bestedge: -1, bslack: +inf since the reset above)
                        if (b < n) queue_w[pos] = b;
                    }
                    const int anyb = coop.first_true(q && b >= n, nl);
                    if (anyb < nl) {
                        red_i[lane][0] = q && fits && b >= n ? b : -1; red_i[lane][1] = pos;
                        sync();
                        if (lane == 0 && !error)
                            for (int l = anyb; l < nl; l++) if (red_i[l][0] != -1) leaves<FAST>(red_i[l][0], SQ_LP(queue) + red_i[l][1]);
                        sync();
                    }
                    qbase += total;
                }
                if (lane == 0) qn = qbase;
            }
            sync();
#ifdef SQ_MWM_PROF
                if (lane == 0) { const long long _n = wall_clock64(); pt[5] += _n - _tp; _tp = _n; }
#endif
            for (;;) {                                          // substages
                if (lane == 0) {
#ifdef SQ_MWM_PROF
                    pc[2]++;
#endif
                    f_augmented = 0;
                }
                // ---- queue of S-vertices (LIFO order matters).  Every lane classifies one neighbour of a popped vertex
                // against the current state; the neighbours before the first one that changes shared state (a label
                // assignment, a new blossom, an augmentation) only touch their own w (allowedge, SQ_LP(label)[w],
                // SQ_LP(bestedge)[w]) or compete for SQ_LP(bestedge)[bv] (first strictly smaller slack wins == lexicographic
                // (slack, position) minimum), so they are applied in parallel; the state-changing neighbour is then
                // handled with the sequential code and what is left of its vertex's list is re-classified.
                // the queue length lives in a register of every lane (the queue itself only changes inside the
                // lane-0 sections, which are bracketed by syncs and followed by a reload)
                sync();
                int qn_r = qn;
                bool stopq = f_augmented || error;
                // SEVERAL VERTICES PER PASS.  The queue is a stack that only events push onto, so until an event happens
                // the vertices popped next are the ones below the top: as long as their whole neighbour lists fit the
                // lanes that are left, up to SEGMAX of them are scanned in one pass, each lane classifying one neighbour
                // of "its" vertex against the state before the pass.  Lane order == the sequential scan order, so the
                // neighbours before the first state-changing one are applied together and the state-changing one
                // splits the pass as before.  What the segments of one pass can share, and how the sequential result is
                // kept: the same w from two vertices -- the first lane that labels w (cat 2) wins and silences every
                // later lane on w; best-edge competitors (cat 3 for w, cat 4 for the scanning vertex's blossom, which
                // several segments may share) take the minimum slack through an LDS minimum on the cached slack and the
                // lowest lane among the ties wins (stores issued in reverse segment order: LDS executes a wave's
                // operations in order, the lowest segment's store lands last).  With the state in global memory
                // (FAST 0) a pass stays one vertex.
                constexpr int SEGMAX = FAST ? Coop::kSegMax : 1;
                // A popped vertex with neighbours left (behind an event, or a list longer than the wave) is the first segment
                // of the next pass.  pf_*: what the next pass scans, one entry per lane 0..SEGMAX-1 (vertex, adjacency start,
                // length; -1: none): lane 0 that vertex in progress if there is one, then the top entries of the queue --
                // fetched while the previous pass's own loads were in flight (valid unless that pass had an event).
                bool hascur = false;
                int cv = 0, ca0 = 0, caend = 0;                  // (the vertex in progress, for the reload behind an event)
                int pf_v = 0, pf_a0 = 0, pf_len = -1;
                bool pf_ok = false;
                while (!stopq && (hascur || qn_r > 0)) {
#ifdef SQ_MWM_PROF2
                    long long _q0 = clock64();
#endif
                    if (!pf_ok) {
                        const int idx = qn_r - 1 - lane + (hascur ? 1 : 0);
                        const bool okc = lane < SEGMAX && idx >= 0 && idx < qn_r;
                        pf_v = okc ? queue_[idx] : 0;
                        pf_a0 = adj_off_[pf_v];
                        pf_len = okc ? adj_off_[pf_v + 1] - pf_a0 : -1;
                        if (hascur && lane == 0) { pf_v = cv; pf_a0 = ca0; pf_len = caend - ca0; }
                    }
                    // ---- the segments of this pass (wave-uniform)
                    int L_v[SEGMAX], L_a0[SEGMAX], L_len[SEGMAX], off[SEGMAX];
                    for (int k = 0; k < SEGMAX; k++) { L_v[k] = coop.readlane(pf_v, k); L_a0[k] = coop.readlane(pf_a0, k); L_len[k] = coop.readlane(pf_len, k); }
                    const bool partial = L_len[0] > nl;                 // a list longer than the wave goes alone, a chunk per pass
                    int total = partial ? nl : L_len[0], nseg = 1;
                    off[0] = 0;
                    {
                        bool open = !partial;
                        for (int k = 1; k < SEGMAX; k++) {
                            off[k] = total;
                            if (open && L_len[k] >= 0 && total + L_len[k] <= nl) { total += L_len[k]; nseg++; }
                            else { open = false; off[k] = nl; }
                        }
                        // (off[k] of a segment that is not taken: past every lane)
                    }
                    const int npop = nseg - (hascur ? 1 : 0);
                    const int qn_after = qn_r - npop;
                    // the entries below the ones this pass pops, for the next pass (behind the rest of a long list): branch-free
                    // (clamped slots), and the two dependent reads are issued WITH the first two levels of the classification's
                    // own loads below
                    const int nx_idx = qn_after - 1 - lane + (partial ? 1 : 0);
                    const bool nx_ok = lane < SEGMAX && nx_idx >= 0 && nx_idx < qn_after;
                    const int nx_q = queue_[nx_idx > 0 ? nx_idx : 0];
                    npass++;
#ifdef SQ_MWM_PROF
                    if (lane == 0) { pc[1] += npop; pc[0] += total; pc[5]++; }
#endif
                    // ---- this lane's vertex and neighbour slot
                    const bool live = lane < total;
                    int seg = 0, v = L_v[0], abase = L_a0[0];
                    for (int k = 1; k < SEGMAX; k++)
                        if (live && lane >= off[k]) { seg = k; v = L_v[k]; abase = L_a0[k] - off[k]; }
                    const int a = live ? abase + lane : L_a0[0];
                    // Branch-free classification: the loads of one dependency level are issued together
                    // (speculatively for lanes past the end, on a clamped slot), three LDS round trips in all.
                    const int bv = inblossom_[v];
                    const double dv = dualvar_[v];
                    const int de = adj_[a], w = adjv_[a];
                    const double wt = adjw_[a];
                    const int nx_v = nx_q < 0 ? 0 : (nx_q >= n ? n - 1 : nx_q);   // (a slot that was never written: any valid vertex)
                    const int nx_a0 = adj_off_[nx_v], nx_a1 = adj_off_[nx_v + 1];
                    const int be_bv = bestedge_[bv];                     // the competitor for SQ_LP(bestedge)[bv]
                    const int bw = inblossom_[w];
                    const double dw = dualvar_[w];
                    const int lw = label_[w];
                    const bool was_allowed = allow_[de >> 1] != 0;
                    const int be_w = bestedge_[w];
                    const int lbw = label_[bw];
                    const double s_bew = bslack_[w];                      // slack<FAST>(bestedge[w]) (+inf when there is none)
                    const double s_bebv = bslack_[bv];
                    const double ks = dv + dw - 2 * wt;                  // slack<FAST>(de): SQ_LP(dualvar)[v] + SQ_LP(dualvar)[w] - 2 weight
                    const bool cons = live && w != v && bw != bv;       // :  `if w == v: continue`, same blossom: continue
                    const bool becomes = cons && !was_allowed && ks <= 0;
                    const bool allowed = was_allowed || becomes;
                    int cat = 0;                                         // 0 none, 1 event, 2 SQ_LP(label)[w] := T, 3 SQ_LP(bestedge)[w], 4 SQ_LP(bestedge)[bv]
                    if (cons) {
                        if (allowed) cat = (lbw == 0 || lbw == 1) ? 1 : (lw == 0 ? 2 : 0);
                        else cat = lbw == 1 ? 4 : (lw == 0 ? 3 : 0);
                    }
#ifdef SQ_MWM_PROF2
                    asm volatile("" :: "v"(cat));
                    const long long _q1 = clock64();
                    p2_cls += _q1 - _q0;
#endif
                    const int f = coop.first_true(cat == 1, nl);        // first state-changing neighbour of the pass
                    const bool act = lane < f && live;
                    if (act && becomes) allow_[de >> 1] = 1;
                    if (nseg == 1) {
                        if (act) {
                            if (cat == 2) { label_[w] = 2; labeledge_[w] = de; }
                            else if (cat == 3) { if (be_w == -1 || ks < s_bew) { bestedge_[w] = de; bslack_[w] = ks; } }
                        }
                        // competitors for SQ_LP(bestedge)[bv]: sequentially "first strictly smaller slack wins", i.e. the
                        // lexicographic (slack, position) minimum among the neighbours that beat the CURRENT best.
                        // SQ_LP(bestedge)[bv] only improves during a stage, so after the first passes usually nobody
                        // does and the pass ends here; one competitor writes directly; several are reduced.
                        const bool comp = act && cat == 4 && (be_bv == -1 || ks < s_bebv);
                        int cfirst;
                        const int ccount = coop.count_true(comp, cfirst);
#ifdef SQ_MWM_PROF2
                        p2_c1 += ccount == 1; p2_cn += ccount > 1;
#endif
                        if (ccount == 1) {
                            if (lane == cfirst) { bestedge_[bv] = de; bslack_[bv] = ks; }
                        } else if (ccount > 1) {
                            // positions grow with the lane: the winner is the lowest lane that holds the minimum
                            double mv = comp ? ks : 1e300;
                            coop.min_plain(mv, total);                  // (lanes past the list hold 1e300)
                            if (lane == coop.first_true(comp && ks == mv, nl)) { bestedge_[bv] = de; bslack_[bv] = ks; }
                        }
                    } else if constexpr (SEGMAX > 1) {
                        const bool c2 = act && cat == 2;
                        bool c3 = act && cat == 3 && ks < s_bew;
                        const bool c4 = act && cat == 4 && ks < s_bebv;
                        if (coop.any(c2)) {
                            // the first lane in scan order labels w: a marker with the segment first (lowest segment lands
                            // last), which silences the best-edge competitors of LATER segments on the same w, then T
                            // (SQ_ORDERED: a compiler barrier -- without it the stores of the segments, whose lanes are disjoint,
                            // may be merged into one instruction, and the order among its lanes is the hardware's)
                            for (int s = SEGMAX - 1; s >= 0; s--) {
                                if (c2 && seg == s) { label_[w] = (int8_t)(8 + s); labeledge_[w] = de; }
                                SQ_ORDERED();
                            }
                            if (coop.any(c3 && lbw == 2)) {
                                const int lnow = c3 ? label_[w] : 0;
                                if (lnow >= 8 && lnow - 8 < seg) c3 = false;
                            }
                            if (c2) label_[w] = 2;
                        }
                        const bool comp = c3 || c4;
#ifdef SQ_MWM_PROF2
                        p2_cn += coop.any(comp);
#endif
                        if (coop.any(comp)) {
                            double *const slot = &bslack_[c4 ? bv : w];
                            if (comp) coop.min_pos_f64(slot, ks);       // slacks of edges that are not allowed: > 0
                            const double now = comp ? *slot : 0.0;              // (behind the atomic: the minimum over all competitors)
                            const bool tie = comp && now == ks;
                            for (int s = SEGMAX - 1; s >= 0; s--) {
                                if (s >= nseg) continue;
                                const bool ts = tie && seg == s;
                                const int l4 = coop.first_true(ts && c4, nl);      // one blossom per segment: its lowest lane
                                if (ts && (c3 || lane == l4)) bestedge_[c4 ? bv : w] = de;
                                SQ_ORDERED();
                            }
                        }
                    }
#ifdef SQ_MWM_PROF2
                    _q0 = clock64();
                    p2_app += _q0 - _q1;
#endif
                    if (f >= nl) {                                      // no event: everything taken is scanned
                        qn_r = qn_after;
                        hascur = partial;
                        pf_v = nx_v; pf_a0 = nx_a0; pf_len = nx_ok ? nx_a1 - nx_a0 : -1; pf_ok = true;
                        if (partial && lane == 0) { pf_v = L_v[0]; pf_a0 = L_a0[0] + nl; pf_len = L_len[0] - nl; }
                        continue;
                    }
                    // ---- the state-changing neighbour (lane f): the vertices of the segments before its own are done, its
                    // own vertex is popped and goes on as the vertex in progress behind the event
                    const int fseg = coop.readlane(seg, f), fa = coop.readlane(a, f), fv = coop.readlane(v, f);
                    int fend = L_a0[0] + L_len[0];
                    for (int k = 1; k < SEGMAX; k++) if (fseg == k) fend = L_a0[k] + L_len[k];
                    qn_r -= fseg + 1 - (hascur ? 1 : 0);
                    nevent++;
                    if (lane == 0) qn = qn_r;
                    sync();
#ifdef SQ_MWM_PROF
                    long long _te = 0;
                    if (lane == 0) { pc[4]++; _te = wall_clock64(); }
#endif
                    // the most frequent event by far -- an allowed edge to an unlabelled blossom: T for it, S for the mate of its
                    // base -- is run by lane f itself, which holds the edge, w and the allow decision in registers (what the
                    // lanes before it applied cannot touch them: their neighbours are labelled or stay unlabelled)
                    const bool tfast = coop.readlane(lbw == 0 ? 1 : 0, f) != 0;
                    if (tfast) {
                        if (lane == f) { if (becomes) allow_[de >> 1] = 1; assignLabel<FAST>(w, 2, de); }
                    } else if (lane == 0) {                     // the sequential body for that neighbour
                        const int de1 = SQ_LQ(adj)[fa];
                        const int w1 = head<FAST>(de1);
                        const int bv1 = SQ_LP(inblossom)[fv], bw1 = SQ_LP(inblossom)[w1];
                        if (w1 != fv && bv1 != bw1) {
                            double kslack = 0;
                            if (!SQ_LP(allow)[de1 >> 1]) {
                                kslack = slack<FAST>(de1);
                                if (kslack <= 0) SQ_LP(allow)[de1 >> 1] = 1;
                            }
                            if (allow[de1 >> 1]) {
                                if (label[bw1] == 0) assignLabel<FAST>(w1, 2, de1);
                                else if (label[bw1] == 1) {
                                    const int bs = scanBlossom<FAST>(fv, w1);
                                    if (bs != -1) addBlossom<FAST>(bs, de1);
                                    else { augmentMatching<FAST>(de1); f_augmented = 1; }
                                } else if (label[w1] == 0) {
                                    SQ_LP(label)[w1] = 2; SQ_LP(labeledge)[w1] = de1;
                                }
                            } else if (label[bw1] == 1) {
                                if (bestedge[bv1] == -1 || kslack < slack<FAST>(bestedge[bv1])) { SQ_LP(bestedge)[bv1] = de1; SQ_LP(bslack)[bv1] = kslack; }
                            } else if (label[w1] == 0) {
                                if (bestedge[w1] == -1 || kslack < slack<FAST>(bestedge[w1])) { SQ_LP(bestedge)[w1] = de1; SQ_LP(bslack)[w1] = kslack; }
                            }
                        }
                    }
                    sync();
#ifdef SQ_MWM_PROF
                    if (lane == 0) pt[7] += wall_clock64() - _te;
#endif
                    qn_r = qn;
                    stopq = f_augmented || error;
#ifdef SQ_MWM_PROF2
                    { const long long _n = clock64(); p2_evt += _n - _q0; _q0 = _n; }
#endif
                    hascur = fa + 1 < fend; cv = fv; ca0 = fa + 1; caend = fend;
                    pf_ok = false;
                }
                if (lane == 0) qn = qn_r;
                sync();
#ifdef SQ_MWM_PROF
                if (lane == 0) { const long long _n = wall_clock64(); pt[0] += _n - _tp; _tp = _n; }
#endif
                if (f_augmented || error) break;
                // ---- the dual step: the four delta candidates, the dual update and the refresh of the cached slacks in ONE
                // sweep over the entries k = 0 .. n + nlive (vertices, then live blossoms in creation order == the order of
                // networkx's blossomparent / blossomdual dicts).  A lane takes three entries per trip and issues the loads of
                // one dependency level for all three together (entry -> its state -> its blossom's label and its best edge's
                // ends): four LDS round trips per trip.  (Until round 4: five loops whose loads the compiler chained behind
                // branches, ~50 dependent round trips per substage, 0.5 of the critical graph's 3.0 ms.)
                // The sequential rule -- delta starts as min(dualvar) and a later candidate replaces it only when strictly
                // smaller -- is the lexicographic minimum of (value, type, iteration index): one f64 and one i32 reduction.
                {
                    const int nent = n + nlive;
                    const int KEYB = 26;                                 // key = type << 26 | index  (n < 2^25: see run()'s head)
                    struct Ent { int x, par, lab, be, lbv; double bs, dual; SqMatchEdge ed; bool valid, isb; int k; };
                    auto load3 = [&](int k0, Ent *en) {
                        int lv[3];
                        for (int j = 0; j < 3; j++) {
                            const int k = k0 + j * nl + lane;
                            en[j].k = k; en[j].valid = k < nent; en[j].isb = k >= n;
                            const int kb = k - n;
                            lv[j] = live_[kb < 0 ? 0 : (kb < nlive ? kb : 0)];
                        }
                        for (int j = 0; j < 3; j++) en[j].x = en[j].valid ? (en[j].isb ? lv[j] : en[j].k) : 0;
                        int inb[3]; double dv[3], db[3];
                        for (int j = 0; j < 3; j++) {
                            const int x = en[j].x;
                            en[j].par = parent_[x]; en[j].lab = label_[x]; en[j].be = bestedge_[x]; en[j].bs = bslack_[x];
                            inb[j] = inblossom_[en[j].isb ? 0 : x]; dv[j] = dualvar_[en[j].isb ? 0 : x]; db[j] = bdual_[x];
                        }
                        for (int j = 0; j < 3; j++) {
                            en[j].dual = en[j].isb ? db[j] : dv[j];
                            en[j].lbv = label_[inb[j]];
                            en[j].ed = E_[en[j].be < 0 ? 0 : en[j].be >> 1];
                        }
                    };
                    double best = 1e300; int bkey = 0x7fffffff;
                    auto consider = [&](bool c, double val, int key) {
                        if (c && (val < best || (val == best && key < bkey))) { best = val; bkey = key; }
                    };
                    auto candidates = [&](const Ent *en) {
                        for (int j = 0; j < 3; j++) {
                            const Ent &e = en[j];
                            const bool vtx = e.valid && !e.isb, blo = e.valid && e.isb;
                            consider(vtx, e.dual, 1 << KEYB);
                            consider(vtx && e.lbv == 0 && e.be != -1, e.bs, (2 << KEYB) | e.k);
                            consider(e.valid && e.par == -1 && e.lab == 1 && e.be != -1, e.bs / 2.0, (3 << KEYB) | e.k);
                            consider(blo && e.par == -1 && e.lab == 2, e.dual, (4 << KEYB) | (e.k - n));
                        }
                    };
                    auto update = [&](const Ent *en, double delta) {
                        for (int j = 0; j < 3; j++) {
                            const Ent &e = en[j];
                            if (!e.valid) continue;
                            if (!e.isb) { if (e.lbv == 1) dualvar_[e.x] = e.dual - delta; else if (e.lbv == 2) dualvar_[e.x] = e.dual + delta; }
                            else if (e.par == -1) { if (e.lab == 1) bdual_[e.x] = e.dual + delta; else if (e.lab == 2) bdual_[e.x] = e.dual - delta; }
                        }
                    };
                    // the duals moved: the cached slacks of the best edges follow (slack(v, w) with the NEW duals: a + b == b + a)
                    auto refresh = [&](const Ent *en) {
                        double sl[3];
                        for (int j = 0; j < 3; j++) sl[j] = dualvar_[en[j].ed.v] + dualvar_[en[j].ed.w] - 2 * en[j].ed.weight;
                        for (int j = 0; j < 3; j++) if (en[j].valid && en[j].be != -1) bslack_[en[j].x] = sl[j];
                    };
                    auto reduce = [&](double &delta, int &key) {
                        double mv = best;
                        coop.min_plain(mv);
                        int kk = best == mv ? bkey : 0x7fffffff;
                        coop.min_i32(kk);
                        delta = mv; key = kk;
                    };
                    Ent en[3];
                    double delta; int key;
                    if (nent <= 3 * nl) {                                // everything in registers between the phases
                        load3(0, en);
                        candidates(en);
                        reduce(delta, key);
                        sync();
                        update(en, delta);
                        sync();
                        refresh(en);
                    } else {
                        for (int k0 = 0; k0 < nent; k0 += 3 * nl) { load3(k0, en); candidates(en); }
                        reduce(delta, key);
                        sync();
                        for (int k0 = 0; k0 < nent; k0 += 3 * nl) { load3(k0, en); update(en, delta); }
                        sync();
                        for (int k0 = 0; k0 < nent; k0 += 3 * nl) { load3(k0, en); refresh(en); }
                    }
                    if (lane == 0) {
                        const int deltatype = key >> KEYB, idx = key & ((1 << KEYB) - 1);
                        int deltaedge = -1, deltablossom = -1;
                        if (deltatype == 2) deltaedge = bestedge_[idx];
                        else if (deltatype == 3) deltaedge = bestedge_[idx < n ? idx : live_[idx - n]];
                        else if (deltatype == 4) deltablossom = live_[idx];
                        red_k[0] = deltatype; red_k[1] = deltaedge; red_k[2] = deltablossom;
                    }
                }
                sync();
#ifdef SQ_MWM_PROF
                if (lane == 0) { const long long _n = wall_clock64(); pt[3] += _n - _tp; _tp = _n; }
#endif
                if (lane == 0) {
                    const int deltatype = red_k[0], deltaedge = red_k[1], deltablossom = red_k[2];
                    f_stop = 0;
                    if (deltatype == 1) f_stop = 1;
                    else if (deltatype == 2 || deltatype == 3) { SQ_LP(allow)[deltaedge >> 1] = 1; qpush<FAST>(tail<FAST>(deltaedge)); }
                    else expandBlossom<FAST>(deltablossom, false);
                }
                sync();
#ifdef SQ_MWM_PROF
                if (lane == 0) { const long long _n = wall_clock64(); pt[4] += _n - _tp; _tp = _n; }
#endif
                if (f_stop || error) break;
            }
            if (!f_augmented || error) break;
            // end of stage: expand S-blossoms with zero dual (snapshot of the dict keys)
            if (lane == 0) {
                int snap = nlive;
                for (int k = 0; k < snap; k++) SQ_LQ(tmp_path)[k] = SQ_LQ(live)[k];
                for (int k = 0; k < snap; k++) {
                    const int b = SQ_LQ(tmp_path)[k];
                    bool alive = false;
                    for (int q = 0; q < nlive; q++) if (live[q] == b) { alive = true; break; }
                    if (!alive) continue;
                    if (parent[b] == -1 && SQ_LP(label)[b] == 1 && SQ_LQ(bdual)[b] == 0) expandBlossom<FAST>(b, true);
                }
            }
            sync();
#ifdef SQ_MWM_PROF
                if (lane == 0) { const long long _n = wall_clock64(); pt[6] += _n - _tp; _tp = _n; }
#endif
            if (error) break;
        }
        if (lane == 0) { stat_pass = npass; stat_event = nevent; }
        sync();
#ifdef SQ_MWM_PROF2
#ifdef __HIP_DEVICE_COMPILE__
        if (lane == 0 && n >= 140)
            printf("mwm2 n=%d m=%d passes=%d events=%d comp1=%d compN=%d | cycles: total %lld classify %lld apply %lld events %lld\n",
                   n, m, npass, nevent, p2_c1, p2_cn, (long long)(clock64() - p2_all), p2_cls, p2_app, p2_evt);
#endif
#endif
#ifdef SQ_MWM_PROF
#ifdef __HIP_DEVICE_COMPILE__
        if (lane == 0 && n >= 140)
            printf("mwm n=%d m=%d stages=%lld substages=%lld popped=%lld visits=%lld events=%lld passes=%lld | us: events %.0f queue %.0f classify %.0f apply %.0f update %.0f act %.0f stageinit %.0f endstage %.0f runinit %.0f\n",
                   n, m, pc[3], pc[2], pc[1], pc[0], pc[4], pc[5], pt[7] * 0.01, pt[0] * 0.01, pt[1] * 0.01, pt[2] * 0.01, pt[3] * 0.01, pt[4] * 0.01, pt[5] * 0.01, pt[6] * 0.01, _trinit * 0.01);
#endif
#endif
    }

    SQ_HD void run()
    {
        run<0>(0, 1, [] {}, SqCoopSingle(), nullptr);
    }
#undef SQ_LP
#undef SQ_LQ
};
