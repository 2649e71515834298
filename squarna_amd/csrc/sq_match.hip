// sq_match.hip -- a-8 / a-9 / Nussinov on device: the matching step of RunAlgo
// (SQRNdbnseq.py:548-595) for the paramsets whose `algorithms` are H, E or N.
//
//   sq_lsap_kernel      Hungarian (SQRNalgos.py:113-135).  The reference calls
//                       scipy.optimize.linear_sum_assignment (scipy 1.15.3, un-vendored C++:
//                       Crouse 2016 shortest augmenting path).  Restated step by step -- same
//                       column order, same tie-break towards unassigned columns -- so the
//                       assignment is the one scipy returns, not just an optimal one.
//                       One wave per job; the column scan of every augmentation step is
//                       parallel over the lanes, everything else is scalar work on lane 0.
//   sq_nussinov_kernel  Nussinov DP + BackTrack (SQRNalgos.py:6-93): anti-diagonal wavefront,
//                       one block per job, fp64, first-best-k tie rule.
//   sq_mwm_kernel       Edmonds (SQRNalgos.py:96-110): networkx 3.4.2 max_weight_matching
//                       restated in sq_blossom.h, one thread per job.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include "sq_match.h"
#include "sq_hostflag.h"
#include <vector>
#include <algorithm>
#include "sq_blossom.h"

// ------------------------------------------------------------------------------------
// LSAP: one wave per job.  Layout of the job's scratch (doubles then ints), nc = n:
//   cost[n*n] u[n] v[n] spc[n] | path[n] col4row[n] row4col[n] remaining[n] | SR[n] SC[n] (bytes)
// ------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(64) void sq_lsap_kernel(const SqMatchJob *jobs, const SqMatchEdge *edges,
                                                                char *scratch, int32_t *col4row_out, int lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) char lsap_lds[];
    const SqMatchJob jb = jobs[blockIdx.x];
    const int n = jb.n, lane = threadIdx.x;
    if (n <= 0) return;
    double *cost = reinterpret_cast<double *>(scratch + jb.scratch_off);
    double *u = cost + (size_t)n * n;
    // the per-row/column vectors are touched in every step of the augmenting path: LDS when they fit
    const size_t vec_bytes = sq_lsap_vec_bytes(n);
    const bool vec_lds = vec_bytes + 64 <= (size_t)lds_bytes;
    if (vec_lds) u = reinterpret_cast<double *>(lsap_lds);
    double *v = u + n, *spc = v + n;
    int32_t *path = reinterpret_cast<int32_t *>(spc + n);
    int32_t *col4row = path + n, *row4col = col4row + n, *remaining = row4col + n;
    uint8_t *SR = reinterpret_cast<uint8_t *>(remaining + n), *SC = SR + n;
    // the cost matrix is zero except for the 2m stem cells: when it fits, LDS holds it in sparse form (sq_match.h) -- a step
    // of the shortest-path scan never leaves the CU, and a zero cell costs one mask word
    const int m = jb.nedges;
    const int nw = (n + 31) >> 5;
    const bool mat_lds = vec_lds && m < 32767 && vec_bytes + sq_lsap_sparse_bytes(n, m) + 64 <= (size_t)lds_bytes;
    double *wts = reinterpret_cast<double *>(lsap_lds + vec_bytes);
    uint32_t *mask = reinterpret_cast<uint32_t *>(wts + m);                                   // [n][nw] non-zero columns of a row
    uint16_t *rowstart = reinterpret_cast<uint16_t *>(mask + (size_t)n * nw);                 // [n + 1] into cids
    uint16_t *pre = rowstart + ((n + 1 + 1) & ~1);                                            // [n][nw] cells of the row in earlier words
    uint16_t *cids = pre + (((size_t)n * nw + 1) & ~(size_t)1);                               // [2m] edge ids, row by row in column order

    // mat = zeros; mat[v,w] = mat[w,v] = -(score**power)   (SQRNalgos.py:119-123)
    if (mat_lds) {
        for (int q = lane; q < n * nw; q += 64) mask[q] = 0u;
    } else {
        for (size_t q = lane; q < (size_t)n * n; q += 64) cost[q] = 0.0;
    }
    for (int q = lane; q < n; q += 64) { u[q] = 0.0; v[q] = 0.0; path[q] = -1; col4row[q] = -1; row4col[q] = -1; }
    __syncthreads();
    // (the cells of the stems are distinct, so the order of these writes does not matter)
    if (mat_lds) {
        for (int e = lane; e < m; e += 64) {
            const SqMatchEdge ed = edges[jb.edge_off + e];
            wts[e] = ed.weight;
            atomicOr(&mask[ed.v * nw + (ed.w >> 5)], 1u << (ed.w & 31));
            atomicOr(&mask[ed.w * nw + (ed.v >> 5)], 1u << (ed.v & 31));
        }
        __syncthreads();
        // per-word prefix counts of every row, the rows' starts (a wave scan over chunks of rows)
        const int per = (n + 63) / 64, r0 = min(lane * per, n), r1 = min(r0 + per, n);
        int mine = 0;
        for (int r = r0; r < r1; r++) {
            int acc = 0;
            for (int w = 0; w < nw; w++) { pre[r * nw + w] = (uint16_t)acc; acc += __popc(mask[r * nw + w]); }
            mine += acc;
        }
        int inc = mine;
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(inc, off); if (lane >= off) inc += y; }
        int run = inc - mine;
        for (int r = r0; r < r1; r++) {
            rowstart[r] = (uint16_t)run;
            run += (int)pre[r * nw + nw - 1] + __popc(mask[r * nw + nw - 1]);
        }
        if (lane == 63) rowstart[n] = (uint16_t)inc;
        __syncthreads();
        for (int e = lane; e < m; e += 64) {
            const SqMatchEdge ed = edges[jb.edge_off + e];
            const int a = ed.v, c = ed.w;
            cids[rowstart[a] + pre[a * nw + (c >> 5)] + __popc(mask[a * nw + (c >> 5)] & ((1u << (c & 31)) - 1u))] = (uint16_t)e;
            cids[rowstart[c] + pre[c * nw + (a >> 5)] + __popc(mask[c * nw + (a >> 5)] & ((1u << (a & 31)) - 1u))] = (uint16_t)e;
        }
    } else {
        for (int e = lane; e < m; e += 64) {
            const SqMatchEdge ed = edges[jb.edge_off + e];
            cost[(size_t)ed.v * n + ed.w] = -ed.weight;
            cost[(size_t)ed.w * n + ed.v] = -ed.weight;
        }
    }
    __syncthreads();

    // The state of a path -- the row in hand, the columns left, the sink, the distance -- lives in registers: every lane
    // takes lane 0's decision from the same broadcast reads (until late round 4 it went through four words of LDS and two
    // barriers per step).  The scan of a step takes up to three columns per lane with the loads of one dependency level
    // issued together (remaining -> mask word, v, shortest path, owner -> edge id -> weight).
    for (int cur = 0; cur < n; cur++) {
        // ---- augmenting_path(cur)
        for (int q = lane; q < n; q += 64) { remaining[q] = n - q - 1; SR[q] = 0; SC[q] = 0; spc[q] = INFINITY; }
        int i = cur, sink = -1, nrem = n;
        double minval = 0.0;
        __syncthreads();
        while (sink == -1) {
            const double ui = u[i];
            const int rs_i = mat_lds ? (int)rowstart[i] : 0;
            double lowest = INFINITY;
            int first = -1, lastun = -1;
            for (int it0 = 0; it0 < nrem; it0 += 192) {
                int jj[3]; bool have[3];
#pragma unroll
                for (int t = 0; t < 3; t++) { const int it = it0 + 64 * t + lane; have[t] = it < nrem; jj[t] = remaining[have[t] ? it : 0]; }
                uint32_t wd[3]; int pw[3], r4[3]; double vj[3], spj[3], cij[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    const int j = jj[t];
                    vj[t] = v[j]; spj[t] = spc[j]; r4[t] = row4col[j];
                    if (mat_lds) { wd[t] = mask[i * nw + (j >> 5)]; pw[t] = (int)pre[i * nw + (j >> 5)]; cij[t] = 0.0; }
                    else { wd[t] = 0u; pw[t] = 0; cij[t] = cost[(size_t)i * n + j]; }
                }
                if (mat_lds) {
                    int cid[3];
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        const int j = jj[t];
                        const bool nz = (wd[t] >> (j & 31)) & 1u;
                        cid[t] = nz ? (int)cids[rs_i + pw[t] + __popc(wd[t] & ((1u << (j & 31)) - 1u))] : -1;
                    }
#pragma unroll
                    for (int t = 0; t < 3; t++) cij[t] = cid[t] >= 0 ? -wts[cid[t]] : 0.0;
                }
#pragma unroll
                for (int t = 0; t < 3; t++) {                       // (a lane's columns in scan order)
                    if (!have[t]) continue;
                    const int it = it0 + 64 * t + lane, j = jj[t];
                    const double r = minval + cij[t] - ui - vj[t];
                    double sp = spj[t];
                    if (r < sp) { path[j] = i; spc[j] = r; sp = r; }
                    const bool un = r4[t] == -1;
                    if (sp < lowest) { lowest = sp; first = it; lastun = un ? it : -1; }
                    else if (sp == lowest && un) lastun = it;       // sequential rule: later unassigned ties win
                }
            }
            // wave combine == the sequential scan over it = 0..nrem-1
            // (DPP row shifts / broadcasts as in the blossom kernel: the three butterfly reductions over the LDS crossbar they
            // replace -- 30 ds_bpermute per step of the path -- were the longest part of a step)
            const double m = SqCoopWave::wave_min_f64(lowest);
            const int f = SqCoopWave::wave_min_i32((lowest == m && first >= 0) ? first : 0x7fffffff);
            const int lu = -SqCoopWave::wave_min_i32(-((lowest == m) ? lastun : -1));
            minval = m;
            if (m == INFINITY) { if (lane == 0) SR[i] = 1; sink = -2; break; }   // infeasible (cannot happen: finite costs)
            const int index = lu >= 0 ? lu : f;
            const int j = remaining[index], lastj = remaining[nrem - 1];
            const int owner = row4col[j];
            __syncthreads();                                        // (every lane has read what lane 0 overwrites)
            if (lane == 0) { SR[i] = 1; SC[j] = 1; remaining[index] = lastj; }
            nrem--;
            if (owner == -1) sink = j; else i = owner;
            // (the vectors may live in global scratch: lane 0's stores are released to the whole wave before the next step's
            // cross-lane reads, not left to same-wave store-to-load ordering through L1)
            if (!vec_lds) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (!vec_lds) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (sink < 0) break;
        // ---- dual update
        if (lane == 0) u[cur] += minval;
        for (int q = lane; q < n; q += 64) {
            if (SR[q] && q != cur) u[q] += minval - spc[col4row[q]];
            if (SC[q]) v[q] -= minval - spc[q];
        }
        __syncthreads();
        // ---- augment
        if (lane == 0) {
            int j = sink;
            for (;;) {
                const int i2 = path[j];
                row4col[j] = i2;
                const int t = col4row[i2]; col4row[i2] = j; j = t;
                if (i2 == cur) break;
            }
        }
        __syncthreads();
    }
    for (int q = lane; q < n; q += 64) col4row_out[jb.out_off + q] = col4row[q];
}

// ------------------------------------------------------------------------------------
// Nussinov (SQRNalgos.py:44-93).  scratch: S[n*n] D[n*n] (doubles) | K[n*n] (int32) | has[n*n] (bytes)
// out: pairs (k, j) appended to pairs_out[out_off..], count in count_out[job]
// ------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(256) void sq_nussinov_kernel(const SqMatchJob *jobs, const SqMatchEdge *edges,
                                                                     const uint8_t *codes, char *scratch,
                                                                     int32_t *pairs_out, int32_t *count_out, int by_pad)
{
    const SqMatchJob jb = jobs[blockIdx.x];
    const int n = jb.n, tid = threadIdx.x, nthr = blockDim.x;   // 64 .. 256 threads: the launch's longest sequence rounded up to waves
    // (by_pad: the rows are a slice of a table sorted by size, row.pad = the job's index, under which its count is filed)
    int32_t *const my_count = count_out + (by_pad ? jb.pad : (int)blockIdx.x);
    // (n == 1: the column DP writes no cell, BackTrack would read K[0, 0] from reused scratch; the reference returns no pairs)
    if (n < 2) { if (tid == 0) *my_count = 0; return; }
    double *S = reinterpret_cast<double *>(scratch + jb.scratch_off);   // region of n*n doubles: holds the column lists
    double *D = S + (size_t)n * n;
    int32_t *K = reinterpret_cast<int32_t *>(D + (size_t)n * n);
    uint8_t *has = reinterpret_cast<uint8_t *>(K + (size_t)n * n);      // n*n bytes: BackTrack's "already queued" marks
    const uint8_t *cd = codes + jb.pos_off;
    // SCORES is sparse (the cells of the stems): per column j the list of (k, SCORES[(k, j)]) in ascending k,
    // so a cell of the DP only visits the k that can pair with j (:73-74) instead of all of i..j-2
    const int m = jb.nedges;
    double *cs = S;                                                     // [m] scores
    int32_t *ck = reinterpret_cast<int32_t *>(cs + m);                  // [m] row k
    int32_t *col_off = ck + m, *cursor = col_off + (n + 1);            // [n + 1], [n]
    // (the DP writes D and K of every cell above the main diagonal before anything reads them: only the diagonal of D and
    // BackTrack's marks start at zero -- zeroing all three tables was 21 bytes per cell of traffic per job)
    for (size_t q = tid; q < (size_t)n * n; q += nthr) has[q] = 0;
    for (int q = tid; q < n; q += nthr) D[(size_t)q * n + q] = 0.0;
    for (int q = tid; q <= n; q += nthr) col_off[q] = 0;
    __syncthreads();
    for (int e = tid; e < m; e += nthr) atomicAdd(&col_off[edges[jb.edge_off + e].w + 1], 1);
    __syncthreads();
    if (tid == 0) for (int j = 0; j < n; j++) col_off[j + 1] += col_off[j];
    __syncthreads();
    for (int q = tid; q < n; q += nthr) cursor[q] = col_off[q];
    __syncthreads();
    for (int e = tid; e < m; e += nthr) {                               // SCORES[(v,w)] = -stem[2]  (:49)
        const SqMatchEdge ed = edges[jb.edge_off + e];
        const int pos = atomicAdd(&cursor[ed.w], 1);
        ck[pos] = ed.v; cs[pos] = -ed.weight;
    }
    __syncthreads();
    for (int j = tid; j < n; j += nthr) {                               // ascending k inside every column (short lists)
        const int a = col_off[j], b = col_off[j + 1];
        for (int x = a + 1; x < b; x++) {
            const int kk = ck[x]; const double vv = cs[x];
            int y = x - 1;
            while (y >= a && ck[y] > kk) { ck[y + 1] = ck[y]; cs[y + 1] = cs[y]; y--; }
            ck[y + 1] = kk; cs[y + 1] = vv;
        }
    }
    __syncthreads();
    // The DP by COLUMNS (round 4; the reference and the kernel until then: by diagonals, :65-83).  Cell (i, j) reads
    // D[i, k-1] (k <= j - 2), D[k+1, j-1] and D[i, j-1]: columns before j only, so the columns can be taken one after the
    // other with all rows of a column at once -- the same values in any such order.  What the order buys: the list of
    // column j -- the k that can pair with j, :73-74 -- is the same for every row (uniform loads, fetched while the column
    // before is computed), D[k+1, j-1] is one value per list entry, and with D and K stored column-major the rows' reads of
    // D[., k-1] and D[., j-1] are consecutive doubles.  By diagonals every cell chased column list -> k -> two cells of D
    // through L2, three dependent trips per diagonal; now one per column.
    // Dc[j * n + i] = D[i, j], Kc[j * n + i] = K[i, j].
    double *const Dc = D;
    int32_t *const Kc = K;
    for (int j = 1; j < n; j++) {
        const int x0 = col_off[j], x1 = col_off[j + 1];
        const double *const dprevcol = Dc + (size_t)(j - 1) * n;
        for (int i = tid; i < j; i += nthr) {
            int bestk = -1; double best = 1e9;                          // :70
            // k in range(i, j - 1) with (k, j) in SCORES, ascending (first best k, :77); four list entries per trip
            for (int x = x0; x < x1; x += 4) {
                int kk[4]; double d1[4], d2[4], cw[4]; bool ok[4];
#pragma unroll
                for (int t = 0; t < 4; t++) kk[t] = x + t < x1 ? ck[x + t] : 0x7fffffff;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    ok[t] = kk[t] >= i && kk[t] < j - 1;
                    d1[t] = 0.0; d2[t] = 0.0; cw[t] = 0.0;
                    if (ok[t]) {
                        // D[i, k-1] with k == i is numpy's D[i, -1] = D[i, n-1], which no cell has written yet at this point: 0 (:76)
                        if (kk[t] > i) d1[t] = Dc[(size_t)(kk[t] - 1) * n + i];
                        d2[t] = dprevcol[kk[t] + 1];
                        cw[t] = cs[x + t];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; t++)
                    if (ok[t]) {
                        const double sc = d1[t] + d2[t] + cw[t];
                        if (sc < best) { bestk = kk[t]; best = sc; }
                    }
                if (kk[3] >= j - 1) break;
            }
            const double dprev = dprevcol[i];                           // D[i, j-1] (i == j - 1: the diagonal, 0)
            const bool take = best <= dprev;                            // :80-83
            Kc[(size_t)j * n + i] = take ? bestk : -2;
            Dc[(size_t)j * n + i] = take ? best : dprev;
        }
        __syncthreads();
    }
    // BackTrack(0, N-1) (:6-41): level-synchronous set of cells; one thread, O(N) cells
    if (tid == 0) {
        const int minloop = 3;
        int32_t *out = pairs_out + 2 * (size_t)jb.out_off;
        // queue storage: the tail of the job's scratch (two arrays of (i, j) cells)
        int32_t *cur = reinterpret_cast<int32_t *>(has + (((size_t)n * n + 15) & ~(size_t)15)), *nxt = cur + 2 * (size_t)(n + 2);
        uint8_t *inq = has;            // zeroed above
        int qn = 1, np = 0;
        cur[0] = 0; cur[1] = n - 1;
        auto sep = [&](int p) { return cd[p] == 26 || cd[p] == 27; };
        auto anysep = [&](int a, int b) { for (int x = a; x < b; x++) if (x >= 0 && x < n && sep(x)) return true; return false; };
        while (qn) {
            int nn = 0;
            for (int t = 0; t < qn; t++) {
                const int i = cur[2 * t], j = cur[2 * t + 1];
                if (i < 0 || j < 0 || i >= n || j >= n) continue;
                auto push = [&](int a, int b) {
                    if (!inq[(size_t)a * n + b]) { inq[(size_t)a * n + b] = 1; nxt[2 * nn] = a; nxt[2 * nn + 1] = b; nn++; }
                };
                const int kk = Kc[(size_t)j * n + i];
                if (kk != -2) {
                    const int k = kk;
                    if (((k - 1) - i > minloop) || ((k - 1) - i > 0 && anysep(i + 1, k - 1))) push(i, k - 1);
                    if (((j - 1) - (k + 1) > minloop) || ((j - 1) - (k + 1) > 0 && anysep(k + 2, j - 1))) push(k + 1, j - 1);
                    out[2 * np] = k; out[2 * np + 1] = j; np++;
                } else {
                    if (((j - 1) - i > minloop) || ((j - 1) - i > 0 && anysep(i + 1, j - 1))) push(i, j - 1);
                }
            }
            for (int t = 0; t < nn; t++) inq[(size_t)nxt[2 * t] * n + nxt[2 * t + 1]] = 0;
            for (int t = 0; t < 2 * nn; t++) cur[t] = nxt[t];
            qn = nn;
        }
        *my_count = np;
    }
}

// ------------------------------------------------------------------------------------
// Edmonds: one thread per job runs the restated networkx blossom algorithm (sq_blossom.h)
// ------------------------------------------------------------------------------------
// One WAVE per graph, several graphs per block ("bin"): the host packs the graphs of a launch into bins by the LDS
// each one needs (sq_mwm_plan), the block's dynamic LDS is the bin capacity and every wave works in its own slice
// [lds_off, lds_off + lds_bytes) of it.  A launch-wide LDS size for one-graph blocks gives every graph the LDS of the
// largest one (SRtest150: 219 graphs, 8 MB of state in total, 29 MB when each gets 133 KB), and the LDS a blossom block
// holds is LDS the scoring kernels of the batches in flight cannot get.
// The waves of a bin never synchronise with each other: the "barrier" of the algorithm is a wave-local fence.
// job_flags (pinned host memory, may be null): job_flags[row] = stamp once the job's mates are in host memory, so the
// host can filter and rank a sequence while the larger graphs are still being matched.
// mate_out: per job 2n + 2 ints -- mate[0..n), the rank of every vertex's first mate assignment (SqBlossom::mord), then
// the run's scan passes and lane-0 events (measurement: the critical path of the kernel is passes x cycles per pass)
// one graph on one wave: bl (the algorithm object) and wlds (lds_bytes of state room) are the wave's LDS, sync its barrier
template <class Sync>
__device__ __forceinline__ void sq_mwm_one(SqBlossom &bl, char *lds_base, char *wlds, size_t lds_bytes, const SqMatchJob *jp, int row,
                                           const SqMatchEdge *edges, char *scratch, int32_t *mate_out, uint32_t *job_flags,
                                           uint32_t stamp, int lane, Sync wsync)
{
    // The job's flag tells the host that its result block in pinned memory is complete.  A system-scope fence between the
    // data stores and the flag store is NOT enough on this platform: both are posted writes on the way to host memory and
    // the later one was seen to overtake the earlier ones (about one job in 10^5: the collector read a result block that
    // was still arriving -- found by tools/fuzz_options.py, shown by SQ_MWM_POSTHOC=1).  A READ from the same memory
    // cannot pass the posted writes ahead of it, so every lane reads back a word of the block before the flag goes out.
    auto publish = [&]() {
        if (job_flags) {
            const int nw = jp->n > 0 ? 2 * jp->n + 2 : 2;
            sq_host_write_flush(mate_out + jp->out_off + (lane < nw ? nw - 1 - lane : 0));
        }
        // (without per-job flags nobody reads the mates before the kernel has ended -- the finish kernel or the flag kernel
        // behind it on the stream: no fence here; a system-scope fence per graph writes the L2 back, thousands of times a launch)
        wsync();
        if (job_flags && lane == 0) job_flags[row] = stamp;
    };
    const int n = jp->n, m = jp->nedges;
    if (n <= 0) { if (lane == 0) { mate_out[jp->out_off] = 0; mate_out[jp->out_off + 1] = 0; } publish(); return; }
    // The algorithm is a long chain of dependent loads: keep its state in LDS -- all of it with the edge list when
    // the slice holds that (tight capacities), else only the hot part (what every scan pass touches; blossom structure
    // and adjacency stay in global memory); on a capacity overflow rerun the job in global memory.
#ifdef SQ_MWM_PROF
    const long long _c0 = clock64(), _w0 = wall_clock64();
#endif
    const size_t ebytes = ((size_t)m * sizeof(SqMatchEdge) + 15) & ~(size_t)15;
    char *gscratch = scratch + jp->scratch_off;
    const bool all_lds = SqBlossom::scratch_bytes(n, m, 1) + ebytes + 16 <= lds_bytes;
    const bool hot_lds = !all_lds && SqBlossom::hot_bytes(n, m, 2) + 16 <= lds_bytes;
    if (all_lds) {
        SqMatchEdge *le = reinterpret_cast<SqMatchEdge *>(wlds);
        for (int e = lane; e < m; e += 64) le[e] = edges[jp->edge_off + e];
        wsync();
#ifdef SQ_MWM_PROF
        const long long _w1 = wall_clock64();
#endif
        if (lane == 0) { bl.init(n, m, le, wlds + ebytes, 1, false); bl.origin = lds_base; }
        wsync();
#ifdef SQ_MWM_PROF
        const long long _w2 = wall_clock64();
#endif
        bl.template build_csr<1>(lane, 64, wsync, SqCoopWave());
#ifdef SQ_MWM_PROF
        const long long _w3 = wall_clock64();
#endif
        bl.template run<1>(lane, 64, wsync, SqCoopWave(), lds_base);
#ifdef SQ_MWM_PROF
        if (lane == 0 && n >= 140) printf("mwm outside: edge load %.0f us, init %.0f, csr %.0f, run %.0f\n", (_w1 - _w0) * 0.01, (_w2 - _w1) * 0.01, (_w3 - _w2) * 0.01, (wall_clock64() - _w3) * 0.01);
#endif
    } else if (hot_lds) {
        if (lane == 0) {
            char *cold = gscratch;
            bl.init(n, m, edges + jp->edge_off, wlds, 2, false, cold, cold + ((SqBlossom::cold_bytes(n, m, 2) + 15) & ~(size_t)15));
            bl.origin = lds_base;
        }
        wsync();
        bl.template build_csr<2>(lane, 64, wsync, SqCoopWave());
        bl.template run<2>(lane, 64, wsync, SqCoopWave(), lds_base);
    }
    if (all_lds || hot_lds) {
        wsync();
        if (!bl.error) {
            for (int q = lane; q < n; q += 64) { mate_out[jp->out_off + q] = bl.mate[q]; mate_out[jp->out_off + n + q] = bl.mord[q]; }
            if (lane == 0) { mate_out[jp->out_off + 2 * n] = bl.stat_pass; mate_out[jp->out_off + 2 * n + 1] = bl.stat_event; }
#ifdef SQ_MWM_PROF
            if (lane == 0 && n >= 140) {
                const long long dc = clock64() - _c0, dw = wall_clock64() - _w0;
                printf("mwm clock: %lld shader cycles in %.0f us -> %.0f MHz\n", dc, dw * 0.01, (double)dc / (dw * 0.01));
            }
#endif
            publish();
            return;
        }
        wsync();
    }
    if (lane == 0) bl.init(n, m, edges + jp->edge_off, gscratch, 0, false);
    wsync();
    bl.template build_csr<0>(lane, 64, wsync, SqCoopWave());
    // lane 0 runs the order-dependent part; all 64 lanes share the O(n) sweeps of every substage
    bl.template run<0>(lane, 64, wsync, SqCoopWave(), nullptr);
    wsync();
    for (int q = lane; q < n; q += 64) { mate_out[jp->out_off + q] = bl.error ? -2 : bl.mate[q]; mate_out[jp->out_off + n + q] = bl.mord[q]; }
    if (lane == 0) { mate_out[jp->out_off + 2 * n] = bl.stat_pass; mate_out[jp->out_off + 2 * n + 1] = bl.stat_event; }
    publish();
}

// several graphs per block (one per wave), each in its slice of the block's dynamic LDS
// (second launch bound: four waves per SIMD, i.e. at most 128 VGPRs -- the compiler otherwise takes the 248 a 512-thread block
// may have and two graphs per SIMD fill the register file; 24 SRtest150 sets in one batch: Edmonds 16.9 -> 13.0 ms)
extern "C" __global__ __launch_bounds__(512, 4) void sq_mwm_kernel(const SqMatchJob *jobs, const int32_t *bin_head,
                                                                const SqMatchEdge *edges, char *scratch, int32_t *mate_out,
                                                                uint32_t *job_flags, uint32_t stamp)
{
    extern __shared__ __attribute__((aligned(16))) char mwm_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int row = bin_head[blockIdx.x];
    for (int k = 0; k < wave && row >= 0; k++) row = jobs[row].next;
    if (row < 0) return;                                  // this bin holds fewer graphs than the block has waves
    // all lanes of a wave run in lockstep; the fence keeps the compiler (and the memory pipeline) from moving accesses
    // of one lane across the point where another lane's data is needed
    auto wsync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); };
    const SqMatchJob *jp = jobs + row;
    // the wave's slice: the algorithm object (shared by the lanes), then the state
    const size_t hdr = (sizeof(SqBlossom) + 15) & ~(size_t)15;
    SqBlossom &bl = *reinterpret_cast<SqBlossom *>(mwm_lds + jp->lds_off);
    const size_t lds_bytes = (size_t)jp->lds_bytes > hdr ? (size_t)jp->lds_bytes - hdr : 0;
    sq_mwm_one(bl, mwm_lds, mwm_lds + jp->lds_off + hdr, lds_bytes, jp, row, edges, scratch, mate_out, job_flags, stamp, lane, wsync);
}

// one graph per block (one batch alone: every graph on its own CU): the algorithm object at a fixed LDS address, the
// block barrier as the wave's barrier; the block's dynamic LDS is the graph's (rows = blocks)
extern "C" __global__ __launch_bounds__(64) void sq_mwm_single_kernel(const SqMatchJob *jobs, const SqMatchEdge *edges, char *scratch,
                                                                     int32_t *mate_out, int lds_bytes, uint32_t *job_flags,
                                                                     uint32_t stamp)
{
    __shared__ SqBlossom bl;
    extern __shared__ __attribute__((aligned(16))) char mwm_lds[];
    sq_mwm_one(bl, mwm_lds, mwm_lds, (size_t)lds_bytes, jobs + blockIdx.x, (int)blockIdx.x, edges, scratch, mate_out, job_flags, stamp,
               (int)threadIdx.x, [] { __syncthreads(); });
}

extern "C" __global__ void sq_flag_kernel(uint32_t *flag, uint32_t value)
{
    sq_host_write_flush(flag);                           // (the results in pinned memory: earlier kernels)
    *flag = value;
}

// Launch configuration of the three matching kernels (dynamic LDS sized for the largest job of the launch), shared
// by the fold path (sq_algos.hip) and the graph-level C entries (sq_graph.hip).  jobs / edges: what the kernels read
// (pinned host or device memory); h_jobs: the same table on the host; dev_edges: device room for the edge list, used
// only when some blossom job has to run in global memory.  Asynchronous on st.
// Packs the blossom graphs of one launch into bins (blocks) by LDS need, first fit in decreasing order.
//   * a graph gets its whole state + edge list in LDS when that is at most `all_cap`, else only the hot part (the rest in
//     its global scratch), else just the algorithm object (global-memory run);
//   * a bin holds at most `waves` graphs and `cap` bytes; with one batch in flight every graph has its own block (spread
//     over the CUs: latency), with several the bins fill up so that all batches' graphs are resident together and the
//     scoring kernels still find LDS (throughput): waves = ceil(graphs x inflight / 256), cap 150 KB -> 96 KB, all_cap
//     150 KB -> 48 KB from three batches in flight on.  SQ_MWM_BIN_WAVES / SQ_MWM_BIN_BYTES / SQ_MWM_ALL_CAP override.
// Writes lds_off / lds_bytes / next of jobs_rw (the table the kernel reads) and bin_head[0..nbins).
void sq_mwm_plan(const SqMatchJob *h_jobs, SqMatchJob *jobs_rw, int nj, int32_t *bin_head, int inflight, int &nbins, int &waves,
                 size_t &lds, bool &all_in_lds)
{
    static const int env_waves = getenv("SQ_MWM_BIN_WAVES") ? atoi(getenv("SQ_MWM_BIN_WAVES")) : 0;
    static const long env_cap = getenv("SQ_MWM_BIN_BYTES") ? atol(getenv("SQ_MWM_BIN_BYTES")) : 0;
    static const long env_all = getenv("SQ_MWM_ALL_CAP") ? atol(getenv("SQ_MWM_ALL_CAP")) : 0;
    static const bool nolds = getenv("SQ_MWM_NOLDS") != nullptr;
    const size_t hdr = (sizeof(SqBlossom) + 15) & ~(size_t)15;
    const bool many = inflight >= 3;
    // A launch (or the launches in flight together) with more graphs than the chip has room for at one graph per CU is a
    // throughput problem: every graph keeps only the hot part of its state in LDS (~4 KB; the rest in its global scratch,
    // +14 % time per graph) and a block holds four of them in 40 KB, so that ALL graphs are resident at once -- the kernel
    // then lasts as long as its slowest graph -- and the scoring kernels of the greedy rounds still find LDS on every CU.
    // Measured on 24 SRtest150 sets in one batch (4,296 graphs): 11.5 ms with 150 KB bins, 6.6 ms this way.
    // ... and so is a launch of hundreds of LARGE graphs (hot part beyond a crowd's 40 KB bin: 500 vertices and 17,000 edges take
    // 50 KB) beside the rounds of the same batch's pools: 822 such graphs fill the LDS of every CU for 75 ms, during which the
    // round kernel does not run (round 6: 1,000 records of 500 nt under pools of a thousand 313 -> 278 ms with the graphs' state
    // in global memory; at 700 records the same either way, at 500 and fewer the longer Edmonds kernel is what the fold waits for)
    size_t hot_max = 0;
    for (int q = 0; q < nj; q++) hot_max = std::max(hot_max, SqBlossom::hot_bytes(h_jobs[q].n, h_jobs[q].nedges, 2));
    const bool crowd = (long long)nj * std::max(1, inflight) >= 1024 || (nj >= 768 && hot_max + 16 + ((sizeof(SqBlossom) + 15) & ~(size_t)15) > 40 * 1024);
    size_t cap = env_cap > 0 ? (size_t)env_cap : (crowd ? 40 * 1024 : many ? 96 * 1024 : 150 * 1024);
    cap = std::min<size_t>(cap, 150 * 1024);
    const size_t all_cap = env_all > 0 ? (size_t)env_all : (crowd ? 1 : many ? 48 * 1024 : 150 * 1024);
    waves = env_waves > 0 ? env_waves : (crowd ? 4 : (int)(((long long)nj * std::max(1, inflight) + 255) / 256));
    waves = std::max(1, std::min(waves, 8));
    std::vector<size_t> need(nj);
    all_in_lds = true;
    for (int q = 0; q < nj; q++) {
        const int n = h_jobs[q].n, m = h_jobs[q].nedges;
        const size_t full = SqBlossom::scratch_bytes(n, m, 1) + (((size_t)m * sizeof(SqMatchEdge) + 15) & ~(size_t)15) + 16;
        const size_t hot = SqBlossom::hot_bytes(n, m, 2) + 16;
        size_t take = hdr;
        if (n > 0 && !nolds) {
            if (hdr + full <= std::min(all_cap, cap)) take = hdr + full;
            else if (hdr + hot <= cap) take = hdr + hot;
        }
        if (n > 0 && take != hdr + full) all_in_lds = false;
        need[q] = (take + 15) & ~(size_t)15;
    }
    std::vector<int> ord(nj);
    for (int q = 0; q < nj; q++) ord[q] = q;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return need[x] > need[y]; });
    std::vector<size_t> used; std::vector<int> cnt, tail;
    nbins = 0; lds = 0;
    size_t first_open = 0;                               // bins before it are full (by count)
    for (int r = 0; r < nj; r++) {
        const int q = ord[r];
        size_t k = first_open;
        while (k < used.size() && (cnt[k] >= waves || used[k] + need[q] > cap)) k++;
        if (k == used.size()) { used.push_back(0); cnt.push_back(0); tail.push_back(-1); bin_head[k] = q; nbins++; }
        else jobs_rw[tail[k]].next = q;
        jobs_rw[q].lds_off = (int32_t)used[k]; jobs_rw[q].lds_bytes = (int32_t)need[q]; jobs_rw[q].next = -1;
        used[k] += need[q]; cnt[k]++; tail[k] = q;
        lds = std::max(lds, used[k]);
        while (first_open < used.size() && cnt[first_open] >= waves) first_open++;
    }
    lds = (lds + 255) & ~(size_t)255;
    if (getenv("SQ_MWM_DUMP")) {
        size_t tot = 0; int nfull = 0;
        for (int q = 0; q < nj; q++) { tot += need[q]; nfull += need[q] > hdr + SqBlossom::hot_bytes(h_jobs[q].n, h_jobs[q].nedges, 2) + 32; }
        fprintf(stderr, "[mwm plan] %d graphs (inflight %d): %d bins of <= %d waves, %zu B of LDS per block, %zu B needed in all, %d graphs fully in LDS\n",
                nj, inflight, nbins, waves, lds, tot, nfull);
    }
}

int sq_launch_matching(int algo, const SqMatchJob *h_jobs, int nj, const SqMatchJob *jobs, const SqMatchEdge *edges,
                       size_t nedges, SqMatchEdge *dev_edges, char *d_scr, int32_t *out, int32_t *cnt,
                       const uint8_t *codes, uint32_t *job_flags, uint32_t flag_val, hipStream_t st,
                       SqMatchJob *jobs_rw, int32_t *bin_head, int inflight)
{
    int maxn = 0, maxm = 0;
    for (int q = 0; q < nj; q++) { maxn = maxn > h_jobs[q].n ? maxn : h_jobs[q].n; maxm = maxm > h_jobs[q].nedges ? maxm : h_jobs[q].nedges; }
    if (algo == 4) {                                     // SQ_ALGO_H
        // LDS: the row/column vectors and, when it fits, the cost matrix in sparse form (sq_match.h); every job of the launch
        // must find room: the largest n and the largest m may belong to different jobs
        size_t lds = 0;
        for (int q = 0; q < nj; q++) lds = std::max(lds, sq_lsap_lds_bytes(h_jobs[q].n, h_jobs[q].nedges));
        if (lds > 64 * 1024) sq_max_dynamic_lds((const void *)sq_lsap_kernel, 150 * 1024);
        if (lds > 150 * 1024) lds = sq_lsap_vec_bytes(maxn) + 64 < 150 * 1024 ? sq_lsap_vec_bytes(maxn) + 64 : 150 * 1024;   // vectors only
        hipLaunchKernelGGL(sq_lsap_kernel, dim3(nj), dim3(64), lds, st, jobs, edges, d_scr, out, (int)lds);
    } else if (algo == 2) {                              // SQ_ALGO_N
        // an anti-diagonal of the DP has at most n cells: no more waves than that keeps busy (the block holds its wave slots
        // through the single-threaded BackTrack too)
        // (with the chip crowded ONE wave per job: a block of three waves holds three wave slots through every barrier-separated
        // diagonal, and wave slots are what a crowded chip runs out of -- a lane then takes up to three cells of a diagonal)
        static const int env_thr = getenv("SQ_NUSS_THREADS") ? std::max(64, std::min(256, atoi(getenv("SQ_NUSS_THREADS")) / 64 * 64)) : 0;
        // ... and so is a launch of hundreds of LARGE graphs (hot part beyond a crowd's 40 KB bin: 500 vertices and 17,000 edges take
    // 50 KB) beside the rounds of the same batch's pools: 822 such graphs fill the LDS of every CU for 75 ms, during which the
    // round kernel does not run (round 6: 1,000 records of 500 nt under pools of a thousand 313 -> 278 ms with the graphs' state
    // in global memory; at 700 records the same either way, at 500 and fewer the longer Edmonds kernel is what the fold waits for)
    size_t hot_max = 0;
    for (int q = 0; q < nj; q++) hot_max = std::max(hot_max, SqBlossom::hot_bytes(h_jobs[q].n, h_jobs[q].nedges, 2));
    const bool crowd = (long long)nj * std::max(1, inflight) >= 1024 || (nj >= 768 && hot_max + 16 + ((sizeof(SqBlossom) + 15) & ~(size_t)15) > 40 * 1024);
        const int nthr = env_thr ? env_thr : crowd ? 64 : std::max(64, std::min(256, (maxn + 63) / 64 * 64));
        hipLaunchKernelGGL(sq_nussinov_kernel, dim3(nj), dim3(nthr), 0, st, jobs, edges, codes, d_scr, out, cnt, jobs_rw ? 1 : 0);
    } else {                                             // SQ_ALGO_E
        // bins: see sq_mwm_plan.  The plan writes each job's LDS slice into the job table the kernel reads.
        int nbins = 0, waves = 1; size_t lds = 0; bool all_in_lds = true;
        sq_mwm_plan(h_jobs, jobs_rw, nj, bin_head, inflight, nbins, waves, lds, all_in_lds);
        sq_max_dynamic_lds((const void *)sq_mwm_kernel, 156 * 1024);
        if (!all_in_lds && dev_edges != edges) {
            // some job keeps its adjacency in global memory and walks the edges in place: give the launch a device copy
            hipError_t e = hipMemcpyAsync(dev_edges, edges, nedges * sizeof(SqMatchEdge), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return (int)e;
            edges = dev_edges;
        }
        if (waves == 1) {
            // one graph per block: blocks in table order, every block with the LDS of the launch's largest graph
            sq_max_dynamic_lds((const void *)sq_mwm_single_kernel, 156 * 1024);
            const size_t hdr = (sizeof(SqBlossom) + 15) & ~(size_t)15;
            const size_t one = lds > hdr ? lds - hdr : 0;
            hipLaunchKernelGGL(sq_mwm_single_kernel, dim3(nj), dim3(64), one, st, jobs, edges, d_scr, out, (int)one, job_flags, flag_val);
        } else
            hipLaunchKernelGGL(sq_mwm_kernel, dim3(nbins), dim3(64 * waves), lds, st, jobs, bin_head, edges, d_scr, out, job_flags, flag_val);
    }
    return (int)hipGetLastError();
}

size_t sq_lsap_scratch_bytes(int n)
{
    return ((size_t)n * n + 3 * (size_t)n) * 8 + 4 * (size_t)n * 4 + 2 * (size_t)n + 64;
}
size_t sq_nussinov_scratch_bytes(int n)
{
    return 2 * (size_t)n * n * 8 + (size_t)n * n * 4 + (((size_t)n * n + 15) & ~(size_t)15) + 4 * (size_t)(n + 2) * 4 + 64;
}
size_t sq_mwm_scratch_bytes(int n, int nedges) { return SqBlossom::scratch_bytes(n, nedges); }
