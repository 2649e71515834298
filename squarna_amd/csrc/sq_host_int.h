// sq_host_int.h -- what the host-side translation units of the library share beyond sq_host.h: sq_host.hip (errors, worker
// pools, stream / event / pinned-buffer caches, profiling), sq_batch.hip (workspace layout, sq_batch_create / destroy),
// sq_round_host.hip (the launches of a round, the host-driven round driver, the per-call C ABI of a-1 .. a-6, alignment
// step 1), sq_fold.hip (sq_fold and the concurrent forms), sq_results.hip (result getters and packing).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unordered_map>
#include "sq_host.h"
#include "sq_rounds.h"
#include "sq_pool_round.h"
#include "sq_algos_dev.h"
#include "sq_match.h"

#define HIPCK(x) do { int _r = sq_check((x), #x); if (_r) return _r; } while (0)

// host phase timers (printed to stderr when SQ_TIMING is set)
extern thread_local double g_t[8];
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct TScope { int k; double t0; TScope(int k_) : k(k_), t0(now_s()) {} ~TScope() { g_t[k] += now_s() - t0; } };

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// most stems a structure of an n-nt sequence can hold (disjoint stems of >= minlen pairs): sizes the chains' / pools' stem slices
static inline int32_t chain_tcap(int n, double minlen)
{
    const int ml = (int)std::max(1.0, std::ceil(minlen));
    return n / (2 * ml) + 1;
}

// profiling bracket on the batch stream (slot k of sq_profile_get); a no-op unless profiling is enabled
struct ProfScope {
    sq_batch *b; int k; hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(sq_batch *b_, int k_, double bytes) : b(b_), k(k_)
    {
        if (!b->prof_on) return;
        ProfSlot &p = b->prof[k];
        auto get = [&]() { hipEvent_t e; if (!p.pool.empty()) { e = p.pool.back(); p.pool.pop_back(); } else hipEventCreate(&e); return e; };
        e0 = get(); e1 = get();
        p.launches++; p.bytes += bytes;
        hipEventRecord(e0, b->stream);
    }
    ~ProfScope()
    {
        if (!e0) return;
        hipEventRecord(e1, b->stream);
        b->prof[k].pending.emplace_back(e0, e1);
    }
};

struct AlignSink {                    // mode 2: where the stems of structure k of the list are added
    const int32_t *col_off, *cols;    // host: columns of list entry k are cols[col_off[k] .. col_off[k+1])
    int L; double *matrix;            // device L x L fp64
};

// sq_host.hip
SqPool *sq_pool_get(int nthr, int device);       // idle worker pools are kept for the next batch that asks for the same
void sq_pool_put(SqPool *p);
extern "C" __global__ void sq_fill_f64_kernel(double *dst, long long n, double v);

// sq_round_host.hip
int sq_fill_impl(sq_batch *b, int full);
void sq_launch_round_kernels(sq_batch *b, hipStream_t st, int S, int maxn, int64_t maxcap, bool need_reacts, double scan_bytes,
                             int mode, const SqRoundIO &io, const SqScanArgs &scan, SqStruct *d_structs, SqStrand *d_strands,
                             bool chained, bool pooled = false, const SqPoolRoundArgs *pool_round = nullptr);
int sq_run_round_impl(sq_batch *b, SqLane &ln, const std::vector<SView> &structs, int mode,
                      std::vector<std::vector<HStem>> &out, const AlignSink *sink);
