// sq_algos_dev.h -- records of the device-side RunAlgo (sq_algos_dev.hip)
#pragma once
#include <stdint.h>
#include "sq_match.h"
#include "sq_device.h"

#define SQ_ALGO_MAXN 4096          // longest sequence the device-side RunAlgo takes (LDS of the sizes / edges kernels)

struct SqAlgoSize {                // per E / H / N job after its AnnotateStems pass (pinned, read by the host)
    int32_t nedges;                // cells of its stems
    int32_t nv;                    // distinct positions on them (Edmonds: graph vertices)
    int32_t nok;                   // stems
    int32_t raw_base;              // jobs whose edge weights need the host libm: where its nok stem scores start in the raw list
                                   // (-1: weights from the paramset's table; -2: the list had no room)
};
// Edmonds / Hungarian jobs whose stem scores are not multiples of 2^-q (reactivity factors, non-dyadic weights): the sizes
// kernel leaves the scores of their stems in pinned host memory, the host raises them to the power 1.7 with ITS libm -- what
// CPython's `**` calls (SQRNalgos.py:101,122) -- in one loop over the array, and the edges kernel reads the weights there
struct SqAlgoRaw {
    const uint8_t *need;           // pinned: per structure of the round, 1: the job's weights go through the host
    double *vals;                  // pinned: the raw list (scores, then their powers, in place)
    uint32_t *ctr;                 // device: entries taken
    uint32_t cap;
};
struct SqAlgoJob {                 // per job, for the edges / finish kernels (pinned, written by the host once the sizes are known)
    int32_t job, algo;             // batch job index, SQ_ALGO_*
    SqMatchEdge *edges;            // device: the job's edge list
    int32_t *vid2pos;              // device, Edmonds: graph vertex -> sequence position
    const double *raw;             // pinned: stemscore ** 1.7 of the job's stems in survivor-list order (nullptr: the paramset's table)
};
struct SqAlgoStat {                // per launch of the finish kernel (device; sq_algo_publish_kernel copies it to the host)
    uint32_t bad;                  // 1: blossom capacity exceeded, 2: stem capacity exceeded
    uint32_t level_ovf;
    unsigned long long max_pass_job;   // (scan passes << 32) | job row: the blossom kernel's critical path
    unsigned long long passes, graphs;
    long long max_events, max_n, max_m;
};

#ifdef __HIPCC__
extern "C" {
__global__ void sq_algo_sizes_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, SqAlgoSize *sizes, SqAlgoRaw raw);
struct SqAlgoStatPtrs { SqAlgoStat *p[3]; };          // the items' counters, zeroed by the edges kernel (no memset launches)
__global__ void sq_algo_edges_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, const SqAlgoJob *jobs, int maxn_lds, SqAlgoStatPtrs zs, int nokcap);
// dynamic LDS of the edges kernel: per-position arrays (6 bytes) and, with nokcap > 0, the stem lists of the fast form
inline size_t sq_algo_edges_lds(int maxn_lds, int nokcap)
{
    size_t b = (((size_t)6 * maxn_lds + 15) & ~(size_t)15) + 16;
    b += (size_t)4 * (2 * (2 * (size_t)maxn_lds + 2));       // the anti-diagonals' starts and fills (both forms order the stems by them)
    if (nokcap > 0) b += (size_t)4 * (5 * (size_t)nokcap);
    return b;
}
__global__ void sq_algo_finish_kernel(SqDevCtx c, const SqAlgoJob *jobs, const SqMatchJob *mj, const int32_t *out, const int32_t *cnt,
                                      int levellimit_opt, SqPoolFin *fin, SqPoolStem *fin_stems, uint32_t *fin_ctr, uint32_t fin_cap,
                                      uint32_t fin_stem_cap, SqAlgoStat *stats, int tcap);
__global__ void sq_algo_publish_kernel(SqAlgoStat *stats, SqAlgoStat *h_stats, const SqMatchJob *mj, const int32_t *out, int is_edmonds,
                                       uint32_t *flag, uint32_t value, int nj);
}
// dynamic LDS of sq_algo_finish_kernel for sequences up to n nt whose structures hold up to tcap stems
static inline size_t sq_algo_finish_lds(int n, int tcap)
{
    const size_t half = (size_t)n / 2 + 2;
    return ((8 * half + 15) & ~(size_t)15) + 8 * half + 2 * (half + 8) + sq_extend_lds_bytes(tcap) + 64;
}
#endif
