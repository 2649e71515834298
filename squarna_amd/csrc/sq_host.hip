// sq_host.hip -- host driver of libsquarna_hip.so: batch set-up, kernel launches, the
// round driver and the greedy pool loop (SQRNdbnseq.py:1102-1199).  Device memory is
// the caller's workspace; the host only orchestrates (one small H2D + D2H per round).
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

static thread_local std::string g_err;
std::atomic<long long> g_cpuacc[12];
bool g_cpuacc_on = getenv("SQ_CPUACC") != nullptr;

// host phase timers (printed to stderr when SQ_TIMING is set)
static thread_local double g_t[8];
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct TScope { int k; double t0; TScope(int k_) : k(k_), t0(now_s()) {} ~TScope() { g_t[k] += now_s() - t0; } };
void sq_set_error(const std::string &msg) { g_err = msg; }
int sq_check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
}
#define HIPCK(x) do { int _r = sq_check((x), #x); if (_r) return _r; } while (0)

// ---- host CPUs this process may really use --------------------------------------------------------
// min(hardware threads, affinity mask, cgroup CPU quota).  A container with cpu.max = "1600000 100000" shows 256
// hardware threads but gets 16 CPUs worth of time per period; threads beyond that (workers, spinning waiters) only
// burn the quota and the whole process is throttled for the rest of the period (measured on the MI355X box: 40-60 ms
// stalls every few steps with 8 batches in flight).
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>
void sq_max_dynamic_lds(const void *fn, int bytes)
{
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;     // (kernel, device) pairs already raised
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto &d : done) if (d.first == fn && d.second == dev) return;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) (void)hipGetLastError();   // (a launch that needs it reports the failure)
    done.emplace_back(fn, dev);
}

int sq_effective_cpus()
{
    static const int n = [] {
        int cpus = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min(cpus, std::max(1, CPU_COUNT(&set)));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                    // cgroup v2: "<quota|max> <period>"
            char q[64]; long long per = 0;
            if (fscanf(f, "%63s %lld", q, &per) == 2 && per > 0 && strcmp(q, "max") != 0)
                cpus = std::min<long long>(cpus, std::max<long long>(1, (atoll(q) + per - 1) / per));
            fclose(f);
        } else {
            long long quota = -1, per = 0;                                       // cgroup v1
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
            if (quota > 0 && per > 0) cpus = std::min<long long>(cpus, std::max<long long>(1, (quota + per - 1) / per));
        }
        if (const char *e = getenv("SQ_CPUS")) cpus = std::max(1, atoi(e));
        return cpus;
    }();
    return n;
}
// One step of a wait loop on a pinned completion word.  Alone, a waiter spins (a round lasts ~100 us; a sleep would
// double it).  With several batches in flight (`relaxed`) it spins for a few microseconds and then sleeps ~10 us at a
// time: a dozen threads spinning through 6 ms blossom kernels would use up the CPU quota the workers need.
// relaxed waiting pays once the waiters alone (about two per batch in flight) would take most of the CPU budget
bool sq_relaxed_waits(const sq_batch *b)
{
    static const int forced = getenv("SQ_RELAX") ? atoi(getenv("SQ_RELAX")) : -1;
    if (forced >= 0) return forced != 0;
    int lws = 1;
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) lws = std::max(1, atoi(e));
    return 2 * b->inflight * lws > (sq_effective_cpus() * 3) / 5;
}
// timer slack of the calling thread before sq_wait_step lowered it (-1: untouched).  The fold entry points restore it on
// return: the thread that calls sq_fold belongs to the caller (Python's main thread), not to the library.
static thread_local long g_slack_saved = -1;
void sq_restore_timerslack()
{
    if (g_slack_saved >= 0) { prctl(PR_SET_TIMERSLACK, (unsigned long)g_slack_saved, 0, 0, 0); g_slack_saved = -1; }
}
void sq_wait_step(uint64_t spins, bool relaxed)
{
    if (relaxed && spins > 512) {
        if (g_slack_saved < 0) {                            // 1 us instead of the default 50 us, for the length of the fold
            const int cur = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
            g_slack_saved = cur > 0 ? cur : 50000;
            prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
        }
        struct timespec ts = {0, 10000};
        nanosleep(&ts, nullptr);
        return;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

// ---- host worker pool -------------------------------------------------------------------------
SqPool::SqPool(int nthreads_, int device_) : nthreads(nthreads_), device(device_)
{
    for (int t = 1; t < nthreads; t++) {
        const int group = t <= 15 ? 0 : 1;
        group_size[group]++;
        // (workers call into HIP -- stream queries, copies of dense matrices: they work on the batch's device)
        workers.emplace_back([this, group] { if (device >= 0) hipSetDevice(device); worker(group); });
    }
}
SqPool::~SqPool()
{
    { std::lock_guard<std::mutex> lk(mu); stop = true; gen[0]++; gen[1]++; }
    cv_start[0].notify_all(); cv_start[1].notify_all();
    for (auto &t : workers) t.join();
}
void SqPool::worker(int group)
{
    uint64_t seen = 0;
    for (;;) {
        const std::function<void(int)> *f;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv_start[group].wait(lk, [&] { return gen[group] != seen; });
            seen = gen[group];
            if (stop) return;
            f = fn;
        }
        for (int i; (i = next.fetch_add(1)) < total;) (*f)(i);
        { std::lock_guard<std::mutex> lk(mu); if (--active == 0) cv_done.notify_one(); }
    }
}
void SqPool::parallel_for(int n, const std::function<void(int)> &f, int wide)
{
    if (n <= 0) return;
    if (workers.empty() || n == 1) { for (int i = 0; i < n; i++) f(i); return; }
    std::lock_guard<std::mutex> one_caller(callers);
    const bool all = (wide < 0 ? n >= 512 : wide != 0) && group_size[1] > 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        fn = &f; total = n; next.store(0); active = group_size[0] + (all ? group_size[1] : 0); gen[0]++;
        if (all) gen[1]++;
    }
    cv_start[0].notify_all();
    if (all) cv_start[1].notify_all();
    for (int i; (i = next.fetch_add(1)) < n;) f(i);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return active == 0; });
}
static SqPool *pool_get(int nthr, int device);
static void pool_put(SqPool *p);
SqPool *sq_pool(sq_batch *b)
{
    if (!b->pool) {
        // up to 32 workers, sharing the CPUs this process may use (sq_effective_cpus) with the other ranks of the node
        // (torchrun's LOCAL_WORLD_SIZE) and the other batches in flight (sq_fold_concurrent)
        // The workers sleep between bursts (tails of finished sequences, pool growth of big rounds), so their number is
        // not tied to the CPU budget as tightly as the spinning waiters are: 4 x the CPUs of this rank, shared among
        // the batches in flight, between 8 and 32 (measured with a 16-CPU quota and 4 batches in flight: 2 workers
        // per batch 18.0 ms per step, 8 -> 12.2 ms, 32 -> 11.3 ms).
        unsigned cores = (unsigned)sq_effective_cpus();
        if (const char *lws = getenv("LOCAL_WORLD_SIZE")) cores = std::max(1u, cores / (unsigned)std::max(1, atoi(lws)));
        int nthr = (int)std::min(std::max(4u * cores / (unsigned)std::max(1, b->inflight), 8u), 32u);
        if (const char *e = getenv("SQ_HOST_THREADS")) nthr = std::max(1, atoi(e));
        b->pool = pool_get(nthr, b->device);
    }
    return b->pool;
}

// the batch's reactivities on the host: the caller's array, or 0.5 everywhere formed on first use (batches created with
// reacts == NULL only need it on the host paths: the host tail, RunAlgo's host filters)
const double *sq_host_reacts(const sq_batch *b)
{
    if (b->reacts_null)
        std::call_once(b->reacts_once, [b] { const_cast<sq_batch *>(b)->reacts.assign((size_t)std::max<int64_t>(b->ltot, 1), 0.5); });
    return b->reacts.data();
}

extern "C" __global__ void sq_fill_f64_kernel(double *dst, long long n, double v)
{
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) dst[q] = v;
}

// ---- stream / event / worker-pool caches ----------------------------------------------------------
namespace {
struct ObjCache {
    std::mutex mu;
    std::unordered_map<int, std::vector<hipStream_t>> streams;
    std::unordered_map<int, std::vector<hipEvent_t>> events;
    std::vector<SqPool *> pools;
} g_objs;
}
hipError_t sq_stream_get(int device, hipStream_t *s)
{
    {
        std::lock_guard<std::mutex> lk(g_objs.mu);
        auto &v = g_objs.streams[device];
        if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
void sq_stream_put(int device, hipStream_t s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(g_objs.mu);
        auto &v = g_objs.streams[device];
        if (v.size() < 64) { v.push_back(s); return; }
    }
    hipStreamDestroy(s);
}
hipError_t sq_event_get(int device, hipEvent_t *e)
{
    {
        std::lock_guard<std::mutex> lk(g_objs.mu);
        auto &v = g_objs.events[device];
        if (!v.empty()) { *e = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipEventCreateWithFlags(e, hipEventDisableTiming);
}
void sq_event_put(int device, hipEvent_t e)
{
    if (!e) return;
    {
        std::lock_guard<std::mutex> lk(g_objs.mu);
        auto &v = g_objs.events[device];
        if (v.size() < 256) { v.push_back(e); return; }
    }
    hipEventDestroy(e);
}
static SqPool *pool_get(int nthr, int device)
{
    {
        std::lock_guard<std::mutex> lk(g_objs.mu);
        for (size_t k = 0; k < g_objs.pools.size(); k++)
            if (g_objs.pools[k]->nthreads == nthr && g_objs.pools[k]->device == device) {
                SqPool *p = g_objs.pools[k];
                g_objs.pools.erase(g_objs.pools.begin() + k);
                return p;
            }
    }
    return new SqPool(nthr, device);
}
static void pool_put(SqPool *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_objs.mu);
        if (g_objs.pools.size() < 16) { g_objs.pools.push_back(p); return; }   // (its workers sleep on a condition variable)
    }
    delete p;
}

// ---- pinned buffer cache --------------------------------------------------------------------------
namespace {
struct PinnedCache {
    std::mutex mu;
    std::vector<std::pair<size_t, void *>> idle;       // (capacity, pointer)
    std::unordered_map<void *, size_t> live;
    size_t idle_bytes = 0;
} g_pinned;
}
int sq_pinned_get(void **p, size_t bytes)
{
    const size_t want = (std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095;
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        int best = -1;
        for (size_t k = 0; k < g_pinned.idle.size(); k++) {
            const size_t cap = g_pinned.idle[k].first;
            if (cap >= want && cap <= 2 * want + 65536 && (best < 0 || cap < g_pinned.idle[best].first)) best = (int)k;
        }
        if (best >= 0) {
            *p = g_pinned.idle[best].second;
            g_pinned.live[*p] = g_pinned.idle[best].first;
            g_pinned.idle_bytes -= g_pinned.idle[best].first;
            g_pinned.idle.erase(g_pinned.idle.begin() + best);
            return 0;
        }
    }
    // portable: the cache is process-wide, a buffer may be reused by a batch on another device
    const int r = sq_check(hipHostMalloc(p, want, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable), "hipHostMalloc");
    if (r) { *p = nullptr; return r; }
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    g_pinned.live[*p] = want;
    return 0;
}
void sq_pinned_put(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        auto it = g_pinned.live.find(p);
        const size_t cap = it == g_pinned.live.end() ? 0 : it->second;
        if (it != g_pinned.live.end()) g_pinned.live.erase(it);
        // (a batch holds ~25 pinned buffers; eight batches of a server's step are created and destroyed together: with room
        // for 64 idle buffers two thirds of them went back to the driver -- hipHostFree + hipHostMalloc: 6.5 ms per batch)
        if (cap && g_pinned.idle.size() < 1024 && g_pinned.idle_bytes + cap <= ((size_t)2 << 30)) {
            g_pinned.idle.emplace_back(cap, p);
            g_pinned.idle_bytes += cap;
            return;
        }
    }
    hipHostFree(p);
}

extern "C" int sq_version(void) { return 100; }
extern "C" const char *sq_last_error(void) { return g_err.c_str(); }

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
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

static inline int32_t chain_tcap(int n, double minlen)
{
    const int ml = (int)std::max(1.0, std::ceil(minlen));
    return n / (2 * ml) + 1;
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
    // device log of final structures + scratch of the device tail (sq_tail_dev.hip)
    size_t off_fin, off_fin_stems, off_fin_ctr, off_jobevals, off_t_jobs, off_t_seqjob0, off_t_ord, off_t_cstems, off_t_csn, off_t_hash,
           off_t_rep, off_t_mask, off_t_scores, off_t_dlist, off_t_rlist, off_t_seqs, off_t_refp, off_t_refn, off_t_pow;
    uint32_t fin_cap, fin_stem_cap; int32_t pow_len;
    size_t off_mulcols;              // alignment columns of every position (shared L x L weighting matrix), else unused
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
        const bool mul = (d->mul_score && d->mul_score[j]) || (d->bpp_term && d->bpp_term[j]) || (d->mul_shared && d->mul_shared[j]);
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
            maxcap = std::max<int64_t>(maxcap, (int64_t)(0.117 * nn * nn * runs[d->job_pset[j]] * 1.6 + 256));
        }
    }
    L.cand_records = std::min<int64_t>((int64_t)L.max_structs * maxcap, (int64_t)160 << 20);   // (5 GiB of 32-byte records at most)
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
    b->stream = (hipStream_t)hip_stream;
    if (hipGetDevice(&b->device) != hipSuccess) b->device = -1;      // the caller's current device: every thread the library spawns adopts it
    b->nseq = d->nseq; b->npset = d->npset; b->njobs = d->njobs; b->maxn = L.maxn; b->ltot = L.ltot;
    b->seq_off.assign(d->seq_off, d->seq_off + d->nseq + 1);
    b->codes.assign(d->codes, d->codes + L.ltot);
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
    for (int j = 0; j < d->njobs; j++) {
        SqJob &J = b->jobs[j];
        const int s = d->job_seq[j];
        J.n = d->seq_off[s + 1] - d->seq_off[s];
        J.ld = ld_of(J.n); J.seq = s; J.pset = d->job_pset[j];
        J.pos_off = d->seq_off[s];
        J.mat64_off = -1; J.has_ext = 0;
        J.nw = bits_nw(J.n); J.bpitch = bits_pitch(J.n); J.bits_off = mbits; mbits += (int64_t)J.nw * J.bpitch;
        J.rb_off = d->rbp_off[s]; J.nrb = d->rbp_off[s + 1] - d->rbp_off[s];
        const bool ext = d->ext_score && d->ext_score[j];
        const bool term = d->bpp_term && d->bpp_term[j];
        const bool shared = d->mul_shared && d->mul_shared[j];
        const bool mul = (d->mul_score && d->mul_score[j]) || term || shared;
        J.ext_add = term && d->psets[J.pset].bpp < 0 ? 1 : 0;
        if (ext) { J.mat64_off = m64; J.has_ext = 1; m64 += 2 * (int64_t)J.n * J.n; }
        else if (mul) { J.mat64_off = m64; J.has_ext = 2; m64 += (int64_t)J.n * J.n; }
        J.mat_off = -1;
        J.mat64_diag = (shared && !ext) ? 1 : 0;          // the gather kernel writes score x weight, diagonal-major (sq_cells.h)
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
        // jobs weighted by the shared L x L matrix: their N x N slices are gathered on the device through the column maps
        int32_t *d_cols = (int32_t *)(base + L.off_mulcols);
        for (int64_t q = 0; q < L.ltot; q++)
            if (d->mul_cols[q] < 0 || d->mul_cols[q] >= d->mul_L) { hipStreamSynchronize(st); delete b; sq_set_error("mul_cols out of range"); return -1; }
        UP(d_cols, d->mul_cols, 4 * (size_t)L.ltot);
        std::vector<int32_t> jl;
        for (int j = 0; j < d->njobs; j++) if (d->mul_shared[j]) jl.push_back(j);
        if (!jl.empty()) {
            // (the job list travels in the candidate arena's first bytes: nothing else uses it before the first fold)
            int32_t *d_jl = (int32_t *)(base + L.off_cands);
            UP(d_jl, jl.data(), 4 * jl.size());
            sq_launch_gather_mul(b->ctx, d->mul_matrix_dev, d->mul_L, d_cols, d_jl, (int)jl.size(), L.maxn, st);
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
    hipStreamSynchronize(b->stream);
    for (int k = 0; k < 3; k++) if (b->side[k]) { hipStreamSynchronize(b->side[k]); sq_stream_put(b->device, b->side[k]); }
    if (b->lane_stream) { hipStreamSynchronize(b->lane_stream); sq_stream_put(b->device, b->lane_stream); }
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
    pool_put(b->pool);
    sq_event_put(b->device, b->lane_ev);
    for (auto &p : b->prof) {
        for (auto &e : p.pending) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
        for (auto &e : p.pool) hipEventDestroy(e);
    }
    delete b;
}

// ---- profiling (HIP events on the batch stream) ----------------------------------------
namespace {
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
void prof_collect(sq_batch *b)
{
    for (auto &p : b->prof) {
        for (auto &e : p.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) p.ms += ms;
            p.pool.push_back(e.first); p.pool.push_back(e.second);
        }
        p.pending.clear();
    }
}
}  // namespace

void sq_prof_begin(sq_batch *b, int k, hipStream_t st, hipEvent_t *e0)
{
    *e0 = nullptr;
    if (!b->prof_on) return;
    ProfSlot &p = b->prof[k];
    if (!p.pool.empty()) { *e0 = p.pool.back(); p.pool.pop_back(); } else hipEventCreate(e0);
    p.launches++;
    hipEventRecord(*e0, st);
}
void sq_prof_end(sq_batch *b, int k, hipStream_t st, hipEvent_t e0)
{
    if (!e0) return;
    ProfSlot &p = b->prof[k];
    hipEvent_t e1;
    if (!p.pool.empty()) { e1 = p.pool.back(); p.pool.pop_back(); } else hipEventCreate(&e1);
    hipEventRecord(e1, st);
    p.pending.emplace_back(e0, e1);
}

extern "C" int sq_profile_enable(sq_batch *b, int32_t on) { b->prof_on = on != 0; return 0; }
extern "C" int sq_profile_reset(sq_batch *b)
{
    hipStreamSynchronize(b->stream);
    prof_collect(b);
    for (auto &p : b->prof) { p.ms = 0; p.launches = 0; p.bytes = 0; }
    { std::lock_guard<std::mutex> lk(b->mwm_mu); for (int64_t &x : b->mwm_stats) x = 0; }
    return 0;
}
extern "C" int sq_profile_counters(sq_batch *b, int32_t kernel, int64_t out[6])
{
    if (!b || !out || kernel != 4) { sq_set_error("counters exist for kernel 4 (Edmonds) only"); return -1; }
    std::lock_guard<std::mutex> lk(b->mwm_mu);
    for (int k = 0; k < 6; k++) out[k] = b->mwm_stats[k];
    return 0;
}
extern "C" int sq_profile_get(sq_batch *b, int32_t k, double *ms, int64_t *launches, double *bytes)
{
    if (k < 0 || k > 7) return -1;
    hipStreamSynchronize(b->stream);
    for (int q = 0; q < 3; q++) if (b->side[q]) hipStreamSynchronize(b->side[q]);
    prof_collect(b);
    *ms = b->prof[k].ms; *launches = b->prof[k].launches; *bytes = b->prof[k].bytes;
    return 0;
}

// ---- a-1 -----------------------------------------------------------------------------------
// full = 1: fp32 score matrices of every job (the API op).  full = 0: only what the fold path reads -- the
// bit matrices, computed straight from the O(N) inputs; jobs with caller / multiplier matrices still go
// through the fp32 fill (it imports the bool matrix and forms score * multiplier in the dense arena).
static int fill_impl(sq_batch *b, int full)
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
    return fill_impl(b, 1);
}

int sq_prepare_scan(sq_batch *b)
{
    return b->bits_ready ? 0 : fill_impl(b, 0);
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

namespace {
struct AlignSink {                    // mode 2: where the stems of structure k of the list are added
    const int32_t *col_off, *cols;    // host: columns of list entry k are cols[col_off[k] .. col_off[k+1])
    int L; double *matrix;            // device L x L fp64
};
}

// the kernels of one round over S structures: state arrays, bit-diagonal scan, exact scoring (mode 0: + ScoreStems),
// and for host-driven greedy rounds the range filter that writes the round's output records
static void launch_round_kernels(sq_batch *b, hipStream_t st, int S, int maxn, int64_t maxcap, bool need_reacts, double scan_bytes,
                                 int mode, const SqRoundIO &io, const SqScanArgs &scan, SqStruct *d_structs, SqStrand *d_strands,
                                 bool chained, bool pooled = false, const SqPoolRoundArgs *pool_round = nullptr)
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
        hipLaunchKernelGGL(sq_pool_round_kernel, dim3(S), dim3(64), lo.total, st, b->ctx, scan, b->pool_io, *pool_round);
        return;
    }
    if (fuse) {
        ProfScope ps(b, 2, scan_bytes);
        const size_t dyn_state = (size_t)7 * ((maxn + 8) & ~7) + 64, dyn_scan = 4 * (size_t)b->state.fbstride;
        hipLaunchKernelGGL(sq_state_scan_kernel, dim3(S), dim3(64), std::max(dyn_state, dyn_scan), st, b->ctx, io, b->state, scan, maxn, chained ? 1 : 0);
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
        const int thr0 = maxn <= 200 ? (crowded ? short_thr : 128) : (maxn <= 400 ? 256 : 512);
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
    launch_round_kernels(b, st, S, maxn, maxcap, need_reacts, scan_bytes, mode, io, scan, ln.d_structs, ln.d_strands, false);
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
                order_free = J.default_reacts && J.mat64_off < 0 && b->pset_dyadic[J.pset];
                maxcap = std::max<int64_t>(maxcap, J.cand_cap);
            }
            if (order_free) {
                std::vector<int32_t> starts(S);
                for (int k = 0; k < S; k++) starts[k] = sink->col_off[lo + k] - c0;
                int32_t *d_starts = d_cols + (c1 - c0);
                HIPCK(hipMemcpyAsync(d_starts, starts.data(), (size_t)S * 4, hipMemcpyHostToDevice, st));
                HIPCK(hipStreamSynchronize(st));             // (starts is a local)
                const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>(maxcap / 4096, 1), 64);
                hipLaunchKernelGGL(sq_scatter_all_kernel, dim3(blocks, S), dim3(256), 0, st, b->ctx, ln.d_structs, scan,
                                   d_cols, d_starts, sink->L, sink->matrix);
            } else
            for (int k = 0; k < S; k++) {
                const SqJob &J = b->jobs[structs[lo + k].job];
                const unsigned blocks = (unsigned)std::min<int64_t>(std::max<int64_t>(J.cand_cap / 1024, 1), 1024);
                hipLaunchKernelGGL(sq_scatter_kernel, dim3(blocks), dim3(256), 0, st, b->ctx, ln.d_structs, scan, k,
                                   d_cols + (sink->col_off[lo + k] - c0), sink->L, sink->matrix);
            }
        }
    }
    const uint32_t seq = ++*ln.round_seq;
    hipLaunchKernelGGL(sq_done_kernel, dim3(1), dim3(1), 0, st, io, scan, seq);
    HIPCK(hipGetLastError());
    if (g_cpuacc_on) { const long long t = CpuScope::now(); g_cpuacc[mode == 1 ? 7 : 5] += t - cpu_t0; cpu_t0 = t; }
    // wait for the round: spin on the sequence number in pinned memory (no driver round trip); a stuck or
    // faulted queue is caught by polling the stream now and then
    {
        volatile uint32_t *flag = ln.h_seq;
        uint64_t spins = 0;
        const bool relaxed = sq_relaxed_waits(b);
        const uint64_t poll_mask = relaxed ? 0x3FFF : 0xFFFFF;
        while (*flag != seq) {
            if ((++spins & poll_mask) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q != hipErrorNotReady) {
                    if (q != hipSuccess) return sq_check(q, "round kernels");
                    if (*flag != seq) { HIPCK(hipStreamSynchronize(st)); if (*flag != seq) { sq_set_error("round did not signal completion"); return 2; } }
                }
            }
            sq_wait_step(spins, relaxed);
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (g_cpuacc_on) g_cpuacc[6] += CpuScope::now() - cpu_t0;
    const SqCounters ctr = *ln.h_ctr;
    if (ctr.cand_ovf) { sq_set_error("candidate capacity exceeded (raise cand_per_nt)"); return -3; }
    if (ctr.out_ovf) { ln.out_ovf_seen = true; sq_set_error("round output capacity exceeded (lower max_structs)"); return -3; }
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
    launch_round_kernels(b, st, S, maxn, maxcap, need_reacts, scan_bytes, 2, io, scan, ln.d_structs, ln.d_strands, false);
    hipLaunchKernelGGL(sq_algo_sizes_kernel, dim3(S), dim3(256), 0, st, b->ctx, ln.d_structs, scan, h_sizes, raw);
    const uint32_t seq = ++*ln.round_seq;
    hipLaunchKernelGGL(sq_done_kernel, dim3(1), dim3(1), 0, st, io, scan, seq);
    HIPCK(hipGetLastError());
    {
        volatile uint32_t *flag = ln.h_seq;
        uint64_t spins = 0;
        const bool relaxed = sq_relaxed_waits(b);
        const uint64_t poll_mask = relaxed ? 0x3FFF : 0xFFFFF;
        while (*flag != seq) {
            if ((++spins & poll_mask) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q != hipErrorNotReady) {
                    if (q != hipSuccess) return sq_check(q, "AnnotateStems round");
                    if (*flag != seq) { HIPCK(hipStreamSynchronize(st)); if (*flag != seq) { sq_set_error("round did not signal completion"); return 2; } }
                }
            }
            sq_wait_step(spins, relaxed);
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    const SqCounters ctr = *ln.h_ctr;
    if (ctr.cand_ovf) { sq_set_error("candidate capacity exceeded (raise cand_per_nt)"); return -3; }
    return 0;
}

static int run_round_impl(sq_batch *b, SqLane &ln, const std::vector<SView> &structs, int mode,
                          std::vector<std::vector<HStem>> &out, const AlignSink *sink);
int sq_run_round(sq_batch *b, const std::vector<SView> &structs, int mode, std::vector<std::vector<HStem>> &out)
{
    return run_round_impl(b, b->lane_full, structs, mode, out, nullptr);
}
// `ln`: the round buffers to use.  The full lane ends where the arena is lent to matching kernels in flight
// (cand_reserved); the half lanes are set up by sq_fold.
static int run_round_impl(sq_batch *b, SqLane &ln, const std::vector<SView> &structs, int mode,
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
    int r = run_round_impl(b, b->lane_full, views, 2, unused, &sink);
    if (!r) {
        const unsigned nt = (unsigned)((L + 31) / 32);
        hipLaunchKernelGGL(sq_mirror_kernel, dim3(nt, nt), dim3(256), 0, b->stream, d_matrix, L);
        r = sq_check(hipStreamSynchronize(b->stream), "sq_mirror_kernel");
    }
    if (getenv("SQ_TIMING"))
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
    hipLaunchKernelGGL(sq_colselect_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, st,
                       d_matrix, L, threshold, minspan, (long long *)d_idx, d_val, (long long)cap, (unsigned long long *)d_count);
    return sq_check(hipGetLastError(), "sq_colselect_kernel");
}

// ---- a-7: greedy pool loop for every job at once (SQRNdbnseq.py:1102-1199) ----------------------
namespace {
struct alignas(128) JobPool {                // (own cache lines: two lanes work on neighbouring jobs)
    std::vector<HStruct> cur;                // curstemsets
    std::vector<HStruct> nxt;                // next round's curstemsets (kept between rounds: no reallocation)
    std::vector<std::vector<HStem>> fin;     // finstemsets (greedy part)
    double cursubopt = 0, suboptinc = 0, suboptmax = 0, maxstemnum = 0;
    size_t cursize = 1;
    int64_t evals = 0;
};
}  // namespace

extern "C" int sq_fold(sq_batch *b, const sq_fold_opts *opts, const int32_t *ref_off, const int32_t *ref_pairs,
                       const uint8_t *has_ref)
{
    if (!b || !opts) { sq_set_error("bad argument"); return -1; }
    const sq_fold_opts &o = *opts;
    if (o.poollim < 1) { sq_set_error("poollim must be positive"); return -1; }
    SqSlackGuard slack_guard;
    const long long cpu_fold0 = g_cpuacc_on ? CpuScope::now() : 0;
    struct FoldTimer { double t0; ~FoldTimer() { if (getenv("SQ_TIMING")) fprintf(stderr, "[sq_fold] total %.3f ms (incl. teardown)\n", (now_s() - t0) * 1e3); } } fold_timer{now_s()};
    // a-1, once per job and per fold (:1076): never reused from an earlier call, a fold is the whole path
    int r = fill_impl(b, 0);
    if (r) return r;
    // The ranking tail runs on the device (sq_tail_dev.hip) over the device log of final structures whenever the options
    // allow; the host tail below is its fallback.  The log and the per-job evaluation counts start empty.
    const bool dev_tail = sq_tail_device_wanted(b, o);
    // the scoring kernel's two short cuts, per fold (tests fold the same batch with and without them)
    b->score_bound = getenv("SQ_NO_SCORE_BOUND") == nullptr;
    b->score_ctx = getenv("SQ_NO_SCORE_CONTEXT") == nullptr;
    b->no_pool_round = getenv("SQ_NO_POOL_ROUND") != nullptr;              // (tests fold both ways in one process)
    b->pool_round_always = getenv("SQ_POOL_ROUND_ALWAYS") != nullptr;      // (also for a small batch alone: measurements)
    b->pool_round_nsurv = getenv("SQ_POOL_ROUND_NSURV") ? std::max(64, std::min(2048, atoi(getenv("SQ_POOL_ROUND_NSURV")))) : 0;   // (tests: survivors spill)
    if (b->any_dense < 0) { b->any_dense = 0; for (const SqJob &J : b->jobs) if (J.mat64_off >= 0 || J.has_ext) b->any_dense = 1; }
    b->packed_ok = false;
    hipLaunchKernelGGL(sq_fold_begin_kernel, dim3((b->njobs + 256) / 256), dim3(256), 0, b->stream, b->d_fin_ctr, b->d_job_evals,
                       b->tail.job_cnt, b->njobs);
    // (the pools -- thousands of small vectors -- are torn down by a helper thread after the fold returns)
    auto *pools_owner = new std::vector<JobPool>(b->njobs);
    struct PoolsDrop {
        std::vector<JobPool> *p;
        ~PoolsDrop()
        {
            static const bool sync_drop = getenv("SQ_SYNC_TEARDOWN") != nullptr;
            if (sync_drop) delete p; else std::thread([q = p] { CpuScope cpu_(11); delete q; }).detach();
        }
    } pools_drop{pools_owner};
    std::vector<JobPool> &pools = *pools_owner;
    std::vector<uint32_t> algos(b->njobs);
    for (int j = 0; j < b->njobs; j++) {
        const sq_paramset &ps = b->psets[b->job_pset[j]];
        algos[j] = o.algos ? o.algos : ps.algorithms;       // :1065-1066
        JobPool &P = pools[j];
        P.cursubopt = ps.suboptmin;                         // :1069
        P.suboptinc = (ps.suboptmax - ps.suboptmin) / ps.suboptsteps;   // :1071
        P.suboptmax = ps.suboptmax; P.maxstemnum = ps.maxstemnum;
    }
    // Edmonds / Hungarian / Nussinov paramsets (:1094-1100); their stemsets precede the greedy ones.
    // The reference iterates a Python set of letters (unspecified order); we use E, H, N.
    SqAlgoAsync *pending = nullptr;
    const double ta = now_s();
    if (getenv("SQ_TIMING")) fprintf(stderr, "[sq_fold] setup before E/H/N begin: %.3f ms (bit matrix launch + job pools)\n", (ta - fold_timer.t0) * 1e3);
    // (with the device tail: RunAlgo's filters on the device too when the batch qualifies, sq_algos_dev.hip)
    { CpuScope cpu_(9); r = sq_algos_begin(b, algos, pending, o.levellimit, dev_tail); }   // AnnotateStems + matching kernels on side streams
    const bool dev_algos = sq_algos_on_device(pending);
    b->last_paths = dev_algos ? 2 : 0;
    if (getenv("SQ_TIMING") && pending) fprintf(stderr, "[sq_fold] RunAlgo for E / H / N: %s\n", dev_algos ? "on the device (sq_algos_dev.hip)" : "host-driven");
    struct PendGuard {                                      // error paths: wait for the side streams, release the arena
        sq_batch *b; SqAlgoAsync *&p;
        ~PendGuard() { if (p) { sq_algos_abandon(b, p); p = nullptr; } }
    } guard{b, pending};
    if (r) return r;
    const double tbegin = now_s() - ta;
    const double tfold0 = now_s();
    // Width-1 pools (poollim == 1): the greedy rounds are chained on the device (sq_chain.hip) when all structures fit
    // the round buffers at once; otherwise (and for wider pools) the host drives the rounds.
    std::vector<int> greedy_jobs;
    for (int j = 0; j < b->njobs; j++) if (algos[j] & SQ_ALGO_G) greedy_jobs.push_back(j);
    const bool no_chain = getenv("SQ_NO_CHAIN") != nullptr;     // (read per fold: tests compare both drivers in one process)
    const bool no_rounds = getenv("SQ_NO_ROUNDS") != nullptr;   // (likewise: the launched rounds instead of the persistent round kernel)
    bool use_chain = o.poollim == 1 && !no_chain && !greedy_jobs.empty();
    if (use_chain)
        for (int j : greedy_jobs)
            if (chain_tcap(b->jobs[j].n, b->psets[b->job_pset[j]].minlen) > SQ_CHAIN_TMAX ||
                b->jobs[j].cand_cap > b->cand_records - b->cand_reserved) use_chain = false;
    // Wider pools: booked on the device as well (sq_pool.hip) when the batch has the slot arrays (structures of at most
    // SQ_CHAIN_TMAX stems) and one structure per greedy job fits the round buffers; any capacity overflow during the fold makes
    // the host repeat it with its own loop.
    const bool no_pool = getenv("SQ_NO_POOL") != nullptr;
    bool use_pool = !use_chain && o.poollim > 1 && !no_pool && !greedy_jobs.empty() && b->pool_io.pt > 0;
    // the jobs each device driver takes.  Pools that may branch (poollim > 1) but almost never do -- range factor 1.0 (only
    // exact ties branch, :769-778) over cells weighted by a dense fp64 matrix (the alignment's rows, bpp terms) -- first run
    // as chains on the persistent round kernel, which stops a structure at the first tie; the device pools then fold what
    // is left (tied_jobs) and every other job
    std::vector<int> chain_jobs, pool_jobs_v, tied_jobs;
    bool chain_ties = false;
    if (use_chain) chain_jobs = greedy_jobs;
    if (use_pool) {
        static const bool no_opt = getenv("SQ_NO_OPT_CHAIN") != nullptr;
        for (int j : greedy_jobs) {
            const SqJob &J = b->jobs[j];
            const sq_paramset &ps = b->psets[b->job_pset[j]];
            const bool opt = !no_opt && !no_rounds && J.mat64_off >= 0 && ps.suboptmin == 1.0 && ps.suboptmax == 1.0 && J.n <= SQ_ROUNDS_MAXN &&
                             chain_tcap(J.n, ps.minlen) <= SQ_CHAIN_TMAX && J.cand_cap <= b->cand_records - b->cand_reserved;
            (opt ? chain_jobs : pool_jobs_v).push_back(j);
        }
        chain_ties = !chain_jobs.empty();
    }
    auto host_pools_init = [&]() {
        for (int j : greedy_jobs) {
            JobPool &P = pools[j];
            P.cur.clear(); P.nxt.clear(); P.fin.clear(); P.evals = 0; P.cursize = 1;
            P.cursubopt = b->psets[b->job_pset[j]].suboptmin;
            P.cur.emplace_back(); P.cur.back().job = j;       // :1105 one empty structure
        }
    };
    if (!use_chain && !use_pool) host_pools_init();
    for (int k = 0; k < 8; k++) g_t[k] = 0;
    const bool timing = getenv("SQ_TIMING") != nullptr;
    auto mark = [&](const char *what) { if (timing) fprintf(stderr, "[sq_fold]   +%.3f ms %s\n", (now_s() - tfold0) * 1e3, what); };
    // a-10 tail per sequence
    std::vector<std::vector<int32_t>> seq_jobs(b->nseq);
    for (int j = 0; j < b->njobs; j++) seq_jobs[b->job_seq[j]].push_back(j);
    std::vector<double> tail_cost(b->nseq, 0.0);
    mark("job lists");
    auto tail_one = [&](int s) {
        CpuScope cpu_(0);
        const double tt0 = timing ? now_s() : 0;
        struct TT { bool on; double t0; double &dst; ~TT() { if (on) dst = now_s() - t0; } } tt{timing, tt0, tail_cost[s]};
        std::vector<const std::vector<std::vector<HStem>> *> per_job;   // (freed later by the thread that allocated them)
        int64_t ev = 0;
        for (int j : seq_jobs[s]) { per_job.push_back(&pools[j].fin); ev += pools[j].evals; }
        const bool hr = has_ref && has_ref[s];
        const int32_t *rp = hr ? ref_pairs + 2 * (size_t)ref_off[s] : nullptr;
        const int nref = hr ? ref_off[s + 1] - ref_off[s] : 0;
        b->results[s] = SeqResult();
        sq_tail(b, s, o, per_job, seq_jobs[s], rp, nref, hr, b->results[s]);
        b->results[s].evals = ev;
    };
    std::vector<char> tailed(b->nseq, 0);
    // Early tails: without E/H/N stemsets a sequence is complete the moment the pools of its greedy jobs are empty;
    // the lanes report such sequences after every round and a helper thread ranks them on the worker pool while
    // the rounds of the other sequences go on.
    static const bool no_early_tail = getenv("SQ_NO_EARLY_TAIL") != nullptr;
    const bool early_tail = pending == nullptr && !no_early_tail && !dev_tail;
    struct TailQueue {
        std::mutex mu; std::condition_variable cv; std::vector<int> items; bool closed = false;
        std::thread worker;
        void push(std::vector<int> &v) { if (v.empty()) return; { std::lock_guard<std::mutex> lk(mu); items.insert(items.end(), v.begin(), v.end()); } cv.notify_one(); v.clear(); }
        void close() { if (!worker.joinable()) return; { std::lock_guard<std::mutex> lk(mu); closed = true; } cv.notify_one(); worker.join(); }
        ~TailQueue() { close(); }
    } tq;
    std::vector<std::atomic<int>> g_left(early_tail ? b->nseq : 0);
    std::vector<char> job_done(early_tail ? b->njobs : 0, 0);
    // chained rounds: entry q of the device's list of finished structures (job | stems << 32 | by-count << 63) becomes
    // the job's final stem list; handled by the queue's workers so that the thread that enqueues the rounds never waits
    auto chain_finish = [&](uint32_t q) {
        const unsigned long long e = b->chain.h_fin[q];
        if ((e >> 62) & 1ull) return;                       // a structure that stopped at a tie: the device pools fold its job
        const int j = (int)(uint32_t)e, nst = (int)((e >> 32) & 0x3FFFFFFFu);
        const bool by_count = (e >> 63) != 0;
        JobPool &P = pools[j];
        static_assert(sizeof(HStem) == sizeof(SqStemOut), "stem records must match");
        std::vector<HStem> stems((size_t)nst);
        if (nst) memcpy(stems.data(), b->chain.h_stems + b->chain_toff[j], sizeof(HStem) * (size_t)nst);
        P.fin.push_back(std::move(stems));
        P.evals += nst + (by_count ? 0 : 1);                // one evaluation per round the structure took part in
        const int s2 = b->job_seq[j];
        if (early_tail && --g_left[s2] == 0) { tail_one(s2); tailed[s2] = 1; }
    };
    if (early_tail) {
        for (int s2 = 0; s2 < b->nseq; s2++) g_left[s2] = 0;
        for (int j : greedy_jobs) g_left[b->job_seq[j]]++;
    }
    if (early_tail || use_chain || chain_ties) {
        sq_pool(b);
        tq.worker = std::thread([&] {
            if (b->device >= 0) hipSetDevice(b->device);
            for (;;) {
                std::vector<int> take;
                {
                    std::unique_lock<std::mutex> lk(tq.mu);
                    tq.cv.wait(lk, [&] { return !tq.items.empty() || tq.closed; });
                    take.swap(tq.items);
                    if (take.empty()) return;               // closed and drained
                }
                sq_pool(b)->parallel_for((int)take.size(), [&](int k) {
                    if (take[k] < 0) chain_finish((uint32_t)(-(take[k] + 1)));       // (items < 0: chain entries)
                    else { tail_one(take[k]); tailed[take[k]] = 1; }
                });
            }
        });
    }
    mark("tail queue");
    // the greedy pool loop (:1102-1199) for a subset of the jobs, on one lane of round buffers
    struct LoopStats { double tround = 0, twall = 0, tstart = 0; int nrounds = 0; int rc = 0; std::string err; };
    auto greedy_loop = [&](SqLane &ln, const std::vector<int> &myjobs, LoopStats &stats) {
        std::vector<SView> round;
        std::vector<int> owner;                             // job of each view
        std::vector<std::vector<HStem>> res;
        std::vector<int> finished;                          // sequences completed since the last report
        auto job_finished = [&](int j) {
            if (!early_tail || job_done[j]) return;
            job_done[j] = 1;
            if (--g_left[b->job_seq[j]] == 0) finished.push_back(b->job_seq[j]);
        };
        const double tl0 = now_s();
        stats.tstart = tl0 - tfold0;
        struct Wall { double t0; double &dst; ~Wall() { dst = now_s() - t0; } } wall{tl0, stats.twall};
        for (;;) {
            round.clear(); owner.clear();
            for (int j : myjobs) {
                JobPool &P = pools[j];
                if (P.cur.empty()) { job_finished(j); continue; }
                if (P.cur.size() > P.cursize) {             // :1162-1165
                    P.cursize = P.cur.size();
                    if (P.cursubopt < P.suboptmax) P.cursubopt += P.suboptinc;
                }
                bool anyfull = false;                       // :1168-1174
                for (auto &s : P.cur) if ((double)s.stems.size() == P.maxstemnum) { anyfull = true; break; }
                if (anyfull) {
                    std::vector<HStruct> keep;
                    for (auto &s : P.cur) {
                        if ((double)s.stems.size() == P.maxstemnum) P.fin.push_back(std::move(s.stems));
                        else keep.push_back(std::move(s));
                    }
                    P.cur.swap(keep);
                    if (P.cur.empty()) { job_finished(j); continue; }
                }
                for (size_t k = 0; k < P.cur.size(); k++) {
                    round.push_back(SView{j, P.cursubopt, &P.cur[k]});
                    owner.push_back(j);
                }
                P.evals += (int64_t)P.cur.size();
            }
            tq.push(finished);
            if (round.empty()) break;
            { const double t0 = now_s(); stats.rc = run_round_impl(b, ln, round, 0, res, nullptr); stats.tround += now_s() - t0; stats.nrounds++; }
            if (stats.rc) { stats.err = sq_last_error(); return; }
            // :1179-1196.  The entries of one job are contiguous in `round` and only touch that job's pool, so jobs
            // are independent; per job the entries are still handled in order.  Big rounds are shared among the
            // worker pool in contiguous slices (children mostly reuse their parent's storage: no allocator traffic).
            auto grow = [&](size_t q0, size_t q1) {
                CpuScope cpu_(3);
                for (size_t q = q0; q < q1; q++) {
                    const int j = owner[q];
                    JobPool &P = pools[j];
                    const std::vector<HStem> &news = res[q];
                    const HStruct &parent = *round[q].st;
                    if (!news.empty()) {
                        const size_t stopper = P.cursize >= (size_t)o.poollim ? 1 : news.size();
                        for (size_t k = 0; k < stopper; k++) {
                            P.nxt.emplace_back();
                            sq_extend_struct(parent, news[k], P.nxt.back(), k + 1 == stopper);   // the last child inherits the vectors
                        }
                    } else {
                        P.fin.push_back(std::move(const_cast<HStruct &>(parent).stems));   // the structure is final and leaves the pool
                    }
                }
                for (size_t q = q0; q < q1; q++)
                    if (q == q0 || owner[q] != owner[q - 1]) {   // once per job of the slice
                        JobPool &P = pools[owner[q]];
                        P.cur.swap(P.nxt);
                        P.nxt.clear();                      // (capacity stays)
                    }
            };
            static const size_t par_min = getenv("SQ_GROW_PAR") ? (size_t)atol(getenv("SQ_GROW_PAR")) : 1024;
            if (round.size() >= par_min) {
                const int nsl = sq_pool(b)->size() * 4;
                std::vector<size_t> cut(nsl + 1);
                for (int t = 0; t <= nsl; t++) {
                    size_t q = round.size() * (size_t)t / (size_t)nsl;
                    while (q > 0 && q < round.size() && owner[q] == owner[q - 1]) q++;   // slices end on job boundaries
                    cut[t] = q;
                }
                sq_pool(b)->parallel_for(nsl, [&](int t) { if (cut[t] < cut[t + 1]) grow(cut[t], cut[t + 1]); }, round.size() >= 2048 ? 1 : 0);
            } else grow(0, round.size());
        }
    };
    // Two lanes when the batch is big enough: the jobs are dealt alternately (by sequence) to two host threads, each
    // driving its rounds on half of the round buffers; the kernels of both queue on the batch stream, so while one
    // lane's host code books a round the other lane's kernels run.  Jobs are independent: same results.
    static const int want_lanes = getenv("SQ_FOLD_LANES") ? atoi(getenv("SQ_FOLD_LANES")) : 2;
    static const int lane_min_jobs = getenv("SQ_LANE_MIN_JOBS") ? atoi(getenv("SQ_LANE_MIN_JOBS")) : 512;
    const bool two_lanes = want_lanes >= 2 && !b->prof_on && (int)greedy_jobs.size() >= lane_min_jobs &&
                           (int)greedy_jobs.size() <= b->max_structs;   // (a lane holds half of the slots)
    LoopStats st0, st1;
    sq_pool(b);                                             // (created before any second thread can ask for it)
    // ---- device-chained rounds ----
    auto chain_fold = [&](LoopStats &stats) {
        SqLane &ln = b->lane_full;
        hipStream_t st = b->stream;
        const double tl0 = now_s();
        stats.tstart = tl0 - tfold0;
        struct Wall { double t0; double &dst; ~Wall() { dst = now_s() - t0; } } wall{tl0, stats.twall};
        auto fail = [&](int rc, const std::string &msg) { stats.rc = rc; stats.err = msg; };
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail(sq_check(e_, #x), sq_last_error()); return; } } while (0)
        if (!b->chain.h_stems) {
            void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *p3 = nullptr;
            if (sq_pinned_get(&p0, sizeof(SqStemOut) * (size_t)std::max<int64_t>(b->chain_T, 1)) ||
                sq_pinned_get(&p1, 8 * (size_t)b->njobs) || sq_pinned_get(&p2, 64) ||
                sq_pinned_get(&p3, sizeof(SqChain) * (size_t)b->njobs)) { fail(2, sq_last_error()); return; }
            b->chain.h_stems = (SqStemOut *)p0; b->chain.h_fin = (unsigned long long *)p1;
            b->chain.h_nfin = (volatile uint32_t *)p2; b->h_chain = (SqChain *)p3;
            b->chain_toff.resize(b->njobs);
            int32_t t = 0;
            for (int j = 0; j < b->njobs; j++) { b->chain_toff[j] = t; t += chain_tcap(b->jobs[j].n, b->psets[b->job_pset[j]].minlen); }
        }
        std::vector<int> finished;                          // queue items: sequences to rank (>= 0), chain entries (< 0)
        auto job_finished = [&](int j) {
            if (early_tail && --g_left[b->job_seq[j]] == 0) finished.push_back(b->job_seq[j]);
        };
        *b->chain.h_nfin = 0;
        bool first_chain = true;
        uint32_t nfin_seen = 0, nfin_goal = 0;              // entries of the finished list: handed on / expected after this chain
        // as many structures per chain as the round buffers hold at once (one chain after the other)
        const int64_t avail = b->cand_records - b->cand_reserved;
        size_t next_job = 0;
        while (next_job < chain_jobs.size() && !stats.rc) {
        std::vector<int> jobs;                              // structure index -> job
        int maxn = 0, maxt = 0; int64_t cand_off = 0, maxcap = 0; bool need_reacts = false;
        for (; next_job < chain_jobs.size(); next_job++) {
            const int j = chain_jobs[next_job];
            JobPool &P = pools[j];
            if (P.maxstemnum == 0) { P.fin.emplace_back(); job_finished(j); continue; }   // :1168-1174 full before the first round
            const SqJob &J = b->jobs[j];
            if ((int)jobs.size() == ln.max_structs || cand_off + J.cand_cap > avail) break;
            const int sx = (int)jobs.size();
            SqStruct &d = ln.h_structs[sx];
            d.job = j; d.slot = sx; d.subopt = P.cursubopt; d.cand_off = cand_off;
            cand_off += J.cand_cap; maxcap = std::max<int64_t>(maxcap, J.cand_cap);
            SqChain &cr = b->h_chain[sx];
            cr.toff = b->chain_toff[j]; cr.tcap = chain_tcap(J.n, b->psets[b->job_pset[j]].minlen);
            cr.nstems = 0; cr.anycross = 0; cr.maxstems = P.maxstemnum;
            d.strand_off = 4 * cr.toff; d.nstrand = 0;
            maxn = std::max(maxn, J.n); maxt = std::max(maxt, cr.tcap);
            need_reacts |= !J.default_reacts && !(J.react_levels > 0 && b->pset_classes[J.pset] * J.react_levels <= 32);
            jobs.push_back(j);
        }
        tq.push(finished);
        const int S = (int)jobs.size();
        if (S == 0) continue;
        nfin_goal += (uint32_t)S;
        SqRoundIO io;
        io.h_structs = ln.d_structs; io.h_strands = b->chain.strands; io.d_structs = ln.d_structs; io.d_strands = b->chain.strands;
        io.h_out = ln.h_out; io.d_out = ln.d_out; io.h_cap = 0; io.out_cap = 0;
        io.h_ctr = ln.h_ctr; io.h_seq = ln.h_seq;
        SqScanArgs scan = b->scan;
        scan.ctr = ln.d_ctr;
        hipLaunchKernelGGL(sq_chain_init_kernel, dim3((S + 255) / 256), dim3(256), 0, st, ln.h_structs, b->h_chain, ln.d_structs,
                           b->chain, scan, S, first_chain ? 1 : 0);
        first_chain = false;
        static const uint32_t depth = getenv("SQ_CHAIN_DEPTH") ? (uint32_t)std::max(1, atoi(getenv("SQ_CHAIN_DEPTH"))) : 3;
        const uint32_t seq0 = *ln.round_seq;
        uint32_t launched = 0, done = 0;
        const bool relaxed = sq_relaxed_waits(b);
        const uint64_t poll_mask = relaxed ? 0x3FFF : 0xFFFFF;
        uint64_t spins = 0;
        volatile uint32_t *flag = ln.h_seq;
        const double tr0 = now_s();
        std::vector<std::pair<int, double>> round_t;
        // ONE launch for all rounds of these structures (sq_rounds.hip: a persistent block per structure) when every job
        // qualifies: no dense matrix behind its cells, per-position arrays that fit the block's LDS
        bool rounds_ok = !no_rounds;
        for (int j : jobs) rounds_ok = rounds_ok && b->jobs[j].n <= SQ_ROUNDS_MAXN;
        if (rounds_ok) {
            static const int thr_env = getenv("SQ_ROUNDS_THREADS") ? std::max(64, std::min(SQ_ROUNDS_THREADS, atoi(getenv("SQ_ROUNDS_THREADS")) / 64 * 64)) : 0;
            // threads per structure: by length -- and, while the launch leaves the chip empty (a shard of a multi-GPU run, a
            // small batch), twice / four times that: a structure's rounds are a chain of dependent passes over its list that
            // more waves shorten (S1000 x 128: 1.21 -> 0.99 ms at 512 threads)
            int thr = maxn <= 320 ? 64 : (maxn <= 450 ? 128 : 256);
            while (!thr_env && thr < SQ_ROUNDS_THREADS && thr < maxn / 2 && (int64_t)S * thr * 2 <= (int64_t)256 * 512) thr *= 2;
            if (thr_env) { thr = 64; while (thr * 2 <= thr_env) thr *= 2; }   // (a power of two: the survivor ring is indexed with a mask)
            SqRoundsArgs ra;
            ra.lds_n = maxn; ra.str_cap = 2 * maxt + 2; ra.tmax = maxt; ra.cell_entries = b->cell_entries;
            ra.bound = b->score_bound ? 1 : 0; ra.ctx_min = 0; ra.ties = chain_ties ? 1 : 0;
            while (thr > 64 && sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, thr).total + 2048 > 158 * 1024) thr /= 2;   // (long sequences: the survivor ring gives way)
            const SqRoundsLds lo = sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, thr);
            if (lo.total + 2048 > 158 * 1024) rounds_ok = false;
            else {
                if (lo.total > 60 * 1024) sq_max_dynamic_lds((const void *)sq_rounds_kernel, 158 * 1024);   // (the kernel has static LDS too: 160 KB in all)
                {
                    ProfScope ps(b, 7, 0);
                    hipLaunchKernelGGL(sq_rounds_kernel, dim3(S), dim3(thr), lo.total, st, b->ctx, ln.d_structs, scan, b->chain, ra);
                }
                { const hipError_t le = hipGetLastError(); if (le != hipSuccess) { hipFuncAttributes fa; memset(&fa, 0, sizeof(fa)); hipFuncGetAttributes(&fa, (const void *)sq_rounds_kernel); fprintf(stderr, "[sq_fold] persistent rounds launch: S %d threads %d LDS %zu | kernel: maxThreadsPerBlock %d numRegs %d static LDS %zu maxDynamic %d local %zu\n", S, thr, lo.total, fa.maxThreadsPerBlock, fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes, fa.localSizeBytes); fail(sq_check(le, "persistent rounds launch"), sq_last_error()); } }
                const uint32_t seq = ++*ln.round_seq;
                hipLaunchKernelGGL(sq_chain_done_kernel, dim3(1), dim3(1), 0, st, io, scan, b->chain, seq);
                launched = 1;
                b->last_paths |= 4;
                while (*flag != seq) {
                    if ((++spins & poll_mask) == 0) {
                        const hipError_t q = hipStreamQuery(st);
                        if (q != hipErrorNotReady && q != hipSuccess) { fail(sq_check(q, "persistent rounds"), sq_last_error()); break; }
                        if (q == hipSuccess && *flag != seq) { fail(2, "persistent rounds did not signal completion"); break; }
                    }
                    sq_wait_step(spins, relaxed);
                }
                if (!stats.rc) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    const SqCounters ctr = *ln.h_ctr;
                    if (ctr.cand_ovf) fail(-3, "candidate capacity exceeded (raise cand_per_nt)");
                    else if (ctr.out_ovf) fail(-3, "stem capacity of a chained structure exceeded");
                    else if (ctr.level_ovf) fail(-3, "more than 64 pseudoknot levels");
                    else {
                        const uint32_t nf = *b->chain.h_nfin;
                        if (nf != nfin_goal) fail(2, "persistent rounds left structures unfinished");
                        if (chain_ties) for (uint32_t q = nfin_seen; q < nf; q++) if ((b->chain.h_fin[q] >> 62) & 1ull) tied_jobs.push_back((int)(uint32_t)b->chain.h_fin[q]);
                        if (!dev_tail) for (uint32_t q = nfin_seen; q < nf; q++) finished.push_back(-(int)q - 1);
                        nfin_seen = nf;
                        tq.push(finished);
                    }
                }
            }
        }
        if (!rounds_ok && chain_ties) {                       // (the launched rounds do not look for ties: the pools take these jobs)
            for (int j : jobs) tied_jobs.push_back(j);
            nfin_goal -= (uint32_t)S;
            continue;
        }
        while (!rounds_ok && nfin_seen < nfin_goal) {
            while (launched - done < depth) {               // rounds enqueued ahead of the device
                if ((int)launched > maxt + 2) { fail(2, "chained rounds do not terminate"); break; }
                // (algorithmic bytes: NOT per launch -- a launch also covers the structures that are already final; they are
                // booked below from the evaluations the list of finished structures records)
                launch_round_kernels(b, st, S, maxn, maxcap, need_reacts, 0.0, 0, io, scan, ln.d_structs, b->chain.strands, true);
                const uint32_t seq = ++*ln.round_seq;
                hipLaunchKernelGGL(sq_chain_done_kernel, dim3(1), dim3(1), 0, st, io, scan, b->chain, seq);
                launched++;
            }
            if (stats.rc) break;
            const uint32_t d2 = *flag - seq0;
            if (d2 != done && d2 <= launched) {
                std::atomic_thread_fence(std::memory_order_acquire);
                done = d2; spins = 0;
                if (timing) round_t.push_back({(int)done, (now_s() - tr0) * 1e3});
                const SqCounters ctr = *ln.h_ctr;
                if (ctr.cand_ovf) { fail(-3, "candidate capacity exceeded (raise cand_per_nt)"); break; }
                if (ctr.out_ovf) { fail(-3, "stem capacity of a chained structure exceeded"); break; }
                if (ctr.level_ovf) { fail(-3, "more than 64 pseudoknot levels"); break; }
                const uint32_t nf = *b->chain.h_nfin;
                if (!dev_tail) for (uint32_t q = nfin_seen; q < nf; q++) finished.push_back(-(int)q - 1);   // (device tail: the log has them)
                nfin_seen = nf;
                tq.push(finished);
                continue;
            }
            if ((++spins & poll_mask) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q != hipErrorNotReady && q != hipSuccess) { fail(sq_check(q, "chained rounds"), sq_last_error()); break; }
                if (q == hipSuccess && *flag - seq0 != launched) { fail(2, "chained round did not signal completion"); break; }
            }
            sq_wait_step(spins, relaxed);
        }
        stats.nrounds += (int)launched;
        if (timing && (now_s() - tr0) > 8e-3) {
            fprintf(stderr, "[sq_fold] slow chain:");
            for (auto &rt : round_t) fprintf(stderr, " r%d@%.2f", rt.first, rt.second);
            fprintf(stderr, "\n");
        }
        // rounds still in flight find no live structure; they must be through before the buffers are used again
        hipStreamSynchronize(st);
        stats.tround += now_s() - tr0;
        if (b->prof_on && !stats.rc) {
            // 2 N^2 bytes per AnnotateStems evaluation = per round a structure was LIVE in (its stems + the
            // round that found none); exactly what sq_result_evals reports
            double bytes = 0;
            for (uint32_t q = nfin_goal - (uint32_t)S; q < nfin_goal; q++) {
                const unsigned long long e = b->chain.h_fin[q];
                if ((e >> 62) & 1ull) continue;
                const double n = b->jobs[(int)(uint32_t)e].n;
                const double ev = (double)((e >> 32) & 0x3FFFFFFFu) + ((e >> 63) ? 0.0 : 1.0);
                bytes += ev * 2.0 * n * n;
            }
            b->prof[rounds_ok ? 7 : 2].bytes += bytes;        // (the persistent round kernel covers the evaluations of all its rounds)
        }
        }
#undef CHK
    };
    // ---- device pools ----
    std::vector<int> pool_jobs;                              // structure slot of generation 0 -> job
    std::function<int()> pool_collect;                       // set by pool_fold: the device log -> pools[].fin (host tail only)
    auto pool_fold = [&](LoopStats &stats) -> int {          // 0: done, 1: capacity overflow (repeat on the host), < 0 / > 1: error in stats
        SqLane &ln = b->lane_full;
        hipStream_t st = b->stream;
        SqPoolIO &PI = b->pool_io;
        const double tl0 = now_s();
        stats.tstart = tl0 - tfold0;
        struct Wall { double t0; double &dst; ~Wall() { dst = now_s() - t0; } } wall{tl0, stats.twall};
        auto fail = [&](int rc, const std::string &msg) { stats.rc = rc; stats.err = msg; return 2; };
        if (!PI.h_hdr) {
            void *p2 = nullptr, *p3 = nullptr, *p4 = nullptr, *p5 = nullptr, *p6 = nullptr;
            if (sq_pinned_get(&p2, 64) || sq_pinned_get(&p3, sizeof(SqPoolJob) * (size_t)b->njobs) ||
                sq_pinned_get(&p4, sizeof(SqChain) * (size_t)b->njobs) || sq_pinned_get(&p5, sizeof(SqPoolJob) * (size_t)b->njobs) ||
                sq_pinned_get(&p6, 4 * (size_t)b->njobs)) return fail(2, sq_last_error());
            PI.h_hdr = (SqPoolHdr *)p2; PI.h_jobs = (SqPoolJob *)p3;
            b->h_pool_recs = (SqChain *)p4; b->h_pool_jobs = (SqPoolJob *)p5; b->h_pool_jobrec = (int32_t *)p6;
        }
        std::vector<int> jobs;
        int maxn = 0; int64_t maxcap = 0; bool need_reacts = false;
        for (int j : pool_jobs_v) {
            JobPool &P = pools[j];
            if (P.maxstemnum == 0) { P.fin.emplace_back(); continue; }   // :1123-1129 full before the first round
            const SqJob &J = b->jobs[j];
            maxn = std::max(maxn, J.n); maxcap = std::max<int64_t>(maxcap, J.cand_cap);
            need_reacts |= !J.default_reacts && !(J.react_levels > 0 && b->pset_classes[J.pset] * J.react_levels <= 32);
            jobs.push_back(j);
        }
        const int S0 = (int)jobs.size();
        if (S0 == 0) return 0;
        const int64_t avail = b->cand_records - b->cand_reserved;
        int slots = std::min(PI.smax, ln.max_structs);
        if (const char *e = getenv("SQ_POOL_SLOTS")) slots = std::min(slots, std::max(1, atoi(e)));   // (tests: force the overflow path)
        // structures whose candidates fit the arena at once; larger generations go through state .. choose in chunks
        int chunk = (int)std::min<int64_t>(slots, avail / std::max<int64_t>(maxcap, 1));
        if (const char *e = getenv("SQ_POOL_CHUNK")) chunk = std::min(chunk, std::max(1, atoi(e)));   // (tests: force chunked rounds)
        if (S0 > slots || chunk < 1) return 1;
        for (int j = 0; j < b->njobs; j++) b->h_pool_jobrec[j] = -1;
        for (int sx = 0; sx < S0; sx++) {
            const int j = jobs[sx];
            const JobPool &P = pools[j];
            const int toff = sx * PI.pt;                     // generation 0, slot sx
            SqStruct &d = ln.h_structs[sx];
            d.job = j; d.strand_off = 2 * toff; d.nstrand = 0; d.slot = sx; d.subopt = P.cursubopt; d.cand_off = (int64_t)(sx % chunk) * maxcap;
            SqChain &cr = b->h_pool_recs[sx];
            cr.toff = toff; cr.tcap = PI.pt; cr.nstems = 0; cr.anycross = 0; cr.maxstems = P.maxstemnum;
            SqPoolJob &pj = b->h_pool_jobs[sx];
            pj.first = sx; pj.count = 1; pj.cursize = 1; pj.job = j;
            pj.cursubopt = P.cursubopt; pj.suboptinc = P.suboptinc; pj.suboptmax = P.suboptmax; pj.maxstems = P.maxstemnum; pj.evals = 0;
            b->h_pool_jobrec[j] = sx;
        }
        PI.slots = slots; PI.chunk = chunk; PI.poollim = o.poollim; PI.maxcap = maxcap; PI.njobs = S0;   // (the kernels take the batch's record)
        const SqPoolIO pio = PI;
        SqScanArgs scan = b->scan;
        scan.ctr = ln.d_ctr;
        hipLaunchKernelGGL(sq_pool_init_kernel, dim3((std::max(S0, b->njobs) + 255) / 256), dim3(256), 0, st, ln.h_structs, b->h_pool_recs,
                           b->h_pool_jobs, b->h_pool_jobrec, (int32_t *)pio.jobrec_of, b->njobs, pio, scan, S0);
        const bool relaxed = sq_relaxed_waits(b);
        const uint64_t poll_mask = relaxed ? 0x3FFF : 0xFFFFF;
        volatile uint32_t *flag = ln.h_seq;
        auto wait_seq = [&](uint32_t seq) -> int {
            uint64_t spins = 0;
            while (*flag != seq) {
                if ((++spins & poll_mask) == 0) {
                    const hipError_t q = hipStreamQuery(st);
                    if (q != hipErrorNotReady) {
                        if (q != hipSuccess) return fail(sq_check(q, "pool rounds"), sq_last_error());
                        if (*flag != seq) { hipStreamSynchronize(st); if (*flag != seq) return fail(2, "pool round did not signal completion"); }
                    }
                }
                sq_wait_step(spins, relaxed);
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            return 0;
        };
        // short sequences on a crowded chip: a round is ONE kernel (sq_pool_round.hip) + the scan kernel
        const bool crowded_fold = b->inflight > 1 || b->njobs >= 4096;
        SqPoolRoundArgs pra;
        bool round_kernel = maxn <= SQ_PR_MAXN && !b->any_dense && (crowded_fold || b->pool_round_always) && !b->no_pool_round;
        if (round_kernel) {
            pra.lds_n = maxn; pra.str_cap = 2 * pio.pt + 2; pra.cell_entries = b->cell_entries;
            pra.surv_cap = b->pool_round_nsurv ? b->pool_round_nsurv : (maxn <= 96 ? 128 : 256); pra.bound = b->score_bound ? 1 : 0;
            pra.tmax = pio.pt; pra.parity = 0; pra.lo = 0;
            if (sq_pool_round_lds(pra.lds_n, pra.str_cap, pra.cell_entries, pra.surv_cap, pra.tmax).total > 60 * 1024) round_kernel = false;
        }
        if (round_kernel) b->last_paths |= 8;
        const size_t ext_lds = sq_extend_lds_bytes(pio.pt);          // the extend kernel's level scratch (dynamic LDS)
        if (ext_lds > 64 * 1024) sq_max_dynamic_lds((const void *)sq_pool_extend_kernel, 160 * 1024);
        const double tr0 = now_s();
        int parity = 0, S = S0, rounds = 0;
        bool overflow = false;
        while (S > 0) {
            b->last_peak = std::max<int64_t>(b->last_peak, S);
            SqStruct *cur = pio.structs + (size_t)parity * pio.smax;
            SqRoundIO io;
            io.h_strands = pio.strands; io.d_strands = pio.strands;
            io.h_out = ln.h_out; io.d_out = ln.d_out; io.h_cap = 0; io.out_cap = 0;
            io.h_ctr = ln.h_ctr; io.h_seq = ln.h_seq;
            for (int lo = 0; lo < S; lo += chunk) {              // (stream order: a chunk's chosen stems are out before the next one reuses the arena)
                io.h_structs = cur + lo; io.d_structs = cur + lo;
                pra.parity = parity; pra.lo = lo;
                launch_round_kernels(b, st, std::min(chunk, S - lo), maxn, maxcap, need_reacts, 0.0, 0, io, scan, cur + lo, pio.strands, true, true,
                                     round_kernel ? &pra : nullptr);
            }
            const uint32_t seq = ++*ln.round_seq;
            hipLaunchKernelGGL(sq_pool_scan_kernel, dim3(1), dim3(1024), 0, st, pio, scan, io, parity, seq);
            // (4 waves share a parent's children; on a crowded chip ONE takes them all: most parents have one or two, and a wave
            // that finds nothing to do still takes a slot for a microsecond or two -- 593 k -> 601 k)
            static const int ext_crowd = getenv("SQ_POOL_EXTEND_WAVES") ? std::max(1, std::min(16, atoi(getenv("SQ_POOL_EXTEND_WAVES")))) : 1;
            const bool crowded = b->inflight > 1 || b->njobs >= 4096;
            if (!round_kernel)       // (the round kernel's structures extend themselves and log themselves)
                hipLaunchKernelGGL(sq_pool_extend_kernel, dim3(S, crowded ? ext_crowd : 4), dim3(64), ext_lds, st, b->ctx, scan, pio, parity);
            if (wait_seq(seq)) return 2;
            rounds++;
            const SqCounters ctr = *ln.h_ctr;
            if (ctr.cand_ovf) return fail(-3, "candidate capacity exceeded (raise cand_per_nt)");
            if (ctr.level_ovf) return fail(-3, "more than 64 pseudoknot levels");
            const SqPoolHdr hh = *pio.h_hdr;
            if (timing && getenv("SQ_POOL_DEBUG")) fprintf(stderr, "[pool] round %d: S %d -> %u, nfin %u, ovf %u, active jobs %u\n", rounds, S, hh.S[parity ^ 1], hh.nfin, hh.ovf, hh.active_jobs);
            if (hh.ovf) { overflow = true; break; }
            parity ^= 1;
            S = (int)hh.S[parity];
            if (rounds > 4 * PI.pt + 8) return fail(2, "pool rounds do not terminate");
        }
        {   // the last extend kernel's log entries and flags, the evaluation counts
            SqRoundIO io;
            io.h_structs = pio.structs; io.h_strands = pio.strands; io.d_structs = pio.structs; io.d_strands = pio.strands;
            io.h_out = ln.h_out; io.d_out = ln.d_out; io.h_cap = 0; io.out_cap = 0; io.h_ctr = ln.h_ctr; io.h_seq = ln.h_seq;
            const uint32_t seq = ++*ln.round_seq;
            hipLaunchKernelGGL(sq_pool_publish_kernel, dim3(1), dim3(256), 0, st, pio, scan, io, seq);
            if (wait_seq(seq)) return 2;
        }
        stats.nrounds = rounds;
        stats.tround = now_s() - tr0;
        const SqPoolHdr hh = *pio.h_hdr;
        if (overflow || hh.ovf) {
            for (int j : greedy_jobs) { pools[j].fin.clear(); pools[j].evals = 0; }
            // (the device log holds the structures the aborted pools had finished: empty it for the host loop's)
            hipLaunchKernelGGL(sq_fold_begin_kernel, dim3((b->njobs + 256) / 256), dim3(256), 0, st, b->d_fin_ctr, b->d_job_evals, b->tail.job_cnt, b->njobs);
            return 1;
        }
        if ((*ln.h_ctr).level_ovf) return fail(-3, "more than 64 pseudoknot levels");
        // finstemsets of every job: its log entries in (round, kind, position) order.  With the device tail the log is
        // consumed where it is; the host needs it only when the batch falls back to the host tail (pool_collect).
        pool_jobs = jobs;
        pool_collect = [&, S0, hh]() -> int {
            std::vector<SqPoolFin> Fv(hh.nfin);
            std::vector<SqPoolStem> Sv(hh.nfin_stems);
            if (hh.nfin) HIPCK(hipMemcpy(Fv.data(), b->d_fin, sizeof(SqPoolFin) * (size_t)hh.nfin, hipMemcpyDeviceToHost));
            if (hh.nfin_stems) HIPCK(hipMemcpy(Sv.data(), b->d_fin_stems, sizeof(SqPoolStem) * (size_t)hh.nfin_stems, hipMemcpyDeviceToHost));
            const SqPoolFin *F = Fv.data();
            std::vector<uint32_t> start((size_t)b->njobs + 1, 0), ord(hh.nfin);
            // (entries below SQ_FIN_KIND_G0 are E / H / N stemsets: not the pools')
            for (uint32_t q = 0; q < hh.nfin; q++) if (F[q].round_kind >= SQ_FIN_KIND_G0) start[(size_t)F[q].job + 1]++;
            for (int j = 0; j < b->njobs; j++) start[(size_t)j + 1] += start[j];
            {
                std::vector<uint32_t> fillp(start.begin(), start.end() - 1);
                for (uint32_t q = 0; q < hh.nfin; q++) if (F[q].round_kind >= SQ_FIN_KIND_G0) ord[fillp[F[q].job]++] = q;
            }
            auto one_job = [&](int sx) {
                const int j = pool_jobs[sx];
                uint32_t *p0 = ord.data() + start[j], *p1 = ord.data() + start[(size_t)j + 1];
                std::sort(p0, p1, [&](uint32_t x, uint32_t y) {
                    if (F[x].round_kind != F[y].round_kind) return F[x].round_kind < F[y].round_kind;
                    return F[x].pos < F[y].pos;
                });
                auto &fin = pools[j].fin;
                fin.reserve(fin.size() + (size_t)(p1 - p0));
                for (uint32_t *p = p0; p < p1; p++) {
                    const SqPoolFin &e = F[*p];
                    const SqPoolStem *src = Sv.data() + e.stem_off;
                    std::vector<HStem> stems((size_t)e.nstems);
                    for (int t = 0; t < e.nstems; t++) stems[t] = HStem{src[t].i, src[t].j, src[t].len, 0.0, 0.0};
                    fin.push_back(std::move(stems));
                }
            };
            if (hh.nfin >= 8192) sq_pool(b)->parallel_for(S0, one_job);
            else for (int sx = 0; sx < S0; sx++) one_job(sx);
            return 0;
        };
        if (!dev_tail) { const int rc2 = pool_collect(); pool_collect = nullptr; if (rc2) return fail(rc2, sq_last_error()); }
        if (!dev_tail) for (int sx = 0; sx < S0; sx++) pools[jobs[sx]].evals += pio.h_jobs[sx].evals;
        if (b->prof_on)                                      // 2 N^2 bytes per evaluation (live structures only)
            for (int sx = 0; sx < S0; sx++) { const double n = b->jobs[jobs[sx]].n; b->prof[2].bytes += (double)pio.h_jobs[sx].evals * 2.0 * n * n; }
        return 0;
    };
    mark("loop start");
    b->last_driver = use_pool ? 2 : use_chain ? 1 : 0;
    b->last_peak = use_chain ? (int64_t)greedy_jobs.size() : 0;
    if (use_pool && chain_ties) {
        // the optimistic chains first; their structures that met a tie hand their jobs to the pools
        chain_fold(st0);
        if (st0.rc) { tq.close(); sq_set_error(st0.err); return st0.rc; }
        b->last_paths |= 16;
        std::sort(tied_jobs.begin(), tied_jobs.end());
        pool_jobs_v.insert(pool_jobs_v.end(), tied_jobs.begin(), tied_jobs.end());
        std::sort(pool_jobs_v.begin(), pool_jobs_v.end());
        if (timing) fprintf(stderr, "[sq_fold] optimistic chains: %zu jobs, %zu met a tie and go to the device pools (with %zu others)\n",
                            chain_jobs.size(), tied_jobs.size(), pool_jobs_v.size() - tied_jobs.size());
        st0 = LoopStats();
    }
    if (use_pool) {
        const int pr = pool_jobs_v.empty() ? 0 : pool_fold(st0);
        if (pr == 1) { b->last_driver = 3; b->last_peak = 0; }
        if (pr == 1 && timing) fprintf(stderr, "[sq_fold] device pools: a capacity was exceeded, the host loop repeats the greedy part\n");
        if (pr == 1) {                                       // a capacity was exceeded: the host's own loop takes the fold
            st0 = LoopStats();
            use_pool = false;
            host_pools_init();
            if (!two_lanes) greedy_loop(b->lane_full, greedy_jobs, st0);
            else { std::vector<int> none; greedy_loop(b->lane_full, greedy_jobs, st0); }
        }
    } else if (use_chain) {
        chain_fold(st0);
    } else if (!two_lanes) {
        greedy_loop(b->lane_full, greedy_jobs, st0);
    } else {
        std::vector<int> part[2];
        // contiguous halves of equal estimated cost (~ n^3: rounds x cells), so that the lanes do not share cache
        // lines of neighbouring jobs' pools
        double total = 0, acc = 0;
        auto cost = [&](int j) { const double n = b->seq_off[b->job_seq[j] + 1] - b->seq_off[b->job_seq[j]]; return n * n * n + 1.0; };
        for (int j : greedy_jobs) total += cost(j);
        for (int j : greedy_jobs) { part[acc * 2 < total ? 0 : 1].push_back(j); acc += cost(j); }
        const int64_t avail = b->cand_records - b->cand_reserved;
        for (int k = 0; k < 2; k++) {
            SqLane &H = b->lane_half[k];
            H.cand0 = k ? avail / 2 : 0;
            H.cand_records = k ? avail - avail / 2 : avail / 2;
        }
        // the second lane has its own stream (its half-size kernels run beside the first lane's), ordered behind
        // everything the batch stream holds so far (bit matrix, uploads)
        static const bool lane_own_stream = !getenv("SQ_LANE_SAME_STREAM");
        if (lane_own_stream) {
            if (!b->lane_stream) {
                HIPCK(sq_stream_get(b->device, &b->lane_stream));
                HIPCK(sq_event_get(b->device, &b->lane_ev));
            }
            HIPCK(hipEventRecord(b->lane_ev, b->stream));
            HIPCK(hipStreamWaitEvent(b->lane_stream, b->lane_ev, 0));
            b->lane_half[1].stream = b->lane_stream;
        } else b->lane_half[1].stream = nullptr;
        std::thread other([&] { if (b->device >= 0) hipSetDevice(b->device); greedy_loop(b->lane_half[1], part[1], st1); });
        greedy_loop(b->lane_half[0], part[0], st0);
        other.join();
        if (!st0.rc && st1.rc) { st0.rc = st1.rc; st0.err = st1.err; }
    }
    tq.close();
    if (st0.rc) { sq_set_error(st0.err); return st0.rc; }
    const double tround = st0.tround + st1.tround;
    const int nrounds = st0.nrounds + st1.nrounds;
    const double tloop = now_s() - tfold0;
    const double ttail0 = now_s();
    // (the known structures go to the device now: the tail's launches then follow the wait for the matching kernels directly)
    b->tail_refs_state = 0;
    if (dev_tail) (void)sq_tail_refs(b, ref_off, ref_pairs, has_ref);
    // E / H / N stemsets precede the greedy ones of their job (:1094-1100), in the order E, H, N.  Hungarian and
    // Nussinov are final first; Edmonds is streamed job by job, and a sequence is ranked (its tail) the moment its
    // last Edmonds graph is matched -- the other sequences do not wait for the largest graph of the batch.
    {
        const double t0 = now_s();
        std::vector<std::atomic<int>> e_left(b->nseq);
        for (int s = 0; s < b->nseq; s++) e_left[s] = 0;
        for (int j = 0; j < b->njobs; j++) if (algos[j] & SQ_ALGO_E) e_left[b->job_seq[j]]++;
        auto take_sets = [&](std::vector<JobSets> &sets, bool edmonds) {
            for (auto it = sets.rbegin(); it != sets.rend(); ++it) {
                if ((it->algo == SQ_ALGO_E) != edmonds || it->streamed) continue;
                for (size_t k = 0; k < it->jobs.size(); k++) {
                    JobPool &P = pools[it->jobs[k]];
                    P.fin.insert(P.fin.begin(), std::move(it->sets[k]));
                    P.evals++;
                }
            }
        };
        SqAlgoEndHooks hooks;
        hooks.after_short = [&](std::vector<JobSets> &sets) { take_sets(sets, false); };
        hooks.on_e_job = [&](int j, std::vector<HStem> &set) {       // pool worker: job j's Edmonds stemset is final
            JobPool &P = pools[j];
            P.fin.insert(P.fin.begin(), std::move(set));
            P.evals++;
            const int s = b->job_seq[j];
            if (!dev_tail && --e_left[s] == 0) { tail_one(s); tailed[s] = 1; }
        };
        std::vector<JobSets> sets;
        { CpuScope cpu_(10); r = sq_algos_end(b, pending, o.levellimit, sets, &hooks); }
        pending = nullptr;
        if (r) return r;
        bool streamed = false;
        for (const JobSets &js : sets) streamed |= js.streamed;
        if (!streamed) take_sets(sets, false);               // (the hook did not run: no Edmonds jobs, or not staged)
        take_sets(sets, true);
        if (timing) fprintf(stderr, "[sq_fold] E/H/N: begin %.3f ms, wait+collect (+ tails of finished sequences) after the greedy loop %.3f ms\n", tbegin * 1e3, (now_s() - t0) * 1e3);
    }
    // ---- the device tail (sq_tail_dev.hip): every final structure the HOST holds -- the E / H / N stemsets, the greedy ones
    // when the host's own loop ran, the empty structure of a job with maxstemnum 0 -- joins the device log, then the
    // tail kernels rank every sequence and write the packed results; no per-sequence host code
    bool tails_done = false;
    if (dev_tail) {
        CpuScope cpu_(0);
        size_t nent = 0, nst = 0;
        for (int j = 0; j < b->njobs; j++) { nent += pools[j].fin.size(); for (const auto &f : pools[j].fin) nst += f.size(); }
        int rt = 0;
        if (nent > (size_t)b->fin_cap || nst > (size_t)b->fin_stem_cap) rt = 1;
        if (!rt && nent) {
            const size_t need = sizeof(SqPoolFin) * nent + sizeof(SqPoolStem) * nst + 8 * (size_t)b->njobs + 64;
            if (b->h_app_cap < need) {
                hipStreamSynchronize(b->stream);
                sq_pinned_put(b->h_app); b->h_app = nullptr; b->h_app_cap = 0;
                void *p = nullptr;
                if (sq_pinned_get(&p, need + need / 2)) return 2;
                b->h_app = (char *)p; b->h_app_cap = need + need / 2;
            }
            SqPoolFin *ef = (SqPoolFin *)b->h_app;
            SqPoolStem *es = (SqPoolStem *)(b->h_app + sizeof(SqPoolFin) * nent);
            long long *ev = (long long *)(b->h_app + sizeof(SqPoolFin) * nent + ((sizeof(SqPoolStem) * nst + 7) & ~(size_t)7));
            size_t qe = 0, qs = 0;
            const bool host_greedy = b->last_driver == 0 || b->last_driver == 3;
            for (int j = 0; j < b->njobs; j++) {
                const JobPool &P = pools[j];
                // (RunAlgo on the device: its stemsets are in the log already, the host lists hold greedy structures only)
                const int nalgo = dev_algos ? 0 : __builtin_popcount(algos[j] & (uint32_t)(SQ_ALGO_E | SQ_ALGO_H | SQ_ALGO_N));
                ev[j] = std::max<int64_t>(P.evals - nalgo, 0);
                for (size_t k = 0; k < P.fin.size(); k++) {           // [E][H][N] first, then the greedy structures, in list order
                    const std::vector<HStem> &f = P.fin[k];
                    ef[qe++] = SqPoolFin{j, (int)k < nalgo ? (uint32_t)k : SQ_FIN_KIND_G0, (int32_t)k, (int32_t)f.size(), (uint32_t)qs, 0u};
                    for (const HStem &t : f) es[qs++] = SqPoolStem{(int16_t)t.i, (int16_t)t.j, (int16_t)t.len, 0};
                }
            }
            if (host_greedy) HIPCK(hipMemcpyAsync(b->d_job_evals, ev, 8 * (size_t)b->njobs, hipMemcpyHostToDevice, b->stream));
            hipLaunchKernelGGL(sq_fin_append_kernel, dim3((unsigned)((nent + 255) / 256)), dim3(256), 0, b->stream, ef, es, (int)nent,
                               b->d_fin, b->d_fin_stems, b->d_fin_ctr, b->fin_cap, b->fin_stem_cap);
        }
        if (!rt) rt = sq_tail_device(b, o, ref_off, ref_pairs, has_ref);
        // the structures the device drivers left in the log as host lists (the host tail's input)
        auto collect_device_lists = [&]() -> int {
            if (dev_algos) {
                // the E / H / N stemsets the device-side RunAlgo logged: to the front of their job's list, in the order E, H, N
                uint32_t ctr[4] = {0, 0, 0, 0};
                HIPCK(hipMemcpy(ctr, b->d_fin_ctr, 16, hipMemcpyDeviceToHost));
                const uint32_t nf = std::min(ctr[0], b->fin_cap), ns2 = std::min(ctr[1], b->fin_stem_cap);
                std::vector<SqPoolFin> Fv(nf);
                std::vector<SqPoolStem> Sv(ns2);
                if (nf) HIPCK(hipMemcpy(Fv.data(), b->d_fin, sizeof(SqPoolFin) * (size_t)nf, hipMemcpyDeviceToHost));
                if (ns2) HIPCK(hipMemcpy(Sv.data(), b->d_fin_stems, sizeof(SqPoolStem) * (size_t)ns2, hipMemcpyDeviceToHost));
                for (uint32_t kind = SQ_FIN_KIND_N + 1; kind-- > 0;)      // N, then H, then E: each goes in front
                    for (uint32_t q = 0; q < nf; q++) {
                        const SqPoolFin &e = Fv[q];
                        if (e.round_kind != kind) continue;
                        std::vector<HStem> stems((size_t)e.nstems);
                        for (int t = 0; t < e.nstems; t++) { const SqPoolStem &x = Sv[e.stem_off + t]; stems[t] = HStem{x.i, x.j, x.len, 0.0, 0.0}; }
                        JobPool &P = pools[e.job];
                        P.fin.insert(P.fin.begin(), std::move(stems));
                        P.evals++;
                    }
            }
            if (b->last_driver == 1 || (b->last_paths & 16)) {
                const uint32_t nf = *b->chain.h_nfin;
                for (uint32_t q = 0; q < nf; q++) chain_finish(q);
            }
            if (b->last_driver == 2 && pool_collect) {
                // (the E / H / N stemsets are already at the front of the lists: the greedy structures go behind them)
                const int rc2 = pool_collect();
                if (rc2) return rc2;
                for (size_t sx = 0; sx < pool_jobs.size(); sx++) pools[pool_jobs[sx]].evals += b->pool_io.h_jobs[sx].evals;
            }
            return 0;
        };
        static const bool tail_check = getenv("SQ_TAIL_CHECK") != nullptr;
        if (rt == 0) {
            tails_done = true;
            b->last_paths |= 1;
            if (tail_check) {
                // debug: the host tail over the same structures must give the same packed bytes for every sequence
                r = collect_device_lists();
                if (r) return r;
                for (int s2 = 0; s2 < b->nseq; s2++) tail_one(s2);
                size_t bad = 0;
                std::vector<char> hb, db;
                for (int s2 = 0; s2 < b->nseq; s2++) {
                    b->packed_ok = false;
                    const int64_t nh = sq_result_pack_size(b, s2);
                    hb.assign((size_t)nh, 0); sq_result_pack(b, s2, hb.data(), nh);
                    b->packed_ok = true;
                    const int64_t nd = sq_result_pack_size(b, s2);
                    db.assign((size_t)nd, 0); sq_result_pack(b, s2, db.data(), nd);
                    if (nh != nd || memcmp(hb.data(), db.data(), (size_t)nh) != 0) {
                        size_t at = 0;
                        while (at < (size_t)std::min(nh, nd) && hb[at] == db[at]) at++;
                        if (bad++ < 8) fprintf(stderr, "[tail check] sequence %d (n = %d): host %lld bytes, device %lld bytes, first difference at byte %zu\n",
                                               s2, b->seq_off[s2 + 1] - b->seq_off[s2], (long long)nh, (long long)nd, at);
                    }
                }
                fprintf(stderr, "[tail check] %d sequences, %zu differ\n", b->nseq, bad);
            }
        }
        else if (rt != 1) return rt;
        else {
            // the host tail takes the batch
            if (timing) fprintf(stderr, "[sq_fold] device tail: not applicable to this batch, the host tail runs\n");
            r = collect_device_lists();
            if (r) return r;
        }
    }
    // the remaining sequences: the batch's worker pool shares the tail, longest first (deterministic output)
    if (!tails_done) {
        std::vector<int> order;
        std::vector<int64_t> cost(b->nseq, 0);
        for (int s = 0; s < b->nseq; s++) {
            if (tailed[s]) continue;
            order.push_back(s);
            for (int j : seq_jobs[s]) cost[s] += (int64_t)pools[j].fin.size() * (b->seq_off[s + 1] - b->seq_off[s]);
        }
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost[x] > cost[y]; });
        sq_pool(b)->parallel_for((int)order.size(), [&](int k) { tail_one(order[k]); });
    }
    if (timing) {
        double mx = 0, sum = 0; int arg = 0;
        for (int q = 0; q < b->nseq; q++) { sum += tail_cost[q]; if (tail_cost[q] > mx) { mx = tail_cost[q]; arg = q; } }

        fprintf(stderr, "[sq_fold] tail: sum %.3f ms, max %.3f ms (seq %d, n=%d, %zu structures kept)\n", sum * 1e3, mx * 1e3, arg,
                b->seq_off[arg + 1] - b->seq_off[arg], b->results[arg].preds.size());
    }
    if (timing)
        fprintf(stderr, "[sq_fold] rounds=%d loop=%.3fms (round driver %.3f: prep %.3f gpu+wait %.3f post %.3f; pool %.3f) tail=%.3fms\n",
                nrounds, tloop * 1e3, tround * 1e3, g_t[0] * 1e3, g_t[1] * 1e3, g_t[2] * 1e3, (tloop - tround) * 1e3,
                (now_s() - ttail0) * 1e3);
    if (g_cpuacc_on) {
        static const char *nm[12] = {"tails", "collect", "edges", "grow|stemfilter", "post", "launch|hook", "wait", "annotate", "caller", "begin", "end", "teardown"};
        fprintf(stderr, "[sq_fold cpu ms]");
        g_cpuacc[8] += CpuScope::now() - cpu_fold0;
        for (int k = 0; k < 12; k++) fprintf(stderr, " %s %.2f", nm[k], g_cpuacc[k].exchange(0) * 1e-6);
        fprintf(stderr, "\n");
    }
    if (timing && use_chain)
        fprintf(stderr, "[sq_fold] chained rounds: start %.3f ms after the E/H/N launch, wall %.3f ms, %d rounds enqueued\n",
                st0.tstart * 1e3, st0.twall * 1e3, st0.nrounds);
    if (timing && two_lanes && !use_chain)
        fprintf(stderr, "[sq_fold] lanes: 0 start %.3f wall %.3f driver %.3f (%d rounds); 1 start %.3f wall %.3f driver %.3f (%d rounds)\n",
                st0.tstart * 1e3, st0.twall * 1e3, st0.tround * 1e3, st0.nrounds, st1.tstart * 1e3, st1.twall * 1e3, st1.tround * 1e3, st1.nrounds);
    return 0;
}

extern "C" int32_t sq_fold_driver(const sq_batch *b) { return b ? b->last_driver : -1; }
extern "C" int32_t sq_fold_paths(const sq_batch *b) { return b ? b->last_paths : -1; }
extern "C" int64_t sq_fold_peak_structs(const sq_batch *b) { return b ? b->last_peak : -1; }

extern "C" int sq_fold_concurrent(sq_batch *const *batches, int32_t nbatch, const sq_fold_opts *opts,
                                  const int32_t *const *ref_off, const int32_t *const *ref_pairs, const uint8_t *const *has_ref)
{
    return sq_fold_concurrent_n(batches, nbatch, opts, ref_off, ref_pairs, has_ref, 1);
}

extern "C" int sq_fold_concurrent_n(sq_batch *const *batches, int32_t nbatch, const sq_fold_opts *opts,
                                    const int32_t *const *ref_off, const int32_t *const *ref_pairs, const uint8_t *const *has_ref,
                                    int32_t reps)
{
    if (!batches || nbatch <= 0 || !opts || reps < 1) { sq_set_error("bad argument"); return -1; }
    for (int k = 0; k < nbatch; k++) if (!batches[k]) { sq_set_error("bad argument"); return -1; }
    std::vector<int> rc(nbatch, 0);
    std::vector<std::string> msg(nbatch);
    // every stream less keeps the long kernels of one batch out of another batch's hardware queue (GPU_MAX_HW_QUEUES)
    for (int k = 0; k < nbatch; k++) if (batches[k]) {
        batches[k]->side_streams = nbatch >= 3 ? 2 : 3;
        batches[k]->inflight = nbatch;
    }
    auto work = [&](int k) {
        if (k > 0 && batches[k]->device >= 0) hipSetDevice(batches[k]->device);
        for (int r = 0; r < reps && !rc[k]; r++)
            rc[k] = sq_fold(batches[k], opts, ref_off ? ref_off[k] : nullptr, ref_pairs ? ref_pairs[k] : nullptr,
                            has_ref ? has_ref[k] : nullptr);
        if (rc[k]) msg[k] = sq_last_error();                 // (the error text is per thread)
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nbatch; k++) th.emplace_back(work, k);
    work(0);
    for (auto &t : th) t.join();
    // (a later fold of one of these batches alone is a fold with one batch in flight)
    for (int k = 0; k < nbatch; k++) if (batches[k]) { batches[k]->inflight = 1; batches[k]->side_streams = 3; }
    for (int k = 0; k < nbatch; k++) if (rc[k]) { sq_set_error(msg[k]); return rc[k]; }
    return 0;
}

// ---- result getters ------------------------------------------------------------------------------
// Two homes of a fold's results: the packed records the device tail wrote into pinned memory (b->packed_ok; the C ABI's
// own layout, so the bulk getters are copies) or the SeqResult objects of the host tail.
// structures of a sequence the getters show: all of them, or the first result_limit in rank order (sq_result_limit)
static inline int64_t shown(const sq_batch *b, const SeqResult &R)
{
    const int64_t ns = (int64_t)R.preds.size();
    return b->result_limit > 0 ? std::min<int64_t>(ns, b->result_limit) : ns;
}
namespace {
struct PackedRec {                       // view of one packed record (sq_result_pack layout)
    const char *p; int64_t ns, n, has_ref, evals;
    const double *met() const { return (const double *)(p + 32); }
    const double *scores() const { return (const double *)(p + 160); }
    const uint64_t *masks() const { return (const uint64_t *)(p + 160 + 24 * ns); }
    const int16_t *levels(int64_t row) const { return (const int16_t *)(p + 160 + 32 * ns) + row * n; }
};
inline PackedRec packed_rec(const sq_batch *b, int seq)
{
    PackedRec R;
    R.p = b->h_rec + b->h_rec_off[seq];
    const int64_t *h = (const int64_t *)R.p;
    R.ns = h[0]; R.n = h[1]; R.has_ref = h[2]; R.evals = h[3];
    return R;
}
inline int64_t packed_shown(const sq_batch *b, const PackedRec &R) { return b->result_limit > 0 ? std::min<int64_t>(R.ns, b->result_limit) : R.ns; }
// every record shows what the fold packed (no lower limit set since): the bulk getters are plain copies
inline bool packed_whole(const sq_batch *b) { return b->result_limit == b->packed_limit || b->result_limit == 0 || (b->packed_limit > 0 && b->result_limit >= b->packed_limit); }
}  // namespace
extern "C" int sq_result_limit(sq_batch *b, int32_t k)
{
    if (!b || k < 0) { sq_set_error("bad argument"); return -1; }
    b->result_limit = k;
    return 0;
}
extern "C" int32_t sq_result_nstruct(const sq_batch *b, int32_t seq)
{
    if (!b || seq < 0 || seq >= b->nseq) return -1;
    if (b->packed_ok) return (int32_t)packed_shown(b, packed_rec(b, seq));
    return (int32_t)shown(b, b->results[seq]);
}
extern "C" int sq_result_consensus(const sq_batch *b, int32_t seq, int16_t *levels)
{
    if (!b || seq < 0 || seq >= b->nseq) return -1;
    if (b->packed_ok) { const PackedRec R = packed_rec(b, seq); memcpy(levels, R.levels(0), 2 * (size_t)R.n); return 0; }
    const auto &c = b->results[seq].cons;
    memcpy(levels, c.data(), c.size() * sizeof(int16_t));
    return 0;
}
extern "C" int sq_result_struct(const sq_batch *b, int32_t seq, int32_t k, int16_t *levels, double scores[3],
                                uint64_t *pset_mask)
{
    if (!b || seq < 0 || seq >= b->nseq) return -1;
    if (b->packed_ok) {
        const PackedRec R = packed_rec(b, seq);
        if (k < 0 || k >= (int)packed_shown(b, R)) return -1;
        memcpy(levels, R.levels(k + 1), 2 * (size_t)R.n);
        for (int t = 0; t < 3; t++) scores[t] = R.scores()[3 * k + t];
        *pset_mask = R.masks()[k];
        return 0;
    }
    const auto &R = b->results[seq];
    if (k < 0 || k >= (int)shown(b, R)) return -1;
    memcpy(levels, R.preds[k].levels.data(), R.preds[k].levels.size() * sizeof(int16_t));
    for (int t = 0; t < 3; t++) scores[t] = R.preds[k].scores[t];
    *pset_mask = R.preds[k].pset_mask;
    return 0;
}
extern "C" int sq_result_metrics(const sq_batch *b, int32_t seq, double cons[6], double best[7])
{
    if (!b || seq < 0 || seq >= b->nseq) return -1;
    if (b->packed_ok) {
        const PackedRec R = packed_rec(b, seq);
        if (!R.has_ref) return 1;
        for (int t = 0; t < 6; t++) cons[t] = R.met()[t];
        for (int t = 0; t < 7; t++) best[t] = R.met()[6 + t];
        return 0;
    }
    const auto &R = b->results[seq];
    if (!R.has_ref) return 1;
    for (int t = 0; t < 6; t++) cons[t] = R.cons_metrics[t];
    for (int t = 0; t < 7; t++) best[t] = R.best_metrics[t];
    return 0;
}
extern "C" int64_t sq_result_evals(const sq_batch *b, int32_t seq)
{
    if (!b || seq < 0 || seq >= b->nseq) return -1;
    if (b->packed_ok) return packed_rec(b, seq).evals;
    return b->results[seq].evals;
}

extern "C" int64_t sq_result_pack_size(const sq_batch *b, int32_t seq)
{
    if (!b || seq < 0 || seq >= b->nseq) return -1;
    if (b->packed_ok) {
        const PackedRec R = packed_rec(b, seq);
        const int64_t ns = packed_shown(b, R);
        return 8 * 4 + 8 * 16 + 8 * 3 * ns + 8 * ns + 2 * (1 + ns) * R.n;
    }
    const auto &R = b->results[seq];
    const int64_t ns = shown(b, R), n = (int64_t)R.cons.size();
    return 8 * 4 + 8 * 16 + 8 * 3 * ns + 8 * ns + 2 * (1 + ns) * n;
}
extern "C" int sq_result_pack(const sq_batch *b, int32_t seq, void *buf, int64_t cap)
{
    const int64_t need = sq_result_pack_size(b, seq);
    if (need < 0 || cap < need) { sq_set_error("result buffer too small"); return -1; }
    if (b->packed_ok) {
        const PackedRec R = packed_rec(b, seq);
        const int64_t ns = packed_shown(b, R);
        if (ns == R.ns) { memcpy(buf, R.p, (size_t)need); return 0; }
        char *p = (char *)buf;                                // a lower limit than the fold packed: the first ns structures
        int64_t hdr[4] = {ns, R.n, R.has_ref, R.evals};
        memcpy(p, hdr, 32); p += 32;
        memcpy(p, R.met(), 128); p += 128;
        memcpy(p, R.scores(), 24 * (size_t)ns); p += 24 * ns;
        memcpy(p, R.masks(), 8 * (size_t)ns); p += 8 * ns;
        memcpy(p, R.levels(0), 2 * (size_t)((1 + ns) * R.n));
        return 0;
    }
    const auto &R = b->results[seq];
    const int64_t ns = shown(b, R), n = (int64_t)R.cons.size();
    char *p = (char *)buf;
    int64_t hdr[4] = {ns, n, R.has_ref ? 1 : 0, R.evals};
    memcpy(p, hdr, 32); p += 32;
    double met[16];
    for (int t = 0; t < 6; t++) met[t] = R.has_ref ? R.cons_metrics[t] : NAN;
    for (int t = 0; t < 7; t++) met[6 + t] = R.has_ref ? R.best_metrics[t] : NAN;
    for (int t = 0; t < 3; t++) met[13 + t] = R.has_ref ? R.ref_scores[t] : NAN;
    memcpy(p, met, 128); p += 128;
    for (int64_t k = 0; k < ns; k++) { memcpy(p, R.preds[k].scores, 24); p += 24; }
    for (int64_t k = 0; k < ns; k++) { memcpy(p, &R.preds[k].pset_mask, 8); p += 8; }
    memcpy(p, R.cons.data(), 2 * n); p += 2 * n;
    for (int64_t k = 0; k < ns; k++) { memcpy(p, R.preds[k].levels.data(), 2 * n); p += 2 * n; }
    return 0;
}

// all sequences of the batch in one call: record s occupies [off[s], off[s+1]) of buf (sq_result_pack layout)
extern "C" int64_t sq_result_pack_all_size(const sq_batch *b)
{
    if (!b) return -1;
    if (b->packed_ok && packed_whole(b)) return b->h_rec_off[b->nseq];
    int64_t tot = 0;
    for (int s = 0; s < b->nseq; s++) tot += (sq_result_pack_size(b, s) + 7) & ~(int64_t)7;
    return tot;
}
extern "C" int sq_result_pack_all(const sq_batch *b, void *buf, int64_t cap, int64_t *off)
{
    if (!b || !buf || !off) { sq_set_error("bad argument"); return -1; }
    if (b->packed_ok && packed_whole(b)) {                    // the device tail's records, as they lie in pinned memory
        const int64_t tot = b->h_rec_off[b->nseq];
        if (tot > cap) { sq_set_error("result buffer too small"); return -1; }
        memcpy(off, b->h_rec_off, 8 * ((size_t)b->nseq + 1));
        memcpy(buf, b->h_rec, (size_t)tot);
        return 0;
    }
    int64_t o = 0;
    for (int s = 0; s < b->nseq; s++) {
        const int64_t need = sq_result_pack_size(b, s);
        off[s] = o;
        if (o + need > cap) { sq_set_error("result buffer too small"); return -1; }
        o += (need + 7) & ~(int64_t)7;
    }
    off[b->nseq] = o;
    // the records are independent: big batches share the copying among the worker pool (15 MB for 10,000 x 300 nt)
    std::atomic<int> rc{0};
    auto one = [&](int s) {
        const int r = sq_result_pack(b, s, (char *)buf + off[s], cap - off[s]);
        if (r) { rc = r; return; }
        const int64_t need = sq_result_pack_size(b, s);       // the pad up to the next record: zeros, as in the device tail's records
        memset((char *)buf + off[s] + need, 0, (size_t)(off[s + 1] - off[s] - need));
    };
    if (b->nseq >= 512 && o >= ((int64_t)1 << 20)) sq_pool(const_cast<sq_batch *>(b))->parallel_for(b->nseq, one);
    else for (int s = 0; s < b->nseq; s++) one(s);
    return rc.load();
}

// Dot-bracket rows of every record as ASCII text (the bulk form of levels -> characters): record s occupies
// [off[s], off[s+1]) with its consensus row and then its nstruct structure rows, n characters each (gap-free
// coordinates, no separators re-inserted: the caller does that for the records that have any).  Levels 1..30 print as
// ( [ { < A..Z / ) ] } > a..z (SQRNdbnseq.py:107-112); deep[s] = 1 when the record uses a level beyond them (the
// reference continues with Cyrillic letters): such records are left to the generic per-record path.
extern "C" int64_t sq_result_dbn_all_size(const sq_batch *b)
{
    if (!b) return -1;
    if (b->packed_ok) {
        if (packed_whole(b)) return b->h_txt_off[b->nseq];
        int64_t tot = 0;
        for (int s = 0; s < b->nseq; s++) { const PackedRec R = packed_rec(b, s); tot += (packed_shown(b, R) + 1) * R.n; }
        return tot;
    }
    int64_t tot = 0;
    for (int s = 0; s < b->nseq; s++) tot += (shown(b, b->results[s]) + 1) * (int64_t)b->results[s].cons.size();
    return tot;
}
extern "C" int sq_result_dbn_all(const sq_batch *b, char *buf, int64_t cap, int64_t *off, uint8_t *deep)
{
    if (!b || !buf || !off || !deep) { sq_set_error("bad argument"); return -1; }
    if (b->packed_ok) {                                       // the ASCII rows the pack kernel wrote
        memcpy(deep, b->h_deep, (size_t)b->nseq);
        if (packed_whole(b)) {
            const int64_t tot = b->h_txt_off[b->nseq];
            if (tot > cap) { sq_set_error("text buffer too small"); return -1; }
            memcpy(off, b->h_txt_off, 8 * ((size_t)b->nseq + 1));
            memcpy(buf, b->h_txt, (size_t)tot);
            return 0;
        }
        int64_t o2 = 0;
        for (int s = 0; s < b->nseq; s++) {
            const PackedRec R = packed_rec(b, s);
            const int64_t bytes = (packed_shown(b, R) + 1) * R.n;
            off[s] = o2;
            if (o2 + bytes > cap) { sq_set_error("text buffer too small"); return -1; }
            memcpy(buf + o2, b->h_txt + b->h_txt_off[s], (size_t)bytes);
            o2 += bytes;
        }
        off[b->nseq] = o2;
        return 0;
    }
    static const char open_ch[31] = {'.', '(', '[', '{', '<', 'A', 'B', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'J', 'K', 'L', 'M', 'N', 'O',
                                     'P', 'Q', 'R', 'S', 'T', 'U', 'V', 'W', 'X', 'Y', 'Z'};
    static const char close_ch[31] = {'.', ')', ']', '}', '>', 'a', 'b', 'c', 'd', 'e', 'f', 'g', 'h', 'i', 'j', 'k', 'l', 'm', 'n', 'o',
                                      'p', 'q', 'r', 's', 't', 'u', 'v', 'w', 'x', 'y', 'z'};
    int64_t o = 0;
    for (int s = 0; s < b->nseq; s++) {
        const SeqResult &R = b->results[s];
        const int64_t n = (int64_t)R.cons.size();
        off[s] = o;
        const int64_t ns = shown(b, R);
        if (o + (ns + 1) * n > cap) { sq_set_error("text buffer too small"); return -1; }
        bool dp = false;
        auto row = [&](const std::vector<int16_t> &lv) {
            for (int64_t i = 0; i < n; i++) {
                const int v = lv[i];
                char ch = '.';
                if (v > 0) { if (v <= 30) ch = open_ch[v]; else dp = true; }
                else if (v < 0) { if (v >= -30) ch = close_ch[-v]; else dp = true; }
                buf[o + i] = ch;
            }
            o += n;
        };
        row(R.cons);
        for (int64_t k = 0; k < ns; k++) row(R.preds[k].levels);
        deep[s] = dp ? 1 : 0;
    }
    off[b->nseq] = o;
    return 0;
}
