// sq_host.hip -- host driver of libsquarna_hip.so: batch set-up, kernel launches, the
// round driver and the greedy pool loop (SQRNdbnseq.py:1102-1199).  Device memory is
// the caller's workspace; the host only orchestrates (one small H2D + D2H per round).
#include "sq_host_int.h"

static thread_local std::string g_err;
std::atomic<long long> g_cpuacc[12];
bool g_cpuacc_on = getenv("SQ_CPUACC") != nullptr;

// host phase timers (printed to stderr when SQ_TIMING is set)
thread_local double g_t[8];
static thread_local int g_cap_kind = 0;
void sq_set_error(const std::string &msg) { g_err = msg; g_cap_kind = 0; }
void sq_set_capacity_error(int kind, const std::string &msg) { g_err = msg; g_cap_kind = kind; }
extern "C" int sq_last_capacity(void) { return g_cap_kind; }
int sq_check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return 0;
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e;
}

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

int sq_wait_word(const sq_batch *b, volatile uint32_t *flag, uint32_t want, hipStream_t st, const char *what, bool at_least)
{
    const bool relaxed = sq_relaxed_waits(b);
    const uint64_t poll_mask = relaxed ? 0x3FFF : 0xFFFFF;
    uint64_t spins = 0;
    auto there = [&]() { const uint32_t v = *flag; return at_least ? (int32_t)(v - want) >= 0 : v == want; };
    while (!there()) {
        if ((++spins & poll_mask) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipErrorNotReady) {
                if (q != hipSuccess) return sq_check(q, what);
                if (!there()) {
                    hipStreamSynchronize(st);
                    if (!there()) { sq_set_error(std::string(what) + " did not signal completion"); return 2; }
                }
            }
        }
        sq_wait_step(spins, relaxed);
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
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
        b->pool = sq_pool_get(nthr, b->device);
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
SqPool *sq_pool_get(int nthr, int device)
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
void sq_pool_put(SqPool *p)
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
    // size classes, four per octave above 64 KB (at most a quarter more than asked for): the windows of a stream of requests
    // differ by a few per cent from step to step -- with exact sizes the cache kept meeting sizes it did not hold yet and
    // traded old buffers for new ones through the driver for nine steps (hipHostFree + hipHostMalloc: 150-380 ms a step)
    size_t want = (std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095;
    if (want > 65536) {
        int k = 63 - __builtin_clzll((unsigned long long)want);
        const size_t step = (size_t)1 << (k - 2);
        want = (want + step - 1) & ~(step - 1);
    }
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
    static const bool trace = getenv("SQ_PINNED_TRACE") != nullptr;      // (one line per trip to the driver)
    const double tm0 = trace ? now_s() : 0;
    const int r = sq_check(hipHostMalloc(p, want, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable), "hipHostMalloc");
    if (trace) fprintf(stderr, "[sq_pinned] hipHostMalloc %zu bytes: %.2f ms (idle %zu buffers, %zu MB)\n", want, (now_s() - tm0) * 1e3, g_pinned.idle.size(), g_pinned.idle_bytes >> 20);
    if (r) { *p = nullptr; return r; }
    std::lock_guard<std::mutex> lk(g_pinned.mu);
    g_pinned.live[*p] = want;
    return 0;
}
void sq_pinned_put(void *p)
{
    // idle buffers kept: two steps' worth of a server that builds the next batches while it folds (16 batches of 12 SRtest150
    // sets pin ~3 GB between them); SQ_PINNED_CACHE_MB overrides
    static const size_t kIdleBytes = (size_t)(getenv("SQ_PINNED_CACHE_MB") ? std::max(0, atoi(getenv("SQ_PINNED_CACHE_MB"))) : 6144) << 20;
    static const size_t kIdleCount = 4096;
    if (!p) return;
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        auto it = g_pinned.live.find(p);
        const size_t cap = it == g_pinned.live.end() ? 0 : it->second;
        if (it != g_pinned.live.end()) g_pinned.live.erase(it);
        // (a batch holds ~25 pinned buffers; eight batches of a server's step are created and destroyed together: with room
        // for 64 idle buffers two thirds of them went back to the driver -- hipHostFree + hipHostMalloc: 6.5 ms per batch)
        // The newest buffers stay: when the cache is full the OLDEST idle ones go back to the driver (a process that folded
        // other shapes before -- the legs of a bench, a server whose inputs change -- otherwise fills the cache with sizes nobody
        // asks for again and then pays hipHostMalloc + hipHostFree for every buffer of every batch: 245 against 55 ms per step)
        if (cap && cap <= kIdleBytes) {
            while (!g_pinned.idle.empty() && (g_pinned.idle.size() >= kIdleCount || g_pinned.idle_bytes + cap > kIdleBytes)) {
                drop.push_back(g_pinned.idle.front().second);
                g_pinned.idle_bytes -= g_pinned.idle.front().first;
                g_pinned.idle.erase(g_pinned.idle.begin());
            }
            g_pinned.idle.emplace_back(cap, p);
            g_pinned.idle_bytes += cap;
            p = nullptr;
        }
    }
    static const bool trace = getenv("SQ_PINNED_TRACE") != nullptr;
    if (trace && (!drop.empty() || p)) fprintf(stderr, "[sq_pinned] hipHostFree x %zu\n", drop.size() + (p ? 1 : 0));
    for (void *q : drop) hipHostFree(q);
    if (p) hipHostFree(p);
}

extern "C" long long sq_host_cache_trim(void)
{
    std::vector<std::pair<size_t, void *>> drop;
    {
        std::lock_guard<std::mutex> lk(g_pinned.mu);
        drop.swap(g_pinned.idle);
        g_pinned.idle_bytes = 0;
    }
    long long bytes = 0;
    for (auto &e : drop) { bytes += (long long)e.first; hipHostFree(e.second); }
    return bytes;
}

extern "C" int sq_version(void) { return 100; }
extern "C" const char *sq_last_error(void) { return g_err.c_str(); }
// ---- profiling (HIP events on the batch stream) ----------------------------------------
namespace {
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
    if (k < 0 || k > 8) return -1;
    hipStreamSynchronize(b->stream);
    for (int q = 0; q < 4; q++) if (b->side[q]) hipStreamSynchronize(b->side[q]);
    prof_collect(b);
    *ms = b->prof[k].ms; *launches = b->prof[k].launches; *bytes = b->prof[k].bytes;
    return 0;
}

