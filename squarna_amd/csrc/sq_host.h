// sq_host.h -- host-side batch object (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <functional>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <time.h>
#include "../../include/squarna_hip.h"
#include "sq_device.h"
#include "sq_internal.h"
#include "sq_tail_dev.h"

struct HStem {            // host stem record: bps (i+k, j-k), k < len
    int32_t i, j, len;
    double bps, fin;
};

struct HStruct {          // partial structure of the greedy pool
    int32_t job = 0;
    double subopt = 1.0;
    std::vector<HStem> stems;
    std::vector<SqStrand> strands;   // both halves of every stem, sorted by start, with levels
    bool anycross = false;           // some pair of stems crosses (pseudoknot): levels need the full rule
};

// view of a structure handed to the round driver (no copies per round)
struct SView {
    int32_t job;
    double subopt;
    const HStruct *st;
};

// (re)build strands + levels of a structure from its stems
void sq_build_strands(HStruct &s);
// child = parent + one stem, strands/levels maintained incrementally
void sq_extend_struct(const HStruct &parent, const HStem &stem, HStruct &child, bool take_parent = false);

struct SeqResult {        // SQRNdbnseq return tuple of one sequence (SQRNdbnseq.py:1285-1286)
    struct Pred {
        std::vector<int16_t> levels;   // signed bracket level per position
        double scores[3];
        uint64_t pset_mask;
    };
    std::vector<int16_t> cons;
    std::vector<Pred> preds;
    double cons_metrics[6], best_metrics[7];
    double ref_scores[3];              // ScoreStruct of the known structure (ReferenceScores, SQRNdbnseq.py:958-970), when given
    bool has_ref = false;
    int64_t evals = 0;
};

struct ProfSlot {
    double ms = 0;
    int64_t launches = 0;
    double bytes = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<hipEvent_t> pool;
};

// Small persistent worker pool (per batch): host phases that are independent per sequence / per job
// (the a-10 tail, the RunAlgo stem filters) are shared among a few threads that stay alive between
// folds, so their allocator arenas stay warm.  parallel_for hands out indices dynamically; the caller
// takes part.  Results never depend on the schedule (every index writes its own slot).
class SqPool {
public:
    explicit SqPool(int nthreads, int device = -1);
    ~SqPool();
    // callers are serialised (two fold lanes share the pool).  The workers form two groups: the first 15 serve every
    // call, the rest only wide ones (wide < 0: n >= 512) -- waking 31 threads for a few hundred short items costs
    // more than it buys.
    void parallel_for(int n, const std::function<void(int)> &fn, int wide = -1);
    int size() const { return (int)workers.size() + 1; }
    const int nthreads, device;          // (idle pools are kept for the next batch that asks for the same: sq_host.hip)
private:
    void worker(int group);
    std::vector<std::thread> workers;
    int group_size[2] = {0, 0};
    std::mutex mu, callers;
    std::condition_variable cv_start[2], cv_done;
    const std::function<void(int)> *fn = nullptr;
    std::atomic<int> next{0};
    int total = 0, active = 0;
    uint64_t gen[2] = {0, 0};
    bool stop = false;
};

// The round buffers one greedy loop works with.  A batch has one lane that spans all of them; sq_fold may split them
// into two lanes (two host threads, each driving the rounds of half of the jobs), so that the bookkeeping of one
// lane overlaps the kernels of the other.
struct SqLane {
    SqStruct *h_structs = nullptr; SqStrand *h_strands = nullptr; SqOut *h_out = nullptr;   // pinned
    SqCounters *h_ctr = nullptr; uint32_t *h_seq = nullptr;
    SqStruct *d_structs = nullptr; SqStrand *d_strands = nullptr; SqOut *d_out = nullptr;   // device
    SqCounters *d_ctr = nullptr;
    uint32_t h_out_cap = 0, out_cap = 0;
    int32_t slot0 = 0, max_structs = 0, strand_cap = 0;
    int64_t cand0 = 0, cand_records = 0;      // records [cand0, cand0 + cand_records) of the candidate arena
    uint32_t *round_seq = nullptr;            // id of the last round sent to h_seq (lanes that share the word share the counter)
    hipStream_t stream = nullptr;             // nullptr: the batch stream
    std::vector<SqOut> big_out;
    bool out_ovf_seen = false;                // the last chunk emitted more stems than out_cap (run_round_impl splits it)
    std::vector<uint32_t> post_cnt, post_idx, post_fill;   // scratch of the round's output bucketing
};

// Behaviour switches of a fold, read from the environment ONCE at the start of every sq_fold (and at sq_batch_create, for the
// per-call ops): diagnostics and test hooks, none changes results; tests flip them between two folds of one process.
// INTEGRATION.md section 5 documents each.  (Switches that size or shape a batch are read at sq_batch_create; tuning knobs of
// the launch shapes are `static const`, read once per process where they are used.)
struct SqFoldSwitches {
    bool timing = false;              // SQ_TIMING: phase timings on stderr
    bool pool_debug = false;          // SQ_POOL_DEBUG: pool sizes per round (with SQ_TIMING)
    bool no_chain = false;            // SQ_NO_CHAIN: poollim = 1 folds driven round by round from the host
    bool no_rounds = false;           // SQ_NO_ROUNDS: the launched rounds instead of the persistent round kernel
    bool no_pool = false;             // SQ_NO_POOL: pools booked on the host
    bool no_opt_chain = false;        // SQ_NO_OPT_CHAIN: pools with a range factor of 1.0 go to the device pools at once (no optimistic chains)
    bool no_fly_bits = false;         // SQ_NO_FLY_BITS: the bit matrices are always written (the round kernel's scan reads them)
    bool no_defer_wait = false;       // SQ_NO_DEFER_WAIT: the host waits for the round kernel before it enqueues the device tail
    bool no_pool_round = false;       // SQ_NO_POOL_ROUND: state / scan / score / choose / extend kernels instead of sq_pool_round_kernel
    bool pool_round_always = false;   // SQ_POOL_ROUND_ALWAYS: (the default since late round 4; the switch is read and ignored)
    int pool_round_nsurv = 0;         // SQ_POOL_ROUND_NSURV: survivors sq_pool_round_kernel keeps in LDS (0: by length)
    int pool_slots = 0;               // SQ_POOL_SLOTS: structure slots the device pools may use (0: max_structs)
    int pool_root = 0;                // SQ_POOL_ROOT: pools on sequences of 257-1,024 nt run the one-wave round kernel over root lists
    bool no_pool_kept = false;        // SQ_NO_POOL_KEPT: ... not over the lists their parents left (sq_device.h: SqKept)
    int pool_ahead = 3;               // SQ_POOL_AHEAD: rounds of the device pools a batch alone enqueues ahead of the host (0: none)
    int pool_chunk = 0;               // SQ_POOL_CHUNK: structures per chunk of a generation (0: what the arena holds)
    bool no_score_bound = false;      // SQ_NO_SCORE_BOUND: ScoreStems on every survivor of :492
    bool no_score_context = false;    // SQ_NO_SCORE_CONTEXT: the strand walk instead of the context tables (launched rounds)
    bool no_device_algos = false;     // SQ_NO_DEVICE_ALGOS: RunAlgo's edge lists and filters on the host
    bool no_edges_lds = false;        // SQ_NO_EDGES_LDS: the edges kernel ranks its stems in global memory (the form for lists beyond LDS)
    bool no_device_tail = false;      // SQ_NO_DEVICE_TAIL: the ranking tail on the host
    bool algo_sync = false;           // SQ_ALGO_SYNC: matching kernels on the batch stream
    int lsap_classes = 0;             // SQ_LSAP_CLASSES: size classes of the Hungarian / Nussinov launches (0: 3 crowded, else 1)
    bool mwm_dump = false;            // SQ_MWM_DUMP: the blossom graphs' sizes and LDS plan on stderr
    bool mwm_posthoc = false;         // SQ_MWM_POSTHOC: verification of streamed Edmonds results
};
void sq_read_fold_switches(SqFoldSwitches &sw);

struct sq_batch {
    SqFoldSwitches sw;                        // (see above: refreshed by every sq_fold)
    hipStream_t stream = nullptr;
    int device = -1;                          // device current at sq_batch_create (adopted by every spawned thread)
    // host copies
    int32_t nseq = 0, npset = 0, njobs = 0, maxn = 0;
    int64_t ltot = 0;
    bool reacts_null = false;             // created without reactivities (0.5 everywhere): `reacts` is formed on first host use
    mutable std::once_flag reacts_once;
    std::vector<int32_t> seq_off, rbp_off, rbps, job_seq, job_pset;
    std::vector<uint8_t> codes, flags;
    std::vector<char> seq_has_sep;            // per sequence: it holds a chain separator (the kernels that keep prefix counts of separators skip them otherwise)
    std::vector<double> reacts;
    std::vector<double> rftab;                // host-libm reactfactor tables, 256 doubles each (SqJob::rf_idx)
    std::vector<uint8_t> ridx;                // per position: reactivity level index (SqDevCtx::ridx)
    std::vector<sq_paramset> psets;
    std::vector<SqPsetDev> psets_dev;         // host copy of the device paramset records (pow_len: has a stemscore ** 1.7 table)
    std::vector<char> pset_dyadic;            // all pair weights are multiples of 2^-10 below 1024 (exact sums in any order)
    std::vector<int> pset_classes;            // letter classes of the scoring kernel's cell table: pairing letters + 1
    std::vector<SqJob> jobs;
    int32_t interchainonly = 0;
    int32_t nletters = 0;                     // distinct letter codes of the batch (sq_bits_masks_kernel)
    int32_t max_structs = 4096;
    int32_t cand_per_nt = 32;
    // device carve
    SqDevCtx ctx{};
    SqState state{};
    SqScanArgs scan{};
    SqStruct *d_structs = nullptr;
    SqStrand *d_strands = nullptr;
    SqOut *d_out = nullptr;
    int64_t cand_records = 0;
    int64_t cand_reserved = 0;            // records at the END of the arena lent to matching kernels in flight
    char *stage_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // pinned job tables + edge lists of the matching kernels
    size_t stage_cap[4] = {0, 0, 0, 0};
    uint32_t algo_seq = 0;                // completion stamps of the matching launches
    hipStream_t lane_stream = nullptr;        // second lane of sq_fold's greedy rounds (created on first use)
    hipEvent_t lane_ev = nullptr;
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};   // side streams of the E / H / N kernels (sq_fold); [3]: the blossom kernel's smaller size classes of a batch folded alone
    int32_t cell_entries = 32;            // doubles of the scoring kernels' cell table (dynamic LDS)
    bool score_bound = true, score_ctx = true;   // the scoring kernel's branch and bound / closed-form sweep (per fold: SQ_NO_SCORE_BOUND, SQ_NO_SCORE_CONTEXT)
    SqCtxTab ctxtab = SqCtxTab{};         // ScoreStems context tables (sq_context.h), rec == nullptr: none
    int32_t chain_tmax = 1;               // most stems a structure of any job can hold (sizes the extend kernels' LDS)
    char *algo_scratch = nullptr;         // device scratch of the Hungarian / Nussinov kernels (Layout::off_algo)
    size_t algo_bytes = 0, algo_used = 0;
    int last_driver = 0;                  // sq_fold_driver
    int last_paths = 0;                   // sq_fold_paths
    int64_t last_peak = 0;                // sq_fold_peak_structs
    int32_t result_limit = 0;             // sq_result_limit (0: the getters show every structure)
    int inflight = 1;                     // batches folded at the same time (sq_fold_concurrent): sizes the pool, relaxes the wait loops
    int side_streams = 3;                 // side streams of E / H / N: 3, or 2 (H and N share one) with many batches in flight
    hipEvent_t class_ev = nullptr;        // joins the blossom kernel's smaller size classes (on side[3], or on the short kernels' stream) into side[0]
    hipEvent_t edges_ev = nullptr;        // device-side RunAlgo: the edge lists are written (batch stream -> side streams)
    uint32_t out_cap = 0;
    int32_t strand_cap = 0;
    size_t mat32_bytes = 0;
    bool has_fp32 = true;                 // fp32 score matrices are part of the workspace
    bool filled = false;                  // fp32 matrices (and bit matrices) of every job are valid
    bool mul_applied = false;             // multiplier matrices already folded into the dense arena
    bool bits_ready = false;              // bit matrices valid (all the fold path needs)
    // pinned staging
    SqStruct *h_structs = nullptr;
    SqStrand *h_strands = nullptr;
    SqCounters *h_ctr = nullptr;
    uint32_t *h_seq = nullptr;            // pinned: id of the last finished round (written by sq_done_kernel)
    uint32_t round_seq = 0, round_seq2 = 0;
    SqOut *h_out = nullptr;
    uint32_t h_out_cap = 0;
    std::vector<SqOut> big_out;
    // results
    std::vector<SeqResult> results;
    SqPool *pool = nullptr;               // lazily created host workers
    SqLane lane_full, lane_half[2];       // see SqLane
    SqCounters *h_ctr2 = nullptr; uint32_t *h_seq2 = nullptr;   // pinned counters / sequence word of the second lane
    // device-chained rounds (width-1 pools): device arrays carved from the workspace, pinned ones created on first use
    SqChainIO chain{};
    SqChain *h_chain = nullptr;           // pinned staging of the per-structure records
    int64_t chain_T = 0;                  // summed stem capacity of all jobs
    std::vector<int32_t> chain_toff;      // per job: start of its slice of the chain's stem arrays
    // device pools (pools of any width, sq_pool.hip): device arrays carved from the workspace, pinned ones on first use
    SqPoolIO pool_io{};
    SqChain *h_pool_recs = nullptr; SqPoolJob *h_pool_jobs = nullptr; int32_t *h_pool_jobrec = nullptr;   // pinned staging
    // ---- device tail (sq_tail_dev.hip): the log of final structures and the tail's scratch are carved from the workspace,
    // the results land in pinned host memory in the C ABI's packed layout
    SqTailIO tail{};                      // device arrays (the per-fold fields are filled in by sq_fold)
    SqPoolFin *d_fin = nullptr; SqPoolStem *d_fin_stems = nullptr;
    uint32_t *d_fin_ctr = nullptr;        // [0] entries, [1] stems, [2] overflow, [3] the tail's fallback flag
    long long *d_job_evals = nullptr;
    int16_t *d_refp = nullptr; int32_t *d_refn = nullptr;
    uint32_t fin_cap = 0, fin_stem_cap = 0;
    long long *h_tail_totals = nullptr;   // pinned: SqTailIO::h_totals
    long long *h_rec_off = nullptr, *h_txt_off = nullptr;   // pinned: [nseq + 1] record / text offsets (sq_tail_offsets_kernel)
    char *h_rec = nullptr, *h_txt = nullptr; uint8_t *h_deep = nullptr;   // pinned: packed records, ASCII rows, deep flags
    size_t h_rec_cap = 0, h_txt_cap = 0;
    SqKept kept = {nullptr, nullptr, nullptr, nullptr, 0u, 0};   // kept lists of the pools (SQ_BATCH_POOL_LISTS): the workspace's region
    char *h_app = nullptr; size_t h_app_cap = 0;   // pinned staging of host-built log entries (sq_fin_append_kernel)
    char *h_ref = nullptr; size_t h_ref_cap = 0;   // pinned staging of the known structures (partner arrays)
    int tail_refs_state = 0;                       // sq_tail_refs: 0 not prepared for this fold, 1 no known structure, 2 uploaded
    bool packed_ok = false;               // the last fold's results are the packed records above (else: `results`)
    int tail_maxshow = 0;                 // most structures shown for one sequence in the last device tail (shapes the next pack launch)
    int32_t packed_limit = 0;             // result_limit in force at that fold
    // profiling
    std::mutex mwm_mu;
    int64_t mwm_stats[6] = {0, 0, 0, 0, 0, 0};   // blossom jobs collected, their scan passes; the job with the most passes:
                                                 // its passes, events, vertices, edges (reset by sq_profile_reset)
    bool prof_on = false;
    ProfSlot prof[9];                     // 0 fill, 1 state, 2 scan, 3 score, 4 Edmonds, 5 Hungarian, 6 Nussinov, 7 persistent rounds (sq_rounds.hip), 8 the alignment's scatter
};

// CPU accounting (SQ_CPUACC=1): thread CPU time spent in the host phases, summed over all threads, printed per fold.
// 0 tails, 1 RunAlgo collect per job (incl. 3 and 5), 2 edge lists (build), 3 pool growth (host loop) / RunAlgo's stem filters,
// 4 round post-processing, 5 round set-up + launches (host loop) / the per-job hook of streamed Edmonds results (incl. tails), 6 waits of the round driver, 7 AnnotateStems rounds of E/H/N, 8 the fold's calling thread in all, 9 sq_algos_begin,
// 10 sq_algos_end (its thread), 11 teardown of the pools
extern std::atomic<long long> g_cpuacc[12];
extern bool g_cpuacc_on;
struct CpuScope {
    int k; long long t0;
    static long long now() { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec * 1000000000ll + ts.tv_nsec; }
    explicit CpuScope(int k_) : k(k_), t0(g_cpuacc_on ? now() : 0) {}
    ~CpuScope() { if (g_cpuacc_on) g_cpuacc[k] += now() - t0; }
};
void sq_set_error(const std::string &msg);
// a status -3: WHICH capacity of the batch did not hold the fold (SQ_CAP_* of include/squarna_hip.h; read back by sq_last_capacity)
void sq_set_capacity_error(int kind, const std::string &msg);
// pinned (mapped, coherent) host buffers from a small process-wide cache: hipHostMalloc / hipHostFree cost milliseconds,
// and a caller that builds one batch per call (Predict) would pay them every time
int sq_pinned_get(void **p, size_t bytes);     // 0 or an error code (message set)
void sq_pinned_put(void *p);                   // the streams that used the buffer must be idle
// Process-wide caches of the objects a batch needs for a fold and that cost HIP a fraction of a millisecond each to make
// and to destroy (measured: 1.2 ms of a 10.6 ms Predict() on SRtest150 was sq_batch_destroy): non-blocking streams,
// timing-free events -- per device -- and the host worker pools.  A batch takes them when it first needs them and hands
// them back (idle) when it is destroyed.
const double *sq_host_reacts(const sq_batch *b);   // per-position reactivities on the host (formed on first use for reacts == NULL batches)
hipError_t sq_stream_get(int device, hipStream_t *s);
void sq_stream_put(int device, hipStream_t s);
hipError_t sq_event_get(int device, hipEvent_t *e);
void sq_event_put(int device, hipEvent_t e);
int sq_check(hipError_t e, const char *what);
// hipFuncAttributeMaxDynamicSharedMemorySize of `fn` on the CURRENT device, set once per (kernel, device): the
// attribute is per device and a process may fold on several (one worker thread per batch); thread-safe
void sq_max_dynamic_lds(const void *fn, int bytes);
int sq_effective_cpus();                       // CPUs this process may really use: hardware threads, affinity, cgroup quota
bool sq_relaxed_waits(const sq_batch *b);       // spin-then-sleep instead of pure spinning (many batches in flight, few CPUs)
void sq_wait_step(uint64_t spins, bool relaxed);   // one step of a wait loop on a pinned completion word
// Waits until the pinned word *flag holds `want` (written by the last kernel of the work enqueued on `st`): spins on the word
// -- no driver round trip, no staged copy --, polls the stream now and then so that a faulted queue is noticed, acquire fence
// at the end.  0, or an error code with sq_last_error set (`what` names the work in the message)
// at_least: the word counts upwards and later work on the stream writes it too (rounds enqueued ahead): wait for *flag - want >= 0
// in modular arithmetic instead of equality
int sq_wait_word(const sq_batch *b, volatile uint32_t *flag, uint32_t want, hipStream_t st, const char *what, bool at_least = false);
// sq_wait_step lowers the calling thread's timer slack while it sleeps in 10 us steps; entry points that may wait on the
// CALLER's thread put the old value back before they return
void sq_restore_timerslack();
struct SqSlackGuard { ~SqSlackGuard() { sq_restore_timerslack(); } };
// profiling bracket on an arbitrary stream (slot k of sq_profile_get); no-ops unless profiling is enabled
void sq_prof_begin(sq_batch *b, int k, hipStream_t st, hipEvent_t *e0);
void sq_prof_end(sq_batch *b, int k, hipStream_t st, hipEvent_t e0);
SqPool *sq_pool(sq_batch *b);             // the batch's worker pool (SQ_HOST_THREADS, default min(32, cores))

// bit matrices for the scan (full fp32 fill only for jobs with caller matrices / legacy scans)
int sq_prepare_scan(sq_batch *b);

// one greedy round (or a raw AnnotateStems pass) for a list of structures; results per structure
int sq_run_round(sq_batch *b, const std::vector<SView> &structs, int mode, std::vector<std::vector<HStem>> &out);

// AnnotateStems of E / H / N jobs with the stems left on the device (sq_host.hip; used by the device-side RunAlgo)
struct SqAlgoSize;
struct SqAlgoRaw;
int sq_round_annotate_dev(sq_batch *b, const std::vector<int> &jobs, SqAlgoSize *h_sizes, int64_t *cands_used, const SqAlgoRaw &raw);

// RunAlgo (SQRNdbnseq.py:548-595) for one of SQ_ALGO_E / H / N over a list of jobs
int sq_run_algo(sq_batch *b, const std::vector<int> &jobs, int algo, int levellimit_opt,
                std::vector<std::vector<HStem>> &out);

// asynchronous E/H/N for sq_fold: begin() annotates + launches on side streams, end() collects
struct SqAlgoAsync;
struct JobSets { int algo; std::vector<int> jobs; std::vector<std::vector<HStem>> sets; bool streamed = false; };
// optional hooks of sq_algos_end: after_short(sets) once the Hungarian / Nussinov stemsets are final; then
// on_e_job(job, set) per Edmonds job as soon as that job is final (called from pool workers; JobSets.streamed is set)
struct SqAlgoEndHooks {
    std::function<void(std::vector<JobSets> &)> after_short;
    std::function<void(int, std::vector<HStem> &)> on_e_job;
};
// want_dev: the whole of RunAlgo on the device (sq_algos_dev.hip) when the batch qualifies -- the stemsets then go to the
// device log of final structures and sq_algos_end returns no sets (sq_algos_on_device(pa) tells)
int sq_algos_begin(sq_batch *b, const std::vector<uint32_t> &algos, SqAlgoAsync *&pa, int levellimit_opt = -1, bool want_dev = false);
bool sq_algos_on_device(const SqAlgoAsync *pa);
int sq_algos_end(sq_batch *b, SqAlgoAsync *pa, int levellimit_opt, std::vector<JobSets> &sets,
                 const SqAlgoEndHooks *hooks = nullptr);
void sq_algos_abandon(sq_batch *b, SqAlgoAsync *pa);      // error paths: waits for the side streams, releases the arena

// device tail (sq_tail_dev.hip).  Launches the tail over the batch's device log on b->stream and waits for it.
// Returns 0: the packed results are in place (b->packed_ok), 1: the batch needs the host tail (nothing changed), else an error.
int sq_tail_device(sq_batch *b, const sq_fold_opts &o, const int32_t *ref_off, const int32_t *ref_pairs, const uint8_t *has_ref);
int sq_tail_refs(sq_batch *b, const int32_t *ref_off, const int32_t *ref_pairs, const uint8_t *has_ref);
// whether the options of this fold are covered by the device tail at all
bool sq_tail_device_wanted(const sq_batch *b, const sq_fold_opts &o);

// host tail: SQRNdbnseq.py:1201-1286
void sq_tail(const sq_batch *b, int seq, const sq_fold_opts &o,
             const std::vector<const std::vector<std::vector<HStem>> *> &per_job_structs,   // [job-of-seq] -> [structure][stem]
             const std::vector<int32_t> &job_ids, const int32_t *ref_pairs, int nref, bool has_ref,
             SeqResult &res);

// pseudoknot levels at pair level (PairsToDBN, SQRNdbnseq.py:104-150); pairs sorted & unique
int sq_pair_levels(const std::vector<std::pair<int, int>> &pairs, std::vector<int> &level);
// same at stem level (exactly equivalent for the stems of one structure)
void sq_stem_levels(const std::vector<HStem> &stems, std::vector<int> &level);
