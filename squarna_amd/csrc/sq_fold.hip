// sq_fold.hip -- sq_fold: the greedy pool loop of every job of a batch on the device drivers (persistent rounds, device pools) with the host loop as their fallback, E / H / N beside it, the ranking tail behind it; sq_fold_concurrent.
#include "sq_host_int.h"
#include "sq_scan.h"

void sq_read_fold_switches(SqFoldSwitches &sw)
{
    auto on = [](const char *name) { return getenv(name) != nullptr; };
    auto num = [](const char *name, int lo, int hi) { const char *e = getenv(name); return e ? std::max(lo, std::min(hi, atoi(e))) : 0; };
    sw.timing = on("SQ_TIMING"); sw.pool_debug = on("SQ_POOL_DEBUG");
    sw.no_chain = on("SQ_NO_CHAIN"); sw.no_rounds = on("SQ_NO_ROUNDS"); sw.no_pool = on("SQ_NO_POOL");
    sw.no_opt_chain = on("SQ_NO_OPT_CHAIN"); sw.no_fly_bits = on("SQ_NO_FLY_BITS"); sw.no_defer_wait = on("SQ_NO_DEFER_WAIT");
    sw.no_pool_round = on("SQ_NO_POOL_ROUND"); sw.pool_round_always = on("SQ_POOL_ROUND_ALWAYS");
    sw.pool_round_nsurv = num("SQ_POOL_ROUND_NSURV", 16, 2048);
    sw.pool_root = getenv("SQ_POOL_ROOT") ? num("SQ_POOL_ROOT", 0, 1) : 0;
    sw.no_pool_kept = on("SQ_NO_POOL_KEPT");
    sw.pool_ahead = getenv("SQ_POOL_AHEAD") ? num("SQ_POOL_AHEAD", 0, SQ_POOL_HDR_RING - 2) : 3;
    sw.pool_slots = num("SQ_POOL_SLOTS", 1, 0x7fffffff); sw.pool_chunk = num("SQ_POOL_CHUNK", 1, 0x7fffffff);
    sw.no_score_bound = on("SQ_NO_SCORE_BOUND"); sw.no_score_context = on("SQ_NO_SCORE_CONTEXT");
    sw.no_edges_lds = on("SQ_NO_EDGES_LDS");
    sw.no_device_algos = on("SQ_NO_DEVICE_ALGOS"); sw.no_device_tail = on("SQ_NO_DEVICE_TAIL");
    sw.algo_sync = on("SQ_ALGO_SYNC"); sw.lsap_classes = num("SQ_LSAP_CLASSES", 1, 64);
    sw.mwm_dump = on("SQ_MWM_DUMP"); sw.mwm_posthoc = on("SQ_MWM_POSTHOC");
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
    sq_read_fold_switches(b->sw);
    const SqFoldSwitches &sw = b->sw;
    struct FoldTimer { double t0; bool on; ~FoldTimer() { if (on) fprintf(stderr, "[sq_fold] total %.3f ms (incl. teardown)\n", (now_s() - t0) * 1e3); } } fold_timer{now_s(), sw.timing};
    int r = 0;
    // The ranking tail runs on the device (sq_tail_dev.hip) over the device log of final structures whenever the options
    // allow; the host tail below is its fallback.  The log and the per-job evaluation counts start empty.
    const bool dev_tail = sq_tail_device_wanted(b, o);
    // the scoring kernel's two short cuts, per fold (tests fold the same batch with and without them)
    b->score_bound = !sw.no_score_bound;
    b->score_ctx = !sw.no_score_context;
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
            // (pools that hold nothing -- chains and pools on the device drivers with the device tail: their structures never
            // reach the host lists -- go at once: starting the helper thread cost ~25 us of a 0.74-ms fold of 128 chains)
            bool empty = true;
            for (const JobPool &P : *p) if (!P.cur.empty() || !P.nxt.empty() || !P.fin.empty()) { empty = false; break; }
            if (sync_drop || empty) delete p; else std::thread([q = p] { CpuScope cpu_(11); delete q; }).detach();
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
    // a-1, once per job and per fold (:1076): never reused from an earlier call, a fold is the whole path.  A fold whose every
    // job is scanned exactly once -- width-1 pools on the persistent round kernel, no E / H / N -- does not write the bit
    // matrices at all: the kernel's only scan forms the words it needs from letter masks in LDS (SqBitsFly, sq_scan.h; the bit
    // kernel was 215 us of the 1.47 ms of an S1000 x 1,024 fold).  Every other path asks for the matrices (sq_prepare_scan).
    // (a job whose dense matrix the FILL forms -- caller matrices, bpp terms, a multiplier of its own: score x mul or score + bpp
    // in the fp64 arena, sq_kernels.hip -- needs that launch: the round kernel reads those cells as they are.  Only the rows
    // weighted by the alignment's shared matrix are formed elsewhere)
    bool any_ehn = false, any_fill = false;
    for (int j = 0; j < b->njobs; j++) {
        any_ehn |= (algos[j] & (uint32_t)(SQ_ALGO_E | SQ_ALGO_H | SQ_ALGO_N)) != 0;
        any_fill |= b->jobs[j].has_ext != 0 && !b->jobs[j].mat64_diag;
    }
    const bool no_fly = sw.no_fly_bits;
    const bool lazy_bits = o.poollim == 1 && !sw.no_chain && !sw.no_rounds && !any_ehn && !any_fill && !b->interchainonly && b->nletters > 0 && !no_fly;
    b->bits_ready = false;
    if (!lazy_bits) { r = sq_fill_impl(b, 0); if (r) return r; }
    // Edmonds / Hungarian / Nussinov paramsets (:1094-1100); their stemsets precede the greedy ones.
    // The reference iterates a Python set of letters (unspecified order); we use E, H, N.
    SqAlgoAsync *pending = nullptr;
    const double ta = now_s();
    if (sw.timing) fprintf(stderr, "[sq_fold] setup before E/H/N begin: %.3f ms (bit matrix launch + job pools)\n", (ta - fold_timer.t0) * 1e3);
    // (with the device tail: RunAlgo's filters on the device too when the batch qualifies, sq_algos_dev.hip)
    { CpuScope cpu_(9); r = sq_algos_begin(b, algos, pending, o.levellimit, dev_tail); }   // AnnotateStems + matching kernels on side streams
    const bool dev_algos = sq_algos_on_device(pending);
    b->last_paths = dev_algos ? 2 : 0;
    if (sw.timing && pending) fprintf(stderr, "[sq_fold] RunAlgo for E / H / N: %s\n", dev_algos ? "on the device (sq_algos_dev.hip)" : "host-driven");
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
    const bool no_chain = sw.no_chain, no_rounds = sw.no_rounds;   // (per fold: tests compare the drivers in one process)
    bool use_chain = o.poollim == 1 && !no_chain && !greedy_jobs.empty();
    if (use_chain)
        for (int j : greedy_jobs)
            if (chain_tcap(b->jobs[j].n, b->psets[b->job_pset[j]].minlen) > SQ_CHAIN_TMAX ||
                b->jobs[j].cand_cap > b->cand_records - b->cand_reserved) use_chain = false;
    if (!use_chain) { r = sq_prepare_scan(b); if (r) return r; }   // (the host-driven lanes and the pools scan the matrices; asked for before any second thread runs)
    // Wider pools: booked on the device as well (sq_pool.hip) when the batch has the slot arrays (structures of at most
    // SQ_CHAIN_TMAX stems) and one structure per greedy job fits the round buffers; any capacity overflow during the fold makes
    // the host repeat it with its own loop.
    const bool no_pool = sw.no_pool;
    bool use_pool = !use_chain && o.poollim > 1 && !no_pool && !greedy_jobs.empty() && b->pool_io.pt > 0;
    // the jobs each device driver takes.  Pools that may branch (poollim > 1) but rarely do -- range factor 1.0: only a run
    // that ties with the best AND shares a base with it branches (:769-789): `fastest` at the default pool limit, the
    // alignment's rows -- first run as chains on the persistent round kernel, which stops a structure at the first such tie;
    // the device pools then fold what is left (tied_jobs) and every other job
    std::vector<int> chain_jobs, pool_jobs_v, tied_jobs;
    bool chain_ties = false;
    if (use_chain) chain_jobs = greedy_jobs;
    if (use_pool) {
        const bool no_opt = sw.no_opt_chain;
        for (int j : greedy_jobs) {
            const SqJob &J = b->jobs[j];
            const sq_paramset &ps = b->psets[b->job_pset[j]];
            const bool opt = !no_opt && !no_rounds && ps.suboptmin == 1.0 && ps.suboptmax == 1.0 && J.n <= SQ_ROUNDS_MAXN &&
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
    const bool timing = sw.timing;
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
        std::mutex mu; std::condition_variable cv, idle_cv; std::vector<int> items; bool closed = false, busy = false;
        std::thread worker;
        std::function<void()> start; std::once_flag once; // (the worker starts with the first push -- the lanes push from threads of their own --:
                                                         // a fold whose drivers and tail stay on the device never feeds the queue, and
                                                         // starting + joining a thread was ~40 us of it)
        void push(std::vector<int> &v) { if (v.empty()) return; if (start) std::call_once(once, start); { std::lock_guard<std::mutex> lk(mu); items.insert(items.end(), v.begin(), v.end()); } cv.notify_one(); v.clear(); }
        // everything pushed so far has been handled when this returns (the worker stays: later pushes are served as before)
        void flush() { if (!worker.joinable()) return; std::unique_lock<std::mutex> lk(mu); idle_cv.wait(lk, [&] { return items.empty() && !busy; }); }
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
        // (optimistic chains in front of the device pools: a sequence's other jobs may still be the pools', and a capacity overflow
        // there hands EVERY greedy job to the host loop -- nothing is ranked before the pools are through)
        if (early_tail && !chain_ties && --g_left[s2] == 0) { tail_one(s2); tailed[s2] = 1; }
    };
    if (early_tail) {
        for (int s2 = 0; s2 < b->nseq; s2++) g_left[s2] = 0;
        for (int j : greedy_jobs) g_left[b->job_seq[j]]++;
    }
    if (early_tail || use_chain || chain_ties) tq.start = [&] {
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
                    tq.busy = true;
                }
                sq_pool(b)->parallel_for((int)take.size(), [&](int k) {
                    if (take[k] < 0) chain_finish((uint32_t)(-(take[k] + 1)));       // (items < 0: chain entries)
                    else { tail_one(take[k]); tailed[take[k]] = 1; }
                });
                { std::lock_guard<std::mutex> lk(tq.mu); tq.busy = false; }
                tq.idle_cv.notify_all();
            }
        });
    };
    mark("tail queue");
    // the greedy pool loop (:1102-1199) for a subset of the jobs, on one lane of round buffers
    struct LoopStats { double tround = 0, twall = 0, tstart = 0; int nrounds = 0; int rc = 0; int cap = 0; std::string err; };   // (cap: SQ_CAP_* of a status -3)
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
            { const double t0 = now_s(); stats.rc = sq_run_round_impl(b, ln, round, 0, res, nullptr); stats.tround += now_s() - t0; stats.nrounds++; }
            if (stats.rc) { stats.err = sq_last_error(); stats.cap = sq_last_capacity(); return; }
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
    // One launch that covers every chain, the ranking tail on the device, no E / H / N beside it: the tail's kernels are
    // enqueued right behind the round kernel and the host waits ONCE, for the tail's last word -- the chain's own completion
    // (capacity flags, the count of finished structures) is looked at afterwards (the wait between the two was 40-65 us of every
    // fold: a flag's way to the host, then seven launches' way back)
    struct { bool on = false; uint32_t goal = 0; } deferred;

    const bool no_defer = sw.no_defer_wait;
    auto chain_fold = [&](LoopStats &stats) {
        SqLane &ln = b->lane_full;
        hipStream_t st = b->stream;
        const double tl0 = now_s();
        stats.tstart = tl0 - tfold0;
        struct Wall { double t0; double &dst; ~Wall() { dst = now_s() - t0; } } wall{tl0, stats.twall};
        auto fail = [&](int rc, const std::string &msg, int cap = 0) { stats.rc = rc; stats.err = msg; stats.cap = cap; };
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
        // ONE launch for all rounds of these structures (sq_rounds.hip: a persistent block per structure) when every job
        // qualifies: per-position arrays and lists that fit the block's LDS.  Decided BEFORE anything is enqueued: a chain of
        // pools that may branch which the kernel cannot take goes to the device pools as it is, and the init kernel -- which
        // reads the pinned records the next chain overwrites -- is then never launched
        bool rounds_ok = !no_rounds;
        for (int j : jobs) rounds_ok = rounds_ok && b->jobs[j].n <= SQ_ROUNDS_MAXN;
        int thr = 64;
        SqRoundsArgs ra;
        memset(&ra, 0, sizeof(ra));
        if (rounds_ok) {
            static const int thr_env = getenv("SQ_ROUNDS_THREADS") ? std::max(64, std::min(SQ_ROUNDS_THREADS, atoi(getenv("SQ_ROUNDS_THREADS")) / 64 * 64)) : 0;
            // threads per structure: by length -- and, while the launch leaves the chip empty (a shard of a multi-GPU run, a
            // small batch), twice / four times that: a structure's rounds are a chain of dependent passes over its list that
            // more waves shorten (S1000 x 128: 1.21 -> 0.99 ms at 512 threads)
            // (end of round 6, 10,000 / 1,024 / 1,000 chains on one box: 300 nt 64 / 128 / 256 threads 1.93 / 2.08 / 2.63 ms; 1,000 nt
            // 128 / 256 / 512: 1.54 / 1.27 / 1.69; 2,000 nt 256 / 512 / 1,024: 6.59 / 6.28 / 8.33)
            thr = maxn <= 320 ? 64 : (maxn <= 450 ? 128 : (maxn < 1800 ? 256 : 512));                      // (1,500 nt x 1,000: 256 / 512 threads 2.93 / 3.21 ms)
            // (up to one block of 1,024 per CU: S2000 x 125 3.10 -> 2.80 ms with 1,024 instead of 512 threads; S1000 x 128 0.86 / 0.76 /
            // 0.75 ms with 256 / 512 / 1,024 -- there the pass over the list is no longer what a round waits for)
            while (!thr_env && thr < SQ_ROUNDS_THREADS && thr < maxn / 2 && ((int64_t)S * thr * 2 <= (int64_t)256 * 1024 || (S <= 256 && thr * 2 <= maxn / 2 + 64))) thr *= 2;   // (the chip's 256 x 16 wave slots: 1,250 chains of 300 nt 0.50 -> 0.47 ms at 128 threads, 512 of 1,000 nt 1.05 -> 0.99 at 512; 2,500 x 300 nt stay at 64: 0.65 against 0.72)
            if (thr_env) { thr = 64; while (thr * 2 <= thr_env) thr *= 2; }   // (a power of two: the survivor ring is indexed with a mask)
            ra.lds_n = maxn; ra.str_cap = 2 * maxt + 2; ra.tmax = maxt; ra.cell_entries = b->cell_entries;
            ra.su = 0;
            for (int j : jobs) if (b->seq_has_sep[(size_t)b->job_seq[j]]) { ra.su = 1; break; }
            // Pools that may branch (their chains can hand a job to the device pools): when the lists sized for a structure's BOUND
            // of stems -- n / (2 minlen): 1,178 at 4,700 nt, where a row takes ~370 -- keep a CU to one block, they are sized for
            // fewer (the largest of a few steps that lets two blocks of 512 threads share a CU); a structure that outgrows them
            // stops like one that meets a tie.  SQ_ROUNDS_TLDS=n: that size by hand (tests force the hand-over)
            if (chain_ties && !thr_env) {
                const int tenv = getenv("SQ_ROUNDS_TLDS") ? std::max(1, atoi(getenv("SQ_ROUNDS_TLDS"))) : 0;
                auto fits2 = [&](int t) { return sq_rounds_lds(ra.lds_n, 2 * t + 2, t, ra.cell_entries, 512, ra.su).total + 2048 <= 80 * 1024; };
                if (tenv) { if (tenv < maxt) { ra.tmax = tenv; ra.str_cap = 2 * tenv + 2; } }
                else if (S > 256 && maxn >= 1024 && !fits2(maxt))
                    for (int t : {1024, 768, 640, 512, 448}) if (t < maxt && fits2(t)) { ra.tmax = t; ra.str_cap = 2 * t + 2; thr = std::max(thr, 512); break; }
            }
            // (long sequences: the per-position arrays and strand lists of ONE block fill most of a CU's LDS -- 90 KB at 4,700 nt --, so
            // the CU holds one block however many there are: it takes the wave slots the others cannot use.  512 rows of an
            // alignment ran as 512 blocks of four waves on 256 CUs)
            while (!thr_env && thr < SQ_ROUNDS_THREADS && thr < maxn / 2) {
                const size_t l1 = sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, thr, ra.su).total + 2048;
                const size_t l2 = sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, 2 * thr, ra.su).total + 2048;
                const size_t cu = 160 * 1024, r1 = std::min<size_t>(cu / l1 * thr, 1024), r2 = l2 <= 158 * 1024 ? std::min<size_t>(cu / l2 * 2 * thr, 1024) : 0;
                // (only while the LDS keeps a CU below half of its wave slots: a dozen one-wave blocks of 300-nt structures per CU
                // are better off as they are -- doubled, 10,000 chains of 300 nt took 2.08 instead of 1.93 ms)
                if (r2 > r1 && r1 <= 512) thr *= 2; else break;
            }
            ra.bound = b->score_bound ? 1 : 0; ra.ctx_min = 0; ra.ties = chain_ties ? 1 : 0;
            {
                const int wmin = getenv("SQ_WAVE_WALK_MIN") ? atoi(getenv("SQ_WAVE_WALK_MIN")) : 192;
                const int wlanes = getenv("SQ_WAVE_WALK_LANES") ? atoi(getenv("SQ_WAVE_WALK_LANES")) : 12;
                ra.wave_min = wmin > 0 ? wmin : 0x7fffffff; ra.wave_lanes = wlanes; ra.no_early = getenv("SQ_NO_EARLY_WALK") ? 1 : 0;
            }
            ra.fly = 0;
            while (thr > 64 && sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, thr, ra.su).total + 2048 > 158 * 1024) thr /= 2;   // (long sequences: the survivor ring gives way)
            if (sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, thr, ra.su).total + 2048 > 158 * 1024) rounds_ok = false;
        }
        if (rounds_ok && lazy_bits) {                         // the masks take the LDS of the strands and stems (the structure is empty during the scan)
            const SqRoundsLds lo = sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, thr, ra.su);
            // (every length since the end of round 6: 10,000 chains of 100 / 150 / 250 / 300 / 350 nt 1.165 -> 1.128 / 1.337 -> 1.256 /
            // 1.787 -> 1.639 / 2.29 -> 2.12 / 2.607 -> 2.444 ms, S1000 x 1,024 1.47 -> 1.31.  Round 5 had measured S300 x 10,000 at
            // 1.78 with the bit kernel's matrices against 1.84 and kept them below 400 nt; the kernel has changed since.  SQ_FLY_MIN_N
            // sets a shortest length again)
            static const int fly_min = getenv("SQ_FLY_MIN_N") ? atoi(getenv("SQ_FLY_MIN_N")) : 0;
            if (maxn >= fly_min && b->nletters <= SQ_FLY_MAXL && sq_bits_fly_bytes(maxn, b->nletters) <= (size_t)(lo.off_tab - lo.off_str)) ra.fly = b->nletters;
        }
        if (ra.fly == 0) { const int pr = sq_prepare_scan(b); if (pr) { fail(pr, sq_last_error()); return; } }
        if (!rounds_ok && chain_ties) {                       // (the launched rounds do not look for ties: the pools take these jobs)
            for (int j : jobs) tied_jobs.push_back(j);
            nfin_goal -= (uint32_t)S;
            continue;
        }
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
        if (rounds_ok) {
            const SqRoundsLds lo = sq_rounds_lds(ra.lds_n, ra.str_cap, ra.tmax, ra.cell_entries, thr, ra.su);
            {
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
                if (dev_tail && !chain_ties && !b->prof_on && !any_ehn && !no_defer && next_job == chain_jobs.size() &&
                    nfin_goal == (uint32_t)S) {
                    deferred.on = true; deferred.goal = nfin_goal;
                    if (timing) fprintf(stderr, "[sq_fold] persistent rounds: the wait is deferred behind the device tail\n");
                    stats.nrounds += 1;
                    stats.tround += now_s() - tr0;
                    return;
                }
                if (timing) fprintf(stderr, "[sq_fold] persistent rounds: waiting (dev_tail %d ties %d prof %d pending %d next_job %zu of %zu goal %u S %d)\n", (int)dev_tail, (int)chain_ties, (int)b->prof_on, pending != nullptr, next_job, chain_jobs.size(), nfin_goal, S);
                { const int wr = sq_wait_word(b, flag, seq, st, "persistent rounds"); if (wr) fail(wr, sq_last_error()); }
                if (!stats.rc) {
                    const SqCounters ctr = *ln.h_ctr;
                    if (ctr.cand_ovf) fail(-3, "candidate capacity exceeded (raise cand_per_nt)", SQ_CAP_CANDIDATES);
                    else if (ctr.out_ovf) fail(-3, "stem capacity of a chained structure exceeded", SQ_CAP_FIXED);
                    else if (ctr.level_ovf) fail(-3, "more than 64 pseudoknot levels", SQ_CAP_FIXED);
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
        while (!rounds_ok && nfin_seen < nfin_goal) {
            while (launched - done < depth) {               // rounds enqueued ahead of the device
                if ((int)launched > maxt + 2) { fail(2, "chained rounds do not terminate"); break; }
                // (algorithmic bytes: NOT per launch -- a launch also covers the structures that are already final; they are
                // booked below from the evaluations the list of finished structures records)
                sq_launch_round_kernels(b, st, S, maxn, maxcap, need_reacts, 0.0, 0, io, scan, ln.d_structs, b->chain.strands, true);
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
                if (ctr.cand_ovf) { fail(-3, "candidate capacity exceeded (raise cand_per_nt)", SQ_CAP_CANDIDATES); break; }
                if (ctr.out_ovf) { fail(-3, "stem capacity of a chained structure exceeded", SQ_CAP_FIXED); break; }
                if (ctr.level_ovf) { fail(-3, "more than 64 pseudoknot levels", SQ_CAP_FIXED); break; }
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
        auto fail = [&](int rc, const std::string &msg, int cap = 0) { stats.rc = rc; stats.err = msg; stats.cap = cap; return 2; };
        if (!PI.h_hdr) {
            void *p2 = nullptr, *p3 = nullptr, *p4 = nullptr, *p5 = nullptr, *p6 = nullptr;
            if (sq_pinned_get(&p2, sizeof(SqPoolHdr) * SQ_POOL_HDR_RING) || sq_pinned_get(&p3, sizeof(SqPoolJob) * (size_t)b->njobs) ||
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
        { const int pr = sq_prepare_scan(b); if (pr) return fail(pr, sq_last_error()); }
        const int64_t avail = b->cand_records - b->cand_reserved;
        int slots = std::min(PI.smax, ln.max_structs);
        if (sw.pool_slots > 0) slots = std::min(slots, sw.pool_slots);   // (tests: force the overflow path)
        // structures whose candidates fit the arena at once; larger generations go through state .. choose in chunks
        // root lists (sequences beyond the scanning round kernel's 256 nt, up to 1,024): 16 bytes per run of the EMPTY structure
        // of every job, behind the structures' regions of the arena
        const int64_t root_units = (maxcap + 1) / 2;
        // kept lists (the batch reserved their pages: SQ_BATCH_POOL_LISTS): the same kernel, the parent's list in the root list's place
        const bool kept_mode = b->kept.on && !sw.no_pool_kept;
        bool root_mode = (sw.pool_root || kept_mode) && !sw.no_pool_round && maxn > SQ_PR_MAXN && maxn <= SQ_PR_ROOT_MAXN &&
                         (int64_t)S0 * root_units + std::max<int64_t>(maxcap, 1) <= avail;
        const int64_t avail_s = root_mode ? avail - (int64_t)S0 * root_units : avail;
        int chunk = (int)std::min<int64_t>(slots, avail_s / std::max<int64_t>(maxcap, 1));
        if (sw.pool_chunk > 0) chunk = std::min(chunk, sw.pool_chunk);   // (tests: force chunked rounds)
        if (S0 > slots || chunk < 1) return 1;
        // On kept lists a structure keeps no candidates in the arena: its region only takes the runs within range of the best
        // finalscore that LDS has no room for -- SQ_KEPT_SLICE units, not the thousands a scan's output needs -- and a round is
        // one launch (at 500 nt the arena held the candidates of 8,224 structures: a generation of 170,000 went through it in
        // twenty launches of four waves of blocks each).  The root kernel still stages a job's runs in a full region: its
        // launches keep the regions' size.
        const int chunk_root = chunk;
        const int64_t kslice = 512;
        const bool kept_round = root_mode && kept_mode &&
                                sq_pool_round_lds(maxn, 2 * PI.pt + 2, b->cell_entries, sw.pool_round_nsurv ? std::max(sw.pool_round_nsurv, 128) : 128, PI.pt).total <= 60 * 1024;
        if (kept_round) {
            chunk = (int)std::min<int64_t>(slots, ((int64_t)chunk_root * maxcap) / kslice);
            if (sw.pool_chunk > 0) chunk = std::min(chunk, sw.pool_chunk);
        }
        for (int j = 0; j < b->njobs; j++) b->h_pool_jobrec[j] = -1;
        for (int sx = 0; sx < S0; sx++) {
            const int j = jobs[sx];
            const JobPool &P = pools[j];
            const int toff = sx * PI.pt;                     // generation 0, slot sx
            SqStruct &d = ln.h_structs[sx];
            d.job = j; d.strand_off = 2 * toff; d.nstrand = 0; d.slot = sx; d.subopt = P.cursubopt; d.cand_off = (int64_t)(sx % chunk_root) * maxcap;
            SqChain &cr = b->h_pool_recs[sx];
            cr.toff = toff; cr.tcap = PI.pt; cr.nstems = 0; cr.anycross = 0; cr.maxstems = P.maxstemnum;
            SqPoolJob &pj = b->h_pool_jobs[sx];
            pj.first = sx; pj.count = 1; pj.cursize = 1; pj.job = j;
            pj.cursubopt = P.cursubopt; pj.suboptinc = P.suboptinc; pj.suboptmax = P.suboptmax; pj.maxstems = P.maxstemnum; pj.evals = 0;
            b->h_pool_jobrec[j] = sx;
        }
        PI.slots = slots; PI.chunk = chunk; PI.poollim = o.poollim; PI.maxcap = kept_round ? kslice : maxcap; PI.njobs = S0;   // (the kernels take the batch's record)
        PI.kept_ctr = kept_round ? b->kept.ctr : nullptr;
        if (PI.kept_ctr) hipMemsetAsync(PI.kept_ctr, 0, 16, st);
        const SqPoolIO pio = PI;
        SqScanArgs scan = b->scan;
        scan.ctr = ln.d_ctr;
        hipLaunchKernelGGL(sq_pool_init_kernel, dim3((std::max(S0, b->njobs) + 255) / 256), dim3(256), 0, st, ln.h_structs, b->h_pool_recs,
                           b->h_pool_jobs, b->h_pool_jobrec, (int32_t *)pio.jobrec_of, b->njobs, pio, scan, S0);
        auto wait_seq = [&](uint32_t seq, bool at_least = false) -> int {
            const int wr = sq_wait_word(b, ln.h_seq, seq, st, "pool round", at_least);
            return wr ? fail(wr, sq_last_error()) : 0;
        };
        // short sequences: a round is ONE kernel (sq_pool_round.hip) + the scan kernel -- on a crowded chip because wave slots
        // are what it runs out of, for a batch alone because two launches per round instead of six shorten the greedy loop
        // (SRtest150: 1.43 -> 1.28 ms, and the loop depends less on how fast the host turns a round around)
        SqPoolRoundArgs pra;
        bool round_kernel = (maxn <= SQ_PR_MAXN || root_mode) && !sw.no_pool_round;   // (jobs with a dense matrix too: sq_cellrun.h reads their cells there)
        if (round_kernel) {
            pra.lds_n = maxn; pra.str_cap = 2 * pio.pt + 2; pra.cell_entries = b->cell_entries;
            // survivors of :492 kept in LDS (the rest spill to the arena): on a crowded chip LDS is what the round kernel's waves
            // and everybody else's compete for -- 22 bytes x 256 survivors were half of a wave's 10 KB, and most structures have
            // a few dozen (round 4, a sweep of the count: 64 -> +5 % on the headline, 16 .. 64 within a per cent of each other)
            const bool crowded_fold = b->inflight > 1 || b->njobs >= 4096;
            pra.surv_cap = sw.pool_round_nsurv ? std::max(sw.pool_round_nsurv, root_mode ? 128 : 0) : (root_mode ? (kept_round ? 128 : 256) : (crowded_fold ? 64 : (maxn <= 96 ? 128 : 256))); pra.bound = b->score_bound ? 1 : 0;
            pra.tmax = pio.pt; pra.parity = 0; pra.lo = 0; pra.ahead = 0;
            pra.root = root_mode ? 1 : 0; pra.root_units = (int32_t)root_units; pra.root_off = (int64_t)chunk_root * maxcap;
            pra.kept = b->kept; pra.kept.on = kept_round ? 1 : 0;
            if (sq_pool_round_lds(pra.lds_n, pra.str_cap, pra.cell_entries, pra.surv_cap, pra.tmax).total > 60 * 1024) round_kernel = false;
        }
        if (round_kernel) b->last_paths |= 8;
        if (round_kernel && root_mode) {
            // the jobs' root lists: AnnotateStems of every job's empty structure, once (one wave per job)
            const size_t rl = sq_pool_root_lds(pra.lds_n, pra.cell_entries);
            if (rl > 60 * 1024) sq_max_dynamic_lds((const void *)sq_pool_root_kernel, 160 * 1024);
            // (in launches of at most `chunk` jobs: the kernel stages a job's runs in its empty structure's region of the arena,
            // and structures a chunk apart share a region)
            SqPoolIO pio_root = pio;
            pio_root.chunk = chunk_root; pio_root.maxcap = maxcap;
            for (int lo = 0; lo < S0; lo += chunk_root) {
                pra.lo = lo;
                hipLaunchKernelGGL(sq_pool_root_kernel, dim3(std::min(chunk_root, S0 - lo)), dim3(64), rl, st, b->ctx, scan, pio_root, pra);
            }
            pra.lo = 0;
            b->last_paths |= 64;
            if (kept_round) b->last_paths |= 128;
        }
        const size_t ext_lds = sq_extend_lds_bytes(pio.pt);          // the extend kernel's level scratch (dynamic LDS)
        if (ext_lds > 64 * 1024) sq_max_dynamic_lds((const void *)sq_pool_extend_kernel, 160 * 1024);
        const double tr0 = now_s();
        int parity = 0, S = S0, rounds = 0;
        bool overflow = false;
        // A batch alone: its rounds are a chain of short kernels, and waiting for a round's size before launching the next put
        // the host's turn-around -- a PCIe round trip and two launch latencies -- between every two of them (half of the greedy
        // loop of one SRtest150 batch).  With the one-kernel round the rounds are enqueued AHEAD instead: every launch covers
        // all the slots, blocks beyond the generation's size leave at once (sq_pool_round_kernel reads the size the scan kernel
        // left), and the host only follows the ring of published headers to learn when the pools have run empty.  Rounds
        // launched behind the last one find an empty generation.  (A crowded chip hides the turn-around behind other batches'
        // work and has tens of thousands of slots: it keeps the exact grids.)
        const int ahead_env = sw.pool_ahead;
        // (the slots in at most four launches per round: a generation larger than the candidate arena goes through it in chunks)
        const bool ahead = round_kernel && ahead_env > 0 && !(b->inflight > 1 || b->njobs >= 4096) && slots <= 8192 && (int64_t)chunk * 4 >= slots;
        if (ahead) {
            SqRoundIO io;
            io.h_strands = pio.strands; io.d_strands = pio.strands;
            io.h_out = ln.h_out; io.d_out = ln.d_out; io.h_cap = 0; io.out_cap = 0;
            io.h_ctr = ln.h_ctr; io.h_seq = ln.h_seq;
            io.h_structs = pio.structs; io.d_structs = pio.structs;
            int launched = 0, par_l = 0;
            bool stop = false;
            b->last_paths |= 32;
            while (!stop || rounds < launched) {
                while (!stop && launched - rounds < ahead_env) {
                    if (launched > 4 * PI.pt + 8) return fail(2, "pool rounds do not terminate");
                    pra.parity = par_l; pra.ahead = 1;
                    for (int lo = 0; lo < slots; lo += chunk) {
                        pra.lo = lo;
                        sq_launch_round_kernels(b, st, std::min(chunk, slots - lo), maxn, maxcap, need_reacts, 0.0, 0, io, scan, pio.structs + (size_t)par_l * pio.smax + lo, pio.strands, true, true, &pra);
                    }
                    const uint32_t seq = ++*ln.round_seq;
                    hipLaunchKernelGGL(sq_pool_scan_kernel, dim3(1), dim3(1024), 0, st, pio, scan, io, par_l, seq);
                    launched++; par_l ^= 1;
                }
                const uint32_t seq = *ln.round_seq - (uint32_t)(launched - rounds - 1);   // the oldest round still out
                if (wait_seq(seq, true)) return 2;                  // (the rounds behind it write the same word: at least this one)
                rounds++;
                const SqCounters ctr = *ln.h_ctr;
                if (ctr.cand_ovf) return fail(-3, "candidate capacity exceeded (raise cand_per_nt)", SQ_CAP_CANDIDATES);
                if (ctr.level_ovf) return fail(-3, "more than 64 pseudoknot levels", SQ_CAP_FIXED);
                const SqPoolHdr hh = pio.h_hdr[seq % SQ_POOL_HDR_RING];
                if (timing && sw.pool_debug) fprintf(stderr, "[pool] round %d (of %d enqueued): next generation %u, nfin %u, ovf %u, active jobs %u\n", rounds, launched, hh.S[(rounds & 1)], hh.nfin, hh.ovf, hh.active_jobs);
                b->last_peak = std::max<int64_t>(b->last_peak, hh.peak);
                if (hh.ovf) { overflow = true; stop = true; }
                if (hh.S[rounds & 1] == 0) stop = true;         // (round r has parity r & 1; its scan kernel wrote the size of round r + 1)
            }
            S = 0;
        }
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
                sq_launch_round_kernels(b, st, std::min(chunk, S - lo), maxn, maxcap, need_reacts, 0.0, 0, io, scan, cur + lo, pio.strands, true, true,
                                     round_kernel ? &pra : nullptr);
            }
            const uint32_t seq = ++*ln.round_seq;
            hipLaunchKernelGGL(sq_pool_scan_kernel, dim3(1), dim3(b->inflight > 1 ? 256 : 1024), 0, st, pio, scan, io, parity, seq);
            // (4 waves share a parent's children; on a crowded chip ONE takes them all: most parents have one or two, and a wave
            // that finds nothing to do still takes a slot for a microsecond or two -- 593 k -> 601 k)
            static const int ext_crowd = getenv("SQ_POOL_EXTEND_WAVES") ? std::max(1, std::min(16, atoi(getenv("SQ_POOL_EXTEND_WAVES")))) : 1;
            const bool crowded = b->inflight > 1 || b->njobs >= 4096;
            if (!round_kernel)       // (the round kernel's structures extend themselves and log themselves)
                hipLaunchKernelGGL(sq_pool_extend_kernel, dim3(S, crowded ? ext_crowd : 4), dim3(64), ext_lds, st, b->ctx, scan, pio, parity);
            if (wait_seq(seq)) return 2;
            rounds++;
            const SqCounters ctr = *ln.h_ctr;
            if (ctr.cand_ovf) return fail(-3, "candidate capacity exceeded (raise cand_per_nt)", SQ_CAP_CANDIDATES);
            if (ctr.level_ovf) return fail(-3, "more than 64 pseudoknot levels", SQ_CAP_FIXED);
            const SqPoolHdr hh = pio.h_hdr[seq % SQ_POOL_HDR_RING];
            if (timing && sw.pool_debug) fprintf(stderr, "[pool] round %d: S %d -> %u, nfin %u, ovf %u, active jobs %u\n", rounds, S, hh.S[parity ^ 1], hh.nfin, hh.ovf, hh.active_jobs);
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
        if (timing && PI.kept_ctr) {
            uint32_t kc[4] = {0, 0, 0, 0};
            hipMemcpy(kc, PI.kept_ctr, 16, hipMemcpyDeviceToHost);
            fprintf(stderr, "[pool] kept lists: %u pages per generation, most taken %u, structures that left no list %u\n", b->kept.npages, kc[2], kc[3]);
        }
        const SqPoolHdr hh = pio.h_hdr[*ln.round_seq % SQ_POOL_HDR_RING];
        if (overflow || hh.ovf) {
            tq.flush();                                      // (the optimistic chains' entries are still being turned into lists by the queue's workers)
            for (int j : greedy_jobs) { pools[j].fin.clear(); pools[j].evals = 0; }
            // (the device log holds the structures the aborted pools had finished: they leave it for the host loop's.  The E / H / N
            // stemsets of the device RunAlgo stay -- their finish kernels append on the side streams: wait for them first, the
            // host loop that follows is the slow path anyway.  Round 3 emptied the whole log here and lost those stemsets)
            for (int q = 0; q < 4; q++) if (b->side[q]) hipStreamSynchronize(b->side[q]);
            hipLaunchKernelGGL(sq_fin_keep_algos_kernel, dim3(1), dim3(1024), 0, st, b->d_fin, b->d_fin_ctr, b->fin_cap, b->d_job_evals, b->tail.job_cnt, b->njobs);
            return 1;
        }
        if ((*ln.h_ctr).level_ovf) return fail(-3, "more than 64 pseudoknot levels", SQ_CAP_FIXED);
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
        if (st0.rc) { tq.close(); sq_set_capacity_error(st0.rc == -3 ? st0.cap : 0, st0.err); return st0.rc; }
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
            tq.flush();                                      // (no worker is still filling the lists the host loop starts from)
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
        if (!st0.rc && st1.rc) { st0.rc = st1.rc; st0.err = st1.err; st0.cap = st1.cap; }
    }
    tq.close();
    if (st0.rc) { sq_set_capacity_error(st0.rc == -3 ? st0.cap : 0, st0.err); return st0.rc; }
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
        if (deferred.on) {
            // (the tail's last word is behind the round kernel's in stream order: after an error of the tail the stream is
            // drained first)
            if (rt) hipStreamSynchronize(b->stream);
            std::atomic_thread_fence(std::memory_order_acquire);
            const SqCounters ctr = *b->lane_full.h_ctr;
            if (ctr.cand_ovf) { sq_set_capacity_error(SQ_CAP_CANDIDATES, "candidate capacity exceeded (raise cand_per_nt)"); return -3; }
            if (ctr.out_ovf) { sq_set_capacity_error(SQ_CAP_FIXED, "stem capacity of a chained structure exceeded"); return -3; }
            if (ctr.level_ovf) { sq_set_error("more than 64 pseudoknot levels"); return -3; }
            if (*b->chain.h_nfin != deferred.goal) { sq_set_error("persistent rounds left structures unfinished"); return 2; }
        }
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

extern "C" int sq_batch_set_inflight(sq_batch *b, int32_t n)
{
    if (!b) { sq_set_error("bad argument"); return -1; }
    b->inflight = n < 1 ? 1 : n;
    b->side_streams = b->inflight >= 3 ? 2 : 3;
    return 0;
}

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

