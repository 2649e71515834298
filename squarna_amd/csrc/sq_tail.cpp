// sq_tail.cpp -- host tail of SQRNdbnseq (a-10): structure dedupe, ScoreStruct, RankStructs,
// consensus, pseudoknot bracket levels and the TP/FP/FN metrics.  SQRNdbnseq.py:861-955,1201-1286.
// Plain host C++ (the reference does this part in Python; it is O(#structures * N)).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iterator>
#include <chrono>
#include <map>
#include <unordered_map>
#include <set>
#include <vector>
#include "sq_host.h"

typedef std::pair<int, int> BP;

// Python round(x, 3): correctly rounded decimal, ties to even on the exact binary value.
static double py_round3(double x)
{
    if (!std::isfinite(x)) return x;
    // Fast path: k = round(1000 x) decides the decimal whenever 1000 x is not within 1e-6 of a tie -- the product is off by
    // at most half an ulp of y = 1000 x, which is <= 6e-8 for |x| < 1e6 (|y| < 2^30: ulp 2^-23), well inside the guard;
    // larger scores (giant sequences reach 1e6 .. 1e7) take the exact path -- and k / 1000.0 -- one correctly rounded
    // division of two exact integers -- is the double nearest to that decimal, i.e. what strtod returns for its text.
    if (std::fabs(x) < 1e6) {
        const double y = x * 1000.0, f = std::floor(y), frac = y - f;
        if (std::fabs(frac - 0.5) > 1e-6) return (frac > 0.5 ? f + 1.0 : f) / 1000.0;
    }
    char buf[512];
    snprintf(buf, sizeof buf, "%.3f", x);
    return strtod(buf, nullptr);
}

static inline bool crosses(const BP &p, const BP &q)                   // :114-116
{
    return (p.first < q.first && q.first < p.second && p.second < q.second) ||
           (q.first < p.first && p.first < q.second && q.second < p.second);
}

// PairsToDBN level assignment (:119-150) for sorted unique pairs.  Returns #groups.
int sq_pair_levels(const std::vector<BP> &pairs, std::vector<int> &level)
{
    const int np = (int)pairs.size();
    level.assign(np, 0);
    if (!np) return 0;
    // scratch kept per thread (this runs per structure in the tail and in RunAlgo's filters): allocation-free when warm
    static thread_local std::vector<int> cc, order, grp, gord, rank, gsize, members, gnext;
    cc.assign(np, 0); order.resize(np);
    for (int a = 0; a < np; a++) {
        for (int b = 0; b < np; b++)
            if (a != b && crosses(pairs[a], pairs[b])) cc[a]++;
        order[a] = a;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {  // :125
        if (cc[a] != cc[b]) return cc[a] < cc[b];
        return pairs[a].first < pairs[b].first;
    });
    // groups as linked lists in insertion order: members[g] = first pair of group g, gnext[p] = next pair of p's group
    grp.resize(np); gnext.assign(np, -1); members.clear(); gsize.clear();
    static thread_local std::vector<int> gtail;
    gtail.clear();
    for (int t = 0; t < np; t++) {                                     // :130-136
        const int p = order[t];
        int placed = -1;
        for (size_t g = 0; g < members.size() && placed < 0; g++) {
            bool ok = true;
            if (cc[p])
                for (int q = members[g]; q >= 0; q = gnext[q]) if (crosses(pairs[p], pairs[q])) { ok = false; break; }
            if (ok) placed = (int)g;
        }
        if (placed < 0) { placed = (int)members.size(); members.push_back(p); gtail.push_back(p); gsize.push_back(0); }
        else { gnext[gtail[placed]] = p; gtail[placed] = p; }
        gsize[placed]++; grp[p] = placed;
    }
    const int ng = (int)members.size();
    gord.resize(ng);
    for (int g = 0; g < ng; g++) gord[g] = g;
    std::stable_sort(gord.begin(), gord.end(), [&](int a, int b) { return gsize[a] > gsize[b]; });  // :139
    rank.resize(ng);
    for (int r = 0; r < ng; r++) rank[gord[r]] = r;
    for (int a = 0; a < np; a++) level[a] = rank[grp[a]] + 1;
    return ng;
}

typedef std::vector<BP> BPV;   // sorted, unique

static void levels_of(const BPV &pairs, int n, int levellimit, std::vector<int16_t> &out)
{
    static thread_local std::vector<int> lv;
    sq_pair_levels(pairs, lv);
    out.assign(n, 0);
    for (size_t k = 0; k < pairs.size(); k++) {
        if (levellimit >= 0 && lv[k] > levellimit) continue;           // :153-154
        out[pairs[k].first] = (int16_t)lv[k];
        out[pairs[k].second] = (int16_t)-lv[k];
    }
}

// same result as levels_of() on the stems' bps, computed per stem
static void levels_of_stems(const std::vector<HStem> &stems, int n, std::vector<int16_t> &out)
{
    static thread_local std::vector<int> lv;
    sq_stem_levels(stems, lv);
    out.assign(n, 0);
    for (size_t k = 0; k < stems.size(); k++)
        for (int t = 0; t < stems[k].len; t++) {
            out[stems[k].i + t] = (int16_t)lv[k];
            out[stems[k].j - t] = (int16_t)-lv[k];
        }
}

namespace {
// One distinct structure of a sequence.  Its stems stay where the fold left them (the job's final list outlives the
// tail); its sorted base pairs are a slice of the call's flat pair list -- no allocation per structure.
struct Entry {
    const std::vector<HStem> *stems;
    uint32_t cs_off, cs_n;        // canonical stems (maximal stacks, ascending i): what identifies the base-pair set
    uint32_t bp_off, bp_n;        // its sorted pairs, expanded only when something needs them (bp_off == ~0u: not yet)
    double scores[3];
    uint64_t mask;
};
struct Stem3 { int32_t i, j, len; };
struct Span {
    const BP *p; size_t n;
    const BP *begin() const { return p; }
    const BP *end() const { return p + n; }
    size_t size() const { return n; }
};
}  // namespace

static inline uint64_t mix_bp(int i, int j)
{
    uint64_t x = ((uint64_t)(uint32_t)i << 32) | (uint32_t)j;          // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// sorted, unique base pairs of a stem list + an order-independent hash of the set.  The stems of a structure are
// disjoint stacks, so walking them by ascending i already yields the sorted list; anything else takes the sort.
static uint64_t bps_of(const std::vector<HStem> &stems, BPV &out)
{
    static thread_local std::vector<int> order;
    order.resize(stems.size());
    for (size_t k = 0; k < stems.size(); k++) {                        // insertion sort by i (a handful of stems)
        size_t q = k;
        while (q > 0 && stems[order[q - 1]].i > stems[k].i) { order[q] = order[q - 1]; q--; }
        order[q] = (int)k;
    }
    out.clear();
    uint64_t h = 0;
    bool sorted = true;
    for (int idx : order) {
        const HStem &s = stems[idx];
        for (int k = 0; k < s.len; k++) {
            const BP bp(s.i + k, s.j - k);
            if (!out.empty() && !(out.back() < bp)) sorted = false;
            out.push_back(bp);
            h += mix_bp(bp.first, bp.second);
        }
    }
    if (!sorted) {
        std::sort(out.begin(), out.end());
        out.erase(std::unique(out.begin(), out.end()), out.end());
        h = 0;
        for (const BP &bp : out) h += mix_bp(bp.first, bp.second);
    }
    return h;
}

// The base-pair set of a structure as its maximal stacks in ascending order of i -- unique for a set of disjoint stems,
// so two stem lists describe the same structure iff their canonical lists are equal: a dozen triples to hash and
// compare instead of a hundred pairs.  Returns false (-> the pair path) when the stems are not disjoint stacks.
static bool canonical_stems(const std::vector<HStem> &stems, std::vector<Stem3> &out, uint64_t &hash)
{
    static thread_local std::vector<int> order;
    order.resize(stems.size());
    for (size_t k = 0; k < stems.size(); k++) {                        // insertion sort by i (a handful of stems)
        size_t q = k;
        while (q > 0 && stems[order[q - 1]].i > stems[k].i) { order[q] = order[q - 1]; q--; }
        order[q] = (int)k;
    }
    out.clear();
    for (int idx : order) {
        const HStem &s = stems[idx];
        if (s.len <= 0) continue;
        if (!out.empty()) {
            Stem3 &p = out.back();
            if (s.i < p.i + p.len) return false;                       // 5' strands overlap: not disjoint
            if (s.i == p.i + p.len && s.j == p.j - p.len) { p.len += s.len; continue; }   // stacked onto the previous stem
        }
        out.push_back(Stem3{s.i, s.j, s.len});
    }
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (const Stem3 &t : out) h = (h ^ mix_bp(t.i, (t.j << 10) ^ t.len)) * 0xD6E8FEB86659FD93ull;
    hash = h;
    return true;
}

template <class A, class B>
static inline size_t count_common(const A &a, const B &b)
{
    auto i = a.begin(); auto j = b.begin();
    size_t c = 0;
    while (i != a.end() && j != b.end()) {
        if (*i < *j) ++i;
        else if (*j < *i) ++j;
        else { c++; ++i; ++j; }
    }
    return c;
}

template <class A, class B>
static BPV merged(const A &a, const B &b)
{
    BPV out;
    std::set_union(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(out));
    return out;
}

// ScoreStruct (:861-899)
static void score_struct(const uint8_t *codes, const double *reacts, int n, const std::vector<HStem> &stems,
                         double out[3])
{
    auto bpscore = [](int a, int b) -> double {
        const int A = 0, C = 2, G = 6, U = 20;
        if ((a == G && b == U) || (a == U && b == G)) return -0.5;
        if ((a == A && b == U) || (a == U && b == A)) return 1.5;
        if ((a == G && b == C) || (a == C && b == G)) return 4.0;
        return 0.0;
    };
    double thescore = 0;
    static thread_local std::vector<char> paired;
    paired.assign(n, 0);
    for (const HStem &s : stems) {
        double bpsum = 0;
        for (int k = 0; k < s.len; k++) {
            const int v = s.i + k, w = s.j - k;
            bpsum += bpscore(codes[v], codes[w]);
            paired[v] = paired[w] = 1;
        }
        if (bpsum > 0) thescore += pow(bpsum, 1.7);                    // :884
    }
    int sepnum = 0;
    double acc = 0;                                                    // :894-896 (sum() from int 0)
    for (int i = 0; i < n; i++) {
        if (codes[i] == SQ_CODE_SEP1 || codes[i] == SQ_CODE_SEP2) { sepnum++; continue; }
        acc += paired[i] ? reacts[i] : 1 - reacts[i];
    }
    const double reactscore = 1 - acc / (n - sepnum);
    out[0] = py_round3(thescore * reactscore);
    out[1] = py_round3(thescore);
    out[2] = py_round3(reactscore);
}

template <class A>
static void prf(const A &pred, const BPV &known, double m[6])         // :1252-1258
{
    const int tp = (int)count_common(pred, known);
    const int fp = (int)pred.size() - tp, fn = (int)known.size() - tp;
    m[0] = tp; m[1] = fp; m[2] = fn;
    m[3] = (2 * tp + fp + fn) ? py_round3(2.0 * tp / (2 * tp + fp + fn)) : 1;
    m[4] = (tp + fp) ? py_round3((double)tp / (tp + fp)) : 1;
    m[5] = (tp + fn) ? py_round3((double)tp / (tp + fn)) : 1;
}

void sq_tail(const sq_batch *b, int seq, const sq_fold_opts &o,
             const std::vector<const std::vector<std::vector<HStem>> *> &per_job, const std::vector<int32_t> &job_ids,
             const int32_t *ref_pairs, int nref, bool has_ref, SeqResult &res)
{
    const int off = b->seq_off[seq], n = b->seq_off[seq + 1] - off;
    const uint8_t *codes = b->codes.data() + off;
    const double *reacts = sq_host_reacts(b) + off;

#ifdef SQ_TAIL_PROF
    auto nowus = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = nowus();
#endif
    // :1201-1220 dedupe across paramsets; the first producer scores the structure
    static thread_local std::vector<Entry> fins_tl;
    static thread_local std::vector<BP> flat;                           // sorted pairs of the structures that needed them
    static thread_local std::vector<Stem3> cflat, ckey;                 // canonical stems of every distinct structure
    static thread_local std::vector<int> table;                         // open addressing on the hash: entry + 1
    static thread_local std::vector<uint64_t> hashes;
    static thread_local BPV key;
    std::vector<Entry> &fins = fins_tl;
    fins.clear(); flat.clear(); cflat.clear(); hashes.clear();
    size_t total = 0;
    for (size_t k = 0; k < per_job.size(); k++) total += per_job[k]->size();
    size_t tsize = 16;
    while (tsize < 2 * total + 2) tsize <<= 1;
    table.assign(tsize, 0);
    // pairs of an entry, expanded on first use (the ranking needs them for a handful of entries only)
    auto ensure_pairs = [&](Entry &e) {
        if (e.bp_off != ~0u) return;
        e.bp_off = (uint32_t)flat.size();
        for (uint32_t t = 0; t < e.cs_n; t++) {
            const Stem3 &c = cflat[e.cs_off + t];
            for (int k = 0; k < c.len; k++) flat.push_back(BP(c.i + k, c.j - k));
        }
        e.bp_n = (uint32_t)flat.size() - e.bp_off;
    };
    auto bps_of_entry = [&](Entry &e) { ensure_pairs(e); return Span{flat.data() + e.bp_off, e.bp_n}; };
    for (size_t k = 0; k < per_job.size(); k++) {
        for (const auto &stems : *per_job[k]) {
            uint64_t h = 0;
            if (!canonical_stems(stems, ckey, h)) {
                // stems that are not disjoint stacks: the maximal stacks of their sorted unique pairs (the same canonical form)
                bps_of(stems, key);
                ckey.clear();
                for (const BP &bp : key) {
                    if (!ckey.empty() && ckey.back().i + ckey.back().len == bp.first && ckey.back().j - ckey.back().len == bp.second) ckey.back().len++;
                    else ckey.push_back(Stem3{bp.first, bp.second, 1});
                }
                h = 0x9E3779B97F4A7C15ull;
                for (const Stem3 &t : ckey) h = (h ^ mix_bp(t.i, (t.j << 10) ^ t.len)) * 0xD6E8FEB86659FD93ull;
            }
            size_t slot = (size_t)(h * 0x9E3779B97F4A7C15ull >> 11) & (tsize - 1);
            int found = -1;
            for (;; slot = (slot + 1) & (tsize - 1)) {
                const int e = table[slot] - 1;
                if (e < 0) break;
                if (hashes[e] != h) continue;
                const Entry &E = fins[e];
                if (E.cs_n == ckey.size() &&
                    std::equal(ckey.begin(), ckey.end(), cflat.data() + E.cs_off,
                               [](const Stem3 &x, const Stem3 &y) { return x.i == y.i && x.j == y.j && x.len == y.len; })) { found = e; break; }
            }
            if (found < 0) {
                Entry e;
                e.stems = &stems; e.mask = 1ull << k; e.bp_off = ~0u; e.bp_n = 0;
                e.cs_off = (uint32_t)cflat.size(); e.cs_n = (uint32_t)ckey.size();
                cflat.insert(cflat.end(), ckey.begin(), ckey.end());
                score_struct(codes, reacts, n, stems, e.scores);
                table[slot] = (int)fins.size() + 1;
                hashes.push_back(h);
                fins.push_back(e);
            } else {
                fins[found].mask |= 1ull << k;
            }
        }
    }
#ifdef SQ_TAIL_PROF
    const double tp1 = nowus();
#endif
    // RankStructs (:902-955)
    auto keyless = [&](const Entry &x, const Entry &y) {               // true when x sorts before y (descending)
        for (int t = 0; t < 3; t++) {
            const double a = x.scores[o.rankby[t]], c = y.scores[o.rankby[t]];
            if (a != c) return a > c;
        }
        return false;
    };
    std::stable_sort(fins.begin(), fins.end(), keyless);               // :907-909
    if (o.priority_mask)
        std::stable_partition(fins.begin(), fins.end(), [&](const Entry &e) { return (e.mask & o.priority_mask) != 0; });  // :912-913
    if (o.rankbydiff && fins.size() >= 3) {                            // :917-955
        BPV allbps, seenbps;
        for (Entry &e : fins) allbps = merged(allbps, bps_of_entry(e));
        { const Span f0 = bps_of_entry(fins[0]); seenbps.assign(f0.begin(), f0.end()); }
        size_t cur = 1;
        while (seenbps != allbps && cur < fins.size() - 1) {
            std::vector<std::pair<size_t, size_t>> novel;              // (#new bps, original position)
            std::vector<Entry> tailv(fins.begin() + cur, fins.end());
            std::vector<size_t> nov(tailv.size()), idx(tailv.size());
            for (size_t t = 0; t < tailv.size(); t++) { const Span sp = bps_of_entry(tailv[t]); nov[t] = sp.size() - count_common(sp, seenbps); idx[t] = t; }
            std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) {
                if (nov[x] != nov[y]) return nov[x] > nov[y];
                return keyless(tailv[x], tailv[y]);
            });
            for (size_t t = 0; t < tailv.size(); t++) fins[cur + t] = tailv[idx[t]];
            seenbps = merged(seenbps, bps_of_entry(fins[cur]));
            cur++;
        }
        std::stable_sort(fins.begin() + cur, fins.end(), keyless);
    }
    // hardrest: restraint bps whose letters form an allowed pair of the LAST paramset (:1226-1228)
    BPV forced;
    if (o.hardrest && !job_ids.empty()) {
        const sq_paramset &ps = b->psets[b->job_pset[job_ids.back()]];
        for (int k = b->rbp_off[seq]; k < b->rbp_off[seq + 1]; k++) {
            const int v = b->rbps[2 * k], w = b->rbps[2 * k + 1];
            if (ps.inbps[codes[v] * 32 + codes[w]]) forced.push_back(BP(v, w));
        }
        std::sort(forced.begin(), forced.end());
    }
#ifdef SQ_TAIL_PROF
    const double tp2 = nowus();
#endif
    // (sq_result_limit in force: the bracket strings -- three quarters of this function's time on pools of a thousand --
    // are only formed for the structures the getters will show)
    const size_t nshow = b->result_limit > 0 ? std::min<size_t>(fins.size(), (size_t)b->result_limit) : fins.size();
    res.preds.clear();
    res.preds.reserve(nshow);
    for (size_t fk = 0; fk < nshow; fk++) {                            // :1232-1234
        Entry &e = fins[fk];
        res.preds.emplace_back();
        SeqResult::Pred &p = res.preds.back();
        if (forced.empty()) levels_of_stems(*e.stems, n, p.levels);
        else levels_of(merged(bps_of_entry(e), forced), n, -1, p.levels);
        for (int t = 0; t < 3; t++) p.scores[t] = e.scores[t];
        p.pset_mask = e.mask;
    }
#ifdef SQ_TAIL_PROF
    const double tp3 = nowus();
#ifndef SQ_TAIL_PROF_MIN
#define SQ_TAIL_PROF_MIN 100
#endif
    if (fins.size() > SQ_TAIL_PROF_MIN) fprintf(stderr, "[sq_tail] seq %d n=%d: %zu structures: dedupe+score %.0f us, rank %.0f us, levels %.0f us\n", seq, n, fins.size(), tp1 - tp0, tp2 - tp1, tp3 - tp2);
#endif
    BPV cons;                                                          // :845-858,1236
    const size_t top = std::min<size_t>(fins.size(), (size_t)std::max(o.conslim, 0));
    if (top) {
        { const Span f0 = bps_of_entry(fins[0]); cons.assign(f0.begin(), f0.end()); }
        for (size_t k = 1; k < top; k++) {
            BPV nx;
            const Span fk = bps_of_entry(fins[k]);
            std::set_intersection(cons.begin(), cons.end(), fk.begin(), fk.end(), std::back_inserter(nx));
            cons.swap(nx);
        }
    }
    if (!forced.empty()) cons = merged(cons, forced);
    if (top == 1 && forced.empty()) res.cons = res.preds[0].levels;
    else levels_of(cons, n, -1, res.cons);
    res.has_ref = has_ref;
    if (has_ref) {                                                     // :1249-1285
        BPV known;
        for (int k = 0; k < nref; k++) known.push_back(BP(ref_pairs[2 * k], ref_pairs[2 * k + 1]));
        std::sort(known.begin(), known.end());
        known.erase(std::unique(known.begin(), known.end()), known.end());
        {   // ReferenceScores (:958-970): ScoreStruct(seq, PairsToStems(sorted(DBNToPairs(ref))), reacts) -- printed on the
            // record's "reference" line; here because the pairs, letters and reactivities are at hand
            std::vector<HStem> rstems;
            for (size_t k = 0; k < known.size(); k++) {
                if (k && known[k - 1].first + 1 == known[k].first && known[k - 1].second == known[k].second + 1) rstems.back().len++;
                else rstems.push_back(HStem{known[k].first, known[k].second, 1, 0, 0});
            }
            score_struct(codes, reacts, n, rstems, res.ref_scores);
        }
        prf(cons, known, res.cons_metrics);
        double best = -1;
        for (int t = 0; t < 7; t++) res.best_metrics[t] = NAN;
        for (size_t rank = 0; rank < fins.size(); rank++) {
            double m[6];
            if (forced.empty()) prf(bps_of_entry(fins[rank]), known, m);
            else prf(merged(bps_of_entry(fins[rank]), forced), known, m);
            if (m[3] > best) {
                best = m[3];
                for (int t = 0; t < 6; t++) res.best_metrics[t] = m[t];
                res.best_metrics[6] = (double)(rank + 1);
            }
            if ((int)rank + 1 >= o.toplim) break;
        }
    }
}
