// sq_results.hip -- result getters and packing of the C ABI (include/squarna_hip.h).
#include "sq_host_int.h"

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

extern "C" int sq_result_view(const sq_batch *b, const void **buf, const int64_t **off, int64_t *nbytes)
{
    if (!b || !buf || !off || !nbytes) { sq_set_error("bad argument"); return -1; }
    if (!(b->packed_ok && packed_whole(b)) || !b->h_rec || !b->h_rec_off) return 1;
    static_assert(sizeof(long long) == sizeof(int64_t), "offsets are 64-bit");
    *buf = b->h_rec; *off = reinterpret_cast<const int64_t *>(b->h_rec_off); *nbytes = (int64_t)b->h_rec_off[b->nseq];
    return 0;
}

extern "C" int sq_result_detach(sq_batch *b, void **buf, int64_t *nbytes)
{
    if (!b || !buf || !nbytes) { sq_set_error("bad argument"); return -1; }
    if (!(b->packed_ok && packed_whole(b)) || !b->h_rec || !b->h_rec_off) return 1;
    *buf = b->h_rec; *nbytes = (int64_t)b->h_rec_off[b->nseq];
    b->h_rec = nullptr; b->h_rec_cap = 0; b->packed_ok = false;       // (the batch has no results any more until it folds again)
    return 0;
}
extern "C" void sq_buffer_release(void *buf) { sq_pinned_put(buf); }

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

