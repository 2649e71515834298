// sq_text.cpp -- host-side text helpers of the drop-in layer, in C++ because they are per-record work that Python pays
// microseconds for and a batch has thousands of records:
//   sq_dbn_pairs     DBNToPairs (SQRNdbnseq.py:172-207) for many dot-bracket lines at once
//   sq_write_blocks  the output block of RunSQRNdbnseq (SQRNdbnseq.py:1301-1406) for every record of a folded batch,
//                    straight from the packed results (sq_result_pack layout) and the ASCII rows (sq_result_dbn_all)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sq_host.h"

// ---- DBNToPairs ----------------------------------------------------------------------------------------------------
// Brackets: ( ) [ ] { } < > and A..Z / a..z (the ASCII part of the reference's alphabet, :108-112) + the Cyrillic letters of
// the levels beyond 30 recoded by the caller to single bytes (squarna_amd/dbn.py: bracket_bytes).  Unmatched closers are
// ignored; the pairs of a line come out sorted.
static inline int bracket_type(unsigned char ch, bool &open)
{
    switch (ch) {
        case '(': open = true; return 0;  case ')': open = false; return 0;
        case '[': open = true; return 1;  case ']': open = false; return 1;
        case '{': open = true; return 2;  case '}': open = false; return 2;
        case '<': open = true; return 3;  case '>': open = false; return 3;
        default: break;
    }
    if (ch >= 'A' && ch <= 'Z') { open = true; return 4 + (ch - 'A'); }
    if (ch >= 'a' && ch <= 'z') { open = false; return 4 + (ch - 'a'); }
    // the alphabet's letters beyond ASCII (19 Cyrillic pairs, levels 31-49) as the caller recodes them: 0x80 + k opens, 0xA0 + k closes
    if (ch >= 0x80 && ch < 0x80 + 19) { open = true; return 30 + (ch - 0x80); }
    if (ch >= 0xA0 && ch < 0xA0 + 19) { open = false; return 30 + (ch - 0xA0); }
    return -1;
}

extern "C" int sq_dbn_pairs(const char *text, const int64_t *off, int32_t nrec, int32_t *pairs, int64_t pair_cap, int64_t *pair_off)
{
    if (!text || !off || !pairs || !pair_off || nrec < 0) { sq_set_error("bad argument"); return -1; }
    std::vector<std::vector<int32_t>> stacks(49);
    std::vector<std::pair<int32_t, int32_t>> cur;
    int64_t np = 0;
    for (int32_t r = 0; r < nrec; r++) {
        pair_off[r] = np;
        for (auto &s : stacks) s.clear();
        cur.clear();
        const char *line = text + off[r];
        const int64_t n = off[r + 1] - off[r];
        for (int64_t i = 0; i < n; i++) {
            bool open = false;
            const int t = bracket_type((unsigned char)line[i], open);
            if (t < 0) continue;
            if (open) stacks[t].push_back((int32_t)i);
            else if (!stacks[t].empty()) { cur.emplace_back(stacks[t].back(), (int32_t)i); stacks[t].pop_back(); }
        }
        std::sort(cur.begin(), cur.end());
        if (np + (int64_t)cur.size() > pair_cap) { sq_set_error("pair buffer too small"); return -3; }
        for (const auto &p : cur) { pairs[2 * np] = p.first; pairs[2 * np + 1] = p.second; np++; }
    }
    pair_off[nrec] = np;
    return 0;
}

// ---- Python's str(float) ---------------------------------------------------------------------------------------------
// repr of a double: the shortest digit string that round-trips, fixed notation for decimal exponents in [-4, 16), else
// scientific; "x.0" for integral values (float_repr_style 'short').
static void py_float_str(double x, std::string &out)
{
    if (x != x) { out += "nan"; return; }
    if (std::isinf(x)) { out += x < 0 ? "-inf" : "inf"; return; }
    if (x == 0) { out += std::signbit(x) ? "-0.0" : "0.0"; return; }
    {
        // the scores and metrics are round(., 3) values: x is the double nearest to k / 1000, and its shortest repr is k / 1000
        // written out (no shorter decimal can hit the same double below 1e12: neighbours are >= 1e-3 apart, doubles 2^-13)
        const double ax = std::fabs(x);
        if (ax < 1e12) {
            const long long k = std::llround(ax * 1000.0);
            if ((double)k / 1000.0 == ax) {
                char t[40];
                int len = snprintf(t, sizeof t, "%s%lld.%03lld", x < 0 ? "-" : "", k / 1000, k % 1000);
                while (t[len - 1] == '0' && t[len - 2] != '.') len--;
                out.append(t, (size_t)len);
                return;
            }
        }
    }
    char buf[40];
    int prec = 1;
    for (; prec <= 17; prec++) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, x);
        if (strtod(buf, nullptr) == x) break;
    }
    // buf = [-]d[.ddd]e[+-]XX
    const char *p = buf;
    if (*p == '-') { out += '-'; p++; }
    std::string digits;
    for (; *p && *p != 'e'; p++) if (*p != '.') digits += *p;
    const int exp10 = atoi(p + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int nd = (int)digits.size();
    if (exp10 >= -4 && exp10 < 16) {
        if (exp10 < 0) { out += "0."; out.append((size_t)(-exp10 - 1), '0'); out += digits; }
        else if (nd <= exp10 + 1) { out += digits; out.append((size_t)(exp10 + 1 - nd), '0'); out += ".0"; }
        else { out.append(digits, 0, (size_t)exp10 + 1); out += '.'; out.append(digits, (size_t)exp10 + 1, std::string::npos); }
    } else {
        out += digits[0];
        if (nd > 1) { out += '.'; out.append(digits, 1, std::string::npos); }
        char e[16];
        snprintf(e, sizeof e, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10));
        out += e;
    }
}

// "TP={},FP={},FN={},FS={},PR={},RC={}" (:1369,1404): counts as ints; a ratio whose denominator is empty is the int 1
static void metrics_str(const double *m, std::string &out)
{
    const long long tp = (long long)m[0], fp = (long long)m[1], fn = (long long)m[2];
    char b[96];
    snprintf(b, sizeof b, "TP=%lld,FP=%lld,FN=%lld,FS=", tp, fp, fn);
    out += b;
    if (2 * tp + fp + fn) py_float_str(m[3], out); else out += '1';
    out += ",PR=";
    if (tp + fp) py_float_str(m[4], out); else out += '1';
    out += ",RC=";
    if (tp + fn) py_float_str(m[5], out); else out += '1';
}

namespace {
struct Lines {                                   // the '\n'-joined lines of one field, one per record (NULL: no record has one)
    const char *p = nullptr; std::vector<int64_t> beg, end;
    void split(const char *text, int nrec)
    {
        p = text; beg.assign((size_t)nrec, 0); end.assign((size_t)nrec, 0);
        if (!text) return;
        int64_t q = 0;
        for (int r = 0; r < nrec; r++) {
            beg[r] = q;
            while (text[q] && text[q] != '\n') q++;
            end[r] = q;
            if (text[q] == '\n') q++;
        }
    }
    const char *begin(int r) const { return p + beg[(size_t)r]; }
    int64_t len(int r) const { return p ? end[(size_t)r] - beg[(size_t)r] : 0; }
};
inline bool is_gap(char c) { return c == '-' || c == '.' || c == '~'; }
inline bool is_sep(char c) { return c == ';' || c == '&'; }
}  // namespace

namespace {
// appends into a caller buffer; past its end only the length is counted (the caller retries with the size it reports)
struct Sink {
    char *p; int64_t cap, len = 0;
    Sink(char *p_, int64_t cap_) : p(p_), cap(p_ ? cap_ : 0) {}
    inline void put(char c) { if (len < cap) p[len] = c; len++; }
    inline void put(const char *s, int64_t n) { if (len + n <= cap) memcpy(p + len, s, (size_t)n); len += n; }
    inline void put(const char *s) { put(s, (int64_t)strlen(s)); }
    inline void put(const std::string &s) { put(s.data(), (int64_t)s.size()); }
    inline void fill(char c, int64_t n) { if (len + n <= cap) memset(p + len, c, (size_t)n); len += n; }
};
}  // namespace

extern "C" int64_t sq_write_blocks(const sq_batch *b, const sq_block_desc *d, char *buf, int64_t cap, int64_t *off, uint8_t *skipped)
{
    if (!b || !d || !off || !skipped || d->nrec != b->nseq || !d->names || !d->seqs) { sq_set_error("bad argument"); return -1; }
    if (!b->packed_ok) { sq_set_error("the batch's results are not in packed form"); return -2; }
    const int nrec = d->nrec;
    Lines names, seqs, reacts, restr, refs;
    names.split(d->names, nrec); seqs.split(d->seqs, nrec); reacts.split(d->reacts, nrec);
    restr.split(d->restr, nrec); refs.split(d->refs, nrec);
    std::vector<std::vector<std::string>> psn((size_t)std::max(d->nsets, 0));
    for (int k = 0; k < d->nsets; k++) {
        const char *t = d->psnames[k];
        std::string cur;
        for (; *t; t++) { if (*t == '\n') { psn[k].push_back(cur); cur.clear(); } else cur += *t; }
        psn[k].push_back(cur);
    }
    char consname[48];
    snprintf(consname, sizeof consname, "top-%d_consensus", d->conslim);
    Sink out(buf, cap);
    std::string num;
    for (int r = 0; r < nrec; r++) {
        off[r] = out.len;
        skipped[r] = 0;
        if (b->h_deep[r]) { skipped[r] = 1; continue; }            // levels beyond the ASCII brackets: the caller's path
        const char *rec = b->h_rec + b->h_rec_off[r];
        const int64_t *hdr = (const int64_t *)rec;
        const int64_t ns = hdr[0], n = hdr[1], has_ref = hdr[2];
        const double *met = (const double *)(rec + 32);
        const double *scores = (const double *)(rec + 160);
        const uint64_t *masks = (const uint64_t *)(rec + 160 + 24 * ns);
        const char *txt = b->h_txt + b->h_txt_off[r];
        const char *seq = seqs.begin(r);
        const int64_t L = seqs.len(r);
        bool plain = L == n;                                       // no gap columns, no separators: the rows are copied as they are
        for (int64_t i = 0; i < L && plain; i++) plain = !is_gap(seq[i]) && !is_sep(seq[i]);
        // a row of the packed text (gap-free coordinates) in the columns of the input sequence: gap columns are dots, the
        // separators are put back (:1239-1246)
        auto put_row = [&](const char *src) {
            if (plain) { out.put(src, n); return; }
            int64_t q = 0;
            for (int64_t i = 0; i < L; i++) {
                const char c = seq[i];
                if (is_gap(c)) { out.put('.'); continue; }
                const char v = q < n ? src[q] : '.';
                q++;
                out.put(is_sep(c) ? c : v);
            }
        };
        auto put_seps = [&](const char *line, int64_t len) {      // a restraints / reference line with the separators of the sequence
            if (plain) { out.put(line, len); return; }
            for (int64_t i = 0; i < len; i++) out.put((i < L && is_sep(seq[i])) ? seq[i] : line[i]);
        };
        auto put_num = [&](double x) { num.clear(); py_float_str(x, num); out.put(num); };
        auto put_metrics = [&](const double *m) { num.clear(); metrics_str(m, num); out.put(num); };
        out.put(names.begin(r), names.len(r)); out.put('\n');
        out.put(seq, L); out.put('\n');
        if (reacts.len(r)) { out.put(reacts.begin(r), reacts.len(r)); out.put("\treactivities\n"); }
        if (restr.len(r)) { put_seps(restr.begin(r), restr.len(r)); out.put("\trestraints\n"); }
        const bool reference = refs.len(r) > 0;
        if (reference) {
            put_seps(refs.begin(r), refs.len(r));
            out.put("\treference\t");
            put_num(met[13]); out.put('\t');
            if (met[14] == 0) out.put('0'); else put_num(met[14]);      // ScoreStruct keeps the int 0 (:871)
            out.put('\t');
            put_num(met[15]); out.put('\n');
        }
        out.fill('_', L); out.put('\n');
        put_row(txt);                                               // consensus = row 0
        out.put('\t'); out.put(consname);
        if (reference) { out.put('\t'); if (has_ref) put_metrics(met); }
        out.put('\n');
        out.fill('=', L); out.put('\n');
        const int nsetr = d->nameset ? d->nameset[r] : 0;
        const std::vector<std::string> *nm = (nsetr >= 0 && nsetr < (int)psn.size()) ? &psn[nsetr] : nullptr;
        const int64_t nshow = std::min<int64_t>(ns, std::max(d->outplim, 0));
        for (int64_t k = 0; k < nshow; k++) {
            put_row(txt + (k + 1) * n);
            char t[32];
            snprintf(t, sizeof t, "\t#%lld\t", (long long)(k + 1));
            out.put(t);
            put_num(scores[3 * k]); out.put('\t');
            if (scores[3 * k + 1] == 0) out.put('0'); else put_num(scores[3 * k + 1]);   // the int 0 of an empty structure (:871)
            out.put('\t');
            put_num(scores[3 * k + 2]); out.put('\t');
            bool first = true;
            for (uint64_t mk = masks[k]; mk; mk &= mk - 1) {
                const int q = __builtin_ctzll(mk);
                if (!first) out.put(',');
                first = false;
                if (nm && q < (int)nm->size()) out.put((*nm)[q]);
            }
            if (reference && has_ref && (double)(k + 1) == met[12]) {
                out.put('\t'); put_metrics(met + 6);
                snprintf(t, sizeof t, ",RK=%lld", (long long)met[12]);
                out.put(t);
            }
            out.put('\n');
        }
    }
    off[nrec] = out.len;
    if (out.len > out.cap) return -out.len - 16;
    return out.len;
}
