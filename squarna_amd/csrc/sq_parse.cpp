// sq_parse.cpp -- the record scan of SQUARNA's default input format (SQUARNA.py:80-203, ParseDefaultInput) for big inputs:
// a file of ten thousand records costs the Python loop ~2 us per record before any folding starts.  Host code only.
//
// The format: a '>' line names a record, the lines behind it are its fields in the order of `inputformat` (q seQuence,
// t reacTivities, r Restraints, f reFerence, x skipped), each stripped of surrounding whitespace; sequence, restraints and
// reference are the first whitespace-delimited token of their line (comments may follow), reactivities the whole line.
// This scan reports where those pieces lie in the text; everything that makes a file unusual -- default lines in front of
// the first record, bytes outside ASCII, carriage returns (universal-newline translation), a record without a sequence
// token -- returns -1 and the caller runs the general Python form, which then also raises what the reference raises.
#include <cstdint>
#include <cstring>
#include <vector>
#include "sq_host.h"

// str.strip() / str.split() whitespace among the ASCII code points (str.isspace): \t \n \v \f \r, 0x1c-0x1f, space
static inline bool sq_py_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

extern "C" SQ_API int64_t sq_parse_default(const char *text, int64_t len, int32_t nfields, int32_t q_ind, int32_t t_ind, int32_t r_ind,
                                           int32_t f_ind, int64_t *out, int64_t cap)
{
    if (!text || len < 0 || nfields <= 0 || q_ind < 0 || q_ind >= nfields || !out) return -1;
    for (int64_t i = 0; i < len; i++) {
        const unsigned char c = (unsigned char)text[i];
        if (c >= 0x80 || c == '\r' || c == 0) return -1;
    }
    int64_t nrec = 0, pos = 0;
    int64_t *cur = nullptr;             // the record whose field lines are being read
    int field = 0;                      // index of the next field line of `cur`
    while (pos < len) {
        const char *nl = (const char *)memchr(text + pos, '\n', (size_t)(len - pos));
        const int64_t end = nl ? (int64_t)(nl - text) : len;       // the line is text[pos, end)
        int64_t a = pos, b = end;                                  // stripped
        while (a < b && sq_py_space((unsigned char)text[a])) a++;
        while (b > a && sq_py_space((unsigned char)text[b - 1])) b--;
        if (text[pos] == '>' && end > pos) {
            if (nrec >= cap) return -1;
            cur = out + 10 * nrec++;
            cur[0] = a; cur[1] = b - a;
            for (int k = 2; k < 10; k += 2) { cur[k] = 0; cur[k + 1] = -1; }
            field = 0;
        } else if (!cur) {
            if (b > a) return -1;                                  // a default line in front of the first record
        } else {
            if (field < nfields) {
                // first token of the stripped line (what .split()[0] gives), or the whole stripped line for the reactivities
                int64_t te = a;
                while (te < b && !sq_py_space((unsigned char)text[te])) te++;
                if (field == q_ind) { if (te == a) return -1; cur[2] = a; cur[3] = te - a; }
                if (field == t_ind && t_ind > 0) { cur[4] = a; cur[5] = b - a; }
                if (field == r_ind && r_ind > 0 && b > a) { cur[6] = a; cur[7] = te - a; }
                if (field == f_ind && f_ind > 0 && b > a) { cur[8] = a; cur[9] = te - a; }
            }
            field++;
        }
        pos = end + 1;
    }
    // every record needs its sequence line (a record cut short: the general form raises)
    for (int64_t k = 0; k < nrec; k++) if (out[10 * k + 3] < 0) return -1;
    return nrec;
}


// MatrixToDBNs' first structure (SQRNdbnali.py:121-192 as its caller uses it, :242): the cells in decreasing order of value (ties:
// flat index ascending) are taken one by one; a cell of span >= minspan whose two columns are both still free joins the structure.
// flat[k] = v * N + w of the k-th cell in that order.  Writes the pairs (v, w) in the order taken; returns their number (or -1).
// Host code: the sequential part of alignment step 1 (hundreds of thousands of cells at 5,000 columns: a Python loop until round 6).
extern "C" SQ_API int64_t sq_align_first_fit(const int64_t *flat, int64_t n, int32_t N, int32_t minspan, int32_t *pairs, int64_t cap)
{
    if (!flat || n < 0 || N <= 0 || !pairs || cap < 0) return -1;
    std::vector<uint8_t> taken((size_t)N, 0);
    int64_t cnt = 0;
    for (int64_t k = 0; k < n; k++) {
        const int64_t v = flat[k] / N, w = flat[k] - v * N;
        if (v < 0 || v >= N || w - v < minspan) continue;                 // :147
        if (taken[(size_t)v] || taken[(size_t)w]) continue;
        taken[(size_t)v] = taken[(size_t)w] = 1;
        if (cnt < cap) { pairs[2 * cnt] = (int32_t)v; pairs[2 * cnt + 1] = (int32_t)w; }
        cnt++;
    }
    return cnt;
}
