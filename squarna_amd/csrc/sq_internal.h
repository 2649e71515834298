// sq_internal.h -- device-visible records shared by the host driver and the kernels.
#pragma once
#include <stdint.h>

#define SQ_SENT_BITS 0x7FC00000u  // quiet NaN in the fp32 scan matrix == "bpboolmatrix cell is 0"
#define SQ_MAXLEVELS 64           // pseudoknot levels tracked in a 64-bit set

// One (sequence, paramset) fold job.
struct SqJob {
    int32_t n;          // gap-free sequence length
    int32_t ld;         // row pitch (floats) of the scan matrix, ld % 32 == 1
    int32_t seq;        // sequence index
    int32_t pset;       // paramset index
    int64_t pos_off;    // offset of the sequence in the per-position arrays
    int64_t mat_off;    // offset (floats) of the scan matrix in the fp32 arena
    int64_t mat64_off;  // offset (doubles) of the dense N x N exact matrix, -1: recompute from O(N) inputs
    int32_t default_reacts;  // set(reacts) == {0.5}  (SQRNdbnseq.py:273)
    int32_t interchainonly;
    int32_t has_ext;    // scan matrix comes from caller matrices (ext_bool/ext_score) or mul_score
    int32_t cand_cap;   // candidate capacity per structure of this job
    float maxabs;       // upper bound of |scoremat cell| (fp32 prefilter margin of the scan)
    int32_t nrb;        // restraint base pairs of the sequence
    int64_t bits_off;   // offset (words) of the diagonal bit matrix: word (w, s) at bits_off + w * bpitch + s
    int32_t bpitch;     // words per word-row (>= 2N + 64, multiple of 64)
    int32_t nw;         // word-rows: ceil(N / 32); bit b of word (w, s) <-> cell (32w + b, s - 32w - b)
    int32_t rb_off;     // into the packed restraint pair list
    int32_t ext_add;    // has_ext == 2: the dense term is ADDED to the score (bpp < 0) instead of multiplied
    int32_t react_levels;  // 1..16: the reactivities take that many distinct values (level index per position in
                           // SqDevCtx::ridx): reactfactors come from a level x level table; 0: computed per cell
    int32_t rf_idx;        // >= 0 (react_levels > 0, reactivities not all 0.5): table rf_idx of SqDevCtx::rftab holds the
                           // sequence's 16 x 16 reactfactors ((1 - (r_a + r_b) / 2) * 2) ** 0.5 evaluated by the HOST's libm
                           // pow, as CPython does (SQRNdbnseq.py:333) -- sqrt differs from it in the last bit now and then
    int32_t mat64_diag;    // the dense fp64 matrix is diagonal-major (sq_m64_index): jobs weighted by the shared stem matrix
    int32_t mulsh;         // 1: every cell is weighted by the alignment's ONE shared L x L matrix, read through the gap map as it is needed
                           // (SqDevCtx::mulM / mulcols; no dense matrix of the job exists: mat64_off == -1).  0 with mat64_diag == 1: the
                           // job's slice of that matrix was materialised (SQ_MUL_GATHER=1, the round-3 form)
    int32_t bits_owner;    // 1: the job's bit matrix is its own (the bit kernels write it); 0: it reads the matrix of an earlier job
                           // of the same sequence whose paramset pairs the same letters (the default configs: one matrix
                           // for a record's five paramsets -- a-1 is 6 % of a crowded step's wave cycles otherwise)
};

// Device image of a paramset (+ host-built pow tables so every pow() is the host libm's).
struct SqPsetDev {
    double w[32 * 32];
    uint8_t inbps[32 * 32];
    double minlen, minbpscore, minfinscore;
    double bracketweight, distcoef, orderpenalty, loopbonus;
    double oftab[SQ_MAXLEVELS + 1];   // (1/(1+k))**orderpenalty   (SQRNdbnseq.py:729)
    int32_t bw_integral;              // bracketweight is an integer -> stemdist index into sdftab
    int32_t sdf_off, sdf_len;         // (1/(1+d))**distcoef table   (SQRNdbnseq.py:726)
    uint32_t lmask;                   // bit a: letter code a has a pair in the paramset (the classes of sq_cellrun.h)
    uint32_t pmask[32];               // pmask[a] bit q: (a, q) is a pair of the paramset, codes 0..28 (SQRNdbnseq.py:300; the bit kernels)
    // stemscore ** 1.7 (SQRNalgos.py:101,122: the Edmonds / Hungarian edge weights) from the host libm, for paramsets whose
    // pair weights are multiples of 2^-q: a stem score is then k 2^-q exactly and SqDevCtx::powtab[pow_off + k] its power
    // (only valid for jobs without reactivity factors or dense matrices).  pow_len == 0: no table (host-built edges)
    int32_t pow_off, pow_len;
    double pow_scale;                 // 2^q
    // Upper bound of a candidate's finalscore from its bpscore alone (the scoring kernel's branch and bound):
    //     finalscore = bpscore * sdf * orderfactor * loopfactor * tetra  (:732)  <=  ((bpscore * ub_of) * ub_lf) * 1.25
    // for bpscore >= 0, with ub_of = max orderfactor, ub_lf = max loopfactor and sdf <= 1 (distcoef >= 0); ub_lf = +inf
    // switches the bound off (paramsets where a factor has no such maximum)
    double ub_of, ub_lf;
    // The cell table of the scoring kernels (sq_cellrun.h) for jobs without reactivity factors: K x (K | 1) doubles, K = the
    // paramset's letter classes (the pairing letters in code order + one class for all others); cell (a, b) = the pair weight of
    // the classes' letters, 0.0 for the class without pairs.  Built by the host once: every block of those kernels used to
    // derive it from w[] anew (6 % of the pool round kernel's vector instructions)
    double celltab[32 * 33];
};

// One strand (half of a selected stem) of a partial structure, sorted by start.
struct SqStrand {
    int16_t start;    // first position of the strand
    int16_t len;
    int16_t pstart;   // partner of `start`; partner(start+t) = pstart - t
    uint8_t level;    // pseudoknot level (1-based) of the stem
    uint8_t left;     // 1: 5' strand (start = i), 0: 3' strand
};

// One partial structure evaluated in a round.
struct SqStruct {
    int32_t job;
    int32_t strand_off;   // into the round's strand array
    int32_t nstrand;
    int32_t slot;         // state / candidate slot
    double subopt;
    int64_t cand_off;     // into the candidate arena (records)
};

// Candidate storage of one structure: a slice of `cand_cap` 32-byte units of the candidate arena, used as
//   [cand_cap x SqKey]   (key, len) of every candidate the scan emits             (8 bytes each)
//   [.. x SqOk]          only the candidates that pass the exact thresholds, appended by the scoring kernel
// so the per-round traffic is 8 bytes per candidate written + read, plus 24 bytes per SURVIVING candidate.
struct SqCand { uint32_t w[8]; };   // the 32-byte unit of the arena (capacity accounting only)
struct SqKey {
    uint32_t key;     // (s << 16) | i_outer, s = i + j: the reference's emission order
    uint32_t len;
};
struct SqOk {
    uint32_t key, len;
    double bps;
    double fin;
};

// Output record of a round (device -> host).
struct SqOut {
    int32_t st;       // structure index in the round
    uint32_t key;
    int32_t len;
    int32_t pad;
    double bps;
    double fin;
};

// Device-chained greedy rounds (width-1 pools, sq_chain_kernel): one record per structure slot.  The stems chosen so
// far live in a per-slot slice [toff, toff + tcap) of the chain's stem arrays; the strands in two buffers of 2 tcap
// entries at 4 toff (the kernel writes the next round's list into the other one).
struct SqChain {
    int32_t toff, tcap;
    int32_t nstems;
    int32_t anycross;     // some pair of the structure's stems crosses: levels follow the full PairsToDBN rule
    double maxstems;      // paramset maxstemnum: the structure is final when it holds that many stems (:1168-1174)
};
struct SqChainStem { int32_t i, j, len, cc; };      // cc: summed length of the stems crossing this one (:121-124)
struct SqStemOut {                                  // == HStem of the host: one chosen stem, written to pinned memory
    int32_t i, j, len, pad;
    double bps, fin;
};

// Device pools (sq_pool.hip): the greedy pool loop of SQRNdbnseq.py:1102-1199 for pools of any width, booked on the device.
struct SqPoolJob {            // one greedy job
    int32_t first, count;     // its structures of the current round: [first, first + count) of the round's list
    int32_t cursize;          // :1108,1117-1118
    int32_t job;              // job index in the batch
    double cursubopt, suboptinc, suboptmax;   // :1069-1071,1119-1120
    double maxstems;          // :1123-1129
    long long evals;          // structures evaluated so far (AnnotateStems calls)
};
#define SQ_POOL_HDR_RING 8
struct SqPoolHdr {
    uint32_t S[2];            // structures of the round, by round parity (the scan kernel writes the next round's)
    uint32_t round;
    uint32_t nfin, nfin_stems;   // entries / stems of the pinned log of final structures
    uint32_t ovf;             // some capacity was exceeded: the host repeats the fold with its own loop
    uint32_t active_jobs;
    uint32_t peak;            // largest generation so far
};
struct SqPoolFin {            // one final structure (pinned): finstemsets order == (round_kind, pos) ascending per job
    int32_t job;
    uint32_t round_kind;      // 2 * round + kind; kind 0: full at the start of that round (:1123-1129), 1: no new stem (:1155-1156)
    int32_t pos;              // position in the round's list
    int32_t nstems;
    uint32_t stem_off;        // into the pinned stem log
    uint32_t pad;
};
struct SqPoolStem {           // a stem of a final structure in the pinned log (positions fit 16 bits: SQ_MAXLEN)
    int16_t i, j, len, pad;
};
struct SqPoolPick {           // a stem chosen for a parent (ChooseStems' output list)
    uint32_t key, len;
    double bps, fin;
};

// Round-level counters.
struct SqCounters {
    uint32_t nout;        // records appended to the out list
    uint32_t cand_ovf;    // some structure exceeded its candidate capacity
    uint32_t out_ovf;     // out list overflowed
    uint32_t level_ovf;   // a level above SQ_MAXLEVELS was seen
};
