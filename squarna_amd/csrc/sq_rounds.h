// sq_rounds.h -- the persistent round kernel of width-1 pools (sq_rounds.hip): records, launch arguments, LDS layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sq_device.h"

// One maximal run of pairable cells on an anti-diagonal of a structure's masked matrix (a candidate stem of
// AnnotateStems, SQRNdbnseq.py:405-418) with its exact bpscore.  A structure's list lives in its slice of the candidate
// arena (two buffers of cand_cap records: the slice's cand_cap 32-byte units).
struct SqRun {
    uint32_t key;     // (s << 16) | i_outer, s = i + j: the reference's emission order
    uint32_t len;
    double bps;       // sum of the cells outer -> inner from int 0 (:416); NaN: not computed yet (the first round's scan)
};

// The list of the rounds after the first, as two arrays: the run itself (SqRunA, lower half of the structure's slice of the
// arena) and what the rounds before found out about it (SqRunB, upper half).  A run's bound (sq_run_upper) only reads the
// prefix counts within six positions of its four ends, its finalscore (ScoreStems) the strands inside its span and within six
// positions outside it: a stem chosen elsewhere leaves both as they are, so they are kept from round to round and dropped
// when a strand of the new stem comes near (SQ_RX_UB / SQ_RX_FIN in SqRunA::lf say which is valid).
struct SqRunA {
    uint32_t key;     // as SqRun
    uint32_t lf;      // length (low 16 bits; 0: dead) | SQ_RX_* flags
    double bps;
};
struct SqRunB {
    double ub;        // sq_run_upper of the run under the structure as it is (SQ_RX_UB)
    double fin;       // its finalscore under the structure as it is (SQ_RX_FIN), before :751's threshold
};
#define SQ_RX_LEN 0xFFFFu
#define SQ_RX_UB 0x10000u
#define SQ_RX_FIN 0x20000u
#define SQ_RX_LVL 0x40000u         // the kept finalscore counted bracket strands: it reads the levels, a round that renumbers them voids it
#define SQ_RX_FB 0x80000u          // SqRunB::fin holds an UPPER BOUND of the finalscore (ScoreStems' walk ended early at the order factor's
                                   // bound), valid as long as a finalscore would be; never together with SQ_RX_FIN

#define SQ_ROUNDS_SDF_LDS 128      // entries of the distance-factor table kept in LDS
#define SQ_RQ_CAP 192              // entries of each of a wave's three work queues (cut / bound / score): a queue is served when it holds 64
                                   // (a step of the stream adds up to 128)
#define SQ_RQ_WORDS (3 * SQ_RQ_CAP)
#define SQ_ROUNDS_STAGE 128        // runs a wave of the first round's scan stages in LDS before it appends them
#define SQ_ROUNDS_MAXN 8192        // longest sequence whose per-position arrays the kernel keeps in LDS (9 bytes each)
#define SQ_ROUNDS_THREADS 1024     // widest block (few structures: a structure's rounds are a latency chain that more waves shorten)

struct SqRoundsArgs {
    int32_t lds_n;          // longest sequence of the launch
    int32_t str_cap;        // strands per structure the LDS lists hold (2 x the most stems a structure of the launch can have + 2)
    int32_t tmax;           // most stems per structure (level scratch, sq_extend.h)
    int32_t cell_entries;   // doubles of the cell table (largest (classes x reactivity levels)^2 of the batch, padded)
    int32_t bound;          // branch and bound on the finalscore (0: every survivor of :492 is scored)
    int32_t ctx_min;        // strands from which a non-crossing structure's sweep is answered from the context tables (0: never)
    int32_t fly;            // > 0: the first round's scan forms the words of the bit matrix itself, from letter masks in LDS (the fold never
                            // wrote the matrix: every job is scanned once); the value is the batch's letter count (room of the masks)
    int32_t su;             // some sequence of the launch holds a chain separator: the blocks keep the separators' prefix counts (2 bytes per position)
    int32_t wave_min;       // strands from which the last walks of a score step are taken by the whole wave (sq_walk_wave)
    int32_t wave_lanes;     // ... when at most so many lanes are still walking
    int32_t no_early;       // ScoreStems' walk never ends early at the order factor's bound (SQ_NO_EARLY_WALK: tests fold both ways)
    int32_t ties;           // the structures belong to pools that MAY branch (poollim > 1, range factor 1.0): a round in which a second
                            // run reaches the best finalscore ends the structure unfinished (h_fin bit 62) -- the device pools redo its job
};

// dynamic LDS of a block: per-position arrays, free-position words of the first round's scan, cell table, the strand
// list + stem indices, skip pointers, the stems with their crossing weights, and one region shared by the phases that
// never overlap (scan staging / bucket counters / the waves' work queues of the list pass)
struct SqRoundsLds {
    int np, fbh;
    int off_P, off_U, off_SU, off_E, off_ci, off_code, off_fg, off_cell, off_str, off_sidx, off_skip, off_stems, off_tab, off_lvl, off_union;
    int t8;                 // stems the stem arrays hold (tmax rounded up to 8)
    int surv_cap;           // (unused)
    size_t total;
};
__host__ __device__ inline SqRoundsLds sq_rounds_lds(int lds_n, int str_cap, int tmax, int cell_entries, int threads, int su = 1)
{
    SqRoundsLds L;
    L.np = (lds_n + 8) & ~7;                                   // arrays of n + 1 entries
    L.fbh = ((lds_n + 2 + 31) >> 5) + 8;                       // words of either free-position array (as SqState::fbstride / 2)
    int o = 0;
    L.off_P = o; o += 2 * L.np;
    L.off_U = o; o += 2 * L.np;
    L.off_SU = o; o += su ? 2 * L.np : 0;
    L.off_E = o; o += L.np;
    L.off_ci = o; o += L.np;
    L.off_code = o; o += L.np;
    L.off_fg = o; o += 8 * L.fbh;
    o = (o + 15) & ~15;
    L.off_cell = o; o += 8 * cell_entries;
    o = (o + 15) & ~15;
    L.off_str = o; o += str_cap * (int)sizeof(SqStrand);       // (one list: the two strands of a new stem are inserted in place)
    L.off_sidx = o; o += str_cap * 2;
    L.off_skip = o; o += str_cap * 2;
    o = (o + 15) & ~15;
    L.t8 = (tmax + 7) & ~7;
    L.off_stems = o; o += 11 * L.t8 + 64 * 4;                  // the structure's stems: crossing weight (int32), i, j, len (int16), level group (uint8); the groups' sizes
    o = (o + 15) & ~15;
    L.off_tab = o; o += 8 * (SQ_ROUNDS_SDF_LDS + 2 * (SQ_MAXLEVELS + 2));   // the head of the distance-factor table and the order factors (ScoreStems reads them at the end of a chain of dependent loads)
    L.off_lvl = o; o += 4 * L.t8 + 64 + 16;                    // level scratch of the extension (order, level, rank): its own region -- the first
    o = (o + 15) & ~15;                                        // wave extends the structure while the others are in the next round's pass
    L.off_union = o;
    L.surv_cap = 0;
    size_t u = (size_t)(threads / 64) * SQ_RQ_WORDS * 4;       // the waves' work queues of the list pass (list indices)
    const size_t stage = (size_t)(threads / 64) * (SQ_ROUNDS_STAGE * 8 + 16);
    if (u < 2048) u = 2048;                                    // (the first round's bucket counters: 2 x 256 words)
    if (stage > u) u = stage;
    L.total = (size_t)o + ((u + 15) & ~(size_t)15);
    return L;
}

extern "C" __global__ void sq_rounds_kernel(SqDevCtx c, SqStruct *structs, SqScanArgs a, SqChainIO cio, SqRoundsArgs ra);
