// sq_pool_round.h -- launch arguments and LDS layout of sq_pool_round_kernel (sq_pool_round.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sq_device.h"

#ifndef SQ_PR_STAGE
#define SQ_PR_STAGE 128            // runs the scan stages in LDS before they are scored (64 at a time)
#endif
#define SQ_PR_ROOT_MAXN 1024       // ... with root lists (SqPoolRoundArgs::root)
#define SQ_PR_MAXN 256             // longest sequence the kernel takes (one wave per structure; measured to 1,024: parity clean, no faster than the launched round kernels from ~300 nt on -- a round of long structures is bound by its scan and ScoreStems work, not by launches)

struct SqPoolRoundArgs {
    int32_t lds_n;          // longest sequence of the launch
    int32_t str_cap;        // strands per structure the LDS list holds
    int32_t cell_entries;   // doubles of the cell table
    int32_t surv_cap;       // survivors of :492 kept in LDS (the rest spill to the structure's slice of the candidate arena)
    int32_t bound;          // branch and bound on the finalscore
    int32_t tmax;           // stems per structure slot (SqPoolIO::pt): the level scratch of the extension
    int32_t parity;         // generation of this round's structures
    int32_t lo;             // position in the round's list of the launch's first structure (chunked rounds)
    int32_t ahead;          // the launch covers every slot: blocks beyond the generation's size (SqPoolHdr::S) leave
    int32_t root;           // 1: AnnotateStems as a pass over the job's ROOT list -- the runs of the empty structure with their exact
                            // bpscores, written once per fold by sq_pool_root_kernel -- checked against the structure's own partner
                            // array, instead of the bit-diagonal scan (sequences beyond SQ_PR_MAXN: there the scan and the bpscores of
                            // its thousands of runs are most of a structure's round)
    int32_t root_units;     // 32-byte units of the candidate arena per job's root list (16 bytes per run)
    int64_t root_off;       // first unit of the root lists in the candidate arena (job record sx: root_off + sx * root_units)
    SqKept kept;            // (root mode) the structures' own lists, handed from parent to child
};

struct SqPoolRoundLds {
    int np, fbh;
    int off_P, off_U, off_SU, off_E, off_ci, off_code, off_fg, off_cell, off_str, off_sidx, off_skip, off_stems, off_stage, off_surv;
    int t8;                 // stems the stem arrays hold (tmax rounded up to 8)
    int choose_cap;         // candidates within range the choose phase sorts (its arrays reuse everything before off_stems)
    size_t total;
};
__host__ __device__ inline SqPoolRoundLds sq_pool_round_lds(int lds_n, int str_cap, int cell_entries, int surv_cap, int tmax)
{
    SqPoolRoundLds L;
    L.np = (lds_n + 8) & ~7;
    L.fbh = ((lds_n + 2 + 31) >> 5) + 8;
    int o = 0;
    L.off_P = o; o += 2 * L.np;
    L.off_U = o; o += 2 * L.np;
    L.off_SU = o; o += 2 * L.np;
    L.off_E = o; o += L.np;
    L.off_ci = o; o += L.np;
    L.off_code = o; o += L.np;
    L.off_fg = o; o += 8 * L.fbh;
    o = (o + 15) & ~15;
    L.off_cell = o; o += 8 * cell_entries;
    L.off_str = o; o += str_cap * (int)sizeof(SqStrand);
    L.off_sidx = o; o += str_cap * 2;
    L.off_skip = o; o += str_cap * 2;
    o = (o + 15) & ~15;
    L.t8 = (tmax + 7) & ~7;
    L.off_stems = o; o += 6 * L.t8;                            // the structure's stems: i, j, len (int16)
    o = (o + 15) & ~15;
    L.off_stage = o; o += SQ_PR_STAGE * 8;                     // (the extension's crossing weights and level scratch borrow this and the survivors' room)
    L.off_surv = o;
    L.choose_cap = L.off_stems / 16 < 512 ? L.off_stems / 16 : 512;   // (the stems stay: a final structure logs them)
    size_t tail = (size_t)24 * surv_cap + 16;              // bpscore, finalscore, key, length, place in the structure's kept list
    const size_t ext = (size_t)8 * L.t8 + 64 * 4 + 64 + 16;   // crossing weights (int32), order, group, level, group sizes, ranks
    if (SQ_PR_STAGE * 8 + tail < ext) tail = ext - SQ_PR_STAGE * 8;
    L.total = (size_t)o + tail;
    return L;
}

extern "C" __global__ void sq_pool_round_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra);
extern "C" __global__ void sq_pool_round_root_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra);
extern "C" __global__ void sq_pool_root_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra);
// dynamic LDS of sq_pool_root_kernel: mask codes + class indices, free-position words, cell table, staging buffer
__host__ __device__ inline size_t sq_pool_root_lds(int lds_n, int cell_entries)
{
    const size_t np = ((size_t)lds_n + 8) & ~(size_t)7, fbh = (((size_t)lds_n + 2 + 31) >> 5) + 8;
    return 2 * np + ((8 * fbh + 15) & ~(size_t)15) + 8 * (size_t)cell_entries + 8 * 1088 + 64;
}
