// sq_pool_round.h -- launch arguments and LDS layout of sq_pool_round_kernel (sq_pool_round.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sq_device.h"

#ifndef SQ_PR_STAGE
#define SQ_PR_STAGE 128            // runs the scan stages in LDS before they are scored (64 at a time)
#endif
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
    size_t tail = (size_t)22 * surv_cap + 16;
    const size_t ext = (size_t)8 * L.t8 + 64 * 4 + 64 + 16;   // crossing weights (int32), order, group, level, group sizes, ranks
    if (SQ_PR_STAGE * 8 + tail < ext) tail = ext - SQ_PR_STAGE * 8;
    L.total = (size_t)o + tail;
    return L;
}

extern "C" __global__ void sq_pool_round_kernel(SqDevCtx c, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra);
