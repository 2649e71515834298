// sq_pool_round.h -- launch arguments and LDS layout of sq_pool_round_kernel (sq_pool_round.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sq_device.h"

#define SQ_PR_STAGE 128            // runs the scan stages in LDS before they are scored (64 at a time)
#define SQ_PR_MAXN 256             // longest sequence the kernel takes (one wave per structure)

struct SqPoolRoundArgs {
    int32_t lds_n;          // longest sequence of the launch
    int32_t str_cap;        // strands per structure the LDS list holds
    int32_t cell_entries;   // doubles of the cell table
    int32_t surv_cap;       // survivors of :492 kept in LDS (the rest spill to the structure's slice of the candidate arena)
    int32_t bound;          // branch and bound on the finalscore
};

struct SqPoolRoundLds {
    int np, fbh;
    int off_P, off_U, off_SU, off_E, off_ci, off_code, off_fg, off_cell, off_str, off_skip, off_stage, off_surv;
    int choose_cap;         // candidates within range the choose phase sorts (its arrays reuse everything before off_surv)
    size_t total;
};
__host__ __device__ inline SqPoolRoundLds sq_pool_round_lds(int lds_n, int str_cap, int cell_entries, int surv_cap)
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
    L.off_skip = o; o += str_cap * 2;
    o = (o + 15) & ~15;
    L.off_stage = o; o += SQ_PR_STAGE * 8;
    L.off_surv = o;
    L.choose_cap = o / 16 < 512 ? o / 16 : 512;
    L.total = (size_t)o + (size_t)22 * surv_cap + 16;
    return L;
}

extern "C" __global__ void sq_pool_round_kernel(SqDevCtx c, const SqStruct *structs, SqScanArgs a, SqPoolIO pio, SqPoolRoundArgs ra);
