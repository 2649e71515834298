// sq_gather.hip -- alignment step 2's weighting matrices without a host round trip.
//
// The reference hands every sequence of an alignment the same L x L stem matrix and deletes the rows and columns of the
// sequence's gap positions (SQRNdbnseq.py:1031-1034) before bpscorematrix *= shortsmat (:1084-1085).  Built per sequence
// on the host that is N x N doubles each through PCIe (48 sequences x 5,000 columns: 8.5 GB, 4.7 of the 8.9 s the
// three steps took).  The matrix is the result of step 1 and already lives on the device: here every job's slice
//     mat64[a][b] = M[cols[a]][cols[b]]
// is gathered straight from it (one row of the slice per block row; the reads follow the column map, which is monotone,
// so they run through M's rows front to back).  Round 3: the kernel forms the PRODUCT bpscorematrix * shortsmat right away
// (0 where bpboolmatrix is 0; the fill kernel's pass over these jobs is gone) and stores it DIAGONAL-major (sq_m64_index):
// the scoring kernel reads the cells of a stem, which lie on one anti-diagonal, from consecutive addresses.
#include <hip/hip_runtime.h>
#include "sq_device.h"
#include "sq_cells.h"

extern "C" __global__ __launch_bounds__(256) void sq_gather_mul_kernel(SqDevCtx c, const double *M, int L, const int32_t *cols,
                                                                      const int32_t *job_list)
{
    const SqJob jb = c.jobs[job_list[blockIdx.z]];
    const int n = jb.n;
    const int a = blockIdx.y;
    if (a >= n) return;
    const int32_t *cl = cols + jb.pos_off;
    const double *row = M + (size_t)cl[a] * (size_t)L;
    const SqPsetDev *ps = c.psets + jb.pset;
    double *dst = c.mat64 + jb.mat64_off;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n; b += gridDim.x * 256) {
        double v = 0.0;                                              // :1084-1085 on the cells whose bool is 1; the others hold 0
        if (b > a && sq_cell_bool(c, jb, ps, a, b)) v = sq_cell_score(c, jb, ps, a, b) * row[cl[b]];
        dst[sq_m64_index(jb, a, b)] = v;
    }
}

void sq_launch_gather_mul(const SqDevCtx &c, const double *M, int L, const int32_t *cols, const int32_t *job_list, int njl, int maxn,
                          hipStream_t st)
{
    // grid.z is limited to 65,535 jobs per launch
    for (int j0 = 0; j0 < njl; j0 += 65535) {
        const int nz = njl - j0 < 65535 ? njl - j0 : 65535;
        const int gx = maxn >= 2048 ? 4 : 1;
        hipLaunchKernelGGL(sq_gather_mul_kernel, dim3(gx, maxn, nz), dim3(256), 0, st, c, M, L, cols, job_list + j0);
    }
}
