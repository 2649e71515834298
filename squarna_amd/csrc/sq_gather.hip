// sq_gather.hip -- alignment step 2's weighting matrices without a host round trip.
//
// The reference hands every sequence of an alignment the same L x L stem matrix and deletes the rows and columns of the
// sequence's gap positions (SQRNdbnseq.py:1031-1034) before bpscorematrix *= shortsmat (:1084-1085).  Built per sequence
// on the host that is N x N doubles each through PCIe (48 sequences x 5,000 columns: 8.5 GB, 4.7 of the 8.9 s the
// three steps took).  The matrix is the result of step 1 and already lives on the device: here every job's slice
//     mat64[a][b] = M[cols[a]][cols[b]]
// is gathered straight from it (one row of the slice per block row, coalesced writes; the reads follow the column map,
// which is monotone, so they run through M's rows front to back).
#include <hip/hip_runtime.h>
#include "sq_device.h"

extern "C" __global__ __launch_bounds__(256) void sq_gather_mul_kernel(SqDevCtx c, const double *M, int L, const int32_t *cols,
                                                                      const int32_t *job_list)
{
    const SqJob jb = c.jobs[job_list[blockIdx.z]];
    const int n = jb.n;
    const int a = blockIdx.y;
    if (a >= n) return;
    const int32_t *cl = cols + jb.pos_off;
    const double *row = M + (size_t)cl[a] * (size_t)L;
    double *dst = c.mat64 + jb.mat64_off + (size_t)a * (size_t)n;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n; b += gridDim.x * 256) dst[b] = row[cl[b]];
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
