// sq_gather.hip -- alignment step 2's weights: the shared stem matrix in the layout the kernels read, and (fallback) per-job slices.
//
// The reference hands every sequence of an alignment the same L x L stem matrix and deletes the rows and columns of the
// sequence's gap positions (SQRNdbnseq.py:1031-1034) before bpscorematrix *= shortsmat (:1084-1085).  The matrix is the result
// of step 1 and already lives on the device.  Since round 6 no per-sequence slice exists any more: a kernel that needs cell
// (a, b) of a sequence reads M[cols[a]][cols[b]] where it needs it (sq_mulsh_weight, sq_cells.h) -- the slices were 8 N^2
// bytes per row (90 GB for 512 rows of 4,700 nt: five sub-batches, a fifth of the chip's blocks each), written once and read
// once.  What this file keeps:
//   * sq_mul_diag_kernel: the caller's row-major matrix -> the batch's DIAGONAL-major copy (sq_diag_index): the cells of a
//     stem (a + k, b - k) are neighbours in it wherever the sequence has no gap inside the stem;
//   * sq_gather_mul_kernel: a job's slice  mat64[a][b] = score(a, b) x M[cols[a]][cols[b]]  (0 where bpboolmatrix is 0) for the
//     paths that still want one -- RunAlgo's host-driven filters on weighted jobs, and SQ_MUL_GATHER=1 (the round-3 form, kept
//     as the parity check of the direct reads).
#include <hip/hip_runtime.h>
#include "sq_device.h"
#include "sq_cells.h"

extern "C" __global__ __launch_bounds__(256) void sq_mul_diag_kernel(const double *M, int L, double *dst)
{
    const int v = blockIdx.y;
    const double *row = M + (size_t)v * (size_t)L;
    for (int w = blockIdx.x * 256 + threadIdx.x; w < L; w += gridDim.x * 256) dst[sq_diag_index(L, v, w)] = row[w];
}

extern "C" __global__ __launch_bounds__(256) void sq_gather_mul_kernel(SqDevCtx c, const int32_t *job_list, double *dst_one)
{
    const SqJob jb = c.jobs[job_list[blockIdx.z]];
    const int n = jb.n;
    const int a = blockIdx.y;
    if (a >= n) return;
    const SqPsetDev *ps = c.psets + jb.pset;
    double *dst = dst_one ? dst_one : c.mat64 + jb.mat64_off;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < n; b += gridDim.x * 256) {
        double v = 0.0;                                              // :1084-1085 on the cells whose bool is 1; the others hold 0
        if (b > a && sq_cell_bool(c, jb, ps, a, b)) v = sq_cell_score(c, jb, ps, a, b) * sq_mulsh_weight(c, jb, a, b);
        dst[sq_m64_index(jb, a, b)] = v;
    }
}

void sq_launch_mul_diag(const double *M, int L, double *dst, hipStream_t st)
{
    hipLaunchKernelGGL(sq_mul_diag_kernel, dim3((L + 1023) / 1024, L), dim3(256), 0, st, M, L, dst);
}

// dst_one != nullptr: ONE job (njl == 1), its slice written there instead of the job's place in the dense arena
void sq_launch_gather_mul(const SqDevCtx &c, const int32_t *job_list, int njl, int maxn, hipStream_t st, double *dst_one)
{
    // grid.z is limited to 65,535 jobs per launch
    for (int j0 = 0; j0 < njl; j0 += 65535) {
        const int nz = njl - j0 < 65535 ? njl - j0 : 65535;
        const int gx = maxn >= 2048 ? 4 : 1;
        hipLaunchKernelGGL(sq_gather_mul_kernel, dim3(gx, maxn, nz), dim3(256), 0, st, c, job_list + j0, dst_one);
    }
}
