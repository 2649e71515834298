// sq_match.h -- job records of the on-device matching step (a-8, a-9, Nussinov).
#pragma once
#include <stddef.h>
#include <stdint.h>

struct SqMatchEdge {      // one cell (v, w), v < w, of a stem with its weight
    int32_t v, w;         // LSAP / Nussinov: sequence positions; blossom: vertex ids in graph order
    double weight;        // H, E: stemscore ** 1.7 (host libm);  N: stemscore
};

struct SqMatchJob {
    int32_t n;            // LSAP / Nussinov: sequence length; blossom: number of graph vertices
    int32_t nedges;
    int64_t edge_off;     // into the edge array
    int64_t scratch_off;  // bytes into the scratch arena
    int64_t out_off;      // ints into the output array (pairs: 2 ints each for Nussinov)
    int64_t pos_off;      // sequence offset into the per-position arrays (Nussinov separators)
    // blossom: the graph's slice of its block's dynamic LDS and the next graph of the same block (sq_mwm_plan)
    int32_t lds_off, lds_bytes;
    int32_t next, pad;
};

// dynamic LDS of one Hungarian job: the row / column vectors (42 bytes per position) and the cost matrix in sparse form --
// the matrix is zero except for the 2m stem cells: per row a bit mask of its non-zero columns with per-word prefix counts,
// the edge ids of a row's cells in column order (16 bits each) and the m edge weights.  150 nt / 1,320 edges: 27 KB (the
// dense form -- 16-bit ids for all n^2 cells -- took 62 KB, and LDS a job holds for its milliseconds is LDS the other
// kernels on the chip do not get)
#ifdef __HIPCC__
#define SQ_HD __host__ __device__
#else
#define SQ_HD
#endif
SQ_HD static inline size_t sq_lsap_vec_bytes(int n) { return ((size_t)n * (3 * 8 + 4 * 4 + 2) + 15) & ~(size_t)15; }
SQ_HD static inline size_t sq_lsap_sparse_bytes(int n, int m)
{
    const size_t nw = ((size_t)n + 31) / 32;
    return (size_t)m * 8 + (size_t)n * nw * 4 + ((((size_t)n + 1) * 2 + 3) & ~(size_t)3) + (((size_t)n * nw * 2 + 3) & ~(size_t)3) + (size_t)m * 4 + 16;
}
SQ_HD static inline size_t sq_lsap_lds_bytes(int n, int m) { return sq_lsap_vec_bytes(n) + sq_lsap_sparse_bytes(n, m) + 64; }
size_t sq_lsap_scratch_bytes(int n);
size_t sq_nussinov_scratch_bytes(int n);
size_t sq_mwm_scratch_bytes(int n, int nedges);

#ifdef __HIPCC__
void sq_max_dynamic_lds(const void *fn, int bytes);      // sq_host.hip: once per (kernel, device)
int sq_launch_matching(int algo, const SqMatchJob *h_jobs, int nj, const SqMatchJob *jobs, const SqMatchEdge *edges,
                       size_t nedges, SqMatchEdge *dev_edges, char *d_scr, int32_t *out, int32_t *cnt,
                       const uint8_t *codes, uint32_t *job_flags, uint32_t flag_val, hipStream_t st,
                       SqMatchJob *jobs_rw = nullptr, int32_t *bin_head = nullptr, int inflight = 1);
extern "C" {
__global__ void sq_lsap_kernel(const SqMatchJob *jobs, const SqMatchEdge *edges, char *scratch, int32_t *col4row_out,
                               int lds_bytes);
__global__ void sq_nussinov_kernel(const SqMatchJob *jobs, const SqMatchEdge *edges, const uint8_t *codes,
                                   char *scratch, int32_t *pairs_out, int32_t *count_out, int by_pad);
__global__ void sq_mwm_kernel(const SqMatchJob *jobs, const int32_t *bin_head, const SqMatchEdge *edges, char *scratch,
                              int32_t *mate_out, uint32_t *job_flags, uint32_t stamp);
__global__ void sq_mwm_single_kernel(const SqMatchJob *jobs, const SqMatchEdge *edges, char *scratch, int32_t *mate_out,
                                     int lds_bytes, uint32_t *job_flags, uint32_t stamp);
// one thread, launched behind a matching kernel on its stream: publishes "the results are in host memory"
__global__ void sq_flag_kernel(uint32_t *flag, uint32_t value);
}
#endif
