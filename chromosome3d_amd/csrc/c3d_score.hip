// c3d_score.hip — K6: assessment and Spearman scoring of every replica on the device (gfx950), so that
// ranking needs no read-back of coordinates.
//   count_satisfied_tbl_rows / sum_noe_dev   chromosome3D.pl:447-485, 581-600 (distances as "%.3f" text)
//   Spearman(IF_ij, d_ij), |i-j| >= range     spearman_IF_pdb.pl:42-70 (average ranks, d as "%.3f")
// The reference reads coordinates back from "%8.3f" PDB text and prints distances with "%.3f": both
// roundings are reproduced exactly with the fma-residual rule (round_milli), so distances are integers in
// thousandths of an Angstrom and their average ranks come from a histogram — no sort:
//   k_score_round   xr = round3(x) in fp64
//   k_score_hist    histogram of dq = round3(|xr_i - xr_j|) over ordered pairs |i-j| >= range (int atomics)
//   k_score_scan    exclusive prefix of the histogram (one workgroup per replica)
//   k_score_corr    sum (ra - ma)(rb - mb), sum (rb - mb)^2 with rb = below[dq] + (cnt[dq] + 1)/2, and the
//                   satisfied / sum-of-deviation tallies over the restrained pairs; per-block fp64 partials
// The IF ranks ra (one N x N fp64 matrix per input matrix) are computed once on the host.
#include "c3d_internal.h"

namespace c3d {

__device__ __forceinline__ long long round_milli_dev(double d) {
    const double p = d * 1000.0;
    const double err = fma(d, 1000.0, -p);
    double r = rint(p);
    const double diff = p - r;
    if (diff == 0.5 || diff == -0.5) {
        if (err > 0) r = floor(p) + 1.0;
        else if (err < 0) r = floor(p);
    }
    return (long long)r;
}

__global__ __launch_bounds__(256) void k_score_round(const float* __restrict__ xin, int n, int npad, double* __restrict__ xr) {
    const int rep = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;    // over 3*n
    if (q >= 3 * n) return;
    const int comp = q / n, i = q - comp * n;
    xr[((size_t)rep * 3 + comp) * n + i] = (double)round_milli_dev((double)xin[((size_t)rep * 3 + comp) * npad + i]) / 1000.0;
}

__device__ __forceinline__ unsigned pair_dq(const double* __restrict__ xr, int n, int i, int j, unsigned nbins, int* overflow) {
    const double dx = xr[i] - xr[j], dy = xr[n + i] - xr[n + j], dz = xr[2 * n + i] - xr[2 * n + j];
    const long long q = round_milli_dev(sqrt(dx * dx + dy * dy + dz * dz));
    if (q >= (long long)nbins) { *overflow = 1; return nbins - 1; }
    return (unsigned)q;
}

__global__ __launch_bounds__(256) void k_score_hist(const double* __restrict__ xr_all, int n, int range, unsigned nbins,
                                                   unsigned* __restrict__ hist_all, int* __restrict__ overflow) {
    const int i = blockIdx.x, rep = blockIdx.y;
    const double* xr = xr_all + (size_t)rep * 3 * n;
    unsigned* hist = hist_all + (size_t)rep * nbins;
    for (int j = threadIdx.x; j < n; j += 256) {
        const int sep = i > j ? i - j : j - i;
        if (sep < range) continue;
        atomicAdd(&hist[pair_dq(xr, n, i, j, nbins, overflow)], 1u);
    }
}

// exclusive prefix sum of hist -> below (one workgroup per replica, 1024 threads, sequential chunks)
__global__ __launch_bounds__(1024) void k_score_scan(const unsigned* __restrict__ hist_all, unsigned nbins,
                                                    unsigned* __restrict__ below_all) {
    __shared__ unsigned part[1024];
    const int rep = blockIdx.x, tid = threadIdx.x;
    const unsigned* hist = hist_all + (size_t)rep * nbins;
    unsigned* below = below_all + (size_t)rep * nbins;
    const unsigned per = (nbins + 1023u) / 1024u;
    const unsigned lo = tid * per, hi = min(lo + per, nbins);
    unsigned s = 0;
    for (unsigned v = lo; v < hi; ++v) s += hist[v];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int k = 0; k < 1024; ++k) { const unsigned t = part[k]; part[k] = run; run += t; }
    }
    __syncthreads();
    unsigned run = part[tid];
    for (unsigned v = lo; v < hi; ++v) { below[v] = run; run += hist[v]; }
}

// partial[rep][row][0..3] = { sum (ra-ma)(rb-mb), sum (rb-mb)^2, satisfied (as double), sum_dev }
__global__ __launch_bounds__(256) void k_score_corr(const double* __restrict__ xr_all, const float* __restrict__ tgt,
                                                   const double* __restrict__ rankA, int n, int npad, int range, int min_sep,
                                                   unsigned nbins, const unsigned* __restrict__ hist_all,
                                                   const unsigned* __restrict__ below_all, double ma, double mb, double relax,
                                                   double* __restrict__ partial, int* __restrict__ overflow) {
    __shared__ double red[4][256];
    const int i = blockIdx.x, rep = blockIdx.y, tid = threadIdx.x;
    const double* xr = xr_all + (size_t)rep * 3 * n;
    const unsigned* hist = hist_all + (size_t)rep * nbins;
    const unsigned* below = below_all + (size_t)rep * nbins;
    double sab = 0, sbb = 0, sat = 0, dev = 0;
    for (int j = tid; j < n; j += 256) {
        const int sep = i > j ? i - j : j - i;
        if (sep == 0) continue;
        const unsigned dq = pair_dq(xr, n, i, j, nbins, overflow);
        if (sep >= range && rankA) {
            const double rb = (double)below[dq] + 0.5 * ((double)hist[dq] + 1.0);
            const double a = rankA[(size_t)i * n + j] - ma, b = rb - mb;
            sab += a * b;
            sbb += b * b;
        }
        const float tv = tgt[(size_t)i * npad + j];
        if (j > i && sep >= min_sep && tv > 0.0f) {
            const double t = (double)lrintf(tv * 10.0f) / 10.0;     // the tbl's "%.2f" value, exactly t10/10
            const double d = (double)dq / 1000.0;
            if (d < t + 0.0 + relax) sat += 1.0;
            if (d < t - 0.0 - relax) sat -= 1.0;
            if (d > t + 0.0 + 0.2) dev += d - (t + 0.0);
            if (d < t - 0.0 - 0.2) dev += (t - 0.0) - d;
        }
    }
    red[0][tid] = sab; red[1][tid] = sbb; red[2][tid] = sat; red[3][tid] = dev;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) for (int c = 0; c < 4; ++c) red[c][tid] += red[c][tid + s];
        __syncthreads();
    }
    if (tid < 4) partial[((size_t)rep * n + i) * 4 + tid] = red[tid][0];
}

hipError_t launch_score(const float* xin, const float* tgt, const double* rankA, int n, int npad, int nrep, int range, int min_sep,
                        unsigned nbins, double ma, double mb, double relax, double* xr, unsigned* hist, unsigned* below,
                        double* partial, int* overflow, hipStream_t s) {
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned) * (size_t)nbins * nrep, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(overflow, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_score_round, dim3((3 * n + 255) / 256, nrep), dim3(256), 0, s, xin, n, npad, xr);
    hipLaunchKernelGGL(k_score_hist, dim3(n, nrep), dim3(256), 0, s, xr, n, range, nbins, hist, overflow);
    hipLaunchKernelGGL(k_score_scan, dim3(nrep), dim3(1024), 0, s, hist, nbins, below);
    hipLaunchKernelGGL(k_score_corr, dim3(n, nrep), dim3(256), 0, s, xr, tgt, rankA, n, npad, range, min_sep, nbins, hist, below, ma,
                       mb, relax, partial, overflow);
    return hipGetLastError();
}

hipError_t preload_score_unit() {
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_score_round));
}

}  // namespace c3d
