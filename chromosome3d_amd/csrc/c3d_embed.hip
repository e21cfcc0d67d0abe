// c3d_embed.hip — A7: metric-matrix distance geometry at bead level (gfx950).
// Reference: the `mmdg` block of the deck, chromosome3D.pl:1471-1525 (knobs :1008-1090): bounds matrix,
// triangle smoothing ("shortest-path-algorithm auto"), random trial distances, embedding from the
// leading eigenvectors of the metric matrix; one embed per model.  Restated for beads:
//   k_dg_bounds          U = L = b0 for (i,i+1); U = L = t for restrained pairs; [lower, inf) otherwise
//   k_fw_u_* / k_fw_l_*  all-pairs shortest paths on U, inverse triangle inequality on L: blocked Floyd-Warshall,
//                        32 x 32 tiles in LDS, three launches per 32 values of k
//   k_dg_trial           per replica: d_ij = L + u (U - L), u ~ Philox4x32-10 counter (i, j, 2); stores d^2
//   k_dg_eig             per replica, one workgroup: orthogonal iteration for the 3 leading eigenpairs of
//                        B = -1/2 J D2 J (never formed: B v = -1/2 J (D2 (J v))), x = sqrt(lambda) v, centred
// Runs once per solve (not part of the SA-step hot loop): O(N^3) smoothing + O(iters N^2) per replica.
#include "c3d_internal.h"

namespace c3d {

constexpr float kDgInf = 1.0e30f;
constexpr int kEigBlock = 1024;

__device__ __forceinline__ void philox4x32_dev(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                               uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void k_dg_bounds(const float* __restrict__ tgt, int n, int npad, float b0, float lower,
                                                  float* __restrict__ U, float* __restrict__ L) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < n; j += 256) {
        const int sep = i > j ? i - j : j - i;
        const float t = tgt[(size_t)i * npad + j];
        float u = kDgInf, l = lower;
        if (sep == 0) { u = 0.0f; l = 0.0f; }
        else if (sep == 1) { u = b0; l = b0; }
        else if (t > 0.0f) { u = t; l = t; }
        U[(size_t)i * n + j] = u;
        L[(size_t)i * n + j] = l;
    }
}
// ---- bound smoothing: blocked Floyd-Warshall, 32 x 32 tiles in LDS, three phases per diagonal block ----------------
// Upper bounds: all-pairs shortest paths, U_ij = min(U_ij, U_ik + U_kj).  Lower bounds (U final): the inverse triangle
// inequality L_ij = max(L_ij, L_ik - U_kj, L_kj - U_ik).  Both have Floyd-Warshall's dependency pattern ((i,j) at step k
// needs (i,k) and (k,j); row k and column k are fixed points of step k), so both run as the classic three-phase blocked
// form: per diagonal block kb   1. the diagonal tile on itself   2. the tiles of block row / block column kb against it
// 3. every other tile against its column-kb and row-kb tiles (no dependency inside the block step).
// 3 launches per 32 values of k instead of 32 (N = 455: 45 + 45 launches instead of 910; N = 2500: 474 instead of 5000).
// Every update is an implied bound, the closure is unique: same result as the one-k-per-launch form up to the rounding
// of the path sums.  Indices >= n read as "no bound" (+inf above, -inf below) and are never written.
constexpr int kFwB = 32;
constexpr float kDgNegInf = -1.0e30f;

template <bool LOWER>
__device__ __forceinline__ void fw_load(const float* __restrict__ M, int n, int bi, int bj, float (*t)[kFwB + 1], int tid) {
    for (int q = tid; q < kFwB * kFwB; q += 256) {
        const int r = q / kFwB, c = q % kFwB, i = bi * kFwB + r, j = bj * kFwB + c;
        t[r][c] = (i < n && j < n) ? M[(size_t)i * n + j] : (LOWER ? kDgNegInf : (i == j ? 0.0f : kDgInf));
    }
}
__device__ __forceinline__ void fw_store(float* __restrict__ M, int n, int bi, int bj, float (*t)[kFwB + 1], int tid) {
    for (int q = tid; q < kFwB * kFwB; q += 256) {
        const int r = q / kFwB, c = q % kFwB, i = bi * kFwB + r, j = bj * kFwB + c;
        if (i < n && j < n) M[(size_t)i * n + j] = t[r][c];
    }
}

// phase 1 (grid 1) and phase 2 (grid 2 (nb - 1): block-row tiles then block-column tiles) of the UPPER bounds
__global__ __launch_bounds__(256) void k_fw_u_12(float* __restrict__ U, int n, int kb, int phase) {
    __shared__ float D[kFwB][kFwB + 1], T[kFwB][kFwB + 1];
    const int tid = threadIdx.x, nb = (n + kFwB - 1) / kFwB;
    fw_load<false>(U, n, kb, kb, D, tid);
    if (phase == 1) {
        __syncthreads();
        for (int k = 0; k < kFwB; ++k) {
            for (int q = tid; q < kFwB * kFwB; q += 256) {
                const int r = q / kFwB, c = q % kFwB;
                D[r][c] = fminf(D[r][c], D[r][k] + D[k][c]);
            }
            __syncthreads();
        }
        fw_store(U, n, kb, kb, D, tid);
        return;
    }
    int o = blockIdx.x % (nb - 1);
    o += o >= kb;                                   // the other block index, skipping kb
    const bool rowtile = blockIdx.x < nb - 1;       // tile (kb, o) else (o, kb)
    const int bi = rowtile ? kb : o, bj = rowtile ? o : kb;
    fw_load<false>(U, n, bi, bj, T, tid);
    __syncthreads();
    for (int k = 0; k < kFwB; ++k) {
        for (int q = tid; q < kFwB * kFwB; q += 256) {
            const int r = q / kFwB, c = q % kFwB;
            T[r][c] = fminf(T[r][c], rowtile ? D[r][k] + T[k][c] : T[r][k] + D[k][c]);
        }
        __syncthreads();
    }
    fw_store(U, n, bi, bj, T, tid);
}
// phase 3 of the upper bounds: grid (nb - 1, nb - 1)
__global__ __launch_bounds__(256) void k_fw_u_3(float* __restrict__ U, int n, int kb) {
    __shared__ float A[kFwB][kFwB + 1], B[kFwB][kFwB + 1];
    const int tid = threadIdx.x;
    int bi = blockIdx.y, bj = blockIdx.x;
    bi += bi >= kb; bj += bj >= kb;
    fw_load<false>(U, n, bi, kb, A, tid);
    fw_load<false>(U, n, kb, bj, B, tid);
    __syncthreads();
    for (int q = tid; q < kFwB * kFwB; q += 256) {
        const int r = q / kFwB, c = q % kFwB, i = bi * kFwB + r, j = bj * kFwB + c;
        if (i >= n || j >= n) continue;
        float v = U[(size_t)i * n + j];
#pragma unroll 8
        for (int k = 0; k < kFwB; ++k) v = fminf(v, A[r][k] + B[k][c]);
        U[(size_t)i * n + j] = v;
    }
}

// the same three phases for the LOWER bounds (U is final): L_ij = max(L_ij, L_ik - U_kj, L_kj - U_ik)
__global__ __launch_bounds__(256) void k_fw_l_12(float* __restrict__ L, const float* __restrict__ U, int n, int kb, int phase) {
    __shared__ float LD[kFwB][kFwB + 1], UD[kFwB][kFwB + 1], LT[kFwB][kFwB + 1], UT[kFwB][kFwB + 1];
    const int tid = threadIdx.x, nb = (n + kFwB - 1) / kFwB;
    fw_load<true>(L, n, kb, kb, LD, tid);
    fw_load<false>(U, n, kb, kb, UD, tid);
    if (phase == 1) {
        __syncthreads();
        for (int k = 0; k < kFwB; ++k) {
            for (int q = tid; q < kFwB * kFwB; q += 256) {
                const int r = q / kFwB, c = q % kFwB;
                LD[r][c] = fmaxf(LD[r][c], fmaxf(LD[r][k] - UD[k][c], LD[k][c] - UD[r][k]));
            }
            __syncthreads();
        }
        fw_store(L, n, kb, kb, LD, tid);
        return;
    }
    int o = blockIdx.x % (nb - 1);
    o += o >= kb;
    const bool rowtile = blockIdx.x < nb - 1;
    const int bi = rowtile ? kb : o, bj = rowtile ? o : kb;
    fw_load<true>(L, n, bi, bj, LT, tid);
    fw_load<false>(U, n, bi, bj, UT, tid);
    __syncthreads();
    for (int k = 0; k < kFwB; ++k) {
        for (int q = tid; q < kFwB * kFwB; q += 256) {
            const int r = q / kFwB, c = q % kFwB;
            // row tile (i in block kb): L_ik, U_ik from the diagonal tile, L_kj, U_kj from this tile; column tile: the mirror
            const float a = rowtile ? LD[r][k] - UT[k][c] : LT[r][k] - UD[k][c];
            const float b = rowtile ? LT[k][c] - UD[r][k] : LD[k][c] - UT[r][k];
            LT[r][c] = fmaxf(LT[r][c], fmaxf(a, b));
        }
        __syncthreads();
    }
    fw_store(L, n, bi, bj, LT, tid);
}
__global__ __launch_bounds__(256) void k_fw_l_3(float* __restrict__ L, const float* __restrict__ U, int n, int kb) {
    __shared__ float LA[kFwB][kFwB + 1], UA[kFwB][kFwB + 1], LB[kFwB][kFwB + 1], UB[kFwB][kFwB + 1];
    const int tid = threadIdx.x;
    int bi = blockIdx.y, bj = blockIdx.x;
    bi += bi >= kb; bj += bj >= kb;
    fw_load<true>(L, n, bi, kb, LA, tid);
    fw_load<false>(U, n, bi, kb, UA, tid);
    fw_load<true>(L, n, kb, bj, LB, tid);
    fw_load<false>(U, n, kb, bj, UB, tid);
    __syncthreads();
    for (int q = tid; q < kFwB * kFwB; q += 256) {
        const int r = q / kFwB, c = q % kFwB, i = bi * kFwB + r, j = bj * kFwB + c;
        if (i >= n || j >= n) continue;
        float v = L[(size_t)i * n + j];
#pragma unroll 8
        for (int k = 0; k < kFwB; ++k) v = fmaxf(v, fmaxf(LA[r][k] - UB[k][c], LB[k][c] - UA[r][k]));
        L[(size_t)i * n + j] = v;
    }
}

__global__ __launch_bounds__(256) void k_dg_clamp(float* __restrict__ L, const float* __restrict__ U, size_t nn) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q < nn) L[q] = fminf(L[q], U[q]);
}

__global__ __launch_bounds__(256) void k_dg_trial(const float* __restrict__ U, const float* __restrict__ L, int n,
                                                 uint32_t key0_base, uint32_t key1_base, uint32_t first_replica,
                                                 float* __restrict__ D2) {
    const int i = blockIdx.x, rep = blockIdx.y;
    const uint32_t rid = first_replica + (uint32_t)rep;
    const uint32_t k0 = key0_base ^ (rid * 0x9E3779B9u), k1 = key1_base + rid;
    float* out = D2 + ((size_t)rep * n + i) * n;
    for (int j = threadIdx.x; j < n; j += 256) {
        float d2 = 0.0f;
        if (i != j) {
            const int a = i < j ? i : j, b = i < j ? j : i;
            const float lo = L[(size_t)a * n + b], hi = U[(size_t)a * n + b];
            uint32_t r[4];
            philox4x32_dev((uint32_t)a, (uint32_t)b, 2u, 0u, k0, k1, r);
            const float u = (float)(((double)r[0] + 0.5) * (1.0 / 4294967296.0));
            const float d = lo + u * (hi - lo);
            d2 = d * d;
        }
        out[j] = d2;
    }
}

// block-wide sum of one value per thread; result in every thread (scratch: kEigBlock/64 floats)
__device__ __forceinline__ float block_sum(float v, float* scratch, int tid) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < kEigBlock / 64; ++w) s += scratch[w];
    return s;
}

// LDS: V[3][n] | W[3][n] | T[3][n] | scratch[16]
__global__ __launch_bounds__(kEigBlock) void k_dg_eig(const float* __restrict__ D2all, const float* __restrict__ v0, int n,
                                                     int npad, int iters, float* __restrict__ x0, float* __restrict__ x1) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int rep = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* V = sm;
    float* W = sm + 3 * n;
    float* T = sm + 6 * n;
    float* scratch = sm + 9 * n;
    const float* D2 = D2all + (size_t)rep * n * n;
    for (int q = tid; q < 3 * n; q += kEigBlock) V[q] = v0[(size_t)rep * 3 * n + q];
    __syncthreads();
    float lam[3] = {0.0f, 0.0f, 0.0f};
    for (int it = 0; it <= iters; ++it) {
        // T = J V  (subtract the mean of each vector)
        for (int k = 0; k < 3; ++k) {
            float s = 0.0f;
            for (int i = tid; i < n; i += kEigBlock) s += V[k * n + i];
            const float mean = block_sum(s, scratch, tid) / (float)n;
            for (int i = tid; i < n; i += kEigBlock) T[k * n + i] = V[k * n + i] - mean;
        }
        __syncthreads();
        // W = D2 T : one wave per row, lanes along j
        for (int i = wave; i < n; i += kEigBlock / 64) {
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
            const float* row = D2 + (size_t)i * n;
            for (int j = lane; j < n; j += 64) {
                const float d = row[j];
                a0 = fmaf(d, T[j], a0); a1 = fmaf(d, T[n + j], a1); a2 = fmaf(d, T[2 * n + j], a2);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64);
            }
            if (lane == 0) { W[i] = a0; W[n + i] = a1; W[2 * n + i] = a2; }
        }
        __syncthreads();
        // W = -1/2 J W
        for (int k = 0; k < 3; ++k) {
            float s = 0.0f;
            for (int i = tid; i < n; i += kEigBlock) s += W[k * n + i];
            const float mean = block_sum(s, scratch, tid) / (float)n;
            for (int i = tid; i < n; i += kEigBlock) W[k * n + i] = -0.5f * (W[k * n + i] - mean);
        }
        __syncthreads();
        if (it == iters) {   // Rayleigh quotients with the current orthonormal V
            for (int k = 0; k < 3; ++k) {
                float s = 0.0f;
                for (int i = tid; i < n; i += kEigBlock) s += V[k * n + i] * W[k * n + i];
                lam[k] = block_sum(s, scratch, tid);
            }
            break;
        }
        for (int k = 0; k < 3; ++k) {   // modified Gram-Schmidt
            for (int q = 0; q < k; ++q) {
                float s = 0.0f;
                for (int i = tid; i < n; i += kEigBlock) s += W[k * n + i] * V[q * n + i];
                const float dot = block_sum(s, scratch, tid);
                for (int i = tid; i < n; i += kEigBlock) W[k * n + i] -= dot * V[q * n + i];
                __syncthreads();
            }
            float s = 0.0f;
            for (int i = tid; i < n; i += kEigBlock) s += W[k * n + i] * W[k * n + i];
            const float nrm = sqrtf(fmaxf(block_sum(s, scratch, tid), 1e-30f));
            for (int i = tid; i < n; i += kEigBlock) V[k * n + i] = W[k * n + i] / nrm;
            __syncthreads();
        }
    }
    // coordinates, centred, into both parity buffers (SoA, padding untouched)
    for (int k = 0; k < 3; ++k) {
        const float sc = sqrtf(fmaxf(lam[k], 0.0f));
        float s = 0.0f;
        for (int i = tid; i < n; i += kEigBlock) s += sc * V[k * n + i];
        const float mean = block_sum(s, scratch, tid) / (float)n;
        for (int i = tid; i < n; i += kEigBlock) {
            const float c = sc * V[k * n + i] - mean;
            x0[((size_t)rep * 3 + k) * npad + i] = c;
            x1[((size_t)rep * 3 + k) * npad + i] = c;
        }
    }
}

// v0: [nrep][3][n] starting vectors (host Philox normals).  U, L: n*n scratch; D2: nrep*n*n scratch.
hipError_t launch_dg_embed(const float* tgt, int n, int npad, int nrep, float b0, float lower, uint64_t seed,
                           uint32_t first_replica, int iters, const float* v0, float* U, float* L, float* D2, float* x0,
                           float* x1, hipStream_t s) {
    hipLaunchKernelGGL(k_dg_bounds, dim3(n), dim3(256), 0, s, tgt, n, npad, b0, lower, U, L);
    const int nb = (n + kFwB - 1) / kFwB;
    for (int kb = 0; kb < nb; ++kb) {
        hipLaunchKernelGGL(k_fw_u_12, dim3(1), dim3(256), 0, s, U, n, kb, 1);
        if (nb > 1) {
            hipLaunchKernelGGL(k_fw_u_12, dim3(2 * (nb - 1)), dim3(256), 0, s, U, n, kb, 2);
            hipLaunchKernelGGL(k_fw_u_3, dim3(nb - 1, nb - 1), dim3(256), 0, s, U, n, kb);
        }
    }
    for (int kb = 0; kb < nb; ++kb) {
        hipLaunchKernelGGL(k_fw_l_12, dim3(1), dim3(256), 0, s, L, U, n, kb, 1);
        if (nb > 1) {
            hipLaunchKernelGGL(k_fw_l_12, dim3(2 * (nb - 1)), dim3(256), 0, s, L, U, n, kb, 2);
            hipLaunchKernelGGL(k_fw_l_3, dim3(nb - 1, nb - 1), dim3(256), 0, s, L, U, n, kb);
        }
    }
    const size_t nn = (size_t)n * n;
    hipLaunchKernelGGL(k_dg_clamp, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s, L, U, nn);
    hipLaunchKernelGGL(k_dg_trial, dim3(n, nrep), dim3(256), 0, s, U, L, n, (uint32_t)(seed & 0xFFFFFFFFu),
                       (uint32_t)(seed >> 32), first_replica, D2);
    const size_t lds = sizeof(float) * ((size_t)9 * n + 16);
    // (above 64 KB of dynamic LDS a kernel needs an allowance: preload_embed_unit set it when the unit was loaded)
    hipLaunchKernelGGL(k_dg_eig, dim3(nrep), dim3(kEigBlock), lds, s, D2, v0, n, npad, iters, x0, x1);
    return hipGetLastError();
}

// loads the unit's code object on the current device and allows k_dg_eig the whole LDS of a CU (c3d_api.cpp "code objects": once per
// device, never beside a launch)
hipError_t preload_embed_unit() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_dg_eig), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace c3d
