// The cluster kernel's pair loop (c3d_step_core.h: tile_pair_sums_reg<3, 4, 2, true, true>) run bare: one 1024-thread
// workgroup per CU, CW compute waves (4 rows each, targets in registers, weights and coordinates in LDS), `iters` passes.
// Options peel the step's structure back on: a barrier per pass, idle or busy helper waves.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -o pair_loop_insitu pair_loop_insitu.hip && ./pair_loop_insitu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../chromosome3d_amd/csrc/c3d_step_core.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace c3d;

template <int MODE>   // 0 bare, 1 + one barrier per pass, 2 + two barriers per pass
__global__ __launch_bounds__(1024) void k_loop(const float* __restrict__ in, float* __restrict__ out, int iters, int CW, int NH, DevModel m, DevStep p) {
    constexpr int RPW = 4, NB = 2, NPAD = 512;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem; float* ys = smem + NPAD; float* zs = smem + 2 * NPAD;
    float* fbuf = smem + 3 * NPAD;
    float4* mwbuf = reinterpret_cast<float4*>(smem + 3 * NPAD + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = tid; b < 3 * NPAD; b += blockDim.x) smem[b] = in[b & 1023] * (1.0f + 0.01f * (b >> 10));
    const bool is_compute = wave >= NH && wave < NH + CW;
    const int cwave = wave - NH;
    float4 tv[RPW][NB];
    for (int r = 0; r < RPW; ++r) for (int jb = 0; jb < NB; ++jb) tv[r][jb] = make_float4(3.0f + in[(tid + r) & 1023], 5.0f, 7.0f, 9.0f + in[(tid + jb) & 1023]);
    float4* const mw = mwbuf + (size_t)(is_compute ? cwave : 0) * (RPW * NB * 64);
    if (is_compute) for (int r = 0; r < RPW; ++r) for (int jb = 0; jb < NB; ++jb) mw[(r * NB + jb) * 64 + lane] = pair_a<false>(m, p, tv[r][jb]);
    __syncthreads();
    float acc = 0.0f;
    const int row0 = cwave * RPW;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE >= 1) __syncthreads();
        if (is_compute) {
            float Fx, Fy, Fz;
            tile_pair_sums_reg<3, RPW, NB, 4, 1>(m, p, tv, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);
            if (lane < RPW) { const int k = cwave * RPW + lane; fbuf[k] = Fx; fbuf[64 + k] = Fy; fbuf[128 + k] = Fz; }
            acc += Fx;
        }
        if constexpr (MODE >= 2) __syncthreads();
        if (tid < 48) { xs[tid] += 1e-4f * fbuf[tid]; }           // something moves between passes
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

int main(int argc, char** argv) {
    const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 1;            // 2: two workgroups per CU (LDS 60 KB each)
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 1024 * prop.multiProcessorCount * 2));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.37f * (i % 29) - 4.0f + 0.01f * i;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    DevModel m{}; m.n = 455; m.npad = 512; m.nmrs = -4.0f; m.mrs = 4.0f; m.rs = 0.5f; m.inv_rs = 2.0f; m.nm_rs = -8.0f; m.wl = 4;
    DevStep p{}; p.kind = 1; p.w_noe2n = -20.0f; p.inv_rep_r2 = 1.0f / 21.0f; p.w_rep4r2 = 4.0f * 21.0f; p.w_rs = -10.0f; p.kq = -8.4f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = (wgs_per_cu > 1 ? 60 : 100) * 1024, iters = 2000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    struct { int mode, cw, nh, threads; const char* what; } cases[] = {
        {0, 12, 0, 768, "12 compute waves (3 per SIMD), bare loop"},
        {0, 12, 4, 1024, "12 compute + 4 idle waves, bare loop"},
        {0, 16, 0, 1024, "16 compute waves (4 per SIMD), bare loop"},
        {0, 8, 0, 512, "8 compute waves (2 per SIMD), bare loop"},
        {0, 4, 0, 256, "4 compute waves (1 per SIMD), bare loop"},
        {1, 12, 4, 1024, "12 compute + 4 idle waves, one barrier per pass"},
        {2, 12, 4, 1024, "12 compute + 4 idle waves, two barriers per pass"},
    };
    for (auto& c : cases) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (c.mode == 0) k_loop<0><<<prop.multiProcessorCount * wgs_per_cu, c.threads, lds>>>(in, out, iters, c.cw, c.nh, m, p);
            else if (c.mode == 1) k_loop<1><<<prop.multiProcessorCount * wgs_per_cu, c.threads, lds>>>(in, out, iters, c.cw, c.nh, m, p);
            else k_loop<2><<<prop.multiProcessorCount * wgs_per_cu, c.threads, lds>>>(in, out, iters, c.cw, c.nh, m, p);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double per_pass_us = best * 1e3 / iters;
        const double pairs_per_simd = (double)c.cw / 4 * 32 * wgs_per_cu;
        printf("%-52s %.3f us per pass, %.1f ns per pair term per SIMD\n", c.what, per_pass_us, per_pass_us * 1e3 / pairs_per_simd);
    }
    return 0;
}
