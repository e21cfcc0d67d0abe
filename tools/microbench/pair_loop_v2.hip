// Pair-loop experiments for round 3 (gfx950): the cluster kernel's register-resident pair loop, bare, in several
// formulations and wave layouts.  ns per pair term per SIMD is the figure of merit (a step of chr1_500kb x 20 evaluates
// 96 pair terms per SIMD on the fullest CUs).
//   F0  round 2's pair_term (16 VALU instructions; 17 when rswitch != 1)
//   F1  scaled form, 15 instructions: coordinates pre-divided by the repel radius R, targets pre-divided by rswitch,
//       NOE weight folded into one per-row factor at the end; constants in VGPRs
//   F2  F1 with the quad (one row x four columns) written as ONE inline-asm block: explicit operand order so that src0 and
//       src1 of no instruction share a VGPR bank is left to the compiler's allocation of the asm operands ("v" constraints)
// Layouts: CW compute waves x RPW rows, NH idle helper waves.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -o pair_loop_v2 pair_loop_v2.hip && ./pair_loop_v2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../chromosome3d_amd/csrc/c3d_step_core.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace c3d;

#pragma clang fp contract(off)
struct K2 { float nm, kq; };      // -mrswitch / rswitch ; repel weight relative to the NOE weight

// one pair, scaled form.  b = target / rswitch (0: none), a = R / rswitch where restrained else 0, d* = scaled differences
__device__ __forceinline__ void pair_term_s(const float nm, const float kq, float b, float a, float dx, float dy, float dz, float& fx, float& fy, float& fz) {
    const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, 1e-12f)));
    const float rinv = __builtin_amdgcn_rsqf(r2);
    const float mu = fmaf(-b, rinv, a);
    const float s = __builtin_amdgcn_fmed3f(mu, nm * rinv, rinv);
    float q01;
    asm("v_sub_f32 %0, 1.0, %1 clamp" : "=v"(q01) : "v"(r2));
    const float c = fmaf(kq, q01, s);
    fx = fmaf(c, dx, fx);
    fy = fmaf(c, dy, fy);
    fz = fmaf(c, dz, fz);
}

template <int RPW, int NB>
__device__ __forceinline__ void tile_s(float nm, float kq, const float4 (&tv)[RPW][NB], const float4* mw_lds, const float* xs, const float* ys,
                                       const float* zs, int row0, int lane, float& Fx, float& Fy, float& Fz) {
    float fx[RPW], fy[RPW], fz[RPW], xi[RPW], yi[RPW], zi[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) { xi[r] = xs[row0 + r]; yi[r] = ys[row0 + r]; zi[r] = zs[row0 + r]; fx[r] = fy[r] = fz[r] = 0.0f; }
    asm volatile("" : "+v"(nm), "+v"(kq));        // constants live in VGPRs (an SGPR source costs a second issue cycle)
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        const int j = 256 * jb + 4 * lane;
        const float4 xj = *reinterpret_cast<const float4*>(xs + j);
        const float4 yj = *reinterpret_cast<const float4*>(ys + j);
        const float4 zj = *reinterpret_cast<const float4*>(zs + j);
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const float4 a = mw_lds[(r * NB + jb) * 64 + lane];
            const float4 b = tv[r][jb];
            pair_term_s(nm, kq, b.x, a.x, xi[r] - xj.x, yi[r] - yj.x, zi[r] - zj.x, fx[r], fy[r], fz[r]);
            pair_term_s(nm, kq, b.y, a.y, xi[r] - xj.y, yi[r] - yj.y, zi[r] - zj.y, fx[r], fy[r], fz[r]);
            pair_term_s(nm, kq, b.z, a.z, xi[r] - xj.z, yi[r] - yj.z, zi[r] - zj.z, fx[r], fy[r], fz[r]);
            pair_term_s(nm, kq, b.w, a.w, xi[r] - xj.w, yi[r] - yj.w, zi[r] - zj.w, fx[r], fy[r], fz[r]);
            asm volatile("" : "+v"(fx[r]), "+v"(fy[r]), "+v"(fz[r]));
        }
    }
    Fx = reduce_rows<RPW>(fx, lane); Fy = reduce_rows<RPW>(fy, lane); Fz = reduce_rows<RPW>(fz, lane);
}
#pragma clang fp contract(fast)

template <int FORM, int RPW>
__global__ __launch_bounds__(1024) void k_loop(const float* __restrict__ in, float* __restrict__ out, int iters, int CW, int NH, DevModel m, DevStep p, K2 k2) {
    constexpr int NB = 2, NPAD = 512;
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
        __syncthreads();
        if (is_compute) {
            float Fx, Fy, Fz;
            if constexpr (FORM == 0) tile_pair_sums_reg<3, RPW, NB, 4, 1>(m, p, tv, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);   // the product's form
            else if constexpr (FORM == 10) tile_pair_sums_reg<3, RPW, NB, 3, 1>(m, p, tv, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);   // last block 3 columns per lane (N = 455)
            else if constexpr (FORM == 11) tile_pair_sums_reg<3, RPW, NB, 3, 2>(m, p, tv, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);  // eight pair terms in flight
            else if constexpr (FORM == 12) tile_pair_sums_reg<3, RPW, NB, 3, 3>(m, p, tv, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);  // a whole block of all rows in flight
            else tile_s<RPW, NB>(k2.nm, k2.kq, tv, mw, xs, ys, zs, row0, lane, Fx, Fy, Fz);
            if (lane < RPW) { const int k = cwave * RPW + lane; fbuf[k] = Fx; fbuf[64 + k] = Fy; fbuf[128 + k] = Fz; }
            acc += Fx;
        }
        if (tid < 48) { xs[tid] += 1e-4f * fbuf[tid]; }
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

template <int FORM, int RPW>
static float run(int grid, int threads, int lds, const float* in, float* out, int iters, int cw, int nh, DevModel m, DevStep p, K2 k2, hipEvent_t e0, hipEvent_t e1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<FORM, RPW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        k_loop<FORM, RPW><<<grid, threads, lds>>>(in, out, iters, cw, nh, m, p, k2);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 1024 * prop.multiProcessorCount * 2));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.37f * (i % 29) - 4.0f + 0.01f * i;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    DevModel m{}; m.n = 455; m.npad = 512; m.nmrs = -4.0f; m.mrs = 4.0f; m.rs = 0.5f; m.inv_rs = 2.0f; m.nm_rs = -8.0f; m.wl = 4;
    DevStep p{}; p.kind = 1; p.w_noe2n = -20.0f; p.inv_rep_r2 = 1.0f / 21.0f; p.w_rep4r2 = 4.0f * 21.0f; p.w_rs = -10.0f; p.kq = -8.4f;
    K2 k2{-8.0f, 0.3f};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 100 * 1024, iters = 2000, grid = prop.multiProcessorCount;
    struct { int form, rpw, cw, nh; const char* what; } cases[] = {
        {0, 4, 12, 4, "product form (c3d_step_core.h, 15 instr)   12 x 4 rows + 4 idle"},
        {10, 4, 12, 4, "product form, 7 column slots (N = 455)     12 x 4 rows + 4 idle"},
        {11, 4, 12, 4, "product form, 7 slots, 8 terms in flight   12 x 4 rows + 4 idle"},
        {12, 4, 12, 4, "product form, 7 slots, 16 terms in flight  12 x 4 rows + 4 idle"},
        {1, 4, 12, 4, "F1 scaled form (15 instr)                  12 x 4 rows + 4 idle"},
        {0, 3, 16, 0, "product form                               16 x 3 rows"},
        {1, 3, 16, 0, "F1 scaled form                             16 x 3 rows"},
        {1, 4, 16, 0, "F1 scaled form                             16 x 4 rows (64 rows)"},
        {1, 2, 16, 0, "F1 scaled form                             16 x 2 rows (32 rows)"},
        {1, 4, 8, 0, "F1 scaled form                              8 x 4 rows (32 rows)"},
    };
    for (auto& c : cases) {
        const int threads = (c.cw + c.nh) * 64;
        float ms = 0;
        if (c.form == 0 && c.rpw == 4) ms = run<0, 4>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        else if (c.form == 10 && c.rpw == 4) ms = run<10, 4>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        else if (c.form == 11 && c.rpw == 4) ms = run<11, 4>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        else if (c.form == 12 && c.rpw == 4) ms = run<12, 4>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        else if (c.form == 1 && c.rpw == 4) ms = run<1, 4>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        else if (c.form == 0 && c.rpw == 3) ms = run<0, 3>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        else if (c.form == 1 && c.rpw == 3) ms = run<1, 3>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        else if (c.form == 1 && c.rpw == 2) ms = run<1, 2>(grid, threads, lds, in, out, iters, c.cw, c.nh, m, p, k2, e0, e1);
        const double us = ms * 1e3 / iters;
        const double pairs_per_simd = (double)c.cw / 4 * c.rpw * 8;
        printf("%-72s %.3f us per pass, %5.1f ns per pair term per SIMD, %d rows per CU: %.1f ns per row\n", c.what, us, us * 1e3 / pairs_per_simd, c.cw * c.rpw, us * 1e3 / (c.cw * c.rpw));
    }
    return 0;
}
