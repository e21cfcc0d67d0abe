// What does one pair term cost per lane, scalar against packed fp32 (v_pk_*_f32 on column pairs)?  Same arithmetic per
// element (c3d_step_core.h pair_term, POT 3, RS1): 3 sub, 3 fma, rsq, u, lower cap, med3, weight, repel (fma clamp + fma),
// 3 accumulate.  Registers only, W waves per SIMD, ns per pair-lane.
//   hipcc -O3 --offload-arch=gfx950 -o pair_rate pair_rate.hip && ./pair_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

struct Par { float nmrs, inv_rep_r2, w_rep4r2; };

__device__ __forceinline__ void term_scalar(const Par& p, float v, float mw, float dx, float dy, float dz, float& fx, float& fy, float& fz) {
    const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, 1e-12f)));
    const float rinv = __builtin_amdgcn_rsqf(r2);
    const float u = fmaf(-v, rinv, 1.0f);
    const float s = __builtin_amdgcn_fmed3f(u, p.nmrs * rinv, rinv);
    float c = mw * s;
    float q01;
    asm("v_fma_f32 %0, -%1, %2, 1.0 clamp" : "=v"(q01) : "v"(r2), "v"(p.inv_rep_r2));
    c = fmaf(p.w_rep4r2, q01, c);
    fx = fmaf(c, dx, fx); fy = fmaf(c, dy, fy); fz = fmaf(c, dz, fz);
}

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_fma_nega(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_fma_nega_clamp(f2 a, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { f2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) { f2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }

// two columns at once; xi2 = {xi, xi} etc. are formed once per row
__device__ __forceinline__ void term_packed(const f2 nmrs2, const f2 invr2, const f2 wrep2, const f2 eps2, const f2 one2, f2 v, f2 mw, f2 xi2, f2 yi2,
                                            f2 zi2, f2 xj, f2 yj, f2 zj, f2& fx, f2& fy, f2& fz) {
    const f2 dx = pk_sub(xi2, xj), dy = pk_sub(yi2, yj), dz = pk_sub(zi2, zj);
    const f2 r2 = pk_fma(dx, dx, pk_fma(dy, dy, pk_fma(dz, dz, eps2)));
    f2 rinv;
    rinv.x = __builtin_amdgcn_rsqf(r2.x); rinv.y = __builtin_amdgcn_rsqf(r2.y);
    const f2 u = pk_fma_nega(v, rinv, one2);
    const f2 lo = pk_mul(nmrs2, rinv);
    f2 s;
    s.x = __builtin_amdgcn_fmed3f(u.x, lo.x, rinv.x); s.y = __builtin_amdgcn_fmed3f(u.y, lo.y, rinv.y);
    f2 c = pk_mul(mw, s);
    const f2 q = pk_fma_nega_clamp(r2, invr2, one2);
    c = pk_fma(wrep2, q, c);
    fx = pk_fma(c, dx, fx); fy = pk_fma(c, dy, fy); fz = pk_fma(c, dz, fz);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_pairs(const float* __restrict__ in, float* __restrict__ out, int iters, Par p) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    // 8 columns per lane (two float4 of x, y, z each), 4 rows: like one compute wave of the cluster kernel at NB = 2
    float xj[8], yj[8], zj[8], tv[4][8], mw[4][8], xi[4], yi[4], zi[4];
    for (int k = 0; k < 8; ++k) { xj[k] = in[(t + k) & 1023]; yj[k] = in[(t + 2 * k + 1) & 1023]; zj[k] = in[(t + 3 * k + 2) & 1023]; }
    for (int r = 0; r < 4; ++r) {
        xi[r] = in[(r + 5) & 1023]; yi[r] = in[(r + 9) & 1023]; zi[r] = in[(r + 13) & 1023];
        for (int k = 0; k < 8; ++k) { tv[r][k] = 3.0f + in[(t + r + k) & 1023]; mw[r][k] = -20.0f; }
    }
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2 || MODE == 3) {
            // MODE 2: the product's fence after every quad (four terms in flight); MODE 3: after every second quad (eight)
            float fx[4] = {0, 0, 0, 0}, fy[4] = {0, 0, 0, 0}, fz[4] = {0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < 8; kb += 4)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int k = kb; k < kb + 4; ++k) term_scalar(p, tv[r][k], mw[r][k], xi[r] - xj[k], yi[r] - yj[k], zi[r] - zj[k], fx[r], fy[r], fz[r]);
                    if constexpr (MODE == 2) asm volatile("" : "+v"(fx[r]), "+v"(fy[r]), "+v"(fz[r]));
                    if constexpr (MODE == 3) { if (r & 1) asm volatile("" : "+v"(fx[r]), "+v"(fy[r]), "+v"(fz[r]), "+v"(fx[r - 1]), "+v"(fy[r - 1]), "+v"(fz[r - 1])); }
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += fx[r] + fy[r] + fz[r];
        } else {
            const f2 nmrs2 = {p.nmrs, p.nmrs}, invr2 = {p.inv_rep_r2, p.inv_rep_r2}, wrep2 = {p.w_rep4r2, p.w_rep4r2}, eps2 = {1e-12f, 1e-12f}, one2 = {1.0f, 1.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f2 fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
                const f2 xi2 = {xi[r], xi[r]}, yi2 = {yi[r], yi[r]}, zi2 = {zi[r], zi[r]};
#pragma unroll
                for (int k = 0; k < 8; k += 2)
                    term_packed(nmrs2, invr2, wrep2, eps2, one2, f2{tv[r][k], tv[r][k + 1]}, f2{mw[r][k], mw[r][k + 1]}, xi2, yi2, zi2,
                                f2{xj[k], xj[k + 1]}, f2{yj[k], yj[k + 1]}, f2{zj[k], zj[k + 1]}, fx, fy, fz);
                acc += (fx.x + fx.y) + (fy.x + fy.y) + (fz.x + fz.y);
            }
        }
        // the positions move a little every iteration, as they do between steps
#pragma unroll
        for (int k = 0; k < 8; ++k) xj[k] += 1e-3f * acc;
    }
    out[t] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *in, *out; CK(hipMalloc(&in, 4096)); CK(hipMalloc(&out, 4 * 256 * 4 * 8 * 256));
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0.37f * (i % 29) - 4.0f;
    CK(hipMemcpy(in, h, 4096, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const Par p{-11.0f, 1.0f / 45.5f, 4.0f * 45.5f};
    const int iters = 2000;
    const char* names[] = {"scalar", "packed", "scalar, 4 in flight", "scalar, 8 in flight"};
    for (int wps : {1, 2, 3, 4})
        for (int mode = 0; mode < 4; ++mode) {
            const int grid = prop.multiProcessorCount * wps;           // 256-thread blocks: 4 waves = one per SIMD
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) k_pairs<0><<<grid, 256>>>(in, out, iters, p);
                else if (mode == 1) k_pairs<1><<<grid, 256>>>(in, out, iters, p);
                else if (mode == 2) k_pairs<2><<<grid, 256>>>(in, out, iters, p);
                else k_pairs<3><<<grid, 256>>>(in, out, iters, p);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double pairs_per_simd = (double)wps * iters * 32;    // pair-lanes: 4 rows x 8 columns per wave and iteration
            printf("%-20s %d wave(s)/SIMD: %.3f ms -> %.2f ns per pair term per SIMD\n", names[mode], wps, best, best * 1e6 / pairs_per_simd);
        }
    return 0;
}
