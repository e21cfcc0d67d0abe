// valu_rate.hip — issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_rsq_f32 on gfx950 (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float b = 1.0001f, c = 1e-6f; const f2 b2 = {b, b}, c2 = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7) F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
        } else if (MODE == 1) {
#define P(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b2), "v"(c2));
            P(p0) P(p1) P(p2) P(p3) P(p4) P(p5) P(p6) P(p7) P(p0) P(p1) P(p2) P(p3) P(p4) P(p5) P(p6) P(p7)
        } else {
#define R(x) asm volatile("v_rsq_f32 %0, %0" : "+v"(x));
            R(a0) R(a1) R(a2) R(a3) R(a4) R(a5) R(a6) R(a7) R(a0) R(a1) R(a2) R(a3) R(a4) R(a5) R(a6) R(a7)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[3] = {"v_fma_f32", "v_pk_fma_f32", "v_rsq_f32"};
    for (int wpsimd = 1; wpsimd <= 4; wpsimd *= 2)
        for (int mode = 0; mode < 3; ++mode) {
            dim3 g(256 * wpsimd), b(256);   // 4 waves per WG = 1 per SIMD; wpsimd WGs per CU
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, iters);
                else hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 16 * wpsimd;
            printf("%-14s %d wave(s)/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", names[mode], wpsimd, ms,
                   1e6 * ms / instr_per_simd, 2.4 * 1e6 * ms / instr_per_simd);
        }
    return 0;
}
