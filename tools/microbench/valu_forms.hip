// Issue cost of VALU instruction FORMS on gfx950: how many VGPR sources, which register banks (reg % 4), VOP2 vs VOP3,
// SGPR / literal operands.  16 independent instructions per block (destinations v40..v55, sources v20..v31 and s20),
// explicit registers, 4 and 3 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_forms valu_forms.hip && ./valu_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP16(OP, A, B, C) \
    OP(40, A, B, C) OP(41, A, B, C) OP(42, A, B, C) OP(43, A, B, C) OP(44, A, B, C) OP(45, A, B, C) OP(46, A, B, C) OP(47, A, B, C) \
    OP(48, A, B, C) OP(49, A, B, C) OP(50, A, B, C) OP(51, A, B, C) OP(52, A, B, C) OP(53, A, B, C) OP(54, A, B, C) OP(55, A, B, C)
#define FMA3(D, A, B, C) "v_fma_f32 v" #D ", v" #A ", v" #B ", v" #C "\n"
#define FMA_ACC(D, A, B, C) "v_fma_f32 v" #D ", v" #A ", v" #B ", v" #D "\n"
#define FMAC(D, A, B, C) "v_fmac_f32_e32 v" #D ", v" #A ", v" #B "\n"
#define MUL2(D, A, B, C) "v_mul_f32_e32 v" #D ", v" #A ", v" #B "\n"
#define MULS(D, A, B, C) "v_mul_f32_e32 v" #D ", s20, v" #B "\n"
#define SUBS(D, A, B, C) "v_sub_f32_e32 v" #D ", s20, v" #B "\n"
#define FMAS(D, A, B, C) "v_fma_f32 v" #D ", s20, v" #B ", v" #C "\n"
#define FMAK(D, A, B, C) "v_fmaak_f32 v" #D ", v" #A ", v" #B ", 0x2b8cbccc\n"
#define MED3(D, A, B, C) "v_med3_f32 v" #D ", v" #A ", v" #B ", v" #C "\n"
#define RSQ(D, A, B, C) "v_rsq_f32_e32 v" #D ", v" #A "\n"
#define FMAC_DPP(D, A, B, C) "v_fmac_f32_dpp v" #D ", v" #A ", v" #B " quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n"
#define MUL_DPP(D, A, B, C) "v_mul_f32_dpp v" #D ", v" #A ", v" #B " quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n"
#define FMA_SQ3(D, A, B, C) "v_fma_f32 v" #D ", v" #A ", v" #A ", v" #C "\n"
#define FMAK_SQ(D, A, B, C) "v_fmaak_f32 v" #D ", v" #A ", v" #A ", 0x2b8cbccc\n"
#define MUL_SDWA(D, A, B, C) "v_mul_f32_sdwa v" #D ", v" #A ", v" #B " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
             "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "s20"

template <int F> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    asm volatile("v_mov_b32 v20, 1.0\nv_mov_b32 v21, 0.5\nv_mov_b32 v22, 2.0\nv_mov_b32 v23, 1.0\nv_mov_b32 v24, 0.5\nv_mov_b32 v25, 1.0\n"
                 "v_mov_b32 v26, 0.5\nv_mov_b32 v27, 1.0\nv_mov_b32 v28, 0.5\nv_mov_b32 v29, 1.0\nv_mov_b32 v30, 0.5\nv_mov_b32 v31, 1.0\ns_mov_b32 s20, 1.0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if constexpr (F == 0) asm volatile(REP16(FMA3, 20, 21, 22) ::: CLOB);        // three sources, banks 0 1 2
        if constexpr (F == 1) asm volatile(REP16(FMA3, 20, 24, 28) ::: CLOB);        // three sources, all bank 0
        if constexpr (F == 2) asm volatile(REP16(FMA3, 20, 21, 25) ::: CLOB);        // banks 0 1 1
        if constexpr (F == 3) asm volatile(REP16(FMA_ACC, 20, 21, 0) ::: CLOB);      // d = a * b + d (VOP3)
        if constexpr (F == 4) asm volatile(REP16(FMAC, 20, 21, 0) ::: CLOB);         // VOP2 fmac
        if constexpr (F == 5) asm volatile(REP16(FMAC, 20, 20, 0) ::: CLOB);         // fmac d += a * a
        if constexpr (F == 6) asm volatile(REP16(MUL2, 20, 21, 0) ::: CLOB);         // two sources
        if constexpr (F == 7) asm volatile(REP16(MULS, 0, 21, 0) ::: CLOB);          // SGPR x VGPR
        if constexpr (F == 8) asm volatile(REP16(SUBS, 0, 21, 0) ::: CLOB);
        if constexpr (F == 9) asm volatile(REP16(FMAS, 0, 21, 22) ::: CLOB);         // SGPR, two VGPR
        if constexpr (F == 10) asm volatile(REP16(FMAK, 20, 21, 0) ::: CLOB);        // literal addend
        if constexpr (F == 11) asm volatile(REP16(MED3, 20, 21, 22) ::: CLOB);
        if constexpr (F == 12) asm volatile(REP16(RSQ, 20, 0, 0) ::: CLOB);
        if constexpr (F == 13) asm volatile(REP16(FMAC_DPP, 20, 20, 0) ::: CLOB);   // square through the DPP path
        if constexpr (F == 14) asm volatile(REP16(FMAC_DPP, 20, 21, 0) ::: CLOB);
        if constexpr (F == 15) asm volatile(REP16(MUL_DPP, 20, 20, 0) ::: CLOB);
        if constexpr (F == 16) asm volatile(REP16(FMA_SQ3, 20, 0, 22) ::: CLOB);     // VOP3 a*a + c, c in another bank
        if constexpr (F == 17) asm volatile(REP16(FMAK_SQ, 20, 0, 0) ::: CLOB);
        if constexpr (F == 18) asm volatile(REP16(MUL_SDWA, 20, 20, 0) ::: CLOB);
        if constexpr (F == 19) asm volatile(REP16(FMAC, 20, 24, 0) ::: CLOB);       // fmac, sources in one bank, different registers
    }
    float r;
    asm volatile("v_add_f32 %0, v40, v55" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float* out; CK(hipMalloc(&out, (size_t)prop.multiProcessorCount * 4 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    const char* names[] = {"v_fma_f32 d,a,b,c   banks 0 1 2", "v_fma_f32 d,a,b,c   banks 0 0 0", "v_fma_f32 d,a,b,c   banks 0 1 1", "v_fma_f32 d,a,b,d (VOP3 acc)",
                           "v_fmac_f32 d,a,b (VOP2)", "v_fmac_f32 d,a,a", "v_mul_f32 d,a,b", "v_mul_f32 d,s,b", "v_sub_f32 d,s,b", "v_fma_f32 d,s,b,c",
                           "v_fmaak_f32 d,a,b,lit", "v_med3_f32 d,a,b,c", "v_rsq_f32 d,a", "v_fmac_f32_dpp d,a,a (identity)", "v_fmac_f32_dpp d,a,b (identity)",
                           "v_mul_f32_dpp d,a,a (identity)", "v_fma_f32 d,a,a,c", "v_fmaak_f32 d,a,a,lit", "v_mul_f32_sdwa d,a,a", "v_fmac_f32 d,a,b  banks 0 0"};
    for (int wps : {4, 3})
        for (int f = 0; f < 20; ++f) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                const dim3 g(prop.multiProcessorCount * wps), b(256);
                switch (f) {
                    case 0: k<0><<<g, b>>>(out, iters); break; case 1: k<1><<<g, b>>>(out, iters); break; case 2: k<2><<<g, b>>>(out, iters); break;
                    case 3: k<3><<<g, b>>>(out, iters); break; case 4: k<4><<<g, b>>>(out, iters); break; case 5: k<5><<<g, b>>>(out, iters); break;
                    case 6: k<6><<<g, b>>>(out, iters); break; case 7: k<7><<<g, b>>>(out, iters); break; case 8: k<8><<<g, b>>>(out, iters); break;
                    case 9: k<9><<<g, b>>>(out, iters); break; case 10: k<10><<<g, b>>>(out, iters); break; case 11: k<11><<<g, b>>>(out, iters); break;
                    case 12: k<12><<<g, b>>>(out, iters); break; case 13: k<13><<<g, b>>>(out, iters); break; case 14: k<14><<<g, b>>>(out, iters); break;
                    case 15: k<15><<<g, b>>>(out, iters); break; case 16: k<16><<<g, b>>>(out, iters); break; case 17: k<17><<<g, b>>>(out, iters); break;
                    case 18: k<18><<<g, b>>>(out, iters); break;
                    default: k<19><<<g, b>>>(out, iters); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double n = (double)iters * 16 * wps;
            printf("%d wave(s)/SIMD  %-34s %.2f ns per instruction per SIMD (%.2f cycles @ 2.4 GHz)\n", wps, names[f], best * 1e6 / n, 2.4 * best * 1e6 / n);
        }
    return 0;
}
