// Issue cost of the PACKED fp32 forms on gfx950 next to the scalar forms they would replace in the pair term: does a packed
// square (src0 = src1) pay the bank-conflict cycle that v_fmac_f32 d,a,a pays?  does the op_sel broadcast cost anything?
// 8 independent packed instructions per block (destinations v[40:41] .. v[54:55]), explicit registers, 4 and 3 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_packed valu_packed.hip && ./valu_packed
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP8(OP) OP(40) OP(42) OP(44) OP(46) OP(48) OP(50) OP(52) OP(54)
#define REP16S(OP) OP(40) OP(41) OP(42) OP(43) OP(44) OP(45) OP(46) OP(47) OP(48) OP(49) OP(50) OP(51) OP(52) OP(53) OP(54) OP(55)
#define S1(x) #x
#define S(x) S1(x)
#define P(D) "v[" S(D) ":" S1(D+1) "]"
#define PKFMA(D) "v_pk_fma_f32 v[" #D ":" #D "+1], v[20:21], v[22:23], v[24:25]\n"
#define PKFMA_SQ(D) "v_pk_fma_f32 v[" #D ":" #D "+1], v[20:21], v[20:21], v[24:25]\n"
#define PKFMA_ACC_SQ(D) "v_pk_fma_f32 v[" #D ":" #D "+1], v[20:21], v[20:21], v[" #D ":" #D "+1]\n"
#define PKMUL(D) "v_pk_mul_f32 v[" #D ":" #D "+1], v[20:21], v[22:23]\n"
#define PKMUL_SQ(D) "v_pk_mul_f32 v[" #D ":" #D "+1], v[20:21], v[20:21]\n"
#define PKADD(D) "v_pk_add_f32 v[" #D ":" #D "+1], v[20:21], v[22:23]\n"
#define PKSUB_BC(D) "v_pk_add_f32 v[" #D ":" #D "+1], v[20:21], v[22:23] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n"
#define RCP(D) "v_rcp_f32_e32 v" #D ", v20\n"
#define RSQ(D) "v_rsq_f32_e32 v" #D ", v20\n"
#define RCPABS(D) "v_rcp_f32_e64 v" #D ", |v20|\n"
#define FMACSQ(D) "v_fmac_f32_e32 v" #D ", v20, v20\n"
#define MULSQ(D) "v_mul_f32_e32 v" #D ", v20, v20\n"
#define MULNEG(D) "v_mul_f32_e64 v" #D ", -v20, v21\n"
#define FMACLAMP(D) "v_fma_f32 v" #D ", -v20, v21, 1.0 clamp\n"
#define MED3K(D) "v_med3_f32 v" #D ", v20, v21, v22\n"
#define CLOB "v20", "v21", "v22", "v23", "v24", "v25", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"

template <int F> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    asm volatile("v_mov_b32 v20, 1.0\nv_mov_b32 v21, 0.5\nv_mov_b32 v22, 2.0\nv_mov_b32 v23, 1.0\nv_mov_b32 v24, 0.5\nv_mov_b32 v25, 1.0\n" ::: CLOB);
    for (int i = 0; i < iters; ++i) {
        if constexpr (F == 0) asm volatile(REP8(PKFMA) REP8(PKFMA) ::: CLOB);
        if constexpr (F == 1) asm volatile(REP8(PKFMA_SQ) REP8(PKFMA_SQ) ::: CLOB);
        if constexpr (F == 2) asm volatile(REP8(PKFMA_ACC_SQ) REP8(PKFMA_ACC_SQ) ::: CLOB);
        if constexpr (F == 3) asm volatile(REP8(PKMUL) REP8(PKMUL) ::: CLOB);
        if constexpr (F == 4) asm volatile(REP8(PKMUL_SQ) REP8(PKMUL_SQ) ::: CLOB);
        if constexpr (F == 5) asm volatile(REP8(PKADD) REP8(PKADD) ::: CLOB);
        if constexpr (F == 6) asm volatile(REP8(PKSUB_BC) REP8(PKSUB_BC) ::: CLOB);
        if constexpr (F == 7) asm volatile(REP16S(RCP) ::: CLOB);
        if constexpr (F == 8) asm volatile(REP16S(RSQ) ::: CLOB);
        if constexpr (F == 9) asm volatile(REP16S(RCPABS) ::: CLOB);
        if constexpr (F == 10) asm volatile(REP16S(FMACSQ) ::: CLOB);
        if constexpr (F == 11) asm volatile(REP16S(MULSQ) ::: CLOB);
        if constexpr (F == 12) asm volatile(REP16S(MULNEG) ::: CLOB);
        if constexpr (F == 13) asm volatile(REP16S(FMACLAMP) ::: CLOB);
        if constexpr (F == 14) asm volatile(REP16S(MED3K) ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v40, v55" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int F> float run(int g, float* out, int iters, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<F><<<dim3(g), dim3(256)>>>(out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float* out; CK(hipMalloc(&out, (size_t)prop.multiProcessorCount * 4 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    const char* names[] = {"v_pk_fma_f32 d,a,b,c (distinct pairs)", "v_pk_fma_f32 d,a,a,c", "v_pk_fma_f32 d,a,a,d", "v_pk_mul_f32 d,a,b", "v_pk_mul_f32 d,a,a",
                           "v_pk_add_f32 d,a,b", "v_pk_add_f32 d,a.lo(bcast),-b", "v_rcp_f32 d,a", "v_rsq_f32 d,a", "v_rcp_f32_e64 d,|a|", "v_fmac_f32 d,a,a",
                           "v_mul_f32 d,a,a", "v_mul_f32_e64 d,-a,b", "v_fma_f32 d,-a,b,1.0 clamp", "v_med3_f32 d,a,b,c"};
    const int per_iter[] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};
    for (int wps : {4, 3})
        for (int f = 0; f < 15; ++f) {
            const int g = prop.multiProcessorCount * wps;
            float ms = 0;
            switch (f) {
                case 0: ms = run<0>(g, out, iters, e0, e1); break; case 1: ms = run<1>(g, out, iters, e0, e1); break; case 2: ms = run<2>(g, out, iters, e0, e1); break;
                case 3: ms = run<3>(g, out, iters, e0, e1); break; case 4: ms = run<4>(g, out, iters, e0, e1); break; case 5: ms = run<5>(g, out, iters, e0, e1); break;
                case 6: ms = run<6>(g, out, iters, e0, e1); break; case 7: ms = run<7>(g, out, iters, e0, e1); break; case 8: ms = run<8>(g, out, iters, e0, e1); break;
                case 9: ms = run<9>(g, out, iters, e0, e1); break; case 10: ms = run<10>(g, out, iters, e0, e1); break; case 11: ms = run<11>(g, out, iters, e0, e1); break;
                case 12: ms = run<12>(g, out, iters, e0, e1); break; case 13: ms = run<13>(g, out, iters, e0, e1); break; case 14: ms = run<14>(g, out, iters, e0, e1); break;
            }
            // per SIMD: wps waves, each iters * per_iter instructions
            const double ns = 1e6 * ms / ((double)iters * per_iter[f] * wps);
            printf("%d waves/SIMD  %-40s %6.2f ns per instruction per SIMD\n", wps, names[f], ns);
        }
    return 0;
}
