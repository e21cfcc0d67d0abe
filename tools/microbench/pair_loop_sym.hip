// pair_loop_sym.hip — VERDICT round 5, item 3: "evaluate each pair once inside the multi-step kernel — microbenchmark first".
//
// The shipped compute loop of k_cluster<4, 4, 2, 3, true> (chr1_500kb: 12 compute waves x 4 rows, 7 column slots, every pair seen from
// both of its rows: tile_pair_sums_pk of c3d_step_core.h, unchanged) against a SYMMETRIC loop at the same geometry, bare, on every CU:
//
//   window     row i owns the pairs (i, j), j = i + 1 .. i + N/2 (cyclically): a wave's four rows need the 4-aligned window of 256
//              columns that starts at its first row (N <= 504), ONE column block of four columns per lane instead of 7 slots —
//              8 packed pair terms per pass instead of 14; the coordinate arrays are doubled in LDS so that the window never wraps;
//   per term   the shipped packed pair term + a 0/1 mask on the force coefficient (the window's triangular ends: the wave's own 4 x 4
//              diagonal block, and the columns beyond i + N/2) + the COLUMN side: three more packed fma into per-column accumulators
//              (the two halves of a packed accumulator are the wave's two rows of a pair: summed after the loop);
//   columns    a lane ends with 4 columns x 3 components summed over its wave's 4 rows: 3 ds_write_b128 into the wave's slab of LDS,
//              a barrier, and a deterministic cross-wave sum (each of 900 (column, component) outputs adds the <= 12 slabs that cover
//              it in wave order: the canonical order a bit-identical k_step would have to follow), a second barrier.
// What the microbenchmark CANNOT show and the step would pay on top: the part's 300 column sums have to reach the parts that own those
// rows before H0 can integrate — a second trip through the XCD's L2 per step, 0.65-0.77 us bare (profiles/r02_xcd_handoff_microbench.txt,
// r03_cluster_stamps_timeline.txt) — where the shipped step has one.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-kernarg-preload-count=16 -o pair_loop_sym pair_loop_sym.hip && ./pair_loop_sym
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../chromosome3d_amd/csrc/c3d_step_core.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace c3d;

#pragma clang fp contract(off)
// pair_term2 of c3d_step_core.h with the mask and the column side added (same operations, same order on the row side)
template <int SEL>
__device__ __forceinline__ void pair_term2_sym(const PairK2& k, float2v v2, float2v mw2, float2v mask2, float2v xi2, float2v yi2, float2v zi2, float2v xjp,
                                               float2v yjp, float2v zjp, float2v& fx2, float2v& fy2, float2v& fz2, float2v& cx2, float2v& cy2, float2v& cz2) {
    const float xc = SEL ? xjp.y : xjp.x, yc = SEL ? yjp.y : yjp.x, zc = SEL ? zjp.y : zjp.x;
    const float2v dx = xi2 - float2v{xc, xc}, dy = yi2 - float2v{yc, yc}, dz = zi2 - float2v{zc, zc};
    float2v r2 = __builtin_elementwise_fma(dz, dz, float2v{k.k0.x, k.k0.x});
    r2 = __builtin_elementwise_fma(dy, dy, r2);
    r2 = __builtin_elementwise_fma(dx, dx, r2);
    const float2v rinv = float2v{__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
    float2v q01;
    asm("v_pk_fma_f32 %0, %1, %2, 1.0 op_sel:[0,1,0] op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(q01) : "v"(r2), "v"(k.k0));
    const float2v d = r2 * rinv;
    const float2v dl = __builtin_elementwise_fma(d, mw2, -v2);
    const float2v w = float2v{__builtin_amdgcn_rcpf(fabsf(dl.x)), __builtin_amdgcn_rcpf(fabsf(dl.y))};
    const float2v lo = -((w * w) * w);
    const float2v g = float2v{__builtin_amdgcn_fmed3f(dl.x, lo.x, k.k1.y), __builtin_amdgcn_fmed3f(dl.y, lo.y, k.k1.y)};
    const float2v rep = float2v{k.k1.x, k.k1.x} * q01;
    const float2v c = __builtin_elementwise_fma(g, rinv, rep) * mask2;
    fx2 = __builtin_elementwise_fma(c, dx, fx2);
    fy2 = __builtin_elementwise_fma(c, dy, fy2);
    fz2 = __builtin_elementwise_fma(c, dz, fz2);
    cx2 = __builtin_elementwise_fma(-c, dx, cx2);
    cy2 = __builtin_elementwise_fma(-c, dy, cy2);
    cz2 = __builtin_elementwise_fma(-c, dz, cz2);
}
#pragma clang fp contract(fast)

constexpr int NPAD = 512, RPW = 4, NBS = 2, WLS = 3, CW = 12, NH = 4;

// MODE 0: the shipped loop.  1: the symmetric loop, row sums only (column accumulators kept alive, not reduced).  2: + the column side's
// slabs, barriers and the deterministic cross-wave sum.
template <int MODE>
__global__ __launch_bounds__(1024) void k_loop(const float* __restrict__ in, float* __restrict__ out, int iters, DevModel m, DevStep p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem; float* ys = smem + 2 * NPAD; float* zs = smem + 4 * NPAD;          // doubled arrays: index i and i + NPAD hold the same bead
    float* fbuf = smem + 6 * NPAD;                                                        // [3][64] row sums
    float* colsum = fbuf + 256;                                                           // [3][320] the part's column sums
    float* slab = colsum + 3 * 320;                                                       // [CW][3][256] column partials of the compute waves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = tid; b < 6 * NPAD; b += blockDim.x) smem[b] = in[b & 511] * (1.0f + 0.01f * (b / (2 * NPAD))) + 0.003f * (b & 511);
    const bool is_compute = wave >= NH;
    const int cwave = wave - NH, row0 = cwave * RPW;
    PairConsts2<RPW, NBS> pc;
    float2v pcs[2][4], mk[2][4];
    {
        float4 traw[RPW][NBS];
        for (int r = 0; r < RPW; ++r) for (int jb = 0; jb < NBS; ++jb) traw[r][jb] = make_float4(3.0f + in[(tid + r) & 511], 5.0f, 7.0f, 9.0f + in[(tid + jb) & 511]);
        pair_consts2_build<RPW, NBS>(m, traw, false, lane, pc, nullptr);
        for (int q = 0; q < 2; ++q)
            for (int c = 0; c < 4; ++c) {
                pcs[q][c] = pc.p[q][0][c];
                // window mask: row (row0 + 2q + h) owns column j = row0 + 4 lane + c iff 0 < j - row <= 227
                const int j = 4 * lane + c, ra = 2 * q, rb = 2 * q + 1;
                mk[q][c] = float2v{(j - ra > 0 && j - ra <= 227) ? 1.0f : 0.0f, (j - rb > 0 && j - rb <= 227) ? 1.0f : 0.0f};
            }
    }
    // the constants live in registers for the whole launch, as in the kernel (left alone, the compiler rebuilds them from `in` in every pass)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            asm volatile("" : "+v"(pcs[q][c]), "+v"(mk[q][c]));
#pragma unroll
            for (int jb = 0; jb < NBS; ++jb) asm volatile("" : "+v"(pc.p[q][jb][c]));
        }
    __syncthreads();
    float acc = 0.0f;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (is_compute) {
            float Fx, Fy, Fz;
            if constexpr (MODE == 0) {
                tile_pair_sums_pk<RPW, NBS, WLS>(m, p, pc, nullptr, xs, ys, zs, row0, lane, Fx, Fy, Fz);
            } else {
                const PairK2 k2 = pair_k2(m, p);
                float2v on2 = float2v{m.inv_rs, m.inv_rs};
                asm volatile("" : "+v"(on2));
                float2v fx2[2], fy2[2], fz2[2], xi2[2], yi2[2], zi2[2], cx2[4], cy2[4], cz2[4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    xi2[q] = *reinterpret_cast<const float2v*>(xs + row0 + 2 * q); yi2[q] = *reinterpret_cast<const float2v*>(ys + row0 + 2 * q);
                    zi2[q] = *reinterpret_cast<const float2v*>(zs + row0 + 2 * q);
                    fx2[q] = fy2[q] = fz2[q] = float2v{0.0f, 0.0f};
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) cx2[c] = cy2[c] = cz2[c] = float2v{0.0f, 0.0f};
                const int j = row0 + 4 * lane;                           // the wave's window: 4-aligned, inside the doubled arrays
                const float4 xj = *reinterpret_cast<const float4*>(xs + j), yj = *reinterpret_cast<const float4*>(ys + j), zj = *reinterpret_cast<const float4*>(zs + j);
                const float2v x01 = float2v{xj.x, xj.y}, x23 = float2v{xj.z, xj.w}, y01 = float2v{yj.x, yj.y}, y23 = float2v{yj.z, yj.w};
                const float2v z01 = float2v{zj.x, zj.y}, z23 = float2v{zj.z, zj.w};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    pair_term2_sym<0>(k2, pcs[q][0], on2, mk[q][0], xi2[q], yi2[q], zi2[q], x01, y01, z01, fx2[q], fy2[q], fz2[q], cx2[0], cy2[0], cz2[0]);
                    pair_term2_sym<1>(k2, pcs[q][1], on2, mk[q][1], xi2[q], yi2[q], zi2[q], x01, y01, z01, fx2[q], fy2[q], fz2[q], cx2[1], cy2[1], cz2[1]);
                    pair_term2_sym<0>(k2, pcs[q][2], on2, mk[q][2], xi2[q], yi2[q], zi2[q], x23, y23, z23, fx2[q], fy2[q], fz2[q], cx2[2], cy2[2], cz2[2]);
                    pair_term2_sym<1>(k2, pcs[q][3], on2, mk[q][3], xi2[q], yi2[q], zi2[q], x23, y23, z23, fx2[q], fy2[q], fz2[q], cx2[3], cy2[3], cz2[3]);
                    asm volatile("" : "+v"(fx2[q]), "+v"(fy2[q]), "+v"(fz2[q]));
                }
                float fx[RPW], fy[RPW], fz[RPW];
#pragma unroll
                for (int q = 0; q < 2; ++q) { fx[2 * q] = fx2[q].x; fx[2 * q + 1] = fx2[q].y; fy[2 * q] = fy2[q].x; fy[2 * q + 1] = fy2[q].y; fz[2 * q] = fz2[q].x; fz[2 * q + 1] = fz2[q].y; }
                Fx = reduce_rows<RPW>(fx, lane); Fy = reduce_rows<RPW>(fy, lane); Fz = reduce_rows<RPW>(fz, lane);
                // column side: halves of a packed accumulator = the two rows of a pair
                const float4 sx = make_float4(cx2[0].x + cx2[0].y, cx2[1].x + cx2[1].y, cx2[2].x + cx2[2].y, cx2[3].x + cx2[3].y);
                const float4 sy = make_float4(cy2[0].x + cy2[0].y, cy2[1].x + cy2[1].y, cy2[2].x + cy2[2].y, cy2[3].x + cy2[3].y);
                const float4 sz = make_float4(cz2[0].x + cz2[0].y, cz2[1].x + cz2[1].y, cz2[2].x + cz2[2].y, cz2[3].x + cz2[3].y);
                if constexpr (MODE == 2) {
                    float4* sl = reinterpret_cast<float4*>(slab + (size_t)cwave * 3 * 256);
                    sl[lane] = sx; sl[64 + lane] = sy; sl[128 + lane] = sz;
                } else {
                    acc += sx.x + sy.y + sz.z + sx.w;
                }
            }
            if (lane < RPW) { const int k = cwave * RPW + lane; fbuf[k] = Fx; fbuf[64 + k] = Fy; fbuf[128 + k] = Fz; }
            acc += Fx;
        }
        if constexpr (MODE == 2) {
            __syncthreads();
            // the part's window of columns: [0, 44 + 256) relative to its first row; output u = (component, column)
            if (tid < 900) {
                const int comp = tid / 300, col = tid - 300 * comp;
                float s = 0.0f;
#pragma unroll
                for (int w = 0; w < CW; ++w) {
                    const int k = col - 4 * w;
                    if (k >= 0 && k < 256) s += slab[((size_t)w * 3 + comp) * 256 + k];
                }
                colsum[comp * 320 + col] = s;
            }
            __syncthreads();
            if (tid < 48) xs[tid] += 1e-4f * (fbuf[tid] + colsum[tid]);
        } else {
            if (tid < 48) xs[tid] += 1e-4f * fbuf[tid];
        }
    }
    out[blockIdx.x * 1024 + tid] = acc;
}

template <int MODE>
static float run(int grid, int lds, const float* in, float* out, int iters, DevModel m, DevStep p, hipEvent_t e0, hipEvent_t e1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        k_loop<MODE><<<grid, (CW + NH) * 64, lds>>>(in, out, iters, m, p);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    float *in, *out; CK(hipMalloc(&in, 2048)); CK(hipMalloc(&out, 4 * 1024 * prop.multiProcessorCount));
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = 0.37f * (i % 29) - 4.0f + 0.01f * i;
    CK(hipMemcpy(in, h, 2048, hipMemcpyHostToDevice));
    DevModel m{}; m.n = 455; m.npad = 512; m.noe_pot = 4; m.mrs = 10.0f; m.nmrs = -10.0f; m.rs = 0.5f; m.inv_rs = 0.1f; m.nm_rs = 0.05f; m.wl = 3; m.nleft = 7; m.jl0 = 448;
    DevStep p{}; p.kind = 1; p.w_noe2n = -20.0f; p.inv_rep_r2 = 1.0f / 21.0f; p.w_rep4r2 = 4.0f * 21.0f; p.w_rs = -200.0f; p.kq = -0.42f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 100 * 1024, iters = 2000, grid = prop.multiProcessorCount;
    const float a = run<0>(grid, lds, in, out, iters, m, p, e0, e1), b = run<1>(grid, lds, in, out, iters, m, p, e0, e1), c = run<2>(grid, lds, in, out, iters, m, p, e0, e1);
    printf("geometry: 12 compute waves x 4 rows + 4 idle waves per CU, every CU, %d passes, best of 5 launches (one pass = the pair loop of one step)\n", iters);
    printf("shipped loop   tile_pair_sums_pk<4, 2, 3>: 14 packed pair terms per pass, every pair from both rows     %.3f us per pass\n", a * 1e3 / iters);
    printf("symmetric loop 8 packed pair terms per pass + mask + column accumulators, row sums only                %.3f us per pass\n", b * 1e3 / iters);
    printf("symmetric loop + column slabs in LDS, 2 barriers, deterministic cross-wave sum of 900 column sums      %.3f us per pass\n", c * 1e3 / iters);
    printf("difference shipped - symmetric (all in): %.3f us per step; still to pay in a step: a second L2 hand-off (~0.65-0.77 us bare)\n", (a - c) * 1e3 / iters);
    return 0;
}
