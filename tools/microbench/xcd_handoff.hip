// xcd_handoff.hip — price of the per-step all-gather between the workgroups that own one replica when those
// workgroups are FEW and LARGE (P workgroups of 1024 or 512 threads, one per CU) and, optionally, all on ONE XCD.
//
// Geometry of the cluster kernel (c3d_cluster.hip): 256 workgroups, one per CU.  A workgroup reads HW_REG_XCC_ID, takes a
// slot from that XCD's counter and becomes part `slot % P` of cluster (replica) `slot / P` of that XCD, so every
// cluster lives inside one XCD whatever the dispatcher did (placement = 1).  placement = 0 is the naive
// blockIdx -> (cluster, part) map that spreads a cluster over the XCDs.
// Every step a workgroup publishes a record of U 16-byte units {tag, v, tag, v} and gathers the P records of its
// cluster, re-reading with L1-bypassing (sc1) loads until every tag matches.
//   store mode 0: write-through `sc1` stores (valid for any placement)
//   store mode 1: plain stores (line stays in the XCD's L2; valid only when the readers share the XCD)
// Prints microseconds per step, timeouts and the number of payload words with an unexpected value (must be 0).
//   hipcc --offload-arch=gfx950 -O3 xcd_handoff.hip -o xcd_handoff && ./xcd_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Params {
    int P;          // workgroups per cluster
    int U;          // 16-byte units per record
    int ncl;        // clusters in all
    int nsteps, work, placement, store_mode, poll_sleep;
};

__device__ __forceinline__ unsigned expect_val(unsigned step, int cl, int part, int k) {
    return step * 2654435761u + cl * 7919u + part * 40503u + k * 97u;
}

template <int KU>
__global__ __launch_bounds__(1024) void k_handoff(u32x4* __restrict__ rec, Params p, unsigned* __restrict__ claim,
                                                  unsigned* __restrict__ tmo, unsigned long long* __restrict__ ticks,
                                                  unsigned* __restrict__ errs, unsigned* __restrict__ xcd_of_cluster) {
    __shared__ float vals[4096];
    __shared__ int s_slot;
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
    int cl, part;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
    if (p.placement) {
        if (tid == 0) s_slot = (int)atomicAdd(&claim[xcc], 1u);
        __syncthreads();
        const int slot = s_slot;
        const int k = slot / p.P;                  // cluster index inside this XCD
        part = slot % p.P;
        cl = (int)xcc + 8 * k;                     // clusters are dealt to XCDs round-robin
        if (cl >= p.ncl) return;                   // this CU has nothing to do
    } else {
        if ((int)blockIdx.x >= p.ncl * p.P) return;
        cl = blockIdx.x % p.ncl;
        part = blockIdx.x / p.ncl;
    }
    if (tid == 0 && part == 0) xcd_of_cluster[cl] = xcc;
    const int units = p.P * p.U;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rec, 0, (int)(sizeof(u32x4) * 2 * p.ncl * units), 0x00020000);
    unsigned nerr = 0;
    float acc = (float)tid;
    unsigned long long t0 = 0;
    for (int s = 0; s < p.nsteps; ++s) {
        const unsigned tag = (unsigned)s + 1u;
        if (s == 1 && tid == 0) t0 = wall_clock64();
        if (s > 0) {
            const int base = ((s & 1) * p.ncl + cl) * units;
            u32x4 v[KU];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < KU; ++k) {
                    const int u = min(tid + nthr * k, units - 1);
                    v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (base + u) * 16, 0, 16);   // sc1: bypass L1
                }
#pragma unroll
                for (int k = 0; k < KU; ++k) ok &= v[k].x == tag && v[k].z == tag;
                if (__all(ok)) break;
                if (p.poll_sleep) __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 16)) { if (lane == 0) atomicAdd(tmo, 1u); return; }
            }
#pragma unroll
            for (int k = 0; k < KU; ++k) {
                const int u = tid + nthr * k;
                if (u < units) {
                    const int r = u / p.U, q = u % p.U;
                    if (v[k].y != expect_val(tag, cl, r, 2 * q) || v[k].w != expect_val(tag, cl, r, 2 * q + 1)) ++nerr;
                    vals[(2 * u) & 4095] = __uint_as_float(v[k].y);
                    vals[(2 * u + 1) & 4095] = __uint_as_float(v[k].w);
                }
            }
        }
        __syncthreads();
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = acc + vals[(tid * 7 + k) & 4095];
        for (int w = 0; w < p.work; w += 32) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = fmaf(a[k], 1.0000001f, 0.5f);
        }
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
        // publish the record tagged s+2 (read at step s+1) into the other parity buffer: threads 0..U-1
        if (tid < p.U) {
            const unsigned nt = tag + 1u;
            u32x4 o;
            o.x = nt; o.y = expect_val(nt, cl, part, 2 * tid); o.z = nt; o.w = expect_val(nt, cl, part, 2 * tid + 1);
            const int base = (((s + 1) & 1) * p.ncl + cl) * units;
            if (p.store_mode == 0) __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (base + part * p.U + tid) * 16, 0, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (base + part * p.U + tid) * 16, 0, 0);
        }
    }
    if (tid == 0 && part == 0) ticks[cl] = wall_clock64() - t0;
    if (nerr) atomicAdd(errs, nerr);
    if (acc == 12345.678f) errs[1] = 1;
}

int main(int argc, char** argv) {
    int ncl = 20, threads = 1024;
    if (argc > 1) ncl = atoi(argv[1]);
    if (argc > 2) threads = atoi(argv[2]);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const int per_xcd_cl = (ncl + 7) / 8;                // clusters on the fullest XCD
    const int Pmax = (ncu / 8) / per_xcd_cl;             // workgroups per cluster there
    unsigned* claim; unsigned* tmo; unsigned long long* ticks; unsigned* errs; unsigned* xcdc;
    CK(hipMalloc(&claim, 64)); CK(hipMalloc(&tmo, 16)); CK(hipMalloc(&ticks, sizeof(unsigned long long) * ncl));
    CK(hipMalloc(&errs, 16)); CK(hipMalloc(&xcdc, 4 * ncl));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d CUs, %d clusters, up to %d per XCD -> P = %d workgroups of %d threads per cluster\n", ncu, ncl, per_xcd_cl, Pmax, threads);
    const int Ps[] = {Pmax, 8, 4};
    for (int P : Ps) {
        if (P > Pmax) continue;
        const int rows = (455 + P - 1) / P;
        const int U = (3 * rows + 4 * ((rows + 7) / 8) + 1) / 2;     // positions + per-8-row partial sums, 2 values per unit
        const int units = P * U;
        const int KU = (units + threads - 1) / threads;
        if (KU > 4) { printf("P %d: KU %d > 4, skipped\n", P, KU); continue; }
        const size_t nunits = (size_t)2 * ncl * units;
        u32x4* rec; CK(hipMalloc(&rec, nunits * sizeof(u32x4)));
        const int work_real = (rows + threads / 64 - 1) / (threads / 64) * 8 * 19;   // pair loop: rows per wave x 8 columns x 19 VALU
        const int works[] = {0, work_real};
        for (int placement = 1; placement >= 0; --placement)
            for (int mode = 0; mode < 2; ++mode) {
                if (!placement && mode == 1) continue;       // plain stores are only valid inside one XCD
                for (int sl = 1; sl >= 0; --sl)
                for (int w : works) {
                    Params p{P, U, ncl, 2000, w, placement, mode, sl};
                    CK(hipMemsetAsync(rec, 0, nunits * sizeof(u32x4), 0));
                    CK(hipMemsetAsync(claim, 0, 64, 0)); CK(hipMemsetAsync(tmo, 0, 16, 0)); CK(hipMemsetAsync(errs, 0, 16, 0));
                    CK(hipMemsetAsync(xcdc, 0xff, 4 * ncl, 0));
                    CK(hipEventRecord(e0, 0));
                    const int grid = placement ? ncu : ncl * P;
                    switch (KU) {
                        case 1: hipLaunchKernelGGL(k_handoff<1>, dim3(grid), dim3(threads), 0, 0, rec, p, claim, tmo, ticks, errs, xcdc); break;
                        case 2: hipLaunchKernelGGL(k_handoff<2>, dim3(grid), dim3(threads), 0, 0, rec, p, claim, tmo, ticks, errs, xcdc); break;
                        case 3: hipLaunchKernelGGL(k_handoff<3>, dim3(grid), dim3(threads), 0, 0, rec, p, claim, tmo, ticks, errs, xcdc); break;
                        default: hipLaunchKernelGGL(k_handoff<4>, dim3(grid), dim3(threads), 0, 0, rec, p, claim, tmo, ticks, errs, xcdc); break;
                    }
                    CK(hipGetLastError());
                    CK(hipEventRecord(e1, 0));
                    CK(hipDeviceSynchronize());
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned h_tmo = 0, h_err[2] = {0, 0}, h_claim[8];
                    std::vector<unsigned long long> h_t(ncl);
                    CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_err, errs, 8, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(h_t.data(), ticks, sizeof(unsigned long long) * ncl, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(h_claim, claim, 32, hipMemcpyDeviceToHost));
                    unsigned long long worst = 0; for (auto t : h_t) worst = t > worst ? t : worst;
                    printf("P %2d U %3d KU %d placement %s store %s sleep %d work %4d: %.3f us/step (event) %.3f (slowest cluster), timeouts %u, bad words %u",
                           P, U, KU, placement ? "xcd " : "flat", mode ? "plain" : "sc1  ", sl, w, 1e3 * ms / p.nsteps,
                           (double)worst * 0.01 / (p.nsteps - 1), h_tmo, h_err[0]);
                    if (placement) { printf("  wgs/xcd"); for (int k = 0; k < 8; ++k) printf(" %u", h_claim[k]); }
                    printf("\n");
                }
            }
        CK(hipFree(rec));
    }
    return 0;
}
