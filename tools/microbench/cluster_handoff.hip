// cluster_handoff.hip — price of an in-launch all-gather between the workgroups that own one replica,
// in the geometry a resident (multi-step) anneal kernel would have: NCL clusters (replicas) x NT
// workgroups (row tiles) of 256 threads, every step each workgroup publishes a 256-byte record
// (28 values as {tag, value} granules, written write-through by ONE store instruction of wave 0) and
// then gathers all NT records of its cluster (sc1 loads, re-read until every tag matches).
// Prints microseconds per step for several amounts of fake VALU work between gather and publish, and
// the number of payload words that did not have the expected value (must be 0).
//   hipcc --offload-arch=gfx950 -O3 cluster_handoff.hip -o cluster_handoff && ./cluster_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kUnits = 16;   // 16-byte units per record (2 granules each)

struct Params { int ncl, nt, nsteps, work, stagger; };   // work = VALU instructions per lane per step; stagger = start offset (64-cycle sleeps) x (cluster % 4)

__device__ __forceinline__ unsigned expect_val(unsigned step, int tile, int k) { return step * 2654435761u + tile * 40503u + k * 97u; }

__global__ __launch_bounds__(256) void k_handoff(u32x4* __restrict__ rec, Params p, unsigned* __restrict__ tmo,
                                                 unsigned long long* __restrict__ ticks, unsigned* __restrict__ errs) {
    __shared__ float vals[64 * 32];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = b % p.ncl, tile = b / p.ncl;
    const int units = p.nt * kUnits;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rec, 0, (int)(sizeof(u32x4) * 2 * p.ncl * units), 0x00020000);
    unsigned nerr = 0;
    float acc = (float)tid;
    unsigned long long t0 = 0;
    for (int d = 0; d < p.stagger * (cl & 3); ++d) __builtin_amdgcn_s_sleep(1);
    for (int s = 0; s < p.nsteps; ++s) {
        const unsigned tag = (unsigned)s + 1u;
        if (s == 1 && tid == 0) t0 = wall_clock64();
        if (s > 0) {
            // gather: every thread owns units tid, tid+256, ... of the cluster's region for this parity
            const int base = ((s & 1) * p.ncl + cl) * units;
            u32x4 v[4];
            bool ok;
            unsigned spins = 0;
            do {
                ok = true;
                asm volatile("" ::: "memory");   // re-issue the loads on every sweep
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int u = tid + 256 * k;
                    if (u < units) {
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (base + u) * 16, 0, 16);   // aux 16 = sc1
                        ok &= v[k].x == tag && v[k].z == tag;
                    }
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 16)) { if (lane == 0) atomicAdd(tmo, 1u); return; }
            } while (true);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = tid + 256 * k;
                if (u < units) {
                    const int r = u / kUnits, q = u % kUnits;
                    if (v[k].y != expect_val(tag, r, 2 * q) || v[k].w != expect_val(tag, r, 2 * q + 1)) ++nerr;
                    vals[r * 32 + 2 * q] = __uint_as_float(v[k].y);
                    vals[r * 32 + 2 * q + 1] = __uint_as_float(v[k].w);
                }
            }
        }
        __syncthreads();
        // fake pair loop: `work` dependent fmas per lane
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = acc + vals[(tid * 7 + k) & 2047];
        for (int w = 0; w < p.work; w += 32) {   // 32 independent-ish fmas per trip (8 chains x 4)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = fmaf(a[k], 1.0000001f, 0.5f);
        }
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
        // publish the record tagged s+2 (read at step s+1) into the other parity buffer
        if (wave == 0 && lane < kUnits) {
            const unsigned nt = tag + 1u;
            u32x4 o;
            o.x = nt; o.y = expect_val(nt, tile, 2 * lane); o.z = nt; o.w = expect_val(nt, tile, 2 * lane + 1);
            const int base = (((s + 1) & 1) * p.ncl + cl) * units;
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, (base + tile * kUnits + lane) * 16, 0, 16);
        }
    }
    if (tid == 0 && tile == 0) ticks[cl] = wall_clock64() - t0;
    if (nerr) atomicAdd(errs, nerr);
    if (acc == 12345.678f) errs[1] = 1;   // keep the fake work alive
}

int main(int argc, char** argv) {
    Params p{20, 57, 2000, 0, 0};
    if (argc > 1) p.ncl = atoi(argv[1]);
    if (argc > 2) p.nt = atoi(argv[2]);
    if (p.nt > 64) { fprintf(stderr, "nt <= 64\n"); return 1; }
    const size_t nunits = (size_t)2 * p.ncl * p.nt * kUnits;
    u32x4* rec; unsigned* tmo; unsigned long long* ticks; unsigned* errs;
    CK(hipMalloc(&rec, nunits * sizeof(u32x4)));
    CK(hipMalloc(&tmo, 16)); CK(hipMalloc(&ticks, sizeof(unsigned long long) * p.ncl)); CK(hipMalloc(&errs, 16));
    int maxb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_handoff, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = p.ncl * p.nt;
    printf("grid %d workgroups, occupancy API %d per CU x %d CUs\n", grid, maxb, prop.multiProcessorCount);
    if (grid > (maxb > 8 ? 8 : maxb) * prop.multiProcessorCount) { fprintf(stderr, "grid not co-resident\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int works[] = {0, 256, 448, 640, 1280};
    const int staggers[] = {0, 16, 32, 64};
    for (int st : staggers)
        for (int w : works) {
            p.work = w; p.stagger = st;
            CK(hipMemsetAsync(rec, 0, nunits * sizeof(u32x4), 0));
            CK(hipMemsetAsync(tmo, 0, 16, 0)); CK(hipMemsetAsync(errs, 0, 16, 0));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_handoff, dim3(grid), dim3(256), 0, 0, rec, p, tmo, ticks, errs);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h_tmo = 0, h_err[2] = {0, 0};
            std::vector<unsigned long long> h_t(p.ncl);
            CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_err, errs, 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h_t.data(), ticks, sizeof(unsigned long long) * p.ncl, hipMemcpyDeviceToHost));
            unsigned long long worst = 0; for (auto t : h_t) worst = t > worst ? t : worst;
            printf("stagger %3d work %4d valu/lane: %.3f us/step (event), %.3f us/step (slowest cluster, 100 MHz clock), timeouts %u, bad words %u\n",
                   st, w, 1e3 * ms / p.nsteps, (double)worst * 0.01 / (p.nsteps - 1), h_tmo, h_err[0]);
        }
    return 0;
}
