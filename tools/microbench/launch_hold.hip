// launch_hold.hip — round 6: why the FAST one of two anneals on disjoint XCD halves is held to the SLOW one's pace (tools/xcd_budget_proxy.py,
// scratch probes: a launch of context B dispatched while context A's long launch is resident does not finish before A's does).
//
// The multi-step kernel's launch is one workgroup per CU over ALL XCDs; workgroups that land on an XCD outside the context's set exit at
// once.  Here, bare: kernel A keeps 30 of the 32 CUs of XCDs 0-3 busy for ~5 ms (one 1024-thread workgroup with 84 KB of LDS each, the
// rest of its grid exits at once), and while it runs kernel B is launched on a second stream — same shape, its work (1 ms) on XCDs 4-7,
// its workgroups on XCDs 0-3 exiting at once.  B's duration (events on its stream) says whether its exit-at-once workgroups found a CU:
//   case 1  B alone                                   (reference)
//   case 2  B launched 1 ms after A                   (the pairing of c3d_batch / batch.py: two contexts, two streams)
//   case 3  the same with B's stream CU-masked to XCDs 4-7 (hipExtStreamCreateWithCUMask; KFD deals mask bit i to XCC i % 8)
//   case 4  the same with B's workgroups asking for 8 KB of LDS and 64 threads on the foreign XCDs... not expressible: one shape per launch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o launch_hold launch_hold.hip && ./launch_hold
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// a workgroup on an XCD of [x0, x0 + nx) whose per-XCD slot is below `busy` spins for `ticks` of the 100 MHz clock; every other one returns
__global__ __launch_bounds__(1024) void k_hold(unsigned* claim, int x0, int nx, int busy, unsigned long long ticks, unsigned* out) {
    extern __shared__ float smem[];
    const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0x7);
    __shared__ int slot;
    if (threadIdx.x == 0) slot = (int)atomicAdd(&claim[xcc], 1u);
    __syncthreads();
    if (xcc < x0 || xcc >= x0 + nx || slot >= busy) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float acc = smem[threadIdx.x & 255];
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { acc = acc * 1.0001f + 1.0f; __builtin_amdgcn_s_sleep(8); }
    if (acc == 12345.0f) out[0] = 1;
    if (threadIdx.x == 0) atomicAdd(&out[1], 1u);
}

static float run_b(hipStream_t sb, unsigned* claim_b, unsigned* out, int lds, unsigned long long ticks_b, hipEvent_t e0, hipEvent_t e1) {
    (void)hipMemsetAsync(claim_b, 0, 64, sb);
    (void)hipEventRecord(e0, sb);
    hipLaunchKernelGGL(k_hold, dim3(256), dim3(1024), lds, sb, claim_b, 4, 4, 30, ticks_b, out);
    (void)hipEventRecord(e1, sb);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv) {
    const int busy_a = argc > 1 ? atoi(argv[1]) : 30, lds_a = (argc > 2 ? atoi(argv[2]) : 84) * 1024, lds_b = (argc > 3 ? atoi(argv[3]) : 84) * 1024;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int lds = 84 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hold), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    unsigned *claim_a, *claim_b, *out;
    CK(hipMalloc(&claim_a, 64)); CK(hipMalloc(&claim_b, 64)); CK(hipMalloc(&out, 64));
    CK(hipMemset(out, 0, 64));
    hipStream_t sa, sb, sm;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    // CU mask for XCDs 4-7: bit i of the mask belongs to XCC i % 8 (KFD, GFX 9.4.3+: the mask is dealt to the XCCs bit by bit)
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u);
    const int mask_mode = argc > 4 ? atoi(argv[4]) : 0;   // 0: bit i -> XCC i % 8 (interleaved); 1: contiguous upper half
    for (int i = 0; i < prop.multiProcessorCount; ++i) if (mask_mode == 0 ? (i % 8) >= 4 : i >= prop.multiProcessorCount / 2) mask[i / 32] |= 1u << (i % 32);
    const hipError_t me = hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data());
    hipEvent_t e0, e1, a0, a1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    const unsigned long long ms_ticks = 100000ull;        // 1 ms of the 100 MHz counter
    // warm up both streams
    run_b(sb, claim_b, out, lds, ms_ticks / 10, e0, e1);
    if (me == hipSuccess) run_b(sm, claim_b, out, lds, ms_ticks / 10, e0, e1);
    printf("B alone (1 ms of work on 30 CUs of each of XCDs 4-7, 256 workgroups of 1024 threads + 84 KB LDS): %.3f ms\n", run_b(sb, claim_b, out, lds, ms_ticks, e0, e1));
    if (me == hipSuccess) {
        printf("B alone on the CU-masked stream: %.3f ms\n", run_b(sm, claim_b, out, lds, ms_ticks, e0, e1));
        unsigned hc[16]; CK(hipMemcpy(hc, claim_b, 64, hipMemcpyDeviceToHost));
        printf("  workgroups per XCC on the masked stream: %u %u %u %u | %u %u %u %u\n", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7]);
    }
    else printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(me));
    for (int masked = 0; masked < (me == hipSuccess ? 2 : 1); ++masked) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(claim_a, 0, 64, sa));
            CK(hipEventRecord(a0, sa));
            hipLaunchKernelGGL(k_hold, dim3(256), dim3(1024), lds_a, sa, claim_a, 0, 4, busy_a, 5 * ms_ticks, out);
            CK(hipEventRecord(a1, sa));
            // wait ~1 ms on the host, then launch B while A is resident
            const unsigned long long spin_until = 1000;
            for (volatile unsigned long long k = 0; k < 300000ull * spin_until / 1000ull; ++k) { }
            const float bms = run_b(masked ? sm : sb, claim_b, out, lds_b, ms_ticks, e0, e1);
            CK(hipEventSynchronize(a1));
            float ams = 0, gap = 0; CK(hipEventElapsedTime(&ams, a0, a1)); CK(hipEventElapsedTime(&gap, a0, e0));
            printf("A (%d busy CUs per XCD, %d KB LDS) resident for %.3f ms on XCDs 0-3; B (%d KB LDS) launched %.3f ms after A on %s: B took %.3f ms\n", busy_a, lds_a / 1024, ams, lds_b / 1024, gap, masked ? "the CU-masked stream" : "a plain second stream", bms);
        }
    }
    unsigned h[16]; CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost));
    printf("workgroups that did work in all launches: %u\n", h[1]);
    return 0;
}
