// Can a 16-byte aligned buffer_store_dwordx4 of one lane be observed half-written by a buffer_load_dwordx4 (sc1) of another
// CU?  The cluster kernel's hand-off units carry ONE tag word and three payload words (round 3); this stress test is the
// evidence that the four words of such a unit travel together on gfx950.  One producer workgroup rewrites 64 x 16 units
// with {i, i, i, i} for i = 1 .. iters (plain stores, as the kernel's), every other workgroup (one per CU, all XCDs)
// reads them back in a loop (L1-bypassing loads, as the kernel's) and counts units whose four words differ.
//   hipcc -O3 --offload-arch=gfx950 -o tear16 tear16.hip && ./tear16
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k(u32x4* buf, unsigned* stop, unsigned long long* stats, int iters) {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1024 * 16, 0x00020000);
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        for (int i = 1; i <= iters; ++i) {
            u32x4 v; v.x = v.y = v.z = v.w = (unsigned)i;
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, tid * 16, 0, 0);
        }
        __threadfence();
        if (tid == 0) atomicExch(stop, 1u);
        return;
    }
    unsigned long long reads = 0, torn = 0, fresh = 0;
    unsigned lastv = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, 0, 16);
        ++reads;
        if (!(v.x == v.y && v.y == v.z && v.z == v.w)) ++torn;
        if (v.x != lastv) { ++fresh; lastv = v.x; }
        if ((reads & 255) == 0 && __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(stop, 0, 4, 0x00020000), 0, 0, 16) != 0u) break;
    }
    atomicAdd(&stats[0], reads); atomicAdd(&stats[1], torn); atomicAdd(&stats[2], fresh);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    u32x4* buf; unsigned* stop; unsigned long long* stats;
    CK(hipMalloc(&buf, 1024 * 16)); CK(hipMalloc(&stop, 4)); CK(hipMalloc(&stats, 24));
    for (int round = 0; round < 3; ++round) {
        CK(hipMemset(buf, 0, 1024 * 16)); CK(hipMemset(stop, 0, 4)); CK(hipMemset(stats, 0, 24));
        k<<<prop.multiProcessorCount, 1024>>>(buf, stop, stats, 400000);
        CK(hipDeviceSynchronize());
        unsigned long long h[3]; CK(hipMemcpy(h, stats, 24, hipMemcpyDeviceToHost));
        printf("round %d: %llu unit reads by %d consumer workgroups, %llu saw a new value, %llu TORN\n", round, h[0], prop.multiProcessorCount - 1, h[2], h[1]);
    }
    return 0;
}
