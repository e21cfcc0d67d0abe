// How are the workgroups of a launch dealt to the 8 XCDs of an MI355X?  For several launches of one-workgroup-per-CU grids
// (1024 threads, 90 KB of LDS: the cluster kernel's shape): per launch the histogram of (XCC_ID - blockIdx) mod 8, and
// whether blockIdx / 8 numbers the workgroups of every XCD without collision.
//   hipcc -O3 --offload-arch=gfx950 -o xcc_dispatch xcc_dispatch.hip && ./xcc_dispatch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k(int* out) {
    extern __shared__ float smem[];
    if (threadIdx.x == 0) { smem[0] = 1.0f; out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0x7); }
}
__global__ void small(int* x) { if (threadIdx.x == 0) atomicAdd(x, 1); }
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int G = prop.multiProcessorCount;
    int *out, *cnt; CK(hipMalloc(&out, 4 * G)); CK(hipMalloc(&cnt, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024));
    int h[1024];
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int launch = 0; launch < 16; ++launch) {
        const bool user = launch >= 4;                                  // launches 4.. on a user stream, as the library's
        for (int j = 0; j < launch % 5; ++j) small<<<3 + 7 * j, 64, 0, user ? st : 0>>>(cnt);      // other work in between
        if (launch % 3 == 2) CK(hipMemsetAsync(out, 0, 4 * G, user ? st : 0));
        k<<<G, 1024, 90 * 1024, user ? st : 0>>>(out);
        CK(hipStreamSynchronize(user ? st : 0));
        CK(hipMemcpy(h, out, 4 * G, hipMemcpyDeviceToHost));
        int hist[8] = {0}, perx[8] = {0}, collide = 0;
        unsigned seen[8] = {0};
        for (int i = 0; i < G; ++i) {
            ++hist[(h[i] - i) & 7]; ++perx[h[i]];
            if (seen[h[i]] & (1u << (i >> 3))) ++collide;
            seen[h[i]] |= 1u << (i >> 3);
        }
        printf("launch %2d: (xcc - blockIdx) mod 8 histogram", launch);
        for (int v = 0; v < 8; ++v) printf(" %3d", hist[v]);
        printf("   workgroups per XCD");
        for (int v = 0; v < 8; ++v) printf(" %2d", perx[v]);
        printf("   slot collisions with blockIdx/8: %d\n", collide);
    }
    // duration of this (empty) kernel: what a launch of the cluster kernel's shape costs before it has done anything
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {1024, 512, 256, 64}) {
        float best = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipExtLaunchKernelGGL(k, dim3(G), dim3(threads), 90 * 1024, st, e0, e1, 0, out);
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("empty kernel, %d workgroups x %4d threads, 90 KB LDS: %.2f us (start/stop events of the launch)\n", G, threads, best * 1e3);
    }
    return 0;
}
