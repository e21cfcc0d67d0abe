// Which SIMD does wave w of a 1024-thread workgroup run on?  (HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13])
//   hipcc -O3 --offload-arch=gfx950 -o wave_simd_map wave_simd_map.hip && ./wave_simd_map [threads = 1024] [dynamic LDS bytes = 0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_map(unsigned* out) {
    extern __shared__ float lds[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw | (xcc << 24);
    if (threadIdx.x == 99999) lds[0] = 0;
}
int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 1024;
    const int ldsb = argc > 2 ? atoi(argv[2]) : 0;
    const int blocks = 8;
    unsigned* d; CK(hipMalloc(&d, blocks * 16 * 4)); CK(hipMemset(d, 0xff, blocks * 16 * 4));
    if (ldsb > 64 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_map), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    k_map<<<blocks, threads, ldsb>>>(d);
    CK(hipDeviceSynchronize());
    unsigned h[blocks * 16]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    for (int b = 0; b < blocks; ++b) {
        printf("block %d (XCD %u, SE %u, CU %2u): wave -> SIMD:", b, (h[b * 16] >> 24) & 7, (h[b * 16] >> 13) & 7, (h[b * 16] >> 8) & 15);
        for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
        printf("   slots:");
        for (int w = 0; w < threads / 64; ++w) printf(" %u", h[b * 16 + w] & 15);
        printf("\n");
    }
    return 0;
}
