// HBM stream rates of this GPU ("state the measured stream peak next to the 8 TB/s figure"): read-only sum,
// copy and triad over arrays far larger than the 256 MB Infinity Cache; float4 per lane, grid-stride, best of 10.
//   hipcc -O3 --offload-arch=gfx950 -o hbm_stream hbm_stream.hip && ./hbm_stream [GiB per array = 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_read(const float4* __restrict__ a, size_t n, float* out) {
    float s = 0.0f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) out[0] = s;               // never true: keeps the loads
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 x = a[i], y = b[i];
        c[i] = make_float4(x.x + 3.0f * y.x, x.y + 3.0f * y.y, x.z + 3.0f * y.z, x.w + 3.0f * y.w);
    }
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n = (size_t)(gib * (1u << 30)) / sizeof(float4);
    float4 *a, *b, *c; float* out;
    CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMalloc(&b, n * sizeof(float4))); CK(hipMalloc(&c, n * sizeof(float4))); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 0, n * sizeof(float4))); CK(hipMemset(b, 0, n * sizeof(float4))); CK(hipMemset(c, 0, n * sizeof(float4)));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {prop.multiProcessorCount * 8, prop.multiProcessorCount * 16, prop.multiProcessorCount * 32};
    for (int kind = 0; kind < 3; ++kind) {
        double best = 0; int bg = 0;
        for (int grid : grids)
            for (int rep = 0; rep < 10; ++rep) {
                CK(hipEventRecord(e0));
                if (kind == 0) k_read<<<grid, 256>>>(a, n, out);
                else if (kind == 1) k_copy<<<grid, 256>>>(a, b, n);
                else k_triad<<<grid, 256>>>(a, b, c, n);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (kind == 0 ? 1.0 : kind == 1 ? 2.0 : 3.0) * n * sizeof(float4);
                const double gbs = bytes / (ms * 1e-3) / 1e9;
                if (gbs > best) { best = gbs; bg = grid; }
            }
        printf("%-5s %.2f GiB per array: best %.0f GB/s (grid %d x 256)\n", kind == 0 ? "read" : kind == 1 ? "copy" : "triad", gib, best, bg);
    }
    return 0;
}
