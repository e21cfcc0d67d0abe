// launch_rate.hip — what does one dependent kernel launch cost on this box?  (diagnostic, not product)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { float a[57]; };
__global__ void k_empty(float* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_big(Big b, float* p) { if (p && threadIdx.x == 9999) p[0] = b.a[3]; }
__global__ void k_lds(float* p) { extern __shared__ float s[]; s[threadIdx.x] = 1; __syncthreads(); if (p && threadIdx.x == 9999) p[0] = s[0]; }
__global__ void k_rw(const float* __restrict__ in, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] + 1.0f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> double timeit(hipStream_t s, int n, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) f(i); hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    hipEventRecord(a, s); for (int i = 0; i < n; ++i) f(i); hipEventRecord(b, s); hipStreamSynchronize(s);
    double host = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("   device %.3f us/launch, host %.3f us/launch\n", 1e3 * ms / n, host);
    return 1e3 * ms / n;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *x, *y; CK(hipMalloc(&x, 1 << 22)); CK(hipMalloc(&y, 1 << 22));
    const int N = 5000; Big big{};
    printf("empty 1x64 eager\n");      timeit(s, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, (float*)nullptr); });
    printf("empty 640x256 eager\n");   timeit(s, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(640), dim3(256), 0, s, (float*)nullptr); });
    printf("bigarg 640x256 eager\n");  timeit(s, N, [&](int) { hipLaunchKernelGGL(k_big, dim3(640), dim3(256), 0, s, big, (float*)nullptr); });
    printf("lds6k 640x256 eager\n");   timeit(s, N, [&](int) { hipLaunchKernelGGL(k_lds, dim3(640), dim3(256), 6400, s, (float*)nullptr); });
    printf("rw pingpong 640x256 eager (dependent data)\n");
    timeit(s, N, [&](int i) { hipLaunchKernelGGL(k_rw, dim3(640), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, 640 * 256); });
    // graph of 256 dependent rw kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 256; ++i) hipLaunchKernelGGL(k_rw, dim3(640), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, 640 * 256);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("graph of 256 rw kernels (per kernel)\n");
    { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s); hipStreamSynchronize(s);
      hipEventRecord(a, s); for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s); hipEventRecord(b, s); hipStreamSynchronize(s);
      float ms; hipEventElapsedTime(&ms, a, b); printf("   device %.3f us/kernel\n", 1e3 * ms / (20 * 256)); }
    // default (null) stream for comparison
    printf("rw pingpong on the NULL stream\n");
    timeit(nullptr, N, [&](int i) { hipLaunchKernelGGL(k_rw, dim3(640), dim3(256), 0, 0, (i & 1) ? y : x, (i & 1) ? x : y, 640 * 256); });
    return 0;
}
