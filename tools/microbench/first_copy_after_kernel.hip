// Does the first device-to-host copy that FOLLOWS a kernel pay a one-time cost, and does an earlier copy of the same kind remove it?
// hipcc -O2 --offload-arch=gfx950 tools/microbench/first_copy_after_kernel.hip -o /tmp/fcak && /tmp/fcak [pre] [bytes]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
#define T(label, call) do { double t0 = now(); auto e = (call); (void)hipStreamSynchronize(s); printf("%-44s %8.3f ms  (%d)\n", label, now() - t0, (int)e); } while (0)
int main(int argc, char** argv) {
    const bool pre = argc > 1 && !strcmp(argv[1], "pre");
    const size_t bytes = argc > 2 ? (size_t)atol(argv[2]) : 110 * 1024;
    hipStream_t s;
    (void)hipSetDevice(0);
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t big = 1656200;
    float* d = nullptr;
    (void)hipMalloc(&d, big);
    std::vector<char> host(big, 1), back(big, 0);
    T("H2D 1.6 MB pageable (first copy)", hipMemcpyAsync(d, host.data(), big, hipMemcpyHostToDevice, s));
    if (pre) T("D2H before any kernel", hipMemcpyAsync(back.data(), d, bytes, hipMemcpyDeviceToHost, s));
    { double t0 = now(); hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, d, (int)(big / 4)); (void)hipStreamSynchronize(s); printf("%-44s %8.3f ms\n", "kernel #0 (loads the code object)", now() - t0); }
    T("D2H after kernel #0", hipMemcpyAsync(back.data(), d, bytes, hipMemcpyDeviceToHost, s));
    T("D2H again", hipMemcpyAsync(back.data(), d, bytes, hipMemcpyDeviceToHost, s));
    { double t0 = now(); hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, d, (int)(big / 4)); (void)hipStreamSynchronize(s); printf("%-44s %8.3f ms\n", "kernel #1", now() - t0); }
    T("D2H after kernel #1", hipMemcpyAsync(back.data(), d, bytes, hipMemcpyDeviceToHost, s));
    T("D2H 18 KB", hipMemcpyAsync(back.data(), d, 18 * 1024, hipMemcpyDeviceToHost, s));
    T("D2H 1 MB", hipMemcpyAsync(back.data(), d, 1 << 20, hipMemcpyDeviceToHost, s));
    return 0;
}
