// Where the ~0.2 s of a process's first HIP calls go (c3d_create's sequence, timed call by call).
// hipcc -O2 tools/microbench/hip_init_phases.cpp -o /tmp/hip_init_phases && /tmp/hip_init_phases
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(label, call) do { double t0 = now(); auto e = (call); printf("%-34s %8.3f ms  (%d)\n", label, now() - t0, (int)e); } while (0)
int main() {
    int n = 0, v = 0;
    hipDeviceProp_t prop;
    hipStream_t s, s2;
    hipEvent_t ev;
    void *p = nullptr, *h = nullptr;
    T("hipGetDeviceCount (hipInit)", hipGetDeviceCount(&n));
    T("hipSetDevice", hipSetDevice(0));
    T("hipGetDeviceProperties", hipGetDeviceProperties(&prop, 0));
    T("hipDeviceGetAttribute(NumberOfXccs)", hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, 0));
    T("hipStreamCreateWithFlags", hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    T("hipStreamCreateWithFlags #2", hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    T("hipEventCreate", hipEventCreate(&ev));
    T("hipHostMalloc 64 B mapped", hipHostMalloc(&h, 64, hipHostMallocMapped));
    T("hipMalloc 4 KB", hipMalloc(&p, 4096));
    T("hipMemsetAsync", hipMemsetAsync(p, 0, 4096, s));
    T("hipStreamSynchronize", hipStreamSynchronize(s));
    printf("arch %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}
