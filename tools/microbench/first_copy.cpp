// First-use cost of the copy paths: pinned (hipHostMalloc) against pageable host memory, both directions, two sizes.
// hipcc -O2 tools/microbench/first_copy.cpp -o /tmp/first_copy && /tmp/first_copy pinned-first | pageable-first
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(label, call) do { double t0 = now(); auto e = (call); (void)hipStreamSynchronize(s); printf("%-44s %8.3f ms  (%d)\n", label, now() - t0, (int)e); } while (0)
int main(int argc, char** argv) {
    const bool pinned_first = argc > 1 && !strcmp(argv[1], "pinned-first");
    hipStream_t s;
    (void)hipSetDevice(0);
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const size_t big = 1656200, small = 160 * 1024;
    void *d = nullptr, *p = nullptr;
    (void)hipMalloc(&d, big);
    (void)hipMemsetAsync(d, 0, big, s);
    (void)hipStreamSynchronize(s);
    std::vector<char> host(big, 1);
    { double t0 = now(); auto e = hipHostMalloc(&p, big, hipHostMallocDefault); printf("%-44s %8.3f ms  (%d)\n", "hipHostMalloc 1.6 MB", now() - t0, (int)e); }
    memset(p, 1, big);
    for (int pass = 0; pass < 2; ++pass) {
        const bool pin = pinned_first ? pass == 0 : pass == 1;
        void* h = pin ? p : (void*)host.data();
        const char* w = pin ? "pinned  " : "pageable";
        char l[96];
        for (int rep = 0; rep < 2; ++rep) {
            snprintf(l, sizeof l, "%s H2D 1.6 MB  #%d", w, rep); T(l, hipMemcpyAsync(d, h, big, hipMemcpyHostToDevice, s));
            snprintf(l, sizeof l, "%s D2H 160 KB  #%d", w, rep); T(l, hipMemcpyAsync(h, d, small, hipMemcpyDeviceToHost, s));
            snprintf(l, sizeof l, "%s D2H 4 B     #%d", w, rep); T(l, hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, s));
            snprintf(l, sizeof l, "%s H2D 120 KB  #%d", w, rep); T(l, hipMemcpyAsync(d, h, 120 * 1024, hipMemcpyHostToDevice, s));
        }
    }
    return 0;
}
