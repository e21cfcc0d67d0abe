// hip_stub.cpp — a fake HIP layer for ThreadSanitizer runs of libc3d's HOST code on a box without a GPU (tools/sanitize/run.sh).
//
// c3d_api.cpp (context, code-object loader, launch program, executor of c3d_run) and c3d_batch_main.cpp (per-device lists, lanes, the XCD
// broker) are compiled as they are, with -fsanitize=thread, and linked against THIS file instead of libamdhip64 and the kernels'
// translation units: "device" memory is host memory, a stream is a counter, a copy is a memcpy, every kernel launcher returns success and
// computes nothing — except K1, which is restated on the host so that the executor has restraints to write, and the multi-step launcher,
// which writes the completion mark its kernel would write (every seventh launch does not: the abandoned-launch path runs too).
//
// Beyond what TSan sees by itself, the stub CHECKS the loader's contract (c3d_api.cpp "code objects"): a unit's load function and a launch
// must never overlap in time, whatever the thread — c3d_stub_violations() counts the overlaps, the harness fails on any.
// Test infrastructure; never linked into the product.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "../../chromosome3d_amd/csrc/c3d_internal.h"

namespace {
std::atomic<int> g_launching{0}, g_loading{0};
std::atomic<long> g_violations{0}, g_launches{0}, g_loads{0}, g_cluster_launches{0};
struct LaunchScope {
    LaunchScope() {
        g_launching.fetch_add(1);
        if (g_loading.load() != 0) g_violations.fetch_add(1);
        g_launches.fetch_add(1);
        std::this_thread::yield();                    // widen the window
    }
    ~LaunchScope() { g_launching.fetch_sub(1); }
};
std::atomic<int> g_fail_next_load{0};          // c3d_stub_fail_next_loads(n): the next n unit loads report an error (the loader's error path)
struct LoadScope {
    LoadScope() {
        g_loading.fetch_add(1);
        if (g_launching.load() != 0) g_violations.fetch_add(1);
        g_loads.fetch_add(1);
        std::this_thread::sleep_for(std::chrono::microseconds(300));     // a load takes milliseconds on the device: stay inside for a while
        if (g_launching.load() != 0) g_violations.fetch_add(1);
    }
    ~LoadScope() { g_loading.fetch_sub(1); }
};
int stub_devices() {
    const char* e = getenv("C3D_STUB_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n < 1 ? 1 : (n > 64 ? 64 : n);
}
thread_local int t_device = 0;
}  // namespace

extern "C" long c3d_stub_violations() { return g_violations.load(); }
extern "C" void c3d_stub_fail_next_loads(int n) { g_fail_next_load.store(n); }
static hipError_t load_result() {
    int left = g_fail_next_load.load();
    while (left > 0 && !g_fail_next_load.compare_exchange_weak(left, left - 1)) { }
    return left > 0 ? hipErrorSharedObjectInitFailed : hipSuccess;
}
extern "C" long c3d_stub_launches() { return g_launches.load(); }
extern "C" long c3d_stub_loads() { return g_loads.load(); }
extern "C" long c3d_stub_cluster_launches() { return g_cluster_launches.load(); }

// ---- the HIP API the host code calls ----------------------------------------------------------------------------------------------
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = stub_devices(); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= stub_devices()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) { *v = a == hipDeviceAttributeNumberOfXccs ? 8 : 0; return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { LaunchScope ls; memcpy(dst, src, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) { LaunchScope ls; memset(dst, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(calloc(1, 8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(calloc(1, 8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
// stream capture / graphs: the per-step path replays graphs; here a capture records nothing and a graph launch is one "launch"
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = reinterpret_cast<hipGraph_t>(calloc(1, 8)); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* ge, hipGraph_t, hipGraphNode_t*, char*, size_t) { *ge = reinterpret_cast<hipGraphExec_t>(calloc(1, 8)); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t g) { free(g); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { LaunchScope ls; return hipSuccess; }
}

// ---- the kernels' translation units: launchers that launch nothing, loaders that load nothing -------------------------------------
namespace c3d {

hipError_t preload_device_unit() { LoadScope l; return load_result(); }
hipError_t preload_cluster_base_unit() { LoadScope l; return load_result(); }
hipError_t preload_cluster_unit(int pot, bool) { LoadScope l; return pot >= 0 && pot <= 4 ? load_result() : hipErrorInvalidValue; }
hipError_t preload_score_unit() { LoadScope l; return load_result(); }
hipError_t preload_embed_unit() { LoadScope l; return load_result(); }
hipError_t preload_f64_unit() { LoadScope l; return load_result(); }
hipError_t preload_sym_unit() { LoadScope l; return load_result(); }

hipError_t launch_step(const DevModel&, const DevStep&, const DevFire&, const DevBuffers&, int, bool, bool, hipStream_t) { LaunchScope ls; return hipSuccess; }
hipError_t launch_eval_forces(const DevModel&, const DevStep&, const DevBuffers&, int, float*, bool, int, hipStream_t) { LaunchScope ls; return hipSuccess; }
hipError_t launch_energy(const DevModel& m, const DevStep&, const DevBuffers& b, int, float, float, double, hipStream_t) {
    LaunchScope ls;
    for (int r = 0; r < m.nrep; ++r) { b.E[4 * r] = 1000.0 + 7.0 * ((r * 5) % m.nrep); b.E[4 * r + 1] = 1.0; b.E[4 * r + 2] = 2.0; b.E[4 * r + 3] = 0.0; }   // distinct "energies": c3d_rank has something to order
    return hipSuccess;
}
hipError_t launch_centre(const DevModel&, const DevBuffers&, int, hipStream_t) { LaunchScope ls; return hipSuccess; }
size_t pair_targets_floats(int n, int npad) { return (size_t)n * npad; }
hipError_t launch_pair_targets(const DevModel&, const float*, float*, hipStream_t) { LaunchScope ls; return hipSuccess; }
AnnealIO anneal_io(const DevBuffers& b, int parity) {
    const int q = parity ^ 1;
    AnnealIO io;
    io.pin = b.P[parity]; io.xin = b.X[parity]; io.vin = b.V[parity]; io.vinit = b.Vinit; io.sin = b.S[parity];
    io.xout = b.X[q]; io.vout = b.V[q]; io.pout = b.P[q]; io.sout = b.S[q];
    return io;
}
bool cluster_plan(const DevModel& m, int num_cus, int num_xcc, int, int, int xcd_count, ClusterPlan* plan) {
    if (m.npad > 1024 || num_xcc != 8) return false;
    ClusterPlan pl{};
    pl.rpw = 4; pl.cw = 12; pl.helpers = 4; pl.wgs_per_cu = 1; pl.parts = (m.n + 47) / 48; pl.per_xcd = (m.nrep_g + xcd_count - 1) / xcd_count;
    pl.grid = num_cus; pl.threads = 1024; pl.units = 96; pl.device = 0; pl.lds = 84 * 1024; pl.expected = (unsigned)(m.nrep_g * pl.parts);
    pl.xcd_count = xcd_count;
    *plan = pl;
    return pl.per_xcd * pl.parts <= num_cus / 8;
}
size_t cluster_record_bytes(const DevModel& m, const ClusterPlan&) { return (size_t)2 * m.nrep_g * (m.npad + m.npad / 4) * 16; }
hipError_t launch_cluster(const DevModel&, const DevFire&, const ClusterPlan&, const AnnealIO&, const float*, void*, const StepRun*, int, int, int,
                          unsigned tag_base, unsigned* timeout, unsigned*, hipStream_t) {
    LaunchScope ls;
    // the kernel's last workgroup writes the completion mark into the host-mapped word; every seventh launch "loses" a workgroup
    if (g_cluster_launches.fetch_add(1) % 7 != 6) __atomic_store_n(&timeout[1], tag_base | 1u, __ATOMIC_RELEASE);
    return hipSuccess;
}
hipError_t launch_tear16(int, void*, unsigned*, unsigned long long*, int, hipStream_t) { LaunchScope ls; return hipSuccess; }
void sym_geometry(const DevModel&, int* Q, int* G, int* od, int* dg) { *Q = 1; *G = 1; *od = 0; *dg = 1; }
size_t sym_scratch_floats(const DevModel& m) { return (size_t)m.npad * 8; }
void sym_tile_list(const DevModel&, int2* out) { out[0].x = 0; out[0].y = 0; }
hipError_t launch_step_sym(const DevModel&, const DevStep&, const DevFire&, const DevBuffers&, int, const void*, float*, hipStream_t) { LaunchScope ls; return hipSuccess; }
int cols64(int n) { return (n + 127) / 128 * 128; }
hipError_t launch_step64(const DevModel&, const double*, const double*, const double*, int, const Buffers64&, int, hipStream_t) { LaunchScope ls; return hipSuccess; }
hipError_t launch_targets64(const DevModel&, const double*, int, const int32_t*, double*, hipStream_t) { LaunchScope ls; return hipSuccess; }
hipError_t launch_import64(const DevModel&, const float*, const Buffers64&, hipStream_t) { LaunchScope ls; return hipSuccess; }
hipError_t launch_export64(const DevModel&, const Buffers64&, int, float*, float*, float*, hipStream_t) { LaunchScope ls; return hipSuccess; }
size_t fire_state64_bytes() { return 32; }
// K1 restated on the host (chromosome3D.pl:110-162 as c3d_api.cpp's own near-tie redo does it): the executor needs real restraints
hipError_t launch_if_to_target(const double* IF, int n, int npad, double alpha, double K, int min_sep, int, double*, double*, int,
                               int32_t* dist10, float* tgt, unsigned char*, unsigned*, hipStream_t) {
    LaunchScope ls;
    const size_t nn = (size_t)n * n;
    double sum = 0;
    for (size_t k = 0; k < nn; ++k) sum += pow(IF[k], alpha);
    const double mean = sum / ((double)n * (double)n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double v = pow(IF[(size_t)i * n + j], alpha) / mean;
            long long t = -10;
            if (v != 0) t = llround(10.0 * K / v);
            if (t > 2000000000LL) t = 2000000000LL;
            dist10[(size_t)i * n + j] = (int32_t)t;
            const int sep = i > j ? i - j : j - i;
            tgt[(size_t)i * npad + j] = (sep >= min_sep && t > 0) ? (float)((double)t / 10.0) : 0.0f;
        }
    return hipSuccess;
}
hipError_t launch_dg_embed(const float*, int, int, int, float, float, uint64_t, uint32_t, int, const float*, float*, float*, float*, float*, float*, hipStream_t) { LaunchScope ls; return hipSuccess; }
hipError_t launch_score(const float*, const float*, const double*, int, int, int, int, int, unsigned, double, double, double, double*, unsigned*, unsigned*,
                        double* partial, int* overflow, hipStream_t) {
    LaunchScope ls;
    (void)partial;
    *overflow = 0;
    return hipSuccess;
}

}  // namespace c3d
