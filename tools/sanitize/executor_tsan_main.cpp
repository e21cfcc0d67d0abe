// executor_tsan_main.cpp — libc3d's context code and the c3d_batch executor under ThreadSanitizer, on the CPU, against the fake HIP layer
// of hip_stub.cpp (tools/sanitize/run.sh builds and runs it).
//
//   0. a code object that fails to load: c3d_create reports it (C3D_ERR_HIP, message naming the unit) and does not remember it as loaded
//   1. c3d_batch --devices 4 --lanes 2 --map-devices-to 0   eight contexts of one process starting together on one device: the start that
//                                                           met a device exception on the GPU box in round 5
//   2. c3d_batch --devices 8 --lanes 3                      the production shape on an 8-GPU node: 24 contexts (the stub shows 8 devices)
//   3. an API storm: twelve threads, each with its own context, through configurations that want DIFFERENT code objects (the shipped
//      potential, the three other potentials' multi-step units, fp64, symmetric tiles, the embedding, the tear16 hook) while the others
//      launch — the loader must take each unit once, and never beside a launch.
// Passes when TSan reports nothing (halt_on_error) and the stub counted no load that overlapped a launch.
#include <sys/stat.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/c3d.h"

int c3d_batch_main(int argc, char** argv);          // c3d_batch_main.cpp built with -Dmain=c3d_batch_main
extern "C" long c3d_stub_violations();
extern "C" void c3d_stub_fail_next_loads(int n);
extern "C" long c3d_stub_launches();
extern "C" long c3d_stub_loads();
extern "C" long c3d_stub_cluster_launches();

#define REQ(x) do { if (!(x)) { fprintf(stderr, "FAILED line %d: %s (%s)\n", __LINE__, #x, c3d_last_error()); exit(1); } } while (0)

static void write_matrix(const std::string& path, int n, unsigned seed) {
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<double> u(0.5, 1.5);
    std::vector<double> m((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            const double v = i == j ? 5000.0 : 900.0 * u(g) / (double)(j - i);
            m[(size_t)i * n + j] = m[(size_t)j * n + i] = v;
        }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(1); }
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) fprintf(f, "%.10f ", m[(size_t)i * n + j]);
        fputs("\r\n", f);
    }
    fclose(f);
}

static int run_batch(std::vector<std::string> args) {
    std::vector<char*> argv;
    args.insert(args.begin(), "c3d_batch");
    for (std::string& a : args) argv.push_back(&a[0]);
    return c3d_batch_main((int)argv.size(), argv.data());
}

static void api_storm(int device_count) {
    std::atomic<int> failures{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 12; ++t)
        th.emplace_back([t, device_count, &failures] {
            c3d_ctx* c = nullptr;
            if (c3d_create(t % device_count, &c) != C3D_OK) { ++failures; return; }
            const int n = 40 + 7 * t;
            std::vector<double> IF((size_t)n * n);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) IF[(size_t)i * n + j] = i == j ? 4000.0 : 700.0 / (double)std::abs(i - j);
            c3d_model m;
            c3d_default_model(&m);
            bool ok = true;
            for (int round = 0; round < 3 && ok; ++round) {
                const int what = (t + round) % 6;
                if (what == 1) { m.noe_pot = 0; }                       // other potentials: their own multi-step units, on demand
                else if (what == 2) { m.noe_pot = 1; }
                else if (what == 3) { m.noe_pot = 2; }
                else { c3d_default_model(&m); }
                ok = ok && c3d_set_model(c, &m) == C3D_OK;
                ok = ok && c3d_set_option(c, "precision", what == 4 ? 64 : 32) == C3D_OK;
                ok = ok && c3d_set_option(c, "symmetric", what == 5 ? 1 : 0) == C3D_OK;
                ok = ok && c3d_set_if_matrix(c, IF.data(), n, 0.5, 11.0) == C3D_OK;
                ok = ok && c3d_init_replicas(c, 4, 82364, 0) == C3D_OK;
                if (what == 0) ok = ok && c3d_embed_replicas(c, 10) == C3D_OK;
                long done = 0;
                ok = ok && c3d_run_steps(c, 30, &done) == C3D_OK;
                ok = ok && c3d_run(c) == C3D_OK;
                std::vector<double> e(12), rho(4), dev(4);
                std::vector<int32_t> rank(4), sat(4);
                ok = ok && c3d_rank(c, rank.data()) == C3D_OK;
                ok = ok && c3d_score_replicas(c, IF.data(), 3, sat.data(), dev.data(), rho.data()) == C3D_OK;
                if (what == 2) { unsigned long long a, b, d; ok = ok && c3d_debug_tear16(c, 4, &a, &b, &d) == C3D_OK; }
            }
            if (!ok) { fprintf(stderr, "api storm thread %d: %s\n", t, c3d_last_error()); ++failures; }
            c3d_destroy(c);
        });
    for (std::thread& x : th) x.join();
    REQ(failures.load() == 0);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: executor_tsan <scratch dir>\n"); return 2; }
    const std::string tmp = argv[1];
    std::vector<std::string> mats;
    const int sizes[] = {37, 35, 60, 96, 73, 113, 150, 57, 80};
    for (int k = 0; k < 9; ++k) {
        const std::string p = tmp + "/chrS" + std::to_string(k + 1) + "_1mb_matrix.txt";
        write_matrix(p, sizes[k], 100u + (unsigned)k);
        mats.push_back(p);
    }
    // 0. a code object that fails to load is reported by c3d_create — and not remembered as loaded: the next c3d_create loads it
    setenv("C3D_STUB_DEVICES", "1", 1);
    {
        c3d_stub_fail_next_loads(2);                    // the next two load attempts report an error: two c3d_create calls fail at their first unit
        c3d_ctx* c = nullptr;
        REQ(c3d_create(0, &c) == C3D_ERR_HIP && c == nullptr);
        REQ(strstr(c3d_last_error(), "loading the code object of the") != nullptr);
        REQ(c3d_create(0, &c) == C3D_ERR_HIP && c == nullptr);
        REQ(c3d_create(0, &c) == C3D_OK && c != nullptr);
        double loaded = -1;
        REQ(c3d_get_stat(c, "units_loaded", &loaded) == C3D_OK && loaded == 4.0);
        c3d_destroy(c);
    }
    const long failed_attempts = 2;
    // 1. the start of round 5's exception: eight contexts, one device
    {
        std::vector<std::string> a = mats;
        a.insert(a.end(), {"--out", tmp + "/four", "-m", "6", "--devices", "4", "--lanes", "2", "--map-devices-to", "0"});
        REQ(run_batch(a) == 0);
    }
    const long loads_one_device = c3d_stub_loads();
    REQ(loads_one_device == 4 + failed_attempts);     // the default job's four units, once for the process (+ the two attempts that were made to fail): nobody loaded anything later
    // 2. the production shape: 8 devices x 3 lanes
    setenv("C3D_STUB_DEVICES", "8", 1);
    {
        std::vector<std::string> a = mats;
        a.insert(a.end(), {"--out", tmp + "/eight", "-m", "6", "--devices", "8", "--lanes", "3"});
        REQ(run_batch(a) == 0);
    }
    REQ(c3d_stub_loads() == 4 + failed_attempts + 4 * 7);   // device 0 had them; seven more devices x four units
    // 3. configurations that want other units, from twelve threads at once
    api_storm(8);
    printf("executor under TSan: %ld launches (%ld multi-step), %ld unit loads, %ld load/launch overlaps\n", c3d_stub_launches(),
           c3d_stub_cluster_launches(), c3d_stub_loads(), c3d_stub_violations());
    REQ(c3d_stub_violations() == 0);
    REQ(c3d_stub_cluster_launches() > 0);
    return 0;
}
