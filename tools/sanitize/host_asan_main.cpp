// host_asan_main.cpp — drives the host-only part of libc3d (c3d_host.cpp: parser, writers, readers,
// assessment, Spearman) under AddressSanitizer + UBSan on the CPU.  (GPU sanitizers are not available
// on this pool; the HIP translation units are not part of this build.)
//   tools/sanitize/run.sh
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/c3d.h"

#define REQ(x) do { if (!(x)) { fprintf(stderr, "FAILED: %s (%s)\n", #x, c3d_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: host_asan_main <matrix.txt> <model.pdb> <tmpdir>\n"); return 2; }
    const std::string tmp = argv[3];
    double* IF = nullptr;
    int n = 0;
    REQ(c3d_parse_if_file(argv[1], &IF, &n) == C3D_OK && n > 1);
    // a crude stand-in for K1 (the device kernel is not in this build): any int32 matrix exercises the writers
    std::vector<int32_t> d10((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) d10[(size_t)i * n + j] = IF[(size_t)i * n + j] > 0 ? 10 + ((i * 31 + j * 17) % 900) : -10;
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) d10[(size_t)i * n + j] = d10[(size_t)j * n + i];
    int R = 0;
    REQ(c3d_write_front_half(d10.data(), n, 5, (tmp + "/a.dist").c_str(), (tmp + "/a.rr").c_str(), (tmp + "/contact.tbl").c_str(), &R) == C3D_OK);
    int32_t *ri = nullptr, *rj = nullptr, *rt = nullptr;
    int R2 = 0;
    REQ(c3d_read_tbl((tmp + "/contact.tbl").c_str(), &ri, &rj, &rt, &R2) == C3D_OK && R2 == R);
    float* xyz = nullptr;
    int m = 0;
    REQ(c3d_read_pdb_ca(argv[2], &xyz, &m) == C3D_OK && m == n);
    int sat = 0;
    double dev = 0, rho = 0;
    REQ(c3d_assess(xyz, n, R, ri, rj, rt, 0.5, &sat, &dev) == C3D_OK);
    REQ(c3d_spearman_if_dist(IF, xyz, n, 3, &rho) == C3D_OK);
    std::vector<float> two(xyz, xyz + (size_t)3 * n);
    two.insert(two.end(), xyz, xyz + (size_t)3 * n);
    double rr[2];
    REQ(c3d_spearman_if_dist_batch(IF, two.data(), n, 2, 3, rr) == C3D_OK && rr[0] == rho && rr[1] == rho);
    REQ(c3d_write_pdb((tmp + "/m.pdb").c_str(), xyz, n, 1.0, 2.0, 3.0, "m.pdb") == C3D_OK);
    float* back = nullptr;
    REQ(c3d_read_pdb_ca((tmp + "/m.pdb").c_str(), &back, &m) == C3D_OK && m == n);
    // output shaping (in place and to a second file, with and without the log), model reduction, similarity
    REQ(c3d_shape_pdb((tmp + "/m.pdb").c_str(), (tmp + "/shaped.pdb").c_str(), (tmp + "/model_info.log").c_str()) == C3D_OK);
    REQ(c3d_shape_pdb((tmp + "/m.pdb").c_str(), (tmp + "/m.pdb").c_str(), nullptr) == C3D_OK);
    REQ(c3d_shape_pdb((tmp + "/nope.pdb").c_str(), (tmp + "/x.pdb").c_str(), nullptr) != C3D_OK);
    {
        std::vector<double> xd((size_t)3 * n), red((size_t)3 * ((n + 1) / 2));
        for (int i = 0; i < 3 * n; ++i) xd[i] = xyz[i];
        REQ(c3d_reduce_model(xd.data(), n, red.data()) == C3D_OK);
        double sp = 0, rm = 0;
        REQ(c3d_model_similarity(xd.data(), xd.data(), n, &sp, &rm) == C3D_OK && sp > 0.999999 && rm < 1e-9);
    }
    // error paths must not leak or crash
    { double* none = nullptr; int nn = 0; REQ(c3d_parse_if_file((tmp + "/does-not-exist").c_str(), &none, &nn) != C3D_OK); }
    int32_t *xi = nullptr, *xj = nullptr, *xt = nullptr;
    REQ(c3d_read_tbl((tmp + "/a.dist").c_str(), &xi, &xj, &xt, &R2) != C3D_OK);
    int bad_i[1] = {9999}, bad_j[1] = {1};
    int32_t bad_t[1] = {10};
    REQ(c3d_assess(xyz, n, 1, bad_i, bad_j, bad_t, 0.5, &sat, &dev) != C3D_OK);
    printf("host sanitizer run ok: n=%d R=%d satisfied=%d sumdev=%.2f spearman=%.4f\n", n, R, sat, dev, rho);
    c3d_free(back); c3d_free(xyz); c3d_free(IF); c3d_free(ri); c3d_free(rj); c3d_free(rt);
    return 0;
}
