// host_asan_main.cpp — drives the host-only part of libc3d (c3d_host.cpp: parser, writers, readers,
// assessment, violation table, Spearman, similarity) under AddressSanitizer + UBSan (+ float-cast-overflow)
// on the CPU, and — built with -fsanitize=thread, argument "tsan" — the threaded parser and concurrent callers.
// (GPU sanitizers are not available on this pool; the HIP translation units are not part of this build.)
//   tools/sanitize/run.sh
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/c3d.h"

#define REQ(x) do { if (!(x)) { fprintf(stderr, "FAILED line %d: %s (%s)\n", __LINE__, #x, c3d_last_error()); return 1; } } while (0)

static bool write_text(const std::string& path, const std::string& body) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(body.data(), 1, body.size(), f) == body.size();
    return fclose(f) == 0 && ok;
}
static std::string slurp(const std::string& path) {
    std::string out;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return out;
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, k);
    fclose(f);
    return out;
}

// a matrix text of n x n numbers with 17 significant digits: ~19 bytes a number (n = 200: 760 KB -> 4 parser threads; n = 700: 9.3 MB -> 16)
static std::string matrix_text(int n, unsigned seed, std::vector<double>* values, bool symmetric, const char* eol) {
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<double> u(0.0, 1e6);
    std::vector<double> m((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = symmetric ? i : 0; j < n; ++j) {
            double v = u(g);
            if ((i * 7 + j) % 11 == 0) v = std::floor(v / 1000.0);          // ties
            if ((i + j) % 97 == 0 && i != j) v = 0.0;
            m[(size_t)i * n + j] = v;
            if (symmetric) m[(size_t)j * n + i] = v;
        }
    std::string t;
    char b[40];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) { snprintf(b, sizeof b, "%.17g ", m[(size_t)i * n + j]); t += b; }
        t += eol;
    }
    if (values) *values = m;
    return t;
}

static int parser_cases(const std::string& tmp) {
    // threaded parser: 4-thread and 16-thread splits return what a single strtod pass returns
    for (int n : {200, 700}) {
        std::vector<double> want;
        REQ(write_text(tmp + "/big.txt", matrix_text(n, 7u + n, &want, false, "\r\n")));
        double* IF = nullptr;
        int got = 0;
        REQ(c3d_parse_if_file((tmp + "/big.txt").c_str(), &IF, &got) == C3D_OK && got == n);
        REQ(memcmp(IF, want.data(), sizeof(double) * want.size()) == 0);
        c3d_free(IF);
    }
    // forms from_chars does not take whole go to strtod; the last token may end the file
    REQ(write_text(tmp + "/odd.txt", "+1.5 0x10\t1e400\n-0.0 inf 7"));
    double* IF = nullptr;
    int n = 0;
    REQ(c3d_parse_if_file((tmp + "/odd.txt").c_str(), &IF, &n) != C3D_OK);        // 2 fields on line 1, 6 numbers
    REQ(write_text(tmp + "/odd.txt", "+1.5 0x10\n1e400 7"));
    REQ(c3d_parse_if_file((tmp + "/odd.txt").c_str(), &IF, &n) == C3D_OK && n == 2 && IF[0] == 1.5 && IF[1] == 16.0 && std::isinf(IF[2]) && IF[3] == 7.0);
    c3d_free(IF);
    IF = nullptr;
    for (const char* bad : {"1 2\n3 x\n", "1 2\n3\n", "\n", "", "1 2 3\n4 5 6\n", "1,2\n3 4\n", "1 2\n3 4abc\n"})
    {
        REQ(write_text(tmp + "/bad.txt", bad));
        REQ(c3d_parse_if_file((tmp + "/bad.txt").c_str(), &IF, &n) != C3D_OK && strlen(c3d_last_error()) > 0);
    }
    return 0;
}

static int tbl_cases(const std::string& tmp) {
    int32_t *ri = nullptr, *rj = nullptr, *rt = nullptr;
    int R = -1;
    // good rows, with and without a final newline, CRLF, blank lines, four-digit residues (N > 999: "%3d" overflows its width)
    REQ(write_text(tmp + "/t.tbl", "assign45 (resid   1 and name ca) (resid  10 and name ca) 24.80 0.00 0.00\r\n\n"
                                   "assign45 (resid 1000 and name ca) (resid 1200 and name ca) 3.10 0.00 0.00"));
    REQ(c3d_read_tbl((tmp + "/t.tbl").c_str(), &ri, &rj, &rt, &R) == C3D_OK && R == 2 && ri[1] == 1000 && rj[1] == 1200 && rt[0] == 248 && rt[1] == 31);
    c3d_free(ri); c3d_free(rj); c3d_free(rt);
    REQ(write_text(tmp + "/t.tbl", ""));
    REQ(c3d_read_tbl((tmp + "/t.tbl").c_str(), &ri, &rj, &rt, &R) == C3D_OK && R == 0);
    c3d_free(ri); c3d_free(rj); c3d_free(rt);
    for (const char* bad : {"assign45 (resid 1 and name ca) (resid 10 and name ca) 24.80 0.00\n",            // short
                            "hello world\n",
                            "assign45 (resid 1 and name ca) (resid 10 and name ca) nan 0.00 0.00\n",
                            "assign45 (resid 1 and name ca) (resid 10 and name ca) 1e300 0.00 0.00\n",
                            "assign45 (resid x and name ca) (resid 10 and name ca) 2.0 0.00 0.00\n",
                            "assign45 (resid -4 and name ca) (resid 10 and name ca) 2.0 0.00 0.00\n",
                            "assign45 ((((((((((((((((((((((((((((((\n",
                            "assign45 (resid 1 and name ca) (resid 10 and name ca) 24.8x 0.00 0.00"}) {
        REQ(write_text(tmp + "/t.tbl", bad));
        const int rc = c3d_read_tbl((tmp + "/t.tbl").c_str(), &ri, &rj, &rt, &R);
        if (rc == C3D_OK) { c3d_free(ri); c3d_free(rj); c3d_free(rt); }
        // "24.8x": strtod reads 24.8 as Perl's numeric conversion of the token does; everything else is refused
        REQ((rc != C3D_OK) == (strstr(bad, "24.8x") == nullptr));
    }
    return 0;
}

static int violation_cases(const std::string& tmp) {
    // tie-heavy: beads on a 0.005 A lattice along x, so distances and deviations sit on "%.2f" rounding ties; N = 1200 > 999
    const int n = 1200;
    std::vector<float> xyz((size_t)3 * n, 0.0f);
    for (int i = 0; i < n; ++i) { xyz[3 * i] = 0.005f * (float)(i * 37 % 2000); xyz[3 * i + 1] = (i & 1) ? 0.125f : -0.375f; }
    std::vector<int32_t> ri, rj, rt;
    for (int i = 1; i <= n; i += 3)
        for (int j = i + 5; j <= n; j += 101) { ri.push_back(i); rj.push_back(j); rt.push_back((i * 13 + j) % 97 + 1); }
    const int R = (int)ri.size();
    int sat = 0, sat2 = 0;
    double dev = 0, dev2 = 0;
    const std::string out = tmp + "/viol.txt";
    remove(out.c_str());
    REQ(c3d_write_violations(xyz.data(), n, R, ri.data(), rj.data(), rt.data(), 0.5, "m.pdb", "contact.tbl", out.c_str(), &sat, &dev) == C3D_OK);
    REQ(c3d_assess(xyz.data(), n, R, ri.data(), rj.data(), rt.data(), 0.5, &sat2, &dev2) == C3D_OK && sat == sat2 && dev == dev2);
    // every row equals what C's printf makes of the same numbers
    const std::string body = slurp(out);
    size_t p = body.find('\n', body.find('\n') + 1) + 1, rows = 0;
    while (p < body.size()) {
        const size_t e = body.find('\n', p);
        REQ(e != std::string::npos);
        int flag, i, j;
        double d1, d2, t;
        REQ(sscanf(body.c_str() + p, "%d %lf %lf # assign45 resid %d and name ca resid %d and name ca %lf", &flag, &d1, &d2, &i, &j, &t) == 6);
        char want[200];
        // the deviation printed is +-|d - t| or 0: rebuild the row with printf from the parsed values of the hand-rolled row, then compare text
        snprintf(want, sizeof want, "%3d\t%.2f\t%.2f # assign45  resid %3d and name ca   resid %3d and name ca  %.2f 0.00 0.00", flag, d1, d2, i, j, t);
        REQ(body.compare(p, e - p, want) == 0 || (d1 == 0 && body[p + 4] == '-'));   // "-0.00" keeps its sign as printf's would
        p = e + 1;
        ++rows;
    }
    REQ((int)rows == R);
    // empty restraint list: two header lines, zero rows
    remove(out.c_str());
    REQ(c3d_write_violations(xyz.data(), n, 0, nullptr, nullptr, nullptr, 0.5, nullptr, nullptr, out.c_str(), &sat, &dev) == C3D_OK && sat == 0 && dev == 0);
    REQ(slurp(out).size() > 20);
    // coordinates a PDB cannot hold: refused with a message, nothing written, no float -> integer conversion out of range
    const float inf = std::numeric_limits<float>::infinity(), nan = std::numeric_limits<float>::quiet_NaN();
    for (float badv : {inf, -inf, nan, 1e30f, -1e30f, 1e7f, 10000.0f, -1000.0f}) {
        std::vector<float> b(xyz);
        b[3 * 17 + 1] = badv;
        remove(out.c_str());
        REQ(c3d_write_violations(b.data(), n, R, ri.data(), rj.data(), rt.data(), 0.5, "m", "t", out.c_str(), &sat, &dev) == C3D_ERR_INVALID);
        REQ(slurp(out).empty() && strstr(c3d_last_error(), "bead 18") != nullptr);
        REQ(c3d_assess(b.data(), n, R, ri.data(), rj.data(), rt.data(), 0.5, &sat, &dev) == C3D_ERR_INVALID);
        REQ(c3d_write_pdb((tmp + "/bad.pdb").c_str(), b.data(), n, 1, 2, 3, "x") == C3D_ERR_INVALID);
        std::vector<double> IF((size_t)50 * 50, 1.0);
        double rho;
        REQ(c3d_spearman_if_dist(IF.data(), b.data(), 50, 3, &rho) == C3D_ERR_INVALID);
    }
    REQ(c3d_write_pdb((tmp + "/bad.pdb").c_str(), xyz.data(), n, std::nan(""), 2, 3, "x") == C3D_ERR_INVALID);
    REQ(c3d_assess(xyz.data(), n, R, ri.data(), rj.data(), rt.data(), std::nan(""), &sat, &dev) == C3D_ERR_INVALID);
    int32_t huge_t[1] = {std::numeric_limits<int32_t>::min()}, one[1] = {1}, six[1] = {6};
    REQ(c3d_write_violations(xyz.data(), n, 1, one, six, huge_t, 0.5, "m", "t", out.c_str(), &sat, &dev) == C3D_ERR_INVALID);
    // the largest values that ARE accepted still fit their columns
    {
        std::vector<float> b(xyz);
        b[0] = 9999.999f; b[1] = -999.999f;
        REQ(c3d_write_pdb((tmp + "/edge.pdb").c_str(), b.data(), n, 1e12, -2, 3, "x") == C3D_OK);
        float* back = nullptr;
        int m = 0;
        REQ(c3d_read_pdb_ca((tmp + "/edge.pdb").c_str(), &back, &m) == C3D_OK && m == n && back[0] == 9999.999f && back[1] == -999.999f);
        c3d_free(back);
    }
    return 0;
}

static int ranker_cases() {
    // symmetric (ranked from the upper half, copies = 2) and asymmetric (ordered pairs) branch of the IF ranker agree where both apply:
    // perturbing one lower-triangle element inside the |i-j| < range band leaves the matrix symmetric for the ranker; outside it does not
    const int n = 60;
    std::vector<double> IF;
    (void)matrix_text(n, 3u, &IF, true, "\n");
    IF[5 * n + 9] = IF[9 * n + 5] = -0.0;                   // signed zeros tie with +0.0
    IF[6 * n + 12] = IF[12 * n + 6] = 0.0;
    std::vector<float> xyz((size_t)3 * n);
    std::mt19937 g(11);
    std::normal_distribution<float> nd(0.f, 8.f);
    for (auto& v : xyz) v = nd(g);
    double rho_sym = 0, rho_asym = 0, rho0 = 0;
    REQ(c3d_spearman_if_dist(IF.data(), xyz.data(), n, 3, &rho_sym) == C3D_OK);
    std::vector<double> A(IF);
    A[(size_t)40 * n + 2] = std::nextafter(A[(size_t)40 * n + 2], 1e300);         // asymmetric by one ulp: the other branch, ranks may move by one tie
    REQ(c3d_spearman_if_dist(A.data(), xyz.data(), n, 3, &rho_asym) == C3D_OK && std::fabs(rho_sym - rho_asym) < 1e-3);
    REQ(c3d_spearman_if_dist(IF.data(), xyz.data(), n, 0, &rho0) == C3D_OK);        // range 0: the diagonal counts once
    A[7] = std::nan("");
    REQ(c3d_spearman_if_dist(A.data(), xyz.data(), n, 3, &rho_asym) == C3D_ERR_INVALID);
    std::vector<double> flat((size_t)n * n, 1.0);                                   // every IF equal: one tie group, correlation undefined but no crash
    (void)c3d_spearman_if_dist(flat.data(), xyz.data(), n, 3, &rho0);
    REQ(c3d_spearman_if_dist(IF.data(), xyz.data(), n, n, &rho0) != C3D_OK);        // range leaves no pairs
    // similarity / reduction refuse non-finite input
    std::vector<double> xd(xyz.begin(), xyz.end()), red((size_t)3 * ((n + 1) / 2));
    double sp, rm;
    REQ(c3d_model_similarity(xd.data(), xd.data(), n, &sp, &rm) == C3D_OK);
    xd[4] = std::numeric_limits<double>::infinity();
    REQ(c3d_model_similarity(xd.data(), xd.data(), n, &sp, &rm) == C3D_ERR_INVALID && c3d_reduce_model(xd.data(), n, red.data()) == C3D_ERR_INVALID);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: host_asan_main <matrix.txt> <model.pdb> <tmpdir> [tsan]\n"); return 2; }
    const std::string tmp = argv[3];
    const bool tsan = argc > 4 && !strcmp(argv[4], "tsan");
    double* IF = nullptr;
    int n = 0;
    REQ(c3d_parse_if_file(argv[1], &IF, &n) == C3D_OK && n > 1);
    if (tsan) {
        // the threaded parser, and four host threads using the helpers at once (each with its own thread-local error string)
        if (parser_cases(tmp)) return 1;
        float* xyz = nullptr;
        int m = 0;
        REQ(c3d_read_pdb_ca(argv[2], &xyz, &m) == C3D_OK && m == n);
        std::vector<std::thread> th;
        std::vector<int> rc(4, 0);
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                double rho = 0, *again = nullptr;
                int nn = 0;
                rc[t] |= c3d_spearman_if_dist(IF, xyz, n, 3, &rho) != C3D_OK;
                rc[t] |= c3d_parse_if_file(argv[1], &again, &nn) != C3D_OK;
                c3d_free(again);
                rc[t] |= c3d_parse_if_file("/nonexistent", &again, &nn) == C3D_OK || strlen(c3d_last_error()) == 0;
                rc[t] |= c3d_write_pdb((tmp + "/t" + std::to_string(t) + ".pdb").c_str(), xyz, n, 1, 2, 3, "t") != C3D_OK;
            });
        for (auto& x : th) x.join();
        REQ(rc[0] + rc[1] + rc[2] + rc[3] == 0);
        c3d_free(xyz); c3d_free(IF);
        printf("host thread-sanitizer run ok\n");
        return 0;
    }
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
    {
        int sat_v = 0;
        double dev_v = 0;
        REQ(c3d_write_violations(xyz, n, R, ri, rj, rt, 0.5, "m.pdb", "contact.tbl", (tmp + "/cv.txt").c_str(), &sat_v, &dev_v) == C3D_OK && sat_v == sat && dev_v == dev);
    }
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
    // round 5: the host code added in round 4
    if (parser_cases(tmp) || tbl_cases(tmp) || violation_cases(tmp) || ranker_cases()) return 1;
    printf("host sanitizer run ok: n=%d R=%d satisfied=%d sumdev=%.2f spearman=%.4f\n", n, R, sat, dev, rho);
    c3d_free(back); c3d_free(xyz); c3d_free(IF); c3d_free(ri); c3d_free(rj); c3d_free(rt);
    return 0;
}
