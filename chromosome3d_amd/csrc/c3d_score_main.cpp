// c3d_score — command-line twin of the reference's spearman_IF_pdb.pl (:15-76):
//   c3d_score <IF matrix> <pdb file | directory of *.pdb> [range=3]
// and of its satisfaction report (count_satisfied_tbl_rows :447-485, sum_noe_dev :581-600), for drivers without the in-process binding:
//   c3d_score --assess <contact.tbl> <relax> <violation file | -> <pdb> [<pdb> ...]
// prints one row "count/total\tsum_of_deviations\tpdb" per model, in the order given, and appends every model's violation table
// prints "SRCC\tPDB" rows sorted by descending coefficient (3 decimals), computed by
// c3d_spearman_if_dist (average-rank Spearman over ordered pairs |i-j| >= range).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/c3d.h"

static int assess_main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: c3d_score --assess <contact.tbl> <relax> <violation file | -> <pdb> [<pdb> ...]\n"); return 2; }
    int32_t *ri = nullptr, *rj = nullptr, *rt = nullptr;
    int R = 0;
    if (c3d_read_tbl(argv[2], &ri, &rj, &rt, &R) != C3D_OK) { fprintf(stderr, "ERROR! %s\n", c3d_last_error()); return 1; }
    const double relax = atof(argv[3]);
    const bool viol = std::string(argv[4]) != "-";
    for (int k = 5; k < argc; ++k) {
        float* xyz = nullptr;
        int n = 0, sat = 0;
        double dev = 0;
        if (c3d_read_pdb_ca(argv[k], &xyz, &n) != C3D_OK) { fprintf(stderr, "ERROR! %s\n", c3d_last_error()); return 1; }
        const int rc = viol ? c3d_write_violations(xyz, n, R, ri, rj, rt, relax, argv[k], argv[2], argv[4], &sat, &dev)
                            : c3d_assess(xyz, n, R, ri, rj, rt, relax, &sat, &dev);
        c3d_free(xyz);
        if (rc != C3D_OK) { fprintf(stderr, "ERROR! %s\n", c3d_last_error()); return 1; }
        printf("%d/%d\t%.2f\t%s\n", sat, R, dev, argv[k]);
    }
    c3d_free(ri); c3d_free(rj); c3d_free(rt);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "--assess") return assess_main(argc, argv);
    if (argc < 3) { fprintf(stderr, "usage: c3d_score <IF matrix> <pdb|dir> [range=3]\n"); return 2; }
    const int range = argc > 3 ? atoi(argv[3]) : 3;
    double* IF = nullptr;
    int n = 0;
    if (c3d_parse_if_file(argv[1], &IF, &n) != C3D_OK) { fprintf(stderr, "ERROR! %s\n", c3d_last_error()); return 1; }
    std::vector<std::string> pdbs;
    struct stat st;
    if (stat(argv[2], &st) != 0) { fprintf(stderr, "ERROR! %s not found\n", argv[2]); return 1; }
    if (S_ISDIR(st.st_mode)) {
        DIR* d = opendir(argv[2]);
        while (dirent* e = d ? readdir(d) : nullptr) {
            const std::string f = e->d_name;
            if (f.size() > 4 && f.substr(f.size() - 4) == ".pdb") pdbs.push_back(std::string(argv[2]) + "/" + f);
        }
        if (d) closedir(d);
        std::sort(pdbs.begin(), pdbs.end());
    } else {
        pdbs.push_back(argv[2]);
    }
    if (pdbs.empty()) { fprintf(stderr, "ERROR! no pdb files in %s\n", argv[2]); return 1; }
    std::vector<std::pair<double, std::string>> rows;
    for (const std::string& p : pdbs) {
        float* xyz = nullptr;
        int m = 0;
        if (c3d_read_pdb_ca(p.c_str(), &xyz, &m) != C3D_OK) { fprintf(stderr, "ERROR! %s\n", c3d_last_error()); return 1; }
        if (m != n) { fprintf(stderr, "ERROR! mismatch in size! %s has %d CA atoms, matrix is %d x %d\n", p.c_str(), m, n, n); return 1; }
        if (range >= m) { printf("Spearman Correlation coefficient = -\n"); return 0; }
        double rho = 0;
        if (c3d_spearman_if_dist(IF, xyz, n, range, &rho) != C3D_OK) { fprintf(stderr, "ERROR! %s\n", c3d_last_error()); return 1; }
        c3d_free(xyz);
        rows.emplace_back(rho, p);
    }
    std::stable_sort(rows.begin(), rows.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    printf("SRCC\tPDB\n");
    for (const auto& r : rows) printf("%.3f\t%s\n", r.first, r.second.c_str());
    c3d_free(IF);
    return 0;
}
