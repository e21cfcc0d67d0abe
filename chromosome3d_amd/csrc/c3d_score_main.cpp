// c3d_score — command-line twin of the reference's spearman_IF_pdb.pl (:15-76):
//   c3d_score <IF matrix> <pdb file | directory of *.pdb> [range=3]
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

int main(int argc, char** argv) {
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
