// c3d_host.h — error plumbing shared by the host translation units of libc3d.so
#pragma once
#include <cstddef>
#include <string>
#include <vector>
namespace c3d {
int fail(int code, const std::string& msg);
// average ranks of IF over the ordered pairs |i-j| >= range (spearman_IF_pdb.pl:50-63) as an n x n matrix
// (0 elsewhere); m = number of pairs, mean_rank = (m+1)/2, saa = sum (rank - mean)^2
void if_pair_ranks(const double* IF, int n, int range, std::vector<double>& rank_matrix, size_t& m, double& mean_rank, double& saa);
}
