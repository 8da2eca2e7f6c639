// Host-only fuzz of the sparse analysis (cedarsim.jl_amd/csrc/ch_sparse_host.hpp): random small matrices with zero, tiny and
// missing entries; every index of the plan must stay inside its array.  Built with sanitizers by tests/test_host_analysis_fuzz.py.
#include <cstdio>
#include <vector>
#include <map>
#include <cmath>
#include <set>
#include <algorithm>
#include <numeric>
#include <string>
#include <cstdint>
#include <random>
#include <queue>
#include <functional>
#include "cedarhip.h"
#include "ch_sparse_host.hpp"
using namespace chip;
// emulate the GPU refactor+solve on host using the plan, check against dense solve
int main() {
  std::mt19937 rng(123);
  int nfail = 0, nsing = 0, ntot = 0;
  for (int trial = 0; trial < 1200; ++trial) {
    int n = 1 + rng() % 14;
    double dens = (rng() % 100) / 100.0;
    std::vector<int> rp(1, 0), ci; std::vector<double> av;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) {
        bool diag = i == j;
        if (diag || (rng() % 1000) / 1000.0 < dens) {
          ci.push_back(j);
          int r = rng() % 10;
          double v = r < 2 ? 0.0 : r < 4 ? 1e-15 : ((int)(rng() % 2000) - 1000) / 100.0;
          av.push_back(v);
        }
      }
      rp.push_back((int)ci.size());
    }
    SparsePlan P;
    int rc = sparse_analyse(n, rp, ci, av, P);
    ++ntot;
    if (rc != CH_OK) { ++nsing; continue; }
    // consistency checks of the plan
    if ((int)P.prow.size() != n || (int)P.pcol.size() != n) { printf("bad perm size\n"); ++nfail; }
    for (size_t k = 0; k < P.a2lu.size(); ++k) if (P.a2lu[k] < 0 || P.a2lu[k] >= P.nnz_lu) { printf("a2lu oob\n"); ++nfail; break; }
    for (int x : P.upd_dst) if (x < 0 || x >= P.nnz_lu) { printf("upd_dst oob\n"); ++nfail; break; }
    for (int x : P.upd_src) if (x < 0 || x >= P.nnz_lu) { printf("upd_src oob\n"); ++nfail; break; }
    for (int x : P.l_pos) if (x < 0 || x >= P.nnz_lu) { printf("l_pos oob\n"); ++nfail; break; }
    for (int x : P.u_pos) if (x < 0 || x >= P.nnz_lu) { printf("u_pos oob\n"); ++nfail; break; }
    for (int x : P.diag_pos) if (x < 0 || x >= P.nnz_lu) { printf("diag oob\n"); ++nfail; break; }
    for (int x : P.l_k) if (x < 0 || x >= n) { printf("l_k oob\n"); ++nfail; break; }
    for (int x : P.u_col) if (x < 0 || x >= n) { printf("u_col oob\n"); ++nfail; break; }
    for (int x : P.lvl_rows) if (x < 0 || x >= n) { printf("lvl_rows oob\n"); ++nfail; break; }
    for (int x : P.ulvl_rows) if (x < 0 || x >= n) { printf("ulvl_rows oob\n"); ++nfail; break; }
    if ((int)P.lrow_ptr.size() != n + 1 || (int)P.urow_ptr.size() != n + 1) { printf("rowptr size\n"); ++nfail; }
    if (P.l_upd_ptr.size() != P.l_pos.size() + 1) { printf("l_upd_ptr size %zu vs %zu\n", P.l_upd_ptr.size(), P.l_pos.size()); ++nfail; }
  }
  printf("trials %d singular %d fail %d\n", ntot, nsing, nfail);
}
