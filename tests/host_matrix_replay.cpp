// Replays the sparse plan of ch_sparse_host.hpp on a matrix read from a file (tests/golden/mna_jacobian_*.txt) the way the GPU kernels
// do — static pivot sequence, no search — and checks the residual of a solve.  Regression test for the zero pivots that ideal and
// controlled sources produce in MNA matrices after a few eliminations (scripts/extended_fuzz.py, seed 20095).
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
#include "../cedarsim.jl_amd/csrc/ch_sparse_host.hpp"
using namespace chip;
int main(int argc, char** argv) {
  int bad = 0;
  for (int a = 1; a < argc; ++a) {
    FILE* f = fopen(argv[a], "r"); int n;
    if (!f || fscanf(f, "%d", &n) != 1) { printf("cannot read %s\n", argv[a]); return 2; }
    std::vector<int> rp(1, 0), ci; std::vector<double> av;
    for (int i = 0; i < n; ++i) { int m; if (fscanf(f, "%d", &m) != 1) return 2; for (int q = 0; q < m; ++q) { int j; double v; if (fscanf(f, "%d %lf", &j, &v) != 2) return 2; ci.push_back(j); av.push_back(v); } rp.push_back((int)ci.size()); }
    fclose(f);
    SparsePlan P;
    if (sparse_analyse(n, rp, ci, av, P) != CH_OK) { printf("%s: analysis failed\n", argv[a]); ++bad; continue; }
    std::vector<double> LU(P.nnz_lu, 0.0), b(n), y(n, 0.0), dx(n, 0.0);
    std::mt19937 rng(1); for (int i = 0; i < n; ++i) b[i] = ((int)(rng() % 200) - 100) / 10.0;
    for (size_t i = 0; i < ci.size(); ++i) LU[P.a2lu[i]] = av[i];
    double minpiv = 1e300;
    for (size_t lv = 0; lv + 1 < P.lvl_ptr.size(); ++lv) for (int r = P.lvl_ptr[lv]; r < P.lvl_ptr[lv + 1]; ++r) {
      const int k = P.lvl_rows[r];
      for (int e = P.lrow_ptr[k]; e < P.lrow_ptr[k + 1]; ++e) {
        const double l = LU[P.l_pos[e]] / LU[P.diag_pos[P.l_k[e]]];
        for (int p = P.l_upd_ptr[e]; p < P.l_upd_ptr[e + 1]; ++p) LU[P.upd_dst[p]] -= l * LU[P.upd_src[p]];
        LU[P.l_pos[e]] = l;
      }
      minpiv = std::min(minpiv, std::fabs(LU[P.diag_pos[k]]));
    }
    for (size_t lv = 0; lv + 1 < P.lvl_ptr.size(); ++lv) for (int r = P.lvl_ptr[lv]; r < P.lvl_ptr[lv + 1]; ++r) {
      const int k = P.lvl_rows[r]; double s2 = b[P.prow[k]];
      for (int e = P.lrow_ptr[k]; e < P.lrow_ptr[k + 1]; ++e) s2 -= LU[P.l_pos[e]] * y[P.l_k[e]];
      y[k] = s2;
    }
    for (size_t lv = 0; lv + 1 < P.ulvl_ptr.size(); ++lv) for (int r = P.ulvl_ptr[lv]; r < P.ulvl_ptr[lv + 1]; ++r) {
      const int k = P.ulvl_rows[r]; double s2 = y[k];
      for (int e = P.urow_ptr[k]; e < P.urow_ptr[k + 1]; ++e) s2 -= LU[P.u_pos[e]] * dx[P.pcol[P.u_col[e]]];
      dx[P.pcol[k]] = s2 / LU[P.diag_pos[k]];
    }
    double rmax = 0, bmax = 0; bool finite = true;
    for (int i = 0; i < n; ++i) { double s = -b[i]; for (int p = rp[i]; p < rp[i + 1]; ++p) s += av[p] * dx[ci[p]]; if (!std::isfinite(s)) finite = false; rmax = std::max(rmax, std::fabs(s)); bmax = std::max(bmax, std::fabs(b[i])); }
    printf("%s: n %d nnz(L+U) %d min |pivot| %.3e residual %.3e of %.3e\n", argv[a], n, P.nnz_lu, minpiv, rmax, bmax);
    if (!finite || !(minpiv > 0.0) || !(rmax <= 1e-8 * bmax)) ++bad;
  }
  printf("%d bad\n", bad);
  return bad ? 1 : 0;
}
