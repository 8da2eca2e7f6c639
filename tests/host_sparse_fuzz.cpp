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
  int nreplay = 0; double worst = 0.0;
  for (int trial = 0; trial < 2400; ++trial) {
    // the first half of the trials are the degenerate sizes of the forced-sparse DC tests (1..3 unknowns: a V-R circuit is
    // [[g, 1], [1, 0]] — a structurally present but numerically ZERO diagonal in the branch row), the rest 1..14
    const bool tiny = trial < 1200;
    int n = tiny ? 1 + rng() % 3 : 1 + rng() % 14;
    double dens = (rng() % 100) / 100.0;
    std::vector<int> rp(1, 0), ci; std::vector<double> av;
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) {
        bool diag = i == j;
        if (diag || (rng() % 1000) / 1000.0 < dens) {
          ci.push_back(j);
          int r = rng() % 10;
          double v = r < 2 ? 0.0 : r < 4 ? 1e-15 : ((int)(rng() % 2000) - 1000) / 100.0;
          if (tiny && diag && rng() % 2) v = 0.0;   // zero diagonal: the transversal has to move the pivot off it
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
    // replay the plan the way sp_lu_solve_kernel does (scatter, level-ordered row elimination with static pivots, forward and
    // backward substitution) and compare with a dense partial-pivoting solve of the same system
    {
      std::vector<double> LU(P.nnz_lu, 0.0), b(n), y(n, 0.0), dx(n, 0.0);
      for (int i = 0; i < n; ++i) b[i] = ((int)(rng() % 200) - 100) / 10.0;
      for (size_t i = 0; i < ci.size(); ++i) LU[P.a2lu[i]] = av[i];
      bool sing = false;
      for (size_t lv = 0; lv + 1 < P.lvl_ptr.size(); ++lv) for (int r = P.lvl_ptr[lv]; r < P.lvl_ptr[lv + 1]; ++r) {
        const int k = P.lvl_rows[r];
        for (int e = P.lrow_ptr[k]; e < P.lrow_ptr[k + 1]; ++e) {
          const double l = LU[P.l_pos[e]] / LU[P.diag_pos[P.l_k[e]]];
          for (int p = P.l_upd_ptr[e]; p < P.l_upd_ptr[e + 1]; ++p) LU[P.upd_dst[p]] -= l * LU[P.upd_src[p]];
          LU[P.l_pos[e]] = l;
        }
        const double ukk = LU[P.diag_pos[k]];
        if (!(std::fabs(ukk) > 0.0) || !(std::fabs(ukk) < 1e300)) sing = true;
      }
      if (!sing) {
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
        // dense reference with partial pivoting; only well-conditioned systems are compared
        std::vector<double> D((size_t)n * (n + 1), 0.0);
        for (int i = 0; i < n; ++i) { for (int p = rp[i]; p < rp[i + 1]; ++p) D[(size_t)i * (n + 1) + ci[p]] += av[p]; D[(size_t)i * (n + 1) + n] = b[i]; }
        bool ok = true; double pmin = 1e300, pmax = 0.0;
        for (int k = 0; k < n && ok; ++k) {
          int bi = k; for (int i = k + 1; i < n; ++i) if (std::fabs(D[(size_t)i * (n + 1) + k]) > std::fabs(D[(size_t)bi * (n + 1) + k])) bi = i;
          if (bi != k) for (int j = 0; j <= n; ++j) std::swap(D[(size_t)k * (n + 1) + j], D[(size_t)bi * (n + 1) + j]);
          const double pv = D[(size_t)k * (n + 1) + k];
          if (std::fabs(pv) < 1e-9) { ok = false; break; }
          pmin = std::min(pmin, std::fabs(pv)); pmax = std::max(pmax, std::fabs(pv));
          for (int i = k + 1; i < n; ++i) { const double l = D[(size_t)i * (n + 1) + k] / pv; for (int j = k; j <= n; ++j) D[(size_t)i * (n + 1) + j] -= l * D[(size_t)k * (n + 1) + j]; }
        }
        if (ok && pmax / pmin < 1e6) {
          std::vector<double> xr(n);
          for (int k = n - 1; k >= 0; --k) { double s2 = D[(size_t)k * (n + 1) + n]; for (int j = k + 1; j < n; ++j) s2 -= D[(size_t)k * (n + 1) + j] * xr[j]; xr[k] = s2 / D[(size_t)k * (n + 1) + k]; }
          double err = 0, nrm = 1e-300; bool finite = true;
          for (int i = 0; i < n; ++i) { if (!std::isfinite(dx[i])) finite = false; err = std::max(err, std::fabs(dx[i] - xr[i])); nrm = std::max(nrm, std::fabs(xr[i])); }
          // static pivots chosen from |a| >= 1e-3 * row max can lose ~3 digits against partial pivoting; more is a plan bug
          if (finite) { ++nreplay; worst = std::max(worst, err / nrm); if (err / nrm > 1e-5) { printf("replay mismatch n=%d err=%g\n", n, err / nrm); ++nfail; } }
          // the level-synchronous form (sp2_* kernels): A-items and B-items level by level, then the solves with the scaled L
          if (finite) {
            std::vector<double> a2(P.nnz_lu, 0.0), Lv(P.nnz_lu, 0.0), y2(n, 0.0), z2(n, 0.0);
            for (size_t i = 0; i < ci.size(); ++i) a2[P.a2lu[i]] = av[i];
            const int nl = (int)P.lvl_ptr.size() - 1;
            for (int l = 0; l < P.n_rlvl; ++l) {
              std::vector<double> before = a2;   // a launch reads only values no item of the same launch writes: check it
              for (int t = P.la_ptr[l]; t < P.la_ptr[l + 1]; ++t) Lv[P.la_pos[t]] = before[P.la_pos[t]] / before[P.la_diag[t]];
              for (int it = P.lb_ptr[l]; it < P.lb_ptr[l + 1]; ++it) {
                double s2 = 0;
                for (int q = P.lb_sptr[it]; q < P.lb_sptr[it + 1]; ++q) {
                  if (a2[P.lb_l[q]] != before[P.lb_l[q]] || a2[P.lb_u[q]] != before[P.lb_u[q]] || a2[P.lb_d[q]] != before[P.lb_d[q]]) { printf("level %d reads a value it also writes\n", l); ++nfail; }
                  s2 += (before[P.lb_l[q]] / before[P.lb_d[q]]) * before[P.lb_u[q]];
                }
                a2[P.lb_dst[it]] -= s2;
              }
              if (P.lb_ptr[l + 1] - P.lb_ptr[l] < P.lb_nheavy[l]) { printf("heavy count\n"); ++nfail; }
            }
            for (int l = 0; l < nl; ++l) for (int r = P.fl_ptr[l]; r < P.fl_ptr[l + 1]; ++r) {
              const int k = P.fl_rows[r]; double s2 = b[P.prow[k]];
              for (int e = P.lrow_ptr[k]; e < P.lrow_ptr[k + 1]; ++e) s2 -= Lv[P.l_pos[e]] * y2[P.l_k[e]];
              y2[k] = s2;
            }
            for (size_t l = 0; l + 1 < P.bl_ptr.size(); ++l) for (int r = P.bl_ptr[l]; r < P.bl_ptr[l + 1]; ++r) {
              const int k = P.bl_rows[r]; double s2 = y2[k];
              for (int e = P.urow_ptr[k]; e < P.urow_ptr[k + 1]; ++e) s2 -= a2[P.u_pos[e]] * z2[P.pcol[P.u_col[e]]];
              z2[P.pcol[k]] = s2 / a2[P.diag_pos[k]];
            }
            double e2 = 0;
            for (int i = 0; i < n; ++i) e2 = std::max(e2, std::fabs(z2[i] - dx[i]));
            if (!(e2 <= 1e-9 * nrm + 1e-300)) { printf("level-synchronous replay differs from the row-wise one: n=%d err=%g\n", n, e2 / nrm); ++nfail; }
          }
        }
      }
    }
  }
  printf("replayed %d worst %.3g\n", nreplay, worst);
  // ---- MNA-shaped systems: a conductance network, ideal voltage sources (branch row and column of +-1 around a ZERO diagonal),
  //      VCVS constraint rows and VCCS entries.  Entries that are large in A cancel to exactly zero after a few eliminations here, which
  //      is what a static pivot sequence taken from a matching alone divides by (scripts/extended_fuzz.py, seed 20095): the sequence has
  //      to come from an elimination of the values (numeric_pivot_rows).  Checked by the residual of a solve with the replayed plan.
  int mna_done = 0, mna_bad = 0; double mna_worst = 0.0;
  for (int trial = 0; trial < 400; ++trial) {
    const int nn = 4 + rng() % 40, nv = 1 + rng() % 6, ne = rng() % 4, n = nn + nv + ne;
    std::vector<std::vector<double>> D(n, std::vector<double>(n, 0.0));
    auto u = [&]() { return (rng() % 100000) / 100000.0; };
    auto stamp_g = [&](int a, int b, double g) { if (a >= 0) D[a][a] += g; if (b >= 0) D[b][b] += g; if (a >= 0 && b >= 0) { D[a][b] -= g; D[b][a] -= g; } };
    for (int i = 0; i < nn; ++i) stamp_g(i, i == 0 ? -1 : (int)(rng() % i), std::pow(10.0, -5.0 + 3.0 * u()));      // spanning tree to ground
    for (int k = 0; k < 2 * nn; ++k) { const int a = rng() % nn, b = (rng() % 5 == 0) ? -1 : (int)(rng() % nn); if (a != b) stamp_g(a, b, std::pow(10.0, -5.0 + 3.0 * u())); }
    for (int k = 0; k < nn / 3; ++k) { const int a = rng() % nn, b = rng() % nn, c1 = rng() % nn, c2 = rng() % nn; const double g = 1e-4 * (u() - 0.5); D[a][c1] += g; D[a][c2] -= g; D[b][c1] -= g; D[b][c2] += g; }   // VCCS
    for (int k = 0; k < nv; ++k) { const int r = nn + k, a = rng() % nn, b = (rng() % 2) ? -1 : (int)(rng() % nn); D[r][a] += 1.0; D[a][r] += 1.0; if (b >= 0 && b != a) { D[r][b] -= 1.0; D[b][r] -= 1.0; } }   // V source
    for (int k = 0; k < ne; ++k) { const int r = nn + nv + k, a = rng() % nn, c1 = rng() % nn, c2 = rng() % nn; const double gain = u() - 0.5; D[r][a] += 1.0; D[a][r] += 1.0; D[r][c1] -= gain; D[r][c2] += gain; }   // VCVS
    std::vector<int> rp(1, 0), ci; std::vector<double> av;
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) if (D[i][j] != 0.0 || i == j) { ci.push_back(j); av.push_back(D[i][j]); } rp.push_back((int)ci.size()); }
    // dense partial pivoting: is the system regular at all?
    std::vector<std::vector<double>> W = D; bool reg = true; double pmin = 1e300, pmax = 0.0;
    for (int k = 0; k < n && reg; ++k) {
      int bi = k; for (int i = k + 1; i < n; ++i) if (std::fabs(W[i][k]) > std::fabs(W[bi][k])) bi = i;
      std::swap(W[k], W[bi]);
      if (std::fabs(W[k][k]) < 1e-14) { reg = false; break; }
      pmin = std::min(pmin, std::fabs(W[k][k])); pmax = std::max(pmax, std::fabs(W[k][k]));
      for (int i = k + 1; i < n; ++i) { const double l = W[i][k] / W[k][k]; if (l != 0.0) for (int j = k; j < n; ++j) W[i][j] -= l * W[k][j]; }
    }
    if (!reg || pmax / pmin > 1e9) continue;
    SparsePlan P;
    if (sparse_analyse(n, rp, ci, av, P) != CH_OK) { printf("MNA trial %d: analysis failed on a regular system\n", trial); ++mna_bad; continue; }
    std::vector<double> LU(P.nnz_lu, 0.0), b(n), y(n, 0.0), dx(n, 0.0);
    for (int i = 0; i < n; ++i) b[i] = ((int)(rng() % 200) - 100) / 10.0;
    for (size_t i = 0; i < ci.size(); ++i) LU[P.a2lu[i]] = av[i];
    for (size_t lv = 0; lv + 1 < P.lvl_ptr.size(); ++lv) for (int r = P.lvl_ptr[lv]; r < P.lvl_ptr[lv + 1]; ++r) {
      const int k = P.lvl_rows[r];
      for (int e = P.lrow_ptr[k]; e < P.lrow_ptr[k + 1]; ++e) {
        const double l = LU[P.l_pos[e]] / LU[P.diag_pos[P.l_k[e]]];
        for (int p = P.l_upd_ptr[e]; p < P.l_upd_ptr[e + 1]; ++p) LU[P.upd_dst[p]] -= l * LU[P.upd_src[p]];
        LU[P.l_pos[e]] = l;
      }
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
    double rmax = 0.0, scale = 1e-300; bool finite = true;
    for (int i = 0; i < n; ++i) {
      double s2 = -b[i], rowabs = std::fabs(b[i]);
      for (int j = 0; j < n; ++j) { s2 += D[i][j] * dx[j]; rowabs += std::fabs(D[i][j] * dx[j]); }
      if (!std::isfinite(s2)) finite = false;
      rmax = std::max(rmax, std::fabs(s2)); scale = std::max(scale, rowabs);
    }
    ++mna_done; mna_worst = std::max(mna_worst, rmax / scale);
    if (!finite || rmax > 1e-9 * scale) { printf("MNA trial %d (n %d): residual %.3e of %.3e\n", trial, n, rmax, scale); ++mna_bad; }
  }
  printf("MNA systems replayed %d worst relative residual %.3g bad %d\n", mna_done, mna_worst, mna_bad);
  nfail += mna_bad;
  printf("trials %d singular %d fail %d\n", ntot, nsing, nfail);
}
