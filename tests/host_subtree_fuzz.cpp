// Host-only fuzz of the SUBTREE form of the sparse analysis (cedarsim.jl_amd/csrc/ch_sparse_host.hpp, SubtreePlan): "arrow" matrices —
// many independent diagonal blocks (the tiles of an array) under a border of one to three rows and columns that touch every block
// (shared rails) — analysed, then replayed on the host exactly the way sp3_group_kernel / sp3_top_kernel / sp3_back_kernel of
// ch_sparse.hpp do (one group at a time: staged values, flat elimination steps, forward rows, Schur slots; dense top solve; backward
// rows) and compared with a dense solve with partial pivoting.  Every index of every blob must stay inside its arrays (the sanitizers
// of tests/test_host_analysis_fuzz.py see to that).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "cedarhip.h"
#include "ch_sparse_host.hpp"
using namespace chip;

static bool dense_solve(int n, std::vector<double> a, std::vector<double> b, std::vector<double>& x) {
  for (int k = 0; k < n; ++k) {
    int bi = k; for (int i = k + 1; i < n; ++i) if (std::fabs(a[i * n + k]) > std::fabs(a[bi * n + k])) bi = i;
    if (!(std::fabs(a[bi * n + k]) > 1e-300)) return false;
    for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[bi * n + j]);
    std::swap(b[k], b[bi]);
    for (int i = k + 1; i < n; ++i) { const double l = a[i * n + k] / a[k * n + k]; for (int j = k; j < n; ++j) a[i * n + j] -= l * a[k * n + j]; b[i] -= l * b[k]; }
  }
  x.assign(n, 0.0);
  for (int k = n - 1; k >= 0; --k) { double s = b[k]; for (int j = k + 1; j < n; ++j) s -= a[k * n + j] * x[j]; x[k] = s / a[k * n + k]; }
  return true;
}

int main() {
  std::mt19937 rng(4242);
  auto uni = [&](double lo, double hi) { return lo + (hi - lo) * (rng() % 100000) / 100000.0; };
  int nfail = 0, nvalid = 0; double worst = 0.0;
  for (int trial = 0; trial < 60; ++trial) {
    const int ng = 64 + rng() % 40, nb = 1 + rng() % 3;
    std::vector<int> bsz(ng), bofs(ng);
    int n = 0;
    for (int g = 0; g < ng; ++g) { bsz[g] = 1 + rng() % 9; bofs[g] = n; n += bsz[g]; }
    const int border0 = n; n += nb;
    std::vector<double> M((size_t)n * n, 0.0);
    std::vector<char> nzp((size_t)n * n, 0);
    auto set = [&](int i, int j, double v) { M[(size_t)i * n + j] = v; nzp[(size_t)i * n + j] = 1; };
    for (int g = 0; g < ng; ++g) {
      for (int i = 0; i < bsz[g]; ++i) for (int j = 0; j < bsz[g]; ++j) {
        const int I = bofs[g] + i, J = bofs[g] + j;
        if (i == j) set(I, J, uni(4.0, 9.0) * (rng() % 2 ? 1 : -1));
        else if (rng() % 100 < 45) set(I, J, uni(-1.0, 1.0));
      }
      for (int t = 0; t < nb; ++t) {   // every block touches every border row and column at least once
        const int i = bofs[g] + rng() % bsz[g], j = bofs[g] + rng() % bsz[g];
        set(i, border0 + t, uni(-1.0, 1.0)); set(border0 + t, j, uni(-1.0, 1.0));
        if (rng() % 3 == 0) { const int i2 = bofs[g] + rng() % bsz[g]; set(i2, border0 + t, uni(-1.0, 1.0)); }
      }
    }
    for (int t = 0; t < nb; ++t) for (int u = 0; u < nb; ++u) if (t == u || rng() % 2) set(border0 + t, border0 + u, t == u ? uni(30.0, 60.0) : uni(-1.0, 1.0));
    std::vector<int> rp(1, 0), ci; std::vector<double> av;
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) if (nzp[(size_t)i * n + j]) { ci.push_back(j); av.push_back(M[(size_t)i * n + j]); } rp.push_back((int)ci.size()); }
    std::vector<double> rhs(n); for (double& v : rhs) v = uni(-1.0, 1.0);
    SparsePlan P;
    if (sparse_analyse(n, rp, ci, av, P) != CH_OK) { printf("analysis failed\n"); ++nfail; continue; }
    const SubtreePlan& T = P.sub;
    if (!T.valid) { printf("trial %d: subtree form not found (groups %d, top %d)\n", trial, T.n_groups, T.nT); ++nfail; continue; }
    ++nvalid;
    // (the top set is closed under "needed by a top row", and a block that is not connected inside splits: never fewer than these)
    if (T.nT < nb || T.nT > 16 || T.n_groups < ng) { printf("trial %d: %d groups / %d top rows for %d blocks / %d border rows\n", trial, T.n_groups, T.nT, ng, nb); ++nfail; }
    // ---- replay ----
    const int nT = T.nT, ne = nT * nT + nT;
    std::vector<double> LUv((size_t)P.nnz_lu, 0.0), y(n, 0.0), dx(n, 0.0), schur((size_t)ne * T.n_groups, 0.0), xT(nT, 0.0);
    bool sing = false;
    for (int g = 0; g < T.n_groups; ++g) {   // sp3_group_kernel
      const int* B = T.blob.data() + T.blob_ptr[g];
      const Sp3Blob b(B, nT);
      if (b.nv > T.max_nv || T.blob_ptr[g + 1] - T.blob_ptr[g] > T.max_blob) { printf("blob larger than the recorded maximum\n"); ++nfail; }
      if (b.dx_idx + b.np > T.blob.data() + T.blob_ptr[g + 1]) { printf("blob views run past the blob\n"); ++nfail; }
      std::vector<double> val(b.nv, 0.0);
      for (int v = 0; v < b.nv; ++v) { const int ai = b.a_idx[v]; val[v] = ai >= 0 ? av.at(ai) : 0.0; }
      for (int i = 0; i < b.np; ++i) val.at(b.y0 + i) = rhs.at(b.rhs_idx[i]);
      for (int pi = 0; pi < b.np; ++pi) {
        const double pv = val.at(b.piv_dp[pi]);
        if (!(std::fabs(pv) > 0.0)) sing = true;
        std::vector<char> written(b.nv, 0), read(b.nv, 0);   // a step must not write what it reads (one barrier per pivot on the GPU)
        for (int u = b.fu_ptr[pi]; u < b.fu_ptr[pi + 1]; ++u) { const int w = b.fu_ds[u]; read.at(b.fu_lp[u]) = 1; read.at(w & 0xffff) = 1; read.at(b.piv_dp[pi]) = 1; }
        for (int u = b.fu_ptr[pi]; u < b.fu_ptr[pi + 1]; ++u) {
          const int w = b.fu_ds[u];
          if (read.at(w >> 16) || written.at(w >> 16)) { printf("pivot step %d of group %d writes a slot it reads or writes twice\n", pi, g); ++nfail; }
          written.at(w >> 16) = 1;
          const double l = val.at(b.fu_lp[u]) / pv;
          val.at(w >> 16) -= l * val.at(w & 0xffff);
        }
      }
      for (int v = 0; v < b.n_own; ++v) LUv.at(b.lu_pos[v]) = val[v];
      for (int i = 0; i < b.np; ++i) y.at(b.rowk[i]) = val.at(b.y0 + i);
      for (int j = 0; j < ne; ++j) schur[(size_t)j * T.n_groups + g] = j < nT * nT ? val.at(b.schur0 + j) : val.at(b.acc0 + (j - nT * nT));
    }
    {   // sp3_top_kernel
      std::vector<double> S((size_t)nT * nT), g(nT);
      for (int j = 0; j < ne; ++j) {
        double a = 0.0; for (int gg = 0; gg < T.n_groups; ++gg) a += schur[(size_t)j * T.n_groups + gg];
        if (j < nT * nT) { const int ai = T.top_a_idx[j]; S[j] = (ai >= 0 ? av.at(ai) : 0.0) + a; }
        else g[j - nT * nT] = rhs.at(P.prow[T.top_rows[j - nT * nT]]) + a;
      }
      if (!dense_solve(nT, S, g, xT)) sing = true;
      for (int t = 0; t < nT; ++t) dx.at(P.pcol[T.top_rows[t]]) = xT[t];
    }
    for (int g = 0; g < T.n_groups && !sing; ++g) {   // sp3_back_kernel
      const Sp3Blob b(T.blob.data() + T.blob_ptr[g], nT);
      std::vector<double> val(b.nv, 0.0);
      for (int v = 0; v < b.n_own; ++v) val[v] = LUv.at(b.lu_pos[v]);
      for (int i = 0; i < b.np; ++i) val.at(b.y0 + i) = y.at(b.rowk[i]);
      for (int t = 0; t < nT; ++t) val.at(b.acc0 + t) = xT[t];
      for (int t = 0; t < nT; ++t) {
        std::vector<char> hit(b.nv, 0);
        for (int e = b.bt_ptr[t]; e < b.bt_ptr[t + 1]; ++e) { if (hit.at(b.bt_y[e])) { printf("two entries of one top column in one row\n"); ++nfail; } hit.at(b.bt_y[e]) = 1; val.at(b.bt_y[e]) -= val.at(b.bt_up[e]) * val.at(b.acc0 + t); }
      }
      for (int r = 0; r < b.np; ++r) {
        const int pi = b.np - 1 - r;
        const double x = val.at(b.y0 + pi) / val.at(b.piv_dp[pi]);
        for (int e = b.bc_ptr[r]; e < b.bc_ptr[r + 1]; ++e) { if (b.bc_y[e] == b.y0 + pi) { printf("a pivot's own y among its column targets\n"); ++nfail; } val.at(b.bc_y[e]) -= val.at(b.bc_up[e]) * x; }
        val.at(b.y0 + pi) = x; dx.at(b.dx_idx[pi]) = x;
      }
    }
    std::vector<double> xr;
    if (sing || !dense_solve(n, M, rhs, xr)) continue;
    double err = 0.0, nrm = 0.0;
    for (int i = 0; i < n; ++i) { err = std::max(err, std::fabs(dx[i] - xr[i])); nrm = std::max(nrm, std::fabs(xr[i])); }
    worst = std::max(worst, err / nrm);
    if (!(err <= 1e-8 * nrm)) { printf("trial %d: subtree replay differs from the dense solve: %g\n", trial, err / nrm); ++nfail; }
  }
  printf("subtree form: %d plans replayed, worst relative error %.3g, %d failures\n", nvalid, worst, nfail);
  return nfail ? 1 : 0;
}
