// ch_sparse_host.hpp — host half of the sparse path (pure C++, no HIP): the KLU-style analysis that runs once per circuit
// and analysis kind.  Maximum transversal on the numerically significant entries, minimum-degree ordering of the symmetrised
// pattern, row-wise symbolic factorisation with fill, dependency levels and the flat per-row operation lists the GPU kernels of
// ch_sparse.hpp replay at every Newton iteration.  Kept free of device code so that the CPU test-suite can run it under
// AddressSanitizer (tests/test_host_analysis_fuzz.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <vector>

#include "../../include/cedarhip.h"

namespace chip {


struct SparsePlan {
  int n = 0;
  // CSR pattern of A in unknown space + gather lists
  std::vector<int> rowptr, colidx;
  std::vector<int> mat_gptr, mat_gsrc;  // per nnz: staging offsets (G slot; C slot = +16)
  std::vector<int> vec_gptr, vec_gsrc;  // per row: staging offsets (F slot; Q slot = +4)
  // LU structure in pivot space
  std::vector<int> prow, pcol;          // pivot step k -> original row / column
  std::vector<int> a2lu;                // per nnz of A: position in LU values
  int nnz_lu = 0;
  std::vector<int> diag_pos;            // per pivot row: position of u_kk
  // rows grouped by level (factorisation and forward solve) and by reverse level (backward solve)
  std::vector<int> lvl_ptr, lvl_rows, ulvl_ptr, ulvl_rows;
  // per pivot row: its L entries in ascending k
  std::vector<int> lrow_ptr;            // [n+1] into the L-entry arrays
  std::vector<int> l_pos, l_k, l_upd_ptr;   // per L entry: position of l_ik, pivot row k, [e..e+1] into upd arrays
  std::vector<int> upd_dst, upd_src;    // positions in LU values
  // per pivot row: its U entries (excluding the diagonal): positions and pivot-space columns
  std::vector<int> urow_ptr, u_pos, u_col;
  bool valid = false;
};

// Maximum transversal (augmenting DFS) restricted to "usable" entries; returns row_of_col or empty on failure.
inline bool max_transversal(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<char>& usable,
                            std::vector<int>& row_of_col) {
  std::vector<int> col_of_row(n, -1);
  row_of_col.assign(n, -1);
  // cheap assignment: prefer the diagonal
  for (int i = 0; i < n; ++i) for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) if (usable[p] && colidx[p] == i && row_of_col[i] < 0) { row_of_col[i] = i; col_of_row[i] = i; }
  std::vector<int> visited(n, -1), par_row(n, -1), stack_row, stack_p;
  for (int r0 = 0; r0 < n; ++r0) {
    if (col_of_row[r0] >= 0) continue;
    // iterative DFS for an augmenting path from the unmatched row r0 (MC21-style)
    stack_row.assign(1, r0); stack_p.assign(1, rowptr[r0]);
    int found = -1;
    while (!stack_row.empty() && found < 0) {
      const int r = stack_row.back();
      int& p = stack_p.back();
      bool pushed = false;
      while (p < rowptr[r + 1]) {
        const int q = p++;
        if (!usable[q]) continue;
        const int c = colidx[q];
        if (visited[c] == r0) continue;
        visited[c] = r0; par_row[c] = r;
        if (row_of_col[c] < 0) { found = c; break; }
        stack_row.push_back(row_of_col[c]); stack_p.push_back(rowptr[row_of_col[c]]);
        pushed = true;
        break;
      }
      if (found < 0 && !pushed) { stack_row.pop_back(); stack_p.pop_back(); }
    }
    if (found < 0) return false;
    for (int c = found;;) {  // flip the matching along the path back to r0
      const int r = par_row[c], prev_c = col_of_row[r];
      row_of_col[c] = r; col_of_row[r] = c;
      if (r == r0) break;
      c = prev_c;
    }
  }
  return true;
}

// Build the plan.  aval: numeric values of A (same order as colidx) used to pick significant entries.
inline int sparse_analyse(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<double>& aval, SparsePlan& P) {
  P.n = n; P.rowptr = rowptr; P.colidx = colidx;
  const int nnz = (int)colidx.size();
  // 1. zero-free diagonal on significant entries (|a| >= 1e-3 of the row maximum), fall back to any structural entry
  std::vector<char> usable(nnz, 0);
  for (int i = 0; i < n; ++i) {
    double mx = 0; for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) mx = std::max(mx, std::fabs(aval[p]));
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) usable[p] = (mx > 0 && std::fabs(aval[p]) >= 1e-3 * mx);
  }
  std::vector<int> row_of_col;
  if (!max_transversal(n, rowptr, colidx, usable, row_of_col)) {
    std::fill(usable.begin(), usable.end(), 1);
    if (!max_transversal(n, rowptr, colidx, usable, row_of_col)) return CH_ERR_SINGULAR;
  }
  // B = Pr*A with B(c,:) = A(row_of_col[c],:)  → diagonal entry (c,c) present
  // 2. minimum-degree ordering on the pattern of B + B^T
  std::vector<std::set<int>> adj(n);
  for (int c = 0; c < n; ++c) { const int r = row_of_col[c]; for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) { const int j = colidx[p]; if (j != c) { adj[c].insert(j); adj[j].insert(c); } } }
  std::vector<int> order; order.reserve(n);
  {
    std::vector<char> gone(n, 0);
    std::set<std::pair<int, int>> pq;
    for (int v = 0; v < n; ++v) pq.insert({(int)adj[v].size(), v});
    while (!pq.empty()) {
      const int v = pq.begin()->second; pq.erase(pq.begin());
      gone[v] = 1; order.push_back(v);
      std::vector<int> nb(adj[v].begin(), adj[v].end());
      for (int u : nb) { pq.erase({(int)adj[u].size(), u}); adj[u].erase(v); }
      for (size_t i = 0; i < nb.size(); ++i) for (size_t j = i + 1; j < nb.size(); ++j) { adj[nb[i]].insert(nb[j]); adj[nb[j]].insert(nb[i]); }
      for (int u : nb) pq.insert({(int)adj[u].size(), u});
      adj[v].clear();
    }
  }
  P.pcol = order;
  P.prow.resize(n);
  std::vector<int> pos_of_col(n);
  for (int k = 0; k < n; ++k) { pos_of_col[order[k]] = k; P.prow[k] = row_of_col[order[k]]; }
  // 3. symbolic row-wise factorisation in pivot space
  std::vector<std::vector<int>> rowpat(n);  // sorted pivot-space columns of row k of L+U
  for (int k = 0; k < n; ++k) {
    std::set<int> pat;
    const int r = P.prow[k];
    for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) pat.insert(pos_of_col[colidx[p]]);
    pat.insert(k);
    for (auto it = pat.begin(); it != pat.end() && *it < k; ++it) {
      const int kk = *it;
      for (int j : rowpat[kk]) if (j > kk) pat.insert(j);
    }
    rowpat[k].assign(pat.begin(), pat.end());
  }
  std::vector<int> lu_ptr(n + 1, 0);
  for (int k = 0; k < n; ++k) lu_ptr[k + 1] = lu_ptr[k] + (int)rowpat[k].size();
  P.nnz_lu = lu_ptr[n];
  auto find_pos = [&](int row, int col) { const auto& rp = rowpat[row]; return lu_ptr[row] + (int)(std::lower_bound(rp.begin(), rp.end(), col) - rp.begin()); };
  P.a2lu.assign(nnz, -1);
  for (int k = 0; k < n; ++k) { const int r = P.prow[k]; for (int p = rowptr[r]; p < rowptr[r + 1]; ++p) P.a2lu[p] = find_pos(k, pos_of_col[colidx[p]]); }
  P.diag_pos.resize(n);
  for (int k = 0; k < n; ++k) P.diag_pos[k] = find_pos(k, k);
  // 4. op lists and levels
  std::vector<int> level(n, 0), ulevel(n, 0);
  P.lrow_ptr.assign(n + 1, 0); P.urow_ptr.assign(n + 1, 0);
  P.l_pos.clear(); P.l_k.clear(); P.l_upd_ptr.clear(); P.upd_dst.clear(); P.upd_src.clear(); P.u_pos.clear(); P.u_col.clear();
  P.l_upd_ptr.push_back(0);
  for (int k = 0; k < n; ++k) {
    int lv = 0;
    for (int j : rowpat[k]) {
      if (j < k) {
        lv = std::max(lv, level[j] + 1);
        P.l_pos.push_back(find_pos(k, j)); P.l_k.push_back(j);
        for (int c : rowpat[j]) if (c > j) { P.upd_dst.push_back(find_pos(k, c)); P.upd_src.push_back(find_pos(j, c)); }
        P.l_upd_ptr.push_back((int)P.upd_dst.size());
      } else if (j > k) { P.u_pos.push_back(find_pos(k, j)); P.u_col.push_back(j); }
    }
    level[k] = lv;
    P.lrow_ptr[k + 1] = (int)P.l_pos.size();
    P.urow_ptr[k + 1] = (int)P.u_pos.size();
  }
  for (int k = n - 1; k >= 0; --k) { int lv = 0; for (int j : rowpat[k]) if (j > k) lv = std::max(lv, ulevel[j] + 1); ulevel[k] = lv; }
  auto group = [&](const std::vector<int>& lev, std::vector<int>& ptr, std::vector<int>& rows) {
    int nl = 0; for (int v : lev) nl = std::max(nl, v + 1);
    ptr.assign(nl + 1, 0);
    for (int v : lev) ptr[v + 1]++;
    for (int i = 0; i < nl; ++i) ptr[i + 1] += ptr[i];
    rows.resize(n);
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int k = 0; k < n; ++k) rows[fill[lev[k]]++] = k;
  };
  group(level, P.lvl_ptr, P.lvl_rows);
  group(ulevel, P.ulvl_ptr, P.ulvl_rows);
  P.valid = true;
  return CH_OK;
}

// ------------------------------------------------------------------------------------------------

}  // namespace chip
