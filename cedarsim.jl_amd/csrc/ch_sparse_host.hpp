// ch_sparse_host.hpp — host half of the sparse path (pure C++, no HIP): the KLU-style analysis that runs once per circuit
// and analysis kind.  Maximum transversal on the numerically significant entries, minimum-degree ordering of the symmetrised
// pattern, row-wise symbolic factorisation with fill, dependency levels and the flat per-row operation lists the GPU kernels of
// ch_sparse.hpp replay at every Newton iteration.  Kept free of device code so that the CPU test-suite can run it under
// AddressSanitizer (tests/test_host_analysis_fuzz.py).
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <numeric>
#include <set>
#include <string>
#include <vector>

#include "../../include/cedarhip.h"

namespace chip {


// ---- subtree form ("sp3"): independent subtrees of the elimination tree under a small top separator --------------------------------
// A tiled array behind shared rails factors as ~1000 identical 11-row subtrees hanging under a 2-row separator.  The
// level-synchronous kernels pay one launch per level for that (13 levels x factor / forward / backward = 40 launches per Newton
// iteration); the elimination tree says the subtrees never talk to each other below the separator.  Here:
//   top set T   rows with a long L part, and everything that depends on them (closed under L and U): the separator;
//   groups      connected components of the remaining rows (edges: every L and U entry between two of them): the subtrees;
//   group g     ONE wavefront, its values staged in LDS: up-looking elimination of its own rows AND of the segments of the top rows
//               that lie in its columns (the top rows' L entries whose pivots belong to g), forward substitution, and its
//               contribution to the Schur complement on T — written to its own slot, so the sum over groups has a fixed order;
//   top         one workgroup: S = A_TT + sum_g contributions, dense LU with partial pivoting, x_T;
//   groups      backward substitution with x_T.
// Three launches per refactorisation + solve.  Everything a group does is described by ONE int blob (copied to LDS first).  The
// elimination inside a group is COLUMN-oriented (right-looking): one step per pivot, in which every lane takes one update
// a_ic -= (a_ik / u_kk) u_kc — or the matching step of the forward substitution, y_i -= (a_ik / u_kk) y_k, which rides along (y_k is
// final when pivot k is reached) — and nothing a step reads is written by it, so a step is one barrier; the multipliers are never
// stored (the backward substitution needs U and the pivots only).  Eleven steps for an 11-row tile instead of 77 L entries.
// blob: header | a_idx[nv] | lu_pos[nv] | piv_dp[np] | fu_ptr[np+1] | fu_ds[] = dst << 16 | src | fu_lp[] | rhs_idx[np] | rowk[np]
//       | bt_ptr[nT+1] | bt_up[] | bt_y[] | bc_ptr[np+1] (pivots descending) | bc_up[] | bc_y[] | dx_idx[np]
// Value slots (doubles in LDS): [entries of own rows | entries (r, k) of the top rows, k in g | Schur accumulators nT x nT | y of
// own rows | forward accumulators of the top rows nT].
struct SubtreePlan {
  bool valid = false;
  int n_groups = 0, nT = 0, max_blob = 0, max_nv = 0, max_rows = 0;
  std::vector<int> top_rows;                 // pivot indices of T, ascending
  std::vector<int> blob, blob_ptr;           // per group [blob_ptr[g], blob_ptr[g+1])
  // top block: S[t][t'] starts from A's own entry (or 0), its right-hand side from rhs[prow[top_rows[t]]]
  std::vector<int> top_a_idx;                // [nT * nT] index into A's values, or -1
  enum { H_NV = 0, H_NOWN, H_NROWS, H_NFU, H_NBT, H_NBC, H_SCHUR, H_Y, H_ACC, H_WORDS = 12 };
};

#if defined(__HIPCC__)
#define CH_SP_HD __host__ __device__
#else
#define CH_SP_HD
#endif
struct Sp3Blob {   // views into a group's blob (LDS)
  int nv, n_own, np, nfu, nbt, nbc, schur0, y0, acc0;
  const int *a_idx, *lu_pos, *piv_dp, *fu_ptr, *fu_ds, *fu_lp, *rhs_idx, *rowk, *bt_ptr, *bt_up, *bt_y, *bc_ptr, *bc_up, *bc_y, *dx_idx;
  CH_SP_HD Sp3Blob(const int* B, int nT) {
    nv = B[SubtreePlan::H_NV]; n_own = B[SubtreePlan::H_NOWN]; np = B[SubtreePlan::H_NROWS]; nfu = B[SubtreePlan::H_NFU]; nbt = B[SubtreePlan::H_NBT]; nbc = B[SubtreePlan::H_NBC];
    schur0 = B[SubtreePlan::H_SCHUR]; y0 = B[SubtreePlan::H_Y]; acc0 = B[SubtreePlan::H_ACC];
    a_idx = B + SubtreePlan::H_WORDS; lu_pos = a_idx + nv; piv_dp = lu_pos + nv; fu_ptr = piv_dp + np; fu_ds = fu_ptr + np + 1; fu_lp = fu_ds + nfu;
    rhs_idx = fu_lp + nfu; rowk = rhs_idx + np; bt_ptr = rowk + np; bt_up = bt_ptr + nT + 1; bt_y = bt_up + nbt;
    bc_ptr = bt_y + nbt; bc_up = bc_ptr + np + 1; bc_y = bc_up + nbc; dx_idx = bc_y + nbc;
  }
};

struct SparsePlan {
  int n = 0;
  SubtreePlan sub;
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
  // ---- level-synchronous right-looking form (the multi-workgroup kernels of ch_sparse.hpp) ----
  // At level l every pivot of the level is final.  A-items: the L entries (i, k) with level(k) == l — l_ik = a_ik / u_kk.
  // B-items: every position that receives updates from pivots of level l, with the list of its (l_pos, u_pos, diag_pos)
  // source triples in a fixed order: a_dst -= sum (a[l_pos] / a[diag_pos]) * a[u_pos].  Items with long lists (a supply rail
  // collects one product per tile) are reduced by a whole wavefront; the others by one thread.
  std::vector<int> la_ptr, la_pos, la_diag;                 // [n_rlvl+1], per A-item
  std::vector<int> lb_ptr;                                  // [n_rlvl+1] into the B-item arrays, light items first inside a level
  std::vector<int> lb_nheavy;                               // [n_rlvl] heavy items of the level (they follow the light ones)
  std::vector<int> lb_dst, lb_sptr;                         // per B-item: destination position, [items+1] into the source arrays
  std::vector<int> lb_l, lb_u, lb_d;                        // per source: l_pos, u_pos, diag_pos
  // triangular solves by level: rows of a level, light rows first, then heavy rows (long L rows: one wavefront each)
  std::vector<int> fl_ptr, fl_rows, fl_nheavy;              // forward:  [n_lvl+1], rows, heavy count per level
  std::vector<int> bl_ptr, bl_rows;                         // backward: [n_ulvl+1], rows (U rows are short here: one thread each)
  int max_level_width = 0, n_rlvl = 0;                      // n_rlvl: levels of the factorisation (la_ptr / lb_ptr have n_rlvl + 1 entries)
  bool wide_levels = false;                                 // true: few, wide levels — use the multi-workgroup kernels
  bool valid = false;
};
constexpr int SP_HEAVY = 48;   // sources per B-item / L entries per row above which a wavefront takes the item

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

// Pivot ROWS from an actual elimination of the given values (what KLU's first factorisation does; the GPU refactorisation then
// reuses the sequence and never searches).  Columns are taken in the fill-reducing order `order`; at step k the pivot of column
// order[k] is the matched row (`row_of_col`, the zero-free-diagonal matching: keeps the pattern the ordering was made for) when its
// current value is at least `tol` of the largest candidate in the column — KLU's rule, tol = 1e-3 — and the largest candidate
// otherwise.  "Current value": after the updates of the earlier pivots, which is where a matching alone goes wrong — an entry that
// is large in A can cancel to exactly zero (ideal sources and controlled sources in MNA do that), and a static sequence built on it
// divides by zero whatever the refactorisation does afterwards.  Right-looking on sparse rows; returns false when a column has no
// non-zero candidate left (numerically singular).
inline bool numeric_pivot_rows(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<double>& aval,
                               const std::vector<int>& order, const std::vector<int>& row_of_col, std::vector<int>& prow, double tol = 1e-3) {
  std::vector<std::map<int, double>> R(n);
  std::vector<std::vector<int>> col_rows(n);          // rows that hold (or held) an entry in the column: filtered when used
  std::vector<double> rowmax(n, 0.0);
  for (int i = 0; i < n; ++i) for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) {
    auto it = R[i].find(colidx[p]);
    if (it == R[i].end()) { R[i][colidx[p]] = aval[p]; col_rows[colidx[p]].push_back(i); } else it->second += aval[p];
    rowmax[i] = std::max(rowmax[i], std::fabs(aval[p]));
  }
  std::vector<char> done(n, 0);
  prow.assign(n, -1);
  for (int k = 0; k < n; ++k) {
    const int c = order[k];
    std::vector<int> cand;
    double amax = 0.0;
    std::sort(col_rows[c].begin(), col_rows[c].end());
    col_rows[c].erase(std::unique(col_rows[c].begin(), col_rows[c].end()), col_rows[c].end());
    for (int i : col_rows[c]) {
      if (done[i]) continue;
      auto it = R[i].find(c);
      if (it == R[i].end()) continue;
      cand.push_back(i);
      const double v = std::fabs(it->second);
      if (v > 1e-13 * rowmax[i]) amax = std::max(amax, v);     // below that: what a cancellation left behind, never a pivot
    }
    if (!(amax > 0.0) || !(amax < 1e300)) return false;
    int pick = -1;
    { const int m = row_of_col[c];
      if (m >= 0 && !done[m]) { auto it = R[m].find(c); if (it != R[m].end() && std::fabs(it->second) >= tol * amax && std::fabs(it->second) > 1e-13 * rowmax[m]) pick = m; } }
    if (pick < 0) for (int i : cand) { const double v = std::fabs(R[i][c]); if (v == amax) { pick = i; break; } }   // the largest, lowest row on ties
    if (pick < 0) return false;
    prow[k] = pick; done[pick] = 1;
    const std::map<int, double>& pr = R[pick];
    const double pv = pr.at(c);
    for (int i : cand) {
      if (i == pick) continue;
      std::map<int, double>& ri = R[i];
      const double l = ri[c] / pv;
      ri.erase(c);
      if (l == 0.0) continue;
      for (const auto& e : pr) {
        if (e.first == c) continue;
        auto it = ri.find(e.first);
        if (it == ri.end()) { ri[e.first] = -l * e.second; col_rows[e.first].push_back(i); } else it->second -= l * e.second;
      }
    }
  }
  return true;
}

// Build the plan.  aval: numeric values of A (same order as colidx) used to pick significant entries.
inline int sparse_analyse(int n, const std::vector<int>& rowptr, const std::vector<int>& colidx, const std::vector<double>& aval, SparsePlan& P) {
  P.n = n; P.rowptr = rowptr; P.colidx = colidx;
  const int nnz = (int)colidx.size();
  // 1. zero-free diagonal on entries that are LARGE in their row: the pivots are static (the GPU refactorisation never searches), so
  //    the matching is all the pivoting there is.  A bottleneck matching by thresholds: the largest theta of the ladder for which a
  //    perfect matching exists on the entries with |a| >= theta * (row maximum) — every pivot then starts at least that large
  //    relative to its row (with the single 1e-3 rung of rounds 1-3, three random RLC / controlled-source networks of 20-70 nodes in
  //    a hundred lost a transient to element growth: scripts/extended_fuzz.py).  Falls back to any structural entry.
  std::vector<char> usable(nnz, 0);
  std::vector<double> rowmax(n, 0.0);
  for (int i = 0; i < n; ++i) for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) rowmax[i] = std::max(rowmax[i], std::fabs(aval[p]));
  std::vector<int> row_of_col;
  bool matched = false;
  for (double theta : {0.9, 0.5, 0.2, 0.05, 0.01, 1e-3}) {
    for (int i = 0; i < n; ++i) for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) usable[p] = (rowmax[i] > 0 && std::fabs(aval[p]) >= theta * rowmax[i]);
    if (max_transversal(n, rowptr, colidx, usable, row_of_col)) { matched = true; break; }
  }
  if (!matched) {
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
  // 2b. the pivot rows from an elimination of these values (the matching is the starting point and wins wherever it is sound)
  { std::vector<int> pr;
    if (numeric_pivot_rows(n, rowptr, colidx, aval, order, row_of_col, pr)) P.prow = pr; }   // numerically singular: keep the matching, the refactorisation will say so
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
  // ---- level-synchronous form ----
  {
    const int nl = (int)P.lvl_ptr.size() - 1, nul = (int)P.ulvl_ptr.size() - 1;
    // Right-looking levels: pivot k waits for every earlier pivot j that touches row k (j in the L pattern of k) OR column k
    // (k in the U pattern of j: pivot j then updates the entries (i, k) below the diagonal).  On a structurally symmetric
    // pattern this is the row level; circuit matrices are only nearly symmetric (controlled sources, branch rows).
    std::vector<int> rlevel(n, 0);
    for (int k = 0; k < n; ++k) {
      for (int j : rowpat[k]) {
        if (j < k) rlevel[k] = std::max(rlevel[k], rlevel[j] + 1);
      }
      for (int c : rowpat[k]) if (c > k) rlevel[c] = std::max(rlevel[c], rlevel[k] + 1);   // rlevel[k] is final here: all its predecessors are < k
    }
    int nrl = 0; for (int v : rlevel) nrl = std::max(nrl, v + 1);
    P.n_rlvl = nrl;
    std::vector<std::vector<std::pair<int, int>>> a_items(nrl);                 // (l_pos, diag_pos)
    std::vector<std::map<int, std::vector<std::array<int, 3>>>> b_items(nrl);    // dst -> sources, in row / entry order
    for (int i = 0; i < n; ++i) {
      for (int e = P.lrow_ptr[i]; e < P.lrow_ptr[i + 1]; ++e) {
        const int k = P.l_k[e], lk = rlevel[k];
        a_items[lk].push_back({P.l_pos[e], P.diag_pos[k]});
        for (int q = P.l_upd_ptr[e]; q < P.l_upd_ptr[e + 1]; ++q) b_items[lk][P.upd_dst[q]].push_back({P.l_pos[e], P.upd_src[q], P.diag_pos[k]});
      }
    }
    P.la_ptr.assign(1, 0); P.la_pos.clear(); P.la_diag.clear();
    P.lb_ptr.assign(1, 0); P.lb_nheavy.clear(); P.lb_dst.clear(); P.lb_sptr.assign(1, 0); P.lb_l.clear(); P.lb_u.clear(); P.lb_d.clear();
    P.max_level_width = 0;
    for (int l = 0; l < nrl; ++l) {
      for (auto& it : a_items[l]) { P.la_pos.push_back(it.first); P.la_diag.push_back(it.second); }
      P.la_ptr.push_back((int)P.la_pos.size());
      int nheavy = 0;
      for (int pass = 0; pass < 2; ++pass)   // light items first, heavy ones after them
        for (auto& kv : b_items[l]) {
          const bool heavy = (int)kv.second.size() > SP_HEAVY;
          if (heavy != (pass == 1)) continue;
          nheavy += heavy ? 1 : 0;
          P.lb_dst.push_back(kv.first);
          for (auto& sx : kv.second) { P.lb_l.push_back(sx[0]); P.lb_u.push_back(sx[1]); P.lb_d.push_back(sx[2]); }
          P.lb_sptr.push_back((int)P.lb_l.size());
        }
      P.lb_ptr.push_back((int)P.lb_dst.size());
      P.lb_nheavy.push_back(nheavy);
    }
    for (int l = 0; l < nl; ++l) P.max_level_width = std::max(P.max_level_width, P.lvl_ptr[l + 1] - P.lvl_ptr[l]);
    P.fl_ptr.assign(1, 0); P.fl_rows.clear(); P.fl_nheavy.clear();
    for (int l = 0; l < nl; ++l) {
      int nheavy = 0;
      for (int pass = 0; pass < 2; ++pass)
        for (int r = P.lvl_ptr[l]; r < P.lvl_ptr[l + 1]; ++r) {
          const int k = P.lvl_rows[r];
          const bool heavy = P.lrow_ptr[k + 1] - P.lrow_ptr[k] > SP_HEAVY;
          if (heavy != (pass == 1)) continue;
          nheavy += heavy ? 1 : 0;
          P.fl_rows.push_back(k);
        }
      P.fl_ptr.push_back((int)P.fl_rows.size());
      P.fl_nheavy.push_back(nheavy);
    }
    P.bl_ptr = P.ulvl_ptr; P.bl_rows = P.ulvl_rows;
    // worth a launch per level only when the levels are few and wide (a tiled array behind shared rails: 13 levels of ~1000
    // rows); a chain (RC ladder: one row per level) stays on the single-workgroup kernel
    P.wide_levels = nrl + nl + nul <= 144 && 3 * n >= 8 * (nrl + nl + nul);
  }
  // ---- subtree form ----
  {
    SubtreePlan& T = P.sub;
    T = SubtreePlan();
    std::vector<char> top(n, 0);
    for (int k = 0; k < n; ++k) {
      bool t = top[k] || (P.lrow_ptr[k + 1] - P.lrow_ptr[k] > SP_HEAVY);
      for (int j : rowpat[k]) if (j < k && top[j]) t = true;
      top[k] = t ? 1 : 0;
      if (t) for (int j : rowpat[k]) if (j > k) top[j] = 1;   // the backward substitution of a top row needs x_j first
    }
    for (int k = 0; k < n; ++k) if (top[k]) T.top_rows.push_back(k);
    T.nT = (int)T.top_rows.size();
    std::vector<int> tix(n, -1);
    for (int t = 0; t < T.nT; ++t) tix[T.top_rows[t]] = t;
    // groups: union-find over the non-top rows
    std::vector<int> par(n);
    std::iota(par.begin(), par.end(), 0);
    auto find = [&](int x) { while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; } return x; };
    for (int k = 0; k < n; ++k) if (!top[k]) for (int j : rowpat[k]) if (j != k && !top[j]) { const int a = find(k), b = find(j); if (a != b) par[std::max(a, b)] = std::min(a, b); }
    std::vector<int> gid(n, -1);
    std::vector<std::vector<int>> rows_of;
    for (int k = 0; k < n; ++k) if (!top[k]) { const int r = find(k); if (gid[r] < 0) { gid[r] = (int)rows_of.size(); rows_of.emplace_back(); } gid[k] = gid[r]; rows_of[gid[k]].push_back(k); }
    T.n_groups = (int)rows_of.size();
    bool ok = T.nT <= 16 && T.n_groups >= 64;
    // the top rows' L entries by group
    std::vector<std::vector<std::pair<int, int>>> seg(ok ? T.n_groups : 0);   // per group: (top index, L-entry index e)
    if (ok) for (int t = 0; t < T.nT; ++t) { const int r = T.top_rows[t]; for (int e = P.lrow_ptr[r]; e < P.lrow_ptr[r + 1]; ++e) { const int k = P.l_k[e]; if (!top[k]) seg[gid[k]].push_back({t, e}); } }
    T.blob_ptr.assign(1, 0);
    for (int g = 0; g < T.n_groups && ok; ++g) {
      const std::vector<int>& R = rows_of[g];
      std::map<int, int> slot;   // global LU position -> local slot
      std::vector<int> a_idx, lu_pos;
      auto add_slot = [&](int gpos) { auto it = slot.find(gpos); if (it != slot.end()) return it->second; const int v = (int)lu_pos.size(); slot[gpos] = v; lu_pos.push_back(gpos); a_idx.push_back(-1); return v; };
      for (int k : R) for (int c : rowpat[k]) add_slot(find_pos(k, c));
      const int n_own = (int)lu_pos.size();
      for (auto& se : seg[g]) add_slot(P.l_pos[se.second]);
      const int n_ent = (int)lu_pos.size();
      const int schur0 = n_ent, y0 = schur0 + T.nT * T.nT, acc0 = y0 + (int)R.size(), nv = acc0 + T.nT;
      for (int v = n_ent; v < nv; ++v) { lu_pos.push_back(-1); a_idx.push_back(-1); }
      std::map<int, int> yslot; for (size_t i = 0; i < R.size(); ++i) yslot[R[i]] = y0 + (int)i;
      // fused updates, bucketed by pivot (the group's own rows in ascending order): for every L entry (i, k) — i an own row or a top
      // row — its matrix updates {dst (i, c), src (k, c)} and the forward step {dst y_i or the top row's accumulator, src y_k}
      std::map<int, int> pix; for (size_t i = 0; i < R.size(); ++i) pix[R[i]] = (int)i;
      std::vector<std::vector<std::array<int, 3>>> fu(R.size());   // per pivot: {dst, src, lp}
      auto push_entry = [&](int e, int t_of_row, int ydst) {
        const int k = P.l_k[e], lp = slot.at(P.l_pos[e]), pi = pix.at(k);
        for (int q = P.l_upd_ptr[e]; q < P.l_upd_ptr[e + 1]; ++q) {
          int dst;
          auto it = slot.find(P.upd_dst[q]);
          if (it != slot.end()) dst = it->second;
          else {   // (top row, top column): accumulate the contribution
            if (t_of_row < 0) { ok = false; return; }
            const auto& rp = rowpat[k]; const int c = rp[P.upd_src[q] - lu_ptr[k]];   // the column: same as the source's in pivot row k
            if (tix[c] < 0) { ok = false; return; }
            dst = schur0 + t_of_row * T.nT + tix[c];
          }
          fu[pi].push_back({dst, slot.at(P.upd_src[q]), lp});
        }
        fu[pi].push_back({ydst, yslot.at(k), lp});
      };
      for (int k : R) for (int e = P.lrow_ptr[k]; e < P.lrow_ptr[k + 1] && ok; ++e) push_entry(e, -1, yslot.at(k));
      for (auto& se : seg[g]) { if (!ok) break; push_entry(se.second, se.first, acc0 + se.first); }
      if (!ok || nv >= 32768) { ok = false; break; }
      // original values of A
      for (int k : R) { const int r = P.prow[k]; for (int pp = rowptr[r]; pp < rowptr[r + 1]; ++pp) a_idx[slot.at(P.a2lu[pp])] = pp; }
      for (int t = 0; t < T.nT; ++t) { const int r = P.prow[T.top_rows[t]]; for (int pp = rowptr[r]; pp < rowptr[r + 1]; ++pp) { auto it = slot.find(P.a2lu[pp]); if (it != slot.end() && it->second >= n_own) a_idx[it->second] = pp; } }
      std::vector<int> piv_dp, fu_ptr(1, 0), fu_ds, fu_lp, rhs_idx, rowk, dx_idx;
      for (size_t i = 0; i < R.size(); ++i) {
        piv_dp.push_back(slot.at(P.diag_pos[R[i]]));
        for (auto& u : fu[i]) { fu_ds.push_back((u[0] << 16) | u[1]); fu_lp.push_back(u[2]); }
        fu_ptr.push_back((int)fu_ds.size());
        rhs_idx.push_back(P.prow[R[i]]); rowk.push_back(R[i]); dx_idx.push_back(P.pcol[R[i]]);
      }
      // backward, column-oriented: first the top unknowns (y_i -= u_iT x_T), then the pivots in descending order
      // (x_k = y_k / u_kk ; y_i -= u_ik x_k for the rows i < k of the group that hold column k)
      std::vector<std::vector<std::pair<int, int>>> bt(T.nT), bc(R.size());   // (u slot, y slot of row i)
      for (int k : R) for (int e = P.urow_ptr[k]; e < P.urow_ptr[k + 1]; ++e) {
        const int j = P.u_col[e];
        if (top[j]) bt[tix[j]].push_back({slot.at(P.u_pos[e]), yslot.at(k)});
        else bc[pix.at(j)].push_back({slot.at(P.u_pos[e]), yslot.at(k)});
      }
      std::vector<int> bt_ptr(1, 0), bt_up, bt_y, bc_ptr(1, 0), bc_up, bc_y;
      for (int t = 0; t < T.nT; ++t) { for (auto& x : bt[t]) { bt_up.push_back(x.first); bt_y.push_back(x.second); } bt_ptr.push_back((int)bt_up.size()); }
      for (int i = (int)R.size() - 1; i >= 0; --i) { for (auto& x : bc[i]) { bc_up.push_back(x.first); bc_y.push_back(x.second); } bc_ptr.push_back((int)bc_up.size()); }
      // blob
      std::vector<int> b(SubtreePlan::H_WORDS, 0);
      b[SubtreePlan::H_NV] = nv; b[SubtreePlan::H_NOWN] = n_own; b[SubtreePlan::H_NROWS] = (int)R.size(); b[SubtreePlan::H_NFU] = (int)fu_ds.size();
      b[SubtreePlan::H_NBT] = (int)bt_up.size(); b[SubtreePlan::H_NBC] = (int)bc_up.size();
      b[SubtreePlan::H_SCHUR] = schur0; b[SubtreePlan::H_Y] = y0; b[SubtreePlan::H_ACC] = acc0;
      auto app = [&](const std::vector<int>& v) { b.insert(b.end(), v.begin(), v.end()); };
      app(a_idx); app(lu_pos); app(piv_dp); app(fu_ptr); app(fu_ds); app(fu_lp); app(rhs_idx); app(rowk);
      app(bt_ptr); app(bt_up); app(bt_y); app(bc_ptr); app(bc_up); app(bc_y); app(dx_idx);
      while (b.size() & 3) b.push_back(0);
      T.max_blob = std::max(T.max_blob, (int)b.size()); T.max_nv = std::max(T.max_nv, nv); T.max_rows = std::max(T.max_rows, (int)R.size());
      T.blob.insert(T.blob.end(), b.begin(), b.end());
      T.blob_ptr.push_back((int)T.blob.size());
    }
    if (ok) {
      T.top_a_idx.assign((size_t)T.nT * T.nT, -1);
      for (int t = 0; t < T.nT; ++t) { const int r = P.prow[T.top_rows[t]]; for (int pp = rowptr[r]; pp < rowptr[r + 1]; ++pp) { const int c = pos_of_col[colidx[pp]]; if (tix[c] >= 0) T.top_a_idx[(size_t)t * T.nT + tix[c]] = pp; } }
      // LDS of a group's wavefront: the blob + the values; a workgroup is one wavefront
      ok = (size_t)T.max_blob * 4 + (size_t)T.max_nv * 8 <= 60 * 1024;
    }
    T.valid = ok;
    if (!ok) { T.blob.clear(); T.blob_ptr.clear(); }
  }
  P.valid = true;
  return CH_OK;
}

// ------------------------------------------------------------------------------------------------

}  // namespace chip
