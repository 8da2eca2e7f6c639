// ch_sparse.hpp — sparse path for Jacobian blocks that do not fit one CU's LDS (large coupled circuits).
//
// north_star: "KLU-ordered … sparse LU re-factor + triangular solve in HIP".  KLU's recipe is:
// analyse once on the host (permute to a zero-free diagonal, fill-reducing ordering, symbolic
// factorisation), then RE-factor numerically with the pivot sequence fixed.  Here:
//   host  (once per circuit and analysis kind): maximum transversal on the numerically significant
//          entries (zero-free diagonal for branch rows), minimum-degree ordering of the symmetrised
//          pattern, row-wise symbolic factorisation with fill, dependency levels, and a flat
//          "op list" per row: for every L entry (l_ik) the list of (destination, source) positions
//          of the updates a_ij -= l_ik * u_kj;
//   GPU   (every Newton iteration): device evaluation → stamps in HBM, CSR gather assembly of
//          A = G + α0·C, C, F, Q (deterministic, no atomics), numeric re-factorisation + forward /
//          backward substitution level by level inside ONE workgroup (one wavefront per row, lanes
//          over the updates of an L entry, __syncthreads between levels), update and norms.
// KLU is Gilbert-Peierls without supernodes; no dense supernode blocks ≥16x16
// arise in the circuits tested, so MFMA is not used.  The reference itself never selects a sparse
// solver; this path is checked against the oracle's dense LU and closed forms.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <set>
#include <vector>

#include "ch_analysis.hpp"
#include "ch_kernels.hpp"
#include "ch_sparse_host.hpp"

namespace chip {

struct SparseDev {
  // structure
  const int* rowptr; const int* colidx; const int* mat_gptr; const int* mat_gsrc; const int* vec_gptr; const int* vec_gsrc;
  const int* prow; const int* pcol; const int* a2lu; const int* diag_pos;
  const int* lvl_ptr; const int* lvl_rows; const int* ulvl_ptr; const int* ulvl_rows;
  const int* lrow_ptr; const int* l_pos; const int* l_k; const int* l_upd_ptr; const int* upd_dst; const int* upd_src;
  const int* urow_ptr; const int* u_pos; const int* u_col;
  // level-synchronous form (multi-workgroup kernels)
  const int* lu2a; const int* la_pos; const int* la_diag; const int* lb_dst; const int* lb_sptr; const int* lb_l; const int* lb_u; const int* lb_d;
  const int* fl_rows; const int* bl_rows; double* Lv;
  const int* heavy_rows; int n_heavy_rows;   // CSR rows with more than 256 entries
  const int* heavy_mat; const int* heavy_vec; int n_heavy_mat, n_heavy_vec;   // CSR entries / rows with more than SP_ASM_HEAVY gather sources
  int n, nnz, nnz_lu, n_lvl, n_ulvl, n_dev;
  int s; long xofs;                // sample handled by this workgroup; its offset s*n inside a slot of the state ring
  long st_stage, st_nnz, st_lu, st_n;  // per-sample strides of the work arrays (the struct is built for sample 0)
  int stride, q_ofs, c_ofs, wide;  // stamp record layout (40/4/16 narrow, 144/8/64 with compiled Verilog-A devices)
  // work arrays
  double* stage; double* Aval; double* Cval; double* LUv; double* F; double* Q; double* rhs; double* y; double* dx;
  double* xcur; double* xpred; double* hq; double* w; double* qn;
  double* red;   // mapped host memory: [8] reductions
  int* flag;     // mapped host memory: [2] {singular, bad}
  int* dflag;    // device memory: [1] singular flag of the last factorisation (read by the update kernel)
};

// Every kernel of this path runs all active samples in one launch: blockIdx.y walks the list `act` of sample indices and
// the workgroup works on that sample's slices of the arrays.
__device__ __forceinline__ SparseDev sp_pick(SparseDev d, const int* act) {
  const int sm = act[blockIdx.y];
  d.s = sm; d.xofs = (long)sm * d.st_n;
  d.stage += sm * d.st_stage; d.Aval += sm * d.st_nnz; d.Cval += sm * d.st_nnz; if (d.LUv) d.LUv += sm * d.st_lu; if (d.Lv) d.Lv += sm * d.st_lu;
  d.F += sm * d.st_n; d.Q += sm * d.st_n; d.rhs += sm * d.st_n; d.y += sm * d.st_n; d.dx += sm * d.st_n;
  d.xcur += sm * d.st_n; d.xpred += sm * d.st_n; d.hq += sm * d.st_n; d.w += sm * d.st_n; d.qn += sm * d.st_n;
  d.red += (long)sm * 8; d.flag += (long)sm * 2; d.dflag += sm;
  return d;
}

// predictor / history term / Newton weights
__global__ void sp_predict_kernel(const NewtonArgs a, const SparseDev d0, const int* act) {
  const SparseDev d = sp_pick(d0, act);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n) return;
  const double x0 = a.X[(long)a.hist_slot[0] * a.slot_stride + d.xofs + i];
  double p = x0, h = 0.0;
  if (a.mode == MODE_TRAN) {
    p = 0.0;
    for (int j = 0; j < a.npred; ++j) p += a.wpred[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + d.xofs + i];
    for (int j = 1; j <= a.k; ++j) h += a.alpha[j] * a.Qh[(long)a.hist_slot[j - 1] * a.slot_stride + d.xofs + i];
  }
  d.xpred[i] = p; d.xcur[i] = p; d.hq[i] = h; d.qn[i] = 0.0;
  d.w[i] = 1.0 / (a.reltol * fabs(x0) + a.abstol);
}

// BSIM4 by function: PART 0 stores the current half of the record (I, dI/dV), PART 1 the charge half (Q, dQ/dV); the compiler drops
// what a half does not store (3 925 and 3 395 instructions against 6 289 — the split of the device-resident stepper's wave pairs)
template <int PART>
__device__ __noinline__ void sp_mos_half(const B4Col P, double v0, double v1, double v2, double v3, double gmin, double m, double* st) {
  double o[40];
  b4_device(P, v0, v1, v2, v3, gmin, o);
  if (PART != 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st[j] = m * o[j];
#pragma unroll
    for (int j = 8; j < 24; ++j) st[j] = m * o[j];
  }
  if (PART != 0) {
#pragma unroll
    for (int j = 4; j < 8; ++j) st[j] = m * o[j];
#pragma unroll
    for (int j = 24; j < 40; ++j) st[j] = m * o[j];
  }
}
// one thread per device instance: stamps to HBM.  Narrow records (no compiled Verilog-A device): two workgroups per 64 devices, the even
// one evaluates the current halves of the MOSFETs and every other device, the odd one the charge halves (grid.x = 2 * ceil(n_dev / 64)).
__global__ __launch_bounds__(64) void sp_eval_kernel(const NewtonArgs a, const SparseDev d0, const int* act) {
  const SparseDev d = sp_pick(d0, act);
  const int part = d.wide ? -1 : (int)(blockIdx.x & 1);
  const int dev = (d.wide ? blockIdx.x : blockIdx.x >> 1) * blockDim.x + threadIdx.x;
  if (dev >= d.n_dev) return;
  const int sm = d.s;
  const double* kvl = a.inline_vals ? nullptr : a.kv + (long)(a.Ssrc > 1 ? sm : 0) * a.nk;
  const double* svg = a.srcv + (long)(a.Ssrc > 1 ? sm : 0) * a.nsrc;
  const int kind = a.dkind[dev];
  const int* tm = a.dterm + NTERM * dev;
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = tm[k];
    if (t >= 0) v[k] = d.xcur[t];
    else if (a.inline_vals) { double kvv = 0.0; for (int q = 0; q < KV_INLINE; ++q) if (q == -t - 1) kvv = a.vals_inline[q]; v[k] = kvv; }
    else v[k] = kvl[-t - 1];
  }
  const int hd = a.dhdev[dev];
  const long pi = (long)hd * a.Spar + (a.Spar > 1 ? sm : 0);
  const double m = a.dmult[pi];
  const double gmin = a.gmin_s[a.Sgmin > 1 ? sm : 0];
  double* st_final = d.stage + (size_t)dev * d.stride;
  auto srcval = [&](int si) { if (a.inline_vals) { double r = 0.0; for (int q = 0; q < KV_INLINE; ++q) if (q == a.nk + si) r = a.vals_inline[q]; return r; } return svg[si]; };
  if (kind == K_VA) {
    double vv[NTERM];
    for (int k = 0; k < NTERM; ++k) {
      const int t = tm[k];
      if (t >= 0) vv[k] = d.xcur[t];
      else if (a.inline_vals) { double kvv = 0.0; for (int q = 0; q < KV_INLINE; ++q) if (q == -t - 1) kvv = a.vals_inline[q]; vv[k] = kvv; }
      else vv[k] = kvl[-t - 1];
    }
    const va::Env env{a.temp_s[a.Stemp > 1 ? sm : 0] + 273.15, gmin};
    va_gen::stamp_c(a.dcls_local[dev], a.vapar + (long)sm * a.va_stride + a.dsrc[dev], a.vacache + (long)sm * a.vac_stride + a.dvac[dev], vv, env, m, st_final);
    return;
  }
  double tmp40[40];
  double* st = d.wide ? tmp40 : st_final;
  if (kind == K_MOS) {
    const B4Col P = b4_col(a.mosp, (long)a.dcls[dev] * a.Smos + (a.Smos > 1 ? sm : 0));
    if (part == 0) sp_mos_half<0>(P, v[0], v[1], v[2], v[3], gmin, m, st);
    else if (part == 1) sp_mos_half<1>(P, v[0], v[1], v[2], v[3], gmin, m, st);
    else { sp_mos_half<-1>(P, v[0], v[1], v[2], v[3], gmin, m, st); widen_stamp(st, st_final); }
    return;
  }
  if (part == 1) return;   // every other device belongs to the even workgroup
  for (int j = 0; j < 40; ++j) st[j] = 0.0;
  switch (kind) {
    case K_R: { const double g = m / a.dpar[pi], i = g * (v[0] - v[1]); st[0] = i; st[1] = -i; st[8] = g; st[9] = -g; st[12] = -g; st[13] = g; } break;
    case K_C: { const double c = m * a.dpar[pi], q = c * (v[0] - v[1]); st[4] = q; st[5] = -q; st[24] = c; st[25] = -c; st[28] = -c; st[29] = c; } break;
    case K_I: { const double i = m * srcval(a.dsrc[dev]); st[0] = i; st[1] = -i; } break;
    case K_V: case K_L: case K_VCVS_A: {
      const double ib = v[2], src = kind == K_V ? srcval(a.dsrc[dev]) : 0.0, l = kind == K_L ? a.dpar[pi] : 0.0;
      st[0] = m * ib; st[1] = -m * ib; st[2] = v[0] - v[1] - src; st[6] = -l * ib;
      st[8 + 2] = m; st[8 + 6] = -m; st[8 + 8] = 1.0; st[8 + 9] = -1.0; st[24 + 10] = -l;
    } break;
    case K_VCVS_B: { const double g = a.dpar[pi]; st[0] = -g * (v[1] - v[2]); st[8 + 1] = -g; st[8 + 2] = g; } break;
    case K_VCCS: { const double g = m * a.dpar[pi], i = g * (v[2] - v[3]); st[0] = i; st[1] = -i; st[8 + 2] = g; st[8 + 3] = -g; st[8 + 6] = -g; st[8 + 7] = g; } break;
  }
  if (d.wide) widen_stamp(st, st_final);
}

// CSR gather assembly: one thread per nnz and per row; an entry or row with more than 64 sources (a supply rail shared by
// every tile collects one stamp per attached device) is summed by the whole wavefront, lanes striding over the sources and a
// fixed-order tree at the end — deterministic either way.
__device__ __forceinline__ double sp_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
constexpr int SP_ASM_HEAVY = 64;
__global__ __launch_bounds__(256) void sp_assemble_kernel(const NewtonArgs a, const SparseDev d0, const int* act) {
  const SparseDev d = sp_pick(d0, act);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double alpha0 = a.mode == MODE_DC ? 0.0 : a.alpha[0];
  if (i < d.nnz) {
    const int p0 = d.mat_gptr[i], p1 = d.mat_gptr[i + 1];
    if (p1 - p0 <= SP_ASM_HEAVY) {
      double g = 0.0, c = 0.0;
      for (int p = p0; p < p1; ++p) { const int o = d.mat_gsrc[p]; g += d.stage[o]; c += d.stage[o + d.c_ofs]; }
      d.Aval[i] = g + alpha0 * c; d.Cval[i] = c;
    }
  }
  if (i < d.n) {
    const int p0 = d.vec_gptr[i], p1 = d.vec_gptr[i + 1];
    if (p1 - p0 <= SP_ASM_HEAVY) {
      double f = 0.0, q = 0.0;
      for (int p = p0; p < p1; ++p) { const int o = d.vec_gsrc[p]; f += d.stage[o]; q += d.stage[o + d.q_ofs]; }
      if (a.gshunt != 0.0 && !(a.dmask[i] & 2)) f += a.gshunt * d.xcur[i];
      d.Q[i] = q;
      const double F = f + alpha0 * q + d.hq[i];
      d.F[i] = F; d.rhs[i] = -F;
    }
  }
}
// fixed-order block reduction of two values (256 threads): wave trees, then the four wave results in order
__device__ __forceinline__ void sp_block_sum2(double& x, double& y, double* sh) {
  x = sp_wave_sum(x); y = sp_wave_sum(y);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sh[wv] = x; sh[4 + wv] = y; }
  __syncthreads();
  x = ((sh[0] + sh[1]) + sh[2]) + sh[3]; y = ((sh[4] + sh[5]) + sh[6]) + sh[7];
}
// the entries and rows with long gather lists (a supply rail shared by every tile collects one stamp per attached device): one
// 256-thread workgroup per item, four sources in flight per thread, fixed-order reduction — deterministic like the light path
// One workgroup cannot pull a rail's 2 - 4 MB of stamps through one CU's memory pipe in less than ~50 us (measured), so an item is
// spread over SP_HB workgroups: each sums a contiguous slice (four sources in flight per thread, fixed order) into
// hpart[sample][item][workgroup][2]; sp_assemble_heavy_finish_kernel adds the slices in order and writes the entry / row.
constexpr int SP_HB = 64;
__global__ __launch_bounds__(256) void sp_assemble_heavy_kernel(const NewtonArgs a, const SparseDev d0, const int* act, double* hpart) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double sh[8];
  const int item = blockIdx.x / SP_HB, slice = blockIdx.x % SP_HB, t = threadIdx.x;
  const bool vec = item >= d.n_heavy_mat;
  const int e = vec ? d.heavy_vec[item - d.n_heavy_mat] : d.heavy_mat[item];
  const int* gp = vec ? d.vec_gptr : d.mat_gptr; const int* gs = vec ? d.vec_gsrc : d.mat_gsrc;
  const int off2 = vec ? d.q_ofs : d.c_ofs;
  const int q0 = gp[e], q1 = gp[e + 1], per = (q1 - q0 + SP_HB - 1) / SP_HB;
  const int p0 = q0 + slice * per, p1 = min(q1, p0 + per);
  double s1 = 0.0, s2 = 0.0;
  for (int p = p0 + t; p < p1; p += 4 * 256) {
    const int l = p1 - 1;
    const int o0 = gs[p], o1 = gs[min(p + 256, l)], o2 = gs[min(p + 512, l)], o3 = gs[min(p + 768, l)];
    const double a0 = d.stage[o0], b0 = d.stage[o0 + off2], a1 = d.stage[o1], b1 = d.stage[o1 + off2], a2 = d.stage[o2], b2 = d.stage[o2 + off2], a3 = d.stage[o3], b3 = d.stage[o3 + off2];
    s1 += a0; s2 += b0;
    if (p + 256 < p1) { s1 += a1; s2 += b1; }
    if (p + 512 < p1) { s1 += a2; s2 += b2; }
    if (p + 768 < p1) { s1 += a3; s2 += b3; }
  }
  sp_block_sum2(s1, s2, sh);
  if (t == 0) { double* hp = hpart + (((size_t)d.s * (d.n_heavy_mat + d.n_heavy_vec) + item) * SP_HB + slice) * 2; hp[0] = s1; hp[1] = s2; }
}
__global__ __launch_bounds__(64) void sp_assemble_heavy_finish_kernel(const NewtonArgs a, const SparseDev d0, const int* act, const double* hpart) {
  const SparseDev d = sp_pick(d0, act);
  const int item = blockIdx.x, lane = threadIdx.x;
  const double alpha0 = a.mode == MODE_DC ? 0.0 : a.alpha[0];
  const bool vec = item >= d.n_heavy_mat;
  const int e = vec ? d.heavy_vec[item - d.n_heavy_mat] : d.heavy_mat[item];
  const double* hp = hpart + (((size_t)d.s * (d.n_heavy_mat + d.n_heavy_vec) + item) * SP_HB) * 2;
  double s1 = lane < SP_HB ? hp[lane * 2] : 0.0, s2 = lane < SP_HB ? hp[lane * 2 + 1] : 0.0;
  s1 = sp_wave_sum(s1); s2 = sp_wave_sum(s2);
  if (lane == 0) {
    if (vec) {
      if (a.gshunt != 0.0 && !(a.dmask[e] & 2)) s1 += a.gshunt * d.xcur[e];
      d.Q[e] = s2;
      const double F = s1 + alpha0 * s2 + d.hq[e];
      d.F[e] = F; d.rhs[e] = -F;
    } else { d.Aval[e] = s1 + alpha0 * s2; d.Cval[e] = s2; }
  }
}
__global__ void sp_diag_shunt_kernel(const NewtonArgs a, const SparseDev d0, const int* act) {  // gmin stepping: + gshunt on node diagonals
  const SparseDev d = sp_pick(d0, act);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.n || (a.dmask[i] & 2)) return;
  for (int p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) if (d.colidx[p] == i) d.Aval[p] += a.gshunt;
}

// Numeric re-factorisation + both triangular solves inside ONE workgroup.
__global__ __launch_bounds__(1024) void sp_lu_solve_kernel(const SparseDev d0, const int* act) {
  const SparseDev d = sp_pick(d0, act);
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  for (int i = tid; i < d.nnz_lu; i += nthr) d.LUv[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < d.nnz; i += nthr) d.LUv[d.a2lu[i]] = d.Aval[i];
  __shared__ int s_sing;
  if (tid == 0) s_sing = 0;
  __syncthreads();
  // factorisation: rows of a level are independent; one wavefront per row, lanes over the updates of one L entry
  for (int lv = 0; lv < d.n_lvl; ++lv) {
    for (int r = d.lvl_ptr[lv] + wave; r < d.lvl_ptr[lv + 1]; r += nwave) {
      const int k = d.lvl_rows[r];
      for (int e = d.lrow_ptr[k]; e < d.lrow_ptr[k + 1]; ++e) {
        const double ukk = d.LUv[d.diag_pos[d.l_k[e]]];
        const double l = d.LUv[d.l_pos[e]] / ukk;
        for (int p = d.l_upd_ptr[e] + lane; p < d.l_upd_ptr[e + 1]; p += 64) d.LUv[d.upd_dst[p]] -= l * d.LUv[d.upd_src[p]];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane == 0) d.LUv[d.l_pos[e]] = l;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      const double ukk = d.LUv[d.diag_pos[k]];
      if (lane == 0 && (!(fabs(ukk) > 0.0) || !(fabs(ukk) < 1e300))) s_sing = 1;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (tid == 0) { d.flag[0] = s_sing; d.dflag[0] = s_sing; }
  if (s_sing) return;
  // forward: y_k = b(prow[k]) - sum_j l_kj y_j   (rows of a level independent)
  for (int lv = 0; lv < d.n_lvl; ++lv) {
    for (int r = d.lvl_ptr[lv] + tid; r < d.lvl_ptr[lv + 1]; r += nthr) {
      const int k = d.lvl_rows[r];
      double s = d.rhs[d.prow[k]];
      for (int e = d.lrow_ptr[k]; e < d.lrow_ptr[k + 1]; ++e) s -= d.LUv[d.l_pos[e]] * d.y[d.l_k[e]];
      d.y[k] = s;
    }
    __threadfence_block();
    __syncthreads();
  }
  // backward: z_k = (y_k - sum_{j>k} u_kj z_j) / u_kk ; dx(pcol[k]) = z_k
  for (int lv = 0; lv < d.n_ulvl; ++lv) {
    for (int r = d.ulvl_ptr[lv] + tid; r < d.ulvl_ptr[lv + 1]; r += nthr) {
      const int k = d.ulvl_rows[r];
      double s = d.y[k];
      for (int e = d.urow_ptr[k]; e < d.urow_ptr[k + 1]; ++e) s -= d.LUv[d.u_pos[e]] * d.dx[d.pcol[d.u_col[e]]];
      d.dx[d.pcol[k]] = s / d.LUv[d.diag_pos[k]];
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ---- level-synchronous, multi-workgroup refactorisation and solves ------------------------------------------------------------
// For Jacobians whose elimination levels are few and wide (a tiled array behind shared, non-ideal supply rails: ~13 levels of
// ~1000 independent rows, plus two rail rows that collect one product per tile): one launch per level, every CU busy.
//   factor, level l (right-looking):  A-items  l_ik = a_ik / u_kk  for the L entries whose pivot k is in level l;
//                                     B-items  a_dst -= sum (a_ik / u_kk) * u_kc  over the level's pivots, per destination,
//                                              in a fixed order (deterministic); long lists are reduced by a whole wavefront.
//   forward / backward substitution by the same levels; a row with a long L part (a rail) is reduced by a wavefront.
// Nothing is updated in place that the same launch reads (see SparsePlan), so a level needs no synchronisation inside it.
__global__ void sp2_scatter_kernel(const SparseDev d0, const int* act) {
  const SparseDev d = sp_pick(d0, act);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { d.flag[0] = 0; d.dflag[0] = 0; }
  if (i >= d.nnz_lu) return;
  const int a = d.lu2a[i];
  d.LUv[i] = a >= 0 ? d.Aval[a] : 0.0;
}
__device__ __forceinline__ double sp2_wave_sum(double v) {   // fixed-order tree: deterministic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// grid.x: [thread items: A-items then light B-items, 256 per block] then [heavy B-items, 4 wavefronts per block]
__global__ __launch_bounds__(256) void sp2_factor_level_kernel(const SparseDev d0, const int* act, int a0, int nA, int b0, int nBl, int nBh, int thread_blocks) {
  const SparseDev d = sp_pick(d0, act);
  if ((int)blockIdx.x < thread_blocks) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < nA) {
      const int p = d.la_pos[a0 + t];
      d.Lv[p] = d.LUv[p] / d.LUv[d.la_diag[a0 + t]];
    } else if (t < nA + nBl) {
      const int it = b0 + (t - nA);
      double s = 0.0;
      for (int q = d.lb_sptr[it]; q < d.lb_sptr[it + 1]; ++q) s += (d.LUv[d.lb_l[q]] / d.LUv[d.lb_d[q]]) * d.LUv[d.lb_u[q]];
      d.LUv[d.lb_dst[it]] -= s;
    }
  } else {
    const int w = ((int)blockIdx.x - thread_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nBh) return;
    const int it = b0 + nBl + w;
    double s = 0.0;
    for (int q = d.lb_sptr[it] + lane; q < d.lb_sptr[it + 1]; q += 64) s += (d.LUv[d.lb_l[q]] / d.LUv[d.lb_d[q]]) * d.LUv[d.lb_u[q]];
    s = sp2_wave_sum(s);
    if (lane == 0) d.LUv[d.lb_dst[it]] -= s;
  }
}
// forward substitution, one level: y_k = b(prow[k]) - sum_j l_kj y_j; also the pivot check of the level's rows
__global__ __launch_bounds__(256) void sp2_fwd_level_kernel(const SparseDev d0, const int* act, int r0, int nl, int nh, int thread_blocks) {
  const SparseDev d = sp_pick(d0, act);
  if ((int)blockIdx.x < thread_blocks) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nl) return;
    const int k = d.fl_rows[r0 + t];
    double s = d.rhs[d.prow[k]];
    for (int e = d.lrow_ptr[k]; e < d.lrow_ptr[k + 1]; ++e) s -= d.Lv[d.l_pos[e]] * d.y[d.l_k[e]];
    d.y[k] = s;
    const double ukk = d.LUv[d.diag_pos[k]];
    if (!(fabs(ukk) > 0.0) || !(fabs(ukk) < 1e300)) { d.flag[0] = 1; d.dflag[0] = 1; }
  } else {   // a row with a long L part: one workgroup, four entries in flight per thread
    __shared__ double sh[8];
    const int w = (int)blockIdx.x - thread_blocks, t = threadIdx.x;
    if (w >= nh) return;
    const int k = d.fl_rows[r0 + nl + w];
    const int e0 = d.lrow_ptr[k], e1 = d.lrow_ptr[k + 1];
    double s = 0.0, dummy = 0.0;
    for (int e = e0 + t; e < e1; e += 256) s += d.Lv[d.l_pos[e]] * d.y[d.l_k[e]];
    sp_block_sum2(s, dummy, sh);
    if (t == 0) {
      d.y[k] = d.rhs[d.prow[k]] - s;
      const double ukk = d.LUv[d.diag_pos[k]];
      if (!(fabs(ukk) > 0.0) || !(fabs(ukk) < 1e300)) { d.flag[0] = 1; d.dflag[0] = 1; }
    }
  }
}
// backward substitution, one level: z_k = (y_k - sum_{j>k} u_kj z_j) / u_kk ; dx(pcol[k]) = z_k
__global__ __launch_bounds__(256) void sp2_bwd_level_kernel(const SparseDev d0, const int* act, int r0, int nr) {
  const SparseDev d = sp_pick(d0, act);
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nr) return;
  const int k = d.bl_rows[r0 + t];
  double s = d.y[k];
  for (int e = d.urow_ptr[k]; e < d.urow_ptr[k + 1]; ++e) s -= d.LUv[d.u_pos[e]] * d.dx[d.pcol[d.u_col[e]]];
  d.dx[d.pcol[k]] = s / d.LUv[d.diag_pos[k]];
}

// ---- subtree form (SubtreePlan, ch_sparse_host.hpp): independent subtrees in LDS, one wavefront each; a dense top block ---------------
struct Sp3Dev {
  const int* blob; const int* blob_ptr;     // per group
  const int* top_a_idx; const int* top_rows;   // top_rows: [nT] pivot index | [nT] rhs index prow[.] | [nT] dx index pcol[.]
  double* schur;                            // [S][nT*nT + nT][n_groups]: entry-major, so that the top kernel sums a contiguous run per entry
  double* xT;                               // [S][nT]
  double* top_base;                         // [S][nT*nT + nT]: A's own top entries and right-hand sides, gathered by sp3_reset_kernel
  double* top_sum;                          // [S][nT*nT + nT]: base + the groups' contributions, one entry per wavefront of sp3_top_kernel
  unsigned* top_cnt;                        // [S]: workgroups of sp3_top_kernel that have delivered their entries (zeroed by sp3_reset_kernel)
  int n_groups, nT, max_nv;
};
// the singular flags of this factorisation; and the top block's own entries A_TT and right-hand sides gathered into one contiguous
// run (two dependent loads each — index, then value — that the top kernel would otherwise wait for in its own chain)
__global__ void sp3_reset_kernel(const SparseDev d0, const int* act, const Sp3Dev q) {
  const SparseDev d = sp_pick(d0, act);
  if (threadIdx.x == 0) { d.flag[0] = 0; d.dflag[0] = 0; q.top_cnt[d.s] = 0u; }
  const int nT = q.nT, ne = nT * nT + nT;
  for (int j = threadIdx.x; j < ne; j += blockDim.x) {
    double b;
    if (j < nT * nT) { const int ai = q.top_a_idx[j]; b = ai >= 0 ? d.Aval[ai] : 0.0; }
    else b = d.rhs[q.top_rows[nT + (j - nT * nT)]];
    q.top_base[(size_t)d.s * ne + j] = b;
  }
}
// one wavefront per group: load; ONE step per pivot — every lane takes fused updates a_ic -= (a_ik / u_kk) u_kc and
// y_i -= (a_ik / u_kk) y_k (nothing a step reads is written by it: one barrier per pivot); Schur contributions.
// grid (n_groups, samples), 64 threads, dynamic LDS = max_nv doubles + the largest blob.
__global__ __launch_bounds__(64) void sp3_group_kernel(const SparseDev d0, const int* act, const Sp3Dev q) {
  const SparseDev d = sp_pick(d0, act);
  extern __shared__ double sp3_sh[];
  double* val = sp3_sh;
  int* B = (int*)(sp3_sh + q.max_nv);
  const int g = blockIdx.x, lane = threadIdx.x;
  {
    const int4* src = (const int4*)(q.blob + q.blob_ptr[g]); int4* dst = (int4*)B;   // blobs are padded to 16 bytes
    const int n4 = (q.blob_ptr[g + 1] - q.blob_ptr[g]) >> 2;
    for (int i = lane; i < n4; i += 64) dst[i] = src[i];
  }
  __syncthreads();
  const Sp3Blob b(B, q.nT);
  for (int v0 = lane; v0 < b.nv; v0 += 64 * 4) {   // four loads in flight per lane
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int v = v0 + 64 * u; const int ai = v < b.nv ? b.a_idx[v] : -1; x[u] = ai >= 0 ? d.Aval[ai] : 0.0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int v = v0 + 64 * u; if (v < b.nv) val[v] = x[u]; }
  }
  __syncthreads();   // (a y slot is a value slot: zeroed above, then set)
  for (int i = lane; i < b.np; i += 64) val[b.y0 + i] = d.rhs[b.rhs_idx[i]];   // own rows: y starts as the right-hand side
  __syncthreads();
  bool sing = false;
  for (int pi = 0; pi < b.np; ++pi) {
    const double pv = val[b.piv_dp[pi]];
    if (!(fabs(pv) > 0.0) || !(fabs(pv) < 1e300)) sing = true;
    for (int u = b.fu_ptr[pi] + lane; u < b.fu_ptr[pi + 1]; u += 64) {
      const int w = b.fu_ds[u];
      const double l = val[b.fu_lp[u]] / pv;
      val[w >> 16] -= l * val[w & 0xffff];
    }
    __syncthreads();
  }
  for (int v = lane; v < b.n_own; v += 64) d.LUv[b.lu_pos[v]] = val[v];
  for (int i = lane; i < b.np; i += 64) d.y[b.rowk[i]] = val[b.y0 + i];
  const int ne = q.nT * q.nT + q.nT;
  double* so = q.schur + (size_t)d.s * ne * q.n_groups;
  for (int j = lane; j < ne; j += 64) so[(size_t)j * q.n_groups + g] = j < q.nT * q.nT ? val[b.schur0 + j] : val[b.acc0 + (j - q.nT * q.nT)];
  if (sing && lane == 0) { d.flag[0] = 1; d.dflag[0] = 1; }
}
// the top block: S = A_TT + sum over groups (fixed order), right-hand side likewise, dense LU with partial pivoting, x_T.
// grid (SP3_TOP_WG, samples), 256 threads.  The sums are spread over the grid — ONE wavefront per entry, 16 loads in flight per lane,
// fixed order (lane, wavefront tree): for the coupled 1024-DFF array nT = 6, i.e. 42 entries x 1024 groups = 344 KB, which a single
// workgroup pulled through one CU in six dependent batches (29-34 us).  The workgroup that delivers last (an agent-scope counter;
// which one it is does not enter the result: every entry is summed by exactly one wavefront) factors and solves the top system:
// one wavefront, an element of the trailing block per lane and elimination step.
constexpr int SP3_TOP_WG = 16;
// lanes of ONE wavefront hand values to each other through LDS: s_waitcnt lgkmcnt(0), and the compiler keeps the accesses on their sides
#define SP_WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
__device__ __forceinline__ double sp_ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sp_st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ __launch_bounds__(256) void sp3_top_kernel(const SparseDev d0, const int* act, const Sp3Dev q) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double S[16 * 17];
  __shared__ int s_flag[2];   // [0]: this workgroup delivered last; [1]: singular
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nT = q.nT, ne = nT * nT + nT, ld = nT + 1;
  const double* so = q.schur + (size_t)d.s * ne * q.n_groups;
  const double* tb = q.top_base + (size_t)d.s * ne;
  double* ts = q.top_sum + (size_t)d.s * ne;
  // requested before anything is waited for (a memory round trip is ~2 us on a cold L2): flag, the index of this lane's result
  const int sing0 = d.dflag[0];
  const int dxi = t < nT ? q.top_rows[2 * nT + t] : 0;
  for (int j = blockIdx.x * 4 + wave; j < ne; j += SP3_TOP_WG * 4) {
    const double bj = tb[j];
    double a = 0.0;
    for (int g0 = lane; g0 < q.n_groups; g0 += 64 * 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int g = g0 + 64 * u; v[u] = g < q.n_groups ? so[(size_t)j * q.n_groups + g] : 0.0; }
#pragma unroll
      for (int u = 0; u < 16; u += 4) a += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
    }
    a = sp2_wave_sum(a);
    if (lane == 0) sp_st_agent(ts + j, bj + a);
  }
  __threadfence();
  __syncthreads();
  if (t == 0) {
    const unsigned before = __hip_atomic_fetch_add(q.top_cnt + d.s, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_flag[0] = before == (unsigned)(gridDim.x - 1);
    s_flag[1] = sing0;
  }
  __syncthreads();
  if (!s_flag[0]) return;
  __threadfence();
  for (int j = t; j < ne; j += 256) {
    const double v = sp_ld_agent(ts + j);
    if (j < nT * nT) S[(j / nT) * ld + (j % nT)] = v; else S[(j - nT * nT) * ld + nT] = v;
  }
  __syncthreads();
  if (wave == 0 && nT > 0 && !sing0) {
    bool sing = false;
    for (int k = 0; k < nT; ++k) {
      // pivot: the largest |S[i][k]|, i >= k (the lowest row on ties)
      double best = (lane >= k && lane < nT) ? fabs(S[lane * ld + k]) : -1.0;
      int bi = lane;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      best = __shfl(best, 0); bi = __shfl(bi, 0);
      if (!(best > 0.0) || !(best < 1e300)) { sing = true; break; }
      if (bi != k && lane <= nT) { const double tmp = S[k * ld + lane]; S[k * ld + lane] = S[bi * ld + lane]; S[bi * ld + lane] = tmp; }
      SP_WAVE_LDS_SYNC();   // one wavefront, LDS in order — the swapped rows are what the next reads see
      const int rows = nT - 1 - k, cols = nT - k;   // trailing rows k+1..nT-1, columns k+1..nT (the right-hand side is column nT)
      const double pk = S[k * ld + k];
      double upd[5]; int pos[5];                      // (nT-1) * nT <= 240 elements: at most four per lane (+1 spare)
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const int idx = lane + 64 * r;
        pos[r] = -1; upd[r] = 0.0;
        if (idx < rows * cols) {
          const int i = k + 1 + idx / cols, c = k + 1 + idx % cols;
          pos[r] = i * ld + c;
          upd[r] = S[pos[r]] - (S[i * ld + k] / pk) * S[k * ld + c];
        }
      }
      SP_WAVE_LDS_SYNC();
#pragma unroll
      for (int r = 0; r < 5; ++r) if (pos[r] >= 0) S[pos[r]] = upd[r];
      SP_WAVE_LDS_SYNC();
    }
    if (sing) { if (lane == 0) { d.flag[0] = 1; d.dflag[0] = 1; s_flag[1] = 1; } }
    else for (int k = nT - 1; k >= 0; --k) {          // column-oriented: x_k, then every row above takes its share
      const double xk = S[k * ld + nT] / S[k * ld + k];
      SP_WAVE_LDS_SYNC();
      if (lane == 0) S[k * ld + nT] = xk;
      if (lane < k) S[lane * ld + nT] -= S[lane * ld + k] * xk;
      SP_WAVE_LDS_SYNC();
    }
  }
  __syncthreads();
  if (t < nT && !s_flag[1]) { const double x = S[t * ld + nT]; q.xT[(size_t)d.s * nT + t] = x; d.dx[dxi] = x; }
}
// backward substitution of the groups with x_T, column-oriented: first y_i -= u_iT x_T for every top unknown, then the pivots in
// descending order: x_k = y_k / u_kk, y_i -= u_ik x_k for the rows of the group that hold column k.  grid (n_groups, samples), 64 threads.
__global__ __launch_bounds__(64) void sp3_back_kernel(const SparseDev d0, const int* act, const Sp3Dev q) {
  const SparseDev d = sp_pick(d0, act);
  extern __shared__ double sp3_sh[];
  double* val = sp3_sh;
  int* B = (int*)(sp3_sh + q.max_nv);
  const int g = blockIdx.x, lane = threadIdx.x;
  if (d.dflag[0]) return;
  {
    const int4* src = (const int4*)(q.blob + q.blob_ptr[g]); int4* dst = (int4*)B;
    const int n4 = (q.blob_ptr[g + 1] - q.blob_ptr[g]) >> 2;
    for (int i = lane; i < n4; i += 64) dst[i] = src[i];
  }
  __syncthreads();
  const Sp3Blob b(B, q.nT);
  for (int v0 = lane; v0 < b.n_own; v0 += 64 * 4) {   // four loads in flight per lane
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int v = v0 + 64 * u; x[u] = v < b.n_own ? d.LUv[b.lu_pos[v]] : 0.0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int v = v0 + 64 * u; if (v < b.n_own) val[v] = x[u]; }
  }
  for (int i = lane; i < b.np; i += 64) val[b.y0 + i] = d.y[b.rowk[i]];            // y, overwritten by x pivot by pivot
  for (int t = lane; t < q.nT; t += 64) val[b.acc0 + t] = q.xT[(size_t)d.s * q.nT + t];
  __syncthreads();
  for (int t = 0; t < q.nT; ++t) {   // (a row holds a top column at most once: the targets of one t are distinct)
    const double xt = val[b.acc0 + t];
    for (int e = b.bt_ptr[t] + lane; e < b.bt_ptr[t + 1]; e += 64) val[b.bt_y[e]] -= val[b.bt_up[e]] * xt;
    __syncthreads();
  }
  for (int r = 0; r < b.np; ++r) {
    const int pi = b.np - 1 - r;
    const double x = val[b.y0 + pi] / val[b.piv_dp[pi]];
    for (int e = b.bc_ptr[r] + lane; e < b.bc_ptr[r + 1]; e += 64) val[b.bc_y[e]] -= val[b.bc_up[e]] * x;   // rows above the pivot: never y_pi itself
    __syncthreads();
    if (lane == 0) { val[b.y0 + pi] = x; d.dx[b.dx_idx[pi]] = x; }
  }
}

// reductions: red[0]=max|F|, red[1]=max|dx| over node rows, red[2]=sum (dx*w)^2, red[3]=bad flag
__global__ __launch_bounds__(1024) void sp_norms_kernel(const NewtonArgs a, const SparseDev d0, const int* act, int what) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double s0[1024], s1[1024], s2[1024];
  const int t = threadIdx.x;
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = t; i < d.n; i += 1024) {
    if (what == 0) m0 = fmax(m0, fabs(d.F[i]));
    else { if (!(a.dmask[i] & 2)) m1 = fmax(m1, fabs(d.dx[i])); }
  }
  s0[t] = m0; s1[t] = m1; s2[t] = m2;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (t < o) { s0[t] = fmax(s0[t], s0[t + o]); s1[t] = fmax(s1[t], s1[t + o]); } __syncthreads(); }
  if (t == 0) { if (what == 0) d.red[0] = s0[0]; else d.red[1] = s1[0]; }
}

// x += scale*dx ; qn = Q + C*(scale*dx) ; e2 = sum (scale*dx*w)^2 ; bad flag
__global__ __launch_bounds__(1024) void sp_update_kernel(const NewtonArgs a, const SparseDev d0, const int* act, const double* scale_v) {
  const SparseDev d = sp_pick(d0, act);
  const double scale = scale_v ? scale_v[d.s] : 1.0;
  __shared__ double s2[1024]; __shared__ int sbad;
  const int t = threadIdx.x;
  if (d.dflag[0]) return;  // factorisation failed: leave the iterate untouched (the host re-analyses and retries)
  if (t == 0) sbad = 0;
  __syncthreads();
  double e2 = 0;
  for (int i = t; i < d.n; i += 1024) {
    const double dxi = scale * d.dx[i];
    const double xn = d.xcur[i] + dxi;
    if (!(xn == xn) || fabs(xn) > 1e300) sbad = 1;
    const double tt = dxi * d.w[i]; e2 += tt * tt;
    if (a.mode == MODE_TRAN && d.rowptr[i + 1] - d.rowptr[i] <= 256) { double q = d.Q[i]; for (int p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) q += d.Cval[p] * scale * d.dx[d.colidx[p]]; d.qn[i] = q; }
  }
  if (a.mode == MODE_TRAN) {   // long rows (rails): the whole workgroup strides over the row, fixed-order reduction
    for (int h = 0; h < d.n_heavy_rows; ++h) {
      const int i = d.heavy_rows[h];
      const int r0 = d.rowptr[i], r1 = d.rowptr[i + 1];
      double q = 0.0;
      for (int p = r0 + t; p < r1; p += 1024) q += d.Cval[p] * scale * d.dx[d.colidx[p]];
      __syncthreads();
      s2[t] = q;
      __syncthreads();
      for (int o = 512; o > 0; o >>= 1) { if (t < o) s2[t] += s2[t + o]; __syncthreads(); }
      if (t == 0) d.qn[i] = d.Q[i] + s2[0];
    }
    __syncthreads();
  }
  s2[t] = e2;
  __syncthreads();
  for (int i = t; i < d.n; i += 1024) d.xcur[i] += scale * d.dx[i];
  for (int o = 512; o > 0; o >>= 1) { if (t < o) s2[t] += s2[t + o]; __syncthreads(); }
  if (t == 0) { d.red[2] = s2[0]; d.flag[1] = sbad; }
}

// ---- the O(n) passes on many workgroups ----------------------------------------------------------------------------------------
// sp_norms / sp_update / sp_commit above run ONE workgroup per sample — fine for a chain of a few hundred unknowns, 76 us for the
// 11 266 rows of the coupled 1024-DFF array (one CU of 256).  Stage 1: up to SP_NP workgroups of 256 rows each, every workgroup
// leaves its fixed-order partial sums in part[sample][k][workgroup]; stage 2 (sp_finish_kernel, one workgroup): the partials summed
// in workgroup order — deterministic — into the mapped reduction record.  Rows with long C rows (rails) get a workgroup of their own.
constexpr int SP_NP = 1024;
__device__ __forceinline__ double sp_block_sum1(double x, double* sh) {   // 256 threads, fixed order
  x = sp_wave_sum(x);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = x;
  __syncthreads();
  return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}
constexpr int SP_RB = 16;
__global__ __launch_bounds__(256) void sp_update2_kernel(const NewtonArgs a, const SparseDev d0, const int* act, const double* scale_v, double* part, int nb_rows, double* hrow) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double sh[4];
  double* pp = part + (size_t)d.s * 8 * SP_NP;
  const int t = threadIdx.x, b = blockIdx.x;
  const double scale = scale_v ? scale_v[d.s] : 1.0;
  const bool dead = d.dflag[0] != 0;   // factorisation failed: leave the iterate untouched (the host re-analyses and retries)
  if (b < nb_rows) {
    double e2 = 0.0, bad = 0.0;
    if (!dead) for (int i = b * 256 + t; i < d.n; i += nb_rows * 256) {
      const double dxi = scale * d.dx[i];
      const double xn = d.xcur[i] + dxi;
      if (!(xn == xn) || fabs(xn) > 1e300) bad = 1.0;
      const double tt = dxi * d.w[i]; e2 += tt * tt;
      if (a.mode == MODE_TRAN && d.rowptr[i + 1] - d.rowptr[i] <= 256) { double q = d.Q[i]; for (int p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) q += d.Cval[p] * scale * d.dx[d.colidx[p]]; d.qn[i] = q; }
      d.xcur[i] = xn;   // (the charge rows read dx, never xcur: no ordering between rows)
    }
    e2 = sp_block_sum1(e2, sh);
    bad = sp_block_sum1(bad, sh);
    if (t == 0) { pp[2 * SP_NP + b] = e2; pp[3 * SP_NP + b] = bad; }
  } else if (a.mode == MODE_TRAN) {   // a long row (a rail): SP_RB workgroups, a contiguous slice each, four entries in flight per thread
    const int h = (b - nb_rows) / SP_RB, slice = (b - nb_rows) % SP_RB;
    const int i = d.heavy_rows[h];
    const int q0 = d.rowptr[i], q1 = d.rowptr[i + 1], per = (q1 - q0 + SP_RB - 1) / SP_RB;
    const int r0 = q0 + slice * per, r1 = min(q1, r0 + per);
    double q = 0.0;
    if (!dead) for (int p = r0 + t; p < r1; p += 4 * 256) {
      const int l = r1 - 1;
      const int p1 = min(p + 256, l), p2 = min(p + 512, l), p3 = min(p + 768, l);
      const double c0 = d.Cval[p], c1 = d.Cval[p1], c2 = d.Cval[p2], c3 = d.Cval[p3];
      const double x0 = d.dx[d.colidx[p]], x1 = d.dx[d.colidx[p1]], x2 = d.dx[d.colidx[p2]], x3 = d.dx[d.colidx[p3]];
      q += c0 * scale * x0;
      if (p + 256 < r1) q += c1 * scale * x1;
      if (p + 512 < r1) q += c2 * scale * x2;
      if (p + 768 < r1) q += c3 * scale * x3;
    }
    q = sp_block_sum1(q, sh);
    if (t == 0) hrow[((size_t)d.s * d.n_heavy_rows + h) * SP_RB + slice] = q;
  }
}
__global__ __launch_bounds__(256) void sp_commit2_kernel(const NewtonArgs a, const SparseDev d0, const int* act, int use_q_of_eval, double* part, int nb_rows) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double sh[4];
  double* pp = part + (size_t)d.s * 8 * SP_NP;
  const int t = threadIdx.x, b = blockIdx.x;
  double e2k = 0, e2m = 0, e2p = 0, nd = 0;
  for (int i = b * 256 + t; i < d.n; i += nb_rows * 256) {
    const double xn = d.xcur[i];
    a.X[(long)a.cand_slot * a.slot_stride + d.xofs + i] = xn;
    a.Qh[(long)a.cand_slot * a.slot_stride + d.xofs + i] = use_q_of_eval ? d.Q[i] : d.qn[i];
    if (a.obs_row) { const int ob = a.unk_obs[i]; if (ob >= 0) a.obs_row[(long)ob * a.S + d.s] = xn; }
    if (a.mode == MODE_TRAN && (a.dmask[i] & 1)) {
      const double x0 = a.X[(long)a.hist_slot[0] * a.slot_stride + d.xofs + i];
      const double w = 1.0 / (a.reltol * fmax(fabs(x0), fabs(xn)) + a.abstol);
      nd += 1.0;
      double tt = (xn - d.xpred[i]) * w; e2k += tt * tt;
      if (a.nkm1 > 0) { double p = 0.0; for (int j = 0; j < a.nkm1; ++j) p += a.wkm1[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + d.xofs + i]; tt = (xn - p) * w; e2m += tt * tt; }
      if (a.nkp1 > 0) { double p = 0.0; for (int j = 0; j < a.nkp1; ++j) p += a.wkp1[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + d.xofs + i]; tt = (xn - p) * w; e2p += tt * tt; }
    }
  }
  e2k = sp_block_sum1(e2k, sh); e2m = sp_block_sum1(e2m, sh); e2p = sp_block_sum1(e2p, sh); nd = sp_block_sum1(nd, sh);
  if (t == 0) { pp[4 * SP_NP + b] = e2k; pp[5 * SP_NP + b] = e2m; pp[6 * SP_NP + b] = e2p; pp[7 * SP_NP + b] = nd; }
}
// what: 0 max|F| -> part[0], 1 max|dx| over node rows -> part[1]
__global__ __launch_bounds__(256) void sp_norms2_kernel(const NewtonArgs a, const SparseDev d0, const int* act, int what, double* part, int nb_rows) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double sh[4];
  double* pp = part + (size_t)d.s * 8 * SP_NP;
  const int t = threadIdx.x, b = blockIdx.x;
  double m = 0.0;
  for (int i = b * 256 + t; i < d.n; i += nb_rows * 256) {
    if (what == 0) m = fmax(m, fabs(d.F[i]));
    else if (!(a.dmask[i] & 2)) m = fmax(m, fabs(d.dx[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  __syncthreads();
  if ((t & 63) == 0) sh[t >> 6] = m;
  __syncthreads();
  if (t == 0) pp[what * SP_NP + b] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
// stage 2: kind 0 = update (red[2] = sum, flag[1] = any bad), 1 = commit (red[4..7] = sums), 2 / 3 = norms (red[0] / red[1] = max)
__global__ __launch_bounds__(256) void sp_finish_kernel(const SparseDev d0, const int* act, const double* part, int nb_rows, int kind, const double* hrow, int tran) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double sh[4];
  const double* pp = part + (size_t)d.s * 8 * SP_NP;
  const int t = threadIdx.x;
  if (kind == 0 && tran && !d.dflag[0]) for (int h = t; h < d.n_heavy_rows; h += 256) {   // the long rows of the charge update: their slices in order
    const double* hp = hrow + ((size_t)d.s * d.n_heavy_rows + h) * SP_RB;
    double q = 0.0;
    for (int k = 0; k < SP_RB; ++k) q += hp[k];
    const int i = d.heavy_rows[h];
    d.qn[i] = d.Q[i] + q;
  }
  if (kind >= 2) {
    const int k = kind - 2;
    double m = 0.0;
    for (int b = t; b < nb_rows; b += 256) m = fmax(m, pp[k * SP_NP + b]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
    __syncthreads();
    if ((t & 63) == 0) sh[t >> 6] = m;
    __syncthreads();
    if (t == 0) d.red[k] = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
    return;
  }
  const int k0 = kind == 0 ? 2 : 4, k1 = kind == 0 ? 4 : 8;
  for (int k = k0; k < k1; ++k) {
    double s = 0.0;
    for (int b = t; b < nb_rows; b += 256) s += pp[k * SP_NP + b];
    s = sp_block_sum1(s, sh);
    if (t == 0) { if (kind == 0 && k == 3) d.flag[1] = s > 0.0 ? 1 : 0; else d.red[k] = s; }
    __syncthreads();
  }
}

// commit candidate state, observables, local-error sums: red[4..7] = e2k, e2km1, e2kp1, ndiff
__global__ __launch_bounds__(1024) void sp_commit_kernel(const NewtonArgs a, const SparseDev d0, const int* act, int use_q_of_eval) {
  const SparseDev d = sp_pick(d0, act);
  __shared__ double s0[1024], s1[1024], s2[1024], s3[1024];
  const int t = threadIdx.x;
  double e2k = 0, e2m = 0, e2p = 0, nd = 0;
  for (int i = t; i < d.n; i += 1024) {
    const double xn = d.xcur[i];
    a.X[(long)a.cand_slot * a.slot_stride + d.xofs + i] = xn;
    a.Qh[(long)a.cand_slot * a.slot_stride + d.xofs + i] = use_q_of_eval ? d.Q[i] : d.qn[i];
    if (a.obs_row) { const int ob = a.unk_obs[i]; if (ob >= 0) a.obs_row[(long)ob * a.S + d.s] = xn; }
    if (a.mode == MODE_TRAN && (a.dmask[i] & 1)) {
      const double x0 = a.X[(long)a.hist_slot[0] * a.slot_stride + d.xofs + i];
      const double w = 1.0 / (a.reltol * fmax(fabs(x0), fabs(xn)) + a.abstol);
      nd += 1.0;
      double tt = (xn - d.xpred[i]) * w; e2k += tt * tt;
      if (a.nkm1 > 0) { double p = 0.0; for (int j = 0; j < a.nkm1; ++j) p += a.wkm1[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + d.xofs + i]; tt = (xn - p) * w; e2m += tt * tt; }
      if (a.nkp1 > 0) { double p = 0.0; for (int j = 0; j < a.nkp1; ++j) p += a.wkp1[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + d.xofs + i]; tt = (xn - p) * w; e2p += tt * tt; }
    }
  }
  s0[t] = e2k; s1[t] = e2m; s2[t] = e2p; s3[t] = nd;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (t < o) { s0[t] += s0[t + o]; s1[t] += s1[t + o]; s2[t] += s2[t + o]; s3[t] += s3[t + o]; } __syncthreads(); }
  if (t == 0) { d.red[4] = s0[0]; d.red[5] = s1[0]; d.red[6] = s2[0]; d.red[7] = s3[0]; }
}

}  // namespace chip
