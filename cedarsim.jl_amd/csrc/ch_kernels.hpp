// ch_kernels.hpp — HIP kernels of the Newton engine (gfx950, wave64).
//
// newton_block_kernel: ONE launch = the whole Newton solve of one time-step attempt (or one DC
// attempt) for every independent Jacobian block of every sample.  One 64-lane wavefront owns one
// block (component c, sample s):
//   prologue   predictor x_pred = Σ w_j x_{n-j} and BDF history term hq = Σ_{j>=1} α_j q_{n+1-j},
//              gathered from the history ring in HBM (coalesced: unknowns of a block are contiguous)
//   iteration  (1) device evaluation, one lane per device instance (BSIM4 with dual numbers; R, C,
//                  L, V, I, VCVS, VCCS), terminal voltages read from LDS / the known-node table,
//                  40-double element stamps staged in LDS
//              (2) deterministic gather (segmented sum over precomputed lists, no atomics) of the
//                  stamps into the dense LDS-resident block Jacobian A = G + α0·C, C, F and Q
//              (3) in-LDS LU with partial pivoting + forward/back substitution by the same wave
//              (4) update, first-order consistent charge q(x+dx) ≈ q(x) + C·dx, WRMS test
//   epilogue   candidate state/charge written to the ring, local-error sums for orders k-1, k, k+1
// This replaces, per Newton iteration of the reference: prob.f.f + prob.f.jac (DAECompiler-generated,
// benchmarks/benchmark_common.jl:138,155) and IDA's dense LU + solve.
#pragma once
#include <hip/hip_runtime.h>

#include "ch_analysis.hpp"
#include "ch_bsim4.hpp"

namespace chip {

struct ClassMeta {
  int nc, ndev, nonlinear, pad;
  int mat_ptr_ofs, mat_src_ofs, vec_ptr_ofs, vec_src_ofs;  // offsets into the pooled gather arrays
};

struct BlockOut {
  int status;  // 0 converged, 1 not converged, 2 singular / non-finite
  int iters;
  int ndiff;
  int pad;
  double e2k, e2km1, e2kp1;  // sums of squares of weighted local-error terms (differential unknowns)
  double fnorm;              // DC: final residual inf-norm
};

struct Summary {
  int n_fail, max_iters, n_singular, pad;
  long long sum_iters;
  long long sum_block_iters;
  double errk, errkm1, errkp1;  // max over samples of WRMS
  double fnorm;
};

enum { MODE_DC = 0, MODE_TRAN = 1, MODE_EVAL = 2 };

struct NewtonArgs {
  // ---- circuit structure ----
  const int* comp_class; const int* comp_uofs; const int* comp_dofs;
  const ClassMeta* classes; const int* gl_ptr; const uint16_t* gl_src;
  const int* dkind; const int* dterm; const int* dsrc; const int* dcls; const int* dhdev;
  const double* dpar; const double* dmult;   // [n_hdev * Spar]
  const double* mosp; long mos_cols;         // packed BSIM4 table [B4I_COUNT][mos_cols]
  const double* kv; const double* srcv;      // known-node values [Ssrc][nk], source values [Ssrc][nsrc]
  const unsigned char* dmask;                // per unknown: bit0 differential, bit1 branch current
  const unsigned char* active;               // per block (DC restarts) or null
  const double* gmin_s;                      // [Sgmin]
  int n_comp, S, Spar, Ssrc, Smos, Sgmin, nk, nsrc, n_unk, n_mos_cls;
  // ---- state ring: X, Qh are [n_slots][S][n_unk] ----
  double* X; double* Qh; long slot_stride;
  int hist_slot[8]; int cand_slot;
  // ---- step ----
  int mode, k, npred, nkm1, nkp1, maxit, lte_valid;
  double alpha[8], wpred[8], wkm1[8], wkp1[8];
  double ck, ckm1, ckp1;
  double abstol, reltol, newton_tol, dc_abstol, dv_max, gshunt;
  // ---- outputs ----
  BlockOut* out;
  double* dumpA; double* dumpF; double* dumpQ; int dump_stride;  // MODE_EVAL: per-block dense dumps
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Evaluate one device and write its 40-slot stamp record.
__device__ __forceinline__ void eval_device(const NewtonArgs& a, int s, int d, const double* xl, int uofs, double* st) {
  const int kind = a.dkind[d];
  const int* tm = a.dterm + 4 * d;
  const double* kvs = a.kv + (long)(a.Ssrc > 1 ? s : 0) * a.nk;
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int t = tm[k]; v[k] = t >= 0 ? xl[t - uofs] : kvs[-t - 1]; }
  const int hd = a.dhdev[d];
  const long pi = (long)hd * a.Spar + (a.Spar > 1 ? s : 0);
  const double m = a.dmult[pi];
  switch (kind) {
    case K_R: {
      const double g = m / a.dpar[pi], i = g * (v[0] - v[1]);
      st[0] = i; st[1] = -i; st[4] = 0.0; st[5] = 0.0;
      st[8] = g; st[9] = -g; st[12] = -g; st[13] = g;
      st[24] = 0.0; st[25] = 0.0; st[28] = 0.0; st[29] = 0.0;
    } break;
    case K_C: {
      const double c = m * a.dpar[pi], q = c * (v[0] - v[1]);
      st[0] = 0.0; st[1] = 0.0; st[4] = q; st[5] = -q;
      st[8] = 0.0; st[9] = 0.0; st[12] = 0.0; st[13] = 0.0;
      st[24] = c; st[25] = -c; st[28] = -c; st[29] = c;
    } break;
    case K_I: {
      const double i = m * a.srcv[(long)(a.Ssrc > 1 ? s : 0) * a.nsrc + a.dsrc[d]];
      st[0] = i; st[1] = -i; st[4] = 0.0; st[5] = 0.0;
    } break;
    case K_V: case K_L: case K_VCVS_A: {
      // terminals (a, b, branch): KCL rows get ±m·i, branch row: va - vb - V(t) [- d/dt(L i)]
      const double ib = v[2];
      const double src = kind == K_V ? a.srcv[(long)(a.Ssrc > 1 ? s : 0) * a.nsrc + a.dsrc[d]] : 0.0;
      const double l = kind == K_L ? a.dpar[pi] : 0.0;
      st[0] = m * ib; st[1] = -m * ib; st[2] = v[0] - v[1] - src;
      st[4] = 0.0; st[5] = 0.0; st[6] = -l * ib;
      st[8 + 2] = m; st[8 + 6] = -m; st[8 + 8] = 1.0; st[8 + 9] = -1.0; st[8 + 10] = 0.0;
      st[24 + 2] = 0.0; st[24 + 6] = 0.0; st[24 + 8] = 0.0; st[24 + 9] = 0.0; st[24 + 10] = -l;
    } break;
    case K_VCVS_B: {
      // terminals (branch, c, d): branch row gets -gain·(vc - vd)
      const double g = a.dpar[pi];
      st[0] = -g * (v[1] - v[2]); st[4] = 0.0;
      st[8 + 1] = -g; st[8 + 2] = g; st[24 + 1] = 0.0; st[24 + 2] = 0.0;
    } break;
    case K_VCCS: {
      const double g = m * a.dpar[pi], i = g * (v[2] - v[3]);
      st[0] = i; st[1] = -i; st[4] = 0.0; st[5] = 0.0;
      st[8 + 2] = g; st[8 + 3] = -g; st[8 + 6] = -g; st[8 + 7] = g;
      st[24 + 2] = 0.0; st[24 + 3] = 0.0; st[24 + 6] = 0.0; st[24 + 7] = 0.0;
    } break;
    case K_MOS: {
      B4Col P{a.mosp, a.mos_cols, (long)a.dcls[d] * a.Smos + (a.Smos > 1 ? s : 0)};
      double o[40];
      b4_device(P, v[0], v[1], v[2], v[3], a.gmin_s[a.Sgmin > 1 ? s : 0], o);
#pragma unroll
      for (int j = 0; j < 40; ++j) st[j] = m * o[j];
    } break;
  }
}

// LDS layout (doubles): st[ndev_max*40] | A[nc*(nc+1)] | Cm[nc*nc] | xl xp F Q hq dx w qn [8*nc]
__global__ __launch_bounds__(64) void newton_block_kernel(const NewtonArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const int blk = blockIdx.x;
  const int c = blk / a.S, s = blk - c * a.S;
  const ClassMeta cm = a.classes[a.comp_class[c]];
  const int nc = cm.nc, ndev = cm.ndev, uofs = a.comp_uofs[c], dofs = a.comp_dofs[c];
  const int lda = nc + 1;
  double* st = lds;
  double* A = st + (size_t)ndev * 40;
  double* Cm = A + (size_t)nc * lda;
  double* xl = Cm + (size_t)nc * nc;
  double* xp = xl + nc; double* Fv = xp + nc; double* Qv = Fv + nc; double* hq = Qv + nc;
  double* dxv = hq + nc; double* wv = dxv + nc; double* qn = wv + nc;
  const int* mptr = a.gl_ptr + cm.mat_ptr_ofs; const uint16_t* msrc = a.gl_src + cm.mat_src_ofs;
  const int* vptr = a.gl_ptr + cm.vec_ptr_ofs; const uint16_t* vsrc = a.gl_src + cm.vec_src_ofs;
  const long sofs = (long)s * a.n_unk + uofs;
  const double* X0 = a.X + (long)a.hist_slot[0] * a.slot_stride + sofs;
  BlockOut bo; bo.status = 1; bo.iters = 0; bo.ndiff = 0; bo.pad = 0; bo.e2k = bo.e2km1 = bo.e2kp1 = 0.0; bo.fnorm = 0.0;

  if (a.active && !a.active[blk]) {  // DC restart pass: this block already converged
    return;
  }
  // ---- prologue ----
  for (int i = lane; i < nc; i += 64) {
    double x0 = X0[i];
    double p = x0, h = 0.0;
    if (a.mode == MODE_TRAN) {
      p = 0.0;
      for (int j = 0; j < a.npred; ++j) p += a.wpred[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + sofs + i];
      for (int j = 1; j <= a.k; ++j) h += a.alpha[j] * a.Qh[(long)a.hist_slot[j - 1] * a.slot_stride + sofs + i];
    }
    xp[i] = p; xl[i] = p; hq[i] = h;
    wv[i] = 1.0 / (a.reltol * fabs(x0) + a.abstol);
  }
  __syncthreads();
  const double alpha0 = a.mode == MODE_DC ? 0.0 : a.alpha[0];
  int status = 1, iters = 0;
  double fnorm = 0.0;
  const int maxit = a.mode == MODE_EVAL ? 1 : a.maxit;
  for (int it = 0; it <= maxit; ++it) {
    // (1) device evaluation → staging
    for (int d = lane; d < ndev; d += 64) eval_device(a, s, dofs + d, xl, uofs, st + (size_t)d * 40);
    __syncthreads();
    // (2) gather
    for (int e = lane; e < nc * nc; e += 64) {
      double g = 0.0, cc = 0.0;
      for (int p = mptr[e]; p < mptr[e + 1]; ++p) { const int o = msrc[p]; g += st[o]; cc += st[o + 16]; }
      const int r = e / nc, col = e - r * nc;
      if (r == col && a.gshunt != 0.0 && !(a.dmask[uofs + r] & 2)) g += a.gshunt;  // node rows only
      A[r * lda + col] = g + alpha0 * cc;
      Cm[e] = cc;
    }
    for (int i = lane; i < nc; i += 64) {
      double f = 0.0, q = 0.0;
      for (int p = vptr[i]; p < vptr[i + 1]; ++p) { const int o = vsrc[p]; f += st[o]; q += st[o + 4]; }
      if (a.gshunt != 0.0 && !(a.dmask[uofs + i] & 2)) f += a.gshunt * xl[i];
      Qv[i] = q;
      const double F = f + alpha0 * q + hq[i];
      Fv[i] = F;
      A[i * lda + nc] = -F;
    }
    __syncthreads();
    if (a.mode == MODE_EVAL) {
      if (a.dumpA) {
        double* dA = a.dumpA + (long)blk * a.dump_stride * a.dump_stride;
        for (int e = lane; e < nc * nc; e += 64) dA[e] = A[(e / nc) * lda + (e % nc)];
        for (int i = lane; i < nc; i += 64) { a.dumpF[(long)blk * a.dump_stride + i] = Fv[i]; a.dumpQ[(long)blk * a.dump_stride + i] = Qv[i]; }
      }
      for (int i = lane; i < nc; i += 64) qn[i] = Qv[i];
      status = 0;
      break;
    }
    // residual norm (DC convergence test happens on the residual, src/dcop.jl:171-173)
    {
      double m = 0.0;
      for (int i = lane; i < nc; i += 64) m = fmax(m, fabs(Fv[i]));
      fnorm = wave_max(m);
    }
    if (!(fnorm == fnorm) || fnorm > 1e300) { status = 2; break; }
    if (a.mode == MODE_DC && fnorm < a.dc_abstol) { for (int i = lane; i < nc; i += 64) qn[i] = Qv[i]; status = 0; break; }
    if (it == maxit) break;
    // (3) LU with partial pivoting on the augmented matrix [A | -F] in LDS
    bool singular = false;
    for (int k = 0; k < nc; ++k) {
      // pivot search over rows k..nc-1 of column k
      double best = -1.0; int bi = k;
      for (int i = k + lane; i < nc; i += 64) { const double v = fabs(A[i * lda + k]); if (v > best) { best = v; bi = i; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (!(best > 0.0)) { singular = true; break; }
      if (bi != k) {
        for (int j = lane; j <= nc; j += 64) { const double t = A[k * lda + j]; A[k * lda + j] = A[bi * lda + j]; A[bi * lda + j] = t; }
        __syncthreads();
      }
      const double inv = 1.0 / A[k * lda + k];
      const int rem = nc - k - 1;
      // multipliers
      for (int i = k + 1 + lane; i < nc; i += 64) A[i * lda + k] *= inv;
      __syncthreads();
      // rank-1 update of the trailing block and the rhs column
      const int w = rem + 1;  // columns k+1..nc (incl. rhs)
      for (int e = lane; e < rem * w; e += 64) {
        const int i = k + 1 + e / w, j = k + 1 + e % w;
        A[i * lda + j] -= A[i * lda + k] * A[k * lda + j];
      }
      __syncthreads();
    }
    if (singular) { status = 2; break; }
    // back substitution on the rhs column
    for (int k = nc - 1; k >= 0; --k) {
      const double xk = A[k * lda + nc] / A[k * lda + k];
      __syncthreads();
      if (lane == 0) A[k * lda + nc] = xk;
      for (int i = lane; i < k; i += 64) A[i * lda + nc] -= A[i * lda + k] * xk;
      __syncthreads();
    }
    ++iters;
    // (4) update
    double scale = 1.0;
    if (a.mode == MODE_DC && a.dv_max > 0.0 && cm.nonlinear) {  // linear blocks take the full Newton step
      double m = 0.0;
      for (int i = lane; i < nc; i += 64) if (!(a.dmask[uofs + i] & 2)) m = fmax(m, fabs(A[i * lda + nc]));
      m = wave_max(m);
      if (m > a.dv_max) scale = a.dv_max / m;
    }
    double e2 = 0.0; bool bad = false;
    for (int i = lane; i < nc; i += 64) {
      const double dx = scale * A[i * lda + nc];
      dxv[i] = dx;
      const double xn = xl[i] + dx;
      xl[i] = xn;
      if (!(xn == xn) || fabs(xn) > 1e300) bad = true;
      const double t = dx * wv[i];
      e2 += t * t;
    }
    __syncthreads();
    if (a.mode == MODE_TRAN) {
      for (int i = lane; i < nc; i += 64) {
        double q = Qv[i];
        for (int j = 0; j < nc; ++j) q += Cm[i * nc + j] * dxv[j];
        qn[i] = q;
      }
    }
    e2 = wave_sum(e2);
    const unsigned long long anybad = __ballot(bad);
    if (anybad) { status = 2; break; }
    if (a.mode == MODE_TRAN && sqrt(e2 / nc) <= a.newton_tol) { status = 0; __syncthreads(); break; }
    __syncthreads();
  }
  // ---- epilogue ----
  double* Xc = a.X + (long)a.cand_slot * a.slot_stride + sofs;
  double* Qc = a.Qh + (long)a.cand_slot * a.slot_stride + sofs;
  double e2k = 0.0, e2m = 0.0, e2p = 0.0; int nd = 0;
  for (int i = lane; i < nc; i += 64) {
    const double xn = xl[i];
    Xc[i] = xn;
    Qc[i] = qn[i];
    if (a.mode == MODE_TRAN && (a.dmask[uofs + i] & 1)) {
      const double x0 = X0[i];
      const double w = 1.0 / (a.reltol * fmax(fabs(x0), fabs(xn)) + a.abstol);
      ++nd;
      double t = (xn - xp[i]) * w; e2k += t * t;
      if (a.nkm1 > 0) { double p = 0.0; for (int j = 0; j < a.nkm1; ++j) p += a.wkm1[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + sofs + i]; t = (xn - p) * w; e2m += t * t; }
      if (a.nkp1 > 0) { double p = 0.0; for (int j = 0; j < a.nkp1; ++j) p += a.wkp1[j + 1] * a.X[(long)a.hist_slot[j] * a.slot_stride + sofs + i]; t = (xn - p) * w; e2p += t * t; }
    }
  }
  e2k = wave_sum(e2k); e2m = wave_sum(e2m); e2p = wave_sum(e2p);
  nd = (int)wave_sum((double)nd);
  if (lane == 0) {
    bo.status = status; bo.iters = iters; bo.ndiff = nd; bo.e2k = e2k; bo.e2km1 = e2m; bo.e2kp1 = e2p; bo.fnorm = fnorm;
    a.out[blk] = bo;
  }
}

// Reduce per-block outputs: WRMS per sample (over all blocks of the sample), max over samples.
// Many samples: one thread per sample loops over the sample's blocks.  Few samples (a single large
// circuit): the 256 threads split the blocks of one sample and tree-reduce in LDS.
struct RedAcc {
  double a, b, c, f; long long nd, bits; int nfail, nsing, mx;
};
__device__ __forceinline__ void red_add(RedAcc& x, const BlockOut& o) {
  x.a += o.e2k; x.b += o.e2km1; x.c += o.e2kp1; x.nd += o.ndiff; x.bits += o.iters;
  if (o.status != 0) ++x.nfail;
  if (o.status == 2) ++x.nsing;
  x.mx = max(x.mx, o.iters); x.f = fmax(x.f, o.fnorm);
}
__global__ __launch_bounds__(256) void reduce_blocks_kernel(const BlockOut* out, int n_comp, int S, double ck, double ckm1, double ckp1,
                                                            const unsigned char* active, Summary* sum) {
  __shared__ double sk[256], sm[256], sp[256], sf[256];
  __shared__ int sfail[256], smax[256], ssing[256];
  __shared__ long long sit[256], sbi[256], snd[256];
  const int t = threadIdx.x;
  double mk_ = 0.0, mm = 0.0, mp = 0.0, mf = 0.0; int nfail = 0, mx = 0, nsing = 0; long long its = 0, bits = 0;
  if (S >= 64) {
    for (int s = t; s < S; s += 256) {
      RedAcc x{0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < n_comp; ++k) { const int blk = k * S + s; if (active && !active[blk]) continue; red_add(x, out[blk]); }
      its += x.mx; mx = max(mx, x.mx); nfail += x.nfail; nsing += x.nsing; bits += x.bits; mf = fmax(mf, x.f);
      if (x.nd > 0) { mk_ = fmax(mk_, ck * sqrt(x.a / x.nd)); mm = fmax(mm, ckm1 * sqrt(x.b / x.nd)); mp = fmax(mp, ckp1 * sqrt(x.c / x.nd)); }
    }
  } else {
    for (int s = 0; s < S; ++s) {
      RedAcc x{0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = t; k < n_comp; k += 256) { const int blk = k * S + s; if (active && !active[blk]) continue; red_add(x, out[blk]); }
      sk[t] = x.a; sm[t] = x.b; sp[t] = x.c; sf[t] = x.f; snd[t] = x.nd; sbi[t] = x.bits; sfail[t] = x.nfail; ssing[t] = x.nsing; smax[t] = x.mx;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
          sk[t] += sk[t + o]; sm[t] += sm[t + o]; sp[t] += sp[t + o]; sf[t] = fmax(sf[t], sf[t + o]); snd[t] += snd[t + o]; sbi[t] += sbi[t + o];
          sfail[t] += sfail[t + o]; ssing[t] += ssing[t + o]; smax[t] = max(smax[t], smax[t + o]);
        }
        __syncthreads();
      }
      if (t == 0) {
        its += smax[0]; mx = max(mx, smax[0]); nfail += sfail[0]; nsing += ssing[0]; bits += sbi[0]; mf = fmax(mf, sf[0]);
        if (snd[0] > 0) { mk_ = fmax(mk_, ck * sqrt(sk[0] / snd[0])); mm = fmax(mm, ckm1 * sqrt(sm[0] / snd[0])); mp = fmax(mp, ckp1 * sqrt(sp[0] / snd[0])); }
      }
      __syncthreads();
    }
  }
  sk[t] = mk_; sm[t] = mm; sp[t] = mp; sf[t] = mf; sfail[t] = nfail; smax[t] = mx; ssing[t] = nsing; sit[t] = its; sbi[t] = bits;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      sk[t] = fmax(sk[t], sk[t + o]); sm[t] = fmax(sm[t], sm[t + o]); sp[t] = fmax(sp[t], sp[t + o]); sf[t] = fmax(sf[t], sf[t + o]);
      sfail[t] += sfail[t + o]; smax[t] = max(smax[t], smax[t + o]); ssing[t] += ssing[t + o]; sit[t] += sit[t + o]; sbi[t] += sbi[t + o];
    }
    __syncthreads();
  }
  if (t == 0) {
    Summary r; r.n_fail = sfail[0]; r.max_iters = smax[0]; r.n_singular = ssing[0]; r.pad = 0; r.sum_iters = sit[0]; r.sum_block_iters = sbi[0];
    r.errk = sk[0]; r.errkm1 = sm[0]; r.errkp1 = sp[0]; r.fnorm = sf[0];
    *sum = r;
  }
}

// Save observables: out[o*S + s] = Σ_j w[j] X[slot_j][s][idx_o]  (idx < 0 → NaN placeholder, filled on host)
struct ObsArgs { const double* X; long slot_stride; int slots[8]; double w[8]; int nw; int n_unk, S, n_obs; const int* obs_unk; double* dst; };
__global__ void save_obs_kernel(const ObsArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_obs * a.S) return;
  const int o = i / a.S, s = i - o * a.S;
  const int u = a.obs_unk[o];
  double v = 0.0;
  if (u >= 0) for (int j = 0; j < a.nw; ++j) v += a.w[j] * a.X[(long)a.slots[j] * a.slot_stride + (long)s * a.n_unk + u];
  a.dst[i] = v;
}

// Stand-alone BSIM4 evaluation (roofline / parity kernel): one lane per (instance, sample-0).
__global__ __launch_bounds__(64) void mos_eval_kernel(const double* mosp, long mos_cols, const int* cls, int Smos, int sample, int n_mos, const double* v, double gmin, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_mos) return;
  B4Col P{mosp, mos_cols, (long)cls[i] * Smos + (Smos > 1 ? sample : 0)};
  double o[40];
  b4_device(P, v[4 * i + 0], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3], gmin, o);
  for (int j = 0; j < 40; ++j) out[(long)i * 40 + j] = o[j];
}

}  // namespace chip
