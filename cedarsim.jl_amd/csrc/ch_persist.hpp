// ch_persist.hpp — device-resident transient: ONE launch integrates the whole time span.
//
// The host stepper of ch_engine.hip pays a launch + completion round trip per step attempt (17.7 us of the 54 us an
// attempt of the 1024-DFF array took in round 1).  Here the step controller itself runs on the GPU: every wavefront owns one
// Jacobian block for the whole transient, keeps the block's BDF history ring, class lists and BSIM4 columns in LDS, and
// after each attempt the per-block outcomes (Newton status, local-error sums of orders k-1, k, k+1) are combined across
// the grid by an XCD-hierarchical reduction that doubles as the grid barrier.  Every wave then takes the SAME
// accept / reject / order / step-size decision from the same reduced numbers — it is still one sequential controller
// ("the outer adaptive timestepper stays sequential", BASELINE.json north_star; IDA's job in the reference,
// src/sweeps.jl:456), replicated instead of broadcast.  Source waveforms (src/spectre_env.jl:15-77,144-176) and the
// variable-coefficient BDF / predictor weights are evaluated on the device, lane-parallel.
//
// Residency: n_wg workgroups of 4 waves, one workgroup per CU (<= 1024 blocks); the launch is cooperative, so a grid that
// cannot be co-resident is refused by the runtime instead of dead-locking, and every spin is bounded by a wall-clock limit.
// Shapes handled: S == 1 (one circuit of independent blocks: the error norm is the WRMS over ALL blocks) or n_comp == 1
// (a batch of single-block samples: per-sample WRMS, maximum over samples).  Everything else keeps the host stepper.
#pragma once
#include <hip/hip_runtime.h>

#include "ch_kernels.hpp"

namespace chip {

constexpr int PW = 4;          // blocks (= wavefronts) per workgroup
constexpr int P_MAXSRC = 64;   // sources evaluated per attempt: one lane each
constexpr int P_NREC = 8;      // doubles per reduction record
enum { PX_RUNNING = 0, PX_DONE = 1, PX_ROWS_FULL = 2, PX_ABORT = 3 };

// controller state + statistics: written by workgroup 0 at exit, read back at start when `resume` is set
struct TranCtl {
  double t, h;
  double tslot[8];            // times of the history points, newest first (canonical order at exit)
  int k, nhist, steps_at_order, reset_rate;
  int ibp, isave, status, exit_reason;
  long long nsaved, step;
  long long naccept, nreject, nconvfail, sum_iters, sum_block_iters, n_attempts;
  long long t_cycles_total, t_cycles_barrier;   // wave 0 of workgroup 0: cycles inside the kernel / inside the grid reductions
};

struct PersistArgs {
  NewtonArgs a;               // structure pointers, tolerances, state ring (read at start, written back at exit)
  int nblk, n_wg, red_max;    // red_max 1: n_comp == 1 (max over samples of the per-sample WRMS); 0: S == 1 (WRMS over all blocks)
  int wave_doubles;           // LDS doubles per wave region
  double t1, dtmin, dtmax, first_frac;
  int kmax, max_steps, nbp, n_saveat;
  const double* bps; const double* saveat;
  const int* ci; const double* cd; int n_ci, n_cd;   // constants blob (sources, known-node definitions), copied to LDS
  double* out_times; double* out_rows; long long max_rows; int n_obs;
  TranCtl* ctl; int resume;
  double* wg_rec; double* grp_rec; unsigned* counters;   // grid reduction: [n_wg][8], [8][8], 10 counters on 128-byte lines
  long long spin_ticks;       // bound of every spin, in wall_clock64 ticks (100 MHz)
};

// ---- constants blob layout -------------------------------------------------------------------
// ints:    [0] n_need  [1] nk  [2] nds  [3] n_pwl  | per needed source {kind, pwl_ofs, pwl_len} | kn_ptr[nk+1] | kn_idx[..] | ds_idx[nds]
// doubles: per needed source par[8] | kn_coef[..] | pwl_t[n_pwl] | pwl_y[n_pwl]
struct PConst {
  const int* ci; const double* cd;
  __device__ int n_need() const { return ci[0]; }
  __device__ int nk() const { return ci[1]; }
  __device__ int nds() const { return ci[2]; }
  __device__ int n_pwl() const { return ci[3]; }
  __device__ const int* src_rec(int i) const { return ci + 4 + 3 * i; }
  __device__ const int* kn_ptr() const { return ci + 4 + 3 * n_need(); }
  __device__ const int* kn_idx() const { return kn_ptr() + nk() + 1; }
  __device__ const int* ds_idx() const { return kn_idx() + kn_ptr()[nk()]; }
  __device__ const double* par(int i) const { return cd + 8 * i; }
  __device__ const double* kn_coef() const { return cd + 8 * n_need(); }
  __device__ const double* pwl_t() const { return kn_coef() + kn_ptr()[nk()]; }
  __device__ const double* pwl_y() const { return pwl_t() + n_pwl(); }
};

// pwl_at_time (src/spectre_env.jl:15-21,43-69): a break point belongs to the NEXT segment; ends are held; flat and
// zero-width segments have their special cases.  ts/ys live in LDS.
__device__ inline double p_pwl(const double* ts, const double* ys, int n, double t) {
  if (n == 0) return 0.0;
  int lo = 0, hi = n;   // lower_bound: first index with ts[idx] >= t
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (ts[mid] < t) lo = mid + 1; else hi = mid; }
  int i = lo + 1;
  if (i <= n && ts[i - 1] == t) ++i;
  if (i <= 1) return ys[0];
  if (i > n) return ys[n - 1];
  const double y0 = ys[i - 2], y1 = ys[i - 1], t0 = ts[i - 2], t1 = ts[i - 1];
  if (y0 == y1) return y1;
  if (t1 == t0) return 0.5 * (y0 + y1);
  return y0 + (t - t0) * ((y1 - y0) / (t1 - t0));
}
// the same for the four corners of a pulse held in registers
__device__ inline double p_pwl4(double c0, double c1, double c2, double c3, double v1, double v2, double t) {
  // ts = {c0,c1,c2,c3}, ys = {v1,v2,v2,v1}
  int lb = (c0 < t) + (c1 < t) + (c2 < t) + (c3 < t);   // lower_bound (the corners are non-decreasing)
  int i = lb + 1;
  const double tlb = lb == 0 ? c0 : lb == 1 ? c1 : lb == 2 ? c2 : c3;
  if (i <= 4 && tlb == t) ++i;
  if (i <= 1) return v1;
  if (i > 4) return v1;
  const double t0 = i == 2 ? c0 : i == 3 ? c1 : c2, t1 = i == 2 ? c1 : i == 3 ? c2 : c3;
  const double y0 = i == 2 ? v1 : v2, y1 = i == 4 ? v1 : v2;
  if (y0 == y1) return y1;
  if (t1 == t0) return 0.5 * (y0 + y1);
  return y0 + (t - t0) * ((y1 - y0) / (t1 - t0));
}
// transient value of needed source i at time t (mode :tran; pulse :153-166, spsin :169-176)
__device__ inline double p_source(const PConst& C, int i, double t) {
  const int* r = C.src_rec(i);
  const double* par = C.par(i);
  switch (r[0]) {
    case CH_SRC_DC: return par[0];
    case CH_SRC_PWL: return p_pwl(C.pwl_t() + r[1], C.pwl_y() + r[1], r[2], t);
    case CH_SRC_PULSE: {
      const double td = par[2], tr = par[3], tf = par[4], pw = par[5], per = par[6];
      const double tt = (per == per && fabs(per) < __builtin_inf()) ? fmod(t, per) : t;
      return p_pwl4(td, td + tr, td + tr + pw, td + tr + pw + tf, par[0], par[1], tt);
    }
    case CH_SRC_SIN: {
      const double vo = par[0], va = par[1], f = par[2], td = par[3], th = par[4], ph = par[5], nc = par[6];
      const double rad = 3.14159265358979323846 / 180.0;
      if (td < t && t < nc / f) return vo + va * exp(-(t - td) * th) * sin(fmod(360.0 * f * (t - td) + ph, 360.0) * rad);
      return vo + va * sin(fmod(ph, 360.0) * rad);
    }
  }
  return 0.0;
}

// relaxed agent-scope accesses: global_load/store ... sc1 — L2-coherent hand-off without cache-wide fences
__device__ __forceinline__ double ld_agent(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait until *cnt >= target; false when the abort flag was raised or the spin bound was hit (then the flag is raised here)
__device__ inline bool p_wait(const unsigned* cnt, unsigned target, unsigned* abort_flag, long long spin_ticks) {
  const long long t0 = wall_clock64();
  for (;;) {
    if (ld_agent_u(cnt) >= target) return true;
    if (ld_agent_u(abort_flag) != 0u) return false;
    if (wall_clock64() - t0 > spin_ticks) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    __builtin_amdgcn_s_sleep(2);
  }
}

// combine the values held by lanes (m, f) = (lane >> 3, lane & 7) over m in a fixed order (xor tree over lane bits 3..5)
__device__ __forceinline__ double p_tree(double v, bool is_max) {
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) { const double u = __shfl_xor(v, o); v = is_max ? fmax(v, u) : v + u; }
  return v;
}

// Grid-wide reduction of one 8-double record per wave; every wave of the grid returns with the combined record in summ[].
// Record: [0..2] error sums (or per-sample ratios when red_max), [3] ndiff, [4] max iterations, [5] sum of iterations, [6] failures.
// Fields 0..2 are combined with + (S == 1) or max (n_comp == 1); 3, 5, 6 with +; 4 with max.  Returns false on abort.
__device__ inline bool p_grid_reduce(const PersistArgs& p, unsigned gen, const double* rec /* 8 wave-uniform values */, double* part, double* summ,
                                     int* s_abort, int wave, int lane, int wg) {
  const int f = lane & 7, m = lane >> 3;
  const bool fmaxop = (f == 4) || (p.red_max && f < 3);
  if (lane < P_NREC) {
    double v = rec[0];
#pragma unroll
    for (int q = 1; q < P_NREC; ++q) v = (f == q) ? rec[q] : v;
    part[wave * P_NREC + f] = v;
  }
  __syncthreads();
  const int n_grp = p.n_wg < 8 ? p.n_wg : 8;
  const int grp = wg & 7;
  unsigned* abort_flag = p.counters + 9 * 32;
  if (wave == 0) {
    bool ok = true;
    double v = 0.0;
    if (lane < P_NREC) {
      v = part[f];
#pragma unroll
      for (int w = 1; w < PW; ++w) { const double u = part[w * P_NREC + f]; v = fmaxop ? fmax(v, u) : v + u; }
    }
    if (p.n_wg > 1) {
      if (lane < P_NREC) st_agent(p.wg_rec + (long)wg * P_NREC + f, v);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(p.counters + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (wg < n_grp) {   // group leader: fold the records of the workgroups wg, wg + 8, wg + 16, ...
        const int members = (p.n_wg - grp + 7) >> 3;
        ok = p_wait(p.counters + grp * 32, gen * (unsigned)members, abort_flag, p.spin_ticks);
        double acc = 0.0;
        for (int base = 0; base < members; base += 8) {
          const int mm = base + m;
          double u = (ok && mm < members) ? ld_agent(p.wg_rec + (long)(grp + 8 * mm) * P_NREC + f) : 0.0;
          u = p_tree(u, fmaxop);
          acc = fmaxop ? fmax(acc, u) : acc + u;
        }
        if (lane < P_NREC) st_agent(p.grp_rec + grp * P_NREC + f, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(p.counters + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ok = p_wait(p.counters + 8 * 32, gen * (unsigned)n_grp, abort_flag, p.spin_ticks) && ok;
      double u = (ok && m < n_grp) ? ld_agent(p.grp_rec + m * P_NREC + f) : 0.0;
      v = p_tree(u, fmaxop);
    }
    if (lane < P_NREC) summ[f] = v;
    if (lane == 0) *s_abort = ok ? 0 : 1;
  }
  __syncthreads();
  return *s_abort == 0;
}

// (num, den) of one coefficient of the attempt, by lane (all lanes then divide once):
//   set 0, j = 0..7   BDF: alpha_j (j >= 1); the j = 0 slot is filled afterwards from the reciprocals of set 4
//   set 1..3, j = 1..7 Lagrange extrapolation weights of the predictors of order k (np points), k-1 (kk points), k+1 (kk+2 points)
//   set 4, j = 1..5   1 / (tau_0 - tau_j) (summed into alpha_0) ; j = 6: ck, j = 7: ckm1 ; set 5, j = 0: ckp1
// tau[0] = t_new, tau[1..] = history times (newest first).
__device__ inline void p_coefficients(const double* tauv, int kk, int np, int nkm1, int nkp1, double hh, double* coef, int lane) {
  double tau[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) tau[i] = tauv[i];
  const int set = lane >> 3, j = lane & 7;
  double tj = tau[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) tj = (j == i) ? tau[i] : tj;
  double num = 0.0, den = 1.0;
  const int n = set == 1 ? np : set == 2 ? nkm1 : set == 3 ? nkp1 : 0;
  if (set >= 1 && set <= 3 && j >= 1 && j <= n) {
    num = 1.0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i <= n && i != j) { num *= (tau[0] - tau[i]); den *= (tj - tau[i]); }
  } else if (set == 0 && j >= 1 && j <= kk) {
    num = 1.0;
#pragma unroll
    for (int i = 1; i < 8; ++i) if (i <= kk && i != j) num *= (tau[0] - tau[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i <= kk && i != j) den *= (tj - tau[i]);
  } else if (set == 4 && j >= 1 && j <= 5) {
    if (j <= kk) { num = 1.0; den = tau[0] - tj; }
  } else if (set == 4 && j == 6) {   // ck = hh / (tn - tau[kk+1]) when the error estimate exists
    double tq = tau[1];
#pragma unroll
    for (int i = 2; i < 9; ++i) tq = (kk + 1 == i) ? tau[i] : tq;
    if (np >= kk + 1) { num = hh; den = tau[0] - tq; }
  } else if (set == 4 && j == 7) {   // ckm1 = hh / (tn - tau[kk])
    double tq = tau[1];
#pragma unroll
    for (int i = 2; i < 9; ++i) tq = (kk == i) ? tau[i] : tq;
    if (nkm1 > 0) { num = hh; den = tau[0] - tq; }
  } else if (set == 5 && j == 0) {   // ckp1 = hh / (tn - tau[kk+2])
    double tq = tau[1];
#pragma unroll
    for (int i = 2; i < 9; ++i) tq = (kk + 2 == i) ? tau[i] : tq;
    if (nkp1 > 0) { num = hh; den = tau[0] - tq; }
  }
  const double q = num / den;
  if (lane < 48) coef[lane] = q;
  wave_fence();
  if (lane == 0) { double a0 = 0.0; for (int mq = 1; mq <= kk; ++mq) a0 += coef[32 + mq]; coef[0] = a0; }
  wave_fence();
}

// LDS per workgroup: [consts: cd doubles | ci ints] [part PW*8 | summ 8] [PW wave regions]
// wave region (doubles): st[ndev*41] | A[nc*(nc+1)] | Cm[nc*nc] | xl xp F Q hq w qn pm pp x0 dm [11*nc] | Xh[8*nc] | Qh[8*nc] | tauv[10] | coef[48] | asrc[64]
//                        | kvl[nk] svl[nsrc] | pl[max_mc*B4L_STRIDE] | ints: class blob, MOS class list
template <int NC>
__global__ __launch_bounds__(PW * 64, 1) void tran_persistent_kernel(const PersistArgs p) {
  typedef StampLayout<false> SL;
  extern __shared__ double lds[];
  __shared__ int s_abort;
  const NewtonArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  const long long cyc0 = wall_clock64();
  long long cyc_bar = 0;
  // ---- workgroup-shared constants ----
  double* cdl = lds;
  int* cil = (int*)(cdl + p.n_cd);
  double* part = cdl + p.n_cd + ((p.n_ci + 1) >> 1);
  double* summ = part + PW * P_NREC;
  double* W = summ + P_NREC + (size_t)wave * p.wave_doubles;
  for (int i = tid; i < p.n_cd; i += PW * 64) cdl[i] = p.cd[i];
  for (int i = tid; i < p.n_ci; i += PW * 64) cil[i] = p.ci[i];
  if (tid == 0) s_abort = 0;
  __syncthreads();
  const PConst C{cil, cdl};

  const int blk = wg * PW + wave;
  const bool live = blk < p.nblk;
  const int bq = live ? blk : 0;
  const int c = bq / a.S, s = bq - c * a.S;
  const BlockMeta bm = a.bmeta[c];
  const ClassMeta cm = bm.cm;
  const int nc = cm.nc, ndev = cm.ndev, uofs = bm.uofs, dofs = bm.dofs, lda = nc + 1;
  double* st = W;
  double* A = st + (size_t)ndev * SL::STRIDE;
  double* Cm = A + (size_t)nc * lda;
  double* xl = Cm + (size_t)nc * nc;
  double* xp = xl + nc; double* Fv = xp + nc; double* Qv = Fv + nc; double* hq = Qv + nc;
  double* wv = hq + nc; double* qn = wv + nc; double* pm = qn + nc; double* pp = pm + nc; double* x0l = pp + nc;
  int* dml = (int*)(x0l + nc);
  double* Xh = x0l + 2 * nc; double* Qh = Xh + 8 * nc;
  double* tauv = Qh + 8 * nc; double* coef = tauv + 10; double* asrc = coef + 48;
  double* kvl = asrc + P_MAXSRC; double* svl = kvl + a.nk;
  double* pl = svl + a.nsrc;
  int* mptr = (int*)(pl + (size_t)a.max_mc * B4L_STRIDE);
  int* slots = mptr + (nc * nc + 1) + (nc + 1);
  uint16_t* msrc = (uint16_t*)(slots + cm.nslots);
  const int2* wl = (const int2*)(mptr + cm.wl_ofs);
  const long sofs = (long)s * a.n_unk + uofs;
  const bool mine = live && lane < nc;

  // ---- once per transient: class blob, BSIM4 columns, flags, history ring ----
  if (live) {
    const int* bsrc = a.blob + cm.blob_ofs;
    for (int i = lane; i < cm.blob_ints; i += 64) mptr[i] = bsrc[i];
    const long scol = a.Smos > 1 ? s : 0;
    const int total = bm.mc_n * B4I_COUNT;
    for (int e = lane; e < total; e += 64) {
      const int j = e / B4I_COUNT, i = e - j * B4I_COUNT;
      int cl = bm.mc[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) cl = (j == q) ? bm.mc[q] : cl;
      pl[j * B4L_STRIDE + i] = a.mosp[((long)cl * a.Smos + scol) * (long)B4I_COUNT + i];
    }
    if (mine) {
      dml[lane] = a.dmask[uofs + lane] | ((a.unk_obs[uofs + lane] + 1) << 8);
#pragma unroll
      for (int j = 0; j < 7; ++j) {   // global slot j holds the j-th newest point (canonical order); with head = 0 that is ring slot (-j) & 7
        Xh[((8 - j) & 7) * nc + lane] = a.X[(long)j * a.slot_stride + sofs + lane];
        Qh[((8 - j) & 7) * nc + lane] = a.Qh[(long)j * a.slot_stride + sofs + lane];
      }
      Xh[1 * nc + lane] = 0.0; Qh[1 * nc + lane] = 0.0;   // ring slot 1 = first candidate
    }
  }
  // ---- controller state (identical in every wave) ----
  double t, h; int k, nhist, steps_at_order, ibp, isave, status = CH_OK, exit_reason = PX_RUNNING;
  bool reset_rate;
  long long nsaved, step, naccept, nreject, nconvfail, sum_iters, sum_block_iters, n_attempts;
  int head = 0;   // ring slot of the newest point; slot (head - j) & 7 holds the j-th newest, (head + 1) & 7 the candidate
  {
    const TranCtl* cs = p.ctl;
    t = cs->t; h = cs->h; k = cs->k; nhist = cs->nhist; steps_at_order = cs->steps_at_order; reset_rate = cs->reset_rate != 0;
    ibp = cs->ibp; isave = cs->isave; nsaved = cs->nsaved; step = cs->step;
    naccept = cs->naccept; nreject = cs->nreject; nconvfail = cs->nconvfail; sum_iters = cs->sum_iters; sum_block_iters = cs->sum_block_iters; n_attempts = cs->n_attempts;
    if (lane < 8) tauv[lane] = 0.0;   // placeholder; history times live in tsl below
  }
  double tsl[8];   // times by ring slot
#pragma unroll
  for (int j = 0; j < 8; ++j) tsl[(8 - j) & 7] = p.ctl->tslot[j];   // canonical (newest first) -> ring slots, head = 0
  double rate_prev = 1.0;
  wave_fence();
  __syncthreads();

  const EvalCtx ectx{a.dkind, a.dterm, a.dsrc, a.dcls_local, a.dhdev, a.dpar, a.dmult, a.Spar, a.gmin_s[a.Sgmin > 1 ? s : 0], a.vapar, 300.15};
  const int n_need = C.n_need(), nkk = C.nk(), nds = C.nds();
  const long long row_stride = (long long)p.n_obs * a.S;
  unsigned gen = 0;

  // rows due at t0 (the initial state): the host leaves them to the kernel only when it starts fresh
  if (!p.resume) {
    if (p.n_saveat == 0) {
      if (mine) { const int ob = (dml[lane] >> 8) - 1; if (ob >= 0) p.out_rows[nsaved * row_stride + (long long)ob * a.S + s] = Xh[lane]; }
      if (blk == 0 && lane == 0) p.out_times[nsaved] = t;
      ++nsaved;
    } else {
      while (isave < p.n_saveat && p.saveat[isave] <= t) {
        if (mine) { const int ob = (dml[lane] >> 8) - 1; if (ob >= 0) p.out_rows[nsaved * row_stride + (long long)ob * a.S + s] = Xh[lane]; }
        if (blk == 0 && lane == 0) p.out_times[nsaved] = p.saveat[isave];
        ++nsaved; ++isave;
      }
    }
  }

  while (step < p.max_steps && t < p.t1) {
    if (p.n_saveat == 0 && nsaved >= p.max_rows) { exit_reason = PX_ROWS_FULL; break; }
    while (ibp < p.nbp && p.bps[ibp] <= t * (1 + 1e-15) + 1e-300) ++ibp;
    const double tb = ibp < p.nbp ? p.bps[ibp] : p.t1;
    bool hit_bp = false;
    double tn = t + h;
    if (tn >= tb - 1e-3 * h) { tn = tb; hit_bp = true; }
    const double hh = tn - t;
    if (hh < p.dtmin) { status = CH_ERR_DTMIN; break; }
    const int nh = nhist, kk = k < nh ? k : nh, np = (kk + 1) < nh ? (kk + 1) : nh;
    const bool lte = np >= kk + 1;
    const bool try_up = lte && kk < p.kmax && nh >= kk + 2 && steps_at_order + 1 >= kk + 1;
    const int nkm1 = (lte && kk > 1) ? kk : 0, nkp1 = try_up ? kk + 2 : 0;
    // history times, newest first
    if (lane < 9) {
      double v = tn;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int sl = (head - j) & 7; double ts = tsl[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) ts = (sl == q) ? tsl[q] : ts;
        v = (lane == j + 1) ? ts : v; }
      tauv[lane] = v;
    }
    wave_fence();
    p_coefficients(tauv, kk, np, nkm1, nkp1, hh, coef, lane);
    const double alpha0 = coef[0], ck = coef[32 + 6], ckm1 = coef[32 + 7], ckp1 = coef[40];
    // sources at t_new (their left limit when the step lands on a break point), then the known-node values
    {
      const double ts = hit_bp ? nextafter(tn, -__builtin_inf()) : tn;
      if (lane < n_need) asrc[lane] = p_source(C, lane, ts);
      wave_fence();
      if (lane < nkk) { double v = 0.0; for (int q = C.kn_ptr()[lane]; q < C.kn_ptr()[lane + 1]; ++q) v += C.kn_coef()[q] * asrc[C.kn_idx()[q]]; kvl[lane] = v; }
      if (lane < nds) svl[lane] = asrc[C.ds_idx()[lane]];
    }
    // ---- predictor and history term from the ring ----
    const int cand = (head + 1) & 7;
    double x0 = 0.0;
    if (mine) {
      double pr = 0.0, hs = 0.0, m1 = 0.0, p1 = 0.0;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int sl = (head - j) & 7;
        const double xv = Xh[sl * nc + lane];
        if (j == 0) x0 = xv;
        pr += (j < np ? coef[8 + j + 1] : 0.0) * xv;
        m1 += (j < nkm1 ? coef[16 + j + 1] : 0.0) * xv;
        p1 += (j < nkp1 ? coef[24 + j + 1] : 0.0) * xv;
        if (j < 5) hs += (j < kk ? coef[j + 1] : 0.0) * Qh[sl * nc + lane];
      }
      xp[lane] = pr; xl[lane] = pr; hq[lane] = hs; qn[lane] = 0.0; pm[lane] = m1; pp[lane] = p1;
      wv[lane] = 1.0 / (a.reltol * fabs(x0) + a.abstol);
      x0l[lane] = x0;
    }
    if (live) for (int i = lane; i < nc * lda + nc * nc; i += 64) A[i] = 0.0;
    wave_fence();

    // ---- Newton iteration (the register-LU branch of newton_block_kernel, one wave, no workgroup barriers) ----
    int nstat = 1, iters = 0;
    double rate_new = -1.0, dn_prev = 0.0;
    const double rp = reset_rate ? 1.0 : rate_prev;
    if (live) {
      for (int it = 0; it <= a.maxit; ++it) {
        if (lane < cm.nslots) { const int sl = slots[lane]; if (sl >= 0) eval_slot<false>(ectx, s, dofs, sl, xl, uofs, kvl, svl, pl, st); }
        wave_fence();
        for (int w = lane; w < cm.n_work; w += 64) {
          const int2 itw = wl[w];
          const int p0 = itw.x, pe = p0 + (int)((unsigned)itw.y >> 16), e = itw.y & 0x7fff;
          const bool vec = itw.y & 0x8000;
          const int off2 = vec ? SL::QO : SL::CO;
          double s1 = 0.0, s2 = 0.0;
          for (int q = p0; q < pe; q += 4) {
            const int l = pe - 1;
            const int o0 = msrc[q], o1 = msrc[min(q + 1, l)], o2 = msrc[min(q + 2, l)], o3 = msrc[min(q + 3, l)];
            const double a0 = st[o0], b0 = st[o0 + off2], a1 = st[o1], b1 = st[o1 + off2], a2 = st[o2], b2 = st[o2 + off2], a3 = st[o3], b3 = st[o3 + off2];
            s1 += a0; s2 += b0;
            if (q + 1 < pe) { s1 += a1; s2 += b1; }
            if (q + 2 < pe) { s1 += a2; s2 += b2; }
            if (q + 3 < pe) { s1 += a3; s2 += b3; }
          }
          if (vec) {
            Qv[e] = s2;
            const double F = s1 + alpha0 * s2 + hq[e];
            Fv[e] = F;
            A[e * lda + nc] = -F;
          } else {
            const int r = e / nc, col = e - r * nc;
            A[r * lda + col] = s1 + alpha0 * s2;
            Cm[e] = s2;
          }
        }
        wave_fence();
        constexpr int NCR = NC;
        double r[NCR + 1], cr[NCR];
#pragma unroll
        for (int j = 0; j <= NCR; ++j) r[j] = (mine && j <= nc) ? A[lane * lda + j] : 0.0;
#pragma unroll
        for (int j = 0; j < NCR; ++j) cr[j] = (mine && j < nc) ? Cm[lane * nc + j] : 0.0;
        const double Fi = mine ? Fv[lane] : 0.0, Qi = mine ? Qv[lane] : 0.0, xi = mine ? xl[lane] : 0.0, wi = mine ? wv[lane] : 0.0;
        const double fnorm = bcast(row_max<NCR>(fabs(Fi)), 0);
        bool stop = false;
        if (!(fnorm == fnorm) || fnorm > 1e300) { nstat = 2; stop = true; }
        else if (it == a.maxit) { stop = true; }
        else {
          double dx = 0.0;
          const bool ok = lu_solve_regs<NCR>(r, nc, lane, dx);
          if (!ok) { nstat = 2; stop = true; }
          else {
            const double xn = xi + dx;
            if (mine) xl[lane] = xn;
            const bool bad = mine && (!(xn == xn) || fabs(xn) > 1e300);
            const double tq = dx * wi;
            const double e2 = bcast(row_sum<NCR>(mine ? tq * tq : 0.0), 0);
            double q = Qi;
#pragma unroll
            for (int j = 0; j < NCR; ++j) if (j < nc) q = fma(cr[j], bcast(dx, j), q);
            if (mine) qn[lane] = q;
            ++iters;
            if (__ballot(bad)) { nstat = 2; stop = true; }
            else {
              const double dn = sqrt(e2 / nc);
              if (it == 0) { if (dn <= a.newton_tol || (rp < 0.9 && 2.0 * fmax(rp, 0.02) * dn <= a.newton_tol)) { nstat = 0; stop = true; } }
              else { rate_new = dn_prev > 0.0 ? dn / dn_prev : 0.0; if (dn <= a.newton_tol) { nstat = 0; stop = true; } }
              dn_prev = dn;
            }
          }
        }
        wave_fence();
        if (stop) break;
      }
    }
    // ---- candidate into the ring, local-error sums ----
    double e2k = 0.0, e2m = 0.0, e2p = 0.0, ndf = 0.0;
    if (mine) {
      const double xn = xl[lane];
      Xh[cand * nc + lane] = xn;
      Qh[cand * nc + lane] = qn[lane];
      if (dml[lane] & 1) {
        const double w = 1.0 / (a.reltol * fmax(fabs(x0), fabs(xn)) + a.abstol);
        ndf = 1.0;
        double tq = (xn - xp[lane]) * w; e2k = tq * tq;
        if (nkm1 > 0) { tq = (xn - pm[lane]) * w; e2m = tq * tq; }
        if (nkp1 > 0) { tq = (xn - pp[lane]) * w; e2p = tq * tq; }
      }
    }
    e2k = wave_sum(e2k); e2m = wave_sum(e2m); e2p = wave_sum(e2p); ndf = wave_sum(ndf);
    if (live && nstat == 0) rate_prev = iters >= 2 ? fmin(1.0, fmax(rate_new, 1e-4)) : fmin(1.0, rp * 1.5);
    double rec[P_NREC];
    if (p.red_max) { const double inv = ndf > 0.0 ? 1.0 / ndf : 0.0; rec[0] = e2k * inv; rec[1] = e2m * inv; rec[2] = e2p * inv; }
    else { rec[0] = e2k; rec[1] = e2m; rec[2] = e2p; }
    rec[3] = ndf; rec[4] = (double)iters; rec[5] = (double)iters; rec[6] = (live && nstat != 0) ? 1.0 : 0.0; rec[7] = 0.0;
    if (!live) { for (int q = 0; q < P_NREC; ++q) rec[q] = 0.0; }
    ++gen; ++n_attempts;
    const long long cb0 = wall_clock64();
    const bool okr = p_grid_reduce(p, gen, rec, part, summ, &s_abort, wave, lane, wg);
    cyc_bar += wall_clock64() - cb0;
    if (!okr) { exit_reason = PX_ABORT; status = CH_ERR_DEVICE; break; }
    const double sA = summ[0], sB = summ[1], sC = summ[2], sN = summ[3], sItMax = summ[4], sItSum = summ[5], sFail = summ[6];
    __syncthreads();   // summ is rewritten by the next reduction
    double errk = 0.0, errkm1 = 0.0, errkp1 = 0.0;
    if (p.red_max) { errk = ck * sqrt(sA); errkm1 = ckm1 * sqrt(sB); errkp1 = ckp1 * sqrt(sC); }
    else if (sN > 0.0) { errk = ck * sqrt(sA / sN); errkm1 = ckm1 * sqrt(sB / sN); errkp1 = ckp1 * sqrt(sC / sN); }
    sum_block_iters += (long long)sItSum;
    sum_iters += p.red_max ? (long long)sItSum : (long long)sItMax;
    // ---- the step controller (same policy as ch_circuit::tran_solve) ----
    if (sFail > 0.0) {
      ++nconvfail; reset_rate = true;
      h = hh * 0.25; k = 1; steps_at_order = 0;
      if (nhist > 2) nhist = 2;
      continue;
    }
    if (!lte) errk = 0.0;
    // the three candidate factors (2 err + 1e-4)^(-1/(order+1)) are formed by three lanes at once
    double fk, fm, fp;
    {
      const double ev = lane == 0 ? errk : lane == 1 ? errkm1 : errkp1;
      const double ex = lane == 0 ? (double)(kk + 1) : lane == 1 ? (double)kk : (double)(kk + 2);
      const double fv = exp(-log(2.0 * ev + 1e-4) / ex);
      fk = bcast(fv, 0); fm = bcast(fv, 1); fp = bcast(fv, 2);
    }
    if (errk > 1.0) {
      ++nreject;
      h = hh * fmin(0.9, fmax(0.25, 0.9 * fk));
      steps_at_order = 0;
      continue;
    }
    // ---- accept ----
    ++naccept; ++step; reset_rate = false;
    head = cand;
#pragma unroll
    for (int q = 0; q < 8; ++q) tsl[q] = (head == q) ? tn : tsl[q];
    nhist = nhist + 1 < p.kmax + 2 ? nhist + 1 : p.kmax + 2;
    if (p.n_saveat > 0) {
      while (isave < p.n_saveat && p.saveat[isave] <= tn * (1 + 1e-15)) {
        const double tsv = p.saveat[isave];
        const int msv = (kk < nh ? kk : nh) + 1;
        // interpolation weights through the msv newest points (the accepted one included)
        if (lane < 9) {
          double v = tsv;
#pragma unroll
          for (int j = 0; j < 8; ++j) { const int sl = (head - j) & 7; double ts = tsl[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) ts = (sl == q) ? tsl[q] : ts;
            v = (lane == j + 1) ? ts : v; }
          tauv[lane] = v;
        }
        wave_fence();
        p_coefficients(tauv, 0, msv, 0, 0, 0.0, coef, lane);
        if (mine) {
          const int ob = (dml[lane] >> 8) - 1;
          if (ob >= 0) {
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < 7; ++j) if (j < msv) v += coef[8 + j + 1] * Xh[((head - j) & 7) * nc + lane];
            p.out_rows[nsaved * row_stride + (long long)ob * a.S + s] = v;
          }
        }
        if (blk == 0 && lane == 0) p.out_times[nsaved] = tsv;
        ++nsaved; ++isave;
        wave_fence();
      }
    } else {
      if (mine) { const int ob = (dml[lane] >> 8) - 1; if (ob >= 0) p.out_rows[nsaved * row_stride + (long long)ob * a.S + s] = Xh[head * nc + lane]; }
      if (blk == 0 && lane == 0) p.out_times[nsaved] = tn;
      ++nsaved;
    }
    // ---- order / step selection ----
    double best = fk; int knew = kk;
    if (lte) {
      ++steps_at_order;
      if (kk > 1 && fm > best) { best = fm; knew = kk - 1; }
      if (try_up && fp > 1.1 * best) { best = fp; knew = kk + 1; }
    } else knew = 1;
    if (knew != kk) steps_at_order = 0;
    k = knew;
    if (best > 1.0 && best < 1.2) best = 1.0;
    h = fmin(p.dtmax, hh * fmin(kk == 1 ? 10.0 : 2.0, fmax(0.5, best)));
    t = tn;
    if (hit_bp && t < p.t1) {
      nhist = 1; k = 1; steps_at_order = 0; reset_rate = true;
      double nb = p.t1;
      for (int b = ibp; b < p.nbp; ++b) if (p.bps[b] > t * (1 + 1e-15)) { nb = p.bps[b]; break; }
      h = fmax(p.dtmin * 10, fmin(h, (nb - t) / 50.0) * p.first_frac);
    }
  }
  if (exit_reason == PX_RUNNING) {
    exit_reason = PX_DONE;
    if (status == CH_OK && t < p.t1) status = CH_ERR_MAXSTEPS;
  }
  // ---- write the ring back in canonical order (slot j = j-th newest) and the controller state ----
  if (mine) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int sl = (head - j) & 7;
      a.X[(long)j * a.slot_stride + sofs + lane] = Xh[sl * nc + lane];
      a.Qh[(long)j * a.slot_stride + sofs + lane] = Qh[sl * nc + lane];
    }
  }
  if (wg == 0 && wave == 0 && lane == 0) {
    TranCtl* cs = p.ctl;
    cs->t = t; cs->h = h; cs->k = k; cs->nhist = nhist; cs->steps_at_order = steps_at_order; cs->reset_rate = reset_rate ? 1 : 0;
    cs->ibp = ibp; cs->isave = isave; cs->status = status; cs->exit_reason = exit_reason; cs->nsaved = nsaved; cs->step = step;
    cs->naccept = naccept; cs->nreject = nreject; cs->nconvfail = nconvfail; cs->sum_iters = sum_iters; cs->sum_block_iters = sum_block_iters; cs->n_attempts = n_attempts;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int sl = (head - j) & 7; double ts = tsl[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) ts = (sl == q) ? tsl[q] : ts;
      cs->tslot[j] = ts; }
    cs->t_cycles_total = wall_clock64() - cyc0; cs->t_cycles_barrier = cyc_bar;
  }
}

}  // namespace chip
