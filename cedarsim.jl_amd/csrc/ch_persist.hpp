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
// cannot be co-resident is refused by the runtime instead of dead-locking, and every spin is bounded by a wall-clock limit
// (a launch that shared the GPU with another process gives up there and the host repeats the solve on the host stepper).
// Three forms, separate instantiations of one kernel (template parameter MODE):
//   PM_LOCKSTEP  one step sequence for the whole grid — S == 1 (one circuit of independent blocks: the error norm is the WRMS over
//                ALL blocks) or n_comp == 1 (a batch of single-block samples: per-sample WRMS, maximum over samples);
//   PM_OWN       output on a saveat grid: every sample / block (pair) its own controller and step sequence, no grid-wide
//                reduction, any number of queued workgroups, per-workgroup source tables and break points for one circuit;
//   PM_BORDER    a coupled array torn at one or two border unknowns (ch_analysis.hpp): register LU per block + a grid-wide Schur
//                complement per Newton iteration, transient and operating point.
// Everything else keeps the host stepper.
#pragma once
#include <hip/hip_runtime.h>

#include "ch_kernels.hpp"

#ifdef CH_STAMPS
#define P_STAMP(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); pacc_[slot] += now_ - plast_; plast_ = now_; } while (0)
#else
#define P_STAMP(slot) do { } while (0)
#endif

namespace chip {

constexpr int PW = 4;          // blocks (= wavefronts) per workgroup
constexpr int P_MAXSRC = 64;   // sources evaluated per attempt: one lane each
constexpr int P_NREC = 8;      // doubles per reduction record
constexpr int P_SCR = 64;      // doubles of cross-row scratch of the grid reduction (wave 0 of the workgroup)
enum { PX_RUNNING = 0, PX_DONE = 1, PX_ROWS_FULL = 2, PX_ABORT = 3 };

// controller state + statistics: written by workgroup 0 at exit, read back at start when `resume` is set
struct TranCtl {
  double t, h;
  double tslot[8];            // times of the history points, newest first (canonical order at exit)
  int k, nhist, steps_at_order, reset_rate;
  int ibp, isave, status, exit_reason;
  long long nsaved, step;
  long long naccept, nreject, nconvfail, sum_iters, sum_block_iters, n_attempts;
  long long max_iters;                          // per-block / per-sample steps: Newton iterations of the slowest block
  long long t_cycles_total, t_cycles_barrier;   // wave 0 of workgroup 0: cycles inside the kernel / inside the grid reductions
  long long stamps[12];                         // diagnostic build (-DCH_STAMPS): shader-clock cycles per phase, wave 0 of workgroup 0
};

struct PersistArgs {
  NewtonArgs a;               // structure pointers, tolerances, state ring (read at start, written back at exit)
  int nblk, n_wg, red_max;    // red_max 1: n_comp == 1 (max over samples of the per-sample WRMS); 0: S == 1 (WRMS over all blocks)
  int bpw;                    // blocks per workgroup: PW, or 2 (one wave pair per CU, the other two waves idle) to spread few, heavy blocks over more CUs
  int wide_l, wide_other;     // WIDE + PAIR: lanes of one half of the block's split compiled devices, and its other (unsplit) evaluation slots
  int wave_doubles;           // LDS doubles per wave region
  int va_arena;               // WIDE: LDS doubles behind the wave regions for the workgroup's copies of its compiled devices' parameter and constant blocks
  double t1, dtmin, dtmax, first_frac;
  int kmax, max_steps, nbp, n_saveat;
  const double* bps; const double* saveat;   // bps: [nbp times | nbp codes (< 0: the sources jump there; else the length of the source segment starting there)]; per workgroup the same pair at its offset
  const int* ci; const double* cd; int n_ci, n_cd;   // constants blob (sources, known-node definitions), copied to LDS
  // per-block steps of ONE circuit: a blob per workgroup with the sources of its own blocks only.  wgc[6 wg ..] = {ci offset,
  // cd offset, n_ci, n_cd, break-point offset, break-point count}; n_ci / n_cd above are then the largest sizes (LDS layout);
  // wgk[wg][entry] = the circuit-wide known-node index (entries 0 .. nk_local-1) or device-source slot (the rest) of the workgroup's entries
  const int* wgc; const int* wgk;
  double* out_times; double* out_rows; long long max_rows; int n_obs;
  TranCtl* ctl; int resume;
  double* wg_rec; double* grp_rec; unsigned* counters;   // grid reduction: [n_wg][8], [8][8], 10 counters on 128-byte lines
  long long spin_ticks;       // bound of every spin, in wall_clock64 ticks (100 MHz)
  int pair_dbg;               // diagnostic: 1 = the first wave of a pair evaluates both halves itself (A/B of the function split)
  int indep;                  // 1: a batch of single-block samples on a common saveat grid — every sample (pair of samples when PAIR)
                              //    runs its own controller: no grid-wide reduction at all, each takes exactly the steps it needs
  // Bordered block-diagonal form (a coupled array torn at its supply rails, ch_analysis.hpp): every block's last nb local unknowns
  // are replicas of the border unknowns.  Per Newton iteration: partial LU of every block, grid-wide SUM of the blocks' Schur
  // contributions, the nb x nb border system solved by every wave, back substitution; a second reduction carries the update
  // norm and the local-error sums, so convergence and step acceptance are decided from the same global numbers by every wave.
  int nb, n_glob, n_bdev;     // border unknowns (0: independent blocks), unknowns of the untorn system, devices on the border alone
  int bd_kind[8], bd_ta[8], bd_tb[8];   // K_R / K_C; terminals: >= 0 border index, < 0 -(known index + 1)
  double bd_val[8];           // conductance / capacitance, multiplicity included
  // Operating point of the bordered form (dc_mode): ONE damped Newton solve at alpha0 = 0 from the state in ring slot 0, entries
  // (known-node and source values in the operating-point mode) evaluated by the host; result back in slot 0, status in ctl
  int dc_mode, dc_maxit; double dc_abstol, dv_max; const double* dc_entries;
};

// ---- constants blob layout -------------------------------------------------------------------
// The values a block reads per attempt are the ENTRIES [known-node values | device source values]; entry e = sum of
// coef * source(t) over its terms (a supply node: one source; a well tied to a supply through a 0 V source: two).
// ints:    [0] n_need  [1] n_ent  [2] n_pwl  [3] -  | per needed source {kind, pwl_ofs, pwl_len} | ent_ptr[n_ent+1] | ent_idx[..]
// doubles: per needed source par[8] | ent_coef[..] | pwl_t[n_pwl] | pwl_y[n_pwl]
struct PConst {
  const int* ci; const double* cd;
  __device__ int n_need() const { return ci[0]; }
  __device__ int n_ent() const { return ci[1]; }
  __device__ int n_pwl() const { return ci[2]; }
  __device__ const int* src_rec(int i) const { return ci + 4 + 3 * i; }
  __device__ const int* ent_ptr() const { return ci + 4 + 3 * n_need(); }
  __device__ const int* ent_idx() const { return ent_ptr() + n_ent() + 1; }
  __device__ const double* par(int i) const { return cd + 8 * i; }
  __device__ const double* ent_coef() const { return cd + 8 * n_need(); }
  __device__ const double* pwl_t() const { return ent_coef() + ent_ptr()[n_ent()]; }
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

// Linear piece of needed source i around time t: on [t_lo, t_hi) the source equals y_lo + (t - t_lo) * slope with the SAME
// arithmetic as p_pwl (so cached and uncached evaluations agree bit for bit).  Sources that are not piecewise linear in t
// (pulse with its period wrap, sine) report an empty window and are evaluated in full at every attempt.
__device__ inline void p_source_piece(const PConst& C, int i, double t, double& t_lo, double& t_hi, double& y_lo, double& slope) {
  const int* r = C.src_rec(i);
  const double inf = __builtin_inf();
  t_lo = inf; t_hi = -inf; y_lo = 0.0; slope = 0.0;
  if (r[0] == CH_SRC_DC) { t_lo = -inf; t_hi = inf; y_lo = C.par(i)[0]; return; }
  if (r[0] != CH_SRC_PWL) return;
  const double* ts = C.pwl_t() + r[1]; const double* ys = C.pwl_y() + r[1];
  const int n = r[2];
  if (n == 0) { t_lo = -inf; t_hi = inf; return; }
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (ts[mid] < t) lo = mid + 1; else hi = mid; }
  int k = lo + 1;
  if (k <= n && ts[k - 1] == t) ++k;
  if (k <= 1) { t_lo = -inf; t_hi = ts[0]; y_lo = ys[0]; return; }
  if (k > n) { t_lo = ts[n - 1]; t_hi = inf; y_lo = ys[n - 1]; return; }
  const double y0 = ys[k - 2], y1 = ys[k - 1], t0 = ts[k - 2], t1 = ts[k - 1];
  if (t1 == t0) return;   // zero-width segment: no window
  t_lo = t0; t_hi = t1;
  if (y0 == y1) { y_lo = y1; slope = 0.0; } else { y_lo = y0; slope = (y1 - y0) / (t1 - t0); }
}

// Ordering of one wave's own LDS traffic (cross-lane hand-offs inside the wave): LDS executes a wave's instructions in order,
// so only the compiler has to be held back.  NOT wave_fence(): a workgroup-scope release also waits for the wave's global
// stores (vmcnt(0)) — the saved-row stores of an accepted step then stall the next attempt by a memory round trip.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

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
// Fields 0..2 are combined with + (S == 1) or max (n_comp == 1); 3, 5, 6, 7 with +; 4 with max.  Returns false on abort.
//
// Transport: data-tagged 8-byte granules {32-bit half of a double | 32-bit generation}, written with ONE relaxed agent-scope
// store each and polled in place — no payload-then-flag pair, no wait for a store to land, no counter (the price list of
// MI355X_MICROARCH.md: a granule hand-off is one fabric latency, a store + drained flag two to three).  A record is 16 granules
// (128 bytes).  Two levels: every workgroup publishes its record; the leaders (workgroups 0..7) sweep the records of workgroups
// g, g+8, g+16, ... and publish a group record; every workgroup sweeps the (at most) eight group records.  All sums are taken
// in a fixed order.  Records are double-buffered by the parity of the generation: a group may publish generation g+1 while
// a slow workgroup is still sweeping generation g, but never g+2.
typedef unsigned long long p_u64;
__device__ __forceinline__ p_u64 ld_gran(const p_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(p_u64* p, p_u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// lanes 2f and 2f+1 of every 16-lane group hold the low and the high half of field f: both get the double back
// (the partner's half comes through a DPP quad permutation — one VALU instruction, no LDS crossbar trip)
__device__ __forceinline__ double p_join(p_u64 gran, int lane) {
  const unsigned mine = (unsigned)gran, other = (unsigned)dpp_i<DPP_QP_1032>((int)mine);
  const unsigned lo = (lane & 1) ? other : mine, hi = (lane & 1) ? mine : other;
  return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ p_u64 p_split(double v, int lane, unsigned gen) {
  const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
  return ((p_u64)gen << 32) | half;
}
// combine the four 16-lane rows of a wave lane by lane, in a fixed order: (row0 op row1) op (row2 op row3).  Through LDS (one store,
// four loads in flight) — the cross-row shuffles of the compiler (ds_bpermute pairs per double and step) were a fifth of the
// reduction's instruction stream.  scr: 64 doubles of the calling wave's own scratch.
__device__ __forceinline__ double p_rows4(double v, bool is_max, double* scr, int lane) {
  scr[lane] = v;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int c = lane & 15;
  const double a0 = scr[c], a1 = scr[16 + c], a2 = scr[32 + c], a3 = scr[48 + c];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return is_max ? fmax(fmax(a0, a1), fmax(a2, a3)) : (a0 + a1) + (a2 + a3);
}
template <bool ALL_SUM = false>   // ALL_SUM: every field combined with + (the Schur-complement and norm records of the bordered form)
__device__ inline bool p_grid_reduce(const PersistArgs& p, unsigned gen, const double* rec /* 8 wave-uniform values */, double* part, double* summ,
                                     int* s_abort, int wave, int lane, int wg) {
  const int g16 = lane & 15, f = g16 >> 1, m4 = lane >> 4;   // granule of the record, its field, member slot of a sweep
  const bool fmaxop = !ALL_SUM && ((f == 4) || (p.red_max && f < 3));
  if (lane < P_NREC) {
    double v = rec[0];
#pragma unroll
    for (int q = 1; q < P_NREC; ++q) v = (lane == q) ? rec[q] : v;
    part[wave * P_NREC + lane] = v;
  }
  __syncthreads();
  const int n_grp = p.n_wg < 8 ? p.n_wg : 8;
  const int grp = wg & 7;
  unsigned* abort_flag = p.counters + 9 * 32;
  p_u64* wrec = (p_u64*)p.wg_rec + (size_t)(gen & 1) * p.n_wg * 16;
  p_u64* grec = (p_u64*)p.grp_rec + (size_t)(gen & 1) * 8 * 16;
  double* scr = part + PW * P_NREC + P_NREC + 4;   // P_SCR doubles behind [part | summ | pair flags]: see the LDS layout of the kernel
  if (wave == 0) {
    bool ok = true;
    double v = part[f];
#pragma unroll
    for (int w = 1; w < PW; ++w) { const double u = part[w * P_NREC + f]; v = fmaxop ? fmax(v, u) : v + u; }
    if (p.n_wg > 1) {
      if (lane < 16) st_gran(wrec + (size_t)wg * 16 + g16, p_split(v, lane, gen));
      const long long t0 = wall_clock64();
      if (wg < n_grp) {   // group leader: the records of workgroups wg, wg + 8, wg + 16, ... (at most 32), four per sweep load
        const int members = (p.n_wg - grp + 7) >> 3;
        p_u64 x[8];
        for (;;) {
          bool all = true;
          const unsigned ab = ld_agent_u(abort_flag);   // travels with the sweep: one round trip per poll, not two
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int mm = q * 4 + m4;
            x[q] = mm < members ? ld_gran(wrec + (size_t)(grp + 8 * mm) * 16 + g16) : ((p_u64)gen << 32);
            all = all && (unsigned)(x[q] >> 32) == gen;
          }
          if (__all(all)) break;
          if (ab != 0u) { ok = false; break; }
          if (wall_clock64() - t0 > p.spin_ticks) { __hip_atomic_store(abort_flag, 1u | (gen << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }   // diagnostic: site 1, generation
          __builtin_amdgcn_s_sleep(1);
        }
        // members q*4 + m4 of this lane's row, q ascending, then the four rows: a fixed order
        double acc = (m4 < members) ? p_join(x[0], lane) : 0.0;
#pragma unroll
        for (int q = 1; q < 8; ++q) {
          const double u = (q * 4 + m4 < members) ? p_join(x[q], lane) : 0.0;
          acc = fmaxop ? fmax(acc, u) : acc + u;
        }
        acc = p_rows4(acc, fmaxop, scr, lane);
        if (lane < 16) st_gran(grec + (size_t)grp * 16 + g16, p_split(acc, lane, gen));
      }
      // every workgroup: the group records (8 x 16 granules = two per lane)
      p_u64 y[2];
      for (;;) {
        bool all = true;
        const unsigned ab = ok ? ld_agent_u(abort_flag) : 1u;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int gg = q * 4 + m4;
          y[q] = (ok && gg < n_grp) ? ld_gran(grec + (size_t)gg * 16 + g16) : ((p_u64)gen << 32);
          all = all && (unsigned)(y[q] >> 32) == gen;
        }
        if (__all(all) || !ok) break;
        if (ab != 0u) { ok = false; break; }
        if (wall_clock64() - t0 > p.spin_ticks) { __hip_atomic_store(abort_flag, 2u | (gen << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }   // site 2
        __builtin_amdgcn_s_sleep(1);
      }
      double acc = (m4 < n_grp) ? p_join(y[0], lane) : 0.0;
      { const double u = (4 + m4 < n_grp) ? p_join(y[1], lane) : 0.0; acc = fmaxop ? fmax(acc, u) : acc + u; }
      v = p_rows4(acc, fmaxop, scr, lane);
    }
    if (lane < 16 && !(lane & 1)) summ[f] = v;
    if (lane == 0) *s_abort = ok ? 0 : 1;
  }
  __syncthreads();
  return *s_abort == 0;
}

// One coefficient of the attempt per lane, all lanes dividing once.  tau0 = the new time (or the dense-output time),
// tsl[] = history times by ring slot, head = slot of the newest point.  lane = 8*set + j:
//   set 0, j = 1..kk      BDF coefficients alpha_j   (alpha_0 = sum of the reciprocals of set 4, returned in a0)
//   set 1..3, j = 1..n    Lagrange extrapolation weights of the predictors of order k (np points), k-1 (kk), k+1 (kk+2)
//   set 4, j = 1..5       1 / (tau0 - tau_j) ; j = 6: ck = hh / (tau0 - tau_{kk+1}) ; j = 7: ckm1 = hh / (tau0 - tau_kk)
//   set 5, j = 0          ckp1 = hh / (tau0 - tau_{kk+2})
// Lanes outside their set's range return 0, so callers sum over all seven history points unconditionally.
__device__ __forceinline__ double p_coef(double tau0, const double* tsl, int head, int kk, int np, int nkm1, int nkp1, bool lte, double hh, int lane, double& a0) {
  double tau[9];
  tau[0] = tau0;
#pragma unroll
  for (int i = 1; i < 9; ++i) tau[i] = tsl[(head - (i - 1)) & 7];
  const int set = lane >> 3, j = lane & 7;
  const double tj = j == 0 ? tau0 : tsl[(head - (j - 1)) & 7];
  const int n = set == 0 ? kk : set == 1 ? np : set == 2 ? nkm1 : set == 3 ? nkp1 : 0;
  const int lo = set == 0 ? 0 : 1;
  double num = 1.0, den = 1.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool in_n = i >= 1 && i <= n && i != j, in_d = i >= lo && i <= n && i != j;
    num *= in_n ? tau[0] - tau[i] : 1.0;
    den *= in_d ? tj - tau[i] : 1.0;
  }
  if (!(set <= 3 && j >= 1 && j <= n)) { num = 0.0; den = 1.0; }
  // uniform picks of tau_{kk}, tau_{kk+1}, tau_{kk+2}
  double tk0 = tau[1], tk1 = tau[1], tk2 = tau[1];
#pragma unroll
  for (int i = 1; i < 9; ++i) { tk0 = (kk == i) ? tau[i] : tk0; tk1 = (kk + 1 == i) ? tau[i] : tk1; tk2 = (kk + 2 == i) ? tau[i] : tk2; }
  if (set == 4) {
    if (j >= 1 && j <= 5) { num = j <= kk ? 1.0 : 0.0; den = tau[0] - tj; }
    else if (j == 6) { num = lte ? hh : 0.0; den = tau[0] - tk1; }
    else if (j == 7) { num = nkm1 > 0 ? hh : 0.0; den = tau[0] - tk0; }
  } else if (set == 5 && j == 0) { num = nkp1 > 0 ? hh : 0.0; den = tau[0] - tk2; }
  if (num == 0.0) den = 1.0;   // never divide by a history time that is not there
  const double q = num / den;
  a0 = bcast(row_sum<16>((set == 4 && j >= 1 && j <= 5) ? q : 0.0), 32);
  return q;
}

// LDS per workgroup: [consts: cd doubles | ci ints] [part PW*8 | summ 8 | pair flags 4 | reduction scratch P_SCR] [PW wave regions]
// wave region (doubles): st[ndev*41] | A[nc*(nc+1)] | Cm[nc*nc] | xl xp F Q hq w qn pm pp x0 dm [11*nc] | Xh[8*nc] | Qh[8*nc] | tsl[8] | coef[48]
//                        | kvl[nk] svl[nsrc] | pl[max_mc*B4L_STRIDE] | ints: class blob, MOS class list
// PAIR: the two waves of a pair (2q, 2q+1) share the device evaluation of their two blocks BY FUNCTION (eval_cached): wave
// 2q evaluates the current half (I, G) of every MOSFET of both blocks plus the linear devices, wave 2q+1 the charge half (Q, C);
// each wave then gathers, factors and updates its OWN block.  The device evaluation — 45 of the 100 thousand cycles of an
// attempt — shrinks to its longer half, with one wave per SIMD as before.  The pair meets twice per Newton iteration through
// two counters in LDS (in-order LDS pipeline: a wave's stamp writes precede its counter write).  Requires one block class and
// at most 32 evaluation slots per block.
// MODE: 0 = one step sequence for the whole grid (grid-wide reduction per attempt); 1 = every sample / block its own steps
// (PersistArgs::indep, optionally per-workgroup constants); 2 = the bordered block-diagonal form (PersistArgs::nb > 0).  Separate
// instantiations, so that the lock-step kernel of the headline workload carries none of the others' state (the shared-code
// versions cost the 1024-DFF array 3 - 8 %).
enum { PM_LOCKSTEP = 0, PM_OWN = 1, PM_BORDER = 2 };
// WIDE: the circuit holds compiled Verilog-A devices (144-double stamp records, ch_kernels.hpp eval_slot<true>: one lane per unknown
// terminal and half).  With PAIR the two waves of a pair split every large compiled device of their two blocks by FUNCTION like the
// BSIM4 halves: wave 2q the resistive halves (and every unsplit slot), wave 2q+1 the charge halves.
template <int NC, bool PAIR, int MODE = PM_LOCKSTEP, bool WIDE = false>
__global__ __launch_bounds__(PW * 64, 1) void tran_persistent_kernel(const PersistArgs p) {
  typedef StampLayout<WIDE> SL;
  extern __shared__ double lds[];
  __shared__ int s_abort;
  __shared__ double s_bdec[4];   // bordered form: the border step and residual every wave of the workgroup decides from
  const NewtonArgs& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  const long long cyc0 = wall_clock64();
  long long cyc_bar = 0;
#ifdef CH_STAMPS
  unsigned long long pacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, plast_ = __builtin_readcyclecounter();
#endif
  // ---- workgroup-shared constants ----
  double* cdl = lds;
  int* cil = (int*)(cdl + p.n_cd);
  double* part = cdl + p.n_cd + ((p.n_ci + 1) >> 1);
  double* summ = part + PW * P_NREC;
  int* pfl = (int*)(summ + P_NREC);   // per pair: {sequence of wave 2q, of wave 2q+1, done flag of block 2q, of block 2q+1}
  double* W = summ + P_NREC + 4 + P_SCR + (size_t)wave * p.wave_doubles;
  constexpr bool own = MODE == PM_OWN;
  const int* const wgc = own ? p.wgc : nullptr;
  const double* bps_own = nullptr; int nbp_own = 0;   // per-workgroup break points (the other modes read p.bps / p.nbp where they need them)
  if (own && wgc) {
    const int* e = wgc + 6 * wg;
    const int* gci = p.ci + e[0]; const double* gcd = p.cd + e[1]; const int n_ci = e[2], n_cd = e[3];
    bps_own = p.bps + e[4]; nbp_own = e[5];
    for (int i = tid; i < n_cd; i += PW * 64) cdl[i] = gcd[i];
    for (int i = tid; i < n_ci; i += PW * 64) cil[i] = gci[i];
  } else {
    for (int i = tid; i < p.n_cd; i += PW * 64) cdl[i] = p.cd[i];
    for (int i = tid; i < p.n_ci; i += PW * 64) cil[i] = p.ci[i];
  }
#define P_BPS ((own && wgc) ? bps_own : p.bps)
#define P_NBP ((own && wgc) ? nbp_own : p.nbp)
  if (tid == 0) s_abort = 0;
  if (tid < 8) pfl[tid] = 0;
  __syncthreads();
  const PConst C{cil, cdl};

  const int blk = wg * p.bpw + wave;
  const bool live = wave < p.bpw && blk < p.nblk;
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
  double* tsl = Qh + 8 * nc; double* coef = tsl + 8;
  long long* stl = (long long*)(coef + 40);   // statistics counters (lane 0 updates them): they would only crowd the scalar registers
  double* kvl = coef + 48; double* svl = kvl + (wgc ? C.ci[3] : a.nk);     // a workgroup blob numbers its own entries
  double* pl = wgc ? kvl + P_MAXSRC : svl + a.nsrc;
  int* mptr = (int*)(pl + (size_t)a.max_mc * B4L_STRIDE);
  int* slots = mptr + (nc * nc + 1) + (nc + 1);
  uint16_t* msrc = (uint16_t*)(slots + cm.nslots);
  const int2* wl = (const int2*)(mptr + cm.wl_ofs);
  const long sofs = (long)s * a.n_unk + uofs;
  const bool mine = live && lane < nc;
  // paired evaluation: lanes 0..31 work on the pair's first block, lanes 32..63 on its second
  const int role = wave & 1, pairq = wave >> 1, half = lane >> 5;
  int* pf = pfl + pairq * 4;
  const int blk_h = wg * p.bpw + (wave & ~1) + half;
  const bool live_h = PAIR && (wave & ~1) + half < p.bpw && blk_h < p.nblk;
  const int bqh = live_h ? blk_h : 0, c_h = bqh / a.S, s_h = bqh - c_h * a.S;
  const int uofs_h = PAIR ? a.bmeta[c_h].uofs : uofs, dofs_h = PAIR ? a.bmeta[c_h].dofs : dofs;
  double* Wh = summ + P_NREC + 4 + P_SCR + (size_t)((wave & ~1) + half) * p.wave_doubles;   // region of this lane's block (same class, same layout)
  double* st_h = Wh; double* xl_h = Wh + (xl - W); double* pl_h = Wh + (pl - W);
  // the slot table too comes from the region of the lane's block: a wave without a block of its own (odd block count, single
  // circuit) has staged nothing into its own region and only helps its partner
  const int* slots_h = (const int*)((const char*)slots + ((const char*)Wh - (const char*)W));
  int pseq = 0; bool pair_broken = false;
  auto pair_sync = [&]() {   // both waves of the pair arrive; bounded like every other wait of this kernel
    ++pseq;
    lds_fence();
    if (lane == 0) __hip_atomic_store(pf + role, pseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const long long t0w = wall_clock64();
    while (__hip_atomic_load(pf + (1 - role), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < pseq) {
      if (wall_clock64() - t0w > p.spin_ticks) { pair_broken = true; __hip_atomic_store(p.counters + 9 * 32, 3u | ((unsigned)pseq << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }   // site 3
      __builtin_amdgcn_s_sleep(0);
    }
    lds_fence();
  };

  // ---- once per transient: class blob, BSIM4 columns, flags, history ring; A and C are zeroed once (the register LU never
  //      writes them back, the gather rewrites every structural non-zero each iteration) ----
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
    for (int i = lane; i < nc * lda + nc * nc; i += 64) A[i] = 0.0;
    if (mine) {
      dml[lane] = a.dmask[uofs + lane] | ((a.unk_obs[uofs + lane] + 1) << 8);
#pragma unroll
      for (int j = 0; j < 7; ++j) {   // global slot j holds the j-th newest point (canonical order); with head = 0 that is ring slot (-j) & 7
        const bool have = j < p.ctl->nhist;   // slots beyond the history are left over from earlier solves
        Xh[((8 - j) & 7) * nc + lane] = have ? a.X[(long)j * a.slot_stride + sofs + lane] : 0.0;
        Qh[((8 - j) & 7) * nc + lane] = have ? a.Qh[(long)j * a.slot_stride + sofs + lane] : 0.0;
      }
      Xh[1 * nc + lane] = 0.0; Qh[1 * nc + lane] = 0.0;   // ring slot 1 = first candidate
    }
  }
  if (lane < 8) tsl[(8 - lane) & 7] = p.ctl->tslot[lane];   // canonical (newest first) -> ring slots, head = 0
  // ---- controller state (identical in every wave) ----
  double t, h; int k, nhist, steps_at_order, ibp, isave, status = CH_OK, exit_reason = PX_RUNNING;
  bool reset_rate;
  int nsaved, step;
  enum { ST_ACC = 0, ST_REJ, ST_FAIL, ST_ITERS, ST_BITERS, ST_ATT };
  int head = 0;   // ring slot of the newest point; slot (head - j) & 7 holds the j-th newest, (head + 1) & 7 the candidate
  {
    const TranCtl* cs = p.ctl;
    t = cs->t; h = cs->h; k = cs->k; nhist = cs->nhist; steps_at_order = cs->steps_at_order; reset_rate = cs->reset_rate != 0;
    ibp = cs->ibp; isave = cs->isave; nsaved = (int)cs->nsaved; step = (int)cs->step;
    if (lane == 0) { stl[ST_ACC] = cs->naccept; stl[ST_REJ] = cs->nreject; stl[ST_FAIL] = cs->nconvfail; stl[ST_ITERS] = cs->sum_iters; stl[ST_BITERS] = cs->sum_block_iters; stl[ST_ATT] = cs->n_attempts; }
  }
  double rate_prev = 1.0;
  // pivot order of the block's register LU (lu_solve_block): lane i loads row myrow; kept for the whole transient, and across
  // launches / step controllers through a.perm (row ^ lane per unknown, 0 = identity)
  // A transient starts from the identity (the pivot order is then a function of the block's own history alone: identical blocks —
  // the tiles of an array, equal samples of a batch — stay bit-identical whatever the operating-point search did to each of them);
  // a relaunch of the same transient (drained row buffer) continues with the order it had.
  int myrow = lane;
  if (MODE != PM_BORDER && a.perm && mine && p.resume) myrow = lane ^ (int)a.perm[(long)blk * 16 + lane];
  lds_fence();
  __syncthreads();

  const int s_e = PAIR ? s_h : s;   // the sample of the block this lane evaluates for
  const EvalCtx ectx{a.dkind, a.dterm, a.dsrc, a.dcls_local, a.dhdev, a.dpar, a.dmult, a.Spar, a.gmin_s[a.Sgmin > 1 ? s_e : 0],
                     WIDE ? a.vapar + (long)s_e * a.va_stride : a.vapar, WIDE ? a.temp_s[a.Stemp > 1 ? s_e : 0] + 273.15 : 300.15,
                     WIDE ? a.vacache + (long)s_e * a.vac_stride : nullptr, WIDE ? a.dvac : nullptr};
  // this lane's device for the whole transient (block `half` of the pair when PAIR, the wave's own block otherwise)
  const double gmin_h = a.gmin_s[a.Sgmin > 1 ? (PAIR ? s_h : s) : 0];
  SlotMeta smeta;
  int wslot = -1;   // WIDE: this lane's evaluation slot (eval_slot<true> reads the device tables itself)
  if (WIDE) {
    smeta = load_slot_meta(ectx, 0, 0, 0, -1);   // an idle narrow slot
    if (PAIR) {
      // slot list of a class with split devices: [first halves (wide_l) | idle up to 64 | second halves (wide_l) | unsplit slots]
      const int sq = lane & 31;
      if (live_h) {
        if (sq < p.wide_l) wslot = slots_h[role == 0 ? sq : 64 + sq];
        else if (role == 0 && sq - p.wide_l < p.wide_other) wslot = slots_h[64 + p.wide_l + (sq - p.wide_l)];
      }
    } else if (live && lane < cm.nslots) wslot = slots[lane];
  }
  // WIDE: the lane's compiled device resolved once, and the workgroup's distinct parameter / constant blocks copied to LDS (first
  // come first served while the arena lasts; a block that does not fit stays in global memory).  One wave after the other, so that
  // the directory needs no atomics; wave-uniform control flow throughout.
  WideMeta wmeta = load_wide_meta(ectx, PAIR ? s_h : s, PAIR ? dofs_h : dofs, WIDE ? wslot : -1);
  bool va_lds = false;   // workgroup-uniform: every compiled device of this workgroup reads its blocks from LDS (stamp_dir_lds)
  if (WIDE && p.va_arena > 0) {
    __shared__ unsigned long long s_va_ptr[32];
    __shared__ int s_va_ofs[32];
    __shared__ int s_va_n, s_va_top, s_va_fail;
    double* arena = summ + P_NREC + 4 + P_SCR + (size_t)PW * p.wave_doubles;
    if (tid == 0) { s_va_n = 0; s_va_top = 0; s_va_fail = 0; }
    __syncthreads();
    const double* staged[2] = {nullptr, nullptr};
    for (int w = 0; w < PW; ++w) {
      if (wave == w) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const double* g = which ? wmeta.C : wmeta.P;
          const int n = wmeta.mod < 0 ? 0 : (which ? va_gen::cache_doubles(wmeta.mod) : va_gen::param_doubles(wmeta.mod));
          bool pend = n > 0;
          for (;;) {
            const unsigned long long bal = __ballot(pend);
            if (!bal) break;
            const int srcl = __ffsll((long long)bal) - 1;
            const unsigned long long gp = ((unsigned long long)(unsigned)__shfl((int)((unsigned long long)g >> 32), srcl) << 32) | (unsigned)__shfl((int)(unsigned long long)g, srcl);
            const int nn = __shfl(n, srcl);
            const int ndir = s_va_n, top = s_va_top;
            int ofs = -1;
            for (int e = 0; e < ndir; ++e) if (s_va_ptr[e] == gp) ofs = s_va_ofs[e];
            if (ofs < 0 && ndir < 32 && top + nn <= p.va_arena) {
              ofs = top;
              const double* gsrc = (const double*)gp;
              for (int i = lane; i < nn; i += 64) arena[top + i] = gsrc[i];
              lds_fence();
              if (lane == 0) { s_va_ptr[ndir] = gp; s_va_ofs[ndir] = top; s_va_n = ndir + 1; s_va_top = top + ((nn + 1) & ~1); }
              lds_fence();
            }
            if (ofs < 0 && lane == 0) s_va_fail = 1;
            if (pend && (unsigned long long)g == gp) { staged[which] = ofs >= 0 ? arena + ofs : nullptr; pend = false; }
          }
        }
      }
      __syncthreads();
    }
    va_lds = s_va_fail == 0;
    if (va_lds && wmeta.mod >= 0) { wmeta.P = staged[0] ? staged[0] : arena; wmeta.C = staged[1] ? staged[1] : arena; }   // a block of zero doubles is never read
  }
  if (!WIDE)
  {
    int slot = -1;
    if (PAIR) { const int sq = lane & 31; if (live_h && sq < cm.nslots) slot = slots_h[sq]; }
    else if (live && lane < cm.nslots) slot = slots[lane];
    smeta = load_slot_meta(ectx, PAIR ? s_h : s, PAIR ? dofs_h : dofs, PAIR ? uofs_h : uofs, slot);
    if (wgc && smeta.kind != 0) {   // known nodes and device sources by the workgroup's own entry numbers
      const int* wk = p.wgk + (size_t)wg * P_MAXSRC;
      const int nkl = C.ci[3], nel = C.n_ent();
#pragma unroll
      for (int k = 0; k < 4; ++k) if (smeta.t[k] < 0) {
        const int g = -smeta.t[k] - 1; int j = 0;
        while (j < nkl - 1 && wk[j] != g) ++j;     // once per transient: at most 64 entries
        smeta.t[k] = -(j + 1);
      }
      if (smeta.kind == K_I || smeta.kind == K_V) { int j = nkl; while (j < nel - 1 && wk[j] != smeta.src) ++j; smeta.src = j - nkl; }
    }
  }
  const int n_ent = C.n_ent();
  // this lane's entry (a known-node or device-source value): its terms, and the linear piece of up to two of them
  const int e_p0 = lane < n_ent ? C.ent_ptr()[lane] : 0, e_nt = lane < n_ent ? C.ent_ptr()[lane + 1] - e_p0 : 0;
  const int e_i0 = e_nt > 0 ? C.ent_idx()[e_p0] : 0, e_i1 = e_nt > 1 ? C.ent_idx()[e_p0 + 1] : 0;
  const double e_c0 = e_nt > 0 ? C.ent_coef()[e_p0] : 0.0, e_c1 = e_nt > 1 ? C.ent_coef()[e_p0 + 1] : 0.0;
  double s0_lo = __builtin_inf(), s0_hi = -__builtin_inf(), s0_y = 0.0, s0_m = 0.0, s1_lo = __builtin_inf(), s1_hi = -__builtin_inf(), s1_y = 0.0, s1_m = 0.0;
  double bp_next = 0.0; int bp_at = -1;        // cached p.bps[ibp]
  double bp_code = -1.0;                       // ... and its code: < 0 the sources JUMP there (restart at order 1); >= 0 a continuous corner, the
                                               //     length of the source segment that starts there
  double sv_next = 0.0; int sv_at = -1;        // cached p.saveat[isave]
  const long long row_stride = (long long)p.n_obs * a.S;
  const int my_ob = mine ? (dml[lane] >> 8) - 1 : -1;
  unsigned gen = 0;

  // rows due at t0 (the initial state): the host leaves them to the kernel only when it starts fresh
  if (!p.resume) {
    if (p.n_saveat == 0) {
      if (my_ob >= 0) p.out_rows[(long long)nsaved * row_stride + (long long)my_ob * a.S + s] = Xh[lane];
      if (blk == 0 && lane == 0) p.out_times[nsaved] = t;
      ++nsaved;
    } else {
      while (isave < p.n_saveat && p.saveat[isave] <= t) {
        if (my_ob >= 0) p.out_rows[(long long)nsaved * row_stride + (long long)my_ob * a.S + s] = Xh[lane];
        if (blk == 0 && lane == 0) p.out_times[nsaved] = p.saveat[isave];
        ++nsaved; ++isave;
      }
    }
  }

  P_STAMP(0);   // set-up
  while (step < p.max_steps && t < p.t1) {
    if (p.n_saveat == 0 && nsaved >= p.max_rows) { exit_reason = PX_ROWS_FULL; break; }
    // The lane id of the controller code is made opaque once per attempt: otherwise every lane predicate of the select chains
    // below (lane == j, set == q, ...) is hoisted out of the time loop as a 64-bit mask in scalar registers — dozens of
    // pairs that spill, each use then costing two v_readlane instead of the one v_cmp that recomputes it.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    // next break point: global loads only when the index moves
    for (;;) {
      if (bp_at != ibp) { bp_next = ibp < P_NBP ? P_BPS[ibp] : p.t1; bp_code = ibp < P_NBP ? P_BPS[P_NBP + ibp] : -1.0; bp_at = ibp; }
      if (ibp < P_NBP && bp_next <= t * (1 + 1e-15) + 1e-300) ++ibp; else break;
    }
    const double tb = bp_next;
    bool hit_bp = false;
    double tn = t + h;
    if (tn >= tb - 1e-3 * h) { tn = tb; hit_bp = true; }
    const double hh = tn - t;
    const bool dcm = MODE == PM_BORDER && p.dc_mode != 0;
    if (!dcm && hh < p.dtmin) { status = CH_ERR_DTMIN; break; }
    const int nh = nhist, kk = k < nh ? k : nh, np = (kk + 1) < nh ? (kk + 1) : nh;
    const bool lte = np >= kk + 1;
    const bool try_up = lte && kk < p.kmax && nh >= kk + 2 && steps_at_order + 1 >= kk + 1;
    const int nkm1 = (lte && kk > 1) ? kk : 0, nkp1 = try_up ? kk + 2 : 0;
    double alpha0;
    const double cq = p_coef(tn, tsl, head, kk, np, nkm1, nkp1, lte, hh, ln, alpha0);
    const double ck = bcast(cq, 38), ckm1 = bcast(cq, 39), ckp1 = bcast(cq, 40);
    if (dcm) alpha0 = 0.0;   // operating point: no time derivative
    if (ln < 32) coef[ln] = cq;   // the 26 weights go through LDS: as scalars they would take 52 SGPRs and spill
    lds_fence();
    P_STAMP(1);   // break points, BDF / predictor coefficients
    // entries [known-node values | device source values] at t_new (the sources' left limit when the step lands on a break point)
    {
      const double ts = (hit_bp && tn > 0.0) ? __longlong_as_double(__double_as_longlong(tn) - 1) : (hit_bp ? nextafter(tn, -__builtin_inf()) : tn);
      if (dcm) { if (lane < n_ent) kvl[lane] = p.dc_entries[lane]; }
      else if (lane < n_ent) {
        double v = 0.0;
        if (e_nt <= 2) {   // the usual case: one or two piecewise-linear sources, their current pieces cached in registers
          if (e_nt > 0) {
            if (!(s0_lo <= ts && ts < s0_hi)) p_source_piece(C, e_i0, ts, s0_lo, s0_hi, s0_y, s0_m);
            const double u = (s0_lo <= ts && ts < s0_hi) ? ((s0_m == 0.0) ? s0_y : s0_y + (ts - s0_lo) * s0_m) : p_source(C, e_i0, ts);
            v += e_c0 * u;
          }
          if (e_nt > 1) {
            if (!(s1_lo <= ts && ts < s1_hi)) p_source_piece(C, e_i1, ts, s1_lo, s1_hi, s1_y, s1_m);
            const double u = (s1_lo <= ts && ts < s1_hi) ? ((s1_m == 0.0) ? s1_y : s1_y + (ts - s1_lo) * s1_m) : p_source(C, e_i1, ts);
            v += e_c1 * u;
          }
        } else for (int q = e_p0; q < e_p0 + e_nt; ++q) v += C.ent_coef()[q] * p_source(C, C.ent_idx()[q], ts);
        kvl[lane] = v;   // kvl and svl are contiguous
      }
    }
    P_STAMP(2);   // source waveforms, known-node values
    // ---- predictor and history term from the ring (weights straight from the lanes that formed them) ----
    const int cand = (head + 1) & 7;
    double x0 = 0.0;
    {
      double pr = 0.0, hs = 0.0, m1 = 0.0, p1 = 0.0;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const int sl = (head - j) & 7;
        const double xv = (mine && j < nh) ? Xh[sl * nc + lane] : 0.0;   // points beyond the history in use may hold anything (0 * NaN)
        if (j == 0) x0 = xv;
        pr = fma(coef[8 + j + 1], xv, pr);
        m1 = fma(coef[16 + j + 1], xv, m1);
        p1 = fma(coef[24 + j + 1], xv, p1);
        if (j < 5) hs = fma(coef[j + 1], (mine && j < kk) ? Qh[sl * nc + lane] : 0.0, hs);
      }
      if (mine) {
        xp[lane] = pr; xl[lane] = pr; hq[lane] = hs; qn[lane] = 0.0; pm[lane] = m1; pp[lane] = p1;
        wv[lane] = 1.0 / (a.reltol * fabs(x0) + a.abstol);
      }
    }
    if (dcm && mine) { xp[lane] = x0; xl[lane] = x0; hq[lane] = 0.0; pm[lane] = x0; pp[lane] = x0; }   // the iterate starts at the state in the ring
    lds_fence();
    P_STAMP(3);   // predictor

    // ---- Newton iteration (the register-LU branch of newton_block_kernel, one wave, no workgroup barriers) ----
    int nstat = 1, iters = 0;
    double rate_new = -1.0, dn_prev = 0.0;
    const double rp = reset_rate ? 1.0 : rate_prev;
    bool done_own = !live;
    constexpr bool bbd = MODE == PM_BORDER;
    const int no = nc - p.nb;                            // the block's own unknowns; lanes no .. nc-1 hold its border replicas
    bool stop_all = false, abort_all = false;            // bordered form: grid-uniform decisions
    bool dc_bad = false;                                 // operating point: this block's last update left a non-finite value
    double bA = 0.0, bB = 0.0, bC = 0.0, bN = 0.0;       // bordered form: the local-error sums of the last iteration's reduction
    if (live || PAIR || bbd) {
      for (int it = 0; PAIR || bbd || it <= a.maxit; ++it) {
        if (bbd && stop_all) break;
        if (PAIR) {
          if (lane == 0) pf[2 + role] = done_own ? 1 : 0;
          pair_sync();                                   // both iterates (or their done flags) are in LDS
          const int d0 = pf[2], d1 = pf[3];
          if (!bbd && ((d0 && d1) || pair_broken)) break;
          if (!(half ? d1 : d0)) {
            if (WIDE) {
              if (wmeta.mod >= 0) { if (va_lds) eval_wide_cached<true>(wmeta, ectx, wslot, xl_h, uofs_h, kvl, st_h); else eval_wide_cached<false>(wmeta, ectx, wslot, xl_h, uofs_h, kvl, st_h); }
              else if (wslot >= 0) eval_slot<true>(ectx, s_h, dofs_h, wslot, xl_h, uofs_h, kvl, svl, pl_h, st_h);
            }
            else {
            if (role == 0) eval_cached<0>(smeta, gmin_h, xl_h, kvl, svl, pl_h, st_h);
            if (role == 1 ? (p.pair_dbg & 1) == 0 : (p.pair_dbg & 1) == 1) eval_cached<1>(smeta, gmin_h, xl_h, kvl, svl, pl_h, st_h);
            }
          }
          pair_sync();                                   // both halves of every stamp record are in LDS
          P_STAMP(4);   // device evaluation (one half of it)
          if (!bbd) {
            if (pair_broken) break;
            if (done_own) continue;
          }
        } else {
        if (WIDE) {
          if (wmeta.mod >= 0) { if (va_lds) eval_wide_cached<true>(wmeta, ectx, wslot, xl, uofs, kvl, st); else eval_wide_cached<false>(wmeta, ectx, wslot, xl, uofs, kvl, st); }
          else if (wslot >= 0) eval_slot<true>(ectx, s, dofs, wslot, xl, uofs, kvl, svl, pl, st);
        }
        else if (!bbd || live) eval_cached<-1>(smeta, gmin_h, xl, kvl, svl, pl, st);
        lds_fence();
        P_STAMP(4);   // device evaluation
        }
        for (int w = lane; w < ((bbd && !live) ? 0 : cm.n_work); w += 64) {
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
        lds_fence();
        P_STAMP(5);   // gather
        constexpr int NCR = NC;
        double r[NCR + 1], cr[NCR];
        if constexpr (MODE == PM_BORDER) {   // (the other forms load their rows in pivot order inside lu_solve_block)
#pragma unroll
          for (int j = 0; j <= NCR; ++j) r[j] = (mine && j <= nc) ? A[lane * lda + j] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < NCR; ++j) cr[j] = (mine && j < nc) ? Cm[lane * nc + j] : 0.0;
        const double Fi = mine ? Fv[lane] : 0.0, Qi = mine ? Qv[lane] : 0.0, xi = mine ? xl[lane] : 0.0, wi = mine ? wv[lane] : 0.0;
        const double fnorm = bcast(row_max<NCR>(fabs(Fi)), 0);
        if (bbd) {
          // ---- (a) partial LU of the block, this block's Schur contribution, grid-wide sum ----
          bool lfail = live && (!(fnorm == fnorm) || fnorm > 1e300 || dc_bad);
          // operating point: does any OWN row of this block still violate the residual tolerance?
          const double fown = bcast(row_max<NCR>((mine && lane < no) ? fabs(Fi) : 0.0), 0);
          int piv[NCR]; int mystep; double ipiv;
          if (!lu_bbd_factor<NCR>(r, no, nc, lane, piv, mystep, ipiv)) lfail = live;
          double srec[P_NREC];
          {
            double c0 = 0.0, c1 = 0.0, cr_ = 0.0;   // this lane's entries in the border columns and the right-hand side
#pragma unroll
            for (int j = 0; j <= NCR; ++j) { if (j == no) c0 = r[j]; if (j == no + 1 && j < nc) c1 = r[j]; if (j == nc) cr_ = r[j]; }
            const bool use = live && !lfail;
            srec[0] = use ? bcast(c0, no) : 0.0; srec[1] = use ? bcast(c1, no) : 0.0; srec[2] = use ? bcast(cr_, no) : 0.0;
            const int l1 = p.nb > 1 ? no + 1 : no;
            srec[3] = (use && p.nb > 1) ? bcast(c0, l1) : 0.0; srec[4] = (use && p.nb > 1) ? bcast(c1, l1) : 0.0; srec[5] = (use && p.nb > 1) ? bcast(cr_, l1) : 0.0;
            srec[6] = (lfail ? 1.0 : 0.0) + (pair_broken ? 1048576.0 : 0.0);   // failures; a broken pair wait counts 2^20 (abort)
            srec[7] = (dcm && live && !(fown < p.dc_abstol)) ? 1.0 : 0.0;        // operating point: blocks whose own rows have not converged
          }
          ++gen;
          const long long cbA = wall_clock64();
          bool okr = p_grid_reduce<true>(p, gen, srec, part, summ, &s_abort, wave, lane, wg);
          double S00 = summ[0], S01 = summ[1], g0 = summ[2], S10 = summ[3], S11 = summ[4], g1 = summ[5];
          const bool gfail = summ[6] != 0.0;
          const double n_viol = summ[7];
          if (!okr || summ[6] >= 1048576.0) abort_all = true;
          __syncthreads();   // summ is rewritten by the next reduction
          cyc_bar += wall_clock64() - cbA;
          if (abort_all) break;
          if (gfail) { nstat = 2; break; }
          // ---- (b) the devices that sit on the border alone (rail resistance, decoupling capacitors): every wave adds them ----
          {
            const double xb0 = xl[no], xb1 = p.nb > 1 ? xl[no + 1] : 0.0;
            // history term of the border charges: sum_j alpha_j * C (xb_a - xb_b)(t_j) from the ring of the replicas (linear elements)
            double hb0 = 0.0, hb1 = 0.0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
              const int sl = (head - j) & 7;
              const double w = (j < kk) ? coef[j + 1] : 0.0;
              hb0 = fma(w, (j < kk) ? Xh[sl * nc + no] : 0.0, hb0);
              if (p.nb > 1) hb1 = fma(w, (j < kk) ? Xh[sl * nc + no + 1] : 0.0, hb1);
            }
            if (dcm) { hb0 = 0.0; hb1 = 0.0; }   // operating point: capacitors carry no current
            for (int q = 0; q < p.n_bdev; ++q) {
              const int ta = p.bd_ta[q], tb = p.bd_tb[q];
              const double va = ta >= 0 ? (ta == 0 ? xb0 : xb1) : kvl[-ta - 1], vb = tb >= 0 ? (tb == 0 ? xb0 : xb1) : kvl[-tb - 1];
              double gj, f;   // Jacobian entry and residual current of the element, a -> b
              if (p.bd_kind[q] == K_R) { gj = p.bd_val[q]; f = gj * (va - vb); }
              else {            // capacitor (terminals: border or ground): alpha0 q + history
                const double ha = ta >= 0 ? (ta == 0 ? hb0 : hb1) : 0.0, hb = tb >= 0 ? (tb == 0 ? hb0 : hb1) : 0.0;
                gj = alpha0 * p.bd_val[q]; f = gj * (va - vb) + p.bd_val[q] * (ha - hb);
              }
              // rows: J dx = -F
              if (ta == 0) { S00 += gj; g0 -= f; if (tb == 1) S01 -= gj; }
              if (ta == 1) { S11 += gj; g1 -= f; if (tb == 0) S10 -= gj; }
              if (tb == 0) { S00 += gj; g0 += f; if (ta == 1) S01 -= gj; }
              if (tb == 1) { S11 += gj; g1 += f; if (ta == 0) S10 -= gj; }
            }
          }
          // ---- (c) the border system ----
          double dxb0, dxb1 = 0.0;
          if (p.nb > 1) {
            const double det = S00 * S11 - S01 * S10;
            dxb0 = (g0 * S11 - S01 * g1) / det; dxb1 = (S00 * g1 - S10 * g0) / det;
          } else dxb0 = g0 / S00;
          // Every wave of the grid must take the same decisions from here on.  Waves that own a block hold identical replicas of the
          // border, so they computed identical numbers; a wave WITHOUT a block (block count not a multiple of 4, or the idle half of
          // a pair) has no replicas — its LDS region holds whatever an earlier kernel left — and would decide from garbage, leave
          // the loop at another iteration and dead-lock the reductions.  Wave 0 of every workgroup always owns a block: its
          // numbers are the workgroup's.  (The write of the next iteration is separated from these reads by the barriers of the
          // reductions in between.)
          if (wave == 0 && lane == 0) { s_bdec[0] = dxb0; s_bdec[1] = dxb1; s_bdec[2] = g0; s_bdec[3] = g1; }
          __syncthreads();
          dxb0 = s_bdec[0]; dxb1 = s_bdec[1]; g0 = s_bdec[2]; g1 = s_bdec[3];
          if (!(fabs(dxb0) < 1e300) || !(fabs(dxb1) < 1e300)) { nstat = 2; break; }   // singular border (uniform: every wave holds the same numbers)
          if (dcm) {
            // ---- operating point: converged when no block reports a violating own row and the reduced border residual is
            //      below the tolerance (with the own rows converged, g is the border's residual up to O(tolerance)); otherwise a
            //      damped Newton step (CedarDCOp's voltage limiting) ----
            if (n_viol == 0.0 && fmax(fabs(g0), fabs(g1)) < p.dc_abstol) { nstat = 0; stop_all = true; continue; }
            // the full Newton step of every block, then ONE scale for the whole system from the largest node-voltage change
            // (a second reduction, field 4 = maximum): the same damped iteration as the sparse path's, so the same operating
            // point of a multi-stable circuit is reached from the same start
            double dx = 0.0, mo = 0.0;
            if (live) {
              dx = lu_bbd_back<NCR>(r, no, nc, lane, piv, mystep, ipiv, dxb0, dxb1);
              mo = bcast(row_max<NCR>((mine && lane < no && !(dml[lane] & 2)) ? fabs(dx) : 0.0), 0);
              if (__ballot(mine && !(dx == dx))) dc_bad = true;
            }
            double brec[P_NREC] = {0.0, 0.0, 0.0, 0.0, mo, 0.0, 0.0, 0.0};
            ++gen;
            const long long cbD = wall_clock64();
            const bool okd = p_grid_reduce(p, gen, brec, part, summ, &s_abort, wave, lane, wg);
            const double gmax = fmax(summ[4], fmax(fabs(dxb0), fabs(dxb1)));
            __syncthreads();
            cyc_bar += wall_clock64() - cbD;
            if (!okd) { abort_all = true; break; }
            const double sc = (p.dv_max > 0.0 && gmax > p.dv_max) ? p.dv_max / gmax : 1.0;
            if (live) {
              const double xn = xi + sc * dx;
              if (mine) xl[lane] = xn;
              if (__ballot(mine && (!(xn == xn) || fabs(xn) > 1e300))) dc_bad = true;
            }
            ++iters;
            if (it + 1 >= p.dc_maxit) stop_all = true;   // not converged: nstat stays 1
            lds_fence();
            continue;
          }
          // ---- (d) back substitution, update, charges, norms ----
          double e2own = 0.0, l2k = 0.0, l2m = 0.0, l2p = 0.0, lnd = 0.0, lbad = 0.0;
          if (live) {
            const double dx = lu_bbd_back<NCR>(r, no, nc, lane, piv, mystep, ipiv, dxb0, dxb1);
            const double xn = xi + dx;
            if (mine) xl[lane] = xn;
            const bool bad = mine && (!(xn == xn) || fabs(xn) > 1e300);
            double q = Qi;
#pragma unroll
            for (int j = 0; j < NCR; ++j) if (j < nc) q = fma(cr[j], bcast(dx, j), q);
            if (mine) qn[lane] = q;
            const bool counted = mine && !(dml[lane] & 4);   // own unknowns, and block 0's replicas of the border
            const double tq = dx * wi;
            e2own = bcast(row_sum<NCR>(counted ? tq * tq : 0.0), 0);
            lbad = __ballot(bad) ? 1.0 : 0.0;
            double v2k = 0.0, v2m = 0.0, v2p = 0.0, vnd = 0.0;
            if (counted && (dml[lane] & 1)) {
              const double w = 1.0 / (a.reltol * fmax(fabs(x0), fabs(xn)) + a.abstol);
              vnd = 1.0;
              double tq2 = (xn - xp[lane]) * w; v2k = tq2 * tq2;
              if (nkm1 > 0) { tq2 = (xn - pm[lane]) * w; v2m = tq2 * tq2; }
              if (nkp1 > 0) { tq2 = (xn - pp[lane]) * w; v2p = tq2 * tq2; }
            }
            l2k = bcast(row_sum<16>(v2k), 0); l2m = bcast(row_sum<16>(v2m), 0); l2p = bcast(row_sum<16>(v2p), 0); lnd = bcast(row_sum<16>(vnd), 0);
          }
          ++iters;
          double brec[P_NREC] = {l2k, l2m, l2p, lnd, e2own, lbad, 0.0, pair_broken ? 1.0 : 0.0};
          ++gen;
          const long long cbB = wall_clock64();
          okr = p_grid_reduce<true>(p, gen, brec, part, summ, &s_abort, wave, lane, wg);
          bA = summ[0]; bB = summ[1]; bC = summ[2]; bN = summ[3];
          const double e2g = summ[4]; const bool gbad = summ[5] != 0.0;
          if (!okr || summ[7] != 0.0) abort_all = true;
          __syncthreads();
          cyc_bar += wall_clock64() - cbB;
          if (abort_all) break;
          if (gbad) { nstat = 2; break; }
          const double dn = sqrt(e2g / (double)p.n_glob);
          if (it == 0) { if (dn <= a.newton_tol || (rp < 0.9 && 2.0 * fmax(rp, 0.02) * dn <= a.newton_tol)) { nstat = 0; stop_all = true; } }
          else { rate_new = dn_prev > 0.0 ? dn / dn_prev : 0.0; if (dn <= a.newton_tol) { nstat = 0; stop_all = true; } }
          dn_prev = dn;
          if (it + 1 >= a.maxit) stop_all = true;   // not converged within maxit: nstat stays 1
          lds_fence();
          continue;
        }
        bool stop = false;
        if (!(fnorm == fnorm) || fnorm > 1e300) { nstat = 2; stop = true; }
        else if (it == a.maxit) { stop = true; }
        else {
          double dx = 0.0;
          P_STAMP(6);   // row loads, residual norm
          const bool ok = lu_solve_block<NCR>(A, lda, nc, lane, myrow, dx);
          P_STAMP(7);   // LU + triangular solves
          if (!ok) { nstat = 2; stop = true; }
          else {
            const double xn = xi + dx;
            if (mine) xl[lane] = xn;
            const bool bad = mine && (!(xn == xn) || fabs(xn) > 1e300);
            const double tq = dx * wi;
            const double e2 = bcast(row_sum<NCR>(mine ? tq * tq : 0.0), 0);
            const double q = RowDot<NCR, 0>::run(cr, dx, nc, Qi);
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
        lds_fence();
        P_STAMP(8);   // update, charge, convergence test
        if (stop) { if (PAIR) done_own = true; else break; }
      }
    }
    if (dcm) {   // the operating point goes back to the ring slot it came from; the host reads the status
      if (abort_all) { exit_reason = PX_ABORT; status = CH_ERR_DEVICE; break; }
      if (mine) Xh[head * nc + lane] = xl[lane];
      if (lane == 0) { stl[ST_ITERS] += iters; stl[ST_BITERS] += (long long)iters * p.nblk; }
      lds_fence();
      status = nstat == 0 ? CH_OK : (nstat == 2 ? CH_ERR_SINGULAR : CH_ERR_MAXITERS);
      exit_reason = PX_DONE;
      break;
    }
    // ---- candidate into the ring, local-error sums (the block's unknowns sit in the first 16 lanes: DPP row sums) ----
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
    e2k = bcast(row_sum<16>(e2k), 0); e2m = bcast(row_sum<16>(e2m), 0); e2p = bcast(row_sum<16>(e2p), 0); ndf = bcast(row_sum<16>(ndf), 0);
    if ((live || bbd) && nstat == 0) rate_prev = iters >= 2 ? fmin(1.0, fmax(rate_new, 1e-4)) : fmin(1.0, rp * 1.5);   // bordered form: the same in every wave
    double rec[P_NREC];
    if (p.red_max) { const double inv = ndf > 0.0 ? 1.0 / ndf : 0.0; rec[0] = e2k * inv; rec[1] = e2m * inv; rec[2] = e2p * inv; }
    else { rec[0] = e2k; rec[1] = e2m; rec[2] = e2p; }
    rec[3] = ndf; rec[4] = (double)iters; rec[5] = (double)iters; rec[6] = (live && nstat != 0) ? 1.0 : 0.0; rec[7] = 0.0;
    if (!live) { for (int q = 0; q < P_NREC; ++q) rec[q] = 0.0; }
    if (pair_broken) rec[7] = 1.0;   // a pair wait ran into its bound: every wave of the grid leaves at this attempt
    if (!bbd) ++gen;   // (the bordered form has made its reductions inside the Newton loop: consecutive reductions must alternate parity)
    P_STAMP(9);   // candidate, local-error sums
    const long long cb0 = wall_clock64();
    double sA, sB, sC, sN, sItMax, sItSum, sFail;
    if (own) {
      // per-sample step acceptance: the record is combined over the wave pair only (the two samples that share a device
      // evaluation stay in lock-step), or not at all
      double r8[P_NREC];
#pragma unroll
      for (int q = 0; q < P_NREC; ++q) r8[q] = rec[q];
      if (PAIR) {
        if (lane < P_NREC) { double v = rec[0];
#pragma unroll
          for (int q = 1; q < P_NREC; ++q) v = (lane == q) ? rec[q] : v;
          part[wave * P_NREC + lane] = v; }
        pair_sync();
        const double* other = part + (wave ^ 1) * P_NREC;
#pragma unroll
        for (int q = 0; q < P_NREC; ++q) { const double o = other[q]; r8[q] = (q < 3 || q == 4) ? fmax(r8[q], o) : r8[q] + o; }
        if (pair_broken) r8[7] = 1.0;
      }
      sA = r8[0]; sB = r8[1]; sC = r8[2]; sN = r8[3]; sItMax = r8[4]; sItSum = (double)iters; sFail = r8[6];
      if (r8[7] != 0.0) { exit_reason = PX_ABORT; status = CH_ERR_DEVICE; break; }
    } else if (bbd) {
      // the last iteration's reduction already carried the local-error sums; every wave holds the same Newton outcome
      if (abort_all) { exit_reason = PX_ABORT; status = CH_ERR_DEVICE; break; }
      sA = bA; sB = bB; sC = bC; sN = bN; sItMax = (double)iters; sItSum = (double)iters * (double)p.nblk; sFail = nstat != 0 ? 1.0 : 0.0;
    } else {
      const bool okr = p_grid_reduce(p, gen, rec, part, summ, &s_abort, wave, lane, wg);
      if (!okr || summ[7] != 0.0) { exit_reason = PX_ABORT; status = CH_ERR_DEVICE; break; }
      sA = summ[0]; sB = summ[1]; sC = summ[2]; sN = summ[3]; sItMax = summ[4]; sItSum = summ[5]; sFail = summ[6];
      __syncthreads();   // summ is rewritten by the next reduction
    }
    cyc_bar += wall_clock64() - cb0;
    P_STAMP(10);  // grid reduction
    // error norms of orders k, k-1, k+1 in lanes 0, 1, 2 (one division and one square root for the three)
    double errk, errkm1, errkp1;
    double ev3;
    {
      const double sv = ln == 0 ? sA : ln == 1 ? sB : sC, cv = ln == 0 ? ck : ln == 1 ? ckm1 : ckp1;
      ev3 = p.red_max ? cv * sqrt(sv) : (sN > 0.0 ? cv * sqrt(sv / sN) : 0.0);
      errk = bcast(ev3, 0); errkm1 = bcast(ev3, 1); errkp1 = bcast(ev3, 2);
    }
    if (lane == 0) { stl[ST_ATT] += 1; stl[ST_BITERS] += (long long)sItSum; stl[ST_ITERS] += p.red_max ? (long long)sItSum : (long long)sItMax; }
    // ---- the step controller (same policy as ch_circuit::tran_solve) ----
    if (sFail > 0.0) {
      if (lane == 0) stl[ST_FAIL] += 1;
      reset_rate = true;
      h = hh * 0.25; k = 1; steps_at_order = 0;
      if (nhist > 2) nhist = 2;
      P_STAMP(11);
      continue;
    }
    if (!lte) errk = 0.0;
    // the three candidate factors (2 err + 1e-4)^(-1/(order+1)) are formed by three lanes at once
    double fk, fm, fp;
    {
      const double ev = (ln == 0 && !lte) ? 0.0 : ev3;
      const double ex = ln == 0 ? (double)(kk + 1) : ln == 1 ? (double)kk : (double)(kk + 2);
      const double fv = exp(-flog(2.0 * ev + 1e-4) / ex);
      fk = bcast(fv, 0); fm = bcast(fv, 1); fp = bcast(fv, 2);
    }
    if (errk > 1.0) {
      if (lane == 0) stl[ST_REJ] += 1;
      h = hh * fmin(0.9, fmax(0.25, 0.9 * fk));
      steps_at_order = 0;
      P_STAMP(11);
      continue;
    }
    // ---- accept ----
    if (lane == 0) stl[ST_ACC] += 1;
    ++step; reset_rate = false;
    head = cand;
    if (lane == 0) tsl[head] = tn;
    lds_fence();
    nhist = nhist + 1 < p.kmax + 2 ? nhist + 1 : p.kmax + 2;
    if (p.n_saveat > 0) {
      for (;;) {
        if (sv_at != isave) { sv_next = isave < p.n_saveat ? p.saveat[isave] : __builtin_inf(); sv_at = isave; }
        if (!(isave < p.n_saveat && sv_next <= tn * (1 + 1e-15))) break;
        const double tsv = sv_next;
        const int msv = (kk < nh ? kk : nh) + 1;
        // interpolation weights through the msv newest points (the accepted one included)
        double dummy;
        const double wq = p_coef(tsv, tsl, head, 0, msv, 0, 0, false, 0.0, ln, dummy);
        if (ln < 32) coef[ln] = wq;
        lds_fence();
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) v = fma(coef[8 + j + 1], (mine && j < msv) ? Xh[((head - j) & 7) * nc + lane] : 0.0, v);
        lds_fence();
        if (my_ob >= 0) p.out_rows[(long long)nsaved * row_stride + (long long)my_ob * a.S + s] = v;
        if (blk == 0 && lane == 0) p.out_times[nsaved] = tsv;
        ++nsaved; ++isave;
      }
    } else {
      if (my_ob >= 0) p.out_rows[(long long)nsaved * row_stride + (long long)my_ob * a.S + s] = Xh[head * nc + lane];
      if (blk == 0 && lane == 0) { p.out_times[nsaved] = tn; p.out_times[p.max_rows + nsaved] = (double)((kk < nh ? kk : nh) + 1); }   // + the points of the step's dense output
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
      // behind a JUMP of a source: restart at order 1 with a tiny first step; at a continuous corner (IDA's tstops) history and order
      // are kept and the first step is capped at a tenth of the source segment that starts there (same policy as ch_circuit::tran_solve)
      reset_rate = true;
      if (bp_code < 0) {
        nhist = 1; k = 1; steps_at_order = 0;
        double nb = p.t1;
        for (int b = ibp; b < P_NBP; ++b) if (P_BPS[b] > t * (1 + 1e-15)) { nb = P_BPS[b]; break; }
        h = fmax(p.dtmin * 10, fmin(h, (nb - t) / 50.0) * p.first_frac);
      } else h = fmax(p.dtmin * 10, fmin(h, bp_code * 0.1));
    }
    P_STAMP(11);  // controller, saved rows
  }
  if (exit_reason == PX_RUNNING) {
    exit_reason = PX_DONE;
    if (status == CH_OK && t < p.t1) status = CH_ERR_MAXSTEPS;
  }
  // ---- write the ring back in canonical order (slot j = j-th newest) and the controller state ----
  if (MODE != PM_BORDER && a.perm && mine) a.perm[(long)blk * 16 + lane] = (unsigned char)(myrow ^ lane);
  if (mine) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int sl = (head - j) & 7;
      a.X[(long)j * a.slot_stride + sofs + lane] = Xh[sl * nc + lane];
      a.Qh[(long)j * a.slot_stride + sofs + lane] = Qh[sl * nc + lane];
    }
  }
  if (own) {
    // every sample reports for itself: sums of iterations, the longest chain of attempts, the worst status
    // (into the second record: workgroups queued behind this one still read their initial state from the first)
    if (live && lane == 0) {
      TranCtl* cs = p.ctl + 1;
      typedef unsigned long long u64;
      atomicAdd((u64*)&cs->sum_iters, (u64)stl[ST_ITERS]); atomicAdd((u64*)&cs->sum_block_iters, (u64)stl[ST_BITERS]);
      atomicMax((u64*)&cs->naccept, (u64)stl[ST_ACC]); atomicMax((u64*)&cs->nreject, (u64)stl[ST_REJ]); atomicMax((u64*)&cs->nconvfail, (u64)stl[ST_FAIL]);
      atomicMax((u64*)&cs->n_attempts, (u64)stl[ST_ATT]); atomicMax((u64*)&cs->nsaved, (u64)nsaved); atomicMax((u64*)&cs->max_iters, (u64)stl[ST_ITERS]);
      atomicMin(&cs->status, status); atomicMax(&cs->exit_reason, exit_reason);
      if (blk == 0) { cs->t = t; cs->h = h; cs->isave = isave; cs->t_cycles_total = wall_clock64() - cyc0; cs->t_cycles_barrier = cyc_bar; }
    }
  } else
  if (wg == 0 && wave == 0 && lane == 0) {
    TranCtl* cs = p.ctl;
    cs->t = t; cs->h = h; cs->k = k; cs->nhist = nhist; cs->steps_at_order = steps_at_order; cs->reset_rate = reset_rate ? 1 : 0;
    cs->ibp = ibp; cs->isave = isave; cs->status = status; cs->exit_reason = exit_reason; cs->nsaved = nsaved; cs->step = step;
    cs->naccept = stl[ST_ACC]; cs->nreject = stl[ST_REJ]; cs->nconvfail = stl[ST_FAIL]; cs->sum_iters = stl[ST_ITERS]; cs->sum_block_iters = stl[ST_BITERS]; cs->n_attempts = stl[ST_ATT];
    for (int j = 0; j < 8; ++j) cs->tslot[j] = tsl[(head - j) & 7];
    cs->t_cycles_total = wall_clock64() - cyc0; cs->t_cycles_barrier = cyc_bar;
#ifdef CH_STAMPS
    for (int q = 0; q < 12; ++q) cs->stamps[q] = (long long)pacc_[q];
#endif
  }
}

#undef P_BPS
#undef P_NBP

}  // namespace chip
