// oracle.cpp — CPU restatement of the CedarSim hot path (device evaluation + Jacobian assembly +
// LU solve inside DC/transient Newton), used ONLY as the parity checker and as bench.py's
// cpu_baseline ("port").  See oracle/README.md.  Nothing in cedarsim.jl_amd/ may link this.
//
// What it restates, and from where (all paths relative to /root/reference):
//   * device equations ............ src/simpledevices.jl:65-77 (R), :105-109 (C), :128-132 (L),
//                                   :288-300 (V), :327-339 (I), :347-356 (VCVS), :364-373 (VCCS);
//                                   multiplicity src/simulate_ir.jl:54-75; branch sign convention
//                                   src/simulate_ir.jl:112-120
//   * source waveforms ............ src/spectre_env.jl:15-21 (find_t_in_ts), :43-69 (pwl_at_time),
//                                   :153-166 (pulse), :169-176 (spsin), :190-196 ($time in dcop)
//   * DC operating point .......... src/dcop.jl:53-94 (bootstrapped_nlsolve: 10 restarts from
//                                   1e-7*randn, maxiters 200), :157-200 (du=0, mode=:dcop, abstol)
//   * transient ................... solve(prob, IDA()) (src/sweeps.jl:456): variable-order
//                                   variable-step BDF(1..5) + Newton + dense LU.  IDA itself is an
//                                   un-vendored dependency (Sundials 4.24.0, Manifest.toml:2783);
//                                   the published algorithm (BDF in divided-difference form with
//                                   WRMS local-error control) is restated in variable-coefficient
//                                   Lagrange form.  Step sequences are NOT expected to match IDA;
//                                   parity is on observables at tolerance (SURVEY §7 hard parts).
//   * the reference has no MNA: DAECompiler reduces the equations (doc/circuit_simulation.jmd:211).
//     The oracle solves the full, unreduced MNA system (every node voltage and every branch
//     current), so comparing it with the engine also checks the engine's structural reduction.
//
// Pinning: closed-form answers of the reference's own tests (tests/golden/closed_form.json);
// BSIM4 arithmetic is PARITY UNPINNED (see oracle_bsim4.hpp).
#include <algorithm>
#include <chrono>
#include <complex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/cedarhip.h"
#include "oracle_bsim4.hpp"
#include "oracle_dual.hpp"
// compiled Verilog-A modules: the generated device functions (host instantiation) — shared with the engine by
// construction; their independent check is the Python AST interpreter (tests/test_va_compiler.py)
#include "../cedarsim.jl_amd/csrc/_generated/va_models.hpp"

namespace oracle {

// ---------------------------------------------------------------------------------------------
// RNG shared (by specification, not by code) with the engine: splitmix64 + Box-Muller.
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
  double normal() {
    double u1 = uniform(), u2 = uniform();
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

// ---------------------------------------------------------------------------------------------
struct Source {
  int kind;
  double dc;
  double ac = 0.0;  // |ac| (src/simpledevices.jl:293: spec.ϵω * abs(VS.ac))
  double par[CH_SRC_NPAR];
  std::vector<double> ts, ys;
};

// src/spectre_env.jl:15-21 — a break point belongs to the NEXT segment
static int find_t_in_ts(const std::vector<double>& ts, double t) {
  int idx = (int)(std::lower_bound(ts.begin(), ts.end(), t) - ts.begin()) + 1;  // 1-based searchsortedfirst
  if (idx <= (int)ts.size() && ts[idx - 1] == t) return idx + 1;
  return idx;
}
// src/spectre_env.jl:43-69
static double pwl_at_time(const std::vector<double>& ts, const std::vector<double>& ys, double t) {
  int n = (int)ts.size();
  if (n == 0) return 0.0;
  int i = find_t_in_ts(ts, t);
  if (i <= 1) return ys[0];
  if (i > n) return ys[n - 1];
  if (ys[i - 2] == ys[i - 1]) return ys[i - 1];
  if (ts[i - 1] == ts[i - 2]) return 0.5 * (ys[i - 2] + ys[i - 1]);
  double slope = (ys[i - 1] - ys[i - 2]) / (ts[i - 1] - ts[i - 2]);
  return ys[i - 2] + (t - ts[i - 2]) * slope;
}
static double sind(double deg) {
  // Base.sind: exact at multiples of 180; reduce first for accuracy
  double r = std::fmod(deg, 360.0);
  return std::sin(r * (3.14159265358979323846 / 180.0));
}
// mode: 0 = :dcop (dc value), 1 = :tran at time t, 2 = :tranop (tran value at $time()=0)
static double source_value(const Source& s, double t, int mode) {
  if (mode == 0) return s.dc;
  if (mode == 2) t = 0.0;
  switch (s.kind) {
    case CH_SRC_DC: return s.par[0];
    case CH_SRC_PWL: return pwl_at_time(s.ts, s.ys, t);
    case CH_SRC_PULSE: {  // src/spectre_env.jl:153-166
      double v1 = s.par[0], v2 = s.par[1], td = s.par[2], tr = s.par[3], tf = s.par[4], pw = s.par[5], per = s.par[6];
      std::vector<double> ts = {td, td + tr, td + tr + pw, td + tr + pw + tf};
      std::vector<double> ys = {v1, v2, v2, v1};
      double tt = std::isfinite(per) ? std::fmod(t, per) : t;
      return pwl_at_time(ts, ys, tt);
    }
    case CH_SRC_SIN: {  // src/spectre_env.jl:169-176
      double vo = s.par[0], va = s.par[1], freq = s.par[2], td = s.par[3], theta = s.par[4], phase = s.par[5], ncyc = s.par[6];
      if (td < t && t < ncyc / freq) return vo + va * std::exp(-(t - td) * theta) * sind(360.0 * freq * (t - td) + phase);
      return vo + va * sind(phase);
    }
  }
  return 0.0;
}
// Does the source VALUE jump at t (left limit != value at t), or is t only a corner (slope discontinuity)?  The integrator
// restarts at order 1 only behind a jump; a continuous corner is landed on exactly and stepped over with the history kept
// (what IDA does with `tstops`: src/spectre_env.jl:71-77 only announces the times).  A corner so steep that the two values differ
// by rounding alone is taken for a jump: that is the old, always safe behaviour.
static bool source_jumps_at(const Source& s, double t) {
  double amp = 0.0;
  if (s.kind == CH_SRC_PWL) for (double y : s.ys) amp = std::max(amp, std::fabs(y));
  else if (s.kind == CH_SRC_PULSE) amp = std::max(std::fabs(s.par[0]), std::fabs(s.par[1]));
  else if (s.kind == CH_SRC_SIN) amp = std::fabs(s.par[0]) + std::fabs(s.par[1]);
  const double a = source_value(s, std::nextafter(t, -INFINITY), 1), b = source_value(s, t, 1);
  return std::fabs(a - b) > 1e-9 * amp;
}
// break points inside (t0, t1] — what time_periodic_singularities! announces (spectre_env.jl:71-77)
static void source_breakpoints(const Source& s, double t0, double t1, std::vector<double>& out) {
  if (s.kind == CH_SRC_PWL) {
    for (double t : s.ts) if (t > t0 && t < t1) out.push_back(t);
  } else if (s.kind == CH_SRC_PULSE) {
    double td = s.par[2], tr = s.par[3], tf = s.par[4], pw = s.par[5], per = s.par[6];
    double c[4] = {td, td + tr, td + tr + pw, td + tr + pw + tf};
    if (!std::isfinite(per) || per <= 0) { for (double t : c) if (t > t0 && t < t1) out.push_back(t); }
    else {
      long k0 = (long)std::floor(t0 / per) - 1;
      if (k0 < 0) k0 = 0;
      for (long k = k0; k * per < t1 && (k - k0) < 10000000; ++k) {
        for (double tc : c) { double t = tc + k * per; if (t > t0 && t < t1) out.push_back(t); }
        // the periodic extension `t mod period` may jump where it wraps (e.g. td+tr+pw > period,
        // test/inverter.jl:84-99): make the wrap a break point so that it is crossed by a restart step
        if (k >= 1 && k * per > t0 && k * per < t1) out.push_back(k * per);
      }
    }
  } else if (s.kind == CH_SRC_SIN) {
    double td = s.par[3], te = s.par[6] / s.par[2];
    if (td > t0 && td < t1) out.push_back(td);
    if (std::isfinite(te) && te > t0 && te < t1) out.push_back(te);
  }
}

// ---------------------------------------------------------------------------------------------
struct Device {
  int kind;
  int node[CH_DEV_NNODE];
  int ipar[CH_DEV_NIPAR];
  double par[CH_DEV_NPAR];
  double mult;
  int branch;  // index into branch unknowns or -1
  int mos;     // index into mos_size or -1
};

struct Circuit {
  int n_nodes = 0, n_branch = 0, n = 0;
  std::vector<Device> dev;
  std::vector<Source> src;
  std::vector<std::vector<double>> model;
  double temp = 27.0, gmin = 1e-12, scale = 1.0;
  std::vector<int> slot_kind, slot_a, slot_b;
  std::vector<int> obs_kind, obs_index;
  std::vector<B4Size> mos_size;
  std::vector<int> mos_dev;  // device index of each MOS
  std::vector<double> va_par; // parameter blocks of the Verilog-A instances
  bool sizes_dirty = true;
  // "reference-like" proxy of the transient Newton loop: finite-difference Jacobian from
  // n+1 residual evaluations (AutoFiniteDiff, src/dcop.jl:28,53-94), reused across iterations and steps like IDA's
  // modified Newton (solve(prob, IDA()), src/sweeps.jl:456; benchmarks/gf180_dff_solver_bench.jl:60-81).  A cost model, not CedarSim.
  bool proxy_fd_reuse = false;
  std::string err;

  int refresh_sizes() {
    if (!sizes_dirty) return CH_OK;
    for (size_t k = 0; k < mos_dev.size(); ++k) {
      Device& d = dev[mos_dev[k]];
      double ip[CH_DEV_NPAR];
      for (int j = 0; j < CH_DEV_NPAR; ++j) ip[j] = d.par[j];
      ip[CH_MOS_W] *= scale; ip[CH_MOS_L] *= scale;  // SPICE .option scale (src/spectre.jl:1162-1176)
      int rc = b4_setup(model[d.ipar[0]].data(), ip, temp, mos_size[k]);
      if (rc != CH_OK) { err = "invalid MOS geometry/model"; return rc; }
    }
    sizes_dirty = false;
    return CH_OK;
  }
};

struct Eval {
  std::vector<double> F, Q, G, C;  // G, C dense row-major n*n
  void resize(int n) { F.assign(n, 0.0); Q.assign(n, 0.0); G.assign((size_t)n * n, 0.0); C.assign((size_t)n * n, 0.0); }
};

// One residual + Jacobian evaluation: F = i(x,t), Q = q(x), G = di/dx, C = dq/dx.
static void evaluate(Circuit& c, const double* x, double t, int mode, Eval& e, bool resid_only = false) {
  const int n = c.n;
  if (resid_only) { e.F.assign(n, 0.0); e.Q.assign(n, 0.0); if (e.G.size() != (size_t)n * n) { e.G.assign((size_t)n * n, 0.0); e.C.assign((size_t)n * n, 0.0); } }
  else e.resize(n);
  auto V = [&](int node) { return node == 0 ? 0.0 : x[node - 1]; };
  auto addF = [&](int node, double v) { if (node) e.F[node - 1] += v; };
  auto addQ = [&](int node, double v) { if (node) e.Q[node - 1] += v; };
  auto addG = [&](int r, int cc, double v) { if (r >= 0 && cc >= 0) e.G[(size_t)r * n + cc] += v; };
  auto addC = [&](int r, int cc, double v) { if (r >= 0 && cc >= 0) e.C[(size_t)r * n + cc] += v; };
  for (const Device& d : c.dev) {
    const int a = d.node[0], b = d.node[1];
    const int ra = a - 1, rb = b - 1;
    const double m = d.mult;
    switch (d.kind) {
      case CH_DEV_R: {
        double g = 1.0 / d.par[0];
        double i = g * (V(a) - V(b));
        addF(a, m * i); addF(b, -m * i);
        addG(ra, ra, m * g); addG(ra, rb, -m * g); addG(rb, ra, -m * g); addG(rb, rb, m * g);
      } break;
      case CH_DEV_C: {
        double cap = d.par[0];
        double q = cap * (V(a) - V(b));
        addQ(a, m * q); addQ(b, -m * q);
        addC(ra, ra, m * cap); addC(ra, rb, -m * cap); addC(rb, ra, -m * cap); addC(rb, rb, m * cap);
      } break;
      case CH_DEV_L: {
        int br = c.n_nodes + d.branch;
        double i = x[br];
        addF(a, m * i); addF(b, -m * i);
        addG(ra, br, m); addG(rb, br, -m);
        e.F[br] += V(a) - V(b);
        addG(br, ra, 1.0); addG(br, rb, -1.0);
        e.Q[br] += -d.par[0] * i;
        addC(br, br, -d.par[0]);
      } break;
      case CH_DEV_V: {
        int br = c.n_nodes + d.branch;
        double i = x[br];
        addF(a, m * i); addF(b, -m * i);
        addG(ra, br, m); addG(rb, br, -m);
        e.F[br] += V(a) - V(b) - source_value(c.src[d.ipar[0]], t, mode);
        addG(br, ra, 1.0); addG(br, rb, -1.0);
      } break;
      case CH_DEV_I: {
        double i = source_value(c.src[d.ipar[0]], t, mode);
        addF(a, m * i); addF(b, -m * i);
      } break;
      case CH_DEV_VCVS: {
        int br = c.n_nodes + d.branch;
        int cc = d.node[2], dd = d.node[3];
        double i = x[br], gain = d.par[0];
        addF(a, m * i); addF(b, -m * i);
        addG(ra, br, m); addG(rb, br, -m);
        e.F[br] += V(a) - V(b) - gain * (V(cc) - V(dd));
        addG(br, ra, 1.0); addG(br, rb, -1.0); addG(br, cc - 1, -gain); addG(br, dd - 1, gain);
      } break;
      case CH_DEV_VCCS: {
        int cc = d.node[2], dd = d.node[3];
        double gain = d.par[0];
        double i = gain * (V(cc) - V(dd));
        addF(a, m * i); addF(b, -m * i);
        addG(ra, cc - 1, m * gain); addG(ra, dd - 1, -m * gain); addG(rb, cc - 1, -m * gain); addG(rb, dd - 1, m * gain);
      } break;
      case CH_DEV_VA: {
        const int mod = d.ipar[0];
        const int nt = va_gen::MODULES[mod].n_nodes;
        double vv[8] = {0}, st[144];
        for (int k = 0; k < nt; ++k) vv[k] = V(d.node[k]);
        for (int k = 0; k < 144; ++k) st[k] = 0.0;
        const va::Env env{c.temp + 273.15, c.gmin};
        va_gen::stamp(mod, c.va_par.data() + d.ipar[1], vv, env, m, st);
        for (int k = 0; k < nt; ++k) {
          addF(d.node[k], st[k]);
          addQ(d.node[k], st[8 + k]);
          for (int j = 0; j < nt; ++j) { addG(d.node[k] - 1, d.node[j] - 1, st[16 + k * 8 + j]); addC(d.node[k] - 1, d.node[j] - 1, st[80 + k * 8 + j]); }
        }
      } break;
      case CH_DEV_MOS: {
        if (resid_only) {   // plain doubles: what one residual call of the reference costs
          double I[4], Qt[4];
          b4_eval<double>(c.mos_size[d.mos], V(d.node[0]), V(d.node[1]), V(d.node[2]), V(d.node[3]), c.gmin, I, Qt);
          for (int k = 0; k < 4; ++k) { addF(d.node[k], m * I[k]); addQ(d.node[k], m * Qt[k]); }
          break;
        }
        typedef Dual<4> D4;
        D4 vt[4];
        for (int k = 0; k < 4; ++k) vt[k] = D4::var(V(d.node[k]), k);
        D4 I[4], Qt[4];
        b4_eval<D4>(c.mos_size[d.mos], vt[0], vt[1], vt[2], vt[3], c.gmin, I, Qt);
        for (int k = 0; k < 4; ++k) {
          addF(d.node[k], m * I[k].v);
          addQ(d.node[k], m * Qt[k].v);
          for (int j = 0; j < 4; ++j) {
            addG(d.node[k] - 1, d.node[j] - 1, m * I[k].d[j]);
            addC(d.node[k] - 1, d.node[j] - 1, m * Qt[k].d[j]);
          }
        }
      } break;
    }
  }
}

// Dense LU with partial pivoting (what Sundials' default :Dense linear solver does).
// A is overwritten by its factors; returns false when singular.
static bool lu_factor(std::vector<double>& A, int n, std::vector<int>& piv) {
  piv.resize(n);
  for (int k = 0; k < n; ++k) {
    int p = k; double best = std::fabs(A[(size_t)k * n + k]);
    for (int i = k + 1; i < n; ++i) { double v = std::fabs(A[(size_t)i * n + k]); if (v > best) { best = v; p = i; } }
    if (!(best > 0.0) || !std::isfinite(best)) return false;
    piv[k] = p;
    if (p != k) for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
    double inv = 1.0 / A[(size_t)k * n + k];
    for (int i = k + 1; i < n; ++i) {
      double l = A[(size_t)i * n + k] * inv;
      if (l == 0.0) continue;
      A[(size_t)i * n + k] = l;
      double* ri = &A[(size_t)i * n];
      const double* rk = &A[(size_t)k * n];
      for (int j = k + 1; j < n; ++j) ri[j] -= l * rk[j];
    }
  }
  return true;
}
static void lu_solve(const std::vector<double>& A, int n, const std::vector<int>& piv, std::vector<double>& b) {
  // rows of L were swapped along with later pivots (LAPACK convention): permute b first
  for (int k = 0; k < n; ++k) if (piv[k] != k) std::swap(b[k], b[piv[k]]);
  for (int k = 0; k < n; ++k) {
    double bk = b[k];
    if (bk != 0.0) for (int i = k + 1; i < n; ++i) b[i] -= A[(size_t)i * n + k] * bk;
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int j = k + 1; j < n; ++j) s -= A[(size_t)k * n + j] * b[j];
    b[k] = s / A[(size_t)k * n + k];
  }
}

static double inf_norm(const std::vector<double>& v) { double m = 0; for (double x : v) m = std::max(m, std::fabs(x)); return m; }

// ---------------------------------------------------------------------------------------------
// DC operating point — CedarDCOp (src/dcop.jl:157-200) + bootstrapped_nlsolve (:53-94).
// gshunt: extra conductance to ground on every node (homotopy fallback, "TODO: Cedar specific
// homotopies" at dcop.jl:176).
static int dc_newton(Circuit& c, std::vector<double>& x, const ch_dc_opts& o, int mode, double gshunt, ch_stats* st) {
  const int n = c.n;
  Eval e; std::vector<double> A, rhs; std::vector<int> piv;
  for (int it = 0; it <= o.maxiters; ++it) {
    evaluate(c, x.data(), 0.0, mode, e);
    if (st) { st->nf++; st->njacs++; }
    if (gshunt > 0) for (int i = 0; i < c.n_nodes; ++i) { e.F[i] += gshunt * x[i]; e.G[(size_t)i * n + i] += gshunt; }
    double fn = inf_norm(e.F);
    if (!std::isfinite(fn)) return CH_ERR_SINGULAR;
    if (fn < o.abstol) return CH_OK;
    if (it == o.maxiters) break;
    A = e.G; rhs.resize(n);
    for (int i = 0; i < n; ++i) rhs[i] = -e.F[i];
    if (!lu_factor(A, n, piv)) return CH_ERR_SINGULAR;
    lu_solve(A, n, piv, rhs);
    if (st) { st->nfactors++; st->nsolve++; st->nnonliniter++; }
    double scale = 1.0;
    bool has_va = false;
    for (const Device& d : c.dev) if (d.kind == CH_DEV_VA) { has_va = true; break; }
    if (o.dv_max > 0 && (!c.mos_dev.empty() || has_va)) {  // damping only where nonlinear devices exist; linear circuits take the full Newton step
      double mx = 0; for (int i = 0; i < c.n_nodes; ++i) mx = std::max(mx, std::fabs(rhs[i]));
      if (mx > o.dv_max) scale = o.dv_max / mx;
    }
    for (int i = 0; i < n; ++i) x[i] += scale * rhs[i];
  }
  return CH_ERR_MAXITERS;
}

static int dc_solve(Circuit& c, const ch_dc_opts& o, std::vector<double>& x, ch_stats* st) {
  int rc = c.refresh_sizes();
  if (rc != CH_OK) return rc;
  const int n = c.n;
  const int mode = o.tran_mode ? 2 : 0;
  Rng rng(o.seed);
  int last = CH_ERR_MAXITERS;
  for (int r = 0; r < std::max(1, o.n_restarts); ++r) {
    x.assign(n, 0.0);
    if (r == 0 && o.x0) for (int i = 0; i < n; ++i) x[i] = o.x0[i];
    else for (int i = 0; i < n; ++i) x[i] = 1e-7 * rng.normal();
    last = dc_newton(c, x, o, mode, 0.0, st);
    if (last == CH_OK) return CH_OK;
    if (st) { st->nrestarts++; st->nnonlinconvfail++; }
  }
  // homotopy fallback: gmin stepping from a stiff shunt down to none
  x.assign(n, 0.0);
  bool ok = true;
  for (double g = 1e-2; g >= 1e-13; g *= 0.1) {
    int rcg = dc_newton(c, x, o, mode, g, st);
    if (rcg != CH_OK) { ok = false; break; }
  }
  if (ok) { last = dc_newton(c, x, o, mode, 0.0, st); if (last == CH_OK) return CH_OK; }
  return last;
}

// ---------------------------------------------------------------------------------------------
// variable-coefficient BDF helpers.  tau[0] = t_new, tau[1..] = history times (newest first).
static void bdf_coeffs(const double* tau, int k, double* alpha) {
  // derivative at tau[0] of the Lagrange basis through tau[0..k]
  double a0 = 0;
  for (int m = 1; m <= k; ++m) a0 += 1.0 / (tau[0] - tau[m]);
  alpha[0] = a0;
  for (int j = 1; j <= k; ++j) {
    double num = 1, den = 1;
    for (int m = 1; m <= k; ++m) if (m != j) num *= (tau[0] - tau[m]);
    for (int m = 0; m <= k; ++m) if (m != j) den *= (tau[j] - tau[m]);
    alpha[j] = num / den;
  }
}
// weights of the polynomial through tau[1..np] evaluated at tau[0]
static void extrap_weights(const double* tau, int np, double* w) {
  for (int j = 1; j <= np; ++j) {
    double v = 1;
    for (int i = 1; i <= np; ++i) if (i != j) v *= (tau[0] - tau[i]) / (tau[j] - tau[i]);
    w[j] = v;
  }
}

struct Result {
  std::vector<double> times;
  std::vector<double> values;  // [n_obs][n_times]
  std::vector<double> final_state;
  ch_stats stats;
  int status;
  int n_obs;
};

struct HistPoint { double t; std::vector<double> x, q; };

static double wrms(const std::vector<double>& e, const std::vector<double>& w) {
  double s = 0; size_t n = e.size();
  for (size_t i = 0; i < n; ++i) { double v = e[i] * w[i]; s += v * v; }
  return std::sqrt(s / (double)n);
}
// WRMS norm over the differential unknowns only (mask[i] != 0).  Algebraic unknowns (e.g. the
// branch current of a source that drives a capacitor) may jump at waveform corners and carry no
// local truncation error of their own; IDA offers the same exclusion (IDASetSuppressAlg).
static double wrms_masked(const std::vector<double>& e, const std::vector<double>& w, const std::vector<char>& mask) {
  double s = 0; size_t n = e.size(), cnt = 0;
  for (size_t i = 0; i < n; ++i) if (mask[i]) { double v = e[i] * w[i]; s += v * v; ++cnt; }
  return cnt ? std::sqrt(s / (double)cnt) : 0.0;
}

static int tran_solve(Circuit& c, double t0, double t1, const ch_tran_opts& o, Result& R) {
  using clk = std::chrono::steady_clock;
  auto tstart = clk::now();
  std::memset(&R.stats, 0, sizeof(R.stats));
  const int n = c.n;
  R.n_obs = (int)c.obs_kind.size();
  int rc = c.refresh_sizes();
  if (rc != CH_OK) { R.status = rc; return rc; }
  const int kmax = std::min(5, std::max(1, o.max_order));
  const double span = t1 - t0;
  const double dtmax = o.dtmax > 0 ? o.dtmax : span / 10.0;
  const double dtmin = o.dtmin > 0 ? o.dtmin : 1e-15 * span;
  const int max_steps = o.max_steps > 0 ? o.max_steps : 100000;   // Sundials.jl's default maxiters of solve(prob, IDA()) (src/sweeps.jl:456 passes none)
  const int nmaxit = o.newton_maxiters > 0 ? o.newton_maxiters : 10;

  // ---- initialisation: CedarDCOp, then the problem's own mode at t0 ----
  std::vector<double> x(n, 0.0);
  if (o.skip_dc) { if (o.dc.x0) for (int i = 0; i < n; ++i) x[i] = o.dc.x0[i]; }
  else {
    rc = dc_solve(c, o.dc, x, &R.stats);
    if (rc != CH_OK) { R.status = rc; return rc; }
  }
  R.stats.dc_seconds = std::chrono::duration<double>(clk::now() - tstart).count();

  // differential unknowns: those that appear under d/dt (structurally non-zero columns of C)
  std::vector<char> dmask(n, 0);
  for (const Device& d : c.dev) {
    auto mark = [&](int node) { if (node) dmask[node - 1] = 1; };
    if (d.kind == CH_DEV_C) { mark(d.node[0]); mark(d.node[1]); }
    else if (d.kind == CH_DEV_L) dmask[c.n_nodes + d.branch] = 1;
    else if (d.kind == CH_DEV_MOS) for (int k = 0; k < 4; ++k) mark(d.node[k]);
    else if (d.kind == CH_DEV_VA) { const va_gen::ModuleInfo& mi = va_gen::MODULES[d.ipar[0]]; for (int k = 0; k < mi.n_nodes; ++k) if (mi.q_mask & (1u << k)) mark(d.node[k]); }
  }

  // break points with their kind: bpc[i] < 0: some source VALUE jumps there (restart at order 1 behind it); bpc[i] >= 0: a continuous
  // corner, and bpc[i] is the length of the shortest source segment that starts there (the first step behind the corner is capped
  // at a tenth of it).  A source is asked only about its own times.
  std::vector<double> bps, bpc;
  {
    const bool restart_all = std::getenv("CEDARHIP_BP_RESTART_ALL") != nullptr;   // the policy of rounds 1-2, for A/B comparisons
    std::vector<std::pair<double, double>> pts;
    std::vector<double> own;
    for (const Source& s : c.src) {
      own.clear();
      source_breakpoints(s, t0, t1, own);
      std::sort(own.begin(), own.end());
      for (size_t j = 0; j < own.size(); ++j) {
        const double seg = (j + 1 < own.size() ? own[j + 1] : t1) - own[j];
        pts.emplace_back(own[j], (restart_all || source_jumps_at(s, own[j])) ? -1.0 : seg);
      }
    }
    pts.emplace_back(t1, -1.0);
    std::sort(pts.begin(), pts.end());
    for (const auto& pt : pts) {
      if (!bps.empty() && bps.back() == pt.first) { bpc.back() = (bpc.back() < 0 || pt.second < 0) ? -1.0 : std::min(bpc.back(), pt.second); continue; }
      bps.push_back(pt.first); bpc.push_back(pt.second);
    }
  }
  size_t ibp = 0;

  auto obs_of = [&](const std::vector<double>& xs, int k) {
    int idx = c.obs_index[k];
    if (c.obs_kind[k] == 0) return idx == 0 ? 0.0 : xs[idx - 1];
    return xs[c.n_nodes + c.dev[idx].branch];
  };
  std::vector<std::vector<double>> cols(R.n_obs);
  auto save = [&](double t, const std::vector<double>& xs) {
    R.times.push_back(t);
    for (int k = 0; k < R.n_obs; ++k) cols[k].push_back(obs_of(xs, k));
  };

  Eval e;
  evaluate(c, x.data(), t0, 1, e);
  R.stats.nf++;
  std::vector<HistPoint> hist;  // newest first
  hist.push_back({t0, x, e.Q});
  int isave = 0;
  if (o.n_saveat == 0) save(t0, x);
  else while (isave < o.n_saveat && o.saveat[isave] <= t0) { save(o.saveat[isave], x); ++isave; }

  double t = t0;
  // After (re)start only one history point exists, so no local-error estimate is possible: the
  // first step is a tiny backward-Euler step accepted on Newton convergence alone; step sizes then
  // grow by at most 10x per step while the order is 1.
  const double kFirstFrac = 1e-3;
  double h = o.dt0 > 0 ? o.dt0 : std::min(dtmax, 1e-3 * span);
  h = std::min(h, (bps[0] - t0) / 50.0) * kFirstFrac;
  h = std::max(h, 10 * dtmin);
  int k = 1, steps_at_order = 0;
  double newton_rate = 1.0;  // last observed Newton convergence rate (1 = unknown: forces two iterations)
  std::vector<double> xp(n), xn(n), dx(n), w(n), ev(n), A, hq(n), qn(n);
  std::vector<int> piv;
  double tau[8], alpha[8], wts[8];
  int status = CH_OK;
  const bool trace = std::getenv("ORACLE_TRACE") != nullptr;

  for (int step = 0; step < max_steps && t < t1; ) {
    // clip to next break point
    while (ibp < bps.size() && bps[ibp] <= t * (1 + 1e-15) + 1e-300) ++ibp;
    double tb = ibp < bps.size() ? bps[ibp] : t1;
    const double tb_code = ibp < bps.size() ? bpc[ibp] : -1.0;
    const bool tb_jump = tb_code < 0;
    bool hit_bp = false;
    double tn = t + h;
    if (tn >= tb - 1e-3 * h) { tn = tb; hit_bp = true; }
    double hh = tn - t;
    if (hh < dtmin) { status = CH_ERR_DTMIN; break; }

    const int nh = (int)hist.size();
    int kk = std::min(k, nh);  // BDF order actually usable
    tau[0] = tn;
    for (int j = 0; j < nh && j < 7; ++j) tau[j + 1] = hist[j].t;
    // predictor: polynomial through the last min(kk+1, nh) points
    int np = std::min(kk + 1, nh);
    extrap_weights(tau, np, wts);
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 1; j <= np; ++j) s += wts[j] * hist[j - 1].x[i]; xp[i] = s; }
    bdf_coeffs(tau, kk, alpha);
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 1; j <= kk; ++j) s += alpha[j] * hist[j - 1].q[i]; hq[i] = s; }
    for (int i = 0; i < n; ++i) w[i] = 1.0 / (o.reltol * std::fabs(hist[0].x[i]) + o.abstol);

    // a step that lands on a break point sees the sources' LEFT limit there (waveforms may jump);
    // the right limit takes effect in the restart step that follows
    const double tsrc = hit_bp ? std::nextafter(tn, -INFINITY) : tn;
    // ---- Newton ----
    xn = xp;
    bool conv = false;
    double dn_prev = 0.0, rate_new = -1.0;
    int its_this = 0;
    if (c.proxy_fd_reuse) {
      // Modified Newton on a finite-difference Jacobian that is kept while alpha0 stays within 25 % of the value it was
      // built with and the iteration converges (IDA: cj ratio test, at most 4 iterations, refresh + one retry on failure).
      static thread_local std::vector<double> Jlu; static thread_local std::vector<int> Jpiv; static thread_local double J_alpha0 = 0.0; static thread_local int J_n = -1;
      std::vector<double> r0(n), r1(n), xs(n);
      auto resid = [&](const std::vector<double>& xx, std::vector<double>& r) {
        evaluate(c, xx.data(), tsrc, 1, e, true); R.stats.nf++;
        for (int i = 0; i < n; ++i) r[i] = e.F[i] + alpha[0] * e.Q[i] + hq[i];
      };
      auto build_jac = [&]() -> bool {
        resid(xn, r0);
        Jlu.assign((size_t)n * n, 0.0);
        xs = xn;
        for (int j = 0; j < n; ++j) {
          const double sg = 1.4901161193847656e-08 * std::max(std::fabs(xn[j]), 1.0 / w[j]);
          xs[j] = xn[j] + sg;
          resid(xs, r1);
          const double inv = 1.0 / (xs[j] - xn[j]);
          for (int i = 0; i < n; ++i) Jlu[(size_t)i * n + j] = (r1[i] - r0[i]) * inv;
          xs[j] = xn[j];
        }
        R.stats.njacs++; R.stats.nfactors++;
        J_alpha0 = alpha[0]; J_n = n;
        return lu_factor(Jlu, n, Jpiv);
      };
      bool fresh = false;
      if (J_n != n || !(std::fabs(alpha[0] / J_alpha0 - 1.0) <= 0.25) || step == 0 || hist.size() == 1) { if (!build_jac()) { J_n = -1; } fresh = true; }
      for (int attempt = 0; attempt < 2 && !conv && J_n == n; ++attempt) {
        xn = xp; dn_prev = 0.0; its_this = 0;
        for (int it = 0; it < 4; ++it) {
          resid(xn, r0);
          for (int i = 0; i < n; ++i) dx[i] = -r0[i];
          lu_solve(Jlu, n, Jpiv, dx);
          R.stats.nsolve++; R.stats.nnonliniter++; ++its_this;
          bool finite = true;
          for (int i = 0; i < n; ++i) { xn[i] += dx[i]; if (!std::isfinite(xn[i])) finite = false; }
          if (!finite) break;
          const double dn = wrms(dx, w);
          if (it == 0) { if (dn <= 1e-3) { conv = true; break; } }
          else { const double rate = dn_prev > 0 ? dn / dn_prev : 0.0; if (rate > 0.9) break; if (dn * std::min(1.0, rate / (1.0 - rate)) <= 0.1 || dn <= 1e-3) { conv = true; break; } }
          dn_prev = dn;
        }
        if (!conv) { if (fresh) break; xn = xp; if (!build_jac()) { J_n = -1; break; } fresh = true; }
      }
      if (conv) { evaluate(c, xn.data(), tsrc, 1, e, true); R.stats.nf++; qn = e.Q; }
    } else
    for (int it = 0; it < nmaxit; ++it) {
      evaluate(c, xn.data(), tsrc, 1, e);
      R.stats.nf++; R.stats.njacs++;
      A.resize((size_t)n * n);
      for (size_t i = 0; i < (size_t)n * n; ++i) A[i] = e.G[i] + alpha[0] * e.C[i];
      for (int i = 0; i < n; ++i) dx[i] = -(e.F[i] + alpha[0] * e.Q[i] + hq[i]);
      if (!lu_factor(A, n, piv)) break;
      lu_solve(A, n, piv, dx);
      R.stats.nfactors++; R.stats.nsolve++; R.stats.nnonliniter++;
      bool finite = true;
      for (int i = 0; i < n; ++i) { xn[i] += dx[i]; if (!std::isfinite(xn[i])) finite = false; }
      if (!finite) break;
      // first-order consistent charge at the updated point: q(x+dx) ~ q(x) + C dx
      for (int i = 0; i < n; ++i) { double s = e.Q[i]; const double* ci = &e.C[(size_t)i * n]; for (int j = 0; j < n; ++j) s += ci[j] * dx[j]; qn[i] = s; }
      // Convergence (IDA's rate test, IDANls): after the first iteration the remaining error is
      // estimated from the convergence rate observed in earlier solves, so a well-predicted step
      // needs ONE Newton iteration; the rate is refreshed whenever a second iteration runs and is
      // aged (x1.5) otherwise, which forces a periodic two-iteration check.
      const double dn = wrms(dx, w);
      ++its_this;
      if (it == 0) {
        if (dn <= 0.1 || (newton_rate < 0.9 && 2.0 * std::max(newton_rate, 0.02) * dn <= 0.1)) { conv = true; break; }
      } else {
        rate_new = dn_prev > 0 ? dn / dn_prev : 0.0;
        if (dn <= 0.1) { conv = true; break; }
      }
      dn_prev = dn;
    }
    if (conv) newton_rate = its_this >= 2 ? std::min(1.0, std::max(rate_new, 1e-4)) : std::min(1.0, newton_rate * 1.5);
    if (!conv) {
      R.stats.nnonlinconvfail++;
      h = hh * 0.25; k = 1; steps_at_order = 0; newton_rate = 1.0;
      if (hist.size() > 2) hist.resize(2);
      continue;
    }
    // ---- local error test ----
    for (int i = 0; i < n; ++i) { ev[i] = xn[i] - xp[i]; w[i] = 1.0 / (o.reltol * std::max(std::fabs(hist[0].x[i]), std::fabs(xn[i])) + o.abstol); }
    double errk;
    if (np >= kk + 1) errk = (hh / (tn - tau[kk + 1])) * wrms_masked(ev, w, dmask);
    else errk = 0.0;  // (re)start step: accepted on Newton convergence alone
    if (trace) std::fprintf(stderr, "t=%.6e h=%.3e k=%d np=%d err=%.3e %s\n", tn, hh, kk, np, errk, errk > 1.0 ? "REJECT" : "ok");
    if (errk > 1.0) {
      R.stats.nreject++;
      // IDA-style: aim at half the tolerance after a failed error test, shrink by at most 4x
      double fac = 0.9 * std::pow(2.0 * errk + 1e-4, -1.0 / (kk + 1));
      h = hh * std::min(0.9, std::max(0.25, fac));
      steps_at_order = 0;
      continue;
    }
    // ---- accept ----
    R.stats.naccept++;
    ++step;
    // dense output for saveat
    if (o.n_saveat > 0) {
      while (isave < o.n_saveat && o.saveat[isave] <= tn * (1 + 1e-15)) {
        double ts = o.saveat[isave];
        // polynomial through new point + last kk history points
        double tt[8]; int m = std::min(kk, nh) + 1;
        tt[0] = ts; tt[1] = tn; for (int j = 1; j < m; ++j) tt[j + 1] = hist[j - 1].t;
        double ww[8]; extrap_weights(tt, m, ww);
        std::vector<double> xs(n);
        for (int i = 0; i < n; ++i) { double s = ww[1] * xn[i]; for (int j = 2; j <= m; ++j) s += ww[j] * hist[j - 2].x[i]; xs[i] = s; }
        save(ts, xs); ++isave;
      }
    } else save(tn, xn);

    // ---- order / step selection ----
    double fac_k = std::pow(2.0 * errk + 1e-4, -1.0 / (kk + 1));  // step factor that puts the error at half the tolerance
    double best = fac_k; int knew = kk;
    if (np >= kk + 1) {
      ++steps_at_order;
      if (kk > 1) {
        extrap_weights(tau, kk, wts);
        for (int i = 0; i < n; ++i) { double s = 0; for (int j = 1; j <= kk; ++j) s += wts[j] * hist[j - 1].x[i]; ev[i] = xn[i] - s; }
        double em = (hh / (tn - tau[kk])) * wrms_masked(ev, w, dmask);
        double f = std::pow(2.0 * em + 1e-4, -1.0 / kk);
        if (f > best) { best = f; knew = kk - 1; }
      }
      if (kk < kmax && nh >= kk + 2 && steps_at_order >= kk + 1) {
        extrap_weights(tau, kk + 2, wts);
        for (int i = 0; i < n; ++i) { double s = 0; for (int j = 1; j <= kk + 2; ++j) s += wts[j] * hist[j - 1].x[i]; ev[i] = xn[i] - s; }
        double ep = (hh / (tn - tau[kk + 2])) * wrms_masked(ev, w, dmask);
        double f = std::pow(2.0 * ep + 1e-4, -1.0 / (kk + 2));
        if (f > 1.1 * best) { best = f; knew = kk + 1; }
      }
    } else {
      knew = 1;
    }
    if (knew != kk) steps_at_order = 0;
    k = knew;
    // dead band: keep h when the suggested change is small (IDA keeps h unless it can double)
    if (best > 1.0 && best < 1.2) best = 1.0;
    h = hh * std::min(kk == 1 ? 10.0 : 2.0, std::max(0.5, best));
    h = std::min(h, dtmax);

    hist.insert(hist.begin(), HistPoint{tn, xn, qn});
    if ((int)hist.size() > kmax + 2) hist.pop_back();
    t = tn;
    if (hit_bp && t < t1 && !tb_jump) {
      // continuous corner: history and order are kept; the slope of a source has changed, so the first step behind the corner is
      // capped at a tenth of that source's new segment (without the cap it is the old step size, and the Newton iteration from a
      // predictor that knows nothing of the new slope fails there: 7 failures per DFF transient, none with it)
      newton_rate = 1.0;
      h = std::max(dtmin * 10, std::min(h, tb_code / 10.0));
    }
    if (hit_bp && t < t1 && tb_jump) {
      // the sources jump here: restart at order 1 from this point
      hist.resize(1);
      k = 1; steps_at_order = 0; newton_rate = 1.0;
      double nb = t1;
      for (size_t b = ibp; b < bps.size(); ++b) if (bps[b] > t * (1 + 1e-15)) { nb = bps[b]; break; }
      h = std::max(dtmin * 10, std::min(h, (nb - t) / 50.0) * kFirstFrac);
    }
  }
  if (status == CH_OK && t < t1) status = CH_ERR_MAXSTEPS;
  R.values.resize((size_t)R.n_obs * R.times.size());
  for (int kx = 0; kx < R.n_obs; ++kx) std::copy(cols[kx].begin(), cols[kx].end(), R.values.begin() + (size_t)kx * R.times.size());
  R.final_state = hist[0].x;
  R.status = status;
  R.stats.wall_seconds = std::chrono::duration<double>(clk::now() - tstart).count();
  return status;
}


// ---------------------------------------------------------------------------------------------
// Small-signal analyses (src/ac.jl).  ACSol: dss(Ju, M, B, I, 0) with B = ∂F/∂ϵω, sampled as
// C·(jωE − A)⁻¹·B (:75-102, :267-284); here the same linear system in MNA form:
// (G + jωC)·X = −∂F/∂ϵ, the excitation being ϵ·|ac| added to every source value (simpledevices.jl:292-294).
typedef std::complex<double> cplx;
static bool csolve(std::vector<cplx>& A, int n, std::vector<cplx>& b) {  // dense complex LU, partial pivoting
  for (int k = 0; k < n; ++k) {
    int p = k; double best = std::abs(A[(size_t)k * n + k]);
    for (int i = k + 1; i < n; ++i) { double v = std::abs(A[(size_t)i * n + k]); if (v > best) { best = v; p = i; } }
    if (!(best > 0.0) || !std::isfinite(best)) return false;
    if (p != k) { for (int j = 0; j < n; ++j) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]); std::swap(b[k], b[p]); }
    for (int i = k + 1; i < n; ++i) {
      cplx l = A[(size_t)i * n + k] / A[(size_t)k * n + k];
      if (l == cplx(0.0)) continue;
      for (int j = k; j < n; ++j) A[(size_t)i * n + j] -= l * A[(size_t)k * n + j];
      b[i] -= l * b[k];
    }
  }
  for (int k = n - 1; k >= 0; --k) { cplx sacc = b[k]; for (int j = k + 1; j < n; ++j) sacc -= A[(size_t)k * n + j] * b[j]; b[k] = sacc / A[(size_t)k * n + k]; }
  return true;
}
// −∂F/∂ϵ, analytic: V source branch row F = va − vb − (dc + ϵ|ac|); I source KCL rows ±m(dc + ϵ|ac|)
static void ac_rhs(const Circuit& c, std::vector<double>& b) {
  b.assign(c.n, 0.0);
  for (const Device& d : c.dev) {
    if (d.kind == CH_DEV_V) b[c.n_nodes + d.branch] += c.src[d.ipar[0]].ac;
    else if (d.kind == CH_DEV_I) { const double i = d.mult * c.src[d.ipar[0]].ac; if (d.node[0]) b[d.node[0] - 1] -= i; if (d.node[1]) b[d.node[1] - 1] += i; }
  }
}
}  // namespace oracle

// =============================================================================================
// C entry points (ctypes-friendly); same shapes as the ch_* C-ABI so tests read symmetrically.
using namespace oracle;

extern "C" {

void* oracle_build(const ch_desc* d) {
  Circuit* c = new Circuit();
  c->n_nodes = d->n_nodes;
  c->temp = d->temp; c->gmin = d->gmin; c->scale = d->scale;
  for (int i = 0; i < d->n_src; ++i) {
    Source s; s.kind = d->src_kind[i]; s.dc = d->src_dc[i];
    for (int k = 0; k < CH_SRC_NPAR; ++k) s.par[k] = d->src_par[i * CH_SRC_NPAR + k];
    if (d->src_pwl_ofs) for (int k = d->src_pwl_ofs[i]; k < d->src_pwl_ofs[i + 1]; ++k) { s.ts.push_back(d->pwl_t[k]); s.ys.push_back(d->pwl_y[k]); }
    s.ac = d->src_ac ? std::fabs(d->src_ac[i]) : 0.0;
    c->src.push_back(s);
  }
  for (int i = 0; i < d->n_model; ++i) c->model.emplace_back(d->model_par + (size_t)i * CH_B4_NPAR, d->model_par + (size_t)(i + 1) * CH_B4_NPAR);
  for (int i = 0; i < d->n_dev; ++i) {
    Device v; v.kind = d->dev_kind[i];
    for (int k = 0; k < CH_DEV_NNODE; ++k) v.node[k] = d->dev_node[i * CH_DEV_NNODE + k];
    for (int k = 0; k < CH_DEV_NIPAR; ++k) v.ipar[k] = d->dev_ipar[i * CH_DEV_NIPAR + k];
    for (int k = 0; k < CH_DEV_NPAR; ++k) v.par[k] = d->dev_par[i * CH_DEV_NPAR + k];
    v.mult = d->dev_mult[i];
    v.branch = -1; v.mos = -1;
    if (v.kind == CH_DEV_L || v.kind == CH_DEV_V || v.kind == CH_DEV_VCVS) v.branch = c->n_branch++;
    if (v.kind == CH_DEV_MOS) { v.mos = (int)c->mos_dev.size(); c->mos_dev.push_back(i); }
    c->dev.push_back(v);
  }
  if (d->va_par && d->n_va_par > 0) c->va_par.assign(d->va_par, d->va_par + d->n_va_par);
  c->mos_size.resize(c->mos_dev.size());
  c->n = c->n_nodes + c->n_branch;
  for (int i = 0; i < d->n_slot; ++i) { c->slot_kind.push_back(d->slot_kind[i]); c->slot_a.push_back(d->slot_a[i]); c->slot_b.push_back(d->slot_b[i]); }
  for (int i = 0; i < d->n_obs; ++i) { c->obs_kind.push_back(d->obs_kind[i]); c->obs_index.push_back(d->obs_index[i]); }
  return c;
}
void oracle_free(void* h) { delete (Circuit*)h; }
int oracle_n_mna(void* h) { return ((Circuit*)h)->n; }
// 1: transients of this circuit run the finite-difference / Jacobian-reuse cost proxy (see Circuit::proxy_fd_reuse)
int oracle_set_proxy(void* h, int on) { ((Circuit*)h)->proxy_fd_reuse = on != 0; return CH_OK; }
int oracle_n_mos(void* h) { return (int)((Circuit*)h)->mos_dev.size(); }

// remake(prob, p=sim) for one sweep point (src/sweeps.jl:476-478)
int oracle_set_param(void* h, int slot, double value) {
  Circuit* c = (Circuit*)h;
  if (slot < 0 || slot >= (int)c->slot_kind.size()) return CH_ERR_INVALID;
  int a = c->slot_a[slot], b = c->slot_b[slot];
  switch (c->slot_kind[slot]) {
    case CH_SLOT_DEV_PAR: c->dev[a].par[b] = value; if (c->dev[a].kind == CH_DEV_MOS) c->sizes_dirty = true; break;
    case CH_SLOT_DEV_MULT: c->dev[a].mult = value; break;
    case CH_SLOT_MODEL_PAR: c->model[a][b] = value; c->sizes_dirty = true; break;
    case CH_SLOT_SRC_DC: c->src[a].dc = value; if (c->src[a].kind == CH_SRC_DC) c->src[a].par[0] = value; break;
    case CH_SLOT_SRC_PAR: c->src[a].par[b] = value; break;
    case CH_SLOT_TEMP: c->temp = value; c->sizes_dirty = true; break;
    case CH_SLOT_GMIN: c->gmin = value; break;
    case CH_SLOT_VA_PAR: if (a < 0 || (size_t)a >= c->va_par.size()) return CH_ERR_INVALID; c->va_par[a] = value; break;
    default: return CH_ERR_INVALID;
  }
  return CH_OK;
}

int oracle_dc(void* h, const ch_dc_opts* o, double* x_out, ch_stats* st) {
  Circuit* c = (Circuit*)h;
  ch_stats local; std::memset(&local, 0, sizeof(local));
  auto t0 = std::chrono::steady_clock::now();
  std::vector<double> x;
  int rc = dc_solve(*c, *o, x, &local);
  local.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  local.dc_seconds = local.wall_seconds;
  if (x_out) for (int i = 0; i < c->n; ++i) x_out[i] = (i < (int)x.size()) ? x[i] : 0.0;
  if (st) *st = local;
  return rc;
}

// ac!(circ) + freqresp(ac, ·, ωs): x_out[n_freq][n][2] (re, im), MNA order
int oracle_ac(void* h, const ch_dc_opts* o, int n_freq, const double* freqs_hz, double* x_out) {
  Circuit* c = (Circuit*)h;
  std::vector<double> x;
  int rc = dc_solve(*c, *o, x, nullptr);
  if (rc != CH_OK) return rc;
  const int n = c->n;
  Eval e; evaluate(*c, x.data(), 0.0, o->tran_mode ? 2 : 0, e);
  std::vector<double> b; ac_rhs(*c, b);
  for (int f = 0; f < n_freq; ++f) {
    const double w = 6.283185307179586 * freqs_hz[f];
    std::vector<cplx> A((size_t)n * n), r(n);
    for (size_t k = 0; k < A.size(); ++k) A[k] = cplx(e.G[k], w * e.C[k]);
    for (int i = 0; i < n; ++i) r[i] = b[i];
    if (!csolve(A, n, r)) { c->err = "singular small-signal matrix"; return CH_ERR_SINGULAR; }
    for (int i = 0; i < n; ++i) { x_out[((size_t)f * n + i) * 2] = r[i].real(); x_out[((size_t)f * n + i) * 2 + 1] = r[i].imag(); }
  }
  return CH_OK;
}

// noise!(circ) + PSD(noise, sym, ωs) (src/ac.jl:286-305): Σ_k |H_k(jω)|²·pwr_k with pwr = 4kT/res per resistor
// (src/simpledevices.jl:72-76).  Direct method: one solve per noise source (the engine uses the adjoint).
int oracle_noise(void* h, const ch_dc_opts* o, int out_mna, int n_freq, const double* freqs_hz, double* psd_out) {
  Circuit* c = (Circuit*)h;
  std::vector<double> x;
  int rc = dc_solve(*c, *o, x, nullptr);
  if (rc != CH_OK) return rc;
  const int n = c->n;
  if (out_mna < 0 || out_mna >= n) return CH_ERR_INVALID;
  Eval e; evaluate(*c, x.data(), 0.0, o->tran_mode ? 2 : 0, e);
  const double kB = 1.380649e-23, T = c->temp + 273.15;
  // noise sources at the operating point: (node a, node b) as circuit nodes (0 = ground), power, flicker exponent
  struct Src { int a, b; double pwr, ex; };
  std::vector<Src> srcs;
  for (const Device& d : c->dev) {
    if (d.kind == CH_DEV_R) srcs.push_back({d.node[0], d.node[1], 4.0 * kB * T * d.mult / d.par[0], 0.0});
    else if (d.kind == CH_DEV_VA) {
      double vv[8] = {0};
      const int nt = va_gen::MODULES[d.ipar[0]].n_nodes;
      for (int k = 0; k < nt; ++k) vv[k] = d.node[k] ? x[d.node[k] - 1] : 0.0;
      va::NoiseRec rec[va::MAX_NOISE];
      const va::Env env{T, c->gmin};
      const int nn = va_gen::noise(d.ipar[0], c->va_par.data() + d.ipar[1], vv, env, rec);
      for (int k = 0; k < nn; ++k) srcs.push_back({d.node[rec[k].a], rec[k].b >= 0 ? d.node[rec[k].b] : 0, d.mult * rec[k].pwr, rec[k].ex});
    }
  }
  for (int f = 0; f < n_freq; ++f) {
    const double w = 6.283185307179586 * freqs_hz[f];
    double acc = 0.0;
    for (const Src& sr : srcs) {
      if (sr.pwr == 0.0) continue;
      std::vector<cplx> A((size_t)n * n), r(n, cplx(0.0));
      for (size_t k = 0; k < A.size(); ++k) A[k] = cplx(e.G[k], w * e.C[k]);
      // unit noise current from node a to node b through the source: leaves a (F_a += 1), enters b
      if (sr.a) r[sr.a - 1] -= 1.0;
      if (sr.b) r[sr.b - 1] += 1.0;
      if (!csolve(A, n, r)) { c->err = "singular small-signal matrix"; return CH_ERR_SINGULAR; }
      acc += std::norm(r[out_mna]) * (sr.ex == 0.0 ? sr.pwr : sr.pwr / std::pow(freqs_hz[f], sr.ex));
    }
    psd_out[f] = acc;
  }
  return CH_OK;
}

// noise records of one compiled module at given node voltages: out[4*k..] = (a, b, pwr, exp); returns the count
int oracle_va_noise(int mod, const double* par, const double* v, double temperature_k, double gmin, double* out) {
  if (mod < 0 || mod >= va_gen::N_MODULES) return -1;
  double vv[8] = {0};
  for (int k = 0; k < va_gen::MODULES[mod].n_nodes; ++k) vv[k] = v[k];
  va::NoiseRec rec[va::MAX_NOISE];
  const va::Env env{temperature_k, gmin};
  const int n = va_gen::noise(mod, par, vv, env, rec);
  for (int k = 0; k < n; ++k) { out[4 * k] = rec[k].a; out[4 * k + 1] = rec[k].b; out[4 * k + 2] = rec[k].pwr; out[4 * k + 3] = rec[k].ex; }
  return n;
}

// one compiled Verilog-A module at given node voltages (host instantiation of the generated code)
int oracle_va_eval(int mod, const double* par, const double* v, double temperature_k, double gmin, double* st) {
  if (mod < 0 || mod >= va_gen::N_MODULES) return CH_ERR_INVALID;
  double vv[8] = {0};
  for (int k = 0; k < va_gen::MODULES[mod].n_nodes; ++k) vv[k] = v[k];
  for (int k = 0; k < 144; ++k) st[k] = 0.0;
  const va::Env env{temperature_k, gmin};
  va_gen::stamp(mod, par, vv, env, 1.0, st);
  return CH_OK;
}
int oracle_va_opvars(int mod, const double* par, const double* v, double temperature_k, double gmin, double* op) {
  if (mod < 0 || mod >= va_gen::N_MODULES) return -1;
  double vv[8] = {0};
  for (int k = 0; k < va_gen::MODULES[mod].n_nodes; ++k) vv[k] = v[k];
  const va::Env env{temperature_k, gmin};
  va_gen::opvars(mod, par, vv, env, op);
  return va_gen::N_OPVARS[mod];
}
const char* oracle_va_opvar_name(int mod, int k) { return (mod >= 0 && mod < va_gen::N_MODULES && k >= 0 && k < va_gen::N_OPVARS[mod]) ? va_gen::OPNAMES[mod][k] : nullptr; }
int oracle_va_n_modules(void) { return va_gen::N_MODULES; }
const char* oracle_va_module_name(int i) { return (i >= 0 && i < va_gen::N_MODULES) ? va_gen::MODULES[i].name : nullptr; }

void* oracle_tran(void* h, double t0, double t1, const ch_tran_opts* o) {
  Circuit* c = (Circuit*)h;
  Result* R = new Result();
  tran_solve(*c, t0, t1, *o, *R);
  return R;
}
int64_t oracle_result_n_times(void* r) { return (int64_t)((Result*)r)->times.size(); }
const double* oracle_result_times(void* r) { return ((Result*)r)->times.data(); }
const double* oracle_result_values(void* r) { return ((Result*)r)->values.data(); }
const double* oracle_result_final_state(void* r) { return ((Result*)r)->final_state.data(); }
int oracle_result_status(void* r) { return ((Result*)r)->status; }
int oracle_result_stats(void* r, ch_stats* s) { *s = ((Result*)r)->stats; return CH_OK; }
void oracle_result_free(void* r) { delete (Result*)r; }

// F = i(x,t) + alpha0*q(x), Q = q(x), J = G + alpha0*C (dense row-major)
int oracle_eval(void* h, const double* x, double t, double alpha0, int mode, double* F, double* Q, double* J) {
  Circuit* c = (Circuit*)h;
  int rc = c->refresh_sizes();
  if (rc != CH_OK) return rc;
  Eval e;
  evaluate(*c, x, t, mode == 0 ? 0 : 1, e);
  const int n = c->n;
  for (int i = 0; i < n; ++i) { if (F) F[i] = e.F[i] + alpha0 * e.Q[i]; if (Q) Q[i] = e.Q[i]; }
  if (J) for (size_t i = 0; i < (size_t)n * n; ++i) J[i] = e.G[i] + alpha0 * e.C[i];
  return CH_OK;
}

// per-instance BSIM4 stamps: out[n_mos][40] = {i[4], q[4], g[16], c[16]} (multiplier NOT applied)
int oracle_mos_eval(void* h, const double* v, double* out) {
  Circuit* c = (Circuit*)h;
  int rc = c->refresh_sizes();
  if (rc != CH_OK) return rc;
  typedef Dual<4> D4;
  for (size_t k = 0; k < c->mos_dev.size(); ++k) {
    D4 vt[4], I[4], Q[4];
    for (int j = 0; j < 4; ++j) vt[j] = D4::var(v[k * 4 + j], j);
    b4_eval<D4>(c->mos_size[k], vt[0], vt[1], vt[2], vt[3], c->gmin, I, Q);
    double* o = out + k * 40;
    for (int j = 0; j < 4; ++j) { o[j] = I[j].v; o[4 + j] = Q[j].v; for (int m = 0; m < 4; ++m) { o[8 + j * 4 + m] = I[j].d[m]; o[24 + j * 4 + m] = Q[j].d[m]; } }
  }
  return CH_OK;
}
// value-only evaluation in plain double arithmetic (to cross-check the dual derivatives by FD)
int oracle_mos_eval_values(void* h, const double* v, double* out) {
  Circuit* c = (Circuit*)h;
  int rc = c->refresh_sizes();
  if (rc != CH_OK) return rc;
  for (size_t k = 0; k < c->mos_dev.size(); ++k) {
    double I[4], Q[4];
    b4_eval<double>(c->mos_size[k], v[k * 4 + 0], v[k * 4 + 1], v[k * 4 + 2], v[k * 4 + 3], c->gmin, I, Q);
    for (int j = 0; j < 4; ++j) { out[k * 8 + j] = I[j]; out[k * 8 + 4 + j] = Q[j]; }
  }
  return CH_OK;
}

// waveform value, for pinning against test/transients.jl:66-96
double oracle_source_value(void* h, int src, double t, int mode) { return source_value(((Circuit*)h)->src[src], t, mode); }

const char* oracle_last_error(void* h) { return ((Circuit*)h)->err.c_str(); }

}  // extern "C"
