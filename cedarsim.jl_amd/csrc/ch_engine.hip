// ch_engine.hip — host side of the engine: contexts, circuits, parameter tables, the sequential
// adaptive time stepper and the C-ABI of include/cedarhip.h.
//
// Division of labour (BASELINE.json north_star): the outer adaptive time stepper and all one-off
// analysis stay on the host; every Newton solve runs on the GPU (ch_kernels.hpp).  Per step
// attempt the host (1) evaluates the source waveforms at t_new and uploads the few known-node /
// source values, (2) launches ONE fused Newton kernel + a tiny reduction, (3) reads back a 64-byte
// summary (converged?, iterations, local-error norms for orders k-1,k,k+1) and decides
// accept/reject, next step and next order — the job IDA does in the reference (src/sweeps.jl:456).
// There is NO CPU fallback: without a HIP device ch_create fails.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <array>
#include <vector>

#include "../../include/cedarhip.h"
#include "ch_analysis.hpp"
#include "ch_bsim4.hpp"
#include "ch_kernels.hpp"
#include "ch_persist.hpp"
#include "ch_sparse.hpp"

using namespace chip;
using hclock = std::chrono::steady_clock;

#define HIPCHK(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) { set_err(std::string(#call) + ": " + hipGetErrorString(e_)); return CH_ERR_DEVICE; } \
  } while (0)

struct ch_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
};

namespace {

// ---- RNG: splitmix64 + Box-Muller, as specified for the DC initial guess u0 = 1e-7*randn (dcop.jl:60)
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
  double normal() { double u1 = uniform(), u2 = uniform(); return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
};

// ---- source waveforms (src/spectre_env.jl:15-21, :43-69, :153-166, :169-176) ----
int find_t_in_ts(const std::vector<double>& ts, double t) {
  int idx = (int)(std::lower_bound(ts.begin(), ts.end(), t) - ts.begin()) + 1;
  if (idx <= (int)ts.size() && ts[idx - 1] == t) return idx + 1;  // a break point belongs to the next segment
  return idx;
}
double pwl_at_time(const double* ts, const double* ys, int n, double t) {
  if (n == 0) return 0.0;
  int i = (int)(std::lower_bound(ts, ts + n, t) - ts) + 1;
  if (i <= n && ts[i - 1] == t) ++i;
  if (i <= 1) return ys[0];
  if (i > n) return ys[n - 1];
  if (ys[i - 2] == ys[i - 1]) return ys[i - 1];
  if (ts[i - 1] == ts[i - 2]) return 0.5 * (ys[i - 2] + ys[i - 1]);
  return ys[i - 2] + (t - ts[i - 2]) * ((ys[i - 1] - ys[i - 2]) / (ts[i - 1] - ts[i - 2]));
}
double sind(double deg) { return std::sin(std::fmod(deg, 360.0) * (3.14159265358979323846 / 180.0)); }
// mode 0 :dcop, 1 :tran at t, 2 :tranop (t = 0)
double source_value(const HSource& s, const double* par, double dc, double t, int mode) {
  if (mode == 0) return dc;
  if (mode == 2) t = 0.0;
  switch (s.kind) {
    case CH_SRC_DC: return par[0];
    case CH_SRC_PWL: return pwl_at_time(s.ts.data(), s.ys.data(), (int)s.ts.size(), t);
    case CH_SRC_PULSE: {
      const double td = par[2], tr = par[3], tf = par[4], pw = par[5], per = par[6];
      const double ts[4] = {td, td + tr, td + tr + pw, td + tr + pw + tf}, ys[4] = {par[0], par[1], par[1], par[0]};
      return pwl_at_time(ts, ys, 4, std::isfinite(per) ? std::fmod(t, per) : t);
    }
    case CH_SRC_SIN: {
      const double vo = par[0], va = par[1], f = par[2], td = par[3], th = par[4], ph = par[5], nc = par[6];
      if (td < t && t < nc / f) return vo + va * std::exp(-(t - td) * th) * sind(360.0 * f * (t - td) + ph);
      return vo + va * sind(ph);
    }
  }
  return 0.0;
}
void source_breakpoints(const HSource& s, const double* par, double t0, double t1, std::vector<double>& out) {
  if (s.kind == CH_SRC_PWL) { for (double t : s.ts) if (t > t0 && t < t1) out.push_back(t); }
  else if (s.kind == CH_SRC_PULSE) {
    const double td = par[2], tr = par[3], tf = par[4], pw = par[5], per = par[6];
    const double c[4] = {td, td + tr, td + tr + pw, td + tr + pw + tf};
    if (!std::isfinite(per) || per <= 0) { for (double t : c) if (t > t0 && t < t1) out.push_back(t); }
    else {
      long k0 = std::max(0L, (long)std::floor(t0 / per) - 1);
      for (long k = k0; k * per < t1 && (k - k0) < 10000000; ++k) {
        for (double tc : c) { double t = tc + k * per; if (t > t0 && t < t1) out.push_back(t); }
        if (k >= 1 && k * per > t0 && k * per < t1) out.push_back(k * per);  // wrap of `t mod period` may jump
      }
    }
  } else if (s.kind == CH_SRC_SIN) {
    const double te = par[6] / par[2];
    if (par[3] > t0 && par[3] < t1) out.push_back(par[3]);
    if (std::isfinite(te) && te > t0 && te < t1) out.push_back(te);
  }
}

// Does the source VALUE jump at t, or is t only a corner (slope discontinuity)?  Same rule as oracle.cpp source_jumps_at.
bool source_jumps_at(const HSource& s, const double* par, double t) {
  double amp = 0.0;
  if (s.kind == CH_SRC_PWL) for (double y : s.ys) amp = std::max(amp, std::fabs(y));
  else if (s.kind == CH_SRC_PULSE) amp = std::max(std::fabs(par[0]), std::fabs(par[1]));
  else if (s.kind == CH_SRC_SIN) amp = std::fabs(par[0]) + std::fabs(par[1]);
  const double a = source_value(s, par, 0.0, std::nextafter(t, -INFINITY), 1), b = source_value(s, par, 0.0, t, 1);
  return std::fabs(a - b) > 1e-9 * amp;
}

// (time, code) of one source's break points in (t0, t1): code < 0 = the value jumps, else the length of the segment that starts there
void source_breakpoint_codes(const HSource& s, const double* par, double t0, double t1, std::vector<std::pair<double, double>>& out) {
  const bool restart_all = std::getenv("CEDARHIP_BP_RESTART_ALL") != nullptr;   // the policy of rounds 1-2 (A/B switch; read per call: the tests flip it)
  std::vector<double> own;
  source_breakpoints(s, par, t0, t1, own);
  std::sort(own.begin(), own.end());
  for (size_t j = 0; j < own.size(); ++j) {
    const double seg = (j + 1 < own.size() ? own[j + 1] : t1) - own[j];
    out.emplace_back(own[j], (restart_all || source_jumps_at(s, par, own[j])) ? -1.0 : seg);
  }
}
// sorted unique times with merged codes (a jump wins, otherwise the shortest segment); t1 closes the list
void merge_breakpoints(std::vector<std::pair<double, double>>& pts, double t1, std::vector<double>& bps, std::vector<double>& bpc) {
  pts.emplace_back(t1, -1.0);
  std::sort(pts.begin(), pts.end());
  bps.clear(); bpc.clear();
  for (const auto& pt : pts) {
    if (!bps.empty() && bps.back() == pt.first) { bpc.back() = (bpc.back() < 0 || pt.second < 0) ? -1.0 : std::min(bpc.back(), pt.second); continue; }
    bps.push_back(pt.first); bpc.push_back(pt.second);
  }
}

// variable-coefficient BDF helpers: tau[0] = t_new, tau[1..] history (newest first)
void bdf_coeffs(const double* tau, int k, double* alpha) {
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
void extrap_weights(const double* tau, int np, double* w) {
  for (int j = 1; j <= np; ++j) { double v = 1; for (int i = 1; i <= np; ++i) if (i != j) v *= (tau[0] - tau[i]) / (tau[j] - tau[i]); w[j] = v; }
}

// All device buffers of a circuit are carved out of a few large allocations: the Newton kernel's
// prologue touches a dozen different arrays, and with one hipMalloc per array every launch paid a
// cold translation miss per array (measured: prologue 9 us -> see profiles/r01_notes.md).
struct Arena {
  std::vector<char*> chunks; size_t used = 0, cap = 0;
  static constexpr size_t CHUNK = 32u << 20;
  ~Arena() { for (char* c : chunks) (void)hipFree(c); }
  void* take(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes > CHUNK / 2) { char* p = nullptr; if (hipMalloc((void**)&p, bytes) != hipSuccess) return nullptr; chunks.insert(chunks.begin(), p); return p; }
    if (chunks.empty() || used + bytes > cap) { char* p = nullptr; if (hipMalloc((void**)&p, CHUNK) != hipSuccess) return nullptr; chunks.push_back(p); used = 0; cap = CHUNK; }
    void* r = chunks.back() + used; used += bytes; return r;
  }
};
static thread_local Arena* g_arena = nullptr;  // set while a circuit builds / rebuilds its buffers
// Every entry point that may (re)allocate device buffers of a circuit opens one of these: allocations made inside the
// call come from THAT circuit's arena and the pointer never outlives the call (a stale pointer would let a later call on
// another circuit carve its buffers out of this circuit's arena, which is freed with this circuit).
struct ArenaScope {
  Arena* prev;
  explicit ArenaScope(Arena* a) : prev(g_arena) { g_arena = a; }
  ~ArenaScope() { g_arena = prev; }
  ArenaScope(const ArenaScope&) = delete; ArenaScope& operator=(const ArenaScope&) = delete;
};

template <class T>
struct DevBuf {
  T* p = nullptr; size_t n = 0; bool owned = false;
  ~DevBuf() { if (p && owned) (void)hipFree(p); }
  hipError_t alloc(size_t count) {
    if (p && count <= n && count > 0) { return hipSuccess; }  // reuse
    if (p && owned) (void)hipFree(p);
    p = nullptr; n = count;
    const size_t bytes = std::max<size_t>(1, count) * sizeof(T);
    if (g_arena) { p = (T*)g_arena->take(bytes); owned = false; return p ? hipSuccess : hipErrorOutOfMemory; }
    owned = true;
    return hipMalloc((void**)&p, bytes);
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t st) {
    hipError_t e = hipSuccess;
    if (h.size() > n || !p) e = alloc(h.size());
    if (e != hipSuccess) return e;
    if (h.empty()) return hipSuccess;
    e = hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
  }
};

constexpr int NSLOT = 8;  // 7 history points + 1 candidate

}  // namespace

struct ch_result {
  std::vector<double> times, values, final_state;
  std::vector<int32_t> pts;   // per saved row: newest saved points (this row included) the step's dense-output polynomial runs through; 0 = none
  const double* dev_values = nullptr; int64_t dev_n = 0;   // the same values still in HBM ([n_obs][n_times][n_samples]), owned by the circuit, or null
  ch_stats stats;
  int status = CH_OK;
  int n_obs = 0, S = 1;
};

struct ch_circuit {
  Arena arena;  // declared first: destroyed last, after every DevBuf that points into it
  ch_ctx* ctx = nullptr;
  // ---- description ----
  int n_nodes = 0;
  std::vector<HDev> dev;
  std::vector<HSource> src;
  std::vector<std::vector<double>> model;
  double temp = 27, gmin = 1e-12, scale = 1;
  std::vector<int> slot_kind, slot_a, slot_b, obs_kind, obs_index;
  Analysis A;
  // ---- samples ----
  int S = 1;
  std::vector<std::vector<double>> slot_val;  // [n_slot][S] or empty
  bool dirty = true;
  int Spar = 1, Ssrc = 1, Smos = 1, Sgmin = 1;
  std::vector<double> h_src_dc, h_src_par;  // [Ssrc][nsrc], [Ssrc][nsrc][8]
  std::vector<int> needed_src;              // sources that define a known node or feed a surviving device (others, e.g. merged 0 V ammeters, are never evaluated)
  std::vector<int> dev_src;                 // sources whose value the kernels read (referenced by a surviving V / I device)
  int n_dev_src() const { return std::max<int>(1, (int)dev_src.size()); }
  std::vector<int> mos_cls;                 // [n_mos]
  int n_cls = 0;
  // ---- device buffers ----
  DevBuf<int> d_comp_class, d_comp_uofs, d_comp_dofs, d_gl_ptr, d_dkind, d_dterm, d_dsrc, d_dcls, d_dhdev, d_obs_unk, d_moscls_inst, d_slot_tab, d_unk_obs;
  DevBuf<unsigned long long> d_stamps;
  std::vector<int> obs_primary;  // per observable: the observable whose device row it shares (itself if primary)
  DevBuf<int> d_mc_ofs, d_mc_n, d_mc_list, d_dcls_local;
  DevBuf<BlockMeta> d_bmeta;
  std::vector<ClassMeta> h_cms;
  int block_threads = 64, lu_variant = 16, max_mc = 0;
  bool wide_split = false; int wide_l = 0, wide_other = 0;   // class 0: its large compiled devices are evaluated in two halves; lanes per half; unsplit slots
  bool host_reduce = true;      // block outputs land in mapped host memory and the host reduces them
  BlockOut* h_out = nullptr;    // mapped pinned [n_comp*S]
  size_t h_out_n = 0;
  DevBuf<ClassMeta> d_classes;
  DevBuf<uint16_t> d_gl_src;
  DevBuf<double> d_rate;
  DevBuf<unsigned char> d_perm;   // pivot order of every block's register LU (NewtonArgs::perm)
  DevBuf<double> d_dpar, d_dmult, d_mosp, d_kv, d_srcv, d_gmin, d_X, d_Q, d_dumpA, d_dumpF, d_dumpQ, d_dumpC, d_dumpG, d_dumpF0, d_temp, d_omega, d_xac, d_psd;
  DevBuf<int> d_noise_a, d_noise_b, d_noise_h, d_acfail;
  DevBuf<double> d_noise_pwr, d_noise_exp;
  DevBuf<double> d_vapar, d_vacache;
  DevBuf<int> d_dvac, d_va_mod, d_va_pofs, d_va_cofs;   // constant blocks of the Verilog-A instances: offset per flattened device; setup work list
  int n_va_inst = 0; size_t vac_total = 0; long vac_stride_ = 0;
  std::vector<double> va_par;   // parameter blocks of the Verilog-A instances
  int Stemp = 1, Sva = 1;
  double ac_scale = 0.0;        // eval_sources adds ac_scale*|ac| to every source value (AC right-hand side)  // (d_srcv unused: source values share d_kv)
  DevBuf<unsigned char> d_dmask, d_active;
  DevBuf<BlockOut> d_out;
  DevBuf<Summary> d_sum;
  Summary* h_sum = nullptr;     // pinned
  double* h_stage = nullptr;    // pinned staging for kv/srcv uploads
  size_t h_stage_n = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  size_t lds_bytes = 0, lds_doubles_fixed = 0, lds_extra_bytes = 0;
  NewtonArgs base;              // structure pointers filled once
  // ---- sparse path (blocks too large for LDS) ----
  int path = 1;                 // 1 = fused block kernel, 2 = sparse level-scheduled LU
  // A coupled array behind a border of one or two unknowns (supply rails with a series resistance): the same description analysed
  // with tearing (ch_analysis.hpp) — independent blocks + border replicas.  It only ever runs transients on the device-resident
  // stepper, started from this circuit's (sparse-path) operating point; everything else stays on this circuit.
  std::unique_ptr<ch_circuit> torn_c;
  bool is_torn = false;
  std::string torn_note;
  std::vector<int> dsrc_host;               // per flattened device: source slot (or Verilog-A parameter offset), as uploaded
  std::vector<double> h_dpar0, h_dmult0;   // main parameter and multiplicity of every device as uploaded (sample 0)
  SparsePlan plan[2];           // [0] DC (alpha0 = 0), [1] transient
  struct PlanDev { DevBuf<int> prow, pcol, a2lu, diag_pos, lvl_ptr, lvl_rows, ulvl_ptr, ulvl_rows, lrow_ptr, l_pos, l_k, l_upd_ptr, upd_dst, upd_src, urow_ptr, u_pos, u_col;
                   DevBuf<int> lu2a, la_pos, la_diag, lb_dst, lb_sptr, lb_l, lb_u, lb_d, fl_rows, bl_rows; DevBuf<double> LUv, Lv;
                   DevBuf<int> s3_blob, s3_ptr, s3_topa, s3_topr; DevBuf<double> s3_schur, s3_xT, s3_base, s3_sum; DevBuf<unsigned> s3_cnt; bool s3 = false; } plan_dev[2];
  DevBuf<int> sp_dflag;
  DevBuf<double> sp_part;   // [S][8][SP_NP] per-workgroup partial reductions of the O(n) passes (ch_sparse.hpp)
  DevBuf<double> sp_hpart, sp_hrow;   // slices of the heavy assembly items [S][items][SP_HB][2] and of the heavy rows of the charge update [S][rows][SP_RB]
  DevBuf<int> sp_rowptr, sp_colidx, sp_mat_gptr, sp_mat_gsrc, sp_vec_gptr, sp_vec_gsrc, sp_heavy_mat, sp_heavy_vec, sp_heavy_rows;
  int n_heavy_mat = 0, n_heavy_vec = 0, n_heavy_rows = 0;
  DevBuf<double> sp_stage, sp_Aval, sp_Cval, sp_F, sp_Q, sp_rhs, sp_y, sp_dx, sp_xcur, sp_xpred, sp_hq, sp_w, sp_qn;
  std::vector<int> h_rowptr, h_colidx;
  double* h_red = nullptr; int* h_flag = nullptr;  // mapped pinned: [S][8], [S][2]
  size_t h_red_n = 0;
  std::vector<double> sp_rate_v; std::vector<int> sp_status_v;  // per sample: last Newton rate, status of the last solve
  double sp_rate = 1.0;
  // stats
  double device_ms = 0; long n_launch = 0, n_timed = 0;
  double prof_launch = 0, prof_wait = 0, prof_reduce = 0;  // host seconds inside run_newton (CEDARHIP_HOST_PROFILE)
  bool host_profile = std::getenv("CEDARHIP_HOST_PROFILE") != nullptr;
  long time_every = std::getenv("CEDARHIP_TIME_EVERY") ? std::max(1L, std::atol(std::getenv("CEDARHIP_TIME_EVERY"))) : 8;  // device_ms sums the sampled launches only

  std::string& err() { return ctx->err; }
  void set_err(const std::string& s) { ctx->err = s; }

  ~ch_circuit() {
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);  // a polled launch may still be retiring
    if (g_arena == &arena) g_arena = nullptr;
    if (h_sum) (void)hipHostFree(h_sum);
    if (h_out) (void)hipHostFree(h_out);
    if (h_red) (void)hipHostFree(h_red);
    if (h_flag) (void)hipHostFree(h_flag);
    if (h_stage) (void)hipHostFree(h_stage);
    if (h_act) (void)hipHostFree(h_act);
    if (h_scale) (void)hipHostFree(h_scale);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }

  // ------------------------------------------------------------------------------------------
  int upload_structure() {
    g_arena = &arena;
    hipStream_t st = ctx->stream;
    std::vector<ClassMeta> cms; std::vector<int> blob;
    int max_slots = 0;
    for (size_t ci = 0; ci < A.classes.size(); ++ci) {
      const CompClass& c = A.classes[ci];
      ClassMeta m; std::memset(&m, 0, sizeof(m));
      m.nc = c.nc; m.ndev = c.ndev; m.nonlinear = c.nonlinear ? 1 : 0;
      m.n_mat_src = (int)c.mat_src.size(); m.n_vec_src = (int)c.vec_src.size();
      // one contiguous blob per class, copied verbatim into LDS: mat_ptr | vec_ptr | slots | mat_src | vec_src
      while (blob.size() & 3) blob.push_back(0);   // class blobs start on 16-byte boundaries (the kernel copies them with 16-byte loads)
      m.blob_ofs = (int)blob.size();
      blob.insert(blob.end(), c.mat_ptr.begin(), c.mat_ptr.end());
      blob.insert(blob.end(), c.vec_ptr.begin(), c.vec_ptr.end());
      int rep = -1;
      for (int k = 0; k < A.n_comp; ++k) if (A.comp_class[k] == (int)ci) { rep = k; break; }
      int ns = 0;
      // lane slots, expensive devices first so that they share wavefronts: compiled Verilog-A devices take one lane
      // per unknown terminal (direction-parallel duals: lane j computes column j of the stamp Jacobians; derivatives
      // with respect to known nodes are never gathered), MOSFETs and all other devices one lane each
      // A block whose compiled devices are large models (BSIM-CMG class) evaluates them in two halves on separate wavefronts:
      // bit 29 of a slot = split, bit 30 = which half (0: resistive sums and dI/dV, 1: charge sums and dQ/dV); the first halves
      // fill whole wavefronts (padded with idle slots) so that a wavefront never runs both code paths
      {
        std::vector<int> lanes;
        bool big = false;
        for (int d = 0; d < c.ndev; ++d) {
          const EDev& e = A.edev[A.comp_dofs[rep] + d];
          if (e.kind != K_VA) continue;
          if (va_gen::MODULES[dev[e.hdev].ipar[0]].n_params >= 64) big = true;
          bool first = true;
          for (int j = 0; j < e.nt; ++j) if (e.term[j] >= 0) { lanes.push_back((d << 4) | (first ? 8 : 0) | j); first = false; }
        }
        const bool split = big && std::getenv("CEDARHIP_VA_NOSPLIT") == nullptr;
        if (ci == 0) { wide_split = split; wide_l = (int)lanes.size(); wide_other = 0; }
        if (!split) { for (int v : lanes) { blob.push_back(v); ++ns; } }
        else {
          for (int v : lanes) { blob.push_back(v | (1 << 29)); ++ns; }
          while (ns & 63) { blob.push_back(-1); ++ns; }
          for (int v : lanes) { blob.push_back(v | (1 << 29) | (1 << 30)); ++ns; }
        }
      }
      for (int d = 0; d < c.ndev; ++d) if (A.edev[A.comp_dofs[rep] + d].kind == K_MOS) { blob.push_back(d << 4); ++ns; if (ci == 0) ++wide_other; }
      for (int d = 0; d < c.ndev; ++d) { const int kd = A.edev[A.comp_dofs[rep] + d].kind; if (kd != K_MOS && kd != K_VA) { blob.push_back(d << 4); ++ns; if (ci == 0) ++wide_other; } }
      m.nslots = ns;
      std::vector<uint16_t> h16(c.mat_src); h16.insert(h16.end(), c.vec_src.begin(), c.vec_src.end());
      if (h16.size() & 1) h16.push_back(0);
      for (size_t i = 0; i < h16.size(); i += 2) blob.push_back((int)((uint32_t)h16[i] | ((uint32_t)h16[i + 1] << 16)));
      // gather work list of the register-LU variants: structural non-zeros of A/C and all diagonals, then the rows of F/Q;
      // item = {first source (index into mat_src|vec_src), (sources << 16) | vector flag << 15 | entry}
      if (((int)blob.size() - m.blob_ofs) & 1) blob.push_back(0);
      m.wl_ofs = (int)blob.size() - m.blob_ofs;
      if (c.nc <= 64 && !c.vec_ptr.empty()) {
        // heaviest items first: when there are more items than lanes, the second pass of a wave holds only the lightest ones
        std::vector<std::array<int, 3>> items;   // {sources, first source, code}
        for (int i = 0; i < c.nc; ++i) {
          const int start = c.vec_ptr[i], cnt = c.vec_ptr[i + 1] - start;
          items.push_back({cnt, m.n_mat_src + start, (int)(0x8000u | (uint32_t)i)});
        }
        for (int e = 0; e < c.nc * c.nc; ++e) {
          const int start = c.mat_ptr[e], cnt = c.mat_ptr[e + 1] - start;
          if (cnt == 0 && e / c.nc != e % c.nc) continue;
          items.push_back({cnt, start, e});
        }
        std::stable_sort(items.begin(), items.end(), [](const std::array<int, 3>& x, const std::array<int, 3>& y) { return x[0] > y[0]; });
        for (const auto& it : items) { blob.push_back(it[1]); blob.push_back((int)(((uint32_t)it[0] << 16) | (uint32_t)it[2])); ++m.n_work; }
      }
      while ((blob.size() - m.blob_ofs) & 3) blob.push_back(0);
      m.blob_ints = (int)blob.size() - m.blob_ofs;
      if (std::getenv("CEDARHIP_DEBUG_BLOB")) std::fprintf(stderr, "[blob] class %zu: nc %d ndev %d slots %d mat_src %d vec_src %d work %d blob_ints %d\n", ci, m.nc, m.ndev, m.nslots, m.n_mat_src, m.n_vec_src, m.n_work, m.blob_ints);
      max_slots = std::max(max_slots, m.nslots);
      cms.push_back(m);
    }
    block_threads = std::min(256, std::max(64, ((max_slots + 63) / 64) * 64));
    lu_variant = A.max_nc <= 8 ? 8 : (A.max_nc <= 12 ? 12 : (A.max_nc <= 16 ? 16 : (A.max_nc <= 32 ? 32 : 0)));
    if (A.wide) lu_variant = A.max_nc <= 16 ? 16 : 0;  // wide (Verilog-A) stamp records: two instantiations only
    std::vector<int> dkind, dterm, dhdev;
    std::vector<int>& dsrc = dsrc_host; dsrc.clear();
    dev_src.clear();
    { std::vector<int> slot_of(src.size(), -1);
      for (const EDev& e : A.edev) {
        dkind.push_back(e.kind); for (int k = 0; k < NTERM; ++k) dterm.push_back(e.term[k]); dhdev.push_back(e.hdev);
        int si = e.kind == K_VA ? dev[e.hdev].ipar[1] : 0;
        if (e.src >= 0) { if (slot_of[e.src] < 0) { slot_of[e.src] = (int)dev_src.size(); dev_src.push_back(e.src); } si = slot_of[e.src]; }
        dsrc.push_back(si);
      }
      std::vector<char> need(src.size(), 0);
      for (int si : dev_src) need[si] = 1;
      for (const KnownDef& kd : A.known) for (auto& tm : kd.terms) need[tm.first] = 1;
      needed_src.clear();
      for (size_t i = 0; i < src.size(); ++i) if (need[i]) needed_src.push_back((int)i); }
    { // constant blocks (va_gen::setup) of the compiled Verilog-A instances
      std::vector<int> dvac(A.edev.size(), 0), vmod, vpofs, vcofs;
      vac_total = 0;
      for (size_t i = 0; i < A.edev.size(); ++i) {
        const EDev& e = A.edev[i];
        if (e.kind != K_VA) continue;
        const int mod = dev[e.hdev].ipar[0];
        dvac[i] = (int)vac_total; vmod.push_back(mod); vpofs.push_back(dev[e.hdev].ipar[1]); vcofs.push_back((int)vac_total);
        vac_total += (size_t)va_gen::N_CACHE[mod];
      }
      n_va_inst = (int)vmod.size();
      if (vmod.empty()) { vmod.push_back(0); vpofs.push_back(0); vcofs.push_back(0); }
      HIPCHK(d_dvac.upload(dvac, st)); HIPCHK(d_va_mod.upload(vmod, st)); HIPCHK(d_va_pofs.upload(vpofs, st)); HIPCHK(d_va_cofs.upload(vcofs, st));
    }
    std::vector<unsigned char> dm(A.n_unk, 0);
    for (int u = 0; u < A.n_unk; ++u) dm[u] = (A.diff_mask[u] ? 1 : 0) | (A.unk_mna[u] >= n_nodes ? 2 : 0) | (A.replica[u] ? 4 : 0);   // bit 2: border replica outside block 0 (not counted in norms)
    std::vector<int> obs_unk, unk_obs(A.n_unk, -1);
    obs_primary.clear();
    for (size_t o = 0; o < obs_kind.size(); ++o) {
      int u = -1;
      if (obs_kind[o] == 0) u = A.node_unknown[obs_index[o]];
      else { int b = dev[obs_index[o]].branch; u = b >= 0 ? A.branch_unknown[b] : -1; }
      obs_unk.push_back(u);
      int prim = (int)o;
      if (u >= 0) { if (unk_obs[u] < 0) unk_obs[u] = (int)o; else prim = unk_obs[u]; }
      obs_primary.push_back(prim);
    }
    HIPCHK(d_unk_obs.upload(unk_obs, st));
    { std::vector<unsigned long long> z(8, 0ull); HIPCHK(d_stamps.upload(z, st)); }
    h_cms = cms;
    HIPCHK(d_classes.upload(cms, st)); HIPCHK(d_gl_ptr.upload(blob, st));
    HIPCHK(d_comp_class.upload(A.comp_class, st)); HIPCHK(d_comp_uofs.upload(A.comp_uofs, st)); HIPCHK(d_comp_dofs.upload(A.comp_dofs, st));
    HIPCHK(d_dkind.upload(dkind, st)); HIPCHK(d_dterm.upload(dterm, st)); HIPCHK(d_dsrc.upload(dsrc, st)); HIPCHK(d_dhdev.upload(dhdev, st));
    HIPCHK(d_dmask.upload(dm, st)); HIPCHK(d_obs_unk.upload(obs_unk, st));
    HIPCHK(d_sum.alloc(1));
    HIPCHK(hipHostMalloc((void**)&h_sum, sizeof(Summary), hipHostMallocMapped));
    HIPCHK(hipEventCreate(&ev0)); HIPCHK(hipEventCreate(&ev1));
    lds_doubles_fixed = 0; lds_extra_bytes = 0;
    for (size_t ci = 0; ci < A.classes.size(); ++ci) {
      const CompClass& c = A.classes[ci];
      lds_doubles_fixed = std::max(lds_doubles_fixed, (size_t)c.ndev * A.stride() + (size_t)c.nc * (c.nc + 1) + (size_t)c.nc * c.nc + 12 * (size_t)c.nc);
      lds_extra_bytes = std::max(lds_extra_bytes, ((size_t)cms[ci].blob_ints + 64) * 4 + 16);  // blob + the block's MOS class list
    }
    return CH_OK;
  }

  // value of slot `kind` for sample s or the base value
  bool slot_set(int i) const { return !slot_val[i].empty(); }

  // ------------------------------------------------------------------------------------------
  // (Re)build all per-sample parameter tables: remake(prob, p = sim) for every sample at once.
  int finalize_params() {
    if (!dirty) return CH_OK;
    g_arena = &arena;
    hipStream_t st = ctx->stream;
    const int nslot = (int)slot_kind.size();
    {  // the state rings alone take 2 * NSLOT * S * n_unk doubles: refuse a sample count the device cannot hold before any
       // host table is sized by it (the caller gets an error code, not a std::bad_alloc or a half-built circuit)
      size_t free_b = 0, total_b = 0;
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      const double need = 2.0 * NSLOT * (double)S * (double)std::max(1, A.n_unk) * sizeof(double) + 64.0 * (double)S * (double)std::max(1, A.n_comp);
      if (need > 0.9 * (double)total_b) { set_err("sample count does not fit the device: " + std::to_string(S) + " samples of " + std::to_string(A.n_unk) + " unknowns need " + std::to_string((long long)(need / 1048576.0)) + " MiB for the state rings"); return CH_ERR_NOMEM; }
    }
    bool any_par = false, any_src = false, any_mos = false, any_gmin = false;
    for (int i = 0; i < nslot; ++i) if (slot_set(i)) {
      switch (slot_kind[i]) {
        case CH_SLOT_DEV_PAR: if (dev[slot_a[i]].kind == CH_DEV_MOS) any_mos = true; else any_par = true; break;
        case CH_SLOT_DEV_MULT: any_par = true; break;
        case CH_SLOT_MODEL_PAR: case CH_SLOT_TEMP: any_mos = true; break;
        case CH_SLOT_SRC_DC: case CH_SLOT_SRC_PAR: any_src = true; break;
        case CH_SLOT_GMIN: any_gmin = true; break;
      }
    }
    Spar = any_par ? S : 1; Ssrc = any_src ? S : 1; Smos = any_mos ? S : 1; Sgmin = any_gmin ? S : 1;
    const int nh = (int)dev.size(), nsrc = (int)src.size();
    // linear device parameters and multipliers
    std::vector<double> hpar((size_t)nh * Spar), hmult((size_t)nh * Spar);
    for (int d = 0; d < nh; ++d) for (int s = 0; s < Spar; ++s) { hpar[(size_t)d * Spar + s] = dev[d].par[0]; hmult[(size_t)d * Spar + s] = dev[d].mult; }
    // sources
    h_src_dc.assign((size_t)Ssrc * nsrc, 0.0); h_src_par.assign((size_t)Ssrc * nsrc * CH_SRC_NPAR, 0.0);
    for (int s = 0; s < Ssrc; ++s) for (int i = 0; i < nsrc; ++i) { h_src_dc[(size_t)s * nsrc + i] = src[i].dc; for (int k = 0; k < CH_SRC_NPAR; ++k) h_src_par[((size_t)s * nsrc + i) * CH_SRC_NPAR + k] = src[i].par[k]; }
    std::vector<double> hg(Sgmin, gmin);
    bool any_temp = false;
    for (int i = 0; i < nslot; ++i) if (slot_set(i) && slot_kind[i] == CH_SLOT_TEMP) any_temp = true;
    Stemp = any_temp ? S : 1;
    std::vector<double> htemp(Stemp, temp);
    for (int i = 0; i < nslot; ++i) if (slot_set(i)) {
      const int a = slot_a[i], b = slot_b[i];
      for (int s = 0; s < S; ++s) {
        const double v = slot_val[i][s];
        switch (slot_kind[i]) {
          case CH_SLOT_DEV_PAR: if (dev[a].kind != CH_DEV_MOS && b == 0) hpar[(size_t)a * Spar + s] = v; break;
          case CH_SLOT_DEV_MULT: hmult[(size_t)a * Spar + s] = v; break;
          case CH_SLOT_SRC_DC: h_src_dc[(size_t)s * nsrc + a] = v; if (src[a].kind == CH_SRC_DC) h_src_par[((size_t)s * nsrc + a) * CH_SRC_NPAR] = v; break;
          case CH_SLOT_SRC_PAR: h_src_par[((size_t)s * nsrc + a) * CH_SRC_NPAR + b] = v; break;
          case CH_SLOT_GMIN: hg[s] = v; break;
          case CH_SLOT_TEMP: htemp[s] = v; break;
          default: break;
        }
      }
    }
    h_dpar0.assign(nh, 0.0); h_dmult0.assign(nh, 1.0);
    for (int i = 0; i < nh; ++i) { h_dpar0[i] = hpar[(size_t)i * Spar]; h_dmult0[i] = hmult[(size_t)i * Spar]; }
    HIPCHK(d_dpar.upload(hpar, st)); HIPCHK(d_dmult.upload(hmult, st)); HIPCHK(d_gmin.upload(hg, st)); HIPCHK(d_temp.upload(htemp, st));
    {  // Verilog-A parameter blocks: one copy, or one per sample when a CH_SLOT_VA_PAR slot is set
      bool any_va = false;
      for (int i = 0; i < nslot; ++i) if (slot_set(i) && slot_kind[i] == CH_SLOT_VA_PAR) any_va = true;
      Sva = any_va ? S : 1;
      const size_t nvp = std::max<size_t>(1, va_par.size());
      std::vector<double> vp(nvp * Sva, 0.0);
      for (int s = 0; s < Sva; ++s) std::copy(va_par.begin(), va_par.end(), vp.begin() + (size_t)s * nvp);
      for (int i = 0; i < nslot; ++i) if (slot_set(i) && slot_kind[i] == CH_SLOT_VA_PAR) for (int s = 0; s < Sva; ++s) vp[(size_t)s * nvp + slot_a[i]] = slot_val[i][s];
      HIPCHK(d_vapar.upload(vp, st));
      // the bias-independent part of every instance, once per parameter / temperature change
      const int Svac = (Sva > 1 || Stemp > 1) ? S : 1;
      HIPCHK(d_vacache.alloc(std::max<size_t>(1, vac_total * (size_t)Svac)));
      if (n_va_inst > 0) {
        const long nthr = (long)n_va_inst * Svac;
        hipLaunchKernelGGL(va_setup_kernel, dim3((unsigned)((nthr + 63) / 64)), dim3(64), 0, st, n_va_inst, Svac, d_va_mod.p, d_va_pofs.p, d_va_cofs.p,
                           d_vapar.p, Sva > 1 ? (long)nvp : 0L, d_temp.p, Stemp, d_vacache.p, Svac > 1 ? (long)vac_total : 0L);
        HIPCHK(hipGetLastError());
      }
      vac_stride_ = Svac > 1 ? (long)vac_total : 0L;
    }
    // MOS classes: instances with identical (model, geometry, overriding slots) share a column
    const int nmos = (int)A.mos_hdev.size();
    mos_cls.assign(nmos, 0);
    std::map<std::vector<double>, int> cls_of;
    std::vector<int> cls_rep;
    for (int m = 0; m < nmos; ++m) {
      const HDev& d = dev[A.mos_hdev[m]];
      std::vector<double> key;
      key.push_back(d.ipar[0]);
      for (int k = 0; k < 7; ++k) key.push_back(std::isnan(d.par[k]) ? -1e300 : d.par[k]);
      // an overriding slot: which field, and its values in every sample — instances whose overrides are EQUAL (a global W / L delta
      // of a Monte-Carlo sweep gives every device its own slot, but devices of one geometry the same values) still share a column
      for (int i = 0; i < nslot; ++i) if (slot_set(i) && slot_kind[i] == CH_SLOT_DEV_PAR && slot_a[i] == A.mos_hdev[m]) {
        key.push_back(1e6 + slot_b[i]);
        key.insert(key.end(), slot_val[i].begin(), slot_val[i].end());
      }
      auto it = cls_of.find(key);
      if (it == cls_of.end()) { it = cls_of.insert({key, (int)cls_rep.size()}).first; cls_rep.push_back(m); }
      mos_cls[m] = it->second;
    }
    n_cls = (int)cls_rep.size();
    const long cols = (long)std::max(1, n_cls) * Smos;
    std::vector<double> table((size_t)B4I_COUNT * cols + 2, 0.0), col(B4I_COUNT);   // + padding: the kernel reads the columns in 16-byte pairs
    for (int c = 0; c < n_cls; ++c) {
      const int hd = A.mos_hdev[cls_rep[c]];
      for (int s = 0; s < Smos; ++s) {
        std::vector<double> card = model[dev[hd].ipar[0]];
        double ip[CH_DEV_NPAR]; for (int k = 0; k < CH_DEV_NPAR; ++k) ip[k] = dev[hd].par[k];
        double tc = temp;
        for (int i = 0; i < nslot; ++i) if (slot_set(i)) {
          const double v = slot_val[i][Smos > 1 ? s : 0];
          if (slot_kind[i] == CH_SLOT_MODEL_PAR && slot_a[i] == dev[hd].ipar[0]) card[slot_b[i]] = v;
          else if (slot_kind[i] == CH_SLOT_DEV_PAR && slot_a[i] == hd) ip[slot_b[i]] = v;
          else if (slot_kind[i] == CH_SLOT_TEMP) tc = v;
        }
        ip[CH_MOS_W] *= scale; ip[CH_MOS_L] *= scale;
        int rc = b4_pack(card.data(), ip, tc, col.data());
        if (rc != CH_OK) { set_err(rc == CH_ERR_UNSUPPORTED ? "BSIM4 card selects a sub-model the engine does not implement (rdsmod/rgatemod/rbodymod/igcmod/igbmod/trnqsmod/geomod != 0, diomod != 1, mobmod > 2, capmod not 0/2)" : "invalid MOS geometry or model card"); return rc; }
        for (int k = 0; k < B4I_COUNT; ++k) table[((size_t)c * Smos + s) * B4I_COUNT + k] = col[k];
      }
    }
    HIPCHK(d_mosp.upload(table, st));
    std::vector<int> dcls;
    for (const EDev& e : A.edev) dcls.push_back(e.mos >= 0 ? mos_cls[e.mos] : 0);
    HIPCHK(d_dcls.upload(dcls, st));
    {  // per block: the distinct MOS classes its devices use, and each device's index into that list
      std::vector<int> mc_ofs(A.n_comp), mc_n(A.n_comp), mc_list, dloc(A.edev.size(), 0);
      max_mc = 0;
      for (int k = 0; k < A.n_comp; ++k) {
        mc_ofs[k] = (int)mc_list.size();
        std::vector<int> seen;
        for (int d = 0; d < A.comp_ndev[k]; ++d) {
          const EDev& e = A.edev[A.comp_dofs[k] + d];
          if (e.kind == K_VA) { dloc[A.comp_dofs[k] + d] = dev[e.hdev].ipar[0]; continue; }
          if (e.mos < 0) continue;
          const int cl = mos_cls[e.mos];
          int j = (int)(std::find(seen.begin(), seen.end(), cl) - seen.begin());
          if (j == (int)seen.size()) seen.push_back(cl);
          dloc[A.comp_dofs[k] + d] = j;
        }
        mc_n[k] = (int)seen.size();
        mc_list.insert(mc_list.end(), seen.begin(), seen.end());
        max_mc = std::max(max_mc, mc_n[k]);
      }
      // (a block with more than 64 distinct MOSFET classes takes the sparse path: see the path decision below)
      std::vector<BlockMeta> bmv(A.n_comp);
      for (int k = 0; k < A.n_comp; ++k) {
        BlockMeta& b = bmv[k]; std::memset(&b, 0, sizeof(b));
        b.uofs = A.comp_uofs[k]; b.dofs = A.comp_dofs[k]; b.mc_n = mc_n[k]; b.mc_ofs = mc_ofs[k]; b.cm = h_cms[A.comp_class[k]];
        for (int j = 0; j < mc_n[k] && j < 8; ++j) b.mc[j] = mc_list[mc_ofs[k] + j];
      }
      HIPCHK(d_bmeta.upload(bmv, st));
      HIPCHK(d_mc_ofs.upload(mc_ofs, st)); HIPCHK(d_mc_n.upload(mc_n, st)); HIPCHK(d_mc_list.upload(mc_list, st)); HIPCHK(d_dcls_local.upload(dloc, st));
    }
    HIPCHK(d_moscls_inst.upload(mos_cls, st));
    // state ring and outputs
    const size_t slot_elems = (size_t)S * A.n_unk;
    HIPCHK(d_X.alloc(slot_elems * NSLOT)); HIPCHK(d_Q.alloc(slot_elems * NSLOT));
    HIPCHK(hipMemsetAsync(d_X.p, 0, slot_elems * NSLOT * sizeof(double), st));
    HIPCHK(hipMemsetAsync(d_Q.p, 0, slot_elems * NSLOT * sizeof(double), st));
    HIPCHK(d_out.alloc((size_t)A.n_comp * S));
    { std::vector<double> ones((size_t)A.n_comp * S, 1.0); HIPCHK(d_rate.upload(ones, st)); }
    HIPCHK(d_perm.alloc((size_t)A.n_comp * S * 16)); HIPCHK(hipMemsetAsync(d_perm.p, 0, (size_t)A.n_comp * S * 16, st));   // identity
    host_reduce = (size_t)A.n_comp * S <= 4096 && std::getenv("CEDARHIP_DEVICE_REDUCE") == nullptr;
    if (host_reduce && h_out_n < (size_t)A.n_comp * S) {
      if (h_out) { (void)hipHostFree(h_out); h_out = nullptr; h_out_n = 0; }
      HIPCHK(hipHostMalloc((void**)&h_out, (size_t)A.n_comp * S * sizeof(BlockOut), hipHostMallocMapped));
      std::memset(h_out, 0, (size_t)A.n_comp * S * sizeof(BlockOut));   // sequence numbers start at 1
      h_out_n = (size_t)A.n_comp * S;
    }
    HIPCHK(d_active.alloc((size_t)A.n_comp * S));
    const size_t need = (size_t)Ssrc * (A.known.size() + n_dev_src());
    HIPCHK(d_kv.alloc(need));  // [kv | srcv] contiguous: one upload per step
    if (need > h_stage_n) { if (h_stage) (void)hipHostFree(h_stage); HIPCHK(hipHostMalloc((void**)&h_stage, need * sizeof(double))); h_stage_n = need; }
    // argument template
    NewtonArgs& a = base;
    std::memset(&a, 0, sizeof(a));
    a.comp_class = d_comp_class.p; a.comp_uofs = d_comp_uofs.p; a.comp_dofs = d_comp_dofs.p; a.classes = d_classes.p;
    a.blob = d_gl_ptr.p; a.dkind = d_dkind.p; a.dterm = d_dterm.p; a.dsrc = d_dsrc.p; a.dcls = d_dcls.p; a.dhdev = d_dhdev.p;
    a.dpar = d_dpar.p; a.dmult = d_dmult.p; a.mosp = d_mosp.p; a.mos_cols = cols; a.kv = d_kv.p; a.srcv = d_kv.p + (size_t)Ssrc * A.known.size(); a.dmask = d_dmask.p;
    a.active = nullptr; a.gmin_s = d_gmin.p; a.vapar = d_vapar.p; a.va_stride = Sva > 1 ? (long)std::max<size_t>(1, va_par.size()) : 0; a.temp_s = d_temp.p; a.Stemp = Stemp;
    a.vacache = d_vacache.p; a.vac_stride = vac_stride_; a.dvac = d_dvac.p;
    a.n_comp = A.n_comp; a.S = S; a.Spar = Spar; a.Ssrc = Ssrc; a.Smos = Smos; a.Sgmin = Sgmin; a.nk = (int)A.known.size(); a.nsrc = n_dev_src();
    a.n_unk = A.n_unk; a.n_mos_cls = n_cls;
    a.X = d_X.p; a.Qh = d_Q.p; a.slot_stride = (long)slot_elems; a.out = host_reduce ? h_out : d_out.p;
    a.unk_obs = d_unk_obs.p; a.n_obs = (int)obs_kind.size();
    a.bmeta = d_bmeta.p; a.dcls_local = d_dcls_local.p; a.comp_mc_ofs = d_mc_ofs.p; a.comp_mc_n = d_mc_n.p; a.mc_list = d_mc_list.p; a.max_mc = max_mc;
    a.summary = h_sum; a.rate = d_rate.p; a.perm = d_perm.p;
#ifdef CH_STAMPS
    a.stamps = d_stamps.p;
#endif
    lds_bytes = (lds_doubles_fixed + A.known.size() + n_dev_src() + (size_t)max_mc * B4L_STRIDE) * sizeof(double) + lds_extra_bytes;
    lds_bytes = std::max(lds_bytes, (size_t)9 * block_threads * sizeof(double));  // scratch of the in-kernel reduction
    path = (lds_bytes > 150 * 1024 || A.max_nc > 64 || max_mc > 64 || A.force_sparse || std::getenv("CEDARHIP_FORCE_SPARSE") != nullptr) ? 2 : 1;
    if (path == 2) {
      int rcs = build_sparse_structure();
      if (rcs != CH_OK) return rcs;
      lds_bytes = 0;
    }
    // (the kernels' dynamic-LDS ceiling is raised once per device in ch_create: a per-circuit setting would be lowered again
    //  by the next, smaller circuit of the same process)
    HIPCHK(hipStreamSynchronize(st));
    dirty = false;
    return CH_OK;
  }

  // host evaluation of source and known-node values for sample set s at time t
  mutable std::vector<double> all_src;
  void eval_sources(double t, int mode, std::vector<double>& sv, std::vector<double>& kv) const {
    const int nsrc = (int)src.size(), nk = (int)A.known.size(), nds = n_dev_src();
    sv.assign((size_t)Ssrc * nds, 0.0); kv.assign((size_t)Ssrc * nk, 0.0);
    all_src.resize(std::max(1, nsrc));
    for (int s = 0; s < Ssrc; ++s) {
      for (int i : needed_src) all_src[i] = source_value(src[i], &h_src_par[((size_t)s * nsrc + i) * CH_SRC_NPAR], h_src_dc[(size_t)s * nsrc + i], t, mode) + ac_scale * src[i].ac;
      for (size_t j = 0; j < dev_src.size(); ++j) sv[(size_t)s * nds + j] = all_src[dev_src[j]];
      for (int k = 0; k < nk; ++k) { double v = 0; for (auto& tm : A.known[k].terms) v += tm.second * all_src[tm.first]; kv[(size_t)s * nk + k] = v; }
    }
  }
  std::vector<double> sv_buf, kv_buf;
  int set_sources(NewtonArgs& a, double t, int mode) {
    std::vector<double>& sv = sv_buf; std::vector<double>& kv = kv_buf;
    eval_sources(t, mode, sv, kv);
    if (Ssrc == 1 && kv.size() + sv.size() <= (size_t)KV_INLINE) {
      a.inline_vals = 1;
      for (size_t i = 0; i < kv.size(); ++i) a.vals_inline[i] = kv[i];
      for (size_t i = 0; i < sv.size(); ++i) a.vals_inline[kv.size() + i] = sv[i];
      return CH_OK;
    }
    a.inline_vals = 0;
    // the pinned staging buffer is reused every step: the stream sync at the end of each step protects it
    std::memcpy(h_stage, kv.data(), kv.size() * sizeof(double));
    std::memcpy(h_stage + kv.size(), sv.data(), sv.size() * sizeof(double));
    HIPCHK(hipMemcpyAsync(d_kv.p, h_stage, (kv.size() + sv.size()) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return CH_OK;
  }

  // launch the fused Newton kernel + reduction and wait for the summary
  // ------------------------------------------------------------------------------------------
  // Sparse path: CSR pattern + gather lists over the global unknown numbering
  int build_sparse_structure() {
    hipStream_t st = ctx->stream;
    const int n = A.n_unk, nd = (int)A.edev.size();
    std::vector<std::map<int, std::vector<int>>> rows(n);
    std::vector<std::vector<int>> vrows(n);
    for (int d = 0; d < nd; ++d) {
      const EDev& e = A.edev[d];
      bool vm[NTERM], mm[NTERM * NTERM]; kind_mask(e.kind, vm, mm, e.nt);
      const int stride = A.stride(), gofs = A.g_ofs(), gld = A.g_ld();
      for (int k = 0; k < NTERM; ++k) if (vm[k] && e.term[k] >= 0) vrows[e.term[k]].push_back(d * stride + k);
      for (int k = 0; k < NTERM; ++k) for (int j = 0; j < NTERM; ++j) if (mm[k * NTERM + j] && e.term[k] >= 0 && e.term[j] >= 0) rows[e.term[k]][e.term[j]].push_back(d * stride + gofs + k * gld + j);
    }
    for (int i = 0; i < n; ++i) rows[i][i];  // structural diagonal (gmin stepping, pivots)
    h_rowptr.assign(1, 0); h_colidx.clear();
    std::vector<int> mgp(1, 0), mgs, vgp(1, 0), vgs;
    for (int i = 0; i < n; ++i) {
      for (auto& kv : rows[i]) { h_colidx.push_back(kv.first); mgs.insert(mgs.end(), kv.second.begin(), kv.second.end()); mgp.push_back((int)mgs.size()); }
      h_rowptr.push_back((int)h_colidx.size());
      vgs.insert(vgs.end(), vrows[i].begin(), vrows[i].end()); vgp.push_back((int)vgs.size());
    }
    const size_t nnz = h_colidx.size();
    { std::vector<int> hm, hv;
      for (size_t i = 0; i < nnz; ++i) if (mgp[i + 1] - mgp[i] > SP_ASM_HEAVY) hm.push_back((int)i);
      for (int i = 0; i < n; ++i) if (vgp[i + 1] - vgp[i] > SP_ASM_HEAVY) hv.push_back(i);
      n_heavy_mat = (int)hm.size(); n_heavy_vec = (int)hv.size();
      HIPCHK(sp_heavy_mat.upload(hm, st)); HIPCHK(sp_heavy_vec.upload(hv, st));
      std::vector<int> hr;
      for (int i = 0; i < n; ++i) if (h_rowptr[i + 1] - h_rowptr[i] > 256) hr.push_back(i);
      n_heavy_rows = (int)hr.size();
      HIPCHK(sp_heavy_rows.upload(hr, st)); }
    HIPCHK(sp_rowptr.upload(h_rowptr, st)); HIPCHK(sp_colidx.upload(h_colidx, st)); HIPCHK(sp_mat_gptr.upload(mgp, st)); HIPCHK(sp_mat_gsrc.upload(mgs, st));
    HIPCHK(sp_vec_gptr.upload(vgp, st)); HIPCHK(sp_vec_gsrc.upload(vgs, st));
    HIPCHK(sp_stage.alloc((size_t)S * nd * A.stride())); HIPCHK(sp_Aval.alloc((size_t)S * nnz)); HIPCHK(sp_Cval.alloc((size_t)S * nnz));
    for (DevBuf<double>* b : {&sp_F, &sp_Q, &sp_rhs, &sp_y, &sp_dx, &sp_xcur, &sp_xpred, &sp_hq, &sp_w, &sp_qn}) HIPCHK(b->alloc((size_t)S * n));
    if (h_red_n < (size_t)S) {
      if (h_red) (void)hipHostFree(h_red);
      if (h_flag) (void)hipHostFree(h_flag);
      h_red = nullptr; h_flag = nullptr;
      HIPCHK(hipHostMalloc((void**)&h_red, (size_t)S * 8 * sizeof(double), hipHostMallocMapped)); HIPCHK(hipHostMalloc((void**)&h_flag, (size_t)S * 2 * sizeof(int), hipHostMallocMapped));
      h_red_n = (size_t)S;
    }
    { std::vector<int> z((size_t)S, 0); HIPCHK(sp_dflag.upload(z, st)); }
    sp_rate_v.assign(S, 1.0); sp_status_v.assign(S, 0);
    plan[0].valid = plan[1].valid = false;
    return CH_OK;
  }
  SparseDev sparse_dev(int which, int sm = 0) {
    SparseDev d; std::memset(&d, 0, sizeof(d));
    PlanDev& pd = plan_dev[which]; const SparsePlan& P = plan[which];
    const size_t n = A.n_unk, nnz = h_colidx.size(), nd = A.edev.size();
    d.rowptr = sp_rowptr.p; d.colidx = sp_colidx.p; d.mat_gptr = sp_mat_gptr.p; d.mat_gsrc = sp_mat_gsrc.p; d.vec_gptr = sp_vec_gptr.p; d.vec_gsrc = sp_vec_gsrc.p;
    d.prow = pd.prow.p; d.pcol = pd.pcol.p; d.a2lu = pd.a2lu.p; d.diag_pos = pd.diag_pos.p; d.lvl_ptr = pd.lvl_ptr.p; d.lvl_rows = pd.lvl_rows.p;
    d.ulvl_ptr = pd.ulvl_ptr.p; d.ulvl_rows = pd.ulvl_rows.p; d.lrow_ptr = pd.lrow_ptr.p; d.l_pos = pd.l_pos.p; d.l_k = pd.l_k.p; d.l_upd_ptr = pd.l_upd_ptr.p;
    d.upd_dst = pd.upd_dst.p; d.upd_src = pd.upd_src.p; d.urow_ptr = pd.urow_ptr.p; d.u_pos = pd.u_pos.p; d.u_col = pd.u_col.p;
    d.lu2a = pd.lu2a.p; d.la_pos = pd.la_pos.p; d.la_diag = pd.la_diag.p; d.lb_dst = pd.lb_dst.p; d.lb_sptr = pd.lb_sptr.p; d.lb_l = pd.lb_l.p; d.lb_u = pd.lb_u.p; d.lb_d = pd.lb_d.p;
    d.fl_rows = pd.fl_rows.p; d.bl_rows = pd.bl_rows.p;
    d.heavy_rows = sp_heavy_rows.p; d.n_heavy_rows = n_heavy_rows;
    d.heavy_mat = sp_heavy_mat.p; d.heavy_vec = sp_heavy_vec.p; d.n_heavy_mat = n_heavy_mat; d.n_heavy_vec = n_heavy_vec;
    d.Lv = pd.Lv.p ? pd.Lv.p + (size_t)sm * (size_t)std::max(0, P.nnz_lu) : nullptr;
    d.s = sm; d.xofs = (long)sm * (long)n;
    d.st_stage = (long)(nd * A.stride()); d.st_nnz = (long)nnz; d.st_lu = (long)std::max(0, P.nnz_lu); d.st_n = (long)n;
    d.stride = A.stride(); d.q_ofs = A.wide ? 8 : 4; d.c_ofs = A.wide ? 64 : 16; d.wide = A.wide ? 1 : 0;
    d.n = A.n_unk; d.nnz = (int)nnz; d.nnz_lu = P.nnz_lu; d.n_lvl = P.valid ? (int)P.lvl_ptr.size() - 1 : 0; d.n_ulvl = P.valid ? (int)P.ulvl_ptr.size() - 1 : 0; d.n_dev = (int)nd;
    // per-sample slices of the work arrays
    d.stage = sp_stage.p + (size_t)sm * nd * A.stride(); d.Aval = sp_Aval.p + (size_t)sm * nnz; d.Cval = sp_Cval.p + (size_t)sm * nnz;
    d.LUv = pd.LUv.p ? pd.LUv.p + (size_t)sm * (size_t)std::max(0, P.nnz_lu) : nullptr;
    d.F = sp_F.p + sm * n; d.Q = sp_Q.p + sm * n; d.rhs = sp_rhs.p + sm * n; d.y = sp_y.p + sm * n; d.dx = sp_dx.p + sm * n;
    d.xcur = sp_xcur.p + sm * n; d.xpred = sp_xpred.p + sm * n; d.hq = sp_hq.p + sm * n; d.w = sp_w.p + sm * n; d.qn = sp_qn.p + sm * n;
    d.red = h_red + (size_t)sm * 8; d.flag = h_flag + (size_t)sm * 2; d.dflag = sp_dflag.p + sm;
    return d;
  }
  // host analysis from the current numeric values of A (KLU-style: analyse once, refactor many times)
  int sparse_plan_from_current(int which, int sm = 0) {
    g_arena = &arena;
    hipStream_t st = ctx->stream;
    std::vector<double> aval(h_colidx.size());
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(aval.data(), sp_Aval.p + (size_t)sm * aval.size(), aval.size() * sizeof(double), hipMemcpyDeviceToHost));
    SparsePlan& P = plan[which];
    int rc = sparse_analyse(A.n_unk, h_rowptr, h_colidx, aval, P);
    if (rc != CH_OK) { set_err("sparse analysis: structurally singular Jacobian"); return rc; }
    PlanDev& pd = plan_dev[which];
    HIPCHK(pd.prow.upload(P.prow, st)); HIPCHK(pd.pcol.upload(P.pcol, st)); HIPCHK(pd.a2lu.upload(P.a2lu, st)); HIPCHK(pd.diag_pos.upload(P.diag_pos, st));
    HIPCHK(pd.lvl_ptr.upload(P.lvl_ptr, st)); HIPCHK(pd.lvl_rows.upload(P.lvl_rows, st)); HIPCHK(pd.ulvl_ptr.upload(P.ulvl_ptr, st)); HIPCHK(pd.ulvl_rows.upload(P.ulvl_rows, st));
    HIPCHK(pd.lrow_ptr.upload(P.lrow_ptr, st)); HIPCHK(pd.l_pos.upload(P.l_pos, st)); HIPCHK(pd.l_k.upload(P.l_k, st)); HIPCHK(pd.l_upd_ptr.upload(P.l_upd_ptr, st));
    HIPCHK(pd.upd_dst.upload(P.upd_dst, st)); HIPCHK(pd.upd_src.upload(P.upd_src, st)); HIPCHK(pd.urow_ptr.upload(P.urow_ptr, st)); HIPCHK(pd.u_pos.upload(P.u_pos, st)); HIPCHK(pd.u_col.upload(P.u_col, st));
    HIPCHK(pd.LUv.alloc((size_t)S * (size_t)P.nnz_lu));
    // subtree form: many independent subtrees under a small separator (ch_sparse_host.hpp SubtreePlan) — three launches per solve
    pd.s3 = P.sub.valid && std::getenv("CEDARHIP_SPARSE_NO_SUBTREE") == nullptr && std::getenv("CEDARHIP_SPARSE_ONE_WG") == nullptr;
    if (pd.s3) {
      const SubtreePlan& T = P.sub;
      HIPCHK(pd.s3_blob.upload(T.blob, st)); HIPCHK(pd.s3_ptr.upload(T.blob_ptr, st));
      { std::vector<int> ta = T.top_a_idx; if (ta.empty()) ta.push_back(-1); HIPCHK(pd.s3_topa.upload(ta, st)); }
      { std::vector<int> tr = T.top_rows;   // [pivot index | rhs index | dx index] of every top row: one load level in the kernel
        for (int k : T.top_rows) tr.push_back(P.prow[k]);
        for (int k : T.top_rows) tr.push_back(P.pcol[k]);
        if (tr.empty()) tr.push_back(0);
        HIPCHK(pd.s3_topr.upload(tr, st)); }
      HIPCHK(pd.s3_schur.alloc((size_t)S * (size_t)std::max(1, T.nT * T.nT + T.nT) * (size_t)T.n_groups));
      HIPCHK(pd.s3_xT.alloc((size_t)S * (size_t)std::max(1, T.nT)));
      HIPCHK(pd.s3_base.alloc((size_t)S * (size_t)std::max(1, T.nT * T.nT + T.nT)));
      HIPCHK(pd.s3_sum.alloc((size_t)S * (size_t)std::max(1, T.nT * T.nT + T.nT))); HIPCHK(pd.s3_cnt.alloc((size_t)S));
      const int lds3 = T.max_nv * 8 + T.max_blob * 4;
      HIPCHK(hipFuncSetAttribute((const void*)sp3_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(lds3, 64 * 1024)));
      HIPCHK(hipFuncSetAttribute((const void*)sp3_back_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, std::max(lds3, 64 * 1024)));
      P.wide_levels = false;
      return CH_OK;
    }
    if (P.wide_levels && std::getenv("CEDARHIP_SPARSE_ONE_WG") == nullptr) {
      std::vector<int> lu2a((size_t)P.nnz_lu, -1);
      for (size_t i = 0; i < P.a2lu.size(); ++i) lu2a[P.a2lu[i]] = (int)i;
      HIPCHK(pd.lu2a.upload(lu2a, st)); HIPCHK(pd.la_pos.upload(P.la_pos, st)); HIPCHK(pd.la_diag.upload(P.la_diag, st));
      HIPCHK(pd.lb_dst.upload(P.lb_dst, st)); HIPCHK(pd.lb_sptr.upload(P.lb_sptr, st)); HIPCHK(pd.lb_l.upload(P.lb_l, st)); HIPCHK(pd.lb_u.upload(P.lb_u, st)); HIPCHK(pd.lb_d.upload(P.lb_d, st));
      HIPCHK(pd.fl_rows.upload(P.fl_rows, st)); HIPCHK(pd.bl_rows.upload(P.bl_rows, st));
      HIPCHK(pd.Lv.alloc((size_t)S * (size_t)P.nnz_lu));
    } else P.wide_levels = false;
    return CH_OK;
  }
  // refactorisation + both triangular solves for the samples of `wl`: one workgroup per sample (chains, small systems), or
  // one launch per elimination level across the whole chip (few wide levels)
  void launch_lu_solve(int which, const int* wl, size_t n_work) {
    hipStream_t st = ctx->stream;
    const SparsePlan& P = plan[which];
    const SparseDev d = sparse_dev(which);
    if (plan_dev[which].s3) {
      PlanDev& pd = plan_dev[which]; const SubtreePlan& T = P.sub;
      Sp3Dev q; q.blob = pd.s3_blob.p; q.blob_ptr = pd.s3_ptr.p; q.top_a_idx = pd.s3_topa.p; q.top_rows = pd.s3_topr.p; q.schur = pd.s3_schur.p; q.xT = pd.s3_xT.p; q.top_base = pd.s3_base.p; q.top_sum = pd.s3_sum.p; q.top_cnt = pd.s3_cnt.p;
      q.n_groups = T.n_groups; q.nT = T.nT; q.max_nv = T.max_nv;
      const unsigned lds3 = (unsigned)(T.max_nv * 8 + T.max_blob * 4);
      hipLaunchKernelGGL(sp3_reset_kernel, dim3(1, (unsigned)n_work), dim3(64), 0, st, d, wl, q);
      hipLaunchKernelGGL(sp3_group_kernel, dim3((unsigned)T.n_groups, (unsigned)n_work), dim3(64), lds3, st, d, wl, q);
      hipLaunchKernelGGL(sp3_top_kernel, dim3(SP3_TOP_WG, (unsigned)n_work), dim3(256), 0, st, d, wl, q);
      hipLaunchKernelGGL(sp3_back_kernel, dim3((unsigned)T.n_groups, (unsigned)n_work), dim3(64), lds3, st, d, wl, q);
      n_launch += 4;
      return;
    }
    if (!P.wide_levels) { hipLaunchKernelGGL(sp_lu_solve_kernel, dim3(1, (unsigned)n_work), dim3(1024), 0, st, d, wl); return; }
    const unsigned ny = (unsigned)n_work;
    hipLaunchKernelGGL(sp2_scatter_kernel, dim3((unsigned)((P.nnz_lu + 255) / 256), ny), dim3(256), 0, st, d, wl);
    const int nl = (int)P.lvl_ptr.size() - 1, nul = (int)P.ulvl_ptr.size() - 1;
    for (int l = 0; l < P.n_rlvl; ++l) {
      const int nA = P.la_ptr[l + 1] - P.la_ptr[l], nB = P.lb_ptr[l + 1] - P.lb_ptr[l], nBh = P.lb_nheavy[l], nBl = nB - nBh;
      if (nA + nB == 0) continue;
      const int tb = (nA + nBl + 255) / 256, hb = (nBh + 3) / 4;
      hipLaunchKernelGGL(sp2_factor_level_kernel, dim3((unsigned)std::max(1, tb + hb), ny), dim3(256), 0, st, d, wl, P.la_ptr[l], nA, P.lb_ptr[l], nBl, nBh, tb);
    }
    for (int l = 0; l < nl; ++l) {
      const int nr = P.fl_ptr[l + 1] - P.fl_ptr[l], nh = P.fl_nheavy[l], nlg = nr - nh;
      const int tb = (nlg + 255) / 256, hb = nh;   // one workgroup per heavy row
      hipLaunchKernelGGL(sp2_fwd_level_kernel, dim3((unsigned)std::max(1, tb + hb), ny), dim3(256), 0, st, d, wl, P.fl_ptr[l], nlg, nh, tb);
    }
    for (int l = 0; l < nul; ++l) {
      const int nr = P.bl_ptr[l + 1] - P.bl_ptr[l];
      hipLaunchKernelGGL(sp2_bwd_level_kernel, dim3((unsigned)((nr + 255) / 256), ny), dim3(256), 0, st, d, wl, P.bl_ptr[l], nr);
    }
    n_launch += 1 + P.n_rlvl + nl + nul;
  }
  // Small-signal analyses see the circuit as dense blocks: the Jacobian blocks of the fused path, or — on the sparse path —
  // the whole system as one block per sample (up to 96 unknowns the complex LU runs in LDS, above that in a global workspace)
  DevBuf<BlockMeta> d_bmeta_all;
  int ac_ncomp() const { return path == 2 ? 1 : A.n_comp; }
  int ac_ds() const { return path == 2 ? A.n_unk : A.max_nc; }
  int ac_comp_of(int u) const { return path == 2 ? 0 : (int)(std::upper_bound(A.comp_uofs.begin(), A.comp_uofs.end(), u) - A.comp_uofs.begin()) - 1; }
  int ac_uofs(int comp) const { return path == 2 ? 0 : A.comp_uofs[comp]; }
  int ac_nc(int comp) const { return path == 2 ? A.n_unk : A.comp_nc[comp]; }
  int ac_dofs(int comp) const { return path == 2 ? 0 : A.comp_dofs[comp]; }
  int ac_ndev(int comp) const { return path == 2 ? (int)A.edev.size() : A.comp_ndev[comp]; }
  const BlockMeta* ac_bmeta() {
    if (path != 2) return d_bmeta.p;
    BlockMeta b; std::memset(&b, 0, sizeof(b)); b.uofs = 0; b.dofs = 0; b.cm.nc = A.n_unk; b.cm.ndev = (int)A.edev.size();
    std::vector<BlockMeta> v(1, b);
    g_arena = &arena;
    if (d_bmeta_all.upload(v, ctx->stream) != hipSuccess) return nullptr;
    return d_bmeta_all.p;
  }
  int sp_sync() {
    hipStream_t st = ctx->stream;
    hipError_t q = hipErrorNotReady;
    for (int spin = 0; spin < 200000 && q == hipErrorNotReady; ++spin) q = hipStreamQuery(st);
    if (q == hipErrorNotReady) q = hipStreamSynchronize(st);
    if (q != hipSuccess) { set_err(std::string("sparse path: ") + hipGetErrorString(q)); return CH_ERR_DEVICE; }
    return CH_OK;
  }
  // One Newton solve per sample (same contract as the fused kernel: reads the history ring, writes the candidate
  // slot).  Samples share the symbolic plan and the pivot order; every phase is queued for all active samples and
  // the host synchronises once per phase, so the number of round trips does not grow with the sample count.
  // device copies of a sample list / per-sample scales, staged through pinned memory (rewritten only after a stream sync)
  DevBuf<int> sp_act[3]; DevBuf<double> sp_scale;
  int* h_act = nullptr; double* h_scale = nullptr; size_t h_act_n = 0;
  int stage_list(int slot, const std::vector<int>& list) {
    if (h_act_n < (size_t)S) {
      if (h_act) (void)hipHostFree(h_act);
      if (h_scale) (void)hipHostFree(h_scale);
      HIPCHK(hipHostMalloc((void**)&h_act, 3 * (size_t)S * sizeof(int))); HIPCHK(hipHostMalloc((void**)&h_scale, (size_t)S * sizeof(double)));
      h_act_n = (size_t)S;
    }
    g_arena = &arena;
    HIPCHK(sp_act[slot].alloc((size_t)S));
    std::memcpy(h_act + (size_t)slot * S, list.data(), list.size() * sizeof(int));
    HIPCHK(hipMemcpyAsync(sp_act[slot].p, h_act + (size_t)slot * S, list.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return CH_OK;
  }
  int run_sparse(NewtonArgs a, const unsigned char* host_active, Summary& out) {
    hipStream_t st = ctx->stream;
    const int which = a.mode == MODE_DC ? 0 : 1;
    const int n = A.n_unk, nd = (int)A.edev.size(), nnz = (int)h_colidx.size();
    // source values always through the device buffer on this path
    if (a.inline_vals) {
      std::memcpy(h_stage, a.vals_inline, (size_t)(a.nk + a.nsrc) * sizeof(double));
      HIPCHK(hipMemcpyAsync(d_kv.p, h_stage, (size_t)(a.nk + a.nsrc) * sizeof(double), hipMemcpyHostToDevice, st));
      a.inline_vals = 0;
    }
    std::memset(&out, 0, sizeof(out));
    const int maxit = a.mode == MODE_EVAL ? 0 : a.maxit;
    std::vector<int> todo;   // samples taking part in this solve
    for (int sm = 0; sm < S; ++sm) {   // a sample takes part when any of its blocks is active (the sparse system spans all blocks)
      bool on = !host_active;
      for (int k = 0; k < A.n_comp && !on; ++k) on = host_active[(size_t)k * S + sm] != 0;
      if (on) todo.push_back(sm);
    }
    if (todo.empty()) return CH_OK;
    std::vector<int> status(S, 1), iters(S, 0);
    std::vector<double> rate_prev(S, 1.0), rate_new(S, -1.0), dn_prev(S, 0.0), fnorm(S, 0.0), scale(S, 1.0);
    for (int sm : todo) rate_prev[sm] = (a.mode == MODE_TRAN && !a.reset_rate) ? sp_rate_v[sm] : 1.0;
    const dim3 b256(256), b1k(1024);
    auto grid = [&](int nx, size_t nl) { return dim3((unsigned)nx, (unsigned)nl); };
    const int gn = (n + 255) / 256, gd = (nd + 63) / 64, ga = (std::max(n, nnz) + 255) / 256;
    // the O(n) passes: one workgroup per sample for small systems, up to SP_NP workgroups + a finishing pass from 4096 rows
    const bool many = n >= 4096;
    const int nbr = std::min(SP_NP, gn);
    g_arena = &arena;
    if (many) { HIPCHK(sp_part.alloc((size_t)S * 8 * SP_NP)); HIPCHK(sp_hrow.alloc((size_t)S * std::max(1, n_heavy_rows) * SP_RB)); }
    if (n_heavy_mat + n_heavy_vec > 0) HIPCHK(sp_hpart.alloc((size_t)S * (n_heavy_mat + n_heavy_vec) * SP_HB * 2));
    // slot 0: every sample of this solve (predict, commit); slot 1: samples still iterating; slot 2: samples to (re)factor
    int rc = stage_list(0, todo); if (rc != CH_OK) return rc;
    hipLaunchKernelGGL(sp_predict_kernel, grid(gn, todo.size()), b256, 0, st, a, sparse_dev(which), (const int*)sp_act[0].p);
    std::vector<int> act = todo;
    rc = stage_list(1, act); if (rc != CH_OK) return rc;
    const bool damp = a.mode == MODE_DC && a.dv_max > 0.0 && (!A.mos_hdev.empty() || A.wide);
    for (int it = 0; it <= maxit && !act.empty(); ++it) {
      {
        const SparseDev d = sparse_dev(which); const int* al = sp_act[1].p;
        hipLaunchKernelGGL(sp_eval_kernel, grid(A.wide ? gd : 2 * gd, act.size()), dim3(64), 0, st, a, d, al);
        hipLaunchKernelGGL(sp_assemble_kernel, grid(ga, act.size()), b256, 0, st, a, d, al);
        if (n_heavy_mat + n_heavy_vec > 0) {
          hipLaunchKernelGGL(sp_assemble_heavy_kernel, grid((n_heavy_mat + n_heavy_vec) * SP_HB, act.size()), b256, 0, st, a, d, al, sp_hpart.p);
          hipLaunchKernelGGL(sp_assemble_heavy_finish_kernel, grid(n_heavy_mat + n_heavy_vec, act.size()), dim3(64), 0, st, a, d, al, (const double*)sp_hpart.p);
        }
        if (a.gshunt != 0.0) hipLaunchKernelGGL(sp_diag_shunt_kernel, grid(gn, act.size()), b256, 0, st, a, d, al);
        if (a.mode == MODE_DC) {
          if (many) { hipLaunchKernelGGL(sp_norms2_kernel, grid(nbr, act.size()), b256, 0, st, a, d, al, 0, sp_part.p, nbr); hipLaunchKernelGGL(sp_finish_kernel, grid(1, act.size()), b256, 0, st, d, al, (const double*)sp_part.p, nbr, 2, (const double*)sp_hrow.p, 0); }
          else hipLaunchKernelGGL(sp_norms_kernel, grid(1, act.size()), b1k, 0, st, a, d, al, 0);
        }
        n_launch += 2;
      }
      if (a.mode == MODE_EVAL) { for (int sm : act) status[sm] = 0; break; }
      if (a.mode == MODE_DC) {
        rc = sp_sync(); if (rc != CH_OK) return rc;
        std::vector<int> keep;
        for (int sm : act) {
          fnorm[sm] = h_red[(size_t)sm * 8];
          if (!(fnorm[sm] == fnorm[sm]) || fnorm[sm] > 1e300) status[sm] = 2;
          else if (fnorm[sm] < a.dc_abstol) status[sm] = 0;
          else keep.push_back(sm);
        }
        if (keep.size() != act.size()) { act.swap(keep); if (!act.empty()) { rc = stage_list(1, act); if (rc != CH_OK) return rc; } }
        if (act.empty()) break;
      }
      if (it == maxit) break;
      bool fresh = false;
      if (!plan[which].valid) { rc = sparse_plan_from_current(which, act[0]); if (rc != CH_OK) { for (int sm : act) status[sm] = 2; act.clear(); break; } fresh = true; }
      std::vector<int> work = act;   // samples whose factorisation is still to be done in this iteration
      const int* wl = sp_act[1].p;
      for (int attempt = 0; attempt < 2 && !work.empty(); ++attempt) {
        const SparseDev d = sparse_dev(which);
        launch_lu_solve(which, wl, work.size());
        const double* sc = nullptr;
        if (damp) {
          if (many) { hipLaunchKernelGGL(sp_norms2_kernel, grid(nbr, work.size()), b256, 0, st, a, d, wl, 1, sp_part.p, nbr); hipLaunchKernelGGL(sp_finish_kernel, grid(1, work.size()), b256, 0, st, d, wl, (const double*)sp_part.p, nbr, 3, (const double*)sp_hrow.p, 0); }
          else hipLaunchKernelGGL(sp_norms_kernel, grid(1, work.size()), b1k, 0, st, a, d, wl, 1);
          rc = sp_sync(); if (rc != CH_OK) return rc;
          for (int sm : work) { scale[sm] = 1.0; const double mx = h_red[(size_t)sm * 8 + 1]; if (!h_flag[(size_t)sm * 2] && mx > a.dv_max) scale[sm] = a.dv_max / mx; }
          g_arena = &arena;
          HIPCHK(sp_scale.alloc((size_t)S));
          std::memcpy(h_scale, scale.data(), (size_t)S * sizeof(double));
          HIPCHK(hipMemcpyAsync(sp_scale.p, h_scale, (size_t)S * sizeof(double), hipMemcpyHostToDevice, st));
          sc = sp_scale.p;
        }
        if (many) {   // no-op where the factorisation failed
          hipLaunchKernelGGL(sp_update2_kernel, grid(nbr + n_heavy_rows * SP_RB, work.size()), b256, 0, st, a, d, wl, sc, sp_part.p, nbr, sp_hrow.p);
          hipLaunchKernelGGL(sp_finish_kernel, grid(1, work.size()), b256, 0, st, d, wl, (const double*)sp_part.p, nbr, 0, (const double*)sp_hrow.p, a.mode == MODE_TRAN ? 1 : 0);
        } else hipLaunchKernelGGL(sp_update_kernel, grid(1, work.size()), b1k, 0, st, a, d, wl, sc);
        rc = sp_sync(); if (rc != CH_OK) return rc;
        n_launch += 2;
        std::vector<int> failed;
        for (int sm : work) if (h_flag[(size_t)sm * 2]) failed.push_back(sm);
        if (failed.empty()) break;
        // a static pivot became zero: re-analyse once with the current values of the first failing sample (KLU would
        // re-pivot here too) and redo the failing samples; the others have already taken their step
        auto drop_failed = [&]() { for (int sm : failed) status[sm] = 2; act.erase(std::remove_if(act.begin(), act.end(), [&](int q) { return status[q] == 2; }), act.end()); };
        if (fresh || attempt == 1) { drop_failed(); work.clear(); break; }
        rc = sparse_plan_from_current(which, failed[0]);
        if (rc != CH_OK) { drop_failed(); break; }
        fresh = true;
        work.swap(failed);
        rc = stage_list(2, work); if (rc != CH_OK) return rc;
        wl = sp_act[2].p;
      }
      std::vector<int> keep;
      for (int sm : act) {
        if (status[sm] == 2) continue;
        ++iters[sm];
        if (h_flag[(size_t)sm * 2 + 1]) { status[sm] = 2; continue; }
        if (a.mode == MODE_TRAN) {
          const double dn = std::sqrt(h_red[(size_t)sm * 8 + 2] / n);
          bool conv = false;
          if (it == 0) conv = dn <= a.newton_tol || (rate_prev[sm] < 0.9 && 2.0 * std::max(rate_prev[sm], 0.02) * dn <= a.newton_tol);
          else { rate_new[sm] = dn_prev[sm] > 0 ? dn / dn_prev[sm] : 0.0; conv = dn <= a.newton_tol; }
          dn_prev[sm] = dn;
          if (conv) { status[sm] = 0; continue; }
        }
        keep.push_back(sm);
      }
      if (keep.size() != act.size()) { act.swap(keep); if (!act.empty()) { rc = stage_list(1, act); if (rc != CH_OK) return rc; } }
    }
    for (int sm : todo) if (a.mode == MODE_TRAN && status[sm] == 0) sp_rate_v[sm] = iters[sm] >= 2 ? std::min(1.0, std::max(rate_new[sm], 1e-4)) : std::min(1.0, rate_prev[sm] * 1.5);
    if (many) {
      hipLaunchKernelGGL(sp_commit2_kernel, grid(nbr, todo.size()), b256, 0, st, a, sparse_dev(which), (const int*)sp_act[0].p, (a.mode == MODE_TRAN) ? 0 : 1, sp_part.p, nbr);
      hipLaunchKernelGGL(sp_finish_kernel, grid(1, todo.size()), b256, 0, st, sparse_dev(which), (const int*)sp_act[0].p, (const double*)sp_part.p, nbr, 1, (const double*)sp_hrow.p, 0);
    } else hipLaunchKernelGGL(sp_commit_kernel, grid(1, todo.size()), b1k, 0, st, a, sparse_dev(which), (const int*)sp_act[0].p, (a.mode == MODE_TRAN) ? 0 : 1);
    rc = sp_sync(); if (rc != CH_OK) return rc;
    n_launch += 1;
    for (int sm : todo) {
      sp_status_v[sm] = status[sm];
      if (status[sm] != 0) ++out.n_fail;
      if (status[sm] == 2) ++out.n_singular;
      out.max_iters = std::max(out.max_iters, iters[sm]); out.sum_iters += iters[sm]; out.sum_block_iters += iters[sm]; out.fnorm = std::max(out.fnorm, fnorm[sm]);
      const double* r = h_red + (size_t)sm * 8;
      if (a.mode == MODE_TRAN && r[7] > 0) {
        out.errk = std::max(out.errk, a.ck * std::sqrt(r[4] / r[7])); out.errkm1 = std::max(out.errkm1, a.ckm1 * std::sqrt(r[5] / r[7])); out.errkp1 = std::max(out.errkp1, a.ckp1 * std::sqrt(r[6] / r[7]));
      }
    }
    return CH_OK;
  }

  // host_active: host copy of the per-block active mask given to the kernel (DC restart passes, ch_eval), or null
  int run_newton(const NewtonArgs& a, const unsigned char* host_active, Summary& out) {
    if (path == 2) return run_sparse(a, host_active, out);
    hipStream_t st = ctx->stream;
    const int nblk = A.n_comp * S;
    // kernel duration from the dispatch's own start/stop events on one launch in CEDARHIP_TIME_EVERY (default 8; timing every
    // launch costs ~5 us of host time per step)
    // sampled pseudo-randomly (a fixed stride aliases with the accept/reject rhythm of the stepper and biased the mean by 9 %)
    const bool timed = time_every <= 1 || ((uint64_t)(n_launch + 1) * 0x9E3779B97F4A7C15ull >> 33) % (uint64_t)time_every == 0;
    const auto tp0 = host_profile ? hclock::now() : hclock::time_point();
    // timed launches carry their start/stop events in the dispatch itself (hipExtLaunchKernelGGL): the elapsed time is the
    // kernel's own begin-to-end, the quantity rocprofv3 --kernel-trace reports
    hipEvent_t e0 = timed ? ev0 : nullptr, e1 = timed ? ev1 : nullptr;
    const dim3 g(nblk), b(block_threads);
    if (A.wide) {
      if (lu_variant == 16) hipExtLaunchKernelGGL((newton_block_kernel<16, true>), g, b, (uint32_t)lds_bytes, st, e0, e1, 0, a);
      else hipExtLaunchKernelGGL((newton_block_kernel<0, true>), g, b, (uint32_t)lds_bytes, st, e0, e1, 0, a);
    }
    else if (lu_variant == 8) hipExtLaunchKernelGGL(newton_block_kernel<8>, g, b, (uint32_t)lds_bytes, st, e0, e1, 0, a);
    else if (lu_variant == 12) hipExtLaunchKernelGGL(newton_block_kernel<12>, g, b, (uint32_t)lds_bytes, st, e0, e1, 0, a);
    else if (lu_variant == 16) hipExtLaunchKernelGGL(newton_block_kernel<16>, g, b, (uint32_t)lds_bytes, st, e0, e1, 0, a);
    else if (lu_variant == 32) hipExtLaunchKernelGGL(newton_block_kernel<32>, g, b, (uint32_t)lds_bytes, st, e0, e1, 0, a);
    else hipExtLaunchKernelGGL(newton_block_kernel<0>, g, b, (uint32_t)lds_bytes, st, e0, e1, 0, a);
    if (!host_reduce) hipLaunchKernelGGL(reduce_blocks_kernel, dim3(1), dim3(256), 9 * 256 * sizeof(double), st, a);
    const auto tp1 = host_profile ? hclock::now() : hclock::time_point();
    // the host thread has nothing else to do: poll for completion instead of sleeping on an interrupt.
    // (Watching the block records in mapped memory for a per-launch sequence number instead of the stream signal was
    // tried: the system-scope release each block then needs costs ~28 us per launch; profiles/r01_notes.md.)
    {
      hipError_t q = hipErrorNotReady;
      for (int spin = 0; spin < 200000 && q == hipErrorNotReady; ++spin) q = hipStreamQuery(st);
      if (q == hipErrorNotReady) q = hipStreamSynchronize(st);
      if (q != hipSuccess) { set_err(std::string("newton kernel: ") + hipGetErrorString(q)); return CH_ERR_DEVICE; }
    }
    HIPCHK(hipGetLastError());
    const auto tp2 = host_profile ? hclock::now() : hclock::time_point();
    if (timed) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); device_ms += ms; n_timed += 1; }
    n_launch += 1;
    if (!host_reduce) out = *h_sum;
    else {
      // host-side reduction of the per-block records (same arithmetic as reduce_all_blocks)
      Summary r; std::memset(&r, 0, sizeof(r));
      for (int s = 0; s < S; ++s) {
        double e2[3] = {0, 0, 0}; long nd = 0; int smx = 0;
        for (int k = 0; k < A.n_comp; ++k) {
          const size_t b = (size_t)k * S + s;
          if (host_active && !host_active[b]) continue;
          const BlockOut& o = h_out[b];
          e2[0] += o.e2k; e2[1] += o.e2km1; e2[2] += o.e2kp1; nd += o.ndiff;
          if (o.status != 0) ++r.n_fail;
          if (o.status == 2) ++r.n_singular;
          smx = std::max(smx, o.iters); r.sum_block_iters += o.iters; r.fnorm = std::max(r.fnorm, o.fnorm);
        }
        r.sum_iters += smx; r.max_iters = std::max(r.max_iters, smx);
        if (nd > 0) { r.errk = std::max(r.errk, a.ck * std::sqrt(e2[0] / nd)); r.errkm1 = std::max(r.errkm1, a.ckm1 * std::sqrt(e2[1] / nd)); r.errkp1 = std::max(r.errkp1, a.ckp1 * std::sqrt(e2[2] / nd)); }
      }
      out = r;
    }
    if (host_profile) {
      const auto tp3 = hclock::now();
      prof_launch += std::chrono::duration<double>(tp1 - tp0).count(); prof_wait += std::chrono::duration<double>(tp2 - tp1).count();
      prof_reduce += std::chrono::duration<double>(tp3 - tp2).count();
    }
    return CH_OK;
  }

  // unknown-space state of slot -> MNA vectors [S][n_mna]
  int download_mna(int slot, double t, int mode, double* x_out) {
    std::vector<double> xs((size_t)S * A.n_unk);
    HIPCHK(hipMemcpy(xs.data(), d_X.p + (size_t)slot * S * A.n_unk, xs.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> sv, kv; eval_sources(t, mode, sv, kv);
    const int nk = (int)A.known.size();
    for (int s = 0; s < S; ++s) {
      double* xo = x_out + (size_t)s * A.n_mna;
      for (int n = 1; n <= n_nodes; ++n) xo[n - 1] = A.node_unknown[n] >= 0 ? xs[(size_t)s * A.n_unk + A.node_unknown[n]] : kv[(size_t)(Ssrc > 1 ? s : 0) * nk + A.node_known[n]];
      for (int b = 0; b < A.n_branch; ++b) xo[n_nodes + b] = A.branch_unknown[b] >= 0 ? xs[(size_t)s * A.n_unk + A.branch_unknown[b]] : CH_NAN;
    }
    return CH_OK;
  }
  int upload_from_mna(int slot, const double* x_mna) {
    std::vector<double> xs((size_t)S * A.n_unk);
    for (int s = 0; s < S; ++s) for (int u = 0; u < A.n_unk; ++u) xs[(size_t)s * A.n_unk + u] = x_mna[(size_t)s * A.n_mna + A.unk_mna[u]];
    HIPCHK(hipMemcpy(d_X.p + (size_t)slot * S * A.n_unk, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice));
    return CH_OK;
  }

  // ------------------------------------------------------------------------------------------
  // DC operating point: CedarDCOp + bootstrapped_nlsolve, restarted per block (src/dcop.jl:53-94).
  // Leaves the solution in ring slot `slot`.
  int dc_solve(const ch_dc_opts& o, int slot, std::vector<int>* status_out, ch_stats* stt) {
    const int mode = o.tran_mode ? 2 : 0;
    const int nblk = A.n_comp * S;
    int rc = CH_OK;
    std::vector<unsigned char> active(nblk, 1);
    std::vector<double> xs((size_t)S * A.n_unk), xm(A.n_mna);
    std::vector<BlockOut> bo(nblk);
    NewtonArgs a = base;
    a.mode = MODE_DC; a.maxit = std::max(1, o.maxiters); a.dc_abstol = o.abstol; a.dv_max = o.dv_max; a.gshunt = 0.0;
    a.abstol = 1e-6; a.reltol = 1e-3; a.newton_tol = 0.1;
    a.hist_slot[0] = slot; a.cand_slot = slot; a.active = d_active.p;
    rc = set_sources(a, 0.0, mode);
    if (rc != CH_OK) return rc;
    std::vector<Rng> rngs; for (int s = 0; s < S; ++s) rngs.emplace_back(o.seed + (uint64_t)s);
    auto block_of_unknown = [&](int u) { int c = (int)(std::upper_bound(A.comp_uofs.begin(), A.comp_uofs.end(), u) - A.comp_uofs.begin()) - 1; return c; };
    int n_active = nblk;
    std::vector<unsigned char> donor_ok(nblk, 0);   // blocks that converged in this call (their state is a valid starting point for their siblings)
    const int nrest = std::max(1, o.n_restarts);
    for (int r = 0; r < nrest + 1 && n_active > 0; ++r) {
      const bool homotopy = (r == nrest);
      // initial guess for the active blocks
      HIPCHK(hipMemcpy(xs.data(), d_X.p + (size_t)slot * S * A.n_unk, xs.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (int s = 0; s < S; ++s) {
        bool any = false;
        for (int c = 0; c < A.n_comp; ++c) if (active[(size_t)c * S + s]) any = true;
        if (!any) continue;
        if (!homotopy) {
          if (r == 0 && o.x0) for (int i = 0; i < A.n_mna; ++i) xm[i] = o.x0[(size_t)s * A.n_mna + i];
          else for (int i = 0; i < A.n_mna; ++i) xm[i] = 1e-7 * rngs[s].normal();
        } else std::fill(xm.begin(), xm.end(), 0.0);
        for (int u = 0; u < A.n_unk; ++u) if (active[(size_t)block_of_unknown(u) * S + s]) xs[(size_t)s * A.n_unk + u] = xm[A.unk_mna[u]];
      }
      // First restart of a batch: a block that did not converge from the random start is started from the operating point of the
      // same block in a sample that did (continuation from a neighbouring parameter set) — a few iterations instead of another
      // `maxiters` spent from 1e-7*randn; later restarts are random again as in the reference (src/dcop.jl:53-94).
      if (r == 1 && S > 1) {
        for (int c = 0; c < A.n_comp; ++c) {
          int donor = -1;
          for (int s = 0; s < S && donor < 0; ++s) if (!active[(size_t)c * S + s] && donor_ok[(size_t)c * S + s]) donor = s;
          if (donor < 0) continue;
          for (int s = 0; s < S; ++s) if (active[(size_t)c * S + s])
            for (int u = A.comp_uofs[c]; u < A.comp_uofs[c] + A.comp_nc[c]; ++u) xs[(size_t)s * A.n_unk + u] = xs[(size_t)donor * A.n_unk + u];
        }
      }
      HIPCHK(hipMemcpy(d_X.p + (size_t)slot * S * A.n_unk, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(d_active.p, active.data(), nblk, hipMemcpyHostToDevice));
      Summary sm;
      if (!homotopy) {
        rc = run_newton(a, active.data(), sm);
        if (rc != CH_OK) return rc;
      } else {
        // gmin stepping: a shunt conductance on every node, relaxed decade by decade, then removed
        for (double g = 1e-2; g >= 1e-13 * 0.99; g *= 0.1) { a.gshunt = g; rc = run_newton(a, active.data(), sm); if (rc != CH_OK) return rc; }
        a.gshunt = 0.0;
        rc = run_newton(a, active.data(), sm);
        if (rc != CH_OK) return rc;
      }
      if (stt) { stt->n_block_iters += sm.sum_block_iters; stt->nnonliniter += sm.sum_iters; stt->nf += sm.sum_iters; stt->njacs += sm.sum_iters; stt->nfactors += sm.sum_iters; stt->nsolve += sm.sum_iters; }
      if (path == 2) { for (int b = 0; b < nblk; ++b) bo[b].status = sp_status_v[b % S]; }
      else if (host_reduce) std::memcpy(bo.data(), h_out, nblk * sizeof(BlockOut)); else HIPCHK(hipMemcpy(bo.data(), d_out.p, nblk * sizeof(BlockOut), hipMemcpyDeviceToHost));
      n_active = 0;
      for (int b = 0; b < nblk; ++b) if (active[b]) { if (bo[b].status == 0) { active[b] = 0; donor_ok[b] = 1; } else ++n_active; }
      if (n_active > 0 && stt) { stt->nrestarts++; stt->nnonlinconvfail++; }
    }
    if (status_out) { status_out->assign(S, CH_OK); for (int b = 0; b < nblk; ++b) if (active[b]) (*status_out)[b % S] = bo[b].status == 2 ? CH_ERR_SINGULAR : CH_ERR_MAXITERS; }
    if (n_active > 0) { set_err("DC operating point analysis failed for " + std::to_string(n_active) + " block(s)"); return CH_ERR_MAXITERS; }
    return CH_OK;
  }

  // ------------------------------------------------------------------------------------------
  // Device-resident step controller: which circuits qualify (ch_persist.hpp header), and the launch.
  DevBuf<int> d_pci; DevBuf<double> d_pcd, d_pbps, d_psave, d_ptimes, d_prows, d_wgrec, d_grprec; DevBuf<unsigned> d_pcnt; DevBuf<TranCtl> d_pctl; DevBuf<int> d_pwgc, d_pwgk; DevBuf<double> d_pdcent, d_ptrans;
  int n_cu = 0, persist_mode = 0;
  bool persist_aborted = false;   // the last device-stepper launch gave up on a wait (its workgroups were not co-resident: another process's kernel held part of the GPU)
  // `own_steps`: the batch would run with per-sample step acceptance (no grid-wide wait anywhere in the kernel), so the workgroups
  // need not be co-resident and any number of samples can be queued behind each other
  bool persist_eligible(std::string& why, bool own_steps) {
    auto no = [&](const char* m) { why = m; return false; };
    if (path != 1) return no("the circuit takes the sparse path");
    if (A.n_comp < 1) return no("the circuit has no unknowns");
    if (!(lu_variant == 8 || lu_variant == 12 || lu_variant == 16)) return no("a Jacobian block has more than 16 unknowns");
    if (A.wide) {
      // compiled Verilog-A devices: every block of ONE class; a class with split (large) devices needs both halves of two blocks in
      // one wavefront each (wave pairs), any other class all its slots in one wavefront
      if (A.classes.size() != 1) return no("compiled Verilog-A devices in blocks of several classes");
      if (A.nb > 0) return no("compiled Verilog-A devices in a bordered form");
      if (wide_split ? (wide_l + wide_other > 32) : (h_cms[0].nslots > 64)) return no("a block's compiled devices need more evaluation lanes than a wave pair offers");
      if (own_steps && S == 1 && A.n_comp > 1) return no("per-block steps of one circuit with compiled Verilog-A devices");
    } else if (block_threads != 64) return no("a block needs more than one wavefront of device slots");
    if (max_mc > 8) return no("more than 8 MOSFET classes in a block");
    if (Ssrc != 1) return no("per-sample source parameters");
    const bool wg_consts = own_steps && S == 1 && A.n_comp > 1;   // per-block steps: every workgroup gets the sources of ITS blocks only (checked there)
    if (!wg_consts && (needed_src.size() > (size_t)P_MAXSRC || A.known.size() + (size_t)n_dev_src() > 64)) return no("more than 64 sources / known-node and source values per attempt");
    if (!(S == 1 || A.n_comp == 1)) return no("several blocks per sample in a multi-sample batch");
    if (A.nb > 0 && (S != 1 || A.border_dev.size() > 8)) return no("bordered form: one sample and at most 8 devices on the border alone");
    for (const ClassMeta& m : h_cms) if ((!A.wide && m.nslots > 64) || m.nc > lu_variant || m.n_work <= 0) return no("a block class does not fit the one-wave register path");
    if (n_cu == 0) { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return no("hipGetDeviceProperties failed"); n_cu = prop.multiProcessorCount; }
    const long nblk = (long)A.n_comp * S;
    const int bpw = persist_bpw(nblk);
    if (nblk > (long)bpw * n_cu && !own_steps) return no("more blocks than resident wavefronts (4 per CU)");
    // p_grid_reduce: 8 group leaders sweep at most 32 member workgroups each
    if (!own_steps && (nblk + bpw - 1) / bpw > 256) return no("more than 256 workgroups in a grid-wide reduction");
    size_t npwl = 0; for (int i : needed_src) npwl += src[i].ts.size();
    if (!wg_consts && npwl > 2048) return no("piecewise-linear tables above 2048 points");
    return true;
  }
  // Blocks per workgroup of the device-resident stepper: four (one per wavefront), or two — one wave pair per CU, the other two
  // wavefronts idle — for few, heavy blocks (compiled Verilog-A devices: 57 k instructions per evaluation and a 5 kB constant block
  // per instance): they then spread over twice the CUs and do not share a CU's vector L1 four ways.
  int persist_bpw(long nblk) const {
    if (n_cu > 0 && A.wide && wide_split && nblk <= 2L * n_cu) return 2;
    return PW;
  }
  // Every sample of a batch (n_comp == 1) or every block of one circuit of independent blocks (S == 1) takes its own steps when
  // the output is wanted on a common `saveat` grid: independent blocks ARE independent problems, a shared step size only makes
  // each pay for the others' break points and dilutes its local error in the array-wide norm.  (Not for the bordered form.)
  bool persist_own_steps(const ch_tran_opts& o) const {
    return ((A.n_comp == 1 && S > 1) || (S == 1 && A.n_comp > 1 && A.nb == 0 && !A.wide)) && o.n_saveat > 0 && o.step_control != CH_STEPS_SHARED && std::getenv("CEDARHIP_LOCKSTEP") == nullptr;
  }
  size_t persist_wave_doubles(bool wg_consts) const {
    const size_t n_ent = wg_consts ? (size_t)P_MAXSRC : A.known.size() + n_dev_src();
    return lds_doubles_fixed + 16 * (size_t)A.max_nc + 10 + 48 + P_MAXSRC + n_ent + (size_t)max_mc * B4L_STRIDE + (lds_extra_bytes + 7) / 8 + 2;
  }
  // dcm != nullptr: the operating point of the bordered form instead of a transient (PersistArgs::dc_mode) — the state in ring
  // slot 0 is the initial iterate and receives the result; no rows, no finish_tran; returns the solve's status
  int tran_persistent(double t0, double t1, const ch_tran_opts& o, ch_result& R, const std::vector<double>& bps, int kmax, double dtmin, double dtmax,
                      int max_steps, int nmaxit, hclock::time_point tstart, bool& used, const ch_dc_opts* dcm = nullptr, long long* dc_iters = nullptr,
                      const std::vector<double>* bpc = nullptr) {
    used = false;
    hipStream_t st = ctx->stream;
    g_arena = &arena;
    const int n_obs = R.n_obs;
    const int nblk = A.n_comp * S, bpw = persist_bpw(nblk), n_wg = (nblk + bpw - 1) / bpw;
    // ---- constants blob: needed sources, known-node definitions, device-source map, PWL tables ----
    // One blob for the whole grid, or — per-block steps of one circuit (wg_consts) — one per workgroup holding only what its
    // blocks reference, with a map from the circuit's known-node / device-source indices to the workgroup's entries: a block with
    // its own clock source neither evaluates nor stops at the other 1023 clocks.
    const bool own_steps = persist_own_steps(o);
    const bool wg_consts = own_steps && S == 1 && A.n_comp > 1;
    std::vector<int> ci; std::vector<double> cd;
    const int nsrc = (int)src.size(), nk = (int)A.known.size(), nds = n_dev_src();
    // entries `kn` (known-node indices) then `ds` (device-source slots) -> blob; false when a limit of the kernel is exceeded
    auto build_blob = [&](const std::vector<int>& kn, const std::vector<int>& ds, std::vector<int>& bi, std::vector<double>& bd, std::vector<int>& need) -> bool {
      std::vector<char> nd(std::max(1, nsrc), 0);
      for (int k : kn) for (auto& tm : A.known[k].terms) nd[tm.first] = 1;
      for (int j : ds) if (j < (int)dev_src.size()) nd[dev_src[j]] = 1;
      need.clear();
      for (int i = 0; i < nsrc; ++i) if (nd[i]) need.push_back(i);
      std::vector<int> pos(std::max(1, nsrc), -1);
      for (size_t i = 0; i < need.size(); ++i) pos[need[i]] = (int)i;
      std::vector<double> pt, py;
      bi = {(int)need.size(), (int)(kn.size() + ds.size()), 0, (int)kn.size()};
      for (int i : need) { bi.push_back(src[i].kind); bi.push_back((int)pt.size()); bi.push_back((int)src[i].ts.size()); pt.insert(pt.end(), src[i].ts.begin(), src[i].ts.end()); py.insert(py.end(), src[i].ys.begin(), src[i].ys.end()); }
      bi[2] = (int)pt.size();
      // entries: the known-node values, then the device source values (kvl and svl are contiguous in LDS)
      std::vector<int> eptr(1, 0), eidx; std::vector<double> ecoef;
      for (int k : kn) { for (auto& tm : A.known[k].terms) { eidx.push_back(pos[tm.first]); ecoef.push_back(tm.second); } eptr.push_back((int)eidx.size()); }
      for (int j : ds) { if (j < (int)dev_src.size()) { eidx.push_back(pos[dev_src[j]]); ecoef.push_back(1.0); } eptr.push_back((int)eidx.size()); }
      bi.insert(bi.end(), eptr.begin(), eptr.end()); bi.insert(bi.end(), eidx.begin(), eidx.end());
      bd.clear();
      for (int i : need) for (int k = 0; k < CH_SRC_NPAR; ++k) bd.push_back(h_src_par[(size_t)i * CH_SRC_NPAR + k]);
      bd.insert(bd.end(), ecoef.begin(), ecoef.end()); bd.insert(bd.end(), pt.begin(), pt.end()); bd.insert(bd.end(), py.begin(), py.end());
      for (size_t q = 4; q < bi.size(); ++q) if (bi[q] < 0) return false;
      return need.size() <= (size_t)P_MAXSRC && kn.size() + ds.size() <= 64 && pt.size() <= 2048;
    };
    std::vector<int> wgc, wgk;                 // wg_consts: per workgroup {ci offset, cd offset, n_ci, n_cd, bps offset, nbp}; its entries' circuit-wide ids
    std::vector<double> bps_all;
    size_t max_ci = 0, max_cd = 0;
    if (!wg_consts) {
      std::vector<int> kn(nk), ds(nds), need;
      std::iota(kn.begin(), kn.end(), 0); std::iota(ds.begin(), ds.end(), 0);
      if (!build_blob(kn, ds, ci, cd, need)) { set_err("device-resident stepper: the source tables exceed the kernel's limits"); return CH_OK; }
      max_ci = ci.size(); max_cd = cd.size();
    } else {
      wgk.assign((size_t)n_wg * P_MAXSRC, -1);   // [wg][entry]: known-node index (entries 0 .. nk_local-1), then device-source slot
      for (int w = 0; w < n_wg; ++w) {
        std::vector<char> uk(nk, 0), ud(nds, 0);
        for (int b = w * bpw; b < std::min(nblk, (w + 1) * bpw); ++b)
          for (int i = 0; i < A.comp_ndev[b]; ++i) {
            const EDev& e = A.edev[A.comp_dofs[b] + i];
            for (int k = 0; k < NTERM; ++k) if (e.term[k] < 0) uk[-e.term[k] - 1] = 1;
            if (e.src >= 0) ud[dsrc_host[A.comp_dofs[b] + i]] = 1;
          }
        std::vector<int> kn, ds, need, bi; std::vector<double> bd;
        for (int k = 0; k < nk; ++k) if (uk[k]) kn.push_back(k);
        for (int j = 0; j < nds; ++j) if (ud[j]) ds.push_back(j);
        if (ds.empty()) ds.push_back(0);   // the kernel's source-value array is never empty
        if (!build_blob(kn, ds, bi, bd, need)) { set_err("device-resident stepper: a workgroup's blocks reference more than 64 sources / known nodes"); return CH_OK; }
        for (size_t q = 0; q < kn.size(); ++q) wgk[(size_t)w * P_MAXSRC + q] = kn[q];
        for (size_t q = 0; q < ds.size(); ++q) wgk[(size_t)w * P_MAXSRC + kn.size() + q] = ds[q];
        std::vector<double> wb, wc;
        { std::vector<std::pair<double, double>> pts;
          for (int i : need) source_breakpoint_codes(src[i], &h_src_par[(size_t)i * CH_SRC_NPAR], t0, t1, pts);
          merge_breakpoints(pts, t1, wb, wc); }
        wgc.insert(wgc.end(), {(int)ci.size(), (int)cd.size(), (int)bi.size(), (int)bd.size(), (int)bps_all.size(), (int)wb.size()});
        ci.insert(ci.end(), bi.begin(), bi.end()); cd.insert(cd.end(), bd.begin(), bd.end());
        bps_all.insert(bps_all.end(), wb.begin(), wb.end());
        bps_all.insert(bps_all.end(), wc.begin(), wc.end());   // the codes of these times follow them (the kernel reads code i at [count + i])
        max_ci = std::max(max_ci, bi.size()); max_cd = std::max(max_cd, bd.size());
      }
    }
    const size_t wave_d = persist_wave_doubles(wg_consts);
    size_t lds = (max_cd + (max_ci + 1) / 2 + PW * P_NREC + P_NREC + 4 + P_SCR + PW * wave_d) * sizeof(double);
    // compiled Verilog-A devices: room for the workgroup's parameter and constant blocks in LDS (what the blocks of the heaviest
    // component need, without counting shared blocks once; the kernel shares them and stops staging when the arena is full)
    size_t va_arena = 0;
    if (A.wide && std::getenv("CEDARHIP_VA_NO_LDS") == nullptr) {
      size_t worst = 0;
      for (int cpt = 0; cpt < A.n_comp; ++cpt) {
        size_t need = 0;
        for (int i = 0; i < A.comp_ndev[cpt]; ++i) {
          const EDev& e = A.edev[A.comp_dofs[cpt] + i];
          if (e.kind != K_VA) continue;
          const int mod = dev[e.hdev].ipar[0];
          need += (size_t)((va_gen::param_doubles(mod) + 1) & ~1) + (size_t)((va_gen::cache_doubles(mod) + 1) & ~1);
        }
        worst = std::max(worst, need);
      }
      const size_t room = lds < 148 * 1024 ? (148 * 1024 - lds) / sizeof(double) : 0;
      va_arena = std::min(worst * (size_t)bpw, room);
      lds += va_arena * sizeof(double);
      if (std::getenv("CEDARHIP_DEBUG_STEPPER")) std::fprintf(stderr, "[stepper] compiled devices: %zu doubles per block, LDS arena %zu doubles, workgroup LDS %zu bytes\n", worst, va_arena, lds);
    }
    // wave pairs share the device evaluation by function when every block has the same class and at most 32 evaluation slots
    const bool pair = A.wide ? wide_split
                             : (A.classes.size() == 1 && h_cms[0].nslots <= 32 && std::getenv("CEDARHIP_PERSIST_NOPAIR") == nullptr);
    if (lds > 150 * 1024) { set_err("device-resident stepper: the workgroup's LDS footprint exceeds 150 KB"); return CH_OK; }
    // ---- output rows ----
    const size_t row_d = std::max<size_t>(1, (size_t)n_obs * S);
    long long max_rows;
    if (o.n_saveat > 0) max_rows = (long long)o.n_saveat + 1;
    else max_rows = std::min<long long>((long long)max_steps + 2, std::max<long long>(1024, std::min<long long>(1 << 20, (long long)((256u << 20) / (row_d * sizeof(double))))));
    if (o.n_saveat == 0 && std::getenv("CEDARHIP_PERSIST_MAXROWS")) max_rows = std::max(2L, std::atol(std::getenv("CEDARHIP_PERSIST_MAXROWS")));   // test hook: forces the drain-and-resume path
    if (dcm) max_rows = 2;
    HIPCHK(d_pci.upload(ci, st)); HIPCHK(d_pcd.upload(cd, st)); {
      std::vector<double> bpu = wg_consts ? bps_all : bps;   // [times | codes]
      if (!wg_consts) for (size_t b = 0; b < bps.size(); ++b) bpu.push_back(bpc ? (*bpc)[b] : -1.0);
      HIPCHK(d_pbps.upload(bpu, st));
    }
    if (wg_consts) { HIPCHK(d_pwgc.upload(wgc, st)); HIPCHK(d_pwgk.upload(wgk, st)); }
    { std::vector<double> sv(o.saveat, o.saveat + std::max(0, o.n_saveat)); if (sv.empty()) sv.push_back(0.0); HIPCHK(d_psave.upload(sv, st)); }
    HIPCHK(d_ptimes.alloc((size_t)2 * max_rows)); /* [times | dense-output point counts] */ HIPCHK(d_prows.alloc((size_t)max_rows * row_d));
    HIPCHK(d_wgrec.alloc((size_t)2 * n_wg * 16)); HIPCHK(d_grprec.alloc(2 * 8 * 16)); /* 16 granules per record, double-buffered by generation parity */ HIPCHK(d_pcnt.alloc(10 * 32)); HIPCHK(d_pctl.alloc(2));   /* controller state in; [1]: exit state of a batch with per-sample steps */
    PersistArgs pa; std::memset(&pa, 0, sizeof(pa));
    pa.a = base;
    pa.a.mode = MODE_TRAN; pa.a.maxit = nmaxit; pa.a.abstol = o.abstol; pa.a.reltol = o.reltol; pa.a.newton_tol = 0.1; pa.a.active = nullptr; pa.a.gshunt = 0.0;
    pa.bpw = bpw; pa.wide_l = wide_l; pa.wide_other = wide_other; pa.va_arena = (int)va_arena;
    pa.nblk = nblk; pa.n_wg = n_wg; pa.red_max = (S > 1 || own_steps) ? 1 : 0; pa.wave_doubles = (int)wave_d;
    pa.t1 = t1; pa.dtmin = dtmin; pa.dtmax = dtmax; pa.first_frac = 1e-3; pa.kmax = kmax; pa.max_steps = max_steps;
    pa.bps = d_pbps.p; pa.nbp = (int)bps.size(); pa.saveat = d_psave.p; pa.n_saveat = o.n_saveat;
    pa.ci = d_pci.p; pa.cd = d_pcd.p; pa.n_ci = (int)max_ci; pa.n_cd = (int)max_cd;   // layout sizes (the largest workgroup blob)
    pa.wgc = wg_consts ? d_pwgc.p : nullptr; pa.wgk = wg_consts ? d_pwgk.p : nullptr;
    pa.out_times = d_ptimes.p; pa.out_rows = d_prows.p; pa.max_rows = max_rows; pa.n_obs = n_obs;
    pa.ctl = d_pctl.p; pa.wg_rec = d_wgrec.p; pa.grp_rec = d_grprec.p; pa.counters = d_pcnt.p;
    pa.spin_ticks = 200000000LL;   // 2 s at 100 MHz
    if (const char* sp = std::getenv("CEDARHIP_SPIN_TICKS")) pa.spin_ticks = std::max(1LL, std::atoll(sp));   // test hook: makes every wait give up (exercises the fallback)
    // a batch of single-block samples on a common output grid: every sample its own step sequence (no lock-step, no grid reduction)
    pa.indep = own_steps ? 1 : 0;
    persist_mode = A.nb > 0 ? CH_MODE_BORDERED : (own_steps ? CH_MODE_OWN_STEPS : CH_MODE_LOCKSTEP);
    if (dcm) {
      if (A.nb == 0 || wg_consts) { set_err("internal: operating point on the device stepper is for the bordered form"); return CH_ERR_INTERNAL; }
      std::vector<double> sv, kv, ent;
      eval_sources(0.0, dcm->tran_mode ? 2 : 0, sv, kv);
      ent.assign(kv.begin(), kv.begin() + nk); ent.insert(ent.end(), sv.begin(), sv.begin() + nds);
      HIPCHK(d_pdcent.upload(ent, st));
      pa.dc_mode = 1; pa.dc_maxit = std::max(1, dcm->maxiters); pa.dc_abstol = dcm->abstol; pa.dc_entries = d_pdcent.p;
      pa.dv_max = (!A.mos_hdev.empty() || A.wide) ? dcm->dv_max : 0.0;   // linear circuits take the full Newton step (as on the other paths)
    }
    pa.nb = A.nb; pa.n_glob = A.n_glob; pa.n_bdev = (int)A.border_dev.size();
    for (int q = 0; q < pa.n_bdev; ++q) {
      const Analysis::BorderDev& bd = A.border_dev[q];
      pa.bd_kind[q] = bd.kind; pa.bd_ta[q] = bd.ta; pa.bd_tb[q] = bd.tb;
      pa.bd_val[q] = bd.kind == K_R ? h_dmult0[bd.hdev] / h_dpar0[bd.hdev] : h_dmult0[bd.hdev] * h_dpar0[bd.hdev];
    }
    // Own steps: no grid-wide wait anywhere in the kernel, so the workgroups need not be co-resident — an ordinary launch whose
    // workgroups may queue (behind each other, or behind another process's kernel: a cooperative launch would be refused there)
    const bool coop = !pa.indep;
    pa.pair_dbg = std::getenv("CEDARHIP_PAIR_DBG") ? std::atoi(std::getenv("CEDARHIP_PAIR_DBG")) : 0;
    // initial controller state (same first step as the host stepper)
    TranCtl cs; std::memset(&cs, 0, sizeof(cs));
    const double span = t1 - t0;
    double h = o.dt0 > 0 ? o.dt0 : std::min(dtmax, 1e-3 * span);
    h = std::max(10 * dtmin, std::min(h, (bps[0] - t0) / 50.0) * 1e-3);
    cs.t = t0; cs.h = h; cs.k = 1; cs.nhist = 1; cs.reset_rate = 1; cs.tslot[0] = t0;
    const void* fn = A.wide ? (own_steps ? (pair ? (const void*)tran_persistent_kernel<16, true, PM_OWN, true> : (const void*)tran_persistent_kernel<16, false, PM_OWN, true>)
                                         : (pair ? (const void*)tran_persistent_kernel<16, true, PM_LOCKSTEP, true> : (const void*)tran_persistent_kernel<16, false, PM_LOCKSTEP, true>))
                   : A.nb > 0 ? (pair ? (const void*)tran_persistent_kernel<16, true, PM_BORDER> : (const void*)tran_persistent_kernel<16, false, PM_BORDER>)
                   : own_steps ? (lu_variant <= 12 ? (pair ? (const void*)tran_persistent_kernel<12, true, PM_OWN> : (const void*)tran_persistent_kernel<12, false, PM_OWN>)
                                                   : (pair ? (const void*)tran_persistent_kernel<16, true, PM_OWN> : (const void*)tran_persistent_kernel<16, false, PM_OWN>))
                   : lu_variant <= 12 ? (pair ? (const void*)tran_persistent_kernel<12, true> : (const void*)tran_persistent_kernel<12, false>)
                                      : (pair ? (const void*)tran_persistent_kernel<16, true> : (const void*)tran_persistent_kernel<16, false>);
    {
      hipFuncAttributes fa;
      HIPCHK(hipFuncGetAttributes(&fa, fn));
      if (lds + fa.sharedSizeBytes > 160 * 1024) { set_err("device-resident stepper: LDS footprint"); return CH_OK; }
      HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((160 * 1024 - (int)fa.sharedSizeBytes) & ~255)));
    }
    std::vector<double> hrows, htimes, hpts;
    bool single_batch = false;
    std::vector<std::vector<double>> row_store;   // drained batches when the row buffer fills (no saveat)
    int resume = 0, status = CH_OK;
    for (;;) {
      HIPCHK(hipMemcpyAsync(d_pctl.p, &cs, sizeof(cs), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(d_pctl.p + 1, &cs, sizeof(cs), hipMemcpyHostToDevice, st));
      HIPCHK(hipMemsetAsync(d_pcnt.p, 0, 10 * 32 * sizeof(unsigned), st));
      HIPCHK(hipMemsetAsync(d_ptimes.p + max_rows, 0, (size_t)max_rows * sizeof(double), st));
      // own steps: a block that stops early (DtLessThanMin, MaxIters) never writes its later saveat rows; they read as NaN
      if (pa.indep) HIPCHK(hipMemsetAsync(d_prows.p, 0xff, (size_t)max_rows * row_d * sizeof(double), st));
      HIPCHK(hipMemsetAsync(d_wgrec.p, 0, (size_t)2 * n_wg * 16 * sizeof(double), st)); HIPCHK(hipMemsetAsync(d_grprec.p, 0, 2 * 8 * 16 * sizeof(double), st));   // generation tags start at 0
      pa.resume = resume;
      void* kargs[] = {(void*)&pa};
      HIPCHK(hipEventRecord(ev0, st));
      const hipError_t le = coop ? hipLaunchCooperativeKernel(fn, dim3(n_wg), dim3(PW * 64), kargs, (unsigned)lds, st)
                                 : hipLaunchKernel(fn, dim3(n_wg), dim3(PW * 64), kargs, lds, st);
      if (le != hipSuccess) {
        (void)hipGetLastError();
        if (resume == 0) { set_err(std::string("cooperative launch refused: ") + hipGetErrorString(le)); return CH_OK; }   // fall back to the host stepper
        set_err(std::string("device-resident stepper: relaunch failed: ") + hipGetErrorString(le)); used = true; return CH_ERR_DEVICE;
      }
      HIPCHK(hipEventRecord(ev1, st));
      used = true;
      { const hipError_t se = hipStreamSynchronize(st); if (se != hipSuccess) { set_err(std::string("device-resident stepper: ") + hipGetErrorString(se)); return CH_ERR_DEVICE; } }
      { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); persist_ms += ms; persist_launches += 1; }
      HIPCHK(hipMemcpy(&cs, d_pctl.p + (pa.indep ? 1 : 0), sizeof(cs), hipMemcpyDeviceToHost));
      // rows of this launch
      const size_t nr = (size_t)cs.nsaved;
      const size_t base_t = htimes.size();
      // The usual case — the whole transient in one launch: the rows are transposed on the device into the result's layout
      // [observable][time][sample] and cross PCIe once, straight into the result (the host-side transposition of a result with
      // every node observed, 100 MB for the 1024-DFF array, cost several times the solve).  Drained batches keep the host path.
      single_batch = resume == 0 && cs.exit_reason != PX_ROWS_FULL && !dcm;
      htimes.resize(base_t + nr); hpts.resize(base_t + nr);
      if (!single_batch) hrows.resize((base_t + nr) * row_d);
      if (nr > 0) {
        HIPCHK(hipMemcpy(htimes.data() + base_t, d_ptimes.p, nr * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hpts.data() + base_t, d_ptimes.p + max_rows, nr * sizeof(double), hipMemcpyDeviceToHost));
        if (!single_batch) HIPCHK(hipMemcpy(hrows.data() + base_t * row_d, d_prows.p, nr * row_d * sizeof(double), hipMemcpyDeviceToHost));
      }
      if (cs.exit_reason == PX_ROWS_FULL) { cs.nsaved = 0; resume = 1; continue; }
      if (cs.exit_reason == PX_ABORT) {
        persist_aborted = true;
        unsigned code = 0; (void)hipMemcpy(&code, d_pcnt.p + 9 * 32, sizeof(code), hipMemcpyDeviceToHost);
        set_err("device-resident stepper: a wait exceeded its bound (site " + std::to_string(code & 255u) + ", sequence " + std::to_string(code >> 8) +
                ", attempts " + std::to_string((long long)cs.n_attempts) + "; workgroups not co-resident?)");
        status = CH_ERR_DEVICE;
      }
      else status = cs.status;
      break;
    }
    if (dcm) {
      if (dc_iters) *dc_iters = cs.sum_iters;
      if (cs.exit_reason == PX_ABORT && status == CH_OK) status = CH_ERR_DEVICE;
      return status;
    }
    persist_attempts = cs.n_attempts;
    persist_barrier_s = (double)cs.t_cycles_barrier * 1e-8;
#ifdef CH_STAMPS
    { static const char* nm[12] = {"set-up", "coefficients", "sources", "predictor", "eval", "gather", "rows+norm", "LU+solves", "update", "candidate", "grid-reduce", "controller"};
      std::fprintf(stderr, "[pstamps] attempts %lld; cycles per attempt (wave 0 of workgroup 0):", cs.n_attempts);
      for (int q = 0; q < 12; ++q) std::fprintf(stderr, " %s %.0f", nm[q], (double)cs.stamps[q] / (double)std::max<long long>(1, cs.n_attempts));
      std::fprintf(stderr, "\n"); }
#endif
    R.stats.naccept += cs.naccept; R.stats.nreject += cs.nreject; R.stats.nnonlinconvfail += cs.nconvfail;
    const long long arr_iters = (own_steps && S == 1) ? cs.max_iters : cs.sum_iters;   // one circuit: Newton iterations of its slowest block
    R.stats.n_block_iters += cs.sum_block_iters; R.stats.nnonliniter += arr_iters; R.stats.nf += arr_iters; R.stats.njacs += arr_iters;
    R.stats.nfactors += arr_iters; R.stats.nsolve += arr_iters;
    const size_t nt = htimes.size();
    R.times = htimes;
    R.pts.assign(nt, 0);
    if (own_steps == false) for (size_t r = 0; r < nt; ++r) R.pts[r] = (int32_t)hpts[r];
    R.values.assign((size_t)n_obs * nt * S, 0.0);
    if (single_batch) {
      const size_t n = (size_t)n_obs * nt * S;
      if (n > 0) {
        HIPCHK(d_ptrans.alloc(n));
        hipLaunchKernelGGL(transpose_rows_kernel, dim3((unsigned)std::min<size_t>(65535, (n + 255) / 256)), dim3(256), 0, st, (const double*)d_prows.p, d_ptrans.p, (long)nt, (long)n_obs, S);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(R.values.data(), d_ptrans.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        // the rows stay in HBM for a device-side consumer (the RCCL gather of a sharded sweep) when every observable is a plain
        // unknown — the host-side fix-ups of finish_tran (merged nodes, known nodes, eliminated branches) do not reach this buffer
        bool plain = true;
        for (int ob = 0; ob < n_obs && plain; ++ob) {
          if (obs_primary[ob] != ob) plain = false;
          else if (obs_kind[ob] == 0 && A.node_unknown[obs_index[ob]] < 0) plain = false;
          else if (obs_kind[ob] == 1) { const int br = dev[obs_index[ob]].branch; if (br < 0 || A.branch_unknown[br] < 0) plain = false; }
        }
        if (plain) { R.dev_values = d_ptrans.p; R.dev_n = (int64_t)n; }
      }
    } else
    for (size_t r = 0; r < nt; ++r) for (int ob = 0; ob < n_obs; ++ob) std::memcpy(&R.values[((size_t)ob * nt + r) * S], &hrows[(r * n_obs + ob) * S], S * sizeof(double));
    return finish_tran(R, 0, cs.t, status, tstart);
  }

  // Operating point of the torn form on the device stepper: ONE damped Newton solve (CedarDCOp's first attempt: from o.x0 or
  // 1e-7*randn) with the Schur complement on the border per iteration.  Any other outcome than CH_OK sends the caller to the
  // sparse path's full CedarDCOp (restarts, gmin stepping).  The result stays in ring slot 0 of THIS (torn) circuit.
  bool keep_slot0 = false;
  int dc_border(const ch_dc_opts& o, long long* iters) {
    const auto td0 = hclock::now();
    auto lap = [&](const char* what) { if (std::getenv("CEDARHIP_DEBUG_TORN")) std::fprintf(stderr, "[torn] dc_border %s at %.3f ms\n", what, 1e3 * std::chrono::duration<double>(hclock::now() - td0).count()); };
    int rc = finalize_params();
    if (rc != CH_OK) return rc;
    lap("finalized");
    std::string why;
    if (!is_torn || !persist_eligible(why, false)) { set_err("bordered operating point: " + why); return CH_ERR_UNSUPPORTED; }
    std::vector<double> xm((size_t)S * A.n_mna, 0.0);
    if (o.x0) std::copy(o.x0, o.x0 + xm.size(), xm.begin());
    else { Rng rng(o.seed); for (double& v : xm) v = 1e-7 * rng.normal(); }
    lap("start vector");
    rc = upload_from_mna(0, xm.data());
    if (rc != CH_OK) return rc;
    lap("uploaded");
    ch_tran_opts to; std::memset(&to, 0, sizeof(to));
    to.abstol = 1e-6; to.reltol = 1e-3; to.max_order = 1;
    ch_result tmp; tmp.S = S; tmp.n_obs = (int)obs_kind.size();
    const std::vector<double> one_bp{1.0};
    const double save_ms = persist_ms; const long save_l = persist_launches;
    bool used = false;
    const auto tq0 = hclock::now();
    rc = tran_persistent(0.0, 1.0, to, tmp, one_bp, 1, 1e-15, 0.1, 10, 10, hclock::now(), used, &o, iters);
    if (std::getenv("CEDARHIP_DEBUG_TORN")) std::fprintf(stderr, "[torn] dc_border: kernel %.3f ms, call %.3f ms\n", persist_ms - save_ms, 1e3 * std::chrono::duration<double>(hclock::now() - tq0).count());
    persist_ms = save_ms; persist_launches = save_l;
    if (!used && rc == CH_OK) return CH_ERR_UNSUPPORTED;
    return rc;
  }

  // A coupled array behind a border of one or two unknowns: operating point and transient on the torn companion's device-resident
  // stepper (operating point on this circuit's sparse path when the single damped Newton solve there does not converge).
  // used = false: the companion does not take the problem (reason in torn_note).
  int tran_torn(double t0, double t1, const ch_tran_opts& o, ch_result& R, bool& used) {
    used = false;
    auto tstart = hclock::now();
    int rc = finalize_params();
    if (rc != CH_OK) return rc;
    std::vector<double> x_mna((size_t)S * A.n_mna, 0.0);
    ch_stats dcst; std::memset(&dcst, 0, sizeof(dcst));
    device_ms = 0; n_launch = 0; n_timed = 0;
    ch_circuit* tc = torn_c.get();
    bool dc_on_torn = false;
    if (o.skip_dc) { if (o.dc.x0) std::copy(o.dc.x0, o.dc.x0 + x_mna.size(), x_mna.begin()); }
    else {
      if (std::getenv("CEDARHIP_TORN_DC_SPARSE") == nullptr) {
        long long it = 0;
        ArenaScope sc(&tc->arena);
        const int r = tc->dc_border(o.dc, &it);
        if (r == CH_OK) { dc_on_torn = true; dcst.nnonliniter = it; dcst.nf = dcst.njacs = dcst.nfactors = dcst.nsolve = it; dcst.n_block_iters = it * tc->A.n_comp; }
        else { torn_note = "bordered operating point: " + err(); ctx->err.clear(); }
        if (std::getenv("CEDARHIP_DEBUG_TORN")) std::fprintf(stderr, "[torn] operating point on the device stepper: rc %d, %lld iterations, %.3f ms%s%s\n", r, it,
                                                             1e3 * std::chrono::duration<double>(hclock::now() - tstart).count(), r == CH_OK ? "" : " -> sparse path: ", r == CH_OK ? "" : torn_note.c_str());
      }
      if (!dc_on_torn) {
        rc = dc_solve(o.dc, 0, nullptr, &dcst);
        if (rc != CH_OK) { used = true; return rc; }
        rc = download_mna(0, t0, 1, x_mna.data());
        if (rc != CH_OK) return rc;
      }
    }
    const double dc_s = std::chrono::duration<double>(hclock::now() - tstart).count();
    const long dc_l = n_launch;
    ch_tran_opts o2 = o; o2.skip_dc = 1; o2.dc.x0 = dc_on_torn ? nullptr : x_mna.data(); o2.stepper = CH_STEPPER_DEVICE;
    { ArenaScope sc(&tc->arena); tc->keep_slot0 = dc_on_torn; rc = tc->tran_solve(t0, t1, o2, R); tc->keep_slot0 = false; }
    if (rc == CH_ERR_UNSUPPORTED || (rc == CH_ERR_DEVICE && tc->persist_aborted)) { torn_note = err(); ctx->err.clear(); return CH_OK; }   // the sparse path takes it
    used = true;
    R.stats.dc_seconds = dc_s; R.stats.wall_seconds += dc_s; R.stats.n_kernel_launches += dc_l;
    R.stats.nf += dcst.nf; R.stats.njacs += dcst.njacs; R.stats.nfactors += dcst.nfactors; R.stats.nsolve += dcst.nsolve;
    R.stats.nnonliniter += dcst.nnonliniter; R.stats.nrestarts += dcst.nrestarts; R.stats.n_block_iters += dcst.n_block_iters;
    return rc;
  }

  // ------------------------------------------------------------------------------------------
  // what no solver below should have to defend against: non-finite spans and tolerances, an output grid that is not a grid
  int check_tran_opts(double t0, double t1, const ch_tran_opts& o) {
    if (!std::isfinite(t0) || !std::isfinite(t1) || !(t1 > t0)) { set_err("tspan must be finite and increasing"); return CH_ERR_INVALID; }
    if (!(o.abstol >= 0) || !(o.reltol >= 0) || !std::isfinite(o.abstol) || !std::isfinite(o.reltol) || o.abstol + o.reltol == 0) { set_err("abstol and reltol must be finite, non-negative and not both zero"); return CH_ERR_INVALID; }
    if (!(o.dtmin >= 0) || !(o.dtmax >= 0) || !(o.dt0 >= 0) || !std::isfinite(o.dtmin) || !std::isfinite(o.dtmax) || !std::isfinite(o.dt0)) { set_err("dtmin, dtmax and dt0 must be finite and non-negative (0 = automatic)"); return CH_ERR_INVALID; }
    if (o.n_saveat < 0 || (o.n_saveat > 0 && !o.saveat)) { set_err("saveat: n_saveat points announced, none given"); return CH_ERR_INVALID; }
    for (int i = 0; i < o.n_saveat; ++i)
      if (!std::isfinite(o.saveat[i]) || (i > 0 && o.saveat[i] < o.saveat[i - 1])) { set_err("saveat must be finite and non-decreasing"); return CH_ERR_INVALID; }
    if (o.stepper < CH_STEPPER_AUTO || o.stepper > CH_STEPPER_DEVICE) { set_err("stepper must be CH_STEPPER_AUTO, _HOST or _DEVICE"); return CH_ERR_INVALID; }
    if (o.step_control != CH_STEPS_AUTO && o.step_control != CH_STEPS_SHARED) { set_err("step_control must be CH_STEPS_AUTO or CH_STEPS_SHARED"); return CH_ERR_INVALID; }
    return CH_OK;
  }
  int tran_solve(double t0, double t1, const ch_tran_opts& o, ch_result& R) {
    { const int vrc = check_tran_opts(t0, t1, o); if (vrc != CH_OK) return vrc; }
    if (torn_c && !is_torn && S == 1 && std::getenv("CEDARHIP_NO_TEAR") == nullptr) {
      const char* ev = std::getenv("CEDARHIP_STEPPER");
      int want = o.stepper;
      if (want == CH_STEPPER_AUTO && ev) want = std::strcmp(ev, "host") == 0 ? CH_STEPPER_HOST : CH_STEPPER_AUTO;
      if (want != CH_STEPPER_HOST) {
        bool used = false;
        const int rc = tran_torn(t0, t1, o, R, used);
        if (used || rc != CH_OK) return rc;
      }
    }
    auto tstart = hclock::now();
    R.times.clear(); R.pts.clear(); R.values.clear(); R.final_state.clear();   // (a launch that gave up may have left the rows of its first attempt)
    std::memset(&R.stats, 0, sizeof(R.stats));
    R.S = S; R.n_obs = (int)obs_kind.size();
    device_ms = 0; n_launch = 0; n_timed = 0;
    persist_ms = 0; persist_launches = 0; persist_attempts = 0; persist_barrier_s = 0;
    int rc = finalize_params();
    if (rc != CH_OK) return rc;
    hipStream_t st = ctx->stream;
    const int kmax = std::min(5, std::max(1, o.max_order));
    const double span = t1 - t0;
    if (!(span > 0)) { set_err("tspan must be increasing"); return CH_ERR_INVALID; }
    const double dtmax = o.dtmax > 0 ? o.dtmax : span / 10.0, dtmin = o.dtmin > 0 ? o.dtmin : 1e-15 * span;
    const int max_steps = o.max_steps > 0 ? o.max_steps : 100000 /* Sundials.jl's default maxiters of solve(prob, IDA()) */, nmaxit = o.newton_maxiters > 0 ? o.newton_maxiters : 10;
    const int n_obs = R.n_obs;

    // ring bookkeeping: order[] lists slots newest-first
    int order[NSLOT]; for (int i = 0; i < NSLOT; ++i) order[i] = i;
    double htime[NSLOT] = {0};
    int nhist = 1;
    // ---- initialisation ----
    if (o.skip_dc) {
      if (o.dc.x0) { rc = upload_from_mna(order[0], o.dc.x0); if (rc != CH_OK) return rc; }
      else if (!keep_slot0) HIPCHK(hipMemsetAsync(d_X.p + (size_t)order[0] * S * A.n_unk, 0, (size_t)S * A.n_unk * sizeof(double), st));
    } else {
      rc = dc_solve(o.dc, order[0], nullptr, &R.stats);
      if (rc != CH_OK) return rc;
    }
    R.stats.dc_seconds = std::chrono::duration<double>(hclock::now() - tstart).count();
    dc_block_iters = R.stats.n_block_iters;
    // charges at t0 in the problem's own mode
    {
      NewtonArgs a = base; a.mode = MODE_EVAL; a.maxit = 1; a.hist_slot[0] = order[0]; a.cand_slot = order[0]; a.abstol = o.abstol; a.reltol = o.reltol;
      rc = set_sources(a, t0, 1); if (rc != CH_OK) return rc;
      Summary sm; rc = run_newton(a, nullptr, sm); if (rc != CH_OK) return rc;
      R.stats.nf += S;
    }
    htime[0] = t0;
    dc_device_ms = device_ms; dc_launches = n_launch; dc_timed = n_timed;   // everything so far was initialisation

    // break points of every sample's sources, each with its code: < 0 = some source VALUE jumps there (the integrator restarts at order 1
    // behind it); >= 0 = a continuous corner (landed on exactly, stepped over with the history kept — IDA's treatment of `tstops`,
    // src/spectre_env.jl:71-77) and the code is the length of the shortest source segment that starts there (the first step behind
    // the corner is capped at a tenth of it).  A source is asked only about its own times: linear in the number of points.
    std::vector<double> bps, bpc;
    {
      const int nsrc = (int)src.size();
      std::vector<std::pair<double, double>> pts;
      for (int s = 0; s < Ssrc; ++s) for (int i = 0; i < nsrc; ++i) source_breakpoint_codes(src[i], &h_src_par[((size_t)s * nsrc + i) * CH_SRC_NPAR], t0, t1, pts);
      merge_breakpoints(pts, t1, bps, bpc);
    }
    size_t ibp = 0;
    // every transient starts its blocks' LU pivot orders from the identity (see ch_persist.hpp: identical blocks stay identical)
    if (d_perm.p) HIPCHK(hipMemsetAsync(d_perm.p, 0, (size_t)A.n_comp * S * 16, ctx->stream));

    // ---- device-resident step controller (ch_persist.hpp) where the circuit qualifies ----
    {
      const char* ev = std::getenv("CEDARHIP_STEPPER");
      int want = o.stepper;
      if (want == CH_STEPPER_AUTO && ev) want = std::strcmp(ev, "host") == 0 ? CH_STEPPER_HOST : (std::strcmp(ev, "device") == 0 ? CH_STEPPER_DEVICE : CH_STEPPER_AUTO);
      if (want != CH_STEPPER_HOST) {
        std::string why;
        if (persist_eligible(why, persist_own_steps(o))) {
          bool used = false;
          persist_aborted = false;
          rc = tran_persistent(t0, t1, o, R, bps, kmax, dtmin, dtmax, max_steps, nmaxit, tstart, used, nullptr, nullptr, &bpc);
          if (used && persist_aborted && want != CH_STEPPER_DEVICE) {
            // A grid-wide wait ran into its bound: the cooperative launch shared the GPU with another process's kernel and
            // its workgroups were not all resident.  The solve is repeated on the host stepper (whose launches need no
            // co-residency); the torn form hands back to the sparse path of its parent.
            const std::string msg = err();
            ctx->err.clear();
            if (is_torn) { set_err(msg); return CH_ERR_UNSUPPORTED; }
            ch_tran_opts o3 = o; o3.stepper = CH_STEPPER_HOST;
            return tran_solve(t0, t1, o3, R);
          }
          if (used) return rc;
          why = err();
        }
        if (want == CH_STEPPER_DEVICE || is_torn) { set_err("device-resident stepper not available for this circuit: " + why); return CH_ERR_UNSUPPORTED; }
        if (std::getenv("CEDARHIP_DEBUG_STEPPER")) std::fprintf(stderr, "[stepper] host stepper because: %s\n", why.c_str());
      }
      if (is_torn) { set_err("the torn form of a circuit runs on the device-resident stepper only"); return CH_ERR_UNSUPPORTED; }
    }

    // saved observables live on the device until the end
    struct ChunkList : std::vector<double*> { ~ChunkList() { for (double* p : *this) (void)hipFree(p); } } chunks;   // freed on every exit, exceptions included
    const int CH = 512; long nsaved = 0;
    auto row_ptr = [&](long row, double** out) -> int {  // device address of saved-row `row`, growing the buffer by chunks
      while ((size_t)(row / CH) >= chunks.size()) { double* p = nullptr; HIPCHK(hipMalloc((void**)&p, std::max<size_t>(1, (size_t)CH * n_obs * S) * sizeof(double))); chunks.push_back(p); }
      *out = chunks[row / CH] + (size_t)(row % CH) * n_obs * S;
      return CH_OK;
    };
    auto save = [&](double ts, const int* slots, const double* w, int nw) -> int {
      double* dst = nullptr;
      int r0 = row_ptr(nsaved, &dst); if (r0 != CH_OK) return r0;
      if (n_obs > 0) {
        ObsArgs oa; oa.X = d_X.p; oa.slot_stride = (long)S * A.n_unk; oa.nw = nw; oa.n_unk = A.n_unk; oa.S = S; oa.n_obs = n_obs; oa.obs_unk = d_obs_unk.p;
        for (int j = 0; j < nw; ++j) { oa.slots[j] = slots[j]; oa.w[j] = w[j]; }
        oa.dst = dst;
        const int n = n_obs * S;
        hipLaunchKernelGGL(save_obs_kernel, dim3((n + 255) / 256), dim3(256), 0, st, oa);
      }
      R.times.push_back(ts); R.pts.push_back(0); ++nsaved;
      return CH_OK;
    };
    auto free_chunks = [&]() { for (double* p : chunks) (void)hipFree(p); chunks.clear(); };
    int isave = 0;
    { const double one = 1.0; const int s0 = order[0];
      if (o.n_saveat == 0) { rc = save(t0, &s0, &one, 1); if (rc) { free_chunks(); return rc; } }
      else while (isave < o.n_saveat && o.saveat[isave] <= t0) { rc = save(o.saveat[isave], &s0, &one, 1); if (rc) { free_chunks(); return rc; } ++isave; } }

    const double kFirstFrac = 1e-3;
    double t = t0;
    double h = o.dt0 > 0 ? o.dt0 : std::min(dtmax, 1e-3 * span);
    h = std::max(10 * dtmin, std::min(h, (bps[0] - t0) / 50.0) * kFirstFrac);
    int k = 1, steps_at_order = 0, status = CH_OK;
    double tau[9];
    NewtonArgs a = base;
    a.mode = MODE_TRAN; a.maxit = nmaxit; a.abstol = o.abstol; a.reltol = o.reltol; a.newton_tol = 0.1;
    bool reset_rate = true;  // convergence rates unknown at the start and after every restart

    for (int step = 0; step < max_steps && t < t1;) {
      while (ibp < bps.size() && bps[ibp] <= t * (1 + 1e-15) + 1e-300) ++ibp;
      const double tb = ibp < bps.size() ? bps[ibp] : t1;
      const double tb_code = ibp < bps.size() ? bpc[ibp] : -1.0;
      const bool tb_jump = tb_code < 0;
      bool hit_bp = false;
      double tn = t + h;
      if (tn >= tb - 1e-3 * h) { tn = tb; hit_bp = true; }
      const double hh = tn - t;
      if (hh < dtmin) { status = CH_ERR_DTMIN; break; }
      const int nh = nhist, kk = std::min(k, nh), np = std::min(kk + 1, nh);
      tau[0] = tn; for (int j = 0; j < nh && j < 7; ++j) tau[j + 1] = htime[j];
      extrap_weights(tau, np, a.wpred); a.npred = np;
      bdf_coeffs(tau, kk, a.alpha); a.k = kk;
      const bool lte = np >= kk + 1;
      a.ck = lte ? hh / (tn - tau[kk + 1]) : 0.0;
      a.nkm1 = 0; a.nkp1 = 0; a.ckm1 = 0; a.ckp1 = 0;
      const bool try_up = lte && kk < kmax && nh >= kk + 2 && steps_at_order + 1 >= kk + 1;
      if (lte && kk > 1) { extrap_weights(tau, kk, a.wkm1); a.nkm1 = kk; a.ckm1 = hh / (tn - tau[kk]); }
      if (try_up) { extrap_weights(tau, kk + 2, a.wkp1); a.nkp1 = kk + 2; a.ckp1 = hh / (tn - tau[kk + 2]); }
      for (int j = 0; j < 7; ++j) a.hist_slot[j] = order[std::min(j, nh - 1)];
      a.cand_slot = order[NSLOT - 1];
      // landing on a break point uses the sources' left limit there; the jump (if any) is crossed by the restart step
      rc = set_sources(a, hit_bp ? std::nextafter(tn, -INFINITY) : tn, 1); if (rc != CH_OK) { status = rc; break; }
      a.reset_rate = reset_rate ? 1 : 0;
      a.obs_row = nullptr;
      if (o.n_saveat == 0 && n_obs > 0) { rc = row_ptr(nsaved, &a.obs_row); if (rc != CH_OK) { status = rc; break; } }  // candidate row, kept on accept
      Summary sm;
      rc = run_newton(a, nullptr, sm); if (rc != CH_OK) { status = rc; break; }
      R.stats.n_step_attempts++;
      R.stats.n_block_iters += sm.sum_block_iters; R.stats.nnonliniter += sm.sum_iters; R.stats.nf += sm.sum_iters; R.stats.njacs += sm.sum_iters; R.stats.nfactors += sm.sum_iters; R.stats.nsolve += sm.sum_iters;
      if (sm.n_fail > 0) {
        R.stats.nnonlinconvfail++; reset_rate = true;
        h = hh * 0.25; k = 1; steps_at_order = 0;
        if (nhist > 2) nhist = 2;
        continue;
      }
      const double errk = lte ? sm.errk : 0.0;
      if (errk > 1.0) {
        R.stats.nreject++;
        // IDA-style: aim at half the tolerance after a failed error test, shrink by at most 4x
        const double fac = 0.9 * std::pow(2.0 * errk + 1e-4, -1.0 / (kk + 1));
        h = hh * std::min(0.9, std::max(0.25, fac));
        steps_at_order = 0;
        continue;
      }
      // ---- accept: candidate slot becomes the newest history point ----
      R.stats.naccept++; ++step; reset_rate = false;
      { int cand = order[NSLOT - 1]; for (int j = NSLOT - 1; j > 0; --j) { order[j] = order[j - 1]; htime[j] = htime[j - 1]; } order[0] = cand; htime[0] = tn; nhist = std::min(nhist + 1, kmax + 2); }
      if (o.n_saveat > 0) {
        while (isave < o.n_saveat && o.saveat[isave] <= tn * (1 + 1e-15)) {
          const double ts = o.saveat[isave];
          double tt[9], ww[9]; const int m = std::min(kk, nh) + 1;
          tt[0] = ts; for (int j = 0; j < m; ++j) tt[j + 1] = htime[j];
          extrap_weights(tt, m, ww);
          rc = save(ts, order, ww + 1, m); if (rc) break;
          ++isave;
        }
      } else if (n_obs > 0) { R.times.push_back(tn); R.pts.push_back(std::min(kk, nh) + 1); ++nsaved; }  // the kernel's epilogue already wrote this row
      else { const double one = 1.0; rc = save(tn, order, &one, 1); if (rc == CH_OK) R.pts.back() = std::min(kk, nh) + 1; }
      if (rc != CH_OK) { status = rc; break; }
      // ---- order / step selection ----
      const double fac_k = std::pow(2.0 * errk + 1e-4, -1.0 / (kk + 1));  // puts the error at half the tolerance
      double best = fac_k; int knew = kk;
      if (lte) {
        ++steps_at_order;
        if (kk > 1) { const double f = std::pow(2.0 * sm.errkm1 + 1e-4, -1.0 / kk); if (f > best) { best = f; knew = kk - 1; } }
        if (try_up) { const double f = std::pow(2.0 * sm.errkp1 + 1e-4, -1.0 / (kk + 2)); if (f > 1.1 * best) { best = f; knew = kk + 1; } }
      } else knew = 1;
      if (knew != kk) steps_at_order = 0;
      k = knew;
      if (best > 1.0 && best < 1.2) best = 1.0;  // dead band: keep h when the suggested change is small
      h = std::min(dtmax, hh * std::min(kk == 1 ? 10.0 : 2.0, std::max(0.5, best)));
      t = tn;
      if (hit_bp && t < t1 && !tb_jump) {   // continuous corner: history and order are kept, the first step behind it is capped (oracle.cpp)
        reset_rate = true;
        h = std::max(dtmin * 10, std::min(h, tb_code / 10.0));
      }
      if (hit_bp && t < t1 && tb_jump) {
        nhist = 1; k = 1; steps_at_order = 0; reset_rate = true;
        double nb = t1;
        for (size_t b = ibp; b < bps.size(); ++b) if (bps[b] > t * (1 + 1e-15)) { nb = bps[b]; break; }
        h = std::max(dtmin * 10, std::min(h, (nb - t) / 50.0) * kFirstFrac);
      }
    }
    if (status == CH_OK && t < t1) status = CH_ERR_MAXSTEPS;
    // ---- collect results ----
    (void)hipStreamSynchronize(st);
    const size_t nt = R.times.size();
    R.values.assign((size_t)n_obs * nt * S, 0.0);
    {
      std::vector<double> buf((size_t)CH * n_obs * S);
      for (size_t cidx = 0; cidx < chunks.size(); ++cidx) {
        const size_t rows = std::min<size_t>(CH, nt - cidx * CH);
        if (n_obs == 0 || rows == 0) continue;
        (void)hipMemcpy(buf.data(), chunks[cidx], rows * n_obs * S * sizeof(double), hipMemcpyDeviceToHost);
        for (size_t r = 0; r < rows; ++r) for (int ob = 0; ob < n_obs; ++ob)
          std::memcpy(&R.values[((size_t)ob * nt + cidx * CH + r) * S], &buf[(r * n_obs + ob) * S], S * sizeof(double));
      }
      free_chunks();
    }
    return finish_tran(R, order[0], t, status, tstart);
  }

  // shared end of both step controllers: derived observables, final state, statistics
  double persist_ms = 0; long persist_launches = 0; long long persist_attempts = 0; double persist_barrier_s = 0;
  double dc_device_ms = 0; long dc_launches = 0, dc_timed = 0; long long dc_block_iters = 0;
  int finish_tran(ch_result& R, int newest_slot, double t, int status, hclock::time_point tstart) {
    const int n_obs = R.n_obs;
    const size_t nt = R.times.size();
    {
      // several observables fed by one unknown (merged nodes): the kernel writes the primary one only
      for (int ob = 0; ob < n_obs; ++ob) if (obs_primary[ob] != ob)
        std::memcpy(&R.values[(size_t)ob * nt * S], &R.values[(size_t)obs_primary[ob] * nt * S], nt * S * sizeof(double));
      // observables that are known nodes / ground are evaluated on the host
      const int nk = (int)A.known.size();
      std::vector<double> sv, kv;
      for (int ob = 0; ob < n_obs; ++ob) {
        if (obs_kind[ob] == 0 && A.node_unknown[obs_index[ob]] < 0) {
          const int kn = A.node_known[obs_index[ob]];
          for (size_t it = 0; it < nt; ++it) { eval_sources(R.times[it], 1, sv, kv); for (int s = 0; s < S; ++s) R.values[((size_t)ob * nt + it) * S + s] = kv[(size_t)(Ssrc > 1 ? s : 0) * nk + kn]; }
        } else if (obs_kind[ob] == 1) {
          const int b = dev[obs_index[ob]].branch;
          if (b < 0 || A.branch_unknown[b] < 0) for (size_t it = 0; it < nt; ++it) for (int s = 0; s < S; ++s) R.values[((size_t)ob * nt + it) * S + s] = CH_NAN;
        }
      }
    }
    R.final_state.assign((size_t)S * A.n_mna, 0.0);
    download_mna(newest_slot, t, 1, R.final_state.data());
#ifdef CH_STAMPS
    { unsigned long long hs[8]; (void)hipMemcpy(hs, d_stamps.p, sizeof(hs), hipMemcpyDeviceToHost);
      std::fprintf(stderr, "[stamps] cycles summed over blocks: prologue %llu eval %llu gather %llu solve %llu epilogue %llu arrival %llu pre-LU %llu LU %llu ; launches %ld blocks %d\n", hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7], n_launch, A.n_comp * S); }
#endif
    R.status = status;
    R.stats.wall_seconds = std::chrono::duration<double>(hclock::now() - tstart).count();
    if (host_profile) { std::fprintf(stderr, "[host profile] wall %.3f ms; in run_newton: launch %.3f ms, wait %.3f ms, events+reduce %.3f ms; launches %ld\n", 1e3 * R.stats.wall_seconds, 1e3 * prof_launch, 1e3 * prof_wait, 1e3 * prof_reduce, n_launch); prof_launch = prof_wait = prof_reduce = 0; }
    R.stats.device_seconds = (n_timed > 0 ? device_ms * 1e-3 * (double)n_launch / (double)n_timed : 0.0) + persist_ms * 1e-3;  // sampled launches scaled + the persistent launches (exact)
    R.stats.n_kernel_launches = n_launch + persist_launches;
    R.stats.n_step_attempts += persist_attempts; R.stats.barrier_seconds = persist_barrier_s;
    R.stats.stepper = persist_launches > 0 ? CH_STEPPER_DEVICE : CH_STEPPER_HOST;
    R.stats.stepper_mode = persist_launches > 0 ? persist_mode : 0;
    R.stats.step_block_iters = R.stats.n_block_iters - dc_block_iters;
    if (persist_launches > 0) { R.stats.step_kernel_seconds = persist_ms * 1e-3; R.stats.step_kernel_launches = persist_launches; }
    else {
      const long nl = n_launch - dc_launches, ntm = n_timed - dc_timed;
      R.stats.step_kernel_launches = nl;
      R.stats.step_kernel_seconds = ntm > 0 ? (device_ms - dc_device_ms) * 1e-3 * (double)nl / (double)ntm : 0.0;
    }
    if (status != CH_OK && err().empty()) set_err(status == CH_ERR_DTMIN ? "step size underflow (DtLessThanMin)" : "transient did not reach t1");
    return status;
  }
};

// Exception barrier of the C-ABI (include/cedarhip.h: "they never throw").  Host-side containers sized by the caller's
// input (samples x unknowns, blocks x ds^2, ...) can throw std::bad_alloc / std::length_error; letting that unwind through
// ctypes, a C client or a Julia ccall would end the host process in std::terminate.  The message goes to ch_last_error without
// allocating when memory is the problem.
static void set_err_nothrow(ch_ctx* ctx, const char* what, const char* detail) noexcept {
  if (!ctx) return;
  try { ctx->err = what; if (detail && *detail) { ctx->err += ": "; ctx->err += detail; } }
  catch (...) { ctx->err.clear(); }   // clear() does not allocate
}
template <class F>
static int guard_rc(ch_ctx* ctx, F&& body) noexcept {
  try { return body(); }
  catch (const std::bad_alloc&) { set_err_nothrow(ctx, "out of host memory inside the engine call", ""); return CH_ERR_NOMEM; }
  catch (const std::length_error& e) { set_err_nothrow(ctx, "a host container would exceed its maximum size (sample count x circuit size too large)", e.what()); return CH_ERR_NOMEM; }
  catch (const std::exception& e) { set_err_nothrow(ctx, "internal error (C++ exception caught at the C-ABI)", e.what()); return CH_ERR_INTERNAL; }
  catch (...) { set_err_nothrow(ctx, "internal error (unknown C++ exception caught at the C-ABI)", ""); return CH_ERR_INTERNAL; }
}

// =============================================================================================
extern "C" {

void ch_dc_opts_default(ch_dc_opts* o) { std::memset(o, 0, sizeof(*o)); o->abstol = 1e-10; o->maxiters = 200; o->n_restarts = 10; o->seed = 10; o->tran_mode = 0; o->dv_max = 2.0; o->x0 = nullptr; }
void ch_tran_opts_default(ch_tran_opts* o) { std::memset(o, 0, sizeof(*o)); o->abstol = 1e-6; o->reltol = 1e-3; o->max_order = 5; o->newton_maxiters = 10; ch_dc_opts_default(&o->dc); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the (process-global) kernel function, not of a launch: it is set
// once per device to the largest size the path decision admits (finalize_params: 150 KB; ac_block_kernel<64>: 2*96*97 doubles),
// so circuits with different LDS footprints can live side by side in one process.
static hipError_t raise_lds_ceilings() {
  const int lds_cu = 160 * 1024;   // LDS of one gfx950 CU; a kernel's static __shared__ variables come out of the same budget
  const void* fns[] = {(const void*)newton_block_kernel<8>, (const void*)newton_block_kernel<12>, (const void*)newton_block_kernel<16>,
                       (const void*)newton_block_kernel<32>, (const void*)newton_block_kernel<0>, (const void*)newton_block_kernel<16, true>,
                       (const void*)newton_block_kernel<0, true>, (const void*)ac_block_kernel<64>};
  for (const void* f : fns) {
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, f);
    if (e != hipSuccess) return e;
    const int cap = (lds_cu - (int)fa.sharedSizeBytes) & ~255;
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

static ch_ctx* ch_create_impl(int device_id, char* err, size_t errlen) {
  auto fail = [&](const std::string& m) -> ch_ctx* { if (err && errlen) { std::snprintf(err, errlen, "%s", m.c_str()); } return nullptr; };
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(std::string("cedarhip needs a HIP device (gfx950); none available: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
  if (device_id < 0 || device_id >= n) return fail("invalid device id");
  if ((e = hipSetDevice(device_id)) != hipSuccess) return fail(hipGetErrorString(e));
  if ((e = raise_lds_ceilings()) != hipSuccess) return fail(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
  ch_ctx* c = new ch_ctx();
  c->device = device_id;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { delete c; return fail(hipGetErrorString(e)); }
  return c;
}
void ch_destroy(ch_ctx* c) { if (!c) return; if (c->stream) (void)hipStreamDestroy(c->stream); delete c; }
const char* ch_last_error(ch_ctx* c) { return c ? c->err.c_str() : "null context"; }

static ch_circuit* ch_circuit_build_impl(ch_ctx* ctx, const ch_desc* d, bool tear = false) {
  if (!ctx || !d) return nullptr;
  ctx->err.clear();
  (void)hipSetDevice(ctx->device);
  std::unique_ptr<ch_circuit> owner(new ch_circuit());   // released on every early return and on an exception
  ch_circuit* c = owner.get();
  c->ctx = ctx;
  ArenaScope arena_scope(&c->arena);
  c->n_nodes = d->n_nodes; c->temp = d->temp; c->gmin = d->gmin; c->scale = d->scale;
  for (int i = 0; i < d->n_src; ++i) {
    HSource s; s.kind = d->src_kind[i]; s.dc = d->src_dc[i];
    for (int k = 0; k < CH_SRC_NPAR; ++k) s.par[k] = d->src_par[i * CH_SRC_NPAR + k];
    if (d->src_pwl_ofs) for (int k = d->src_pwl_ofs[i]; k < d->src_pwl_ofs[i + 1]; ++k) { s.ts.push_back(d->pwl_t[k]); s.ys.push_back(d->pwl_y[k]); }
    s.ac = d->src_ac ? std::fabs(d->src_ac[i]) : 0.0;
    c->src.push_back(s);
  }
  for (int i = 0; i < d->n_model; ++i) c->model.emplace_back(d->model_par + (size_t)i * CH_B4_NPAR, d->model_par + (size_t)(i + 1) * CH_B4_NPAR);
  auto bad = [&](const char* m) -> ch_circuit* { ctx->err = m; return nullptr; };
  for (int i = 0; i < d->n_dev; ++i) {
    HDev v; v.kind = d->dev_kind[i]; v.branch = -1; v.eliminated = false;
    for (int k = 0; k < CH_DEV_NNODE; ++k) { v.node[k] = d->dev_node[i * CH_DEV_NNODE + k]; if (v.node[k] < 0 || v.node[k] > d->n_nodes) return bad("device node id out of range"); }
    for (int k = 0; k < CH_DEV_NIPAR; ++k) v.ipar[k] = d->dev_ipar[i * CH_DEV_NIPAR + k];
    for (int k = 0; k < CH_DEV_NPAR; ++k) v.par[k] = d->dev_par[i * CH_DEV_NPAR + k];
    v.mult = d->dev_mult[i];
    if (v.mult < 0) return bad("Cannot construct a ParallelInstances with non-positive multiplier");
    if ((v.kind == CH_DEV_V || v.kind == CH_DEV_I) && (v.ipar[0] < 0 || v.ipar[0] >= d->n_src)) return bad("source index out of range");
    if (v.kind == CH_DEV_MOS && (v.ipar[0] < 0 || v.ipar[0] >= d->n_model)) return bad("model index out of range");
    if (v.kind < CH_DEV_R || v.kind > CH_DEV_VA) return bad("unknown device kind");
    if (v.kind == CH_DEV_VA) {
      if (v.ipar[0] < 0 || v.ipar[0] >= va_gen::N_MODULES) return bad("Verilog-A module id out of range (is the module compiled into this library?)");
      const va_gen::ModuleInfo& mi = va_gen::MODULES[v.ipar[0]];
      if (v.ipar[1] < 0 || (int64_t)v.ipar[1] + 2 * mi.n_params > d->n_va_par || !d->va_par) return bad("Verilog-A parameter block out of range");
      v.va_nt = mi.n_nodes; v.va_qmask = mi.q_mask;
    }
    c->dev.push_back(v);
  }
  if (d->va_par && d->n_va_par > 0) c->va_par.assign(d->va_par, d->va_par + d->n_va_par);
  for (int i = 0; i < d->n_slot; ++i) { c->slot_kind.push_back(d->slot_kind[i]); c->slot_a.push_back(d->slot_a[i]); c->slot_b.push_back(d->slot_b[i]); }
  for (int i = 0; i < d->n_obs; ++i) { c->obs_kind.push_back(d->obs_kind[i]); c->obs_index.push_back(d->obs_index[i]); }
  c->slot_val.assign(c->slot_kind.size(), {});
  std::vector<char> protect(c->dev.size(), 0), swept(c->src.size(), 0);
  for (size_t o = 0; o < c->obs_kind.size(); ++o) if (c->obs_kind[o] == 1) { if (c->obs_index[o] < 0 || c->obs_index[o] >= (int)c->dev.size()) return bad("observable device index out of range"); protect[c->obs_index[o]] = 1; }
  for (size_t i = 0; i < c->slot_kind.size(); ++i) {
    const int k = c->slot_kind[i], sa = c->slot_a[i], sb = c->slot_b[i];
    bool ok = true;
    switch (k) {
      case CH_SLOT_DEV_PAR: ok = sa >= 0 && sa < (int)c->dev.size() && sb >= 0 && sb < CH_DEV_NPAR; break;
      case CH_SLOT_DEV_MULT: ok = sa >= 0 && sa < (int)c->dev.size(); break;
      case CH_SLOT_MODEL_PAR: ok = sa >= 0 && sa < (int)c->model.size() && sb >= 0 && sb < CH_B4_NPAR; break;
      case CH_SLOT_SRC_DC: ok = sa >= 0 && sa < (int)c->src.size(); break;
      case CH_SLOT_SRC_PAR: ok = sa >= 0 && sa < (int)c->src.size() && sb >= 0 && sb < CH_SRC_NPAR; break;
      case CH_SLOT_TEMP: case CH_SLOT_GMIN: break;
      case CH_SLOT_VA_PAR: ok = sa >= 0 && (size_t)sa < c->va_par.size(); break;
      default: ok = false;
    }
    if (!ok) return bad("parameter slot refers to a device, model, source or field that does not exist");
    if (k == CH_SLOT_SRC_DC || k == CH_SLOT_SRC_PAR) swept[sa] = 1;
  }
  // an AC-driven voltage source keeps its node and branch unknowns: the small-signal excitation enters one linear row
  for (size_t i = 0; i < c->dev.size(); ++i) if (c->dev[i].kind == CH_DEV_V && c->src[c->dev[i].ipar[0]].ac != 0.0) { protect[i] = 1; swept[c->dev[i].ipar[0]] = 1; }
  int rc = analyse(c->n_nodes, c->dev, c->src, protect, swept, c->A, tear);
  if (rc != CH_OK) { ctx->err = c->A.err; return nullptr; }
  c->is_torn = tear;
  rc = c->upload_structure();
  if (rc != CH_OK) return nullptr;
  if (!tear && c->A.max_nc > 64 && std::getenv("CEDARHIP_NO_TEAR") == nullptr) {
    // one large coupled block: try the bordered block-diagonal form (refused, with a reason, for most circuits)
    c->torn_c.reset(ch_circuit_build_impl(ctx, d, true));
    c->torn_note = c->torn_c ? "torn companion built" : ctx->err;
    ctx->err.clear();
  }
  return owner.release();
}
void ch_circuit_free(ch_circuit* c) { delete c; }

int ch_circuit_info(ch_circuit* c, ch_info* o) {
  if (!c || !o) return CH_ERR_INVALID;
  std::memset(o, 0, sizeof(*o));
  const Analysis& A = c->A;
  o->n_nodes = c->n_nodes; o->n_branches = A.n_branch; o->n_mna = A.n_mna; o->n_unknowns = A.n_unk; o->n_known = (int)A.known.size() - 1;
  o->n_alias = A.n_alias; o->n_components = A.n_comp; o->max_component = A.max_nc; o->n_classes = (int)A.classes.size();
  o->n_mos = (int)A.mos_hdev.size(); o->n_mos_classes = c->n_cls; o->path = c->path; o->n_samples = c->S;
  o->nnz_jac = (int64_t)c->h_colidx.size(); o->nnz_lu = c->plan[1].valid ? c->plan[1].nnz_lu : (c->plan[0].valid ? c->plan[0].nnz_lu : 0);
  return CH_OK;
}
// maps for tests / host mirrors: MNA index -> unknown (or -1) for nodes 0..n_nodes and branches
int ch_circuit_maps(ch_circuit* c, int32_t* node_unknown, int32_t* node_known, int32_t* branch_unknown) {
  if (!c) return CH_ERR_INVALID;
  for (int n = 0; n <= c->n_nodes; ++n) { if (node_unknown) node_unknown[n] = c->A.node_unknown[n]; if (node_known) node_known[n] = c->A.node_known[n]; }
  for (int b = 0; b < c->A.n_branch; ++b) if (branch_unknown) branch_unknown[b] = c->A.branch_unknown[b];
  return CH_OK;
}

static int ch_set_samples_impl(ch_circuit* c, int32_t n) {
  if (!c) return CH_ERR_INVALID;
  if (n < 1) { c->set_err("ch_set_samples: at least one sample"); return CH_ERR_INVALID; }
  c->S = n;
  for (auto& v : c->slot_val) v.clear();
  c->dirty = true;
  if (c->torn_c) return ch_set_samples_impl(c->torn_c.get(), n);
  return CH_OK;
}
static int ch_set_params_impl(ch_circuit* c, int32_t lo, int32_t hi, int32_t n_slots, const int32_t* ids, const double* values) {
  if (!c) return CH_ERR_INVALID;
  if (lo < 0 || hi > c->S || lo >= hi || n_slots < 0 || (n_slots > 0 && (!ids || !values))) { c->set_err("ch_set_params: sample range outside [0, n_samples) or missing arrays"); return CH_ERR_INVALID; }
  for (int i = 0; i < n_slots; ++i) {
    const int id = ids[i];
    if (id < 0 || id >= (int)c->slot_kind.size()) { c->set_err("slot id out of range"); return CH_ERR_INVALID; }
    auto& v = c->slot_val[id];
    if (v.empty()) {
      // initialise with the description's base value
      double base = 0; const int a = c->slot_a[id], b = c->slot_b[id];
      switch (c->slot_kind[id]) {
        case CH_SLOT_DEV_PAR: base = c->dev[a].par[b]; break;
        case CH_SLOT_DEV_MULT: base = c->dev[a].mult; break;
        case CH_SLOT_MODEL_PAR: base = c->model[a][b]; break;
        case CH_SLOT_SRC_DC: base = c->src[a].dc; break;
        case CH_SLOT_SRC_PAR: base = c->src[a].par[b]; break;
        case CH_SLOT_TEMP: base = c->temp; break;
        case CH_SLOT_GMIN: base = c->gmin; break;
        case CH_SLOT_VA_PAR: base = c->va_par[a]; break;
      }
      v.assign(c->S, base);
    }
    for (int s = lo; s < hi; ++s) v[s] = values[(size_t)i * (hi - lo) + (s - lo)];
  }
  c->dirty = true;
  if (c->torn_c) return ch_set_params_impl(c->torn_c.get(), lo, hi, n_slots, ids, values);
  return CH_OK;
}

static int ch_dc_impl(ch_circuit* c, const ch_dc_opts* o, double* x_out, int32_t* status_out, ch_stats* stats) {
  if (!c || !o) return CH_ERR_INVALID;
  ArenaScope arena_scope(&c->arena);
  c->ctx->err.clear();
  (void)hipSetDevice(c->ctx->device);
  auto t0 = hclock::now();
  ch_stats st; std::memset(&st, 0, sizeof(st));
  c->device_ms = 0; c->n_launch = 0; c->n_timed = 0;
  int rc = c->finalize_params();
  if (rc != CH_OK) return rc;
  std::vector<int> status;
  rc = c->dc_solve(*o, 0, &status, &st);
  if (x_out) { int r2 = c->download_mna(0, 0.0, o->tran_mode ? 2 : 0, x_out); if (r2 != CH_OK) return r2; }
  if (status_out) for (int s = 0; s < c->S; ++s) status_out[s] = status.empty() ? rc : status[s];
  st.wall_seconds = st.dc_seconds = std::chrono::duration<double>(hclock::now() - t0).count();
  st.device_seconds = c->n_timed > 0 ? c->device_ms * 1e-3 * (double)c->n_launch / (double)c->n_timed : 0.0; st.n_kernel_launches = c->n_launch;
  if (stats) *stats = st;
  return rc;
}

static int ch_tran_impl(ch_circuit* c, double t0, double t1, const ch_tran_opts* o, ch_result** out) {
  if (!c || !o || !out) return CH_ERR_INVALID;
  ArenaScope arena_scope(&c->arena);
  c->ctx->err.clear();
  (void)hipSetDevice(c->ctx->device);
  std::unique_ptr<ch_result> R(new ch_result());
  int rc = c->tran_solve(t0, t1, *o, *R);
  R->status = rc;
  *out = R.release();
  return rc;
}
int64_t ch_result_n_times(const ch_result* r) { return r ? (int64_t)r->times.size() : 0; }
const double* ch_result_times(const ch_result* r) { return r ? r->times.data() : nullptr; }
const int32_t* ch_result_dense_points(const ch_result* r) { return (r && r->pts.size() == r->times.size()) ? r->pts.data() : nullptr; }
int ch_result_device_values(const ch_result* r, const double** ptr, int64_t* n_doubles) {
  if (!r || !ptr || !n_doubles) return CH_ERR_INVALID;
  *ptr = r->dev_values; *n_doubles = r->dev_n;
  return r->dev_values ? CH_OK : CH_ERR_UNSUPPORTED;
}
const double* ch_result_values(const ch_result* r) { return r ? r->values.data() : nullptr; }
const double* ch_result_final_state(const ch_result* r) { return r ? r->final_state.data() : nullptr; }
int ch_result_stats(const ch_result* r, ch_stats* s) { if (!r || !s) return CH_ERR_INVALID; *s = r->stats; return CH_OK; }
int ch_result_status(const ch_result* r) { return r ? r->status : CH_ERR_INVALID; }
void ch_result_free(ch_result* r) { delete r; }

static int ch_eval_impl(ch_circuit* c, int32_t sample, const double* x_mna, double t, double alpha0, int32_t mode, double* F_out, double* Q_out, double* J_out) {
  if (!c || !x_mna || sample < 0 || sample >= c->S) return CH_ERR_INVALID;
  c->ctx->err.clear();
  (void)hipSetDevice(c->ctx->device);
  ArenaScope arena_scope(&c->arena);
  int rc = c->finalize_params();
  if (rc != CH_OK) return rc;
  const Analysis& A = c->A;
  const int S = c->S, nblk = A.n_comp * S, ds = A.max_nc;
  // state: only the requested sample matters
  std::vector<double> xs((size_t)S * A.n_unk, 0.0);
  for (int u = 0; u < A.n_unk; ++u) xs[(size_t)sample * A.n_unk + u] = x_mna[A.unk_mna[u]];
  if (hipMemcpy(c->d_X.p, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return CH_ERR_DEVICE;
  std::vector<unsigned char> act(nblk, 0);
  for (int k = 0; k < A.n_comp; ++k) act[(size_t)k * S + sample] = 1;
  if (hipMemcpy(c->d_active.p, act.data(), nblk, hipMemcpyHostToDevice) != hipSuccess) return CH_ERR_DEVICE;
  if (c->d_dumpA.alloc((size_t)nblk * ds * ds) != hipSuccess || c->d_dumpF.alloc((size_t)nblk * ds) != hipSuccess || c->d_dumpQ.alloc((size_t)nblk * ds) != hipSuccess) return CH_ERR_DEVICE;
  NewtonArgs a = c->base;
  rc = c->set_sources(a, t, mode == 0 ? 0 : 1);
  if (rc != CH_OK) return rc;
  a.mode = MODE_EVAL; a.maxit = 1; a.alpha[0] = alpha0; a.hist_slot[0] = 0; a.cand_slot = 1; a.active = c->d_active.p; a.abstol = 1e-6; a.reltol = 1e-3;
  a.dumpA = c->d_dumpA.p; a.dumpF = c->d_dumpF.p; a.dumpQ = c->d_dumpQ.p; a.dumpC = nullptr; a.dump_stride = ds;
  Summary sm;
  rc = c->run_newton(a, act.data(), sm);
  if (rc != CH_OK) return rc;
  const int n = A.n_mna;
  std::vector<char> has(n, 0);
  if (J_out) std::fill(J_out, J_out + (size_t)n * n, 0.0);
  if (F_out) std::fill(F_out, F_out + n, 0.0);
  if (Q_out) std::fill(Q_out, Q_out + n, 0.0);
  if (c->path == 2) {
    const size_t nnz = c->h_colidx.size();
    std::vector<double> av(nnz), fv(A.n_unk), qv(A.n_unk);
    (void)hipMemcpy(av.data(), c->sp_Aval.p + (size_t)sample * nnz, nnz * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipMemcpy(fv.data(), c->sp_F.p + (size_t)sample * A.n_unk, fv.size() * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipMemcpy(qv.data(), c->sp_Q.p + (size_t)sample * A.n_unk, qv.size() * sizeof(double), hipMemcpyDeviceToHost);
    for (int u = 0; u < A.n_unk; ++u) {
      const int ri = A.unk_mna[u]; has[ri] = 1;
      if (F_out) F_out[ri] = fv[u];
      if (Q_out) Q_out[ri] = qv[u];
      if (J_out) for (int p = c->h_rowptr[u]; p < c->h_rowptr[u + 1]; ++p) J_out[(size_t)ri * n + A.unk_mna[c->h_colidx[p]]] = av[p];
    }
  } else {
  std::vector<double> hA((size_t)nblk * ds * ds), hF((size_t)nblk * ds), hQ((size_t)nblk * ds);
  (void)hipMemcpy(hA.data(), c->d_dumpA.p, hA.size() * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipMemcpy(hF.data(), c->d_dumpF.p, hF.size() * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipMemcpy(hQ.data(), c->d_dumpQ.p, hQ.size() * sizeof(double), hipMemcpyDeviceToHost);
  for (int k = 0; k < A.n_comp; ++k) {
    const int blk = k * S + sample, nc = A.comp_nc[k], uo = A.comp_uofs[k];
    for (int i = 0; i < nc; ++i) {
      const int ri = A.unk_mna[uo + i];
      has[ri] = 1;
      if (F_out) F_out[ri] = hF[(size_t)blk * ds + i];
      if (Q_out) Q_out[ri] = hQ[(size_t)blk * ds + i];
      if (J_out) for (int j = 0; j < nc; ++j) J_out[(size_t)ri * n + A.unk_mna[uo + j]] = hA[(size_t)blk * ds * ds + (size_t)i * nc + j];
    }
  }
  }
  if (J_out) for (int i = 0; i < n; ++i) if (!has[i]) J_out[(size_t)i * n + i] = 1.0;
  return CH_OK;
}

// ---- small-signal analyses -------------------------------------------------------------------
// Shared front half of ch_ac / ch_noise: DC operating point (slot 0), then G, C (and the AC right-hand
// side b = -(F(src + ac) - F(src)), exact because every source enters F linearly) as per-block dense dumps.
static int ac_linearise(ch_circuit* c, const ch_dc_opts* o, ch_stats* st, bool want_b) {
  int rc = c->finalize_params();
  if (rc != CH_OK) return rc;
  if (c->path == 2 && c->A.n_unk > 4096) { c->set_err("AC / noise analysis: the coupled system has more than 4096 unknowns (dense complex LU)"); return CH_ERR_UNSUPPORTED; }
  g_arena = &c->arena;
  rc = c->dc_solve(*o, 0, nullptr, st);
  if (rc != CH_OK) return rc;
  const Analysis& A = c->A;
  const int S = c->S, nblk = c->ac_ncomp() * S, ds = c->ac_ds();
  const size_t nA = (size_t)nblk * ds * ds, nF = (size_t)nblk * ds;
  if (c->d_dumpG.alloc(nA) != hipSuccess || c->d_dumpC.alloc(nA) != hipSuccess || c->d_dumpF0.alloc(nF) != hipSuccess ||
      c->d_dumpF.alloc(nF) != hipSuccess || c->d_dumpQ.alloc(nF) != hipSuccess || c->d_dumpA.alloc(nA) != hipSuccess) return CH_ERR_DEVICE;
  const int mode = o->tran_mode ? 2 : 0;
  for (int pass = 0; pass < (want_b ? 2 : 1); ++pass) {
    NewtonArgs a = c->base;
    c->ac_scale = pass == 0 ? 0.0 : 1.0;
    rc = c->set_sources(a, 0.0, mode);
    c->ac_scale = 0.0;
    if (rc != CH_OK) return rc;
    a.mode = MODE_EVAL; a.maxit = 1; a.alpha[0] = 0.0; a.hist_slot[0] = 0; a.cand_slot = 1; a.active = nullptr; a.abstol = 1e-6; a.reltol = 1e-3;
    a.dumpA = pass == 0 ? c->d_dumpG.p : c->d_dumpA.p; a.dumpC = pass == 0 ? c->d_dumpC.p : nullptr;
    a.dumpF = pass == 0 ? c->d_dumpF0.p : c->d_dumpF.p; a.dumpQ = c->d_dumpQ.p; a.dump_stride = ds;
    Summary sm;
    rc = c->run_newton(a, nullptr, sm);
    if (rc != CH_OK) return rc;
    if (c->path == 2) {   // the sparse evaluation left G (alpha0 = 0), C and F in CSR / vector form: expand to the dense blocks
      const int n = A.n_unk, nnz = (int)c->h_colidx.size();
      hipLaunchKernelGGL(csr_to_dense_kernel, dim3((n + 63) / 64, S), dim3(64), 0, c->ctx->stream, (const int*)c->sp_rowptr.p, (const int*)c->sp_colidx.p,
                         (const double*)c->sp_Aval.p, (const double*)c->sp_Cval.p, (const double*)c->sp_F.p, n, nnz, S, c->d_dumpG.p, c->d_dumpC.p,
                         pass == 0 ? c->d_dumpF0.p : c->d_dumpF.p, pass == 0 ? 1 : 0);
    }
  }
  if (want_b) {  // b = F0 - F1 (device side, in place in d_dumpF)
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((nF + 255) / 256)), dim3(256), 0, c->ctx->stream, c->d_dumpF.p, (const double*)c->d_dumpF0.p, (long)nF);
  }
  return CH_OK;
}

// (G + jwC) solves of one analysis: `ny` systems per frequency.  Up to 96 unknowns the complex LU of a system runs in one
// wavefront with its matrices in LDS; larger coupled systems (sparse path) take the 256-thread variant with a global
// workspace, a chunk of frequencies per launch so that the workspace stays below 1 GiB.
static int launch_ac(ch_circuit* c, AcArgs& a, int n_freq, int ny, int ds) {
  const size_t per = (size_t)2 * ds * (ds + 1);
  if (ds <= 96) {
    const size_t lds = per * sizeof(double);
    a.work = nullptr; a.f0 = 0;
    hipLaunchKernelGGL(ac_block_kernel<64>, dim3(n_freq, ny), dim3(64), lds, c->ctx->stream, a);
    return hipGetLastError() == hipSuccess ? CH_OK : CH_ERR_DEVICE;
  }
  const size_t cap = (size_t)1 << 27;   // doubles
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_freq, cap / (per * (size_t)ny)));
  double* work = nullptr;
  if (hipMalloc((void**)&work, (size_t)chunk * ny * per * sizeof(double)) != hipSuccess) { c->set_err("AC / noise analysis: out of device memory for the dense complex LU workspace"); return CH_ERR_DEVICE; }
  int rc = CH_OK;
  for (int f0 = 0; f0 < n_freq && rc == CH_OK; f0 += chunk) {
    a.work = work; a.f0 = f0;
    hipLaunchKernelGGL(ac_block_kernel<256>, dim3(std::min(chunk, n_freq - f0), ny), dim3(256), 0, c->ctx->stream, a);
    if (hipGetLastError() != hipSuccess) rc = CH_ERR_DEVICE;
  }
  if (hipStreamSynchronize(c->ctx->stream) != hipSuccess) rc = CH_ERR_DEVICE;
  (void)hipFree(work);
  return rc;
}

static int upload_omega(ch_circuit* c, int n_freq, const double* freqs_hz) {
  std::vector<double> w(n_freq);
  for (int i = 0; i < n_freq; ++i) { if (!(freqs_hz[i] >= 0.0) || !std::isfinite(freqs_hz[i])) { c->set_err("frequencies must be finite and non-negative"); return CH_ERR_INVALID; } w[i] = 6.283185307179586 * freqs_hz[i]; }
  if (c->d_omega.upload(w, c->ctx->stream) != hipSuccess) return CH_ERR_DEVICE;
  std::vector<int> z(1, 0);
  if (c->d_acfail.upload(z, c->ctx->stream) != hipSuccess) return CH_ERR_DEVICE;
  return CH_OK;
}

static int ch_ac_impl(ch_circuit* c, const ch_dc_opts* o, int32_t n_freq, const double* freqs_hz, double* x_ac_out, ch_stats* stats) {
  if (!c || !o || n_freq < 1 || !freqs_hz || !x_ac_out) return CH_ERR_INVALID;
  ArenaScope arena_scope(&c->arena);
  c->ctx->err.clear();
  (void)hipSetDevice(c->ctx->device);
  auto t0 = hclock::now();
  ch_stats st; std::memset(&st, 0, sizeof(st));
  c->device_ms = 0; c->n_launch = 0; c->n_timed = 0;
  int rc = ac_linearise(c, o, &st, true);
  st.dc_seconds = std::chrono::duration<double>(hclock::now() - t0).count();
  if (rc != CH_OK) { if (stats) *stats = st; return rc; }
  const Analysis& A = c->A;
  const int S = c->S, nblk = c->ac_ncomp() * S, ds = c->ac_ds();
  g_arena = &c->arena;
  rc = upload_omega(c, n_freq, freqs_hz); if (rc != CH_OK) return rc;
  const size_t nx = (size_t)S * n_freq * A.n_unk * 2;
  if (c->d_xac.alloc(nx) != hipSuccess) return CH_ERR_DEVICE;
  AcArgs a; std::memset(&a, 0, sizeof(a));
  a.bmeta = c->ac_bmeta(); a.G = c->d_dumpG.p; a.C = c->d_dumpC.p; a.b = c->d_dumpF.p; a.ds = ds; a.S = S; a.n_unk = A.n_unk; a.n_freq = n_freq; a.n_comp = c->ac_ncomp();
  if (!a.bmeta) return CH_ERR_DEVICE;
  a.omega = c->d_omega.p; a.x_out = c->d_xac.p; a.noise = 0; a.fail = c->d_acfail.p;
  if (std::getenv("CEDARHIP_DEBUG_AC") && ds <= 8 && nblk == 1) {   // diagnostic: the linearisation the complex solves start from
    std::vector<double> hg((size_t)ds * ds), hc((size_t)ds * ds), hb(ds);
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)hipMemcpy(hg.data(), c->d_dumpG.p, hg.size() * sizeof(double), hipMemcpyDeviceToHost); (void)hipMemcpy(hc.data(), c->d_dumpC.p, hc.size() * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb.data(), c->d_dumpF.p, hb.size() * sizeof(double), hipMemcpyDeviceToHost);
    { std::vector<double> hx(A.n_unk); (void)hipMemcpy(hx.data(), c->d_X.p, hx.size() * sizeof(double), hipMemcpyDeviceToHost);
      std::fprintf(stderr, "[ac] state:"); for (double v : hx) std::fprintf(stderr, " %.12e", v); std::fprintf(stderr, "\n"); }
    for (int i = 0; i < ds; ++i) { std::fprintf(stderr, "[ac] G row %d:", i); for (int j = 0; j < ds; ++j) std::fprintf(stderr, " %.9e", hg[(size_t)i * ds + j]); std::fprintf(stderr, " | C:"); for (int j = 0; j < ds; ++j) std::fprintf(stderr, " %.9e", hc[(size_t)i * ds + j]); std::fprintf(stderr, " | b %.9e\n", hb[i]); }
  }
  rc = launch_ac(c, a, n_freq, nblk, ds);
  if (rc != CH_OK) return rc;
  std::vector<double> xs(nx);
  int fail = 0;
  if (hipMemcpyAsync(xs.data(), c->d_xac.p, nx * sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream) != hipSuccess ||
      hipMemcpyAsync(&fail, c->d_acfail.p, sizeof(int), hipMemcpyDeviceToHost, c->ctx->stream) != hipSuccess ||
      hipStreamSynchronize(c->ctx->stream) != hipSuccess) { c->set_err("AC solve failed on the device"); return CH_ERR_DEVICE; }
  st.n_kernel_launches = c->n_launch + 2; st.nfactors += (int64_t)n_freq * S; st.nsolve += (int64_t)n_freq * S;
  // unknown space -> MNA order (known nodes carry no small signal: AC-driven sources are never eliminated)
  const int n_nodes = c->n_nodes, nm = A.n_mna;
  for (int s = 0; s < S; ++s) for (int f = 0; f < n_freq; ++f) {
    const double* xu = &xs[(((size_t)s * n_freq + f) * A.n_unk) * 2];
    double* xo = x_ac_out + (((size_t)s * n_freq + f) * nm) * 2;
    for (int n = 1; n <= n_nodes; ++n) { const int u = A.node_unknown[n]; xo[2 * (n - 1)] = u >= 0 ? xu[2 * u] : 0.0; xo[2 * (n - 1) + 1] = u >= 0 ? xu[2 * u + 1] : 0.0; }
    for (int b = 0; b < A.n_branch; ++b) { const int u = A.branch_unknown[b]; xo[2 * (n_nodes + b)] = u >= 0 ? xu[2 * u] : CH_NAN; xo[2 * (n_nodes + b) + 1] = u >= 0 ? xu[2 * u + 1] : CH_NAN; }
  }
  st.wall_seconds = std::chrono::duration<double>(hclock::now() - t0).count();
  if (stats) *stats = st;
  if (fail) { c->set_err("AC analysis: singular small-signal matrix G + jwC"); return CH_ERR_SINGULAR; }
  return CH_OK;
}

static int ch_noise_impl(ch_circuit* c, const ch_dc_opts* o, int32_t out_kind, int32_t out_index, int32_t n_freq, const double* freqs_hz, double* psd_out, ch_stats* stats) {
  if (!c || !o || n_freq < 1 || !freqs_hz || !psd_out) return CH_ERR_INVALID;
  ArenaScope arena_scope(&c->arena);
  c->ctx->err.clear();
  (void)hipSetDevice(c->ctx->device);
  auto t0 = hclock::now();
  ch_stats st; std::memset(&st, 0, sizeof(st));
  c->device_ms = 0; c->n_launch = 0; c->n_timed = 0;
  const Analysis& A = c->A;
  int u_out = -1;
  if (out_kind == 0) { if (out_index < 0 || out_index > c->n_nodes) { c->set_err("noise: output node out of range"); return CH_ERR_INVALID; } u_out = out_index == 0 ? -1 : A.node_unknown[out_index]; }
  else if (out_kind == 1) {
    if (out_index < 0 || out_index >= (int)c->dev.size() || c->dev[out_index].branch < 0) { c->set_err("noise: output device has no branch current"); return CH_ERR_INVALID; }
    u_out = A.branch_unknown[c->dev[out_index].branch];
    if (u_out < 0) { c->set_err("noise: the output branch current was eliminated (observe it when building the circuit)"); return CH_ERR_INVALID; }
  } else return CH_ERR_INVALID;
  int rc = ac_linearise(c, o, &st, false);
  st.dc_seconds = std::chrono::duration<double>(hclock::now() - t0).count();
  if (rc != CH_OK) { if (stats) *stats = st; return rc; }
  const int S = c->S, ds = c->ac_ds();
  if (u_out < 0) {  // a node held by ideal sources carries no noise
    std::fill(psd_out, psd_out + (size_t)S * n_freq, 0.0);
    if (stats) *stats = st;
    return CH_OK;
  }
  const int comp = c->ac_comp_of(u_out);
  const int uofs = c->ac_uofs(comp), ncb = c->ac_nc(comp);
  g_arena = &c->arena;
  rc = upload_omega(c, n_freq, freqs_hz); if (rc != CH_OK) return rc;
  // noise table of the output block at the operating point (device side)
  const int ndev_b = c->ac_ndev(comp), n_tab = ndev_b * va::MAX_NOISE;
  if (c->d_noise_a.alloc((size_t)S * n_tab) != hipSuccess || c->d_noise_b.alloc((size_t)S * n_tab) != hipSuccess ||
      c->d_noise_pwr.alloc((size_t)S * n_tab) != hipSuccess || c->d_noise_exp.alloc((size_t)S * n_tab) != hipSuccess ||
      c->d_psd.alloc((size_t)S * n_freq) != hipSuccess) return CH_ERR_DEVICE;
  {
    NewtonArgs na0 = c->base;
    rc = c->set_sources(na0, 0.0, o->tran_mode ? 2 : 0); if (rc != CH_OK) return rc;
    if (na0.inline_vals) {  // the table kernel reads the known-node values from the device buffer
      std::memcpy(c->h_stage, na0.vals_inline, (size_t)(na0.nk + na0.nsrc) * sizeof(double));
      if (hipMemcpyAsync(c->d_kv.p, c->h_stage, (size_t)(na0.nk + na0.nsrc) * sizeof(double), hipMemcpyHostToDevice, c->ctx->stream) != hipSuccess) return CH_ERR_DEVICE;
    }
    NoiseTabArgs t; std::memset(&t, 0, sizeof(t));
    t.dkind = c->d_dkind.p; t.dterm = c->d_dterm.p; t.dsrc = c->d_dsrc.p; t.dcls_local = c->d_dcls_local.p; t.dhdev = c->d_dhdev.p;
    t.dpar = c->d_dpar.p; t.dmult = c->d_dmult.p; t.vapar = c->d_vapar.p; t.va_stride = c->base.va_stride; t.temp_s = c->d_temp.p; t.gmin_s = c->d_gmin.p;
    t.X = c->d_X.p; t.kv = c->d_kv.p;  // slot 0 holds the operating point
    t.Spar = c->Spar; t.Stemp = c->Stemp; t.Sgmin = c->Sgmin; t.Ssrc = c->Ssrc; t.nk = (int)A.known.size(); t.S = S; t.n_unk = A.n_unk;
    t.dofs = c->ac_dofs(comp); t.ndev = ndev_b; t.uofs = uofs; t.nc = ncb;
    t.na = c->d_noise_a.p; t.nb = c->d_noise_b.p; t.pwr = c->d_noise_pwr.p; t.ex = c->d_noise_exp.p;
    hipLaunchKernelGGL(noise_table_kernel, dim3((ndev_b * S + 63) / 64), dim3(64), 0, c->ctx->stream, t);
  }
  AcArgs a; std::memset(&a, 0, sizeof(a));
  a.bmeta = c->ac_bmeta(); a.G = c->d_dumpG.p; a.C = c->d_dumpC.p; a.b = nullptr; a.ds = ds; a.S = S; a.n_unk = A.n_unk; a.n_freq = n_freq; a.n_comp = c->ac_ncomp();
  if (!a.bmeta) return CH_ERR_DEVICE;
  a.omega = c->d_omega.p; a.noise = 1; a.comp_out = comp; a.row_out = u_out - uofs; a.n_noise = n_tab;
  a.noise_a = c->d_noise_a.p; a.noise_b = c->d_noise_b.p; a.noise_pwr = c->d_noise_pwr.p; a.noise_exp = c->d_noise_exp.p;
  a.psd_out = c->d_psd.p; a.fail = c->d_acfail.p;
  { const int rc_l = launch_ac(c, a, n_freq, S, ds); if (rc_l != CH_OK) return rc_l; }
  int fail = 0;
  if (hipMemcpyAsync(psd_out, c->d_psd.p, (size_t)S * n_freq * sizeof(double), hipMemcpyDeviceToHost, c->ctx->stream) != hipSuccess ||
      hipMemcpyAsync(&fail, c->d_acfail.p, sizeof(int), hipMemcpyDeviceToHost, c->ctx->stream) != hipSuccess ||
      hipStreamSynchronize(c->ctx->stream) != hipSuccess) { c->set_err("noise solve failed on the device"); return CH_ERR_DEVICE; }
  st.n_kernel_launches = c->n_launch + 1; st.nfactors += (int64_t)n_freq * S; st.nsolve += (int64_t)n_freq * S;
  st.wall_seconds = std::chrono::duration<double>(hclock::now() - t0).count();
  if (stats) *stats = st;
  if (fail) { c->set_err("noise analysis: singular small-signal matrix G + jwC"); return CH_ERR_SINGULAR; }
  return CH_OK;
}

static int mos_eval_impl(ch_circuit* c, int32_t sample, const double* v, double* out, bool quad) {
  if (!c || !v || !out || sample < 0 || sample >= c->S) return CH_ERR_INVALID;
  c->ctx->err.clear();
  (void)hipSetDevice(c->ctx->device);
  ArenaScope arena_scope(&c->arena);
  int rc = c->finalize_params();
  if (rc != CH_OK) return rc;
  const int nm = (int)c->A.mos_hdev.size();
  if (nm == 0) return CH_OK;
  double *dv = nullptr, *dout = nullptr;
  if (hipMalloc((void**)&dv, (size_t)nm * 4 * sizeof(double)) != hipSuccess || hipMalloc((void**)&dout, (size_t)nm * 40 * sizeof(double)) != hipSuccess) return CH_ERR_DEVICE;
  (void)hipMemcpy(dv, v, (size_t)nm * 4 * sizeof(double), hipMemcpyHostToDevice);
  std::vector<double> hg(c->Sgmin);
  (void)hipMemcpy(hg.data(), c->d_gmin.p, hg.size() * sizeof(double), hipMemcpyDeviceToHost);
  const double gm = hg[c->Sgmin > 1 ? sample : 0];
  if (quad) hipLaunchKernelGGL(mos_eval_quad_kernel, dim3((nm * 4 + 63) / 64), dim3(64), 0, c->ctx->stream, (const double*)c->d_mosp.p, c->base.mos_cols, (const int*)c->d_moscls_inst.p, c->Smos, (int)sample, nm, (const double*)dv, gm, dout);
  else hipLaunchKernelGGL(mos_eval_kernel, dim3((nm + 63) / 64), dim3(64), 0, c->ctx->stream, (const double*)c->d_mosp.p, c->base.mos_cols, (const int*)c->d_moscls_inst.p, c->Smos, (int)sample, nm, (const double*)dv, gm, dout);
  hipError_t e = hipStreamSynchronize(c->ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)nm * 40 * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dv); (void)hipFree(dout);
  if (e != hipSuccess) { c->set_err(hipGetErrorString(e)); return CH_ERR_DEVICE; }
  return CH_OK;
}
int ch_mos_eval(ch_circuit* c, int32_t sample, const double* v, double* out) { return guard_rc(c ? c->ctx : nullptr, [&] { return mos_eval_impl(c, sample, v, out, false); }); }
int ch_mos_eval_quad(ch_circuit* c, int32_t sample, const double* v, double* out) { return guard_rc(c ? c->ctx : nullptr, [&] { return mos_eval_impl(c, sample, v, out, true); }); }

static const char* const k_b4_names[] = {
#define P(n, d) #n,
#define B(n, d) #n, "l" #n, "w" #n, "p" #n,
#define I(n)
#include "../../include/cedarhip_bsim4_params.def"
};
static const char* const k_b4_ignored[] = {
#define P(n, d)
#define B(n, d)
#define I(n) #n,
#include "../../include/cedarhip_bsim4_params.def"
    nullptr};
int32_t ch_bsim4_npar(void) { return CH_B4_NPAR; }
const char* ch_bsim4_param_name(int32_t i) { return (i >= 0 && i < CH_B4_NPAR) ? k_b4_names[i] : nullptr; }
int32_t ch_bsim4_param_ignored(const char* name) {
  if (!name) return 0;
  for (int i = 0; k_b4_ignored[i]; ++i) if (std::strcmp(k_b4_ignored[i], name) == 0) return 1;
  return 0;
}
static int ch_bench_triad_impl(ch_ctx* ctx, int64_t n, int32_t iters, double* gbps_out) {
  if (!ctx || n < 1024 || iters < 1 || !gbps_out) return CH_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  double *a = nullptr, *b = nullptr, *c = nullptr;
  const size_t bytes = (size_t)n * sizeof(double);
  if (hipMalloc((void**)&a, bytes) != hipSuccess || hipMalloc((void**)&b, bytes) != hipSuccess || hipMalloc((void**)&c, bytes) != hipSuccess) {
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); ctx->err = "triad: out of device memory"; return CH_ERR_DEVICE;
  }
  (void)hipMemsetAsync(b, 0, bytes, ctx->stream); (void)hipMemsetAsync(c, 0, bytes, ctx->stream);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int threads = 256; const long n2 = n / 2;
  const int blocks = (int)std::min<long>((n2 + threads - 1) / threads, 256L * 32);
  double best = 0;
  for (int it = 0; it <= iters; ++it) {
    (void)hipEventRecord(e0, ctx->stream);
    hipLaunchKernelGGL(triad_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, (double2*)a, (const double2*)b, (const double2*)c, 3.0, n2);
    (void)hipEventRecord(e1, ctx->stream);
    if (hipEventSynchronize(e1) != hipSuccess) { ctx->err = "triad kernel failed"; break; }
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms > 0) best = std::max(best, 3.0 * (double)(n2 * 2) * sizeof(double) / (ms * 1e-3) / 1e9);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
  *gbps_out = best;
  return best > 0 ? CH_OK : CH_ERR_DEVICE;
}
// Test hook: fills the LDS of every CU with a pattern that is a NaN as a double and a large negative number as an int, so that a
// kernel which reads LDS it has not staged itself (LDS keeps whatever the previous kernel left there) gets garbage deterministically
// instead of the zeros of a fresh process.  The round-2 abort of test_gpu_stepper.py (gpurun_out/r02_stepper6.log) was exactly
// that: the helper wave of a pair read its OWN, unstaged slot table after a host-stepper kernel had used the CU.
__global__ void poison_lds_kernel(unsigned* sink) {
  extern __shared__ unsigned pl_[];
  const int n = 160 * 1024 / 4 - 64;
  for (int i = threadIdx.x; i < n; i += blockDim.x) pl_[i] = 0xfff7a5a5u;
  __syncthreads();
  if (threadIdx.x == 0 && pl_[blockIdx.x % n] != 0xfff7a5a5u) *sink = 1u;   // keeps the stores alive
}
static int ch_debug_poison_lds_impl(ch_ctx* ctx) {
  if (!ctx) return CH_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return CH_ERR_DEVICE;
  unsigned* sink = nullptr;
  if (hipMalloc((void**)&sink, sizeof(unsigned)) != hipSuccess) return CH_ERR_DEVICE;
  const int lds = 160 * 1024 - 256;
  (void)hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  // one workgroup fills a CU's LDS; several rounds of n_cu workgroups so that every CU is reached whatever the dispatcher does
  hipLaunchKernelGGL(poison_lds_kernel, dim3(prop.multiProcessorCount * 8), dim3(256), lds, ctx->stream, sink);
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(sink);
  if (e != hipSuccess) { ctx->err = std::string("poison_lds: ") + hipGetErrorString(e); return CH_ERR_DEVICE; }
  return CH_OK;
}
// test hook: the device's own exp / ln (va::v_exp, va::v_ln of va_rt.hpp and the BSIM4 code's flog) over a vector
__global__ void debug_math_kernel(int which, int n, const double* x, double* y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = which == 0 ? va::v_exp(x[i]) : which == 1 ? va::v_ln(x[i]) : flog(x[i]);
}
static int ch_debug_math_impl(ch_ctx* ctx, int32_t which, int32_t n, const double* x, double* y) {
  if (!ctx || which < 0 || which > 2 || n < 1 || !x || !y) return CH_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  double *dx = nullptr, *dy = nullptr;
  if (hipMalloc((void**)&dx, (size_t)n * sizeof(double)) != hipSuccess || hipMalloc((void**)&dy, (size_t)n * sizeof(double)) != hipSuccess) { (void)hipFree(dx); return CH_ERR_NOMEM; }
  hipError_t e = hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, which, n, (const double*)dx, dy); e = hipStreamSynchronize(ctx->stream); }
  if (e == hipSuccess) e = hipMemcpy(y, dy, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy);
  if (e != hipSuccess) { ctx->err = std::string("debug_math: ") + hipGetErrorString(e); return CH_ERR_DEVICE; }
  return CH_OK;
}
static int ch_bench_fp64_impl(ch_ctx* ctx, int32_t iters, double* tflops_out) {
  if (!ctx || iters < 1 || !tflops_out) return CH_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return CH_ERR_DEVICE;
  const int blocks = prop.multiProcessorCount * 8, threads = 256, n_outer = 2000;  // 8 waves per SIMD
  double* out = nullptr;
  if (hipMalloc((void**)&out, (size_t)blocks * threads * sizeof(double)) != hipSuccess) return CH_ERR_DEVICE;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 0;
  for (int it = 0; it <= iters; ++it) {
    (void)hipEventRecord(e0, ctx->stream);
    hipLaunchKernelGGL(fp64_peak_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, out, n_outer, 0.999999, 1e-6);
    (void)hipEventRecord(e1, ctx->stream);
    if (hipEventSynchronize(e1) != hipSuccess) { ctx->err = "fp64 peak kernel failed"; break; }
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 8 * (double)n_outer * blocks * threads;
    if (it > 0 && ms > 0) best = std::max(best, flop / (ms * 1e-3) / 1e12);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
  *tflops_out = best;
  return best > 0 ? CH_OK : CH_ERR_DEVICE;
}
int32_t ch_va_n_modules(void) { return va_gen::N_MODULES; }
int32_t ch_va_find(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < va_gen::N_MODULES; ++i) if (std::strcmp(va_gen::MODULES[i].name, name) == 0) return i;
  return -1;
}
const char* ch_va_module_name(int32_t id) { return (id >= 0 && id < va_gen::N_MODULES) ? va_gen::MODULES[id].name : nullptr; }
int32_t ch_va_module_info(int32_t id, int32_t* n_ports, int32_t* n_nodes, int32_t* n_params) {
  if (id < 0 || id >= va_gen::N_MODULES) return CH_ERR_INVALID;
  const va_gen::ModuleInfo& mi = va_gen::MODULES[id];
  if (n_ports) *n_ports = mi.n_ports; if (n_nodes) *n_nodes = mi.n_nodes; if (n_params) *n_params = mi.n_params;
  return CH_OK;
}
const char* ch_va_node_name(int32_t id, int32_t k) { return (id >= 0 && id < va_gen::N_MODULES && k >= 0 && k < va_gen::MODULES[id].n_nodes) ? va_gen::MODULES[id].node_names[k] : nullptr; }
const char* ch_va_param_name(int32_t id, int32_t k) { return (id >= 0 && id < va_gen::N_MODULES && k >= 0 && k < va_gen::MODULES[id].n_params) ? va_gen::MODULES[id].param_names[k] : nullptr; }
static int ch_va_eval_impl(ch_ctx* ctx, int32_t id, const double* par, const double* v, double temperature_k, double gmin, double* st_out) {
  if (!ctx || !par || !v || !st_out || id < 0 || id >= va_gen::N_MODULES) return CH_ERR_INVALID;
  (void)hipSetDevice(ctx->device);
  const va_gen::ModuleInfo& mi = va_gen::MODULES[id];
  const size_t np = (size_t)std::max(1, 2 * mi.n_params);
  double *dp = nullptr, *dv = nullptr, *ds = nullptr;
  if (hipMalloc((void**)&dp, np * sizeof(double)) != hipSuccess || hipMalloc((void**)&dv, NTERM * sizeof(double)) != hipSuccess || hipMalloc((void**)&ds, 144 * sizeof(double)) != hipSuccess) return CH_ERR_DEVICE;
  double vv[NTERM] = {0}; for (int k = 0; k < mi.n_nodes; ++k) vv[k] = v[k];
  (void)hipMemcpy(dp, par, (size_t)2 * mi.n_params * sizeof(double), hipMemcpyHostToDevice);
  (void)hipMemcpy(dv, vv, sizeof(vv), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(va_eval_kernel, dim3(1), dim3(64), 0, ctx->stream, (int)id, (const double*)dp, (const double*)dv, temperature_k, gmin, ds);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(st_out, ds, 144 * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dp); (void)hipFree(dv); (void)hipFree(ds);
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return CH_ERR_DEVICE; }
  return CH_OK;
}
int32_t ch_va_n_opvars(int32_t id) { return (id >= 0 && id < va_gen::N_MODULES) ? va_gen::N_OPVARS[id] : 0; }
const char* ch_va_opvar_name(int32_t id, int32_t k) { return (id >= 0 && id < va_gen::N_MODULES && k >= 0 && k < va_gen::N_OPVARS[id]) ? va_gen::OPNAMES[id][k] : nullptr; }
static int ch_va_opvars_impl(ch_ctx* ctx, int32_t id, const double* par, const double* v, double temperature_k, double gmin, double* op_out) {
  if (!ctx || !par || !v || !op_out || id < 0 || id >= va_gen::N_MODULES) return CH_ERR_INVALID;
  const int nop = va_gen::N_OPVARS[id];
  if (nop == 0) return CH_OK;
  (void)hipSetDevice(ctx->device);
  const va_gen::ModuleInfo& mi = va_gen::MODULES[id];
  double *dp = nullptr, *dv = nullptr, *dop = nullptr;
  if (hipMalloc((void**)&dp, (size_t)std::max(1, 2 * mi.n_params) * sizeof(double)) != hipSuccess || hipMalloc((void**)&dv, NTERM * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&dop, (size_t)nop * sizeof(double)) != hipSuccess) return CH_ERR_DEVICE;
  double vv[NTERM] = {0}; for (int k = 0; k < mi.n_nodes; ++k) vv[k] = v[k];
  (void)hipMemcpy(dp, par, (size_t)2 * mi.n_params * sizeof(double), hipMemcpyHostToDevice);
  (void)hipMemcpy(dv, vv, sizeof(vv), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(va_opvars_kernel, dim3(1), dim3(64), 0, ctx->stream, (int)id, (const double*)dp, (const double*)dv, temperature_k, gmin, dop);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(op_out, dop, (size_t)nop * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dp); (void)hipFree(dv); (void)hipFree(dop);
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return CH_ERR_DEVICE; }
  return CH_OK;
}
// ---- the exported entry points: every body that can allocate runs behind an exception barrier ----
ch_ctx* ch_create(int device_id, char* err, size_t errlen) {
  try { return ch_create_impl(device_id, err, errlen); }
  catch (const std::exception& e) { if (err && errlen) std::snprintf(err, errlen, "ch_create: %s", e.what()); }
  catch (...) { if (err && errlen) std::snprintf(err, errlen, "ch_create: unknown C++ exception"); }
  return nullptr;
}
ch_circuit* ch_circuit_build(ch_ctx* ctx, const ch_desc* d) {
  ch_circuit* c = nullptr;
  const int rc = guard_rc(ctx, [&] { c = ch_circuit_build_impl(ctx, d); return c ? CH_OK : CH_ERR_INVALID; });
  return rc == CH_OK ? c : nullptr;   // on an exception the partially built circuit was already released by its owner (see ch_circuit_build_impl)
}
int ch_set_samples(ch_circuit* c, int32_t n) { return guard_rc(c ? c->ctx : nullptr, [&] { return ch_set_samples_impl(c, n); }); }
int ch_set_params(ch_circuit* c, int32_t lo, int32_t hi, int32_t n_slots, const int32_t* ids, const double* values) {
  return guard_rc(c ? c->ctx : nullptr, [&] { return ch_set_params_impl(c, lo, hi, n_slots, ids, values); });
}
int ch_dc(ch_circuit* c, const ch_dc_opts* o, double* x_out, int32_t* status_out, ch_stats* stats) {
  return guard_rc(c ? c->ctx : nullptr, [&] { return ch_dc_impl(c, o, x_out, status_out, stats); });
}
int ch_tran(ch_circuit* c, double t0, double t1, const ch_tran_opts* o, ch_result** out) {
  if (out) *out = nullptr;
  return guard_rc(c ? c->ctx : nullptr, [&] { return ch_tran_impl(c, t0, t1, o, out); });
}
int ch_eval(ch_circuit* c, int32_t sample, const double* x_mna, double t, double alpha0, int32_t mode, double* F_out, double* Q_out, double* J_out) {
  return guard_rc(c ? c->ctx : nullptr, [&] { return ch_eval_impl(c, sample, x_mna, t, alpha0, mode, F_out, Q_out, J_out); });
}
int ch_ac(ch_circuit* c, const ch_dc_opts* o, int32_t n_freq, const double* freqs_hz, double* x_ac_out, ch_stats* stats) {
  return guard_rc(c ? c->ctx : nullptr, [&] { return ch_ac_impl(c, o, n_freq, freqs_hz, x_ac_out, stats); });
}
int ch_noise(ch_circuit* c, const ch_dc_opts* o, int32_t out_kind, int32_t out_index, int32_t n_freq, const double* freqs_hz, double* psd_out, ch_stats* stats) {
  return guard_rc(c ? c->ctx : nullptr, [&] { return ch_noise_impl(c, o, out_kind, out_index, n_freq, freqs_hz, psd_out, stats); });
}
int ch_bench_triad(ch_ctx* ctx, int64_t n, int32_t iters, double* gbps_out) { return guard_rc(ctx, [&] { return ch_bench_triad_impl(ctx, n, iters, gbps_out); }); }
int ch_bench_fp64(ch_ctx* ctx, int32_t iters, double* tflops_out) { return guard_rc(ctx, [&] { return ch_bench_fp64_impl(ctx, iters, tflops_out); }); }
int ch_debug_poison_lds(ch_ctx* ctx) { return guard_rc(ctx, [&] { return ch_debug_poison_lds_impl(ctx); }); }
int ch_debug_math(ch_ctx* ctx, int32_t which, int32_t n, const double* x, double* y) { return guard_rc(ctx, [&] { return ch_debug_math_impl(ctx, which, n, x, y); }); }
int ch_va_eval(ch_ctx* ctx, int32_t id, const double* par, const double* v, double temperature_k, double gmin, double* st_out) {
  return guard_rc(ctx, [&] { return ch_va_eval_impl(ctx, id, par, v, temperature_k, gmin, st_out); });
}
int ch_va_opvars(ch_ctx* ctx, int32_t id, const double* par, const double* v, double temperature_k, double gmin, double* op_out) {
  return guard_rc(ctx, [&] { return ch_va_opvars_impl(ctx, id, par, v, temperature_k, gmin, op_out); });
}
const char* ch_version(void) { return "cedarhip 0.3 (gfx950; fused block Newton, device-resident step controller)"; }

}  // extern "C"
