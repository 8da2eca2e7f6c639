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
#include "_generated/va_models.hpp"

namespace chip {

struct ClassMeta {
  int nc, ndev, nonlinear, nslots;   // nslots: lane slots of the evaluation phase (4 per MOSFET, 1 otherwise)
  int wl_ofs, n_work, spare0, spare1;  // gather work list (register-LU variants): offset inside the blob (ints, even), items
  int n_mat_src, n_vec_src, blob_ofs, blob_ints;            // list lengths; this class's packed list blob (ints)
};

// Everything a block needs to address its data, in ONE 96-byte record (one dependent load level):
// offsets, a copy of its class record and (when they are at most 8) its MOS class ids.
struct BlockMeta {
  int uofs, dofs, mc_n, mc_ofs;
  ClassMeta cm;
  int mc[8];
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
constexpr int KV_INLINE = 24;
// BSIM4 columns in LDS: one class per B4L_STRIDE doubles (even, so a column is a whole number of 16-byte pairs)
constexpr int B4L_STRIDE = (B4I_COUNT + 1) & ~1;

struct NewtonArgs {
  // ---- circuit structure ----
  const BlockMeta* bmeta;                    // [n_comp]
  const int* comp_class; const int* comp_uofs; const int* comp_dofs;
  const ClassMeta* classes; const int* blob;  // per class: [mat_ptr | vec_ptr | slots | mat_src u16 | vec_src u16], copied verbatim to LDS
  const int* dkind; const int* dterm; const int* dsrc; const int* dcls; const int* dhdev;
  const int* dcls_local;                     // per device: index into its block's MOS class list
  const int* comp_mc_ofs; const int* comp_mc_n; const int* mc_list;  // per block: distinct MOS classes
  const double* dpar; const double* dmult;   // [n_hdev * Spar]
  const double* vapar; long va_stride;       // parameter blocks of the compiled Verilog-A instances [Sva][va_stride] (dsrc[d] = offset, dcls_local[d] = module; va_stride = 0 when shared by all samples)
  const double* temp_s; int Stemp;           // Celsius per sample ($temperature of Verilog-A modules)
  const double* vacache; long vac_stride;    // per-instance constants of the Verilog-A instances (va_gen::setup), [Svac][vac_stride]; dvac[d] = offset
  const int* dvac;
  const double* mosp; long mos_cols;         // packed BSIM4 table [mos_cols][B4I_COUNT]
  const double* kv; const double* srcv;      // known-node values [Ssrc][nk], source values [Ssrc][nsrc]
  const unsigned char* dmask;                // per unknown: bit0 differential, bit1 branch current
  const unsigned char* active;               // per block (DC restarts) or null
  const double* gmin_s;                      // [Sgmin]
  const int* unk_obs;                        // per unknown: observable row it feeds, or -1
  int n_comp, S, Spar, Ssrc, Smos, Sgmin, nk, nsrc, n_unk, n_mos_cls, n_obs, max_mc;
  // ---- state ring: X, Qh are [n_slots][S][n_unk] ----
  double* X; double* Qh; long slot_stride;
  int hist_slot[8]; int cand_slot;
  // ---- step ----
  int mode, k, npred, nkm1, nkp1, maxit, inline_vals;
  double alpha[8], wpred[8], wkm1[8], wkp1[8];
  double ck, ckm1, ckp1;
  double abstol, reltol, newton_tol, dc_abstol, dv_max, gshunt;
  double vals_inline[KV_INLINE];             // [kv | srcv] when they fit and are sample-independent
  // ---- outputs ----
  BlockOut* out;
  Summary* summary;                          // output of reduce_blocks_kernel (mapped pinned host memory)
  double* rate;                              // [n_comp*S] last observed Newton convergence rate per block
  unsigned char* perm;                       // [n_comp*S][16] pivot order of the block's register LU, stored as row ^ lane (0 = identity), or null
  int reset_rate;                            // 1: ignore the stored rates (after a restart / failure)
  double* obs_row;                           // [n_obs][S] candidate row of the observable buffer, or null
  double* dumpA; double* dumpF; double* dumpQ; double* dumpC; int dump_stride;  // MODE_EVAL: per-block dense dumps (A = G + alpha0*C, C)
  unsigned long long* stamps;                // diagnostic build (-DCH_STAMPS): [block][8] cycle sums, else null
};

#ifdef CH_STAMPS
#define CH_STAMP(slot) do { if (tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); acc_[slot] += now_ - last_; last_ = now_; } } while (0)
#else
#define CH_STAMP(slot) do { } while (0)
#endif

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
// broadcast from a wave-uniform lane: v_readlane_b32 x2 (result lives in SGPRs)
__device__ __forceinline__ double bcast(double v, int src_lane) {
  const int l = __builtin_amdgcn_readfirstlane(src_lane);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// What the evaluation phase needs from the launch arguments (passed in registers to the
// non-inlined evaluator, so that its register allocation is independent of the solve phase).
struct EvalCtx {
  const int* dkind; const int* dterm; const int* dsrc; const int* dcls_local; const int* dhdev;
  const double* dpar; const double* dmult;
  int Spar;
  double gmin;
  const double* vapar;
  double temp_k;
  const double* vacache; const int* dvac;   // this sample's constant blocks of the compiled Verilog-A instances
};

// stamp record layouts: narrow [I(4)|Q(4)|G(4x4)|C(4x4)] = 40 doubles; wide (a compiled Verilog-A device is
// present) [I(8)|Q(8)|G(8x8)|C(8x8)] = 144 doubles
template <bool WIDE> struct StampLayout {
  static constexpr int STRIDE = WIDE ? 145 : 41, QO = WIDE ? 8 : 4, GO = WIDE ? 16 : 8, CO = WIDE ? 64 : 16, LD = WIDE ? 8 : 4;
};
__device__ __forceinline__ void widen_stamp(const double* t, double* st) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    st[k] = t[k]; st[8 + k] = t[4 + k];
#pragma unroll
    for (int j = 0; j < 4; ++j) { st[16 + k * 8 + j] = t[8 + k * 4 + j]; st[80 + k * 8 + j] = t[24 + k * 4 + j]; }
  }
}

// Linear devices (R, C, L, V, I, VCVS, VCCS; src/simpledevices.jl:49-373) into a narrow 40-double record.
__device__ __forceinline__ void eval_linear(const EvalCtx& a, int kind, int d, long pi, double m, const double (&v)[4], const double* svl, double* st) {
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
      const double i = m * svl[a.dsrc[d]];
      st[0] = i; st[1] = -i; st[4] = 0.0; st[5] = 0.0;
    } break;
    case K_V: case K_L: case K_VCVS_A: {
      // terminals (a, b, branch): KCL rows get ±m·i, branch row: va - vb - V(t) [- d/dt(L i)]
      const double ib = v[2];
      const double src = kind == K_V ? svl[a.dsrc[d]] : 0.0;
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
  }
}

// Evaluate one lane slot of the device-evaluation phase.  slot = (local device << 4) | (first << 3) | direction:
// compiled Verilog-A devices take one lane per unknown terminal (direction-parallel duals), every other device one lane.
__device__ __noinline__ void b4_device_call(const B4Col P, double vd, double vg, double vs, double vb, double gmin, double* out) {
  VA_KEEP_RETURN_ADDRESS;
  b4_device(P, vd, vg, vs, vb, gmin, out);
}
template <bool WIDE>
__device__ __forceinline__ void eval_slot(const EvalCtx a, int s, int dofs, int slot, const double* xl, int uofs,
                                          const double* kvl, const double* svl, const double* pl, double* stage) {
  const int dl = (slot >> 4) & 0x01ffffff;
  const int d = dofs + dl;
  double* st_final = stage + (size_t)dl * StampLayout<WIDE>::STRIDE;
  const int kind = a.dkind[d];
  const int* tm = a.dterm + NTERM * d;
  if (WIDE && kind == K_VA) {
    double vv[NTERM];
#pragma unroll
    for (int k = 0; k < NTERM; ++k) { const int t = tm[k]; vv[k] = t >= 0 ? xl[t - uofs] : kvl[-t - 1]; }
    const long pi = (long)a.dhdev[d] * a.Spar + (a.Spar > 1 ? s : 0);
    const va::Env env{a.temp_k, a.gmin};
    va_gen::stamp_dir_c(a.dcls_local[d], a.vapar + a.dsrc[d], a.vacache + a.dvac[d], vv, env, a.dmult[pi], slot & 7, (slot & 8) != 0,
                        (slot & (1 << 29)) ? ((slot >> 30) & 1) : -1, st_final);
    return;
  }
  double tmp40[WIDE ? 40 : 1];
  double* st = WIDE ? tmp40 : st_final;
  if (WIDE) {
#pragma unroll
    for (int j = 0; j < 40; ++j) tmp40[j] = 0.0;
  }
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int t = tm[k]; v[k] = t >= 0 ? xl[t - uofs] : kvl[-t - 1]; }
  const int hd = a.dhdev[d];
  const long pi = (long)hd * a.Spar + (a.Spar > 1 ? s : 0);
  const double m = a.dmult[pi];
  if (kind == K_MOS) {
    // parameters come from the block's LDS copy of its classes' packed columns (staged in the prologue)
    const B4Col P{pl + (size_t)a.dcls_local[d] * B4L_STRIDE};
    double o[40];
    // circuits with compiled Verilog-A devices (WIDE) reach the BSIM4 code through a call: inlined, its 6 k instructions and ~460
    // registers sit in every wide kernel three or four times over, whether the circuit holds a MOSFET or not
    if (WIDE) b4_device_call(P, v[0], v[1], v[2], v[3], a.gmin, o);
    else b4_device(P, v[0], v[1], v[2], v[3], a.gmin, o);
#pragma unroll
    for (int j = 0; j < 40; ++j) st[j] = m * o[j];
    if (WIDE) widen_stamp(st, st_final);
    return;
  }
  eval_linear(a, kind, d, pi, m, v, svl, st);
  if (WIDE) widen_stamp(st, st_final);
}

// A compiled Verilog-A lane slot of the device-resident stepper, resolved ONCE per transient: module, the instance's parameter and
// constant blocks (global memory, or the workgroup's LDS copies: tran_persistent_kernel stages them — a BSIM-CMG evaluation reads
// some hundred of these constants one dependent load at a time, an L2 round trip each when they stay in global memory),
// multiplicity and terminals.  mod < 0: not such a slot (eval_slot<true> handles it).
struct WideMeta {
  int mod, dl;
  const double* P; const double* C;
  double m;
  int t[NTERM];
};
__device__ __forceinline__ WideMeta load_wide_meta(const EvalCtx& a, int s, int dofs, int slot) {
  WideMeta q;
  q.mod = -1; q.dl = 0; q.P = nullptr; q.C = nullptr; q.m = 0.0;
#pragma unroll
  for (int k = 0; k < NTERM; ++k) q.t[k] = -1;
  if (slot < 0) return q;
  const int dl = (slot >> 4) & 0x01ffffff, d = dofs + dl;
  if (a.dkind[d] != K_VA) return q;
  q.dl = dl; q.mod = a.dcls_local[d];
  q.P = a.vapar + a.dsrc[d]; q.C = a.vacache + a.dvac[d];
  q.m = a.dmult[(long)a.dhdev[d] * a.Spar + (a.Spar > 1 ? s : 0)];
#pragma unroll
  for (int k = 0; k < NTERM; ++k) q.t[k] = a.dterm[NTERM * d + k];
  return q;
}
template <bool LDS>
__device__ __forceinline__ void eval_wide_cached(const WideMeta& q, const EvalCtx& a, int slot, const double* xl, int uofs, const double* kvl, double* stage) {
  double vv[NTERM];
#pragma unroll
  for (int k = 0; k < NTERM; ++k) { const int t = q.t[k]; vv[k] = t >= 0 ? xl[t - uofs] : kvl[-t - 1]; }
  const va::Env env{a.temp_k, a.gmin};
  const int part = (slot & (1 << 29)) ? ((slot >> 30) & 1) : -1;
  double* st = stage + (size_t)q.dl * StampLayout<true>::STRIDE;
  if (LDS) va_gen::stamp_dir_lds(q.mod, (va::lds_cptr)q.P, (va::lds_cptr)q.C, vv, env, q.m, slot & 7, (slot & 8) != 0, part, st);
  else va_gen::stamp_dir_c(q.mod, q.P, q.C, vv, env, q.m, slot & 7, (slot & 8) != 0, part, st);
}

// What one lane of the device-resident stepper evaluates, loaded ONCE per transient: a lane keeps its device for the whole time
// span, so the device tables are not read again inside the time loop (in the per-attempt kernel they cost a chain of three
// dependent global loads per evaluation, and seven table pointers held in scalar registers across the BSIM4 code).
struct SlotMeta {
  int kind;       // engine device kind, or 0 for an idle lane
  int dl;         // device index inside the block (its stamp record)
  int t[4];       // terminals: >= 0 block-local unknown, < 0: -(known index + 1)
  int cls, src;   // BSIM4 class inside the block's column list; device source slot
  double m, par;  // multiplier; first parameter of a linear device
};
__device__ __forceinline__ SlotMeta load_slot_meta(const EvalCtx& a, int s, int dofs, int uofs, int slot) {
  SlotMeta q;
  q.kind = 0; q.dl = 0; q.cls = 0; q.src = 0; q.m = 0.0; q.par = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) q.t[k] = -1;
  if (slot < 0) return q;
  const int dl = slot >> 4, d = dofs + dl;
  q.dl = dl; q.kind = a.dkind[d];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int t = a.dterm[NTERM * d + k]; q.t[k] = t >= 0 ? t - uofs : t; }
  const long pi = (long)a.dhdev[d] * a.Spar + (a.Spar > 1 ? s : 0);
  q.m = a.dmult[pi]; q.par = a.dpar[pi]; q.cls = a.dcls_local[d]; q.src = a.dsrc[d];
  return q;
}
// One HALF (or, PART < 0, the whole) of a narrow slot's evaluation from cached metadata.  For the wave pairs of the
// device-resident stepper (ch_persist.hpp) a BSIM4 instance is split by FUNCTION — PART 0 = threshold / mobility / drain current /
// output resistance / substrate and junction currents (the I and G entries of the record), PART 1 = the same threshold front end,
// then the capMod-2 intrinsic charges, junction and overlap charges (Q and C entries).  The value path dominates the model
// (r01_notes: 3 -> 1 derivative directions removes 1 000 of 6 300 instructions), so the split is along outputs, not along
// derivative directions: the compiler drops what a half does not store (3 925 and 3 395 instructions against 6 289).  Linear
// devices are evaluated whole by PART 0.
template <int PART>
__device__ __forceinline__ void eval_cached(const SlotMeta& q, double gmin, const double* xl, const double* kvl, const double* svl, const double* pl, double* stage) {
  if (q.kind == 0 || (PART == 1 && q.kind != K_MOS)) return;
  double* st = stage + (size_t)q.dl * StampLayout<false>::STRIDE;
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = q.t[k] >= 0 ? xl[q.t[k]] : kvl[-q.t[k] - 1];
  if (q.kind == K_MOS) {
    const B4Col P{pl + (size_t)q.cls * B4L_STRIDE};
    double o[40];
    b4_device(P, v[0], v[1], v[2], v[3], gmin, o);
    if (PART != 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) st[j] = q.m * o[j];
#pragma unroll
      for (int j = 8; j < 24; ++j) st[j] = q.m * o[j];
    }
    if (PART != 0) {
#pragma unroll
      for (int j = 4; j < 8; ++j) st[j] = q.m * o[j];
#pragma unroll
      for (int j = 24; j < 40; ++j) st[j] = q.m * o[j];
    }
    return;
  }
  const double m = q.m;
  switch (q.kind) {
    case K_R: {
      const double g = m / q.par, i = g * (v[0] - v[1]);
      st[0] = i; st[1] = -i; st[4] = 0.0; st[5] = 0.0;
      st[8] = g; st[9] = -g; st[12] = -g; st[13] = g;
      st[24] = 0.0; st[25] = 0.0; st[28] = 0.0; st[29] = 0.0;
    } break;
    case K_C: {
      const double c = m * q.par, qq = c * (v[0] - v[1]);
      st[0] = 0.0; st[1] = 0.0; st[4] = qq; st[5] = -qq;
      st[8] = 0.0; st[9] = 0.0; st[12] = 0.0; st[13] = 0.0;
      st[24] = c; st[25] = -c; st[28] = -c; st[29] = c;
    } break;
    case K_I: {
      const double i = m * svl[q.src];
      st[0] = i; st[1] = -i; st[4] = 0.0; st[5] = 0.0;
    } break;
    case K_V: case K_L: case K_VCVS_A: {
      const double ib = v[2];
      const double src = q.kind == K_V ? svl[q.src] : 0.0;
      const double l = q.kind == K_L ? q.par : 0.0;
      st[0] = m * ib; st[1] = -m * ib; st[2] = v[0] - v[1] - src;
      st[4] = 0.0; st[5] = 0.0; st[6] = -l * ib;
      st[8 + 2] = m; st[8 + 6] = -m; st[8 + 8] = 1.0; st[8 + 9] = -1.0; st[8 + 10] = 0.0;
      st[24 + 2] = 0.0; st[24 + 6] = 0.0; st[24 + 8] = 0.0; st[24 + 9] = 0.0; st[24 + 10] = -l;
    } break;
    case K_VCVS_B: {
      const double g = q.par;
      st[0] = -g * (v[1] - v[2]); st[4] = 0.0;
      st[8 + 1] = -g; st[8 + 2] = g; st[24 + 1] = 0.0; st[24 + 2] = 0.0;
    } break;
    case K_VCCS: {
      const double g = m * q.par, i = g * (v[2] - v[3]);
      st[0] = i; st[1] = -i; st[4] = 0.0; st[5] = 0.0;
      st[8 + 2] = g; st[8 + 3] = -g; st[8 + 6] = -g; st[8 + 7] = g;
      st[24 + 2] = 0.0; st[24 + 3] = 0.0; st[24 + 6] = 0.0; st[24 + 7] = 0.0;
    } break;
  }
}

// ---- DPP reductions over the 16-lane rows of a wavefront (no LDS round trip, ~8 cycles a step) ----
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
constexpr int DPP_QP_1032 = 0xB1, DPP_QP_2301 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
// result valid in every lane of rows 0 (NCV <= 16) or 0 and 1 (NCV == 32)
template <int NCV> __device__ __forceinline__ double row_max(double v) {
  v = fmax(v, dpp_d<DPP_QP_1032>(v)); v = fmax(v, dpp_d<DPP_QP_2301>(v));
  v = fmax(v, dpp_d<DPP_HALF_MIRROR>(v)); v = fmax(v, dpp_d<DPP_MIRROR>(v));
  if (NCV > 16) v = fmax(v, __shfl_xor(v, 16));
  return v;
}
template <int NCV> __device__ __forceinline__ double row_sum(double v) {
  v += dpp_d<DPP_QP_1032>(v); v += dpp_d<DPP_QP_2301>(v); v += dpp_d<DPP_HALF_MIRROR>(v); v += dpp_d<DPP_MIRROR>(v);
  if (NCV > 16) v += __shfl_xor(v, 16);
  return v;
}
template <int NCV> __device__ __forceinline__ int row_min_i(int v) {
  v = min(v, dpp_i<DPP_QP_1032>(v)); v = min(v, dpp_i<DPP_QP_2301>(v)); v = min(v, dpp_i<DPP_HALF_MIRROR>(v)); v = min(v, dpp_i<DPP_MIRROR>(v));
  if (NCV > 16) v = min(v, __shfl_xor(v, 16));
  return v;
}

template <int NCV> __device__ __forceinline__ int row_max_i(int v) {
  v = max(v, dpp_i<DPP_QP_1032>(v)); v = max(v, dpp_i<DPP_QP_2301>(v)); v = max(v, dpp_i<DPP_HALF_MIRROR>(v)); v = max(v, dpp_i<DPP_MIRROR>(v));
  if (NCV > 16) v = max(v, __shfl_xor(v, 16));
  return v;
}

// In-register dense LU with partial pivoting + solve by ONE wavefront: lane i owns row i of the
// augmented matrix [A | rhs] (r[0..NC], column indices are compile-time constants).  Pivot rows
// are not moved: a lane that has served as pivot is retired and its row is broadcast with
// v_readlane.  The pivot search is ONE 32-bit DPP max-reduction per step: the key is the high word
// of |r[k]| (monotonic for non-negative doubles: exponent + 14 leading mantissa bits) with its low
// 6 bits replaced by 63 - lane, so the largest magnitude wins and ties go to the lowest lane; the
// winner's lane comes out of the same word.  (A 64-bit fmax reduction plus a second reduction for
// the lane was 70 of the 130 instructions of a step.)  Returns false when the pivot column is zero
// or not finite.  On return lane i holds x_i in `sol` (i < nc).
template <int NC>
__device__ __forceinline__ bool lu_solve_regs(double (&r)[NC + 1], int nc, int lane, double& sol, int* myrow_out = nullptr) {
  bool done = lane >= nc;
  int mystep = -1;       // elimination step at which this lane's row became the pivot row
  double ipiv = 0.0;     // reciprocal of this lane's pivot
  int piv[NC];           // pivot lane of every step (wave-uniform)
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    piv[k] = 0;
    if (k < nc) {
      const int hi = __double2hiint(r[k]) & 0x7fffffff;
      const int key = done ? -1 : ((hi & ~63) | (63 - lane));
      const int best = __builtin_amdgcn_readfirstlane(row_max_i<NC>(key));
      if (best < 64 || best >= 0x7e300000) return false;   // |pivot| below the normal range, or above 2^996 / NaN
      const int bi = 63 - (best & 63);
      piv[k] = bi;
      const double ipk = frcp(bcast(r[k], bi));
      const double l = (!done && lane != bi) ? r[k] * ipk : 0.0;
#pragma unroll
      for (int j = k + 1; j <= NC; ++j) r[j] = fma(-l, bcast(r[j], bi), r[j]);
      if (lane == bi) { done = true; mystep = k; ipiv = ipk; }
    }
  }
  // back substitution: x_k is formed in the lane whose row was the k-th pivot, broadcast, and eliminated from the earlier pivot rows
  double rhs = 0.0;
#pragma unroll
  for (int j = 0; j <= NC; ++j) if (j == nc) rhs = r[j];
  double out = 0.0;
#pragma unroll
  for (int k = NC - 1; k >= 0; --k) {
    if (k < nc) {
      const double xk = bcast(rhs * ipiv, piv[k]);
      if (lane == k) out = xk;
      if (mystep >= 0 && mystep < k) rhs = fma(-r[k], xk, rhs);
    }
  }
  sol = out;
  if (myrow_out) {   // the row that was the pivot of step `lane`: loading the rows in this order puts every pivot on the diagonal
    int mr = lane;
#pragma unroll
    for (int k = 0; k < NC; ++k) if (k < nc && lane == k) mr = piv[k];
    *myrow_out = mr;
  }
  return true;
}

// ---- LU without a pivot search: rows pre-permuted into pivot order ----------------------------------------------------------
// The pivot search is what the register LU above spends its instructions on (1 500 for NC = 12: 258 v_readlane with a
// scalar-register lane select, 117 s_nop hazard pads, the retired-lane bookkeeping).  A circuit's Jacobian keeps its pivot order
// from one Newton iteration to the next almost always (KLU's refactorisation reuses it for the same reason), so the rows are loaded
// from LDS IN PIVOT ORDER — lane i takes row myrow(i), the row partial pivoting chose at step i the last time it ran — and the
// elimination runs with the pivot of step k in lane k: a compile-time lane, so every broadcast is ONE v_mov_b64_dpp row_newbcast:k
// (a DPP broadcast inside the 16-lane row; 64-bit DPP exists for exactly this control) and nothing is searched or retired.
// What partial pivoting guarantees, |multiplier| <= 1, is CHECKED instead of enforced: every lane keeps the largest multiplier
// magnitude it formed (one integer max per step on the high word), and ONE ballot at the end compares it with 8: beyond that, or
// with a vanishing / non-finite pivot, the solve is repeated by lu_solve_regs (natural row order, full search), which also hands
// back the new pivot order.  Threshold pivoting with |l| <= 8 loses at most three bits against |l| <= 1 — irrelevant for a Newton
// correction, and the same bound sparse direct solvers run with (KLU's default is 1000).
// Layout: r[0 .. NC-1] = matrix columns (zero beyond nc), r[NC] = right-hand side; rows beyond nc are zero.  NC <= 16.
template <int K> __device__ __forceinline__ double row_bcast(double v) {
  static_assert(K >= 0 && K < 16, "row_newbcast lane");
  // (keep the builtin's result in a typed temporary: used directly as an argument of an overloaded function, clang 19 took it for a
  //  64-bit INTEGER and converted it numerically)
  const double b = __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xF, 0xF, true);   // v_mov_b64_dpp row_newbcast:K (bound_ctrl: every lane is written, the old value is dead)
  return b;
}
// q += sum_j c[j] * x_j with x_j = the value lane j holds (j < nc <= NC <= 16): one DPP row broadcast per term
template <int NC, int J> struct RowDot {
  static __device__ __forceinline__ double run(const double (&c)[NC], double x, int nc, double q) {
    if constexpr (J < NC) {
      if (J < nc) { const double xj = row_bcast<J>(x); q = fma(c[J], xj, q); }
      return RowDot<NC, J + 1>::run(c, x, nc, q);
    } else return q;
  }
};

template <int NC, int K>
struct LuStaticStep {
  static __device__ __forceinline__ void fwd(double (&r)[NC + 1], int nc, int lane, int& lkey, double& ipiv) {
    if constexpr (K < NC) {
      if (K < nc) {
        const double ipk = frcp(row_bcast<K>(r[K]));      // every lane of the row forms the same reciprocal
        double pj[NC + 1];                                 // the pivot row, broadcast under the full exec mask
#pragma unroll
        for (int j = K + 1; j <= NC; ++j) pj[j] = row_bcast<K>(r[j]);
        if (lane == K) ipiv = ipk;
        if (lane > K) {                                    // one exec-masked region: the multiplier and the row update
          const double l = r[K] * ipk;
          lkey = max(lkey, __double2hiint(l) & 0x7fffffff);
#pragma unroll
          for (int j = K + 1; j <= NC; ++j) r[j] = fma(-l, pj[j], r[j]);
        }
      }
      LuStaticStep<NC, K + 1>::fwd(r, nc, lane, lkey, ipiv);
    }
  }
  // back substitution, Q from NC-1 down to 0 (instantiated as K = NC-1-Q)
  static __device__ __forceinline__ void bwd(const double (&r)[NC + 1], int nc, int lane, double& rhs, double ipiv) {
    if constexpr (K < NC) {
      constexpr int Q = NC - 1 - K;
      if (Q < nc && Q > 0) {
        const double xk = row_bcast<Q>(rhs * ipiv);        // lane Q holds x_Q = rhs_Q / pivot_Q
        if (lane < Q) rhs = fma(-r[Q], xk, rhs);
      }
      LuStaticStep<NC, K + 1>::bwd(r, nc, lane, rhs, ipiv);
    }
  }
};
template <int NC>
__device__ __forceinline__ bool lu_solve_static(double (&r)[NC + 1], int nc, int lane_in, double& sol) {
  static_assert(NC <= 16, "row_newbcast spans one 16-lane row");
  // The lane id is made opaque: otherwise every predicate below (lane == K, lane > K, lane < Q for every K) is hoisted out of the
  // caller's loops as a 64-bit mask in scalar registers — three dozen pairs that spill, each use then costing v_readlane pairs
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  int lkey = 0; double ipiv = 0.0;
  LuStaticStep<NC, 0>::fwd(r, nc, lane, lkey, ipiv);
  // |l| <= 8 everywhere: 0x40200000 is the high word of 8.0; a NaN or an infinity (vanished or non-finite pivot) compares above it.
  // (The LAST pivot has no multipliers: a vanishing one shows as a non-finite solution, which every caller tests for.)
  if (__ballot(lane < nc && lkey > 0x40200000)) return false;
  double rhs = r[NC];
  LuStaticStep<NC, 0>::bwd(r, nc, lane, rhs, ipiv);
  sol = rhs * ipiv;
  return true;
}
// The solve of one block from its LDS image A[nc][lda] (column nc = right-hand side): static pivot order first, full search on
// failure.  `myrow` (per lane, kept by the caller across solves) is the pivot order; it starts as the identity.
template <int NC>
__device__ __forceinline__ bool lu_solve_block(const double* A, int lda, int nc, int lane, int& myrow, double& sol) {
  const bool mine = lane < nc;
  double r[NC + 1];
  if constexpr (NC <= 16) {
#pragma unroll
    for (int j = 0; j < NC; ++j) r[j] = (mine && j < nc) ? A[myrow * lda + j] : 0.0;
    r[NC] = mine ? A[myrow * lda + nc] : 0.0;
    if (lu_solve_static<NC>(r, nc, lane, sol)) return true;
  }
#pragma unroll
  for (int j = 0; j <= NC; ++j) r[j] = (mine && j <= nc) ? A[lane * lda + j] : 0.0;
  return lu_solve_regs<NC>(r, nc, lane, sol, &myrow);
}

// Bordered block-diagonal form (ch_analysis.hpp, tearing): the block's last nb rows / columns are its replicas of the border
// unknowns.  lu_bbd_factor eliminates the block's OWN unknowns only (pivots among lanes < no, columns 0..no-1) from every row,
// the border rows included: what is left in the border lanes, columns no..nc, is this block's contribution [S_i | g_i] to the
// Schur complement on the border.  After the contributions of all blocks have been summed and the border solved (dxb),
// lu_bbd_back substitutes back through the block's pivot rows.  Same pivot rule and broadcasts as lu_solve_regs.
template <int NC>
__device__ __forceinline__ bool lu_bbd_factor(double (&r)[NC + 1], int no, int nc, int lane, int (&piv)[NC], int& mystep, double& ipiv) {
  bool done = lane >= nc;
  mystep = -1; ipiv = 0.0;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    piv[k] = 0;
    if (k < no) {
      const int hi = __double2hiint(r[k]) & 0x7fffffff;
      const int key = (done || lane >= no) ? -1 : ((hi & ~63) | (63 - lane));
      const int best = __builtin_amdgcn_readfirstlane(row_max_i<NC>(key));
      if (best < 64 || best >= 0x7e300000) return false;
      const int bi = 63 - (best & 63);
      piv[k] = bi;
      const double ipk = frcp(bcast(r[k], bi));
      const double l = (!done && lane != bi) ? r[k] * ipk : 0.0;
#pragma unroll
      for (int j = k + 1; j <= NC; ++j) r[j] = fma(-l, bcast(r[j], bi), r[j]);
      if (lane == bi) { done = true; mystep = k; ipiv = ipk; }
    }
  }
  return true;
}
// r[] as left by lu_bbd_factor; dxb0 / dxb1 = the border solution (wave-uniform).  On return lane i < no holds dx_i, lane no + b holds dxb_b.
template <int NC>
__device__ __forceinline__ double lu_bbd_back(const double (&r)[NC + 1], int no, int nc, int lane, const int (&piv)[NC], int mystep, double ipiv, double dxb0, double dxb1) {
  double rhs = 0.0, c0 = 0.0, c1 = 0.0;
#pragma unroll
  for (int j = 0; j <= NC; ++j) { if (j == nc) rhs = r[j]; if (j == no) c0 = r[j]; if (j == no + 1 && j < nc) c1 = r[j]; }
  rhs = fma(-c0, dxb0, rhs);
  rhs = fma(-c1, dxb1, rhs);
  double out = lane == no ? dxb0 : (lane == no + 1 ? dxb1 : 0.0);
#pragma unroll
  for (int k = NC - 1; k >= 0; --k) {
    if (k < no) {
      const double xk = bcast(rhs * ipiv, piv[k]);
      if (lane == k) out = xk;
      if (mystep >= 0 && mystep < k) rhs = fma(-r[k], xk, rhs);
    }
  }
  return out;
}

// LDS fallback for blocks larger than the register variant (single wave, barriers are wave-local fences)
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
__device__ inline bool lu_solve_lds(double* A, int lda, int nc, int lane) {
  for (int k = 0; k < nc; ++k) {
    double best = -1.0; int bi = k;
    for (int i = k + lane; i < nc; i += 64) { const double v = fabs(A[i * lda + k]); if (v > best) { best = v; bi = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (!(best > 0.0) || !(best < 1e300)) return false;
    if (bi != k) { for (int j = lane; j <= nc; j += 64) { const double t = A[k * lda + j]; A[k * lda + j] = A[bi * lda + j]; A[bi * lda + j] = t; } wave_fence(); }
    const double inv = 1.0 / A[k * lda + k];
    const int rem = nc - k - 1, w = rem + 1;
    for (int i = k + 1 + lane; i < nc; i += 64) A[i * lda + k] *= inv;
    wave_fence();
    for (int e = lane; e < rem * w; e += 64) { const int i = k + 1 + e / w, j = k + 1 + e % w; A[i * lda + j] -= A[i * lda + k] * A[k * lda + j]; }
    wave_fence();
  }
  for (int k = nc - 1; k >= 0; --k) {
    const double xk = A[k * lda + nc] / A[k * lda + k];
    wave_fence();
    if (lane == 0) A[k * lda + nc] = xk;
    for (int i = lane; i < k; i += 64) A[i * lda + nc] -= A[i * lda + k] * xk;
    wave_fence();
  }
  return true;
}

// LDS layout (doubles): st[ndev*41] | A[nc*(nc+1)] | Cm[nc*nc] | xl xp F Q hq dx w qn pm pp x0 dm [12*nc] | kvl[nk] svl[nsrc]
//                       | ints: mptr[nc*nc+1] vptr[nc+1] slots[nslots] | u16: msrc[] vsrc[]
template <int NC, bool WIDE = false>
__global__ __launch_bounds__(256, 1) void newton_block_kernel(const NewtonArgs a) {
  typedef StampLayout<WIDE> SL;
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int blk = blockIdx.x;
  const int c = blk / a.S, s = blk - c * a.S;
  const BlockMeta bm = a.bmeta[c];
  const ClassMeta cm = bm.cm;
  const int nc = cm.nc, ndev = cm.ndev, uofs = bm.uofs, dofs = bm.dofs;
  const int lda = nc + 1;
  double* st = lds;
  double* A = st + (size_t)ndev * SL::STRIDE;
  double* Cm = A + (size_t)nc * lda;
  double* xl = Cm + (size_t)nc * nc;
  double* xp = xl + nc; double* Fv = xp + nc; double* Qv = Fv + nc; double* hq = Qv + nc;
  double* dxv = hq + nc; double* wv = dxv + nc; double* qn = wv + nc;
  double* pm = qn + nc; double* pp = pm + nc;          // predictors of order k-1 / k+1 (local-error estimates of the epilogue)
  double* x0l = pp + nc;                               // accepted state of the previous step (error weights of the epilogue)
  int* dml = (int*)(x0l + nc);                         // unknown flags (bit 0 differential, bit 1 branch row): read once per launch
  double* kvl = x0l + 2 * nc; double* svl = kvl + a.nk;
  double* pl = svl + a.nsrc;                       // [max_mc][B4L_STRIDE] packed BSIM4 columns of this block's classes
  int* mptr = (int*)(pl + (size_t)a.max_mc * B4L_STRIDE);  // start of the class blob copy
  int* vptr = mptr + (nc * nc + 1);
  int* slots = vptr + (nc + 1);
  uint16_t* msrc = (uint16_t*)(slots + cm.nslots);
  uint16_t* vsrc = msrc + cm.n_mat_src;
  int* mcl = mptr + cm.blob_ints;                  // [64] this block's MOS class list (behind the blob)
  const int2* wl = (const int2*)(mptr + cm.wl_ofs);  // gather work list: {first source, (sources << 16) | vector flag << 15 | entry}
  const bool use_wl = NC > 0 && nc <= NC;          // register LU: A and C in LDS are never overwritten, so only structural non-zeros are gathered
  __shared__ int s_ctl[4];  // [0] loop control (0 continue, 1 stop), [1] status, [2] iters
  __shared__ double s_fnorm;
  const long sofs = (long)s * a.n_unk + uofs;
  const double* X0 = a.X + (long)a.hist_slot[0] * a.slot_stride + sofs;
  const bool is_active = !(a.active && !a.active[blk]);
#ifdef CH_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif

  if (is_active) {
    // ---- prologue: stage lists / known values / BSIM4 columns in LDS, predictor and history term ----
    // All global loads of one level are issued before any is consumed (batches of 8 per thread), so
    // the prologue costs a few memory latencies instead of one per element.
    constexpr int PB = 18;  // generic path: doubles per thread and batch
    constexpr int PP = 9;   // fast path: 16-byte pairs per lane (8 DFF classes x 69 pairs / 64 lanes = 8.6)
    constexpr int NPAIR = B4L_STRIDE / 2;
    const long scol = a.Smos > 1 ? s : 0;
    const bool fast = nthr == 64 && bm.mc_n <= 8 && cm.blob_ints <= 2 * 256 && bm.mc_n * NPAIR <= PP * 64 && nc <= 64;
    if (fast) {
      // One wave, everything fits one batch per lane: ALL global loads of the prologue (class blob, BSIM4 columns of the
      // block's classes, state history) are issued back to back before the first one is consumed — one memory latency
      // instead of three dependent ones (the class ids come from the BlockMeta registers, not from LDS).  The cost of this
      // phase was measured to be per load INSTRUCTION (~280 cycles each with 1024 waves loading at once), not per byte, so
      // the loads are as wide as the data allows: 16 bytes per lane for the blob and the BSIM4 columns, and history points
      // the current order does not need are skipped.
      typedef int i4u __attribute__((ext_vector_type(4), aligned(16)));
      typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
      const i4u* src = (const i4u*)(a.blob + cm.blob_ofs);    // class blobs start on 16-byte boundaries and are padded to 4 ints
      const int nq = cm.blob_ints >> 2, npair = bm.mc_n * NPAIR;
      i4u bv[2]; d2u pv[PP]; double x0 = 0.0, xv[7], qv[5];
      const int nxh = max(a.npred, max(a.nkm1, a.nkp1));   // history points any of the three predictors needs
#pragma unroll
      for (int u = 0; u < 2; ++u) { const int i = tid + u * 64; bv[u] = src[i < nq ? i : nq - 1]; }
      if (npair > 0) {
        // Pair q = 64u + lane of the concatenated columns (69 pairs per class; the last pair of a column reads one double
        // past it, the table is padded for that): lane 0's class j0 and offset r0 are compile-time constants of the
        // unrolled loop and a batch of 64 pairs straddles at most one class boundary, so the address is one compare and
        // one select between two wave-uniform bases (no per-lane division / 64-bit multiply chain).
        int cl = bm.mc[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) cl = (bm.mc_n - 1 == q) ? bm.mc[q] : cl;
        const long last = ((long)cl * a.Smos + scol) * (long)B4I_COUNT + 2 * (NPAIR - 1);
#pragma unroll
        for (int u = 0; u < PP; ++u) {
          const int j0 = (u * 64) / NPAIR, r0 = u * 64 - j0 * NPAIR;
          const int ja = j0 < 8 ? j0 : 7, jb = j0 + 1 < 8 ? j0 + 1 : 7;
          const long oa = ((long)bm.mc[ja] * a.Smos + scol) * (long)B4I_COUNT + 2 * r0;
          const long ob = ((long)bm.mc[jb] * a.Smos + scol) * (long)B4I_COUNT + 2 * (r0 - NPAIR);
          long idx = (r0 + tid >= NPAIR ? ob : oa) + 2 * tid;
          if (tid + u * 64 >= npair) idx = last;   // clamped: branch-free loads
          pv[u] = *(const d2u*)(a.mosp + idx);
        }
      }
      const int iu = tid < nc ? tid : 0;
      x0 = X0[iu];
      const int dmr = a.dmask[uofs + iu] | (a.obs_row ? (a.unk_obs[uofs + iu] + 1) << 8 : 0);   // flags | (observable row + 1) << 8
      if (a.mode == MODE_TRAN) {
#pragma unroll
        for (int j = 0; j < 7; ++j) { xv[j] = 0.0; if (j < nxh) xv[j] = a.X[(long)a.hist_slot[j] * a.slot_stride + sofs + iu]; }
#pragma unroll
        for (int j = 0; j < 5; ++j) { qv[j] = 0.0; if (j < a.k) qv[j] = a.Qh[(long)a.hist_slot[j] * a.slot_stride + sofs + iu]; }
      }
      if (a.inline_vals) {
        // known-node and source values travel in the kernel arguments: lane i reads entry i of that array with one
        // vector load from the kernarg segment (kvl and svl are contiguous in LDS)
        typedef const double __attribute__((address_space(4)))* karg_d;
        const karg_d kp = (karg_d)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(NewtonArgs, vals_inline));
        const double kin = kp[tid < KV_INLINE ? tid : 0];
        if (tid < a.nk + a.nsrc) kvl[tid] = kin;
      } else {
        const double* kg = a.kv + (long)(a.Ssrc > 1 ? s : 0) * a.nk;
        const double* sg = a.srcv + (long)(a.Ssrc > 1 ? s : 0) * a.nsrc;
        for (int i = tid; i < a.nk; i += nthr) kvl[i] = kg[i];
        for (int i = tid; i < a.nsrc; i += nthr) svl[i] = sg[i];
      }
      // branch-free stores: an out-of-range lane holds the (clamped) last element and rewrites it in place
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 64, q = (i < nq ? i : nq - 1) * 4;
        mptr[q] = bv[u].x; mptr[q + 1] = bv[u].y; mptr[q + 2] = bv[u].z; mptr[q + 3] = bv[u].w;
      }
      if (npair > 0) {
#pragma unroll
        for (int u = 0; u < PP; ++u) { const int e = tid + u * 64, q = (e < npair ? e : npair - 1) * 2; pl[q] = pv[u].x; pl[q + 1] = pv[u].y; }
      }
      if (tid < nc) {
        double p = x0, h = 0.0;
        if (a.mode == MODE_TRAN) {
          p = 0.0;
#pragma unroll
          for (int j = 0; j < 7; ++j) p += (j < a.npred ? a.wpred[j + 1] : 0.0) * xv[j];
#pragma unroll
          for (int j = 0; j < 5; ++j) h += (j < a.k ? a.alpha[j + 1] : 0.0) * qv[j];
          double m1 = 0.0, p1 = 0.0;
#pragma unroll
          for (int j = 0; j < 7; ++j) { m1 += (j < a.nkm1 ? a.wkm1[j + 1] : 0.0) * xv[j]; p1 += (j < a.nkp1 ? a.wkp1[j + 1] : 0.0) * xv[j]; }
          pm[tid] = m1; pp[tid] = p1;
        }
        xp[tid] = p; xl[tid] = p; hq[tid] = h; qn[tid] = 0.0;
        wv[tid] = 1.0 / (a.reltol * fabs(x0) + a.abstol);
        x0l[tid] = x0; dml[tid] = dmr;
      }
    } else {
    {
      const int nmc = bm.mc_n, mco = bm.mc_ofs;
      if (nmc <= 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) if (tid == j && j < nmc) mcl[j] = bm.mc[j];
      } else if (tid < nmc) mcl[tid] = a.mc_list[mco + tid];
      // the class's gather lists and slot table: one contiguous int blob, one batch of loads
      {
        const int* src = a.blob + cm.blob_ofs;
        const int n = cm.blob_ints;
        for (int base = tid; base < n; base += nthr * 8) {
          int v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int i = base + u * nthr; v[u] = src[i < n ? i : n - 1]; }  // clamped: branch-free, all 8 loads in flight
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int i = base + u * nthr; if (i < n) mptr[i] = v[u]; }
        }
      }
      if (a.inline_vals) {
#pragma unroll
        for (int i = 0; i < KV_INLINE; ++i) if (tid == i && i < a.nk + a.nsrc) kvl[i] = a.vals_inline[i];  // kvl and svl are contiguous
      } else {
        const double* kg = a.kv + (long)(a.Ssrc > 1 ? s : 0) * a.nk;
        const double* sg = a.srcv + (long)(a.Ssrc > 1 ? s : 0) * a.nsrc;
        for (int i = tid; i < a.nk; i += nthr) kvl[i] = kg[i];
        for (int i = tid; i < a.nsrc; i += nthr) svl[i] = sg[i];
      }
      __syncthreads();  // mcl visible
      const int total = nmc * B4I_COUNT;
      if (total > 0) for (int base = tid; base < total; base += nthr * PB) {
        double v[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int e0 = base + u * nthr, e = e0 < total ? e0 : total - 1;  // clamped: branch-free loads
          const int j = e / B4I_COUNT, i = e - j * B4I_COUNT;
          v[u] = a.mosp[((long)mcl[j] * a.Smos + scol) * (long)B4I_COUNT + i];
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) { const int e = base + u * nthr; if (e < total) { const int j = e / B4I_COUNT; pl[e + j * (B4L_STRIDE - B4I_COUNT)] = v[u]; } }
      }
    }
    for (int i = tid; i < nc; i += nthr) {
      double x0 = X0[i];
      double p = x0, h = 0.0;
      if (a.mode == MODE_TRAN) {
        double xv[7], qv[5];  // all history loads are issued before any is used
        const int nxh2 = max(a.npred, max(a.nkm1, a.nkp1));
#pragma unroll
        for (int j = 0; j < 7; ++j) xv[j] = a.X[(long)a.hist_slot[j < nxh2 ? j : 0] * a.slot_stride + sofs + i];
#pragma unroll
        for (int j = 0; j < 5; ++j) qv[j] = a.Qh[(long)a.hist_slot[j < a.k ? j : 0] * a.slot_stride + sofs + i];
        p = 0.0;
        double m1 = 0.0, p1 = 0.0;
#pragma unroll
        for (int j = 0; j < 7; ++j) { p += (j < a.npred ? a.wpred[j + 1] : 0.0) * xv[j]; m1 += (j < a.nkm1 ? a.wkm1[j + 1] : 0.0) * xv[j]; p1 += (j < a.nkp1 ? a.wkp1[j + 1] : 0.0) * xv[j]; }
        pm[i] = m1; pp[i] = p1;
#pragma unroll
        for (int j = 0; j < 5; ++j) h += (j < a.k ? a.alpha[j + 1] : 0.0) * qv[j];
      }
      xp[i] = p; xl[i] = p; hq[i] = h; qn[i] = 0.0;
      wv[i] = 1.0 / (a.reltol * fabs(x0) + a.abstol);
      x0l[i] = x0; dml[i] = a.dmask[uofs + i] | (a.obs_row ? (a.unk_obs[uofs + i] + 1) << 8 : 0);
    }
    }
    if (use_wl) for (int i = tid; i < nc * lda + nc * nc; i += nthr) A[i] = 0.0;   // A and Cm are contiguous
    if (tid == 0) { s_ctl[0] = 0; s_ctl[1] = 1; s_ctl[2] = 0; }
    __syncthreads();
    CH_STAMP(0);
    const double alpha0 = a.mode == MODE_DC ? 0.0 : a.alpha[0];
    const int maxit = a.mode == MODE_EVAL ? 1 : a.maxit;
    const double rate_prev = (a.mode == MODE_TRAN && !a.reset_rate) ? a.rate[blk] : 1.0;
    double rate_new = -1.0, dn_prev = 0.0;  // wave 0 only
    // pivot order of this block's register LU (lu_solve_block), kept from launch to launch
    constexpr bool keep_order = NC > 0 && NC <= 16;
    int myrow = lane;
    if (keep_order && a.perm && wave == 0 && lane < nc && nc <= NC) myrow = lane ^ (int)a.perm[(long)blk * 16 + lane];
    const int myrow_in = myrow;
    const EvalCtx ectx{a.dkind, a.dterm, a.dsrc, a.dcls_local, a.dhdev, a.dpar, a.dmult, a.Spar, a.gmin_s[a.Sgmin > 1 ? s : 0], a.vapar + (long)s * a.va_stride,
                       WIDE ? a.temp_s[a.Stemp > 1 ? s : 0] + 273.15 : 300.15, a.vacache + (long)s * a.vac_stride, a.dvac};
    for (int it = 0; it <= maxit; ++it) {
      // (1) device evaluation → staging (all waves)
      for (int q = tid; q < cm.nslots; q += nthr) { const int sl = slots[q]; if (sl >= 0) eval_slot<WIDE>(ectx, s, dofs, sl, xl, uofs, kvl, svl, pl, st); }
      __syncthreads();
      CH_STAMP(1);
      // (2) gather (all waves)
      if (use_wl) {
        // one work item per structural non-zero of A/C (plus every diagonal) and per row of F/Q: a DFF block has ~55 items,
        // one pass of one wave, instead of two passes over the 121 matrix entries and a third over the rows
        for (int w = tid; w < cm.n_work; w += nthr) {
          const int2 it = wl[w];
          const int p0 = it.x, pe = p0 + (int)((unsigned)it.y >> 16), e = it.y & 0x7fff;
          const bool vec = it.y & 0x8000;
          const int off2 = vec ? SL::QO : SL::CO;
          double s1 = 0.0, s2 = 0.0;
          for (int p = p0; p < pe; p += 4) {   // four sources in flight per trip (clamped indices, predicated adds): sequential summation order
            const int l = pe - 1;
            const int o0 = msrc[p], o1 = msrc[min(p + 1, l)], o2 = msrc[min(p + 2, l)], o3 = msrc[min(p + 3, l)];
            const double a0 = st[o0], b0 = st[o0 + off2], a1 = st[o1], b1 = st[o1 + off2], a2 = st[o2], b2 = st[o2 + off2], a3 = st[o3], b3 = st[o3 + off2];
            s1 += a0; s2 += b0;
            if (p + 1 < pe) { s1 += a1; s2 += b1; }
            if (p + 2 < pe) { s1 += a2; s2 += b2; }
            if (p + 3 < pe) { s1 += a3; s2 += b3; }
          }
          if (vec) {
            if (a.gshunt != 0.0 && !(dml[e] & 2)) s1 += a.gshunt * xl[e];
            Qv[e] = s2;
            const double F = s1 + alpha0 * s2 + hq[e];
            Fv[e] = F;
            A[e * lda + nc] = -F;
          } else {
            const int r = e / nc, col = e - r * nc;
            if (r == col && a.gshunt != 0.0 && !(dml[r] & 2)) s1 += a.gshunt;  // node rows only
            A[r * lda + col] = s1 + alpha0 * s2;
            Cm[e] = s2;
          }
        }
      } else {
      for (int e = tid; e < nc * nc; e += nthr) {
        double g = 0.0, cc = 0.0;
        // four sources in flight per trip (clamped indices, predicated adds): same summation order, a quarter of the LDS round trips
        for (int p = mptr[e], pe = mptr[e + 1]; p < pe; p += 4) {
          const int l = pe - 1;
          const int o0 = msrc[p], o1 = msrc[min(p + 1, l)], o2 = msrc[min(p + 2, l)], o3 = msrc[min(p + 3, l)];
          const double g0 = st[o0], c0 = st[o0 + SL::CO], g1 = st[o1], c1 = st[o1 + SL::CO], g2 = st[o2], c2 = st[o2 + SL::CO], g3 = st[o3], c3 = st[o3 + SL::CO];
          g += g0; cc += c0;
          if (p + 1 < pe) { g += g1; cc += c1; }
          if (p + 2 < pe) { g += g2; cc += c2; }
          if (p + 3 < pe) { g += g3; cc += c3; }
        }
        const int r = e / nc, col = e - r * nc;
        if (r == col && a.gshunt != 0.0 && !(dml[r] & 2)) g += a.gshunt;  // node rows only
        A[r * lda + col] = g + alpha0 * cc;
        Cm[e] = cc;
      }
      for (int i = tid; i < nc; i += nthr) {
        double f = 0.0, q = 0.0;
        for (int p = vptr[i], pe = vptr[i + 1]; p < pe; p += 4) {
          const int l = pe - 1;
          const int o0 = vsrc[p], o1 = vsrc[min(p + 1, l)], o2 = vsrc[min(p + 2, l)], o3 = vsrc[min(p + 3, l)];
          const double f0 = st[o0], q0 = st[o0 + SL::QO], f1 = st[o1], q1 = st[o1 + SL::QO], f2 = st[o2], q2 = st[o2 + SL::QO], f3 = st[o3], q3 = st[o3 + SL::QO];
          f += f0; q += q0;
          if (p + 1 < pe) { f += f1; q += q1; }
          if (p + 2 < pe) { f += f2; q += q2; }
          if (p + 3 < pe) { f += f3; q += q3; }
        }
        if (a.gshunt != 0.0 && !(dml[i] & 2)) f += a.gshunt * xl[i];
        Qv[i] = q;
        const double F = f + alpha0 * q + hq[i];
        Fv[i] = F;
        A[i * lda + nc] = -F;
      }
      }
      __syncthreads();
      CH_STAMP(2);
      // (3)+(4) solve, update, convergence: wave 0 only.  Register variant (nc <= NC): lane i keeps
      // row i of [A | -F], row i of C, F_i, Q_i, x_i in VGPRs; reductions are DPP; no LDS round trips.
      if (wave == 0) {
        int status = 1, stop = 0;
        const bool regs = (NC > 0 && nc <= NC);
        if (a.mode == MODE_EVAL) {
          if (a.dumpA) {
            double* dA = a.dumpA + (long)blk * a.dump_stride * a.dump_stride;
            for (int e = lane; e < nc * nc; e += 64) dA[e] = A[(e / nc) * lda + (e % nc)];
            if (a.dumpC) { double* dC = a.dumpC + (long)blk * a.dump_stride * a.dump_stride; for (int e = lane; e < nc * nc; e += 64) dC[e] = Cm[e]; }
            for (int i = lane; i < nc; i += 64) { a.dumpF[(long)blk * a.dump_stride + i] = Fv[i]; a.dumpQ[(long)blk * a.dump_stride + i] = Qv[i]; }
          }
          for (int i = lane; i < nc; i += 64) qn[i] = Qv[i];
          status = 0; stop = 1;
        } else if (regs) {
          constexpr int NCR = NC > 0 ? NC : 1;
          const bool mine = lane < nc;
          double cr[NCR];
#pragma unroll
          for (int j = 0; j < NCR; ++j) cr[j] = (mine && j < nc) ? Cm[lane * nc + j] : 0.0;
          const double Fi = mine ? Fv[lane] : 0.0, Qi = mine ? Qv[lane] : 0.0, xi = mine ? xl[lane] : 0.0, wi = mine ? wv[lane] : 0.0;
          const bool is_node = mine && !(dml[mine ? lane : 0] & 2);
          const double fnorm = bcast(row_max<NCR>(fabs(Fi)), 0);
          if (lane == 0) s_fnorm = fnorm;
          CH_STAMP(6);
          if (!(fnorm == fnorm) || fnorm > 1e300) { status = 2; stop = 1; }
          else if (a.mode == MODE_DC && fnorm < a.dc_abstol) { if (mine) qn[lane] = Qi; status = 0; stop = 1; }
          else if (it == maxit) { stop = 1; }
          else {
            double dx = 0.0;
            const bool ok = lu_solve_block<NCR>(A, lda, nc, lane, myrow, dx);
            CH_STAMP(7);
            if (!ok) { status = 2; stop = 1; }
            else {
              if (a.mode == MODE_DC && a.dv_max > 0.0 && cm.nonlinear) {  // linear blocks take the full Newton step
                const double mm = bcast(row_max<NCR>(is_node ? fabs(dx) : 0.0), 0);
                if (mm > a.dv_max) dx *= a.dv_max / mm;
              }
              const double xn = xi + dx;
              if (mine) xl[lane] = xn;
              const bool bad = mine && (!(xn == xn) || fabs(xn) > 1e300);
              const double t = dx * wi;
              const double e2 = bcast(row_sum<NCR>(mine ? t * t : 0.0), 0);
              if (a.mode == MODE_TRAN) {
                double q = Qi;
                if constexpr (NCR <= 16) q = RowDot<NCR, 0>::run(cr, dx, nc, q);
                else {
#pragma unroll
                  for (int j = 0; j < NCR; ++j) if (j < nc) q = fma(cr[j], bcast(dx, j), q);
                }
                if (mine) qn[lane] = q;
              }
              if (lane == 0) s_ctl[2] += 1;
              if (__ballot(bad)) { status = 2; stop = 1; }
              else if (a.mode == MODE_TRAN) {
                // IDA-style rate test: see oracle.cpp (same policy)
                const double dn = sqrt(e2 / nc);
                if (it == 0) { if (dn <= a.newton_tol || (rate_prev < 0.9 && 2.0 * fmax(rate_prev, 0.02) * dn <= a.newton_tol)) { status = 0; stop = 1; } }
                else { rate_new = dn_prev > 0.0 ? dn / dn_prev : 0.0; if (dn <= a.newton_tol) { status = 0; stop = 1; } }
                dn_prev = dn;
              }
            }
          }
        } else {
          double m = 0.0;
          for (int i = lane; i < nc; i += 64) m = fmax(m, fabs(Fv[i]));
          const double fnorm = wave_max(m);
          if (lane == 0) s_fnorm = fnorm;
          if (!(fnorm == fnorm) || fnorm > 1e300) { status = 2; stop = 1; }
          else if (a.mode == MODE_DC && fnorm < a.dc_abstol) { for (int i = lane; i < nc; i += 64) qn[i] = Qv[i]; status = 0; stop = 1; }
          else if (it == maxit) { stop = 1; }
          else {
            const bool ok = lu_solve_lds(A, lda, nc, lane);
            if (!ok) { status = 2; stop = 1; }
            else {
              double scale = 1.0;
              if (a.mode == MODE_DC && a.dv_max > 0.0 && cm.nonlinear) {
                double mm = 0.0;
                for (int i = lane; i < nc; i += 64) if (!(dml[i] & 2)) mm = fmax(mm, fabs(A[i * lda + nc]));
                mm = wave_max(mm);
                if (mm > a.dv_max) scale = a.dv_max / mm;
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
              wave_fence();
              if (a.mode == MODE_TRAN) {
                for (int i = lane; i < nc; i += 64) {
                  double q = Qv[i];
                  for (int j = 0; j < nc; ++j) q += Cm[i * nc + j] * dxv[j];
                  qn[i] = q;
                }
              }
              e2 = wave_sum(e2);
              if (lane == 0) s_ctl[2] += 1;
              if (__ballot(bad)) { status = 2; stop = 1; }
              else if (a.mode == MODE_TRAN) {
                const double dn = sqrt(e2 / nc);
                if (it == 0) { if (dn <= a.newton_tol || (rate_prev < 0.9 && 2.0 * fmax(rate_prev, 0.02) * dn <= a.newton_tol)) { status = 0; stop = 1; } }
                else { rate_new = dn_prev > 0.0 ? dn / dn_prev : 0.0; if (dn <= a.newton_tol) { status = 0; stop = 1; } }
                dn_prev = dn;
              }
            }
          }
        }
        if (lane == 0) { s_ctl[0] = stop; s_ctl[1] = status; }
      }
      __syncthreads();
      CH_STAMP(3);
      if (s_ctl[0]) break;
    }
    // ---- epilogue ----
    double* Xc = a.X + (long)a.cand_slot * a.slot_stride + sofs;
    double* Qc = a.Qh + (long)a.cand_slot * a.slot_stride + sofs;
    if (wave == 0) {
      double e2k = 0.0, e2m = 0.0, e2p = 0.0; int nd = 0;
      for (int i = lane; i < nc; i += 64) {
        const double xn = xl[i];
        Xc[i] = xn;
        Qc[i] = qn[i];
        if (a.obs_row) { const int ob = (dml[i] >> 8) - 1; if (ob >= 0) a.obs_row[(long)ob * a.S + s] = xn; }
        if (a.mode == MODE_TRAN && (dml[i] & 1)) {
          const double x0 = x0l[i];
          const double w = 1.0 / (a.reltol * fmax(fabs(x0), fabs(xn)) + a.abstol);
          ++nd;
          double t = (xn - xp[i]) * w; e2k += t * t;
          if (a.nkm1 > 0) { t = (xn - pm[i]) * w; e2m += t * t; }   // predictors of order k-1 / k+1 were formed in the prologue,
          if (a.nkp1 > 0) { t = (xn - pp[i]) * w; e2p += t * t; }   // from the history values already in registers there
        }
      }
      e2k = wave_sum(e2k); e2m = wave_sum(e2m); e2p = wave_sum(e2p);
      nd = (int)wave_sum((double)nd);
      if (lane == 0) {
        BlockOut bo; bo.status = s_ctl[1]; bo.iters = s_ctl[2]; bo.ndiff = nd; bo.pad = 0; bo.e2k = e2k; bo.e2km1 = e2m; bo.e2kp1 = e2p; bo.fnorm = s_fnorm;
        a.out[blk] = bo;
        if (a.mode == MODE_TRAN && s_ctl[1] == 0) a.rate[blk] = s_ctl[2] >= 2 ? fmin(1.0, fmax(rate_new, 1e-4)) : fmin(1.0, rate_prev * 1.5);
      }
      if (keep_order && a.perm && lane < nc && nc <= NC && myrow != myrow_in) a.perm[(long)blk * 16 + lane] = (unsigned char)(myrow ^ lane);
    }
  }
  CH_STAMP(4);
  CH_STAMP(5);
#ifdef CH_STAMPS
  if (tid == 0 && a.stamps) for (int i = 0; i < 8; ++i) atomicAdd(&a.stamps[i], acc_[i]);
#endif
}

// Reduce per-block outputs: WRMS per sample (over all blocks of the sample), max over samples.
// Many samples: one thread per sample loops over the sample's blocks.  Few samples (a single large
// circuit): the threads split the blocks of one sample and tree-reduce in LDS.
// Used when there are many blocks; for up to a few thousand blocks the Newton kernel writes its
// 48-byte BlockOut records straight into mapped pinned host memory and the HOST reduces them — no
// second launch, no inter-block hand-off (an in-kernel last-block reduction was measured at
// 8-12 us per launch for its agent-scope fences; profiles/r01_notes.md).
struct RedAcc {
  double a, b, c, f; long long nd, bits; int nfail, nsing, mx;
};
__device__ __forceinline__ void red_add(RedAcc& x, const BlockOut& o) {
  x.a += o.e2k; x.b += o.e2km1; x.c += o.e2kp1; x.nd += o.ndiff; x.bits += o.iters;
  if (o.status != 0) ++x.nfail;
  if (o.status == 2) ++x.nsing;
  x.mx = max(x.mx, o.iters); x.f = fmax(x.f, o.fnorm);
}
__device__ inline void reduce_all_blocks(const NewtonArgs& a, int t, int T, double* red) {
  // LDS scratch carved from the kernel's dynamic LDS (the Newton phase is over): 9 arrays of T doubles
  double* sk = red; double* sm = sk + T; double* sp = sm + T; double* sf = sp + T;
  double* snd = sf + T; double* sbi = snd + T; double* sfail = sbi + T; double* ssing = sfail + T; double* smax = ssing + T;
  const BlockOut* out = a.out; const unsigned char* active = a.active;
  const int n_comp = a.n_comp, S = a.S;
  double mk_ = 0.0, mm = 0.0, mp = 0.0, mf = 0.0; double nfail = 0, mx = 0, nsing = 0, its = 0, bits = 0;
  __syncthreads();
  if (S >= 64) {
    for (int s = t; s < S; s += T) {
      RedAcc x{0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = 0; k < n_comp; ++k) { const int blk = k * S + s; if (active && !active[blk]) continue; red_add(x, out[blk]); }
      its += x.mx; mx = fmax(mx, (double)x.mx); nfail += x.nfail; nsing += x.nsing; bits += (double)x.bits; mf = fmax(mf, x.f);
      if (x.nd > 0) { mk_ = fmax(mk_, a.ck * sqrt(x.a / x.nd)); mm = fmax(mm, a.ckm1 * sqrt(x.b / x.nd)); mp = fmax(mp, a.ckp1 * sqrt(x.c / x.nd)); }
    }
  } else {
    for (int s = 0; s < S; ++s) {
      RedAcc x{0, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int k = t; k < n_comp; k += T) { const int blk = k * S + s; if (active && !active[blk]) continue; red_add(x, out[blk]); }
      sk[t] = x.a; sm[t] = x.b; sp[t] = x.c; sf[t] = x.f; snd[t] = (double)x.nd; sbi[t] = (double)x.bits; sfail[t] = x.nfail; ssing[t] = x.nsing; smax[t] = x.mx;
      __syncthreads();
      for (int o = T >> 1; o > 0; o >>= 1) {
        if (t < o) {
          sk[t] += sk[t + o]; sm[t] += sm[t + o]; sp[t] += sp[t + o]; sf[t] = fmax(sf[t], sf[t + o]); snd[t] += snd[t + o]; sbi[t] += sbi[t + o];
          sfail[t] += sfail[t + o]; ssing[t] += ssing[t + o]; smax[t] = fmax(smax[t], smax[t + o]);
        }
        __syncthreads();
      }
      if (t == 0) {
        its += smax[0]; mx = fmax(mx, smax[0]); nfail += sfail[0]; nsing += ssing[0]; bits += sbi[0]; mf = fmax(mf, sf[0]);
        if (snd[0] > 0) { mk_ = fmax(mk_, a.ck * sqrt(sk[0] / snd[0])); mm = fmax(mm, a.ckm1 * sqrt(sm[0] / snd[0])); mp = fmax(mp, a.ckp1 * sqrt(sp[0] / snd[0])); }
      }
      __syncthreads();
    }
  }
  sk[t] = mk_; sm[t] = mm; sp[t] = mp; sf[t] = mf; sfail[t] = nfail; smax[t] = mx; ssing[t] = nsing; snd[t] = its; sbi[t] = bits;
  __syncthreads();
  for (int o = T >> 1; o > 0; o >>= 1) {
    if (t < o) {
      sk[t] = fmax(sk[t], sk[t + o]); sm[t] = fmax(sm[t], sm[t + o]); sp[t] = fmax(sp[t], sp[t + o]); sf[t] = fmax(sf[t], sf[t + o]);
      sfail[t] += sfail[t + o]; smax[t] = fmax(smax[t], smax[t + o]); ssing[t] += ssing[t + o]; snd[t] += snd[t + o]; sbi[t] += sbi[t + o];
    }
    __syncthreads();
  }
  if (t == 0) {
    Summary r; r.n_fail = (int)sfail[0]; r.max_iters = (int)smax[0]; r.n_singular = (int)ssing[0]; r.pad = 0;
    r.sum_iters = (long long)snd[0]; r.sum_block_iters = (long long)sbi[0];
    r.errk = sk[0]; r.errkm1 = sm[0]; r.errkp1 = sp[0]; r.fnorm = sf[0];
    *a.summary = r;
  }
}

__global__ __launch_bounds__(256) void reduce_blocks_kernel(const NewtonArgs a) {
  extern __shared__ double rlds[];
  reduce_all_blocks(a, threadIdx.x, blockDim.x, rlds);
}

// Save observables: out[o*S + s] = Σ_j w[j] X[slot_j][s][idx_o]  (idx < 0 → NaN placeholder, filled on host)
// saved rows [time][observable][sample] -> the result's layout [observable][time][sample] (writes coalesced)
__global__ void transpose_rows_kernel(const double* in, double* out, long nt, long ncol, int S) {
  const long n = nt * ncol * S;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
    const long s = o % S, rc = o / S, r = rc % nt, c = rc / nt;
    out[o] = in[(r * ncol + c) * S + s];
  }
}

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
  const B4Col P = b4_col(mosp, (long)cls[i] * Smos + (Smos > 1 ? sample : 0));
  double o[40];
  b4_device(P, v[4 * i + 0], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3], gmin, o);
  for (int j = 0; j < 40; ++j) out[(long)i * 40 + j] = o[j];
}

// Same evaluation with four lanes per instance (the path the Newton kernel uses); writes the
// 40-slot records straight to `out` (multiplier 1).
__global__ __launch_bounds__(64) void mos_eval_quad_kernel(const double* mosp, long mos_cols, const int* cls, int Smos, int sample, int n_mos, const double* v, double gmin, double* out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = q >> 2, sub = q & 3;
  if (i >= n_mos) return;
  const B4Col P = b4_col(mosp, (long)cls[i] * Smos + (Smos > 1 ? sample : 0));
  b4_device_quad(P, v[4 * i + 0], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3], gmin, sub, 1.0, out + (long)i * 40);
}


// ---------------------------------------------------------------------------------------------
// Small-signal frequency sweep: (G + jωC) x = b per (block, sample, frequency) — ac! / freqresp
// (src/ac.jl:75-102, 267-284) and, transposed with a unit right-hand side, the adjoint solve behind
// noise! / PSD (src/ac.jl:136-163, 286-305).  G, C and b come from MODE_EVAL dumps of the Newton kernel
// at the DC operating point, so the device arithmetic is the one the transient uses.
// One wavefront per (frequency, block): complex dense LU with partial pivoting in LDS; one 256-thread workgroup with a global
// workspace for coupled systems that do not fit LDS.
struct AcArgs {
  const BlockMeta* bmeta;
  const double* G; const double* C; const double* b;  // dumps: [nblk][ds][ds], [nblk][ds]
  int ds, S, n_unk, n_freq, n_comp;
  const double* omega;                                // [n_freq] rad/s
  double* x_out;                                      // AC: [S][n_freq][n_unk][2]
  // noise (adjoint) mode
  int noise, comp_out, row_out;                       // block and block-local row of the output unknown
  int n_noise;                                        // entries of the noise table per sample (unused entries: pwr = 0)
  const int* noise_a; const int* noise_b;             // [S][n_noise] block-local unknown of each terminal or -1 (known node)
  const double* noise_pwr; const double* noise_exp;   // [S][n_noise] power at the operating point, flicker exponent (0 = white)
  double* psd_out;                                    // [S][n_freq]
  int* fail;                                          // set to 1 when a factorisation breaks down
  double* work; int f0;                               // global workspace [systems of this launch][2*nc*(nc+1)] (T = 256 variant); first frequency of this launch
};

// T = 64: one wavefront, matrices in LDS (blocks of the fused path, coupled systems up to 96 unknowns).
// T = 256: one workgroup, matrices in a global workspace (a.work; larger coupled systems of the sparse path) — same algorithm,
// workgroup barriers instead of wave-local fences.
template <int T>
__global__ __launch_bounds__(T) void ac_block_kernel(const AcArgs a) {
  extern __shared__ double lds[];
  constexpr bool MULTI = T > 64;
  __shared__ double s_best[4]; __shared__ int s_bi[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, f = a.f0 + blockIdx.x;
  const int blk = a.noise ? a.comp_out * a.S + (int)blockIdx.y : (int)blockIdx.y;
  const int c = blk / a.S, s = blk - c * a.S;
  const BlockMeta bm = a.bmeta[c];
  const int nc = bm.cm.nc, lda = nc + 1;
  double* Ar = a.work ? a.work + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * (size_t)nc * lda : lds;
  double* Ai = Ar + (size_t)nc * lda;
  auto fence = [&]() { if (MULTI) __syncthreads(); else wave_fence(); };
  const double w = a.omega[f];
  const double* G = a.G + (long)blk * a.ds * a.ds; const double* Cg = a.C + (long)blk * a.ds * a.ds;
  for (int e = tid; e < nc * nc; e += T) {
    const int r = e / nc, col = e - r * nc;
    const int src = a.noise ? col * nc + r : e;  // adjoint: transpose
    Ar[r * lda + col] = G[src]; Ai[r * lda + col] = w * Cg[src];
  }
  for (int i = tid; i < nc; i += T) { Ar[i * lda + nc] = a.noise ? (i == a.row_out ? 1.0 : 0.0) : a.b[(long)blk * a.ds + i]; Ai[i * lda + nc] = 0.0; }
  fence();
  bool ok = true;
  for (int k = 0; k < nc && ok; ++k) {
    double best = -1.0; int bi = k;
    for (int i = k + tid; i < nc; i += T) { const double v = fabs(Ar[i * lda + k]) + fabs(Ai[i * lda + k]); if (v > best) { best = v; bi = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (MULTI) {
      if (lane == 0) { s_best[wv] = best; s_bi[wv] = bi; }
      __syncthreads();
      best = s_best[0]; bi = s_bi[0];
#pragma unroll
      for (int q = 1; q < T / 64; ++q) { const double ob = s_best[q]; const int oi = s_bi[q]; if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; } }
      __syncthreads();   // s_best / s_bi are rewritten in the next step
    }
    if (!(best > 0.0) || !(best < 1e300)) { ok = false; break; }
    if (bi != k) {
      for (int j = tid; j <= nc; j += T) {
        double t = Ar[k * lda + j]; Ar[k * lda + j] = Ar[bi * lda + j]; Ar[bi * lda + j] = t;
        t = Ai[k * lda + j]; Ai[k * lda + j] = Ai[bi * lda + j]; Ai[bi * lda + j] = t;
      }
      fence();
    }
    const double pr = Ar[k * lda + k], pi = Ai[k * lda + k];
    const double den = 1.0 / (pr * pr + pi * pi), ir = pr * den, ii = -pi * den;  // 1/pivot
    for (int i = k + 1 + tid; i < nc; i += T) {
      const double lr = Ar[i * lda + k], li = Ai[i * lda + k];
      Ar[i * lda + k] = lr * ir - li * ii; Ai[i * lda + k] = lr * ii + li * ir;
    }
    fence();
    const int rem = nc - k - 1, wd = rem + 1;
    for (int e = tid; e < rem * wd; e += T) {
      const int i = k + 1 + e / wd, j = k + 1 + e % wd;
      const double lr = Ar[i * lda + k], li = Ai[i * lda + k], ur = Ar[k * lda + j], ui = Ai[k * lda + j];
      Ar[i * lda + j] -= lr * ur - li * ui; Ai[i * lda + j] -= lr * ui + li * ur;
    }
    fence();
  }
  if (ok) {
    for (int k = nc - 1; k >= 0; --k) {
      const double pr = Ar[k * lda + k], pi = Ai[k * lda + k], br = Ar[k * lda + nc], bi_ = Ai[k * lda + nc];
      const double den = 1.0 / (pr * pr + pi * pi);
      const double xr = (br * pr + bi_ * pi) * den, xi = (bi_ * pr - br * pi) * den;
      fence();
      if (tid == 0) { Ar[k * lda + nc] = xr; Ai[k * lda + nc] = xi; }
      for (int i = tid; i < k; i += T) {
        const double ur = Ar[i * lda + k], ui = Ai[i * lda + k];
        Ar[i * lda + nc] -= ur * xr - ui * xi; Ai[i * lda + nc] -= ur * xi + ui * xr;
      }
      fence();
    }
  } else if (tid == 0) *a.fail = 1;
  if (!a.noise) {
    double* xo = a.x_out + (((long)s * a.n_freq + f) * a.n_unk + bm.uofs) * 2;
    for (int i = tid; i < nc; i += T) { xo[2 * i] = ok ? Ar[i * lda + nc] : CH_NAN; xo[2 * i + 1] = ok ? Ai[i * lda + nc] : CH_NAN; }
  } else {
    // output PSD = Σ_k |y_a - y_b|² · pwr_k / f^exp_k  (PSD(dss, ωs, pwr, exp), src/ac.jl:286-298)
    const double fhz = w * 0.15915494309189535;
    double acc = 0.0;
    for (int k = tid; k < a.n_noise; k += T) {
      const long e = (long)s * a.n_noise + k;
      const double pw = a.noise_pwr[e];
      if (pw == 0.0) continue;
      const int na = a.noise_a[e], nb = a.noise_b[e];
      const double yr = (na >= 0 ? Ar[na * lda + nc] : 0.0) - (nb >= 0 ? Ar[nb * lda + nc] : 0.0);
      const double yi = (na >= 0 ? Ai[na * lda + nc] : 0.0) - (nb >= 0 ? Ai[nb * lda + nc] : 0.0);
      const double ex = a.noise_exp[e];
      acc += (yr * yr + yi * yi) * (ex == 0.0 ? pw : pw / pow(fhz, ex));
    }
    acc = wave_sum(acc);
    if (MULTI) {
      __syncthreads();
      if (lane == 0) s_best[wv] = acc;
      __syncthreads();
      acc = 0.0;
#pragma unroll
      for (int q = 0; q < T / 64; ++q) acc += s_best[q];
    }
    if (tid == 0) a.psd_out[(long)s * a.n_freq + f] = ok ? acc : CH_NAN;
  }
}

// Noise sources of the output block at the DC operating point (slot `x_slot` of the state ring): resistors
// (white_noise(dscope, 4kT/res, :thermal), src/simpledevices.jl:72-76) and the white/flicker sources of compiled
// Verilog-A modules.  One thread per (device of the block, sample); every device owns va::MAX_NOISE table entries.
struct NoiseTabArgs {
  const int* dkind; const int* dterm; const int* dsrc; const int* dcls_local; const int* dhdev;
  const double* dpar; const double* dmult; const double* vapar; long va_stride; const double* temp_s; const double* gmin_s;
  const double* X; const double* kv;   // state (slot already applied) [S][n_unk]; known-node values [Ssrc][nk]
  int Spar, Stemp, Sgmin, Ssrc, nk, S, n_unk, dofs, ndev, uofs, nc;
  int* na; int* nb; double* pwr; double* ex;
};
__global__ void noise_table_kernel(const NoiseTabArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.ndev * a.S) return;
  const int dl = i % a.ndev, s = i / a.ndev, d = a.dofs + dl;
  const long base = ((long)s * a.ndev + dl) * va::MAX_NOISE;
  for (int k = 0; k < va::MAX_NOISE; ++k) { a.na[base + k] = -1; a.nb[base + k] = -1; a.pwr[base + k] = 0.0; a.ex[base + k] = 0.0; }
  const int kind = a.dkind[d];
  const int* tm = a.dterm + NTERM * d;
  auto loc = [&](int t) { return (t >= a.uofs && t < a.uofs + a.nc) ? t - a.uofs : -1; };
  const long pi = (long)a.dhdev[d] * a.Spar + (a.Spar > 1 ? s : 0);
  const double T = a.temp_s[a.Stemp > 1 ? s : 0] + 273.15;
  if (kind == K_R) {
    a.na[base] = loc(tm[0]); a.nb[base] = loc(tm[1]);
    a.pwr[base] = 4.0 * 1.380649e-23 * T * a.dmult[pi] / a.dpar[pi];
  } else if (kind == K_VA) {
    double vv[NTERM];
    for (int k = 0; k < NTERM; ++k) { const int t = tm[k]; vv[k] = t >= 0 ? a.X[(long)s * a.n_unk + t] : a.kv[(long)(a.Ssrc > 1 ? s : 0) * a.nk + (-t - 1)]; }
    va::NoiseRec rec[va::MAX_NOISE];
    const va::Env env{T, a.gmin_s[a.Sgmin > 1 ? s : 0]};
    const int n = va_gen::noise(a.dcls_local[d], a.vapar + (long)s * a.va_stride + a.dsrc[d], vv, env, rec);
    for (int k = 0; k < n && k < va::MAX_NOISE; ++k) {
      a.na[base + k] = rec[k].a >= 0 ? loc(tm[rec[k].a]) : -1;
      a.nb[base + k] = rec[k].b >= 0 ? loc(tm[rec[k].b]) : -1;
      a.pwr[base + k] = a.dmult[pi] * rec[k].pwr; a.ex[base + k] = rec[k].ex;
    }
  }
}

// one compiled Verilog-A module at given node voltages (ch_va_eval: stamp-level parity entry point)
// Per-instance constants of the compiled Verilog-A devices: one thread per (instance, sample).  Runs whenever parameters or
// the temperature change (ch_circuit::finalize_params), never inside the Newton loop.
__global__ void va_setup_kernel(int n_va, int Svac, const int* vmod, const int* vpofs, const int* vcofs, const double* vapar, long va_stride,
                                const double* temp_s, int Stemp, double* cache, long vac_stride) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n_va * Svac) return;
  const int d = (int)(i % n_va), s = (int)(i / n_va);
  const va::Env env{temp_s[Stemp > 1 ? s : 0] + 273.15, 0.0};   // $simparam("gmin") is not a setup-time quantity (gmin stepping)
  va_gen::setup(vmod[d], vapar + (long)s * va_stride + vpofs[d], env, cache + (long)s * vac_stride + vcofs[d]);
}

__global__ void va_eval_kernel(int mod, const double* P, const double* v, double temp_k, double gmin, double* st) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double vv[NTERM], out[144];
  for (int k = 0; k < NTERM; ++k) vv[k] = v[k];
  for (int k = 0; k < 144; ++k) out[k] = 0.0;
  const va::Env env{temp_k, gmin};
  va_gen::stamp(mod, P, vv, env, 1.0, out);
  for (int k = 0; k < 144; ++k) st[k] = out[k];
}

__global__ void va_opvars_kernel(int mod, const double* P, const double* v, double temp_k, double gmin, double* op) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double vv[NTERM];
  for (int k = 0; k < NTERM; ++k) vv[k] = v[k];
  const va::Env env{temp_k, gmin};
  va_gen::opvars(mod, P, vv, env, op);
}

// sparse path -> dense small-signal blocks: CSR values of G and C (one system per sample) into [S][n][n], F into [S][n]
__global__ void csr_to_dense_kernel(const int* rowptr, const int* colidx, const double* Aval, const double* Cval, const double* F,
                                    int n, int nnz, int S, double* G, double* C, double* Fd, int fill_gc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, sm = blockIdx.y;
  if (i >= n || sm >= S) return;
  if (fill_gc) {
    double* g = G + ((long)sm * n + i) * n; double* c = C + ((long)sm * n + i) * n;
    for (int j = 0; j < n; ++j) { g[j] = 0.0; c[j] = 0.0; }
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) { g[colidx[p]] = Aval[(long)sm * nnz + p]; c[colidx[p]] = Cval[(long)sm * nnz + p]; }
  }
  Fd[(long)sm * n + i] = F[(long)sm * n + i];
}

// y = x - y (AC right-hand side b = F(src) - F(src + ac))
__global__ void axpby_kernel(double* y, const double* x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] - y[i];
}

// fp64 FMA peak (measurement utility, ch_bench_fp64): 16 independent chains per lane
__global__ __launch_bounds__(256) void fp64_peak_kernel(double* out, int n_outer, double a, double b) {
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  for (int it = 0; it < n_outer; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  if (s == 123.456) out[blockIdx.x * blockDim.x + threadIdx.x] = s;  // never true: keeps the chains alive
}

// STREAM triad (measurement utility, ch_bench_triad): grid-stride, 16-byte accesses
__global__ __launch_bounds__(256) void triad_kernel(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, double s, long n2) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {  // 8 loads in flight per lane before the first use
    double2 x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { x[u] = b[i + u * stride]; y[u] = c[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) a[i + u * stride] = make_double2(x[u].x + s * y[u].x, x[u].y + s * y[u].y);
  }
  for (; i < n2; i += stride) { const double2 x = b[i], y = c[i]; a[i] = make_double2(x.x + s * y.x, x.y + s * y.y); }
}

}  // namespace chip
