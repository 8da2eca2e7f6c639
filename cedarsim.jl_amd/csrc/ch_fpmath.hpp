// ch_fpmath.hpp — fp64 exp and ln for the device functions (BSIM4: ch_bsim4.hpp; compiled Verilog-A: va_rt.hpp).
//
// Why its own implementation: a compact model calls exp / ln a few hundred times per evaluation, each call is a polynomial with a
// dozen fp64 coefficients, and gfx9 encodes no 64-bit literal — every coefficient written as a literal costs two moves (a quarter of
// the static code of the compiled BSIM-CMG evaluation was v_mov_b32 / s_mov_b32 of literal halves, profiles/r03_notes.md section 5).
// Here the coefficients live in __constant__ tables: a call site loads them with one or two scalar loads (s_load_dwordx16 /
// s_load_dwordx8 through the scalar cache) straight into SGPR pairs that v_fma_f64 takes as operands.  A __constant__ variable is
// "externally initialised" for the compiler (the host may overwrite it), so the loads are never folded back into literals.
//
// Accuracy (tests/test_gpu_va.py::test_device_exp_and_ln_accuracy through ch_debug_math): exp and ln within 2 ulp of libm over
// their whole ranges; the special values (0, negative, NaN, +-inf, overflow, underflow into denormals) as libm returns them.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>

namespace chfp {

// exp: x = k ln2 + r, |r| <= ln2/2 (Cody-Waite in two steps; ln2_hi has 21 trailing zero bits, so k ln2_hi is exact),
// exp(r) = 1 + r + r^2 (c2 + c3 r + ... + c13 r^11), c_n = 1/n!: the truncation error is r^14/14! < 5e-18.
//   [0] log2(e)  [1] ln2_hi  [2] ln2_lo  [3..14] c13 .. c2  [15] overflow threshold  [16] underflow-to-zero threshold
__device__ __constant__ double EXP_TAB[20] = {
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0,
    1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5,
    709.782712893384, -745.1332191019412, 0.0, 0.0, 0.0};
// ln for positive finite x = m 2^e, m in [sqrt(1/2), sqrt(2)): s = (m-1)/(m+1), ln m = 2s (1 + s^2/3 + ... + s^18/19), |s| <= 0.1716
//   [0..8] 1/19 .. 1/3  [9] ln2_hi  [10] ln2_lo  [11] sqrt(1/2)
__device__ __constant__ double LN_TAB[12] = {
    1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0,
    6.93147180369123816490e-01, 1.90821492927058770002e-10, 0.70710678118654752440};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double exp_tab(double x) {
  const double* T = EXP_TAB;
  const double kf = __builtin_rint(x * T[0]);
  double r = __builtin_fma(-kf, T[1], x);
  r = __builtin_fma(-kf, T[2], r);
  double p = T[3];
#pragma unroll
  for (int i = 4; i <= 14; ++i) p = __builtin_fma(p, r, T[i]);
  const double e = __builtin_fma(r * r, p, r) + 1.0;
  double y = __builtin_ldexp(e, (int)kf);       // v_cvt_i32_f64 saturates and maps NaN to 0; v_ldexp_f64 rounds into the denormals
  y = x > T[15] ? __builtin_inf() : y;            // also +inf (r would be NaN there)
  y = x < T[16] ? 0.0 : y;                        // also -inf
  return y;                                       // NaN in, NaN out: both comparisons are false
}
// positive finite x only: callers handle x <= 0, NaN and +inf (va_rt.hpp v_ln, ch_bsim4.hpp flog)
__device__ __forceinline__ double ln_pos_tab(double x) {
  const double* T = LN_TAB;
  double m = __builtin_amdgcn_frexp_mant(x);     // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < T[11];
  m = lo ? 2.0 * m : m;
  e = lo ? e - 1 : e;
  const double q = m + 1.0;                      // in [1.70, 2.42): the hardware seed plus one third-order step, no special cases
  const double r0 = __builtin_amdgcn_rcp(q), e0 = __builtin_fma(-q, r0, 1.0);
  const double s = (m - 1.0) * __builtin_fma(r0, __builtin_fma(e0, e0, e0), r0);
  const double z = s * s;
  double p = T[0];
#pragma unroll
  for (int i = 1; i <= 8; ++i) p = __builtin_fma(p, z, T[i]);
  const double lm = __builtin_fma(2.0 * s, p * z, 2.0 * s);
  const double de = (double)e;
  return __builtin_fma(de, T[9], __builtin_fma(de, T[10], lm));
}
#endif

}  // namespace chfp
#endif
