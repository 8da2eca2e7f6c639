// va_rt.hpp — runtime of the generated Verilog-A device functions (cedarsim.jl_amd/va/codegen.py).
// Self-contained (host and gfx950 device): forward-mode dual numbers generic over their scalar type, so a
// module that uses ddx() is instantiated over nested duals (outer = Jacobian directions, one per device
// node; inner = the ddx probe nodes) and gets exact second derivatives — what the reference obtains from
// ForwardDiff.Dual{SimTag} inside DAECompiler's own derivative pass (src/vasim.jl:347-357, 392-412).
// Math follows src/va_env.jl:35-47 (NaNMath: ln/sqrt/pow return NaN outside their domain).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VA_HD __host__ __device__ __forceinline__
#define VA_HD_NOINLINE __host__ __device__
#else
#define VA_HD inline
#define VA_HD_NOINLINE inline
#endif

// Work-around for the gfx950 code generator of ROCm 7.2: a large LEAF device function gets its out-of-range branches
// expanded (s_getpc_b64 / s_add / s_setpc_b64) through s[30:31] — the register pair that holds the function's return address —
// without a save, and then returns to the last branch target (memory fault on address nil; seen on bsimcmg's eval once the
// library calls had gone from it).  Declaring the pair clobbered at the top of the function makes the prologue keep the return
// address in a VGPR lane, as it does for every function that uses the pair itself.  scripts/check_return_address.py scans the
// device assembly for functions that still have the pattern.
#if defined(__HIP_DEVICE_COMPILE__)
#define VA_KEEP_RETURN_ADDRESS asm volatile("" ::: "s30", "s31")
#else
#define VA_KEEP_RETURN_ADDRESS ((void)0)
#endif

namespace va {

// pointer to a parameter / constant block that a kernel has copied into LDS (generated eval<R, PART, lds_cptr>)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(3))) double* lds_cptr;
#else
typedef const double* lds_cptr;
#endif

struct Env {
  double temperature;  // kelvin: $temperature (src/va_env.jl:123)
  double gmin;         // $simparam("gmin")
};

constexpr int MAX_NOISE = 16;  // noise sources one module instance may report
struct NoiseRec {
  int a, b;        // module node indices of the branch (b = -1: to ground)
  double pwr, ex;  // white: pwr [A²/Hz]; flicker: pwr / f^ex
};

template <int N, class S>
struct VD {
  S v;
  S d[N];
  VA_HD VD() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
  VA_HD VD(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
  VA_HD VD(int c) : v((double)c) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
  VA_HD VD(const S& c, int) : v(c) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
};

// ---- fp64 primitives ----
// Division, square root, logarithm and pow are a third of the instructions of a compiled compact model when left to the
// IEEE / library sequences (11, 15, 118 and ~300 VALU instructions on gfx950).  On the device they are built from the hardware
// seeds (v_rcp_f64, v_rsq_f64, frexp) plus Newton / polynomial steps: 1-2 ulp, far inside the 1e-6 / 1e-4 parity bars
// (SURVEY 8).  The host instantiation (oracle, tests against the interpreter) keeps the library functions.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VA_NO_FAST_MATH)
VA_HD double rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r, 1.0);
  const double y = __builtin_fma(r, __builtin_fma(e, e, e), r);
  return (r != 0.0 && __builtin_fabs(r) < __builtin_inf()) ? y : r;   // 1/0, 1/inf: the seed is already the answer (e is NaN there)
}
VA_HD double sqrt_pos(double x) {   // x > 0, finite
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  return __builtin_fma(__builtin_fma(-g, g, x), h, g);
}
// ln x for x > 0 finite: x = m 2^e, m in [sqrt(1/2), sqrt(2)); s = (m-1)/(m+1); ln m = 2s(1 + s^2/3 + ... + s^18/19), |s| <= 0.1716
VA_HD double ln_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? 2.0 * m : m;
  e = lo ? e - 1 : e;
  const double s = (m - 1.0) * rcp(m + 1.0);
  const double z = s * s;
  double p = 1.0 / 19.0;
  p = __builtin_fma(p, z, 1.0 / 17.0); p = __builtin_fma(p, z, 1.0 / 15.0); p = __builtin_fma(p, z, 1.0 / 13.0); p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0); p = __builtin_fma(p, z, 1.0 / 7.0); p = __builtin_fma(p, z, 1.0 / 5.0); p = __builtin_fma(p, z, 1.0 / 3.0);
  const double lm = __builtin_fma(2.0 * s, p * z, 2.0 * s);
  const double de = (double)e;
  return __builtin_fma(de, 6.93147180369123816490e-01, __builtin_fma(de, 1.90821492927058770002e-10, lm));
}
VA_HD double v_div(double a, double b) { return a * rcp(b); }
#else
VA_HD double rcp(double x) { return 1.0 / x; }
VA_HD double v_div(double a, double b) { return a / b; }
#endif

// plain value at the bottom of any nesting
VA_HD double val(double x) { return x; }
VA_HD double val(int x) { return (double)x; }
template <int N, class S> VA_HD double val(const VD<N, S>& x) { return val(x.v); }

template <int N, class S> VA_HD VD<N, S> operator+(const VD<N, S>& a, const VD<N, S>& b) { VD<N, S> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N, class S> VA_HD VD<N, S> operator-(const VD<N, S>& a, const VD<N, S>& b) { VD<N, S> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N, class S> VA_HD VD<N, S> operator-(const VD<N, S>& a) { VD<N, S> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N, class S> VA_HD VD<N, S> operator*(const VD<N, S>& a, const VD<N, S>& b) { VD<N, S> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N, class S> VA_HD VD<N, S> operator/(const VD<N, S>& a, const VD<N, S>& b) {
  VD<N, S> r; const S ib = rcp(b.v); r.v = a.v * ib;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
#define VA_MIXED(OP) \
  template <int N, class S> VA_HD VD<N, S> operator OP(const VD<N, S>& a, double b) { return a OP VD<N, S>(b); } \
  template <int N, class S> VA_HD VD<N, S> operator OP(double a, const VD<N, S>& b) { return VD<N, S>(a) OP b; } \
  template <int N, class S> VA_HD VD<N, S> operator OP(const VD<N, S>& a, int b) { return a OP VD<N, S>((double)b); } \
  template <int N, class S> VA_HD VD<N, S> operator OP(int a, const VD<N, S>& b) { return VD<N, S>((double)a) OP b; }
VA_MIXED(+) VA_MIXED(-) VA_MIXED(*) VA_MIXED(/)
#undef VA_MIXED
template <int N, class S> VA_HD VD<N, S> rcp(const VD<N, S>& b) { VD<N, S> r; r.v = rcp(b.v); const S m = -(r.v * r.v); for (int i = 0; i < N; ++i) r.d[i] = m * b.d[i]; return r; }
// the code generator writes every Verilog-A `/` as v_div
template <int N, class S> VA_HD VD<N, S> v_div(const VD<N, S>& a, const VD<N, S>& b) { return a / b; }
template <int N, class S> VA_HD VD<N, S> v_div(const VD<N, S>& a, double b) { return a * rcp(b); }
template <int N, class S> VA_HD VD<N, S> v_div(double a, const VD<N, S>& b) { return rcp(b) * a; }
template <int N, class S> VA_HD VD<N, S>& operator+=(VD<N, S>& a, const VD<N, S>& b) { a = a + b; return a; }
template <int N, class S> VA_HD VD<N, S>& operator-=(VD<N, S>& a, const VD<N, S>& b) { a = a - b; return a; }

// f(x) with derivative g = f'(x.v): chain rule, generic over the nesting
template <int N, class S> VA_HD VD<N, S> chain(const VD<N, S>& x, const S& f, const S& g) { VD<N, S> r; r.v = f; for (int i = 0; i < N; ++i) r.d[i] = g * x.d[i]; return r; }

VA_HD double v_exp(double x) { return ::exp(x); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VA_NO_FAST_MATH)
VA_HD double v_ln(double x) { return (x > 0.0 && x < INFINITY) ? ln_pos(x) : (x == 0.0 ? -INFINITY : (x > 0.0 ? x : NAN)); }
#else
VA_HD double v_ln(double x) { return x > 0.0 ? ::log(x) : (x == 0.0 ? -INFINITY : NAN); }
#endif
VA_HD double v_log10(double x) { return x > 0.0 ? ::log10(x) : (x == 0.0 ? -INFINITY : NAN); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VA_NO_FAST_MATH)
VA_HD double v_sqrt(double x) { return (x > 0.0 && x < INFINITY) ? sqrt_pos(x) : (x == 0.0 || x > 0.0 ? x : NAN); }
#else
VA_HD double v_sqrt(double x) { return x >= 0.0 ? ::sqrt(x) : NAN; }
#endif
VA_HD double v_sin(double x) { return ::sin(x); }
VA_HD double v_cos(double x) { return ::cos(x); }
VA_HD double v_tan(double x) { return ::tan(x); }
VA_HD double v_sinh(double x) { return ::sinh(x); }
VA_HD double v_cosh(double x) { return ::cosh(x); }
VA_HD double v_tanh(double x) { return ::tanh(x); }
VA_HD double v_atan(double x) { return ::atan(x); }
VA_HD double v_asin(double x) { return (x >= -1.0 && x <= 1.0) ? ::asin(x) : NAN; }
VA_HD double v_acos(double x) { return (x >= -1.0 && x <= 1.0) ? ::acos(x) : NAN; }
VA_HD double v_asinh(double x) { return ::asinh(x); }
VA_HD double v_acosh(double x) { return x >= 1.0 ? ::acosh(x) : NAN; }
VA_HD double v_atanh(double x) { return (x > -1.0 && x < 1.0) ? ::atanh(x) : NAN; }
VA_HD double v_abs(double x) { return ::fabs(x); }
VA_HD int v_abs(int x) { return x < 0 ? -x : x; }
VA_HD double v_floor(double x) { return ::floor(x); }
VA_HD double v_ceil(double x) { return ::ceil(x); }
VA_HD double v_limexp(double x) { return x < 80.0 ? ::exp(x) : ::exp(80.0) * (1.0 + (x - 80.0)); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VA_NO_FAST_MATH)
// a^b = exp(b ln a) for a > 0 (relative error ~ |b ln a| ulp); the sign of a negative base with an integer exponent is restored;
// everything else (zero / infinite base, NaN) goes to the library
VA_HD double v_pow(double a, double b) {
  const double fa = __builtin_fabs(a);
  if (fa > 0.0 && fa < INFINITY && __builtin_fabs(b) < 1e15) {
    if (a < 0.0 && b != ::floor(b)) return NAN;
    const double p = ::exp(b * ln_pos(fa));
    return (a < 0.0 && ::fmod(b, 2.0) != 0.0) ? -p : p;
  }
  return (a < 0.0 && b != ::floor(b)) ? NAN : ::pow(a, b);
}
#else
VA_HD double v_pow(double a, double b) { return (a < 0.0 && b != ::floor(b)) ? NAN : ::pow(a, b); }
#endif
VA_HD double v_min(double a, double b) { return a < b ? a : b; }
VA_HD double v_max(double a, double b) { return a > b ? a : b; }
VA_HD int v_min(int a, int b) { return a < b ? a : b; }
VA_HD int v_max(int a, int b) { return a > b ? a : b; }
VA_HD double v_atan2(double y, double x) { return ::atan2(y, x); }
VA_HD double v_hypot(double a, double b) { return v_sqrt(a * a + b * b); }

template <int N, class S> VA_HD VD<N, S> v_exp(const VD<N, S>& x) { const S e = v_exp(x.v); return chain(x, e, e); }
template <int N, class S> VA_HD VD<N, S> v_ln(const VD<N, S>& x) { return chain(x, v_ln(x.v), rcp(x.v)); }
template <int N, class S> VA_HD VD<N, S> v_log10(const VD<N, S>& x) { return chain(x, v_log10(x.v), S(0.43429448190325182765) * rcp(x.v)); }
template <int N, class S> VA_HD VD<N, S> v_sqrt(const VD<N, S>& x) { const S r = v_sqrt(x.v); return chain(x, r, S(0.5) * rcp(r)); }
template <int N, class S> VA_HD VD<N, S> v_sin(const VD<N, S>& x) { return chain(x, v_sin(x.v), v_cos(x.v)); }
template <int N, class S> VA_HD VD<N, S> v_cos(const VD<N, S>& x) { return chain(x, v_cos(x.v), -v_sin(x.v)); }
template <int N, class S> VA_HD VD<N, S> v_tan(const VD<N, S>& x) { const S t = v_tan(x.v); return chain(x, t, S(1.0) + t * t); }
template <int N, class S> VA_HD VD<N, S> v_sinh(const VD<N, S>& x) { return chain(x, v_sinh(x.v), v_cosh(x.v)); }
template <int N, class S> VA_HD VD<N, S> v_cosh(const VD<N, S>& x) { return chain(x, v_cosh(x.v), v_sinh(x.v)); }
template <int N, class S> VA_HD VD<N, S> v_tanh(const VD<N, S>& x) { const S t = v_tanh(x.v); return chain(x, t, S(1.0) - t * t); }
template <int N, class S> VA_HD VD<N, S> v_atan(const VD<N, S>& x) { return chain(x, v_atan(x.v), rcp(S(1.0) + x.v * x.v)); }
template <int N, class S> VA_HD VD<N, S> v_asin(const VD<N, S>& x) { return chain(x, v_asin(x.v), S(1.0) / v_sqrt(S(1.0) - x.v * x.v)); }
template <int N, class S> VA_HD VD<N, S> v_acos(const VD<N, S>& x) { return chain(x, v_acos(x.v), S(-1.0) / v_sqrt(S(1.0) - x.v * x.v)); }
template <int N, class S> VA_HD VD<N, S> v_asinh(const VD<N, S>& x) { return chain(x, v_asinh(x.v), S(1.0) / v_sqrt(x.v * x.v + S(1.0))); }
template <int N, class S> VA_HD VD<N, S> v_acosh(const VD<N, S>& x) { return chain(x, v_acosh(x.v), S(1.0) / v_sqrt(x.v * x.v - S(1.0))); }
template <int N, class S> VA_HD VD<N, S> v_atanh(const VD<N, S>& x) { return chain(x, v_atanh(x.v), S(1.0) / (S(1.0) - x.v * x.v)); }
template <int N, class S> VA_HD VD<N, S> v_abs(const VD<N, S>& x) { return val(x) < 0.0 ? -x : x; }
template <int N, class S> VA_HD double v_floor(const VD<N, S>& x) { return ::floor(val(x)); }
template <int N, class S> VA_HD double v_ceil(const VD<N, S>& x) { return ::ceil(val(x)); }
template <int N, class S> VA_HD VD<N, S> v_limexp(const VD<N, S>& x) { if (val(x) < 80.0) return v_exp(x); return (x - 79.0) * ::exp(80.0); }
template <int N, class S> VA_HD VD<N, S> v_min(const VD<N, S>& a, const VD<N, S>& b) { return val(a) < val(b) ? a : b; }
template <int N, class S> VA_HD VD<N, S> v_max(const VD<N, S>& a, const VD<N, S>& b) { return val(a) > val(b) ? a : b; }
template <int N, class S> VA_HD VD<N, S> v_hypot(const VD<N, S>& a, const VD<N, S>& b) { return v_sqrt(a * a + b * b); }
template <int N, class S> VA_HD VD<N, S> v_atan2(const VD<N, S>& y, const VD<N, S>& x) {
  VD<N, S> r; const S r2 = x.v * x.v + y.v * y.v; r.v = v_atan2(y.v, x.v);
  for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) / r2;
  return r;
}
// pow: constant exponent → b·a^(b-1) with a strong zero for the exponent's tangent (src/va_env.jl:60-70)
template <int N, class S> VA_HD VD<N, S> v_pow(const VD<N, S>& a, double b) {
  if (b == 0.0) return VD<N, S>(1.0);
  const S p = v_pow(a.v, b);
  if (val(a.v) != 0.0) return chain(a, p, S(b) * p * rcp(a.v));   // one pow instead of two
  return chain(a, p, S(b) * v_pow(a.v, b - 1.0));
}
template <int N, class S> VA_HD VD<N, S> v_pow(const VD<N, S>& a, int b) { return v_pow(a, (double)b); }
template <int N, class S> VA_HD VD<N, S> v_pow(const VD<N, S>& a, const VD<N, S>& b) {
  const double av = val(a), bv = val(b);
  if (av > 0.0) return v_exp(b * v_ln(a));
  if (av == 0.0 && bv > 0.0) return VD<N, S>(0.0);
  return VD<N, S>(NAN);
}
template <int N, class S> VA_HD VD<N, S> v_pow(double a, const VD<N, S>& b) { return v_pow(VD<N, S>(a), b); }

// array index: offset by the lower bound; out-of-range indices are clamped (the interpreter raises instead)
VA_HD int clamp_index(int i, int lo, int hi) { return (i < lo ? lo : (i > hi ? hi : i)) - lo; }

// VA real → integer: round half away from zero (LRM 4.2.1.1, src/va_env.jl:107)
VA_HD int to_int(double x) { return (int)(x >= 0.0 ? ::floor(x + 0.5) : -::floor(-x + 0.5)); }
VA_HD int to_int(int x) { return x; }
template <int N, class S> VA_HD int to_int(const VD<N, S>& x) { return to_int(val(x)); }
VA_HD bool truth(double x) { return x != 0.0; }
VA_HD bool truth(int x) { return x != 0; }
VA_HD bool truth(bool x) { return x; }
template <int N, class S> VA_HD bool truth(const VD<N, S>& x) { return val(x) != 0.0; }

// ddx(e, V(a)) / ddx(e, V(a,b)) on the nested type R = VD<NT, VD<ND,double>> (src/vasim.jl:392-412)
template <int NT, int ND> VA_HD VD<NT, VD<ND, double>> ddx1(const VD<NT, VD<ND, double>>& x, int ia) {
  VD<NT, VD<ND, double>> r; r.v = VD<ND, double>(x.v.d[ia]);
  for (int k = 0; k < NT; ++k) r.d[k] = VD<ND, double>(x.d[k].d[ia]);
  return r;
}
template <int NT, int ND> VA_HD VD<NT, VD<ND, double>> ddx2(const VD<NT, VD<ND, double>>& x, int ia, int ib) {
  VD<NT, VD<ND, double>> r; r.v = VD<ND, double>(0.5 * (x.v.d[ia] - x.v.d[ib]));
  for (int k = 0; k < NT; ++k) r.d[k] = VD<ND, double>(0.5 * (x.d[k].d[ia] - x.d[k].d[ib]));
  return r;
}
VA_HD double ddx1(double, int) { return 0.0; }
VA_HD double ddx2(double, int, int) { return 0.0; }

// seeds: node voltage k of NT as the outer variable; with ND > 0 also the inner variable `dk` (or -1)
template <int NT> VA_HD VD<NT, double> seed(double v, int k, int /*dk*/, VD<NT, double>*) { VD<NT, double> r(v); r.d[k] = 1.0; return r; }
template <int NT, int ND> VA_HD VD<NT, VD<ND, double>> seed(double v, int k, int dk, VD<NT, VD<ND, double>>*) {
  VD<NT, VD<ND, double>> r; r.v = VD<ND, double>(v); if (dk >= 0) r.v.d[dk] = 1.0;
  r.d[k] = VD<ND, double>(1.0);
  return r;
}

// one-directional seeds (direction-parallel evaluation)
VA_HD VD<1, double> seed1(double v, bool is_dir, int /*dk*/, VD<1, double>*) { VD<1, double> r(v); r.d[0] = is_dir ? 1.0 : 0.0; return r; }
template <int ND> VA_HD VD<1, VD<ND, double>> seed1(double v, bool is_dir, int dk, VD<1, VD<ND, double>>*) {
  VD<1, VD<ND, double>> r; r.v = VD<ND, double>(v); if (dk >= 0) r.v.d[dk] = 1.0;
  r.d[0] = VD<ND, double>(is_dir ? 1.0 : 0.0);
  return r;
}

}  // namespace va
