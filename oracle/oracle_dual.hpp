// oracle_dual.hpp — forward-mode dual numbers for the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY (see oracle/README.md): nothing in the product path may include this.
//
// The reference differentiates device code with ForwardDiff.Dual{SimTag} (src/vasim.jl:29-35,
// :347-357, src/simulate_ir.jl:142-156); this is the same idea in plain C++: a value plus N
// partial derivatives, propagated through + - * / sqrt exp log pow.
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  Dual(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  static Dual var(double x, int k) { Dual r(x); r.d[k] = 1.0; return r; }
};

template <int N> inline double val(const Dual<N>& a) { return a.v; }
inline double val(double a) { return a; }

template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
  Dual<N> r; double inv = 1.0 / b.v; r.v = a.v * inv;
  for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double a, const Dual<N>& b) { return b + a; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double a, const Dual<N>& b) { Dual<N> r = -b; r.v += a; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double a, const Dual<N>& b) { return b * a; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, const Dual<N>& b) { a = a - b; return a; }
template <int N> inline Dual<N>& operator*=(Dual<N>& a, const Dual<N>& b) { a = a * b; return a; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, double b) { a.v += b; return a; }
template <int N> inline Dual<N>& operator-=(Dual<N>& a, double b) { a.v -= b; return a; }
template <int N> inline Dual<N>& operator*=(Dual<N>& a, double b) { a = a * b; return a; }

template <int N> inline Dual<N> sqrt(const Dual<N>& a) {
  Dual<N> r; r.v = std::sqrt(a.v); double k = 0.5 / r.v;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k;
  return r;
}
template <int N> inline Dual<N> exp(const Dual<N>& a) {
  Dual<N> r; r.v = std::exp(a.v);
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v;
  return r;
}
template <int N> inline Dual<N> log(const Dual<N>& a) {
  Dual<N> r; r.v = std::log(a.v); double k = 1.0 / a.v;
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k;
  return r;
}
// a^p for constant exponent p
template <int N> inline Dual<N> pow(const Dual<N>& a, double p) {
  Dual<N> r; r.v = std::pow(a.v, p); double k = p * std::pow(a.v, p - 1.0);
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k;
  return r;
}
inline double sqrt(double a) { return std::sqrt(a); }
inline double exp(double a) { return std::exp(a); }
inline double log(double a) { return std::log(a); }
inline double pow(double a, double p) { return std::pow(a, p); }

}  // namespace oracle
