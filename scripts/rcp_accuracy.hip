// Measures the raw accuracy of v_rcp_f64 / v_rsq_f64 on the GPU (relative error vs correctly rounded fp64),
// with 0, 1 and 2 Newton refinement steps.  Build: hipcc --offload-arch=gfx950 -O3 -o rcp_accuracy rcp_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* s0, double* s1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = fma(fma(-v, r, 1.0), r, r); r1[i] = r;
  r = fma(fma(-v, r, 1.0), r, r); r2[i] = r;
  double y = __builtin_amdgcn_rsq(v);
  s0[i] = y;
  // one Newton step on rsqrt: y = y*(1.5 - 0.5*v*y*y)
  y = y * fma(-0.5 * v * y, y, 1.5);
  s1[i] = y;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), r0(n), r1(n), r2(n), s0(n), s1(n);
  std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-20.0, 20.0);
  for (int i = 0; i < n; ++i) x[i] = std::exp2(u(g)) * (1.0 + (g() % 1000) / 1000.0);
  double *dx, *d0, *d1, *d2, *e0, *e1;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&e0, n * 8); hipMalloc(&e1, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, e0, e1, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(s0.data(), e0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(s1.data(), e1, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0, q0 = 0, q1 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i], ts = 1.0L / sqrtl((long double)x[i]);
    m0 = fmax(m0, (double)fabsl((r0[i] - t) / t)); m1 = fmax(m1, (double)fabsl((r1[i] - t) / t)); m2 = fmax(m2, (double)fabsl((r2[i] - t) / t));
    q0 = fmax(q0, (double)fabsl((s0[i] - ts) / ts)); q1 = fmax(q1, (double)fabsl((s1[i] - ts) / ts));
  }
  printf("rcp: raw %.3e  1 step %.3e  2 steps %.3e  (eps = %.3e)\n", m0, m1, m2, 1.11e-16);
  printf("rsq: raw %.3e  1 step %.3e\n", q0, q1);
  return 0;
}
