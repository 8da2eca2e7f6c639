// LU probe: the register LU with a pivot search (lu_solve_regs) against the pivot-order-reusing form (lu_solve_block) of
// ch_kernels.hpp, on random 11x11 systems (diagonally dominant, general, and with a permuted diagonal so that the identity order
// fails and the fallback runs): max error against a host LU with partial pivoting, and shader cycles per solve.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icedarsim.jl_amd/csrc -Iinclude scripts/lu_probe.hip -o scripts/_bin/lu_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "ch_kernels.hpp"
using namespace chip;

template <int NC, int WHICH>
__global__ __launch_bounds__(64) void lu_k(const double* Ain, double* X, long long* cyc, int* order, int nc, int reps) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x, lda = nc + 1;
  for (int i = lane; i < nc * lda; i += 64) lds[i] = Ain[(size_t)blockIdx.x * 16 * 17 + (i / lda) * 17 + (i % lda)];
  __syncthreads();
  double sol = 0.0; long long t = 0; int myrow = lane; bool ok = true;
  for (int rep = 0; rep < reps; ++rep) {
    const long long t0 = __builtin_readcyclecounter();
    if (WHICH == 0) {
      double r[NC + 1];
#pragma unroll
      for (int j = 0; j <= NC; ++j) r[j] = (lane < nc && j <= nc) ? lds[lane * lda + j] : 0.0;
      ok = lu_solve_regs<NC>(r, nc, lane, sol);
    } else ok = lu_solve_block<NC>(lds, lda, nc, lane, myrow, sol);
    const long long t1 = __builtin_readcyclecounter();
    if (rep > 0) t += t1 - t0;
  }
  if (lane < nc) { X[blockIdx.x * 16 + lane] = ok ? sol : NAN; order[blockIdx.x * 16 + lane] = myrow; }
  if (lane == 0) cyc[blockIdx.x] = t / (reps - 1);
}

static void host_solve(const double* A17, int nc, double* x) {
  std::vector<double> a(nc * (nc + 1));
  for (int i = 0; i < nc; ++i) for (int j = 0; j <= nc; ++j) a[i * (nc + 1) + j] = A17[i * 17 + j];
  for (int k = 0; k < nc; ++k) {
    int bi = k; for (int i = k + 1; i < nc; ++i) if (std::fabs(a[i * (nc + 1) + k]) > std::fabs(a[bi * (nc + 1) + k])) bi = i;
    for (int j = 0; j <= nc; ++j) std::swap(a[k * (nc + 1) + j], a[bi * (nc + 1) + j]);
    for (int i = k + 1; i < nc; ++i) { const double l = a[i * (nc + 1) + k] / a[k * (nc + 1) + k]; for (int j = k; j <= nc; ++j) a[i * (nc + 1) + j] -= l * a[k * (nc + 1) + j]; }
  }
  for (int k = nc - 1; k >= 0; --k) { double s = a[k * (nc + 1) + nc]; for (int j = k + 1; j < nc; ++j) s -= a[k * (nc + 1) + j] * x[j]; x[k] = s / a[k * (nc + 1) + k]; }
}

int main() {
  const int nb = 1024, nc = 11, reps = 50;
  std::mt19937_64 g(7);
  std::normal_distribution<double> N(0.0, 1.0);
  for (int kind = 0; kind < 3; ++kind) {
    std::vector<double> A((size_t)nb * 16 * 17, 0.0);
    for (int b = 0; b < nb; ++b) {
      int perm[16]; for (int i = 0; i < 16; ++i) perm[i] = i;
      if (kind == 2) for (int i = nc - 1; i > 0; --i) std::swap(perm[i], perm[g() % (i + 1)]);
      for (int i = 0; i < nc; ++i) {
        double* row = &A[((size_t)b * 16 + perm[i]) * 17];
        for (int j = 0; j <= nc; ++j) row[j] = N(g) * std::pow(10.0, N(g));
        if (kind != 1) row[i] += (row[i] >= 0 ? 1 : -1) * 40.0 * std::pow(10.0, std::fabs(N(g)));   // dominant "diagonal" (of the unpermuted system)
      }
    }
    double *dA, *dX; long long* dC; int* dO;
    hipMalloc((void**)&dA, A.size() * 8); hipMalloc((void**)&dX, nb * 16 * 8); hipMalloc((void**)&dC, nb * 8); hipMalloc((void**)&dO, nb * 16 * 4);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    const char* names[3] = {"diagonally dominant", "general random", "dominant entries on a permuted diagonal"};
    for (int which = 0; which < 2; ++which) {
      hipMemset(dX, 0, nb * 16 * 8);
      if (which == 0) hipLaunchKernelGGL((lu_k<12, 0>), dim3(nb), dim3(64), 16 * 17 * 8, 0, dA, dX, dC, dO, nc, reps);
      else hipLaunchKernelGGL((lu_k<12, 1>), dim3(nb), dim3(64), 16 * 17 * 8, 0, dA, dX, dC, dO, nc, reps);
      hipDeviceSynchronize();
      std::vector<double> X(nb * 16); std::vector<long long> C(nb); std::vector<int> O(nb * 16);
      hipMemcpy(X.data(), dX, nb * 16 * 8, hipMemcpyDeviceToHost); hipMemcpy(C.data(), dC, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(O.data(), dO, nb * 16 * 4, hipMemcpyDeviceToHost);
      double worst = 0, cyc = 0; int nan = 0, moved = 0;
      for (int b = 0; b < nb; ++b) {
        double x[16]; host_solve(&A[(size_t)b * 16 * 17], nc, x);
        double xm = 0; for (int i = 0; i < nc; ++i) xm = std::fmax(xm, std::fabs(x[i]));
        for (int i = 0; i < nc; ++i) { const double v = X[b * 16 + i]; if (!(v == v)) ++nan; else worst = std::fmax(worst, std::fabs(v - x[i]) / xm); }
        bool mv = false; for (int i = 0; i < nc; ++i) if (O[b * 16 + i] != i) mv = true;
        moved += mv; cyc += (double)C[b];
      }
      printf("%-42s %-16s cycles/solve %8.0f   max |dx|/|x|max vs host LU %.2e   NaN %d   blocks with a non-identity order %d\n", names[kind],
             which == 0 ? "lu_solve_regs" : "lu_solve_block", cyc / nb, worst, nan, moved);
    }
    hipFree(dA); hipFree(dX); hipFree(dC); hipFree(dO);
  }
  return 0;
}
