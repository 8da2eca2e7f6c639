// fp64 issue probe (gfx950): cycles per wave-instruction of the operations the Newton kernels are made of, for ONE wave alone on
// its SIMD and for 2 / 4 co-resident waves, with 1 / 2 / 4 / 8 independent dependency chains per lane.  Answers: what does a wave
// that is alone on its SIMD (the persistent kernel: 4 waves per CU) give away against two waves per SIMD, and what does a
// dependent fp64 chain cost per link.  Build: hipcc --offload-arch=gfx950 -O3 scripts/fp64_issue_probe.hip -o scripts/_bin/fp64_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH, int OP>
__global__ void probe(double* out, long long* cyc, int n_iter, double a, double b) {
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = 1.0 + 1e-3 * (threadIdx.x + c);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n_iter; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (OP == 0) x[c] = __builtin_fma(x[c], a, b);                         // v_fma_f64
        else if (OP == 1) x[c] = x[c] * a;                                     // v_mul_f64
        else if (OP == 2) x[c] = x[c] + b;                                     // v_add_f64
        else if (OP == 3) x[c] = __builtin_amdgcn_rcp(x[c]) + b;               // v_rcp_f64 + add
        else if (OP == 4) x[c] = __builtin_fma(__shfl_xor(x[c], 1), a, b);     // DPP quad_perm / ds_swizzle + fma
        else if (OP == 5) x[c] = __builtin_fma(__shfl(x[c], 3), a, b);         // broadcast of one lane (readlane or bpermute) + fma
        else if (OP == 6) { float f = (float)x[c]; f = __builtin_fmaf(f, (float)a, (float)b); x[c] = (double)f; }   // cvt + f32 fma + cvt
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH, int OP>
static double run(int waves_per_simd, int n_cu) {
  const int threads = 256 * waves_per_simd;   // 4 SIMDs per CU: 256 threads = one wave per SIMD
  const int n_iter = 2000;
  double* out; long long* cyc;
  hipMalloc((void**)&out, sizeof(double) * n_cu * threads); hipMalloc((void**)&cyc, sizeof(long long) * n_cu);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<CH, OP>), dim3(n_cu), dim3(threads), 0, 0, out, cyc, n_iter, 0.999999, 1e-6);
  hipDeviceSynchronize();
  std::vector<long long> h(n_cu);
  hipMemcpy(h.data(), cyc, sizeof(long long) * n_cu, hipMemcpyDeviceToHost);
  hipFree(out); hipFree(cyc);
  double m = 0; for (long long v : h) m += (double)v; m /= n_cu;
  return m / ((double)n_iter * 8 * CH);   // shader cycles per wave-instruction of THIS wave (elapsed / instructions it issued)
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  const char* names[7] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64+add", "shfl_xor1+fma", "shfl(lane3)+fma", "cvt+fma_f32+cvt"};
  printf("cycles per wave-instruction group as seen by one wave (elapsed cycles / its own instruction count)\n");
  printf("%-18s %-7s %8s %8s %8s\n", "op", "chains", "1w/SIMD", "2w/SIMD", "4w/SIMD");
#define ROW(CH, OP) printf("%-18s %-7d %8.2f %8.2f %8.2f\n", names[OP], CH, run<CH, OP>(1, n_cu), run<CH, OP>(2, n_cu), run<CH, OP>(4, n_cu));
  ROW(1, 0) ROW(2, 0) ROW(4, 0) ROW(8, 0)
  ROW(1, 1) ROW(4, 1) ROW(1, 2) ROW(4, 2)
  ROW(1, 3) ROW(4, 3)
  ROW(1, 4) ROW(4, 4) ROW(1, 5) ROW(4, 5)
  ROW(1, 6) ROW(4, 6)
  return 0;
}
