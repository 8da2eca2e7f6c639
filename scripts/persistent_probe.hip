// Feasibility probe for device-resident stepping: 1024 one-wave workgroups stay resident and wait for a command sequence
// number in mapped host memory; on each command every workgroup does `work_us` of busy work, then writes a 48-byte record to
// mapped host memory followed by the sequence number (system-scope release).  The host measures the round trip per command.
// Every wait has a bounded spin count, so the kernel always drains.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
struct Rec { double a, b, c, d, e; int pad, seq; };
__global__ __launch_bounds__(64) void worker(volatile int* cmd, int* dev_cmd, int relay, Rec* out, int n_cmd, long work_cycles, long max_spin) {
  const int blk = blockIdx.x;
  for (int k = 1; k <= n_cmd; ++k) {
    long spin = 0;
    int seen = 0;
    if (threadIdx.x == 0) {
      if (!relay || blk == 0) {
        while ((seen = __hip_atomic_load((int*)cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) < k && ++spin < max_spin) __builtin_amdgcn_s_sleep(1);
        if (relay) __hip_atomic_store(dev_cmd, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // one poller on PCIe, the rest watch L2
      } else {
        while ((seen = __hip_atomic_load(dev_cmd, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < k && ++spin < max_spin) __builtin_amdgcn_s_sleep(1);
      }
    }
    seen = __shfl(seen, 0);
    if (seen < k) return;   // host went away: drain
    const long t0 = clock64();
    double x = 1.0 + threadIdx.x;
    while (clock64() - t0 < work_cycles) x = fma(x, 1.0000001, 1e-9);
    if (threadIdx.x == 0) {
      Rec r; r.a = x; r.b = r.c = r.d = r.e = 0.0; r.pad = 0; r.seq = 0;
      out[blk] = r;
      __hip_atomic_store(&out[blk].seq, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
int main(int argc, char** argv) {
  const int nblk = 1024, n_cmd = 500;
  int* cmd; Rec* out;
  hipHostMalloc((void**)&cmd, 64, hipHostMallocMapped); hipHostMalloc((void**)&out, nblk * sizeof(Rec), hipHostMallocMapped);
  int* dev_cmd; hipMalloc((void**)&dev_cmd, 64);
  for (int relay : {0, 1}) for (double work_us : {0.0, 40.0}) {
    *cmd = 0; memset(out, 0, nblk * sizeof(Rec)); hipMemset(dev_cmd, 0, 64); hipDeviceSynchronize();
    const long work_cycles = (long)(work_us * 100.0);   // clock64 ticks at 100 MHz
    hipLaunchKernelGGL(worker, dim3(nblk), dim3(64), 0, 0, (volatile int*)cmd, dev_cmd, relay, out, n_cmd, work_cycles, 20000000L);
    auto t0 = std::chrono::steady_clock::now();
    bool ok = true;
    for (int k = 1; k <= n_cmd && ok; ++k) {
      __atomic_store_n(cmd, k, __ATOMIC_RELEASE);
      int next = 0; long spin = 0;
      while (next < nblk) { while (next < nblk && __atomic_load_n(&out[next].seq, __ATOMIC_ACQUIRE) >= k) ++next; if (++spin > 400000000L) { ok = false; break; } }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (!ok) { __atomic_store_n(cmd, n_cmd + 1, __ATOMIC_RELEASE); printf("TIMEOUT\n"); }
    hipDeviceSynchronize();
    printf("relay %d work %.0f us: %.2f us per command round trip (%d commands)%s\n", relay, work_us, std::chrono::duration<double>(t1 - t0).count() * 1e6 / n_cmd, n_cmd, ok ? "" : " [aborted]");
  }
  return 0;
}
