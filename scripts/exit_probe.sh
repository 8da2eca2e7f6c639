#!/bin/bash
# Control experiments for the exit-time SIGSEGV of processes profiled with rocprofv3 (VERDICT round 2, item 5 ii): a 20-line HIP shared
# library behind a plain-C main, nothing of cedarhip in the process.  Variant A makes one ordinary launch, variant B one COOPERATIVE
# launch (hipLaunchCooperativeKernel, what the device-resident stepper uses).  Each runs plain and under rocprofv3 --kernel-trace --stats.
# -> gpurun_out/<tag>_exit_probe.log.   usage: bash scripts/exit_probe.sh TAG   (on the GPU box)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
cat > /tmp/probe_lib.hip <<'PROBE'
#include <hip/hip_runtime.h>
__global__ void probe_k(int* p) { *p = 42; }
extern "C" int probe_run(int cooperative) {
  int* d = nullptr; int h = 0;
  if (hipMalloc((void**)&d, sizeof(int)) != hipSuccess) return 2;
  if (cooperative) {
    void* args[] = {(void*)&d};
    if (hipLaunchCooperativeKernel((const void*)probe_k, dim3(1), dim3(64), args, 0, 0) != hipSuccess) return 4;
  } else hipLaunchKernelGGL(probe_k, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 3;
  (void)hipFree(d);
  return h == 42 ? 0 : 1;
}
PROBE
printf 'int probe_run(int);\nint main(int argc, char** argv) { (void)argv; return probe_run(argc > 1); }\n' > /tmp/probe_main.c
: > "$OUT/${TAG}_exit_probe.log"
if /opt/rocm/bin/hipcc -O2 -fPIC -shared --offload-arch=gfx950 -o /tmp/libprobe.so /tmp/probe_lib.hip > /dev/null 2>&1 && gcc -O2 /tmp/probe_main.c -L/tmp -lprobe -Wl,-rpath,/tmp -o /tmp/probe_main; then
  for variant in "" "cooperative"; do
    name=${variant:-ordinary}
    timeout -k 10 120 /tmp/probe_main $variant; echo "control (no cedarhip), $name launch, plain: status $?" | tee -a "$OUT/${TAG}_exit_probe.log"
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_exit_probe_stats" -o run -- /tmp/probe_main $variant > /dev/null 2>&1; echo "control (no cedarhip), $name launch, under rocprofv3 --kernel-trace --stats: status $?" | tee -a "$OUT/${TAG}_exit_probe.log"
  done
else
  echo "control build failed" | tee -a "$OUT/${TAG}_exit_probe.log"
fi
