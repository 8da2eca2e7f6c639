/* Plain-C client of the C-ABI (include/cedarhip.h): what a Julia `ccall` binding, or any other FFI, does.
 * Circuit: V1 (PWL step 0 -> 1 V in 1 ns) - R1 (1 kOhm) - C1 (1 nF) to ground.  DC operating point, then a transient to
 * 5 tau with the node voltage saved at five times; compared with the closed form 1 - exp(-t/RC).
 * Build:  gcc -O2 -Iinclude examples/c_abi_demo.c -Lcedarsim.jl_amd/lib -lcedarhip -Wl,-rpath,$PWD/cedarsim.jl_amd/lib -lm -o c_abi_demo */
#define _GNU_SOURCE
#include <math.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <unistd.h>

#include "cedarhip.h"

/* CEDARHIP_DEMO_MAPS=1: attribute a crash at process exit (seen under rocprofv3: SIGSEGV inside __cxa_finalize after main has
 * returned 0).  A SIGSEGV handler prints the faulting address and instruction pointer, then /proc/self/maps, so that the frame
 * can be assigned to a mapped object (libcedarhip.so's fat-binary unregistration, the HIP runtime, or the profiler's tool
 * library); the same dump is also written once at the normal end of main.  Async-signal-safe calls only. */
static void dump_maps(void) {
  char buf[4096];
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return;
  while (fgets(buf, sizeof buf, f)) if (strstr(buf, " r-xp ") || strstr(buf, "cedarhip") || strstr(buf, "rocprof") || strstr(buf, "amdhip")) fputs(buf, stderr);
  fclose(f);
}
static void on_segv(int sig, siginfo_t* si, void* uc_) {
  ucontext_t* uc = (ucontext_t*)uc_;
  char line[160];
  int n = snprintf(line, sizeof line, "[c_abi_demo] signal %d: fault address %p, instruction pointer %p\n", sig, si->si_addr,
                   (void*)uc->uc_mcontext.gregs[REG_RIP]);
  if (n > 0) (void)!write(2, line, (size_t)n);
  int fd = open("/proc/self/maps", 0);
  if (fd >= 0) { char b[4096]; ssize_t k; while ((k = read(fd, b, sizeof b)) > 0) (void)!write(2, b, (size_t)k); close(fd); }
  _exit(128 + sig);
}

int main(void) {
  if (getenv("CEDARHIP_DEMO_MAPS")) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, NULL);
  }
  char err[512];
  ch_ctx* ctx = ch_create(0, err, sizeof err);
  if (!ctx) { fprintf(stderr, "ch_create: %s\n", err); return 2; }   /* no CPU fallback */

  /* nodes: 1 = in, 2 = out; devices: V1(in,0), R1(in,out), C1(out,0) */
  int32_t dev_kind[3] = {CH_DEV_V, CH_DEV_R, CH_DEV_C};
  int32_t dev_node[3 * CH_DEV_NNODE] = {0};
  int32_t dev_ipar[3 * CH_DEV_NIPAR] = {0};
  double dev_par[3 * CH_DEV_NPAR], dev_mult[3] = {1.0, 1.0, 1.0};
  for (int i = 0; i < 3 * CH_DEV_NPAR; ++i) dev_par[i] = CH_NAN;
  dev_node[0 * CH_DEV_NNODE + 0] = 1; dev_node[0 * CH_DEV_NNODE + 1] = 0; dev_ipar[0 * CH_DEV_NIPAR] = 0; /* source 0 */
  dev_node[1 * CH_DEV_NNODE + 0] = 1; dev_node[1 * CH_DEV_NNODE + 1] = 2; dev_par[1 * CH_DEV_NPAR] = 1e3;
  dev_node[2 * CH_DEV_NNODE + 0] = 2; dev_node[2 * CH_DEV_NNODE + 1] = 0; dev_par[2 * CH_DEV_NPAR] = 1e-9;
  int32_t src_kind[1] = {CH_SRC_PWL};
  double src_dc[1] = {0.0}, src_par[CH_SRC_NPAR] = {0};
  int32_t pwl_ofs[2] = {0, 3};
  double pwl_t[3] = {0.0, 1e-9, 1.0}, pwl_y[3] = {0.0, 1.0, 1.0};
  int32_t obs_kind[1] = {0}, obs_index[1] = {2};

  ch_desc d;
  memset(&d, 0, sizeof d);
  d.n_nodes = 2; d.n_dev = 3;
  d.dev_kind = dev_kind; d.dev_node = dev_node; d.dev_ipar = dev_ipar; d.dev_par = dev_par; d.dev_mult = dev_mult;
  d.n_src = 1; d.src_kind = src_kind; d.src_dc = src_dc; d.src_par = src_par; d.src_pwl_ofs = pwl_ofs; d.pwl_t = pwl_t; d.pwl_y = pwl_y;
  d.temp = 27.0; d.gmin = 1e-12; d.scale = 1.0;
  d.n_obs = 1; d.obs_kind = obs_kind; d.obs_index = obs_index;

  ch_circuit* c = ch_circuit_build(ctx, &d);
  if (!c) { fprintf(stderr, "ch_circuit_build: %s\n", ch_last_error(ctx)); return 3; }
  ch_info info;
  ch_circuit_info(c, &info);
  printf("unknowns after structural reduction: %d of %d MNA unknowns\n", info.n_unknowns, info.n_mna);

  double saveat[5] = {1e-6, 2e-6, 3e-6, 4e-6, 5e-6};
  ch_tran_opts o;
  ch_tran_opts_default(&o);
  o.abstol = 1e-9; o.reltol = 1e-7; o.n_saveat = 5; o.saveat = saveat;
  ch_result* r = NULL;
  int rc = ch_tran(c, 0.0, 5e-6, &o, &r);
  if (rc != CH_OK || !r) { fprintf(stderr, "ch_tran: rc=%d %s\n", rc, ch_last_error(ctx)); return 4; }
  const int64_t nt = ch_result_n_times(r);
  const double *t = ch_result_times(r), *v = ch_result_values(r);
  double worst = 0.0;
  for (int64_t i = 0; i < nt; ++i) {
    /* ramp of 1 ns in front of the step: v(t) = 1 - (tau/tr)(1 - exp(-tr/tau)) exp(-(t - tr)/tau) for t >= tr */
    const double tau = 1e-6, tr = 1e-9;
    const double want = 1.0 - (tau / tr) * (1.0 - exp(-tr / tau)) * exp(-(t[i] - tr) / tau);
    printf("t = %.1e s   v(out) = %.9f   closed form %.9f\n", t[i], v[i], want);
    if (fabs(v[i] - want) > worst) worst = fabs(v[i] - want);
  }
  ch_stats st;
  ch_result_stats(r, &st);
  printf("accepted steps %lld, Newton iterations %lld, max |error| %.2e\n", (long long)st.naccept, (long long)st.nnonliniter, worst);
  ch_result_free(r);
  ch_circuit_free(c);
  ch_destroy(ctx);
  if (getenv("CEDARHIP_DEMO_MAPS")) { fprintf(stderr, "[c_abi_demo] main ends normally; executable mappings:\n"); dump_maps(); }
  return worst < 1e-5 && nt == 5 ? 0 : 1;
}
