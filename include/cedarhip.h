/* cedarhip.h — C-ABI of the MI355X-native transient/DC Newton engine for CedarSim-style circuits.
 *
 * This is the drop-in boundary described in DESIGN.md §2.  CedarSim (Julia) has NO FFI for this
 * path today (the only ccall in the reference is :jl_generating_output, src/CedarSim.jl:44); the
 * hot path runs inside un-vendored Julia/C packages (DAECompiler, Sundials IDA, NonlinearSolve).
 * Each entry point below therefore cites the reference *interface* whose work it replaces; the
 * Julia-side binding a maintainer would add is shown in INTEGRATION.md and cedarsim.jl_amd/julia/.
 *
 * Conventions
 *   - plain C, no torch types; all arrays are caller-owned and copied by the engine unless stated.
 *   - node ids: 0 is ground, 1..n_nodes are circuit nodes.
 *   - MNA solution vectors ("x_mna") have length n_nodes + n_branches:
 *       x_mna[0..n_nodes)            node voltages of node 1..n_nodes
 *       x_mna[n_nodes + k]           current of the k-th branch device (V, L, E in device order),
 *                                    flowing from net+ to net- through the device — the sign
 *                                    convention of branch!() (src/simulate_ir.jl:112-120).
 *   - all functions return 0 on success or a negative CH_ERR_* code; they never throw: every entry point catches
 *     C++ exceptions at the boundary (CH_ERR_NOMEM / CH_ERR_INTERNAL; builders return NULL), the way CedarDCOp
 *     swallows solver exceptions and reports a retcode (src/dcop.jl:58-82).
 *   - a ch_ctx / ch_circuit is not thread-safe; use one host thread per GPU (one process per GPU).
 */
#ifndef CEDARHIP_H
#define CEDARHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CH_NAN (__builtin_nan(""))

/* ---- error codes (map to SciML retcodes on the Julia side) ---- */
enum {
  CH_OK = 0,
  CH_ERR_INVALID = -1,      /* bad argument / malformed description (CedarError)            */
  CH_ERR_SINGULAR = -2,     /* singular Jacobian (LinearAlgebra.SingularException)          */
  CH_ERR_MAXITERS = -3,     /* DC did not converge: ReturnCode.InitialFailure / MaxIters    */
  CH_ERR_DTMIN = -4,        /* step size underflow: ReturnCode.DtLessThanMin                */
  CH_ERR_DEVICE = -5,       /* HIP runtime failure                                          */
  CH_ERR_UNSUPPORTED = -6,  /* feature outside the engine's scope                           */
  CH_ERR_MAXSTEPS = -7,     /* transient exceeded max_steps: ReturnCode.MaxIters            */
  CH_ERR_NOMEM = -8,        /* host allocation failed inside the call (std::bad_alloc / length_error) */
  CH_ERR_INTERNAL = -9      /* any other C++ exception caught at the boundary (message in ch_last_error) */
};

/* ---- device kinds (reference device functors, src/simpledevices.jl, src/vasim.jl) ---- */
enum {
  CH_DEV_R = 1,    /* SimpleResistor   simpledevices.jl:65-77   par[0]=r                      */
  CH_DEV_C = 2,    /* SimpleCapacitor  simpledevices.jl:105-109 par[0]=c                      */
  CH_DEV_L = 3,    /* SimpleInductor   simpledevices.jl:128-132 par[0]=l     (branch)         */
  CH_DEV_V = 4,    /* VoltageSource    simpledevices.jl:288-300 ipar[0]=source (branch)       */
  CH_DEV_I = 5,    /* CurrentSource    simpledevices.jl:327-339 ipar[0]=source                */
  CH_DEV_VCVS = 6, /* vcvs             simpledevices.jl:347-356 par[0]=gain  (branch)         */
  CH_DEV_VCCS = 7, /* vccs             simpledevices.jl:364-373 par[0]=gain                   */
  CH_DEV_MOS = 8,  /* BSIM4 functor (VA-generated in the reference, vasim.jl:853-867):
                      nodes d,g,s,b ; ipar[0]=model ; par = {w,l,nf,as,ad,ps,pd,-}            */
  CH_DEV_VA = 9    /* compiled Verilog-A module (device functor of make_spice_device, vasim.jl:649-867):
                      nodes = ports then internal nets (the caller allocates the internal nets as
                      circuit nodes) ; ipar[0] = module id (ch_va_find) ; ipar[1] = offset of the
                      instance's parameter block in ch_desc.va_par                              */
};
#define CH_DEV_NNODE 8
#define CH_DEV_NPAR 8
#define CH_DEV_NIPAR 2

/* MOS instance parameter slots inside par[] */
enum { CH_MOS_W = 0, CH_MOS_L = 1, CH_MOS_NF = 2, CH_MOS_AS = 3, CH_MOS_AD = 4, CH_MOS_PS = 5, CH_MOS_PD = 6 };

/* ---- source waveforms (src/spectre_env.jl:144-176) ---- */
enum {
  CH_SRC_DC = 0,
  CH_SRC_PWL = 1,   /* pwl(wave)            spectre_env.jl:43-69,144-151 */
  CH_SRC_PULSE = 2, /* pulse(v1,v2,td,tr,tf,pw,period)  :153-166 ; par = v1 v2 td tr tf pw period */
  CH_SRC_SIN = 3    /* spsin(vo,va,freq,td,theta,phase,ncycles) :169-176                           */
};
#define CH_SRC_NPAR 8

/* ---- BSIM4 model-card parameter indices ---- */
enum {
#define P(n, d) CH_B4_##n,
#define B(n, d) CH_B4_##n, CH_B4_l##n, CH_B4_w##n, CH_B4_p##n,
#define I(n)
#include "cedarhip_bsim4_params.def"
  CH_B4_NPAR
};

/* ---- sweepable parameter slots: the runtime fields of ParamSim's `p` (circuitodesystem.jl:66-97) ---- */
enum {
  CH_SLOT_DEV_PAR = 1,   /* a = device index, b = par index            */
  CH_SLOT_MODEL_PAR = 2, /* a = model index,  b = CH_B4_* index        */
  CH_SLOT_SRC_DC = 3,    /* a = source index  (dc AND constant tran)   */
  CH_SLOT_SRC_PAR = 4,   /* a = source index, b = par index            */
  CH_SLOT_TEMP = 5,      /* SimSpec.temp  (simulate_ir.jl:15)          */
  CH_SLOT_GMIN = 6,      /* SimSpec.gmin  (simulate_ir.jl:16)          */
  CH_SLOT_DEV_MULT = 7,  /* a = device index: ParallelInstances m      */
  CH_SLOT_VA_PAR = 8     /* a = index into ch_desc.va_par (one entry of a compiled Verilog-A instance's parameter block) */
};

/* Flat circuit description = what a StampExtract overlay over the netlist closure records
 * (SURVEY §3.5): device type, field values, net ids, multiplier. */
typedef struct ch_desc {
  int32_t n_nodes; /* excluding ground */
  /* devices */
  int32_t n_dev;
  const int32_t* dev_kind; /* [n_dev] CH_DEV_*                                   */
  const int32_t* dev_node; /* [n_dev*CH_DEV_NNODE] node ids (unused = 0)         */
  const int32_t* dev_ipar; /* [n_dev*CH_DEV_NIPAR]                               */
  const double* dev_par;   /* [n_dev*CH_DEV_NPAR] (NaN = not given)              */
  const double* dev_mult;  /* [n_dev] multiplicity m (simulate_ir.jl:56-75)      */
  /* sources */
  int32_t n_src;
  const int32_t* src_kind;    /* [n_src] CH_SRC_* : transient waveform            */
  const double* src_dc;       /* [n_src] value in :dcop mode (simpledevices.jl:290-291) */
  const double* src_par;      /* [n_src*CH_SRC_NPAR]                              */
  const int32_t* src_pwl_ofs; /* [n_src+1] offsets into pwl_t/pwl_y               */
  const double* pwl_t;
  const double* pwl_y;
  /* BSIM4 model cards */
  int32_t n_model;
  const double* model_par; /* [n_model*CH_B4_NPAR], NaN = not given */
  /* SimSpec (simulate_ir.jl:12-20) */
  double temp;  /* Celsius, default 27 */
  double gmin;  /* default 1e-12       */
  double scale; /* default 1           */
  /* sweepable slots */
  int32_t n_slot;
  const int32_t* slot_kind; /* [n_slot] CH_SLOT_* */
  const int32_t* slot_a;    /* [n_slot] */
  const int32_t* slot_b;    /* [n_slot] */
  /* observables saved by ch_tran: node voltages and branch currents */
  int32_t n_obs;
  const int32_t* obs_kind;  /* [n_obs] 0 = node voltage (index = node id), 1 = branch current (index = device index) */
  const int32_t* obs_index; /* [n_obs] */
  /* small-signal analysis: |ac| of every source (VoltageSource/CurrentSource `ac`, the phase is ignored as
   * in src/simpledevices.jl:293,332), or NULL when no source has one.  A voltage source with ac != 0 keeps
   * its node and branch unknowns (it is never folded into a known node). */
  const double* src_ac;     /* [n_src] or NULL */
  /* parameter blocks of the CH_DEV_VA instances: per instance 2*n_params doubles — the module's parameters in
   * declaration order (defaults already applied, integers as doubles) followed by the $param_given flags */
  int64_t n_va_par;
  const double* va_par;     /* [n_va_par] or NULL */
} ch_desc;

/* Statistics — the fields CedarSim accumulates from NLStats/DEStats (src/dcop.jl:63-67,134-139) */
typedef struct ch_stats {
  int64_t nf;              /* residual (device) evaluations                     */
  int64_t njacs;           /* Jacobian evaluations                              */
  int64_t nfactors;        /* LU factorisations                                 */
  int64_t nsolve;          /* triangular solves                                 */
  int64_t nnonliniter;     /* Newton iterations (max over blocks, summed over samples: see DESIGN.md) */
  int64_t nnonlinconvfail; /* Newton convergence failures                       */
  int64_t naccept;         /* accepted time steps                               */
  int64_t nreject;         /* rejected time steps (LTE)                         */
  int64_t nrestarts;       /* DC random restarts used                           */
  double wall_seconds;     /* host wall-clock inside the call                   */
  double dc_seconds;       /* of which DC initialisation                        */
  double device_seconds;   /* sum of dominant-kernel durations measured with HIP events (0 for oracle) */
  int64_t n_kernel_launches;
  int64_t n_block_iters;   /* Newton iterations summed over every Jacobian block (exact work count) */
  int64_t n_step_attempts; /* time-step attempts (accepted + rejected + Newton failures)                */
  double barrier_seconds;  /* device-resident stepper: time one wave spent inside the grid-wide reductions (0 otherwise) */
  int32_t stepper;         /* which step controller ran: CH_STEPPER_HOST or CH_STEPPER_DEVICE (0 for DC-only calls) */
  int32_t stepper_mode;    /* device-resident stepper: CH_MODE_LOCKSTEP (one step sequence for the whole circuit / batch), CH_MODE_OWN_STEPS
                              (every independent block / sample its own sequence, output on the saveat grid), CH_MODE_BORDERED (coupled array
                              torn at its rail nodes: Schur complement on the border inside the kernel); 0 on the host stepper */
  /* the dominant kernel of the call, for roofline accounting: the time-stepping kernel(s) only, DC initialisation excluded */
  double step_kernel_seconds;    /* HIP-event time of those launches (device stepper: exact; host stepper: sampled launches, scaled) */
  int64_t step_kernel_launches;  /* their number (1 per transient on the device stepper, 1 per attempt on the host stepper)         */
  int64_t step_block_iters;      /* Newton iterations summed over blocks inside them                                               */
} ch_stats;

/* CedarDCOp options (src/dcop.jl:24-28, 53-94) */
typedef struct ch_dc_opts {
  double abstol;      /* residual inf-norm tolerance, default 1e-10 (dcop.jl:28)          */
  int32_t maxiters;   /* default 200 (dcop.jl:53)                                          */
  int32_t n_restarts; /* default 10 (num_trajectories, dcop.jl:53)                         */
  uint64_t seed;      /* RNG seed for u0 = 1e-7*randn (dcop.jl:60)                         */
  int32_t tran_mode;  /* 0: :dcop (sources at .dc), 1: :tranop (sources at tran(t=0))      */
  double dv_max;      /* Newton damping: max node-voltage change per iteration (0 = off)   */
  const double* x0;   /* optional initial guess [n_samples][n_mna] or NULL                 */
} ch_dc_opts;

/* transient options — solve(prob, IDA(); abstol, reltol) (src/sweeps.jl:456) */
typedef struct ch_tran_opts {
  double abstol;     /* default 1e-6 */
  double reltol;     /* default 1e-3 */
  int32_t max_order; /* 1..5 variable-order BDF, default 5 (IDA) */
  double dtmin;      /* default 1e-18*(t1-t0)... 0 = auto */
  double dtmax;      /* 0 = auto ((t1-t0)/10) */
  double dt0;        /* initial step, 0 = auto */
  int32_t max_steps; /* accepted steps before CH_ERR_MAXSTEPS (ReturnCode.MaxIters); 0 = 100 000, the default maxiters of solve(prob, IDA()) in Sundials.jl (the
                        reference passes none, src/sweeps.jl:456).  A transient that creeps (Newton failing whenever the step grows) ends there. */
  int32_t newton_maxiters; /* default 10 */
  int32_t n_saveat;        /* 0: save every accepted step (ONE step sequence and time vector for the whole circuit / batch) */
  const double* saveat;    /* [n_saveat] increasing times to save (dense output by interpolation).  With a saveat grid the samples of a batch,
                              and the structurally independent blocks of one circuit, may each take their own step sequence (what
                              sol(t) of separate solves would give; error control per sample / block) */
  ch_dc_opts dc;           /* initialisation (CedarDCOp) */
  int32_t skip_dc;         /* 1: start from dc.x0 as given (u0 passed by the caller, test/common.jl:36-43) */
  int32_t stepper;         /* CH_STEPPER_AUTO (default): device-resident controller where the circuit qualifies, host otherwise;
                              CH_STEPPER_HOST / CH_STEPPER_DEVICE force one (DEVICE fails with CH_ERR_UNSUPPORTED when it cannot run) */
  int32_t step_control;    /* CH_STEPS_AUTO (default): with a saveat grid, independent blocks / samples take their own steps under their
                              own error norm (above).  CH_STEPS_SHARED: ONE step sequence and ONE error norm over the whole circuit /
                              batch whatever the output grid — the answer of a single IDA() integrator over the whole system
                              (src/sweeps.jl:456), and the same controller a run without saveat uses: results of the two are then
                              comparable point for point */
} ch_tran_opts;

/* Where the sequential step controller of ch_tran runs (the policy is the same): on the host with one kernel
 * launch per step attempt, or inside ONE persistent cooperative launch per transient (blocks stay resident, the
 * accept/reject/order/step decision is taken from a grid-wide reduction by every wavefront identically). */
enum { CH_STEPPER_AUTO = 0, CH_STEPPER_HOST = 1, CH_STEPPER_DEVICE = 2 };
enum { CH_MODE_LOCKSTEP = 1, CH_MODE_OWN_STEPS = 2, CH_MODE_BORDERED = 3 };
enum { CH_STEPS_AUTO = 0, CH_STEPS_SHARED = 1 };

typedef struct ch_ctx ch_ctx;
typedef struct ch_circuit ch_circuit;
typedef struct ch_result ch_result;

typedef struct ch_info {
  int32_t n_nodes, n_branches, n_mna; /* MNA sizes of the description                          */
  int32_t n_unknowns;                 /* unknowns after structural simplification              */
  int32_t n_known;                    /* nodes eliminated as known (grounded-source chains)    */
  int32_t n_alias;                    /* nodes merged by 0 V ammeter sources                   */
  int32_t n_components;               /* independent diagonal blocks of the Jacobian           */
  int32_t max_component;              /* unknowns in the largest block                         */
  int32_t n_classes;                  /* structurally distinct blocks                          */
  int32_t n_mos, n_mos_classes;       /* MOS instances, distinct (model,geometry) classes      */
  int32_t path;                       /* 1 = fused block kernel, 2 = sparse level-scheduled    */
  int64_t nnz_jac, nnz_lu;            /* sparse path only                                      */
  int32_t n_samples;
} ch_info;

/* ---- defaults ---- */
void ch_dc_opts_default(ch_dc_opts*);
void ch_tran_opts_default(ch_tran_opts*);

/* ---- context: one per GPU.  Replaces: nothing in the reference (CPU only). ---- */
ch_ctx* ch_create(int device_id, char* err, size_t errlen);
void ch_destroy(ch_ctx*);
const char* ch_last_error(ch_ctx*);

/* ---- circuit: replaces CircuitIRODESystem(circuit) + DAEProblem(sys, …) construction
 *      (src/circuitodesystem.jl:147-164, src/sweeps.jl:444): structural analysis done once. ---- */
ch_circuit* ch_circuit_build(ch_ctx*, const ch_desc*);
void ch_circuit_free(ch_circuit*);
int ch_circuit_info(ch_circuit*, ch_info*);
/* Result of the structural analysis, for host mirrors and tests: for node 0..n_nodes the index of
 * the unknown that carries its voltage (or -1) and of the known value (or -1); for each MNA branch
 * the unknown of its current (or -1 when the source was eliminated). */
int ch_circuit_maps(ch_circuit*, int32_t* node_unknown, int32_t* node_known, int32_t* branch_unknown);

/* ---- samples: replaces remake(prob, p=sim) over the sweep (src/sweeps.jl:471-482). ----
 * values is slot-major: values[slot_i * (sample_hi-sample_lo) + (s - sample_lo)]. */
int ch_set_samples(ch_circuit*, int32_t n_samples);
int ch_set_params(ch_circuit*, int32_t sample_lo, int32_t sample_hi, int32_t n_slots,
                  const int32_t* slot_ids, const double* values);

/* ---- DC operating point: replaces initialize_dae!(…, ::CedarDCOp) + bootstrapped_nlsolve
 *      (src/dcop.jl:53-94,157-200).  x_out: [n_samples][n_mna]. status_out: [n_samples] or NULL. ---- */
int ch_dc(ch_circuit*, const ch_dc_opts*, double* x_out, int32_t* status_out, ch_stats* stats);

/* ---- transient: replaces solve(prob, IDA(); abstol, reltol, initializealg=CedarDCOp())
 *      (src/sweeps.jl:456, test/gf180_dff.jl:25). ---- */
int ch_tran(ch_circuit*, double t0, double t1, const ch_tran_opts*, ch_result** out);

int64_t ch_result_n_times(const ch_result*);
const double* ch_result_times(const ch_result*);  /* [n_times]                                  */
const double* ch_result_values(const ch_result*); /* [n_obs][n_times][n_samples]                */
/* Dense output of a run WITHOUT a saveat grid (`sol(t, idxs=…)`, test/gf180_dff.jl:29-33): entry i is the number m of newest
 * saved points, row i included, through which the polynomial of the step that ended at times[i] runs — for t in
 * (times[i-1], times[i]] the solution is the Lagrange interpolant through rows i-m+1 .. i, which is what a saveat grid would
 * have returned there.  0: no polynomial for this row (the initial state; every row of a run on a saveat grid).  [n_times] */
const int32_t* ch_result_dense_points(const ch_result*);
/* The values of ch_result_values STILL IN HBM, same layout [n_obs][n_times][n_samples], for a device-side consumer — the result gather of
 * a sharded sweep (src/sweeps.jl:471-502 collects the per-point solutions on the host; here RCCL moves the rows GPU to GPU).  *ptr is a
 * device pointer on the context's GPU, owned by the circuit and valid until the next call on that circuit or ch_circuit_free.
 * CH_ERR_UNSUPPORTED (and *ptr = NULL) when the rows are not kept: the host stepper ran, the row buffer was drained in batches, or an
 * observable is a known node / merged node / eliminated branch (those are filled in on the host). */
int ch_result_device_values(const ch_result*, const double** ptr, int64_t* n_doubles);
const double* ch_result_final_state(const ch_result*); /* [n_samples][n_mna] at the last time   */
int ch_result_stats(const ch_result*, ch_stats*);
int ch_result_status(const ch_result*);           /* CH_OK or CH_ERR_*                          */
void ch_result_free(ch_result*);

/* ---- one residual/Jacobian evaluation: replaces prob.f.f(out,du,u,p,t) and
 *      prob.f.jac(J,du,u,p,γ,t) (benchmarks/benchmark_common.jl:138,155).
 *      Evaluates on the GPU at x_mna for sample `sample`:
 *        F = i(x,t) + alpha0*q(x)   (history term excluded),  Q = q(x),
 *        J = di/dx + alpha0*dq/dx as dense row-major [n_mna*n_mna] (rows/cols of eliminated
 *        unknowns are returned as identity rows so the matrix stays usable by a host LU).
 *      mode: 0 = :dcop source values, 1 = :tran source values at t. ---- */
int ch_eval(ch_circuit*, int32_t sample, const double* x_mna, double t, double alpha0, int32_t mode,
            double* F_out, double* Q_out, double* J_out);

/* ---- BSIM4 device evaluation only (roofline kernel): evaluates every MOS instance at the
 *      given terminal voltages.  v: [n_mos][4] (d,g,s,b); out: [n_mos][40] =
 *      {i[4], q[4], g[16], c[16]} per instance.  Used by the parity tests and bench. ---- */
int ch_mos_eval(ch_circuit*, int32_t sample, const double* v, double* out);
/* Same result computed the way the Newton kernel does it: four lanes per instance, one partial
 * derivative per lane, columns exchanged with quad shuffles. */
int ch_mos_eval_quad(ch_circuit*, int32_t sample, const double* v, double* out);

/* ---- names of BSIM4 parameters (for host-side card parsing) ---- */
int32_t ch_bsim4_npar(void);
const char* ch_bsim4_param_name(int32_t idx);
int32_t ch_bsim4_param_ignored(const char* name); /* 1 if accepted-and-ignored */

/* ---- small-signal AC analysis.  Replaces: ac!(circ) + freqresp(ac, sym, ωs) (src/ac.jl:166-177, 75-102,
 * 267-284): DC operating point with CedarDCOp, linearisation A = ∂F/∂u, E = mass matrix, B = ∂F/∂ϵω, then
 * C·(jωE − A)⁻¹·B per frequency.  Here: G = ∂i/∂x and C = ∂q/∂x from the same device kernels at the DC
 * point, and one batched complex LU (G + jωC)·x = b per (block, sample, frequency) on the GPU.
 * freqs_hz[n_freq]; x_ac_out[n_samples][n_freq][n_mna][2] (re, im) in MNA order, like ch_dc's x_out
 * (eliminated branch currents are NaN; nodes tied to ground through non-AC sources are 0).  A circuit on the sparse path is
 * taken as one dense block per sample and may have up to 4096 unknowns (CH_ERR_UNSUPPORTED above that). ---- */
int ch_ac(ch_circuit*, const ch_dc_opts*, int32_t n_freq, const double* freqs_hz, double* x_ac_out, ch_stats* stats);

/* ---- output-noise PSD.  Replaces: noise!(circ) + PSD(noise, sym, ωs) (src/ac.jl:136-163, 178-186, 286-305).
 * Noise sources built: resistor thermal noise, 4kT/R per instance (src/simpledevices.jl:72-76), T = temp + 273.15, and the
 * white / flicker sources of compiled Verilog-A modules (BSIM-CMG: test/ac.jl:155-237).  The noise sources of the hand-written
 * BSIM4 functor are NOT built (test/inverter_noise.jl:56-124 would pin them, with GF180 cards that are not in the reference tree):
 * CH_DEV_MOS devices enter the analysis with their small-signal conductances and capacitances only and contribute no noise power of
 * their own — the PSD of such a circuit is that of its resistors and compiled devices seen through the MOSFETs.
 * out_kind/out_index select the observed unknown like ch_desc.obs_kind/obs_index (0 = node voltage, 1 = branch
 * current); psd_out[n_samples][n_freq] in V²/Hz (A²/Hz), computed with one adjoint solve per frequency. ---- */
int ch_noise(ch_circuit*, const ch_dc_opts*, int32_t out_kind, int32_t out_index, int32_t n_freq, const double* freqs_hz,
             double* psd_out, ch_stats* stats);

/* ---- compiled Verilog-A modules.  Replaces: the device functors `make_spice_device` generates from a
 * parsed module (src/vasim.jl:649-867).  Modules are compiled ahead of time into the library by
 * cedarsim.jl_amd/va (front-end + code generator) — the counterpart of the reference's precompiled model
 * packages (src/ModelLoader.jl).  ch_va_eval evaluates one module instance on the GPU at given node voltages:
 * st[144] = [I(8) | Q(8) | dI/dV (8x8) | dQ/dV (8x8)] (stamp-level parity entry point, like ch_mos_eval). ---- */
int32_t ch_va_n_modules(void);
int32_t ch_va_find(const char* module_name);                                  /* id or -1 */
const char* ch_va_module_name(int32_t id);
int32_t ch_va_module_info(int32_t id, int32_t* n_ports, int32_t* n_nodes, int32_t* n_params);
const char* ch_va_node_name(int32_t id, int32_t k);
const char* ch_va_param_name(int32_t id, int32_t k);
int ch_va_eval(ch_ctx*, int32_t id, const double* par_and_given, const double* v_nodes, double temperature_k, double gmin,
               double* st_out);
/* Operating-point observables of a module: the variables declared with a (* desc = "..." *) attribute
 * (src/vasim.jl:742-753, 841-843: `observed!(var, DScope(dscope, name))`), evaluated on the GPU at given node voltages. */
int32_t ch_va_n_opvars(int32_t id);
const char* ch_va_opvar_name(int32_t id, int32_t k);
int ch_va_opvars(ch_ctx*, int32_t id, const double* par_and_given, const double* v_nodes, double temperature_k, double gmin,
                 double* op_out);

/* ---- library/kernel introspection ---- */
const char* ch_version(void);

/* ---- measurement utility ("report against ... the builder's own measured STREAM-triad").
 * a[i] = b[i] + s*c[i] over three fp64 arrays of n_doubles each, `iters` timed passes after one warm-up;
 * returns the best pass in GB/s (24 bytes per element) in *gbps_out.  No reference counterpart. ---- */
int ch_bench_triad(ch_ctx*, int64_t n_doubles, int32_t iters, double* gbps_out);
/* Vector fp64 FMA peak of this GPU, measured: 16 independent v_fma_f64 chains per lane, 8 waves per SIMD;
 * best of `iters` passes in TFLOP/s (the eval kernel's fp64 rate is reported against this). */
int ch_bench_fp64(ch_ctx*, int32_t iters, double* tflops_out);

/* ---- test hook: fill the LDS of every CU with a NaN / negative-integer pattern (LDS keeps what the previous kernel left; a
 * kernel that reads LDS it did not stage must not get away with a fresh process's zeros).  No reference counterpart. ---- */
int ch_debug_poison_lds(ch_ctx*);
/* Test hook: y[i] = f(x[i]) with the DEVICE's own fp64 functions (hardware seeds + polynomial / Newton steps instead of the library
 * sequences: cedarsim.jl_amd/csrc/va_rt.hpp, ch_bsim4.hpp) — which = 0: exp as compiled Verilog-A code calls it, 1: ln likewise
 * (incl. the special values), 2: the ln of the BSIM4 device code.  Host arrays of n doubles. */
int ch_debug_math(ch_ctx*, int32_t which, int32_t n, const double* x, double* y);

#ifdef __cplusplus
}
#endif
#endif /* CEDARHIP_H */
