"""Secondary measurements on one MI355X (not the driver's bench line): SURVEY §8(d) config 4 (Monte-Carlo samples of one
DFF as one batched solve: 8192 on one GPU, and the 1024-sample share one GPU gets on an 8-GPU node, on both step controllers),
config 5 (128 BSIM-CMG inverters, ASAP7 TT cards) and the COUPLED form of config 3 (supply rails behind 1-ohm resistors: one
Jacobian block, sparse path).  Prints one JSON object; pass a config name to run only that one."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cedarsim_jl_amd import bsim4_params as B4  # noqa: E402
from cedarsim_jl_amd import dc_opts, tran_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from cedarsim_jl_amd.workloads import CMG_TSPAN, DFF_CHECK_Q, DFF_CHECK_TIMES, cmg_inverter_array, dff_array, inverter  # noqa: E402

out = {}
only = sys.argv[1] if len(sys.argv) > 1 else "all"
if only in ("all", "config1", "config2"):
    # config 1 (the reference's plumbing case: one CMOS inverter, DC + transient, gate at 50/150/250/350 ns) and config 2 (one DFF)
    # are latency cases: one Jacobian block = one wavefront (pair) of the chip; listed for completeness, on both step controllers
    for name, ckt, tspan, gate_t, gate_v, tol in (("config1_inverter", inverter(), (0.0, 4e-7), (50e-9, 150e-9, 250e-9, 350e-9), (5.0, 0.0, 5.0, 0.0), 1e-8),
                                                  ("config2_single_dff", dff_array(1, observe="q0"), (0.0, 7e-7), DFF_CHECK_TIMES, DFF_CHECK_Q, 1e-4)):
        e = EngineCircuit(ckt)
        entry = {}
        for stepper in ("host", "device"):
            opts = tran_opts(abstol=tol, reltol=tol, dc=dc_opts(abstol=1e-14), stepper=stepper)
            e.tran(tspan[0], tspan[1], opts)
            t0 = time.perf_counter()
            rc, t, v, xf, st = e.tran(tspan[0], tspan[1], opts)
            el = time.perf_counter() - t0
            q = [float(np.interp(tt, t, v[0, :, 0])) for tt in gate_t]
            entry[stepper] = {"rc": rc, "wall_seconds": el, "dc_seconds": st["dc_seconds"], "accepted": st["naccept"], "rejected": st["nreject"], "newton_iters": st["nnonliniter"],
                              "newton_iters_per_sec": st["nnonliniter"] / el, "us_per_attempt": 1e6 * (el - st["dc_seconds"]) / max(1, st["n_step_attempts"]),
                              "launches": st["n_kernel_launches"], "gate_ok": bool(all(abs(a - b) <= max(10 * tol, 1e-3) for a, b in zip(q, gate_v)))}
        out[name] = entry
if only in ("all", "config4"):
    S = 8192
    c = dff_array(1)
    slots, names = [], []
    for m in ("nfet_06v0", "pfet_06v0"):
        for p in ("vth0", "u0", "toxe"):
            slots.append(c.slot(m, p))
            names.append((m, p))
    rng = np.random.default_rng(2024)
    base = np.array([c.models[c.model_names.index(m)][B4.PARAM_INDEX[p]] for m, p in names])
    vals = base[:, None] * (1.0 + 0.03 * rng.standard_normal((len(slots), S)))
    e = EngineCircuit(c)
    e.set_samples(S)
    e.set_params(slots, vals)
    e1 = EngineCircuit(dff_array(1))
    rc0, x_nom, _, _ = e1.dc(dc_opts(abstol=1e-14))
    res = {}
    # host: one launch per attempt, lock-step batch; device: per-sample step acceptance, 2048 queued workgroups of 4 samples;
    # device_dc_from_nominal: the same with every sample's DC started from the nominal operating point (CircuitSweep(warm_start=True))
    for label, stepper, dco in (("host", "host", dc_opts(abstol=1e-14)), ("device", "device", dc_opts(abstol=1e-14)),
                                ("device_dc_from_nominal", "device", dc_opts(abstol=1e-14, x0=np.tile(x_nom[0], (S, 1))))):
        opts = tran_opts(abstol=1e-4, reltol=1e-4, dc=dco, saveat=np.array(DFF_CHECK_TIMES), stepper=stepper)
        e.tran(0.0, 7e-7, opts)
        t0 = time.perf_counter()
        rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
        el = time.perf_counter() - t0
        ok = np.abs(v[0] - np.array(DFF_CHECK_Q)[:, None]) < 1e-3
        res[label] = {"rc": rc, "wall_seconds": el, "dc_seconds": st["dc_seconds"], "launches": st["n_kernel_launches"],
                      "block_iterations": st["n_block_iters"], "block_iterations_per_second": st["n_block_iters"] / el,
                      "samples_passing_reference_gate": int(ok.all(axis=0).sum())}
    out["config4_mc8192"] = {"samples": S, **res}
if only in ("all", "config4", "config4_sweep"):
    # config 4 THROUGH the reference's sweep surface: an explicit 8192-point Monte-Carlo TandemSweep of one DFF (src/sweeps.jl:278-290,
    # 473-480), CircuitSweep learns the name -> table-entry map from a handful of builds (setup reported apart) and runs ONE batched solve
    from cedarsim_jl_amd import CircuitSweep
    from cedarsim_jl_amd.workloads import dff_mc_builder, mc_tandem_sweep
    S = 8192
    build, names = dff_mc_builder(observe=("q",))
    t0 = time.perf_counter()
    sweep = mc_tandem_sweep(S)
    draw_s = time.perf_counter() - t0
    res = {}
    for label in ("first_call", "second_call"):
        cs = CircuitSweep(build, sweep)
        t0 = time.perf_counter()
        rc, t, rows, st = cs.tran_arrays((0.0, 7e-7), abstol=1e-4, reltol=1e-4, dc_abstol=1e-14, saveat=np.array(DFF_CHECK_TIMES))
        el = time.perf_counter() - t0
        ok = np.abs(rows[:, 0, :] - np.array(DFF_CHECK_Q)[None, :]) < 1e-3
        res[label] = {"rc": rc, "wall_seconds_incl_setup": el, "setup": cs.setup, "dc_seconds": st["dc_seconds"], "stepper": st["stepper"], "stepper_mode": st["stepper_mode"],
                      "block_iterations": st["n_block_iters"], "block_iterations_per_second_of_the_solve": st["n_block_iters"] / max(1e-12, el - cs.setup["seconds"]),
                      "samples_passing_reference_gate": int(ok.all(axis=1).sum())}
    out["config4_mc8192_through_CircuitSweep"] = {"samples": S, "swept_names": list(names), "draw_seconds": draw_s, **res}
if only in ("all", "config4_share"):
    S = 1024
    c = dff_array(1)
    slots, names = [], []
    for m in ("nfet_06v0", "pfet_06v0"):
        for p in ("vth0", "u0", "toxe"):
            slots.append(c.slot(m, p))
            names.append((m, p))
    rng = np.random.default_rng(2024)
    base = np.array([c.models[c.model_names.index(m)][B4.PARAM_INDEX[p]] for m, p in names])
    vals = base[:, None] * (1.0 + 0.03 * rng.standard_normal((len(slots), S)))
    e = EngineCircuit(c)
    e.set_samples(S)
    e.set_params(slots, vals)
    res = {}
    for stepper in ("host", "device"):
        opts = tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14), saveat=np.array(DFF_CHECK_TIMES), stepper=stepper)
        e.tran(0.0, 7e-7, opts)
        t0 = time.perf_counter()
        rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
        el = time.perf_counter() - t0
        ok = np.abs(v[0] - np.array(DFF_CHECK_Q)[:, None]) < 1e-3
        res[stepper] = {"rc": rc, "wall_seconds": el, "dc_seconds": st["dc_seconds"], "transient_seconds": el - st["dc_seconds"], "step_attempts": st["n_step_attempts"],
                        "us_per_attempt": 1e6 * (el - st["dc_seconds"]) / max(1, st["n_step_attempts"]),
                        "block_iterations_per_second": st["n_block_iters"] / el, "samples_passing_reference_gate": int(ok.all(axis=0).sum())}
    # the same batch with every sample's DC started from the nominal operating point instead of 1e-7*randn restarts
    e1 = EngineCircuit(dff_array(1))
    rc0, x_nom, _, _ = e1.dc(dc_opts(abstol=1e-14))
    opts = tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14, x0=np.tile(x_nom[0], (S, 1))), saveat=np.array(DFF_CHECK_TIMES), stepper="device")
    e.tran(0.0, 7e-7, opts)
    t0 = time.perf_counter()
    rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
    el = time.perf_counter() - t0
    ok = np.abs(v[0] - np.array(DFF_CHECK_Q)[:, None]) < 1e-3
    res["device_dc_from_nominal"] = {"rc": rc, "wall_seconds": el, "dc_seconds": st["dc_seconds"], "transient_seconds": el - st["dc_seconds"],
                                     "block_iterations_per_second": st["n_block_iters"] / el, "samples_passing_reference_gate": int(ok.all(axis=0).sum())}
    out["config4_share_mc1024"] = {"samples": S, "note": "the per-GPU share of the 8192-sample Monte-Carlo on an 8-GPU node", **res}
if only in ("all", "coupled"):
    # config 3 with non-ideal rails: ONE coupled block for the structural analysis.  Two solvers of the same Newton systems:
    #   torn   — bordered block-diagonal form on the device-resident stepper: a register LU per tile and a Schur complement on the
    #            two rail unknowns per Newton iteration (two grid-wide reductions), DC on the sparse path
    #   sparse — CSR assembly + level-synchronous LU refactor + solves, one launch per level (CEDARHIP_NO_TEAR=1)
    for tiles in [int(x) for x in os.environ.get("CEDARHIP_COUPLED_TILES", "64,256,1024").split(",")]:
        c = dff_array(tiles, observe="q0", supply_r=1.0)
        e = EngineCircuit(c)
        opts = tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-12), saveat=np.array(DFF_CHECK_TIMES))
        entry = {}
        for label in os.environ.get("CEDARHIP_COUPLED_FORMS", "torn,sparse").split(","):
            if label == "sparse":
                os.environ["CEDARHIP_NO_TEAR"] = "1"
            else:
                os.environ.pop("CEDARHIP_NO_TEAR", None)
            e.tran(0.0, 7e-7, opts)   # warm-up: symbolic analysis, first-launch costs
            t0 = time.perf_counter()
            rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
            el = time.perf_counter() - t0
            info = e.info()
            nnz, nlu, n = info["nnz_jac"], info["nnz_lu"], info["n_unknowns"]
            # algorithmic bytes per Newton iteration: eval 400 B + assembly 288 B per MOSFET + 8(nnz + n); LU 8(nnz + 2 nnz(L+U)); solves 8(2 nnz(L+U) + 4 n)
            bpi = tiles * 30 * 688 + 8 * (nnz + n) + 8 * (nnz + 2 * nlu) + 8 * (2 * nlu + 4 * n)
            tran_s = el - st["dc_seconds"]
            tran_iters = st["step_block_iters"] / tiles if label == "torn" else st["nnonliniter"]
            entry[label] = {"rc": rc, "stepper": st["stepper"], "wall_seconds": el, "dc_seconds": st["dc_seconds"], "transient_seconds": tran_s,
                            "accepted": st["naccept"], "rejected": st["nreject"], "newton_iters": st["nnonliniter"], "step_attempts": st["n_step_attempts"],
                            "newton_iters_per_sec": st["nnonliniter"] / el, "us_per_attempt_transient": 1e6 * tran_s / max(1, st["n_step_attempts"]),
                            "launches": st["n_kernel_launches"], "achieved_GBps_wall": bpi * st["nnonliniter"] / el / 1e9,
                            "gate_q": [float(v[0, k, 0]) for k in range(len(DFF_CHECK_TIMES))] if rc == 0 else None}
        os.environ.pop("CEDARHIP_NO_TEAR", None)
        out["config3_coupled_%d_tiles" % tiles] = {"path": info["path"], "unknowns": n, "blocks": info["n_components"], "nnz_jacobian": nnz, "nnz_lu": nlu,
                                                    "algorithmic_bytes_per_iteration": bpi, **entry}
if only in ("all", "config5"):
    cards = json.load(open(os.path.join(ROOT, "cedarsim.jl_amd", "data", "asap7_tt_lvt_cards.json")))["cards"]
    c = cmg_inverter_array(128, cards)
    e = EngineCircuit(c)
    opts = tran_opts(abstol=1e-7, reltol=1e-7, dc=dc_opts(abstol=1e-10, tran_mode=1))
    e.tran(CMG_TSPAN[0], CMG_TSPAN[1], opts)
    t0 = time.perf_counter()
    rc, t, v, xf, st = e.tran(CMG_TSPAN[0], CMG_TSPAN[1], opts)
    el = time.perf_counter() - t0
    # roofline of the dominant kernel (T = 6 terminals -> 792 B per BSIM-CMG instance and evaluation, 2 instances per block iteration)
    kb = 2 * 792 * st["step_block_iters"]
    out["config5_bsimcmg_x256"] = {"rc": rc, "inverters": 128, "bsimcmg_instances": 256, "wall_seconds": el, "accepted": st["naccept"], "rejected": st["nreject"],
                                   "launches": st["n_kernel_launches"], "avg_launch_us": 1e6 * st["step_kernel_seconds"] / max(1, st["step_kernel_launches"]),
                                   "newton_iters_per_sec": st["nnonliniter"] / el, "block_iterations": st["n_block_iters"],
                                   "roofline": {"bound": "hbm", "algorithmic_bytes_per_instance_evaluation": 792, "achieved_GBps": kb / max(1e-12, st["step_kernel_seconds"]) / 1e9,
                                                "frac_of_8TBps": kb / max(1e-12, st["step_kernel_seconds"]) / 8e12,
                                                "limiter": "single-wave instruction issue of the generated dual-number code: about 17.5 k instructions per evaluation half and Newton iteration (12 k VALU, 5.6 k of them fp64; profiles/r03_cfg5_wait_counters.json) at about 7 cycles each; 128 blocks on 64 CUs, one persistent launch"}}
print(json.dumps(out))
