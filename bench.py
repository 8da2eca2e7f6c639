#!/usr/bin/env python3
"""bench.py — Newton iterations/sec and wall-clock per transient on the GF180 DFF array.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
`python -m torch.distributed.run --nproc-per-node N …`, one rank per GPU.

Workload (BASELINE.json metric / SURVEY §8(d) config 3): tiled array of 1024 GF180 D-flip-flops
(30 720 BSIM4 MOSFETs, 11 264 unknowns after structural reduction = 1024 independent 11x11 Jacobian
blocks), full transient 0..700 ns with the reference test bench's CLKN/D stimuli,
abstol = reltol = 1e-4 (benchmarks/gf180_dff_solver_bench.jl:27,62), DC operating point included.
A "step" is ONE such transient.  With N GPUs every rank integrates its own process-variation sample
of the array (weak scaling, no data-path collective; the per-rank Q waveforms at the reference's
check times and the iteration counts are gathered once at the end over RCCL).

value = Newton iterations (of the whole array, i.e. max over blocks per step attempt, summed over
ranks) / wall seconds, with the circuit description already resident on the GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TILES = 1024
TOL = 1e-4
MOS_PER_TILE = 30


def algorithmic_bytes_per_block_iteration(nc, n_mos):
    """eval 400 B/instance (T=4), assembly 288 B/instance + 8(nnz(A)+n),
    LU refactor 8(nnz(A)+2 nnz(L+U)) + solve 8(2 nnz(L+U)+4n); dense block: nnz = nc^2."""
    nnz = nc * nc
    return n_mos * 400 + n_mos * 288 + 8 * (nnz + nc) + 8 * (nnz + 2 * nnz) + 8 * (2 * nnz + 4 * nc)


def _oracle_worker(args):
    """One host thread: full transients of one decoupled DFF tile until the deadline (ctypes releases the GIL)."""
    deadline, max_reps = args[0], args[1]
    proxy = len(args) > 2 and args[2]
    from oracle_binding import Oracle
    from cedarsim_jl_amd import dc_opts, tran_opts
    from cedarsim_jl_amd.workloads import DFF_TSPAN, dff_array
    o = Oracle(dff_array(1))
    o.set_proxy(proxy)
    iters = reps = 0
    while True:
        rc, t, v, xf, st = o.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(abstol=TOL, reltol=TOL, dc=dc_opts(abstol=1e-14)))
        assert rc == 0
        iters += st["nnonliniter"]
        reps += 1
        if time.perf_counter() > deadline or reps >= max_reps:
            break
    return iters, reps


def cpu_baseline(seconds_single=5.0, seconds_multi=10.0, seconds_proxy=5.0):
    """Oracle ("port") timed on the host cores on a bounded sample: full transients of single decoupled tiles
    (1 of the 1024), first on one core, then one tile per thread on every core this process may use;
    converted to array-level iterations/s by dividing the tile rate by the tile count."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    t0 = time.perf_counter()
    it1, reps1 = _oracle_worker((t0 + seconds_single, 200))
    el1 = time.perf_counter() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(_oracle_worker, [(t0 + seconds_multi, 400)] * cores))
    elm = time.perf_counter() - t0
    itm, repsm = sum(r[0] for r in res), sum(r[1] for r in res)
    tile_rate_1, tile_rate_m = it1 / el1, itm / elm
    # B0: "reference-like" cost proxy on one core — finite-difference Jacobian from n+1 residuals, reused like
    # IDA's modified Newton (src/dcop.jl:28,53-94; benchmarks/gf180_dff_solver_bench.jl:60-81), one tile, scaled by the tile
    # count.  Flattering to the reference: its dense LU is O(n^3) of the WHOLE 13.3k system, here it is 25x25 per tile.
    t0 = time.perf_counter()
    itp, repsp = _oracle_worker((t0 + seconds_proxy, 200, True))
    elp = time.perf_counter() - t0
    proxy = {"kind": "proxy, not CedarSim", "value": itp / elp / N_TILES, "unit": "newton_iters/s (1024-DFF array equivalent)", "cores": 1,
             "seconds_per_tile_transient": elp / repsp, "seconds_per_array_transient_scaled": elp / repsp * N_TILES,
             "sample": "%d full transients of one decoupled tile with a finite-difference Jacobian (26 residuals each) reused across "
                       "iterations and steps, %.1f s" % (repsp, elp)}
    return {"value": tile_rate_m / N_TILES, "reference_like_proxy_B0": proxy, "unit": "newton_iters/s (1024-DFF array equivalent)", "cores": cores, "kind": "port",
            "sample": "%d full transients of single decoupled DFF tiles (1 of the %d; dense-LU MNA oracle, n=25) on %d threads, "
                      "%.1f s; tile rate %.0f iters/s divided by %d" % (repsm, N_TILES, cores, elm, tile_rate_m, N_TILES),
            "single_core_value": tile_rate_1 / N_TILES, "seconds_per_tile_transient_single_core": el1 / reps1}


def combine_ranks(dist, world, device, iters, el, gate_ok, q):
    """The only collective of the bench: one all_gather of a small per-rank vector AFTER the timed region (RCCL over xGMI when
    `device` is a GPU, gloo on the CPU rehearsal).  Returns whole-job iterations (sum over ranks), the slowest rank's time
    (max), the AND of the ranks' gates, and every rank's gate values."""
    import torch
    if dist is None or world <= 1:
        return float(iters), el, bool(gate_ok), [list(q)]
    buf = torch.tensor([float(iters), el, float(gate_ok)] + list(q), dtype=torch.float64, device=device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)  # result gather only
    res = torch.stack(out).cpu().numpy()
    return float(res[:, 0].sum()), float(res[:, 1].max()), bool(res[:, 2].min() > 0.5), res[:, 3:].tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=N_TILES, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-skew", action="store_true", help="skip the second, clearly labelled measurement with per-tile clock skew")
    ap.add_argument("--stepper", default="auto", choices=["auto", "host", "device"], help=argparse.SUPPRESS)
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU fallback")
    # CEDARHIP_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices, the result
    # gather goes through host memory); the driver's multi-GPU runs use the default: one GPU per rank, RCCL over xGMI
    backend = os.environ.get("CEDARHIP_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from cedarsim_jl_amd import dc_opts, tran_opts
    from cedarsim_jl_amd.engine import Context, EngineCircuit
    from cedarsim_jl_amd.workloads import DFF_CHECK_Q, DFF_CHECK_TIMES, DFF_TSPAN, dff_array

    ctx = Context(local_rank)
    ckt = dff_array(args.tiles, observe="q0")
    # one process-variation sample per rank (seed 2024 + rank): multipliers on vth0/u0 of both cards
    if world > 1:
        rng = np.random.default_rng(2024 + rank)
        slots, vals = [], []
        from cedarsim_jl_amd import bsim4_params as B4
        for mname in ("nfet_06v0", "pfet_06v0"):
            for par in ("vth0", "u0"):
                bv = ckt.models[ckt.model_names.index(mname)][B4.PARAM_INDEX[par]]
                slots.append(ckt.slot(mname, par))
                vals.append([bv * (1.0 + 0.03 * rng.standard_normal())])
    eng = EngineCircuit(ckt, ctx)
    if world > 1:
        eng.set_samples(1)
        eng.set_params(slots, vals)
    info = eng.info()
    opts = tran_opts(abstol=TOL, reltol=TOL, dc=dc_opts(abstol=1e-14), stepper=args.stepper)

    def one_transient(eng=eng):
        rc, t, v, xf, st = eng.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts)
        if rc != 0:
            raise SystemExit("transient failed: rc=%d %s" % (rc, ctx.last_error()))
        q = [float(np.interp(tt, t, v[0, :, 0])) for tt in DFF_CHECK_TIMES]
        return st, q

    for _ in range(args.warmup):
        one_transient()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    iters = block_iters = launches = attempts = 0
    dev_s = dc_s = bar_s = 0.0
    naccept = nreject = 0
    q = None
    stepper_used = 0
    for _ in range(args.steps):
        st, q = one_transient()
        iters += st["nnonliniter"]
        block_iters += st["step_block_iters"]          # inside the time-stepping kernel(s): what the roofline line is about
        launches += st["step_kernel_launches"]
        dev_s += st["step_kernel_seconds"]
        dc_s += st["dc_seconds"]
        bar_s += st["barrier_seconds"]
        attempts += st["n_step_attempts"]
        naccept += st["naccept"]
        nreject += st["nreject"]
        stepper_used = st["stepper"]
    barrier()
    el = time.perf_counter() - t0

    # correctness gate of the reference harness (benchmarks/gf180_dff_solver_bench.jl:84-96)
    gate_ok = all(abs(a - b) <= 10 * TOL for a, b in zip(q, DFF_CHECK_Q))

    tot_iters, max_el, gate_ok, all_q = combine_ranks(dist, world, "cuda" if backend == "nccl" else "cpu", iters, el, gate_ok, q)

    if rank == 0:
        nc, n_mos = info["max_component"], MOS_PER_TILE
        bpi = algorithmic_bytes_per_block_iteration(nc, n_mos)
        avg_launch = dev_s / max(1, launches)
        bytes_per_launch = bpi * block_iters / max(1, launches)
        achieved = bytes_per_launch / avg_launch / 1e9 if avg_launch > 0 else 0.0
        # HBM traffic and the fp64 instruction mix come from separate rocprofv3 --pmc passes (scripts/profile_round.sh); they
        # are quoted only when that file was measured on THIS build of the library (hash of the .so) and this stepper
        traffic = flops = None
        pmc_note = "no PMC file for this build"
        import hashlib
        from cedarsim_jl_amd import engine as _eng
        lib_hash = hashlib.sha256(open(_eng.LIB_PATH, "rb").read()).hexdigest()
        for cand in sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")):
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            except Exception:  # noqa: BLE001
                continue
            if pj.get("lib_sha256") == lib_hash and pj.get("tiles", N_TILES) == args.tiles and "bench.py" in pj.get("workload", "bench.py") \
                    and pj.get("kernel", "") in ("tran_persistent_kernel", "newton_block_kernel"):
                traffic = pj.get("hbm_bytes_per_launch")
                flops = pj.get("fp64_flop_per_launch")
                pmc_note = "profiles/%s (same library build, sha256 %s...)" % (cand, lib_hash[:12])
        try:
            triad, fp64_peak = ctx.triad_gbps(), ctx.fp64_tflops()
        except RuntimeError:
            triad = fp64_peak = None
        line = {
            "metric": "newton_iters_per_sec", "value": tot_iters / max_el, "unit": "newton_iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * max_el / max(1, args.steps),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "gf180_dff_array_%d tiles transient 0..700ns abstol=reltol=1e-4 (SURVEY 8d config 3)" % args.tiles,
                       "tiles": args.tiles, "mosfets": args.tiles * MOS_PER_TILE, "unknowns": info["n_unknowns"],
                       "blocks": info["n_components"], "block_size": nc, "device_cards": "substitute BSIM4 cards (GF180MCUPDK unavailable)",
                       "wall_seconds_per_transient": max_el / max(1, args.steps),
                       "dc_seconds_per_transient": dc_s / max(1, args.steps),
                       "tile_newton_iters_per_sec": tot_iters * args.tiles / max_el,
                       "accepted_steps": naccept // max(1, args.steps), "rejected_steps": nreject // max(1, args.steps),
                       "reference_gate_q": all_q[0], "reference_gate_ok": gate_ok,
                       "step_controller": "device-resident (one persistent cooperative launch per transient)" if stepper_used == 2 else "host (one launch per attempt)",
                       "multi_gpu": "independent process-variation samples per rank; all_gather of results only"},
            "roofline": {"bound": "hbm", "kernel": "tran_persistent_kernel (one launch per transient)" if stepper_used == 2 else "newton_block_kernel (one launch per step attempt)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": pmc_note,
                         "measured_hbm_gbps": (traffic / avg_launch / 1e9) if (traffic and avg_launch > 0) else None,
                         "limiter": "fp64 VALU issue + dependency latency (one wave per SIMD); HBM is the stated bound of the path, not the limiter here",
                         "step_attempts_per_launch": attempts / max(1, launches), "us_per_step_attempt": 1e6 * dev_s / max(1, attempts),
                         "grid_reduction_us_per_attempt": 1e6 * bar_s / max(1, attempts),
                         "algorithmic_bytes_per_block_iteration": bpi, "block_iterations_per_launch": block_iters / max(1, launches),
                         "avg_launch_us": 1e6 * avg_launch, "launches": launches,
                         "peak_measured_triad": triad, "frac_of_measured_triad": (achieved / triad) if triad else None,
                         "fp64_tflops_achieved": (flops / avg_launch / 1e12) if (flops and avg_launch > 0) else None,
                         "fp64_vector_peak_tflops_measured": fp64_peak,
                         "note": "stamps and the block Jacobian stay in LDS, so HBM is not the limiter; the kernel is "
                                 "fp64-VALU/latency bound (see DESIGN.md)"},
        }
        # The two variants below are separate, clearly labelled measurements; a failure inside one of them is reported in its field
        # and never costs the headline line.
        if world == 1 and not args.no_skew and args.tiles == N_TILES:
            try:
                # SURVEY 8(d) config 3 variant: per-tile clock skew U(0, 50 ps), seed 1234 — every tile has its own clock source, the
                # tiles stop being bit-identical (the worst case for one shared step size).  A second, separate measurement: NOT `value`.
                rngs = np.random.default_rng(1234)
                ck = dff_array(args.tiles, skew=rngs.uniform(0.0, 50e-12, args.tiles), observe="q")
                es = EngineCircuit(ck, ctx)
                # output on a saveat grid (the gate times): independent blocks then take their own steps — no tile pays for the
                # other 1023 clocks' corners (lock-step, every accepted step saved: 62 292 steps, 4.3 s; profiles/r02_notes.md)
                opts_s = tran_opts(abstol=TOL, reltol=TOL, dc=dc_opts(abstol=1e-14), stepper=args.stepper, saveat=np.array(DFF_CHECK_TIMES))
                es.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts_s)
                t0s = time.perf_counter()
                rc_s, t_s, v_s, _, st_s = es.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts_s)
                el_s = time.perf_counter() - t0s
                qs = np.array([[np.interp(tt, t_s, v_s[k, :, 0]) for tt in DFF_CHECK_TIMES] for k in range(v_s.shape[0])])
                # for comparison, ONE run with every accepted step of ONE step sequence saved (what a caller without saveat gets):
                # 1024 private clocks are more than the device-resident controller's 64 lock-step sources, so this is the host stepper
                t0l = time.perf_counter()
                rc_l, t_l, v_l, _, st_l = es.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts)
                el_l = time.perf_counter() - t0l
                line["skewed_clock_variant"] = {"workload": "same array, per-tile clock skew U(0,50 ps) seed 1234 (%d private clock sources)" % args.tiles,
                                                "rc": rc_s, "ms_per_transient": 1e3 * el_s, "newton_iters_per_sec": st_s["nnonliniter"] / el_s,
                                                "accepted_steps": st_s["naccept"], "rejected_steps": st_s["nreject"], "step_attempts": st_s["n_step_attempts"],
                                                "step_controller": "device-resident, per-block step acceptance on the saveat grid of the gate times" if st_s["stepper"] == 2 else "host, lock-step",
                                                "block_iterations": st_s["n_block_iters"],
                                                "lockstep_every_step_saved": {"rc": rc_l, "ms_per_transient": 1e3 * el_l, "accepted_steps": st_l["naccept"], "rejected_steps": st_l["nreject"],
                                                                              "block_iterations": st_l["n_block_iters"], "step_controller": "device-resident" if st_l["stepper"] == 2 else "host"},
                                                "every_tile_meets_reference_gate": bool(np.max(np.abs(qs - np.array(DFF_CHECK_Q)[None, :])) <= 10 * TOL)}
            except Exception as ex:  # noqa: BLE001
                line["skewed_clock_variant"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world == 1 and not args.no_skew and args.tiles == N_TILES:
            try:
                # The same array behind NON-IDEAL rails (1 ohm in series with VDD and VSS): structurally ONE coupled block of 11 266
                # unknowns — "assembly + LU refactor" of BASELINE.json config 3 taken literally.  The engine tears it at the two rail
                # unknowns: register LU per tile + Schur complement on the rails inside the device-resident stepper, DC on the sparse
                # path.  A third, separate measurement: NOT `value`.
                cc = dff_array(args.tiles, observe="q", supply_r=1.0)
                ec = EngineCircuit(cc, ctx)
                # DC tolerance 1e-12 A: a rail row sums the currents of 15 360 MOSFET terminals, its residual has a rounding floor near 1e-13
                # (output on the gate times: with every accepted step of 1024 observables saved, 9 MB of rows per transient cross PCIe)
                opts_c = tran_opts(abstol=TOL, reltol=TOL, dc=dc_opts(abstol=1e-12), stepper=args.stepper, saveat=np.array(DFF_CHECK_TIMES))
                ec.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts_c)
                t0c = time.perf_counter()
                rc_c, t_c, v_c, _, st_c = ec.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts_c)
                el_c = time.perf_counter() - t0c
                qc = np.array([[np.interp(tt, t_c, v_c[k, :, 0]) for tt in DFF_CHECK_TIMES] for k in range(v_c.shape[0])])
                ic = ec.info()
                line["coupled_rails_variant"] = {"workload": "same array, 1 ohm in series with the VDD and VSS sources (one coupled block, %d unknowns, nnz(J) %d)" % (ic["n_unknowns"], ic["nnz_jac"]),
                                                 "rc": rc_c, "ms_per_transient": 1e3 * el_c, "dc_ms": 1e3 * st_c["dc_seconds"], "newton_iters_per_sec": st_c["nnonliniter"] / el_c,
                                                 "accepted_steps": st_c["naccept"], "rejected_steps": st_c["nreject"], "step_attempts": st_c["n_step_attempts"],
                                                 "solver": "torn at the rails: register LU per tile + Schur complement, device-resident stepper" if st_c["stepper"] == 2 else "sparse path (level-synchronous LU), host stepper",
                                                 "every_tile_meets_reference_gate": bool(np.max(np.abs(qc - np.array(DFF_CHECK_Q)[None, :])) <= 10 * TOL)}
            except Exception as ex:  # noqa: BLE001
                line["coupled_rails_variant"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if not args.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as ex:  # noqa: BLE001  (the oracle library is test infrastructure: its absence must not cost the GPU line)
                line["cpu_baseline"] = {"value": None, "unit": "newton_iters/s", "cores": 0, "kind": "port", "sample": "not measured", "error": "%s: %s" % (type(ex).__name__, ex)}
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
