"""Secondary measurements on one MI355X (not the driver's bench line): SURVEY §8(d) config 4 (8192 Monte-Carlo samples
of one DFF, one batched solve) and config 5 (128 BSIM-CMG inverters, ASAP7 TT cards).  Prints one JSON object."""
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
from cedarsim_jl_amd.workloads import CMG_TSPAN, DFF_CHECK_Q, DFF_CHECK_TIMES, cmg_inverter_array, dff_array  # noqa: E402

out = {}
only = sys.argv[1] if len(sys.argv) > 1 else "all"
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
    opts = tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14), saveat=np.array(DFF_CHECK_TIMES))
    e.tran(0.0, 7e-7, opts)
    t0 = time.perf_counter()
    rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
    el = time.perf_counter() - t0
    ok = np.abs(v[0] - np.array(DFF_CHECK_Q)[:, None]) < 1e-3
    out["config4_mc8192"] = {"rc": rc, "samples": S, "wall_seconds": el, "launches": st["n_kernel_launches"],
                             "avg_launch_us": 1e6 * st["device_seconds"] / max(1, st["n_kernel_launches"]),
                             "block_iterations": st["n_block_iters"], "block_iterations_per_second": st["n_block_iters"] / el,
                             "samples_passing_reference_gate": int(ok.all(axis=0).sum())}
if only in ("all", "config5"):
    cards = json.load(open(os.path.join(ROOT, "tests", "golden", "asap7_tt_lvt_cards.json")))["cards"]
    c = cmg_inverter_array(128, cards)
    e = EngineCircuit(c)
    opts = tran_opts(abstol=1e-7, reltol=1e-7, dc=dc_opts(abstol=1e-10, tran_mode=1))
    e.tran(CMG_TSPAN[0], CMG_TSPAN[1], opts)
    t0 = time.perf_counter()
    rc, t, v, xf, st = e.tran(CMG_TSPAN[0], CMG_TSPAN[1], opts)
    el = time.perf_counter() - t0
    out["config5_bsimcmg_x256"] = {"rc": rc, "inverters": 128, "bsimcmg_instances": 256, "wall_seconds": el, "accepted": st["naccept"], "rejected": st["nreject"],
                                   "launches": st["n_kernel_launches"], "avg_launch_us": 1e6 * st["device_seconds"] / max(1, st["n_kernel_launches"]),
                                   "newton_iters_per_sec": st["nnonliniter"] / el, "block_iterations": st["n_block_iters"]}
print(json.dumps(out))
