#!/usr/bin/env python3
"""Per-attempt cost of the two step controllers on the DFF array: python scripts/stepper_profile.py [tiles ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedarsim_jl_amd import dc_opts, tran_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from cedarsim_jl_amd.workloads import DFF_TSPAN, dff_array  # noqa: E402

tiles = [int(x) for x in sys.argv[1:]] or [1, 256, 1024]
out = []
for n in tiles:
    e = EngineCircuit(dff_array(n, observe="q0"))
    for stepper in ("host", "device"):
        o = tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14), stepper=stepper)
        e.tran(*DFF_TSPAN, o)
        t0 = time.perf_counter()
        rc, t, v, xf, st = e.tran(*DFF_TSPAN, o)
        el = time.perf_counter() - t0
        att = max(1, st["n_step_attempts"])
        out.append({"tiles": n, "stepper": stepper, "rc": rc, "wall_ms": 1e3 * el, "attempts": att, "naccept": st["naccept"], "nreject": st["nreject"],
                    "us_per_attempt_wall": 1e6 * el / att, "device_us_per_attempt": 1e6 * st["device_seconds"] / att,
                    "barrier_us_per_attempt": 1e6 * st["barrier_seconds"] / att, "iters": st["nnonliniter"], "block_iters": st["n_block_iters"],
                    "launches": st["n_kernel_launches"]})
        print(json.dumps(out[-1]), flush=True)
