"""Coupled DFF array (non-ideal rails): transient through the torn (bordered block-diagonal) form on the device-resident stepper
against the sparse path of the same circuit.  Usage: python scripts/torn_check.py [tiles ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cedarsim_jl_amd import dc_opts, tran_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit, load_library  # noqa: E402
from cedarsim_jl_amd.workloads import DFF_CHECK_Q, DFF_CHECK_TIMES, dff_array  # noqa: E402

load_library()
for tiles in [int(x) for x in (sys.argv[1:] or ["8"])]:
    c = dff_array(tiles, observe="q0", supply_r=1.0)
    c.observe_node("vdd")
    c.observe_node("vss")
    e = EngineCircuit(c)
    sv = np.linspace(0.0, 7e-7, 141)
    res = {}
    for label, env in (("torn", None), ("sparse", "1")):
        if label == "sparse" and tiles > 256:
            continue
        if env:
            os.environ["CEDARHIP_NO_TEAR"] = env
        else:
            os.environ.pop("CEDARHIP_NO_TEAR", None)
        opts = tran_opts(abstol=1e-4, reltol=1e-4, saveat=sv, dc=dc_opts(abstol=1e-12))
        t0 = time.perf_counter()
        rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
        el = time.perf_counter() - t0
        print(tiles, label, "rc", rc, e.ctx.last_error() if rc else "", "stepper", st["stepper"], "wall %.4f s" % el, "dc %.4f" % st["dc_seconds"], "acc/rej/fail", st["naccept"], st["nreject"], st["nnonlinconvfail"],
              "iters", st["nnonliniter"], "attempts", st["n_step_attempts"], "launches", st["n_kernel_launches"], flush=True)
        if rc == 0:
            res[label] = v
            q = [float(np.interp(tt, t, v[0, :, 0])) for tt in DFF_CHECK_TIMES]
            print("   q at the check times", np.round(q, 5), "rails min/max", v[1].min(), v[1].max(), v[2].min(), v[2].max())
    os.environ.pop("CEDARHIP_NO_TEAR", None)
    if len(res) == 2:
        print("   max |torn - sparse| per observable:", [float(np.max(np.abs(res["torn"][k] - res["sparse"][k]))) for k in range(3)])
