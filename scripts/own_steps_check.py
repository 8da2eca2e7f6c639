"""Per-block step acceptance for ONE circuit of independent blocks on a saveat grid: the tiled DFF array with per-tile clock skew
(1024 private clock sources) on the device-resident stepper against the lock-step host stepper."""
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
tol = float(os.environ.get("TOL", "1e-4"))
for tiles in [int(x) for x in (sys.argv[1:] or ["16"])]:
    rng = np.random.default_rng(1234)
    c = dff_array(tiles, skew=rng.uniform(0.0, 50e-12, tiles), observe="q")
    e = EngineCircuit(c)
    sv = np.linspace(0.0, 7e-7, 141)
    res = {}
    for label, env in (("own", None), ("lockstep", "1")):
        if env:
            os.environ["CEDARHIP_LOCKSTEP"] = env
        else:
            os.environ.pop("CEDARHIP_LOCKSTEP", None)
        opts = tran_opts(abstol=tol, reltol=tol, saveat=sv, dc=dc_opts(abstol=1e-14))
        e.tran(0.0, 7e-7, opts)
        t0 = time.perf_counter()
        rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
        el = time.perf_counter() - t0
        print(tiles, label, "rc", rc, e.ctx.last_error() if rc else "", "stepper", st["stepper"], "wall %.4f s" % el, "dc %.4f" % st["dc_seconds"], "acc/rej/fail", st["naccept"], st["nreject"], st["nnonlinconvfail"],
              "iters", st["nnonliniter"], "block iters", st["n_block_iters"], "attempts", st["n_step_attempts"], "launches", st["n_kernel_launches"], flush=True)
        if rc == 0:
            res[label] = v
            q = np.array([[np.interp(tt, t, v[k, :, 0]) for tt in DFF_CHECK_TIMES] for k in range(v.shape[0])])
            print("   worst gate deviation", float(np.max(np.abs(q - np.array(DFF_CHECK_Q)[None, :]))))
    os.environ.pop("CEDARHIP_LOCKSTEP", None)
    if len(res) == 2:
        print("   max |own - lockstep|:", float(np.max(np.abs(res["own"] - res["lockstep"]))))
