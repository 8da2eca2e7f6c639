"""Extended fuzz of the bordered (torn) form against the sparse path (not part of the suite): the seeded loop of
tests/test_gpu_torn.py::test_torn_form_differential_fuzz with many more seeds, tile counts up to 200 (so that the sparse side runs
the subtree form too).  usage: python scripts/extended_fuzz_torn.py [first_seed] [n_seeds] [seconds]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from cedarsim_jl_amd import dc_opts, tran_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from cedarsim_jl_amd.workloads import dff_array  # noqa: E402
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 100
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
t_start = time.time()
fails, done = [], 0
sv = np.linspace(0.0, 3e-7, 61)
for seed in range(first, first + n_seeds):
    if time.time() - t_start > budget:
        break
    if (seed - first) % 20 == 0:
        print("progress: seed %d, %d compared, %d failures, %.0f s" % (seed, done, len(fails), time.time() - t_start), flush=True)
    rng = np.random.default_rng(seed)
    tiles = int(rng.integers(5, 200))
    r1 = float(10.0 ** rng.uniform(-1.5, 2.0))
    r2 = float(10.0 ** rng.uniform(-1.5, 2.0)) if rng.random() < 0.7 else None
    caps = rng.random() < 0.5
    tol = float(rng.choice([1e-4, 1e-5, 1e-6]))
    c = dff_array(tiles, observe="q", supply_r=(r1, r2))
    for n in ("vdd", "vss"):
        c.observe_node(n)
    if caps:
        c.C("cd1", "vdd", 0, 1e-12)
        if r2 is not None:
            c.C("cd2", "vdd", "vss", 5e-13)
    tag = (seed, tiles, round(r1, 3), None if r2 is None else round(r2, 3), caps, tol)
    try:
        e = EngineCircuit(c)
        os.environ.pop("CEDARHIP_NO_TEAR", None)
        os.environ["CEDARHIP_TORN_DC_SPARSE"] = "1"
        try:
            rc, t, v, xf, st = e.tran(0.0, 3e-7, tran_opts(abstol=tol, reltol=tol, saveat=sv, dc=dc_opts(abstol=1e-12)))
        finally:
            del os.environ["CEDARHIP_TORN_DC_SPARSE"]
        os.environ["CEDARHIP_NO_TEAR"] = "1"
        try:
            rc2, t2, v2, xf2, st2 = e.tran(0.0, 3e-7, tran_opts(abstol=tol, reltol=tol, saveat=sv, dc=dc_opts(abstol=1e-12)))
        finally:
            del os.environ["CEDARHIP_NO_TEAR"]
        if rc != 0 or rc2 != 0:
            fails.append((tag, "rc", rc, rc2, e.ctx.last_error()[:80]))
        elif st["stepper_mode"] != 3 or st2["stepper"] != 1:
            fails.append((tag, "path", st["stepper_mode"], st2["stepper"]))
        elif abs(st["naccept"] - st2["naccept"]) > 2 or abs(st["nreject"] - st2["nreject"]) > 2:
            fails.append((tag, "steps", st["naccept"], st2["naccept"], st["nreject"], st2["nreject"], float(np.max(np.abs(v - v2)))))
        elif not np.max(np.abs(v - v2)) < 1e-5:
            fails.append((tag, "v", float(np.max(np.abs(v - v2)))))
        done += 1
    except Exception as ex:  # noqa: BLE001
        fails.append((tag, "raised", type(ex).__name__, str(ex)[:160]))
for f in fails:
    print("FAIL", f)
print("%d compared, %d failures, %.0f s" % (done, len(fails), time.time() - t_start))
sys.exit(1 if fails else 0)
