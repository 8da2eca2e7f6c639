"""Extended differential fuzz of circuits made of several INDEPENDENT blocks (not part of the suite): 2-5 random networks that share
only ground, transient on a saveat grid (device stepper: every block its own steps, per-workgroup constant blobs) and without one
(lock-step) against the oracle, which integrates the whole system with one step sequence.
usage: python scripts/extended_fuzz_blocks.py [first_seed] [n_seeds] [seconds]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cedarsim_jl_amd import dc_opts, tran_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
from test_gpu_parity import _random_circuit  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 90000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
t_start = time.time()
fails, done, skipped, modes = [], 0, 0, {}
sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
for seed in range(first, first + n_seeds):
    if time.time() - t_start > budget:
        break
    if (seed - first) % 100 == 0:
        print("progress: seed %d, %d compared, %d failures, %.0f s, stepper modes %s" % (seed, done, len(fails), time.time() - t_start, modes), flush=True)
    rng = np.random.default_rng(seed)
    c = None
    for b in range(int(rng.integers(2, 6))):
        c = _random_circuit(rng, int(rng.integers(3, 11)), with_mos=(seed + b) % 3 == 0, c=c, prefix="b%d_" % b)
    c.observe_all_nodes()
    try:
        o = Oracle(c)
        ref = {}
        for key, grid in (("grid", sv), ("all", None)):
            rco, to, vo, _, _ = o.tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=grid, dc=dc_opts(abstol=1e-12, tran_mode=1)))
            ref[key] = (rco, to, vo if vo.ndim == 2 else vo[:, :, 0])
        if ref["grid"][0] != 0:
            skipped += 1
            continue
        e = EngineCircuit(c)
        for stp in ("host", "device"):
            for key, grid in (("grid", sv), ("all", None)):
                rce, te, ve, _, ste = e.tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=grid, dc=dc_opts(abstol=1e-12, tran_mode=1), stepper=stp))
                if stp == "device" and rce == -6:
                    continue
                modes[(stp, key, ste["stepper_mode"])] = modes.get((stp, key, ste["stepper_mode"]), 0) + 1
                if rce != 0:
                    fails.append((seed, "tran rc", stp, key, rce, e.ctx.last_error()[:80]))
                    continue
                rco, to, vo = ref[key]
                if key == "grid":
                    err = float(np.abs(ve[:, :, 0] - vo).max())
                else:   # own time points: compare at the oracle's times by linear interpolation of the engine's rows (dense enough at these tolerances)
                    err = max(float(np.abs(np.interp(to, te, ve[k, :, 0]) - vo[k]).max()) for k in range(vo.shape[0]))
                tol = (1e-4 if key == "grid" else 5e-3) * max(1.0, float(np.abs(vo).max()))
                if not err < tol:
                    fails.append((seed, "tran v", stp, key, err, float(np.abs(vo).max())))
        done += 1
    except Exception as ex:  # noqa: BLE001
        fails.append((seed, "raised", type(ex).__name__, str(ex)[:160]))
for fl in fails:
    print("FAIL", fl)
print("seeds %d..%d: %d compared, %d skipped, %d failures, %.0f s, stepper modes %s" % (first, seed, done, skipped, len(fails), time.time() - t_start, modes))
sys.exit(1 if fails else 0)
