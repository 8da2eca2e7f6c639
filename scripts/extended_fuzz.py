"""Extended differential fuzz (not part of the test suite): the random networks of tests/test_gpu_parity.py with many more seeds,
DC and transient on BOTH step controllers against the oracle.  usage: python scripts/extended_fuzz.py [first_seed] [n_seeds] [seconds]
Prints one line per failure and a summary; exit code 1 when anything differed."""
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

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 420.0
t_start = time.time()
fails, done, skipped = [], 0, 0
sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
for seed in range(first, first + n_seeds):
    if time.time() - t_start > budget:
        break
    if (seed - first) % 250 == 0:
        print("progress: seed %d, %d compared, %d failures, %.0f s" % (seed, done, len(fails), time.time() - t_start), flush=True)
    rng = np.random.default_rng(seed)
    c = _random_circuit(rng, int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "16")))), with_mos=(seed % 2 == 0 or os.environ.get("FUZZ_ALWAYS_MOS") is not None))
    c.observe_all_nodes()
    try:
        o = Oracle(c)
        rc_o, x_o, _ = o.dc(dc_opts(abstol=1e-12))
        e = EngineCircuit(c)
        rc, x, status, st = e.dc(dc_opts(abstol=1e-12))
        if rc_o != 0 or rc != 0:
            if rc_o != rc:
                fails.append((seed, "dc rc", rc_o, rc))
            else:
                skipped += 1
            continue
        xe = x[0]
        known = ~np.isnan(xe)
        if not np.allclose(xe[known], x_o[known], rtol=1e-6, atol=1e-9):
            # a multi-stable operating point is not a bug: check KCL of the engine's own solution in the oracle's residual
            xf = xe.copy(); xf[~known] = x_o[~known]
            F, Q, J = o.eval(xf, 0.0, 0.0, 0)
            fails.append((seed, "dc x", float(np.abs(xe[known] - x_o[known]).max()), "KCL of the engine's point in the oracle: %.2e" % float(np.abs(F).max())))
            continue
        opts = lambda stp: tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1), stepper=stp)  # noqa: E731
        rco, to, vo, _, _ = o.tran(0.0, 1e-6, opts("auto"))
        vo = vo if vo.ndim == 2 else vo[:, :, 0]
        for stp in ("host", "device"):
            try:
                rce, te, ve, _, ste = e.tran(0.0, 1e-6, opts(stp))
            except Exception as ex:  # noqa: BLE001
                fails.append((seed, "tran raised", stp, str(ex)[:120]))
                continue
            if stp == "device" and rce == -6:
                continue       # not eligible for the device stepper: refused, as documented
            if rce != 0 and rco != 0:
                skipped += 1   # both give up (DtLessThanMin here, MaxIters there: a transient neither can integrate)
            elif rce != rco:
                fails.append((seed, "tran rc", stp, rco, rce))
            elif rce == 0:
                err = float(np.abs(ve[:, :, 0] - vo).max())
                if not err < 1e-4 * max(1.0, float(np.abs(vo).max())):
                    fails.append((seed, "tran v", stp, err))
        done += 1
    except Exception as ex:  # noqa: BLE001
        fails.append((seed, "raised", type(ex).__name__, str(ex)[:160]))
for f in fails:
    print("FAIL", f)
print("seeds %d..%d: %d compared, %d skipped (both solvers failed alike), %d failures, %.0f s" % (first, seed, done, skipped, len(fails), time.time() - t_start))
sys.exit(1 if fails else 0)
