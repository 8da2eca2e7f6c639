"""Extended differential fuzz of the small-signal analyses (not part of the suite): random networks with an AC current injection,
`ch_ac` and `ch_noise` against the oracle.  usage: [FUZZ_MAX_NODES=40] python scripts/extended_fuzz_ac.py [first_seed] [n_seeds] [seconds]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cedarsim_jl_amd import dc_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
from test_gpu_parity import _random_circuit  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
t_start = time.time()
fails, done, skipped = [], 0, 0
f = np.array([1e2, 1e4, 1e6, 1e8, 1e10])
for seed in range(first, first + n_seeds):
    if time.time() - t_start > budget:
        break
    if (seed - first) % 250 == 0:
        print("progress: seed %d, %d compared, %d failures, %.0f s" % (seed, done, len(fails), time.time() - t_start), flush=True)
    rng = np.random.default_rng(seed)
    nn = int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "40"))))
    c = _random_circuit(rng, nn, with_mos=seed % 2 == 0)
    c.I("iac_fuzz", "n%d" % (1 + int(rng.integers(nn))), 0, dc=0.0, ac=1.0)
    c.observe_all_nodes()
    out = "n%d" % (1 + int(rng.integers(nn)))
    try:
        o, e = Oracle(c), EngineCircuit(c, small_signal=True)
        rc_o, xo = o.ac(f, dc_opts(abstol=1e-12))
        rc, xe, st = e.ac(f, dc_opts(abstol=1e-12))
        if rc_o != 0 or rc != 0:
            if (rc_o == 0) != (rc == 0):
                fails.append((seed, "ac rc", rc_o, rc, e.ctx.last_error()[:80]))
            else:
                skipped += 1
            continue
        xe = xe[0]
        ok = ~np.isnan(xe)
        scale = max(1e-30, float(np.abs(xo[ok]).max()))
        if scale < 1e-9:      # the injection went into a node an ideal source holds: the response is rounding noise
            skipped += 1
            continue
        err = float(np.abs(xe[ok] - xo[ok]).max()) / scale
        if not err < 1e-6:
            fails.append((seed, "ac x", err))
            continue
        rc_o, po = o.noise(c._n(out) - 1, f, dc_opts(abstol=1e-12))
        rc, pe, st = EngineCircuit(c).noise(0, c._n(out), f, dc_opts(abstol=1e-12))
        if rc_o != 0 or rc != 0:
            if (rc_o == 0) != (rc == 0):
                fails.append((seed, "noise rc", rc_o, rc))
            else:
                skipped += 1
            continue
        if not np.allclose(pe[0], po, rtol=1e-6, atol=1e-40):
            fails.append((seed, "noise psd", float(np.max(np.abs(pe[0] - po) / np.maximum(np.abs(po), 1e-300)))))
            continue
        done += 1
    except Exception as ex:  # noqa: BLE001
        fails.append((seed, "raised", type(ex).__name__, str(ex)[:160]))
for fl in fails:
    print("FAIL", fl)
print("seeds %d..%d: %d compared, %d skipped, %d failures, %.0f s" % (first, seed, done, skipped, len(fails), time.time() - t_start))
sys.exit(1 if fails else 0)
