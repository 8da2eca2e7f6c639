"""Extended differential fuzz with compiled Verilog-A devices (not part of the suite): the random networks of the other fuzz scripts
plus a handful of va_diode / va_mos1 / va_resistor / va_capacitor / va_inductor instances; DC and transient on both step controllers
against the oracle (wide stamp records: the WIDE kernels, and the sparse path above 16 unknowns per block).
usage: [FUZZ_MAX_NODES=14] python scripts/extended_fuzz_va.py [first_seed] [n_seeds] [seconds]"""
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

first = int(sys.argv[1]) if len(sys.argv) > 1 else 110000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
t_start = time.time()
fails, done, skipped, paths = [], 0, 0, {}
sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
for seed in range(first, first + n_seeds):
    if time.time() - t_start > budget:
        break
    if (seed - first) % 100 == 0:
        print("progress: seed %d, %d compared, %d failures, %.0f s, %s" % (seed, done, len(fails), time.time() - t_start, paths), flush=True)
    rng = np.random.default_rng(seed)
    nn = int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "14"))))
    c = _random_circuit(rng, nn, with_mos=seed % 4 == 0)
    names = ["n%d" % i for i in range(1, nn + 1)]
    pick = lambda: names[rng.integers(nn)] if rng.random() > 0.15 else 0  # noqa: E731
    for q in range(int(rng.integers(1, 5))):
        a, b = pick(), pick()
        if a == b:
            continue
        t = rng.random()
        if t < 0.35:
            c.VA("xd%d" % q, "va_diode", [a, b], {"IS": float(10 ** rng.uniform(-15, -12)), "RS": float(rng.uniform(0.5, 20.0)), "N": float(rng.uniform(1.0, 1.8)), "CJ0": float(10 ** rng.uniform(-14, -12))})
        elif t < 0.6:
            typ = 1.0 if rng.random() < 0.5 else -1.0
            c.VA("xm%d" % q, "va_mos1", [a, pick(), b, 0 if typ > 0 else names[0]], {"TYPE": typ, "W": float(rng.uniform(1e-6, 1e-5)), "L": 1e-6, "VTO": float(typ * rng.uniform(0.4, 0.9)), "KP": 5e-5})
        elif t < 0.75:
            c.VA("xr%d" % q, "va_resistor", [a, b], {"R": float(10 ** rng.uniform(2, 5))})
        elif t < 0.9:
            c.VA("xc%d" % q, "va_capacitor", [a, b], {"C": float(10 ** rng.uniform(-13, -10))})
        else:
            c.VA("xl%d" % q, "va_inductor", [a, b], {"L": float(10 ** rng.uniform(-7, -5)), "RS": float(rng.uniform(1.0, 100.0))})
    c.observe_all_nodes()
    try:
        o, e = Oracle(c), EngineCircuit(c)
        rc_o, x_o, _ = o.dc(dc_opts(abstol=1e-12))
        rc, x, status, st = e.dc(dc_opts(abstol=1e-12))
        if rc_o != 0 or rc != 0:
            if (rc_o == 0) != (rc == 0):
                fails.append((seed, "dc rc", rc_o, rc, e.ctx.last_error()[:80]))
            else:
                skipped += 1
            continue
        xe = x[0]
        known = ~np.isnan(xe)
        if not np.allclose(xe[known], x_o[known], rtol=1e-6, atol=1e-9):
            xf = xe.copy(); xf[~known] = x_o[~known]
            F, Q, J = o.eval(xf, 0.0, 0.0, 0)
            if float(np.abs(F).max()) > 1e-7:   # a different but valid operating point (diodes / MOSFETs: more than one) is not a defect
                fails.append((seed, "dc x", float(np.abs(xe[known] - x_o[known]).max()), "KCL of the engine's point in the oracle: %.2e" % float(np.abs(F).max())))
            else:
                skipped += 1
            continue
        opts = lambda stp: tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1), stepper=stp)  # noqa: E731
        rco, to, vo, _, _ = o.tran(0.0, 1e-6, opts("auto"))
        vo = vo if vo.ndim == 2 else vo[:, :, 0]
        for stp in ("host", "device"):
            rce, te, ve, _, ste = e.tran(0.0, 1e-6, opts(stp))
            if stp == "device" and rce == -6:
                continue
            key = (e.info()["path"], stp, ste["stepper_mode"]); paths[key] = paths.get(key, 0) + 1
            if rce != 0 and rco != 0:
                skipped += 1
            elif rce != rco:
                fails.append((seed, "tran rc", stp, rco, rce, e.ctx.last_error()[:60]))
            elif rce == 0:
                err = float(np.abs(ve[:, :, 0] - vo).max())
                if not err < 1e-4 * max(1.0, float(np.abs(vo).max())):
                    fails.append((seed, "tran v", stp, err, float(np.abs(vo).max())))
        done += 1
    except Exception as ex:  # noqa: BLE001
        fails.append((seed, "raised", type(ex).__name__, str(ex)[:160]))
for fl in fails:
    print("FAIL", fl)
print("seeds %d..%d: %d compared, %d skipped, %d failures, %.0f s, %s" % (first, seed, done, skipped, len(fails), time.time() - t_start, paths))
sys.exit(1 if fails else 0)
