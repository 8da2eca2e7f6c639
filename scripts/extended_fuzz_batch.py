"""Extended differential fuzz of batched samples (not part of the suite): a random network, three resistor / capacitor values swept over
4 samples through parameter slots, ONE batched transient per step controller against one oracle run per sample.
usage: [FUZZ_MAX_NODES=16] python scripts/extended_fuzz_batch.py [first_seed] [n_seeds] [seconds]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cedarsim_jl_amd import dc_opts, tran_opts  # noqa: E402
from cedarsim_jl_amd.circuit import DEV_C, DEV_R  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
from test_gpu_parity import _random_circuit  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
budget = float(sys.argv[3]) if len(sys.argv) > 3 else 300.0
S = 4
t_start = time.time()
fails, done, skipped = [], 0, 0
sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
for seed in range(first, first + n_seeds):
    if time.time() - t_start > budget:
        break
    if (seed - first) % 100 == 0:
        print("progress: seed %d, %d compared, %d failures, %.0f s" % (seed, done, len(fails), time.time() - t_start), flush=True)

    def build():
        rng = np.random.default_rng(seed)
        c = _random_circuit(rng, int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "16")))), with_mos=seed % 2 == 0)
        c.observe_all_nodes()
        return c, rng
    c, rng = build()
    cand = [i for i, k in enumerate(c.dev_kind) if k in (DEV_R, DEV_C)]
    picks = [cand[j] for j in rng.choice(len(cand), size=min(3, len(cand)), replace=False)]
    scale = np.exp(rng.uniform(-0.7, 0.7, (len(picks), S)))
    scale[:, 0] = 1.0
    vals = np.array([[c.dev_par[d][0] * scale[q, s] for s in range(S)] for q, d in enumerate(picks)])
    slots = [c.slot(c.dev_names[d], "r" if c.dev_kind[d] == DEV_R else "c") for d in picks]
    try:
        e = EngineCircuit(c)
        e.set_samples(S)
        e.set_params(slots, [vals[q] for q in range(len(picks))])
        refs = []
        bad_ref = False
        for s in range(S):
            cs, _ = build()
            for q, d in enumerate(picks):
                cs.dev_par[d][0] = float(vals[q, s])
            rco, to, vo, _, _ = Oracle(cs).tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1)))
            if rco != 0:
                bad_ref = True
                break
            refs.append(vo if vo.ndim == 2 else vo[:, :, 0])
        if bad_ref:
            skipped += 1
            continue
        for stp in ("host", "device"):
            rce, te, ve, _, ste = e.tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1), stepper=stp))
            if stp == "device" and rce == -6:
                continue
            if rce != 0:
                fails.append((seed, "tran rc", stp, rce, e.ctx.last_error()[:80]))
                continue
            for s in range(S):
                err = float(np.abs(ve[:, :, s] - refs[s]).max())
                if not err < 2e-4 * max(1.0, float(np.abs(refs[s]).max())):
                    fails.append((seed, "tran v", stp, s, err, float(np.abs(refs[s]).max())))
                    break
        done += 1
    except Exception as ex:  # noqa: BLE001
        fails.append((seed, "raised", type(ex).__name__, str(ex)[:160]))
for fl in fails:
    print("FAIL", fl)
print("seeds %d..%d: %d compared, %d skipped, %d failures, %.0f s" % (first, seed, done, skipped, len(fails), time.time() - t_start))
sys.exit(1 if fails else 0)
