"""One seed of scripts/extended_fuzz.py on one step controller, verbosely.  usage: python scripts/fuzz_one.py SEED host|device [max_steps]"""
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
seed, stp = int(sys.argv[1]), sys.argv[2]
max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
c = _random_circuit(rng, int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "16")))), with_mos=(seed % 2 == 0 or os.environ.get("FUZZ_ALWAYS_MOS") is not None))
c.observe_all_nodes()
sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
o = Oracle(c)
rco, to, vo, _, sto = o.tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1), max_steps=max_steps))
print("oracle rc", rco, "accepted", sto["naccept"], "rejected", sto["nreject"], "convfail", sto["nnonlinconvfail"], "rows", len(to), flush=True)
e = EngineCircuit(c)
print("info", e.info(), flush=True)
t0 = time.time()
rce, te, ve, _, ste = e.tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1), stepper=stp, max_steps=max_steps))
print(stp, "rc", rce, "accepted", ste["naccept"], "rejected", ste["nreject"], "convfail", ste["nnonlinconvfail"], "rows", len(te), "stepper", ste["stepper"], ste["stepper_mode"], "%.2f s" % (time.time() - t0), e.ctx.last_error(), flush=True)
print("info after", {k: v for k, v in e.info().items() if k in ("path", "max_component", "nnz_jac", "nnz_lu", "n_components")}, flush=True)
rcd, xd, std_, _ = e.dc(dc_opts(abstol=1e-12, tran_mode=1))
print("dc alone rc", rcd, e.ctx.last_error(), flush=True)
vo2 = vo if vo.ndim == 2 else vo[:, :, 0]
if rce == 0 and rco == 0:
    d = np.abs(ve[:, :, 0] - vo2)
    print("max |engine| %.3e  max |oracle| %.3e  max diff %.3e at obs %d, time index %d" % (np.abs(ve).max(), np.abs(vo2).max(), d.max(), np.unravel_index(d.argmax(), d.shape)[0], np.unravel_index(d.argmax(), d.shape)[1]))
    k = np.unravel_index(d.argmax(), d.shape)[0]
    print("engine row", ve[k, :, 0]); print("oracle row", vo2[k, :])
print("devices:", [(c.dev_names[i], c.dev_kind[i]) for i in range(len(c.dev_kind))][:60])
