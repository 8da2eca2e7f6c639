"""F / Q / J of the engine against the oracle for one seed of the extended fuzz (structural reduction taken into account the way
tests/test_gpu_parity.py does).  usage: [FUZZ_MAX_NODES=70] python scripts/fuzz_eval.py SEED"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cedarsim_jl_amd import dc_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
from test_gpu_parity import _random_circuit, canon, reduce_rows  # noqa: E402
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
c = _random_circuit(rng, int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "16")))), with_mos=(seed % 2 == 0 or os.environ.get("FUZZ_ALWAYS_MOS") is not None))
c.observe_all_nodes()
e, o = EngineCircuit(c), Oracle(c)
nu, nk, bu = e.maps()
rc, xo, _ = o.dc(dc_opts(abstol=1e-12, tran_mode=1))
print("oracle dc rc", rc, "info", {k: v for k, v in e.info().items() if k in ("n_unknowns", "n_known", "n_alias", "max_component")})
x = canon(c, xo.copy(), nu)
for alpha0 in (0.0, 1e6, 1e9, 1e12):
    Fe, Qe, Je = e.eval(x, t=0.0, alpha0=alpha0, mode=1)
    Fo, Qo, Jo = o.eval(x, t=0.0, alpha0=alpha0, mode=1)
    reps, Fo_r, Jo_r = reduce_rows(c, nu, Fo, Jo)
    _, Qo_r, _ = reduce_rows(c, nu, Qo)
    dJ = np.abs(Je[np.ix_(reps, reps)] - Jo_r)
    print("alpha0 %.0e: max|dF| %.3e (of %.3e)  max|dQ| %.3e (of %.3e)  max|dJ| %.3e (of %.3e) at %s" % (alpha0, np.max(np.abs(Fe[reps] - Fo_r)), np.max(np.abs(Fo_r)), np.max(np.abs(Qe[reps] - Qo_r)), np.max(np.abs(Qo_r)), dJ.max(), np.max(np.abs(Jo_r)), np.unravel_index(dJ.argmax(), dJ.shape)))
    if alpha0 == 1e9:
        Jr = Je[np.ix_(reps, reps)]
        print("cond(J engine) %.3e  cond(J oracle) %.3e" % (np.linalg.cond(Jr), np.linalg.cond(Jo_r)))
print("path", e.info()["path"])
