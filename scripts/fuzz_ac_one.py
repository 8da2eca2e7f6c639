"""One seed of scripts/extended_fuzz_ac.py, verbosely.  usage: [FUZZ_MAX_NODES=40] python scripts/fuzz_ac_one.py SEED"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cedarsim_jl_amd import dc_opts  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
from test_gpu_parity import _random_circuit  # noqa: E402
seed = int(sys.argv[1])
f = np.array([1e2, 1e4, 1e6, 1e8, 1e10])
rng = np.random.default_rng(seed)
nn = int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "40"))))
c = _random_circuit(rng, nn, with_mos=seed % 2 == 0)
inj = "n%d" % (1 + int(rng.integers(nn)))
c.I("iac_fuzz", inj, 0, dc=0.0, ac=1.0)
c.observe_all_nodes()
o, e = Oracle(c), EngineCircuit(c, small_signal=True)
rc_o, xo = o.ac(f, dc_opts(abstol=1e-12))
rc, xe, st = e.ac(f, dc_opts(abstol=1e-12))
xe = xe[0]
print("rc", rc_o, rc, "info", {k: v for k, v in e.info().items() if k in ("n_nodes", "n_unknowns", "n_known", "n_alias", "n_components", "max_component", "path")}, "inject at", inj)
ok = ~np.isnan(xe)
d = np.abs(np.where(ok, xe - xo, 0.0))
fi, ki = np.unravel_index(d.argmax(), d.shape)
print("worst at freq index", fi, "mna index", ki, "engine", xe[fi, ki], "oracle", xo[fi, ki])
names = {c._n(nm) - 1: nm for nm in ["n%d" % i for i in range(1, nn + 1)]}
for k in range(xe.shape[1]):
    if ok[0, k] and np.abs(xe[:, k] - xo[:, k]).max() > 1e-6 * max(1e-30, np.abs(xo).max()):
        print(" mna", k, names.get(k, "branch/internal"), "engine", np.abs(xe[:, k]), "oracle", np.abs(xo[:, k]))
nu, nk, bu = e.maps()
print("node->unknown", list(nu)[:nn + 1])
print("devices:", [(c.dev_names[i], c.dev_kind[i], [c.node_names[n] if n < len(c.node_names) else n for n in c.dev_node[i][:4]]) for i in range(len(c.dev_kind))])
