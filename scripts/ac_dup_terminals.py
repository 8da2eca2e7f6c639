"""AC linearisation of a MOSFET whose drain and gate are the same node (diagnostic).  usage: python scripts/ac_dup_terminals.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from cedarsim_jl_amd import dc_opts  # noqa: E402
from cedarsim_jl_amd.circuit import Circuit  # noqa: E402
from cedarsim_jl_amd.engine import EngineCircuit  # noqa: E402
from cedarsim_jl_amd.workloads import gf180_models  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
f = np.array([1e2, 1e6, 1e10])


def ckt(kind):
    c = Circuit(gmin=1e-12)
    m = gf180_models()
    n = c.add_model(*m["nfet_06v0"])
    c.V("vdd", "vdd", 0, dc=3.0)
    c.R("rl", "vdd", "d", 20e3)
    if kind == "diode":
        c.M("m1", "d", "d", "s", 0, n, 2e-6, 6e-7)
    elif kind == "separate":
        c.R("rg", "d", "g", 1e-3)
        c.M("m1", "d", "g", "s", 0, n, 2e-6, 6e-7)
    c.R("rs", "s", 0, 1e3)
    c.I("iac", "d", 0, dc=0.0, ac=1.0)
    c.observe_all_nodes()
    return c


for kind in ("diode", "separate"):
    c = ckt(kind)
    e, o = EngineCircuit(c, small_signal=True), Oracle(c)
    rc, xe, st = e.ac(f, dc_opts(abstol=1e-12))
    rco, xo = o.ac(f, dc_opts(abstol=1e-12))
    k = c._n("d") - 1
    print(kind, "rc", rc, rco, "|v(d)| engine", np.abs(xe[0][:, k]), "oracle", np.abs(xo[:, k]), "rel", np.abs(xe[0][:, k] - xo[:, k]) / np.abs(xo[:, k]))
