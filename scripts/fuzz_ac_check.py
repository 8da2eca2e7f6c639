"""Is the engine's AC answer the solution of ITS OWN G + jwC?  usage: [FUZZ_MAX_NODES=40] python scripts/fuzz_ac_check.py SEED"""
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
f = np.array([1e2, 1e6, 1e10])
rng = np.random.default_rng(seed)
nn = int(rng.integers(3, int(os.environ.get("FUZZ_MAX_NODES", "40"))))
c = _random_circuit(rng, nn, with_mos=seed % 2 == 0)
inj = "n%d" % (1 + int(rng.integers(nn)))
c.I("iac_fuzz", inj, 0, dc=0.0, ac=1.0)
c.observe_all_nodes()
o = Oracle(c)
for ss in (False, True):
    e = EngineCircuit(c, small_signal=ss)
    nu, nk, bu = e.maps()
    rc, x, status, st = e.dc(dc_opts(abstol=1e-12))
    rco, xo, _ = o.dc(dc_opts(abstol=1e-12))
    xe = x[0].copy(); known = ~np.isnan(xe)
    print("small_signal", ss, "dc diff", float(np.abs(xe[known] - xo[known]).max()), "unknowns", e.info()["n_unknowns"], "known", e.info()["n_known"])
    xq = canon(c, xo.copy(), nu)
    for mode in (0, 1):
        F0, Q0, J0 = e.eval(xq, 0.0, 0.0, mode); F1, Q1, J1 = e.eval(xq, 0.0, 1.0, mode)
        Fo0, Qo0, Jo0 = o.eval(xq, 0.0, 0.0, mode); Fo1, Qo1, Jo1 = o.eval(xq, 0.0, 1.0, mode)
        reps, _, Go = reduce_rows(c, nu, Fo0, Jo0); _, _, J1o = reduce_rows(c, nu, Fo1, Jo1)
        Ce, Co = (J1 - J0)[np.ix_(reps, reps)], J1o - Go
        print("  mode", mode, "max |G_e - G_o| %.3e  max |C_e - C_o| %.3e (|C| max %.3e)" % (np.abs(J0[np.ix_(reps, reps)] - Go).max(), np.abs(Ce - Co).max(), np.abs(Co).max()))
    if ss:
        rca, xa, _ = e.ac(f, dc_opts(abstol=1e-12)); rcb, xb = o.ac(f, dc_opts(abstol=1e-12))
        ok = ~np.isnan(xa[0])
        print("  ac engine vs oracle rel", [float(np.abs(np.where(ok[i], xa[0][i] - xb[i], 0)).max() / np.abs(xb[i]).max()) for i in range(len(f))])
# which matrix does the engine's AC answer solve?  try C, C^T, sym(C), diag-only variants against the engine's own answer
e = EngineCircuit(c, small_signal=True)
nu, nk, bu = e.maps()
rco, xo, _ = o.dc(dc_opts(abstol=1e-12))
xq = canon(c, xo.copy(), nu)
F0, Q0, J0 = e.eval(xq, 0.0, 0.0, 0); F1, Q1, J1 = e.eval(xq, 0.0, 1.0, 0)
Fo0, Qo0, Jo0 = o.eval(xq, 0.0, 0.0, 0)
reps, _, _ = reduce_rows(c, nu, Fo0, Jo0)
G = J0[np.ix_(reps, reps)]; C = (J1 - J0)[np.ix_(reps, reps)]
print("reps", reps, "x at reps", xq[reps], "all x", xq, "G", G, "C", C, sep="\n")
rca, xa, _ = e.ac(f, dc_opts(abstol=1e-12))
k = list(reps).index(c._n(inj) - 1) if (c._n(inj) - 1) in list(reps) else None
print("inject row", k)
if k is not None:
    b = np.zeros(len(reps), complex); b[k] = -1.0
    for name, Cx in (("C", C), ("C^T", C.T), ("sym", 0.5 * (C + C.T)), ("diag", np.diag(np.diag(C)))):
        errs = []
        for i, ff in enumerate(f):
            x = np.linalg.solve(G + 2j * np.pi * ff * Cx, b)
            errs.append(float(np.abs(x - xa[0][i][reps]).max() / np.abs(x).max()))
        print("engine.ac vs numpy with", name, errs)
