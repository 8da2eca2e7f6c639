import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from cedarsim_jl_amd import dc_opts
from cedarsim_jl_amd.engine import EngineCircuit
from cedarsim_jl_amd.workloads import dff_array
from oracle_binding import Oracle
from test_gpu_parity import canon
c = dff_array(1); e, o = EngineCircuit(c), Oracle(c)
nu, nk, bu = e.maps()
rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
x = canon(c, xo + 0.02*np.random.default_rng(5).standard_normal(xo.shape), nu)
for n in range(1, c.n_nodes+1):
    if nu[n] < 0: x[n-1] = xo[n-1]
rows = sorted({int(u): n-1 for n, u in enumerate(nu) if n > 0 and u >= 0}.values())
for it in range(50):
    F, Q, J = e.eval(x, t=0.0, alpha0=0.0, mode=0)
    print(it, np.max(np.abs(F[rows])))
    if np.max(np.abs(F[rows])) < 1e-13: break
    dx = spla.splu(sp.csc_matrix(J)).solve(-F)
    dx *= min(1.0, 1.0/max(1e-30, np.max(np.abs(dx))))
    x = canon(c, x + dx, nu)
np.set_printoptions(precision=5, linewidth=200)
print(c.node_names[1:]); print(x[:c.n_nodes]); print(xo[:c.n_nodes]); print(x[:c.n_nodes]-xo[:c.n_nodes])
