"""Writes the oracle's full MNA Jacobian G + alpha0*C of one random network of tests/test_gpu_parity.py::_random_circuit (seed, alpha0, file) as
sparse rows "count (col value)..." — tests/golden/mna_jacobian_seed20095_{dc,tran}.txt: seed 20095 at alpha0 = 0 and 1e12 (FUZZ_MAX_NODES = 70).
The static pivot sequence of rounds 1-3 (a matching on entries large in their rows) met an exact zero pivot on both."""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np
from cedarsim_jl_amd import dc_opts
from oracle_binding import Oracle
from test_gpu_parity import _random_circuit
seed=int(sys.argv[1]); alpha0=float(sys.argv[2])
rng=np.random.default_rng(seed)
c=_random_circuit(rng, int(rng.integers(3, 70)), with_mos=seed % 2 == 0)
o=Oracle(c)
rc,xo,_=o.dc(dc_opts(abstol=1e-12, tran_mode=1))
F,Q,J=o.eval(xo,0.0,alpha0,1)
n=J.shape[0]
with open(sys.argv[3],'w') as f:
    f.write("%d
"%n)
    for i in range(n):
        nz=[(j,J[i,j]) for j in range(n) if J[i,j]!=0.0 or i==j]
        f.write("%d "%len(nz)+" ".join("%d %.17g"%(j,v) for j,v in nz)+"
")
print("n",n,"cond",np.linalg.cond(J))
