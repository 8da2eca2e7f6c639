"""Development script: stand-alone timing of the BSIM4 evaluation kernels (30720 instances)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cedarsim_jl_amd.engine import EngineCircuit
from cedarsim_jl_amd.workloads import dff_array
c = dff_array(1024, observe="q0")
e = EngineCircuit(c)
rng = np.random.default_rng(0)
v = rng.uniform(0, 5, size=(30720, 4))
for i in range(20):
    out = e.mos_eval(v)
    out = e.mos_eval(v, quad=True)
print("ok", out.shape)
