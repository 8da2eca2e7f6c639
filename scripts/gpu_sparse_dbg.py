import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cedarsim_jl_amd import Circuit, dc, tran, dc_opts
from cedarsim_jl_amd.engine import EngineCircuit
for rep in range(3):
    c = Circuit(); c.V("V", "vcc", 0, dc=5.0); c.R("R", "vcc", 0, 2.0)
    c.observe_all_nodes(); c.observe_branch("V")
    e = EngineCircuit(c); print("built", e.info(), flush=True)
    print(e.dc(), flush=True)
    c = Circuit(); c.I("I", "icc", 0, dc=-5.0); c.R("R", "icc", 0, 2.0); c.observe_all_nodes()
    e2 = EngineCircuit(c); print("built2", e2.info(), flush=True)
    print(e2.dc(), flush=True)
print("done")
