import sys, numpy as np
sys.path.insert(0, '/root/repo')
from cedarsim_jl_amd import dc_opts, tran_opts
from cedarsim_jl_amd.engine import EngineCircuit
from cedarsim_jl_amd.circuit import Circuit, CedarError
from cedarsim_jl_amd.workloads import dff_array
def rc_ckt():
    c = Circuit(gmin=1e-12); c.V("v1", "in", 0, dc=1.0); c.R("r1", "in", "a", 1e3); c.C("c1", "a", 0, 1e-9); c.observe_node("a"); return c
def trial(name, f):
    try:
        r = f(); print(name, "->", r)
    except Exception as ex:
        print(name, "-> raised", type(ex).__name__, str(ex)[:150])
e = EngineCircuit(rc_ckt())
trial("t0 == t1", lambda: e.tran(0.0, 0.0, tran_opts())[0])
trial("t1 < t0", lambda: e.tran(1e-6, 0.0, tran_opts())[0])
trial("saveat unsorted", lambda: e.tran(0.0, 1e-6, tran_opts(saveat=np.array([5e-7, 2e-7])))[0])
trial("saveat outside", lambda: (lambda r: (r[0], r[1][:4]))(e.tran(0.0, 1e-6, tran_opts(saveat=np.array([-1e-7, 5e-7, 2e-6])))))
trial("saveat with NaN", lambda: e.tran(0.0, 1e-6, tran_opts(saveat=np.array([1e-7, np.nan])))[0])
trial("abstol 0", lambda: e.tran(0.0, 1e-6, tran_opts(abstol=0.0, reltol=0.0))[0])
trial("negative tol", lambda: e.tran(0.0, 1e-6, tran_opts(abstol=-1.0))[0])
trial("max_order 9", lambda: e.tran(0.0, 1e-6, tran_opts(max_order=9))[0])
trial("tspan NaN", lambda: e.tran(0.0, float('nan'), tran_opts())[0])
trial("tspan inf", lambda: e.tran(0.0, float('inf'), tran_opts(max_steps=50))[0])
trial("step_control 7", lambda: e.tran(0.0, 1e-6, tran_opts(step_control=7))[0])
trial("stepper 9", lambda: e.tran(0.0, 1e-6, tran_opts(stepper=9))[0])
trial("set_samples 0", lambda: e.set_samples(0))
trial("dc maxiters 0", lambda: e.dc(dc_opts(maxiters=0))[0])
for st in ("host", "device"):
    trial("normal " + st, lambda: (lambda r: (r[0], len(r[1]), float(r[2][0, -1, 0])))(e.tran(0.0, 5e-6, tran_opts(stepper=st))))
