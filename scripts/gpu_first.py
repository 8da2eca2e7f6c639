"""Development script: first end-to-end check of the HIP engine against the oracle (run under gpurun)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cedarsim_jl_amd import Circuit, SIN, PWL, tran_opts, dc_opts
from cedarsim_jl_amd.netlist import parse_spice, parse_spice_file
from cedarsim_jl_amd.engine import EngineCircuit
from oracle_binding import Oracle
np.set_printoptions(linewidth=220, precision=6)
lib = open(os.path.join(ROOT, "cedarsim.jl_amd/data/gf180_substitute.lib")).read()
res = lambda p: lib if p.startswith("jlpkg://GF180MCUPDK") else None

def section(s): print("\n==== %s ====" % s, flush=True)

section("VR")
c = Circuit(); c.V("V", "vcc", 0, dc=5.0); c.R("R", "vcc", 0, 2.0)
e = EngineCircuit(c); print(e.info())
print(e.dc())
c = Circuit(); c.V("V", "vcc", 0, dc=5.0); c.R("R", "vcc", 0, 2.0); c.observe_branch("V")
e = EngineCircuit(c); print(e.info()); print(e.dc()[:2], Oracle(c).dc()[:2])

section("Butterworth")
c = Circuit()
c.V("V1","vin",0,tran=SIN(0,1,1/(2*np.pi))); c.L("L1","vin","n1",1.5); c.C("C2","n1",0,4/3); c.L("L3","n1","vout",0.5); c.R("R4","vout",0,1.0)
c.observe_node("vout")
e = EngineCircuit(c); print(e.info())
t0=time.time(); rc,t,v,xf,st = e.tran(0,100.0,tran_opts(abstol=1e-9,reltol=1e-9,skip_dc=True)); 
an = (np.exp(-t)-np.sin(t)-np.cos(t))/2 + 2*np.sin(np.sqrt(3)*t/2)/(np.sqrt(3)*np.sqrt(np.exp(t)))
print(rc, len(t), "%.2fs"%(time.time()-t0), "maxerr", np.max(np.abs(v[0,:,0]-an)), st)

section("MOS eval parity")
nl = parse_spice_file(os.path.join(ROOT,"tests/golden/DFF_cap_all.cir"), lib_resolver=res)
c = nl.build(); c.observe_node("q")
e = EngineCircuit(c); o = Oracle(c); print(e.info())
rng = np.random.default_rng(0)
v = rng.uniform(-0.5, 5.5, size=(30,4))
a = e.mos_eval(v); b = o.mos_eval(v)
sc = np.maximum(np.abs(b).max(axis=0, keepdims=True), 1e-30)
print("max rel (by column scale):", np.max(np.abs(a-b)/sc), " worst col", np.argmax(np.max(np.abs(a-b)/sc,axis=0)))

section("eval parity (F,Q,J) on DFF")
rc, xo, sto = o.dc(dc_opts(abstol=1e-14))
x = xo + 0.05*rng.standard_normal(xo.shape)
# known nodes must hold their known values: use oracle's dc for those
nu, nk, bu = e.maps()
for n in range(1, c.n_nodes+1):
    if nu[n] < 0: x[n-1] = xo[n-1]
Fe,Qe,Je = e.eval(x, t=0.0, alpha0=1e9, mode=1)
Fo,Qo,Jo = o.eval(x, t=0.0, alpha0=1e9, mode=1)
rows = [n-1 for n in range(1,c.n_nodes+1) if nu[n]>=0]
# alias: merged nodes -> compare after summing oracle rows/cols of merged nodes
print("unknown rows:", len(rows), "n_mna", c.n_mna)
groups = {}
for n in range(1,c.n_nodes+1):
    if nu[n]>=0: groups.setdefault(nu[n], []).append(n-1)
reps = [g[0] for g in groups.values()]
Fo_r = np.array([sum(Fo[i] for i in g) for g in groups.values()])
Jo_r = np.array([[sum(Jo[i,j] for i in g for j in h) for h in groups.values()] for g in groups.values()])
Fe_r = Fe[reps]; Je_r = Je[np.ix_(reps,reps)]
print("F err", np.max(np.abs(Fe_r-Fo_r))/np.max(np.abs(Fo_r)), "J err", np.max(np.abs(Je_r-Jo_r))/np.max(np.abs(Jo_r)))

section("DFF DC + transient")
rc, xe, status, ste = e.dc(dc_opts(abstol=1e-14))
print("engine dc", rc, ste["nnonliniter"], ste["nrestarts"], "oracle iters", sto["nnonliniter"])
print("max |x_e - x_o| nodes:", np.nanmax(np.abs(xe[0][:c.n_nodes]-xo[:c.n_nodes])))
for tol in (1e-4, 1e-6):
    t0=time.time(); rc,t,v,xf,st = e.tran(0,7e-7,tran_opts(abstol=tol,reltol=tol,dc=dc_opts(abstol=1e-14)))
    wall=time.time()-t0
    rco,to,vo,xfo,sto2 = o.tran(0,7e-7,tran_opts(abstol=tol,reltol=tol,dc=dc_opts(abstol=1e-14)))
    q = [float(np.interp(tt,t,v[0,:,0])) for tt in (1.5e-7,2.5e-7,4.5e-7,5.5e-7,7e-7)]
    tt = np.linspace(0,7e-7,2001)
    d = np.abs(np.interp(tt,t,v[0,:,0])-np.interp(tt,to,vo[0]))
    print("tol",tol,"rc",rc,"steps",len(t),"wall %.3fs"%wall, {k:st[k] for k in ("nnonliniter","naccept","nreject","nnonlinconvfail","device_seconds","n_kernel_launches")}, "Q", np.round(q,5), "oracle steps", len(to), "max|dq|", d.max(), "mean|dq|", d.mean())
