"""Development script: config 4 shape on one GPU — S Monte-Carlo samples of one DFF in one batched solve."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from cedarsim_jl_amd import dc_opts, tran_opts
from cedarsim_jl_amd import bsim4_params as B4
from cedarsim_jl_amd.engine import EngineCircuit
from cedarsim_jl_amd.workloads import dff_array, DFF_CHECK_TIMES, DFF_CHECK_Q
for S in (1024, 8192):
    c = dff_array(1)
    slots, names = [], []
    for m in ("nfet_06v0", "pfet_06v0"):
        for p in ("vth0", "u0", "toxe"):
            slots.append(c.slot(m, p)); names.append((m, p))
    rng = np.random.default_rng(2024)
    base = np.array([c.models[c.model_names.index(m)][B4.PARAM_INDEX[p]] for m, p in names])
    vals = base[:, None] * (1.0 + 0.03 * rng.standard_normal((len(slots), S)))
    e = EngineCircuit(c)
    t0 = time.time(); e.set_samples(S); e.set_params(slots, vals); 
    opts = tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14), saveat=np.array(DFF_CHECK_TIMES))
    rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)   # warm-up incl. parameter packing
    t1 = time.time()
    rc, t, v, xf, st = e.tran(0.0, 7e-7, opts)
    t2 = time.time()
    q = v[0]
    ok = np.abs(q - np.array(DFF_CHECK_Q)[:, None]) < 1e-3
    print("S=%d rc=%d first %.3fs second %.3fs | launches %d avg launch %.1f us | sum iters %d block iters %d | gate ok for %d/%d samples | dc %.3fs" % (
        S, rc, t1 - t0, t2 - t1, st["n_kernel_launches"], 1e6 * st["device_seconds"] / max(1, st["n_kernel_launches"]), st["nnonliniter"], st["n_block_iters"], int(ok.all(axis=0).sum()), S, st["dc_seconds"]), flush=True)
