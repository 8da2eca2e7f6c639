import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from cedarsim_jl_amd import CircuitSweep
from cedarsim_jl_amd.workloads import DFF_TSPAN, DFF_CHECK_TIMES, dff_mc_builder, mc_tandem_sweep
build, names = dff_mc_builder(observe=("q", "q_neg"))
cs = CircuitSweep(build, mc_tandem_sweep(1024))
saveat = np.unique(np.concatenate((np.linspace(0, 7e-7, 2001), np.array(DFF_CHECK_TIMES))))
for rep in range(2):
    t0 = time.perf_counter()
    rc, t, rows, st = cs.tran_arrays(DFF_TSPAN, abstol=1e-4, reltol=1e-4, dc_abstol=1e-14, saveat=saveat)
    print(rc, time.perf_counter() - t0, st["stepper"], st["stepper_mode"], st["dc_seconds"], rows.shape)
