"""Writes tests/golden/closed_form.json: the closed-form known answers the reference's own tests
assert for this path.  Every entry cites the reference test it restates.  The values
are formulas evaluated here — the reference cannot run in this image (no Julia)."""
import json
import math
import os

out = {
    "vr": {"cite": "test/basic.jl:21-43", "V": 5.0, "R": 2.0, "R.V": 5.0, "R.I": 2.5},
    "ir": {"cite": "test/basic.jl:81-106", "I": -5.0, "R": 2.0, "R.V": 10.0, "R.I": 5.0},
    "vrc": {"cite": "test/basic.jl:108-141", "V": 5.0, "R": 2000.0, "C": 1e-6, "C.I(0)": 5.0 / 2000.0, "C.V(0)": 0.0, "C.V(end)": 5.0, "C.I(end)": 0.0},
    "parallel": {"cite": "test/basic.jl:144-166", "m": 10, "C.I(0)": 10 * 5.0 / 2000.0, "C.V(end)": 5.0},
    "two_resistor_sweep": {"cite": "test/sweep.jl:326-340", "R1": [100.0 * i for i in range(1, 21)], "R2": [100.0 * i for i in range(1, 21)],
                           "V.I": "-1/(R1+R2)"},
    "spice_sweep": {"cite": "test/sweep.jl:342-371", "v_in": list(range(1, 11)), "r_load": list(range(1, 11)), "r1.I": "v_in/r_load"},
    "pwl_ir": {"cite": "test/transients.jl:17-63", "i_max": 2.0, "r": 2.0, "t0": 1e-3, "t1": 9e-3, "tspan": [0.0, 10e-3]},
    "pwl_slope": {"cite": "test/transients.jl:66-96", "ts": [0.0, 100e-9, 110e-9, 200e-9, 210e-9], "ys": [0.0, 0.0, 5.0, 5.0, 0.0],
                  "t": [0.0, 50e-9, 99e-9, 100e-9, 110e-9, 200e-9], "dydt": [0.0, 0.0, 0.0, 5.0e8, 0.0, -5.0e8]},
    "butterworth": {"cite": "test/transients.jl:98-173", "L1": 1.5, "C2": 4.0 / 3.0, "L3": 0.5, "R4": 1.0, "tspan": [0.0, 100.0],
                    "vout": "(exp(-t)-sin(t)-cos(t))/2 + 2 sin(sqrt(3) t/2)/(sqrt(3) sqrt(exp(t)))",
                    "samples_t": [0.5, 1.0, 2.0, 5.0, 10.0, 50.0, 100.0]},
    "multiplicity": {"cite": "test/basic.jl:556-595", "expect": 10.0 / 11.0},
    "units": {"cite": "test/basic.jl:609-638", "1Meg": 1e6, "1Mil": 25.4e-6, "0.22u": 0.22e-6},
    "dff_gate": {"cite": "test/gf180_dff.jl:29-33", "t": [1.5e-7, 2.5e-7, 4.5e-7, 5.5e-7, 7.0e-7], "q": [0.0, 0.0, 5.0, 5.0, 5.0], "atol": 1e-4},
    "inverter_gate": {"cite": "test/inverter.jl:40-50", "t": [0.5e-7, 1.5e-7, 2.5e-7, 3.5e-7], "d": [0.0, 5.0, 0.0, 5.0], "q": [5.0, 0.0, 5.0, 0.0], "tol": 1e-7},
}
bw = out["butterworth"]
bw["samples_vout"] = [(math.exp(-t) - math.sin(t) - math.cos(t)) / 2 + 2 * math.sin(math.sqrt(3) * t / 2) / (math.sqrt(3) * math.sqrt(math.exp(t)))
                      for t in bw["samples_t"]]
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "closed_form.json"), "w") as f:
    json.dump(out, f, indent=1)
