"""Generates tests/golden/va_closed_form.json: stamps of two library models derived BY HAND from their Verilog-A equations
(cedarsim.jl_amd/va/library/cedar_basic.va) with plain floating-point arithmetic.  Nothing of cedarsim.jl_amd/va is imported:
the file checks the compiler (front end + code generator) against an evaluation that does not share its parser.
Sign convention of a stamp: I[k] = current leaving node k into the device; G[r][c] = dI[r]/dV[c]."""
import json
import math
import os

P_K, P_Q = 1.3806503e-23, 1.602176462e-19   # constants.vams
cases = []

# ---- va_diode(a, c) with internal node ai, LEVEL = 1: I(a,ai) <+ V(a,ai)/RS ; I(ai,c) <+ IS*(exp(vd/vt)-1) + gmin*vd, vd = V(ai,c)
# parameter block: values in declaration order IS N RS CJ0 VJ M TT FC LEVEL, then the nine $param_given flags, then one pad
for (va, vc, vai, IS, N, RS, T) in ((0.7, 0.0, 0.62, 1e-14, 1.0, 10.0, 300.15), (0.2, -0.1, 0.15, 2e-13, 1.3, 3.0, 350.0), (-1.0, 0.3, -0.9, 1e-14, 1.0, 10.0, 250.0)):
    vt = N * P_K * T / P_Q
    vd = vai - vc
    ex = math.exp(vd / vt)
    idd, gd = IS * (ex - 1.0), IS * ex / vt
    ir, gr = (va - vai) / RS, 1.0 / RS
    cases.append({"module": "va_diode", "temperature_k": T, "v": [va, vc, vai],
                  "par_block": [IS, N, RS, 1e-12, 0.8, 0.5, 1e-9, 0.5, 1.0] + [1, 1, 1, 0, 0, 0, 0, 0, 0] + [0.0],
                  "I": [[0, ir], [1, -idd], [2, idd - ir]],
                  "G": [[0, 0, gr], [0, 2, -gr], [2, 0, -gr], [2, 2, gr + gd], [2, 1, -gd], [1, 2, -gd], [1, 1, gd]]})

# ---- va_mos1(d, g, s, b): square law with channel-length modulation, gmin = 0 here
# block: TYPE W L VTO KP LAMBDA COX CGSO CGDO
for (vd, vg, vs, vb, TYPE, W, L, VTO, KP, LAM) in ((1.5, 1.2, 0.0, 0.0, 1, 2e-6, 1e-6, 0.7, 1e-4, 0.02),      # saturation
                                                   (0.2, 1.8, 0.1, 0.0, 1, 2e-6, 1e-6, 0.7, 1e-4, 0.02),      # triode
                                                   (-1.0, -2.0, 0.3, 0.3, -1, 3e-6, 1e-6, 0.8, 4e-5, 0.05)):   # p-channel, saturation
    vgs, vds = TYPE * (vg - vs), TYPE * (vd - vs)
    assert vds >= 0.0
    vov, beta = vgs - VTO, KP * W / L
    if vov <= 0:
        ids = dg = dd = 0.0
    elif vds < vov:
        ids = beta * (vov - 0.5 * vds) * vds * (1 + LAM * vds)
        dg = beta * vds * (1 + LAM * vds)                                        # d ids / d vgs
        dd = beta * ((vov - vds) * (1 + LAM * vds) + (vov - 0.5 * vds) * vds * LAM)  # d ids / d vds
    else:
        ids = 0.5 * beta * vov * vov * (1 + LAM * vds)
        dg = beta * vov * (1 + LAM * vds)
        dd = 0.5 * beta * vov * vov * LAM
    i = TYPE * ids                       # I(d,s)
    # dI/dV(g) = TYPE*dg*TYPE = dg ; dI/dV(d) = dd ; dI/dV(s) = -(dg + dd)
    cases.append({"module": "va_mos1", "temperature_k": 300.15, "v": [vd, vg, vs, vb],
                  "par_block": [TYPE, W, L, VTO, KP, LAM, 3e-3, 0.0, 0.0] + [1, 1, 1, 1, 1, 1, 0, 0, 0] + [0.0],
                  "I": [[0, i], [2, -i], [1, 0.0], [3, 0.0]],
                  "G": [[0, 0, dd], [0, 1, dg], [0, 2, -(dg + dd)], [2, 0, -dd], [2, 1, -dg], [2, 2, dg + dd]]})

json.dump({"note": "hand-derived closed forms (tests/golden/make_va_closed_form.py); gmin = 0", "cases": cases},
          open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "va_closed_form.json"), "w"), indent=1)
print(len(cases), "cases")
