"""Small-signal AC and noise: the oracle against the reference's closed forms and its ngspice table.

Reference tests restated: test/ac.jl:17-57 (3rd-order Butterworth low-pass: vout = 1/((s+1)(s²+s+1)),
vin = 1, l3.V = s·L3·H) and test/ac.jl:62-148 (output noise of R4 ∥ R5 at 23 °C: analytic PSD and the
61-point ngspice table, rtol 1e-6).  CPU only."""
import json
import math
import os

import numpy as np

from cedarsim_jl_amd import acdec, parse_spice
from oracle_binding import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
L1, C2, L3, R4 = 1.5, 4.0 / 3.0, 0.5, 1.0

BUTTERWORTH = """*Third order low pass filter, butterworth, with wc = 1
.param res=%g
V1 vin 0 AC 1 SIN (0, 1, %r)
L1 vin n1 %r
C2 n1 0 %r
L3 n1 vout %r
R4 vout 0 '2*res'
R5 vout 0 '2*res'
""" % (R4, 1 / (2 * math.pi), L1, C2, L3)


def butterworth_circuit():
    ckt = parse_spice(BUTTERWORTH).build()
    ckt.temp, ckt.gmin = 23.0, 0.0
    return ckt


def test_acdec_matches_reference_definition():
    f = acdec(20, 0.01, 10)          # test/ac.jl:38 → 61 points, 20 per decade
    assert len(f) == 61 and abs(f[0] - 0.01) < 1e-15 and abs(f[-1] - 10) < 1e-12
    assert np.allclose(f[20] / f[0], 10.0)


def test_netlist_reads_ac_magnitude():
    ckt = butterworth_circuit()
    assert ckt.source_ac == [1.0]
    assert ckt.to_desc().src_ac[0] == 1.0


def test_oracle_ac_butterworth_closed_form():
    ckt = butterworth_circuit()
    o = Oracle(ckt)
    f = acdec(20, 0.01, 10)
    rc, x = o.ac(f)
    assert rc == 0
    s = 2j * math.pi * f
    H = 1.0 / ((s + 1) * (s * s + s + 1))
    vout, vin, n1 = (x[:, ckt._n(n) - 1] for n in ("vout", "vin", "n1"))
    assert np.allclose(vout, H, rtol=1e-9, atol=0)           # test/ac.jl:47
    assert np.allclose(vin, 1.0, rtol=1e-12)                  # :49
    assert np.allclose(n1 - vout, s * L3 * H, rtol=1e-9)      # :61-64 (sys.l3.V)
    # bode: magnitude and phase (:53-58)
    assert np.allclose(np.abs(vout), np.abs(H), rtol=1e-9) and np.allclose(np.angle(vout), np.angle(H), atol=1e-9)


def test_oracle_noise_matches_analytic_and_ngspice():
    ckt = butterworth_circuit()
    o = Oracle(ckt)
    gold = json.load(open(os.path.join(HERE, "golden", "ac_butterworth_noise_ngspice.json")))
    f = np.array([r[0] for r in gold["rows"]])
    ng = np.array([r[1] for r in gold["rows"]])
    assert np.allclose(f, acdec(20, 0.01, 10), rtol=1e-6)
    rc, psd = o.noise(ckt._n("vout") - 1, acdec(20, 0.01, 10))
    assert rc == 0
    # analytic (test/ac.jl:71-81): H = (sL1 ∥ 1/sC2 + sL3) ∥ R4, Hn = 4kT/R4 · H²
    s = 2j * math.pi * acdec(20, 0.01, 10)
    par = lambda a, b: a * b / (a + b)  # noqa: E731
    H = par(par(s * L1, 1 / (s * C2)) + s * L3, R4)
    k, T = 1.380649e-23, 23 + 273.15
    apsd = np.sqrt(np.abs(4 * k * T / R4 * H * H))
    assert np.allclose(apsd, ng, rtol=1e-6)                   # :147 (pins the fixture itself)
    assert np.allclose(np.sqrt(psd), apsd, rtol=1e-6)         # :148
    assert np.allclose(np.sqrt(psd), ng, rtol=1e-6)


def test_bsimcmg_inverter_noise_matches_reference_ngspice_table():
    """test/ac.jl:155-237: output noise at node q of the ASAP7 BSIM-CMG inverter against the 61-point ngspice table
    (rtol 1e-6 in the reference).  This pins the compiled BSIM-CMG 107 device arithmetic — operating point, ∂i/∂v,
    ∂q/∂v and the thermal / flicker / shot noise powers — against an independent simulator."""
    from cedarsim_jl_amd import dc_opts
    from cedarsim_jl_amd.va.registry import load_modules
    from cedarsim_jl_amd.workloads import cmg_inverter_array
    if "bsimcmg" not in load_modules()[1]:
        import pytest
        pytest.skip("bsimcmg was not in the model library build")
    gold = json.load(open(os.path.join(HERE, "golden", "ac_bsimcmg_inverter_noise_ngspice.json")))
    f = np.array([r[0] for r in gold["rows"]])
    ng = np.array([r[1] for r in gold["rows"]])
    assert np.allclose(f, acdec(5, 1e3, 1e15), rtol=1e-8)
    c = cmg_inverter_array(1, json.load(open(os.path.join(HERE, "golden", "asap7_tt_lvt_cards.json")))["cards"])
    rc, psd = Oracle(c).noise(c._n("q") - 1, f, dc_opts(abstol=1e-12))
    assert rc == 0
    assert np.allclose(np.sqrt(psd), ng, rtol=1e-6)
