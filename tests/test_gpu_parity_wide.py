"""Round-2 additions to the GPU parity net: SimSpec fields `temp` and `scale`
(src/simulate_ir.jl:12-20, test/basic.jl:470-512), the damped / delayed / phase-shifted / cycle-limited sine
(src/spectre_env.jl:169-176), PWL corner semantics on the device (test/transients.jl:66-96) and a Verilog-A check that does not
pass through the compiler's own front end: hand-derived closed-form stamps of `va_diode` / `va_mos1`-style equations."""
import math

import numpy as np
import pytest

from cedarsim_jl_amd import PWL, SIN, Circuit, dc_opts, tran_opts
from cedarsim_jl_amd.workloads import dff_array, gf180_models, inverter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from cedarsim_jl_amd.engine import EngineCircuit, load_library
    load_library()
    return EngineCircuit


@pytest.fixture(scope="module")
def O(oracle_lib):
    from oracle_binding import Oracle
    return Oracle


def two_fets(temp=27.0, scale=1.0):
    c = Circuit(temp=temp, gmin=1e-12, scale=scale)
    m = gf180_models()
    n = c.add_model(*m["nfet_06v0"])
    p = c.add_model(*m["pfet_06v0"])
    c.V("vdd", "vdd", 0, dc=5.0)
    c.V("vin", "in", 0, dc=2.2)
    c.M("mn", "out", "in", 0, 0, n, 0.8e-6 / scale, 0.6e-6 / scale)
    c.M("mp", "out", "in", "vdd", "vdd", p, 1.2e-6 / scale, 0.5e-6 / scale)
    c.R("rl", "out", 0, 2e5)
    c.observe_node("out")
    return c


def test_bsim4_stamps_at_cold_and_hot_match_oracle(E, O):
    """BSIM4 at -40 and 125 C (temperature-dependent vth0, mobility, vsat, junctions — b4_pack) against the oracle's own set-up."""
    v = np.array([[3.0, 2.0, 0.1, -0.2], [1.0, 4.5, 4.9, 5.0]])
    for temp in (-40.0, 27.0, 125.0):
        c = two_fets(temp=temp)
        se, so = E(c).mos_eval(v), O(c).mos_eval(v)
        assert np.max(np.abs(se - so)) <= 1e-10 * np.max(np.abs(so)), temp
    cold, hot = E(two_fets(temp=-40.0)).mos_eval(v), E(two_fets(temp=125.0)).mos_eval(v)
    assert abs(cold[0, 0]) > 1.2 * abs(hot[0, 0])   # mobility: the cold device carries clearly more current at high Vgs


def test_temperature_sweep_as_samples_matches_per_point_oracle(E, O):
    """CH_SLOT_TEMP: five temperatures as five samples of ONE batched DC solve against five oracle solves (rtol 1e-6)."""
    temps = np.array([-40.0, 0.0, 27.0, 85.0, 125.0])
    c = two_fets()
    st = c.slot("temp")
    e = E(c)
    e.set_samples(len(temps))
    e.set_params([st], [temps])
    rc, x, status, _ = e.dc(dc_opts(abstol=1e-13))
    assert rc == 0 and not status.any()
    io = c.mna_index("v", "out")
    outs = []
    for k, tc in enumerate(temps):
        o = O(two_fets(temp=tc))
        rco, xo, _ = o.dc(dc_opts(abstol=1e-13))
        assert rco == 0
        assert abs(x[k, io] - xo[io]) <= 1e-6 * max(1.0, abs(xo[io])), tc
        outs.append(xo[io])
    assert max(outs) - min(outs) > 1e-3   # the operating point really moves with temperature


def test_option_scale_matches_pre_scaled_geometry(E, O):
    """.option scale (src/spectre.jl:1162-1176): W and L given in units of `scale` give the same circuit as metres with scale 1."""
    ref = E(two_fets()).dc(dc_opts(abstol=1e-13))
    for scale in (1e-6, 0.5):
        c = two_fets(scale=scale)
        rc, x, _, _ = E(c).dc(dc_opts(abstol=1e-13))
        rco, xo, _ = O(c).dc(dc_opts(abstol=1e-13))
        assert rc == 0 and rco == 0
        io = c.mna_index("v", "out")
        assert abs(x[0, io] - ref[1][0, io]) < 1e-9 and abs(x[0, io] - xo[io]) < 1e-6


def test_spsin_with_delay_damping_phase_and_cycle_limit(E):
    """spsin(vo, va, f, td, theta, phase, ncycles) across a resistor: the device-resident and the host stepper both reproduce
    vo + va*exp(-(t-td)*theta)*sind(360 f (t-td) + phase) for td < t < ncycles/f and vo + va*sind(phase) outside."""
    vo, va, f, td, th, ph, ncy = 0.3, 1.5, 2e5, 2e-6, 1.5e5, 40.0, 3.0
    c = Circuit()
    c.V("v", "a", 0, dc=vo + va * math.sin(math.radians(ph)), tran=SIN(vo, va, f, td, th, ph, ncy))   # the DC point is the value before the delay
    c.R("r1", "a", "b", 1e3)
    c.R("r2", "b", 0, 1e3)
    c.observe_node("b")
    e = E(c)
    sind = lambda d: np.sin(np.fmod(d, 360.0) * math.pi / 180.0)  # noqa: E731
    for stepper in ("host", "device"):
        # a purely resistive circuit has no local error: the integrator strides to dtmax, so the check is made AT the accepted
        # points (every one of them is an exact algebraic solve), not on an interpolated grid
        rc, t, v, _, st = e.tran(0.0, 2.2e-5, tran_opts(abstol=1e-9, reltol=1e-7, dtmax=2e-7, stepper=stepper))
        assert rc == 0 and len(t) > 100
        inside = (t > td) & (t < ncy / f)
        want = np.where(inside, vo + va * np.exp(-(t - td) * th) * sind(360.0 * f * (t - td) + ph), vo + va * sind(ph)) / 2
        m = (t != td) & (t != ncy / f)     # landing on a break point uses the source's left limit there
        assert np.max(np.abs(v[0, m, 0] - want[m])) < 1e-9, stepper
        assert np.any(t == td) and np.any(t == ncy / f)   # both break points are hit exactly


def test_pwl_corner_belongs_to_the_next_segment_on_the_device(E):
    """test/transients.jl:66-96: at a break point the source already follows the NEXT segment.  A current source into a
    capacitor integrates the waveform: q(t) = integral of the PWL, exact for a piecewise-linear input up to the tolerance."""
    ts = [0.0, 1e-6, 1e-6, 3e-6, 4e-6, 4e-6, 6e-6]
    ys = [0.0, 0.0, 1e-3, 1e-3, 0.0, -5e-4, -5e-4]   # two jumps (zero-width segments) and a ramp
    c = Circuit()
    c.I("i", 0, "a", dc=0.0, tran=PWL(ts, ys))
    c.C("c", "a", 0, 1e-9)
    c.R("r", "a", 0, 1e12)
    c.observe_node("a")
    sv = np.linspace(0.0, 6e-6, 601)
    # integral of the waveform (right-continuous at the jumps)
    def wave(t):
        if t < 1e-6:
            return 0.0
        if t < 3e-6:
            return 1e-3
        if t < 4e-6:
            return 1e-3 * (4e-6 - t) / 1e-6
        return -5e-4
    q = np.array([np.trapezoid([wave(u) for u in np.linspace(0.0, t, 4001)], np.linspace(0.0, t, 4001)) if t > 0 else 0.0 for t in sv])
    e = E(c)
    for stepper in ("host", "device"):
        rc, t, v, _, _ = e.tran(0.0, 6e-6, tran_opts(abstol=1e-9, reltol=1e-8, saveat=sv, stepper=stepper))
        assert rc == 0
        assert np.max(np.abs(v[0, :, 0] - q / 1e-9)) < 2e-3, stepper   # volts on a 0..2 V excursion (quadrature of the reference integral)


def test_compiled_va_stamps_against_hand_derived_closed_forms():
    """A Verilog-A check that shares nothing with va/frontend.py: the library's diode I = is*(exp(V/(n*vt)) - 1) + gmin-free
    conductance and a square-law MOS-level-1 style current, differentiated BY HAND, at three biases (values committed in
    tests/golden/va_closed_form.json)."""
    import json
    import os
    from cedarsim_jl_amd.engine import Context, load_library
    L = load_library()
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "va_closed_form.json")))
    ctx = Context(0)
    for case in g["cases"]:
        mid = L.ch_va_find(case["module"].encode())
        assert mid >= 0, case["module"]
        st = ctx.va_eval(mid, np.array(case["par_block"], float), case["v"], temperature_k=case["temperature_k"], gmin=0.0)
        I, G = st[0:8], st[16:80].reshape(8, 8)
        for (k, want) in case["I"]:
            assert abs(I[k] - want) <= 1e-12 * max(1.0, abs(want)) + 1e-25, (case["module"], "I", k, I[k], want)
        for (r, cc, want) in case["G"]:
            assert abs(G[r, cc] - want) <= 1e-11 * max(1.0, abs(want)) + 1e-25, (case["module"], "G", r, cc, G[r, cc], want)
