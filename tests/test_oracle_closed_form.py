"""Pins the CPU oracle against the closed-form known answers of the reference's own tests
(tests/golden/closed_form.json).  CPU only."""
import json
import math
import os

import numpy as np
import pytest

from cedarsim_jl_amd import (PWL, SIN, Circuit, dc_opts, parse_spice, tran_opts)
from oracle_binding import Oracle

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "closed_form.json")))
DEFTOL = 1e-7  # test/common.jl:21-24 isapprox_deftol


def approx(a, b, tol=DEFTOL):
    return abs(a - b) <= max(tol, tol * max(abs(a), abs(b)))


def test_vr():  # test/basic.jl:21-43
    g = G["vr"]
    c = Circuit()
    c.V("V", "vcc", 0, dc=g["V"])
    c.R("R", "vcc", 0, g["R"])
    rc, x, _ = Oracle(c).dc()
    assert rc == 0
    assert approx(x[c.mna_index("v", "vcc")], g["R.V"])
    assert approx(x[c.mna_index("v", "vcc")] / g["R"], g["R.I"])
    assert approx(x[c.mna_index("i", "V")], -g["R.I"])  # branch current flows + to - through the source


def test_ir():  # test/basic.jl:81-106 (SPICE sign convention of the current source)
    g = G["ir"]
    c = Circuit()
    c.I("I", "icc", 0, dc=g["I"])
    c.R("R", "icc", 0, g["R"])
    rc, x, _ = Oracle(c).dc()
    assert rc == 0 and approx(x[0], g["R.V"]) and approx(x[0] / g["R"], g["R.I"])


@pytest.mark.parametrize("m", [1, 10])
def test_vrc_and_parallel_instances(m):  # test/basic.jl:108-166
    g = G["vrc"]
    c = Circuit()
    c.V("V", "vcc", 0, dc=g["V"])
    c.R("R", "vcc", "vrc", g["R"], m=m)
    c.C("C", "vrc", 0, g["C"])
    c.observe_node("vrc")
    c.observe_node("vcc")
    o = Oracle(c)
    x0 = np.zeros(o.n)
    x0[c.mna_index("v", "vcc")] = g["V"]  # u0 = [0.0]: capacitor starts uncharged
    rc, t, v, xf, st = o.tran(0.0, 1.0, tran_opts(abstol=1e-9, reltol=1e-9, skip_dc=True, dc=dc_opts(x0=x0)))
    assert rc == 0
    ic0 = m * (v[1][0] - v[0][0]) / g["R"]
    assert approx(ic0, m * g["C.I(0)"])
    assert approx(v[0][0], 0.0) and approx(v[0][-1], g["C.V(end)"])
    assert approx(m * (v[1][-1] - v[0][-1]) / g["R"], 0.0)


def test_two_resistor_sweep():  # test/sweep.jl:326-340: 400 points, I = -1/(R1+R2)
    g = G["two_resistor_sweep"]
    c = Circuit()
    c.V("V", "vcc", 0, dc=1.0)
    c.R("R1", "vcc", "mid", 100.0)
    c.R("R2", "mid", 0, 100.0)
    s1, s2 = c.slot("R1"), c.slot("R2")
    o = Oracle(c)
    for r1 in g["R1"]:
        for r2 in g["R2"]:
            o.set_param(s1, r1)
            o.set_param(s2, r2)
            rc, x, _ = o.dc()
            assert rc == 0 and approx(x[c.mna_index("i", "V")], -1.0 / (r1 + r2))


def test_pwl_current_into_resistor():  # test/transients.jl:17-63
    g = G["pwl_ir"]
    nl = parse_spice("""* PWL test
.param pval=-1
i1 vout 0 PWL(1m 0 9m 'pval*%g')
R1 vout 0 r=%g
""" % (g["i_max"], g["r"]))
    c = nl.build()
    c.observe_node("vout")
    rc, t, v, _, _ = Oracle(c).tran(0.0, 10e-3, tran_opts(abstol=1e-8, reltol=1e-8))
    assert rc == 0 and len(t) > 5
    pw = np.clip((t - g["t0"]) / (g["t1"] - g["t0"]), 0, 1)
    assert np.max(np.abs(v[0] - pw * g["i_max"] * g["r"])) < DEFTOL
    assert any(abs(t - g["t0"]) < 1e-15) and any(abs(t - g["t1"]) < 1e-15)  # break points are hit exactly


def test_pwl_break_point_semantics():  # test/transients.jl:66-96: the corner belongs to the next segment
    g = G["pwl_slope"]
    c = Circuit()
    c.V("V", "a", 0, tran=PWL(g["ts"], g["ys"]))
    c.R("R", "a", 0, 1.0)
    o = Oracle(c)
    for t, want in zip(g["t"], g["dydt"]):
        h = 1e-12
        slope = (o.source_value(0, t + h) - o.source_value(0, t)) / h  # right derivative at t
        assert abs(slope - want) <= 1e-3 * max(1.0, abs(want))


def test_butterworth_closed_form():  # test/transients.jl:98-173
    g = G["butterworth"]
    c = Circuit()
    c.V("V1", "vin", 0, tran=SIN(0, 1, 1 / (2 * math.pi)))
    c.L("L1", "vin", "n1", g["L1"])
    c.C("C2", "n1", 0, g["C2"])
    c.L("L3", "n1", "vout", g["L3"])
    c.R("R4", "vout", 0, g["R4"])
    c.observe_node("vout")
    rc, t, v, _, st = Oracle(c).tran(0.0, 100.0, tran_opts(abstol=1e-9, reltol=1e-9, skip_dc=True))
    assert rc == 0
    an = (np.exp(-t) - np.sin(t) - np.cos(t)) / 2 + 2 * np.sin(np.sqrt(3) * t / 2) / (np.sqrt(3) * np.sqrt(np.exp(t)))
    assert np.max(np.abs(v[0] - an)) < DEFTOL
    half = v[0][len(t) // 2:]
    assert abs(math.sqrt(np.mean(half ** 2)) - 0.5) < 0.1 + 0.05  # RMS 0.5 ± 0.1 (test/transients.jl:150)
    # dense output at the golden sample times
    rc, t2, v2, _, _ = Oracle(c).tran(0.0, 100.0, tran_opts(abstol=1e-9, reltol=1e-9, skip_dc=True, saveat=np.array(g["samples_t"])))
    assert np.max(np.abs(v2[0] - np.array(g["samples_vout"]))) < 1e-6


def test_multiplicities():  # test/basic.jl:556-595: every divider node == 10/11
    spice = """* multiplicities
v1 vcc 0 DC 1
r1a vcc 1 1 m=10
r1b 1 0 1
.subckt r10 a b m=10
r2a a b 1
.ends
x2a vcc 2 r10
r2b 2 0 1
x3a1 vcc 3 r10 m=5
x3a2 vcc 3 r10 m=5
r3b 3 0 1
.subckt r5t2 a b
x5r1 a b r10 m=5
x5r2 a b r10 m=5
.ends
x4a1 vcc 4 r5t2
r4b 4 0 1
.subckt r2 a b
r2 a b 1 m=2
.ends
x5a vcc 5 r2 m=5
r5b 5 0 1
.model rm r R=1
r6a vcc 6 rm m=10 l=1u
r6b 6 0 1
"""
    c = parse_spice(spice).build()
    rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0
    for n in "123456":
        assert abs(x[c.mna_index("v", n)] - G["multiplicity"]["expect"]) < 1e-12


def test_spice_sources_and_controlled_sources():  # test/basic.jl:207-235 style: B/E/G sources
    spice = """* sources
v1 1 0 2
e1 2 0 1 0 2
r2 2 0 1
g1 3 0 1 0 2
r3 3 0 1k
b1 4 0 v=4
r4 4 0 1
"""
    c = parse_spice(spice).build()
    rc, x, _ = Oracle(c).dc()
    assert rc == 0
    assert approx(x[c.mna_index("v", "1")], 2.0) and approx(x[c.mna_index("v", "2")], 4.0)
    assert approx(x[c.mna_index("v", "4")], 4.0) and approx(x[c.mna_index("v", "3")], -4000.0)


def test_dc_init_of_time_varying_source():  # test/basic.jl:534-554: dc=5 used in :dcop, SIN offset at t=0 in tran
    c = Circuit()
    c.V("V", "vcc", 0, dc=5.0, tran=SIN(10, 3, 1e3))
    c.R("R", "vcc", 0, 1.0)
    c.observe_node("vcc")
    o = Oracle(c)
    rc, x, _ = o.dc()
    assert approx(x[0], 5.0)
    rc, x, _ = o.dc(dc_opts(tran_mode=True))
    assert approx(x[0], 10.0)
    rc, t, v, _, _ = o.tran(0.0, 1e-3, tran_opts(abstol=1e-6, reltol=1e-6))
    assert rc == 0 and approx(v[0][-1], 10.0 + 3 * math.sin(2 * math.pi * 1.0), 1e-6)


def test_singular_circuit_reports_error():
    c = Circuit()
    c.I("I", "a", 0, dc=1.0)
    c.C("C", "a", 0, 1e-9)  # no DC path: singular G
    rc, x, _ = Oracle(c).dc(dc_opts(n_restarts=2, maxiters=5))
    assert rc != 0


def test_remaining_basic_jl_netlist_cases_solve_to_their_closed_forms():
    """test/basic.jl:686-737 on the oracle: `device == param` (a resistor named like its parameter inside a subcircuit,
    with the top-level x1 overridden), the semiconductor resistor (rsh·l/w = 1 kΩ) next to a parameterised one,
    `.model` case-insensitivity (:597-607) and the `.option` line (:640-649)."""
    from cedarsim_jl_amd import parse_spice
    c = parse_spice("""* device == param
.param x1=1
.subckt myres p n
    .param rload=1k
    rload p n 'rload*x1'
.ends
i1 vcc 0 DC -1
x1 vcc 0 myres
""").build(x1=2.0)
    rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0 and x[c._n("vcc") - 1] == pytest.approx(2000.0, rel=1e-12)       # sol[sys.x1.rload.V] with x1 = 2
    c = parse_spice("""* semiconductor resistor
.model myres r rsh=500
.param res=1k
v1 vcc 0 1
R1 vcc 0 myres w=1m l=2m
R2 vcc 0 res
""").build()
    rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0 and x[c.mna_index("i", "v1")] == pytest.approx(-2e-3, rel=1e-12)   # r1.I = r2.I = 1 mA
    assert c.dev_par[c.dev_names.index("r1")][0] == pytest.approx(1000.0)
    c = parse_spice("* .option\n.option temp=10 filemode=ascii noinit\n").build()
    assert c.temp == 10.0 and c.dev_names == []
    c = parse_spice("""* .model case sensitivity
.MODEL MyRes R RSH=500
V1 vcc 0 1
r1 VCC 0 myres W=1m L=2m
""").build()
    rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0 and x[c.mna_index("i", "v1")] == pytest.approx(-1e-3, rel=1e-12)
