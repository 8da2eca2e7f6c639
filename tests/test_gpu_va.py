"""GPU parity tests of compiled Verilog-A devices (CH_DEV_VA through the C-ABI): stamp-level against the
host instantiation and the Python interpreter, circuit-level against the reference's VA tests
(test/ddx.jl, test/varegress.jl, test/basic.jl:359-381, test/bsimcmg/inverter.jl) and the CPU oracle."""
import ctypes as C
import json
import math

import numpy as np
import pytest

from cedarsim_jl_amd import PULSE, SIN, Circuit, dc, dc_opts, parse_spice, tran, tran_opts
from cedarsim_jl_amd.va.interp import Interp
from cedarsim_jl_amd.va.registry import find_module, load_modules

pytestmark = pytest.mark.gpu
_pd = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def E():
    from cedarsim_jl_amd.engine import EngineCircuit, load_library
    load_library()
    return EngineCircuit


@pytest.fixture(scope="module")
def ctx():
    from cedarsim_jl_amd.engine import default_context
    return default_context()


@pytest.fixture(scope="module")
def O(oracle_lib):
    from oracle_binding import Oracle
    return Oracle


def _pblock(mod, params, temp_c=27.0):
    it = Interp(mod, params, temperature_c=temp_c)
    P = np.array([float(it.params[p[0]]) if p[1] != "string" else 0.0 for p in mod.params] + [1.0 if p[0] in it.given else 0.0 for p in mod.params] + [0.0])
    return it, P


def _stamp_parity(ctx, oracle_lib, name, params, biases, tol):
    mid, mod = find_module(name)
    assert ctx.L.ch_va_find(mod.name.encode()) == mid and ctx.L.ch_va_module_name(mid).decode() == mod.name
    it, P = _pblock(mod, params, 35.0)
    n = len(mod.nodes)
    for vb in biases:
        v = np.zeros(8)
        v[:n] = [vb.get(x, 0.0) for x in mod.nodes]
        got = ctx.va_eval(mid, P, v[:n], 35.0 + 273.15, 1e-12)
        ref = np.zeros(144)
        assert oracle_lib.oracle_va_eval(mid, P.ctypes.data_as(_pd), v.ctypes.data_as(_pd), 35.0 + 273.15, 1e-12, ref.ctypes.data_as(_pd)) == 0
        for lo, hi in ((0, 8), (8, 16), (16, 80), (80, 144)):
            scale = max(np.abs(ref[lo:hi]).max(), 1e-300)
            assert np.allclose(got[lo:hi], ref[lo:hi], rtol=tol, atol=tol * scale), (name, vb, lo)
        I, Q, G, Cc = it.evaluate(vb)   # independent evaluator
        assert np.allclose(got[:n], I, rtol=1e-8, atol=1e-8 * max(np.abs(I).max(), 1e-300))


def test_va_stamps_match_host_and_interpreter(ctx, oracle_lib):
    rng = np.random.default_rng(3)
    for name, params in (("va_resistor", {"R": 2e3}), ("va_nlvcr", {"R": 2.0}), ("va_diode", {"LEVEL": 2, "RS": 3.0}), ("va_mos1", {"TYPE": -1})):
        _, mod = find_module(name)
        _stamp_parity(ctx, oracle_lib, name, params, [{x: float(rng.uniform(-0.3, 0.9)) for x in mod.nodes} for _ in range(4)], 1e-12)


def test_bsimcmg_stamps_match_host_and_interpreter(ctx, oracle_lib):
    if "bsimcmg" not in load_modules()[1]:
        pytest.skip("bsimcmg was not in the model library build")
    rng = np.random.default_rng(4)
    _, mod = find_module("bsimcmg")
    for params in ({"DEVTYPE": 1, "L": 2e-8, "NFIN": 2, "IGCMOD": 1, "GIDLMOD": 1}, {"DEVTYPE": 0, "L": 3e-8}):
        biases = []
        for _ in range(3):
            b = {x: float(rng.uniform(-0.2, 0.8)) for x in mod.nodes}
            b["di"], b["si"] = b["d"] + 1e-3, b["s"] - 1e-3
            biases.append(b)
        _stamp_parity(ctx, oracle_lib, "bsimcmg", params, biases, 1e-9)


def test_reference_va_circuit_tests_on_the_gpu():
    # test/basic.jl:370-381
    sol = dc(parse_spice('* Verilog Include 2\n.hdl "cedar_basic.va"\nx1 vcc 0 va_resistor r=2k\nv1 vcc 0 dc=1\n'), abstol=1e-14)
    assert sol.rc == 0 and sol["v1.i"][0] == pytest.approx(-1 / 2e3, rel=1e-12)
    # test/ddx.jl:21
    c = Circuit()
    c.V("v1", "vcc", 0, dc=5.0)
    c.V("v2", "vg", 0, dc=3.0)
    c.VA("r", "va_nlvcr", ["vcc", "vg", 0], {"R": 2.0})
    sol = dc(c, abstol=1e-14)
    assert sol.rc == 0 and sol["v1.i"][0] == pytest.approx(-5 * 2 * 2 * 3, rel=1e-12)
    # test/varegress.jl
    for mod in ("va_resistor", "va_resistor_rev"):
        c = Circuit()
        c.V("v", "vcc", 0, dc=1.0)
        c.VA("r", mod, ["vcc", "out"], {"R": 1000.0})
        c.C("c", "out", 0, 1e-9)
        sol = tran(c, (0.0, 1e-5), abstol=1e-9, reltol=1e-6, u0=np.zeros(c.n_mna), initializealg="none") if False else None
        from cedarsim_jl_amd.engine import EngineCircuit
        c.observe_node("out")
        rc, t, v, xf, st = EngineCircuit(c).tran(0.0, 1e-5, tran_opts(abstol=1e-9, reltol=1e-6, skip_dc=1))
        assert rc == 0
        vout = v[0, :, 0]
        assert np.all((1.0 - vout) / 1000.0 >= -1e-12)
        assert vout[-1] == pytest.approx(1 - math.exp(-10.0), rel=1e-4)


def rectifier():
    c = Circuit(gmin=1e-12)
    c.V("vin", "in", 0, tran=SIN(0.0, 2.0, 1e6))
    c.VA("d1", "va_diode", ["in", "out"], {"IS": 1e-13, "RS": 5.0, "CJ0": 2e-12, "LEVEL": 2, "TT": 2e-9})
    c.R("rl", "out", 0, 1e3)
    c.C("cl", "out", 0, 2e-9)
    c.observe_node("out")
    c.observe_node("d1.ai")
    return c


def test_va_diode_rectifier_transient_matches_oracle(E, O):
    c = rectifier()
    ts = np.linspace(0, 3e-6, 121)
    opts = lambda: tran_opts(abstol=1e-9, reltol=1e-7, saveat=ts)  # noqa: E731
    rc_o, t_o, v_o, _, _ = O(c).tran(0.0, 3e-6, opts())
    rc, t, v, xf, st = E(c).tran(0.0, 3e-6, opts())
    assert rc == 0 and rc_o == 0
    assert v[0, :, 0].max() > 0.8           # it rectifies
    v_o = v_o if v_o.ndim == 3 else v_o[:, :, None]
    assert np.allclose(v[:, :, 0], v_o[:, :, 0], rtol=0, atol=1e-4 * 2.0)


def mixed_inverter():
    """VA square-law NMOS pull-down + BSIM4 PMOS pull-up + VA capacitor load: legacy and compiled devices in one block."""
    from cedarsim_jl_amd.workloads import gf180_models
    c = Circuit(gmin=1e-12)
    m = gf180_models()
    p = c.add_model(*m["pfet_06v0"])
    c.V("vdd", "vdd", 0, dc=5.0)
    c.V("vin", "in", 0, tran=PULSE(0.0, 5.0, 2e-9, 1e-9, 1e-9, 10e-9, 30e-9))
    c.R("rg", "in", "g", 100.0)
    c.VA("mn", "va_mos1", ["out", "g", 0, 0], {"W": 2e-6, "L": 6e-7, "KP": 1.2e-4, "VTO": 0.8, "CGSO": 2e-10, "CGDO": 2e-10})
    c.M("mp", "out", "g", "vdd", "vdd", p, 4e-6, 5e-7)
    c.VA("cl", "va_capacitor", ["out", 0], {"C": 2e-14})
    c.observe_node("out")
    c.observe_node("g")
    return c


def test_mixed_va_and_bsim4_block_matches_oracle(E, O):
    c = mixed_inverter()
    x_o = O(c).dc(dc_opts(abstol=1e-12))[1]
    rc, x, status, st = E(c).dc(dc_opts(abstol=1e-12))
    assert rc == 0
    assert np.allclose(x[0][:c.n_nodes], x_o[:c.n_nodes], rtol=1e-6, atol=1e-9)
    ts = np.linspace(0, 60e-9, 241)
    opts = lambda: tran_opts(abstol=1e-8, reltol=1e-6, saveat=ts)  # noqa: E731
    rc_o, t_o, v_o, _, _ = O(c).tran(0.0, 60e-9, opts())
    rc, t, v, xf, st = E(c).tran(0.0, 60e-9, opts())
    assert rc == 0 and rc_o == 0
    v_o = v_o if v_o.ndim == 3 else v_o[:, :, None]
    out = v[0, :, 0]
    assert out.max() > 4.5 and out.min() < 0.5          # it switches rail to rail
    assert np.allclose(v[:, :, 0], v_o[:, :, 0], rtol=0, atol=1e-4 * 5.0)


def cmg_inverter_netlist():
    # test/bsimcmg/inverter_cmg_cedar.cir:6-14 with in-line cards (the ASAP7 card file is not used here: DESIGN.md)
    return """* BSIM-CMG inverter
.model nmos_lvt nmos level=72 l=2.1e-8 nfin=2 tfin=6.5e-9 hfin=3.2e-8 eot=1e-9 phig=4.3 igcmod=1 gidlmod=1
.model pmos_lvt pmos level=72 l=2.1e-8 nfin=3 tfin=6.5e-9 hfin=3.2e-8 eot=1e-9 phig=4.8
mneg Q D VSS VSS nmos_lvt
mpos Q D VDD VDD pmos_lvt
VVDD VDD 0 1.0
VVSS VSS 0 0.0
CQ D 0 1e-15
VD D 0 AC 1 SIN (0.5 0.4 1e7)
.TRAN 1e-9 4.0e-7
.END
"""


def test_bsimcmg_inverter_runs_and_matches_oracle(E, O):
    """test/bsimcmg/inverter.jl:22 asserts only `retcode == Success`; here additionally engine == oracle."""
    if "bsimcmg" not in load_modules()[1]:
        pytest.skip("bsimcmg was not in the model library build")
    c = parse_spice(cmg_inverter_netlist()).build()
    c.observe_node("q")
    c.observe_node("d")
    ts = np.linspace(0, 4e-7, 161)
    opts = lambda: tran_opts(abstol=1e-7, reltol=1e-7, saveat=ts, dc=dc_opts(abstol=1e-10, tran_mode=1))  # noqa: E731
    rc, t, v, xf, st = E(c).tran(0.0, 4e-7, opts())
    assert rc == 0                                    # ReturnCode.Success
    q = v[0, :, 0]
    assert q.max() > 0.9 and q.min() < 0.1            # the inverter switches
    rc_o, t_o, v_o, _, _ = O(c).tran(0.0, 4e-7, opts())
    assert rc_o == 0
    v_o = v_o if v_o.ndim == 3 else v_o[:, :, None]
    assert np.allclose(v[:, :, 0], v_o[:, :, 0], rtol=0, atol=1e-4 * 1.0)   # 1e-4 of the 1 V swing


def test_config5_bsimcmg_asap7_inverter_array(E, O):
    """128 BSIM-CMG inverters (256 instances) with the ASAP7 TT cards the reference's parser
    tests hold (SpectreNetlistParser.jl/test/examples/7nm_TT.scs → tests/golden/asap7_tt_lvt_cards.json) and the deck of
    test/bsimcmg/inverter_cmg_cedar.cir (VDD = 1, SIN(0.5 0.01 1e7), 4e-7 s, abstol = reltol = 1e-7 as in
    test/bsimcmg/inverter.jl:20).  The reference asserts `retcode == Success`; here also tile 0 == oracle and
    identical tiles give identical waveforms."""
    import os
    import time
    from cedarsim_jl_amd.workloads import CMG_TSPAN, cmg_inverter_array
    if "bsimcmg" not in load_modules()[1]:
        pytest.skip("bsimcmg was not in the model library build")
    cards = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "asap7_tt_lvt_cards.json")))["cards"]
    ts = np.linspace(0, 4e-7, 81)
    opts = lambda: tran_opts(abstol=1e-7, reltol=1e-7, saveat=ts, dc=dc_opts(abstol=1e-10, tran_mode=1))  # noqa: E731
    c = cmg_inverter_array(128, cards, observe="q")
    eng = E(c)
    info = eng.info()
    assert info["n_mos"] == 0 and info["n_components"] == 128 and info["max_component"] == 5
    t0 = time.perf_counter()
    rc, t, v, xf, st = eng.tran(CMG_TSPAN[0], CMG_TSPAN[1], opts())
    wall = time.perf_counter() - t0
    assert rc == 0                                                     # ReturnCode.Success
    assert np.abs(v[:, :, 0] - v[0:1, :, 0]).max() < 1e-12             # identical tiles
    rc_o, t_o, v_o, _, st_o = O(cmg_inverter_array(1, cards, observe="q")).tran(CMG_TSPAN[0], CMG_TSPAN[1], opts())
    v_o = v_o if v_o.ndim == 3 else v_o[:, :, None]
    assert rc_o == 0
    assert np.allclose(v[0, :, 0], v_o[0, :, 0], rtol=0, atol=1e-4)
    print("config5: 128 inverters, %d accepted / %d rejected steps, %.3f s wall, %d block iterations, stepper %d mode %d" % (st["naccept"], st["nreject"], wall, st["n_block_iters"], st["stepper"], st["stepper_mode"]))


def test_sweep_over_verilog_a_parameters_is_batched(E, O):
    """CircuitSweep over parameters of compiled modules (ParamSim fields of a VA device, src/circuitodesystem.jl:66-97):
    all points become samples of one batched solve through CH_SLOT_VA_PAR; each sample must equal its own oracle run."""
    from cedarsim_jl_amd import CircuitSweep, ProductSweep

    def build(r=1e3, isat=1e-14):
        c = Circuit(gmin=1e-12)
        c.V("v1", "in", 0, dc=1.5)
        c.VA("rs", "va_resistor", ["in", "a"], {"R": r})
        c.VA("d1", "va_diode", ["a", 0], {"IS": isat, "RS": 2.0})
        c.observe_node("a")
        return c

    cs = CircuitSweep(build, ProductSweep(r=[500.0, 1e3, 2e3, 4e3], isat=[1e-15, 1e-13]))
    sols = dc(cs, abstol=1e-13)
    assert len(sols) == 8
    va = []
    for sol, p in zip(sols, cs):
        ref = O(build(**p)).dc(dc_opts(abstol=1e-13))[1]
        assert sol.rc == 0
        assert sol["node_a"][0] == pytest.approx(ref[build(**p)._n("a") - 1], rel=1e-6)
        va.append(sol["node_a"][0])
    assert len(set(np.round(va, 9))) == 8            # every point really got its own parameters
    # explicit slot on a VA instance parameter
    c = build()
    slot = c.slot("rs", "r")
    eng = E(c)
    eng.set_samples(3)
    eng.set_params([slot], [[500.0, 1e3, 4e3]])
    rc, x, status, st = eng.dc(dc_opts(abstol=1e-13))
    assert rc == 0 and x[0][c._n("a") - 1] > x[1][c._n("a") - 1] > x[2][c._n("a") - 1]


def cmg_chain(n_stages):
    """Chain of BSIM-CMG inverters (ASAP7 cards): one coupled Jacobian block of 5 unknowns per stage."""
    import os
    from cedarsim_jl_amd.netlist import parse_spice as ps
    cards = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "asap7_tt_lvt_cards.json")))["cards"]
    lines = ["* chain", "VVDD VDD 0 0.7", "VIN n0 0 PULSE(0 0.7 0.1n 20p 20p 0.4n 1n)"]
    for k in range(n_stages):
        lines.append("mn%d n%d n%d 0 0 nmos_lvt" % (k, k + 1, k))
        lines.append("mp%d n%d n%d VDD VDD pmos_lvt" % (k, k + 1, k))
        lines.append("cw%d n%d 0 2e-16" % (k, k + 1))
    nl = ps("\n".join(lines) + "\n.END\n")
    nl.add_model_cards(cards)
    c = nl.build()
    for k in (1, n_stages // 2, n_stages):
        c.observe_node("n%d" % k)
    return c


@pytest.mark.parametrize("n_stages,path", [(10, 1), (16, 2)])
def test_coupled_bsimcmg_chain_dense_lds_and_sparse_paths_match_oracle(E, O, n_stages, path):
    """A coupled block of compiled devices: 10 stages (50 unknowns) runs in the fused kernel with the LDS LU, 16 stages
    (80 unknowns) takes the sparse path; both against the oracle."""
    if "bsimcmg" not in load_modules()[1]:
        pytest.skip("bsimcmg was not in the model library build")
    c = cmg_chain(n_stages)
    eng = E(c)
    ts = np.linspace(0, 1.2e-9, 61)
    opts = lambda: tran_opts(abstol=1e-7, reltol=1e-6, saveat=ts, dc=dc_opts(abstol=1e-10, tran_mode=1))  # noqa: E731
    rc, t, v, xf, st = eng.tran(0.0, 1.2e-9, opts())
    assert rc == 0, eng.ctx.last_error()
    assert eng.info()["path"] == path and eng.info()["max_component"] == 5 * n_stages
    rc_o, t_o, v_o, _, _ = O(c).tran(0.0, 1.2e-9, opts())
    assert rc_o == 0
    v_o = v_o if v_o.ndim == 3 else v_o[:, :, None]
    last = v[2, :, 0]
    assert last.max() > 0.6 and last.min() < 0.1       # the edge propagates to the last stage
    assert np.allclose(v[:, :, 0], v_o[:, :, 0], rtol=0, atol=1e-4 * 0.7)


def test_voltage_contributions_on_the_gpu(E, O):
    c = Circuit()
    c.V("v1", "in", 0, dc=1.0)
    c.R("r1", "in", "a", 100.0)
    c.VA("l1", "va_inductor", ["a", 0], {"L": 1e-3, "RS": 0.0})
    c.observe_node("a")
    c.observe_node("l1.i(p,n)")
    rc, t, v, xf, st = E(c).tran(0.0, 5e-5, tran_opts(abstol=1e-10, reltol=1e-7, skip_dc=1))
    assert rc == 0
    assert np.allclose(v[1, :, 0], 1.0 / 100.0 * (1 - np.exp(-t / 1e-5)), rtol=1e-4, atol=1e-8)
    c = Circuit()
    c.V("vc", "c", 0, dc=0.25)
    c.VA("e1", "va_vcvs", ["out", 0, "c", 0], {"VDC": 1.5, "GAIN": -2.0})
    c.R("rl", "out", 0, 50.0)
    rc, x, status, st = E(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0 and x[0][c._n("out") - 1] == pytest.approx(1.0, rel=1e-12)
    assert x[0][c._n("e1.i(p,n)") - 1] == pytest.approx(-1.0 / 50.0, rel=1e-12)


def test_operating_point_observables_on_the_gpu(ctx):
    """`sol.op(dev)` ≙ `sol[sys.dev.var]` for (* desc *) variables: evaluated on the GPU at the DC solution."""
    c = Circuit()
    c.V("vd", "d", 0, dc=2.0)
    c.V("vg", "g", 0, dc=1.7)
    c.VA("m1", "va_mos1", ["d", "g", 0, 0], {"KP": 2e-4, "W": 2e-6, "L": 1e-6, "VTO": 0.7, "LAMBDA": 0.0})
    sol = dc(c, abstol=1e-14)
    op = sol.op("m1")
    assert op["REGION"] == 2 and op["VOV"] == pytest.approx(1.0, rel=1e-12) and op["IDS"] == pytest.approx(2e-4, rel=1e-12)
    assert sol["vd.i"][0] == pytest.approx(-(2e-4 + 1e-12 * 2.0), rel=1e-9)
    if "bsimcmg" in load_modules()[1]:
        from cedarsim_jl_amd.workloads import cmg_inverter_array
        import os
        cards = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "asap7_tt_lvt_cards.json")))["cards"]
        ci = cmg_inverter_array(1, cards)
        s2 = dc(ci, abstol=1e-12)
        opn = s2.op("mneg0")
        mod, par = ci.va_instances["mneg0"]
        it = Interp(mod, par, temperature_c=ci.temp, gmin=ci.gmin)
        it.evaluate({n: float(s2._node_series(k)[0]) for n, k in zip(mod.nodes, ci.dev_node[ci.dev_names.index("mneg0")])})
        assert set(opn) == set(it.opvars) and len(opn) == 70
        for k, v in opn.items():
            assert v == pytest.approx(it.opvars[k], rel=1e-8, abs=1e-30), k
        assert opn["IDS"] > 1e-6                                    # the n-FET conducts at the switching point


def test_switch_branch_on_the_gpu(E):
    for vc, want in ((1.0, 1000.0 / 1001.0), (0.0, 1e-9 * 1000.0 / (1 + 1e-9 * 1000.0))):
        c = Circuit()
        c.V("v1", "in", 0, dc=1.0)
        c.V("vc", "ctl", 0, dc=vc)
        c.VA("s1", "va_switch", ["in", "out", "ctl"], {})
        c.R("rl", "out", 0, 1e3)
        rc, x, status, st = E(c).dc(dc_opts(abstol=1e-15))
        assert rc == 0 and x[0][c._n("out") - 1] == pytest.approx(want, rel=1e-9)


def test_temperature_sweep_of_a_compiled_device_as_samples(E):
    """CH_SLOT_TEMP over the samples of one batch with a compiled Verilog-A device: every sample has its own constant block
    (`va_setup_kernel` per sample: vt = $vt depends on the temperature) — against one-sample solves at each temperature."""
    temps = [-20.0, 27.0, 100.0]

    def ckt(temp=27.0):
        c = Circuit(gmin=1e-12)
        c.temp = temp
        c.V("v1", "in", 0, dc=0.9)
        c.R("r1", "in", "a", 100.0)
        c.VA("d1", "va_diode", ["a", 0], {"IS": 1e-13, "RS": 2.0, "N": 1.2})
        c.observe_node("a")
        return c
    c = ckt()
    slot = c.slot("temp")
    e = E(c)
    e.set_samples(len(temps))
    e.set_params([slot], [np.array(temps)])
    rc, x, status, st = e.dc(dc_opts(abstol=1e-13))
    assert rc == 0
    nu, _, _ = e.maps()
    va = [x[k][c._n("a") - 1] for k in range(len(temps))]
    assert va[0] < va[1] < va[2]                       # no IS(T) in this model: the drop n vt ln(i / IS) grows with the temperature
    for k, T in enumerate(temps):
        c1 = ckt(T)
        rc1, x1, _, _ = E(c1).dc(dc_opts(abstol=1e-13))
        assert rc1 == 0 and abs(x1[0][c1._n("a") - 1] - va[k]) < 1e-9, (T, x1[0][c1._n("a") - 1], va[k])


def test_compiled_device_constants_in_lds_equal_global_memory(E, monkeypatch):
    """The device-resident stepper copies the parameter and constant blocks of a workgroup's compiled instances into LDS and evaluates
    the large models through `eval<R, PART, va::lds_cptr>` (ch_persist.hpp, `PersistArgs::va_arena`); `CEDARHIP_VA_NO_LDS=1` keeps the
    blocks in global memory (the same generated function over generic pointers).  Same steps, same waveform — on an array that
    leaves waves without a block (6 blocks: two workgroups of two pairs, the second half empty) and on a batch of samples with their
    own temperatures (one constant block per sample and instance, shared parameter blocks)."""
    import os
    from cedarsim_jl_amd.workloads import CMG_TSPAN, cmg_inverter_array
    if "bsimcmg" not in load_modules()[1]:
        pytest.skip("bsimcmg was not in the model library build")
    cards = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "asap7_tt_lvt_cards.json")))["cards"]
    ts = np.linspace(0, 1e-7, 41)
    opts = lambda: tran_opts(abstol=1e-6, reltol=1e-6, saveat=ts, dc=dc_opts(abstol=1e-10, tran_mode=1), stepper="device")  # noqa: E731

    def both(make):
        out = []
        for no_lds in (False, True):
            if no_lds:
                monkeypatch.setenv("CEDARHIP_VA_NO_LDS", "1")
            else:
                monkeypatch.delenv("CEDARHIP_VA_NO_LDS", raising=False)
            eng = make()
            rc, t, v, xf, st = eng.tran(CMG_TSPAN[0], 1e-7, opts())
            assert rc == 0 and st["stepper"] == 2
            out.append((v.copy(), st["naccept"], st["nreject"], st["nnonliniter"]))
        monkeypatch.delenv("CEDARHIP_VA_NO_LDS", raising=False)
        (v0, *c0), (v1, *c1) = out
        assert c0 == c1, (c0, c1)
        assert np.abs(v0 - v1).max() < 1e-12
        return v0

    v = both(lambda: E(cmg_inverter_array(6, cards, amp=0.2, observe="q")))
    assert v[0, :, 0].max() - v[0, :, 0].min() > 0.05                   # the output moves
    assert np.abs(v[:, :, 0] - v[0:1, :, 0]).max() < 1e-12              # identical tiles

    temps = np.array([-20.0, 0.0, 27.0, 60.0, 100.0, 125.0])

    def batch():
        c = cmg_inverter_array(1, cards, amp=0.2, observe="q")
        slot = c.slot("temp")          # before the engine sees the description
        e = E(c)
        e.set_samples(len(temps))
        e.set_params([slot], [temps])
        return e
    vb = both(batch)
    assert vb.shape[2] == len(temps)
    assert len(set(np.round(vb[0, -1, :], 7))) == len(temps)            # every sample really ran at its own temperature
    c1 = cmg_inverter_array(1, cards, amp=0.2, observe="q")
    c1.temp = 100.0
    rc1, t1, v1, _, _ = E(c1).tran(CMG_TSPAN[0], 1e-7, opts())
    assert rc1 == 0 and np.abs(v1[0, :, 0] - vb[0, :, 4]).max() < 1e-5


def test_device_exp_and_ln_accuracy(ctx):
    """The device functions do not call the library's log (118 instructions on gfx950) but build ln from frexp, the hardware
    reciprocal seed and a polynomial (va_rt.hpp `ln_pos`, ch_bsim4.hpp `flog`); exp is the library's.  Through `ch_debug_math`:
    within 2 ulp of libm over the whole range, and libm's answers at the special values — what `va_env.jl:35-47` (NaNMath) asks of
    ln, and what IEEE asks of exp."""
    rng = np.random.default_rng(11)
    x = np.concatenate((rng.uniform(-745.0, 709.7, 20000), rng.uniform(-2.0, 2.0, 20000), rng.uniform(-1e-8, 1e-8, 2000),
                        np.array([0.0, -0.0, 1.0, -1.0, 709.782712893384, 709.79, 720.0, 1e6, -745.2, -746.0, -1e6, 0.5 * np.log(2.0), -0.5 * np.log(2.0)])))
    y = ctx.debug_math(0, x)
    with np.errstate(over="ignore"):
        ref = np.exp(x)
    ok = np.isfinite(ref) & (ref > 0)
    ulp = np.abs(y[ok] - ref[ok]) / np.spacing(ref[ok])
    assert ulp.max() <= 2.0, (ulp.max(), x[ok][np.argmax(ulp)])
    assert np.array_equal(y[~ok], ref[~ok])                       # overflow -> inf, deep underflow -> 0
    sp = ctx.debug_math(0, np.array([np.inf, -np.inf, np.nan]))
    assert sp[0] == np.inf and sp[1] == 0.0 and np.isnan(sp[2])
    for which in (1, 2):                                          # compiled Verilog-A's ln, the BSIM4 code's ln
        xl = np.concatenate((np.exp(rng.uniform(-700.0, 700.0, 20000)), rng.uniform(0.5, 2.0, 20000), 1.0 + rng.uniform(-1e-6, 1e-6, 2000),
                             np.array([1.0, 2.0, 0.5, np.sqrt(0.5), np.sqrt(2.0), 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308])))
        yl = ctx.debug_math(which, xl)
        rl = np.log(xl)
        err = np.abs(yl - rl) / np.maximum(np.spacing(np.abs(rl)), 1e-17)     # near ln 1 = 0 an absolute 1e-17 stands in for the ulp
        assert err.max() <= 2.0, (which, err.max(), xl[np.argmax(err)])
        sp = ctx.debug_math(which, np.array([0.0, -1.0, np.inf, np.nan]))
        assert sp[0] == -np.inf and np.isnan(sp[1]) and sp[2] == np.inf and np.isnan(sp[3])
