"""GPU parity tests of the small-signal analyses (ch_ac / ch_noise through the C-ABI): the reference's
closed forms and ngspice table (test/ac.jl:17-148), and the CPU oracle on nonlinear circuits."""
import json
import math
import os

import numpy as np
import pytest

from cedarsim_jl_amd import Circuit, ac, acdec, dc_opts, noise
from cedarsim_jl_amd.workloads import gf180_models

from test_ac_noise_oracle import C2, L1, L3, R4, butterworth_circuit

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def E():
    from cedarsim_jl_amd.engine import EngineCircuit, load_library
    load_library()
    return EngineCircuit


@pytest.fixture(scope="module")
def O(oracle_lib):
    from oracle_binding import Oracle
    return Oracle


def test_ac_butterworth_closed_form():
    sol = ac(butterworth_circuit())
    w = 2 * math.pi * acdec(20, 0.01, 10)
    s = 1j * w
    H = 1.0 / ((s + 1) * (s * s + s + 1))
    assert np.allclose(sol.freqresp("node_vout", w), H, rtol=1e-9, atol=0)        # test/ac.jl:47
    assert np.allclose(sol.freqresp("node_vin", w), 1.0, rtol=1e-12)               # :49
    assert np.allclose(sol.freqresp("l3.v", w), s * L3 * H, rtol=1e-9, atol=0)     # :61-64
    mag, ph, _ = sol.bode("node_vout", w)
    assert np.allclose(mag, np.abs(H), rtol=1e-9) and np.allclose(ph, np.degrees(np.unwrap(np.angle(H))), atol=1e-7)


def test_noise_butterworth_analytic_and_ngspice():
    gold = json.load(open(os.path.join(HERE, "golden", "ac_butterworth_noise_ngspice.json")))
    ng = np.array([r[1] for r in gold["rows"]])
    w = 2 * math.pi * acdec(20, 0.01, 10)
    psd = noise(butterworth_circuit()).psd("node_vout", w)
    s = 1j * w
    par = lambda a, b: a * b / (a + b)  # noqa: E731
    H = par(par(s * L1, 1 / (s * C2)) + s * L3, R4)
    apsd = np.sqrt(np.abs(4 * 1.380649e-23 * (23 + 273.15) / R4 * H * H))
    assert np.allclose(np.sqrt(psd), apsd, rtol=1e-6)     # test/ac.jl:148
    assert np.allclose(np.sqrt(psd), ng, rtol=1e-6)       # against the ngspice table itself


def amp_circuit():
    """Common-source NMOS stage with a resistive load, a current-source bias and an RC-coupled AC input."""
    c = Circuit(gmin=1e-12)
    c.temp = 40.0
    m = gf180_models()
    n = c.add_model(*m["nfet_06v0"])
    p = c.add_model(*m["pfet_06v0"])
    c.V("vdd", "vdd", 0, dc=5.0)
    c.V("vin", "in", 0, dc=0.0, ac=1.0)
    c.C("cin", "in", "g", 1e-9)
    c.R("rb1", "vdd", "g", 300e3, m=2.0)
    c.R("rb2", "g", 0, 100e3)
    c.M("m1", "d", "g", "s", 0, n, 2e-6, 6e-7)
    c.R("rd", "vdd", "d", 20e3)
    c.R("rs", "s", 0, 1e3)
    c.C("cs", "s", 0, 1e-9)
    c.M("m2", "o", "d", "vdd", "vdd", p, 4e-6, 5e-7)   # PMOS follower-ish second stage
    c.R("ro", "o", 0, 50e3)
    c.C("co", "o", 0, 1e-12)
    c.I("iinj", "d", 0, dc=1e-6, ac=0.0)
    return c


def test_ac_nonlinear_amplifier_matches_oracle(E, O):
    ckt = amp_circuit()
    ckt.observe_all_nodes()
    f = acdec(5, 1e2, 1e10)
    rc_o, xo = O(ckt).ac(f, dc_opts(abstol=1e-12))
    rc, xe, st = E(ckt, small_signal=True).ac(f, dc_opts(abstol=1e-12))
    assert rc == 0 and rc_o == 0
    xe = xe[0]
    gain = np.abs(xe[:, ckt._n("o") - 1])
    assert gain.max() > 1.0                                     # it does amplify
    for node in ("g", "d", "s", "o", "in"):
        a, b = xe[:, ckt._n(node) - 1], xo[:, ckt._n(node) - 1]
        assert np.allclose(a, b, rtol=1e-6, atol=1e-12 * np.abs(b).max()), node
    # branch current of the AC source (kept as an unknown)
    k = ckt.mna_index("i", "vin")
    assert np.allclose(xe[:, k], xo[:, k], rtol=1e-6, atol=1e-18)


def test_noise_nonlinear_amplifier_matches_oracle(E, O):
    ckt = amp_circuit()
    ckt.observe_all_nodes()
    f = acdec(5, 1e2, 1e10)
    for node in ("o", "d"):
        rc_o, po = O(ckt).noise(ckt._n(node) - 1, f, dc_opts(abstol=1e-12))
        rc, pe, st = E(ckt).noise(0, ckt._n(node), f, dc_opts(abstol=1e-12))
        assert rc == 0 and rc_o == 0
        assert np.all(pe[0] > 0)
        assert np.allclose(pe[0], po, rtol=1e-6), node
    # a node held by an ideal source is noiseless
    rc, pe, st = E(ckt).noise(0, ckt._n("vdd"), f)
    assert rc == 0 and np.all(pe == 0.0)


def test_ac_current_source_and_batched_samples(E, O):
    """I-source excitation into a parallel RC; R swept over 8 samples in one batched call: Z = R/(1+jwRC)."""
    c = Circuit()
    c.I("i1", 0, "a", dc=0.0, ac=2.0)
    c.R("r1", "a", 0, 1e3)
    c.C("c1", "a", 0, 1e-9)
    c.observe_node("a")
    slot = c.slot("r1", "r")
    eng = E(c, small_signal=True)
    rs = np.linspace(500.0, 4000.0, 8)
    eng.set_samples(8)
    eng.set_params([slot], [list(rs)])
    f = acdec(10, 1e3, 1e7)
    rc, x, st = eng.ac(f)
    assert rc == 0
    w = 2 * math.pi * f
    for s, r in enumerate(rs):
        z = 2.0 * r / (1 + 1j * w * r * 1e-9)
        assert np.allclose(x[s, :, 0], z, rtol=1e-10)
    # thermal noise of the same R: 4kTR/(1+(wRC)^2)
    rc, psd, st = eng.noise(0, c._n("a"), f)
    assert rc == 0
    for s, r in enumerate(rs):
        assert np.allclose(psd[s], 4 * 1.380649e-23 * (27 + 273.15) * r / (1 + (w * r * 1e-9) ** 2), rtol=1e-10)


def test_ac_requires_a_source_and_small_blocks():
    from cedarsim_jl_amd import CedarError
    c = Circuit()
    c.V("v1", "a", 0, dc=1.0)
    c.R("r1", "a", 0, 1.0)
    with pytest.raises(CedarError):
        ac(c)


def test_bsimcmg_inverter_noise_matches_ngspice_table_on_the_gpu(E, O):
    """test/ac.jl:155-237 through ch_noise: compiled BSIM-CMG 107 + ASAP7 TT cards, output noise at node q against the
    reference's 61-point ngspice table at the reference's own tolerance (rtol 1e-6)."""
    from cedarsim_jl_amd.va.registry import load_modules
    from cedarsim_jl_amd.workloads import cmg_inverter_array
    if "bsimcmg" not in load_modules()[1]:
        pytest.skip("bsimcmg was not in the model library build")
    gold = json.load(open(os.path.join(HERE, "golden", "ac_bsimcmg_inverter_noise_ngspice.json")))
    f = np.array([r[0] for r in gold["rows"]])
    ng = np.array([r[1] for r in gold["rows"]])
    c = cmg_inverter_array(1, json.load(open(os.path.join(HERE, "golden", "asap7_tt_lvt_cards.json")))["cards"])
    rc, psd, st = E(c).noise(0, c._n("q"), f, dc_opts(abstol=1e-12))
    assert rc == 0
    assert np.allclose(np.sqrt(psd[0]), ng, rtol=1e-6)
    rc_o, po = O(c).noise(c._n("q") - 1, f, dc_opts(abstol=1e-12))
    assert rc_o == 0 and np.allclose(psd[0], po, rtol=1e-8)
    # the AC transfer of the same stage: engine == oracle
    rc, xe, st = E(c, small_signal=True).ac(f[:40], dc_opts(abstol=1e-12))
    rc_o, xo = O(c).ac(f[:40], dc_opts(abstol=1e-12))
    assert rc == 0 and rc_o == 0
    assert np.allclose(xe[0][:, c._n("q") - 1], xo[:, c._n("q") - 1], rtol=1e-7)
    assert np.abs(xe[0][0, c._n("q") - 1]) > 3.0      # an inverter biased at its switching point has gain


def test_va_noise_sources_white_and_flicker(E):
    """Compiled module with white_noise + flicker_noise: V²/Hz at the node of a current-biased noisy resistor."""
    c = Circuit()
    c.temp = 50.0
    c.I("ib", 0, "a", dc=1e-3)
    c.VA("r1", "va_noisy_resistor", ["a", 0], {"R": 2e3, "KF": 1e-10, "AF": 2.0, "EF": 1.2}, m=2.0)
    c.observe_node("a")
    f = acdec(4, 1.0, 1e6)
    rc, psd, st = E(c).noise(0, c._n("a"), f)
    assert rc == 0
    k, T, R = 1.3806503e-23, 50.0 + 273.15, 2e3          # the module uses `P_K of constants.vams
    i_each = 0.5e-3                                       # two parallel instances share the bias
    s_i = 2.0 * (4 * k * T / R + 1e-10 * i_each ** 2 / f ** 1.2)
    assert np.allclose(psd[0], s_i * (R / 2.0) ** 2, rtol=1e-10)


def _self_biased_chain(n_stages):
    """AC-coupled chain of self-biased BSIM-CMG inverters (1 MOhm feedback, 1 fF coupling): every stage sits at its switching
    point, so the DC point is easy and the small-signal transfer is large; the coupling capacitors make it ONE Jacobian block."""
    from cedarsim_jl_amd import parse_spice
    cards = json.load(open(os.path.join(HERE, "golden", "asap7_tt_lvt_cards.json")))["cards"]
    lines = ["* chain", "VVDD VDD 0 0.7", "VIN src 0 DC 0 AC 1", "cc0 src i1 1e-15"]
    for k in range(1, n_stages + 1):
        lines += ["mn%d o%d i%d 0 0 nmos_lvt" % (k, k, k), "mp%d o%d i%d VDD VDD pmos_lvt" % (k, k, k), "rf%d o%d i%d 1e6" % (k, k, k)]
        if k < n_stages:
            lines.append("cc%d o%d i%d 1e-15" % (k, k, k + 1))
    lines.append("cl o%d 0 1e-15" % n_stages)
    nl = parse_spice("\n".join(lines) + "\n.END\n")
    nl.add_model_cards(cards)
    return nl.build()


def test_ac_and_noise_on_the_sparse_path_match_oracle(E, O):
    """A coupled system too large for the fused kernel (13 self-biased BSIM-CMG stages, 81 unknowns with the AC source):
    the sparse path evaluates G and C, which are expanded to one dense block per sample for the complex LU."""
    from cedarsim_jl_amd.va.registry import load_modules
    if "bsimcmg" not in load_modules()[1]:
        pytest.skip("bsimcmg was not in the model library build")
    c = _self_biased_chain(13)
    c.observe_all_nodes()
    f = acdec(4, 1e6, 1e12)
    eng = E(c, small_signal=True)
    rc, xe, st = eng.ac(f, dc_opts(abstol=1e-11))
    assert rc == 0, eng.ctx.last_error()
    assert eng.info()["path"] == 2 and 64 < eng.info()["n_unknowns"] <= 96
    rc_o, xo = O(c).ac(f, dc_opts(abstol=1e-11))
    assert rc_o == 0
    for node in ("o1", "o6", "o13"):
        a, b = xe[0][:, c._n(node) - 1], xo[:, c._n(node) - 1]
        assert np.allclose(a, b, rtol=1e-6, atol=1e-9 * max(1.0, np.abs(b).max())), node
    assert np.abs(xe[0][:, c._n("o13") - 1]).max() > 10.0      # the chain amplifies in its pass band
    rc, pe, st = E(c).noise(0, c._n("o3"), f, dc_opts(abstol=1e-11))
    rc_o, po = O(c).noise(c._n("o3") - 1, f, dc_opts(abstol=1e-11))
    assert rc == 0 and rc_o == 0 and np.all(pe[0] > 0)
    assert np.allclose(pe[0], po, rtol=1e-6)


def _ac_ladder(n):
    c = Circuit()
    c.V("vin", "n0", 0, dc=0.0, ac=1.0)
    for i in range(n):
        c.R("r%d" % i, "n%d" % i, "n%d" % (i + 1), 1e3)
        c.C("c%d" % i, "n%d" % (i + 1), 0, 1e-12)
    c.observe_node("n%d" % n)
    return c


def test_ac_and_noise_of_a_large_coupled_system_match_oracle(E, O):
    """More than 96 coupled unknowns: the complex LU leaves LDS and runs as one 256-thread workgroup per (frequency, sample)
    on a global workspace.  A 300-section RC ladder (sparse path), two samples with different values of one resistor,
    against the oracle's host LU; the thermal noise of its 300 resistors at the far end likewise.  A system above 4096
    unknowns is declined."""
    n = 300
    c = _ac_ladder(n)
    slot = c.slot("r7")
    e = E(c, small_signal=True)
    e.set_samples(2)
    e.set_params([slot], [[1e3, 2.5e3]])
    freqs = np.logspace(2, 5.5, 17)
    rc, x, st = e.ac(freqs, dc_opts(abstol=1e-12))
    assert rc == 0, e.ctx.last_error()
    assert e.info()["path"] == 2 and e.info()["n_unknowns"] > 96
    out = c._n("n%d" % n)
    rce, pe, _ = e.noise(0, out, freqs, dc_opts(abstol=1e-12))
    assert rce == 0, e.ctx.last_error()
    for s_, r7 in enumerate((1e3, 2.5e3)):
        o = O(c)
        o.set_param(slot, r7)
        rco, xo = o.ac(freqs, dc_opts(abstol=1e-12))
        assert rco == 0
        nodes = [c._n("n%d" % k) - 1 for k in (1, 7, 8, n // 2, n)]
        assert np.abs(x[s_][:, nodes] - xo[:, nodes]).max() <= 1e-6 * np.abs(xo[:, nodes]).max()
        rcn, psd = o.noise(out - 1, freqs, dc_opts(abstol=1e-12))
        assert rcn == 0 and np.all(psd > 0)
        assert np.allclose(pe[s_], psd, rtol=1e-6, atol=0.0)
    big = E(_ac_ladder(4200), small_signal=True)
    rc, _, _ = big.ac(freqs[:2], dc_opts(abstol=1e-12))
    assert rc != 0 and "4096" in big.ctx.last_error()
