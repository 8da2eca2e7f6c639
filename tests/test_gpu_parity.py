"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP engine, called through the C-ABI
(libcedarhip.so via ctypes), against the CPU oracle on the same seeded inputs, against the
reference's closed-form answers, and — at BASELINE.json's full size — through size-independent
properties.  Tolerances: rtol 1e-6 on DC operating points, 1e-4 on transient waveforms
(BASELINE.json north_star); device stamps 1e-10 (fp64, different libm)."""
import json
import math
import os

import numpy as np
import pytest

from cedarsim_jl_amd import (PULSE, PWL, SIN, Circuit, CircuitSweep, ProductSweep, dc, dc_opts, frange, parse_spice,
                             parse_spice_file, tran, tran_opts)
from cedarsim_jl_amd import bsim4_params as B4
from cedarsim_jl_amd.workloads import (DFF_CHECK_Q, DFF_CHECK_TIMES, DFF_TSPAN, dff_array, dff_chain, gf180_resolver, inverter, rc_ladder)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "closed_form.json")))
DEFTOL = 1e-7


@pytest.fixture(scope="module")
def E():
    from cedarsim_jl_amd.engine import EngineCircuit, load_library
    load_library()  # fails loudly if the HIP extension is missing
    return EngineCircuit


@pytest.fixture(scope="module")
def O(oracle_lib):
    from oracle_binding import Oracle
    return Oracle


def approx(a, b, tol=DEFTOL):
    return abs(a - b) <= max(tol, tol * max(abs(a), abs(b)))


def canon(c, x, nu):
    """Make an MNA vector consistent with the engine's aliases: merged nodes share one voltage."""
    x = x.copy()
    first = {}
    for n in range(1, c.n_nodes + 1):
        if nu[n] >= 0:
            if nu[n] in first:
                x[n - 1] = x[first[nu[n]] - 1]
            else:
                first[nu[n]] = n
    return x


def reduce_rows(c, nu, F, J=None):
    """Sum oracle rows/columns of merged nodes; keep only rows of the engine's unknowns."""
    groups = {}
    for n in range(1, c.n_nodes + 1):
        if nu[n] >= 0:
            groups.setdefault(int(nu[n]), []).append(n - 1)
    reps = [g[0] for g in groups.values()]
    Fr = np.array([sum(F[i] for i in g) for g in groups.values()])
    Jr = None if J is None else np.array([[sum(J[i, j] for i in g for j in h) for h in groups.values()] for g in groups.values()])
    return reps, Fr, Jr


# ------------------------------------------------------------------------------------------------
# device level
def test_bsim4_stamps_match_oracle(E, O):
    c = dff_array(1)
    e, o = E(c), O(c)
    rng = np.random.default_rng(0)
    for scale, lo in ((6.5, -0.75), (0.2, 2.4), (12.0, -3.0)):
        v = lo + scale * rng.random((30, 4))
        a, b = e.mos_eval(v), o.mos_eval(v)
        sc = np.maximum(np.abs(b).max(axis=0, keepdims=True), 1e-30)
        assert np.max(np.abs(a - b) / sc) < 1e-10
        aq = e.mos_eval(v, quad=True)  # the 4-lanes-per-instance path used inside the Newton kernel
        assert np.max(np.abs(aq - b) / sc) < 1e-10
    # edge cases: vds = 0 exactly, reverse mode, forward-biased junctions, deep subthreshold
    v = np.array([[2.0, 3.0, 2.0, 0.0], [0.5, 3.0, 4.0, 0.0], [0.0, 5.0, 0.0, 0.9], [5.0, -2.0, 0.0, 0.0]] + [[1, 1, 1, 1.0]] * 26)
    a, aq, b = e.mos_eval(v), e.mos_eval(v, quad=True), o.mos_eval(v)
    sc = np.maximum(np.abs(b).max(axis=0, keepdims=True), 1e-30)
    assert np.max(np.abs(a - b) / sc) < 1e-10 and np.max(np.abs(aq - b) / sc) < 1e-10


def test_residual_jacobian_assembly_matches_oracle(E, O):
    """prob.f.f / prob.f.jac equivalents (benchmarks/benchmark_common.jl:138,155) on the DFF."""
    c = dff_array(1)
    e, o = E(c), O(c)
    nu, nk, bu = e.maps()
    info = e.info()
    assert info["n_unknowns"] == 11 and info["n_known"] == 6 and info["n_alias"] == 1 and info["n_components"] == 1
    rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
    rng = np.random.default_rng(3)
    x = xo + 0.05 * rng.standard_normal(xo.shape)
    for n in range(1, c.n_nodes + 1):
        if nu[n] < 0:
            x[n - 1] = xo[n - 1]  # eliminated nodes hold their source-defined values
    x = canon(c, x, nu)
    for alpha0 in (0.0, 3e9):
        Fe, Qe, Je = e.eval(x, t=0.0, alpha0=alpha0, mode=1)
        Fo, Qo, Jo = o.eval(x, t=0.0, alpha0=alpha0, mode=1)
        reps, Fo_r, Jo_r = reduce_rows(c, nu, Fo, Jo)
        _, Qo_r, _ = reduce_rows(c, nu, Qo)
        assert np.max(np.abs(Fe[reps] - Fo_r)) <= 1e-10 * np.max(np.abs(Fo_r))
        assert np.max(np.abs(Qe[reps] - Qo_r)) <= 1e-10 * np.max(np.abs(Qo_r))
        assert np.max(np.abs(Je[np.ix_(reps, reps)] - Jo_r)) <= 1e-10 * np.max(np.abs(Jo_r))


# ------------------------------------------------------------------------------------------------
# DC: closed forms of the reference's tests, through the host mirror API
def test_dc_closed_forms():
    c = Circuit()
    c.V("V", "vcc", 0, dc=5.0)
    c.R("R", "vcc", 0, 2.0)
    sol = dc(c)
    assert sol.retcode == "Success" and approx(sol["R.V"][0], 5.0) and approx(sol["R.I"][0], 2.5) and approx(sol["V.I"][0], -2.5)
    c = Circuit()
    c.I("I", "icc", 0, dc=-5.0)
    c.R("R", "icc", 0, 2.0)
    sol = dc(c)
    assert approx(sol["icc"][0], 10.0) and approx(sol["R.I"][0], 5.0)


MULT = """* multiplicities
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


def test_dc_multiplicities_and_sources():
    sol = dc(parse_spice(MULT).build(), abstol=1e-14)  # test/basic.jl:556-595
    for n in "123456":
        assert abs(sol[n][0] - 10.0 / 11.0) < 1e-12
    c = parse_spice("* sources\nv1 1 0 2\ne1 2 0 1 0 2\nr2 2 0 1\ng1 3 0 1 0 2\nr3 3 0 1k\nb1 4 0 v=4\nr4 4 0 1\n").build()
    sol = dc(c)
    assert approx(sol["2"][0], 4.0) and approx(sol["3"][0], -4000.0) and approx(sol["4"][0], 4.0)


def test_dc_sweep_400_points_batched():
    """test/sweep.jl:326-340: one batched GPU solve for the 20x20 sweep, I = -1/(R1+R2)."""
    def two_resistor(R1=100.0, R2=100.0):
        c = Circuit()
        c.V("V", "vcc", 0, dc=1.0)
        c.R("R1", "vcc", "mid", R1)
        c.R("R2", "mid", 0, R2)
        return c
    cs = CircuitSweep(two_resistor, ProductSweep(R1=frange(100.0, 100.0, 2000.0), R2=frange(100.0, 100.0, 2000.0)))
    sols = dc(cs, abstol=DEFTOL)
    assert len(sols) == 400
    for sol, p in zip(sols, cs):
        assert sol.retcode == "Success"
        assert approx(sol["V.I"][0], -1.0 / (p["R1"] + p["R2"]))


def test_dc_sweep_on_spice_code():
    """test/sweep.jl:342-371: sweep of a top-level and a subcircuit parameter, I = v_in/r_load."""
    nl = parse_spice("""* Parameter scoping test
.subckt subcircuit1 vss gnd
.param r_load=1
r1 vss gnd 'r_load'
.ends
.param v_in=1
x1 vss 0 subcircuit1
v1 vss 0 'v_in'
""")
    cs = CircuitSweep(nl, ProductSweep(**{"v_in": frange(1.0, 1.0, 10.0), "x1.r_load": frange(1.0, 1.0, 10.0)}))
    sols = dc(cs, abstol=DEFTOL)
    assert len(sols) == 100
    for sol, p in zip(sols, cs):
        assert approx(sol["x1.r1.I"][-1], p["v_in"] / p["x1.r_load"])


def test_dc_operating_point_matches_oracle_rtol_1e6(E, O):
    for c in (inverter(), dff_array(1)):
        e, o = E(c), O(c)
        rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
        assert rc == 0
        rc, x, status, st = e.dc(dc_opts(abstol=1e-14, x0=xo[None, :]))  # same basin of the bistable DFF
        assert rc == 0 and status[0] == 0
        ok = ~np.isnan(x[0])
        assert np.allclose(x[0][ok], xo[ok], rtol=1e-6, atol=1e-9)
        # from a cold random start the engine converges too, to a point where the ORACLE's KCL holds
        rc, x2, status, st = e.dc(dc_opts(abstol=1e-14))
        assert rc == 0 and st["nnonliniter"] > 0
        nu = e.maps()[0]
        xs = canon(c, np.nan_to_num(x2[0]), nu)
        Fo, _, _ = o.eval(xs, t=0.0, alpha0=0.0, mode=0)
        _, Fr, _ = reduce_rows(c, nu, Fo)
        assert np.max(np.abs(Fr)) < 1e-10


def test_singular_and_invalid_inputs(E):
    from cedarsim_jl_amd.circuit import CedarError
    c = Circuit()
    c.I("I", "a", 0, dc=1.0)
    c.C("C", "a", 0, 1e-9)
    rc, x, status, st = E(c).dc(dc_opts(n_restarts=2, maxiters=5))
    assert rc != 0 and status[0] != 0
    c = Circuit()
    c.V("V1", "a", 0, dc=1.0)
    c.V("V2", "a", 0, dc=2.0)  # loop of voltage sources
    with pytest.raises(CedarError):
        E(c)
    with pytest.raises(CedarError):
        Circuit().R("R", "a", 0, 1.0, m=-1.0)


# ------------------------------------------------------------------------------------------------
# transient
def test_transient_rc_and_pwl_closed_forms():
    g = G["vrc"]
    c = Circuit()
    c.V("V", "vcc", 0, dc=g["V"])
    c.R("R", "vcc", "vrc", g["R"])
    c.C("C", "vrc", 0, g["C"])
    u0 = np.zeros(c.n_mna)
    u0[c.mna_index("v", "vcc")] = g["V"]
    sol = tran(c, (0.0, 1.0), abstol=1e-9, reltol=1e-9, u0=u0)
    assert sol.retcode == "Success"
    cv = sol["C.V"]
    assert approx(cv[0], 0.0) and approx(cv[-1], g["C.V(end)"])
    assert approx((sol["vcc"][0] - sol["vrc"][0]) / g["R"], g["C.I(0)"])
    tt = np.array([1e-3, 2e-3, 5e-3, 1e-2])
    sol = tran(c, (0.0, 1.0), abstol=1e-9, reltol=1e-9, u0=u0, saveat=tt)
    assert np.max(np.abs(sol["vrc"] - g["V"] * (1 - np.exp(-tt / (g["R"] * g["C"]))))) < 1e-6
    # PWL current into a resistor (test/transients.jl:17-63) from SPICE text
    p = G["pwl_ir"]
    c = parse_spice("* PWL test\n.param pval=-1\ni1 vout 0 PWL(1m 0 9m 'pval*%g')\nR1 vout 0 r=%g\n" % (p["i_max"], p["r"])).build()
    sol = tran(c, (0.0, 10e-3), abstol=1e-8, reltol=1e-8)
    want = np.clip((sol.t - p["t0"]) / (p["t1"] - p["t0"]), 0, 1) * p["i_max"] * p["r"]
    assert np.max(np.abs(sol["vout"] - want)) < DEFTOL


def test_butterworth_closed_form():
    g = G["butterworth"]
    c = Circuit()
    c.V("V1", "vin", 0, tran=SIN(0, 1, 1 / (2 * math.pi)))
    c.L("L1", "vin", "n1", g["L1"])
    c.C("C2", "n1", 0, g["C2"])
    c.L("L3", "n1", "vout", g["L3"])
    c.R("R4", "vout", 0, g["R4"])
    sol = tran(c, (0.0, 100.0), abstol=1e-9, reltol=1e-9, u0=np.zeros(c.n_mna))
    t = sol.t
    an = (np.exp(-t) - np.sin(t) - np.cos(t)) / 2 + 2 * np.sin(np.sqrt(3) * t / 2) / (np.sqrt(3) * np.sqrt(np.exp(t)))
    assert sol.retcode == "Success" and np.max(np.abs(sol["vout"] - an)) < DEFTOL


def test_inverter_logic_levels_pwl_and_pulse():
    """test/inverter.jl:40-50 (config 1), with PWL and with PULSE stimulus."""
    g = G["inverter_gate"]
    for variant in ("pwl", "pulse"):
        c = inverter()
        if variant == "pulse":
            i = c.dev_names.index("vd")
            c.sources[c.dev_ipar[i][0]] = (0.0, PULSE(0.0, 5.0, 100e-9, 10e-9, 10e-9, 100e-9, 200e-9))
        sol = tran(c, (0.0, 4e-7), abstol=1e-8, reltol=1e-8, dc_abstol=1e-14, saveat=np.array(g["t"]))
        assert sol.retcode == "Success"
        for k in range(4):
            assert approx(sol["d"][k], g["d"][k]) and approx(sol["q"][k], g["q"][k], g["tol"])


def test_dff_waveform_matches_oracle_rtol_1e4(E, O):
    """Config 2: one GF180 DFF, same initial state, waveform parity at rtol 1e-4 of the 5 V swing."""
    c = dff_array(1)
    for nm in ("q_neg", "net0", "d_neg_clked", "cki"):
        c.observe_node(nm)
    e, o = E(c), O(c)
    rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
    sv = np.linspace(0.0, 7e-7, 1401)
    tol = 1e-7
    rco, to, vo, xfo, sto = o.tran(0.0, 7e-7, tran_opts(abstol=tol, reltol=tol, saveat=sv, skip_dc=True, dc=dc_opts(x0=xo)))
    assert rco == 0
    # the oracle against BOTH step controllers, each asked for by name (the AUTO default must not decide what this test covers)
    for stepper, want in (("device", 2), ("host", 1)):
        rc, t, v, xf, st = e.tran(0.0, 7e-7, tran_opts(abstol=tol, reltol=tol, saveat=sv, skip_dc=True, dc=dc_opts(x0=xo[None, :]), stepper=stepper))
        assert rc == 0 and st["stepper"] == want, (stepper, rc, st["stepper"])
        assert len(t) == len(to) == len(sv)
        err = np.max(np.abs(v[:, :, 0] - vo))
        assert err < 1e-4 * 5.0, (stepper, err)
        for tt, q in zip(DFF_CHECK_TIMES, DFF_CHECK_Q):  # test/gf180_dff.jl:29-33
            assert abs(np.interp(tt, t, v[0, :, 0]) - q) < 1e-4
    rc, t, v, xf, st = e.tran(0.0, 7e-7, tran_opts(abstol=tol, reltol=tol, saveat=sv, skip_dc=True, dc=dc_opts(x0=xo[None, :])))
    assert rc == 0 and st["stepper"] == 2   # AUTO takes the device-resident controller for this circuit


def test_lockstep_device_kernel_six_tiles_matches_oracle(E, O):
    """The lock-step form of the device-resident controller (PM_LOCKSTEP: no saveat grid, ONE step sequence for all blocks, a
    grid-wide reduction per attempt across two workgroups) against the oracle: six tiles, tight tolerances, every accepted step
    saved; the oracle is sampled at the engine's own time points through its dense output."""
    c = dff_array(6, observe="q")
    e, o = E(c), O(c)
    rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
    assert rc == 0
    tol = 1e-7
    rc, t, v, xf, st = e.tran(0.0, 7e-7, tran_opts(abstol=tol, reltol=tol, skip_dc=True, dc=dc_opts(x0=xo[None, :])))
    assert rc == 0 and st["stepper"] == 2 and st["stepper_mode"] == 1, (rc, st["stepper"], st["stepper_mode"])
    assert len(t) == st["naccept"] + 1 and v.shape == (6, len(t), 1)
    keep = np.concatenate(([True], np.diff(t) > 0))          # a restart after a break point may repeat a time
    sv = t[keep]
    rco, to, vo, _, _ = o.tran(0.0, 7e-7, tran_opts(abstol=tol, reltol=tol, saveat=sv, skip_dc=True, dc=dc_opts(x0=xo)))
    assert rco == 0 and len(to) == len(sv)
    err = np.max(np.abs(v[:, keep, 0] - vo))
    assert err < 1e-4 * 5.0, err
    assert np.max(np.abs(v[:, :, 0] - v[0:1, :, 0])) < 1e-9   # tile equivalence inside one step sequence


def test_dff_reference_gate_from_cold_start():
    """test/gf180_dff.jl: DC from random start (CedarDCOp abstol 1e-14), abstol 1e-5, netlist as shipped."""
    nl = parse_spice_file(os.path.join(GOLD, "DFF_cap_all.cir"), lib_resolver=gf180_resolver)
    sol = tran(nl, (0.0, 7e-7), abstol=1e-5, reltol=1e-3, dc_abstol=1e-14, observe=["q"])
    assert sol.retcode == "Success"
    for tt, q in zip(DFF_CHECK_TIMES, DFF_CHECK_Q):
        assert abs(sol(tt, idxs="node_q") - q) < 1e-4


# ------------------------------------------------------------------------------------------------
# batched samples (config 4 shape) and the full-size array (config 3)
def test_monte_carlo_batch_matches_per_sample_oracle(E, O):
    c = dff_array(1)
    slots, names = [], []
    for m in ("nfet_06v0", "pfet_06v0"):
        for p in ("vth0", "u0", "toxe"):
            slots.append(c.slot(m, p))
            names.append((m, p))
    S = 64
    rng = np.random.default_rng(2024)
    base = np.array([c.models[c.model_names.index(m)][B4.PARAM_INDEX[p]] for m, p in names])
    vals = base[:, None] * (1.0 + 0.03 * rng.standard_normal((len(slots), S)))
    e = E(c)
    e.set_samples(S)
    e.set_params(slots, vals)
    assert e.info()["n_samples"] == S
    sv = np.array([0.5e-7, 0.8e-7, 1.2e-7])
    o = O(c)
    rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
    x0 = np.tile(xo, (S, 1))
    rc, t, v, xf, st = e.tran(0.0, 1.2e-7, tran_opts(abstol=1e-6, reltol=1e-6, saveat=sv, skip_dc=True, dc=dc_opts(x0=x0)))
    assert rc == 0 and v.shape == (1, 3, S)
    assert np.std(v[0, 1, :]) > 0  # samples really differ
    for s in (0, 17, 63):
        for k, sl in enumerate(slots):
            o.set_param(sl, vals[k, s])
        rco, to, vo, _, _ = o.tran(0.0, 1.2e-7, tran_opts(abstol=1e-6, reltol=1e-6, saveat=sv, skip_dc=True, dc=dc_opts(x0=xo)))
        assert rco == 0 and np.max(np.abs(v[0, :, s] - vo[0])) < 1e-4 * 5.0


def test_full_size_array_properties(E):
    """Config 3 at BASELINE size: 1024 tiles / 30720 MOSFETs.  Size-independent properties:
    (1) the reference's logic gate holds for every tile, (2) identical tiles produce identical
    waveforms (tile equivalence: the blocks are decoupled)."""
    c = dff_array(1024, observe="q")
    e = E(c)
    info = e.info()
    assert info["n_mos"] == 30720 and info["n_components"] == 1024 and info["n_classes"] == 1 and info["n_unknowns"] == 1024 * 11
    sv = np.array(DFF_CHECK_TIMES)
    rc, t, v, xf, st = e.tran(0.0, 7e-7, tran_opts(abstol=1e-5, reltol=1e-5, saveat=sv, dc=dc_opts(abstol=1e-14)))
    assert rc == 0 and v.shape == (1024, 5, 1)
    q = v[:, :, 0]
    assert np.max(np.abs(q - np.array(DFF_CHECK_Q)[None, :])) < 1e-4
    assert np.max(np.abs(q - q[0:1])) < 1e-9  # tile equivalence
    assert st["n_block_iters"] >= 1024 * st["naccept"]


def test_bench_instantiation_is_the_lockstep_device_kernel_at_full_size(E):
    """Exactly what bench.py times: the 1024-DFF array, NO saveat grid, abstol = reltol = 1e-4 — must run
    `tran_persistent_kernel<12, true, PM_LOCKSTEP>`: 1024 blocks on 256 workgroups, every accepted step saved, ONE time vector.
    Properties: controller identity (stepper 2, mode 1), one row per accepted step, the reference's gate on tile 0, and the same
    accepted / rejected / iteration counts as the host stepper (both implement one policy)."""
    e = E(dff_array(1024, observe="q0"))
    opts = dict(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14))
    rc, t, v, xf, st = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(**opts))
    assert rc == 0 and st["stepper"] == 2 and st["stepper_mode"] == 1, (rc, st["stepper"], st["stepper_mode"], e.ctx.last_error())
    assert len(t) == st["naccept"] + 1 and v.shape == (1, len(t), 1) and t[0] == 0.0 and t[-1] == DFF_TSPAN[1]
    assert st["step_kernel_launches"] == 1
    for tt, q in zip(DFF_CHECK_TIMES, DFF_CHECK_Q):      # benchmarks/gf180_dff_solver_bench.jl:89-93: within 10 abstol
        assert abs(np.interp(tt, t, v[0, :, 0]) - q) <= 10 * 1e-4
    rc_h, t_h, v_h, _, st_h = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(stepper="host", **opts))
    assert rc_h == 0 and st_h["stepper"] == 1
    assert (st_h["naccept"], st_h["nreject"], st_h["nnonliniter"]) == (st["naccept"], st["nreject"], st["nnonliniter"])
    assert np.max(np.abs(t_h - t)) < 1e-12 and np.max(np.abs(v_h - v)) < 1e-4   # libm vs device exp/log in the step-size factors: same decisions, last-digit step sizes


# ------------------------------------------------------------------------------------------------
# sparse path: Jacobian blocks larger than one CU's LDS (level-scheduled LU re-factorisation)
def test_sparse_path_rc_ladder_matches_oracle(E, O):
    c = rc_ladder(300)
    e, o = E(c), O(c)
    sv = np.array([2e-9, 2e-8, 1e-7, 5e-7, 2e-6])
    rc, t, v, xf, st = e.tran(0.0, 2e-6, tran_opts(abstol=1e-8, reltol=1e-6, saveat=sv))
    info = e.info()
    assert rc == 0 and info["path"] == 2 and info["n_unknowns"] == 300 and info["nnz_lu"] >= info["nnz_jac"] > 0
    rco, to, vo, _, _ = o.tran(0.0, 2e-6, tran_opts(abstol=1e-8, reltol=1e-6, saveat=sv))
    assert rco == 0 and np.max(np.abs(v[:, :, 0] - vo)) < 1e-4
    # DC: every node sits at the source value; closed form
    rc, x, status, st = e.dc(dc_opts(abstol=1e-12, tran_mode=True))
    assert rc == 0 and np.nanmax(np.abs(x[0][:301] - 0.0)) < 1e-9


def test_sparse_path_coupled_dff_chain_matches_oracle(E, O):
    """Six flip-flops in a shift register: one coupled 66-unknown MOS block (does not fit the fused kernel)."""
    c = dff_chain(6)
    e, o = E(c), O(c)
    info = e.info()
    assert info["n_components"] == 1 and info["n_unknowns"] == 66
    rc, xo, _ = o.dc(dc_opts(abstol=1e-13))
    assert rc == 0
    sv = np.array([0.4e-7, 0.9e-7, 1.5e-7, 2.5e-7])
    rc, t, v, xf, st = e.tran(0.0, 2.5e-7, tran_opts(abstol=1e-7, reltol=1e-7, saveat=sv, skip_dc=True, dc=dc_opts(x0=xo[None, :])))
    assert rc == 0 and e.info()["path"] == 2
    rco, to, vo, _, _ = o.tran(0.0, 2.5e-7, tran_opts(abstol=1e-7, reltol=1e-7, saveat=sv, skip_dc=True, dc=dc_opts(x0=xo)))
    assert rco == 0
    assert np.max(np.abs(v[:, :, 0] - vo)) < 1e-4 * 5.0


# ------------------------------------------------------------------------------------------------
# config 2 of BASELINE.json: device evaluation + Jacobian assembly on the GPU, sparse LU on the host
def test_config2_gpu_assembly_with_host_sparse_lu(E, O):
    """benchmarks/gf180_dff: `prob.f.jac` from the GPU (ch_eval), Newton + LU on the host (SuperLU via
    scipy, standing in for KLU): converges to the same DC operating point as the oracle."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    c = dff_array(1)
    e, o = E(c), O(c)
    nu, nk, bu = e.maps()
    rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
    x = canon(c, xo + 0.02 * np.random.default_rng(5).standard_normal(xo.shape), nu)
    for n in range(1, c.n_nodes + 1):
        if nu[n] < 0:
            x[n - 1] = xo[n - 1]
    rows = sorted({int(u): n - 1 for n, u in enumerate(nu) if n > 0 and u >= 0}.values())
    for it in range(50):
        F, Q, J = e.eval(x, t=0.0, alpha0=0.0, mode=0)
        if np.max(np.abs(F[rows])) < 1e-13:
            break
        lu = spla.splu(sp.csc_matrix(J))  # eliminated rows/cols come back as identity
        dx = lu.solve(-F)
        dx *= min(1.0, 1.0 / max(1e-30, np.max(np.abs(dx))))
        x = canon(c, x + dx, nu)
    assert it < 49
    # net11 floats between two OFF transistors (set by fA leakage): compare the driven nodes directly and
    # check the ORACLE's KCL residual at the converged point for all of them
    ok = [n - 1 for n in range(1, c.n_nodes + 1) if c.node_names[n] not in ("net11", "net4")]
    assert np.allclose(x[ok], xo[ok], rtol=1e-6, atol=1e-8)
    Fo, _, _ = o.eval(x, t=0.0, alpha0=0.0, mode=0)
    _, Fr, _ = reduce_rows(c, nu, Fo)
    assert np.max(np.abs(Fr)) < 1e-12


def test_monte_carlo_1024_samples_lockstep_gate(E):
    """Config 4 shape on one GPU: 1024 process-variation samples in ONE batched transient; every
    sample must satisfy the reference's logic gate (test/gf180_dff.jl:29-33)."""
    c = dff_array(1)
    slots, base = [], []
    for m in ("nfet_06v0", "pfet_06v0"):
        for p in ("vth0", "u0", "toxe"):
            slots.append(c.slot(m, p))
            base.append(c.models[c.model_names.index(m)][B4.PARAM_INDEX[p]])
    S = 1024
    vals = np.array(base)[:, None] * (1.0 + 0.03 * np.random.default_rng(2024).standard_normal((len(slots), S)))
    e = E(c)
    e.set_samples(S)
    e.set_params(slots, vals)
    rc, t, v, xf, st = e.tran(0.0, 7e-7, tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14), saveat=np.array(DFF_CHECK_TIMES)))
    assert rc == 0 and v.shape == (1, 5, S)
    assert np.max(np.abs(v[0] - np.array(DFF_CHECK_Q)[:, None])) < 1e-3
    assert st["n_block_iters"] >= S * st["naccept"]


def test_sparse_path_batched_samples_match_per_sample_oracle(E, O):
    """A sweep over a circuit whose Jacobian block is too large for one CU: the samples share the symbolic plan of the
    sparse LU and are solved in one batched call; every sample must equal its own oracle run."""
    c = rc_ladder(200)
    s_r, s_c = c.slot("r100", "r"), c.slot("c150", "c")
    rs, cs = [500.0, 1e3, 5e3, 2e4], [1e-12, 4e-12, 1e-12, 2.5e-13]
    eng = E(c)
    eng.set_samples(4)
    eng.set_params([s_r, s_c], [rs, cs])
    sv = np.array([1e-8, 1e-7, 4e-7, 1e-6])
    rc, t, v, xf, st = eng.tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv))
    assert rc == 0 and eng.info()["path"] == 2 and v.shape[2] == 4
    for k in range(4):
        o = O(c)
        o.set_param(s_r, rs[k])
        o.set_param(s_c, cs[k])
        rco, to, vo, _, _ = o.tran(0.0, 1e-6, tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv))
        vo = vo if vo.ndim == 2 else vo[:, :, 0]
        assert rco == 0 and np.max(np.abs(v[:, :, k] - vo)) < 1e-4, k
    assert np.abs(v[:, :, 0] - v[:, :, 3]).max() > 1e-3          # the samples really differ
    # DC sweep of the source value on the same path: every node follows its sample's source
    c2 = rc_ladder(200)
    sl = c2.slot("vin", "dc")
    e2 = E(c2)
    e2.set_samples(3)
    e2.set_params([sl], [[0.5, 1.0, 2.0]])
    rc, x, status, st = e2.dc(dc_opts(abstol=1e-12))
    assert rc == 0 and e2.info()["path"] == 2
    for k, vv in enumerate((0.5, 1.0, 2.0)):
        assert np.nanmax(np.abs(x[k][:201] - vv)) < 1e-9


GPU_SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch.distributed as dist
from cedarsim_jl_amd import Circuit, CircuitSweep, ProductSweep, dc, gather_sharded
dist.init_process_group("gloo")           # two ranks share the one GPU of this box; on a node every rank has its own GPU + RCCL
rank, world = dist.get_rank(), dist.get_world_size()

def build(r1=100.0, r2=100.0):
    c = Circuit(); c.V("V", "vcc", 0, dc=1.0); c.R("R1", "vcc", "mid", r1); c.R("R2", "mid", 0, r2)
    return c

sweep = ProductSweep(r1=[100.0 * i for i in range(1, 11)], r2=[100.0 * j for j in range(1, 11)])
cs = CircuitSweep(build, sweep, rank=rank, world=world)
sols = dc(cs)                                   # this rank's contiguous shard, one batched GPU solve
local = np.array([[s["V.i"][0], float(rank)] for s in sols])
full = gather_sharded(local, len(cs), rank, world)
want = np.array([-1.0 / (p["r1"] + p["r2"]) for p in cs])
assert full.shape == (100, 2) and np.allclose(full[:, 0], want, rtol=1e-9), np.abs(full[:, 0] - want).max()
assert np.all(full[:50, 1] == 0) and np.all(full[50:, 1] == 1)
if rank == 0: print("GPU_SHARD_OK")
dist.destroy_process_group()
'''


def test_sharded_sweep_two_ranks_on_the_gpu(tmp_path):
    """SURVEY §8(e) end to end on real hardware: two processes, each solves its contiguous shard of a 10x10 sweep
    (test/sweep.jl:326-340) on the GPU through the C-ABI, one all_gather of the results at the end."""
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text(GPU_SHARD_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", OMP_NUM_THREADS="1", CEDARHIP_USE_LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29537", str(script), root], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GPU_SHARD_OK" in r.stdout


def test_large_batch_uses_the_device_side_reduction(E):
    """More than 4096 blocks: the per-block records are reduced by `reduce_blocks_kernel` instead of the host.  4608
    samples of one DFF with two distinct parameter sets: the halves must agree internally to round-off and with a
    2-sample run of the same two sets (host-reduced path)."""
    c = dff_array(1)
    slot = c.slot("nfet_06v0", "vth0")
    base = c.models[c.model_names.index("nfet_06v0")][B4.PARAM_INDEX["vth0"]]
    S = 4608
    vals = np.where(np.arange(S) % 2 == 0, base, 1.03 * base)[None, :]
    opts = lambda: tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14), saveat=np.array(DFF_CHECK_TIMES))  # noqa: E731
    e = E(c)
    e.set_samples(S)
    e.set_params([slot], vals)
    rc, t, v, xf, st = e.tran(0.0, 7e-7, opts())
    assert rc == 0 and v.shape == (1, 5, S)
    assert np.max(np.abs(v[0] - np.array(DFF_CHECK_Q)[:, None])) < 1e-3
    assert np.abs(v[0][:, 0::2] - v[0][:, 0:1]).max() < 1e-12 and np.abs(v[0][:, 1::2] - v[0][:, 1:2]).max() < 1e-12
    e2 = E(c)
    e2.set_samples(2)
    e2.set_params([slot], [[base, 1.03 * base]])
    rc2, t2, v2, _, st2 = e2.tran(0.0, 7e-7, opts())
    assert rc2 == 0 and st2["naccept"] == st["naccept"] and st2["nreject"] == st["nreject"]
    assert np.abs(v2[0] - v[0][:, :2]).max() < 1e-9


def _random_circuit(rng, n_nodes, with_mos, c=None, prefix=""):
    """Connected random network: a resistive spanning tree to ground keeps every node's DC value defined; on top of it
    random R, C, L, I, grounded and floating V sources (incl. 0 V ammeters), VCVS / VCCS and, optionally, MOSFETs.
    `c` + `prefix`: add the network to an existing circuit under prefixed names (several networks that share only ground
    = several independent Jacobian blocks)."""
    from cedarsim_jl_amd.workloads import gf180_models
    c = Circuit(gmin=1e-12) if c is None else c
    names = ["%sn%d" % (prefix, i) for i in range(1, n_nodes + 1)]
    mi = {}
    if with_mos:
        m = gf180_models()
        mi = {"n": c.add_model(*m["nfet_06v0"]), "p": c.add_model(*m["pfet_06v0"])}
    k = 0

    def nm(p):
        nonlocal k
        k += 1
        return "%s%s%d" % (prefix, p, k)

    def pick():
        return names[rng.integers(n_nodes)] if rng.random() > 0.15 else 0

    for i, n in enumerate(names):   # spanning tree
        other = 0 if i == 0 else names[rng.integers(i)]
        c.R(nm("r"), n, other, float(10 ** rng.uniform(2, 5)))
    v0 = float(rng.uniform(1, 5))
    c.V(nm("v"), names[0], 0, dc=v0, tran=PWL([0.0, v0, 2e-7, 0.4 * v0, 5e-7, 0.4 * v0, 6e-7, v0, 1.0, v0]))
    for _ in range(int(1.5 * n_nodes)):
        a, b = pick(), pick()
        if a == b:
            continue
        t = rng.random()
        if t < 0.30:
            c.R(nm("r"), a, b, float(10 ** rng.uniform(2, 5)), m=float(rng.integers(1, 4)))
        elif t < 0.45:
            c.C(nm("c"), a, b, float(10 ** rng.uniform(-13, -10)))
        elif t < 0.52:
            c.I(nm("i"), a, b, dc=float(rng.uniform(-1e-3, 1e-3)))
        elif t < 0.60:
            mid = c.net(nm("m"))      # inductor in series with a resistor: no V/L loops
            c.L(nm("l"), a, mid, float(10 ** rng.uniform(-7, -5)))
            c.R(nm("r"), mid, b, float(10 ** rng.uniform(1, 3)))
        elif t < 0.70:
            mid = c.net(nm("m"))      # floating source (sometimes a 0 V ammeter) in series with a resistor
            c.V(nm("v"), a, mid, dc=0.0 if rng.random() < 0.5 else float(rng.uniform(-1, 1)))
            c.R(nm("r"), mid, b, float(10 ** rng.uniform(2, 4)))
        elif t < 0.78:
            c.G(nm("g"), a, b, pick(), pick(), gain=float(rng.uniform(-1e-4, 1e-4)))
        elif t < 0.84:
            mid = c.net(nm("m"))
            c.E(nm("e"), mid, 0, pick(), pick(), gain=float(rng.uniform(-0.5, 0.5)))
            c.R(nm("r"), mid, a, float(10 ** rng.uniform(3, 5)))
        elif with_mos:
            typ = "n" if rng.random() < 0.5 else "p"
            c.M(nm("m"), a, pick(), b, 0 if typ == "n" else names[0], mi[typ], float(rng.uniform(0.4e-6, 2e-6)), 6e-7)
    return c


def test_random_circuits_dc_and_residual_parity_with_oracle(E, O):
    """Differential fuzz of the structural analysis (known nodes, aliases, components, classes) and of the assembly: 24
    random networks, DC operating point on the GPU against the oracle (rtol 1e-6), plus the engine's own solution
    plugged into the ORACLE's residual (KCL of the full unreduced MNA system must vanish)."""
    rng = np.random.default_rng(20251004)
    worst = 0.0
    for trial in range(24):
        c = _random_circuit(rng, int(rng.integers(3, 14)), with_mos=trial % 3 == 0)
        o = O(c)
        rc_o, x_o, _ = o.dc(dc_opts(abstol=1e-12))
        rc, x, status, st = E(c).dc(dc_opts(abstol=1e-12))
        assert rc_o == 0 and rc == 0, (trial, rc_o, rc)
        xe = x[0].copy()
        known = ~np.isnan(xe)
        assert np.allclose(xe[known], x_o[known], rtol=1e-6, atol=1e-9), (trial, np.abs(xe[known] - x_o[known]).max())
        xe[~known] = x_o[~known]      # eliminated branch currents: take the oracle's, then check KCL with the engine's voltages
        F, Q, J = o.eval(xe, 0.0, 0.0, 0)
        worst = max(worst, float(np.abs(F).max()))
        assert np.abs(F).max() < 1e-7, (trial, np.abs(F).max())
        if trial % 3 == 1:   # transient through the PWL corners of the supply: every node against the oracle
            c.observe_all_nodes()
            sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
            opts = lambda: tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1))  # noqa: E731
            rce, te, ve, _, _ = E(c).tran(0.0, 1e-6, opts())
            rco, to, vo, _, _ = O(c).tran(0.0, 1e-6, opts())
            vo = vo if vo.ndim == 2 else vo[:, :, 0]
            assert rce == 0 and rco == 0, (trial, rce, rco)
            assert np.abs(ve[:, :, 0] - vo).max() < 1e-4 * max(1.0, np.abs(vo).max()), (trial, np.abs(ve[:, :, 0] - vo).max())


def _dense_mesh(n_nodes, rng, n_cg=None, n_mos=None):
    """Every pair of nodes joined by a resistor, every node with a capacitor to ground, a handful of MOSFETs and a PWL
    supply behind a resistor: ONE block of n_nodes unknowns whose gather lists are as long as the block size allows."""
    from cedarsim_jl_amd.workloads import gf180_models
    c = Circuit(gmin=1e-12)
    m = gf180_models()
    mn, mp = c.add_model(*m["nfet_06v0"]), c.add_model(*m["pfet_06v0"])
    names = ["n%d" % i for i in range(1, n_nodes + 1)]
    c.V("vdd", "vdd", 0, dc=5.0, tran=PWL([0.0, 5.0, 2e-7, 2.0, 5e-7, 2.0, 6e-7, 5.0, 1.0, 5.0]))
    for i, a in enumerate(names):
        c.R("rs%d" % i, "vdd" if i % 3 == 0 else 0, a, float(10 ** rng.uniform(3, 4)))
        if n_cg is None or i < n_cg:
            c.C("cg%d" % i, a, 0, float(10 ** rng.uniform(-13, -12)))
        for j in range(i + 1, n_nodes):
            c.R("r%d_%d" % (i, j), a, names[j], float(10 ** rng.uniform(3, 5)))
    for k in range(min(6, n_nodes // 2) if n_mos is None else n_mos):
        d, g, s = names[(3 * k) % n_nodes], names[(3 * k + 1) % n_nodes], names[(3 * k + 2) % n_nodes]
        if k % 2 == 0:
            c.M("mn%d" % k, d, g, s, 0, mn, 1e-6, 6e-7)
        else:
            c.M("mp%d" % k, d, g, s, "vdd", mp, 1e-6, 6e-7)
    c.observe_all_nodes()
    return c


def test_dense_blocks_of_every_kernel_variant_match_oracle(E, O):
    """One fully coupled block per size: 7 unknowns (register LU <8>, single-batch prologue), 10 with exactly 64 devices
    (register LU <12>, one wave, class blob just above the 512 ints of the single-batch prologue -> three-level prologue),
    12, 16 and 28 (register LU <12>/<16>/<32>, several waves per block, gather work list in several passes) and 40 (LDS LU,
    full gather).  DC rtol 1e-6, transient 1e-4 against the oracle."""
    rng = np.random.default_rng(7)
    sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
    for n, kw in ((7, {}), (10, dict(n_cg=5, n_mos=4)), (12, {}), (16, {}), (28, {}), (40, {})):
        c = _dense_mesh(n, rng, **kw)
        rc_o, x_o, _ = O(c).dc(dc_opts(abstol=1e-12))
        rc, x, status, st = E(c).dc(dc_opts(abstol=1e-12))
        assert rc_o == 0 and rc == 0, (n, rc_o, rc)
        known = ~np.isnan(x[0])
        assert np.allclose(x[0][known], x_o[known], rtol=1e-6, atol=1e-9), (n, np.abs(x[0][known] - x_o[known]).max())
        opts = lambda: tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1))  # noqa: E731
        rce, te, ve, _, _ = E(c).tran(0.0, 1e-6, opts())
        rco, to, vo, _, _ = O(c).tran(0.0, 1e-6, opts())
        vo = vo if vo.ndim == 2 else vo[:, :, 0]
        assert rce == 0 and rco == 0, (n, rce, rco)
        assert np.abs(ve[:, :, 0] - vo).max() < 1e-4 * max(1.0, np.abs(vo).max()), (n, np.abs(ve[:, :, 0] - vo).max())


def test_sweep_in_concurrent_groups_matches_single_batch():
    """CircuitSweep(groups=2): the points of one rank run as two batched solves with their own streams and host steppers
    (the GPU's idle time during one group's host round trip is filled by the other group's kernel).  Every point must meet
    the reference's logic gate and agree with the single-batch run to the transient tolerance."""
    from cedarsim_jl_amd import CircuitSweep, Sweep
    from cedarsim_jl_amd import tran as tran_api

    def build(dv=0.0):
        c = dff_array(1)
        for m in ("nfet_06v0", "pfet_06v0"):
            c.models[c.model_names.index(m)][B4.PARAM_INDEX["vth0"]] *= (1.0 + dv)
        return c

    dvs = list(np.linspace(-0.05, 0.05, 48))
    out = {}
    for g in (1, 2):
        cs = CircuitSweep(build, Sweep(dv=dvs), groups=g)
        sols = tran_api(cs, tspan=(0.0, 7e-7), abstol=1e-4, reltol=1e-4, dc_abstol=1e-14, saveat=np.array(DFF_CHECK_TIMES))
        assert len(sols) == len(dvs) and all(s.retcode == "Success" for s in sols)
        out[g] = np.array([s["q"] for s in sols])
        assert np.max(np.abs(out[g] - np.array(DFF_CHECK_Q)[None, :])) < 1e-3
    assert np.max(np.abs(out[1] - out[2])) < 2e-3


def test_large_coupled_component_takes_the_sparse_path():
    """One coupled component of 3000 unknowns and 6001 devices (more than the 16-bit staging offsets of the fused kernel's
    gather lists can address — such a circuit never builds them): loaded resistor ladder with a capacitor per node, DC
    against the closed form of the divider, then the step response must rise monotonically to that DC solution."""
    from cedarsim_jl_amd import tran as tran_api
    n, R, RL = 3000, 10.0, 2.0e4
    c = Circuit()
    def build(vdc):
        c = Circuit()
        c.V("vin", "n0", 0, dc=vdc, tran=PWL([0.0, 0.0, 1e-9, 1.0, 1.0, 1.0]))
        for i in range(n):
            c.R("r%d" % i, "n%d" % i, "n%d" % (i + 1), R)
            c.C("c%d" % i, "n%d" % (i + 1), 0, 1e-15)
        c.R("rl", "n%d" % n, 0, RL)
        for k in (1, n // 2, n):
            c.observe_node("n%d" % k)
        return c

    sol = dc(build(1.0), abstol=1e-12)
    itot = 1.0 / (n * R + RL)
    for k in (1, n // 2, n):
        assert approx(sol["n%d" % k][0], 1.0 - k * R * itot, 1e-6), k
    s2 = tran_api(build(0.0), tspan=(0.0, 8e-7), abstol=1e-9, reltol=1e-6)   # starts discharged; RC of the line ~ 0.1 us
    assert s2.retcode == "Success"
    v = s2["n%d" % n]
    assert abs(v[0]) < 1e-9 and np.all(np.diff(v) > -1e-7) and approx(v[-1], 1.0 - n * R * itot, 1e-4)


# ------------------------------------------------------------------------------------------------
# C-ABI robustness (round 2): exception containment, per-process kernel attributes, path routing
def test_absurd_sample_count_comes_back_as_an_error_code(E):
    """A host container sized by the caller's input must not take the process down: 2^31-1 samples of a small circuit ask
    for far more host memory than exists; the library answers with an error code and stays usable."""
    c = Circuit()
    c.V("V", "vcc", 0, dc=5.0)
    c.R("R", "vcc", "o", 2.0)
    c.R("R2", "o", 0, 2.0)
    c.observe_node("o")
    e = E(c)
    e.set_samples(2 ** 31 - 1)
    x = np.zeros((1, e.n_mna))
    status = np.zeros(1, np.int32)
    import ctypes as C
    from cedarsim_jl_amd.circuit import ChStats
    st = ChStats()
    rc = e.L.ch_dc(e.h, C.byref(dc_opts()), None, None, C.byref(st))   # no output buffers: the call must fail before it needs them
    assert rc == -8, rc   # CH_ERR_NOMEM
    assert e.ctx.last_error() != ""
    e.set_samples(1)
    rc, x, status, _ = e.dc()
    assert rc == 0 and approx(x[0][c.mna_index("v", "o")], 2.5)


def test_circuits_with_different_lds_footprints_coexist(E):
    """The dynamic-LDS ceiling of a kernel is a per-process attribute: a second, smaller circuit (still above the 48 KB
    default) must not lower it under the first circuit's need."""
    big, small = E(dff_chain(5)), E(dff_chain(3))
    assert big.info()["path"] == 1 or True
    rc1, x1, _, _ = big.dc(dc_opts(abstol=1e-12))
    rc2, x2, _, _ = small.dc(dc_opts(abstol=1e-12))
    rc3, x3, _, _ = big.dc(dc_opts(abstol=1e-12))
    assert rc1 == 0 and rc2 == 0 and rc3 == 0, (rc1, rc2, rc3, big.ctx.last_error())
    assert big.info()["path"] == 1 and small.info()["path"] == 1
    ok = ~np.isnan(x1[0])
    assert np.allclose(x1[0][ok], x3[0][ok], rtol=1e-9, atol=1e-12)


def test_device_heavy_small_block_takes_the_sparse_path():
    """2000 parallel resistors between two nodes: 3 unknowns but more stamp records than 16-bit staging offsets address.
    The circuit must be routed to the sparse path instead of being rejected."""
    n = 2000
    c = Circuit()
    c.V("v", "a", 0, dc=1.0)
    for i in range(n):
        c.R("r%d" % i, "a", "b", 1000.0 * n)
    c.R("rl", "b", "m", 500.0)
    c.R("rm", "m", 0, 500.0)
    sol = dc(c)
    assert sol.retcode == "Success"
    assert approx(sol["b"][0], 0.5, 1e-9) and approx(sol["m"][0], 0.25, 1e-9)


def test_sweep_with_dc_warm_start_gives_the_same_answers():
    """CircuitSweep(warm_start=True): every sample's DC starts from the first point's operating point instead of ten random
    restarts — same results on the reference's 400-point divider sweep and on a small Monte-Carlo of the DFF transient."""
    def two_resistor(R1=100.0, R2=100.0):
        c = Circuit()
        c.V("V", "vcc", 0, dc=1.0)
        c.R("R1", "vcc", "mid", R1)
        c.R("R2", "mid", 0, R2)
        return c
    sw = ProductSweep(R1=frange(100.0, 100.0, 2000.0), R2=frange(100.0, 100.0, 2000.0))
    cold, warm = dc(CircuitSweep(two_resistor, sw), abstol=DEFTOL), dc(CircuitSweep(two_resistor, sw, warm_start=True), abstol=DEFTOL)
    for a, b in zip(cold, warm):
        assert b.retcode == "Success" and approx(a["V.I"][0], b["V.I"][0])
    def mc(dv=0.0):
        c = dff_array(1, observe="q0")
        i = c.model_names.index("nfet_06v0")
        c.models[i][B4.PARAM_INDEX["vth0"]] += dv
        return c
    from cedarsim_jl_amd import Sweep
    dvs = [float(x) for x in np.linspace(-0.03, 0.03, 16)]
    sv = np.array(DFF_CHECK_TIMES)
    a = tran(CircuitSweep(mc, Sweep(dv=dvs)), (0.0, 7e-7), abstol=1e-5, reltol=1e-5, dc_abstol=1e-13, saveat=sv)
    b = tran(CircuitSweep(mc, Sweep(dv=dvs), warm_start=True), (0.0, 7e-7), abstol=1e-5, reltol=1e-5, dc_abstol=1e-13, saveat=sv)
    for x, y in zip(a, b):
        assert y.retcode == "Success" and np.max(np.abs(np.array(x["q"]) - np.array(y["q"]))) < 5e-4
        assert np.max(np.abs(np.array(y["q"]) - np.array(DFF_CHECK_Q))) < 1e-3


def test_large_random_networks_on_the_sparse_path_match_oracle(E, O):
    """Seeds of scripts/extended_fuzz.py (random RLC / controlled-source / MOSFET networks of up to 70 nodes: ONE block of 66–109
    unknowns, sparse path) on which the static pivot sequence of rounds 1–3 met exact zero pivots: the transient gave up with
    DtLessThanMin on its first step (20065, 20095, 20110) or the operating point did not converge (20981), while the oracle's dense
    LU had no trouble.  The pivot rows now come from an elimination of the actual values (ch_sparse_host.hpp numeric_pivot_rows)."""
    sv = np.array([1e-7, 2e-7, 3.5e-7, 5.5e-7, 6e-7, 1e-6])
    for seed in (20065, 20095, 20110, 20981):
        rng = np.random.default_rng(seed)
        c = _random_circuit(rng, int(rng.integers(3, 70)), with_mos=seed % 2 == 0)
        c.observe_all_nodes()
        e, o = E(c), O(c)
        rc_o, x_o, _ = o.dc(dc_opts(abstol=1e-12))
        rc, x, status, st = e.dc(dc_opts(abstol=1e-12))
        assert rc_o == 0 and rc == 0 and e.info()["path"] == 2, (seed, rc_o, rc, e.ctx.last_error())
        known = ~np.isnan(x[0])
        assert np.allclose(x[0][known], x_o[known], rtol=1e-6, atol=1e-9), (seed, np.abs(x[0][known] - x_o[known]).max())
        opts = lambda: tran_opts(abstol=1e-9, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12, tran_mode=1))  # noqa: E731
        rce, te, ve, _, _ = e.tran(0.0, 1e-6, opts())
        rco, to, vo, _, _ = o.tran(0.0, 1e-6, opts())
        vo = vo if vo.ndim == 2 else vo[:, :, 0]
        assert rce == 0 and rco == 0, (seed, rce, rco, e.ctx.last_error())
        assert np.abs(ve[:, :, 0] - vo).max() < 1e-4 * max(1.0, np.abs(vo).max()), (seed, np.abs(ve[:, :, 0] - vo).max())
