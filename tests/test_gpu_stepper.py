"""Device-resident step controller (cedarsim.jl_amd/csrc/ch_persist.hpp) against the host stepper of the same engine
(same policy, two implementations: one launch per attempt vs one persistent cooperative launch per transient) and against the
closed forms / the oracle.  Tolerances as everywhere: 1e-4 of the swing on transient waveforms."""
import os

import numpy as np
import pytest

from cedarsim_jl_amd import PULSE, PWL, SIN, Circuit, dc_opts, tran_opts
from cedarsim_jl_amd.workloads import DFF_CHECK_Q, DFF_CHECK_TIMES, DFF_TSPAN, dff_array

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


def run(e, tspan, stepper, **kw):
    rc, t, v, xf, st = e.tran(tspan[0], tspan[1], tran_opts(stepper=stepper, **kw))
    assert rc == 0, (rc, e.ctx.last_error())
    return t, v, xf, st


def test_single_dff_device_stepper_matches_host_stepper(E):
    e = E(dff_array(1, observe="q0"))
    sv = np.linspace(0.0, 7e-7, 701)
    kw = dict(abstol=1e-7, reltol=1e-7, saveat=sv, dc=dc_opts(abstol=1e-14))
    th, vh, xh, sth = run(e, DFF_TSPAN, "host", **kw)
    td, vd, xd, std = run(e, DFF_TSPAN, "device", **kw)
    assert sth["stepper"] == 1 and std["stepper"] == 2 and std["stepper_mode"] == 1   # one block: CH_MODE_LOCKSTEP
    assert np.array_equal(th, td) and np.array_equal(td, sv)
    assert np.max(np.abs(vh - vd)) < 1e-4 * 5.0, np.max(np.abs(vh - vd))
    ok = ~np.isnan(xh[0])
    assert np.allclose(xh[0][ok], xd[0][ok], rtol=1e-4, atol=5e-4)
    # same policy: the two controllers take (nearly) the same number of steps
    assert abs(std["naccept"] - sth["naccept"]) <= 0.02 * sth["naccept"] + 3, (std["naccept"], sth["naccept"])
    assert std["n_kernel_launches"] < 20 and sth["n_kernel_launches"] > 500
    assert std["n_step_attempts"] == std["naccept"] + std["nreject"] + std["nnonlinconvfail"]


def test_array_on_the_device_stepper_gate_and_identical_tiles(E):
    e = E(dff_array(64, observe="q"))
    t, v, xf, st = run(e, DFF_TSPAN, "device", abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14))
    assert st["stepper"] == 2 and v.shape[0] == 64
    assert np.max(np.abs(v - v[0:1])) < 1e-9   # tile equivalence (the DC restarts start every block from its own 1e-7*randn)
    q = [float(np.interp(tt, t, v[0, :, 0])) for tt in DFF_CHECK_TIMES]
    assert all(abs(a - b) <= 10 * 1e-4 for a, b in zip(q, DFF_CHECK_Q)), q
    # every accepted step saved: times strictly increasing, ending at t1
    assert np.all(np.diff(t) > 0) and t[0] == 0.0 and t[-1] == DFF_TSPAN[1] and len(t) == st["naccept"] + 1


def test_linear_circuits_with_every_waveform_kind_on_the_device_stepper(E):
    """RC low-pass driven by PWL, PULSE and SIN sources: device-resident vs host stepper on a saveat grid, and the PWL case
    against its closed form (test/transients.jl:17-62)."""
    def rc(wave):
        c = Circuit()
        c.V("v", "in", 0, dc=0.0, tran=wave)
        c.R("r", "in", "o", 1e3)
        c.C("c", "o", 0, 1e-9)
        c.observe_node("o")
        return c
    sv = np.linspace(0.0, 2e-5, 401)
    waves = [PWL([0.0, 0.0, 1e-6, 1.0, 1.2e-5, 1.0, 1.3e-5, 0.0]), PULSE(0.0, 1.0, 1e-6, 1e-7, 2e-7, 3e-6, 8e-6), SIN(0.2, 1.0, 2e5, 1e-6, 1e5, 30.0, 3.0)]
    for w in waves:
        e = E(rc(w))
        kw = dict(abstol=1e-9, reltol=1e-7, saveat=sv)
        th, vh, _, sth = run(e, (0.0, 2e-5), "host", **kw)
        td, vd, _, std = run(e, (0.0, 2e-5), "device", **kw)
        assert std["stepper"] == 2
        assert np.max(np.abs(vh - vd)) < 1e-5, (type(w).__name__, np.max(np.abs(vh - vd)))
    # closed form of the first ramp: v(t) = (t - t0) - tau (1 - exp(-(t - t0)/tau)) per volt/second of slope, tau = 1 us
    e = E(rc(waves[0]))
    td, vd, _, _ = run(e, (0.0, 2e-5), "device", abstol=1e-10, reltol=1e-8, saveat=sv)
    tau, slope = 1e-6, 1e6
    m = (sv >= 0.0) & (sv <= 1e-6)
    ref = slope * (sv[m] - tau * (1.0 - np.exp(-sv[m] / tau)))
    assert np.max(np.abs(vd[0, m, 0] - ref)) < 1e-6


def test_monte_carlo_batch_on_the_device_stepper(E):
    """n_comp == 1, many samples: per-sample error norms, maximum over samples (lock-step batch) — device vs host stepper."""
    from cedarsim_jl_amd import bsim4_params as B4
    ckt = dff_array(1, observe="q0")
    rng = np.random.default_rng(7)
    S = 96
    slots, vals = [], []
    for mname in ("nfet_06v0", "pfet_06v0"):
        bv = ckt.models[ckt.model_names.index(mname)][B4.PARAM_INDEX["vth0"]]
        slots.append(ckt.slot(mname, "vth0"))
        vals.append(bv * (1.0 + 0.02 * rng.standard_normal(S)))
    e = E(ckt)
    e.set_samples(S)
    e.set_params(slots, vals)
    sv = np.linspace(0.0, 7e-7, 141)
    kw = dict(abstol=1e-6, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-14))
    th, vh, _, sth = run(e, DFF_TSPAN, "host", **kw)
    td, vd, _, std = run(e, DFF_TSPAN, "device", **kw)
    assert std["stepper"] == 2 and vd.shape == (1, 141, S)
    assert np.max(np.abs(vh - vd)) < 1e-4 * 5.0, np.max(np.abs(vh - vd))
    assert len({tuple(np.round(vd[0, :, s], 9)) for s in range(S)}) > S // 2   # the samples really differ


def test_more_samples_than_resident_wavefronts_queue_on_the_device_stepper(E):
    """Per-sample step acceptance has no grid-wide wait, so a batch larger than the 1024 co-resident wavefronts is launched as
    an ordinary (queued) grid: 2500 RC samples with their own resistances against the closed form of each."""
    c = Circuit()
    c.V("v", "in", 0, dc=0.0, tran=PWL([0.0, 0.0, 1e-9, 1.0]))
    c.R("r", "in", "o", 1e3)
    c.C("c", "o", 0, 1e-9)
    c.observe_node("o")
    S = 2500
    slot = c.slot("r", "r")
    e = E(c)
    e.set_samples(S)
    rs = np.linspace(500.0, 3000.0, S)
    e.set_params([slot], [rs])
    sv = np.linspace(0.0, 1e-5, 101)
    td, vd, _, std = run(e, (0.0, 1e-5), "device", abstol=1e-9, reltol=1e-7, saveat=sv)
    assert std["stepper"] == 2 and vd.shape == (1, 101, S)
    m = sv > 2e-8
    for s_ in (0, 1, 1023, 1024, 1025, 2047, 2048, 2499):
        tau = rs[s_] * 1e-9
        # ramp of 1 ns, then a plateau: v = 1 - tau/1ns (exp(-(t-1ns)/tau) - exp(-t/tau))
        ref = 1.0 - tau / 1e-9 * (np.exp(-(sv[m] - 1e-9) / tau) - np.exp(-sv[m] / tau))
        assert np.max(np.abs(vd[0, m, s_] - ref)) < 2e-6, (s_, np.max(np.abs(vd[0, m, s_] - ref)))
    th, vh, _, sth = run(e, (0.0, 1e-5), "host", abstol=1e-9, reltol=1e-7, saveat=sv)
    assert np.max(np.abs(vh - vd)) < 1e-5


def test_blocks_of_one_circuit_take_their_own_steps_on_a_saveat_grid(E):
    """Independent blocks of ONE circuit on a common output grid are stepped block by block (no grid-wide reduction, every
    workgroup evaluates and stops at the sources of its own blocks only): the DFF array with per-tile clock skew (a private clock
    source per tile) against the lock-step controller at a tight tolerance, where both must land on the same waveforms."""
    rng = np.random.default_rng(1234)
    tiles = 24
    e = E(dff_array(tiles, skew=rng.uniform(0.0, 50e-12, tiles), observe="q"))
    sv = np.linspace(0.0, 7e-7, 141)
    kw = dict(abstol=1e-7, reltol=1e-7, saveat=sv, dc=dc_opts(abstol=1e-14))
    t1, v1, x1, st1 = run(e, DFF_TSPAN, "device", **kw)
    os.environ["CEDARHIP_LOCKSTEP"] = "1"
    try:
        t2, v2, x2, st2 = run(e, DFF_TSPAN, "auto", **kw)
    finally:
        del os.environ["CEDARHIP_LOCKSTEP"]
    assert st1["stepper"] == 2 and st1["stepper_mode"] == 2 and v1.shape == (tiles, 141, 1)   # CH_MODE_OWN_STEPS
    assert st2["stepper_mode"] in (0, 1)                                                         # lock-step (device) or the host stepper
    assert np.max(np.abs(v1 - v2)) < 1e-5, np.max(np.abs(v1 - v2))
    assert st1["n_step_attempts"] < 0.8 * st2["n_step_attempts"]          # no block pays for the others' clock corners
    assert st1["n_block_iters"] < 0.8 * st2["n_block_iters"]
    ok = ~np.isnan(x2[0])
    assert np.allclose(x1[0][ok], x2[0][ok], rtol=0, atol=1e-5)
    q = np.array([[np.interp(tt, t1, v1[k, :, 0]) for tt in DFF_CHECK_TIMES] for k in range(tiles)])
    assert np.max(np.abs(q - np.array(DFF_CHECK_Q)[None, :])) < 1e-3
    os.environ["CEDARHIP_PERSIST_NOPAIR"] = "1"   # the one-wave-per-block instantiation with per-workgroup source tables
    try:
        t5, v5, _, st5 = run(e, DFF_TSPAN, "device", **kw)
    finally:
        del os.environ["CEDARHIP_PERSIST_NOPAIR"]
    assert st5["stepper"] == 2 and np.max(np.abs(v5 - v1)) < 1e-5
    # full size (the bench's skewed-clock variant): 1024 private clocks, every tile through the reference's gate
    big = E(dff_array(1024, skew=np.random.default_rng(1234).uniform(0.0, 50e-12, 1024), observe="q"))
    tb, vb, _, stb = run(big, DFF_TSPAN, "auto", abstol=1e-4, reltol=1e-4, saveat=np.array(DFF_CHECK_TIMES), dc=dc_opts(abstol=1e-14))
    assert stb["stepper"] == 2 and vb.shape == (1024, 5, 1) and stb["n_step_attempts"] < 3000
    assert np.max(np.abs(vb[:, :, 0] - np.array(DFF_CHECK_Q)[None, :])) <= 10 * 1e-4
    # identical tiles stepped block by block: the same answer in every tile, and the single flip-flop's
    e3 = E(dff_array(6, observe="q"))
    t3, v3, _, st3 = run(e3, DFF_TSPAN, "device", **kw)
    e4 = E(dff_array(1, observe="q0"))
    t4, v4, _, st4 = run(e4, DFF_TSPAN, "device", **kw)
    assert np.max(np.abs(v3 - v3[0:1])) < 1e-9 and np.max(np.abs(v3[0] - v4[0])) < 1e-6


def test_own_steps_against_the_oracle(E):
    """Per-block steps against the oracle (which integrates the whole circuit with one step sequence, dense LU): six tiles with
    private, skewed clocks, waveforms of every tile within 1e-4 of the swing."""
    from oracle_binding import Oracle
    rng = np.random.default_rng(7)
    ckt = dff_array(6, skew=rng.uniform(0.0, 50e-12, 6), observe="q")
    sv = np.linspace(0.0, 7e-7, 141)
    e = E(ckt)
    # both from the SAME operating point: the flip-flops' latches are bistable at t = 0, and the two DC solvers (block-wise restarts
    # here, one system in the oracle) need not pick the same state from their random starts
    rc, x0, _, _ = e.dc(dc_opts(abstol=1e-14))
    assert rc == 0
    x0 = np.nan_to_num(x0, nan=0.0)
    rc, t, v, _, st = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(abstol=1e-6, reltol=1e-6, saveat=sv, skip_dc=True, dc=dc_opts(x0=x0)))
    assert rc == 0 and st["stepper_mode"] == 2
    rc_o, t_o, v_o, _, _ = Oracle(ckt).tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(abstol=1e-6, reltol=1e-6, saveat=sv, skip_dc=True, dc=dc_opts(x0=x0[0])))
    assert rc_o == 0
    assert np.max(np.abs(v[:, :, 0] - v_o)) < 1e-4 * 5.0, np.max(np.abs(v[:, :, 0] - v_o))


def test_own_steps_differential_fuzz(E):
    """Seeded random skewed arrays (tile count incl. counts that leave waves and pairs without a block, skew range, tolerance,
    with and without the function split across wave pairs): per-block steps against the lock-step controller."""
    rng = np.random.default_rng(4242)
    sv = np.linspace(0.0, 4.2e-7, 85)
    for trial in range(5):
        tiles = int(rng.integers(3, 31))
        skew = rng.uniform(0.0, float(rng.choice([10e-12, 50e-12, 300e-12])), tiles)
        tol = float(rng.choice([1e-6, 1e-7]))
        nopair = bool(rng.random() < 0.4)
        e = E(dff_array(tiles, skew=skew, observe="q"))
        kw = dict(abstol=tol, reltol=tol, saveat=sv, dc=dc_opts(abstol=1e-14))
        if nopair:
            os.environ["CEDARHIP_PERSIST_NOPAIR"] = "1"
        try:
            rc, t1, v1, _, st1 = e.tran(0.0, 4.2e-7, tran_opts(**kw))
            os.environ["CEDARHIP_LOCKSTEP"] = "1"
            rc2, t2, v2, _, st2 = e.tran(0.0, 4.2e-7, tran_opts(**kw))
        finally:
            os.environ.pop("CEDARHIP_LOCKSTEP", None)
            os.environ.pop("CEDARHIP_PERSIST_NOPAIR", None)
        tag = (trial, tiles, tol, nopair)
        assert rc == 0 and rc2 == 0 and st1["stepper_mode"] == 2 and st2["stepper_mode"] in (0, 1), tag
        assert np.max(np.abs(v1 - v2)) < 60 * tol * 5.0, (tag, np.max(np.abs(v1 - v2)))


def test_row_buffer_drain_and_resume(E):
    """Without saveat every accepted step is a row; when the device row buffer fills, the kernel stops with its controller state
    and history written back and the host relaunches it (resume): the result must not depend on where the cuts fall."""
    e = E(dff_array(2, observe="q"))
    kw = dict(abstol=1e-5, reltol=1e-5, dc=dc_opts(abstol=1e-14))
    t1, v1, x1, st1 = run(e, DFF_TSPAN, "device", **kw)
    os.environ["CEDARHIP_PERSIST_MAXROWS"] = "100"
    try:
        t2, v2, x2, st2 = run(e, DFF_TSPAN, "device", **kw)
    finally:
        del os.environ["CEDARHIP_PERSIST_MAXROWS"]
    assert st2["n_kernel_launches"] > st1["n_kernel_launches"] + 5
    assert len(t1) > 300 and abs(len(t1) - len(t2)) <= 0.03 * len(t1) + 3
    g = np.linspace(0.0, 7e-7, 1401)
    assert np.max(np.abs(np.interp(g, t1, v1[0, :, 0]) - np.interp(g, t2, v2[0, :, 0]))) < 1e-3 * 5.0


def test_device_stepper_is_refused_where_it_cannot_run(E):
    from cedarsim_jl_amd.workloads import dff_chain
    e = E(dff_chain(2))   # one coupled 22-unknown block: LDS LU, not the register path
    rc, t, v, xf, st = e.tran(0.0, 1e-8, tran_opts(stepper="device"))
    assert rc == -6 and "device-resident stepper" in e.ctx.last_error()
    rc, t, v, xf, st = e.tran(0.0, 1e-8, tran_opts())   # auto: falls back to the host stepper
    assert rc == 0 and st["stepper"] == 1


SHARED_GPU_WORKER = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from cedarsim_jl_amd import dc_opts, tran_opts
from cedarsim_jl_amd.engine import EngineCircuit, load_library
from cedarsim_jl_amd.workloads import DFF_CHECK_Q, DFF_CHECK_TIMES, dff_array
load_library()
e = EngineCircuit(dff_array(1024, observe="q0"))
modes = []
for k in range(3):
    rc, t, v, xf, st = e.tran(0.0, 7e-7, tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14)))
    assert rc == 0, (rc, e.ctx.last_error())
    q = [float(np.interp(tt, t, v[0, :, 0])) for tt in DFF_CHECK_TIMES]
    assert all(abs(a - b) <= 1e-3 for a, b in zip(q, DFF_CHECK_Q)), q
    modes.append(st["stepper"])
print("SHARED_GPU_OK", modes)
"""


def test_two_processes_share_the_gpu(tmp_path):
    """The device-resident stepper is a cooperative launch that takes the whole chip.  When another process's kernel holds part of
    it the workgroups are not all resident, a grid-wide wait runs into its bound, and the solve is repeated on the host stepper:
    two processes solving the 1024-DFF array at the same time must both come back with rc 0 and the reference's gate."""
    import subprocess
    import sys
    script = tmp_path / "worker.py"
    script.write_text(SHARED_GPU_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, str(script), root], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "SHARED_GPU_OK" in so, so[-1500:] + se[-1500:]


def test_a_launch_that_gives_up_is_repeated_on_the_other_path(E):
    """CEDARHIP_SPIN_TICKS=1 makes every grid-wide wait of the device-resident stepper run into its bound at once — what happens
    when another process's kernel keeps part of the GPU and the cooperative launch is not co-resident.  The solve must come back
    correct from the host stepper (independent blocks) or from the sparse path (a torn coupled array)."""
    sv = np.linspace(0.0, 3e-7, 61)
    kw = dict(abstol=1e-5, reltol=1e-5, saveat=sv, dc=dc_opts(abstol=1e-12))
    for ckt in (dff_array(8, observe="q"), dff_array(9, observe="q", supply_r=1.0)):
        e = E(ckt)
        os.environ["CEDARHIP_LOCKSTEP"] = "1"       # the lock-step form has the grid-wide waits
        try:
            rc, t, v, _, st = e.tran(0.0, 3e-7, tran_opts(**kw))
            os.environ["CEDARHIP_SPIN_TICKS"] = "1"
            rc2, t2, v2, _, st2 = e.tran(0.0, 3e-7, tran_opts(**kw))
        finally:
            os.environ.pop("CEDARHIP_SPIN_TICKS", None)
            os.environ.pop("CEDARHIP_LOCKSTEP", None)
        assert rc == 0 and st["stepper"] == 2
        assert rc2 == 0 and st2["stepper"] == 1, (rc2, st2["stepper"], e.ctx.last_error())
        assert np.max(np.abs(v - v2)) < 1e-3
    e = E(dff_array(8, observe="q"))
    os.environ["CEDARHIP_SPIN_TICKS"] = "1"
    os.environ["CEDARHIP_LOCKSTEP"] = "1"
    try:
        rc3, _, _, _, _ = e.tran(0.0, 3e-7, tran_opts(stepper="device", **kw))   # asked for explicitly: the error is reported
    finally:
        os.environ.pop("CEDARHIP_SPIN_TICKS", None)
        os.environ.pop("CEDARHIP_LOCKSTEP", None)
    assert rc3 == -5 and "wait exceeded its bound" in e.ctx.last_error()


def test_blocks_with_too_many_private_sources_keep_the_host_stepper(E):
    """Eight RC blocks, each summing twenty private pulse sources: a workgroup of four blocks would need 80 source entries (the
    device stepper holds 64 per workgroup), so the solve stays on the host stepper — and is right (superposition closed form at
    the end of the plateau)."""
    c = Circuit()
    nsrc, r, cap = 20, 1e3, 1e-12
    for b in range(8):
        for k in range(nsrc):
            c.V("v%d_%d" % (b, k), "s%d_%d" % (b, k), 0, dc=0.0, tran=PULSE(0.0, 0.1 * (k + 1), 1e-9 * (1 + b), 1e-9, 1e-9, 4e-7, 1e-6))
            c.R("r%d_%d" % (b, k), "s%d_%d" % (b, k), "n%d" % b, r)
        c.C("c%d" % b, "n%d" % b, 0, cap)
        c.observe_node("n%d" % b)
    e = E(c)
    sv = np.linspace(0.0, 2e-7, 41)
    rc, t, v, _, st = e.tran(0.0, 2e-7, tran_opts(abstol=1e-9, reltol=1e-7, saveat=sv))
    assert rc == 0 and st["stepper"] == 1, (rc, st["stepper"], e.ctx.last_error())
    want = np.mean([0.1 * (k + 1) for k in range(nsrc)])       # all sources on their plateau, the node settled (tau = r c / nsrc = 50 ps)
    assert np.max(np.abs(v[:, -1, 0] - want)) < 1e-6


def test_return_codes_of_both_step_controllers(E):
    """The SciML return codes the binding maps: MaxIters when the step budget runs out, DtLessThanMin when the step
    size underflows, on the host stepper, the lock-step device stepper and the own-steps device stepper alike; partial results
    come back with the rows reached so far."""
    e = E(dff_array(4, observe="q"))
    sv = np.linspace(0.0, 7e-7, 71)
    for stepper, extra, mode in (("host", {}, 0), ("device", {}, 1), ("device", {"saveat": sv}, 2)):
        rc, t, v, _, st = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(abstol=1e-4, reltol=1e-4, stepper=stepper, max_steps=40, dc=dc_opts(abstol=1e-14), **extra))
        assert rc == -7 and st["stepper_mode"] == mode, (stepper, extra.keys(), rc, st["stepper_mode"])       # MaxIters
        assert st["naccept"] <= 40 and (len(t) == 0 or t[-1] < DFF_TSPAN[1])
        rc, t, v, _, st = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(abstol=1e-4, reltol=1e-4, stepper=stepper, dtmin=1e-9, dc=dc_opts(abstol=1e-14), **extra))
        assert rc == -4 and st["stepper_mode"] == mode, (stepper, extra.keys(), rc)                             # DtLessThanMin: the 1 ns edges need smaller steps
    rc, t, v, _, st = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14)))
    assert rc == 0          # and the circuit is fine afterwards


def test_dense_output_after_the_fact_equals_the_saveat_grid(E):
    """`sol(t, idxs=[sys.node_q])` (test/gf180_dff.jl:29-33) on a run WITHOUT a saveat grid: `ch_result_dense_points` tells, per
    accepted step, through how many newest rows the step's BDF polynomial runs; evaluating it afterwards must give what the
    engine's own dense output writes on a saveat grid for the same step sequence — on both step controllers."""
    from cedarsim_jl_amd.api import Solution
    c = dff_array(1, observe="q")
    e = E(c)
    sv = np.linspace(0.0, 7e-7, 351)[1:-1]
    for stepper in ("device", "host"):
        kw = dict(abstol=1e-5, reltol=1e-5, dc=dc_opts(abstol=1e-14), stepper=stepper)
        rc, t, v, xf, st = e.tran(0.0, 7e-7, tran_opts(**kw))
        assert rc == 0 and st["dense_points"] is not None and len(st["dense_points"]) == len(t)
        assert st["dense_points"][0] == 0 and st["dense_points"][1:].min() >= 2 and st["dense_points"].max() <= 6
        rc2, t2, v2, _, st2 = e.tran(0.0, 7e-7, tran_opts(saveat=sv, **kw))
        assert rc2 == 0 and (st2["naccept"], st2["nreject"]) == (st["naccept"], st["nreject"])   # one block: the same step sequence
        sol = Solution(c, t, {c.obs[0]: v[0, :, 0]}, None, rc, st)
        got = sol(sv, idxs="q")
        assert np.max(np.abs(got - v2[0, :, 0])) < 1e-9, (stepper, np.max(np.abs(got - v2[0, :, 0])))
        lin = np.interp(sv, t, v[0, :, 0])
        assert np.max(np.abs(lin - v2[0, :, 0])) > 10 * np.max(np.abs(got - v2[0, :, 0]))           # and it is not just linear interpolation


def test_waves_without_a_block_never_read_unstaged_lds(E):
    """The abort of round 2 (gpurun_out/r02_stepper6.log: SIGABRT inside ch_tran on the DEVICE run that followed a HOST run of the same
    circuit; the same test alone, in a fresh process, passed): the helper wave of a wave pair that owns no block read the slot table
    of its OWN LDS region, which nothing had staged — zeros in a fresh process, but whatever the previous kernel left on that CU
    otherwise — and formed device-table addresses from it: an out-of-bounds global read, which the runtime answers with abort().
    Fixed in 9f26d32 ("helper waves read their partner's tables").  This test makes the condition deterministic: every CU's LDS is
    filled with a NaN / negative-integer pattern before each device-stepper run of circuits that leave waves without a block — a
    single DFF (one live wave + its helper), 5 tiles on own steps, and torn arrays of 9 and 10 tiles — and the results must equal the
    runs on clean LDS."""
    cases = [(dff_array(1, observe="q"), {}), (dff_array(5, observe="q"), {"saveat": np.linspace(0.0, 3e-7, 31)}),
             (dff_array(9, observe="q", supply_r=1.0), {"saveat": np.linspace(0.0, 3e-7, 31)}),
             (dff_array(10, observe="q", supply_r=1.0), {"saveat": np.linspace(0.0, 3e-7, 31)})]
    for ckt, extra in cases:
        e = E(ckt)
        kw = dict(abstol=1e-5, reltol=1e-5, dc=dc_opts(abstol=1e-12), stepper="device", **extra)
        rc0, t0, v0, _, st0 = e.tran(0.0, 3e-7, tran_opts(**kw))
        assert rc0 == 0 and st0["stepper"] == 2, (rc0, e.ctx.last_error())
        e.ctx.poison_lds()
        rc1, t1, v1, _, st1 = e.tran(0.0, 3e-7, tran_opts(**kw))
        assert rc1 == 0 and st1["stepper"] == 2, (rc1, e.ctx.last_error())
        assert (st1["naccept"], st1["nreject"], st1["nnonliniter"]) == (st0["naccept"], st0["nreject"], st0["nnonliniter"])
        assert np.array_equal(t0, t1) and np.array_equal(v0, v1)


def test_continuous_corners_keep_the_history_jumps_restart(E, O):
    """Break-point policy (IDA `tstops`, src/spectre_env.jl:71-77): every break point is landed on exactly; behind a JUMP of a source
    value the integrator restarts at order 1, at a continuous corner (a PWL knee) it keeps its history and order and only caps the
    next step.  Eight tiles with private, skewed clocks in ONE step sequence (lock-step, every step saved) have 96 private corners:
    with the restart at every corner (CEDARHIP_BP_RESTART_ALL=1, the policy of rounds 1-2) each costs about five steps.  Checked:
    fewer steps, both step controllers and the oracle agree on the waveform, and a PWL with a true jump still restarts (the capacitor
    integrates the jump exactly: test_gpu_parity_wide covers the values; here the step counts show the restart)."""
    rng = np.random.default_rng(1234)
    ckt = dff_array(8, skew=rng.uniform(0.0, 50e-12, 8), observe="q")
    e, o = E(ckt), O(ckt)
    rc, xo, _ = o.dc(dc_opts(abstol=1e-14))
    assert rc == 0
    kw = dict(abstol=1e-5, reltol=1e-5, skip_dc=True)
    res = {}
    for label, env, stepper in (("device", None, "device"), ("host", None, "host"), ("device_restart_all", "1", "device")):
        if env:
            os.environ["CEDARHIP_BP_RESTART_ALL"] = env
        try:
            rcx, t, v, _, st = e.tran(0.0, 7e-7, tran_opts(stepper=stepper, dc=dc_opts(x0=xo[None, :]), **kw))
        finally:
            os.environ.pop("CEDARHIP_BP_RESTART_ALL", None)
        assert rcx == 0, (label, rcx, e.ctx.last_error())
        res[label] = (t, v, st)
    st_d, st_h, st_r = res["device"][2], res["host"][2], res["device_restart_all"][2]
    assert st_d["stepper"] == 2 and st_d["stepper_mode"] == 1 and st_h["stepper"] == 1
    assert (st_d["naccept"], st_d["nreject"]) == (st_h["naccept"], st_h["nreject"])          # one policy, two controllers
    assert st_d["naccept"] + 2 * 96 <= st_r["naccept"], (st_d["naccept"], st_r["naccept"])   # 96 private corners: each restart cost at least two extra steps
    t, v, _ = res["device"]
    keep = np.concatenate(([True], np.diff(t) > 0))
    rco, to, vo, _, sto = o.tran(0.0, 7e-7, tran_opts(saveat=t[keep], dc=dc_opts(x0=xo), **kw))
    assert rco == 0 and np.max(np.abs(v[:, keep, 0] - vo)) < 2e-3       # same policy in the oracle; tolerance-level agreement of two solvers
    for tt, q in zip(DFF_CHECK_TIMES, DFF_CHECK_Q):
        assert np.max(np.abs([np.interp(tt, t, v[k, :, 0]) - q for k in range(8)])) < 1e-3
    # a source with a true jump: the step right behind it is the tiny restart step
    c = Circuit()
    c.V("v1", "a", 0, tran=PWL([0.0, 0.0, 1e-9, 0.0, 1e-9, 1.0, 2e-9, 1.0, 3e-9, 0.0, 5e-9, 0.0]))   # jump at 1 ns, knees at 2 and 3 ns
    c.R("r1", "a", "b", 1e3)
    c.C("c1", "b", 0, 1e-12)
    c.observe_node("b")
    ej = E(c)
    rcj, tj, vj, _, stj = ej.tran(0.0, 5e-9, tran_opts(abstol=1e-9, reltol=1e-6, stepper="device"))
    assert rcj == 0
    after = lambda tb: tj[np.searchsorted(tj, tb, side="right")] - tb          # noqa: E731  first step behind a break point
    assert after(1e-9) < 1e-13 and after(2e-9) > 1e-12 and after(3e-9) > 1e-12, (after(1e-9), after(2e-9), after(3e-9))


HBM_ROWS_WORKER = r"""
import os, sys
import numpy as np
import torch
assert torch.cuda.is_available()          # torch's HIP runtime first, as in bench.py: the engine then binds to the runtime already loaded
torch.cuda.set_device(0)
sys.path.insert(0, sys.argv[1])
from cedarsim_jl_amd import dc_opts, tran_opts, gather_sharded_device
from cedarsim_jl_amd import bsim4_params as B4
from cedarsim_jl_amd.engine import EngineCircuit, load_library
from cedarsim_jl_amd.workloads import dff_array
load_library()
c = dff_array(1, observe="q")
c.observe_node("q_neg")
slots = [c.slot("nfet_06v0", "vth0")]
base = c.models[c.model_names.index("nfet_06v0")]
e = EngineCircuit(c)
S = 16
e.set_samples(S)
e.set_params(slots, [list(base[B4.PARAM_INDEX["vth0"]] * (1.0 + 0.01 * np.arange(S)))])
sv = np.linspace(0.0, 3e-7, 61)
rc, t, v, _, st = e.tran(0.0, 3e-7, tran_opts(abstol=1e-5, reltol=1e-5, saveat=sv, dc=dc_opts(abstol=1e-14)))
assert rc == 0 and st["stepper"] == 2 and st["device_rows"] is not None and st["device_rows"].shape == v.shape
dev = torch.as_tensor(st["device_rows"], device="cuda")
assert dev.data_ptr() == st["device_rows"].ptr, "a view, not a copy"
assert np.array_equal(dev.cpu().numpy(), v)
y = (dev * 2.0).sum().item()               # torch kernels read the engine's buffer
assert abs(y - 2.0 * v.sum()) <= 1e-9 * abs(y)
c2 = dff_array(1, observe="q")
c2.observe_node("vdd")                     # a known node: evaluated on the host, so the rows are not offered
rc2, _, _, _, st2 = EngineCircuit(c2).tran(0.0, 1e-7, tran_opts(abstol=1e-4, reltol=1e-4, saveat=np.linspace(0, 1e-7, 11)))
assert rc2 == 0 and st2["device_rows"] is None
print("HBM_ROWS_OK")
"""


def test_result_rows_stay_in_hbm_for_a_device_side_gather(tmp_path):
    """`ch_result_device_values`: after a device-stepper transient the rows [n_obs][n_times][n_samples] are still in HBM; a torch CUDA
    tensor made from them without a copy (`__cuda_array_interface__`) equals the host result and torch kernels can read it — the
    single-rank half of the RCCL gather of a sharded sweep (bench.py `config4_sharded_sweep`).  Results whose observables are
    filled in on the host (a known node) are not offered.  In a process of its own, torch first: torch ships its own HIP runtime, and
    the engine must bind to the one already loaded for the pointer to mean anything to torch (the order bench.py uses)."""
    import subprocess
    import sys
    script = tmp_path / "hbm_rows.py"
    script.write_text(HBM_ROWS_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HBM_ROWS_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


def test_shared_step_control_on_a_saveat_grid_is_the_run_without_one(E):
    """`ch_tran_opts.step_control = CH_STEPS_SHARED` (a `saveat` grid alone used to change the error norm):
    independent blocks on a `saveat` grid keep ONE step sequence and ONE error norm — the single integrator of
    `solve(prob, IDA(); …)` over the whole system (src/sweeps.jl:456).  The result on the grid is then the dense output of the run
    WITHOUT a grid (same steps), on both controllers; the default takes per-block steps there (mode 2) and may differ by the
    tolerance."""
    from cedarsim_jl_amd.api import Solution
    rng = np.random.default_rng(7)
    c = dff_array(6, skew=rng.uniform(0.0, 50e-12, 6), observe="q")   # six tiles that are NOT identical: per-block steps really differ
    e = E(c)
    sv = np.linspace(0.0, 7e-7, 141)[1:-1]
    kw = dict(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14))
    rc, t, v, _, st = e.tran(0.0, 7e-7, tran_opts(stepper="device", **kw))
    assert rc == 0 and st["stepper"] == 2 and st["stepper_mode"] == 1
    for stepper in ("device", "host"):
        rc1, t1, v1, _, st1 = e.tran(0.0, 7e-7, tran_opts(saveat=sv, stepper=stepper, step_control="shared", **kw))
        assert rc1 == 0 and (st1["naccept"], st1["nreject"]) == (st["naccept"], st["nreject"]), (stepper, st1["naccept"], st["naccept"])
        if stepper == "device":
            assert st1["stepper"] == 2 and st1["stepper_mode"] == 1
        sol = Solution(c, t, {o: v[k, :, 0] for k, o in enumerate(c.obs)}, None, rc, st)
        for k in range(6):
            assert np.max(np.abs(sol(sv, idxs="x%d.q" % k) - v1[k, :, 0])) < 1e-8, (stepper, k)
    rc2, t2, v2, _, st2 = e.tran(0.0, 7e-7, tran_opts(saveat=sv, stepper="device", **kw))
    assert rc2 == 0 and st2["stepper_mode"] == 2                       # the default on a grid: per-block steps
    assert np.max(np.abs(v2 - v1)) < 0.05 * 5.0                        # the same waveforms within what the tolerance allows at the edges


def test_malformed_transient_options_are_refused_not_integrated(E):
    """`ch_tran` answers CH_ERR_INVALID (-1) with a message — before any solver sees them — for a span that is not finite and
    increasing, tolerances that are negative / non-finite / both zero, a `saveat` grid that is not finite and non-decreasing, and
    option codes outside their enums; the circuit stays usable (the torn form and the sparse path share the same check)."""
    e = E(dff_array(2, observe="q"))
    ok = dict(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-14))
    bad_spans = [(0.0, 0.0), (1e-7, 0.0), (0.0, float("nan")), (0.0, float("inf")), (float("-inf"), 1e-7)]
    for t0, t1 in bad_spans:
        rc = e.tran(t0, t1, tran_opts(**ok))[0]
        assert rc == -1 and "tspan" in e.ctx.last_error(), (t0, t1, rc, e.ctx.last_error())
    bad_opts = [dict(abstol=-1e-6), dict(reltol=float("nan")), dict(abstol=0.0, reltol=0.0), dict(abstol=float("inf")), dict(dtmax=-1.0), dict(dt0=float("nan")),
                dict(saveat=np.array([2e-7, 1e-7])), dict(saveat=np.array([1e-7, np.nan])), dict(stepper=7), dict(step_control=5)]
    for extra in bad_opts:
        kw = dict(ok); kw.update(extra)
        rc = e.tran(0.0, 3e-7, tran_opts(**kw))[0]
        assert rc == -1 and e.ctx.last_error(), (extra, rc)
    ec = E(dff_array(8, observe="q", supply_r=1.0))                              # a coupled array: torn form / sparse path
    assert ec.tran(0.0, 3e-7, tran_opts(saveat=np.array([2e-7, 1e-7]), **ok))[0] == -1
    rc, t, v, _, st = e.tran(0.0, 3e-7, tran_opts(saveat=np.array([-1e-8, 0.0, 1e-7, 1e-7, 9e-7]), **ok))   # before t0, repeated, beyond t1: a grid all the same
    assert rc == 0 and len(t) >= 4 and t[-1] <= 3e-7
