"""Bordered block-diagonal form of a coupled array (cedarsim.jl_amd/csrc/ch_analysis.hpp `tear`, ch_persist.hpp): a tiled array
whose tiles share supply rails behind a series resistance is ONE Jacobian block for the structural analysis; torn at the rails it
is independent tiles + a border of one or two unknowns, solved per Newton iteration by a register LU per tile and a Schur
complement on the border.  Checked against the sparse path of the same engine (same controller, same equations) and the oracle."""
import os

import numpy as np
import pytest

from cedarsim_jl_amd import dc_opts, tran_opts
from cedarsim_jl_amd.workloads import DFF_CHECK_Q, DFF_CHECK_TIMES, DFF_TSPAN, dff_array, dff_chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from cedarsim_jl_amd.engine import EngineCircuit, load_library
    load_library()
    return EngineCircuit


@pytest.fixture(scope="module")
def O():
    from oracle_binding import Oracle
    return Oracle


def both_paths(e, opts):
    os.environ.pop("CEDARHIP_NO_TEAR", None)
    rc, t, v, xf, st = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts)
    assert rc == 0, (rc, e.ctx.last_error())
    os.environ["CEDARHIP_NO_TEAR"] = "1"
    try:
        rc2, t2, v2, xf2, st2 = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts)
    finally:
        del os.environ["CEDARHIP_NO_TEAR"]
    assert rc2 == 0, (rc2, e.ctx.last_error())
    return (t, v, xf, st), (t2, v2, xf2, st2)


def coupled(tiles, supply_r, extra=None):
    c = dff_array(tiles, observe="q", supply_r=supply_r)
    for n in ("vdd", "vss"):
        c.observe_node(n)
    if extra:
        extra(c)
    return c


@pytest.mark.parametrize("supply_r", [1.0, (2.0, None), (None, 0.5)])
def test_torn_array_takes_the_same_steps_as_the_sparse_path(E, supply_r):
    """Two rails, only VDD, only VSS behind a resistance (border of 2, 1, 1 unknowns): same accepted / rejected / iteration counts
    and the same waveforms to rounding — the two paths solve the same linear systems in a different order."""
    e = E(coupled(12, supply_r))
    sv = np.linspace(0.0, 7e-7, 141)
    (t, v, xf, st), (t2, v2, xf2, st2) = both_paths(e, tran_opts(abstol=1e-5, reltol=1e-5, saveat=sv, dc=dc_opts(abstol=1e-12)))
    assert e.info()["path"] == 2 and e.info()["n_components"] == 1   # one coupled block for the structural analysis
    assert st["stepper"] == 2 and st2["stepper"] == 1          # device-resident stepper on the torn form / host stepper on the sparse path
    assert (st["naccept"], st["nreject"], st["nnonlinconvfail"]) == (st2["naccept"], st2["nreject"], st2["nnonlinconvfail"])
    assert abs(st["nnonliniter"] - st2["nnonliniter"]) <= 2
    assert np.max(np.abs(v - v2)) < 1e-9, np.max(np.abs(v - v2))
    ok = ~np.isnan(xf2[0])
    assert np.allclose(xf[0][ok], xf2[0][ok], rtol=0, atol=1e-9)
    q = [float(np.interp(tt, t, v[0, :, 0])) for tt in DFF_CHECK_TIMES]
    assert all(abs(a - b) <= 10 * 1e-4 for a, b in zip(q, DFF_CHECK_Q)), q
    assert np.min(v[-2]) < 5.0 - 1e-5 or supply_r == (None, 0.5)   # the VDD rail really droops when it has a resistance


def test_border_capacitors_and_the_oracle(E, O):
    """Decoupling capacitors on the border alone (rail to ground, rail to rail) are stamped into the reduced system by every
    wavefront, their charge history taken from the replicas' ring: against the sparse path and against the oracle."""
    def extra(c):
        c.C("cdec1", "vdd", 0, 2e-12)
        c.C("cdec2", "vdd", "vss", 1e-12)
        c.C("cdec3", 0, "vss", 3e-12)
    ckt = coupled(10, 5.0, extra)
    e = E(ckt)
    sv = np.linspace(0.0, 7e-7, 141)
    opts = tran_opts(abstol=1e-6, reltol=1e-6, saveat=sv, dc=dc_opts(abstol=1e-12))
    (t, v, xf, st), (t2, v2, xf2, st2) = both_paths(e, opts)
    assert st["stepper"] == 2 and st2["stepper"] == 1
    assert np.max(np.abs(v - v2)) < 1e-8, np.max(np.abs(v - v2))
    ora = O(ckt)
    rc_o, t_o, v_o, _, _ = ora.tran(DFF_TSPAN[0], DFF_TSPAN[1], opts)
    assert rc_o == 0
    assert np.max(np.abs(v[:, :, 0] - v_o)) < 1e-4 * 5.0, np.max(np.abs(v[:, :, 0] - v_o))
    assert np.ptp(v[-2]) > 1e-4   # the rail moves


def test_every_accepted_step_is_saved_and_resume_works(E):
    e = E(coupled(9, 1.0))
    kw = dict(abstol=1e-4, reltol=1e-4, dc=dc_opts(abstol=1e-12))
    rc, t1, v1, _, st1 = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(**kw))
    assert rc == 0 and st1["stepper"] == 2 and len(t1) == st1["naccept"] + 1 and t1[-1] == DFF_TSPAN[1]
    os.environ["CEDARHIP_PERSIST_MAXROWS"] = "150"
    try:
        rc, t2, v2, _, st2 = e.tran(DFF_TSPAN[0], DFF_TSPAN[1], tran_opts(**kw))
    finally:
        del os.environ["CEDARHIP_PERSIST_MAXROWS"]
    assert rc == 0 and st2["n_kernel_launches"] > st1["n_kernel_launches"]
    g = np.linspace(0.0, 7e-7, 1401)
    assert np.max(np.abs(np.interp(g, t1, v1[0, :, 0]) - np.interp(g, t2, v2[0, :, 0]))) < 1e-3 * 5.0


def test_circuits_without_a_border_keep_the_sparse_path(E):
    e = E(dff_chain(8))   # stages coupled through their data nets: no one or two nodes that everything hangs on
    rc, t, v, xf, st = e.tran(0.0, 5e-8, tran_opts(abstol=1e-4, reltol=1e-4))
    assert rc == 0 and st["stepper"] == 1 and e.info()["path"] == 2
