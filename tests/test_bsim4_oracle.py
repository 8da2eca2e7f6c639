"""Self-consistency of the oracle's BSIM4 restatement.  The device arithmetic is PARITY UNPINNED
against the reference (bsim4.va is not in /root/reference); these tests pin what can be pinned:
dual-number derivatives vs finite differences, KCL / charge neutrality, source-drain symmetry,
polarity mirror, temperature and gmin behaviour.  CPU only."""
import numpy as np

from cedarsim_jl_amd import Circuit
from cedarsim_jl_amd import bsim4_params as B4
from cedarsim_jl_amd.workloads import gf180_models
from oracle_binding import Oracle


def two_fets(**kw):
    c = Circuit(**kw)
    m = gf180_models()
    n, p = c.add_model(*m["nfet_06v0"]), c.add_model(*m["pfet_06v0"])
    c.M("mn", "d", "g", "s", "b", n, 3.6e-7, 6e-7)
    c.M("mp", "d", "g", "s", "b", p, 4.95e-7, 5e-7)
    return c


def test_dual_derivatives_match_finite_differences():
    o = Oracle(two_fets())
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(100):
        v = rng.uniform(-1, 6, size=(2, 4))
        v[1] = -v[1]
        out = o.mos_eval(v)
        h = 1e-6
        for k in range(2):
            for j in range(4):
                vp, vm = v.copy(), v.copy()
                vp[k, j] += h
                vm[k, j] -= h
                fd = (o.mos_eval_values(vp)[k] - o.mos_eval_values(vm)[k]) / (2 * h)
                g = np.concatenate([out[k, 8:24].reshape(4, 4)[:, j], out[k, 24:40].reshape(4, 4)[:, j]])
                scale = np.concatenate([np.full(4, 1e-9 + np.abs(out[k, 8:24]).max()), np.full(4, 1e-20 + np.abs(out[k, 24:40]).max())])
                worst = max(worst, np.max(np.abs(fd - g) / scale))
    assert worst < 1e-6


def test_kcl_and_charge_neutrality():
    o = Oracle(two_fets())
    rng = np.random.default_rng(1)
    v = rng.uniform(-1, 6, size=(2, 4))
    out = o.mos_eval(v)
    for k in range(2):
        assert abs(out[k, 0:4].sum()) < 1e-12 * np.abs(out[k, 0:4]).max() + 1e-18  # currents sum to zero
        assert abs(out[k, 4:8].sum()) < 1e-12 * np.abs(out[k, 4:8]).max()            # charges sum to zero
        G, C = out[k, 8:24].reshape(4, 4), out[k, 24:40].reshape(4, 4)
        assert np.abs(G.sum(axis=1)).max() < 1e-9 * np.abs(G).max()  # rows sum to zero: only differences matter
        assert np.abs(C.sum(axis=1)).max() < 1e-9 * np.abs(C).max()
        assert np.abs(G.sum(axis=0)).max() < 1e-9 * np.abs(G).max()


def test_source_drain_symmetry_and_zero_vds():
    o = Oracle(two_fets())
    v = np.array([[1.3, 3.0, 0.4, 0.0], [0.0, 0.0, 0.0, 0.0]])
    vs = v.copy()
    vs[0, [0, 2]] = v[0, [2, 0]]  # swap drain and source voltages
    a, b = o.mos_eval(v)[0], o.mos_eval(vs)[0]
    assert abs(a[0] - b[2]) < 1e-12 * abs(a[0]) and abs(a[2] - b[0]) < 1e-12 * abs(a[0])  # Id <-> Is
    assert abs(a[4] - b[6]) < 1e-9 * abs(a[5])  # Qd <-> Qs
    z = o.mos_eval(np.array([[2.0, 3.0, 2.0, 0.0], [0, 0, 0, 0.0]]))[0]
    assert abs(z[0] - z[2]) < 1e-18 and abs(z[0] + z[2] + z[3]) < 1e-18  # vds=0: no channel current, only the two (equal) junction leakages


def test_iv_is_monotonic_and_off_current_small():
    o = Oracle(two_fets())
    ids = [o.mos_eval(np.array([[5.0, vg, 0.0, 0.0], [0, 0, 0, 0.0]]))[0][0] for vg in np.linspace(0, 5, 26)]
    assert all(b > a for a, b in zip(ids, ids[1:]))
    assert ids[0] < 1e-9 and 5e-5 < ids[-1] < 5e-4
    idp = o.mos_eval(np.array([[0, 0, 0, 0.0], [-5.0, -5.0, 0.0, 0.0]]))[1][0]
    assert -5e-4 < idp < -2e-5


def test_pmos_is_mirror_of_nmos_with_mirrored_card():
    m = gf180_models()
    c = Circuit()
    pn = dict(m["nfet_06v0"][2])
    n = c.add_model("n", "nmos", pn)
    pp = dict(pn)
    pp["vth0"] = -pn["vth0"]
    p = c.add_model("p", "pmos", pp)
    c.M("mn", "d", "g", "s", "b", n, 1e-6, 6e-7)
    c.M("mp", "d", "g", "s", "b", p, 1e-6, 6e-7)
    o = Oracle(c)
    v = np.array([[2.0, 3.0, 0.2, -0.3]])
    out = o.mos_eval(np.vstack([v, -v]))
    assert np.allclose(out[0, :8], -out[1, :8], rtol=1e-12, atol=1e-30)
    assert np.allclose(out[0, 8:], out[1, 8:], rtol=1e-12, atol=1e-30)


def test_temperature_and_gmin_slots():
    c = two_fets(gmin=1e-12)
    st, sg = c.slot("temp"), c.slot("gmin")
    o = Oracle(c)
    v = np.array([[5.0, 5.0, 0.0, 0.0], [0, 0, 0, 0.0]])
    i27 = o.mos_eval(v)[0][0]
    o.set_param(st, 125.0)
    i125 = o.mos_eval(v)[0][0]
    assert i125 < i27  # mobility degradation wins at high Vgs
    off = np.array([[5.0, 0.0, 0.0, 0.0], [0, 0, 0, 0.0]])
    o.set_param(st, 27.0)
    a = o.mos_eval(off)[0][0]
    o.set_param(sg, 1e-9)
    b = o.mos_eval(off)[0][0]
    assert abs((b - a) - (1e-9 - 1e-12) * 5.0) < 1e-12  # gmin sits across the drain-bulk junction
