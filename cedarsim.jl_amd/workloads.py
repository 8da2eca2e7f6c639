"""Benchmark / test workloads of BASELINE.json as flat `Circuit`s.

The GF180 D-flip-flop topology below restates, as a Python table, the standard cell the reference
simulates (test/DFF/gf180mcu_fd_sc_mcu7t5v0__dffnq_4.ngspice:4-58: 30 MOSFETs, 15 n + 15 p, 8
distinct geometries) and its test bench (test/DFF/DFF_cap_all.cir:5-36: VDD = 5 V, CQ = 172.05 fF
behind a 0 V ammeter source, well ties, CLKN and D piece-wise-linear stimuli with 1.02 ns edges,
`.option gmin=1e-15`).  Device cards: substitute BSIM4 cards in data/gf180_substitute.lib (the real
GF180MCUPDK cards are not available).
"""
import os

import numpy as np

from .circuit import PWL, Circuit
from .netlist import parse_spice

_HERE = os.path.dirname(os.path.abspath(__file__))
SUBSTITUTE_LIB = os.path.join(_HERE, "data", "gf180_substitute.lib")


def gf180_resolver(path):
    """lib_resolver for `.LIB "jlpkg://GF180MCUPDK/..."` (test/DFF/DFF_cap_all.cir:37)."""
    if path.startswith("jlpkg://GF180MCUPDK"):
        with open(SUBSTITUTE_LIB) as f:
            return f.read()
    return None


_models_cache = {}


def gf180_models():
    """(name, type, params) of the two substitute cards."""
    if not _models_cache:
        nl = parse_spice("* cards\n.lib 'jlpkg://GF180MCUPDK/sm141064.ngspice' typical\n", lib_resolver=gf180_resolver)
        for base, bins in nl.models.items():
            _models_cache[base] = bins[0]
    return _models_cache


# (name, drain, gate, source, bulk, model, W, L) — ports of the cell: D CLKN Q VDD VNW VPW VSS
WN, WP = 3.6e-07, 4.95e-07
DFF_FETS = [
    ("tn10", "VSS", "D", "D_neg", "VPW", "n", WN), ("tp10", "VDD", "D", "D_neg", "VNW", "p", WP),
    ("tn11", "D_neg", "cki", "D_neg_clked", "VPW", "n", WN), ("tp11", "D_neg_clked", "ncki", "D_neg", "VNW", "p", WP),
    ("tn15", "Q_internal", "D_neg_clked", "VSS", "VPW", "n", WN), ("tp15", "Q_internal", "D_neg_clked", "VDD", "VNW", "p", WP),
    ("tn0", "D_neg_clked", "ncki", "net11", "VPW", "n", WN), ("tp0", "net4", "cki", "D_neg_clked", "VNW", "p", WP),
    ("tn1", "VSS", "Q_internal", "net11", "VPW", "n", WN), ("tp1", "VDD", "Q_internal", "net4", "VNW", "p", WP),
    ("tn2", "net0", "ncki", "Q_internal", "VPW", "n", WN), ("tp7", "net0", "cki", "Q_internal", "VNW", "p", WP),
    ("tn3", "net7", "cki", "net0", "VPW", "n", WN), ("tp6", "net7", "ncki", "net0", "VNW", "p", WP),
    ("tn5", "Q_neg", "net0", "VSS", "VPW", "n", 9.45e-07), ("tp3", "Q_neg", "net0", "VDD", "VNW", "p", 1.075e-06),
    ("tn4", "VSS", "Q_neg", "net7", "VPW", "n", 9.45e-07), ("tp2", "VDD", "Q_neg", "net7", "VNW", "p", 1.075e-06),
    ("tn6_7", "Q", "Q_neg", "VSS", "VPW", "n", 8.2e-07), ("tn6", "Q", "Q_neg", "VSS", "VPW", "n", 8.2e-07),
    ("tn6_7_61", "Q", "Q_neg", "VSS", "VPW", "n", 8.2e-07), ("tn6_49", "Q", "Q_neg", "VSS", "VPW", "n", 8.2e-07),
    ("tp4_13", "Q", "Q_neg", "VDD", "VNW", "p", 10.95e-07), ("tp4", "Q", "Q_neg", "VDD", "VNW", "p", 10.95e-07),
    ("tp4_13_64", "Q", "Q_neg", "VDD", "VNW", "p", 10.95e-07), ("tp4_55", "Q", "Q_neg", "VDD", "VNW", "p", 10.95e-07),
    ("tn9", "ncki", "CLKN", "VSS", "VPW", "n", 4.65e-07), ("tp9", "ncki", "CLKN", "VDD", "VNW", "p", 8.65e-07),
    ("tn16", "cki", "ncki", "VSS", "VPW", "n", 4.65e-07), ("tp16", "cki", "ncki", "VDD", "VNW", "p", 8.65e-07),
]
DFF_SHARED = ("VDD", "VSS", "VNW", "VPW", "CLKN", "D")
LN, LP = 6e-07, 5e-07

CLKN_PWL = [0.0, 5.0, 50e-9, 5.0, 51.02e-9, 0.0, 100e-9, 0.0, 101.02e-9, 5.0, 400e-9, 5.0, 401.02e-9, 0.0,
            500e-9, 0.0, 501.02e-9, 5.0, 600e-9, 5.0, 601.02e-9, 0.0, 700e-9, 0.0]
D_PWL = [0.0, 0.0, 200e-9, 0.0, 201.02e-9, 5.0, 300e-9, 5.0, 301.02e-9, 0.0, 400e-9, 0.0, 401.02e-9, 5.0, 600e-9, 5.0]
DFF_TSPAN = (0.0, 7e-7)            # test/gf180_dff.jl:24
DFF_CHECK_TIMES = (1.5e-7, 2.5e-7, 4.5e-7, 5.5e-7, 7.0e-7)
DFF_CHECK_Q = (0.0, 0.0, 5.0, 5.0, 5.0)  # test/gf180_dff.jl:29-33


def dff_array(n_tiles=1, skew=None, observe="q", gmin=1e-15, supply_r=None):
    """SURVEY §8(d) config 2 (n_tiles=1) and config 3 (n_tiles=1024): tiled DFFs sharing the supplies,
    wells, CLKN and D; each tile has its 12 private nodes, its own CQ and its own VQ ammeter.

    skew: optional per-tile clock delay in seconds (array of n_tiles) — gives every tile a private
    clock source (config 3 "optional per-tile clock skew U(0, 50 ps), seed 1234").
    supply_r: optional series resistance (ohms) between the ideal 5 V / 0 V sources and the VDD / VSS rails: the rails become
    unknowns shared by every tile, so structural analysis can no longer split the array — ONE coupled Jacobian block of
    11*n_tiles + 2 unknowns (the literal "assembly + sparse LU" form of config 3; takes the sparse path).
    """
    c = Circuit(gmin=gmin)
    m = gf180_models()
    mi = {"n": c.add_model(*m["nfet_06v0"]), "p": c.add_model(*m["pfet_06v0"])}
    r_vdd, r_vss = supply_r if isinstance(supply_r, (tuple, list)) else (supply_r, supply_r)   # a pair: one rail may stay ideal
    if r_vdd is None:
        c.V("vvdd", "vdd", 0, dc=5.0)
    else:
        c.V("vvdd", "vdd_src", 0, dc=5.0)
        c.R("rvdd", "vdd_src", "vdd", float(r_vdd))
    if r_vss is None:
        c.V("vvss", "vss", 0, dc=0.0)
    else:
        c.V("vvss", "vss_src", 0, dc=0.0)
        c.R("rvss", "vss_src", "vss", float(r_vss))
    c.V("vnw", "vnw", "vdd", dc=0.0)
    c.V("vpw", "vpw", "vss", dc=0.0)
    if skew is None:
        c.V("vclkn", "clkn", 0, tran=PWL(CLKN_PWL))
    c.V("vd", "d", 0, tran=PWL(D_PWL))
    for t in range(n_tiles):
        pre = "" if n_tiles == 1 else "x%d." % t

        def nn(name):
            ln = name.lower()
            if name in DFF_SHARED:
                if name == "CLKN" and skew is not None:
                    return pre + "clkn"
                return ln
            return pre + ln

        if skew is not None:
            w = list(CLKN_PWL)
            for i in range(2, len(w), 2):
                w[i] += float(skew[t])
            c.V(pre + "vclkn", nn("CLKN"), 0, tran=PWL(w))
        for name, d, g, s, b, typ, w in DFF_FETS:
            c.M(pre + "x_" + name, nn(d), nn(g), nn(s), nn(b), mi[typ], w, LN if typ == "n" else LP)
        c.C(pre + "cq", pre + "q_tmp", 0, 1.7205e-13)
        c.V(pre + "vq", nn("Q"), pre + "q_tmp", dc=0.0)
        if observe == "q" or (observe == "q0" and t == 0):
            c.observe_node(nn("Q"))
    return c


def dff_chain(n_stages, gmin=1e-15):
    """Shift register: Q of stage i drives D of stage i+1 (one COUPLED Jacobian block of 11*n unknowns).
    Exercises the sparse path: the block does not fit one CU's LDS for n_stages >= 6."""
    c = Circuit(gmin=gmin)
    m = gf180_models()
    mi = {"n": c.add_model(*m["nfet_06v0"]), "p": c.add_model(*m["pfet_06v0"])}
    c.V("vvdd", "vdd", 0, dc=5.0)
    c.V("vvss", "vss", 0, dc=0.0)
    c.V("vnw", "vnw", "vdd", dc=0.0)
    c.V("vpw", "vpw", "vss", dc=0.0)
    c.V("vclkn", "clkn", 0, tran=PWL(CLKN_PWL))
    c.V("vd", "d", 0, tran=PWL(D_PWL))
    for t in range(n_stages):
        pre = "x%d." % t

        def nn(name):
            if name in ("VDD", "VSS", "VNW", "VPW", "CLKN"):
                return name.lower()
            if name == "D":
                return "d" if t == 0 else "x%d.q" % (t - 1)
            return pre + name.lower()

        for name, d, g, s, b, typ, w in DFF_FETS:
            c.M(pre + "x_" + name, nn(d), nn(g), nn(s), nn(b), mi[typ], w, LN if typ == "n" else LP)
        c.C(pre + "cq", pre + "q_tmp", 0, 1.7205e-13 / 8)
        c.V(pre + "vq", nn("Q"), pre + "q_tmp", dc=0.0)
        c.observe_node(nn("Q"))
    return c


def rc_ladder(n_sections, r=1e3, cap=1e-12, vstep=1.0, trise=1e-9):
    """Uniform RC ladder driven by a ramped step: one coupled block of n_sections unknowns (sparse path)."""
    c = Circuit()
    c.V("vin", "n0", 0, tran=PWL([0.0, 0.0, trise, vstep, 1.0, vstep]))
    for i in range(n_sections):
        c.R("r%d" % i, "n%d" % i, "n%d" % (i + 1), r)
        c.C("c%d" % i, "n%d" % (i + 1), 0, cap)
    for i in (1, n_sections // 2, n_sections):
        c.observe_node("n%d" % i)
    return c


INVERTER_NETLIST = """* Inverter test (test/inverter.jl:58-81)
Xneg VSS D Q VSS nfet_06v0 W=3.6e-07 L=6e-07
Xpos VDD D Q VDD pfet_06v0 W=4.95e-07 L=5e-07
VVDD VDD 0 5.0
VVSS VSS 0 0.0
CQ D 0 1e-15
VD D 0 PWL(
+ 000.0e-9 0.0
+ 100.0e-9 0.0
+ 110.0e-9 5.0
+ 200.0e-9 5.0
+ 210.0e-9 0.0
+ 300.0e-9 0.0
+ 310.0e-9 5.0
+ 400.0e-9 5.0
+ )
.LIB "jlpkg://GF180MCUPDK/sm141064.ngspice" typical
.END
"""


def inverter():
    """SURVEY §8(d) config 1."""
    c = parse_spice(INVERTER_NETLIST, lib_resolver=gf180_resolver).build()
    c.observe_node("q")
    c.observe_node("d")
    return c


def mc_samples(n_samples, seed=2024, sigma=0.03):
    """Config 4 sample table: per-sample multipliers on vth0/u0/toxe of both cards, N(1, sigma),
    generated host-side (the reference has no working sampler: src/simulate_ir.jl:18-19)."""
    rng = np.random.default_rng(seed)
    return {k: 1.0 + sigma * rng.standard_normal(n_samples) for k in ("n.vth0", "n.u0", "n.toxe", "p.vth0", "p.u0", "p.toxe")}


MC_NAMES = ("n_vth0", "n_u0", "n_toxe", "p_vth0", "p_u0", "p_toxe", "dw", "dl")


def dff_mc_builder(gmin=1e-15, observe=("q",)):
    """Config 4 as the reference would see it: a circuit BUILDER whose keyword arguments are the swept names of an explicit
    `TandemSweep` (src/sweeps.jl:278-290) — multipliers on vth0 / u0 / toxe of both cards (N(1, 0.03)) and global W / L
    deltas in metres (N(0, 5 nm)), SURVEY 8(d) config 4.  Returns (build(**point) -> Circuit of one DFF, names)."""
    cards = gf180_models()

    def build(**kw):
        c = Circuit(gmin=gmin)
        mi = {}
        for typ, mname in (("n", "nfet_06v0"), ("p", "pfet_06v0")):
            name, mtype, params = cards[mname]
            pr = dict(params)
            for par in ("vth0", "u0", "toxe"):
                pr[par] = float(pr[par]) * float(kw.get("%s_%s" % (typ, par), 1.0))
            mi[typ] = c.add_model(name, mtype, pr)
        dw, dl = float(kw.get("dw", 0.0)), float(kw.get("dl", 0.0))
        c.V("vvdd", "vdd", 0, dc=5.0)
        c.V("vvss", "vss", 0, dc=0.0)
        c.V("vnw", "vnw", "vdd", dc=0.0)
        c.V("vpw", "vpw", "vss", dc=0.0)
        c.V("vclkn", "clkn", 0, tran=PWL(CLKN_PWL))
        c.V("vd", "d", 0, tran=PWL(D_PWL))
        for name, d, g, s, b, typ, w in DFF_FETS:
            c.M("x_" + name, d.lower(), g.lower(), s.lower(), b.lower(), mi[typ], w + dw, (LN if typ == "n" else LP) + dl)
        c.C("cq", "q_tmp", 0, 1.7205e-13)
        c.V("vq", "q", "q_tmp", dc=0.0)
        for o in observe:
            c.observe_node(o)
        return c

    return build, MC_NAMES


def mc_tandem_sweep(n_samples, seed=2024, sigma=0.03, sigma_wl=5e-9):
    """The explicit TandemSweep of config 4 (seed 2024): one value of every MC_NAMES variable per sample."""
    from .sweeps import TandemSweep
    rng = np.random.default_rng(seed)
    cols = {}
    for k in MC_NAMES:
        cols[k] = list(sigma_wl * rng.standard_normal(n_samples)) if k in ("dw", "dl") else list(1.0 + sigma * rng.standard_normal(n_samples))
    return TandemSweep(**cols)


CMG_INVERTER_DECK = """* BSIM-CMG inverter array (test/bsimcmg/inverter_cmg_cedar.cir:6-14, tiled)
VVDD VDD 0 1.0
VVSS VSS 0 0.0
CQ D 0 1e-15
VD D 0 AC 1 SIN (0.5 %(amp)g 1e7)
%(insts)s
.END
"""
CMG_TSPAN = (0.0, 4e-7)   # .TRAN 1e-9 4.0e-7 (test/bsimcmg/inverter_cmg_cedar.cir:16)


def cmg_inverter_array(n_inverters, cards, amp=0.01, observe="q0"):
    """`n_inverters` BSIM-CMG inverters (nmos_lvt / pmos_lvt of the ASAP7 TT cards: `cards` is
    either Spectre-language card text or a {name: {"master", "params"}} table such as tests/golden/asap7_tt_lvt_cards.json)
    on shared VDD / VSS / D; every inverter has a private output node.
    The reference's deck drives D with SIN(0.5 0.01 1e7); `amp` scales that amplitude."""
    insts = []
    for k in range(n_inverters):
        q = "q" if n_inverters == 1 else "q%d" % k
        insts.append("mneg%d %s D VSS VSS nmos_lvt\nmpos%d %s D VDD VDD pmos_lvt" % (k, q, k, q))
    nl = parse_spice(CMG_INVERTER_DECK % {"amp": amp, "insts": "\n".join(insts)})
    if isinstance(cards, str):
        nl.add_spectre_models(cards)
    else:
        nl.add_model_cards(cards)
    c = nl.build()
    for k in range(n_inverters):
        if observe == "q" or (observe == "q0" and k == 0):
            c.observe_node("q" if n_inverters == 1 else "q%d" % k)
    return c
