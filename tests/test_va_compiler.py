"""Verilog-A front-end, interpreter and code generator (CPU): the reference's VA tests restated
(test/ddx.jl, test/varegress.jl, test/basic.jl:359-381) on the oracle, and the generated C++ (host
instantiation) against the independent Python AST interpreter for every module of the compiled library."""
import math
import os

import numpy as np
import pytest

from cedarsim_jl_amd import Circuit, dc_opts, parse_spice, tran_opts
from cedarsim_jl_amd.va.frontend import Preprocessor, VAError, parse_number, parse_va, tokenize
from cedarsim_jl_amd.va.interp import Interp
from cedarsim_jl_amd.va.registry import find_module, load_modules
from oracle_binding import Oracle, lib


def test_preprocessor_macros_and_conditionals():
    pp = Preprocessor(defines={"FLAG": 1})
    out = pp.process("""`define A 2.0
`define MUL(x, y) ((x)*(y))
`ifdef FLAG
a = `MUL(`A, b+1);   // comment
`else
a = 0;
`endif
`ifndef FLAG
never
`endif
s = "keep `this";
""")
    assert "((2.0)*(b+1))" in out and "never" not in out and "a = 0" not in out and "keep `this" in out
    with pytest.raises(VAError):
        Preprocessor().process("x = `UNDEFINED;")


def test_literals_scale_factors_and_precedence():
    assert parse_number("1k") == ("num", 1000.0, False)           # src/vasim.jl:100-126
    assert parse_number("2.5u")[1] == pytest.approx(2.5e-6) and parse_number("10")[2] is True
    assert parse_number("1.0e-38")[1] == 1.0e-38
    m = parse_va("module t(a); electrical a; real x; analog begin x = -2**2 + 3*4 > 1 ? 1 : 0; end endmodule")[0]
    it = Interp(m)
    it.evaluate({})
    # -(2**2) + 12 = 8 > 1
    assert m.analog[0][3][0][2][0] == "tern"
    assert [t[1] for t in tokenize("a<+b**c")][:5] == ["a", "<+", "b", "**", "c"]


def test_integer_semantics_and_case():
    m = parse_va("""module t(a, b); electrical a, b;
      parameter integer sel = 2; integer k; real y;
      analog begin
        k = 2.5;            // rounds half away from zero (src/va_env.jl:107)
        y = 7/2;            // real division (src/vasim.jl:221-232)
        case (sel)
          1: y = y + 100;
          2, 3: y = y + k;
          default: y = -1;
        endcase
        I(a,b) <+ y;
      end endmodule""")[0]
    I, Q, G, C = Interp(m).evaluate({})
    assert I[0] == 3.5 + 3
    I, _, _, _ = Interp(m, {"sel": 9}).evaluate({})
    assert I[0] == -1


def test_function_output_arguments_and_param_given():
    m = parse_va("""module t(a, b); electrical a, b;
      parameter real R = 5.0; parameter real K = 1.0;
      real g, h;
      analog function real twice; input x; output y; real x, y;
        begin y = 3*x; twice = 2*x; end
      endfunction
      analog begin
        g = twice(V(a,b), h);
        I(a,b) <+ g + h + ($param_given(R) ? R : 100.0);
      end endmodule""")[0]
    I, _, G, _ = Interp(m).evaluate({"a": 2.0})
    assert I[0] == 4 + 6 + 100 and G[0][0] == 5.0
    I, _, _, _ = Interp(m, {"r": 7.0}).evaluate({"a": 2.0})   # case-insensitive instance parameter
    assert I[0] == 4 + 6 + 7


def test_parameter_ranges():
    _, mod = find_module("va_resistor")
    with pytest.raises(VAError):
        Interp(mod, {"R": -1.0}, strict_ranges=True)
    _, nl = find_module("va_nlvcr")
    with pytest.raises(VAError):
        Interp(nl, {"R": 0.0}, strict_ranges=True)
    assert Interp(nl, {"R": 0.0}).warnings   # the reference does not enforce ranges: recorded, not raised


def test_ddx_matches_reference_test_and_has_second_derivatives():
    _, mod = find_module("va_nlvcr")
    I, Q, G, C = Interp(mod, {"R": 2.0}).evaluate({"d": 5.0, "g": 3.0, "s": 0.0})
    assert I[0] == 5 * 2 * 2 * 3                     # test/ddx.jl:21 (seen from the source: -60)
    assert G[0] == [12.0, 20.0, -32.0]               # ∂/∂Vd = 2R·Vgs, ∂/∂Vg = 2R·Vds (needs the nested dual)


def _random_bias(mod, rng, scale=1.0):
    return {n: float(scale * rng.uniform(-0.3, 1.0)) for n in mod.nodes}


def _codegen_vs_interp(name, params, biases, temp_c=27.0, gmin=1e-12, tol=1e-11):
    mid, mod = find_module(name)
    assert lib().oracle_va_module_name(mid).decode() == mod.name
    it = Interp(mod, params, temperature_c=temp_c, gmin=gmin)
    P = np.array([float(it.params[p[0]]) if p[1] != "string" else 0.0 for p in mod.params] + [1.0 if p[0] in it.given else 0.0 for p in mod.params] + [0.0])
    n = len(mod.nodes)
    for vb in biases:
        I, Q, G, C = it.evaluate(vb)
        v = np.zeros(8)
        v[:n] = [vb.get(x, 0.0) for x in mod.nodes]
        st = np.zeros(144)
        rc = lib().oracle_va_eval(mid, P.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)),
                                  v.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)), temp_c + 273.15, gmin,
                                  st.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)))
        assert rc == 0
        for got, want in ((st[:n], I), (st[8:8 + n], Q), (st[16:80].reshape(8, 8)[:n, :n], G), (st[80:144].reshape(8, 8)[:n, :n], C)):
            want = np.array(want, dtype=float)
            assert np.allclose(got, want, rtol=tol, atol=tol * max(1e-300, np.abs(want).max())), (name, vb)


def test_generated_code_matches_interpreter_library_models():
    rng = np.random.default_rng(5)
    for name, params in (("va_resistor", {"R": 2e3}), ("va_resistor_rev", {}), ("va_nlvcr", {"R": 2.0}), ("va_capacitor", {"C": 3e-12}),
                         ("va_diode", {"IS": 2e-14, "RS": 5.0, "LEVEL": 2}), ("va_diode", {}), ("va_mos1", {"TYPE": -1, "W": 4e-6, "CGSO": 1e-10}),
                         ("va_mos1", {"KP": 2e-4}), ("va_inductor", {"L": 2e-6, "RS": 3.0}), ("va_vcvs", {"VDC": 1.5, "GAIN": -2.0}),
                         ("va_noisy_resistor", {"KF": 1e-12}), ("va_switch", {"VTH": 0.3}), ("va_switch", {"VTH": 5.0})):
        _, mod = find_module(name)
        _codegen_vs_interp(name, params, [_random_bias(mod, rng) for _ in range(6)], temp_c=40.0)


def test_generated_code_matches_interpreter_bsimcmg():
    """The CMC BSIM-CMG 107 model of the reference (VerilogAParser.jl/cmc_models/bsimcmg107): 905 parameters,
    ~4k lines, two internal nodes — present when the library was built where the reference checkout exists."""
    mods, ix = load_modules()
    if "bsimcmg" not in ix:
        pytest.skip("bsimcmg was not in the model library build")
    rng = np.random.default_rng(11)
    _, mod = find_module("bsimcmg")
    for params in ({"DEVTYPE": 1, "L": 2e-8, "NFIN": 2, "IGCMOD": 1, "IGBMOD": 1, "GIDLMOD": 1}, {"DEVTYPE": 0, "L": 3e-8, "TFIN": 8e-9, "CGEOMOD": 2},
                   {"DEVTYPE": 1, "BULKMOD": 1, "CAPMOD": 1, "GEOMOD": 1}):
        biases = [_random_bias(mod, rng, 0.8) for _ in range(4)]
        for b in biases:   # internal nodes near their ports
            b["di"] = b["d"] + 1e-3 * rng.standard_normal()
            b["si"] = b["s"] + 1e-3 * rng.standard_normal()
        _codegen_vs_interp("bsimcmg", params, biases, tol=1e-9)


def test_reference_va_circuit_tests_on_the_oracle():
    # test/basic.jl:370-381: `.hdl` + VA resistor 2k across a 1 V source
    c = parse_spice('* Verilog Include 2\n.hdl "cedar_basic.va"\nx1 vcc 0 va_resistor r=2k\nv1 vcc 0 dc=1\n').build()
    rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0 and x[c.mna_index("i", "v1")] == pytest.approx(-1 / 2e3, rel=1e-12)
    # test/ddx.jl
    c = Circuit()
    c.V("v1", "vcc", 0, dc=5.0)
    c.V("v2", "vg", 0, dc=3.0)
    c.VA("r", "va_nlvcr", ["vcc", "vg", 0], {"R": 2.0})
    rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0 and x[c.mna_index("i", "v1")] == pytest.approx(-5 * 2 * 2 * 3, rel=1e-12)
    # test/varegress.jl: both branch orientations charge the capacitor with a non-negative resistor current
    for mod in ("va_resistor", "va_resistor_rev"):
        c = Circuit()
        c.V("v", "vcc", 0, dc=1.0)
        c.VA("r", mod, ["vcc", "out"], {"R": 1000.0})
        c.C("c", "out", 0, 1e-9)
        c.observe_node("out")
        rc, t, v, xf, st = Oracle(c).tran(0.0, 1e-5, tran_opts(abstol=1e-9, reltol=1e-6, skip_dc=1))
        assert rc == 0
        vout = v[0] if v.ndim == 2 else v[0, :, 0]
        i_r = (1.0 - vout) / 1000.0
        assert np.all(i_r >= -1e-12)
        assert vout[-1] == pytest.approx(1 - math.exp(-10.0), rel=1e-4)


def test_netlist_model_card_for_va_module_and_unknown_module():
    nl = parse_spice("""* card
.model dmod va_diode is=3e-14 rs=2
x1 a 0 dmod n=1.2
v1 a 0 0.7
""")
    c = nl.build()
    mod, p = c.va_instances["x1"]
    assert mod.name == "va_diode" and p["IS"] == 3e-14 and p["RS"] == 2.0 and p["N"] == 1.2
    assert "x1.ai" in c.node_names
    with pytest.raises(Exception):
        parse_spice("* t\nx1 a 0 no_such_module r=1\nv1 a 0 1\n").build()


def test_generated_noise_records_match_interpreter():
    import ctypes as C
    pd = C.POINTER(C.c_double)
    cases = [("va_noisy_resistor", {"R": 2e3, "KF": 1e-12, "AF": 1.5, "EF": 0.9}, {"p": 0.7, "n": -0.1})]
    if "bsimcmg" in load_modules()[1]:
        cases.append(("bsimcmg", {"DEVTYPE": 1, "L": 2.1e-8, "NFIN": 2, "IGCMOD": 1, "IGBMOD": 1}, {"d": 0.6, "g": 0.7, "s": 0.0, "e": 0.0, "di": 0.599, "si": 0.001}))
    for name, params, vb in cases:
        mid, mod = find_module(name)
        it = Interp(mod, params, temperature_c=30.0)
        it.evaluate(vb)
        P = np.array([float(it.params[p[0]]) if p[1] != "string" else 0.0 for p in mod.params] + [1.0 if p[0] in it.given else 0.0 for p in mod.params] + [0.0])
        v = np.zeros(8)
        v[:len(mod.nodes)] = [vb.get(x, 0.0) for x in mod.nodes]
        out = np.zeros(64)
        n = lib().oracle_va_noise(mid, P.ctypes.data_as(pd), v.ctypes.data_as(pd), 303.15, 1e-12, out.ctypes.data_as(pd))
        assert n == len(it.noise) and n > 0
        for k, rec in enumerate(it.noise):
            a, b, pwr, ex = out[4 * k:4 * k + 4]
            assert mod.nodes[int(a)] == rec["nodes"][0] and (mod.nodes[int(b)] == rec["nodes"][1] if len(rec["nodes"]) > 1 else b == -1)
            assert pwr == pytest.approx(rec["pwr"], rel=1e-12, abs=1e-300) and ex == pytest.approx(rec["exp"], rel=1e-12)


def test_voltage_contributions_and_branch_current_probe():
    """`V(a,b) <+ expr` with `I(a,b)` probes (voltage form of a branch equation, src/vasim.jl:128-180, 810-822): the branch
    current is an extra unknown of the instance.  RL step response and a VA-defined source against closed forms."""
    _, ind = find_module("va_inductor")
    assert ind.nodes == ["p", "n", "I(p,n)"] and ind.vbranches == [("p", "n")]
    c = Circuit()
    c.V("v1", "in", 0, dc=1.0)
    c.R("r1", "in", "a", 100.0)
    c.VA("l1", "va_inductor", ["a", 0], {"L": 1e-3, "RS": 0.0})
    c.observe_node("a")
    c.observe_node("l1.i(p,n)")
    rc, t, v, xf, st = Oracle(c).tran(0.0, 5e-5, tran_opts(abstol=1e-10, reltol=1e-7, skip_dc=1))
    assert rc == 0
    v = v if v.ndim == 2 else v[:, :, 0]
    tau = 1e-3 / 100.0
    assert np.allclose(v[1], 1.0 / 100.0 * (1 - np.exp(-t / tau)), rtol=1e-4, atol=1e-8)
    # DC: the inductor is a short, a VA source sets a node: out = VDC + GAIN*V(c) = 1.5 - 2*0.25
    c = Circuit()
    c.V("vc", "c", 0, dc=0.25)
    c.VA("e1", "va_vcvs", ["out", 0, "c", 0], {"VDC": 1.5, "GAIN": -2.0})
    c.R("rl", "out", 0, 50.0)
    rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-14))
    assert rc == 0 and x[c._n("out") - 1] == pytest.approx(1.0, rel=1e-12)
    assert x[c._n("e1.i(p,n)") - 1] == pytest.approx(-1.0 / 50.0, rel=1e-12)   # current through the branch p -> n


def test_front_end_rejects_the_reference_parsers_error_cases():
    """VerilogAParser.jl/test/errors/*.va: malformed sources the reference's parser must diagnose.  The front-end has to
    raise VAError on every one (never crash or hang).  Runs where the reference checkout is present."""
    import glob
    from cedarsim_jl_amd.va.frontend import parse_va_file
    files = sorted(glob.glob("/root/reference/VerilogAParser.jl/test/errors/*.va"))
    if not files:
        pytest.skip("reference checkout not present")
    for f in files:
        with pytest.raises(VAError):
            parse_va_file(f)


def test_operating_point_observables_match_interpreter():
    """Variables declared with (* desc = "..." *) are the module's observables (src/vasim.jl:742-753): generated
    `opvars` pass (host instantiation) against the interpreter."""
    import ctypes as C
    pd = C.POINTER(C.c_double)
    cases = [("va_mos1", {"TYPE": 1, "KP": 2e-4, "W": 2e-6}, {"d": 1.2, "g": 1.5, "s": 0.0, "b": 0.0}),
             ("va_mos1", {"TYPE": -1}, {"d": -0.1, "g": -2.0, "s": 0.0, "b": 0.0})]
    if "bsimcmg" in load_modules()[1]:
        cases.append(("bsimcmg", {"DEVTYPE": 1, "L": 2.1e-8, "NFIN": 2}, {"d": 0.6, "g": 0.7, "s": 0.0, "e": 0.0, "di": 0.599, "si": 0.001}))
    for name, params, vb in cases:
        mid, mod = find_module(name)
        it = Interp(mod, params, temperature_c=27.0)
        it.evaluate(vb)
        P = np.array([float(it.params[p[0]]) if p[1] != "string" else 0.0 for p in mod.params] + [1.0 if p[0] in it.given else 0.0 for p in mod.params] + [0.0])
        v = np.zeros(8)
        v[:len(mod.nodes)] = [vb.get(x, 0.0) for x in mod.nodes]
        out = np.zeros(128)
        n = lib().oracle_va_opvars(mid, P.ctypes.data_as(pd), v.ctypes.data_as(pd), 300.15, 1e-12, out.ctypes.data_as(pd))
        assert n == len(mod.var_desc) and n > 0
        for k in range(n):
            nm = lib().oracle_va_opvar_name(mid, k).decode()
            assert out[k] == pytest.approx(it.opvars[nm], rel=1e-12, abs=1e-300), (name, nm)
    # semantic check on the square-law model: saturation, IDS = KP/2 * W/L * Vov^2 * (1 + lambda*Vds)
    _, mod = find_module("va_mos1")
    it = Interp(mod, {"KP": 2e-4, "W": 2e-6, "L": 1e-6, "VTO": 0.7, "LAMBDA": 0.0})
    it.evaluate({"d": 2.0, "g": 1.7})
    assert it.opvars["REGION"] == 2 and it.opvars["VOV"] == pytest.approx(1.0) and it.opvars["IDS"] == pytest.approx(0.5 * 2e-4 * 2.0 * 1.0)


def test_analog_initial_blocks_run_first():
    m = parse_va("""module t(a,b); electrical a,b; parameter real R=2.0; real g, k;
analog begin I(a,b) <+ k*g*V(a,b); end
analog initial begin g = 1.0/R; end
analog initial k = 3.0;
endmodule""")[0]
    I, Q, G, C = Interp(m).evaluate({"a": 3.0})
    assert I[0] == pytest.approx(4.5) and G[0][0] == pytest.approx(1.5)


CODEGEN_TORTURE = """
`define SQ(x) ((x)*(x))
module cg_torture(a, b, c);
  inout a, b, c; electrical a, b, c; electrical n1;
  parameter real G0 = 1e-3 from (0:inf);
  parameter integer N = 3 from [0:8];
  parameter integer MODE = 2;
  parameter real VT = 0.5;
  real g, acc, vx, q1, tmp, w, tab[0:3];
  integer i, flags, k, pick[1:2];

  analog function real softplus;
    input x, s; real x, s;
    begin
      if (x/s > 30.0) softplus = x;
      else softplus = s*ln(1.0 + exp(x/s));
    end
  endfunction

  analog function real split;    // two outputs and an inout
    input x; output pos, neg; inout count;
    real x, pos, neg; integer count;
    begin
      pos = (x > 0.0) ? x : 0.0;
      neg = (x > 0.0) ? 0.0 : -x;
      count = count + 1;
      split = pos - neg;
    end
  endfunction

  analog begin : main
    real lpos, lneg;
    vx = V(a, b);
    // loops with integer control, accumulation of bias-dependent terms
    acc = 0.0;
    for (i = 0; i < N; i = i + 1) acc = acc + `SQ(vx) / (1.0 + i);
    k = 0;
    while (k < 2) begin acc = acc + 0.1*tanh(vx*(k + 1)); k = k + 1; end
    repeat (2) acc = acc * 1.01;
    // integer / bitwise / shift / modulo semantics
    flags = (MODE << 2) | 1;
    flags = flags ^ 2;
    if ((flags & 8) && !(flags & 4)) g = G0; else g = 2.0*G0;
    if (flags % 3 == 2) g = g*1.5;
    // case with several labels and default
    case (MODE)
      0: w = 0.0;
      1, 2: w = softplus(V(c) - VT, 0.05);
      default: w = 1.0;
    endcase
    // function with output arguments (dual instantiation) and an integer inout
    k = 10;
    tmp = split(V(c, b), lpos, lneg, k);
    // nested ternary, pow with dual exponent, hypot, atan2, min/max/abs
    q1 = 1e-12*( (vx > 0.2) ? pow(vx, 1.5) : ((vx < -0.2) ? -hypot(vx, 0.1) : vx) ) + 1e-13*atan2(V(c), 1.0 + abs(vx));
    I(a, b) <+ g*(acc + w*vx) + ddt(q1);
    I(c, b) <+ 1e-4*(lpos - 0.5*lneg) + 1e-6*(k - 10) + 1e-5*max(min(tmp, 0.3), -0.3);
    // array variables with computed indices
    for (i = 0; i <= 3; i = i + 1) tab[i] = tanh((i + 1)*V(c))/(i + 1);
    pick[1] = MODE % 4; pick[2] = 3 - pick[1];
    I(c, a) <+ 1e-5*(tab[pick[1]] - tab[pick[2]] + tab[N % 4]);
    I(a, n1) <+ 1e-2*V(a, n1);
    I(n1, c) <+ 1e-2*limexp(V(n1, c)) - 1e-2 + ddt(2e-12*V(n1, c)*V(n1, c));
  end
endmodule
"""


def test_generated_cpp_matches_interpreter_on_a_torture_module(tmp_path):
    """Loops, case, integer / bitwise operators, analog functions with output and inout arguments (dual instantiation),
    nested ternaries, pow with a dual base, block-scoped variables: generated C++ (compiled here with g++) vs interpreter."""
    import ctypes as C
    import subprocess
    from cedarsim_jl_amd.va.codegen import generate_header
    mods = parse_va(CODEGEN_TORTURE)
    hdr = generate_header(mods).replace('#include "../va_rt.hpp"', '#include "va_rt.hpp"')
    (tmp_path / "gen.hpp").write_text(hdr)
    (tmp_path / "shim.cpp").write_text('#include "gen.hpp"\nextern "C" void stamp(const double* P, const double* v, double T, double gmin, double* st) {\n'
                                       '  for (int k = 0; k < 144; ++k) st[k] = 0.0; const va::Env env{T, gmin}; va_gen::stamp(0, P, v, env, 1.0, st); }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libgen.so")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(root, "cedarsim.jl_amd", "csrc"), "-I", str(tmp_path),
                        str(tmp_path / "shim.cpp"), "-o", so], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    L = C.CDLL(so)
    pd = C.POINTER(C.c_double)
    L.stamp.argtypes = [pd, pd, C.c_double, C.c_double, pd]
    mod = mods[0]
    rng = np.random.default_rng(9)
    for params in ({}, {"MODE": 0, "N": 0}, {"MODE": 1, "N": 5, "G0": 2e-3}, {"MODE": 7, "VT": 0.1}):
        it = Interp(mod, params)
        P = np.array([float(it.params[p[0]]) for p in mod.params] + [1.0 if p[0] in it.given else 0.0 for p in mod.params])
        for _ in range(5):
            vb = {n: float(rng.uniform(-0.6, 0.9)) for n in mod.nodes}
            I, Q, G, Cm = it.evaluate(vb)
            v = np.zeros(8)
            v[:4] = [vb[n] for n in mod.nodes]
            st = np.zeros(144)
            L.stamp(P.ctypes.data_as(pd), v.ctypes.data_as(pd), 300.15, 1e-12, st.ctypes.data_as(pd))
            for got, want in ((st[:4], I), (st[8:12], Q), (st[16:80].reshape(8, 8)[:4, :4], G), (st[80:144].reshape(8, 8)[:4, :4], Cm)):
                want = np.array(want, float)
                assert np.allclose(got, want, rtol=1e-11, atol=1e-11 * max(np.abs(want).max(), 1e-300)), (params, vb)


SPLIT_TORTURE = """
module cg_split(a, b, c);
  inout a, b, c; electrical a, b, c;
  parameter real G0 = 1e-3;
  parameter real K = 0.7;
  parameter integer MODE = 1;
  parameter real TNOM = 27.0;
  real T0, T1, T2, vab, vcb, gt, acc, late, carry, loopv, sel;
  integer i, flag, cnt;

  analog function real twice;
    input x; output sq; real x, sq;
    begin sq = x*x; twice = 2.0*x; end
  endfunction

  analog function real bump;
    input x; inout n; real x; integer n;
    begin n = n + 1; bump = x * n; end
  endfunction

  analog begin : main
    real loc;
    // temporaries that hold bias-independent and bias-dependent values in turn
    T0 = exp(K) / (1.0 + K);             // static
    gt = G0 * T0 * ($temperature / (TNOM + 273.15));
    vab = V(a, b);
    T0 = T0 * vab;                       // now dynamic
    T1 = ln(2.0 + K);                    // static
    I(a, b) <+ gt * (T0 + T1 * vab * vab);
    T0 = sqrt(K + 1.0);                  // static again
    // static condition, both branches static with different values -> merge slot
    if (MODE > 0) begin T2 = T0 * 3.0; flag = 1; end else begin T2 = T0 / 3.0; flag = 2; end
    // static condition, one branch bias-dependent -> the other is materialised
    if (MODE == 2) carry = tanh(vab); else carry = 0.25 * T2;
    // bias-dependent condition around bias-independent right-hand sides (incl. an expensive one that is hoisted)
    vcb = V(c, b);
    late = 0.5;
    if (vcb > 0.1) begin late = pow(K, 1.5) + T1; loc = 2.0; end
    else if (vcb < -0.1) late = -T2;
    // case on a parameter
    case (MODE)
      0: sel = 0.0;
      1, 3: sel = T1 * K;
      default: begin sel = vab * K; end
    endcase
    // variables assigned in a loop are bias-dependent from there on; the loop reads static values
    acc = T2;
    for (i = 0; i < 3; i = i + 1) acc = acc + T1 * (i + 1) * vcb;
    loopv = 0.0;
    cnt = 0;
    while (cnt < flag) begin loopv = loopv + T0; cnt = cnt + 1; end
    // case with a bias-dependent selector; a variable assigned in both branches of a bias-dependent condition
    case (vab > 0.2)
      1: begin T1 = K; cnt = 4; end
      default: begin T1 = 2.0*K; cnt = 5; end
    endcase
    if (vcb > 0.0) T0 = T0 + 1.0; else T0 = T0 - 1.0;
    I(a, c) <+ 1e-6 * (T1 + cnt + T0);
    // a named block whose local shadows an outer variable: the outer one keeps its (bias-independent) value
    begin : inner
      real T2;
      T2 = vab * 3.0;
      I(a, c) <+ 1e-7 * T2;
    end
    I(a, c) <+ 1e-7 * T2 * vcb;
    // an integer inout argument counted through two calls, one of them on a bias-dependent input
    cnt = 0;
    T0 = bump(K, cnt);
    T0 = T0 + bump(vab, cnt);
    I(c, b) <+ 1e-5 * (T0 + cnt);
    // an analog function with an output argument on bias-independent input
    T1 = twice(K, T2);
    I(c, b) <+ 1e-3 * (carry + late + sel + acc + loopv * vcb + T1 * vab + T2 * vcb + loc * 0.0) + ddt(1e-12 * flag * T0 * vcb * vcb);
    if (MODE == 3) I(a, c) <+ 1e-4 * V(a, c) * T0;
  end
endmodule
"""


def test_setup_eval_split_on_a_binding_time_torture_module(tmp_path):
    """The setup/eval split (codegen.ModuleGen._split): temporaries reused for bias-independent and bias-dependent values,
    merges after bias-independent conditions, materialisation where only one branch depends on the bias, a `case` on a
    parameter, hoisted sub-expressions under bias-dependent control, loops, output arguments.  The constant block is built ONCE
    per parameter set and reused for every bias (as the engine does); the result must equal the interpreter's."""
    import ctypes as C
    import subprocess
    from cedarsim_jl_amd.va.codegen import generate_header
    mods = parse_va(SPLIT_TORTURE)
    hdr = generate_header(mods).replace('#include "../va_rt.hpp"', '#include "va_rt.hpp"')
    assert "void setup(" in hdr
    (tmp_path / "gen.hpp").write_text(hdr)
    (tmp_path / "shim.cpp").write_text('#include "gen.hpp"\nextern "C" int n_cache() { return va_gen::N_CACHE[0]; }\n'
                                       'extern "C" void setup(const double* P, double T, double gmin, double* Cc) { const va::Env env{T, gmin}; va_gen::setup(0, P, env, Cc); }\n'
                                       'extern "C" void stamp_c(const double* P, const double* Cc, const double* v, double T, double gmin, double* st) {\n'
                                       '  for (int k = 0; k < 144; ++k) st[k] = 0.0; const va::Env env{T, gmin}; va_gen::stamp_c(0, P, Cc, v, env, 1.0, st); }\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libgen.so")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(root, "cedarsim.jl_amd", "csrc"), "-I", str(tmp_path),
                        str(tmp_path / "shim.cpp"), "-o", so], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    L = C.CDLL(so)
    pd = C.POINTER(C.c_double)
    L.setup.argtypes = [pd, C.c_double, C.c_double, pd]
    L.stamp_c.argtypes = [pd, pd, pd, C.c_double, C.c_double, pd]
    nc = L.n_cache()
    assert 4 <= nc <= 60, nc
    mod = mods[0]
    rng = np.random.default_rng(11)
    for params in ({}, {"MODE": 0}, {"MODE": 2, "K": 0.3}, {"MODE": 3, "G0": 2e-3}, {"MODE": -1, "K": 1.4}):
        it = Interp(mod, params, temperature_c=320.0 - 273.15)
        P = np.array([float(it.params[p[0]]) for p in mod.params] + [1.0 if p[0] in it.given else 0.0 for p in mod.params])
        Cc = np.full(max(1, nc), np.nan)
        L.setup(P.ctypes.data_as(pd), 320.0, 1e-12, Cc.ctypes.data_as(pd))
        for _ in range(6):
            vb = {n: float(rng.uniform(-0.6, 0.9)) for n in mod.nodes}
            I, Q, G, Cm = it.evaluate(vb)
            v = np.zeros(8)
            v[:3] = [vb[n] for n in mod.nodes]
            st = np.zeros(144)
            L.stamp_c(P.ctypes.data_as(pd), Cc.ctypes.data_as(pd), v.ctypes.data_as(pd), 320.0, 1e-12, st.ctypes.data_as(pd))
            for got, want in ((st[:3], I), (st[8:11], Q), (st[16:80].reshape(8, 8)[:3, :3], G), (st[80:144].reshape(8, 8)[:3, :3], Cm)):
                want = np.array(want, float)
                assert np.allclose(got, want, rtol=1e-11, atol=1e-11 * max(np.abs(want).max(), 1e-300)), (params, vb, got, want)


def test_switch_branch_state_semantics():
    """A branch that receives a voltage contribution on one path and a current contribution on the other: the branch state
    follows the last contribution executed (src/vasim.jl:128-180, 810-822)."""
    _, mod = find_module("va_switch")
    assert mod.nodes == ["p", "n", "c", "I(p,n)"]
    # closed: row x_br: V(p,n) - RON*x_br ; open: x_br - GOFF*V(p,n)
    I, Q, G, C = Interp(mod, {"RON": 2.0}).evaluate({"p": 1.0, "n": 0.0, "c": 1.0, "I(p,n)": 0.25})
    assert I == [0.25, -0.25, 0.0, 1.0 - 2.0 * 0.25] and G[3] == [1.0, -1.0, 0.0, -2.0]
    I, Q, G, C = Interp(mod, {"GOFF": 1e-3}).evaluate({"p": 1.0, "n": 0.0, "c": 0.0, "I(p,n)": 0.25})
    assert I == [0.25, -0.25, 0.0, 0.25 - 1e-3] and G[3] == [-1e-3, 1e-3, 0.0, 1.0]
    for vc, want in ((1.0, 1000.0 / 1001.0), (0.0, 1e-9 * 1000.0 / (1 + 1e-9 * 1000.0))):
        c = Circuit()
        c.V("v1", "in", 0, dc=1.0)
        c.V("vc", "ctl", 0, dc=vc)
        c.VA("s1", "va_switch", ["in", "out", "ctl"], {})
        c.R("rl", "out", 0, 1e3)
        rc, x, _ = Oracle(c).dc(dc_opts(abstol=1e-15))
        assert rc == 0 and x[c._n("out") - 1] == pytest.approx(want, rel=1e-9)


def test_generated_code_matches_hand_derived_closed_forms(oracle_lib):
    """The compiler (front end + code generator) against stamps derived by hand from the model equations
    (tests/golden/make_va_closed_form.py shares nothing with cedarsim.jl_amd/va): g++ build of the generated code here,
    the HIP build in tests/test_gpu_parity_wide.py."""
    import ctypes as C
    import json
    import os
    import numpy as np
    from cedarsim_jl_amd.va.registry import find_module
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "va_closed_form.json")))
    pd = C.POINTER(C.c_double)
    for case in g["cases"]:
        mid, _ = find_module(case["module"])
        P = np.array(case["par_block"], float)
        v = np.zeros(8)
        v[:len(case["v"])] = case["v"]
        ref = np.zeros(144)
        assert oracle_lib.oracle_va_eval(mid, P.ctypes.data_as(pd), v.ctypes.data_as(pd), case["temperature_k"], 0.0, ref.ctypes.data_as(pd)) == 0
        I, G = ref[:8], ref[16:80].reshape(8, 8)
        for k, w in case["I"]:
            assert abs(I[k] - w) <= 1e-12 * max(1.0, abs(w)) + 1e-25, (case["module"], "I", k)
        for r, c, w in case["G"]:
            assert abs(G[r, c] - w) <= 1e-11 * max(1.0, abs(w)) + 1e-25, (case["module"], "G", r, c)
