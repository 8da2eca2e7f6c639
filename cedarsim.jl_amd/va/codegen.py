"""Verilog-A → C++ (HIP device + host) code generator — the MI355X counterpart of `make_spice_device`
(src/vasim.jl:649-867), which turns a parsed module into a Julia device functor.

For every module one function template is emitted:

    template <class R> void eval_<module>(const double* P, const R* V, const va::Env& env, R* I, R* Q)

P  : parameter values in declaration order followed by the same number of "given" flags
V  : node voltages (ports then internal nets), seeded as dual numbers by the dispatcher
I/Q: per-node sums of the resistive currents / the charges under ddt() leaving the node through the device
     (`I(a,b) <+ f + ddt(q)` adds f to I[a], −f to I[b], q to Q[a], −q to Q[b])

plus a registry (names of modules, nodes, parameters) and `va_gen::stamp(module, …)`, which seeds the
duals, runs the module and scatters values and derivatives into the engine's wide stamp record
[I(8) | Q(8) | G(8×8) | C(8×8)].  The same header is compiled into libcedarhip.so (device) and into the
test oracle (host), so a model added to the VA library is available to both after a rebuild.

Typing: VA `integer` → int, VA `real` → double unless a fixed-point pass finds that the variable can
depend on a node voltage, in which case it is the dual type R.  Analog functions are templates over one
scalar type S, instantiated with R when any argument is dual.
"""
from .frontend import FLOW_ACCESS, POTENTIAL_ACCESS, VAError, _is_zero

MAX_NODES = 8

_F1 = {"exp", "ln", "sqrt", "sin", "cos", "tan", "sinh", "cosh", "tanh", "atan", "asin", "acos", "asinh", "acosh", "atanh", "abs",
       "floor", "ceil", "limexp"}
_F2 = {"pow", "min", "max", "atan2", "hypot"}
_RENAME = {"log": "log10"}


def _walk(e):
    if isinstance(e, tuple):
        yield e
        for c in e:
            yield from _walk(c)
    elif isinstance(e, (list,)):
        for c in e:
            yield from _walk(c)
    elif isinstance(e, dict):
        for c in e.values():
            yield from _walk(c)


def _has_ddt(e):
    return any(n[0] == "call" and n[1] == "ddt" for n in _walk(e) if len(n) >= 2)


class ModuleGen:
    def __init__(self, module):
        self.m = module
        if len(module.nodes) > MAX_NODES:
            raise VAError("module %s has %d nodes; the engine's stamp record holds %d" % (module.name, len(module.nodes), MAX_NODES))
        self.node_ix = {n: i for i, n in enumerate(module.nodes)}
        self.param_ix = {p[0]: i for i, p in enumerate(module.params)}
        self.param_ty = {p[0]: p[1] for p in module.params}
        self.ddx_nodes = self._ddx_nodes()
        self.all_vars = dict(module.vars)
        for n in _walk(module.analog):
            if n and n[0] == "block" and isinstance(n[2], dict):
                self.all_vars.update(n[2])
        self.dual = set()
        self._infer_dual()
        self.tmp = 0

    # ---- analyses ----
    def _ddx_nodes(self):
        order = []
        for n in _walk([self.m.analog] + [f.body for f in self.m.functions.values()]):
            if len(n) >= 3 and n[0] == "call" and n[1] == "ddx":
                probe = n[2][1]
                if probe[0] != "call" or probe[1] not in POTENTIAL_ACCESS:
                    raise VAError("ddx(): the second argument must be a potential probe V(a) or V(a,b)")
                for a in probe[2]:
                    if a[1] not in order:
                        order.append(a[1])
        return order

    def _is_dual(self, e):
        k = e[0]
        if k in ("num", "str"):
            return False
        if k in ("id", "index"):
            return e[1] in self.dual
        if k == "un":
            return self._is_dual(e[2])
        if k == "bin":
            if e[1] in ("<", "<=", ">", ">=", "==", "!=", "&&", "||", "&", "|", "^", "<<", ">>", "%"):
                return False
            return self._is_dual(e[2]) or self._is_dual(e[3])
        if k == "tern":
            return self._is_dual(e[2]) or self._is_dual(e[3])
        if k == "call":
            name = e[1]
            if name in POTENTIAL_ACCESS or name in FLOW_ACCESS or name == "ddx":
                return True
            if name.startswith("$") and name[1:] not in _F1 and name[1:] not in _F2 and name not in ("$limit",):
                return False
            if name in ("floor", "ceil", "$floor", "$ceil", "white_noise", "flicker_noise"):
                return False
            return any(self._is_dual(a) for a in e[2])
        return False

    def _infer_dual(self):
        changed = True
        while changed:
            changed = False
            for n in _walk(self.m.analog):
                if not n:
                    continue
                if n[0] == "assign" and n[1] not in self.dual and self.all_vars.get(n[1]) == "real" and self._is_dual(n[2]):
                    self.dual.add(n[1])
                    changed = True
                if n[0] == "assign_idx" and n[1] not in self.dual and self.all_vars.get(n[1]) == "real" and self._is_dual(n[3]):
                    self.dual.add(n[1])
                    changed = True
                if n[0] == "call" and n[1] in self.m.functions:
                    f = self.m.functions[n[1]]
                    if any(self._is_dual(a) for a in n[2]):
                        for (nm, kind), a in zip(f.args, n[2]):
                            if kind in ("output", "inout") and a[0] == "id" and a[1] not in self.dual and self.all_vars.get(a[1]) == "real":
                                self.dual.add(a[1])
                                changed = True

    # ---- expressions: returns (code, type) with type in int|real|dual ----
    def cast(self, code, ty, to, S):
        if ty == to:
            return code
        if to == "dual":
            return "%s(%s)" % (S, code) if ty == "real" else "%s((double)(%s))" % (S, code)
        if to == "real":
            if ty == "int":
                return "(double)(%s)" % code
            return "va::val(%s)" % code
        if to == "int":
            return "va::to_int(%s)" % code
        raise VAError("cast %s -> %s" % (ty, to))

    @staticmethod
    def promote(a, b):
        return "dual" if "dual" in (a, b) else ("real" if "real" in (a, b) else "int")

    def expr(self, e, ctx):
        """ctx: dict(vars={name: type}, S=scalar type name for 'dual', infunc=bool)"""
        k = e[0]
        S = ctx["S"]
        if ctx.get("setup_out") is not None and self._worth_hoisting(e) and self.is_static(e, ctx["cache"], ctx["vars"]):
            # a bias-independent sub-expression of a bias-dependent statement: evaluated by setup() at this point of the
            # (static) control flow, loaded here
            c, t = self.expr(e, ctx["setup_ctx"])
            slot = self._new_slot()
            ctx["setup_out"].append((slot, "%sC[@%d@] = (double)(%s);" % (ctx.get("pad", "  "), slot, c)))
            self.used_slots.add(slot)
            return ("(int)C[@%d@]" % slot, "int") if t == "int" else ("C[@%d@]" % slot, "real")
        if k == "num":
            if e[2]:
                return str(e[1]), "int"
            v = e[1]
            if v != v:
                return "NAN", "real"
            if v in (float("inf"), float("-inf")):
                return ("INFINITY" if v > 0 else "-INFINITY"), "real"
            return repr(float(v)), "real"
        if k == "id":
            name = e[1]
            if name in ctx["vars"]:
                cache = ctx.get("cache")
                if cache is not None and cache.get(name) is not None:
                    return self._load(name, cache[name], ctx["vars"][name])
                return "v_" + name, ctx["vars"][name]
            if name in self.param_ix:
                ty = self.param_ty[name]
                if ty == "string":
                    raise VAError("string parameter '%s' cannot be used in an expression" % name)
                return "p_" + name, ("int" if ty == "integer" else "real")
            raise VAError("undefined identifier '%s' in module %s" % (name, self.m.name))
        if k == "str":
            raise VAError("string in an arithmetic expression")
        if k == "index":
            name = e[1]
            if name not in ctx["vars"] or name not in self.m.arrays:
                raise VAError("'%s' is not an array variable" % name)
            ic, it = self.expr(e[2], ctx)
            return "v_%s[va::clamp_index(%s, %d, %d)]" % (name, self.cast(ic, it, "int", S), self.m.arrays[name][0], self.m.arrays[name][1]), ctx["vars"][name]
        if k == "un":
            c, t = self.expr(e[2], ctx)
            if e[1] == "-":
                return "(-%s)" % c, t
            if e[1] == "!":
                return "(va::truth(%s) ? 0 : 1)" % c, "int"
            return "(~%s)" % self.cast(c, t, "int", S), "int"
        if k == "bin":
            op = e[1]
            a, ta = self.expr(e[2], ctx)
            b, tb = self.expr(e[3], ctx)
            if op in ("+", "-", "*"):
                return "(%s %s %s)" % (a, op, b), self.promote(ta, tb)
            if op == "/":   # always real division (the reference maps `/` to Julia's `/`, src/vasim.jl:221-232)
                t = self.promote(self.promote(ta, tb), "real")
                return "va::v_div(%s, %s)" % (self.cast(a, ta, "real", S) if ta == "int" else a, self.cast(b, tb, "real", S) if tb == "int" else b), t
            if op == "**":
                t = self.promote(self.promote(ta, tb), "real")
                if ta == "dual" and tb != "dual":
                    return "va::v_pow(%s, %s)" % (a, b), "dual"
                return "va::v_pow(%s, %s)" % (self.cast(a, ta, t, S), self.cast(b, tb, t, S)), t
            if op == "%":
                if ta == "int" and tb == "int":
                    return "(%s %% %s)" % (a, b), "int"
                return "::fmod(%s, %s)" % (self.cast(a, ta, "real", S), self.cast(b, tb, "real", S)), "real"
            if op in ("<", "<=", ">", ">=", "==", "!="):
                av = a if ta == "int" else self.cast(a, ta, "real", S)
                bv = b if tb == "int" else self.cast(b, tb, "real", S)
                return "((%s %s %s) ? 1 : 0)" % (av, op, bv), "int"
            if op in ("&&", "||"):
                return "((va::truth(%s) %s va::truth(%s)) ? 1 : 0)" % (a, op, b), "int"
            return "(%s %s %s)" % (self.cast(a, ta, "int", S), op, self.cast(b, tb, "int", S)), "int"
        if k == "tern":
            c, _ = self.expr(e[1], ctx)
            a, ta = self.expr(e[2], ctx)
            b, tb = self.expr(e[3], ctx)
            t = self.promote(ta, tb)
            return "(va::truth(%s) ? %s : %s)" % (c, self.cast(a, ta, t, S), self.cast(b, tb, t, S)), t
        if k == "call":
            return self.call(e, ctx)
        raise VAError("cannot generate %r" % (e,))

    def probe(self, acc, nodes, ctx):
        if ctx.get("infunc"):
            raise VAError("branch probes inside analog functions are not supported")
        if len(nodes) == 1 and nodes[0] in self.m.branches:
            nodes = [x for x in self.m.branches[nodes[0]] if x is not None]
        if acc in FLOW_ACCESS:
            vb = self.m.find_vbranch(nodes)
            if vb is None:
                raise VAError("flow probe %s(%s): only the current of a voltage branch can be probed" % (acc, ",".join(nodes)))
            ty = "real" if ctx.get("noise") else "dual"
            return ("n%d_" if vb[1] > 0 else "(-n%d_)") % self.node_ix[vb[0]], ty
        for n in nodes:
            if n not in self.node_ix:
                raise VAError("unknown node '%s' in module %s" % (n, self.m.name))
        ty = "real" if ctx.get("noise") else "dual"
        if len(nodes) == 1:
            return "n%d_" % self.node_ix[nodes[0]], ty
        return "(n%d_ - n%d_)" % (self.node_ix[nodes[0]], self.node_ix[nodes[1]]), ty

    def call(self, e, ctx):
        name, args, S = e[1], e[2], ctx["S"]
        if name in POTENTIAL_ACCESS or name in FLOW_ACCESS:
            return self.probe(name, [a[1] for a in args], ctx)
        if name == "$temperature":
            return "env.temperature", "real"
        if name == "$vt":
            if args:
                c, t = self.expr(args[0], ctx)
                return "(%s * 8.617343e-5)" % c if False else "(%s * (1.3806503e-23 / 1.602176462e-19))" % c, self.promote(t, "real")
            return "(env.temperature * (1.3806503e-23 / 1.602176462e-19))", "real"
        if name in ("$param_given", "$given"):
            pn = args[0][1]
            pn = self.m.aliases.get(pn, pn)
            if pn not in self.param_ix:
                raise VAError("$param_given(%s): no such parameter" % pn)
            return "g_" + pn, "int"
        if name == "$simparam":
            if args[0][0] == "str" and args[0][1] == "gmin":
                return "env.gmin", "real"
            if len(args) > 1:
                return self.expr(args[1], ctx)
            raise VAError("$simparam(\"%s\") has no value" % (args[0][1],))
        if name == "$mfactor":
            return "1.0", "real"
        if name == "$port_connected":
            return "1", "int"
        if name in ("$abstime", "$realtime"):
            return "0.0", "real"
        if name == "$limit":
            return self.expr(args[0], ctx)
        if name in ("white_noise", "flicker_noise"):
            return "0.0", "real"
        if name == "ddt":
            raise VAError("ddt() is only supported as an additive (possibly scaled) term of a contribution")
        if name == "ddx":
            c, t = self.expr(args[0], ctx)
            if ctx.get("noise"):
                raise VAError("ddx() in a module with noise sources is not supported by the noise pass")
            ix = [self.ddx_nodes.index(a[1]) for a in args[1][2]]
            c = self.cast(c, t, "dual", S)
            if len(ix) == 1:
                return "va::ddx1(%s, %d)" % (c, ix[0]), "dual"
            return "va::ddx2(%s, %d, %d)" % (c, ix[0], ix[1]), "dual"
        base = name[1:] if name.startswith("$") else name
        base = _RENAME.get(base, base)
        if base in _F1 and (name in _F1 or name.startswith("$") or name == "log"):
            c, t = self.expr(args[0], ctx)
            if base in ("floor", "ceil"):
                return "va::v_%s(%s)" % (base, self.cast(c, t, "real", S) if t == "int" else c), "real"
            if base == "abs" and t == "int":
                return "va::v_abs(%s)" % c, "int"
            t2 = self.promote(t, "real")
            return "va::v_%s(%s)" % (base, self.cast(c, t, t2, S)), t2
        if base == "log10":
            c, t = self.expr(args[0], ctx)
            t2 = self.promote(t, "real")
            return "va::v_log10(%s)" % self.cast(c, t, t2, S), t2
        if base in _F2:
            a, ta = self.expr(args[0], ctx)
            b, tb = self.expr(args[1], ctx)
            t = self.promote(ta, tb)
            if base in ("min", "max") and t == "int":
                return "va::v_%s(%s, %s)" % (base, a, b), "int"
            t = self.promote(t, "real")
            if base == "pow" and ta == "dual" and tb != "dual":
                return "va::v_pow(%s, %s)" % (a, b), "dual"
            return "va::v_%s(%s, %s)" % (base, self.cast(a, ta, t, S), self.cast(b, tb, t, S)), t
        if name in self.m.functions:
            return self.user_call(self.m.functions[name], args, ctx)
        raise VAError("unknown function '%s' in module %s" % (name, self.m.name))

    def user_call(self, f, args, ctx):
        if len(args) != len(f.args):
            raise VAError("function %s expects %d arguments, got %d" % (f.name, len(f.args), len(args)))
        S = ctx["S"]
        ev = [self.expr(a, ctx) if kind != "output" else (None, None) for (nm, kind), a in zip(f.args, args)]
        # outputs whose target variable is dual force the dual instantiation as well
        anydual = any(t == "dual" for c, t in ev if c is not None)
        FS = S if anydual else "double"
        fty = "dual" if anydual else "real"
        call_args, pre, post = [], [], []
        has_out = any(kind != "input" for _, kind in f.args)
        for (nm, kind), a, (c, t) in zip(f.args, args, ev):
            aty = f.vars.get(nm, "real")
            if kind == "input":
                call_args.append(self.cast(c, t, "int" if aty == "integer" else fty, S))
                continue
            if a[0] != "id" or a[1] not in ctx["vars"]:
                raise VAError("output argument of %s must be a variable" % f.name)
            self.tmp += 1
            tn = "o%d_" % self.tmp
            tt = "int" if aty == "integer" else FS
            init = ""
            if kind == "inout":
                init = " = " + self.cast(c, t, "int" if aty == "integer" else fty, S)
            pre.append("%s %s%s;" % (tt, tn, init))
            call_args.append(tn)
            vty = ctx["vars"][a[1]]
            post.append("v_%s = %s;" % (a[1], self.cast(tn, "int" if aty == "integer" else fty, vty, S)))
        rty = "int" if f.rtype == "integer" else fty
        callc = "f_%s<%s>(%s)" % (f.name, FS, ", ".join(["env"] + call_args))
        if not has_out:
            return callc, rty
        rt = "int" if rty == "int" else FS
        return "([&]() -> %s { %s %s r_ = %s; %s return r_; }())" % (rt, " ".join(pre), rt, callc, " ".join(post)), rty

    # ---- statements ----
    def split_ddt(self, e):
        """(resistive AST or None, reactive AST or None)"""
        k = e[0]
        if k == "call" and e[1] == "ddt":
            return None, e[2][0]
        if not _has_ddt(e):
            return e, None
        if k == "bin" and e[1] in ("+", "-"):
            ar, aq = self.split_ddt(e[2])
            br, bq = self.split_ddt(e[3])

            def comb(x, y):
                if x is None and y is None:
                    return None
                if y is None:
                    return x
                if x is None:
                    return y if e[1] == "+" else ("un", "-", y)
                return ("bin", e[1], x, y)
            return comb(ar, br), comb(aq, bq)
        if k == "un" and e[1] == "-":
            r, q = self.split_ddt(e[2])
            return (None if r is None else ("un", "-", r)), (None if q is None else ("un", "-", q))
        if k == "bin" and e[1] == "*":
            for x, y, left in ((e[2], e[3], True), (e[3], e[2], False)):
                if _has_ddt(x) and not _has_ddt(y):
                    r, q = self.split_ddt(x)

                    def mul(z):
                        return None if z is None else (("bin", "*", z, y) if left else ("bin", "*", y, z))
                    return mul(r), mul(q)
        if k == "bin" and e[1] == "/" and _has_ddt(e[2]) and not _has_ddt(e[3]):
            r, q = self.split_ddt(e[2])
            return (None if r is None else ("bin", "/", r, e[3])), (None if q is None else ("bin", "/", q, e[3]))
        if k == "tern":
            ar, aq = self.split_ddt(e[2])
            br, bq = self.split_ddt(e[3])
            zero = ("num", 0.0, False)
            r = None if ar is None and br is None else ("tern", e[1], ar or zero, br or zero)
            q = None if aq is None and bq is None else ("tern", e[1], aq or zero, bq or zero)
            return r, q
        raise VAError("ddt() must appear as an additive (possibly scaled) term of a contribution")

    def stmt(self, st, ctx, ind):
        k = st[0]
        pad = "  " * ind
        S = ctx["S"]
        if k == "assign_idx":
            name = st[1]
            if name not in ctx["vars"] or name not in self.m.arrays:
                raise VAError("'%s' is not an array variable" % name)
            ic, it = self.expr(st[2], ctx)
            c, t = self.expr(st[3], ctx)
            lo, hi = self.m.arrays[name]
            return ["%sv_%s[va::clamp_index(%s, %d, %d)] = %s;" % (pad, name, self.cast(ic, it, "int", S), lo, hi, self.cast(c, t, ctx["vars"][name], S))]
        if k == "assign":
            name = st[1]
            if name not in ctx["vars"]:
                raise VAError("assignment to undeclared variable '%s' in module %s" % (name, self.m.name))
            c, t = self.expr(st[2], ctx)
            return ["%sv_%s = %s;" % (pad, name, self.cast(c, t, ctx["vars"][name], S))]
        if k == "contrib":
            if ctx.get("infunc"):
                raise VAError("contribution inside an analog function")
            acc, nodes = st[1], st[2]
            if len(nodes) == 1 and nodes[0] in self.m.branches:
                nodes = [x for x in self.m.branches[nodes[0]] if x is not None]
            if acc in POTENTIAL_ACCESS:
                if _is_zero(st[3]):
                    # V(a,b) <+ 0: node collapse, resolved structurally on the host (the two nodes are merged before
                    # the circuit reaches the engine)
                    return ["%s/* V(%s) <+ 0: node collapse handled at circuit build */" % (pad, ",".join(nodes))]
                if ctx.get("noise"):
                    return []
                return self._branch_contrib(nodes, 1, st[3], ctx, pad)
            if acc not in FLOW_ACCESS:
                raise VAError("unknown access function %s" % acc)
            rhs = st[3]
            is_noise = rhs[0] == "call" and rhs[1] in ("white_noise", "flicker_noise")
            if ctx.get("opvars"):
                return []
            if ctx.get("noise"):
                # noise pass: `I(a,b) <+ white_noise(pwr, name)` / `flicker_noise(pwr, exp, name)` become records
                # (src/va_env.jl:92-101: the power is an observable, the source an epsilon of the linearisation)
                if not is_noise:
                    return []
                nargs = [x for x in rhs[2] if x[0] != "str"]
                pc, pt = self.expr(nargs[0], ctx)
                ec, et = (self.expr(nargs[1], ctx) if rhs[1] == "flicker_noise" else ("0.0", "real"))
                a = self.node_ix[nodes[0]]
                b = self.node_ix[nodes[1]] if len(nodes) > 1 else -1
                return ["%sif (n_ < va::MAX_NOISE) { out[n_].a = %d; out[n_].b = %d; out[n_].pwr = %s; out[n_].ex = %s; ++n_; }" %
                        (pad, a, b, self.cast(pc, pt, "real", S), self.cast(ec, et, "real", S))]
            if is_noise:
                return []
            if self.m.find_vbranch(nodes) is not None:   # current contribution to a voltage / switch branch
                return self._branch_contrib(nodes, 0, st[3], ctx, pad)
            r, q = self.split_ddt(st[3])
            a = self.node_ix[nodes[0]]
            b = self.node_ix[nodes[1]] if len(nodes) > 1 else None
            out = []
            for ast, arr in ((r, "I"), (q, "Q")):
                if ast is None:
                    continue
                c, t = self.expr(ast, ctx)
                acc = arr.lower()   # node sums are kept in locals (i0_, q0_, ...) and stored once at the end: the caller's I[] / Q[]
                #                     live in scratch (the function is not inlined), a read-modify-write there per contribution
                out.append("%sif (PART != %d) { const %s c_ = %s; %s%d_ += c_;%s }" % (pad, 1 if arr == "I" else 0, S, self.cast(c, t, "dual", S), acc, a,
                                                                                      (" %s%d_ -= c_;" % (acc, b)) if b is not None else ""))
            return out
        if k == "if":
            c, _ = self.expr(st[1], ctx)
            out = ["%sif (va::truth(%s)) {" % (pad, c)] + self.stmt(st[2], ctx, ind + 1)
            if st[3] is not None:
                out += ["%s} else {" % pad] + self.stmt(st[3], ctx, ind + 1)
            return out + ["%s}" % pad]
        if k == "block":
            out = ["%s{" % pad]
            if st[2]:
                ctx = dict(ctx, vars=dict(ctx["vars"]))
                if ctx.get("cache") is not None:
                    ctx["cache"] = dict(ctx["cache"])
                for nm, ty in st[2].items():
                    t = "int" if ty == "integer" else ("dual" if (ctx.get("infunc") or nm in self.dual) else "real")
                    ctx["vars"][nm] = t
                    if ctx.get("cache") is not None:
                        ctx["cache"][nm] = None
                    out.append("%s  %s v_%s%s;" % (pad, {"int": "int", "real": "double", "dual": S}[t], nm, self._decl_suffix(nm)))
            for s in st[3]:
                out += self.stmt(s, ctx, ind + 1)
            return out + ["%s}" % pad]
        if k == "case":
            c, t = self.expr(st[1], ctx)
            self.tmp += 1
            sv = "sw%d_" % self.tmp
            out = ["%s{ const double %s = %s;" % (pad, sv, self.cast(c, t, "real", S) if t != "real" else c)]
            first, default = True, None
            for conds, body in st[2]:
                if conds is None:
                    default = body
                    continue
                tests = []
                for cd in conds:
                    cc, ct = self.expr(cd, ctx)
                    tests.append("%s == %s" % (sv, self.cast(cc, ct, "real", S) if ct != "real" else cc))
                out.append("%s%sif (%s) {" % (pad, "" if first else "} else ", " || ".join(tests)))
                out += self.stmt(body, ctx, ind + 1)
                first = False
            if default is not None:
                out.append("%s%s{" % (pad, "" if first else "} else "))
                out += self.stmt(default, ctx, ind + 1)
                first = False
            if not first:
                out.append("%s}" % pad)
            return out + ["%s}" % pad]
        if k == "for":
            init = self.stmt(st[1], ctx, 0)[0]
            c, _ = self.expr(st[2], ctx)
            upd = self.stmt(st[3], ctx, 0)[0].rstrip(";")
            return ["%sfor (%s va::truth(%s); %s) {" % (pad, init, c, upd)] + self.stmt(st[4], ctx, ind + 1) + ["%s}" % pad]
        if k == "while":
            c, _ = self.expr(st[1], ctx)
            return ["%swhile (va::truth(%s)) {" % (pad, c)] + self.stmt(st[2], ctx, ind + 1) + ["%s}" % pad]
        if k == "repeat":
            c, t = self.expr(st[1], ctx)
            self.tmp += 1
            return ["%sfor (int r%d_ = 0, n%d_ = %s; r%d_ < n%d_; ++r%d_) {" % (pad, self.tmp, self.tmp, self.cast(c, t, "int", S), self.tmp, self.tmp, self.tmp)] + \
                self.stmt(st[2], ctx, ind + 1) + ["%s}" % pad]
        if k == "event":
            return self.stmt(st[1], ctx, ind)
        if k in ("task", "null"):
            return []
        raise VAError("cannot generate statement %r" % (st,))

    def _branch_contrib(self, nodes, kind, rhs, ctx, pad):
        """Contribution to a voltage / switch branch: state (0 CURRENT, 1 VOLTAGE) and value as in src/vasim.jl:128-180."""
        S = ctx["S"]
        name, sgn = self.m.find_vbranch(nodes)
        k = [self.m.branch_node(key) for key in self.m.vbranches].index(name)
        r, q = self.split_ddt(rhs)
        out = ["%sif (bs%d_ != %d) { bs%d_ = %d; bv%d_ = %s(0.0); bq%d_ = %s(0.0); }" % (pad, k, kind, k, kind, k, S, k, S)]
        for ast, var in ((r, "bv"), (q, "bq")):
            if ast is None:
                continue
            c, t = self.expr(ast, ctx)
            out.append("%sif (PART != %d) %s%d_ %s %s;" % (pad, 1 if var == "bv" else 0, var, k, "+=" if sgn > 0 else "-=", self.cast(c, t, "dual", S)))
        return out

    def _decl_suffix(self, name):
        if name in self.m.arrays:
            lo, hi = self.m.arrays[name]
            return "[%d] = {}" % (hi - lo + 1)
        return " = 0"

    def q_mask(self):
        """Bit k set: node k receives a ddt() contribution somewhere in the analog block."""
        mask = 0
        for n in _walk(self.m.analog):
            if n and n[0] == "contrib" and n[1] in POTENTIAL_ACCESS and not _is_zero(n[3]) and _has_ddt(n[3]):
                nd = list(n[2])
                if len(nd) == 1 and nd[0] in self.m.branches:
                    nd = [x for x in self.m.branches[nd[0]] if x is not None]
                mask |= 1 << self.node_ix[self.m.find_vbranch(nd)[0]]
            if n and n[0] == "contrib" and n[1] in FLOW_ACCESS and _has_ddt(n[3]):
                nodes = n[2]
                if len(nodes) == 1 and nodes[0] in self.m.branches:
                    nodes = [x for x in self.m.branches[nodes[0]] if x is not None]
                vb = self.m.find_vbranch(nodes)
                if vb is not None:
                    mask |= 1 << self.node_ix[vb[0]]
                else:
                    for nd in nodes:
                        mask |= 1 << self.node_ix[nd]
        return mask

    # ---- setup / eval split (binding-time analysis) ----
    # The reference constant-folds everything that depends only on the instance parameters when it compiles the circuit
    # (`DefaultSim` parameters are compile-time constants, src/circuitodesystem.jl:57-62).  Here the analog block is split
    # into `setup(P, env, C)`, which runs the bias-independent statements once per instance and parameter change and stores
    # what the rest needs into the per-instance constant block C, and `eval(P, C, V, ...)`, the bias-dependent remainder.
    # The analysis is flow-sensitive (BSIM code reuses its temporaries T0..T9 for both kinds of value): walking the statements
    # in order, `state[var]` is the slot of C that holds the variable's current value, or None once it depends on a node
    # voltage (by data, or by being assigned under a bias-dependent condition).  Conditions that are themselves
    # bias-independent are kept as control flow on both sides (their truth value is a slot); where the branches disagree
    # about a variable, a merge slot (both static) or a materialisation `v = C[slot]` at the end of the static branch is made.
    def _new_slot(self):
        self.n_slots += 1
        return self.n_slots - 1

    def _load(self, name, slot, ty):
        if slot < 0:   # never assigned so far: Verilog-A variables start at zero
            return ("0", "int") if ty == "int" else ("0.0", "real")
        self.used_slots.add(slot)
        return ("(int)C[@%d@]" % slot, "int") if ty == "int" else ("C[@%d@]" % slot, "real")

    def _materialise(self, name, slot, ty, pad):
        c, t = self._load(name, slot, ty)
        return "%sv_%s = %s;" % (pad, name, self.cast(c, t, ty, "R"))

    @staticmethod
    def _worth_hoisting(e):
        if e[0] == "bin":
            return e[1] in ("/", "**")
        return e[0] == "call" and not e[1].startswith("$") and e[1] not in POTENTIAL_ACCESS and e[1] not in FLOW_ACCESS and e[1] not in ("ddx", "ddt", "white_noise", "flicker_noise")

    def is_static(self, e, state, vars_):
        k = e[0]
        if k in ("num", "str"):
            return True
        if k == "id":
            if e[1] in vars_:
                return state.get(e[1]) is not None
            return e[1] in self.param_ix
        if k == "index":
            return False
        if k == "un":
            return self.is_static(e[2], state, vars_)
        if k == "bin":
            return self.is_static(e[2], state, vars_) and self.is_static(e[3], state, vars_)
        if k == "tern":
            return all(self.is_static(x, state, vars_) for x in e[1:4])
        if k == "call":
            name = e[1]
            if name in POTENTIAL_ACCESS or name in FLOW_ACCESS or name in ("ddx", "ddt", "$simparam", "$limit", "$abstime", "$realtime"):
                return False     # $simparam("gmin") changes between launches of one solve (gmin stepping)
            if name in ("$param_given", "$given", "$mfactor", "$port_connected", "white_noise", "flicker_noise"):
                return True
            if name in self.m.functions and any(kind != "input" for _, kind in self.m.functions[name].args):
                return False
            return all(self.is_static(a, state, vars_) for a in e[2] if a[0] != "str")
        return False

    def _assigned(self, st, out=None):
        """names that a statement subtree may assign (output arguments of analog functions included)"""
        out = set() if out is None else out
        for n in _walk(st):
            if not n or not isinstance(n[0], str):
                continue
            if n[0] in ("assign", "assign_idx"):
                out.add(n[1])
            elif n[0] == "call" and n[1] in self.m.functions:
                for (nm, kind), a in zip(self.m.functions[n[1]].args, n[2]):
                    if kind != "input" and a[0] == "id":
                        out.add(a[1])
        return out

    def _split(self, st, state, dyn, sctx, ectx, ind):
        """-> (setup lines, eval lines); setup lines that only store a slot are (slot, text) pairs, dropped later if unused"""
        if st is None:
            return [], []
        k = st[0]
        pad = "  " * ind
        S, E = [], []

        def ectx_here(hoist=True):
            return dict(ectx, cache=state, setup_out=(S if hoist else None), setup_ctx=sctx, pad=pad)

        def make_dynamic(names):
            for nm in sorted(names):
                if state.get(nm) is not None:
                    if nm in ectx["vars"]:
                        E.append(self._materialise(nm, state[nm], ectx["vars"][nm], pad))
                    state[nm] = None

        if k == "assign":
            name = st[1]
            if name not in ectx["vars"]:
                raise VAError("assignment to undeclared variable '%s' in module %s" % (name, self.m.name))
            outs = self._assigned(st[2])
            if name in state and not dyn and not outs and self.is_static(st[2], state, ectx["vars"]):
                c, t = self.expr(st[2], sctx)
                S.append("%sv_%s = %s;" % (pad, name, self.cast(c, t, sctx["vars"][name], "double")))
                slot = self._new_slot()
                S.append((slot, "%sC[@%d@] = (double)v_%s;" % (pad, slot, name)))
                state[name] = slot
            else:
                c, t = self.expr(st[2], ectx_here())
                E.append("%sv_%s = %s;" % (pad, name, self.cast(c, t, ectx["vars"][name], "R")))
                for nm in outs | {name}:
                    if nm in state:
                        state[nm] = None
            return S, E
        if k in ("assign_idx", "contrib"):
            outs = self._assigned(st)
            E += self.stmt(st, ectx_here(), ind)
            for nm in outs:
                if nm in state:
                    state[nm] = None
            return S, E
        if k == "case":
            sel, default, chain = st[1], None, []
            for conds, body in st[2]:
                if conds is None:
                    default = body
                    continue
                test = None
                for cd in conds:
                    t1 = ("bin", "==", sel, cd)
                    test = t1 if test is None else ("bin", "||", test, t1)
                chain.append((test, body))
            node = default
            for test, body in reversed(chain):
                node = ("if", test, body, node)
            return self._split(node, state, dyn, sctx, ectx, ind)
        if k == "if":
            cond = st[1]
            if not dyn and not self._assigned(cond) and self.is_static(cond, state, ectx["vars"]):
                cs, _ = self.expr(cond, sctx)
                kc = self._new_slot()
                S.append((kc, "%sC[@%d@] = va::truth(%s) ? 1.0 : 0.0;" % (pad, kc, cs)))
                stA, stB = dict(state), dict(state)
                SA, EA = self._split(st[2], stA, False, sctx, ectx, ind + 1)
                SB, EB = self._split(st[3], stB, False, sctx, ectx, ind + 1)
                post = []
                for nm in list(state):
                    a, b = stA.get(nm), stB.get(nm)
                    if a == b:
                        state[nm] = a
                    elif a is not None and b is not None:
                        km = self._new_slot()
                        post.append((km, "%sC[@%d@] = (double)v_%s;" % (pad, km, nm)))
                        state[nm] = km
                    else:
                        (EA if a is not None else EB).append(self._materialise(nm, a if a is not None else b, ectx["vars"][nm], pad + "  "))
                        state[nm] = None
                S += ["%sif (va::truth(%s)) {" % (pad, cs)] + SA + ["%s} else {" % pad] + SB + ["%s}" % pad] + post
                if EA or EB:
                    self.used_slots.add(kc)
                    E += ["%sif (C[@%d@] != 0.0) {" % (pad, kc)] + EA + (["%s} else {" % pad] + EB if EB else []) + ["%s}" % pad]
                return S, E
            make_dynamic(n for n in self._assigned(st) if n in state)
            c, _ = self.expr(cond, ectx_here())
            SA, EA = self._split(st[2], state, True, sctx, ectx, ind + 1)
            SB, EB = self._split(st[3], state, True, sctx, ectx, ind + 1)
            S += SA + SB
            E += ["%sif (va::truth(%s)) {" % (pad, c)] + EA + (["%s} else {" % pad] + EB if EB else []) + ["%s}" % pad]
            return S, E
        if k == "block":
            S.append("%s{" % pad)
            E.append("%s{" % pad)
            saved = {}
            if st[2]:
                sctx = dict(sctx, vars=dict(sctx["vars"]))
                ectx = dict(ectx, vars=dict(ectx["vars"]))
                for nm, ty in st[2].items():
                    t = "int" if ty == "integer" else ("dual" if nm in self.dual else "real")
                    ectx["vars"][nm] = t
                    sctx["vars"][nm] = "int" if t == "int" else "real"
                    E.append("%s  %s v_%s%s;" % (pad, {"int": "int", "real": "double", "dual": "R"}[t], nm, self._decl_suffix(nm)))
                    S.append("%s  %s v_%s%s;" % (pad, "int" if t == "int" else "double", nm, self._decl_suffix(nm)))
                    saved[nm] = state.get(nm, "absent")
                    if nm in self.m.arrays:
                        state.pop(nm, None)
                    else:
                        state[nm] = -1
            n_body = 0
            for s1 in st[3]:
                s_, e_ = self._split(s1, state, dyn, sctx, ectx, ind + 1)
                S += s_
                E += e_
                n_body += len(e_)
            if n_body == 0:
                E = ["#"]     # marker: nothing bias-dependent in this block (dropped below)
            for nm, old in saved.items():
                if old == "absent":
                    state.pop(nm, None)
                else:
                    state[nm] = old
            S.append("%s}" % pad)
            if E == ["#"]:
                E = []
            else:
                E.append("%s}" % pad)
            return S, E
        if k in ("for", "while", "repeat"):
            make_dynamic(n for n in self._assigned(st) if n in state)
            E += self.stmt(st, ectx_here(hoist=False), ind)
            return S, E
        if k == "event":
            return self._split(st[1], state, dyn, sctx, ectx, ind)
        if k in ("task", "null"):
            return S, E
        raise VAError("cannot generate statement %r" % (st,))

    def generate_split(self, param_decls, vars_):
        """lines of `setup` and of the cached `eval`; sets self.n_cache"""
        import re
        m = self.m
        self.n_slots, self.used_slots = 0, set()
        state = {nm: -1 for nm, t in vars_.items() if nm not in m.arrays}
        sctx = {"vars": {k: ("real" if t == "dual" else t) for k, t in vars_.items()}, "S": "double", "noise": True}
        ectx = {"vars": dict(vars_), "S": "R"}
        S, E = [], []
        for st in m.analog:
            s_, e_ = self._split(st, state, False, sctx, ectx, 1)
            S += s_
            E += e_
        order = sorted(self.used_slots)
        final = {k: i for i, k in enumerate(order)}
        self.n_cache = len(order)

        def finish(lines):
            out = []
            for ln in lines:
                if isinstance(ln, tuple):
                    if ln[0] not in final:
                        continue
                    ln = ln[1]
                out.append(re.sub(r"@(\d+)@", lambda mo: str(final[int(mo.group(1))]), ln))
            return out
        S, E = finish(S), finish(E)

        def var_decls(scalar):
            return ["  %s v_%s%s;" % ({"int": "int", "real": "double", "dual": scalar}[t], nm, self._decl_suffix(nm)) for nm, t in vars_.items()]
        out = ["VA_HD_NOINLINE void setup(const double* P, const va::Env& env, double* C) {", "  VA_KEEP_RETURN_ADDRESS;"]
        out += param_decls + var_decls("double") + ["  (void)env; (void)P; (void)C;"] + S + ["}"]
        # PART: -1 everything; 0 the resistive sums I[] only; 1 the charge sums Q[] only — two half-evaluations on two wavefronts
        # (the engine's function split of a compiled device: what a half does not store, the compiler drops)
        # CP: the pointer type of the parameter and constant blocks — `const double*`, or va::lds_cptr when the caller has staged them
        # into LDS (then every P[i] / C[i] is a ds_read that waits on the LDS counter alone instead of a flat load behind every scratch store)
        out.append("template <class R, int PART, class CP = const double*> VA_HD_NOINLINE void eval(CP P, CP C, const R* V, const va::Env& env, R* I, R* Q) {")
        out.append("  VA_KEEP_RETURN_ADDRESS;")
        out += param_decls + var_decls("R")
        out.append("  (void)env; (void)V; (void)P; (void)C;")
        for k in range(len(m.nodes)):   # node voltages read once, node sums accumulated in registers
            out.append("  const R n%d_ = V[%d]; R i%d_ = R(0.0), q%d_ = R(0.0); (void)n%d_;" % (k, k, k, k, k))
        for k in range(len(m.vbranches)):
            out.append("  int bs%d_ = 0; R bv%d_ = R(0.0), bq%d_ = R(0.0);" % (k, k, k))
        out += E
        for k, key in enumerate(m.vbranches):
            kb, a = self.node_ix[m.branch_node(key)], self.node_ix[key[0]]
            vab = "n%d_" % a if len(key) == 1 else "(n%d_ - n%d_)" % (a, self.node_ix[key[1]])
            out.append("  i%d_ += n%d_;%s" % (a, kb, (" i%d_ -= n%d_;" % (self.node_ix[key[1]], kb)) if len(key) > 1 else ""))
            out.append("  i%d_ += (bs%d_ == 1 ? %s : n%d_) - bv%d_; q%d_ -= bq%d_;" % (kb, k, vab, kb, k, kb, k))
        for k in range(len(m.nodes)):
            out.append("  if (PART != 1) I[%d] = i%d_;" % (k, k))
            out.append("  if (PART != 0) Q[%d] = q%d_;" % (k, k))
        out.append("}")
        return out

    # ---- top level ----
    def function(self, f):
        targs = []
        for nm, kind in f.args:
            ty = "int" if f.vars.get(nm) == "integer" else "S"
            targs.append("%s%s v_%s" % (ty, "&" if kind != "input" else "", nm))
        rt = "int" if f.rtype == "integer" else "S"
        out = ["template <class S> VA_HD %s f_%s(%s) {" % (rt, f.name, ", ".join(["const va::Env& env"] + targs))]
        argn = {nm for nm, _ in f.args}
        vars_ = {}
        for nm, ty in f.vars.items():
            t = "int" if ty == "integer" else "dual"
            vars_[nm] = t
            if nm not in argn:
                out.append("  %s v_%s%s;" % ("int" if t == "int" else "S", nm, self._decl_suffix(nm)))
        out.append("  (void)env;")
        ctx = {"vars": vars_, "S": "S", "infunc": True}
        out += self.stmt(f.body, ctx, 1)
        out.append("  return v_%s;" % f.name)
        out.append("}")
        return out

    def generate(self):
        m = self.m
        out = ["// ---- module %s: %d ports, %d internal nodes, %d parameters ----" % (m.name, len(m.ports), len(m.internal), len(m.params))]
        out.append("namespace m_%s {" % m.name)
        for f in m.functions.values():
            out += self.function(f)
        np_ = len(m.params)
        used = set()
        for n in _walk([m.analog]):
            if n and n[0] == "id":
                used.add(n[1])
            if n and n[0] == "call" and n[1] in ("$param_given", "$given"):
                used.add("?" + m.aliases.get(n[2][0][1], n[2][0][1]))
        param_decls = []
        for i, (nm, ty, _, _) in enumerate(m.params):
            if ty == "string":
                continue
            if nm in used:
                param_decls.append("  const %s p_%s = %sP[%d];" % ("int" if ty == "integer" else "double", nm, "(int)" if ty == "integer" else "", i))
            if "?" + nm in used:
                param_decls.append("  const int g_%s = P[%d] != 0.0 ? 1 : 0;" % (nm, np_ + i))
        vars_ = {}
        for nm, ty in m.vars.items():
            if ty == "string":
                continue
            vars_[nm] = "int" if ty == "integer" else ("dual" if nm in self.dual else "real")

        def var_decls(scalar):
            return ["  %s v_%s%s;" % ({"int": "int", "real": "double", "dual": scalar}[t], nm, self._decl_suffix(nm)) for nm, t in vars_.items()]
        # setup(P, env, C): the bias-independent statements; eval(P, C, V, ...): the rest (see _split)
        out += self.generate_split(param_decls, vars_)
        # noise pass: same statements over plain doubles, contributions replaced by noise records
        self.has_noise = any(n and n[0] == "call" and n[1] in ("white_noise", "flicker_noise") for n in _walk(m.analog))
        if self.has_noise:
            out.append("VA_HD_NOINLINE int noise(const double* P, const double* V, const va::Env& env, va::NoiseRec* out) {")
            out.append("  VA_KEEP_RETURN_ADDRESS;")
            out.append("  int n_ = 0;")
            out += param_decls + var_decls("double")
            out.append("  (void)env; (void)V; (void)P;")
            out += ["  const double n%d_ = V[%d]; (void)n%d_;" % (k, k, k) for k in range(len(m.nodes))]
            nctx = {"vars": {k: ("real" if t == "dual" else t) for k, t in vars_.items()}, "S": "double", "noise": True}
            self.dual_saved, self.dual = self.dual, set()
            try:
                for st in m.analog:
                    out += self.stmt(st, nctx, 1)
            finally:
                self.dual = self.dual_saved
            out.append("  return n_;")
            out.append("}")
        # operating-point pass: the analog block over plain doubles; the variables declared with a (* desc *) attribute are
        # the module's observables (src/vasim.jl:742-753, 841-843)
        self.op_names = [nm for nm in m.var_desc if vars_.get(nm) in ("real", "dual", "int")]
        if self.op_names:
            out.append("VA_HD_NOINLINE void opvars(const double* P, const double* V, const va::Env& env, double* op) {")
            out.append("  VA_KEEP_RETURN_ADDRESS;")
            out += param_decls + var_decls("double")
            out.append("  (void)env; (void)V; (void)P;")
            out += ["  const double n%d_ = V[%d]; (void)n%d_;" % (k, k, k) for k in range(len(m.nodes))]
            octx = {"vars": {k: ("real" if t == "dual" else t) for k, t in vars_.items()}, "S": "double", "noise": True, "opvars": True}
            saved, self.dual = self.dual, set()
            try:
                for st in m.analog:
                    out += self.stmt(st, octx, 1)
            finally:
                self.dual = saved
            for k, nm in enumerate(self.op_names):
                out.append("  op[%d] = (double)v_%s;" % (k, nm))
            out.append("}")
        out.append("}  // namespace m_%s" % m.name)
        return out


def generate_header(modules, source_tag=""):
    """C++ header text for a list of parsed modules (module id = position in the list)."""
    gens = [ModuleGen(m) for m in modules]
    out = ["// GENERATED by cedarsim.jl_amd/va/codegen.py — do not edit.  %s" % source_tag,
           "#pragma once", '#include "../va_rt.hpp"', "", "namespace va_gen {", ""]
    for g in gens:
        out += g.generate()
        out.append("")
    out.append("struct ModuleInfo { const char* name; int n_ports, n_nodes, n_params; unsigned q_mask; const char* const* node_names; const char* const* param_names; };")
    for g in gens:
        m = g.m
        out.append("static const char* const nodes_%s[] = {%s};" % (m.name, ", ".join('"%s"' % n for n in m.nodes) or '""'))
        out.append("static const char* const params_%s[] = {%s};" % (m.name, ", ".join('"%s"' % p[0] for p in m.params) or '""'))
    for g in gens:
        ops = getattr(g, "op_names", [])
        out.append("static const char* const opnames_%s[] = {%s};" % (g.m.name, ", ".join('"%s"' % n for n in ops) or '""'))
    out.append("static const int N_OPVARS[] = {%s};" % (", ".join(str(len(getattr(g, "op_names", []))) for g in gens) or "0"))
    out.append("static const char* const* const OPNAMES[] = {%s};" % (", ".join("opnames_%s" % g.m.name for g in gens) or "nullptr"))
    out.append("static const int N_MODULES = %d;" % len(gens))
    out.append("static const ModuleInfo MODULES[] = {")
    for g in gens:
        m = g.m
        out.append('  {"%s", %d, %d, %d, %du, nodes_%s, params_%s},' % (m.name, len(m.ports), len(m.nodes), len(m.params), g.q_mask(), m.name, m.name))
    if not gens:
        out.append('  {"", 0, 0, 0, 0u, nullptr, nullptr},')
    out.append("};")
    out.append("")
    out.append("// Seeds the duals, evaluates module `mod` and scatters into the wide stamp record")
    out.append("// st = [I(8) | Q(8) | G(8x8) | C(8x8)], every entry scaled by the multiplicity m.")
    out.append("template <int NT, class R> VA_HD void scatter(const R* I, const R* Q, double m, double* st) {")
    out.append("  for (int k = 0; k < NT; ++k) {")
    out.append("    st[k] = m * va::val(I[k]); st[8 + k] = m * va::val(Q[k]);")
    out.append("    for (int j = 0; j < NT; ++j) { st[16 + k * 8 + j] = m * va::val(I[k].d[j]); st[80 + k * 8 + j] = m * va::val(Q[k].d[j]); }")
    out.append("  }")
    out.append("}")
    out.append("static const int N_CACHE[] = {%s};" % (", ".join(str(g.n_cache) for g in gens) or "0"))
    out.append("constexpr int MAX_CACHE = %d;" % max([1] + [g.n_cache for g in gens]))
    out.append("// doubles of module `mod`'s parameter block [values | given flags] and of its constant block (device-side sizes: a kernel")
    out.append("// that stages the blocks of its instances into LDS)")
    out.append("VA_HD int param_doubles(int mod) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        out.append("    case %d: return %d;" % (i, 2 * len(g.m.params)))
    out.append("    default: return 0;")
    out.append("  }")
    out.append("}")
    out.append("VA_HD int cache_doubles(int mod) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        out.append("    case %d: return %d;" % (i, g.n_cache))
    out.append("    default: return 0;")
    out.append("  }")
    out.append("}")
    out.append("// Per-instance constants of module `mod`: C[0 .. N_CACHE[mod]) from the parameter block and the temperature")
    out.append("VA_HD_NOINLINE void setup(int mod, const double* P, const va::Env& env, double* C) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        out.append("    case %d: m_%s::setup(P, env, C); break;" % (i, g.m.name))
    out.append("    default: break;")
    out.append("  }")
    out.append("}")
    out.append("VA_HD_NOINLINE void stamp_c(int mod, const double* P, const double* C, const double* v, const va::Env& env, double m, double* st) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        mo = g.m
        nt, nd = len(mo.nodes), len(g.ddx_nodes)
        R = "va::VD<%d, double>" % nt if nd == 0 else "va::VD<%d, va::VD<%d, double>>" % (nt, nd)
        out.append("    case %d: {" % i)
        out.append("      typedef %s R;" % R)
        out.append("      R V[%d], I[%d], Q[%d];" % (nt, nt, nt))
        for k, node in enumerate(mo.nodes):
            dk = g.ddx_nodes.index(node) if node in g.ddx_nodes else -1
            out.append("      V[%d] = va::seed(v[%d], %d, %d, (R*)nullptr);" % (k, k, k, dk))
        out.append("      m_%s::eval<R, -1>(P, C, V, env, I, Q);" % mo.name)
        out.append("      scatter<%d, R>(I, Q, m, st);" % nt)
        out.append("    } break;")
    out.append("    default: break;")
    out.append("  }")
    out.append("}")
    out.append("// the same without a stored constant block (host-side callers, one-off evaluations): setup into a local block first")
    out.append("VA_HD_NOINLINE void stamp(int mod, const double* P, const double* v, const va::Env& env, double m, double* st) {")
    out.append("  double C[MAX_CACHE];")
    out.append("  setup(mod, P, env, C);")
    out.append("  stamp_c(mod, P, C, v, env, m, st);")
    out.append("}")
    out.append("")
    out.append("// Direction-parallel evaluation: one lane per (device, node j) computes the values and the j-th column of the")
    out.append("// Jacobians with one-directional duals (VD<1,·>); the lane flagged `first` also writes I and Q.")
    out.append("// part: -1 the whole record; 0 the resistive half (I, dI/dV); 1 the charge half (Q, dQ/dV) — the two halves of a device may run")
    out.append("// on different wavefronts.")
    out.append("VA_HD_NOINLINE void stamp_dir_c(int mod, const double* P, const double* C, const double* v, const va::Env& env, double m, int dir, bool first, int part, double* st) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        mo = g.m
        nt, nd = len(mo.nodes), len(g.ddx_nodes)
        R = "va::VD<1, double>" if nd == 0 else "va::VD<1, va::VD<%d, double>>" % nd
        out.append("    case %d: {" % i)
        out.append("      typedef %s R;" % R)
        out.append("      R V[%d], I[%d], Q[%d];" % (nt, nt, nt))
        for k, node in enumerate(mo.nodes):
            dk = g.ddx_nodes.index(node) if node in g.ddx_nodes else -1
            out.append("      V[%d] = va::seed1(v[%d], dir == %d, %d, (R*)nullptr);" % (k, k, k, dk))
        if len(mo.params) >= 64:
            # a large model: the engine splits it into a resistive and a charge half (ch_engine.hip, slot bits 29/30); the whole record
            # (CEDARHIP_VA_NOSPLIT, a diagnostic) is the two halves in sequence — no third instantiation of a 57 k-instruction function
            out.append("      if (part != 1) m_%s::eval<R, 0>(P, C, V, env, I, Q);" % mo.name)
            out.append("      if (part != 0) m_%s::eval<R, 1>(P, C, V, env, I, Q);" % mo.name)
        else:
            # a small model is never split: ONE evaluation (two half instantiations would run the shared front end twice)
            out.append("      m_%s::eval<R, -1>(P, C, V, env, I, Q);" % mo.name)
        out.append("      for (int k = 0; k < %d; ++k) {" % nt)
        out.append("        if (part != 1) { if (first) st[k] = m * va::val(I[k]); st[16 + k * 8 + dir] = m * va::val(I[k].d[0]); }")
        out.append("        if (part != 0) { if (first) st[8 + k] = m * va::val(Q[k]); st[80 + k * 8 + dir] = m * va::val(Q[k].d[0]); }")
        out.append("      }")
        out.append("    } break;")
    out.append("    default: break;")
    out.append("  }")
    out.append("}")
    out.append("")
    out.append("// stamp_dir_c with the parameter and constant blocks in LDS (tran_persistent_kernel stages them once per transient): the large")
    out.append("// models get an instantiation that reads them with LDS instructions, every other module goes through the generic pointers")
    out.append("VA_HD_NOINLINE void stamp_dir_lds(int mod, va::lds_cptr P, va::lds_cptr C, const double* v, const va::Env& env, double m, int dir, bool first, int part, double* st) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        mo = g.m
        if len(mo.params) < 64:
            continue
        nt, nd = len(mo.nodes), len(g.ddx_nodes)
        R = "va::VD<1, double>" if nd == 0 else "va::VD<1, va::VD<%d, double>>" % nd
        out.append("    case %d: {" % i)
        out.append("      typedef %s R;" % R)
        out.append("      R V[%d], I[%d], Q[%d];" % (nt, nt, nt))
        for k, node in enumerate(mo.nodes):
            dk = g.ddx_nodes.index(node) if node in g.ddx_nodes else -1
            out.append("      V[%d] = va::seed1(v[%d], dir == %d, %d, (R*)nullptr);" % (k, k, k, dk))
        out.append("      if (part != 1) m_%s::eval<R, 0, va::lds_cptr>(P, C, V, env, I, Q);" % mo.name)
        out.append("      if (part != 0) m_%s::eval<R, 1, va::lds_cptr>(P, C, V, env, I, Q);" % mo.name)
        out.append("      for (int k = 0; k < %d; ++k) {" % nt)
        out.append("        if (part != 1) { if (first) st[k] = m * va::val(I[k]); st[16 + k * 8 + dir] = m * va::val(I[k].d[0]); }")
        out.append("        if (part != 0) { if (first) st[8 + k] = m * va::val(Q[k]); st[80 + k * 8 + dir] = m * va::val(Q[k].d[0]); }")
        out.append("      }")
        out.append("    } break;")
    out.append("    default: stamp_dir_c(mod, (const double*)P, (const double*)C, v, env, m, dir, first, part, st); break;")
    out.append("  }")
    out.append("}")
    out.append("")
    out.append("// operating-point variables (the (* desc *) observables) of module `mod` at node voltages v")
    out.append("VA_HD_NOINLINE void opvars(int mod, const double* P, const double* v, const va::Env& env, double* op) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        if getattr(g, "op_names", []):
            out.append("    case %d: m_%s::opvars(P, v, env, op); break;" % (i, g.m.name))
    out.append("    default: break;")
    out.append("  }")
    out.append("}")
    out.append("")
    out.append("// noise sources of module `mod` at node voltages v: records (node a, node b or -1, power, flicker exponent)")
    out.append("VA_HD_NOINLINE int noise(int mod, const double* P, const double* v, const va::Env& env, va::NoiseRec* out) {")
    out.append("  switch (mod) {")
    for i, g in enumerate(gens):
        if getattr(g, "has_noise", False):
            out.append("    case %d: return m_%s::noise(P, v, env, out);" % (i, g.m.name))
    out.append("    default: return 0;")
    out.append("  }")
    out.append("}")
    out.append("")
    out.append("}  // namespace va_gen")
    return "\n".join(out) + "\n"
