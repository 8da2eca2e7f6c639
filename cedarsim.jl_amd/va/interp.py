"""Reference evaluator of a parsed Verilog-A module (pure Python, forward-mode dual numbers).

Two uses: (1) on the host at circuit-build time — parameter defaults/ranges (`resolve_params`) and the
structure probe that tells which contributions a given parameter set activates; (2) in the tests, as the
independent check of the generated C++ (codegen.py): same AST, different evaluator.  It is slow and is
never on the simulation path.

Semantics follow src/vasim.jl / src/va_env.jl of the reference: `/` is real division, integer assignment
rounds half away from zero (va_env.jl:107), `ln/sqrt/pow` return NaN outside their domain (NaNMath),
`ddx(e, V(a,b)) = (∂e/∂V(a) − ∂e/∂V(b))/2` and `ddx(e, V(a)) = ∂e/∂V(a)` (vasim.jl:392-412),
`$temperature` in kelvin (va_env.jl:123), `$param_given` (vasim.jl:339-343), functions with
output/inout arguments (vasim.jl:426-454), `case` (vasim.jl:603-626).
"""
import math

from .frontend import FLOW_ACCESS, POTENTIAL_ACCESS, VAError, _is_zero

KB, QE = 1.3806503e-23, 1.602176462e-19   # P_K, P_Q of constants.vams ($vt)


class D:
    """Dual number; components may themselves be D (nesting gives the second derivatives ddx needs)."""
    __slots__ = ("v", "d")

    def __init__(self, v, d):
        self.v, self.d = v, d

    def _lift(self, o):
        """Bring `o` to this dual's nesting depth (a shallower operand is a constant at the outer levels)."""
        ds, do = depth(self), depth(o)
        if do == ds:
            return o
        if do > ds:
            raise _Deeper()
        return _const_like(o, self)

    def __add__(self, o):
        if depth(o) > depth(self):
            return o + self
        o = self._lift(o)
        return D(self.v + o.v, [a + b for a, b in zip(self.d, o.d)])
    __radd__ = __add__

    def __sub__(self, o):
        if depth(o) > depth(self):
            return -(o - self)
        o = self._lift(o)
        return D(self.v - o.v, [a - b for a, b in zip(self.d, o.d)])

    def __rsub__(self, o):
        return self._lift(o) - self

    def __neg__(self):
        return D(-self.v, [-a for a in self.d])

    def __mul__(self, o):
        if depth(o) > depth(self):
            return o * self
        o = self._lift(o)
        return D(self.v * o.v, [a * o.v + self.v * b for a, b in zip(self.d, o.d)])
    __rmul__ = __mul__

    def __truediv__(self, o):
        if depth(o) > depth(self):
            return _const_like(self, o) / o
        o = self._lift(o)
        q = self.v / o.v
        return D(q, [(a - q * b) / o.v for a, b in zip(self.d, o.d)])

    def __rtruediv__(self, o):
        return self._lift(o) / self


class _Deeper(Exception):
    pass


def _zero_like(z):
    return D(_zero_like(z.v), [_zero_like(z.v) for _ in z.d]) if isinstance(z, D) else 0.0


def _const_like(x, t):
    """x (shallower) as a constant with the nesting structure of t."""
    inner = _const_like(x, t.v) if depth(t.v) > depth(x) else x
    return D(inner, [_zero_like(t.v) for _ in t.d])


def depth(x):
    return 1 + depth(x.v) if isinstance(x, D) else 0


def val(x):
    while isinstance(x, D):
        x = x.v
    return x


def _chain(x, f, df):
    """f(x) with derivative df(x) (both generic over nested duals)."""
    if not isinstance(x, D):
        return f(x)
    g = df(x.v)
    return D(_chain_f(x.v, f, df), [g * a for a in x.d])


def _chain_f(x, f, df):
    return _chain(x, f, df)


def _safe(fn, dom):
    def w(x):
        try:
            return fn(x) if dom(x) else math.nan
        except OverflowError:
            return math.inf
    return w


def m_exp(x):
    return _chain(x, _safe(math.exp, lambda v: True), m_exp)


def _ln_scalar(v):
    if v > 0:
        return math.log(v)
    return -math.inf if v == 0 else math.nan


def m_ln(x):
    return _chain(x, _ln_scalar, lambda v: 1.0 / v)


def m_log10(x):
    return m_ln(x) * (1.0 / math.log(10.0))


def _sqrt_scalar(v):
    return math.sqrt(v) if v >= 0 else math.nan


def m_sqrt(x):
    return _chain(x, _sqrt_scalar, lambda v: 0.5 / m_sqrt(v))


def m_sin(x):
    return _chain(x, math.sin, m_cos)


def m_cos(x):
    return _chain(x, math.cos, lambda v: -m_sin(v))


def m_tan(x):
    return _chain(x, math.tan, lambda v: 1.0 + m_tan(v) * m_tan(v))


def m_sinh(x):
    return _chain(x, _safe(math.sinh, lambda v: True), m_cosh)


def m_cosh(x):
    return _chain(x, _safe(math.cosh, lambda v: True), m_sinh)


def m_tanh(x):
    return _chain(x, math.tanh, lambda v: 1.0 - m_tanh(v) * m_tanh(v))


def m_atan(x):
    return _chain(x, math.atan, lambda v: 1.0 / (1.0 + v * v))


def m_asin(x):
    return _chain(x, _safe(math.asin, lambda v: -1 <= v <= 1), lambda v: 1.0 / m_sqrt(1.0 - v * v))


def m_acos(x):
    return _chain(x, _safe(math.acos, lambda v: -1 <= v <= 1), lambda v: -1.0 / m_sqrt(1.0 - v * v))


def m_asinh(x):
    return _chain(x, math.asinh, lambda v: 1.0 / m_sqrt(v * v + 1.0))


def m_acosh(x):
    return _chain(x, _safe(math.acosh, lambda v: v >= 1), lambda v: 1.0 / m_sqrt(v * v - 1.0))


def m_atanh(x):
    return _chain(x, _safe(math.atanh, lambda v: -1 < v < 1), lambda v: 1.0 / (1.0 - v * v))


def m_abs(x):
    return -x if val(x) < 0 else x


def m_min(a, b):
    return a if val(a) < val(b) else b


def m_max(a, b):
    return a if val(a) > val(b) else b


def m_pow(a, b):
    av, bv = val(a), val(b)
    if not isinstance(a, D) and not isinstance(b, D):
        try:
            if av < 0 and bv != int(bv):
                return math.nan
            if av == 0 and bv < 0:
                return math.inf
            return float(av) ** bv
        except OverflowError:
            return math.inf
    if not isinstance(b, D):  # constant exponent: d/da = b·a^(b-1)  (strong zero for the exponent, va_env.jl:60-70)
        if bv == 0:
            return a * 0.0 + 1.0
        return _chain(a, lambda v: m_pow(v, b), lambda v: b * m_pow(v, b - 1))
    return m_exp(b * m_ln(a)) if av > 0 else (a * 0.0 if av == 0 and bv > 0 else a * math.nan)


def m_atan2(y, x):
    if not isinstance(y, D) and not isinstance(x, D):
        return math.atan2(y, x)
    r2 = x * x + y * y
    base = math.atan2(val(y), val(x))
    # linearise around the value: d atan2 = (x dy − y dx)/r²
    if isinstance(y, D) or isinstance(x, D):
        n = len(y.d) if isinstance(y, D) else len(x.d)
        yd = y.d if isinstance(y, D) else [0.0] * n
        xd = x.d if isinstance(x, D) else [0.0] * n
        yv = y.v if isinstance(y, D) else y
        xv = x.v if isinstance(x, D) else x
        rv = xv * xv + yv * yv
        return D(m_atan2(yv, xv), [(xv * a - yv * b) / rv for a, b in zip(yd, xd)])
    return base


def m_hypot(a, b):
    return m_sqrt(a * a + b * b)


def m_floor(x):
    return float(math.floor(val(x)))


def m_ceil(x):
    return float(math.ceil(val(x)))


def m_limexp(x):
    return m_exp(x) if val(x) < 80.0 else math.exp(80.0) * (1.0 + (x - 80.0))


FUNCS1 = {"exp": m_exp, "ln": m_ln, "log": m_log10, "sqrt": m_sqrt, "sin": m_sin, "cos": m_cos, "tan": m_tan, "sinh": m_sinh,
          "cosh": m_cosh, "tanh": m_tanh, "atan": m_atan, "asin": m_asin, "acos": m_acos, "asinh": m_asinh, "acosh": m_acosh,
          "atanh": m_atanh, "abs": m_abs, "floor": m_floor, "ceil": m_ceil, "limexp": m_limexp,
          "$ln": m_ln, "$log10": m_log10, "$exp": m_exp, "$sqrt": m_sqrt, "$sin": m_sin, "$cos": m_cos, "$tan": m_tan,
          "$asin": m_asin, "$acos": m_acos, "$atan": m_atan, "$sinh": m_sinh, "$cosh": m_cosh, "$tanh": m_tanh,
          "$asinh": m_asinh, "$acosh": m_acosh, "$atanh": m_atanh, "$abs": m_abs, "$floor": m_floor, "$ceil": m_ceil, "$limexp": m_limexp}
FUNCS2 = {"pow": m_pow, "min": m_min, "max": m_max, "atan2": m_atan2, "hypot": m_hypot,
          "$pow": m_pow, "$min": m_min, "$max": m_max, "$atan2": m_atan2, "$hypot": m_hypot}


def va_round(x):
    """VA real → integer: round half away from zero (LRM 4.2.1.1, src/va_env.jl:107)."""
    x = val(x)
    if isinstance(x, int):
        return x
    if math.isnan(x) or math.isinf(x):
        raise VAError("cannot convert %r to integer" % x)
    return int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


def truth(x):
    return val(x) != 0


class _Frame:
    def __init__(self, types):
        self.types, self.vals = dict(types), {}

    def init(self, arrays):
        for nm, ty in self.types.items():
            z = 0 if ty == "integer" else 0.0
            self.vals[nm] = [z] * (arrays[nm][1] - arrays[nm][0] + 1) if nm in arrays else z


class Interp:
    """One module instance: parameters resolved, then `evaluate(V)` any number of times."""

    def __init__(self, module, params=None, temperature_c=27.0, gmin=1e-12, strict_ranges=False):
        self.m = module
        # the reference does not enforce `from` / `exclude` (make_spice_device keeps only the defaults, src/vasim.jl:702-720):
        # violations are recorded as warnings unless strict_ranges is set
        self.strict_ranges, self.warnings = strict_ranges, []
        self.temperature = temperature_c + 273.15
        self.gmin = gmin
        self.node_ix = {n: i for i, n in enumerate(module.nodes)}
        self.ddx_nodes = self._find_ddx(module)
        self.params, self.given = self.resolve_params(params or {})
        self.structure = None

    # ---- parameters ----
    def resolve_params(self, user):
        m = self.m
        canon = {nm.lower(): nm for nm, *_ in m.params}
        for a, tgt in m.aliases.items():
            canon[a.lower()] = tgt
        given, vals = {}, {}
        for k, v in user.items():
            kk = canon.get(str(k).lower())
            if kk is None:
                raise VAError("module %s has no parameter '%s'" % (m.name, k))
            given[kk] = v
        self.params, self.given = vals, given   # defaults may reference earlier parameters
        for nm, ty, default, ranges in m.params:
            v = given[nm] if nm in given else val(self.ev(default, _Frame({})))
            v = va_round(v) if ty == "integer" else (v if ty == "string" else float(v))
            vals[nm] = v
            for kind, r in ranges:
                if isinstance(r, tuple) and len(r) == 4 and not isinstance(r[0], str):
                    lo, lo_open, hi, hi_open = r
                    lo, hi = val(self.ev(lo, _Frame({}))), val(self.ev(hi, _Frame({})))
                    inside = (v > lo if lo_open else v >= lo) and (v < hi if hi_open else v <= hi)
                    if (kind == "from" and not inside) or (kind == "exclude" and inside):
                        self._range_violation("parameter %s = %r of %s is outside its allowed range" % (nm, v, m.name))
                elif kind == "exclude" and v == val(self.ev(r, _Frame({}))):
                    self._range_violation("parameter %s = %r of %s is an excluded value" % (nm, v, m.name))
        return vals, given

    def _range_violation(self, msg):
        if self.strict_ranges:
            raise VAError(msg)
        self.warnings.append(msg)

    @staticmethod
    def _find_ddx(module):
        order = []

        def walk(n):
            if isinstance(n, (list, tuple)):
                if len(n) >= 3 and n[0] == "call" and n[1] == "ddx":
                    for a in n[2][1][2]:
                        if a[1] not in order:
                            order.append(a[1])
                for c in n:
                    walk(c)
            elif isinstance(n, dict):
                for c in n.values():
                    walk(c)
        walk(module.analog)
        for f in module.functions.values():
            walk(f.body)
        return order

    # ---- evaluation ----
    def evaluate(self, volts):
        """volts: {node: V} (missing → 0).  Returns (I, Q, G, C) with G[a][b] = ∂I_a/∂V_b over module.nodes order."""
        m = self.m
        n = len(m.nodes)
        nd = len(self.ddx_nodes)
        self.V = {}
        for i, node in enumerate(m.nodes):
            v = float(volts.get(node, 0.0))
            if nd:   # inner dual: ddx partials; outer dual: Jacobian
                inner = D(v, [1.0 if self.ddx_nodes[k] == node else 0.0 for k in range(nd)])
                zero_in = D(0.0, [0.0] * nd)
                one_in = D(1.0, [0.0] * nd)
                self.V[node] = D(inner, [one_in if j == i else zero_in for j in range(n)])
            else:
                self.V[node] = D(v, [1.0 if j == i else 0.0 for j in range(n)])
        self.nn, self.nd = n, nd
        self.Ires = [0.0] * n
        self.Qres = [0.0] * n
        self.contribs = []   # (access, nodes, kind) executed, for the structure probe
        self.noise = []
        fr = _Frame(m.vars)
        fr.init(m.arrays)
        # voltage / switch branches (src/vasim.jl:128-180): the branch current x_br is an unknown; the branch carries a state
        # (CURRENT at the start of every evaluation) and a value; a contribution of the other kind resets the value and
        # switches the state; at the end the branch row is `x_br − value` (CURRENT) or `V(a,b) − value` (VOLTAGE)
        self.bstate = {key: ["I", 0.0, None] for key in m.vbranches}
        for st in m.analog:
            self.ex(st, fr)
        for key, (state, value, charge) in self.bstate.items():
            kb = self.node_ix[m.branch_node(key)]
            xb = self.V[m.branch_node(key)]
            a = self.node_ix[key[0]]
            self.Ires[a] = self.Ires[a] + xb
            if len(key) > 1:
                self.Ires[self.node_ix[key[1]]] = self.Ires[self.node_ix[key[1]]] - xb
            lhs = xb if state == "I" else (self.V[key[0]] - (self.V[key[1]] if len(key) > 1 else 0.0))
            self.Ires[kb] = self.Ires[kb] + lhs - value
            if charge is not None:
                self.Qres[kb] = self.Qres[kb] - charge
        self.structure = list(self.contribs)
        self.opvars = {nm: val(fr.vals[nm]) for nm in m.var_desc if nm in fr.vals}   # (* desc *) observables

        def flat(x):
            x = x if isinstance(x, D) else D(x, [0.0] * n)
            return val(x.v), [val(d) for d in x.d]
        I, G, Q, C = [], [], [], []
        for k in range(n):
            v, d = flat(self.Ires[k])
            I.append(v)
            G.append(d)
            v, d = flat(self.Qres[k])
            Q.append(v)
            C.append(d)
        return I, Q, G, C

    def _zero(self):
        return 0.0

    def lookup(self, name, fr):
        f = fr
        while f is not None:
            if name in f.vals:
                return f.vals[name]
            f = getattr(f, "parent", None)
        if name in self.params:
            return self.params[name]
        raise VAError("undefined identifier '%s' in module %s" % (name, self.m.name))

    def assign(self, name, value, fr):
        f = fr
        while f is not None:
            if name in f.types:
                f.vals[name] = va_round(value) if f.types[name] == "integer" else value
                return
            f = getattr(f, "parent", None)
        raise VAError("assignment to undeclared variable '%s'" % name)

    def probe(self, acc, nodes):
        if acc in POTENTIAL_ACCESS:
            if len(nodes) == 1 and nodes[0] in self.m.branches:
                nodes = [x for x in self.m.branches[nodes[0]] if x is not None]
            a = self.V[nodes[0]]
            return a - self.V[nodes[1]] if len(nodes) > 1 else a
        if len(nodes) == 1 and nodes[0] in self.m.branches:
            nodes = [x for x in self.m.branches[nodes[0]] if x is not None]
        vb = self.m.find_vbranch(nodes)
        if vb is not None:   # current of a voltage branch = its branch unknown
            return self.V[vb[0]] * vb[1]
        raise VAError("flow probe %s(%s): only the current of a voltage branch can be probed" % (acc, ",".join(nodes)))

    def ev(self, e, fr):
        k = e[0]
        if k == "num":
            return e[1]
        if k == "id":
            return self.lookup(e[1], fr)
        if k == "str":
            return e[1]
        if k == "index":
            arr = self.lookup(e[1], fr)
            lo, hi = self.m.arrays[e[1]]
            i = va_round(self.ev(e[2], fr))
            if not lo <= i <= hi:
                raise VAError("index %d outside %s[%d:%d]" % (i, e[1], lo, hi))
            return arr[i - lo]
        if k == "un":
            x = self.ev(e[2], fr)
            if e[1] == "-":
                return -x
            if e[1] == "!":
                return 0 if truth(x) else 1
            if e[1] == "~":
                return ~va_round(x)
        if k == "bin":
            op = e[1]
            a, b = self.ev(e[2], fr), self.ev(e[3], fr)
            if op == "+":
                return a + b
            if op == "-":
                return a - b
            if op == "*":
                return a * b
            if op == "/":
                if not isinstance(a, D) and not isinstance(b, D):
                    if b == 0:
                        return math.nan if a == 0 else math.copysign(math.inf, a) * (1 if math.copysign(1, b) > 0 else -1)
                    return a / b
                return a / b
            if op == "**":
                return m_pow(a, b)
            if op == "%":
                return math.fmod(val(a), val(b))
            av, bv = val(a), val(b)
            if op == "<":
                return int(av < bv)
            if op == "<=":
                return int(av <= bv)
            if op == ">":
                return int(av > bv)
            if op == ">=":
                return int(av >= bv)
            if op == "==":
                return int(av == bv)
            if op == "!=":
                return int(av != bv)
            if op == "&&":
                return int(truth(a) and truth(b))
            if op == "||":
                return int(truth(a) or truth(b))
            ia, ib = va_round(a), va_round(b)
            return {"&": ia & ib, "|": ia | ib, "^": ia ^ ib, "<<": ia << ib, ">>": ia >> ib}[op]
        if k == "tern":
            return self.ev(e[2], fr) if truth(self.ev(e[1], fr)) else self.ev(e[3], fr)
        if k == "call":
            return self.call(e[1], e[2], fr)
        raise VAError("cannot evaluate %r" % (e,))

    def call(self, name, args, fr):
        if name in POTENTIAL_ACCESS or name in FLOW_ACCESS:
            return self.probe(name, [a[1] for a in args])
        if name == "$temperature":
            return self.temperature
        if name == "$vt":
            t = self.ev(args[0], fr) if args else self.temperature
            return t * (KB / QE)
        if name in ("$param_given", "$given"):
            pn = args[0][1]
            return int(self.m.aliases.get(pn, pn) in self.given)
        if name == "$simparam":
            key = args[0][1]
            if key == "gmin":
                return self.gmin
            if len(args) > 1:
                return self.ev(args[1], fr)
            raise VAError("$simparam(\"%s\") has no value" % key)
        if name in ("$mfactor", "$port_connected"):
            return 1.0 if name == "$mfactor" else 1
        if name in ("$abstime", "$realtime"):
            return 0.0
        if name == "$limit":
            return self.ev(args[0], fr)
        if name == "ddt":
            raise VAError("ddt() is only supported as an additive term of a contribution")
        if name == "ddx":
            x = self.ev(args[0], fr)
            nodes = [a[1] for a in args[1][2]]
            ix = [self.ddx_nodes.index(nd) for nd in nodes]

            def part(z, j):   # z: inner dual (or plain) → its j-th partial
                return z.d[j] if isinstance(z, D) else 0.0

            def ddx_inner(z):
                if len(ix) == 1:
                    return part(z, ix[0])
                return (part(z, ix[0]) - part(z, ix[1])) / 2
            if isinstance(x, D) and depth(x) == 2:
                return D(D(ddx_inner(x.v), [0.0] * self.nd), [D(ddx_inner(dd), [0.0] * self.nd) for dd in x.d])
            return 0.0
        if name in ("white_noise", "flicker_noise"):
            self.noise.append((name, [val(self.ev(a, fr)) for a in args[:-1]], args[-1][1] if args[-1][0] == "str" else None))
            return 0.0
        if name in FUNCS1:
            return FUNCS1[name](self.ev(args[0], fr))
        if name in FUNCS2:
            return FUNCS2[name](self.ev(args[0], fr), self.ev(args[1], fr))
        if name in self.m.functions:
            return self.call_function(self.m.functions[name], args, fr)
        raise VAError("unknown function '%s'" % name)

    def call_function(self, f, args, fr):
        if len(args) != len(f.args):
            raise VAError("function %s expects %d arguments" % (f.name, len(f.args)))
        loc = _Frame(f.vars)
        loc.init(self.m.arrays)
        for (nm, kind), a in zip(f.args, args):
            if kind in ("input", "inout"):
                self.assign(nm, self.ev(a, fr), loc)
        self.ex(f.body, loc)
        for (nm, kind), a in zip(f.args, args):
            if kind in ("output", "inout"):
                if a[0] != "id":
                    raise VAError("output argument of %s must be a variable" % f.name)
                self.assign(a[1], loc.vals[nm], fr)
        return loc.vals[f.name]

    def split_ddt(self, e, fr):
        """(resistive value, reactive value or None) of a contribution's right-hand side."""
        k = e[0]
        if k == "call" and e[1] == "ddt":
            return 0.0, self.ev(e[2][0], fr)
        if k == "bin" and e[1] in ("+", "-"):
            ar, aq = self.split_ddt(e[2], fr)
            br, bq = self.split_ddt(e[3], fr)
            sgn = 1.0 if e[1] == "+" else -1.0
            q = None if aq is None and bq is None else (aq if aq is not None else 0.0) + sgn * (bq if bq is not None else 0.0)
            return ar + sgn * br, q
        if k == "un" and e[1] == "-":
            r, q = self.split_ddt(e[2], fr)
            return -r, (None if q is None else -q)
        if k == "bin" and e[1] == "*":
            if _has_ddt(e[2]) and not _has_ddt(e[3]):
                r, q = self.split_ddt(e[2], fr)
                f = self.ev(e[3], fr)
                return r * f, q * f
            if _has_ddt(e[3]) and not _has_ddt(e[2]):
                r, q = self.split_ddt(e[3], fr)
                f = self.ev(e[2], fr)
                return f * r, f * q
        if k == "bin" and e[1] == "/" and _has_ddt(e[2]) and not _has_ddt(e[3]):
            r, q = self.split_ddt(e[2], fr)
            f = self.ev(e[3], fr)
            return r / f, q / f
        if k == "tern" and (_has_ddt(e[2]) or _has_ddt(e[3])):
            return self.split_ddt(e[2] if truth(self.ev(e[1], fr)) else e[3], fr)
        if _has_ddt(e):
            raise VAError("ddt() must appear as an additive (possibly scaled) term of a contribution")
        return self.ev(e, fr), None

    def _branch_contrib(self, nodes, kind, rhs, fr):
        name, sgn = self.m.find_vbranch(nodes)
        key = next(k for k in self.m.vbranches if self.m.branch_node(k) == name)
        rec = self.bstate[key]
        if rec[0] != kind:
            rec[0], rec[1], rec[2] = kind, 0.0, None
        r, q = self.split_ddt(rhs, fr)
        rec[1] = rec[1] + sgn * r
        if q is not None:
            rec[2] = (rec[2] if rec[2] is not None else 0.0) + sgn * q

    def ex(self, st, fr):
        k = st[0]
        if k == "assign_idx":
            lo, hi = self.m.arrays[st[1]]
            i = va_round(self.ev(st[2], fr))
            if not lo <= i <= hi:
                raise VAError("index %d outside %s[%d:%d]" % (i, st[1], lo, hi))
            f = fr
            while f is not None and st[1] not in f.types:
                f = getattr(f, "parent", None)
            if f is None:
                raise VAError("assignment to undeclared array '%s'" % st[1])
            v = self.ev(st[3], fr)
            f.vals[st[1]][i - lo] = va_round(v) if f.types[st[1]] == "integer" else v
        elif k == "assign":
            self.assign(st[1], self.ev(st[2], fr), fr)
        elif k == "contrib":
            acc, nodes = st[1], st[2]
            if len(nodes) == 1 and nodes[0] in self.m.branches:
                nodes = [x for x in self.m.branches[nodes[0]] if x is not None]
            if acc in POTENTIAL_ACCESS:
                if _is_zero(st[3]):
                    self.contribs.append(("V", tuple(nodes), "collapse"))
                    return
                self._branch_contrib(nodes, "V", st[3], fr)
                self.contribs.append(("V", tuple(nodes), "branch"))
                return
            if st[3][0] == "call" and st[3][1] in ("white_noise", "flicker_noise"):
                args = st[3][2]
                nargs = args[:-1] if (args and args[-1][0] == "str") else args
                vals = [val(self.ev(a, fr)) for a in nargs]
                self.noise.append({"kind": st[3][1], "nodes": tuple(nodes), "pwr": vals[0], "exp": vals[1] if st[3][1] == "flicker_noise" else 0.0,
                                   "name": args[-1][1] if (args and args[-1][0] == "str") else ""})
                return
            if self.m.find_vbranch(nodes) is not None:   # current contribution to a voltage / switch branch
                self._branch_contrib(nodes, "I", st[3], fr)
                self.contribs.append(("I", tuple(nodes), "branch"))
                return
            r, q = self.split_ddt(st[3], fr)
            a = self.node_ix[nodes[0]]
            b = self.node_ix[nodes[1]] if len(nodes) > 1 else None
            self.Ires[a] = self.Ires[a] + r
            if b is not None:
                self.Ires[b] = self.Ires[b] - r
            if q is not None:
                self.Qres[a] = self.Qres[a] + q
                if b is not None:
                    self.Qres[b] = self.Qres[b] - q
            self.contribs.append(("I", tuple(nodes), "q" if q is not None else "i"))
        elif k == "if":
            if truth(self.ev(st[1], fr)):
                self.ex(st[2], fr)
            elif st[3] is not None:
                self.ex(st[3], fr)
        elif k == "block":
            if st[2]:
                inner = _Frame(st[2])
                inner.parent = fr
                inner.init(self.m.arrays)
                fr = inner
            for s in st[3]:
                self.ex(s, fr)
        elif k == "case":
            x = val(self.ev(st[1], fr))
            default = None
            for conds, body in st[2]:
                if conds is None:
                    default = body
                elif any(val(self.ev(c, fr)) == x for c in conds):
                    self.ex(body, fr)
                    return
            if default is not None:
                self.ex(default, fr)
        elif k == "for":
            self.ex(st[1], fr)
            guard = 0
            while truth(self.ev(st[2], fr)):
                self.ex(st[4], fr)
                self.ex(st[3], fr)
                guard += 1
                if guard > 100000:
                    raise VAError("for loop does not terminate")
        elif k == "while":
            guard = 0
            while truth(self.ev(st[1], fr)):
                self.ex(st[2], fr)
                guard += 1
                if guard > 100000:
                    raise VAError("while loop does not terminate")
        elif k == "repeat":
            for _ in range(va_round(self.ev(st[1], fr))):
                self.ex(st[2], fr)
        elif k == "event":
            self.ex(st[1], fr)
        elif k == "task":
            if st[1] in ("$finish", "$stop", "$fatal", "$error"):
                raise VAError("%s called by module %s" % (st[1], self.m.name))
        elif k == "null":
            pass
        else:
            raise VAError("cannot execute %r" % (st,))


def _has_ddt(e):
    if isinstance(e, tuple):
        if len(e) >= 2 and e[0] == "call" and e[1] == "ddt":
            return True
        return any(_has_ddt(c) for c in e)
    if isinstance(e, list):
        return any(_has_ddt(c) for c in e)
    return False
