"""SPICE-subset netlist → flat `Circuit`.

Covers exactly what the benchmark and test decks of the reference use: `R C L V I E G M X` lines,
`DC / PWL / PULSE / SIN` sources, `.subckt/.ends` with parameters, `.param`, `.model` (BSIM4 level
14/54, with `name.N` binning), `.include`, `.lib` (sections), `.option`, `.temp`, `.tran`, `.if/.else/
.endif`, quoted expressions, SPICE magnitudes.  Semantics follow the reference front-end:

  * magnitudes t g meg k m u mil n p f a, case-insensitive, matched at the END of the token
    (src/spectre.jl:402-415, :441-456) and multiplied in decimal so `0.22u == 0.22e-6` exactly
    (Dec64 there, `decimal.Decimal` here; test/basic.jl:626-637)
  * model binning: bins named `name.N`; bin chosen by `lmin <= scale*l < lmax && wmin <= scale*w < wmax`
    (src/spectre.jl:677, :718-722, :1162-1176); no bin → NoBinException
  * level 14/54 → BSIM4; `nmos`/`pmos` → TYPE=±1 (src/spectre.jl:589-643)
  * `m=` multiplies down the hierarchy (src/spectre.jl:942-952, src/simulate_ir.jl:43-48)
  * `.option temp/gmin/scale` and `.temp` feed SimSpec unless overridden (src/spectre.jl:1529-1544)
  * unsupported statements are ignored with a warning (src/spectre.jl:1391-1393)
"""
import decimal
import math
import os
import re
import warnings

from . import bsim4_params as B4
from .circuit import DC, PULSE, PWL, SIN, CedarError, Circuit

_MAG = {"t": "1e12", "g": "1e9", "meg": "1e6", "k": "1e3", "m": "1e-3", "u": "1e-6", "mil": "25.4e-6",
        "n": "1e-9", "p": "1e-12", "f": "1e-15", "a": "1e-18"}
_NUM_RE = re.compile(r"^([+-]?(?:\d+\.?\d*|\.\d+)(?:e[+-]?\d+)?)([a-z]*)$")


class NoBinException(CedarError):
    pass


def parse_number(tok):
    """SPICE number with magnitude suffix → float, or None if not a number."""
    m = _NUM_RE.match(tok.strip().lower())
    if not m:
        return None
    num, suf = m.group(1), m.group(2)
    sf = None
    if suf:
        if suf.startswith("meg"):
            sf = _MAG["meg"]
        elif suf.startswith("mil"):
            sf = _MAG["mil"]
        elif suf[0] in _MAG:
            sf = _MAG[suf[0]]
        # otherwise: a pure unit such as "v" / "s" / "hz" — ignored
    d = decimal.Decimal(num)
    if sf is not None:
        d *= decimal.Decimal(sf)
    return float(d)


# ---- expression evaluation (SPICE functions: src/spectre.jl source_body / test/basic.jl:651-684) ----
def _nint(x):
    return float(math.floor(x + 0.5)) if x >= 0 else float(-math.floor(-x + 0.5))


_FUNCS = {
    "sqrt": math.sqrt, "exp": math.exp, "ln": math.log, "log": math.log, "log10": math.log10, "abs": abs,
    "min": min, "max": max, "pow": math.pow, "pwr": lambda x, y: math.copysign(abs(x) ** y, x),
    "int": lambda x: float(math.trunc(x)), "nint": _nint, "floor": lambda x: float(math.floor(x)),
    "ceil": lambda x: float(math.ceil(x)), "sin": math.sin, "cos": math.cos, "tan": math.tan, "atan": math.atan,
    "sinh": math.sinh, "cosh": math.cosh, "tanh": math.tanh, "sgn": lambda x: float((x > 0) - (x < 0)),
    "pi": math.pi, "true": 1.0, "false": 0.0,
}
_TOKEN_RE = re.compile(r"\s*(?:(\d+\.?\d*(?:e[+-]?\d+)?[a-z]*|\.\d+(?:e[+-]?\d+)?[a-z]*)|([a-z_][a-z0-9_.]*)|(\*\*|&&|\|\||[<>=!]=|[-+*/^(),<>?:!]))", re.I)


def eval_expr(text, env):
    """Evaluate a SPICE expression against parameter environment `env` (dict, lowercase keys)."""
    s = text.strip().lower()
    if len(s) >= 2 and s[0] in "'{" and s[-1] in "'}":
        s = s[1:-1]
    v = parse_number(s)
    if v is not None:
        return v
    out, pos = [], 0
    while pos < len(s):
        m = _TOKEN_RE.match(s, pos)
        if not m:
            if s[pos:].strip() == "":
                break
            raise CedarError("cannot parse expression %r" % text)
        pos = m.end()
        num, ident, op = m.groups()
        if num is not None:
            out.append(repr(parse_number(num)))
        elif ident is not None:
            if ident in env:
                out.append("(%r)" % float(env[ident]))
            elif ident in _FUNCS:
                out.append("_f[%r]" % ident)
            else:
                raise CedarError("undefined parameter '%s' in expression %r" % (ident, text))
        else:
            out.append({"^": "**", "&&": " and ", "||": " or ", "!": " not ", "?": " if_ ", ":": " else_ "}.get(op, op))
    code = "".join(out)
    if " if_ " in code:  # ternary a ? b : c  →  (b if a else c), single level
        cond, rest = code.split(" if_ ", 1)
        a, b = rest.split(" else_ ", 1)
        code = "((%s) if (%s) else (%s))" % (a, cond, b)
    try:
        return float(eval(code, {"__builtins__": {}}, {"_f": _FUNCS}))
    except CedarError:
        raise
    except Exception as e:  # noqa: BLE001
        raise CedarError("error evaluating %r: %s" % (text, e))


# ---- lexical layer --------------------------------------------------------------------------------
def _logical_lines(text):
    """Join '+' continuations, drop comments; keeps the first line (title) out."""
    lines = []
    for raw in text.splitlines():
        line = raw.rstrip()
        if not line.strip():
            continue
        st = line.lstrip()
        if st.startswith("*"):
            continue
        # inline comments
        for c in (" $", "\t$", ";"):
            i = line.find(c)
            if i >= 0:
                line = line[:i]
        if st.startswith("+"):
            if lines:
                lines[-1] += " " + st[1:]
            continue
        lines.append(line.strip())
    return lines


def _tokenize(line):
    """Split a logical line into tokens; quoted expressions, {...} and (...) groups survive;
    `a = b` becomes `a=b`."""
    line = re.sub(r"\s*=\s*", "=", line)
    toks, cur, depth, quote = [], "", 0, None
    for ch in line:
        if quote:
            cur += ch
            if ch == quote:
                quote = None
            continue
        if ch == "'":
            quote = "'"
            cur += ch
        elif ch in "({":
            depth += 1
            cur += ch
        elif ch in ")}":
            depth -= 1
            cur += ch
        elif ch in " \t," and depth == 0:
            if cur:
                toks.append(cur)
                cur = ""
        else:
            cur += ch
    if cur:
        toks.append(cur)
    return toks


def _split_params(tokens):
    """Separate positional tokens from key=value tokens."""
    pos, kw = [], {}
    for t in tokens:
        if "=" in t and not t.startswith("'"):
            k, v = t.split("=", 1)
            kw[k.lower()] = v
        else:
            pos.append(t)
    return pos, kw


class Subckt:
    def __init__(self, name, ports, params):
        self.name, self.ports, self.params = name, ports, params  # params: ordered dict name → expr text
        self.body = []


class ParsedNetlist:
    """Result of `parse_spice`: statements + models; `build(**overrides)` flattens to a `Circuit`.

    Parameter overrides use the reference's dotted naming: `R1=...` for a top-level `.param`,
    `var"x1.r_load"` ≙ `"x1.r_load"` for a parameter inside instance x1 (test/sweep.jl:342-371).
    """

    def __init__(self):
        self.title = ""
        self.top = Subckt("<top>", [], {})
        self.subckts = {}
        self.models = {}        # base name → list of (full name, type, params dict)
        self.options = {}
        self.tran = None        # (tstep, tstop)
        self.warnings = []

    def add_spectre_models(self, text):
        """Register the `model` cards of a Spectre-language file (e.g. the ASAP7 `7nm_TT.scs` the reference's parser
        tests hold) so that SPICE instance lines can name them."""
        for name, (master, params) in parse_spectre_models(text).items():
            self.models.setdefault(name, []).append((name, master, params))

    def add_model_cards(self, cards):
        """Register model cards given as {name: {"master": module-or-type, "params": {param: value}}}."""
        for name, card in cards.items():
            self.models.setdefault(name.lower(), []).append((name.lower(), card["master"].lower(), dict(card["params"])))

    # -- SimSpec --
    def _spec(self, overrides):
        temp, gmin, scale = 27.0, 1e-12, 1.0
        env = {}
        if "temp" in self.options:
            temp = eval_expr(self.options["temp"], env)
        if "tnom" in self.options and "temp" not in self.options:
            pass
        if "gmin" in self.options:
            gmin = eval_expr(self.options["gmin"], env)
        if "scale" in self.options:
            scale = eval_expr(self.options["scale"], env)
        return (overrides.pop("temp", temp), overrides.pop("gmin", gmin), overrides.pop("scale", scale))

    def find_bin(self, base, l, w, scale=1.0):
        """find_bin (src/spectre.jl:1162-1176): half-open ranges on scale*l, scale*w."""
        bins = self.models[base]
        if len(bins) == 1 and "." not in bins[0][0]:
            return bins[0]
        L, W = scale * l, scale * w
        for b in bins:
            p = b[2]
            if p.get("lmin", 0.0) <= L < p.get("lmax", 1.0) and p.get("wmin", 0.0) <= W < p.get("wmax", 1.0):
                return b
        raise NoBinException("NoBinException: no bin for BinnedModel %s of size (l=%s, w=%s)." % (base, L, W))

    def build(self, **overrides):
        ov = {k.lower(): float(v) for k, v in overrides.items()}
        temp, gmin, scale = self._spec(ov)
        ckt = Circuit(temp=temp, gmin=gmin, scale=scale)
        ckt.title = self.title
        ckt._netlist = self
        self._model_ix = {}
        self._used_ov = set()
        self._expand(ckt, self.top, prefix="", node_map={}, env_outer={}, inst_params={}, mult=1.0, ov=ov)
        unused = set(ov) - self._used_ov
        if unused:
            raise CedarError("unknown parameter(s) in override: %s" % ", ".join(sorted(unused)))
        return ckt

    # -- hierarchy expansion --
    def _expand(self, ckt, sub, prefix, node_map, env_outer, inst_params, mult, ov):
        env = dict(env_outer)
        # defaults declared on the .subckt line, then instance overrides, then sweep overrides
        for k, expr in sub.params.items():
            env[k] = eval_expr(expr, env)
        for k, val in inst_params.items():
            env[k] = val
        for st in sub.body:
            if st[0] == "param":
                for k, expr in st[1].items():
                    key = prefix + k
                    if key in ov:
                        env[k] = ov[key]
                        self._used_ov.add(key)
                    elif k in inst_params and prefix:
                        env[k] = inst_params[k]
                    else:
                        env[k] = eval_expr(expr, env)
        for k in list(inst_params):
            key = prefix + k
            if key in ov:
                env[k] = ov[key]
                self._used_ov.add(key)

        def node(n):
            n = n.lower()
            if n in ("0", "gnd", "gnd!"):
                return 0
            if n in node_map:
                return node_map[n]
            return ckt.net(prefix + n)

        def val(expr):
            return eval_expr(expr, env)

        active = [True]
        for st in sub.body:
            kind = st[0]
            if kind == "if":
                active.append(active[-1] and bool(val(st[1])))
                continue
            if kind == "elseif":
                prev = active.pop()
                active.append(active[-1] and (not prev) and bool(val(st[1])))
                continue
            if kind == "else":
                prev = active.pop()
                active.append(active[-1] and not prev)
                continue
            if kind == "endif":
                active.pop()
                continue
            if not active[-1] or kind == "param":
                continue
            name, toks = st[1], st[2]
            full = prefix + name
            pos, kw = _split_params(toks)
            m_given = "m" in kw
            m = mult * (val(kw.pop("m")) if m_given else 1.0)
            c0 = name[0]
            if c0 == "r":
                a, b = node(pos[0]), node(pos[1])
                rest = pos[2:]
                if "r" in kw:
                    ckt.R(full, a, b, val(kw["r"]), m=m)
                elif "l" in kw or (rest and parse_number(rest[0]) is None and not rest[0].startswith("'") and rest[0].lower() not in env):
                    # semiconductor resistor with a model: r = rsh*(l-short)/(w-narrow) (simpledevices.jl:66-70)
                    mp = {}
                    if rest:
                        base = rest[0].lower()
                        if base in self.models:
                            mp = self.models[base][0][2]
                    if "r" in mp:  # `.model rm r R=1` (test/basic.jl:585-587)
                        ckt.R(full, a, b, mp["r"], m=m)
                        continue
                    ckt.R(full, a, b, None, m=m, rsh=mp.get("rsh", 50.0), w=val(kw["w"]) if "w" in kw else 1e-6,
                          l=val(kw["l"]) if "l" in kw else 1e-6, narrow=mp.get("narrow", 0.0), short=mp.get("short", 0.0))
                else:
                    ckt.R(full, a, b, val(rest[0]), m=m)
            elif c0 == "c":
                ckt.C(full, node(pos[0]), node(pos[1]), val(kw["c"]) if "c" in kw else val(pos[2]), m=m)
            elif c0 == "l":
                ckt.L(full, node(pos[0]), node(pos[1]), val(kw["l"]) if "l" in kw else val(pos[2]), m=m)
            elif c0 in "vi":
                dc, tran, ac = self._source(pos[2:], kw, val)
                (ckt.V if c0 == "v" else ckt.I)(full, node(pos[0]), node(pos[1]), dc=dc, tran=tran, m=m, ac=ac)
            elif c0 == "b":
                # bsource (spectre_env.jl:127-140): v= / i= / r= / c=
                a, b = node(pos[0]), node(pos[1])
                if "v" in kw:
                    ckt.V(full, a, b, tran=DC(val(kw["v"])), m=m)
                elif "i" in kw:
                    ckt.I(full, a, b, tran=DC(val(kw["i"])), m=m)
                elif "r" in kw:
                    ckt.R(full, a, b, val(kw["r"]), m=m)
                elif "c" in kw:
                    ckt.C(full, a, b, val(kw["c"]), m=m)
                else:
                    raise CedarError("BSOURCE with args %s not supported." % kw)
            elif c0 in "eg":
                add = ckt.E if c0 == "e" else ckt.G
                if len(pos) >= 5:
                    add(full, node(pos[0]), node(pos[1]), node(pos[2]), node(pos[3]), gain=val(pos[4]), m=m)
                else:  # two-terminal form: vol=/cur=/value= constant source
                    v = kw.get("vol", kw.get("cur", kw.get("value", "0")))
                    if c0 == "e":
                        ckt.V(full, node(pos[0]), node(pos[1]), dc=val(v), m=m)
                    else:
                        ckt.I(full, node(pos[0]), node(pos[1]), dc=val(v), m=m)
            elif c0 == "m":
                if self._va_model(pos[4]) is not None:
                    self._va_instance(ckt, full, [node(p) for p in pos[:4]], pos[4], kw, val, m)
                else:
                    self._mos(ckt, full, [node(p) for p in pos[:4]], pos[4], kw, val, m, scale=ckt.scale)
            elif c0 == "x":
                target = pos[-1].lower()
                nodes = pos[:-1]
                if target in self.subckts:
                    sc = self.subckts[target]
                    if len(nodes) != len(sc.ports):
                        raise CedarError("subckt %s expects %d nodes, got %d" % (target, len(sc.ports), len(nodes)))
                    nm = {p: node(n) for p, n in zip(sc.ports, nodes)}
                    if not m_given and "m" in sc.params:  # `.subckt r10 a b m=10`: default multiplicity (test/basic.jl:563)
                        m = mult * eval_expr(sc.params["m"], env)
                    ip = {k: val(v) for k, v in kw.items()}
                    self._expand(ckt, sc, full + ".", nm, env, ip, m, ov)
                elif self._va_model(target) is not None:
                    # a compiled Verilog-A module (`.hdl "file.va"`, test/basic.jl:359-381) or a model card of one
                    self._va_instance(ckt, full, [node(p) for p in nodes], target, kw, val, m)
                elif target in self.models:
                    # PDK style "X… nfet_06v0 W= L=": the model used as a 4-terminal subcircuit
                    self._mos(ckt, full, [node(p) for p in nodes[:4]], target, kw, val, m, scale=ckt.scale)
                else:
                    raise CedarError("unknown subcircuit or model '%s'" % target)
            else:
                self.warnings.append("Statement ignored: %s" % name)
                warnings.warn("Statement ignored: %s" % name)

    # level → compiled Verilog-A module (src/spectre.jl:589-630: 17/72 → bsimcmg107); TYPE/DEVTYPE from nmos/pmos (:632-643)
    _VA_LEVELS = {17: "bsimcmg", 72: "bsimcmg"}
    _VA_TYPE_PARAM = {"bsimcmg": ("devtype", {"nmos": 1, "pmos": 0})}

    def _va_model(self, name):
        """(module name, card parameters) when `name` is a compiled Verilog-A module or a `.model` card of one."""
        from .va.registry import has_module
        base = str(name).lower()
        if base in self.models:
            full, mtype, params = self.models[base][0]
            mod = None
            if has_module(mtype):
                mod = mtype
            elif mtype in ("bsimcmg107", "bsimcmg_va", "bsimcmg") and has_module("bsimcmg"):
                mod = "bsimcmg"
            elif mtype in ("nmos", "pmos") and int(params.get("level", 0)) in self._VA_LEVELS and has_module(self._VA_LEVELS[int(params["level"])]):
                mod = self._VA_LEVELS[int(params["level"])]
            if mod is None:
                return None
            card = {k: v for k, v in params.items() if k not in ("level", "version") or mod != "bsimcmg"}
            card.pop("level", None)
            if mod in self._VA_TYPE_PARAM:   # TYPE/DEVTYPE from nmos|pmos or type=n|p (src/spectre.jl:632-643)
                pn, mp = self._VA_TYPE_PARAM[mod]
                ty = card.pop("type", None)
                if mtype in ("nmos", "pmos"):
                    card.setdefault(pn, mp[mtype])
                elif ty in ("n", "p"):
                    card.setdefault(pn, mp["nmos" if ty == "n" else "pmos"])
            return mod, card
        if base not in self.subckts and has_module(base):
            return base, {}
        return None

    def _va_instance(self, ckt, full, nodes, model, kw, val, m):
        mod, card = self._va_model(model)
        params = dict(card)
        params.update({k: val(v) for k, v in kw.items()})
        ckt.VA(full, mod, nodes, params=params, m=m)

    def _mos(self, ckt, full, nodes, model, kw, val, m, scale):
        base = model.lower()
        if base not in self.models:
            raise CedarError("unknown model '%s'" % model)
        p = {k: val(v) for k, v in kw.items()}
        if "w" not in p or "l" not in p:
            raise CedarError("MOSFET %s needs w= and l=" % full)
        full_name, mtype, params = self.find_bin(base, p["l"], p["w"], scale)
        if full_name not in self._model_ix:
            self._model_ix[full_name] = ckt.add_model(full_name, mtype, params)
        ckt.M(full, nodes[0], nodes[1], nodes[2], nodes[3], self._model_ix[full_name], p["w"], p["l"],
              nf=p.get("nf"), m=m, as_=p.get("as"), ad=p.get("ad"), ps=p.get("ps"), pd=p.get("pd"))

    @staticmethod
    def _source(rest, kw, val):
        """`[DC] v` / `DC v` / `PWL(...)` / `PULSE(...)` / `SIN(...)` / `AC mag` (src/spectre.jl:1021-1062)."""
        dc, tran, ac = None, None, 0.0
        i = 0
        rest = list(rest)
        if "dc" in kw:
            dc = val(kw["dc"])
        if "ac" in kw:
            ac = val(kw["ac"])
        while i < len(rest):
            t = rest[i]
            tl = t.lower()
            if tl == "dc":
                dc = val(rest[i + 1])
                i += 2
            elif tl == "ac":  # AC mag [phase]: the phase is parsed and ignored (src/simpledevices.jl:293 "TODO phase")
                ac = val(rest[i + 1]) if i + 1 < len(rest) else 1.0
                i += 2
                while i < len(rest) and parse_number(rest[i]) is not None:
                    i += 1
            elif re.match(r"^(pwl|pulse|sin)\b", tl):
                fn = re.match(r"^(pwl|pulse|sin)", tl).group(1)
                args = tl[len(fn):].strip()
                if not args and i + 1 < len(rest):
                    i += 1
                    args = rest[i]
                args = args.strip()
                if args.startswith("("):
                    args = args[1:-1]
                vals = [val(a) for a in _tokenize(args)]
                if fn == "pwl":
                    tran = PWL(vals)
                elif fn == "pulse":
                    tran = PULSE(*vals)
                else:
                    tran = SIN(*vals)
                i += 1
            else:
                dc = val(t)
                i += 1
        return dc, tran, ac


def parse_spectre_models(text):
    """`model <name> <master> key=value ...` statements of a Spectre-language card file (`+` continuations, `//`
    comments; everything else is ignored).  Returns {name: (master, {param: value})} with lower-case keys; values
    are numbers where they parse (magnitude suffixes as in src/spectre.jl:402-415) and strings otherwise."""
    models, cur = {}, None
    for raw in text.splitlines():
        line = raw.split("//")[0].strip()
        if not line:
            continue
        if line.lower().startswith("model "):
            toks = line.split()
            cur = {}
            models[toks[1].lower()] = (toks[2].lower(), cur)
            line = " ".join(toks[3:])
        elif line.startswith("+") and cur is not None:
            line = line[1:]
        else:
            cur = None
            continue
        for k, v in re.findall(r"([A-Za-z_]\w*)\s*=\s*(\S+)", line):
            num = parse_number(v)
            cur[k.lower()] = v.lower() if num is None else num
    return models


def parse_spice(text, include_dirs=(), lib_resolver=None, _into=None, _section=None):
    """Parse SPICE text.  `lib_resolver(path) -> text or filename or None` lets the caller satisfy
    `.lib "jlpkg://GF180MCUPDK/..."` style references (the reference resolves them through Julia
    packages that are not available here; see DESIGN.md §6 substitute cards)."""
    nl = _into or ParsedNetlist()
    lines = text.splitlines()
    if _into is None and lines:
        nl.title = lines[0].lstrip("* ").strip()
        text = "\n".join(lines[1:])
    stack = [nl.top] if not hasattr(nl, "_stack") else nl._stack
    nl._stack = stack
    in_section = _section is None
    for line in _logical_lines(text):
        low = line.lower()
        toks = _tokenize(line)
        if not toks:
            continue
        head = toks[0].lower()
        if _section is not None:
            if head == ".lib" and len(toks) == 2:
                in_section = toks[1].lower() == _section
                continue
            if head == ".endl":
                in_section = False
                continue
            if not in_section:
                continue
        cur = stack[-1]
        if head == ".end":
            break
        if head in (".subckt",):
            pos, kw = _split_params(toks[2:])
            pos = [p for p in pos if p.lower() != "params:"]
            sc = Subckt(toks[1].lower(), [p.lower() for p in pos], dict(kw))
            nl.subckts[sc.name] = sc
            stack.append(sc)
        elif head == ".ends":
            if len(stack) > 1:
                stack.pop()
        elif head == ".param":
            _, kw = _split_params(toks[1:])
            cur.body.append(("param", kw))
        elif head == ".model":
            name, mtype = toks[1].lower(), toks[2].lower()
            _, kw = _split_params(toks[3:])
            params = {}
            for k, v in kw.items():
                params[k] = eval_expr(v, _global_env(nl))
            base = name.split(".")[0] if re.match(r".*\.\d+$", name) else name
            nl.models.setdefault(base, []).append((name, mtype, params))
        elif head in (".hdl", "ahdl_include"):
            # Verilog-A sources are compiled ahead of time (cedarsim.jl_amd/va/build.py); here only check that every
            # module of the named file is in the compiled library (test/basic.jl:359-381)
            path = toks[1].strip("'\"")
            from .va.build import HERE as _va_dir
            from .va.frontend import parse_va_file
            from .va.registry import has_module
            for c in [path] + [os.path.join(d, path) for d in list(include_dirs) + [os.path.join(_va_dir, "library")]]:
                if os.path.isfile(c):
                    for vm in parse_va_file(c):
                        if not has_module(vm.name):
                            raise CedarError("Verilog-A module '%s' of %s is not compiled into the model library: add the file to "
                                             "CEDARHIP_VA_SOURCES (or va/library) and rebuild" % (vm.name, path))
                    break
            else:
                raise CedarError("cannot resolve %s %r" % (head, path))
        elif head in (".include", ".inc", ".lib"):
            path = toks[1].strip("'\"")
            section = toks[2].lower() if (head == ".lib" and len(toks) > 2) else None
            content = None
            if lib_resolver is not None:
                content = lib_resolver(path)
            if content is None:
                cand = [path] + [os.path.join(d, path) for d in include_dirs]
                for c in cand:
                    if os.path.isfile(c):
                        content = c
                        break
            if content is None:
                raise CedarError("cannot resolve %s %r" % (head, path))
            if os.path.isfile(content):
                inc_dirs = list(include_dirs) + [os.path.dirname(content)]
                with open(content) as f:
                    content = f.read()
            else:
                inc_dirs = include_dirs
            if section is not None and not re.search(r"(?im)^\s*\.lib\s+%s\s*$" % re.escape(section), content):
                section = None  # library without that section: take it whole
            # included files have no title line
            parse_spice(content, inc_dirs, lib_resolver, _into=nl, _section=section)
        elif head in (".option", ".options"):
            _, kw = _split_params(toks[1:])
            nl.options.update(kw)
        elif head == ".temp":
            nl.options["temp"] = toks[1]
        elif head == ".tran":
            env = _global_env(nl)
            nl.tran = (eval_expr(toks[1], env), eval_expr(toks[2], env))
        elif head == ".if":
            cur.body.append(("if", line[3:].strip().strip("()")))
        elif head == ".elseif":
            cur.body.append(("elseif", line[7:].strip().strip("()")))
        elif head == ".else":
            cur.body.append(("else",))
        elif head == ".endif":
            cur.body.append(("endif",))
        elif head in (".global", ".endl", ".control", ".endc", ".ac", ".dc", ".op", ".print", ".plot", ".save", ".ic", ".nodeset", ".noise"):
            continue
        elif head.startswith("."):
            nl.warnings.append("Statement ignored: %s" % head)
        else:
            cur.body.append(("dev", toks[0].lower(), toks[1:]))
    return nl


def _global_env(nl):
    env = {}
    for st in nl.top.body:
        if st[0] == "param":
            for k, expr in st[1].items():
                try:
                    env[k] = eval_expr(expr, env)
                except CedarError:
                    pass
    return env


def parse_spice_file(path, include_dirs=(), lib_resolver=None):
    with open(path) as f:
        text = f.read()
    return parse_spice(text, [os.path.dirname(os.path.abspath(path))] + list(include_dirs), lib_resolver)
