"""Verilog-A front-end: preprocessor, lexer and parser for the compact-model subset the reference
compiles (src/vasim.jl; its parser is the un-vendored-in-spirit VerilogAParser.jl package, whose
behaviour is restated here from the Verilog-AMS LRM 2.4 and from how src/vasim.jl consumes the tree).

Covered: `include / `define (with arguments) / `undef / `ifdef / `ifndef / `else / `endif; modules with
ports, disciplines, internal nets, named branches, `parameter real|integer` with ranges, `aliasparam`,
module-level and block-level real/integer variables, analog functions (input/output/inout arguments),
the analog block with begin/end, if/else, case, for/while/repeat, event-controlled statements,
assignments, contributions `I(a,b) <+`, `V(a,b) <+`, system tasks, and the full expression grammar
(literals with scale factors src/vasim.jl:100-126, `**`, ternary, logical/bitwise/relational operators).

AST nodes are plain tuples:
  expr: ("num", value, is_int) ("str", s) ("id", name) ("call", name, [args]) ("un", op, e)
        ("bin", op, a, b) ("tern", c, a, b)
  stmt: ("assign", name, expr) ("contrib", access, [nodes], expr) ("if", c, then, else|None)
        ("case", e, [([conds]|None, stmt)]) ("block", name|None, {var: type}, [stmts])
        ("for", init, cond, update, body) ("while", cond, body) ("repeat", n, body)
        ("task", name, [args]) ("event", stmt) ("null",)
"""
import os
import re

SCALE = {"T": 1e12, "G": 1e9, "M": 1e6, "K": 1e3, "k": 1e3, "m": 1e-3, "u": 1e-6, "n": 1e-9, "p": 1e-12, "f": 1e-15, "a": 1e-18}

# `include "constants.vams"` — values of LRM 2.4 annex D (mathematical and physical constants)
CONSTANTS_VAMS = """
`define M_E 2.7182818284590452354
`define M_LOG2E 1.4426950408889634074
`define M_LOG10E 0.43429448190325182765
`define M_LN2 0.69314718055994530942
`define M_LN10 2.30258509299404568402
`define M_PI 3.14159265358979323846
`define M_TWO_PI 6.28318530717958647693
`define M_PI_2 1.57079632679489661923
`define M_PI_4 0.78539816339744830962
`define M_1_PI 0.31830988618379067154
`define M_2_PI 0.63661977236758134308
`define M_2_SQRTPI 1.12837916709551257390
`define M_SQRT2 1.41421356237309504880
`define M_SQRT1_2 0.70710678118654752440
`define P_Q 1.602176462e-19
`define P_C 2.99792458e8
`define P_K 1.3806503e-23
`define P_H 6.62606876e-34
`define P_EPS0 8.854187817e-12
`define P_U0 (4.0e-7 * `M_PI)
`define P_CELSIUS0 273.15
"""
BUILTIN_INCLUDES = {"constants.vams": CONSTANTS_VAMS, "constants.h": CONSTANTS_VAMS, "disciplines.vams": "", "discipline.h": "", "disciplines.h": ""}
DISCIPLINES = {"electrical", "thermal", "voltage", "current", "magnetic", "kinematic", "rotational", "kinematic_v", "rotational_omega"}
FLOW_ACCESS = {"I", "Pwr"}
POTENTIAL_ACCESS = {"V", "Temp"}


class VAError(Exception):
    pass


def _is_zero(e):
    return e[0] == "num" and e[1] == 0


# ------------------------------------------------------------------------------------------------
# preprocessor
def _strip_comments(text):
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            if j < 0:
                raise VAError("unterminated /* comment")
            out.append(" " + "\n" * text.count("\n", i, j))
            i = j + 2
        else:
            out.append(c)
            i += 1
    return "".join(out)


_ID = r"[A-Za-z_][A-Za-z0-9_$]*"


class Preprocessor:
    def __init__(self, include_dirs=(), defines=None, reader=None):
        self.include_dirs = list(include_dirs)
        self.macros = {}       # name -> (params or None, body)
        self.reader = reader   # optional callable(path) -> text or None
        for k, v in (defines or {}).items():
            self.macros[k] = (None, str(v))

    def _read(self, name, cur_dir):
        base = os.path.basename(name)
        for d in [cur_dir] + self.include_dirs:
            if d is None:
                continue
            p = os.path.join(d, name)
            if os.path.isfile(p):
                with open(p) as f:
                    return f.read(), os.path.dirname(p)
        if self.reader is not None:
            t = self.reader(name)
            if t is not None:
                return t, cur_dir
        if base in BUILTIN_INCLUDES:
            return BUILTIN_INCLUDES[base], cur_dir
        raise VAError("include file not found: %s" % name)

    def process(self, text, cur_dir=None):
        text = _strip_comments(text)
        # join continuation lines (macro bodies)
        lines, buf = [], ""
        for ln in text.split("\n"):
            if ln.rstrip().endswith("\\"):
                buf += ln.rstrip()[:-1] + " "
            else:
                lines.append(buf + ln)
                buf = ""
        if buf:
            lines.append(buf)
        out, stack = [], []   # stack of [active_parent, taken, active]
        for ln in lines:
            s = ln.strip()
            active = all(e[2] for e in stack)
            m = re.match(r"`(ifdef|ifndef|else|elsif|endif|define|undef|include)\b(.*)", s)
            if m:
                d, rest = m.group(1), m.group(2).strip()
                if d in ("ifdef", "ifndef"):
                    name = rest.split()[0]
                    cond = (name in self.macros) == (d == "ifdef")
                    stack.append([active, cond, active and cond])
                elif d == "elsif":
                    e = stack[-1]
                    cond = rest.split()[0] in self.macros
                    e[2] = e[0] and not e[1] and cond
                    e[1] = e[1] or cond
                elif d == "else":
                    e = stack[-1]
                    e[2] = e[0] and not e[1]
                    e[1] = True
                elif d == "endif":
                    stack.pop()
                elif not active:
                    pass
                elif d == "define":
                    mm = re.match(r"(%s)(\(([^)]*)\))?\s*(.*)" % _ID, rest)
                    if not mm:
                        raise VAError("bad `define: " + s)
                    # a parameter list only counts when '(' follows the name immediately
                    name = mm.group(1)
                    if mm.group(2) is not None and rest[len(name):len(name) + 1] == "(":
                        params = [p.strip() for p in mm.group(3).split(",")] if mm.group(3).strip() else []
                        body = mm.group(4)
                    else:
                        params, body = None, rest[len(name):].strip()
                    self.macros[name] = (params, body)
                elif d == "undef":
                    self.macros.pop(rest.split()[0], None)
                elif d == "include":
                    mm = re.match(r'"([^"]+)"', rest)
                    if not mm:
                        raise VAError("bad `include: " + s)
                    t, dd = self._read(mm.group(1), cur_dir)
                    out.append(self.process(t, dd))
                continue
            if active:
                out.append(self.expand(ln))
        if stack:
            raise VAError("unterminated `ifdef")
        return "\n".join(out)

    def expand(self, s, depth=0):
        if "`" not in s:
            return s
        if depth > 64:
            raise VAError("recursive macro expansion")
        out, i, n = [], 0, len(s)
        while i < n:
            if s[i] == '"':
                j = i + 1
                while j < n and s[j] != '"':
                    j += 2 if s[j] == "\\" else 1
                j = min(j, n - 1)
                out.append(s[i:j + 1])
                i = j + 1
                continue
            if s[i] != "`":
                out.append(s[i])
                i += 1
                continue
            m = re.match(_ID, s[i + 1:])
            if not m:
                raise VAError("stray ` in: " + s)
            name = m.group(0)
            i += 1 + len(name)
            if name not in self.macros:
                raise VAError("undefined macro `%s" % name)
            params, body = self.macros[name]
            if params is not None:
                j = i
                while j < n and s[j].isspace():
                    j += 1
                if j >= n or s[j] != "(":
                    raise VAError("macro `%s needs arguments" % name)
                depth_p, k, args, cur = 0, j, [], ""
                while k < n:
                    c = s[k]
                    if c == "(":
                        depth_p += 1
                        if depth_p > 1:
                            cur += c
                    elif c == ")":
                        depth_p -= 1
                        if depth_p == 0:
                            args.append(cur)
                            break
                        cur += c
                    elif c == "," and depth_p == 1:
                        args.append(cur)
                        cur = ""
                    else:
                        cur += c
                    k += 1
                if depth_p != 0:
                    raise VAError("unterminated arguments of macro `%s" % name)
                i = k + 1
                if len(args) != len(params) and not (len(params) == 0 and args == [""]):
                    raise VAError("macro `%s expects %d arguments, got %d" % (name, len(params), len(args)))
                for p, a in zip(params, args):
                    body = re.sub(r"(?<![A-Za-z0-9_$`])%s(?![A-Za-z0-9_$])" % re.escape(p), lambda _m, a=a: a.strip(), body)
            out.append(self.expand(body, depth + 1))
        return "".join(out)


# ------------------------------------------------------------------------------------------------
# lexer
_TOKEN = re.compile(r"""
  (?P<ws>\s+)
 |(?P<attr>\(\*(?!\s*\)).*?\*\))
 |(?P<num>(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[TGMKkmunpfa]?(?![A-Za-z0-9_]))
 |(?P<sys>\$[A-Za-z_][A-Za-z0-9_$]*)
 |(?P<id>[A-Za-z_][A-Za-z0-9_$]*|\\[^\s]+)
 |(?P<str>"(?:\\.|[^"\\])*")
 |(?P<op><\+|\*\*|&&|\|\||==|!=|<=|>=|<<|>>|[-+*/%<>!~&|^?:;,.()\[\]{}=@\#'])
""", re.X | re.S)


def tokenize(text):
    toks, i, n, line = [], 0, len(text), 1
    while i < n:
        m = _TOKEN.match(text, i)
        if not m:
            raise VAError("line %d: unexpected character %r" % (line, text[i]))
        k = m.lastgroup
        v = m.group(k)
        if k == "attr":
            toks.append(("attr", v, line))   # (* name = value, ... *): kept for `desc` on variable declarations
        elif k != "ws":
            toks.append((k, v, line))
        line += v.count("\n")
        i = m.end()
    toks.append(("eof", "", line))
    return toks


def parse_number(txt):
    sf = 1.0
    if txt[-1] in SCALE and not txt[-1].isdigit():
        sf = SCALE[txt[-1]]
        txt = txt[:-1]
    if re.fullmatch(r"\d+", txt) and sf == 1.0:
        return ("num", int(txt), True)
    return ("num", float(txt) * sf, False)


# ------------------------------------------------------------------------------------------------
class Function:
    def __init__(self, name, rtype):
        self.name, self.rtype = name, rtype
        self.args = []        # [(name, 'input'|'output'|'inout')]
        self.vars = {}        # name -> 'real'|'integer' (arguments and locals)
        self.body = None


class Module:
    def __init__(self, name):
        self.name = name
        self.ports = []
        self.internal = []    # internal nets in declaration order
        self.params = []      # [(name, type, default_expr, ranges)]
        self.aliases = {}
        self.vars = {}        # name -> type
        self.var_desc = {}    # observable variables: name -> text of the (* desc = "..." *) attribute on their declaration
        self.functions = {}
        self.branches = {}    # name -> (a, b|None)
        self.arrays = {}      # array variables: name -> (lo, hi)
        self.analog = []      # analog statements in order (`analog initial` blocks first)
        self.n_initial = 0

    @property
    def vbranches(self):
        """Voltage branches: (a, b) pairs that receive a `V(a,b) <+ expr` contribution other than the literal 0 (which is a
        node collapse).  Each gets a branch-current unknown, carried as a pseudo-node after the internal nets."""
        if getattr(self, "_vb", None) is None:
            out = []

            def walk(st):
                if isinstance(st, tuple):
                    if st and st[0] == "contrib" and st[1] in POTENTIAL_ACCESS and not _is_zero(st[3]):
                        nodes = list(st[2])
                        if len(nodes) == 1 and nodes[0] in self.branches:
                            nodes = [x for x in self.branches[nodes[0]] if x is not None]
                        key = tuple(nodes)
                        if key not in out and tuple(reversed(key)) not in out:
                            out.append(key)
                    for c in st:
                        walk(c)
                elif isinstance(st, list):
                    for c in st:
                        walk(c)
            walk(self.analog)
            self._vb = out
        return self._vb

    @staticmethod
    def branch_node(key):
        return "I(" + ",".join(key) + ")"

    @property
    def nodes(self):
        return self.ports + self.internal + [self.branch_node(k) for k in self.vbranches]

    def find_vbranch(self, nodes):
        """(pseudo-node name, sign) of the voltage branch between `nodes`, or None."""
        key = tuple(nodes)
        for k in self.vbranches:
            if k == key:
                return self.branch_node(k), 1.0
            if tuple(reversed(k)) == key and len(k) == 2:
                return self.branch_node(k), -1.0
        return None


_BINPREC = [("||",), ("&&",), ("|",), ("^",), ("&",), ("==", "!="), ("<", "<=", ">", ">="), ("<<", ">>"), ("+", "-"), ("*", "/", "%"), ("**",)]


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0
        self.arrays = {}   # name -> (lo, hi) of every array variable declared so far (module, block or function level)

    # -- token helpers --
    def peek(self, k=0):
        j, n = self.i, 0
        while True:   # attribute instances are transparent to the grammar
            while self.t[j][0] == "attr":
                j += 1
            if n == k:
                return self.t[j]
            j += 1
            n += 1

    def _skip_attrs(self):
        last = None
        while self.t[self.i][0] == "attr":
            last = self.t[self.i][1]
            self.i += 1
        return last

    def at(self, v):
        tok = self.peek()
        return tok[1] == v and tok[0] in ("op", "id")

    def eat(self, v=None):
        self._skip_attrs()
        tok = self.t[self.i]
        if v is not None and tok[1] != v:
            raise VAError("line %d: expected %r, found %r" % (tok[2], v, tok[1]))
        self.i += 1
        return tok

    def ident(self):
        self._skip_attrs()
        tok = self.t[self.i]
        if tok[0] != "id":
            raise VAError("line %d: expected an identifier, found %r" % (tok[2], tok[1]))
        self.i += 1
        return tok[1][1:] if tok[1].startswith("\\") else tok[1]

    # -- expressions --
    def expr(self):
        c = self.binary(0)
        if self.at("?"):
            self.eat()
            a = self.expr()
            self.eat(":")
            b = self.expr()
            return ("tern", c, a, b)
        return c

    def binary(self, level):
        if level == len(_BINPREC):
            return self.unary()
        ops = _BINPREC[level]
        if ops == ("**",):  # right associative
            a = self.unary()
            if self.peek()[0] == "op" and self.peek()[1] == "**":
                self.eat()
                return ("bin", "**", a, self.binary(level))
            return a
        a = self.binary(level + 1)
        while self.peek()[0] == "op" and self.peek()[1] in ops:
            op = self.eat()[1]
            a = ("bin", op, a, self.binary(level + 1))
        return a

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("-", "+", "!", "~"):
            self.eat()
            e = self.unary()
            # `-a ** b` binds as -(a ** b)
            if self.peek()[0] == "op" and self.peek()[1] == "**":
                self.eat()
                e = ("bin", "**", e, self.binary(len(_BINPREC) - 1))
            return e if tok[1] == "+" else ("un", tok[1], e)
        return self.primary()

    def primary(self):
        k, v, line = self.peek()
        if k == "num":
            self.eat()
            return parse_number(v)
        if k == "str":
            self.eat()
            return ("str", v[1:-1])
        if k == "op" and v == "(":
            self.eat()
            e = self.expr()
            self.eat(")")
            return e
        if k in ("id", "sys"):
            name = self.ident() if k == "id" else self.eat()[1]
            if self.at("("):
                self.eat()
                args = []
                if not self.at(")"):
                    while True:
                        args.append(self.expr())
                        if self.at(","):
                            self.eat()
                            continue
                        break
                self.eat(")")
                return ("call", name, args)
            if k == "id" and self.at("["):
                self.eat()
                ix = self.expr()
                self.eat("]")
                return ("index", name, ix)
            return ("call", name, []) if k == "sys" else ("id", name)
        raise VAError("line %d: unexpected %r in expression" % (line, v))

    # -- statements --
    def statement(self):
        k, v, line = self.peek()
        if v == ";" and k == "op":
            self.eat()
            return ("null",)
        if k == "id" and v == "begin":
            self.eat()
            name, decls, stmts = None, {}, []
            if self.at(":"):
                self.eat()
                name = self.ident()
            while self.peek()[1] in ("real", "integer") and self.peek()[0] == "id":
                ty = self.eat()[1]
                for nm in self.idlist():
                    decls[nm] = ty
                self.eat(";")
            while not (self.peek()[0] == "id" and self.peek()[1] == "end"):
                if self.peek()[0] == "eof":
                    raise VAError("line %d: begin without end" % line)
                stmts.append(self.statement())
            self.eat("end")
            return ("block", name, decls, stmts)
        if k == "id" and v == "if":
            self.eat()
            self.eat("(")
            c = self.expr()
            self.eat(")")
            th = self.statement()
            el = None
            if self.peek()[0] == "id" and self.peek()[1] == "else":
                self.eat()
                el = self.statement()
            return ("if", c, th, el)
        if k == "id" and v == "case":
            self.eat()
            self.eat("(")
            e = self.expr()
            self.eat(")")
            items = []
            while not (self.peek()[0] == "id" and self.peek()[1] == "endcase"):
                if self.peek()[0] == "id" and self.peek()[1] == "default":
                    self.eat()
                    if self.at(":"):
                        self.eat()
                    items.append((None, self.statement()))
                else:
                    conds = [self.expr()]
                    while self.at(","):
                        self.eat()
                        conds.append(self.expr())
                    self.eat(":")
                    items.append((conds, self.statement()))
            self.eat("endcase")
            return ("case", e, items)
        if k == "id" and v == "for":
            self.eat()
            self.eat("(")
            init = self.assignment()
            self.eat(";")
            cond = self.expr()
            self.eat(";")
            upd = self.assignment()
            self.eat(")")
            return ("for", init, cond, upd, self.statement())
        if k == "id" and v == "while":
            self.eat()
            self.eat("(")
            c = self.expr()
            self.eat(")")
            return ("while", c, self.statement())
        if k == "id" and v == "repeat":
            self.eat()
            self.eat("(")
            c = self.expr()
            self.eat(")")
            return ("repeat", c, self.statement())
        if k == "op" and v == "@":
            self.eat()
            self.eat("(")
            depth = 1
            while depth:
                tv = self.eat()[1]
                depth += (tv == "(") - (tv == ")")
            return ("event", self.statement())
        if k == "sys":
            name = self.eat()[1]
            args = []
            if self.at("("):
                self.eat()
                if not self.at(")"):
                    while True:
                        args.append(self.expr())
                        if self.at(","):
                            self.eat()
                            continue
                        break
                self.eat(")")
            self.eat(";")
            return ("task", name, args)
        if k == "id":
            # contribution or assignment
            if self.peek(1)[1] == "(":
                save = self.i
                acc = self.ident()
                self.eat("(")
                nodes = [self.ident()]
                while self.at(","):
                    self.eat()
                    nodes.append(self.ident())
                self.eat(")")
                if self.peek()[1] == "<+":
                    self.eat()
                    e = self.expr()
                    self.eat(";")
                    return ("contrib", acc, nodes, e)
                self.i = save
            st = self.assignment()
            self.eat(";")
            return st
        raise VAError("line %d: unexpected %r at statement start" % (line, v))

    def assignment(self):
        name = self.ident()
        if self.at("["):
            self.eat()
            ix = self.expr()
            self.eat("]")
            self.eat("=")
            return ("assign_idx", name, ix, self.expr())
        self.eat("=")
        return ("assign", name, self.expr())

    def _const_int(self):
        e = self.expr()
        sign = 1
        while e[0] == "un" and e[1] == "-":
            sign, e = -sign, e[2]
        if e[0] != "num" or not e[2]:
            raise VAError("line %d: array bounds must be integer literals" % self.peek()[2])
        return sign * e[1]

    def idlist(self):
        names = []
        while True:
            nm = self.ident()
            if self.at("["):   # array variable: real x[lo:hi]
                self.eat()
                lo = self._const_int()
                self.eat(":")
                hi = self._const_int()
                self.eat("]")
                self.arrays[nm] = (min(lo, hi), max(lo, hi))
            if self.at("="):   # `real x = 1.0` initialisers are not part of the subset
                raise VAError("line %d: variable initialisers are not supported" % self.peek()[2])
            names.append(nm)
            if self.at(","):
                self.eat()
                continue
            return names

    # -- module level --
    def source(self):
        mods = []
        while self.peek()[0] != "eof":
            k, v, line = self.peek()
            if k == "id" and v in ("module", "macromodule"):
                mods.append(self.module())
            elif k == "id" and v in ("discipline", "nature"):
                end = "end" + v
                while not (self.peek()[0] == "id" and self.peek()[1] == end):
                    self.eat()
                self.eat()
            else:
                raise VAError("line %d: unexpected %r at top level" % (line, v))
        return mods

    def range_spec(self):
        lo_open = self.eat()[1] == "("
        lo = None if (self.peek()[1] in ("-", "+") and self.peek(1)[1] == "inf") else None
        def bound():
            if self.peek()[1] in ("-", "+") and self.peek(1)[1] == "inf":
                s = self.eat()[1]
                self.eat()
                return ("num", float("-inf") if s == "-" else float("inf"), False)
            if self.peek()[1] == "inf":
                self.eat()
                return ("num", float("inf"), False)
            return self.expr()
        lo = bound()
        self.eat(":")
        hi = bound()
        hi_open = self.eat()[1] == ")"
        return (lo, lo_open, hi, hi_open)

    def module(self):
        self.eat()
        m = Module(self.ident())
        m.arrays = self.arrays = {}
        if self.at("("):
            self.eat()
            if not self.at(")"):
                m.ports = [self.ident()]
                while self.at(","):
                    self.eat()
                    m.ports.append(self.ident())
            self.eat(")")
        self.eat(";")
        while True:
            attr = self._skip_attrs()
            k, v, line = self.peek()
            if k == "id" and v == "endmodule":
                self.eat()
                break
            if k == "eof":
                raise VAError("module %s: missing endmodule" % m.name)
            if k == "id" and v in ("inout", "input", "output"):
                self.eat()
                if self.peek()[1] in DISCIPLINES:
                    self.eat()
                self.idlist()
                self.eat(";")
            elif k == "id" and v in DISCIPLINES:
                self.eat()
                for nm in self.idlist():
                    if nm not in m.ports and nm not in m.internal:
                        m.internal.append(nm)
                self.eat(";")
            elif k == "id" and v == "ground":
                self.eat()
                self.idlist()
                self.eat(";")
            elif k == "id" and v in ("parameter", "localparam"):
                self.eat()
                ty = "real"
                if self.peek()[1] in ("real", "integer", "string"):
                    ty = self.eat()[1]
                while True:
                    nm = self.ident()
                    self.eat("=")
                    default = self.expr()
                    ranges = []
                    while self.peek()[0] == "id" and self.peek()[1] in ("from", "exclude"):
                        kind = self.eat()[1]
                        if self.peek()[1] in ("[", "("):
                            ranges.append((kind, self.range_spec()))
                        else:
                            ranges.append((kind, self.expr()))
                    m.params.append((nm, ty, default, ranges))
                    if self.at(","):
                        self.eat()
                        continue
                    break
                self.eat(";")
            elif k == "id" and v == "aliasparam":
                self.eat()
                a = self.ident()
                self.eat("=")
                m.aliases[a] = self.ident()
                self.eat(";")
            elif k == "id" and v in ("real", "integer", "string"):
                self.eat()
                desc = None
                if attr:
                    md = re.search(r'desc\s*=\s*"([^"]*)"', attr)
                    desc = md.group(1) if md else None
                for nm in self.idlist():
                    m.vars[nm] = v
                    if desc is not None and v != "string":
                        m.var_desc[nm] = desc   # operating-point observable (src/vasim.jl:742-753)
                self.eat(";")
            elif k == "id" and v == "branch":
                self.eat()
                self.eat("(")
                a = self.ident()
                b = None
                if self.at(","):
                    self.eat()
                    b = self.ident()
                self.eat(")")
                for nm in self.idlist():
                    m.branches[nm] = (a, b)
                self.eat(";")
            elif k == "id" and v == "analog":
                self.eat()
                if self.peek()[0] == "id" and self.peek()[1] == "function":
                    self.eat()
                    rt = "real"
                    if self.peek()[1] in ("real", "integer"):
                        rt = self.eat()[1]
                    f = Function(self.ident(), rt)
                    self.eat(";")
                    f.vars[f.name] = rt
                    while self.peek()[0] == "id" and self.peek()[1] in ("input", "output", "inout", "real", "integer"):
                        kw = self.eat()[1]
                        names = self.idlist()
                        self.eat(";")
                        if kw in ("real", "integer"):
                            for nm in names:
                                f.vars[nm] = kw
                        else:
                            for nm in names:
                                f.args.append((nm, kw))
                                f.vars.setdefault(nm, "real")
                    f.body = self.statement()
                    self.eat("endfunction")
                    m.functions[f.name] = f
                else:
                    if self.peek()[0] == "id" and self.peek()[1] == "initial":   # `analog initial`: runs before the analog block
                        self.eat()
                        m.analog.insert(m.n_initial, ("event", self.statement()))
                        m.n_initial += 1
                    else:
                        m.analog.append(self.statement())
            else:
                raise VAError("line %d: unexpected %r in module %s" % (line, v, m.name))
        return m


def parse_va(text, include_dirs=(), defines=None, cur_dir=None, reader=None):
    """Preprocess and parse Verilog-A source text; returns the list of modules."""
    pp = Preprocessor(include_dirs, defines, reader)
    return Parser(tokenize(pp.process(text, cur_dir))).source()


def parse_va_file(path, include_dirs=(), defines=None):
    with open(path) as f:
        return parse_va(f.read(), include_dirs, defines, cur_dir=os.path.dirname(os.path.abspath(path)))
