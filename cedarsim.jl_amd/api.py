"""User-facing analyses — host-side mirror of CedarSim's problem surface for the hot path.

    dc(circuit; abstol…)                ≙ dc!(circ)                 src/sweeps.jl:437-448
    tran(circuit, tspan; abstol, reltol) ≙ tran!(circ, tspan)        src/sweeps.jl:450-465
    CircuitSweep(builder, sweep)         ≙ CircuitSweep(circuit, sweep)  src/sweeps.jl:390-435
    dc(cs) / tran(cs, tspan)             ≙ dc!(cs) / tran!.(…) broadcast  src/sweeps.jl:448,471-502
    sol.retcode, sol.t, sol["node_q"], sol["r.i"], sol(t, idxs=…), sol.stats   SURVEY §8(b)
    ac(circuit) → ACSolution.freqresp(sym, ωs)  ≙ ac!(circ), freqresp(ac, sym, ωs)  src/ac.jl:166-177, 267-284
    noise(circuit) → NoiseSolution.psd(sym, ωs)  ≙ noise!(circ), PSD(noise, sym, ωs) src/ac.jl:178-186, 286-305
    acdec(nd, fstart, fstop)             ≙ acdec                     src/ac.jl:307-323

The reference loops serially over sweep points, paying a full DC + transient each
(`broadcast(sims) do sim … remake(prob, p=sim)`, src/sweeps.jl:473-480).  Here every point becomes one
*sample* of a single batched GPU solve (`ch_set_params` + one `ch_dc`/`ch_tran`); with several ranks
the points are split into contiguous shards, one per GPU, and results are gathered at the end.
All numerics run in libcedarhip.so; nothing here computes a circuit solution.
"""
import math

import numpy as np

from .circuit import (DEV_C, DEV_L, DEV_MOS, DEV_R, DEV_V, DEV_VCVS, RETCODES, CedarError, Circuit, dc_opts, tran_opts)
from .engine import EngineCircuit
from .sweeps import shard_range, sweepify


class Solution:
    """Solution of one sample: time points, observables by name, retcode and stats."""

    def __init__(self, circuit, t, cols, x_final, rc, stats, params=None):
        self.circuit, self.t, self._cols, self.x_final = circuit, np.asarray(t), cols, x_final
        self.retcode = RETCODES.get(rc, str(rc))
        self.rc = rc
        self.stats = stats
        self.params = params or {}

    # -- symbolic indexing: "node_q" / "q" (node voltage), "r1.i" / "r1.v" (device current / voltage) --
    def _node_series(self, node):
        c = self.circuit
        n = c._n(node) if not isinstance(node, (int, np.integer)) else int(node)
        if n == 0:
            return np.zeros(len(self.t))
        key = ("v", n)
        if key in self._cols:
            return self._cols[key]
        if len(self.t) == 1 and self.x_final is not None:
            return np.array([self.x_final[n - 1]])
        raise KeyError("node '%s' was not observed" % node)

    def __getitem__(self, name):
        c = self.circuit
        nm = str(name).lower()
        if nm.startswith("node_"):
            nm = nm[5:]
        if nm in c._node_ix:
            return self._node_series(nm)
        if "." in nm:
            dev, fld = nm.rsplit(".", 1)
            if dev in c.dev_names and fld in ("i", "v"):
                i = c.dev_names.index(dev)
                a, b = c.dev_node[i][0], c.dev_node[i][1]
                v = self._node_series(a) - self._node_series(b)
                if fld == "v":
                    return v
                k = c.dev_kind[i]
                if k == DEV_R:
                    return v / c.dev_par[i][0]
                if k in (DEV_V, DEV_L, DEV_VCVS):
                    key = ("i", i)
                    if key in self._cols:
                        return self._cols[key]
                    if len(self.t) == 1 and self.x_final is not None:
                        return np.array([self.x_final[c.mna_index("i", dev)]])
                    raise KeyError("branch current of '%s' was not observed" % dev)
                if k == DEV_C:
                    if len(self.t) < 2:
                        return np.zeros(len(self.t))
                    return c.dev_par[i][0] * np.gradient(v, self.t)  # post-processed derivative
        raise KeyError(name)

    def op(self, dev_name, index=-1, ctx=None):
        """Operating-point observables of a compiled Verilog-A instance at saved point `index` — `sol[sys.x1.gm]` for the
        variables a module declares with (* desc *) (src/vasim.jl:742-753).  Evaluated on the GPU from the node voltages."""
        from .engine import default_context
        c = self.circuit
        name = str(dev_name).lower()
        if name not in getattr(c, "va_instances", {}):
            raise KeyError("'%s' is not a compiled Verilog-A instance" % dev_name)
        i = c.dev_names.index(name)
        mid, ofs = c.dev_ipar[i]
        mod, _ = c.va_instances[name]
        v = [float(self._node_series(n)[index]) for n in c.dev_node[i][:len(mod.nodes)]]
        par = c.va_par[ofs:ofs + 2 * len(mod.params)]
        return (ctx or default_context()).va_opvars(mid, par, v, c.temp + 273.15, c.gmin)

    def default_name_map(self):
        """{solution key: column name} for every observed top-level node voltage — default_name_map (src/util.jl:239-260):
        the `node_` prefix is dropped and ground aliases are left out."""
        c = self.circuit
        out = {}
        for kind, idx in self._cols:
            if kind != "v":
                continue
            name = c.node_names[idx]
            if "." in name or name in ("0", "gnd"):
                continue
            out["node_" + name] = name
        return out

    def write_csv(self, path, name_map=None):
        """CSV.write(file, sol; name_map) (ext/CedarSimCSVExt.jl:13-19): column `t` then one column per mapped probe."""
        name_map = name_map or self.default_name_map()
        cols = [("t", self.t)] + [(name, self[key]) for key, name in name_map.items()]
        with open(path, "w") as f:
            f.write(",".join(n for n, _ in cols) + "\n")
            for i in range(len(self.t)):
                f.write(",".join(repr(float(v[i])) for _, v in cols) + "\n")
        return path

    def __call__(self, t, idxs=None):
        """`sol(t, idxs=…)` (test/gf180_dff.jl:29-33).  A run without a `saveat` grid carries, per accepted step, the number of
        newest saved points its BDF dense-output polynomial runs through (`ch_result_dense_points`): `sol(t)` evaluates exactly
        that polynomial — the value a `saveat` grid would have returned at t.  Results that are already on a `saveat` grid are
        interpolated between grid points by a shape-preserving cubic (PCHIP)."""
        names = idxs if isinstance(idxs, (list, tuple)) else [idxs]
        tt = np.asarray(self.t, float)
        pts = self.stats.get("dense_points") if isinstance(self.stats, dict) else None
        tq = np.atleast_1d(np.asarray(t, float))

        def dense(y):
            y = np.asarray(y, float)
            out = np.empty(len(tq))
            for q, x in enumerate(tq):
                if x <= tt[0]:
                    out[q] = y[0]
                    continue
                if x >= tt[-1]:
                    out[q] = y[-1]
                    continue
                i = int(np.searchsorted(tt, x, side="left"))          # tt[i-1] < x <= tt[i]: the step that ended at row i covers x
                m = min(int(pts[i]), i + 1)
                rows = list(range(i - m + 1, i + 1))
                if m < 2 or len(set(tt[rows])) < m:
                    out[q] = np.interp(x, tt[i - 1:i + 1], y[i - 1:i + 1]) if tt[i] > tt[i - 1] else y[i]
                    continue
                acc = 0.0
                for a in rows:
                    w = 1.0
                    for b in rows:
                        if b != a:
                            w *= (x - tt[b]) / (tt[a] - tt[b])
                    acc += w * y[a]
                out[q] = acc
            return out if np.ndim(t) else float(out[0])

        uniq = np.concatenate(([True], np.diff(tt) > 0)) if len(tt) > 1 else np.ones(len(tt), bool)   # restart steps repeat a time

        def pchip(y):
            y = np.asarray(y, float)
            if uniq.sum() < 3:
                return np.interp(t, tt, y)
            from scipy.interpolate import PchipInterpolator
            return PchipInterpolator(tt[uniq], y[uniq], extrapolate=False)(np.clip(t, tt[0], tt[-1]))
        interp = dense if (pts is not None and len(pts) == len(tt) and np.any(np.asarray(pts) >= 2)) else pchip
        out = [interp(self[n]) for n in names]
        out = [float(o) if np.ndim(o) == 0 else o for o in out]
        return out if isinstance(idxs, (list, tuple)) else out[0]


def _prepare(circuit, observe):
    if isinstance(circuit, Circuit):
        ckt = circuit
    elif hasattr(circuit, "build"):
        ckt = circuit.build()
    else:
        raise CedarError("expected a Circuit or a parsed netlist")
    if observe is None and not ckt.obs:
        ckt.observe_all_nodes()
        for i, k in enumerate(ckt.dev_kind):
            if k in (DEV_V, DEV_L, DEV_VCVS):
                ckt.observe_branch(ckt.dev_names[i])
    elif observe:
        for o in observe:
            o = str(o).lower()
            if o.endswith(".i") and o[:-2] in ckt.dev_names:
                ckt.observe_branch(o[:-2])
            else:
                ckt.observe_node(o[5:] if o.startswith("node_") else o)
    return ckt


def _solutions(ckt, t, v, xf, rc, status, st, S, point_params=None):
    sols = []
    for s in range(S):
        cols = {}
        for k, o in enumerate(ckt.obs):
            cols[o] = v[k, :, s] if v.size else np.zeros(0)
        sols.append(Solution(ckt, t, cols, xf[s] if xf is not None else None, int(status[s]) if status is not None else rc, st,
                             point_params[s] if point_params else None))
    return sols


def dc(circuit, abstol=1e-10, maxiters=200, n_restarts=10, seed=10, tran_mode=False, u0=None, observe=None, ctx=None):
    """DC operating point (CedarDCOp: sources at their .dc value, du = 0; src/dcop.jl:157-200)."""
    if isinstance(circuit, CircuitSweep):
        return circuit._run("dc", dict(abstol=abstol, maxiters=maxiters, n_restarts=n_restarts, seed=seed, tran_mode=tran_mode), ctx)
    ckt = _prepare(circuit, observe)
    eng = EngineCircuit(ckt, ctx)
    rc, x, status, st = eng.dc(dc_opts(abstol=abstol, maxiters=maxiters, n_restarts=n_restarts, seed=seed, tran_mode=tran_mode, x0=u0))
    if rc != 0:
        import warnings
        warnings.warn("DC operating point analysis failed. Further failures may follow.")  # src/dcop.jl:141
    cols = {}
    for o in ckt.obs:
        idx = (o[1] - 1) if o[0] == "v" else ckt.mna_index("i", ckt.dev_names[o[1]])
        cols[o] = np.array([x[0][idx]])
    return Solution(ckt, np.array([0.0]), cols, x[0], rc, st)


def tran(circuit, tspan=None, abstol=1e-6, reltol=1e-3, u0=None, initializealg="dcop", dc_abstol=1e-10, saveat=None,
         max_order=5, observe=None, ctx=None, **kw):
    """Transient (solve(prob, IDA(); abstol, reltol, initializealg=CedarDCOp()), src/sweeps.jl:450-465).

    u0: MNA initial state → skips the DC solve (test/common.jl:36-43 `u0=` semantics).
    tspan defaults to the netlist's `.TRAN` (src/circsummary.jl:109-128)."""
    if isinstance(circuit, CircuitSweep):
        return circuit._run("tran", dict(tspan=tspan, abstol=abstol, reltol=reltol, initializealg=initializealg, dc_abstol=dc_abstol,
                                          saveat=saveat, max_order=max_order, **kw), ctx)
    if tspan is None:
        nl = circuit if hasattr(circuit, "tran") and not isinstance(circuit, Circuit) else getattr(circuit, "_netlist", None)
        if nl is None or nl.tran is None:
            raise CedarError("no tspan given and the netlist has no .TRAN statement")
        tspan = (0.0, nl.tran[1])
    ckt = _prepare(circuit, observe)
    eng = EngineCircuit(ckt, ctx)
    dco = dc_opts(abstol=dc_abstol, tran_mode=(initializealg == "tranop"), x0=None if u0 is None else np.asarray(u0, float)[None, :])
    opts = tran_opts(abstol=abstol, reltol=reltol, max_order=max_order, saveat=saveat, dc=dco, skip_dc=u0 is not None, **kw)
    rc, t, v, xf, st = eng.tran(tspan[0], tspan[1], opts)
    return _solutions(ckt, t, v, xf, rc, None, st, 1)[0]


def acdec(nd, fstart, fstop):
    """`.ac dec nd fstart fstop`: log-spaced frequencies in Hz (src/ac.jl:317-322)."""
    a, b = math.log10(fstart), math.log10(fstop)
    points = int(math.ceil((b - a) * nd)) + 1
    return 10.0 ** np.linspace(a, b, points)


def _resolve_sym(ckt, sym):
    """'node_vout' / 'vout' → ('v', node id); 'l3.i' → ('i', device index); 'l3.v' → ('dv', a, b)."""
    nm = str(sym).lower()
    if nm.startswith("node_"):
        nm = nm[5:]
    if nm in ckt._node_ix:
        return ("v", ckt._n(nm))
    if "." in nm:
        dev, fld = nm.rsplit(".", 1)
        if dev in ckt.dev_names:
            i = ckt.dev_names.index(dev)
            if fld == "v":
                return ("dv", ckt.dev_node[i][0], ckt.dev_node[i][1])
            if fld == "i" and ckt.dev_kind[i] in (DEV_V, DEV_L, DEV_VCVS):
                return ("i", i)
    raise KeyError(sym)


class ACSolution:
    """Linearisation around the DC operating point, sampled on demand (ACSol, src/ac.jl:10-13)."""

    def __init__(self, circuit, eng, opts):
        self.circuit, self._eng, self._opts = circuit, eng, opts
        self.stats = None

    def freqresp(self, sym, omegas, sample=0):
        """Complex response of `sym` to the circuit's AC sources at angular frequencies `omegas` (rad/s)."""
        ckt = self.circuit
        w = np.asarray(omegas, dtype=np.float64)
        rc, x, st = self._eng.ac(w / (2.0 * math.pi), self._opts)
        self.stats = st
        if rc != 0:
            raise CedarError("AC analysis failed (%s): %s" % (RETCODES.get(rc, rc), self._eng.ctx.last_error()))
        x = x[sample]

        def node(n):
            return np.zeros(len(w), complex) if n == 0 else x[:, n - 1]

        r = _resolve_sym(ckt, sym)
        if r[0] == "v":
            return node(r[1])
        if r[0] == "dv":
            return node(r[1]) - node(r[2])
        v = x[:, ckt.mna_index("i", ckt.dev_names[r[1]])]
        if np.any(np.isnan(v)):
            raise KeyError("branch current of '%s' was eliminated: observe it when building the circuit" % sym)
        return v

    def bode(self, sym, omegas, sample=0):
        h = self.freqresp(sym, omegas, sample)
        return np.abs(h), np.degrees(np.unwrap(np.angle(h))), np.asarray(omegas)


class NoiseSolution:
    """Output-noise analysis around the DC operating point (NoiseSol, src/ac.jl:15-20)."""

    def __init__(self, circuit, eng, opts):
        self.circuit, self._eng, self._opts = circuit, eng, opts
        self.stats = None

    def psd(self, sym, omegas, sample=0):
        """Power spectral density of `sym` (V²/Hz or A²/Hz) at angular frequencies `omegas` — PSD(noise, sym, ωs)."""
        r = _resolve_sym(self.circuit, sym)
        if r[0] == "dv":
            raise CedarError("noise output must be a node voltage or a branch current")
        w = np.asarray(omegas, dtype=np.float64)
        rc, out, st = self._eng.noise(0 if r[0] == "v" else 1, r[1], w / (2.0 * math.pi), self._opts)
        self.stats = st
        if rc != 0:
            raise CedarError("noise analysis failed (%s): %s" % (RETCODES.get(rc, rc), self._eng.ctx.last_error()))
        return out[sample]


def ac(circuit, abstol=1e-10, maxiters=200, n_restarts=10, seed=10, ctx=None):
    """ac!(circ): DC operating point + linearisation; sample it with `.freqresp(sym, ωs)`."""
    ckt = _prepare(circuit, None)
    if not any(ckt.source_ac):
        raise CedarError("AC analysis needs at least one source with an `ac` magnitude")
    return ACSolution(ckt, EngineCircuit(ckt, ctx, small_signal=True), dc_opts(abstol=abstol, maxiters=maxiters, n_restarts=n_restarts, seed=seed))


def noise(circuit, abstol=1e-10, maxiters=200, n_restarts=10, seed=10, ctx=None):
    """noise!(circ): resistor thermal noise referred to an output with `.psd(sym, ωs)`."""
    ckt = _prepare(circuit, None)
    return NoiseSolution(ckt, EngineCircuit(ckt, ctx, small_signal=True), dc_opts(abstol=abstol, maxiters=maxiters, n_restarts=n_restarts, seed=seed))


class CircuitSweep:
    """Batched sweep over circuit parameters (src/sweeps.jl:390-435).

    builder: a parsed netlist (`.build(**params)`) or a callable `f(**params) -> Circuit`.
    Iterating yields one parameter dict per point (`s.params` in the reference)."""

    def __init__(self, builder, sweep, rank=0, world=1, groups="auto", warm_start=False):
        """warm_start=True: the DC operating point of the range's first point is solved alone and handed to every sample as its
        initial guess (`u0`), instead of ten `1e-7*randn` restarts per sample (src/dcop.jl:53-94).  Not the reference's behaviour
        (every `dc!` there starts cold), so it is opt-in; on the 1024-sample Monte-Carlo share of config 4 the DC phase drops from
        46 ms to 0.14 ms (profiles/r02_configs.json).  For a multistable circuit every sample then starts in the first point's basin."""
        self.warm_start = bool(warm_start)
        self.builder = builder
        self.sweep = sweepify(sweep)
        self.points = [{k: v for k, v in p if v is not None} for p in self.sweep]
        self.shape = self.sweep.shape
        self.rank, self.world = rank, world
        # groups > 1: this rank's points are split into that many contiguous groups, each a batched solve of its own with its
        # own stream and host stepper (one thread each).  A step attempt of one group costs ~36 us of kernel plus ~16 us of
        # host round trip during which the GPU would idle; a second group's kernel fills that gap (profiles/r01_notes.md),
        # and every group steps with the time steps ITS samples need.
        # Measured on one MI355X (scripts/sweep_groups.py, Monte-Carlo batch of one DFF): 8192 samples 0.46 s in one group,
        # 0.38 s in two, 0.35 s in four; no gain at 1024 samples (one launch does not fill the GPU there).  "auto": 4 groups
        # from 4096 points per rank, 2 from 2048, else 1.
        self.groups = groups if groups == "auto" else max(1, int(groups))
        self._build = builder.build if hasattr(builder, "build") else builder

    def __iter__(self):
        return iter(self.points)

    def __len__(self):
        return len(self.points)

    @staticmethod
    def _flat(c):
        """Every sweepable entry of a circuit's tables as one vector, with the engine slot (kind, a, b) of each position."""
        from .circuit import (SLOT_DEV_MULT, SLOT_DEV_PAR, SLOT_GMIN, SLOT_MODEL_PAR, SLOT_SRC_DC, SLOT_SRC_PAR, SLOT_TEMP, SLOT_VA_PAR)
        vals, keys = [], []
        par = np.array(c.dev_par, float).reshape(len(c.dev_par), -1) if len(c.dev_par) else np.zeros((0, 8))
        for d in range(par.shape[0]):
            vals.extend(par[d]); keys.extend((SLOT_DEV_PAR, d, k) for k in range(par.shape[1]))
        vals.extend(float(m) for m in c.dev_mult); keys.extend((SLOT_DEV_MULT, d, 0) for d in range(len(c.dev_mult)))
        for i, sv in enumerate(c.sources):
            vals.append(float(sv[0])); keys.append((SLOT_SRC_DC, i, 0))
            pr = list(sv[1].par) + [0.0] * (8 - len(sv[1].par))
            vals.extend(float(x) for x in pr); keys.extend((SLOT_SRC_PAR, i, k) for k in range(8))
        for m, card in enumerate(c.models):
            vals.extend(float(x) for x in card); keys.extend((SLOT_MODEL_PAR, m, k) for k in range(len(card)))
        vals.extend(float(x) for x in c.va_par); keys.extend((SLOT_VA_PAR, i, 0) for i in range(len(c.va_par)))
        vals.append(float(c.temp)); keys.append((SLOT_TEMP, 0, 0))
        vals.append(float(c.gmin)); keys.append((SLOT_GMIN, 0, 0))
        return np.array(vals, float), keys

    def _same_topology(self, base, c):
        if c.dev_kind != base.dev_kind or c.dev_node != base.dev_node or c.dev_ipar != base.dev_ipar:
            raise CedarError("sweep points must not change the circuit topology")

    def _batch(self, lo, hi):
        """Base circuit + slots + per-sample values for points lo..hi, found by diffing the flat tables of built circuits.

        The reference rebuilds nothing per point: `remake(prob, p=sim)` swaps a parameter struct (src/sweeps.jl:278-290,
        473-480).  Here the netlist is rebuilt only as often as needed to LEARN the swept-name -> table-entry map:
          * a variable with at most four distinct values: one build per value (exact look-up);
          * any other variable (the Monte-Carlo shape: a TandemSweep with as many distinct values as points): three builds —
            two determine an identity / proportional / affine map for every entry the variable moves, the third checks it;
          * the assembled table is then VALIDATED against full builds of the point with the most variables away from the
            base point and of a few seeded random points: an entry that answers to two variables (r = a*b with a base of
            a = 0, a conditional) is not visible from single-axis builds around one point.
        Any failed check (an entry moved by two variables, a non-affine map, a validation mismatch) falls back to one build
        per point, which is always correct.  `self.setup` records what was done (builds, seconds, which way)."""
        import time
        from .circuit import SLOT_SRC_DC, SLOT_SRC_PAR
        t_setup = time.perf_counter()
        pts = self.points[lo:hi]
        base = self._build(**pts[0])
        v0, keys = self._flat(base)
        differs = lambda x, y: (x != y) & ~(np.isnan(x) & np.isnan(y))  # noqa: E731
        n_builds = [1]

        def flat_of(point):
            c = self._build(**point)
            n_builds[0] += 1
            self._same_topology(base, c)
            vv, _ = self._flat(c)
            if len(vv) != len(v0):
                raise CedarError("sweep points must not change the circuit topology")
            return vv

        def close(x, y):
            return np.all((x == y) | (np.isnan(x) & np.isnan(y)) | (np.abs(x - y) <= 1e-13 * np.maximum(np.abs(x), np.abs(y))))

        table, how = None, "one build per point"
        names = sorted({k for p in pts for k in p})
        numeric = lambda v: isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool)  # noqa: E731
        if names and len(pts) > 1 and all(set(p) == set(names) for p in pts):
            distinct = {k: list(dict.fromkeys(p[k] for p in pts)) for k in names}
            n_check = min(4, len(pts) - 1)
            fitted = [k for k in names if len(distinct[k]) > 4 and all(numeric(x) for x in distinct[k])]   # variables whose map is fitted, not looked up
            cost = 1 + sum(min(len(v) - 1, 2) if all(numeric(x) for x in v) else len(v) - 1 for v in distinct.values()) + n_check + 2 * len(fitted)
            if cost < len(pts):
                table = self._learn_table(pts, names, distinct, v0, flat_of, differs, close, numeric)
                if table is not None:
                    # validation: the point farthest from the base point (most variables changed) + seeded random points
                    far = max(range(1, len(pts)), key=lambda r: sum(pts[r][k] != pts[0][k] for k in names))
                    rng = np.random.default_rng(len(pts))
                    picks = {far, len(pts) - 1} | {int(r) for r in rng.integers(1, len(pts), size=max(0, n_check - 2))}
                    # ... and, for every variable whose map was FITTED from three values, the points that hold its smallest and its
                    # largest value: a clipped or saturating entry (max(x, lower bound), a model card's limits) is affine on the
                    # three fitted values and wrong beyond the kink (scripts/extended_fuzz_sweepmap.py, 2 of 1 500 random builders)
                    # Among the points that hold such an extreme, the one with the most OTHER variables away from the base point: an
                    # entry that follows one variable only while another is beyond a threshold shows there and nowhere on the axes.
                    away = lambda r: sum(pts[r][j] != pts[0][j] for j in names)  # noqa: E731
                    for k in fitted:
                        for ext in (min(p[k] for p in pts), max(p[k] for p in pts)):
                            picks.add(max((r for r in range(len(pts)) if pts[r][k] == ext), key=lambda r: (away(r), r)))
                    picks.discard(0)
                    for r in sorted(picks):
                        if not close(flat_of(pts[r]), table[r]):
                            table = None   # e.g. an entry that depends on two swept variables
                            break
                if table is not None:
                    how = "learned map (%d builds for %d points)" % (n_builds[0], len(pts))
        if table is None:
            rows = [v0]
            for p in pts[1:]:
                rows.append(flat_of(p))
            table = np.array(rows)
        ch = np.nonzero(np.any(differs(table, table[0:1]), axis=0))[0]
        slots = [keys[i] for i in ch]
        # a constant source whose dc is swept: SRC_DC already updates the transient value
        drop = {i for i in ch if keys[i][0] == SLOT_SRC_PAR and keys[i][2] == 0 and base.sources[keys[i][1]][1].kind == 0 and (SLOT_SRC_DC, keys[i][1], 0) in slots}
        ch = [i for i in ch if i not in drop]
        slots = [keys[i] for i in ch]
        base.slots = list(slots)
        base.slot_names = [("slot%d" % i, None) for i in range(len(slots))]
        self.setup = {"points": len(pts), "circuit_builds": n_builds[0], "seconds": time.perf_counter() - t_setup, "how": how, "slots": len(slots)}
        return base, list(range(len(slots))), np.ascontiguousarray(table[:, ch].T, float).reshape(len(slots), hi - lo)

    @staticmethod
    def _learn_table(pts, names, distinct, v0, flat_of, differs, close, numeric):
        """Per-point flat tables from single-axis builds around pts[0]; None when the sweep is not separable that way."""
        owner = np.full(len(v0), -1)
        table = np.tile(v0, (len(pts), 1))
        for ki, k in enumerate(names):
            vals = distinct[k]
            x0 = pts[0][k]
            others = [v for v in vals if v != x0]
            if not others:
                continue
            cols = {x0: v0}

            def lookup():
                moved = np.zeros(len(v0), bool)
                for val in others:
                    if val not in cols:
                        cols[val] = flat_of(dict(pts[0], **{k: val}))
                    moved |= differs(cols[val], v0)
                if np.any(moved & (owner >= 0) & (owner != ki)):
                    return False
                owner[moved] = ki
                idx = np.nonzero(moved)[0]
                for r, p in enumerate(pts):
                    table[r, idx] = cols[p[k]][idx]
                return True

            if len(others) <= 3 or not all(numeric(v) for v in vals):
                if not lookup():
                    return None
                continue
            # many distinct numeric values: every entry the variable moves must be an affine function of it
            x1 = max(others, key=lambda v: abs(v - x0))
            x2 = min((v for v in others if v != x1), key=lambda v: abs(v - 0.5 * (x0 + x1)))
            v1, v2 = flat_of(dict(pts[0], **{k: x1})), flat_of(dict(pts[0], **{k: x2}))
            cols[x1], cols[x2] = v1, v2
            moved = differs(v1, v0) | differs(v2, v0)
            if np.any(moved & (owner >= 0)):
                return None
            idx = np.nonzero(moved)[0]
            if not len(idx):
                continue
            xs = np.array([float(p[k]) for p in pts])
            a0, a1, a2 = v0[idx], v1[idx], v2[idx]
            ident = (a0 == x0) & (a1 == x1) & (a2 == x2)
            with np.errstate(all="ignore"):
                cprop = a1 / x1 if x1 != 0 else np.full(len(idx), np.nan)
                prop = ~ident & (cprop * x0 == a0) & (cprop * x2 == a2)
                slope = (a1 - a0) / (float(x1) - float(x0))
                pred2 = a0 + slope * (float(x2) - float(x0))
            if not close(np.where(ident | prop, a2, pred2), a2):
                # not affine in the swept variable (1/x, x^2, a table look-up ...): one build per distinct value, if that is
                # still cheaper than one per point
                if 2 * len(vals) < len(pts) and lookup():
                    continue
                return None
            owner[moved] = ki
            col = a0[None, :] + slope[None, :] * (xs[:, None] - float(x0))
            col = np.where(prop[None, :], cprop[None, :] * xs[:, None], col)   # the builder's own product, bit for bit
            col = np.where(ident[None, :], xs[:, None], col)
            table[:, idx] = col
        return table

    def _run(self, kind, kw, ctx):
        lo, hi = shard_range(len(self.points), self.rank, self.world)
        if hi <= lo:
            return []
        n = hi - lo
        G = (4 if n >= 4096 else 2 if n >= 2048 else 1) if self.groups == "auto" else min(self.groups, n)
        if G <= 1:
            return self._run_range(kind, kw, ctx, lo, hi)
        from concurrent.futures import ThreadPoolExecutor
        from .engine import Context, default_context
        first = ctx if ctx is not None else default_context()
        ctxs = [first] + [Context(first.device_id) for _ in range(G - 1)]

        def work(g):
            a, b = shard_range(hi - lo, g, G)
            return self._run_range(kind, dict(kw), ctxs[g], lo + a, lo + b)

        with ThreadPoolExecutor(G) as ex:
            parts = list(ex.map(work, range(G)))
        return [sol for part in parts for sol in part]

    def _run_range(self, kind, kw, ctx, lo, hi):
        base, slot_ids, vals = self._batch(lo, hi)
        ckt = _prepare(base, None)
        eng = EngineCircuit(ckt, ctx)
        S = hi - lo
        eng.set_samples(S)
        if slot_ids:
            eng.set_params(slot_ids, vals)
        pts = self.points[lo:hi]
        x0 = None
        if self.warm_start and S > 1:
            e0 = EngineCircuit(_prepare(self._build(**pts[0]), None), ctx)
            rc0, xw, _, _ = e0.dc(dc_opts(abstol=kw.get("dc_abstol", kw.get("abstol", 1e-10)) if kind != "dc" else kw.get("abstol", 1e-10)))
            if rc0 == 0:
                x0 = np.tile(np.nan_to_num(xw[0]), (S, 1))
        if kind == "dc":
            rc, x, status, st = eng.dc(dc_opts(x0=x0, **kw))
            sols = []
            for s in range(S):
                cols = {}
                for o in ckt.obs:
                    idx = (o[1] - 1) if o[0] == "v" else ckt.mna_index("i", ckt.dev_names[o[1]])
                    cols[o] = np.array([x[s][idx]])
                ck = self._sample_circuit(ckt, slot_ids, vals, s)
                sols.append(Solution(ck, np.array([0.0]), cols, x[s], int(status[s]), st, pts[s]))
            return sols
        tspan = kw.pop("tspan")
        dco = dc_opts(abstol=kw.pop("dc_abstol", 1e-10), tran_mode=(kw.pop("initializealg", "dcop") == "tranop"), x0=x0)
        opts = tran_opts(dc=dco, **kw)
        rc, t, v, xf, st = eng.tran(tspan[0], tspan[1], opts)
        sols = _solutions(ckt, t, v, xf, rc, None, st, S, pts)
        for s, sol in enumerate(sols):
            sol.circuit = self._sample_circuit(ckt, slot_ids, vals, s)
        return sols

    def tran_arrays(self, tspan, abstol=1e-6, reltol=1e-3, dc_abstol=1e-10, saveat=None, ctx=None, **kw):
        """This rank's share of the sweep as ONE batched transient, results as arrays instead of per-point Solutions: the shape
        the result gather of a sharded sweep moves (`gather_sharded`).  Returns (rc, t[n_save],
        rows[samples, n_obs, n_save], stats); observables in the order of `builder(...).obs`."""
        lo, hi = shard_range(len(self.points), self.rank, self.world)
        base, slot_ids, vals = self._batch(lo, hi)
        ckt = _prepare(base, None)
        eng = EngineCircuit(ckt, ctx)
        eng.set_samples(hi - lo)
        if slot_ids:
            eng.set_params(slot_ids, vals)
        opts = tran_opts(abstol=abstol, reltol=reltol, saveat=saveat, dc=dc_opts(abstol=dc_abstol), **kw)
        rc, t, v, xf, st = eng.tran(tspan[0], tspan[1], opts)
        self._last_engine = eng   # st["device_rows"] (the same rows still in HBM, [n_obs][n_times][samples]) lives as long as this circuit
        return rc, t, np.ascontiguousarray(np.transpose(v, (2, 0, 1))), st

    @staticmethod
    def _sample_circuit(ckt, slot_ids, vals, s):
        """Shallow per-sample view so that post-processing (e.g. R.I = V/r) uses the sample's values."""
        import copy
        from .circuit import SLOT_DEV_PAR
        c = copy.copy(ckt)
        c.dev_par = [list(p) for p in ckt.dev_par]
        for i, sl in enumerate(ckt.slots):
            if sl[0] == SLOT_DEV_PAR:
                c.dev_par[sl[1]][sl[2]] = float(vals[i][s])
        return c


def gather_sharded_device(rows, n_total, rank, world, group=None):
    """The same gather with the rows never leaving HBM: `rows` is a CUDA tensor [n_obs, n_times, n_local] (samples fastest — the
    engine's own layout, e.g. `torch.as_tensor(stats["device_rows"], device="cuda")`); one `all_gather` over RCCL; returns a CUDA
    tensor [n_obs, n_times, n_total] on every rank."""
    import torch
    import torch.distributed as dist
    max_local = -(-n_total // world)
    n_obs, n_t, n_loc = rows.shape
    buf = rows if n_loc == max_local else torch.cat((rows, rows.new_zeros((n_obs, n_t, max_local - n_loc))), dim=2)
    buf = buf.contiguous()
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r][:, :, :hi - lo])
    return torch.cat(parts, dim=2)


def gather_sharded(local, n_total, rank, world, group=None, device=None):
    """All-gather per-rank result rows (one collective at the end, none in the Newton
    loop).  `local` is a float64 array [n_local, ...] for this rank's contiguous shard; returns the
    full [n_total, ...] array on every rank.  Backend nccl (= RCCL over xGMI) needs `device='cuda'`."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=np.float64)
    tail = local.shape[1:]
    width = int(np.prod(tail)) if tail else 1
    max_rows = -(-n_total // world)
    buf = torch.zeros((max_rows, width), dtype=torch.float64, device=device or "cpu")
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local.reshape(local.shape[0], width)).to(buf.device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rows.append(out[r][:hi - lo].cpu().numpy())
    return np.concatenate(rows, axis=0).reshape((n_total,) + tuple(tail))
