"""Flat circuit description: the host-side mirror of what CedarSim's netlist closure describes.

In the reference a circuit is a Julia closure that calls device functors on `Net`s
(src/simulate_ir.jl:28-91, src/simpledevices.jl); SURVEY §3.5 shows how a `StampExtract` overlay
records `(device type, field values, net ids, multiplier)` from it.  `Circuit` is that record, built
either programmatically (mirroring `Named(R(2.), "R")(vcc, gnd)`) or by `netlist.parse_spice`.
`Circuit.to_desc()` produces the `ch_desc` of include/cedarhip.h.
"""
import ctypes as C
import math

import numpy as np

from . import bsim4_params as B4

# ---- enums of include/cedarhip.h -------------------------------------------------------------
DEV_R, DEV_C, DEV_L, DEV_V, DEV_I, DEV_VCVS, DEV_VCCS, DEV_MOS, DEV_VA = 1, 2, 3, 4, 5, 6, 7, 8, 9
DEV_NNODE, DEV_NPAR, DEV_NIPAR = 8, 8, 2
MOS_W, MOS_L, MOS_NF, MOS_AS, MOS_AD, MOS_PS, MOS_PD = 0, 1, 2, 3, 4, 5, 6
SRC_DC, SRC_PWL, SRC_PULSE, SRC_SIN = 0, 1, 2, 3
SRC_NPAR = 8
SLOT_DEV_PAR, SLOT_MODEL_PAR, SLOT_SRC_DC, SLOT_SRC_PAR, SLOT_TEMP, SLOT_GMIN, SLOT_DEV_MULT, SLOT_VA_PAR = 1, 2, 3, 4, 5, 6, 7, 8

OK, ERR_INVALID, ERR_SINGULAR, ERR_MAXITERS, ERR_DTMIN, ERR_DEVICE, ERR_UNSUPPORTED, ERR_MAXSTEPS, ERR_NOMEM, ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7, -8, -9
RETCODES = {0: "Success", -1: "Invalid", -2: "Singular", -3: "InitialFailure", -4: "DtLessThanMin",
            -5: "DeviceError", -6: "Unsupported", -7: "MaxIters", -8: "OutOfMemory", -9: "InternalError"}

_pi32 = C.POINTER(C.c_int32)
_pf64 = C.POINTER(C.c_double)


class ChDesc(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32),
        ("n_dev", C.c_int32),
        ("dev_kind", _pi32), ("dev_node", _pi32), ("dev_ipar", _pi32), ("dev_par", _pf64), ("dev_mult", _pf64),
        ("n_src", C.c_int32),
        ("src_kind", _pi32), ("src_dc", _pf64), ("src_par", _pf64), ("src_pwl_ofs", _pi32), ("pwl_t", _pf64), ("pwl_y", _pf64),
        ("n_model", C.c_int32),
        ("model_par", _pf64),
        ("temp", C.c_double), ("gmin", C.c_double), ("scale", C.c_double),
        ("n_slot", C.c_int32),
        ("slot_kind", _pi32), ("slot_a", _pi32), ("slot_b", _pi32),
        ("n_obs", C.c_int32),
        ("obs_kind", _pi32), ("obs_index", _pi32), ("src_ac", _pf64), ("n_va_par", C.c_int64), ("va_par", _pf64),
    ]


class ChStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("nf", "njacs", "nfactors", "nsolve", "nnonliniter", "nnonlinconvfail",
                                         "naccept", "nreject", "nrestarts")] + \
               [("wall_seconds", C.c_double), ("dc_seconds", C.c_double), ("device_seconds", C.c_double),
                ("n_kernel_launches", C.c_int64), ("n_block_iters", C.c_int64), ("n_step_attempts", C.c_int64),
                ("barrier_seconds", C.c_double), ("stepper", C.c_int32), ("stepper_mode", C.c_int32),
                ("step_kernel_seconds", C.c_double), ("step_kernel_launches", C.c_int64), ("step_block_iters", C.c_int64)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class ChDcOpts(C.Structure):
    _fields_ = [("abstol", C.c_double), ("maxiters", C.c_int32), ("n_restarts", C.c_int32), ("seed", C.c_uint64),
                ("tran_mode", C.c_int32), ("dv_max", C.c_double), ("x0", _pf64)]


class ChTranOpts(C.Structure):
    _fields_ = [("abstol", C.c_double), ("reltol", C.c_double), ("max_order", C.c_int32), ("dtmin", C.c_double),
                ("dtmax", C.c_double), ("dt0", C.c_double), ("max_steps", C.c_int32), ("newton_maxiters", C.c_int32),
                ("n_saveat", C.c_int32), ("saveat", _pf64), ("dc", ChDcOpts), ("skip_dc", C.c_int32), ("stepper", C.c_int32), ("step_control", C.c_int32)]


class ChInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_nodes", "n_branches", "n_mna", "n_unknowns", "n_known", "n_alias",
                                         "n_components", "max_component", "n_classes", "n_mos", "n_mos_classes", "path")] + \
               [("nnz_jac", C.c_int64), ("nnz_lu", C.c_int64), ("n_samples", C.c_int32)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def dc_opts(abstol=1e-10, maxiters=200, n_restarts=10, seed=10, tran_mode=False, dv_max=2.0, x0=None):
    """CedarDCOp defaults (src/dcop.jl:28,53); seed 10 echoes test/runtests.jl `Random.seed!(10)`."""
    o = ChDcOpts()
    o.abstol, o.maxiters, o.n_restarts, o.seed = abstol, maxiters, n_restarts, seed
    o.tran_mode, o.dv_max = int(bool(tran_mode)), dv_max
    keep = None
    if x0 is not None:
        keep = np.ascontiguousarray(x0, dtype=np.float64)
        o.x0 = keep.ctypes.data_as(_pf64)
    o._keep = keep
    return o


def tran_opts(abstol=1e-6, reltol=1e-3, max_order=5, dtmin=0.0, dtmax=0.0, dt0=0.0, max_steps=0,
              newton_maxiters=10, saveat=None, dc=None, skip_dc=False, stepper="auto", step_control="auto"):
    """stepper: "auto" | "host" | "device" — where the sequential step controller runs (include/cedarhip.h CH_STEPPER_*).
    step_control: "auto" | "shared" — "shared" keeps ONE step sequence and error norm over the whole circuit / batch even on a
    `saveat` grid (CH_STEPS_SHARED: what a single IDA() over the whole system does, src/sweeps.jl:456)."""
    o = ChTranOpts()
    o.step_control = {"auto": 0, "shared": 1}[step_control] if isinstance(step_control, str) else int(step_control)
    o.stepper = {"auto": 0, "host": 1, "device": 2}[stepper] if isinstance(stepper, str) else int(stepper)
    o.abstol, o.reltol, o.max_order = abstol, reltol, max_order
    o.dtmin, o.dtmax, o.dt0, o.max_steps, o.newton_maxiters = dtmin, dtmax, dt0, max_steps, newton_maxiters
    keep = None
    if saveat is not None:
        keep = np.ascontiguousarray(saveat, dtype=np.float64)
        o.n_saveat = len(keep)
        o.saveat = keep.ctypes.data_as(_pf64)
    dco = dc if dc is not None else dc_opts()
    o.dc = dco
    o.skip_dc = int(bool(skip_dc))
    o._keep = (keep, dco, getattr(dco, "_keep", None))
    return o


# ---- waveforms (src/spectre_env.jl:144-176) ----------------------------------------------------
class Wave:
    kind = SRC_DC
    par = ()
    ts = ()
    ys = ()


class DC(Wave):
    def __init__(self, v):
        self.kind, self.par = SRC_DC, (float(v),)


class PWL(Wave):
    """pwl(wave): alternating t, y values (spectre_env.jl:36-41,144-151)."""

    def __init__(self, ts, ys=None):
        if ys is None:
            flat = list(ts)
            if len(flat) % 2:
                raise ValueError("PWL must have an equal number of x and y values")  # PWLConstructError
            ts, ys = flat[0::2], flat[1::2]
        if len(ts) != len(ys):
            raise ValueError("PWL must have an equal number of x and y values")
        self.kind, self.ts, self.ys = SRC_PWL, [float(t) for t in ts], [float(y) for y in ys]


class PULSE(Wave):
    def __init__(self, v1, v2, td=0.0, tr=0.0, tf=0.0, pw=math.inf, period=math.inf):
        self.kind = SRC_PULSE
        self.par = tuple(float(v) for v in (v1, v2, td, tr, tf, pw, period))


class SIN(Wave):
    def __init__(self, vo, va, freq, td=0.0, theta=0.0, phase=0.0, ncycles=math.inf):
        self.kind = SRC_SIN
        self.par = tuple(float(v) for v in (vo, va, freq, td, theta, phase, ncycles))


class CedarError(Exception):
    """Mirror of CedarSim.CedarError (src/util.jl:14-21)."""


class Circuit:
    """Programmatic circuit builder.

    >>> c = Circuit(); vcc = c.net("vcc")
    >>> c.V("V", vcc, 0, dc=5.0); c.R("R", vcc, 0, 2.0)          # test/basic.jl:21-27
    """

    def __init__(self, temp=27.0, gmin=1e-12, scale=1.0):
        self.temp, self.gmin, self.scale = float(temp), float(gmin), float(scale)
        self.node_names = ["0"]
        self._node_ix = {"0": 0, "gnd": 0, "gnd!": 0}
        self.dev_names, self.dev_kind, self.dev_node, self.dev_ipar, self.dev_par, self.dev_mult = [], [], [], [], [], []
        self.sources = []  # (dc, Wave)
        self.source_ac = []  # |ac| per source (small-signal magnitude; the phase is ignored, simpledevices.jl:293)
        self.model_names, self.models = [], []
        self.slots, self.slot_names = [], []
        self.obs, self.obs_names = [], []
        self.va_par = []       # parameter blocks of the compiled Verilog-A instances (values, then $param_given flags)
        self.va_instances = {}  # device name -> (module, resolved parameters)

    # -- nets --
    def net(self, name):
        name = str(name).lower()
        if name not in self._node_ix:
            self._node_ix[name] = len(self.node_names)
            self.node_names.append(name)
        return self._node_ix[name]

    def _n(self, x):
        return x if isinstance(x, (int, np.integer)) else self.net(x)

    @property
    def n_nodes(self):
        return len(self.node_names) - 1

    # -- devices --
    def _add(self, name, kind, nodes, par=(), ipar=(), m=1.0):
        if m < 0:
            raise CedarError("Cannot construct a ParallelInstances with non-positive multiplier '%s'" % m)  # simulate_ir.jl:62
        name = str(name).lower()
        self.dev_names.append(name)
        self.dev_kind.append(kind)
        nn = [self._n(x) for x in nodes] + [0] * (DEV_NNODE - len(nodes))
        self.dev_node.append(nn)
        pp = list(par) + [math.nan] * (DEV_NPAR - len(par))
        self.dev_par.append([math.nan if p is None else float(p) for p in pp])
        self.dev_ipar.append(list(ipar) + [0] * (DEV_NIPAR - len(ipar)))
        self.dev_mult.append(float(m))
        return len(self.dev_names) - 1

    def R(self, name, a, b, r=None, m=1.0, rsh=50.0, w=1e-6, l=1e-6, narrow=0.0, short=0.0):
        if r is None:  # semiconductor resistor (simpledevices.jl:66-70)
            r = rsh * (l - short) / (w - narrow)
        return self._add(name, DEV_R, (a, b), (r,), m=m)

    def C(self, name, a, b, c, m=1.0):
        return self._add(name, DEV_C, (a, b), (c,), m=m)

    def L(self, name, a, b, l, m=1.0):
        return self._add(name, DEV_L, (a, b), (l,), m=m)

    def _source(self, dc, tran, ac=0.0):
        # VoltageSource(;dc, tran): dc = something(dc, tran, 0); tran = something(tran, dc) (simpledevices.jl:279-283)
        if tran is not None and not isinstance(tran, Wave):
            tran = DC(tran)
        if dc is None:
            dc = _wave_value_at_zero(tran) if tran is not None else 0.0
        if tran is None:
            tran = DC(dc)
        self.sources.append((float(dc), tran))
        self.source_ac.append(abs(complex(ac)))
        return len(self.sources) - 1

    def V(self, name, a, b, dc=None, tran=None, m=1.0, ac=0.0):
        return self._add(name, DEV_V, (a, b), ipar=(self._source(dc, tran, ac),), m=m)

    def I(self, name, a, b, dc=None, tran=None, m=1.0, ac=0.0):
        return self._add(name, DEV_I, (a, b), ipar=(self._source(dc, tran, ac),), m=m)

    def E(self, name, a, b, c, d, gain=1.0, m=1.0):
        return self._add(name, DEV_VCVS, (a, b, c, d), (gain,), m=m)

    def G(self, name, a, b, c, d, gain=1.0, m=1.0):
        return self._add(name, DEV_VCCS, (a, b, c, d), (gain,), m=m)

    def VA(self, name, module, nodes, params=None, m=1.0):
        """Instance of a compiled Verilog-A module — the device functor `make_spice_device` builds from the module
        (src/vasim.jl:649-867).  `nodes` are the ports in declaration order; internal nets become circuit nodes
        `<name>.<net>`; `V(a,b) <+ 0` contributions that the parameter set activates merge their nodes."""
        from .va.interp import Interp
        from .va.registry import find_module
        mid, mod = find_module(module)
        name = str(name).lower()
        if len(nodes) != len(mod.ports):
            raise CedarError("module %s has %d ports, %d nodes given" % (mod.name, len(mod.ports), len(nodes)))
        it = Interp(mod, params or {}, temperature_c=self.temp, gmin=self.gmin)
        # internal nets and the branch-current unknowns of voltage branches become circuit nodes of this instance
        all_nodes = [self._n(x) for x in nodes] + [self.net("%s.%s" % (name, n)) for n in mod.nodes[len(mod.ports):]]
        ofs = len(self.va_par)
        for pname, ty, _, _ in mod.params:
            self.va_par.append(float(it.params[pname]) if ty != "string" else 0.0)
        for pname, ty, _, _ in mod.params:
            self.va_par.append(1.0 if pname in it.given else 0.0)
        # structure probe: which voltage contributions (node collapses) does this parameter set execute?
        it.evaluate({})
        k = 0
        for acc, nds, kind in it.structure:
            if acc == "V" and kind == "collapse":
                if len(nds) != 2:
                    raise CedarError("V(%s) <+ 0 to ground is not supported" % nds[0])
                a, b = (all_nodes[mod.nodes.index(x)] for x in nds)
                self.V("%s.collapse%d" % (name, k), a, b, dc=0.0)
                k += 1
        self.va_instances[name] = (mod, it.params)
        return self._add(name, DEV_VA, all_nodes, ipar=(mid, ofs), m=m)

    def add_model(self, name, mtype, params):
        """BSIM4 card.  mtype 'nmos'|'pmos' → TYPE=±1 (src/spectre.jl:632-643)."""
        arr = [math.nan] * B4.NPAR
        arr[B4.PARAM_INDEX["type"]] = 1.0 if str(mtype).lower().startswith("n") else -1.0
        for k, v in params.items():
            k = k.lower()
            if k in B4.PARAM_INDEX:
                arr[B4.PARAM_INDEX[k]] = float(v)
            elif k in B4.IGNORED:
                continue
            else:
                raise CedarError("unknown BSIM4 model parameter '%s'" % k)
        lvl = arr[B4.PARAM_INDEX["level"]]
        if not math.isnan(lvl) and int(lvl) not in (14, 54):
            raise CedarError("only BSIM4 (level 14/54) MOS models are supported, got level %s" % lvl)
        self.model_names.append(str(name).lower())
        self.models.append(arr)
        return len(self.models) - 1

    def M(self, name, d, g, s, b, model, w, l, nf=None, m=1.0, as_=None, ad=None, ps=None, pd=None):
        mi = model if isinstance(model, int) else self.model_names.index(str(model).lower())
        return self._add(name, DEV_MOS, (d, g, s, b), (w, l, nf, as_, ad, ps, pd), ipar=(mi,), m=m)

    # -- sweepable parameters (ParamSim fields, src/circuitodesystem.jl:66-97) --
    _MAIN = {DEV_R: 0, DEV_C: 0, DEV_L: 0, DEV_VCVS: 0, DEV_VCCS: 0}

    def slot(self, name, field=None):
        """Declare a runtime parameter.  name: device name, model name, 'temp' or 'gmin'."""
        key = (str(name).lower(), None if field is None else str(field).lower())
        if key in self.slot_names:
            return self.slot_names.index(key)
        nm, fld = key
        if nm == "temp" and fld is None:
            s = (SLOT_TEMP, 0, 0)
        elif nm == "gmin" and fld is None:
            s = (SLOT_GMIN, 0, 0)
        elif nm in self.dev_names:
            i = self.dev_names.index(nm)
            k = self.dev_kind[i]
            if fld in ("m", "mult"):
                s = (SLOT_DEV_MULT, i, 0)
            elif k in (DEV_V, DEV_I):
                si = self.dev_ipar[i][0]
                if fld in (None, "dc"):
                    s = (SLOT_SRC_DC, si, 0)
                else:
                    s = (SLOT_SRC_PAR, si, int(fld))
            elif k == DEV_VA:
                mod, _ = self.va_instances[nm]
                names = [p[0].lower() for p in mod.params]
                s = (SLOT_VA_PAR, self.dev_ipar[i][1] + names.index(mod.aliases.get(fld, fld).lower() if fld not in names else fld), 0)
            elif k == DEV_MOS:
                s = (SLOT_DEV_PAR, i, {"w": MOS_W, "l": MOS_L, "nf": MOS_NF, "as": MOS_AS, "ad": MOS_AD, "ps": MOS_PS, "pd": MOS_PD}[fld])
            else:
                s = (SLOT_DEV_PAR, i, self._MAIN[k])
        elif nm in self.model_names:
            s = (SLOT_MODEL_PAR, self.model_names.index(nm), B4.PARAM_INDEX[fld])
        else:
            raise CedarError("unknown parameter '%s'" % name)
        self.slots.append(s)
        self.slot_names.append(key)
        return len(self.slots) - 1

    # -- observables --
    def observe_node(self, node):
        key = ("v", self._n(node))
        if key not in self.obs:
            self.obs.append(key)
        return self.obs.index(key)

    def observe_branch(self, dev_name):
        i = self.dev_names.index(str(dev_name).lower())
        if self.dev_kind[i] not in (DEV_L, DEV_V, DEV_VCVS):
            raise CedarError("device '%s' has no branch current unknown" % dev_name)
        key = ("i", i)
        if key not in self.obs:
            self.obs.append(key)
        return self.obs.index(key)

    def observe_all_nodes(self):
        for k in range(1, self.n_nodes + 1):
            self.observe_node(k)

    @property
    def branch_devices(self):
        return [i for i, k in enumerate(self.dev_kind) if k in (DEV_L, DEV_V, DEV_VCVS)]

    @property
    def n_mna(self):
        return self.n_nodes + len(self.branch_devices)

    def mna_index(self, what, name):
        """Index into an x_mna vector: what='v' (node name) or 'i' (branch device name)."""
        if what == "v":
            n = self._n(name)
            return None if n == 0 else n - 1
        i = self.dev_names.index(str(name).lower())
        return self.n_nodes + self.branch_devices.index(i)

    # -- flatten --
    def to_desc(self, small_signal=True):
        keep = {}

        def arr(key, data, dtype):
            a = np.ascontiguousarray(np.asarray(data, dtype=dtype).reshape(-1))
            if a.size == 0:
                a = np.zeros(1, dtype=dtype)
            keep[key] = a
            return a.ctypes.data_as(_pi32 if dtype == np.int32 else _pf64)

        d = ChDesc()
        d.n_nodes = self.n_nodes
        d.n_dev = len(self.dev_kind)
        d.dev_kind = arr("dk", self.dev_kind, np.int32)
        d.dev_node = arr("dn", self.dev_node, np.int32)
        d.dev_ipar = arr("di", self.dev_ipar, np.int32)
        d.dev_par = arr("dp", self.dev_par, np.float64)
        d.dev_mult = arr("dm", self.dev_mult, np.float64)
        d.n_src = len(self.sources)
        kinds, dcs, pars, ofs, pt, py = [], [], [], [0], [], []
        for dc, w in self.sources:
            kinds.append(w.kind)
            dcs.append(dc)
            p = list(w.par) + [0.0] * (SRC_NPAR - len(w.par))
            pars.append(p)
            pt.extend(w.ts)
            py.extend(w.ys)
            ofs.append(len(pt))
        d.src_kind = arr("sk", kinds, np.int32)
        d.src_dc = arr("sd", dcs, np.float64)
        d.src_par = arr("sp", pars, np.float64)
        d.src_pwl_ofs = arr("so", ofs, np.int32)
        d.pwl_t = arr("pt", pt, np.float64)
        d.pwl_y = arr("py", py, np.float64)
        d.n_model = len(self.models)
        d.model_par = arr("mp", self.models, np.float64)
        d.temp, d.gmin, d.scale = self.temp, self.gmin, self.scale
        d.n_slot = len(self.slots)
        d.slot_kind = arr("s0", [s[0] for s in self.slots], np.int32)
        d.slot_a = arr("s1", [s[1] for s in self.slots], np.int32)
        d.slot_b = arr("s2", [s[2] for s in self.slots], np.int32)
        d.n_obs = len(self.obs)
        d.obs_kind = arr("o0", [0 if o[0] == "v" else 1 for o in self.obs], np.int32)
        d.obs_index = arr("o1", [o[1] for o in self.obs], np.int32)
        # AC magnitudes travel only for small-signal analyses: an AC-driven voltage source keeps its node and branch as
        # unknowns, which would needlessly couple the blocks of a DC/transient run (e.g. an inverter array on one input)
        d.src_ac = arr("sa", self.source_ac, np.float64) if (small_signal and any(self.source_ac)) else None
        d.n_va_par = len(self.va_par)
        d.va_par = arr("vp", self.va_par, np.float64) if self.va_par else None
        d._keep = keep
        return d


def _wave_value_at_zero(w):
    if w.kind == SRC_DC:
        return w.par[0]
    if w.kind == SRC_PWL:
        return w.ys[0] if len(w.ys) else 0.0
    if w.kind == SRC_PULSE:
        return w.par[0]
    if w.kind == SRC_SIN:
        return w.par[0] + w.par[1] * math.sin(math.radians(w.par[5]))
    return 0.0
