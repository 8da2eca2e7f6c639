"""ctypes binding of libcedarhip.so — the C-ABI of include/cedarhip.h.

The library is built in-tree (cedarsim.jl_amd/lib/libcedarhip.so) by `__graft_entry__.build()` or
`make -C cedarsim.jl_amd/csrc`.  There is no CPU fallback: if the library is missing, or no HIP
device is present, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

from .circuit import (ChDcOpts, ChDesc, ChInfo, ChStats, ChTranOpts, CedarError, RETCODES, dc_opts, tran_opts)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("CEDARHIP_LIB", "libcedarhip.so"))  # CEDARHIP_LIB: diagnostic builds
_pf64 = C.POINTER(C.c_double)
_pi32 = C.POINTER(C.c_int32)

EXPORTS = [
    "ch_dc_opts_default", "ch_tran_opts_default", "ch_create", "ch_destroy", "ch_last_error", "ch_circuit_build",
    "ch_circuit_free", "ch_circuit_info", "ch_circuit_maps", "ch_set_samples", "ch_set_params", "ch_dc", "ch_tran",
    "ch_result_n_times", "ch_result_times", "ch_result_dense_points", "ch_result_device_values", "ch_result_values", "ch_result_final_state", "ch_result_stats",
    "ch_result_status", "ch_result_free", "ch_eval", "ch_ac", "ch_noise", "ch_mos_eval", "ch_mos_eval_quad", "ch_bsim4_npar", "ch_bsim4_param_name",
    "ch_bsim4_param_ignored", "ch_version", "ch_bench_triad", "ch_bench_fp64", "ch_va_n_modules", "ch_va_find", "ch_va_module_name", "ch_va_module_info",
    "ch_va_node_name", "ch_va_param_name", "ch_va_eval", "ch_va_n_opvars", "ch_va_opvar_name", "ch_va_opvars", "ch_debug_poison_lds", "ch_debug_math",
]

_lib = None


def load_library():
    """Load libcedarhip.so and declare every prototype.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libcedarhip.so is not built (%s missing): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C cedarsim.jl_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.ch_create.restype = vp
    L.ch_create.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.ch_destroy.argtypes = [vp]
    L.ch_last_error.restype = C.c_char_p
    L.ch_last_error.argtypes = [vp]
    L.ch_circuit_build.restype = vp
    L.ch_circuit_build.argtypes = [vp, C.POINTER(ChDesc)]
    L.ch_circuit_free.argtypes = [vp]
    L.ch_circuit_info.argtypes = [vp, C.POINTER(ChInfo)]
    L.ch_circuit_maps.argtypes = [vp, _pi32, _pi32, _pi32]
    L.ch_set_samples.argtypes = [vp, C.c_int32]
    L.ch_set_params.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, _pi32, _pf64]
    L.ch_dc.argtypes = [vp, C.POINTER(ChDcOpts), _pf64, _pi32, C.POINTER(ChStats)]
    L.ch_tran.argtypes = [vp, C.c_double, C.c_double, C.POINTER(ChTranOpts), C.POINTER(vp)]
    L.ch_result_n_times.restype = C.c_int64
    L.ch_result_n_times.argtypes = [vp]
    for f in ("ch_result_times", "ch_result_values", "ch_result_final_state"):
        getattr(L, f).restype = _pf64
        getattr(L, f).argtypes = [vp]
    L.ch_result_dense_points.restype = _pi32
    L.ch_result_dense_points.argtypes = [vp]
    L.ch_result_device_values.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.ch_result_stats.argtypes = [vp, C.POINTER(ChStats)]
    L.ch_result_status.argtypes = [vp]
    L.ch_result_free.argtypes = [vp]
    L.ch_eval.argtypes = [vp, C.c_int32, _pf64, C.c_double, C.c_double, C.c_int32, _pf64, _pf64, _pf64]
    L.ch_mos_eval.argtypes = [vp, C.c_int32, _pf64, _pf64]
    L.ch_mos_eval_quad.argtypes = [vp, C.c_int32, _pf64, _pf64]
    L.ch_bsim4_npar.restype = C.c_int32
    L.ch_bsim4_param_name.restype = C.c_char_p
    L.ch_bsim4_param_name.argtypes = [C.c_int32]
    L.ch_bsim4_param_ignored.argtypes = [C.c_char_p]
    L.ch_ac.argtypes = [vp, C.POINTER(ChDcOpts), C.c_int32, _pf64, _pf64, C.POINTER(ChStats)]
    L.ch_noise.argtypes = [vp, C.POINTER(ChDcOpts), C.c_int32, C.c_int32, C.c_int32, _pf64, _pf64, C.POINTER(ChStats)]
    L.ch_va_find.argtypes = [C.c_char_p]
    L.ch_va_module_name.argtypes = [C.c_int32]
    L.ch_va_module_name.restype = C.c_char_p
    L.ch_va_module_info.argtypes = [C.c_int32, _pi32, _pi32, _pi32]
    L.ch_va_node_name.argtypes = [C.c_int32, C.c_int32]
    L.ch_va_node_name.restype = C.c_char_p
    L.ch_va_param_name.argtypes = [C.c_int32, C.c_int32]
    L.ch_va_param_name.restype = C.c_char_p
    L.ch_va_eval.argtypes = [vp, C.c_int32, _pf64, _pf64, C.c_double, C.c_double, _pf64]
    L.ch_va_n_opvars.argtypes = [C.c_int32]
    L.ch_va_opvar_name.argtypes = [C.c_int32, C.c_int32]
    L.ch_va_opvar_name.restype = C.c_char_p
    L.ch_va_opvars.argtypes = [vp, C.c_int32, _pf64, _pf64, C.c_double, C.c_double, _pf64]
    L.ch_version.restype = C.c_char_p
    L.ch_bench_triad.argtypes = [vp, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
    L.ch_bench_fp64.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
    L.ch_dc_opts_default.argtypes = [C.POINTER(ChDcOpts)]
    L.ch_tran_opts_default.argtypes = [C.POINTER(ChTranOpts)]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(_pf64)


class Context:
    """One per GPU (one process per GPU)."""

    def __init__(self, device_id=0):
        self.L = load_library()
        buf = C.create_string_buffer(512)
        self.h = self.L.ch_create(device_id, buf, 512)
        if not self.h:
            raise RuntimeError("ch_create failed: %s" % buf.value.decode())
        self.device_id = device_id

    def last_error(self):
        return self.L.ch_last_error(self.h).decode()

    def fp64_tflops(self, iters=3):
        """Measured vector fp64 FMA peak of this GPU in TFLOP/s (measurement utility)."""
        out = C.c_double(0.0)
        rc = self.L.ch_bench_fp64(self.h, int(iters), C.byref(out))
        if rc != 0:
            raise RuntimeError("ch_bench_fp64 failed: %s" % self.last_error())
        return out.value

    def va_eval(self, module_id, par_and_given, v_nodes, temperature_k=300.15, gmin=1e-12):
        """One compiled Verilog-A module on the GPU: the 144-double wide stamp [I(8)|Q(8)|G(8x8)|C(8x8)]."""
        p = np.ascontiguousarray(par_and_given, dtype=np.float64)
        v = np.zeros(8)
        v[:len(v_nodes)] = v_nodes
        out = np.zeros(144)
        rc = self.L.ch_va_eval(self.h, int(module_id), _p(p), _p(v), float(temperature_k), float(gmin), _p(out))
        if rc != 0:
            raise RuntimeError("ch_va_eval failed: %s" % self.last_error())
        return out

    def va_opvars(self, module_id, par_and_given, v_nodes, temperature_k=300.15, gmin=1e-12):
        """{name: value} of the (* desc *) observables of a compiled module at the given node voltages (on the GPU)."""
        n = self.L.ch_va_n_opvars(int(module_id))
        p = np.ascontiguousarray(par_and_given, dtype=np.float64)
        v = np.zeros(8)
        v[:len(v_nodes)] = v_nodes
        out = np.zeros(max(1, n))
        rc = self.L.ch_va_opvars(self.h, int(module_id), _p(p), _p(v), float(temperature_k), float(gmin), _p(out))
        if rc != 0:
            raise RuntimeError("ch_va_opvars failed: %s" % self.last_error())
        return {self.L.ch_va_opvar_name(int(module_id), k).decode(): float(out[k]) for k in range(n)}

    def debug_math(self, which, x):
        """Test hook: the device's own exp (which=0), ln (1) and the BSIM4 code's ln (2) over a vector (ch_debug_math)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        pf = C.POINTER(C.c_double)
        self.L.ch_debug_math.argtypes = [C.c_void_p, C.c_int32, C.c_int32, pf, pf]
        rc = self.L.ch_debug_math(self.h, int(which), len(x), x.ctypes.data_as(pf), y.ctypes.data_as(pf))
        if rc != 0:
            raise RuntimeError("ch_debug_math failed: %s" % self.last_error())
        return y

    def poison_lds(self):
        """Test hook: leave every CU's LDS full of garbage (ch_debug_poison_lds)."""
        self.L.ch_debug_poison_lds.argtypes = [C.c_void_p]
        rc = self.L.ch_debug_poison_lds(self.h)
        if rc != 0:
            raise RuntimeError("ch_debug_poison_lds failed: %s" % self.last_error())

    def triad_gbps(self, n_doubles=1 << 27, iters=5):
        """Measured STREAM-triad bandwidth of this GPU in GB/s (measurement utility)."""
        out = C.c_double(0.0)
        rc = self.L.ch_bench_triad(self.h, int(n_doubles), int(iters), C.byref(out))
        if rc != 0:
            raise RuntimeError("ch_bench_triad failed: %s" % self.last_error())
        return out.value

    def close(self):
        if getattr(self, "h", None):
            self.L.ch_destroy(self.h)
            self.h = None


_default_ctx = {}


def default_context(device_id=None):
    if device_id is None:
        device_id = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("CEDARHIP_USE_LOCAL_RANK", "1") == "1" else 0
        try:
            import ctypes.util  # noqa: F401
        except Exception:  # noqa: BLE001
            pass
    if device_id not in _default_ctx:
        _default_ctx[device_id] = Context(device_id)
    return _default_ctx[device_id]


class DeviceRows:
    """Result rows of the last transient still in HBM, [n_obs][n_times][n_samples] fp64, exposed through `__cuda_array_interface__`
    (zero-copy: `torch.as_tensor(rows, device="cuda")`).  Owned by the engine circuit: valid until its next call; `owner` keeps it alive."""

    def __init__(self, ptr, shape, owner):
        self.ptr, self.shape, self.owner = int(ptr), tuple(int(x) for x in shape), owner
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 2, "strides": None}


class EngineCircuit:
    """A circuit resident on the GPU: structure analysed once, parameters per sample."""

    def __init__(self, circuit, ctx=None, small_signal=False):
        """small_signal=True keeps the sources' AC magnitudes (needed by .ac(); .noise() works either way)."""
        self.ctx = ctx or default_context()
        self.L = self.ctx.L
        self.circuit = circuit
        self._desc = circuit.to_desc(small_signal=small_signal)
        for nm, (mod, _) in getattr(circuit, "va_instances", {}).items():
            mid = circuit.dev_ipar[circuit.dev_names.index(nm)][0]
            got = self.L.ch_va_module_name(mid)
            if got is None or got.decode() != mod.name:
                raise CedarError("the Verilog-A model library (lib/va_modules.json) and libcedarhip.so are out of step: rebuild both")
        self.h = self.L.ch_circuit_build(self.ctx.h, C.byref(self._desc))
        if not self.h:
            raise CedarError("ch_circuit_build failed: %s" % self.ctx.last_error())
        self.n_mna = circuit.n_mna
        self.n_samples = 1
        # the library writes [n_samples][n_mna] doubles into buffers this class allocates: the two sides must agree on n_mna
        lib_n_mna = self.info()["n_mna"]
        if lib_n_mna != self.n_mna:
            self.L.ch_circuit_free(self.h)
            self.h = None
            raise CedarError("host mirror and engine disagree on the MNA size (%d vs %d)" % (self.n_mna, lib_n_mna))

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.L.ch_circuit_free(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def info(self):
        i = ChInfo()
        self.L.ch_circuit_info(self.h, C.byref(i))
        return i.asdict()

    def maps(self):
        nn, nb = self.circuit.n_nodes, len(self.circuit.branch_devices)
        nu = np.zeros(nn + 1, np.int32)
        nk = np.zeros(nn + 1, np.int32)
        bu = np.zeros(max(1, nb), np.int32)
        self.L.ch_circuit_maps(self.h, nu.ctypes.data_as(_pi32), nk.ctypes.data_as(_pi32), bu.ctypes.data_as(_pi32))
        return nu, nk, bu[:nb]

    def _check(self, rc, what):
        if rc not in (0,):
            raise CedarError("%s failed (%s): %s" % (what, RETCODES.get(rc, rc), self.ctx.last_error()))

    def set_samples(self, n):
        self._check(self.L.ch_set_samples(self.h, int(n)), "ch_set_samples")
        self.n_samples = int(n)

    def set_params(self, slot_ids, values, lo=0, hi=None):
        """values[slot][sample] for samples lo..hi."""
        hi = self.n_samples if hi is None else hi
        ids = np.ascontiguousarray(slot_ids, dtype=np.int32)
        vals = np.ascontiguousarray(values, dtype=np.float64).reshape(len(ids), hi - lo)
        self._check(self.L.ch_set_params(self.h, lo, hi, len(ids), ids.ctypes.data_as(_pi32), _p(vals)), "ch_set_params")

    def dc(self, opts=None, check=False):
        opts = opts or dc_opts()
        x = np.zeros((self.n_samples, self.n_mna))
        status = np.zeros(self.n_samples, np.int32)
        st = ChStats()
        rc = self.L.ch_dc(self.h, C.byref(opts), _p(x), status.ctypes.data_as(_pi32), C.byref(st))
        if check:
            self._check(rc, "ch_dc")
        return rc, x, status, st.asdict()

    def tran(self, t0, t1, opts=None):
        opts = opts or tran_opts()
        r = C.c_void_p()
        rc = self.L.ch_tran(self.h, float(t0), float(t1), C.byref(opts), C.byref(r))
        if not r:
            raise CedarError("ch_tran failed: %s" % self.ctx.last_error())
        try:
            nt = self.L.ch_result_n_times(r)
            nobs = len(self.circuit.obs)
            S = self.n_samples
            t = np.ctypeslib.as_array(self.L.ch_result_times(r), (nt,)).copy() if nt else np.zeros(0)
            v = np.ctypeslib.as_array(self.L.ch_result_values(r), (nobs, nt, S)).copy() if nt and nobs else np.zeros((nobs, nt, S))
            fs = self.L.ch_result_final_state(r)
            xf = np.ctypeslib.as_array(fs, (S, self.n_mna)).copy() if fs and nt else np.zeros((S, self.n_mna))
            st = ChStats()
            self.L.ch_result_stats(r, C.byref(st))
            sd = st.asdict()
            dp = self.L.ch_result_dense_points(r)
            # per saved row: how many newest rows the step's dense-output polynomial runs through (0: none; api.Solution.__call__)
            sd["dense_points"] = np.ctypeslib.as_array(dp, (nt,)).copy() if (dp and nt) else None
            # the same rows still in HBM (ch_result_device_values): a view for torch.as_tensor(..., device="cuda"), valid until the next call on this circuit
            dptr, dn = C.c_void_p(), C.c_int64(0)
            ok = self.L.ch_result_device_values(r, C.byref(dptr), C.byref(dn)) == 0 and dptr.value and dn.value == nobs * nt * S
            sd["device_rows"] = DeviceRows(dptr.value, (nobs, nt, S), self) if ok else None
            return rc, t, v, xf, sd
        finally:
            self.L.ch_result_free(r)

    def ac(self, freqs_hz, opts=None):
        """Small-signal sweep: complex MNA phasors [S][n_freq][n_mna] for unit excitation of the sources' `ac`."""
        opts = opts or dc_opts()
        f = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        out = np.zeros((self.n_samples, len(f), self.n_mna, 2))
        st = ChStats()
        rc = self.L.ch_ac(self.h, C.byref(opts), len(f), _p(f), _p(out), C.byref(st))
        return rc, out[..., 0] + 1j * out[..., 1], st.asdict()

    def noise(self, out_kind, out_index, freqs_hz, opts=None):
        """Output-noise PSD [S][n_freq] at a node (out_kind 0, node id) or branch current (1, device index)."""
        opts = opts or dc_opts()
        f = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        out = np.zeros((self.n_samples, len(f)))
        st = ChStats()
        rc = self.L.ch_noise(self.h, C.byref(opts), int(out_kind), int(out_index), len(f), _p(f), _p(out), C.byref(st))
        return rc, out, st.asdict()

    def eval(self, x_mna, t=0.0, alpha0=0.0, mode=1, sample=0):
        x = np.ascontiguousarray(x_mna, dtype=np.float64)
        n = self.n_mna
        F, Q, J = np.zeros(n), np.zeros(n), np.zeros((n, n))
        self._check(self.L.ch_eval(self.h, sample, _p(x), t, alpha0, mode, _p(F), _p(Q), _p(J)), "ch_eval")
        return F, Q, J

    def mos_eval(self, v, sample=0, quad=False):
        v = np.ascontiguousarray(v, dtype=np.float64)
        nm = self.info()["n_mos"]
        out = np.zeros((nm, 40))
        fn = self.L.ch_mos_eval_quad if quad else self.L.ch_mos_eval
        self._check(fn(self.h, sample, _p(v), _p(out)), "ch_mos_eval")
        return out
