"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from cedarsim_jl_amd.circuit import ChDcOpts, ChDesc, ChStats, ChTranOpts, dc_opts, tran_opts

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")
_pf64 = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def lib():
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    L.oracle_build.restype = C.c_void_p
    L.oracle_build.argtypes = [C.POINTER(ChDesc)]
    L.oracle_free.argtypes = [C.c_void_p]
    L.oracle_n_mna.argtypes = [C.c_void_p]
    L.oracle_n_mos.argtypes = [C.c_void_p]
    L.oracle_set_proxy.argtypes = [C.c_void_p, C.c_int]
    L.oracle_set_param.argtypes = [C.c_void_p, C.c_int, C.c_double]
    L.oracle_dc.argtypes = [C.c_void_p, C.POINTER(ChDcOpts), _pf64, C.POINTER(ChStats)]
    L.oracle_tran.restype = C.c_void_p
    L.oracle_tran.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(ChTranOpts)]
    L.oracle_result_n_times.restype = C.c_int64
    L.oracle_result_n_times.argtypes = [C.c_void_p]
    for f in ("oracle_result_times", "oracle_result_values", "oracle_result_final_state"):
        getattr(L, f).restype = _pf64
        getattr(L, f).argtypes = [C.c_void_p]
    L.oracle_result_status.argtypes = [C.c_void_p]
    L.oracle_result_stats.argtypes = [C.c_void_p, C.POINTER(ChStats)]
    L.oracle_result_free.argtypes = [C.c_void_p]
    L.oracle_ac.argtypes = [C.c_void_p, C.POINTER(ChDcOpts), C.c_int, _pf64, _pf64]
    L.oracle_noise.argtypes = [C.c_void_p, C.POINTER(ChDcOpts), C.c_int, C.c_int, _pf64, _pf64]
    L.oracle_va_eval.argtypes = [C.c_int, _pf64, _pf64, C.c_double, C.c_double, _pf64]
    L.oracle_va_noise.argtypes = [C.c_int, _pf64, _pf64, C.c_double, C.c_double, _pf64]
    L.oracle_va_opvars.argtypes = [C.c_int, _pf64, _pf64, C.c_double, C.c_double, _pf64]
    L.oracle_va_opvar_name.argtypes = [C.c_int, C.c_int]
    L.oracle_va_opvar_name.restype = C.c_char_p
    L.oracle_va_module_name.argtypes = [C.c_int]
    L.oracle_va_module_name.restype = C.c_char_p
    L.oracle_eval.argtypes = [C.c_void_p, _pf64, C.c_double, C.c_double, C.c_int, _pf64, _pf64, _pf64]
    L.oracle_mos_eval.argtypes = [C.c_void_p, _pf64, _pf64]
    L.oracle_mos_eval_values.argtypes = [C.c_void_p, _pf64, _pf64]
    L.oracle_source_value.restype = C.c_double
    L.oracle_source_value.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
    return L


def _p(a):
    return a.ctypes.data_as(_pf64)


class Oracle:
    def __init__(self, circuit):
        self.L = lib()
        self.circuit = circuit
        self._desc = circuit.to_desc()
        self.h = self.L.oracle_build(C.byref(self._desc))
        self.n = self.L.oracle_n_mna(self.h)

    def __del__(self):
        try:
            self.L.oracle_free(self.h)
        except Exception:
            pass

    def set_param(self, slot, value):
        rc = self.L.oracle_set_param(self.h, slot, float(value))
        assert rc == 0, rc

    def dc(self, opts=None):
        opts = opts or dc_opts()
        x = np.zeros(self.n)
        st = ChStats()
        rc = self.L.oracle_dc(self.h, C.byref(opts), _p(x), C.byref(st))
        return rc, x, st.asdict()

    def ac(self, freqs_hz, opts=None):
        opts = opts or dc_opts()
        f = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        out = np.zeros((len(f), self.n, 2))
        rc = self.L.oracle_ac(self.h, C.byref(opts), len(f), _p(f), _p(out))
        return rc, out[..., 0] + 1j * out[..., 1]

    def noise(self, out_mna, freqs_hz, opts=None):
        opts = opts or dc_opts()
        f = np.ascontiguousarray(freqs_hz, dtype=np.float64)
        out = np.zeros(len(f))
        rc = self.L.oracle_noise(self.h, C.byref(opts), int(out_mna), len(f), _p(f), _p(out))
        return rc, out

    def set_proxy(self, on=True):
        """Transients run the "reference-like" cost proxy: finite-difference Jacobian (n+1 residuals), reused like IDA's
        modified Newton.  Off by default: the parity oracle uses exact dual-number Jacobians."""
        self.L.oracle_set_proxy(self.h, int(bool(on)))

    def tran(self, t0, t1, opts=None):
        opts = opts or tran_opts()
        r = self.L.oracle_tran(self.h, t0, t1, C.byref(opts))
        try:
            nt = self.L.oracle_result_n_times(r)
            nobs = len(self.circuit.obs)
            t = np.ctypeslib.as_array(self.L.oracle_result_times(r), (nt,)).copy() if nt else np.zeros(0)
            v = np.ctypeslib.as_array(self.L.oracle_result_values(r), (nobs, nt)).copy() if nt and nobs else np.zeros((nobs, nt))
            pf = self.L.oracle_result_final_state(r)   # NULL when the operating point already failed: no state to report
            xf = np.ctypeslib.as_array(pf, (self.n,)).copy() if pf else np.full(self.n, np.nan)
            st = ChStats()
            self.L.oracle_result_stats(r, C.byref(st))
            return self.L.oracle_result_status(r), t, v, xf, st.asdict()
        finally:
            self.L.oracle_result_free(r)

    def eval(self, x, t=0.0, alpha0=0.0, mode=1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        F, Q, J = np.zeros(self.n), np.zeros(self.n), np.zeros((self.n, self.n))
        rc = self.L.oracle_eval(self.h, _p(x), t, alpha0, mode, _p(F), _p(Q), _p(J))
        assert rc == 0, rc
        return F, Q, J

    def mos_eval(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        nm = self.L.oracle_n_mos(self.h)
        out = np.zeros((nm, 40))
        rc = self.L.oracle_mos_eval(self.h, _p(v), _p(out))
        assert rc == 0, rc
        return out

    def mos_eval_values(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        nm = self.L.oracle_n_mos(self.h)
        out = np.zeros((nm, 8))
        rc = self.L.oracle_mos_eval_values(self.h, _p(v), _p(out))
        assert rc == 0, rc
        return out

    def source_value(self, src, t, mode=1):
        return self.L.oracle_source_value(self.h, src, t, mode)
