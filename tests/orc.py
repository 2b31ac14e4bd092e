"""ctypes binding of oracle/liborc.so (the CPU restatement).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None

i64 = C.c_int64
dbl = C.c_double
dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ROOT, "oracle", "liborc.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("afesp_oracle.c", "afesp_oracle_so.c", "afesp_oracle_blas.c")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liborc.so"])
    L = C.CDLL(so)
    L.orc_neri.restype = i64
    L.orc_neri.argtypes = [i64]
    L.orc_ao2mo.argtypes = [i64, dp, dp, dp]
    L.orc_unpack_eri.argtypes = [i64, dp, dp]
    L.orc_pack_eri.argtypes = [i64, dp, dp]
    L.orc_mp2_energy.restype = dbl
    L.orc_mp2_energy.argtypes = [i64, i64, dp, dp]
    L.orc_cc_create.restype = C.c_void_p
    L.orc_cc_create.argtypes = [i64, i64, dp, dp, C.c_int]
    L.orc_cc_destroy.argtypes = [C.c_void_p]
    for f in ("orc_cc_intermediates", "orc_cc_amplitudes", "orc_cc_diis_save"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.orc_cc_diis_update.argtypes = [C.c_void_p]
    L.orc_cc_diis_update.restype = C.c_int
    L.orc_cc_energy.argtypes = [C.c_void_p, dbl, dbl]
    L.orc_cc_energy.restype = C.c_int
    L.orc_cc_solve.argtypes = [C.c_void_p, C.c_int, dbl, dbl, dp, dp]
    L.orc_cc_solve.restype = C.c_int
    L.orc_cc_get_energy.argtypes = [C.c_void_p]
    L.orc_cc_get_energy.restype = dbl
    L.orc_cc_get_rms.argtypes = [C.c_void_p]
    L.orc_cc_get_rms.restype = dbl
    L.orc_cc_t1.argtypes = [C.c_void_p]
    L.orc_cc_t1.restype = C.POINTER(dbl)
    L.orc_cc_t2.argtypes = [C.c_void_p]
    L.orc_cc_t2.restype = C.POINTER(dbl)
    L.orc_cc_field.argtypes = [C.c_void_p, C.c_int]
    L.orc_cc_field.restype = C.POINTER(dbl)
    L.orc_cc_t1_diagnostic.argtypes = [C.c_void_p, i64]
    L.orc_cc_t1_diagnostic.restype = dbl
    L.orc_ccsd_t.argtypes = [i64, i64, dp, dp, dp, dp, dp, dp, i64, i64, dp]
    L.orc_ccsd_t.restype = None
    L.orc_ccsd_t_cr.argtypes = [i64, i64, dp, dp, dp, dp, dp, dp, dp, dp, i64, i64, dp]
    L.orc_ccsd_t_cr.restype = None
    L.orc_cc_cr_intermediates.argtypes = [C.c_void_p]
    L.orc_cc_cr_intermediates.restype = None
    L.orc_cc_cr_field.argtypes = [C.c_void_p, C.c_int]
    L.orc_cc_cr_field.restype = C.POINTER(dbl)
    L.orc_gemm.argtypes = [C.c_int, C.c_int, i64, i64, i64, dbl, dp, dp, dbl, dp]
    L.orc_gemm.restype = None
    L.orc_permute4.argtypes = [C.POINTER(i64), C.c_char_p, dp, dp, C.c_int, dbl]
    L.orc_permute4.restype = None
    L.orc_linsolve.argtypes = [C.c_int, dp, dp]
    L.orc_linsolve.restype = C.c_int
    L.orc_build_fock.argtypes = [i64, dp, dp, dp, dp]
    L.orc_build_fock.restype = None
    # spin-orbital path (afesp_oracle_so.c)
    L.orc_so_create.restype = C.c_void_p
    L.orc_so_create.argtypes = [i64, i64, dp, dp, C.c_int]
    L.orc_so_destroy.argtypes = [C.c_void_p]
    for f in ("orc_so_iterate", "orc_so_diis_save"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.orc_so_diis_update.argtypes = [C.c_void_p]
    L.orc_so_diis_update.restype = C.c_int
    L.orc_so_energy.argtypes = [C.c_void_p, dbl, dbl]
    L.orc_so_energy.restype = C.c_int
    L.orc_so_solve.argtypes = [C.c_void_p, C.c_int, dbl, dbl, dp, dp]
    L.orc_so_solve.restype = C.c_int
    for f in ("orc_so_get_energy", "orc_so_get_rms", "orc_so_triples"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = dbl
    for f in ("orc_so_t1", "orc_so_t2"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.POINTER(dbl)
    L.orc_so_set_foo_as_published.argtypes = [C.c_void_p, C.c_int]
    L.orc_so_set_foo_as_published.restype = None
    L.orc_so_field.argtypes = [C.c_void_p, C.c_int]
    L.orc_so_field.restype = C.POINTER(dbl)
    L.orcb_load.argtypes = [C.c_char_p]
    L.orcb_ccsd_t.argtypes = [i64, i64, dp, dp, dp, dp, dp, dp, i64, i64, dp]
    L.orcb_gemm.argtypes = [i64, i64, i64, dbl, dp, dp, dbl, dp, C.c_int]
    L.orcb_ring_I_ovov.argtypes = [i64, i64, dp, dp, dp, i64, i64]
    L.orcb_ring_t2.argtypes = [i64, i64, dp, dp, dp, dp, dp, i64, i64]
    _LIB = L
    return L


def blas_lib():
    """liborc.so with the dgemm of the OpenBLAS that numpy bundles loaded into it (oracle/afesp_oracle_blas.c), or None if that
    library cannot be found -- the BLAS-backed CPU baseline of bench.py is then skipped, never replaced by something else."""
    import glob
    L = lib()
    cands = sorted(glob.glob(os.path.join(os.path.dirname(np.__file__), "..", "numpy.libs", "libscipy_openblas64_*.so")))
    for path in cands:
        if L.orcb_load(os.path.abspath(path).encode()) == 0:
            return L
    return None


FIELDS = {"v_oovv": 0, "v_ovov": 1, "v_vvov": 2, "v_oovo": 3, "v_oooo": 4, "v_vvvv": 5, "I_vo": 6, "I_vv": 7,
          "I_oo_p": 8, "I_oo": 9, "c_oovv": 10, "asym_t2": 11, "x_voov": 12, "I_oooo": 13, "I_ovov": 14,
          "I_voov": 15, "I_vovv_p": 16, "I_ooov_p": 17, "r1": 18, "r2": 19, "D1": 20, "D2": 21}


def field_shape(name, o, v):
    return {"v_oovv": (o, o, v, v), "v_ovov": (o, v, o, v), "v_vvov": (v, v, o, v), "v_oovo": (o, o, v, o),
            "v_oooo": (o, o, o, o), "v_vvvv": (v, v, v, v), "I_vo": (v, o), "I_vv": (v, v), "I_oo_p": (o, o),
            "I_oo": (o, o), "c_oovv": (o, o, v, v), "asym_t2": (o, o, v, v), "x_voov": (v, o, o, v),
            "I_oooo": (o, o, o, o), "I_ovov": (o, v, o, v), "I_voov": (v, o, o, v), "I_vovv_p": (v, o, v, v),
            "I_ooov_p": (o, o, o, v), "r1": (o, v), "r2": (o, o, v, v), "D1": (o, v), "D2": (o, o, v, v)}[name]


class OracleCC:
    """Thin owner of an orc_cc handle.  Arrays come back as Fortran-ordered numpy views/copies."""

    def __init__(self, o, v, eri_mo, e, diis_nerr=8):
        self.L = lib()
        self.o, self.v = int(o), int(v)
        self.h = self.L.orc_cc_create(o, v, np.ascontiguousarray(eri_mo), np.ascontiguousarray(e), diis_nerr)

    def close(self):
        if self.h:
            self.L.orc_cc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _view(self, ptr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape, order="F")

    def field(self, name):
        return self._view(self.L.orc_cc_field(self.h, FIELDS[name]), field_shape(name, self.o, self.v))

    @property
    def t1(self):
        return self._view(self.L.orc_cc_t1(self.h), (self.o, self.v))

    @property
    def t2(self):
        return self._view(self.L.orc_cc_t2(self.h), (self.o, self.o, self.v, self.v))

    def solve(self, maxiter, e_tol, t_tol):
        en = np.zeros(maxiter + 1)
        rm = np.zeros(maxiter + 1)
        nit = self.L.orc_cc_solve(self.h, maxiter, e_tol, t_tol, en, rm)
        return nit, en, rm

    @property
    def energy(self):
        return self.L.orc_cc_get_energy(self.h)

    def triples(self, e, t_begin=0, t_end=None):
        o, v = self.o, self.v
        out = np.zeros(4)
        if t_end is None:
            t_end = o ** 3
        f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
        self.L.orc_ccsd_t(o, v, np.ascontiguousarray(e), f(self.t1), f(self.t2), f(self.field("v_vvov")),
                          f(self.field("v_oovo")), f(self.field("v_oovv")), t_begin, t_end, out)
        return out


def _blas_methods():
    def triples_blas(self, e, t_begin=0, t_end=None):
        """The same four sums from the dgemm-per-term form (oracle/afesp_oracle_blas.c); None without numpy's OpenBLAS."""
        L = blas_lib()
        if L is None:
            return None
        o, v = self.o, self.v
        out = np.zeros(4)
        if t_end is None:
            t_end = o ** 3
        f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
        rc = L.orcb_ccsd_t(o, v, np.ascontiguousarray(e), f(self.t1), f(self.t2), f(self.field("v_vvov")),
                           f(self.field("v_oovo")), f(self.field("v_oovv")), t_begin, t_end, out)
        assert rc == 0, rc
        return out
    OracleCC.triples_blas = triples_blas


_blas_methods()


def _cr_methods():
    def cr_intermediates(self):
        self.L.orc_cc_cr_intermediates(self.h)
        o, v = self.o, self.v
        return (self._view(self.L.orc_cc_cr_field(self.h, 0), (v, o, v, v)),
                self._view(self.L.orc_cc_cr_field(self.h, 1), (o, o, o, v)))

    def triples_cr(self, e, t_begin=0, t_end=None):
        """-> out[6]: E[T], E(T), D[T], D(T), sum t_bar.M3, sum (t_bar+z_bar).M3 (call cr_intermediates first)"""
        o, v = self.o, self.v
        out = np.zeros(6)
        if t_end is None:
            t_end = o ** 3
        f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
        ipp = self._view(self.L.orc_cc_cr_field(self.h, 0), (v, o, v, v))
        ioo = self._view(self.L.orc_cc_cr_field(self.h, 1), (o, o, o, v))
        self.L.orc_ccsd_t_cr(o, v, np.ascontiguousarray(e), f(self.t1), f(self.t2), f(self.field("v_vvov")),
                             f(self.field("v_oovo")), f(self.field("v_oovv")), f(ipp), f(ioo), t_begin, t_end, out)
        return out
    OracleCC.cr_intermediates = cr_intermediates
    OracleCC.triples_cr = triples_cr


_cr_methods()


def ao2mo(n, Cmat, eri_packed):
    """Cmat is (MO, AO) as a 2-D numpy array; passed column-major like the Fortran array."""
    L = lib()
    out = np.zeros_like(eri_packed)
    L.orc_ao2mo(n, np.ascontiguousarray(Cmat.ravel(order="F")), np.ascontiguousarray(eri_packed), out)
    return out


def mp2_energy(n, o, eri_mo, e):
    return lib().orc_mp2_energy(n, o, np.ascontiguousarray(eri_mo), np.ascontiguousarray(e))


SO_FIELDS = {"F_vv": 0, "F_oo": 1, "F_ov": 2, "W_oooo": 3, "W_vvvv": 4, "W_ovvo": 5, "tau": 6, "tau_tilde": 7, "oovv": 8,
             "vvvv": 9}


class OracleSO:
    """Spin-orbital CCSD / CCSD(T) restatement (oracle/afesp_oracle_so.c).  o, v are spin-orbital counts."""

    def __init__(self, n, nel, eri_mo, e, diis_nerr=8, foo_as_published=False):
        self.L = lib()
        self.n, self.o, self.v = int(n), int(nel), int(2 * n - nel)
        self.h = self.L.orc_so_create(n, nel, np.ascontiguousarray(eri_mo), np.ascontiguousarray(e), diis_nerr)
        self.L.orc_so_set_foo_as_published(self.h, 1 if foo_as_published else 0)

    def close(self):
        if self.h:
            self.L.orc_so_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _view(self, ptr, shape):
        return np.ctypeslib.as_array(ptr, shape=(int(np.prod(shape)),)).reshape(shape, order="F")

    def field(self, name):
        o, v = self.o, self.v
        shape = {"F_vv": (v, v), "F_oo": (o, o), "F_ov": (o, v), "W_oooo": (o, o, o, o), "W_vvvv": (v, v, v, v),
                 "W_ovvo": (o, v, v, o), "tau": (o, o, v, v), "tau_tilde": (o, o, v, v), "oovv": (o, o, v, v),
                 "vvvv": (v, v, v, v)}[name]
        return self._view(self.L.orc_so_field(self.h, SO_FIELDS[name]), shape)

    @property
    def t1(self):
        return self._view(self.L.orc_so_t1(self.h), (self.o, self.v))

    @property
    def t2(self):
        return self._view(self.L.orc_so_t2(self.h), (self.o, self.o, self.v, self.v))

    def iterate(self):
        self.L.orc_so_iterate(self.h)

    def energy_step(self, e_tol, t_tol):
        conv = self.L.orc_so_energy(self.h, e_tol, t_tol)
        return self.L.orc_so_get_energy(self.h), self.L.orc_so_get_rms(self.h), bool(conv)

    def solve(self, maxiter, e_tol, t_tol):
        en = np.zeros(maxiter + 1)
        rm = np.zeros(maxiter + 1)
        nit = self.L.orc_so_solve(self.h, maxiter, e_tol, t_tol, en, rm)
        return nit, en, rm

    @property
    def energy(self):
        return self.L.orc_so_get_energy(self.h)

    def triples(self):
        return self.L.orc_so_triples(self.h)


def build_fock(n, eri, dens, hcore):
    """hf.f90:349-385; dens/hcore are (n,n) arrays, returned Fock matrix too."""
    L = lib()
    out = np.zeros(n * n)
    L.orc_build_fock(n, np.ascontiguousarray(eri), np.ascontiguousarray(dens.ravel(order="F")),
                     np.ascontiguousarray(hcore.ravel(order="F")), out)
    return out.reshape((n, n), order="F")
