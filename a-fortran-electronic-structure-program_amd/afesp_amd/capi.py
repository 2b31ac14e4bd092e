"""ctypes binding of libafesp_hip.so (include/afesp.h).  Python-side mirror of the three calls the reference's driver
makes into the hot path (src/main.F90:98,105,112): `do_mp2_spatial`, `do_ccsd_spatial`, `do_ccsd_t_spatial`.

There is no CPU fallback: if the shared library is missing, or no GPU is visible, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFESP_LIBRARY: a differently built libafesp_hip (A/B kernel measurements inside one GPU session, tools/ab_gemm.py)
LIB_PATH = os.environ.get("AFESP_LIBRARY") or os.path.join(os.path.dirname(_HERE), "csrc", "libafesp_hip.so")

i64 = C.c_int64
dbl = C.c_double
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_opt = C.c_void_p

EXPORTS = [
    "afesp_ctx_create", "afesp_ctx_destroy", "afesp_last_error", "afesp_version", "afesp_neri", "afesp_ao2mo_mp2",
    "afesp_ccsd_init", "afesp_ccsd_iterate", "afesp_ccsd_energy", "afesp_ccsd_diis", "afesp_ccsd_solve",
    "afesp_ccsd_get_amplitudes", "afesp_ccsd_set_amplitudes", "afesp_ccsd_get_tensor", "afesp_ccsd_update_intermediates",
    "afesp_ccsd_update_amplitudes", "afesp_ccsd_t_ntriples", "afesp_ccsd_t", "afesp_ccsd_t_shard_bounds", "afesp_gemm", "afesp_permute4",
    "afesp_contract", "afesp_synthetic_init", "afesp_time_pp_ladder", "afesp_bench_contract", "afesp_set_tuning", "afesp_bench_stream", "afesp_profile", "afesp_ccsd_cr_intermediates", "afesp_ccsd_t_cr",
    "afesp_ccsd_so_init", "afesp_ccsd_so_energy", "afesp_ccsd_so_iterate", "afesp_ccsd_so_diis", "afesp_ccsd_so_get_amplitudes",
    "afesp_ccsd_so_set_amplitudes", "afesp_ccsd_so_get_tensor", "afesp_ccsd_so_t_ntriples", "afesp_ccsd_so_t",
    "afesp_read_eri_text", "afesp_write_fcidump", "afesp_set_eri", "afesp_build_fock", "afesp_ccsd_t_plain",
    "afesp_synthetic_ao", "afesp_ccsd_pp_ladder_flop", "afesp_ccsd_iteration_flop",
    "afesp_device_count", "afesp_comm_unique_id", "afesp_comm_init", "afesp_comm_destroy", "afesp_allreduce_sum",
    "afesp_ccsd_t_block_size", "afesp_test_inject", "afesp_ccsd_is_split", "afesp_ccsd_set_split", "afesp_ccsd_set_fused", "afesp_ccsd_iteration_launches", "afesp_debug_stamps", "afesp_launch_counts", "afesp_first_use_count", "afesp_test_ring_path", "afesp_arena_stats",
]
COMM_RCCL, COMM_HOST = 0, 1


class AfespError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen the HIP library.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AfespError(f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                         "(there is no CPU fallback for the accelerated path)")
    L = C.CDLL(LIB_PATH)
    L.afesp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.afesp_ctx_destroy.argtypes = [C.c_void_p]
    L.afesp_ctx_destroy.restype = None
    L.afesp_last_error.argtypes = [C.c_void_p]
    L.afesp_last_error.restype = C.c_char_p
    L.afesp_neri.argtypes = [i64]
    L.afesp_neri.restype = i64
    L.afesp_ao2mo_mp2.argtypes = [C.c_void_p, i64, i64, _dp, _dp, _opt, _opt, C.POINTER(dbl)]
    L.afesp_ccsd_init.argtypes = [C.c_void_p, i64, i64, _opt, _dp, C.c_int]
    L.afesp_ccsd_iterate.argtypes = [C.c_void_p, dbl, dbl, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(C.c_int)]
    L.afesp_ccsd_energy.argtypes = [C.c_void_p, dbl, dbl, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(C.c_int)]
    L.afesp_ccsd_diis.argtypes = [C.c_void_p]
    L.afesp_ccsd_solve.argtypes = [C.c_void_p, C.c_int, dbl, dbl, _dp, _dp, C.POINTER(C.c_int)]
    L.afesp_ccsd_get_amplitudes.argtypes = [C.c_void_p, _dp, _dp]
    L.afesp_ccsd_set_amplitudes.argtypes = [C.c_void_p, _dp, _dp]
    L.afesp_ccsd_get_tensor.argtypes = [C.c_void_p, C.c_char_p, _dp, i64]
    L.afesp_ccsd_update_intermediates.argtypes = [C.c_void_p]
    L.afesp_ccsd_update_amplitudes.argtypes = [C.c_void_p]
    L.afesp_ccsd_t_ntriples.argtypes = [i64]
    L.afesp_ccsd_t_ntriples.restype = i64
    L.afesp_ccsd_t_shard_bounds.argtypes = [C.c_void_p, i64, i64, C.c_int, C.c_int, C.POINTER(i64)]
    L.afesp_ccsd_t.argtypes = [C.c_void_p, i64, i64, _dp]
    L.afesp_ccsd_t_cr.argtypes = [C.c_void_p, i64, i64, _dp]
    L.afesp_ccsd_t_plain.argtypes = [C.c_void_p, i64, i64, _dp]
    L.afesp_ccsd_cr_intermediates.argtypes = [C.c_void_p]
    L.afesp_gemm.argtypes = [C.c_void_p, C.c_char, C.c_char, i64, i64, i64, dbl, _dp, _dp, dbl, _dp]
    L.afesp_permute4.argtypes = [C.c_void_p, C.POINTER(i64), C.c_char_p, _dp, _dp, C.c_int, dbl]
    L.afesp_contract.argtypes = [C.c_void_p, dbl, _dp, C.c_char_p, C.POINTER(i64), _dp, C.c_char_p, C.POINTER(i64), dbl,
                                 _dp, C.c_char_p, C.POINTER(i64), C.c_int, C.c_int, C.c_int]
    L.afesp_synthetic_init.argtypes = [C.c_void_p, i64, i64, dbl, C.c_uint64, C.c_int]
    L.afesp_time_pp_ladder.argtypes = [C.c_void_p, C.c_int, C.POINTER(dbl)]
    L.afesp_synthetic_ao.argtypes = [C.c_void_p, i64, dbl, C.c_uint64]
    L.afesp_ccsd_pp_ladder_flop.argtypes = [C.c_void_p, C.POINTER(dbl)]
    L.afesp_ccsd_iteration_flop.argtypes = [C.c_void_p, C.POINTER(dbl)]
    L.afesp_bench_contract.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(i64), C.c_char_p, C.POINTER(i64), C.c_char_p,
                                       C.POINTER(i64), C.c_int, C.POINTER(dbl)]
    L.afesp_set_tuning.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.afesp_bench_stream.argtypes = [C.c_void_p, i64, C.c_int, C.POINTER(dbl)]
    L.afesp_profile.argtypes = [C.c_void_p, C.c_int, _dp]
    L.afesp_ccsd_so_init.argtypes = [C.c_void_p, i64, i64, _opt, _dp, C.c_int, C.c_int]
    L.afesp_ccsd_so_iterate.argtypes = [C.c_void_p, dbl, dbl, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(C.c_int)]
    L.afesp_ccsd_so_energy.argtypes = [C.c_void_p, dbl, dbl, C.POINTER(dbl), C.POINTER(dbl), C.POINTER(C.c_int)]
    L.afesp_ccsd_so_diis.argtypes = [C.c_void_p]
    L.afesp_ccsd_so_get_amplitudes.argtypes = [C.c_void_p, _dp, _dp]
    L.afesp_ccsd_so_set_amplitudes.argtypes = [C.c_void_p, _dp, _dp]
    L.afesp_ccsd_so_get_tensor.argtypes = [C.c_void_p, C.c_char_p, _dp, i64]
    L.afesp_ccsd_so_t_ntriples.argtypes = [i64]
    L.afesp_ccsd_so_t_ntriples.restype = i64
    L.afesp_ccsd_so_t.argtypes = [C.c_void_p, i64, i64, C.POINTER(dbl)]
    L.afesp_read_eri_text.argtypes = [C.c_void_p, C.c_char_p, i64, _opt, C.POINTER(i64)]
    L.afesp_write_fcidump.argtypes = [C.c_void_p, C.c_char_p, i64, C.POINTER(i64)]
    L.afesp_set_eri.argtypes = [C.c_void_p, i64, _dp]
    L.afesp_build_fock.argtypes = [C.c_void_p, i64, _dp, _dp, _dp]
    L.afesp_device_count.argtypes = []
    L.afesp_comm_unique_id.argtypes = [C.c_char_p]
    L.afesp_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    L.afesp_comm_destroy.argtypes = [C.c_void_p]
    L.afesp_allreduce_sum.argtypes = [C.c_void_p, _dp, i64]
    L.afesp_ccsd_t_block_size.argtypes = [C.c_void_p, i64, i64, C.c_int, C.POINTER(C.c_int)]
    L.afesp_test_inject.argtypes = [C.c_void_p, C.c_int]
    L.afesp_arena_stats.argtypes = [C.c_void_p, _dp]
    L.afesp_ccsd_is_split.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.afesp_ccsd_set_split.argtypes = [C.c_void_p, C.c_int]
    L.afesp_ccsd_set_fused.argtypes = [C.c_void_p, C.c_int]
    L.afesp_ccsd_iteration_launches.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.afesp_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    L.afesp_launch_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.afesp_first_use_count.argtypes = []
    L.afesp_first_use_count.restype = C.c_uint64
    L.afesp_test_ring_path.argtypes = [C.c_int64, C.c_int64]
    _lib = L
    return L


def _f(a):
    """Fortran-order flat copy of an array (what the C-ABI expects)."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))


TENSOR_SHAPES = {
    "v_oovv": "oovv", "v_ovov": "ovov", "v_vvov": "vvov", "v_oovo": "oovo", "v_oooo": "oooo", "v_vvvv": "vvvv",
    "I_vo": "vo", "I_vv": "vv", "I_oo_p": "oo", "I_oo": "oo", "c_oovv": "oovv", "asym_t2": "oovv", "x_voov": "voov",
    "I_oooo": "oooo", "I_ovov": "ovov", "I_voov": "voov", "I_vovv_p": "vovv", "I_ooov_p": "ooov", "r1": "ov",
    "r2": "oovv", "D1": "ov", "D2": "oovv", "t1": "ov", "t2": "oovv",
}


def device_count():
    """Number of HIP devices this process sees (afesp_device_count; 0 without a GPU)."""
    return int(load_library().afesp_device_count())


def first_use_count():
    """Launch sites of this process that resolved their kernel under the first-use lock so far (afesp_first_use_count; test hook)."""
    return int(load_library().afesp_first_use_count())


class Engine:
    """One GPU context.  Method names follow the reference routines they replace."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.afesp_ctx_create(device, C.byref(h))
        if rc != 0:
            raise AfespError(f"afesp_ctx_create(device={device}) failed with status {rc} "
                             "(no usable MI355X/HIP device; the accelerated path has no CPU fallback)")
        self.h = h
        self.o = self.v = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.afesp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise AfespError(f"status {rc}: {self.L.afesp_last_error(self.h).decode()}")

    # ---- src/mp2.f90:261-449
    def do_mp2_spatial(self, nbasis, nocc, canon_coeff, canon_levels, eri_packed, want_eri_mo=True):
        e2 = dbl(0.0)
        out = np.zeros(self.L.afesp_neri(nbasis)) if want_eri_mo else None
        src = None   # None: the AO integrals read_eri_text left on the device
        if eri_packed is not None:
            eri_packed = np.ascontiguousarray(eri_packed, dtype=np.float64)
            src = eri_packed.ctypes.data_as(C.c_void_p)
        self._chk(self.L.afesp_ao2mo_mp2(self.h, nbasis, nocc, _f(canon_coeff), _f(canon_levels), src,
                                         out.ctypes.data_as(C.c_void_p) if out is not None else None, C.byref(e2)))
        return e2.value, out

    # ---- src/ccsd.f90:279-402
    def ccsd_init(self, nocc, nvirt, canon_levels, eri_mo_packed=None, diis_n_errmat=8):
        self.o, self.v = int(nocc), int(nvirt)
        ptr = None
        if eri_mo_packed is not None:
            eri_mo_packed = np.ascontiguousarray(eri_mo_packed, dtype=np.float64)
            ptr = eri_mo_packed.ctypes.data_as(C.c_void_p)
        self._chk(self.L.afesp_ccsd_init(self.h, nocc, nvirt, ptr, _f(canon_levels), diis_n_errmat))

    def synthetic_init(self, nocc, nvirt, scale=0.02, seed=12345, diis_n_errmat=8):
        self.o, self.v = int(nocc), int(nvirt)
        self._chk(self.L.afesp_synthetic_init(self.h, nocc, nvirt, scale, seed, diis_n_errmat))

    def ccsd_energy(self, e_tol=1e-6, t_tol=1e-7):
        e, r, c = dbl(), dbl(), C.c_int()
        self._chk(self.L.afesp_ccsd_energy(self.h, e_tol, t_tol, C.byref(e), C.byref(r), C.byref(c)))
        return e.value, r.value, bool(c.value)

    def ccsd_iterate(self, e_tol=1e-6, t_tol=1e-7):
        e, r, c = dbl(), dbl(), C.c_int()
        self._chk(self.L.afesp_ccsd_iterate(self.h, e_tol, t_tol, C.byref(e), C.byref(r), C.byref(c)))
        return e.value, r.value, bool(c.value)

    def ccsd_diis(self):
        self._chk(self.L.afesp_ccsd_diis(self.h))

    def update_intermediates(self):
        self._chk(self.L.afesp_ccsd_update_intermediates(self.h))

    def update_amplitudes(self):
        self._chk(self.L.afesp_ccsd_update_amplitudes(self.h))

    def do_ccsd_spatial(self, maxiter=50, e_tol=1e-6, t_tol=1e-7):
        en = np.zeros(maxiter + 1)
        rm = np.zeros(maxiter + 1)
        nit = C.c_int()
        self._chk(self.L.afesp_ccsd_solve(self.h, maxiter, e_tol, t_tol, en, rm, C.byref(nit)))
        return nit.value, en, rm

    def amplitudes(self):
        o, v = self.o, self.v
        t1 = np.zeros(o * v)
        t2 = np.zeros(o * o * v * v)
        self._chk(self.L.afesp_ccsd_get_amplitudes(self.h, t1, t2))
        return t1.reshape((o, v), order="F"), t2.reshape((o, o, v, v), order="F")

    def set_amplitudes(self, t1, t2):
        self._chk(self.L.afesp_ccsd_set_amplitudes(self.h, _f(t1), _f(t2)))

    def tensor(self, name):
        dims = tuple(self.o if ch == "o" else self.v for ch in TENSOR_SHAPES[name])
        buf = np.zeros(int(np.prod(dims)))
        self._chk(self.L.afesp_ccsd_get_tensor(self.h, name.encode(), buf, buf.size))
        return buf.reshape(dims, order="F")

    # ---- src/ccsd.f90:2018-2293
    def ntriples(self):
        return self.L.afesp_ccsd_t_ntriples(self.o)

    def shard_bounds(self, world, cr=False):
        """Cost-balanced shard boundaries of the (i<=j<=k) list: rank r evaluates [b[r], b[r+1])."""
        b = (i64 * (world + 1))()
        self._chk(self.L.afesp_ccsd_t_shard_bounds(self.h, self.o, self.v, 1 if cr else 0, world, b))
        return [int(x) for x in b]

    def do_ccsd_t_spatial(self, t_begin=0, t_end=None):
        out = np.zeros(4)
        if t_end is None:
            t_end = self.ntriples()
        self._chk(self.L.afesp_ccsd_t(self.h, t_begin, t_end, out))
        return out

    # ---- multi-GPU: the OpenMP reduction of src/ccsd.f90:2091 as a sum over ranks (include/afesp.h)
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.L.afesp_comm_unique_id(buf)
        if rc != 0:
            raise AfespError(f"afesp_comm_unique_id failed with status {rc} (is librccl.so.1 loadable?)")
        return buf.raw

    def comm_init(self, rank, world, transport=COMM_RCCL, bootstrap_path=None, unique_id=None):
        self._chk(self.L.afesp_comm_init(self.h, rank, world, transport,
                                         None if bootstrap_path is None else str(bootstrap_path).encode(), unique_id))

    def comm_destroy(self):
        self._chk(self.L.afesp_comm_destroy(self.h))

    def allreduce_sum(self, values):
        buf = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._chk(self.L.afesp_allreduce_sum(self.h, buf, buf.size))
        return buf

    def t_block_size(self, cr=False):
        sb = C.c_int()
        self._chk(self.L.afesp_ccsd_t_block_size(self.h, self.o, self.v, 1 if cr else 0, C.byref(sb)))
        return sb.value

    def arena_stats(self):
        out = np.zeros(4)
        self._chk(self.L.afesp_arena_stats(self.h, out))
        return dict(driver_calls=int(out[0]), reuse_hits=int(out[1]), idle_gb=out[2] / 1e9, live_gb=out[3] / 1e9)

    def launch_counts(self):
        """launches so far in this context: tall (streamed tall x skinny kernel), gett (gather kernel through the planner), tgemm
        (LDS-DMA GEMM, 128-row tiles), tgemm_mixed (... with 96-row tiles where the rows end)"""
        out = (C.c_uint64 * 4)()
        self._chk(self.L.afesp_launch_counts(self.h, out))
        return dict(tall=int(out[0]), gett=int(out[1]), tgemm=int(out[2]), tgemm_mixed=int(out[3]))

    def ccsd_set_split(self, mode):
        """1: split the CCSD iteration over the ranks, 0: replicas, -1: as AFESP_CC_SHARD says (default replicas)."""
        self._chk(self.L.afesp_ccsd_set_split(self.h, int(mode)))

    def ccsd_set_fused(self, mode):
        """1: the launch-fused iteration of small systems (csrc/fused.h), 0: call by call, -1: as AFESP_FUSED says (default on)."""
        self._chk(self.L.afesp_ccsd_set_fused(self.h, int(mode)))

    def ccsd_iteration_launches(self):
        """Kernel launches of one compiled (launch-fused) iteration; 0 when the iteration runs call by call."""
        n = C.c_int()
        self._chk(self.L.afesp_ccsd_iteration_launches(self.h, C.byref(n)))
        return n.value

    def ccsd_is_split(self):
        f = C.c_int()
        self._chk(self.L.afesp_ccsd_is_split(self.h, C.byref(f)))
        return bool(f.value)

    def test_inject(self, what):
        self._chk(self.L.afesp_test_inject(self.h, what))

    # ---- input / output side: src/integrals.f90:146-161, src/mp2.f90:451-487
    def read_eri_text(self, path, nbasis, want_host_copy=True):
        """-> (packed AO integrals or None, number of lines); the packed array also stays on the device."""
        out = np.zeros(self.L.afesp_neri(nbasis)) if want_host_copy else None
        n = i64()
        self._chk(self.L.afesp_read_eri_text(self.h, str(path).encode(), nbasis,
                                             out.ctypes.data_as(C.c_void_p) if out is not None else None, C.byref(n)))
        return out, n.value

    def set_eri(self, nbasis, eri_packed):
        self._chk(self.L.afesp_set_eri(self.h, nbasis, np.ascontiguousarray(eri_packed, dtype=np.float64)))

    def build_fock(self, nbasis, density, core_hamil):
        """src/hf.f90:349-385 on the device-resident packed AO integrals."""
        out = np.zeros(nbasis * nbasis)
        self._chk(self.L.afesp_build_fock(self.h, nbasis, _f(density), _f(core_hamil), out))
        return out.reshape((nbasis, nbasis), order="F")

    def write_fcidump(self, path, nbasis):
        n = i64()
        self._chk(self.L.afesp_write_fcidump(self.h, str(path).encode(), nbasis, C.byref(n)))
        return n.value

    # ---- spin-orbital path: do_ccsd_spinorb (src/ccsd.f90:71-277), do_ccsd_t_spinorb (:1812-1922)
    SO_SHAPES = {"F_vv": "vv", "F_oo": "oo", "F_ov": "ov", "W_oooo": "oooo", "W_vvvv": "vvvv", "W_ovvo": "ovvo", "tau": "oovv",
                 "tau_tilde": "oovv", "oovv": "oovv", "vvvv": "vvvv", "t1": "ov", "t2": "oovv"}

    def init_cc_spinorb(self, nbasis, nel, canon_levels, eri_mo=None, diis_nerr=8, foo_as_published=False):
        self.so_o, self.so_v = int(nel), int(2 * nbasis - nel)
        eri = None
        if eri_mo is not None:
            eri_mo = np.ascontiguousarray(eri_mo, dtype=np.float64)
            eri = eri_mo.ctypes.data_as(C.c_void_p)
        self._chk(self.L.afesp_ccsd_so_init(self.h, nbasis, nel, eri, np.ascontiguousarray(canon_levels, dtype=np.float64),
                                            diis_nerr, 1 if foo_as_published else 0))

    def _so_step(self, fn, e_tol, t_tol):
        e, r, c = dbl(), dbl(), C.c_int()
        self._chk(fn(self.h, e_tol, t_tol, C.byref(e), C.byref(r), C.byref(c)))
        return e.value, r.value, bool(c.value)

    def so_energy(self, e_tol=1e-6, t_tol=1e-7):
        return self._so_step(self.L.afesp_ccsd_so_energy, e_tol, t_tol)

    def so_iterate(self, e_tol=1e-6, t_tol=1e-7):
        return self._so_step(self.L.afesp_ccsd_so_iterate, e_tol, t_tol)

    def so_diis(self):
        self._chk(self.L.afesp_ccsd_so_diis(self.h))

    def do_ccsd_spinorb(self, maxiter=50, e_tol=1e-6, t_tol=1e-7):
        """The driver loop of src/ccsd.f90:215-275 -> (iterations or -1, energies incl. the MP1 line, un-rooted rms)."""
        en, rm = [], []
        e, r, _ = self.so_energy(e_tol, t_tol)
        en.append(e); rm.append(r)
        for it in range(1, maxiter + 1):
            e, r, conv = self.so_iterate(e_tol, t_tol)
            en.append(e); rm.append(r)
            if conv:
                return it, np.array(en), np.array(rm)
            self.so_diis()
        return -1, np.array(en), np.array(rm)

    def so_amplitudes(self):
        o, v = self.so_o, self.so_v
        t1 = np.zeros(o * v)
        t2 = np.zeros(o * o * v * v)
        self._chk(self.L.afesp_ccsd_so_get_amplitudes(self.h, t1, t2))
        return t1.reshape((o, v), order="F"), t2.reshape((o, o, v, v), order="F")

    def so_set_amplitudes(self, t1, t2):
        self._chk(self.L.afesp_ccsd_so_set_amplitudes(self.h, _f(t1), _f(t2)))

    def so_tensor(self, name):
        dims = tuple(self.so_o if ch == "o" else self.so_v for ch in self.SO_SHAPES[name])
        buf = np.zeros(int(np.prod(dims)))
        self._chk(self.L.afesp_ccsd_so_get_tensor(self.h, name.encode(), buf, buf.size))
        return buf.reshape(dims, order="F")

    def so_ntriples(self):
        return self.L.afesp_ccsd_so_t_ntriples(self.so_o)

    def do_ccsd_t_spinorb(self, t_begin=0, t_end=None):
        if t_end is None:
            t_end = self.so_ntriples()
        e = dbl()
        self._chk(self.L.afesp_ccsd_so_t(self.h, t_begin, t_end, C.byref(e)))
        return e.value

    def do_ccsd_t_spatial_plain(self, t_begin=0, t_end=None):
        """E[T], E(T) only: what plain CCSD(T)_spatial / CCSD[T]_spatial need (no y, no D sums)."""
        out = np.zeros(2)
        if t_end is None:
            t_end = self.ntriples()
        self._chk(self.L.afesp_ccsd_t_plain(self.h, t_begin, t_end, out))
        return out

    # ---- completely renormalised variants (src/ccsd.f90:2338-2551, :2186-2194)
    def build_cr_intermediates(self):
        self._chk(self.L.afesp_ccsd_cr_intermediates(self.h))

    def do_ccsd_t_spatial_cr(self, t_begin=0, t_end=None):
        out = np.zeros(6)
        if t_end is None:
            t_end = self.ntriples()
        self._chk(self.L.afesp_ccsd_t_cr(self.h, t_begin, t_end, out))
        return out

    # ---- src/linalg.fpp operator layer
    def gemm(self, transA, transB, m, n, k, A, B, Cmat=None, alpha=1.0, beta=0.0):
        Cflat = np.zeros(m * n) if Cmat is None else _f(Cmat)
        self._chk(self.L.afesp_gemm(self.h, transA.encode(), transB.encode(), m, n, k, alpha, _f(A), _f(B), beta, Cflat))
        return Cflat.reshape((m, n), order="F")

    def omp_reshape(self, in_arr, order, out_arr=None, beta=None):
        dims = (i64 * 4)(*in_arr.shape)
        oshape = tuple(in_arr.shape[int(ch) - 1] for ch in order)
        out = np.zeros(int(np.prod(oshape))) if out_arr is None else _f(out_arr)
        self._chk(self.L.afesp_permute4(self.h, dims, order.encode(), _f(in_arr), out, 0 if beta is None else 1,
                                        0.0 if beta is None else beta))
        return out.reshape(oshape, order="F")

    def contract(self, alpha, A, la, B, lb, beta, Cmat, lc, force_split=0, force_tm=0, force_tn=0):
        dA, dB, dC = (i64 * len(la))(*A.shape), (i64 * len(lb))(*B.shape), (i64 * len(lc))(*Cmat.shape)
        Cflat = _f(Cmat)
        self._chk(self.L.afesp_contract(self.h, alpha, _f(A), la.encode(), dA, _f(B), lb.encode(), dB, beta, Cflat,
                                        lc.encode(), dC, force_split, force_tm, force_tn))
        return Cflat.reshape(Cmat.shape, order="F")

    def bench_contract(self, la, dA, lb, dB, lc, dC, reps=5):
        ms = dbl()
        self._chk(self.L.afesp_bench_contract(self.h, la.encode(), (i64 * len(dA))(*dA), lb.encode(), (i64 * len(dB))(*dB),
                                              lc.encode(), (i64 * len(dC))(*dC), reps, C.byref(ms)))
        return ms.value

    def profile(self, enable):
        out = np.zeros(8)
        self._chk(self.L.afesp_profile(self.h, 1 if enable else 0, out))
        return dict(gemm_ms=out[0], gemm_launches=int(out[1]), gemm_flop=out[2], orbit_ms=out[3], orbit_launches=int(out[4]),
                    orbit_bytes=out[5], gemm_flop_padded=out[6], gemm_kernel="tgemm_kernel" if out[7] else "gett_kernel")

    def bench_stream(self, n, reps=5):
        ms = dbl()
        self._chk(self.L.afesp_bench_stream(self.h, n, reps, C.byref(ms)))
        return ms.value

    def set_tuning(self, group_m=0, tm=0, tn=0, split=0):
        self.L.afesp_set_tuning(group_m, tm, tn, split)

    def synthetic_ao(self, nbasis, scale=0.02, seed=12345):
        self._chk(self.L.afesp_synthetic_ao(self.h, nbasis, scale, seed))

    def pp_ladder_flop(self):
        f = dbl()
        self._chk(self.L.afesp_ccsd_pp_ladder_flop(self.h, C.byref(f)))
        return f.value

    def iteration_flop(self):
        f = dbl()
        self._chk(self.L.afesp_ccsd_iteration_flop(self.h, C.byref(f)))
        return f.value

    def time_pp_ladder(self, reps=10):
        ms = dbl()
        self._chk(self.L.afesp_time_pp_ladder(self.h, reps, C.byref(ms)))
        return ms.value
