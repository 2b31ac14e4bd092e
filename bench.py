#!/usr/bin/env python3
"""bench.py -- CCSD iteration + (T) on MI355X through the C-ABI (libafesp_hip.so).

One "step" = one pass of the hot path over one system held in HBM:
    one CCSD iteration (src/ccsd.f90:340-395: intermediates, amplitudes, energy, DIIS)  -- replicated on every rank
  + the full (T) correction (src/ccsd.f90:2018-2293), its (i<=j<=k) triples sharded over the ranks,
    followed by ONE all-reduce (RCCL) of the four scalars E[T], E(T), D[T], D(T).
Inputs are generated on the device (afesp_synthetic_init); nothing crosses PCIe inside the timed region.

Workloads (BASELINE.json configs):
  cfg5    config 5: synthetic o=20, v=200 -- the largest single-GPU configuration and the only one where the fp64 MFMA roofline
          means anything: the default, and what `value` is quoted on
  h2o_tz  config 2 shape: H2O/cc-pVTZ extents o=5, v=53 (its eri.dat is not bundled -> synthetic integrals)
  n2      config 3 extents o=7, v=21 (synthetic integrals; the real N2 / F2 inputs run in the `real_molecules_same_run` leg)
value = the fp64 multiply-adds (x 2) the kernels of the step issue / step time, summed over the job (strong scaling: total work
fixed); `value_survey_count` is SURVEY.md 8(d)'s algorithmic count of the same step over the same time (one count per number).

`--gpus N` without a launcher (WORLD_SIZE unset): this process starts N ranks of itself -- before anything touches the GPU -- and
waits for them; under torchrun (WORLD_SIZE set) it must equal the world size.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {"h2o_tz": (5, 53), "cfg5": (20, 200), "n2": (7, 21), "f2": (9, 19), "mid": (10, 100), "mid_large": (12, 120)}
MFMA_F64_PEAK_TFLOPS = 78.6      # v_mfma_f64_16x16x4_f64: 32 FLOP/clk/SIMD x 4 SIMD x 256 CU x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec


def flops_iter(o, v, pp_flop=None):
    """sum over every contraction site of one CCSD iteration -- with the pp-ladder counted as it is
    executed here (`pp_flop` from the engine: o^2 v^2 v(v+1) over the symmetry-unique column pairs, or about o^2 v^4 / 2 in
    the symmetric/antisymmetric pair form, instead of the reference's 2 o^2 v^4), the same "count the algorithm that is
    timed" rule as for (T)."""
    if pp_flop is None:
        pp_flop = o**2 * v**2 * v * (v + 1)
    return (int(pp_flop) + 14 * o**3 * v**3 + 2 * o**4 * v**2 + 2 * o**4 * v + 18 * o**2 * v**3 + 2 * o * v**3
            + 14 * o**3 * v**2)


def flops_t_sym(o, v):
    """(T) with the i<=j<=k restriction (the algorithm timed here): SURVEY.md 8(d)."""
    return (o * (o + 1) * (o + 2) // 6) * 12 * v**3 * (v + o)


def flops_t_ref(o, v):
    """(T) as the reference formulates it: all o^3 ordered triples."""
    return o**3 * 12 * v**3 * (v + o)


def hash_uniform(k, seed):
    """numpy twin of the device generator in csrc/capi.hip (splitmix64)."""
    x = (k.astype(np.uint64) + np.uint64(seed)) + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / 9007199254740992.0


def synthetic_host(o, v, scale, seed):
    n = o + v
    npair = n * (n + 1) // 2
    ne = npair * (npair + 1) // 2
    with np.errstate(over="ignore"):
        eri = scale * (2.0 * hash_uniform(np.arange(ne, dtype=np.uint64), seed) - 1.0)
    e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
    return e, eri


def cpu_baseline(o, v, scale, seed, eng, budget_s=25.0):
    """Time the CPU restatement (oracle/, kind "port") on a bounded sample of the same workload.  (T) and the pp-ladder run in
    the reference's own shape -- one dgemm per permuted term with OpenMP over (i,j,k), a multi-threaded dgemm for the ladder
    (oracle/afesp_oracle_blas.c on the OpenBLAS numpy bundles) -- so the baseline is BLAS-backed like the reference's CPU path
    (src/ccsd.f90:2056-2066, :2091, :1669); the loop form of oracle/afesp_oracle.c is timed beside it on the small workload."""
    # A GPU box advertises many more hardware threads than the CPU share its job gets (16 per GPU on this pool); more
    # OpenMP threads than that only spin against each other.  AFESP_BENCH_THREADS overrides.
    cores = os.cpu_count() or 1
    threads = int(os.environ.get("AFESP_BENCH_THREADS", min(cores, 16)))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import orc
    L = orc.lib()
    LB = orc.blas_lib()
    try:
        omp = ctypes.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(threads)
    except OSError:
        pass
    n = o + v
    f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
    blas_note = ("(T): one dgemm per permuted term, OpenMP over (i,j,k) with serial BLAS inside, as src/ccsd.f90:2056-2066/:2091; "
                 "OpenBLAS bundled with numpy") if LB is not None else "numpy's OpenBLAS not found: loop form only"

    def time_triples(fn, args, per_hint=None):
        """as many ordered triples as fit the budget (at least one per thread), extrapolated to the reference's o^3"""
        out = np.zeros(4)
        t0 = time.perf_counter()
        fn(*args, 0, threads, out)
        per = max(time.perf_counter() - t0, 1e-6)              # one round of `threads` triples
        ns = int(min(o**3, max(threads, threads * int(budget_s * 0.4 / per))))
        if ns > threads:
            t0 = time.perf_counter()
            fn(*args, 0, ns, out)
            per, cnt = time.perf_counter() - t0, ns
        else:
            cnt = threads
        return per * (o**3 / cnt), cnt

    if n <= 64:
        e, eri = synthetic_host(o, v, scale, seed)
        cc = orc.OracleCC(o, v, eri, e, 8)
        t0 = time.perf_counter()
        L.orc_cc_diis_save(cc.h)
        L.orc_cc_intermediates(cc.h)
        L.orc_cc_amplitudes(cc.h)
        L.orc_cc_energy(cc.h, 1e-6, 1e-7)
        L.orc_cc_diis_update(cc.h)
        t_iter = time.perf_counter() - t0
        targs = (o, v, np.ascontiguousarray(e), f(cc.t1), f(cc.t2), f(cc.field("v_vvov")), f(cc.field("v_oovo")), f(cc.field("v_oovv")))
        t_loops, ns = time_triples(L.orc_ccsd_t, targs)
        t_t, nsb = time_triples(LB.orcb_ccsd_t, targs) if LB is not None else (t_loops, ns)
        sample = f"1 full CCSD iteration (loop form) + {nsb}/{o**3} ordered (i,j,k) triples of (T), scaled to o^3"
        return {"value": t_iter + t_t, "unit": "s/step", "ccsd_iter_s": t_iter, "t_s": t_t, "t_s_loop_form": t_loops, "cores": threads,
                "kind": "port", "blas": blas_note, "sample": sample,
                "gflops": (flops_iter(o, v) + flops_t_ref(o, v)) / (t_iter + t_t) / 1e9}
    # large system: the reference formulation cannot run here (n<=99); time slabs of the restatement in the
    # reference's own shape -- the dgemm sites through threaded OpenBLAS, the sites the reference leaves to OpenMP loop nests as
    # those loop nests -- and scale each to its whole
    t1, t2 = eng.amplitudes()
    e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
    targs = (o, v, e, f(t1), f(t2), f(eng.tensor("v_vvov")), f(eng.tensor("v_oovo")), f(eng.tensor("v_oovv")))
    if LB is not None:
        # at least 5 % of the ordered triples (whole rounds of `threads`), more if the budget allows
        out = np.zeros(4)
        t0 = time.perf_counter()
        LB.orcb_ccsd_t(*targs, 0, threads, out)
        per = max(time.perf_counter() - t0, 1e-6)
        want = -(-int(0.05 * o**3) // threads) * threads
        ns = int(min(o**3, max(want, threads * int(budget_s * 0.4 / per))))
        t0 = time.perf_counter()
        LB.orcb_ccsd_t(*targs, 0, ns, out)
        t_t = (time.perf_counter() - t0) * (o**3 / ns)
    else:
        out = np.zeros(4)
        ns = threads
        t0 = time.perf_counter()
        L.orc_ccsd_t(*targs, 0, ns, out)
        t_t = (time.perf_counter() - t0) * (o**3 / ns)
    # (1) dgemm sites.  The pp-ladder (ccsd.f90:1669) is one dgemm of o^2 x v^2 x v^2: a slab of its columns; the other dgemm
    # sites (the I_voov products :1193-1230, I_oooo, the o v^3 and o^2 v^2 terms) at the same rate
    c = f(eng.tensor("c_oovv"))
    ncol = 1024 if LB is not None else 64
    vv = np.full(v * v * ncol, 0.01)
    res = np.zeros(o * o * ncol)
    if LB is not None:
        LB.orcb_gemm(o * o, ncol, v * v, 0.5, c, vv, 0.0, res, threads)      # warm-up (thread pool, pages)
        t0 = time.perf_counter()
        LB.orcb_gemm(o * o, ncol, v * v, 0.5, c, vv, 0.0, res, threads)
    else:
        t0 = time.perf_counter()
        L.orc_gemm(0, 0, o * o, ncol, v * v, 0.5, c, vv, 0.0, res)
    t_lad = (time.perf_counter() - t0) * (v * v / ncol)
    f_lad, f_loops = 2 * o**2 * v**4, 8 * o**3 * v**3       # ladder as the reference's full dgemm; the four o^3 v^3 products in loops
    f_iter = flops_iter(o, v, f_lad)
    t_blas = t_lad * (f_iter - f_loops) / f_lad
    # (2) loop sites: I_ovov's c-term (ccsd.f90:1170-1182) and terms 6-8 of the T2 equation (:1680-1695, "seems hopeless, use
    # OMP") are OpenMP loop nests in the reference, 8 o^3 v^3 flop of the iteration: slabs of their outermost index
    t_loops, nsl = None, 0
    if LB is not None:
        tl = time.perf_counter()
        asym, I_ovov, I_voov = f(eng.tensor("asym_t2")), f(eng.tensor("I_ovov")), f(eng.tensor("I_voov"))
        v_oovv = targs[7]
        buf = I_ovov.copy()
        acc2 = np.zeros(o * o * v * v)
        nsl = max(2, -(-v // 25))                          # >= 4 % of the outermost index of each nest
        t0 = time.perf_counter()
        LB.orcb_ring_I_ovov(o, v, v_oovv, c, buf, 0, nsl)
        t_a = (time.perf_counter() - t0) * (v / nsl)
        t0 = time.perf_counter()
        LB.orcb_ring_t2(o, v, f(t2), asym, I_ovov, I_voov, acc2, 0, nsl)
        t_b = (time.perf_counter() - t0) * (v / nsl)
        t_loops = t_a + t_b
    t_iter = t_blas + (t_loops if t_loops is not None else t_lad * f_loops / f_lad)
    sample = (f"(T): {ns}/{o**3} ordered triples scaled to o^3; CCSD iteration: {ncol}/{v*v} columns of the pp-ladder dgemm scaled to v^2 "
              f"and to the flop of all dgemm sites; {nsl}/{v} slabs of each of the reference's two o^3 v^3 OpenMP loop nests scaled to v")
    return {"value": t_iter + t_t, "unit": "s/step", "ccsd_iter_s": t_iter, "t_s": t_t, "cores": threads, "kind": "port",
            "extrapolated": True, "ccsd_iter_s_blas_sites": t_blas, "ccsd_iter_s_loop_sites": t_loops,
            "sample_fraction": {"t_ordered_triples": ns / o**3, "pp_ladder_columns": ncol / (v * v), "loop_nest_slabs": nsl / v},
            "blas": blas_note, "sample": sample, "gflops": (flops_iter(o, v) + flops_t_ref(o, v)) / (t_iter + t_t) / 1e9}


def time_ao2mo(eng, o, v, reps):
    """AO->MO transform + repack + MP2 energy (src/mp2.f90:261-449) on packed AO integrals already resident in HBM, with a
    seeded random orthogonal coefficient matrix (the identity is degenerate for timing)."""
    n = o + v
    q, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((n, n)))
    e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
    eng.synthetic_ao(n, 0.02, 777)
    emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False)      # warm-up (plans, buffers)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False)  # returns E(MP2) to the host: synchronous
        times.append(time.perf_counter() - t0)
    # median of the calls: each allocates and frees its two n^2 x npair temporaries, and a hipMalloc of several GB now and
    # then takes seconds on this runtime (tools: alloc timing in DESIGN.md section 4.4)
    sec = float(np.median(times))
    return {"nbasis": n, "ms": sec * 1e3, "ms_min": min(times) * 1e3, "ms_max": max(times) * 1e3, "calls": reps,
            "tflops_reference_count_8n5": 8 * n**5 / sec / 1e12,
            "algorithmic_gbs": 8 * (n**4 + (n * (n + 1) // 2) ** 2) / sec / 1e9, "e_mp2": emp2}


# Timing lines of the reference's own bundled outputs (hardware not stated anywhere): context, not a baseline
PUBLISHED = {
    "n2-cc-pvdz": {"ao2mo_mp2": 0.220, "ccsd_iter": 0.065, "cr_t": 0.662, "source": "sample_data/n2-cc-pvdz/2.00_0.00/els.out, unknown CPU"},
    "f2-cc-pvdz": {"ao2mo_mp2": 0.083, "ccsd_iter": 0.027, "cr_t": 1.705, "source": "sample_data/f2-cc-pvdz/1.75_0.00/els.out, unknown CPU"},
    "h2o-cc-pvtz_spinorb": {"ao2mo_mp2": 0.442, "ccsd_iter": 2.3, "t": 123.0,
                            "source": "sample_data/h2o-cc-pvtz/2.00_104.45/els_cpu.out (CCSD(T)_spinorb), unknown CPU, multi-threaded"},
}


def spinorb_h2o_tz(rank, world, local, dist, cdev, torch, backend, jobdir):
    """The configuration of the reference's only published H2O/cc-pVTZ timings: CCSD(T)_spinorb, 10 electrons in 58 spatial
    orbitals (o = 10, v = 106 spin orbitals), here on synthetic integrals of that shape (its eri.dat is not bundled).  The
    i<j<k triples of (T) are split evenly over the ranks, one all-reduce of the scalar."""
    from afesp_amd import dist as adist, inputs
    from afesp_amd.capi import Engine
    n, nel = 58, 10
    o = nel // 2
    e = np.concatenate([-2.0 + np.arange(o) / (o - 1), 1.0 + 2.0 * np.arange(n - o) / (n - o - 1)])
    eri = 0.02 * (2.0 * np.random.default_rng(1).random(inputs.neri(n)) - 1.0)
    eng = Engine(local)
    red = Reducer(eng, rank, world, dist, cdev, torch, backend, jobdir)   # the product's own all-reduce (afesp_allreduce_sum)
    eng.init_cc_spinorb(n, nel, e, eri, 8)
    eng.so_energy()
    per_iter = []
    for _ in range(8):
        t0 = time.perf_counter()
        eng.so_iterate()
        eng.so_diis()
        per_iter.append(time.perf_counter() - t0)
    lo, hi = adist.shard_range(eng.so_ntriples(), rank, world)
    eng.do_ccsd_t_spinorb(lo, hi)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    et = red.sum(np.array([eng.do_ccsd_t_spinorb(lo, hi)]))
    t_t = time.perf_counter() - t0
    red.close()
    eng.close()
    os_, vs_ = nel, 2 * n - nel
    npo, npv = os_ * (os_ - 1) // 2, vs_ * (vs_ - 1) // 2
    t_it = float(np.median(per_iter))
    # what the iteration executes (csrc/ccsd_so.hip): the W_abef term over antisymmetric pairs, Z(ijma) = tau <ma||ef>, the two (ov)^3
    # products of W_mbej / T2, the two o^4 v^2 products of W_mnij; beside it the count of the reference's formulation (stored W_abef:
    # 2 o v^4 to build it, 2 o^2 v^4 to contract it, src/ccsd.f90:849-858, :1021-1024)
    flop_exec = 2.0 * npo * npv * npv + 2.0 * os_**3 * vs_**3 + 4.0 * (os_ * vs_) ** 3 + 4.0 * os_**4 * vs_**2 + 4.0 * os_**2 * vs_**3
    flop_ref = 2.0 * os_**2 * vs_**4 + 2.0 * os_ * vs_**4 + 4.0 * (os_ * vs_) ** 3 + 4.0 * os_**4 * vs_**2 + 4.0 * os_**2 * vs_**3
    # (T): every triple i < j < k of the whole list (this rank's share: 1 / world of them) is three products of v^2 x v x (v + o)
    flop_t = (os_ * (os_ - 1) * (os_ - 2) // 6) * 3 * 2.0 * vs_**3 * (vs_ + os_)
    return {"nocc_spin": nel, "nvirt_spin": 2 * n - nel, "ccsd_iter_s": t_it, "t_s": t_t, "e_t": float(et[0]),
            "t_flop_executed": flop_t, "t_fraction_of_mfma_peak": flop_t / t_t / 1e12 / MFMA_F64_PEAK_TFLOPS / max(world, 1),
            "t_dominant_kernel": "tgemm_kernel (one launch per chunk of triples: Y(b,c;a) blocks over kappa = f (+) m) + triples_so_orbit_kernel; "
                                 "operand copies cached per converged state -- kernel sequence in profiles/r05_so_t_timeline.txt",
            "flop_per_iter_executed": flop_exec, "flop_per_iter_reference_formulation": flop_ref,
            "fraction_of_mfma_peak": flop_exec / t_it / 1e12 / MFMA_F64_PEAK_TFLOPS,
            "tflops_reference_equivalent": flop_ref / t_it / 1e12,
            "dominant_kernels": "gett_kernel: tau(ij,ef) <ab||ef> over antisymmetric pairs (M = K = v(v-1)/2, N = o(o-1)/2; streams 8 [v(v-1)/2]^2 bytes), "
                                "Z = tau <ma||ef>, the two (ov)^3 ring products -- kernel sequence in profiles/r04_so_timeline.txt",
            "t_allreduce": red.kind, "integrals": "synthetic (the reference tree has no eri.dat for H2O/cc-pVTZ: .MISSING_LARGE_BLOBS)",
            "published_reference_s": PUBLISHED["h2o-cc-pvtz_spinorb"]}


def real_molecule(name, rank, world, local, dist, cdev, torch, backend, jobdir):
    """BASELINE configs 3 / 4: the bundled N2 / F2 cc-pVDZ inputs (tests/golden, copies of the reference's sample_data) through
    the whole path -- RHF on the host, AO->MO + MP2, CCSD to convergence, (T) with the (i<=j<=k) triples sharded over the
    ranks and one all-reduce -- with the energies checked against the reference's own outputs."""
    import molecules
    from afesp_amd.capi import Engine
    si, ints, res, _ = molecules.load(name)
    gold = molecules.SURVEY_GOLD[name]
    n, o = ints.nbasis, ints.nel // 2
    v = n - o
    eng = Engine(local)
    red = Reducer(eng, rank, world, dist, cdev, torch, backend, jobdir)   # the product's own all-reduce (afesp_allreduce_sum)
    t0 = time.perf_counter()
    e_mp2, _ = eng.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, ints.eri, want_eri_mo=False)
    t_ao = time.perf_counter() - t0
    eng.ccsd_init(o, v, res.canon_levels, None, si.ccsd_diis_n_errmat)
    # the loop of src/ccsd.f90:340-395 driven per iteration, as the Fortran host does, so that every iteration is timed
    en = [eng.ccsd_energy(si.ccsd_e_tol, si.ccsd_t_tol)[0]]
    per_iter, nit = [], 0
    t0 = time.perf_counter()
    for it in range(1, si.ccsd_maxiter + 1):
        t1 = time.perf_counter()
        e_it, _, conv = eng.ccsd_iterate(si.ccsd_e_tol, si.ccsd_t_tol)
        if not conv:
            eng.ccsd_diis()
        per_iter.append(time.perf_counter() - t1)
        en.append(e_it)
        if conv:
            nit = it
            break
    t_cc = time.perf_counter() - t0
    lo, hi = eng.shard_bounds(world)[rank:rank + 2]
    eng.do_ccsd_t_spatial(lo, hi)                                        # first call builds the (T) plan
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    part = red.sum(np.asarray(eng.do_ccsd_t_spatial(lo, hi), dtype=np.float64))
    t_t = time.perf_counter() - t0
    # the bundled outputs are CRCCSD(T)_spatial runs: also time what they timed (moments + the completely renormalised (T))
    clo, chi = eng.shard_bounds(world, cr=True)[rank:rank + 2]
    eng.build_cr_intermediates()
    eng.do_ccsd_t_spatial_cr(clo, chi)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    eng.build_cr_intermediates()
    crp = red.sum(np.asarray(eng.do_ccsd_t_spatial_cr(clo, chi), dtype=np.float64))
    t_cr = time.perf_counter() - t0
    # the same molecule again in the same context -- what a scan over geometries does (the reference's utils/els_wrapper.py): the
    # state of the same extents is initialised where it lies and its compiled programs stay, so this solve has no recording in it
    eng.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, ints.eri, want_eri_mo=False)
    eng.ccsd_init(o, v, res.canon_levels, None, si.ccsd_diis_n_errmat)
    t0 = time.perf_counter()
    nit2, en2, _ = eng.do_ccsd_spatial(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    t_cc2 = time.perf_counter() - t0
    red.close()
    eng.close()
    ec = float(en[nit])
    if int(nit2) != int(nit) or abs(float(en2[nit2]) - ec) > 1e-12:
        raise RuntimeError(f"{name}: the re-entered solve differs: {nit2} iterations, E = {en2[nit2]!r} against {nit}, {ec!r}")
    got = {"mp2_corr": e_mp2, "ccsd_corr": ec, "ccsd_bt_corr": ec + part[0], "ccsd_pt_corr": ec + part[1],
           "d_bt": part[2], "d_pt": part[3]}
    return {"nocc": o, "nvirt": v, "ao2mo_mp2_s": t_ao, "ccsd_iterations": int(nit), "ccsd_solve_s": t_cc,
            "ccsd_solve_reentered_s": t_cc2,
            "ccsd_iter_s": float(np.median(per_iter)), "ccsd_iter_s_first_three": [float(x) for x in per_iter[:3]],
            "t_s": t_t, "cr_t_s": t_cr, "cr_t_vs_t_max_abs_diff": float(np.max(np.abs(crp[:4] - part[:4]))),
            "t_allreduce": red.kind, "t_shard": [int(lo), int(hi)],
            "published_reference_s": PUBLISHED[name], "energies": {k: float(x) for k, x in got.items()},
            "max_abs_error_vs_reference_Eh": max(abs(float(got[k]) - gold[k]) for k in got)}


DEFAULT_SCALE = {"cfg5": 0.005}     # magnitude of the hashed integrals: keeps the first iterates of every workload finite


class Reducer:
    """The one collective of the path: the sum over ranks of the (T) partial scalars.  It goes through the product's own
    C-ABI (afesp_comm_init / afesp_allreduce_sum: ncclAllReduce on the engine's stream, or the host segment when the ranks of a
    rehearsal share one GPU); the RCCL unique id travels by a torch.distributed broadcast.  If the C-ABI communicator cannot be
    set up the sum falls back to torch.distributed -- the line says which one ran (`t_allreduce`)."""

    def __init__(self, eng, rank, world, dist, cdev, torch, backend, jobdir):
        self.eng, self.world, self.dist, self.cdev, self.torch = eng, world, dist, cdev, torch
        self.kind = "none (one rank)"
        self.own = False
        self.checked = False          # the first sum through the product's own transport has come back on every rank
        if world == 1:
            return
        from afesp_amd import capi
        try:
            if backend == "gloo":
                eng.comm_init(rank, world, capi.COMM_HOST, os.path.join(jobdir, f"afesp_seg_{Reducer.count}"))
                self.kind = "afesp_allreduce_sum (host segment: ranks share a GPU)"
            else:
                # rank 0's failure to produce an id must still reach the broadcast, or the other ranks wait in it for ever
                box = [None]
                if rank == 0:
                    try:
                        box = [eng.comm_unique_id()]
                    except Exception as exc:   # noqa: BLE001
                        box = [f"error: {exc}"]
                dist.broadcast_object_list(box, src=0)
                if not isinstance(box[0], (bytes, bytearray)):
                    raise RuntimeError(f"no RCCL unique id from rank 0 ({box[0]})")
                eng.comm_init(rank, world, capi.COMM_RCCL, None, box[0])
                self.kind = "afesp_allreduce_sum (ncclAllReduce on the engine stream)"
            self.own = True
        except Exception as exc:   # noqa: BLE001 -- any failure here must not lose the measurement
            self.own = False
            self.kind = f"torch.distributed.all_reduce (afesp_comm_init failed: {exc})"
        Reducer.count += 1
        # every rank must have taken the same branch
        flag = torch.tensor([1.0 if self.own else 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if self.own and float(flag.cpu()[0]) == 0.0:
            eng.comm_destroy()
            self.own = False
            self.kind = "torch.distributed.all_reduce (afesp_comm_init failed on another rank)"
        if self.own:
            self._probe(rank)

    def _probe(self, rank):
        """Transport health is agreed BEFORE the first real collective: one tiny afesp_allreduce_sum right behind comm_init (the first time
        ncclAllReduce runs on these N ranks at all), on a helper thread with a time limit, and its outcome -- fine / raised / still waiting
        -- exchanged over torch.distributed.  All fine: the product's own transport serves the run.  All raised (nobody enqueued
        anything): every rank falls back to torch.distributed and the line says so.  Anything one-sided -- a rank that raised while the
        others sit in the library's stream synchronisation behind a collective waiting for it, or a rank that timed out -- cannot be
        repaired from inside: every rank says so on stderr and exits non-zero at once, so the spawner fails fast instead of waiting in a
        barrier."""
        import threading
        box = {}

        def run():
            try:
                box["out"] = self.eng.allreduce_sum(np.array([float(rank + 1), 1.0]))
            except Exception as exc:   # noqa: BLE001
                box["err"] = exc

        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("AFESP_BENCH_PROBE_TIMEOUT", "180")))
        w = self.world
        if th.is_alive():
            state = 2.0                                                            # still waiting in the library
        elif "err" in box or not np.allclose(box.get("out", [0, 0]), [w * (w + 1) / 2, w], rtol=0, atol=1e-9):
            state = 1.0                                                            # raised, or a wrong sum
        else:
            state = 0.0
        hi = self.torch.tensor([state], dtype=self.torch.float64, device=self.cdev)
        lo = hi.clone()
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        hi, lo = float(hi.cpu()[0]), float(lo.cpu()[0])
        if hi == 0.0:
            self.checked = True
            return
        if hi == 1.0 and lo == 1.0:   # every rank raised before anything was enqueued: a clean fall-back
            try:
                self.eng.comm_destroy()
            except Exception:   # noqa: BLE001
                pass
            self.own = False
            self.kind = f"torch.distributed.all_reduce (the probe afesp_allreduce_sum failed on every rank: {box.get('err')})"
            return
        sys.stderr.write(f"bench.py rank {rank}: the probe afesp_allreduce_sum ended one-sided (this rank: "
                         f"{['fine', 'raised: ' + str(box.get('err')), 'no answer within the time limit'][int(state)]}); "
                         "ranks disagree about the transport -- exiting\n")
        sys.stderr.flush()
        os._exit(3)

    count = 0

    def sum(self, values):
        if self.world == 1:
            return np.asarray(values, dtype=np.float64)
        if self.own:
            return self.eng.allreduce_sum(values)
        t = self.torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(self.cdev)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def close(self):
        if self.world > 1 and getattr(self, "own", False):
            self.eng.comm_destroy()


def live_pmc(workload, timeout_s=240.0):
    """HBM traffic and MFMA-pipe busy fraction of the grouped (T) GEMM launches, measured NOW: four rocprofv3 --pmc passes of
    this very command (one counter per pass, kernel trace only -- counters are collected in runs of their own, MI355X_MICROARCH
    guide) as child processes on the same GPU.  FETCH_SIZE counts 64 B per 128-B request on gfx950 (x2; calibrated on a stream of
    known size, profiles/r01_pmc_*).  Returns None if rocprofv3 is missing, a pass fails or the budget runs out."""
    import csv, glob, shutil, subprocess
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    if any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None   # this process is being profiled itself: no profiler inside a profiler
    t_end = time.time() + timeout_s
    # the (T) GEMM launches: tgemm_kernel (LDS-DMA kernel), or gett_kernel<..., GRP = true, RAG = false> under AFESP_T_GEMM=gett
    grouped = lambda name: "tgemm_kernel" in name or ("gett_kernel" in name and ", true, false>" in name)
    mean = {}
    with tempfile.TemporaryDirectory(prefix="afesp_pmc_", dir="/tmp") as td:
        for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            out = os.path.join(td, counter)
            cmd = [exe, "--kernel-trace", "--output-format", "csv", "--pmc", counter, "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                   "--no-extra", "--steps-only", "--no-live-pmc"]
            try:   # (a session of its own: on a time-out the profiler AND the program under it are ended, by process group)
                proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                        stderr=subprocess.DEVNULL, start_new_session=True)
            except OSError:
                return None
            try:
                rc = proc.wait(timeout=max(5.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                return None
            files = glob.glob(out + "/*/*counter_collection.csv")
            if rc != 0 or not files:
                return None
            vals = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                    if row["Counter_Name"] == counter and grouped(row["Kernel_Name"])]
            if not vals:
                return None
            mean[counter] = (sum(vals) / len(vals), len(vals))
    fetch, write = mean["FETCH_SIZE"][0] * 1024 * 2, mean["WRITE_SIZE"][0] * 1024
    return {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
            "dispatches": mean["FETCH_SIZE"][1],
            "mfma_busy_frac": mean["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (256 * 4) / (mean["GRBM_GUI_ACTIVE"][0] / 8)}


def latest_profile(pattern):
    """Newest profiles/rNN_<pattern> file (the PMC passes are separate rocprofv3 runs of this same command, tools/refresh_profiles.sh)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return files[-1] if files else None


def measure(args, workload, steps, warmup, rank, world, local, dist, cdev, torch, jobdir, with_roofline=True, with_cpu=True):
    """Run `warmup` untimed + `steps` timed steps of one workload; returns the result dictionary (rank 0) or None."""
    from afesp_amd.capi import Engine
    o, v = WORKLOADS[workload]
    scale = args.scale if args.scale is not None else DEFAULT_SCALE.get(workload, 0.02)
    seed = 12345
    eng = Engine(local)
    eng.synthetic_init(o, v, scale, seed, 8)
    eng.ccsd_energy()                                   # the "MP1" line: primes t2_old
    nt = eng.ntriples()
    lo, hi = eng.shard_bounds(world)[rank:rank + 2]     # contiguous shard of the i<=j<=k list, balanced by cost
    red = Reducer(eng, rank, world, dist, cdev, torch, args.backend, jobdir)
    sb = float(eng.t_block_size())                      # the ranks must enumerate the triples in the same block order
    chk = red.sum([1.0, sb, sb * sb])
    rccl_ranks = int(round(chk[0]))
    if abs(chk[2] * world - chk[1] ** 2) > 0.5 or rccl_ranks != world:
        raise SystemExit(f"bench.py: ranks disagree (ranks summed {chk[0]}, (T) block sizes sum {chk[1]}, squares {chk[2]})")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The CCSD iteration split over the ranks (ring products + pp-ladder, one all-reduce of [PP | residual]) is opt-in in the
    # library.  Here it is switched on for systems whose iteration is not launch-bound -- AFTER one iteration has been run both
    # ways from the same amplitudes on these very ranks and the two energies agree; otherwise the iterations stay replicas
    # and the line says why.
    split_check = "not applicable (one rank)" if world == 1 else "not attempted"
    if world > 1 and red.own and o * o * v * v > (1 << 20) and os.environ.get("AFESP_CC_SHARD", "1") != "0":
        ok, why = 1.0, ""
        try:
            t1s, t2s = eng.amplitudes()
            eng.ccsd_set_split(1)
            e_split = eng.ccsd_iterate()[0]
            eng.ccsd_set_split(0)
            eng.set_amplitudes(t1s, t2s)
            e_repl = eng.ccsd_iterate()[0]
            eng.set_amplitudes(t1s, t2s)
            if not abs(e_split - e_repl) <= 1e-10 * max(1.0, abs(e_repl)):
                ok, why = 0.0, f"energies differ: split {e_split!r}, replicas {e_repl!r}"
        except Exception as exc:   # noqa: BLE001 -- a failing split must not lose the measurement
            ok, why = 0.0, f"{type(exc).__name__}: {exc}"
        agree = red.sum([ok])
        if int(round(agree[0])) == world:
            eng.ccsd_set_split(1)
            split_check = "one iteration split == replicas to 1e-10 on every rank: split enabled"
        else:
            try:
                eng.ccsd_set_split(0)
            except Exception:   # noqa: BLE001
                pass
            split_check = f"failed on {world - int(round(agree[0]))} rank(s), iterations stay replicas" + (f" (this rank: {why})" if why else "")
    elif world > 1:
        split_check = "not attempted (launch-bound system, no product communicator, or AFESP_CC_SHARD=0)"

    acc = {"iter": 0.0, "trip": 0.0, "shard": 0.0, "allred": 0.0, "last": None}

    def step(timed):
        t0 = time.perf_counter()
        eng.ccsd_iterate()
        eng.ccsd_diis()
        t1 = time.perf_counter()
        # the benchmark configurations are CCSD(T)_spatial: E[T] and E(T) (the renormalised types' y / D sums are extra)
        part = eng.do_ccsd_t_spatial_plain(lo, hi)      # returns the partial sums to the host: this rank's shard is finished
        t1b = time.perf_counter()
        acc["last"] = red.sum(part)                     # the only collective of the path: 2 doubles over xGMI
        t2 = time.perf_counter()
        if timed:
            acc["iter"] += t1 - t0
            acc["trip"] += t2 - t1
            acc["shard"] += t1b - t1                    # (the all-reduce also absorbs the wait for the slowest rank)
            acc["allred"] += t2 - t1b

    for _ in range(warmup):
        step(False)
    eng.profile(True)                                   # HIP-event stamps around the (T) launches from here on
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile(False)
    # per rank, for the record of an N > 1 run: this rank's CCSD iteration, its (T) shard and the all-reduce behind it
    mine = [acc["iter"] / steps * 1e3, acc["shard"] / steps * 1e3, acc["allred"] / steps * 1e6, float(hi - lo)]
    per_rank = None
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    tt = torch.tensor([elapsed, acc["iter"], acc["trip"]], dtype=torch.float64, device=cdev)
    ex = torch.tensor([prof["gemm_flop"]], dtype=torch.float64, device=cdev)   # executed (T) multiply-adds of this rank's shard
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ex)
    elapsed, t_iter, t_trip = [float(x) for x in tt.cpu()]
    sec_per_step = elapsed / steps
    it_flop = int(eng.iteration_flop())     # the iteration as the engine evaluates it (pair forms counted as executed)
    flop_survey = flops_iter(o, v) + flops_t_sym(o, v)   # SURVEY 8(d), both parts algorithmic
    # executed: the (T) GEMMs evaluate o x o(o+1)/2 distinct blocks of 4 v^3 (v+o) flop (half of that where the occupied pair
    # coincides) instead of the symmetric count's 12 v^3 (v+o) per i<=j<=k triple; the CCSD iteration is replicated per rank
    split = world > 1 and red.own and eng.ccsd_is_split()   # the iteration's ring products and ladder run slice by slice on the ranks
    rep_flop = it_flop - (eng.pp_ladder_flop() + 12 * o**3 * v**3) if split else it_flop
    flop_exec = (it_flop - rep_flop) + rep_flop * world + float(ex.cpu()[0]) / steps
    res = None
    if rank == 0:
        res = {
            "value": flop_exec / sec_per_step / 1e12, "ms_per_step": sec_per_step * 1e3,
            "config": {"workload": f"{workload}: nocc={o} nvirt={v}, synthetic hashed ERIs scale {scale}; "
                                   "step = 1 CCSD iteration (replicated) + full (T) over i<=j<=k sharded across ranks",
                       "nocc": o, "nvirt": v, "triples": int(nt), "parallelism": (f"(T) ijk-shard x{world}; CCSD iteration: ring o^3v^3 products + pp-ladder split x{world} with one "
                                       f"all-reduce of [PP | residual], rest replicated" if split else f"(T) ijk-shard x{world}, CCSD replicas")},
            "ccsd_iter_s": t_iter / steps, "t_s": t_trip / steps, "flop_per_step": flop_exec,
            "fraction_of_mfma_peak": flop_exec / sec_per_step / 1e12 / (MFMA_F64_PEAK_TFLOPS * world),
            # one count per number ("report which, never mix"): the headline is what the kernels issue; the survey's
            # algorithmic count of the same step -- every contraction site of the iteration, (T) over i<=j<=k -- under a name of its own
            "flop_per_step_survey_count": flop_survey,
            "value_survey_count": flop_survey / sec_per_step / 1e12,
            "fraction_of_mfma_peak_survey_count": flop_survey / sec_per_step / 1e12 / (MFMA_F64_PEAK_TFLOPS * world),
            "rates_note": "value / fraction_of_mfma_peak: the multiply-adds the kernels issue (iteration as evaluated: pair forms; (T): "
                          "o x o(o+1)/2 blocks of 4 v^3 (v+o), half where the pair coincides), all ranks; *_survey_count: SURVEY 8(d)'s "
                          "algorithmic count of the step (iteration: every contraction site with the ladder over a<=b; (T) = "
                          "[o(o+1)(o+2)/6] 12 v^3 (v+o)) -- larger than what is executed, so its fraction can exceed the kernel's",
            "e_t": [float(x) for x in acc["last"]], "rccl_ranks": rccl_ranks, "t_allreduce": red.kind,
            "ccsd_split": bool(split), "ccsd_split_check": split_check,
        }
        nl = eng.ccsd_iteration_launches()
        if nl and world == 1 and with_roofline:
            # Small systems: the iteration above ran as the compiled sequence of grouped launches (csrc/fused.hip).  Beside it, from the
            # same state, the call-by-call evaluation on parallel streams ("laned", what a 15-30-iteration solve saw until round 4) and
            # its graph replay (captured after AFESP_GRAPH_AFTER = 40 iterations)
            def iter_time(n):
                t0 = time.perf_counter()
                for _ in range(n):
                    eng.ccsd_iterate()
                    eng.ccsd_diis()
                return (time.perf_counter() - t0) / n
            eng.ccsd_set_fused(0)
            iter_time(5)
            laned = iter_time(30)
            iter_time(10)            # (past the 40 calls after which the graph is captured)
            replayed = iter_time(30)
            eng.ccsd_set_fused(1)
            iter_time(3)
            fused = iter_time(30)
            eng.ccsd_set_fused(-1)
            res["ccsd_iter_paths"] = {"launch_fused_s": fused, "launches_per_iteration": nl, "call_by_call_laned_s": laned,
                                      "call_by_call_graph_replayed_s": replayed,
                                      "note": "launch_fused = default path of afesp_ccsd_iterate for o^2 v^2 <= 2^20 (one grouped launch per "
                                              "dependency level, csrc/fused.hip); the other two: afesp_ccsd_set_fused(0)"}
        if per_rank:
            res["per_rank"] = {"ccsd_iter_ms": [round(r[0], 4) for r in per_rank], "t_shard_ms": [round(r[1], 4) for r in per_rank],
                               "allreduce_us_incl_wait_for_slowest_rank": [round(r[2], 1) for r in per_rank],
                               "t_shard_triples": [int(r[3]) for r in per_rank]}
            res["t_shard_ms_min"] = min(r[1] for r in per_rank)
            res["t_shard_ms_max"] = max(r[1] for r in per_rank)
            res["allreduce_us_min_over_ranks"] = min(r[2] for r in per_rank)   # the last rank to arrive waits for nobody: ~ the collective itself
        if with_roofline:
            # Dominant kernel: the (T) GEMM (gett_kernel, X = tt^T vt over kappa = d + l, K = v+o), timed over the timed
            # region with HIP events on the engine's stream (csrc/triples.hip).  achieved = EXECUTED flop per launch
            # (2 M N K of the products it runs) / its average duration.
            nl = max(prof["gemm_launches"], 1)
            roof = {"bound": "mfma", "achieved": prof["gemm_flop"] / max(prof["gemm_ms"], 1e-9) / 1e9,
                    "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s"}
            roof["frac"] = roof["achieved"] / roof["peak"]
            roof["traffic"] = None
            if prof.get("gemm_kernel") == "tgemm_kernel":
                roof["kernel"] = ("tgemm_kernel (csrc/tgemm.hip: LDS-DMA staged, two 4-wave workgroups per CU on 128x128 tiles), the grouped "
                                  "(T) launches, one per chunk: Y(a;b,c|i;jk) = sum over kappa = [d + l ; d + l] of tt(kappa;a,.)*vt(kappa;b,c,.), "
                                  "K = 2(v+o); the groups with j == k run over K = v+o in the same launch (the orbit kernel adds the transpose)")
            else:
                roof["kernel"] = ("gett_kernel<..., GRP = true>, the grouped (T) launches (one per chunk): Y(a;b,c|i;jk) = sum over "
                                  "kappa = [d + l ; d + l] of tt(kappa;a,.)*vt(kappa;b,c,.), K = 2(v+o); "
                                  "pairs j == k in a second launch per chunk over K = v+o (the orbit kernel adds the transpose)")
            roof["launches"] = prof["gemm_launches"]
            roof["ms_per_launch"] = prof["gemm_ms"] / nl
            roof["flop_per_launch"] = prof["gemm_flop"] / nl
            # ... and what the tiles execute with their zero padding (tile edges in M and N, the K steps' tails): the rate of the
            # matrix pipe itself
            roof["flop_per_launch_padded"] = prof.get("gemm_flop_padded", 0.0) / nl
            roof["achieved_padded"] = prof.get("gemm_flop_padded", 0.0) / max(prof["gemm_ms"], 1e-9) / 1e9
            roof["share_of_step_time"] = prof["gemm_ms"] * 1e-3 / elapsed
            # what the traffic below is held against: every Y block written once (the orbit kernel's bytes) + both operand sets read once
            # (vt, vtT: 2 Kc v^2 o doubles; tt: Kc v o^2), per launch
            kc, vp = (v + o + 15) // 16 * 16, (v + 7) // 8 * 8
            c_bytes = 8.0 * vp**3 * (o * o * (o + 1) // 2) if world == 1 else prof["orbit_bytes"]   # distinct blocks Y^{p;qr}, q <= r (one rank: all of them)
            launches_per_t = max(nl / max(steps, 1), 1.0)   # (nl counts the launches of all timed steps)
            if world != 1:
                c_bytes /= max(steps, 1)                     # (orbit_bytes: summed over the timed steps too)
            roof["traffic_algorithmic"] = (c_bytes + 8.0 * (2 * kc * v * v * o + kc * v * o * o)) / launches_per_t
            tfile = latest_profile("traffic.json")
            live = live_pmc(workload) if (args.live_pmc and world == 1 and workload == args.workload) else None
            if live:
                roof["traffic"] = live["hbm_bytes_per_launch"]
                roof["traffic_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES | "
                                          f"GRBM_GUI_ACTIVE, one pass each, of `bench.py --workload {workload} --steps 1 --warmup 0` as child "
                                          f"processes ({live['dispatches']} grouped launches; FETCH_SIZE x2 per the gfx950 correction); "
                                          f"fetched {live['fetch_bytes_per_launch']:.4g} B + written {live['write_bytes_per_launch']:.4g} B per launch")
                roof["mfma_busy"] = live["mfma_busy_frac"]
            elif tfile:
                tr = json.load(open(tfile)).get(workload + "_t_gemm")
                if tr:
                    roof["traffic"] = tr["hbm_bytes_per_launch"]
                    roof["traffic_source"] = (os.path.relpath(tfile, ROOT) + ": separate rocprofv3 --pmc passes of this command (FETCH_SIZE x2 "
                                              "per the gfx950 correction, WRITE_SIZE); " + tr["command"])
                    if "mfma_busy_frac" in tr:
                        roof["mfma_busy"] = tr["mfma_busy_frac"]     # MFMA-pipe busy fraction (PMC), same passes
            second = {"kernel": "triples_orbit_kernel", "bound": "hbm",
                      "achieved": prof["orbit_bytes"] / max(prof["orbit_ms"], 1e-9) / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "ms_per_launch": prof["orbit_ms"] / max(prof["orbit_launches"], 1),
                      "share_of_step_time": prof["orbit_ms"] * 1e-3 / elapsed}
            second["frac"] = second["achieved"] / second["peak"]
            ms_lad = eng.time_pp_ladder(20 if o * v < 2000 else 5)
            res["roofline"] = roof
            res["roofline_second_kernel"] = second
            res["pp_ladder"] = {"ms_per_launch": ms_lad, "tflops_executed": eng.pp_ladder_flop() / (ms_lad * 1e-3) / 1e12,
                                "tflops_reference_equivalent": 2 * o**2 * v**4 / (ms_lad * 1e-3) / 1e12,
                                "algorithmic_gbs": 8 * (v**3 * (v + 1) / 2 + 2 * o**2 * v**2) / (ms_lad * 1e-3) / 1e9}
            if world == 1:
                # the completely renormalised (T) (src/ccsd.f90:2186-2194, :2338-2551) -- the variant of every bundled reference
                # output: its intermediates, then the triples with the M3 moments (a second pool of product blocks, the CR orbit kernel)
                t0 = time.perf_counter()
                eng.build_cr_intermediates()
                t_cri = time.perf_counter() - t0
                eng.do_ccsd_t_spatial_cr(lo, hi)           # (plan, pools)
                t0 = time.perf_counter()
                cr = eng.do_ccsd_t_spatial_cr(lo, hi)
                t_cr = time.perf_counter() - t0
                res["cr_t"] = {"cr_intermediates_s": t_cri, "cr_t_s": t_cr, "out": [float(x) for x in cr],
                               "max_abs_diff_of_E[T]_E(T)_to_the_plain_evaluation": float(np.max(np.abs(np.asarray(cr[:2]) - np.asarray(acc["last"][:2]))))}
            res["ao2mo"] = time_ao2mo(eng, o, v, 21 if o * v < 2000 else 5)
            if world == 1:
                # A whole calculation, so that the weights of its stages are visible (a step of this benchmark is ONE iteration beside the
                # full (T); a solve is 20-30 of them): AO->MO + MP2 (as timed above), CCSD from the MP1 amplitudes to the reference's
                # default thresholds (src/system.f90:46-50) through afesp_ccsd_solve, then the (T) of the benchmark configuration
                eng.synthetic_init(o, v, scale, seed, 8)
                t0 = time.perf_counter()
                nit_w, en_w, _ = eng.do_ccsd_spatial(100, 1e-6, 1e-7)
                t_solve = time.perf_counter() - t0
                t0 = time.perf_counter()
                eng.do_ccsd_t_spatial_plain(0, nt)
                t_tw = time.perf_counter() - t0
                res["whole_calculation"] = {"ao2mo_mp2_s": res["ao2mo"]["ms"] * 1e-3, "ccsd_solve_s": t_solve, "ccsd_iterations": int(nit_w),
                                            "ccsd_converged": bool(nit_w > 0), "ccsd_e_tol": 1e-6, "ccsd_t_tol": 1e-7,
                                            "ccsd_energy": float(en_w[nit_w]) if nit_w > 0 else None,
                                            "t_s": t_tw, "total_s": res["ao2mo"]["ms"] * 1e-3 + t_solve + t_tw,
                                            "note": "the (T) evaluation includes building its operand copies from the converged amplitudes"}
        if args.cpu_baseline and with_cpu and world == 1:   # rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(o, v, scale, seed, eng)
            res["cpu_baseline"]["vs_gpu_step"] = res["cpu_baseline"]["value"] / sec_per_step
    barrier()
    red.close()
    eng.close()
    return res


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`--gpus N` with no launcher: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun would)
    and wait for them.  Runs before this process has imported torch or touched the GPU; nothing is exec'ed."""
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    jobdir = tempfile.mkdtemp(prefix="afesp_bench_")
    env["AFESP_BENCH_JOBDIR"] = jobdir
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + sys.argv[1:],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    codes = [p.wait() for p in procs]
    try:
        for f in os.listdir(jobdir):
            os.unlink(os.path.join(jobdir, f))
        os.rmdir(jobdir)
    except OSError:
        pass
    return next((c for c in codes if c != 0), 0)


def key_paths(d, prefix=""):
    out = set()
    for k, v in d.items():
        out.add(prefix + k)
        if isinstance(v, dict):
            out |= key_paths(v, prefix + k + ".")
    return out


def dry_ranks(n, argv):
    """`--dry-ranks N`: walk the N > 1 control flow on ONE GPU before a multi-GPU node ever sees it -- N ranks of this script over the
    host-segment transport (shard bounds, block-size agreement, the split self-check, the per-rank fields, every same-run leg) on a
    mid-size system past the small-system switch (o = 12, v = 120: the split of the iteration is attempted), then the same command on one rank -- and check that the N-rank line carries every key of the one-rank line
    (`cpu_baseline`, the launch-path comparison, the CR-(T) timing and `whole_calculation` are one-rank measurements by contract).  Prints one JSON report; exit code 1 if
    a key is missing.  At most 6 ranks: a GPU box admits six processes on its card."""
    if not 2 <= n <= 6:
        raise SystemExit("bench.py --dry-ranks: 2 ... 6 ranks (one GPU admits six processes)")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mid_large", "--steps", "2", "--warmup", "1", "--no-live-pmc"]
    lines = {}
    for tag, extra in (("one_rank", ["--gpus", "1"]), ("n_ranks", ["--gpus", str(n), "--backend", "gloo"])):
        r = subprocess.run(base + extra, capture_output=True, text=True, env=dict(os.environ, AFESP_CC_SHARD=os.environ.get("AFESP_CC_SHARD", "1")))
        js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not js:
            print(json.dumps({"dry_ranks": n, "failed": tag, "returncode": r.returncode, "stderr_tail": r.stderr[-2000:]}))
            return 1
        lines[tag] = json.loads(js[-1])
    one_rank_only = tuple(pre + k for pre in ("", "h2o_tz_same_run.") for k in ("cpu_baseline", "ccsd_iter_paths", "cr_t", "whole_calculation"))
    missing = sorted(k for k in key_paths(lines["one_rank"]) - key_paths(lines["n_ranks"])
                     if not any(k == p or k.startswith(p + ".") for p in one_rank_only))
    rep = {"dry_ranks": n, "missing_keys": missing, "n_rank_only_keys": sorted(key_paths(lines["n_ranks"]) - key_paths(lines["one_rank"])),
           "n_ranks_line": lines["n_ranks"]}
    print(json.dumps(rep))
    return 1 if missing else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)      # a config-5 step is ~0.5 s
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg5", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", dest="cpu_baseline", action="store_false")
    ap.add_argument("--no-extra", dest="extra", action="store_false",
                    help="skip the additional measurements appended to the line: the H2O/cc-pVTZ shape (config 2), the bundled "
                         "N2 / F2 inputs (configs 3 / 4) with their energy check, the spin-orbital H2O/cc-pVTZ shape")
    ap.add_argument("--scale", type=float, default=None)
    ap.add_argument("--steps-only", dest="legs", action="store_false",
                    help="only the timed steps: no pp-ladder / AO->MO timing legs behind them (kernel traces of the steps alone)")
    ap.add_argument("--no-live-pmc", dest="live_pmc", action="store_false",
                    help="take roofline.traffic / mfma_busy from the newest profiles/rNN_traffic.json instead of measuring them in "
                         "this run (four rocprofv3 --pmc passes of this command as child processes, N = 1 only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only to rehearse the multi-rank path on a one-GPU box")
    ap.add_argument("--dry-ranks", type=int, default=0,
                    help="rehearse the N-rank control flow on one GPU (host transport, mid-size system) and check the emitted line's keys "
                         "against a one-rank run: see dry_ranks()")
    args = ap.parse_args()

    if args.dry_ranks:
        sys.exit(dry_ranks(args.dry_ranks, sys.argv[1:]))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); refusing to "
                         "report a line for a different number of GPUs than asked for")
    jobdir = os.environ.get("AFESP_BENCH_JOBDIR") or os.path.join(tempfile.gettempdir(), "afesp_bench_" + os.environ.get("MASTER_PORT", "0"))
    os.makedirs(jobdir, exist_ok=True)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            local = 0                                   # rehearsal: every rank shares the one visible GPU
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)
    cdev = "cpu" if (world > 1 and args.backend == "gloo") else f"cuda:{local}"   # where torch's collectives run

    res = measure(args, args.workload, args.steps, args.warmup, rank, world, local, dist, cdev, torch, jobdir, with_roofline=args.legs)
    others = {}
    if args.extra:
        if args.workload != "h2o_tz":
            # BASELINE config 2 shape: a step is ~0.6 ms, latency-bound (no roofline meaning): 50 steps, after enough warm-up
            # for the engine to have captured the iteration's graph (it waits for 40 calls: a real solve is shorter)
            others["h2o_tz_same_run"] = measure(args, "h2o_tz", 50, 45, rank, world, local, dist, cdev, torch, jobdir,
                                                with_roofline=True, with_cpu=True)
        others["real_molecules_same_run"] = {name: real_molecule(name, rank, world, local, dist, cdev, torch, args.backend, jobdir)
                                             for name in ("n2-cc-pvdz", "f2-cc-pvdz")}
        others["spinorb_h2o_tz_same_run"] = spinorb_h2o_tz(rank, world, local, dist, cdev, torch, args.backend, jobdir)
    if rank == 0:
        line = {"metric": "CCSD iter wall-time (s) + (T) wall-time (s); fp64 TFLOP/s vs MFMA peak",
                "value": res.pop("value"), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic"}
        line.update(res)
        for key, val in others.items():
            if key == "h2o_tz_same_run" and val is not None:
                val = dict(val, unit="TFLOP/s", steps=50, warmup=45,
                           integrals="synthetic hashed ERIs of the H2O/cc-pVTZ shape: the reference tree has no eri.dat for BASELINE "
                                     "config 2 (.MISSING_LARGE_BLOBS), so its shipped energies cannot be reproduced")
            line[key] = val
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
