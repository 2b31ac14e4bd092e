"""Pins the CPU restatement (oracle/) to the reference's own bundled outputs."""
import numpy as np
import pytest

import molecules
import orc


@pytest.mark.parametrize("name", ["n2-cc-pvdz", "f2-cc-pvdz"])
def test_oracle_reproduces_bundled_els_out(name):
    si, ints, res, gold = molecules.load(name)
    assert res.converged
    # RHF table (hf.f90:110-113 prints F15.10) and total energy (main.F90:125)
    assert len(res.iters) == len(gold["scf_iters"])
    for (it, e, de, rms), (git, ge, gde, grms) in zip(res.iters, gold["scf_iters"]):
        assert it == git and abs(e - ge) < 2e-9 and abs(rms - grms) < 2e-9
    assert abs(res.e_hf + ints.e_nuc - gold["rhf_total"]) < 1e-9
    n, o = ints.nbasis, ints.nel // 2
    v = n - o
    for k, ge in gold["orbital_energies"].items():
        assert abs(res.canon_levels[k - 1] - ge) < 1e-7
    mo = orc.ao2mo(n, res.canon_coeff, ints.eri)
    assert abs(orc.mp2_energy(n, o, mo, res.canon_levels) - gold["mp2_corr"]) < 1e-9
    cc = orc.OracleCC(o, v, mo, res.canon_levels, si.ccsd_diis_n_errmat)
    nit, en, rm = cc.solve(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    # every iteration line of the bundled output (ccsd.f90:362-363, F15.12): pins equations + DIIS path
    assert nit == gold["cc_iters"][-1][0]
    for (git, ge, gde, grms) in gold["cc_iters"]:
        assert abs(en[git] - ge) < 5e-12, (git, en[git], ge)
        assert abs(rm[git] - grms) < 5e-12
    assert abs(cc.energy - gold["final_ccsd"]) < 5e-12
    t = cc.triples(res.canon_levels)
    ec = cc.energy
    assert abs(ec + t[0] - gold["ccsd_bt_corr"]) < 1e-9
    assert abs(ec + t[1] - gold["ccsd_pt_corr"]) < 1e-9
    assert abs(ec + t[0] / t[2] - gold["r_ccsd_bt_corr"]) < 1e-9
    assert abs(ec + t[1] / t[3] - gold["r_ccsd_pt_corr"]) < 1e-9
    assert abs(t[2] - gold["d_bt"]) < 1e-9 and abs(t[3] - gold["d_pt"]) < 1e-9
    assert abs(cc.L.orc_cc_t1_diagnostic(cc.h, ints.nel) - gold["t1_diag"]) < 1e-9
    # completely renormalised variants (ccsd.f90:2338-2551, :2186-2194): the calc_type of the bundled runs
    cc.cr_intermediates()
    tc = cc.triples_cr(res.canon_levels)
    assert np.allclose(tc[:4], t, atol=1e-12)
    assert abs(ec + tc[4] / tc[2] - gold["cr_ccsd_bt_corr"]) < 1e-9
    assert abs(ec + tc[5] / tc[3] - gold["cr_ccsd_pt_corr"]) < 1e-9
    assert abs(res.e_hf + ints.e_nuc + ec + tc[5] / tc[3] - gold["total"]) < 2e-9


def test_oracle_h2o_matches_survey_recorded_reference_run():
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    g = molecules.SURVEY_GOLD["h2o-cc-pvdz"]
    n, o = ints.nbasis, ints.nel // 2
    assert abs(res.e_hf + ints.e_nuc - g["rhf_total"]) < 1e-9
    mo = orc.ao2mo(n, res.canon_coeff, ints.eri)
    assert abs(orc.mp2_energy(n, o, mo, res.canon_levels) - g["mp2_corr"]) < 1e-9
    cc = orc.OracleCC(o, n - o, mo, res.canon_levels, si.ccsd_diis_n_errmat)
    nit, en, rm = cc.solve(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    assert nit > 0 and abs(cc.energy - g["ccsd_corr"]) < 1e-9
    t = cc.triples(res.canon_levels)
    assert abs(cc.energy + t[0] - g["ccsd_bt_corr"]) < 1e-9
    assert abs(cc.energy + t[1] - g["ccsd_pt_corr"]) < 1e-9
    assert abs(t[2] - g["d_bt"]) < 1e-9 and abs(t[3] - g["d_pt"]) < 1e-9


def test_triples_range_is_additive():
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    n, o = ints.nbasis, ints.nel // 2
    mo = orc.ao2mo(n, res.canon_coeff, ints.eri)
    cc = orc.OracleCC(o, n - o, mo, res.canon_levels, 8)
    cc.solve(50, 1e-6, 1e-7)
    full = cc.triples(res.canon_levels)
    parts = sum(cc.triples(res.canon_levels, b, min(b + 37, o ** 3)) for b in range(0, o ** 3, 37))
    assert np.allclose(full, parts, atol=1e-12)


def test_blas_backed_triples_equal_the_pinned_loops():
    """oracle/afesp_oracle_blas.c (one dgemm per permuted term, the reference's shape; bench.py's CPU baseline) against the loop
    form that the bundled outputs pin, on H2O and on a shard of the list."""
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    n, o = ints.nbasis, ints.nel // 2
    mo = orc.ao2mo(n, res.canon_coeff, ints.eri)
    cc = orc.OracleCC(o, n - o, mo, res.canon_levels, 8)
    cc.solve(50, 1e-6, 1e-7)
    full = cc.triples_blas(res.canon_levels)
    if full is None:
        pytest.skip("numpy's bundled OpenBLAS not found")
    assert np.allclose(full, cc.triples(res.canon_levels), atol=1e-12, rtol=0)
    assert np.allclose(cc.triples_blas(res.canon_levels, 17, 61), cc.triples(res.canon_levels, 17, 61), atol=1e-12, rtol=0)
    # the ladder-shaped product
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal((12, 30)), rng.standard_normal((30, 9))
    c = np.zeros(12 * 9)
    assert orc.blas_lib().orcb_gemm(12, 9, 30, 0.5, np.ascontiguousarray(a.ravel(order="F")), np.ascontiguousarray(b.ravel(order="F")), 0.0, c, 2) == 0
    assert np.allclose(c.reshape((12, 9), order="F"), 0.5 * a @ b, atol=1e-13)


def test_pack_unpack_roundtrip_and_canonical_order():
    L = orc.lib()
    n = 7
    packed = np.random.default_rng(0).standard_normal(L.orc_neri(n))
    full = np.zeros(n ** 4)
    L.orc_unpack_eri(n, packed, full)
    f = full.reshape((n,) * 4, order="F")
    assert np.array_equal(f, f.transpose(1, 0, 2, 3)) and np.array_equal(f, f.transpose(2, 3, 0, 1))
    back = np.zeros_like(packed)
    L.orc_pack_eri(n, full, back)
    assert np.array_equal(back, packed)


def _ref_out_spinorb_table():
    """Iteration energies and the final CCSD energy of the spin-orbital run shipped as h2o-cc-pvdz/.../ref_out."""
    import os
    import re
    it, final = [], None
    for line in open(os.path.join(molecules.GOLDEN, "h2o-cc-pvdz", "ref_out")):
        m = re.match(r"\s*Iteration\s+\d+\s+(-0\.\d{12})\s+[\d.]+ s", line)
        if m:
            it.append(float(m.group(1)))
        m = re.match(r"\s*Final CCSD Energy \(Hartree\):\s+(-?\d+\.\d+)", line)
        if m:
            final = float(m.group(1))
    return it, final


def test_spinorbital_oracle_reproduces_shipped_ref_out():
    """ref_out was produced with ccsd_e_tol 1e-6, ccsd_t_tol 1e-7 and 8 DIIS vectors (its header and system.f90:46-50).
    It predates the dgemm form of build_F: the tau~ term of F_mi is in Stanton's index order there, see
    oracle/afesp_oracle_so.c so_F -- with that order all 19 printed energies are reproduced to 12 decimals."""
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    gold_it, gold_final = _ref_out_spinorb_table()
    assert len(gold_it) == 19
    eri_mo = orc.ao2mo(ints.nbasis, res.canon_coeff, ints.eri)
    s = orc.OracleSO(ints.nbasis, ints.nel, eri_mo, res.canon_levels, 8, foo_as_published=True)
    nit, en, _ = s.solve(40, 1e-6, 1e-7)
    assert nit == 19
    np.testing.assert_allclose(en[1:20], gold_it, rtol=0, atol=2e-12)
    assert abs(s.energy - gold_final) < 2e-12
    assert abs(en[0] - molecules.SURVEY_GOLD["h2o-cc-pvdz"]["mp2_corr"]) < 1e-9   # the "MP1" line is the MP2 energy
    # as coded today (ccsd.f90:791-794) the same term is accumulated transposed: a different fixed point
    s2 = orc.OracleSO(ints.nbasis, 10, eri_mo, res.canon_levels, 8)
    nit2, en2, _ = s2.solve(40, 1e-6, 1e-7)
    assert nit2 > 0 and abs(en2[1] - gold_it[0]) < 2e-12 and 1e-6 < abs(s2.energy - gold_final) < 1e-4


def _spin_expand(t1, t2):
    o, v = t1.shape
    T1 = np.zeros((2 * o, 2 * v))
    T2 = np.zeros((2 * o, 2 * o, 2 * v, 2 * v))
    for s1 in (0, 1):
        T1[s1::2, s1::2] = t1
        for s2 in (0, 1):
            T2[s1::2, s2::2, s1::2, s2::2] += t2
            T2[s1::2, s2::2, s2::2, s1::2] -= t2.transpose(0, 1, 3, 2)
    return T1, T2


def test_spinorbital_triples_equal_spin_free_triples_on_the_same_amplitudes():
    """do_ccsd_t_spinorb and do_ccsd_t_spatial evaluate the same quantity: feed the spin-orbital restatement the
    spin-expanded converged spin-free amplitudes (pinned by the N2/F2 goldens) -> E(T) must agree to rounding."""
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    n, o = ints.nbasis, 5
    eri_mo = orc.ao2mo(n, res.canon_coeff, ints.eri)
    cc = orc.OracleCC(o, n - o, eri_mo, res.canon_levels, 8)
    cc.solve(60, 1e-11, 1e-11)
    out = cc.triples(res.canon_levels)
    so = orc.OracleSO(n, 2 * o, eri_mo, res.canon_levels, 8)
    T1, T2 = _spin_expand(np.array(cc.t1), np.array(cc.t2))
    so.t1[...] = T1
    so.t2[...] = T2
    e, _, _ = so.energy_step(1.0, 1.0)
    assert abs(e - cc.energy) < 1e-13
    assert abs(so.triples() - out[1]) < 1e-13


def test_loop_sites_of_the_cpu_baseline_equal_their_defining_sums():
    """oracle/afesp_oracle_blas.c, the two OpenMP loop nests of the reference's iteration (ccsd.f90:1170-1182, :1680-1695) that
    bench.py times slab by slab as the CPU baseline's loop sites: slabs add up to the whole, and the whole is the defining sum."""
    L = orc.lib()
    o, v = 3, 5
    rng = np.random.default_rng(11)
    f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
    v_oovv, c_oovv, t2, asym = (rng.standard_normal((o, o, v, v)) for _ in range(4))
    I_ovov = rng.standard_normal((o, v, o, v))
    I_voov = rng.standard_normal((v, o, o, v))
    got = f(I_ovov).copy()
    assert L.orcb_ring_I_ovov(o, v, f(v_oovv), f(c_oovv), got, 0, 2) == 0 and L.orcb_ring_I_ovov(o, v, f(v_oovv), f(c_oovv), got, 2, v) == 0
    ref = I_ovov - 0.5 * np.einsum("mibe,mjae->jbia", v_oovv, c_oovv)
    assert np.max(np.abs(got - f(ref))) < 1e-13
    out = np.zeros(o * o * v * v)
    assert L.orcb_ring_t2(o, v, f(t2), f(asym), f(I_ovov), f(I_voov), out, 0, 1) == 0
    assert L.orcb_ring_t2(o, v, f(t2), f(asym), f(I_ovov), f(I_voov), out, 1, v) == 0
    ref = (-np.einsum("mjae,iemb->ijab", t2, I_ovov) - np.einsum("iema,mjeb->ijab", I_ovov, t2)
           + np.einsum("miea,ejmb->ijab", asym, I_voov))
    assert np.max(np.abs(out - f(ref))) < 1e-13


def test_numpy_restatement_equals_the_loop_form():
    """tests/np_cc.py (one dgemm per o^3 v^3 sum, the pp-ladder on sampled column pairs: what tools/big_system_check.py holds the device
    against at sizes past config 5) against the loop-form restatement that the reference's bundled outputs pin, from amplitudes with
    t1 != 0: every intermediate, r1, and r2 / the updated t2 on every column pair."""
    import molecules
    import np_cc
    import orc
    o, v = 3, 7
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    cc = orc.OracleCC(o, v, eri, e, 4)
    rng = np.random.default_rng(5)
    t1 = 0.05 * rng.standard_normal((o, v))
    t2 = 0.05 * rng.standard_normal((o, o, v, v))
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    cc.t1[...] = t1; cc.t2[...] = t2
    cc.L.orc_cc_intermediates(cc.h)
    f = lambda name: np.array(cc.field(name))
    oovv, ovov, vvov, oovo, oooo, vvvv = (f(k) for k in ("v_oovv", "v_ovov", "v_vvov", "v_oovo", "v_oooo", "v_vvvv"))
    I = np_cc.intermediates(t1, t2, oovv, ovov, vvov, oovo, oooo)
    for name in ("asym_t2", "c_oovv", "I_vo", "I_vv", "I_oo_p", "I_oo", "I_oooo", "I_ovov", "I_voov", "x_voov", "I_ooov_p"):
        assert np.max(np.abs(I[name] - f(name))) < 1e-13, name
    ref_vovv = f("I_vovv_p")
    for a in range(v):
        for b in range(v):
            assert np.max(np.abs(np_cc.vovv_p_cols(t1, oovv, ovov, vvov, a, b) - ref_vovv[:, :, a, b])) < 1e-13
    cc.L.orc_cc_amplitudes(cc.h)
    assert np.max(np.abs(np_cc.r1(t1, I, oovv, ovov, vvov, oovo) - f("r1"))) < 1e-13
    r2, D2, t2_new = f("r2"), f("D2"), np.array(cc.t2)
    for a in range(v):
        for b in range(v):
            rab = np_cc.r2_cols(t1, t2, I, oovv, ovov, vvov, vvvv[:, :, a, b], a, b)
            rba = np_cc.r2_cols(t1, t2, I, oovv, ovov, vvov, vvvv[:, :, b, a], b, a)
            assert np.max(np.abs(rab - r2[:, :, a, b])) < 1e-13, (a, b)
            assert np.max(np.abs(np_cc.new_t2_cols(rab, rba, oovv, D2, a, b) - t2_new[:, :, a, b])) < 1e-13, (a, b)
