"""GPU parity of the spin-orbital CCSD / CCSD(T) path (do_ccsd_spinorb, do_ccsd_t_spinorb) through the C-ABI."""
import os
import re

import numpy as np
import pytest

import molecules
import orc

pytestmark = pytest.mark.gpu

SO_FIELDS = ["tau", "tau_tilde", "F_vv", "F_oo", "F_ov", "W_oooo", "W_vvvv", "W_ovvo"]


@pytest.fixture(scope="module")
def eng():
    from afesp_amd.capi import Engine
    e = Engine(0)
    yield e
    e.close()


def _ref_out_table():
    it, final = [], None
    for line in open(os.path.join(molecules.GOLDEN, "h2o-cc-pvdz", "ref_out")):
        m = re.match(r"\s*Iteration\s+\d+\s+(-0\.\d{12})\s+[\d.]+ s", line)
        if m:
            it.append(float(m.group(1)))
        m = re.match(r"\s*Final CCSD Energy \(Hartree\):\s+(-?\d+\.\d+)", line)
        if m:
            final = float(m.group(1))
    return it, final


def test_h2o_spinorbital_run_matches_shipped_ref_out_and_oracle(eng):
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    n, nel = ints.nbasis, ints.nel
    gold_it, gold_final = _ref_out_table()
    e_mp2, eri_mo = eng.do_mp2_spatial(n, nel // 2, res.canon_coeff, res.canon_levels, ints.eri)
    # the shipped run (Feb 2022): F_mi in Stanton's index order, 8 DIIS vectors, tolerances 1e-6 / 1e-7
    eng.init_cc_spinorb(n, nel, res.canon_levels, None, 8, foo_as_published=True)
    nit, en, rm = eng.do_ccsd_spinorb(40, 1e-6, 1e-7)
    assert nit == 19
    np.testing.assert_allclose(en[1:20], gold_it, rtol=0, atol=1e-10)
    assert abs(en[nit] - gold_final) < 1e-10
    assert abs(en[0] - e_mp2) < 1e-11          # the MP1 line is the MP2 energy
    # as coded today (ccsd.f90:789-794): against the oracle, iteration by iteration, then (T)
    eng.init_cc_spinorb(n, nel, res.canon_levels, None, 8)
    so = orc.OracleSO(n, nel, orc.ao2mo(n, res.canon_coeff, ints.eri), res.canon_levels, 8)
    nit, en, rm = eng.do_ccsd_spinorb(60, 1e-9, 1e-9)
    onit, oen, orm = so.solve(60, 1e-9, 1e-9)
    assert nit == onit
    assert np.max(np.abs(en[:nit + 1] - oen[:nit + 1])) < 1e-10
    assert np.max(np.abs(rm[:nit + 1] - orm[:nit + 1])) < 1e-10
    t1, t2 = eng.so_amplitudes()
    assert np.max(np.abs(t1 - so.t1)) < 1e-9 and np.max(np.abs(t2 - so.t2)) < 1e-9
    e_t = eng.do_ccsd_t_spinorb()
    assert abs(e_t - so.triples()) < 1e-10
    # shards add up
    nt = eng.so_ntriples()
    assert nt == nel * (nel - 1) * (nel - 2) // 6
    parts = [eng.do_ccsd_t_spinorb(a, b) for a, b in ((0, nt // 3), (nt // 3, nt // 2), (nt // 2, nt))]
    assert abs(sum(parts) - e_t) < 1e-12


@pytest.mark.parametrize("n,nel", [(6, 4), (7, 2), (9, 6), (5, 8)])
def test_spinorbital_terms_and_triples_match_oracle_on_synthetic_systems(eng, n, nel):
    """Every intermediate after two iterations (t1 != 0 from the second on), then converged energies and (T)."""
    o = nel // 2
    _, e, eri = molecules.synthetic_system(o, n - o, scale=0.05, seed=7 + n)
    for pub in (False, True):
        eng.init_cc_spinorb(n, nel, e, eri, 4, foo_as_published=pub)
        so = orc.OracleSO(n, nel, eri, e, 4, foo_as_published=pub)
        assert np.max(np.abs(eng.so_tensor("oovv") - so.field("oovv"))) < 1e-14
        assert np.max(np.abs(eng.so_tensor("vvvv") - so.field("vvvv"))) < 1e-14
        eng.so_energy(); so.energy_step(1e-9, 1e-9)
        for _ in range(2):
            eng.so_iterate(); so.iterate(); so.energy_step(1e-9, 1e-9)
        for f in SO_FIELDS:
            assert np.max(np.abs(eng.so_tensor(f) - so.field(f))) < 1e-12, f
        t1, t2 = eng.so_amplitudes()
        assert np.max(np.abs(t1 - so.t1)) < 1e-12 and np.max(np.abs(t2 - so.t2)) < 1e-12
    eng.init_cc_spinorb(n, nel, e, eri, 4)
    so = orc.OracleSO(n, nel, eri, e, 4)
    nit, en, rm = eng.do_ccsd_spinorb(80, 1e-10, 1e-10)
    onit, oen, orm = so.solve(80, 1e-10, 1e-10)
    assert nit == onit and nit > 0
    assert np.max(np.abs(en[:nit + 1] - oen[:nit + 1])) < 1e-11
    if nel >= 3:
        assert abs(eng.do_ccsd_t_spinorb() - so.triples()) < 1e-11
    else:
        assert eng.so_ntriples() == 0 and eng.do_ccsd_t_spinorb() == 0.0


def test_spinorbital_triples_from_spin_expanded_amplitudes_equal_spin_free_triples(eng):
    """Same quantity through two independent device paths: the spin-free (T) kernel and the spin-orbital one."""
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    n, o = ints.nbasis, ints.nel // 2
    eng.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, ints.eri)
    eng.ccsd_init(o, n - o, res.canon_levels, None, 8)
    nit, en, _ = eng.do_ccsd_spatial(60, 1e-10, 1e-10)
    out = eng.do_ccsd_t_spatial()
    t1, t2 = eng.amplitudes()
    T1 = np.zeros((2 * o, 2 * (n - o)))
    T2 = np.zeros((2 * o, 2 * o, 2 * (n - o), 2 * (n - o)))
    for s1 in (0, 1):
        T1[s1::2, s1::2] = t1
        for s2 in (0, 1):
            T2[s1::2, s2::2, s1::2, s2::2] += t2
            T2[s1::2, s2::2, s2::2, s1::2] -= t2.transpose(0, 1, 3, 2)
    eng.init_cc_spinorb(n, 2 * o, res.canon_levels, None, 8)
    eng.so_set_amplitudes(T1, T2)
    e, _, _ = eng.so_energy()
    assert abs(e - en[nit]) < 1e-12
    assert abs(eng.do_ccsd_t_spinorb() - out[1]) < 1e-12


def test_spinorbital_errors(eng):
    from afesp_amd.capi import AfespError
    _, e, eri = molecules.synthetic_system(2, 3)
    with pytest.raises(AfespError):
        eng.init_cc_spinorb(5, 3, e, eri)       # odd electron count
    with pytest.raises(AfespError):
        eng.init_cc_spinorb(5, 10, e, eri)      # no virtual orbital


def test_cached_triples_plans_of_both_solvers_do_not_clobber_each_other(eng):
    """(T) of the spin-free and of the spin-orbital solver alternately in one context: each keeps a cached launch plan."""
    o, v = 3, 7
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05, seed=5)
    eng.ccsd_init(o, v, e, eri, 4)
    eng.do_ccsd_spatial(40, 1e-9, 1e-9)
    eng.init_cc_spinorb(n, 2 * o, e, eri, 4)
    eng.do_ccsd_spinorb(60, 1e-9, 1e-9)
    a1 = eng.do_ccsd_t_spatial()
    b1 = eng.do_ccsd_t_spinorb()
    a2 = eng.do_ccsd_t_spatial()
    b2 = eng.do_ccsd_t_spinorb()
    assert np.array_equal(a1, a2) and b1 == b2
    # re-initialising one solver frees the context's cached scratch buffers: the other solver's plan must notice
    eng.init_cc_spinorb(n, 2 * o, e, eri, 4)
    eng.do_ccsd_spinorb(60, 1e-9, 1e-9)
    a3 = eng.do_ccsd_t_spatial()
    assert np.max(np.abs(a3 - a1)) < 1e-14
    assert abs(eng.do_ccsd_t_spinorb() - b1) < 1e-14
