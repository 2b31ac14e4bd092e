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
