"""GPU parity of AO->MO, CCSD and (T) against the oracle and the reference's bundled outputs, through the C-ABI."""
import numpy as np
import pytest

import molecules
import orc

pytestmark = pytest.mark.gpu

def _p(r2):
    """The residual as it enters the amplitudes, r2(ijab) + r2(jiba) (P(ia/jb), src/ccsd.f90:1720-1728): the engine holds some terms as
    their images under (i <-> j, a <-> b) (csrc/ccsd.hip, z_ooov), which only this sum is invariant under."""
    return r2 + r2.transpose(1, 0, 3, 2)


INTERMEDIATES = ["asym_t2", "c_oovv", "I_vo", "I_vv", "I_oo_p", "I_oo", "I_oooo", "I_ovov", "I_voov", "I_vovv_p", "x_voov",
                 "I_ooov_p"]
SLICES = ["v_oovv", "v_ovov", "v_vvov", "v_oovo", "v_oooo", "v_vvvv", "D1", "D2"]


@pytest.fixture(scope="module")
def eng():
    from afesp_amd.capi import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("name", ["h2o-cc-pvdz", "n2-cc-pvdz", "f2-cc-pvdz"])
def test_molecule_full_path_matches_reference(eng, name):
    si, ints, res, gold = molecules.load(name)
    g = dict(molecules.SURVEY_GOLD[name])
    n, o = ints.nbasis, ints.nel // 2
    v = n - o
    # --- AO->MO + MP2 (mp2.f90:261-449): packed MO integrals bit-for-bit comparable to the oracle within rounding
    e_mp2, eri_mo = eng.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, ints.eri)
    ref_mo = orc.ao2mo(n, res.canon_coeff, ints.eri)
    assert np.max(np.abs(eri_mo - ref_mo)) < 1e-11
    assert abs(e_mp2 - g["mp2_corr"]) < 1e-9
    # --- CCSD from the device-resident integrals
    eng.ccsd_init(o, v, res.canon_levels, None, si.ccsd_diis_n_errmat)
    cc = orc.OracleCC(o, v, ref_mo, res.canon_levels, si.ccsd_diis_n_errmat)
    for s in SLICES:
        assert np.max(np.abs(eng.tensor(s) - cc.field(s))) < 1e-11, s
    nit, en, rm = eng.do_ccsd_spatial(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    onit, oen, orm = cc.solve(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    assert nit == onit
    assert np.max(np.abs(en[:nit + 1] - oen[:nit + 1])) < 1e-10
    assert abs(en[nit] - g["ccsd_corr"]) < 1e-8          # north_star tolerance: 1e-8 Eh
    if gold:
        # the reference's own iteration table (ccsd.f90:362-363), printed with 12 decimals
        assert nit == gold["cc_iters"][-1][0]
        for (git, ge, gde, grms) in gold["cc_iters"]:
            assert abs(en[git] - ge) < 1e-10 and abs(rm[git] - grms) < 1e-10
    t1, t2 = eng.amplitudes()
    assert np.max(np.abs(t1 - cc.t1)) < 1e-9 and np.max(np.abs(t2 - cc.t2)) < 1e-9
    # --- (T)
    out = eng.do_ccsd_t_spatial()
    ref = cc.triples(res.canon_levels)
    assert np.max(np.abs(out - ref)) < 1e-10
    ec = en[nit]
    assert abs(ec + out[0] - g["ccsd_bt_corr"]) < 1e-8
    assert abs(ec + out[1] - g["ccsd_pt_corr"]) < 1e-8
    assert abs(ec + out[1] / out[3] - g["r_ccsd_pt_corr"]) < 1e-8
    assert abs(out[2] - g["d_bt"]) < 1e-8 and abs(out[3] - g["d_pt"]) < 1e-8
    # sharded evaluation (what the ranks of a multi-GPU run do) sums to the same numbers
    nt = eng.ntriples()
    parts = sum(eng.do_ccsd_t_spatial(b, min(b + 13, nt)) for b in range(0, nt, 13))
    assert np.max(np.abs(parts - out)) < 1e-12


@pytest.mark.parametrize("name", ["n2-cc-pvdz", "f2-cc-pvdz"])
@pytest.mark.parametrize("ring,tail", [("1", "1"), ("1000000", "1"), ("1", "0")])
def test_large_system_path_walks_the_reference_iteration_table(eng, name, ring, tail, monkeypatch):
    """The path config 5 takes (one stream of whole-tensor products; the o^3 v^3 ring products as two launches of the LDS-DMA GEMM,
    csrc/ring.hip; update + energy + DIIS push in one pass) on the bundled molecules: every CCSD iteration energy and rms of the
    reference's own els.out (src/ccsd.f90:362-363, 12 decimals), with the ring products on the gather kernel and with the
    three-kernel tail as well."""
    monkeypatch.setenv("AFESP_SMALL_MAX", "0")
    monkeypatch.setenv("AFESP_RING_TG_MIN", ring)
    monkeypatch.setenv("AFESP_LARGE_TAIL", tail)
    si, ints, res, gold = molecules.load(name)
    n, o = ints.nbasis, ints.nel // 2
    v = n - o
    eng.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, ints.eri, want_eri_mo=False)
    eng.ccsd_init(o, v, res.canon_levels, None, si.ccsd_diis_n_errmat)
    nit, en, rm = eng.do_ccsd_spatial(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    assert nit == gold["cc_iters"][-1][0]
    for (git, ge, gde, grms) in gold["cc_iters"]:
        assert abs(en[git] - ge) < 1e-10 and abs(rm[git] - grms) < 1e-10, (git, en[git], ge)
    assert abs(en[nit] - molecules.SURVEY_GOLD[name]["ccsd_corr"]) < 1e-8
    # the step-by-step entry points (afesp_ccsd_iterate + afesp_ccsd_diis) walk the same table
    eng.ccsd_init(o, v, res.canon_levels, None, si.ccsd_diis_n_errmat)
    eng.ccsd_energy(si.ccsd_e_tol, si.ccsd_t_tol)
    for (git, ge, gde, grms) in gold["cc_iters"][1:6]:
        e, r, _ = eng.ccsd_iterate(si.ccsd_e_tol, si.ccsd_t_tol)
        eng.ccsd_diis()
        assert abs(e - ge) < 1e-10, (git, e, ge)


def test_same_shape_state_is_initialised_again_where_it_lies():
    """A scan over geometries (utils/els_wrapper.py) re-enters afesp_ccsd_init with the same extents: the state keeps its buffers and its
    compiled programs (no second recording), and walks the reference's iteration table again -- also after other amplitudes, a (T)
    evaluation and a system of another shape in between."""
    from afesp_amd.capi import Engine
    with Engine(0) as e:
        tables = {}
        for name in ("n2-cc-pvdz", "n2-cc-pvdz", "f2-cc-pvdz", "n2-cc-pvdz", "n2-cc-pvdz"):
            si, ints, res, gold = molecules.load(name)
            n, o = ints.nbasis, ints.nel // 2
            v = n - o
            e.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, ints.eri, want_eri_mo=False)
            e.ccsd_init(o, v, res.canon_levels, None, si.ccsd_diis_n_errmat)
            nit, en, rm = e.do_ccsd_spatial(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
            assert nit == gold["cc_iters"][-1][0]
            for (git, ge, gde, grms) in gold["cc_iters"]:
                assert abs(en[git] - ge) < 1e-10 and abs(rm[git] - grms) < 1e-10, (name, git)
            out = e.do_ccsd_t_spatial()
            g = molecules.SURVEY_GOLD[name]
            assert abs(en[nit] + out[1] - g["ccsd_pt_corr"]) < 1e-8
            if name in tables:   # bit for bit the first walk
                assert np.array_equal(tables[name][0], en[:nit + 1]) and np.array_equal(tables[name][1], out)
            tables[name] = (en[:nit + 1].copy(), out.copy())
            # what a state may have grown on request before it is initialised again: the CR moments, the <ef|ab> slice
            e.build_cr_intermediates()
            cr = e.do_ccsd_t_spatial_cr()
            assert abs(en[nit] + cr[5] / cr[3] - gold["cr_ccsd_pt_corr"]) < 1e-8
            vvvv = e.tensor("v_vvvv")
            assert np.max(np.abs(vvvv - vvvv.transpose(1, 0, 3, 2))) == 0.0


def test_amplitudes_set_again_between_intermediates_and_update(eng, monkeypatch):
    """A large system keeps I_ovov / I_voov and copies of the amplitudes in the layout of its ring launches (csrc/ring.hip).
    afesp_ccsd_set_amplitudes between afesp_ccsd_update_intermediates and afesp_ccsd_update_amplitudes -- here with the same values: the
    two calls are one update of ONE set of amplitudes (include/afesp.h) -- turns them back into the reference's layout, and the update
    then runs on the gather kernel: same intermediates, same residuals."""
    monkeypatch.setenv("AFESP_SMALL_MAX", "0")
    monkeypatch.setenv("AFESP_RING_TG_MIN", "1")
    o, v = 4, 9
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    cc = orc.OracleCC(o, v, eri, e, 8)
    eng.ccsd_init(o, v, e, eri, 8)
    rng = np.random.default_rng(9)
    t1 = 0.05 * rng.standard_normal((o, v))
    t2 = 0.05 * rng.standard_normal((o, o, v, v))
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    cc.t1[...] = t1; cc.t2[...] = t2
    eng.set_amplitudes(t1, t2)
    cc.L.orc_cc_intermediates(cc.h)
    eng.update_intermediates()
    eng.set_amplitudes(t1, t2)
    for name in ("I_ovov", "I_voov"):
        assert np.max(np.abs(eng.tensor(name) - cc.field(name))) < 1e-12, name
    cc.L.orc_cc_amplitudes(cc.h)
    eng.update_amplitudes()
    assert np.max(np.abs(eng.tensor("r1") - cc.field("r1"))) < 1e-12
    assert np.max(np.abs(_p(eng.tensor("r2")) - _p(cc.field("r2")))) < 1e-12


def test_solve_from_amplitudes_handed_in_on_the_large_system_path(eng, monkeypatch):
    """afesp_ccsd_set_amplitudes then afesp_ccsd_solve down the large-system path: the error vector of the iteration that starts from the
    handed-in set stays in the DIIS history for nerr iterations, and while it does the two-kernel tail sums the DIIS overlaps over every
    element (CCState::hist_plain) instead of a <= b; afterwards it halves them again.  Every iteration energy against the oracle, which
    sums everything, through more iterations than the history has slots."""
    monkeypatch.setenv("AFESP_SMALL_MAX", "0")
    monkeypatch.setenv("AFESP_RING_TG_MIN", "1")
    o, v, nerr = 4, 9, 3
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    cc = orc.OracleCC(o, v, eri, e, nerr)
    eng.ccsd_init(o, v, e, eri, nerr)
    rng = np.random.default_rng(11)
    t1 = 0.02 * rng.standard_normal((o, v))
    t2 = 0.02 * rng.standard_normal((o, o, v, v))
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    cc.t1[...] = t1; cc.t2[...] = t2
    eng.set_amplitudes(t1, t2)
    nit, en, rm = eng.do_ccsd_spatial(30, 1e-10, 1e-10)
    onit, oen, orm = cc.solve(30, 1e-10, 1e-10)
    assert nit == onit and nit > nerr + 2, (nit, onit)
    assert np.max(np.abs(en[:nit + 1] - oen[:nit + 1])) < 1e-10 and np.max(np.abs(rm[:nit + 1] - orm[:nit + 1])) < 1e-10


@pytest.mark.parametrize("pp_sym", ["0", "1"])
def test_reinitialised_state_follows_new_integrals(pp_sym, monkeypatch):
    """Initialised again where it lies with OTHER integrals of the same extents (the next geometry of a scan), on both forms of the
    pp-ladder and down the large-system path too: iteration energies and (T) against the oracle on the second system, after the first
    one has left its amplitudes, DIIS history, on-request tensors and integral copies behind."""
    from afesp_amd.capi import Engine
    o, v = 5, 13
    for large in ("0", "1"):
        if large == "1":
            monkeypatch.setenv("AFESP_SMALL_MAX", "0")
            monkeypatch.setenv("AFESP_RING_TG_MIN", "1")
        monkeypatch.setenv("AFESP_PP_SYM", pp_sym)
        with Engine(0) as e:
            for seed in (3, 4, 5):
                n, lev, eri = molecules.synthetic_system(o, v, scale=0.05, seed=seed)
                cc = orc.OracleCC(o, v, eri, lev, 4)
                e.ccsd_init(o, v, lev, eri, 4)
                nit, en, rm = e.do_ccsd_spatial(40, 1e-9, 1e-9)
                onit, oen, _ = cc.solve(40, 1e-9, 1e-9)
                assert nit == onit and np.max(np.abs(en[:nit + 1] - oen[:onit + 1])) < 1e-10, (seed, large)
                assert np.max(np.abs(e.do_ccsd_t_spatial() - cc.triples(lev))) < 1e-10
                e.tensor("v_vvvv")
                e.tensor("I_vovv_p")


@pytest.mark.parametrize("name", ["n2-cc-pvdz", "f2-cc-pvdz"])
def test_completely_renormalised_triples_match_bundled_outputs(eng, name):
    """CR-CCSD[T]/(T) (src/ccsd.f90:2338-2551): the numbers of the reference's bundled CRCCSD(T)_spatial runs."""
    si, ints, res, gold = molecules.load(name)
    n, o = ints.nbasis, ints.nel // 2
    v = n - o
    e_mp2, eri_mo = eng.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, ints.eri)
    eng.ccsd_init(o, v, res.canon_levels, None, si.ccsd_diis_n_errmat)
    nit, en, rm = eng.do_ccsd_spatial(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    eng.build_cr_intermediates()
    out = eng.do_ccsd_t_spatial_cr()
    cc = orc.OracleCC(o, v, eri_mo, res.canon_levels, si.ccsd_diis_n_errmat)
    cc.solve(si.ccsd_maxiter, si.ccsd_e_tol, si.ccsd_t_tol)
    cc.cr_intermediates()
    ref = cc.triples_cr(res.canon_levels)
    assert np.max(np.abs(out - ref)) < 1e-9
    ec = en[nit]
    assert abs(ec + out[4] / out[2] - gold["cr_ccsd_bt_corr"]) < 1e-8
    assert abs(ec + out[5] / out[3] - gold["cr_ccsd_pt_corr"]) < 1e-8
    assert abs(ec + out[1] / out[3] - gold["r_ccsd_pt_corr"]) < 1e-8
    nt = eng.ntriples()
    parts = sum(eng.do_ccsd_t_spatial_cr(b, min(b + 17, nt)) for b in range(0, nt, 17))
    assert np.max(np.abs(parts - out)) < 1e-12


@pytest.mark.parametrize("fused", [1, 0, "large", "ring"])
@pytest.mark.parametrize("o,v", [(4, 9), (12, 72), (5, 13), (18, 21)])
def test_one_iteration_term_by_term(eng, o, v, fused, monkeypatch):
    """Every intermediate and both residuals after one update from non-trivial amplitudes (t1 != 0).  The second size is
    the largest the oracle does in seconds and is past the thresholds where the launcher switches to the kernels config 5
    runs on: 256x128 / 128x128 tiles with 16-byte staging, K slicing by the wave-quantisation score, re-laid-out operands,
    the pp-ladder in its symmetric/antisymmetric pair form with M = v(v+1)/2 = 2628 rows.  "large": the single stream of
    whole-tensor products that o^2 v^2 > 2^20 takes (AFESP_SMALL_MAX=0 sends these sizes down it): there the t1 term of I_vv comes
    from the m = i diagonals of the two <eb|ia> products instead of a pass over 2<eb|ma> - <be|ma>."""
    if fused in ("large", "ring"):
        monkeypatch.setenv("AFESP_SMALL_MAX", "0")
        # "ring": the large-system path as config 5 runs it -- the six o^3 v^3 ring products as two launches of the LDS-DMA GEMM
        # (csrc/ring.hip; from o v = 2048 by itself); "large": the same path with those products on the gather kernel
        monkeypatch.setenv("AFESP_RING_TG_MIN", "1" if fused == "ring" else "1000000")
        # the streamed tall x skinny kernel from 64 rows on (default 2^17): at (12, 72) and (18, 21) the products of t1 with <eb|ia> then run
        # as config 5 runs them -- y and x_voov in ONE launch of tall_dual_kernel (12 / 18 columns: one / two fragments, ragged rows)
        monkeypatch.setenv("AFESP_TALL_MIN", "64")
        if fused == "ring":
            # ... and the pair forms at every size: there the T1 equation's asym(m,i,e,f) <ef|ma> is a trace of the t2 <ef|ia> product
            monkeypatch.setenv("AFESP_PP_SYM", "1")
        fused = 0
    if o * v > 100:
        from afesp_amd import inputs
        n = o + v
        e = np.concatenate([-2.0 + np.arange(o) / (o - 1), 1.0 + 2.0 * np.arange(v) / (v - 1)])
        eri = 0.02 * (2.0 * np.random.default_rng(11).random(inputs.neri(n)) - 1.0)
    else:
        n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    cc = orc.OracleCC(o, v, eri, e, 8)
    eng.ccsd_set_fused(fused)       # both evaluation orders: one grouped launch per dependency level (csrc/fused.hip) / call by call
    eng.ccsd_init(o, v, e, eri, 8)
    rng = np.random.default_rng(5)
    t1 = 0.05 * rng.standard_normal((o, v))
    t2 = 0.05 * rng.standard_normal((o, o, v, v))
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    cc.t1[...] = t1
    cc.t2[...] = t2
    eng.set_amplitudes(t1, t2)
    cc.L.orc_cc_intermediates(cc.h)
    eng.update_intermediates()
    tol = 1e-12 if o * v <= 100 else 2e-11      # sums of up to o v^2 = 62 208 terms at the second size
    for name in INTERMEDIATES:
        assert np.max(np.abs(eng.tensor(name) - cc.field(name))) < tol, name
    cc.L.orc_cc_amplitudes(cc.h)
    eng.update_amplitudes()
    assert np.max(np.abs(eng.tensor("r1") - cc.field("r1"))) < tol
    assert np.max(np.abs(_p(eng.tensor("r2")) - _p(cc.field("r2")))) < tol
    g1, g2 = eng.amplitudes()
    eng.ccsd_set_fused(-1)
    assert np.max(np.abs(g1 - cc.t1)) < tol and np.max(np.abs(g2 - cc.t2)) < tol


@pytest.mark.parametrize("pp_sym", ["0", "1"])
@pytest.mark.parametrize("o,v", [(5, 53), (7, 21), (9, 19), (4, 10), (1, 3), (2, 2), (9, 1), (1, 9), (6, 4)])
def test_launch_fused_iteration_equals_the_call_by_call_iteration(o, v, pp_sym, monkeypatch):
    """Small systems run their iteration as a compiled sequence of grouped launches (csrc/fused.hip: the recorded calls of
    update_restricted_intermediates / update_amplitudes_restricted levelled by data dependence, the update / energy / DIIS tail in
    two kernels, the DIIS system solved on the host).  Same DIIS path as the call-by-call evaluation, iteration by iteration; at
    most 25 launches per iteration; and the two may alternate inside one solve (the error overlap matrix lives on the device)."""
    from afesp_amd.capi import Engine
    monkeypatch.setenv("AFESP_PP_SYM", pp_sym)
    runs = {}
    with Engine(0) as e2:
        for mode in ("fused", "plain", "mixed"):
            e2.ccsd_set_fused(0 if mode == "plain" else 1)
            e2.synthetic_init(o, v, 0.03, 4711, 5)
            e2.ccsd_energy()
            en = []
            for it in range(9):
                if mode == "mixed":
                    e2.ccsd_set_fused(1 if (it // 2) % 2 == 0 else 0)
                en.append(e2.ccsd_iterate(1e-14, 1e-14)[:2])
                e2.ccsd_diis()
            if mode == "fused":
                nl = e2.ccsd_iteration_launches()
                assert 0 < nl <= 25, nl
            runs[mode] = (np.array(en), e2.amplitudes())
        e2.ccsd_set_fused(-1)
    for mode in ("plain", "mixed"):
        assert np.max(np.abs(runs["fused"][0] - runs[mode][0])) < 1e-12, mode
        assert np.max(np.abs(runs["fused"][1][0] - runs[mode][1][0])) < 1e-12 and np.max(np.abs(runs["fused"][1][1] - runs[mode][1][1])) < 1e-12


@pytest.mark.parametrize("o,v", [(1, 3), (2, 2), (3, 8), (6, 4), (5, 9), (2, 17), (7, 21), (9, 2)])
def test_pp_ladder_split_form_on_ragged_extents(eng, o, v, monkeypatch):
    """The symmetric/antisymmetric pair form of the particle-particle ladder (chosen by itself from about o = 10, v = 70 on;
    forced here) on shapes with a single occupied orbital (no antisymmetric part), o > v, odd pair counts."""
    monkeypatch.setenv("AFESP_PP_SYM", "1")
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05, seed=o * 100 + v)
    cc = orc.OracleCC(o, v, eri, e, 3)
    eng.ccsd_init(o, v, e, eri, 3)
    rng = np.random.default_rng(o + 31 * v)
    t1 = 0.05 * rng.standard_normal((o, v))
    t2 = 0.05 * rng.standard_normal((o, o, v, v))
    t2 = 0.5 * (t2 + t2.transpose(1, 0, 3, 2))
    cc.t1[...] = t1
    cc.t2[...] = t2
    eng.set_amplitudes(t1, t2)
    cc.L.orc_cc_intermediates(cc.h); cc.L.orc_cc_amplitudes(cc.h)
    eng.update_intermediates(); eng.update_amplitudes()
    assert np.max(np.abs(_p(eng.tensor("r2")) - _p(cc.field("r2")))) < 1e-12
    g1, g2 = eng.amplitudes()
    assert np.max(np.abs(g1 - cc.t1)) < 1e-12 and np.max(np.abs(g2 - cc.t2)) < 1e-12
    monkeypatch.setenv("AFESP_PP_SYM", "0")
    eng.ccsd_init(o, v, e, eri, 3)
    eng.set_amplitudes(t1, t2)
    eng.update_intermediates(); eng.update_amplitudes()
    h1, h2 = eng.amplitudes()
    assert np.max(np.abs(h2 - g2)) < 1e-13


@pytest.mark.parametrize("blocked", ["0", "1", "gather"])
@pytest.mark.parametrize("n,o", [(2, 1), (3, 2), (7, 3), (13, 4), (24, 5), (33, 16), (58, 5), (64, 9), (66, 9)])
def test_ao2mo_pair_symmetric_transform(eng, n, o, blocked, monkeypatch):
    """AO->MO over the unique pairs (kl), then (pq): every packed MO integral and E(MP2) against the restatement of the four
    quarter transforms (src/mp2.f90:321-410), with a general (non-orthogonal) coefficient matrix and odd extents -- the whole
    tensor at once (up to 64 functions: the LDS-resident pair transform, three launches; "gather": the gather-GEMM form it
    replaced, which bases of 65...95 functions still run) and slab by slab through the pair-packed half-transformed array (what
    large bases run; forced here), the latter also starting from the (ij|KL) copy a Fock build leaves on the device."""
    from afesp_amd import inputs
    monkeypatch.setenv("AFESP_AO2MO_BLOCKED", "0" if blocked == "gather" else blocked)
    if blocked == "gather":
        monkeypatch.setenv("AFESP_AO2MO_PAIR", "0")
    rng = np.random.default_rng(100 * n + o)
    eri = rng.standard_normal(inputs.neri(n))
    c = rng.standard_normal((n, n))
    e = np.concatenate([-2.0 - rng.random(o), 1.0 + rng.random(n - o)])
    e_mp2, eri_mo = eng.do_mp2_spatial(n, o, c, e, eri)
    ref = orc.ao2mo(n, c, eri)
    assert np.max(np.abs(eri_mo - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))
    ref_e = orc.mp2_energy(n, o, ref, e)
    assert abs(e_mp2 - ref_e) < 1e-10 * max(1.0, abs(ref_e))
    eng.set_eri(n, eri)
    e2, again = eng.do_mp2_spatial(n, o, c, e, None)       # AO integrals already resident on the device
    assert np.array_equal(again, eri_mo) and e2 == e_mp2
    eng.set_eri(n, eri)
    eng.build_fock(n, np.eye(n), np.zeros((n, n)))          # leaves (ij|KL), ij squared up, as the transform's starting point
    e3, third = eng.do_mp2_spatial(n, o, c, e, None)
    assert np.array_equal(third, eri_mo) and e3 == e_mp2


@pytest.mark.parametrize("n,o", [(16, 3), (24, 5), (58, 5), (130, 9), (100, 7)])
def test_ao2mo_on_the_lds_dma_gemm(n, o, monkeypatch):
    """The whole-tensor AO->MO with its four quarter transforms on tgemm_kernel (what even bases from n = 96 on run; forced here
    for the small ones): K tails of 0, 8, 10 and 2 elements, one and two row tiles, the triangular column list of the last
    transform, the first-128-rows shortcut for the pairs with p < 128 (n = 130) -- every packed MO integral against the
    restatement (n <= 58) and against the gather-GEMM path (all n), also from the (ij|KL) copy of a Fock build.  Rows that end at most
    96 past a multiple of 128 run their last tile 96 rows high (tgemm_mixed_kernel: every n here but 100, whose rows keep 128-row tiles)."""
    from afesp_amd import inputs
    from afesp_amd.capi import Engine
    rng = np.random.default_rng(7 * n + o)
    eri = rng.standard_normal(inputs.neri(n))
    c = rng.standard_normal((n, n))
    e = np.concatenate([-2.0 - rng.random(o), 1.0 + rng.random(n - o)])
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("AFESP_AO2MO_TG", mode)
        monkeypatch.setenv("AFESP_AO2MO_BLOCKED", "0")
        with Engine(0) as eng:
            got[mode] = eng.do_mp2_spatial(n, o, c, e, eri)
            if mode == "1":
                lc = eng.launch_counts()
                assert (lc["tgemm_mixed"] > 0) if n != 100 else (lc["tgemm"] > 0 and lc["tgemm_mixed"] == 0), lc
                eng.set_eri(n, eri)
                eng.build_fock(n, np.eye(n), np.zeros((n, n)))
                e3, third = eng.do_mp2_spatial(n, o, c, e, None)
                assert np.array_equal(third, got[mode][1]) and e3 == got[mode][0]
                # (round 6) the temporaries' columns are 16 ceil(n / 16) doubles long -- the same sums in the same order as with columns of n
                monkeypatch.setenv("AFESP_AO2MO_PAD", "0")
                e4, fourth = eng.do_mp2_spatial(n, o, c, e, eri)
                monkeypatch.delenv("AFESP_AO2MO_PAD")
                assert np.array_equal(fourth, got[mode][1]) and e4 == got[mode][0]
                e5, fifth = eng.do_mp2_spatial(n, o, c, e, eri)      # ... and back: the padding rows are zeroed again
                assert np.array_equal(fifth, got[mode][1]) and e5 == got[mode][0]
    scale = max(1.0, np.max(np.abs(got["0"][1])))
    assert np.max(np.abs(got["1"][1] - got["0"][1])) < 1e-11 * scale
    assert abs(got["1"][0] - got["0"][0]) < 1e-10 * max(1.0, abs(got["0"][0]))
    if n <= 58:
        ref = orc.ao2mo(n, c, eri)
        assert np.max(np.abs(got["1"][1] - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("device_from", ["1", "1000000000"])
def test_offset_tables_built_on_the_device_equal_the_host_enumeration(device_from, monkeypatch):
    """The planner writes the big offset tables of a contraction with a kernel and derives the 16-byte-staging flags from the
    strides; AFESP_PLAN_VERIFY makes every new plan compare both with the host enumeration and the scanned table (an error
    status if they differ).  A fresh context (no cached plans), odd and even extents, AO->MO and a whole iteration + (T)."""
    from afesp_amd import inputs
    from afesp_amd.capi import Engine
    monkeypatch.setenv("AFESP_PLAN_VERIFY", "1")
    monkeypatch.setenv("AFESP_PLAN_DEVICE_FROM", device_from)
    for n, o in ((13, 4), (24, 5)):
        rng = np.random.default_rng(7 * n + o)
        eri = rng.standard_normal(inputs.neri(n)) * 0.05
        c = rng.standard_normal((n, n))
        e = np.concatenate([-2.0 - rng.random(o), 1.0 + rng.random(n - o)])
        with Engine(0) as fresh:
            e_mp2, eri_mo = fresh.do_mp2_spatial(n, o, c, e, eri)
            ref = orc.ao2mo(n, c, eri)
            assert np.max(np.abs(eri_mo - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))
            fresh.ccsd_init(o, n - o, e, eri_mo, 4)
            cc = orc.OracleCC(o, n - o, eri_mo, e, 4)
            for _ in range(2):
                fresh.update_intermediates(); fresh.update_amplitudes()
                cc.L.orc_cc_intermediates(cc.h); cc.L.orc_cc_amplitudes(cc.h)
            t1, t2 = fresh.amplitudes()
            assert np.max(np.abs(t2 - cc.t2)) < 1e-9 * max(1.0, np.max(np.abs(cc.t2)))
            ref_t = cc.triples(e)
            assert np.allclose(fresh.do_ccsd_t_spatial(), ref_t, atol=1e-9 * max(1.0, np.max(np.abs(ref_t))), rtol=0)


def test_failed_graph_capture_leaves_a_working_context():
    """The iteration of a small system is captured into a hipGraph on its second call.  A failure in the middle of the captured
    body -- thrown while a lane other than the main one is selected (afesp_test_inject) -- must end the capture on the origin
    stream and fall back to plain launches: that iteration and every later one still follow the oracle."""
    from afesp_amd.capi import Engine
    o, v = 4, 9
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    cc = orc.OracleCC(o, v, eri, e, 8)
    onit, oen, _ = cc.solve(40, 1e-8, 1e-9)
    with Engine(0) as eng2:
        eng2.ccsd_init(o, v, e, eri, 8)
        en = [eng2.ccsd_energy(1e-8, 1e-9)[0]]
        for it in range(1, onit + 1):
            if it == 2:
                eng2.test_inject(1)                  # the capturing call
            e_it, _, conv = eng2.ccsd_iterate(1e-8, 1e-9)
            en.append(e_it)
            if not conv:
                eng2.ccsd_diis()
        assert conv
        assert np.max(np.abs(np.array(en) - oen[:onit + 1])) < 1e-10
        out = eng2.do_ccsd_t_spatial()
        assert np.max(np.abs(out - cc.triples(e))) < 1e-10


def test_default_graph_policy_short_solve_runs_unreplayed(monkeypatch):
    """Default policy (AFESP_GRAPH_AFTER unset = 40 calls): a solve of ordinary length never captures a graph -- the laned
    launches alone must walk the oracle's iteration path too."""
    from afesp_amd.capi import Engine
    monkeypatch.delenv("AFESP_GRAPH_AFTER", raising=False)
    o, v = 4, 9
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    cc = orc.OracleCC(o, v, eri, e, 8)
    onit, oen, _ = cc.solve(40, 1e-8, 1e-9)
    with Engine(0) as eng2:
        eng2.ccsd_init(o, v, e, eri, 8)
        nit, en, _ = eng2.do_ccsd_spatial(40, 1e-8, 1e-9)
    assert nit == onit and np.max(np.abs(en[:nit + 1] - oen[:onit + 1])) < 1e-10


def test_pp_ladder_split_form_through_the_replayed_iteration(eng, monkeypatch):
    """The pair form inside the laned, graph-replayed iteration of a small system: a whole solve against the oracle."""
    monkeypatch.setenv("AFESP_PP_SYM", "1")
    o, v = 5, 19
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04)
    cc = orc.OracleCC(o, v, eri, e, 8)
    eng.ccsd_init(o, v, e, eri, 8)
    nit, en, rm = eng.do_ccsd_spatial(60, 1e-8, 1e-9)
    onit, oen, orm = cc.solve(60, 1e-8, 1e-9)
    assert nit == onit and nit > 3
    assert np.max(np.abs(en[:nit + 1] - oen[:nit + 1])) < 1e-10


def test_h2o_tz_shape_synthetic(eng):
    """BASELINE config 2 shape (o=5, v=53) on the SURVEY 8(d) synthetic integrals: CCSD path + (T) vs oracle."""
    o, v = 5, 53
    n, e, eri = molecules.synthetic_system(o, v, scale=0.02)
    cc = orc.OracleCC(o, v, eri, e, 8)
    eng.ccsd_init(o, v, e, eri, 8)
    nit, en, rm = eng.do_ccsd_spatial(60, 1e-6, 1e-7)
    onit, oen, orm = cc.solve(60, 1e-6, 1e-7)
    assert nit == onit and nit > 0
    assert np.max(np.abs(en[:nit + 1] - oen[:nit + 1])) < 1e-10
    out = eng.do_ccsd_t_spatial()
    ref = cc.triples(e)
    assert np.max(np.abs(out - ref)) < 1e-9


@pytest.mark.parametrize("o,v", [(1, 3), (2, 2), (3, 8), (6, 4), (4, 16), (8, 8), (2, 17), (5, 9)])
def test_ragged_and_degenerate_extents(eng, o, v):
    """Edge shapes: o > v, single occupied, v below / not a multiple of the 8-wide cube and 16-deep K tiles, even and odd
    extents (16-byte and 8-byte staging paths).  Three iterations with DIIS, then (T) and CR-(T), against the oracle."""
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05, seed=o * 100 + v)
    cc = orc.OracleCC(o, v, eri, e, 3)
    eng.ccsd_init(o, v, e, eri, 3)
    L = cc.L
    eng.ccsd_energy(1e-12, 1e-12)
    L.orc_cc_energy(cc.h, 1e-12, 1e-12)
    for it in range(3):
        ge, gr, _ = eng.ccsd_iterate(1e-12, 1e-12)
        eng.ccsd_diis()
        L.orc_cc_diis_save(cc.h); L.orc_cc_intermediates(cc.h); L.orc_cc_amplitudes(cc.h); L.orc_cc_energy(cc.h, 1e-12, 1e-12)
        assert abs(ge - cc.energy) < 1e-11 and abs(gr - L.orc_cc_get_rms(cc.h)) < 1e-11, (it, ge, cc.energy)
        L.orc_cc_diis_update(cc.h)
    t1, t2 = eng.amplitudes()
    assert np.max(np.abs(t1 - cc.t1)) < 1e-11 and np.max(np.abs(t2 - cc.t2)) < 1e-11
    out = eng.do_ccsd_t_spatial()
    ref = cc.triples(e)
    assert np.max(np.abs(out - ref)) < 1e-10, (out, ref)
    eng.build_cr_intermediates()
    ipp, ioo = cc.cr_intermediates()
    outc = eng.do_ccsd_t_spatial_cr()
    refc = cc.triples_cr(e)
    assert np.max(np.abs(outc - refc)) < 1e-10, (outc, refc)


@pytest.mark.parametrize("o,v", [(1, 1), (2, 1), (9, 1), (1, 9)])
@pytest.mark.parametrize("pp_sym", ["0", "1"])
def test_single_orbital_extents(eng, o, v, pp_sym, monkeypatch):
    """One occupied or one virtual orbital (no antisymmetric pairs, one-element cubes), both forms of the pp-ladder: three
    iterations with DIIS and (T) against the oracle."""
    monkeypatch.setenv("AFESP_PP_SYM", pp_sym)
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05, seed=7 * o + v)
    cc = orc.OracleCC(o, v, eri, e, 3)
    eng.ccsd_init(o, v, e, eri, 3)
    L = cc.L
    eng.ccsd_energy(1e-12, 1e-12)
    L.orc_cc_energy(cc.h, 1e-12, 1e-12)
    for it in range(3):
        ge, gr, _ = eng.ccsd_iterate(1e-12, 1e-12)
        eng.ccsd_diis()
        L.orc_cc_diis_save(cc.h); L.orc_cc_intermediates(cc.h); L.orc_cc_amplitudes(cc.h); L.orc_cc_energy(cc.h, 1e-12, 1e-12)
        assert abs(ge - cc.energy) < 1e-12, (it, ge, cc.energy)
        L.orc_cc_diis_update(cc.h)
    assert np.max(np.abs(eng.do_ccsd_t_spatial() - cc.triples(e))) < 1e-12


def test_ccsd_without_diis_and_nonconvergence_is_silent(eng):
    o, v = 3, 6
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    eng.ccsd_init(o, v, e, eri, 1)        # n_errmat < 2 switches DIIS off (ccsd.f90:593-595)
    cc = orc.OracleCC(o, v, eri, e, 1)
    nit, en, rm = eng.do_ccsd_spatial(3, 1e-14, 1e-14)
    onit, oen, _ = cc.solve(3, 1e-14, 1e-14)
    assert nit == -1 and onit == -1       # falls through the loop like ccsd.f90:396
    assert np.max(np.abs(en - oen)) < 1e-12


def test_errors_are_reported_not_swallowed(eng):
    from afesp_amd.capi import AfespError
    with pytest.raises(AfespError):
        eng.ccsd_init(0, 5, np.zeros(5), np.zeros(int(orc.lib().orc_neri(5))), 8)


def test_eri_text_reader_and_fcidump_writer(eng, tmp_path):
    """Input/output side of the path (integrals.f90:146-161, mp2.f90:451-487): eri.dat -> device -> AO->MO -> FCIDUMP."""
    import os
    from afesp_amd import inputs
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    n, o = ints.nbasis, ints.nel // 2
    path = os.path.join(molecules.GOLDEN, "h2o-cc-pvdz", "eri.dat")
    packed, nlines = eng.read_eri_text(path, n)
    assert nlines == sum(1 for _ in open(path))
    assert np.array_equal(packed, ints.eri)                      # bit-identical to the host-side reader
    e_mp2, eri_mo = eng.do_mp2_spatial(n, o, res.canon_coeff, res.canon_levels, None)   # integrals already resident
    assert abs(e_mp2 - molecules.SURVEY_GOLD["h2o-cc-pvdz"]["mp2_corr"]) < 1e-9
    out = tmp_path / "FCIDUMP"
    nw = eng.write_fcidump(out, n)
    # restate write_fcidump: canonical packed order, |x| > 1e-7, (I3,I3,I3,I3,ES17.9)
    expect = []
    k = 0
    for p in range(1, n + 1):
        for q in range(1, p + 1):
            for r in range(1, p + 1):
                for s in range(1, (q if p == r else r) + 1):
                    x = eri_mo[k]
                    k += 1
                    if abs(x) > 1e-7:
                        expect.append((p, q, r, s, x))
    assert k == inputs.neri(n) and nw == len(expect)
    lines = open(out).read().splitlines()
    assert len(lines) == len(expect)
    for line, (p, q, r, s, x) in zip(lines[::97] + lines[-3:], expect[::97] + expect[-3:]):
        assert len(line) == 29
        assert (int(line[0:3]), int(line[3:6]), int(line[6:9]), int(line[9:12])) == (p, q, r, s)
        assert line[12:] == "%17.9E" % x and abs(float(line[12:]) - x) <= 5e-10 * abs(x)
    # list-directed input (src/integrals.f90:150): Fortran D exponents, comma separators and trailing items are what the
    # reference's `read (ir, *)` accepts -- the same lines through the engine's reader
    lines_in = open(path).read().splitlines()
    alt = tmp_path / "alt.dat"
    rewritten = []
    for li, ln in enumerate(lines_in):
        i, j, a, b, x = ln.split()
        if li % 3 == 0:
            mant, ex = ("%.16E" % float(x)).split("E")      # 17 significant digits: the value survives
            rewritten.append(f"{i} {j} {a} {b} {mant}D{ex}")
        elif li % 3 == 1:
            rewritten.append(f"{i}, {j},{a} ,{b},   {x}  ignored tail")
        else:
            rewritten.append(f"\t{i}\t{j}\t{a}\t{b}\t{x}\r")
    alt.write_text("\n".join(rewritten) + "\n")
    packed_alt, nalt = eng.read_eri_text(alt, n)
    assert nalt == nlines
    assert np.array_equal(packed_alt, ints.eri)
    glued = tmp_path / "glued.dat"
    glued.write_text("1 1 1 1 0.5D-01x\n")
    from afesp_amd.capi import AfespError
    with pytest.raises(AfespError):
        eng.read_eri_text(glued, n)
    # malformed input is an error, not a silent zero
    bad = tmp_path / "bad.dat"
    bad.write_text("1 1 1 1 0.5\n1 1 99 1 0.25\n")
    from afesp_amd.capi import AfespError
    with pytest.raises(AfespError):
        eng.read_eri_text(bad, n)
    with pytest.raises(AfespError):
        eng.read_eri_text(tmp_path / "missing.dat", n)


def test_build_fock_on_device_matches_restatement(eng):
    """hf.f90:349-385 on the resident packed AO integrals, against the CPU restatement, for a converged and a random density."""
    si, ints, res, _ = molecules.load("h2o-cc-pvdz")
    n, o = ints.nbasis, ints.nel // 2
    eng.set_eri(n, ints.eri)
    hcore = ints.core_hamil
    rng = np.random.default_rng(5)
    dens_rand = rng.standard_normal((n, n))
    dens_conv = res.canon_coeff[:o, :].T @ res.canon_coeff[:o, :]
    for dens in (dens_conv, dens_rand + dens_rand.T, dens_rand):
        f_dev = eng.build_fock(n, dens, hcore)
        f_ref = orc.build_fock(n, ints.eri, dens, hcore)
        assert np.max(np.abs(f_dev - f_ref)) < 1e-12 * max(1.0, np.max(np.abs(f_ref)))
    # the converged density reproduces the SCF energy of the reference (E = sum D (H + F), hf.f90:341)
    e = float(np.sum(dens_conv * (hcore + eng.build_fock(n, dens_conv, hcore))))
    assert abs(e + ints.e_nuc - molecules.SURVEY_GOLD["h2o-cc-pvdz"]["rhf_total"]) < 1e-6


@pytest.mark.parametrize("t_gemm", [None, "gett"])
@pytest.mark.parametrize("o,v", [(3, 8), (5, 19), (6, 24)])
def test_plain_triples_equal_the_full_evaluation(eng, o, v, t_gemm, monkeypatch):
    """afesp_ccsd_t_plain (one Z evaluation per element, the symmetriser moved onto W) against afesp_ccsd_t, whole range and
    shards, and against the oracle -- with the products on the LDS-DMA kernel (default) and on the grouped gather kernel
    (AFESP_T_GEMM=gett: what the planner falls back to when 32-bit byte offsets do not reach every operand row)."""
    if t_gemm:
        monkeypatch.setenv("AFESP_T_GEMM", t_gemm)
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04, seed=3 + v)
    eng.ccsd_init(o, v, e, eri, 6)
    cc = orc.OracleCC(o, v, eri, e, 6)
    eng.do_ccsd_spatial(40, 1e-9, 1e-9)
    cc.solve(40, 1e-9, 1e-9)
    full = eng.do_ccsd_t_spatial()
    plain = eng.do_ccsd_t_spatial_plain()
    ref = cc.triples(e)
    assert np.max(np.abs(plain - full[:2])) < 1e-13 * max(1.0, np.max(np.abs(full)))
    assert np.max(np.abs(plain - ref[:2])) < 1e-11
    nt = eng.ntriples()
    parts = eng.do_ccsd_t_spatial_plain(0, nt // 2) + eng.do_ccsd_t_spatial_plain(nt // 2, nt)
    assert np.max(np.abs(parts - plain)) < 1e-13 * max(1.0, np.max(np.abs(plain)))


def test_block_pool_in_pieces_of_idle_memory_equals_one_allocation(monkeypatch):
    """The (T) block pool is assembled from blocks the context holds idle (here: the AO->MO temporaries, which at these extents are
    larger than a few blocks but not more than the pool needs) plus a fresh remainder; the GEMM column tables and the orbit
    kernel address blocks across the pieces.  Same sums as with the pool in one allocation (AFESP_T_ONE_POOL) and as the oracle;
    a second system in the same context finds the pieces of the first."""
    from afesp_amd.capi import Engine
    o, v = 6, 24
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04, seed=29)
    cc = orc.OracleCC(o, v, eri, e, 6)
    cc.solve(40, 1e-9, 1e-9)
    ref = cc.triples(e)
    out = {}
    for mode in ("pieces", "one"):
        if mode == "one":
            monkeypatch.setenv("AFESP_T_ONE_POOL", "1")
        with Engine(0) as eng:
            eng.do_mp2_spatial(n, o, np.eye(n), e, eri, want_eri_mo=False)    # leaves its two temporaries idle at ccsd_init
            eng.ccsd_init(o, v, e, None, 6)
            eng.do_ccsd_spatial(40, 1e-9, 1e-9)
            idle_before = eng.arena_stats()["idle_gb"]
            out[mode] = eng.do_ccsd_t_spatial()
            if mode == "pieces":
                assert eng.arena_stats()["idle_gb"] < idle_before        # the pool took idle blocks
                nt = eng.ntriples()
                parts = eng.do_ccsd_t_spatial(0, nt // 3) + eng.do_ccsd_t_spatial(nt // 3, nt)
                assert np.max(np.abs(parts - out[mode])) < 1e-12 * max(1.0, np.max(np.abs(out[mode])))
                eng.do_mp2_spatial(n, o, np.eye(n), e, eri, want_eri_mo=False)   # next system: gives the pool back first
                eng.ccsd_init(o, v, e, None, 6)
                eng.do_ccsd_spatial(40, 1e-9, 1e-9)
                assert np.array_equal(eng.do_ccsd_t_spatial(), out[mode])
    assert np.array_equal(out["pieces"], out["one"])
    assert np.max(np.abs(out["pieces"] - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))


def test_plain_and_renormalised_triples_do_not_keep_both_block_pools():
    """A plain (T) followed by a completely renormalised one on the same state (what bench.py and the rank tests do) must not keep
    the plain pool's pieces beside the two CR pools, and going back gives the CR pools back: the memory the context holds
    handed out stays that of the variant in use."""
    from afesp_amd.capi import Engine
    o, v = 8, 64
    with Engine(0) as eng:
        eng.synthetic_init(o, v, 0.02, 4711, 6)
        eng.ccsd_energy()
        for _ in range(2):
            eng.ccsd_iterate(); eng.ccsd_diis()
        plain = eng.do_ccsd_t_spatial()
        live_plain = eng.arena_stats()["live_gb"]
        eng.build_cr_intermediates()
        live_cr_state = eng.arena_stats()["live_gb"]          # (the CR intermediates themselves)
        cr = eng.do_ccsd_t_spatial_cr()
        live_cr = eng.arena_stats()["live_gb"]
        pool_gb = 8 * 8 * 9 // 2 * (8 ** 3 * 512) * 8 / 1e9   # o x o(o+1)/2 blocks of 8^3 cubes: one pool of this system
        assert live_cr - live_cr_state < 1.25 * pool_gb + 0.05, (live_plain, live_cr_state, live_cr)   # two pools instead of one, not three
        again = eng.do_ccsd_t_spatial()
        assert np.array_equal(again, plain)
        assert eng.arena_stats()["live_gb"] <= live_cr_state + 0.05
        assert np.max(np.abs(cr[:4] - plain[:4])) < 1e-12 * max(1.0, np.max(np.abs(plain)))


def test_iteration_graph_survives_other_work_in_the_same_context(eng):
    """The small-system iteration is replayed as a captured graph; (T) calls, tensor downloads and a spin-orbital solve in
    the same context (which frees cached scratch buffers) must not leave it replaying stale buffers."""
    o, v = 4, 10
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05, seed=21)
    cc = orc.OracleCC(o, v, eri, e, 6)
    eng.ccsd_init(o, v, e, eri, 6)
    eng.ccsd_energy(); cc.L.orc_cc_energy(cc.h, 1e-9, 1e-9)
    for it in range(8):
        eng.ccsd_iterate(); eng.ccsd_diis()
        cc.L.orc_cc_diis_save(cc.h); cc.L.orc_cc_intermediates(cc.h); cc.L.orc_cc_amplitudes(cc.h)
        cc.L.orc_cc_energy(cc.h, 1e-9, 1e-9); cc.L.orc_cc_diis_update(cc.h)
        if it == 2:
            eng.do_ccsd_t_spatial()
            eng.tensor("r2")
        if it == 4:
            eng.init_cc_spinorb(n, 2 * o, e, eri, 4)      # frees the context's scratch cache
            eng.so_iterate()
        t1, t2 = eng.amplitudes()
        assert np.max(np.abs(t2 - cc.t2)) < 1e-11 and np.max(np.abs(t1 - cc.t1)) < 1e-11, it


@pytest.mark.parametrize("o,v", [(4, 10), (3, 16), (6, 7)])
def test_completely_renormalised_triples_on_synthetic_extents(eng, o, v):
    """CR moments through the grouped 16-byte-staged launches (even v) and the plain ones (odd v), whole range and shards,
    against the oracle (the molecules above all have an odd number of virtuals)."""
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04, seed=11 + o)
    eng.ccsd_init(o, v, e, eri, 6)
    cc = orc.OracleCC(o, v, eri, e, 6)
    nit, _, _ = eng.do_ccsd_spatial(60, 1e-9, 1e-9)
    onit, _, _ = cc.solve(60, 1e-9, 1e-9)
    assert nit == onit and nit > 0
    eng.build_cr_intermediates()
    ipp, ioo = cc.cr_intermediates()
    out = eng.do_ccsd_t_spatial_cr()
    ref = cc.triples_cr(e)
    assert np.max(np.abs(out - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))
    nt = eng.ntriples()
    parts = eng.do_ccsd_t_spatial_cr(0, nt // 3) + eng.do_ccsd_t_spatial_cr(nt // 3, nt)
    assert np.max(np.abs(parts - out)) < 1e-12 * max(1.0, np.max(np.abs(out)))


def test_vvvv_slice_is_formed_on_request_when_the_ladder_runs_in_pair_form(monkeypatch):
    """A system whose pp-ladder runs in pair form builds V+- straight from the packed MO integrals and never forms <ef|ab>
    (12.8 GB at v = 200) unless asked: afesp_ccsd_get_tensor("v_vvvv") and the completely renormalised intermediates
    (src/ccsd.f90:2513-2520) build it then -- from the state's own copy when the host handed the integrals in, from the
    context's resident array after afesp_ao2mo_mp2; once that array has been replaced the request is an error, not stale data."""
    from afesp_amd import inputs
    from afesp_amd.capi import Engine, AfespError
    monkeypatch.setenv("AFESP_PP_SYM", "1")
    o, v = 4, 10
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04, seed=15)
    cc = orc.OracleCC(o, v, eri, e, 6)
    onit, _, _ = cc.solve(60, 1e-9, 1e-9)
    cc.cr_intermediates()
    ref = cc.triples_cr(e)
    with Engine(0) as eng:
        eng.ccsd_init(o, v, e, eri, 6)                       # host-supplied integrals: the state keeps its device copy
        live_without = eng.arena_stats()["live_gb"]
        assert eng.do_ccsd_spatial(60, 1e-9, 1e-9)[0] == onit
        assert np.max(np.abs(eng.tensor("v_vvvv") - cc.field("v_vvvv"))) == 0.0
        assert eng.arena_stats()["live_gb"] > live_without   # formed now, not before
        eng.build_cr_intermediates()
        out = eng.do_ccsd_t_spatial_cr()
        assert np.max(np.abs(out - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))
    with Engine(0) as eng:
        c = np.eye(n)
        eng.do_mp2_spatial(n, o, c, e, eri, want_eri_mo=False)   # identity coefficients: the MO integrals are the AO ones
        eng.ccsd_init(o, v, e, None, 6)
        assert eng.do_ccsd_spatial(60, 1e-9, 1e-9)[0] == onit
        eng.build_cr_intermediates()                           # forms <ef|ab> from the resident array
        out = eng.do_ccsd_t_spatial_cr()
        assert np.max(np.abs(out - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))
    with Engine(0) as eng:
        eng.do_mp2_spatial(n, o, np.eye(n), e, eri, want_eri_mo=False)
        eng.ccsd_init(o, v, e, None, 6)
        eng.do_mp2_spatial(n, o, np.eye(n), e, 2.0 * eri, want_eri_mo=False)   # replaces the resident integrals
        with pytest.raises(AfespError):
            eng.tensor("v_vvvv")
        eng.ccsd_init(o, v, e, None, 6)                        # ... a new initialisation sees the new ones
        assert np.max(np.abs(eng.tensor("v_vvvv") - 2.0 * cc.field("v_vvvv"))) < 1e-15


@pytest.mark.parametrize("t_gemm", [None, "gett"])
@pytest.mark.parametrize("o,v,tiny_pool", [(5, 12, False), (7, 10, True), (4, 17, False)])
def test_coinciding_pair_blocks_in_their_own_launch(o, v, tiny_pool, t_gemm, monkeypatch):
    """Blocks Y^{p;qq} are computed as X over half the summation index in a launch of their own and symmetrised by the orbit
    kernel; at small sizes that launch is normally folded into the main one, so force it (AFESP_T_SPLIT_TILES=1) and check
    the plain, the full and the completely renormalised evaluation against the oracle, whole range and shards.  A tiny
    pool gives one chunk per triple (chunks with and without coinciding pairs)."""
    from afesp_amd.capi import Engine
    monkeypatch.setenv("AFESP_T_SPLIT_TILES", "1")
    if t_gemm:
        monkeypatch.setenv("AFESP_T_GEMM", t_gemm)   # (the knob above only moves launches of the gather kernel: both kernels here)
    if tiny_pool:
        monkeypatch.setenv("AFESP_T_POOL_GIB", "0")   # occupied blocks of one index: a chunk per triple
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04, seed=31 + v)
    cc = orc.OracleCC(o, v, eri, e, 6)
    cc.solve(60, 1e-9, 1e-9)
    ref = cc.triples(e)
    ipp, ioo = cc.cr_intermediates()
    ref_cr = cc.triples_cr(e)
    eng = Engine(0)
    try:
        eng.ccsd_init(o, v, e, eri, 6)
        eng.do_ccsd_spatial(60, 1e-9, 1e-9)
        full = eng.do_ccsd_t_spatial()
        plain = eng.do_ccsd_t_spatial_plain()
        assert np.max(np.abs(full[:4] - ref[:4])) < 1e-11 * max(1.0, np.max(np.abs(ref)))
        assert np.max(np.abs(plain - ref[:2])) < 1e-11
        nt = eng.ntriples()
        parts = eng.do_ccsd_t_spatial_plain(0, nt // 3) + eng.do_ccsd_t_spatial_plain(nt // 3, nt)
        assert np.max(np.abs(parts - plain)) < 1e-13 * max(1.0, np.max(np.abs(plain)))
        eng.build_cr_intermediates()
        out = eng.do_ccsd_t_spatial_cr()
        assert np.max(np.abs(out - ref_cr)) < 1e-10 * max(1.0, np.max(np.abs(ref_cr)))
    finally:
        eng.close()


def test_two_live_contexts_launch_the_lds_dma_gemm_side_by_side():
    """The ticket counters and the grid size of tgemm_kernel's launcher belong to the context (Context::tg; they were process-wide
    statics keyed by device until round 4): two contexts on one device run their (T) at the same time from two host threads, tickets
    forced for these small launches (AFESP_TG_DYNAMIC=2, read once per process -> a process of its own), and each reproduces its
    serial result."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, threading
        import numpy as np
        sys.path.insert(0, %r)
        from afesp_amd.capi import Engine
        engs = [Engine(0), Engine(0)]
        refs = []
        for k, e in enumerate(engs):
            e.synthetic_init(6 + k, 24 - 3 * k, 0.03, 99 + k, 6)
            e.do_ccsd_spatial(12, 1e-9, 1e-9)
            refs.append(e.do_ccsd_t_spatial())
        bad = []
        def work(k):
            for _ in range(12):
                out = engs[k].do_ccsd_t_spatial()
                if not np.max(np.abs(out - refs[k])) < 1e-13 * max(1.0, np.max(np.abs(refs[k]))):
                    bad.append((k, out, refs[k]))
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        for e in engs: e.close()
        assert not bad, bad
        print("ok")
    """) % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "a-fortran-electronic-structure-program_amd")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AFESP_TG_DYNAMIC="2"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_two_contexts_first_use_the_large_system_kernels_from_two_threads():
    """Every first use of a kernel function is made under one process-wide lock (csrc/first_use.h), whoever makes it: here two contexts on
    two host threads enter the large-system iteration -- gather kernel, streamed tall x skinny kernel, the ring launches of the LDS-DMA
    GEMM, the two-kernel tail -- and then (T) as their FIRST calls, with the start-up thread's preload running (AFESP_NO_PRELOAD unset),
    i.e. the translation units the preload list does not name (gett, tall, ring) are first-touched by both threads at once.  Each
    reproduces what a context does alone afterwards.  A process of its own: the threads start behind a barrier right after the
    contexts exist, nothing has been launched before."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, threading
        import numpy as np
        sys.path.insert(0, %r)
        from afesp_amd.capi import Engine, first_use_count
        engs = [Engine(0), Engine(0)]
        gate = threading.Barrier(2)
        res = [None, None]
        def work(k):
            gate.wait()
            e = engs[k]
            e.synthetic_init(6 + k, 22 - 2 * k, 0.03, 7 + k, 6)
            nit, en, rm = e.do_ccsd_spatial(10, 1e-9, 1e-9)
            res[k] = (nit, en.copy(), e.do_ccsd_t_spatial())
        th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        assert all(r is not None for r in res), res
        n_sites = first_use_count()
        assert n_sites > 20, n_sites          # the launch sites resolved their kernels under the lock
        for k in range(2):                    # ... and alone, afterwards, each gets the same numbers
            with Engine(0) as e:
                e.synthetic_init(6 + k, 22 - 2 * k, 0.03, 7 + k, 6)
                nit, en, rm = e.do_ccsd_spatial(10, 1e-9, 1e-9)
                t = e.do_ccsd_t_spatial()
            assert nit == res[k][0] and np.max(np.abs(en - res[k][1])) < 1e-13, (k, en, res[k][1])
            assert np.max(np.abs(t - res[k][2])) < 1e-13 * max(1.0, np.max(np.abs(t))), (k, t, res[k][2])
        for e in engs: e.close()
        print("ok", n_sites)
    """) % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "a-fortran-electronic-structure-program_amd")
    env = dict(os.environ, AFESP_SMALL_MAX="0", AFESP_RING_TG_MIN="1", AFESP_TALL_MIN="64")
    env.pop("AFESP_NO_PRELOAD", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("o,v,tiny_pool", [(6, 16, True), (7, 12, False)])
def test_cost_balanced_shard_bounds_partition_the_list(o, v, tiny_pool, monkeypatch):
    """afesp_ccsd_t_shard_bounds: monotone, exhaustive for every world size (also more ranks than triples), and the shards
    add up to the whole evaluation."""
    from afesp_amd.capi import Engine
    if tiny_pool:
        monkeypatch.setenv("AFESP_T_POOL_GIB", "0")
    n, e, eri = molecules.synthetic_system(o, v, scale=0.04, seed=5 + v)
    eng = Engine(0)
    try:
        eng.ccsd_init(o, v, e, eri, 6)
        eng.do_ccsd_spatial(40, 1e-9, 1e-9)
        nt = eng.ntriples()
        whole = eng.do_ccsd_t_spatial()
        for world in (1, 2, 3, 8, nt + 3):
            b = eng.shard_bounds(world)
            assert len(b) == world + 1 and b[0] == 0 and b[-1] == nt
            assert all(b[r] <= b[r + 1] for r in range(world))
        for world in (3, 8):
            b = eng.shard_bounds(world)
            parts = sum(eng.do_ccsd_t_spatial(b[r], b[r + 1]) for r in range(world))
            assert np.max(np.abs(parts - whole)) < 1e-12 * max(1.0, np.max(np.abs(whole)))
        assert eng.shard_bounds(4, cr=True)[-1] == nt
    finally:
        eng.close()
