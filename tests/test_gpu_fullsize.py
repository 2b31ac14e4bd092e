"""Config 5 (o=20, v=200, the size BASELINE.json's roofline numbers are quoted on): the oracle cannot run a whole step at
this size in test time, so parity is shown through size-independent properties and through oracle evaluations of a few
triples on the tensors the device holds."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
O, V = 20, 200


@pytest.fixture(scope="module")
def big():
    from afesp_amd.capi import Engine
    e = Engine(0)
    e.synthetic_init(O, V, 0.005, 12345, 8)
    e.ccsd_energy()
    energies = []
    for _ in range(2):
        energies.append(e.ccsd_iterate()[0])
        e.ccsd_diis()
    e.fullsize_energies = energies
    yield e
    e.close()


def test_amplitude_symmetry_and_energy_functional(big):
    """t2(i,j,a,b) = t2(j,i,b,a) is preserved by the equations (and by the a<=b evaluation of the pp-ladder); the energy the
    device reports is the functional of ccsd.f90:1775 evaluated on the host from the downloaded tensors."""
    t1, t2 = big.amplitudes()
    assert np.max(np.abs(t2 - t2.transpose(1, 0, 3, 2))) < 1e-13
    v_oovv = big.tensor("v_oovv")
    big.ccsd_energy()
    e, rms, conv = big.ccsd_energy()
    w = 2.0 * v_oovv - v_oovv.transpose(0, 1, 3, 2)
    ref = float(np.sum(w * (t2 + np.einsum("ia,jb->ijab", t1, t1))))
    assert abs(e - ref) < 1e-10 * max(1.0, abs(ref))
    assert rms == 0.0           # second call in a row: t2_old == t2 (ccsd.f90:1803-1806)


def test_triples_shards_add_up_and_match_oracle_on_device_tensors(big):
    nt = big.ntriples()
    assert nt == O * (O + 1) * (O + 2) // 6
    full = big.do_ccsd_t_spatial()
    plain = big.do_ccsd_t_spatial_plain()          # the variant bench.py times: E[T], E(T) with one Z evaluation per element
    assert np.max(np.abs(plain - full[:2])) < 1e-12 * np.max(np.abs(full[:2]))
    cuts = [0, 2, nt // 7, nt // 2, nt - 5, nt]
    parts = sum(big.do_ccsd_t_spatial(a, b) for a, b in zip(cuts[:-1], cuts[1:]))
    assert np.max(np.abs(parts - full)) < 1e-11 * np.max(np.abs(full))
    # first two sorted triples (0,0,0), (0,0,1) = ordered triples 0, 1, o, o^2 of the reference's enumeration
    t1, t2 = big.amplitudes()
    f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
    e = np.concatenate([-2.0 + np.arange(O) / (O - 1), 1.0 + 2.0 * np.arange(V) / (V - 1)])
    vvov, oovo, oovv = f(big.tensor("v_vvov")), f(big.tensor("v_oovo")), f(big.tensor("v_oovv"))
    L = orc.lib()
    ref = np.zeros(4)
    for lo, hi in ((0, 2), (O, O + 1), (O * O, O * O + 1)):
        out = np.zeros(4)
        L.orc_ccsd_t(O, V, e, f(t1), f(t2), vvov, oovo, oovv, lo, hi, out)
        ref += out
    got = big.do_ccsd_t_spatial(0, 2)
    assert np.max(np.abs(got - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("pool_gib,t_gemm", [("6", None), ("6", "gett"), (None, None)])
def test_first_block_of_triples_against_the_blas_backed_restatement(big, monkeypatch, pool_gib, t_gemm):
    """All sorted triples i <= j <= k over the first s occupied orbitals (the first block triple of the engine's enumeration: coinciding
    pairs, a triple with i = j = k, every multiplicity) = all s^3 ordered triples of the reference's loop over those orbitals,
    evaluated by the dgemm-per-term restatement (oracle/afesp_oracle_blas.c, pinned to the loop form by tests/test_oracle_golden.py)
    on the tensors the device holds.  A smaller pool than the default makes s = 3 (27 reference triples); the default pool
    s = 5 (125)."""
    L = orc.blas_lib()
    if L is None:
        pytest.skip("numpy's bundled OpenBLAS not found")
    if pool_gib:
        monkeypatch.setenv("AFESP_T_POOL_GIB", pool_gib)
    if t_gemm:
        monkeypatch.setenv("AFESP_T_GEMM", t_gemm)   # the fallback kernel of the products (csrc/gett_grouped.hip) at full size
    s = big.t_block_size()
    assert 2 <= s <= (4 if pool_gib else 8)
    nsorted = s * (s + 1) * (s + 2) // 6
    got = big.do_ccsd_t_spatial(0, nsorted)
    t1, t2 = big.amplitudes()
    f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
    e = np.concatenate([-2.0 + np.arange(O) / (O - 1), 1.0 + 2.0 * np.arange(V) / (V - 1)])
    args = (O, V, e, f(t1), f(t2), f(big.tensor("v_vvov")), f(big.tensor("v_oovo")), f(big.tensor("v_oovv")))
    ref = np.zeros(4)
    for i in range(s):
        for j in range(s):               # ordered triples (i, j, 0..s-1) are consecutive in the reference's flat order
            out = np.zeros(4)
            lo = (i * O + j) * O
            assert L.orcb_ccsd_t(*args, lo, lo + s, out) == 0
            ref += out
    assert np.max(np.abs(got - ref)) < 1e-11 * max(1.0, np.max(np.abs(ref))), (got, ref)


def test_completely_renormalised_variant_at_full_size_carries_the_same_sums(big):
    """The completely renormalised evaluation (second pool of moment blocks, its own orbit-kernel instantiation at two workgroups
    per CU) returns E[T], E(T), D[T], D(T) of the plain evaluation on a shard of config 5, and finite moment sums that add up
    over sub-shards."""
    big.build_cr_intermediates()
    nt = big.ntriples()
    hi = nt // 12
    ref = big.do_ccsd_t_spatial(0, hi)
    cr = big.do_ccsd_t_spatial_cr(0, hi)
    assert np.max(np.abs(cr[:4] - ref)) < 1e-12 * max(1.0, np.max(np.abs(ref)))
    assert np.all(np.isfinite(cr)) and abs(cr[4]) > 0.0
    parts = big.do_ccsd_t_spatial_cr(0, hi // 3) + big.do_ccsd_t_spatial_cr(hi // 3, hi)
    assert np.max(np.abs(parts - cr)) < 1e-11 * max(1.0, np.max(np.abs(cr)))


def test_one_amplitude_update_at_config_5_against_the_pinned_restatement(capsys):
    """Config 5, element by element: all twelve intermediates, both residuals and the updated t1 / t2 of one CCSD update from
    non-trivial amplitudes against the loop-form restatement (the one the reference's bundled outputs pin) on the same hashed
    integrals -- 1e-10 relative; measured 3.4e-14 (profiles/r02_full_iteration_check.txt).  The restatement needs about two
    minutes on the box's 16 host threads, by far the longest test of the suite (tools/full_iteration_check.py does the work)."""
    import importlib.util, os, sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "full_iteration_check.py")
    spec = importlib.util.spec_from_file_location("full_iteration_check", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv, sys.argv = sys.argv, [path, str(O), str(V)]
    try:
        rc = mod.main()
    finally:
        sys.argv = argv
    assert rc == 0, capsys.readouterr().out[-2000:]


def _hash_uniform(k, seed):
    """numpy twin of the device generator (csrc/capi.hip, splitmix64) used by afesp_synthetic_ao / afesp_synthetic_init"""
    with np.errstate(over="ignore"):
        x = (k.astype(np.uint64) + np.uint64(seed)) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / 9007199254740992.0


def _tri(i, j):
    hi, lo = np.maximum(i, j), np.minimum(i, j)
    return hi * (hi + 1) // 2 + lo


def _packed(i, j, k, l):
    return _tri(_tri(i, j), _tri(k, l))          # integrals.f90:196-210, 0-based


@pytest.mark.parametrize("blocked", [None, "1"])
def test_ao2mo_full_size_with_a_signed_permutation(blocked, monkeypatch):
    """n = 220 (config 5), the whole tensor at once (what this size runs) and slab by slab (what n >= 256 runs; forced):
    with C a signed permutation matrix every MO integral is one AO integral with a sign,
    (pq|rs) = s_p s_q s_r s_s (P(p)P(q)|P(r)P(s)) -- exact in floating point, so the pair-squaring, the four quarter transforms over
    unique pairs and the repack are checked element by element (200 000 sampled index quadruples) at the full size."""
    from afesp_amd.capi import Engine
    if blocked:
        monkeypatch.setenv("AFESP_AO2MO_BLOCKED", blocked)
    n, o, scale, seed = O + V, O, 0.02, 4242
    rng = np.random.default_rng(17)
    perm = rng.permutation(n)
    sign = rng.choice([-1.0, 1.0], n)
    c = np.zeros((n, n))
    c[np.arange(n), perm] = sign                  # canon_coeff(MO, AO)
    e = np.concatenate([-2.0 + np.arange(o) / (o - 1), 1.0 + 2.0 * np.arange(n - o) / (n - o - 1)])
    with Engine(0) as eng:
        eng.synthetic_ao(n, scale, seed)
        e_mp2, mo = eng.do_mp2_spatial(n, o, c, e, None)
    assert np.isfinite(e_mp2)
    p, q, r, s = (rng.integers(0, n, 200_000) for _ in range(4))
    ao_index = _packed(perm[p], perm[q], perm[r], perm[s]).astype(np.uint64)
    expect = sign[p] * sign[q] * sign[r] * sign[s] * scale * (2.0 * _hash_uniform(ao_index, seed) - 1.0)
    assert np.array_equal(mo[_packed(p, q, r, s)], expect)


def test_fock_build_full_size_against_the_defining_sum():
    """n = 220: sampled elements of F = H + sum_kl D(k,l) [2 (ij|kl) - (ik|jl)] (hf.f90:349-385) evaluated on the host from the
    same hashed integrals, a non-symmetric density included; linearity in D."""
    from afesp_amd.capi import Engine
    n, scale, seed = O + V, 0.02, 99
    rng = np.random.default_rng(23)
    h = rng.standard_normal((n, n))
    d1, d2 = rng.standard_normal((n, n)), rng.standard_normal((n, n))
    d2 = d2 + d2.T
    with Engine(0) as eng:
        eng.synthetic_ao(n, scale, seed)
        f1, f2, f12 = eng.build_fock(n, d1, h), eng.build_fock(n, d2, h), eng.build_fock(n, d1 + d2, h)
    assert np.max(np.abs((f12 - h) - (f1 - h) - (f2 - h))) < 1e-11 * np.max(np.abs(f12))
    g2 = f2 - h                                   # the two-electron part is symmetric for a symmetric density
    assert np.max(np.abs(g2 - g2.T)) < 1e-11 * np.max(np.abs(g2))
    k, l = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    val = lambda idx: scale * (2.0 * _hash_uniform(idx.astype(np.uint64), seed) - 1.0)
    for (i, j) in [(0, 0), (3, 217), (219, 5), (111, 112), (219, 219)]:
        ref = h[i, j] + np.sum(d1 * (2.0 * val(_packed(i, j, k, l)) - val(_packed(i, k, j, l))))
        assert abs(f1[i, j] - ref) < 1e-11 * max(1.0, abs(ref)), (i, j)


def test_pp_ladder_pair_form_equals_the_plain_form_at_full_size(monkeypatch):
    """Config 5: one amplitude update with the particle-particle ladder in its symmetric/antisymmetric pair form (what this size
    runs) and one with the a <= b form, from the same amplitudes: two different sets of GEMMs, the same t2."""
    from afesp_amd.capi import Engine
    res = []
    for form in ("1", "0"):
        monkeypatch.setenv("AFESP_PP_SYM", form)
        with Engine(0) as eng:
            eng.synthetic_init(O, V, 0.005, 12345, 2)
            eng.ccsd_energy()
            eng.ccsd_iterate()
            en = eng.ccsd_iterate()[0]
            res.append((en, eng.amplitudes()[1]))
    assert abs(res[0][0] - res[1][0]) < 1e-10 * max(1.0, abs(res[0][0]))
    assert np.max(np.abs(res[0][1] - res[1][1])) < 1e-12 * max(1.0, np.max(np.abs(res[1][1])))
