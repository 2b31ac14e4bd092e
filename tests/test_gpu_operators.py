"""GPU parity of the operator layer (gather-GEMM, permute) against the oracle / numpy, through the C-ABI."""
import itertools

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from afesp_amd.capi import Engine
    e = Engine(0)
    yield e
    e.close()


def _rand(rng, *shape):
    return np.asfortranarray(rng.uniform(-1.0, 1.0, size=shape))


def test_mfma_layout_identity_times_asymmetric(eng):
    # A = I with an asymmetric B catches a transposed C/D register map (cdna_hip_programming.md section 3)
    n = 48
    B = np.asfortranarray(np.arange(n * n, dtype=float).reshape(n, n) * 0.001 + np.arange(n)[:, None] * 0.37)
    C = eng.gemm("N", "N", n, n, n, np.eye(n), B)
    assert np.array_equal(C, B)
    C2 = eng.gemm("N", "N", n, n, n, B, np.eye(n))
    assert np.array_equal(C2, B)


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (5, 7, 3), (16, 16, 4), (33, 65, 17), (128, 128, 64), (130, 257, 45),
                                   (25, 361, 361), (400, 96, 512), (3, 1000, 9)])
@pytest.mark.parametrize("ta,tb", [("N", "N"), ("T", "N"), ("N", "T"), ("T", "T")])
def test_dgemm_wrapper_parity(eng, m, n, k, ta, tb):
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    A = _rand(rng, *((k, m) if ta == "T" else (m, k)))
    B = _rand(rng, *((n, k) if tb == "T" else (k, n)))
    C0 = _rand(rng, m, n)
    ref = np.ascontiguousarray(C0.ravel(order="F")).copy()
    orc.lib().orc_gemm(int(ta == "T"), int(tb == "T"), m, n, k, 0.75, np.ascontiguousarray(A.ravel(order="F")),
                       np.ascontiguousarray(B.ravel(order="F")), -0.5, ref)
    got = eng.gemm(ta, tb, m, n, k, A, B, C0, alpha=0.75, beta=-0.5)
    # fp64, tolerance: K products of |x|<=1 accumulated in a different order -> a few ulp * K
    assert np.max(np.abs(got.ravel(order="F") - ref)) < 1e-13 * max(k, 1) + 1e-14


def test_gemm_k_zero_and_beta_zero_ignores_nan(eng):
    C0 = np.full((4, 5), np.nan, order="F")
    A = np.ones((4, 3), order="F")
    B = np.ones((3, 5), order="F")
    got = eng.gemm("N", "N", 4, 5, 3, A, B, C0, alpha=1.0, beta=0.0)
    assert np.array_equal(got, np.full((4, 5), 3.0))


@pytest.mark.parametrize("split,tm,tn", [(1, 1, 1), (3, 1, 4), (4, 4, 4), (2, 2, 2), (7, 4, 1), (0, 0, 0)])
def test_contract_tile_shapes_and_split_k(eng, split, tm, tn):
    rng = np.random.default_rng(7)
    o, v = 5, 11
    t2 = _rand(rng, o, o, v, v)
    I = _rand(rng, o, v, o, v)
    C0 = _rand(rng, o, o, v, v)
    ref = C0 - 2.0 * np.einsum("mjae,iemb->ijab", t2, I)
    got = eng.contract(-2.0, t2, "mjae", I, "iemb", 1.0, C0, "ijab", force_split=split, force_tm=tm, force_tn=tn)
    assert np.max(np.abs(got - ref)) < 1e-12


@pytest.mark.parametrize("tn,beta", [(7, 0.0), (6, -0.5)])
def test_stream_k_pieces_of_long_tiles(eng, tn, beta):
    """The 256 x 112 / 256 x 96 tiles of the pp-ladder's pair products run stream-K where whole (tile, K slice) items would leave a round
    of the device partly idle (csrc/gett.hip, SK): every workgroup takes the same number of consecutive K steps of the tiles' sequence,
    the pieces of a tile meet in the split-K slabs.  Ragged rows, columns and K tail; 60 tiles of 301 steps (304 in the sequence: workgroups
    eight apart start at the same step of their tiles) -> 76 steps per workgroup, up to five pieces per tile, an accumulating product included."""
    rng = np.random.default_rng(11)
    M, N, K = 7670, 16 * tn * 2 - (18 if tn == 7 else 2), 4806
    A = _rand(rng, K, M)
    B = _rand(rng, N, K)
    # (both orders of the result: the tall extent as the tile's 256 rows -- the ladder's own shape -- and as its columns)
    for lc in ("nm", "mn"):
        C0 = _rand(rng, *((M, N) if lc == "mn" else (N, M)))
        prod = A.T @ B.T
        ref = beta * C0 + 0.5 * (prod if lc == "mn" else prod.T)
        got = eng.contract(0.5, A, "km", B, "nk", beta, C0, lc, force_tm=16, force_tn=tn)
        assert np.max(np.abs(got - ref)) < 1e-13 * K, lc


def test_contract_gemv_shapes(eng):
    rng = np.random.default_rng(3)
    o, v = 4, 9
    w = _rand(rng, o, o, v, v)
    t1 = _rand(rng, o, v)
    got = eng.contract(1.0, w, "miea", t1, "me", 0.0, np.zeros((v, o), order="F"), "ai")
    assert np.max(np.abs(got - np.einsum("miea,me->ai", w, t1))) < 1e-12
    got = eng.contract(1.0, t1, "me", w, "miea", 0.0, np.zeros((v, o), order="F"), "ai")
    assert np.max(np.abs(got - np.einsum("miea,me->ai", w, t1))) < 1e-12


# (o, v): the skinny extent S = o and the summed extent K = v of the streamed products (csrc/tall.hip): one and two 16-column
# fragments, K a multiple of 4 and not, K below / at / above one chunk of 16 loads, a tall extent that is not a multiple of 16
@pytest.mark.parametrize("o,v", [(3, 37), (16, 24), (20, 50), (32, 21), (1, 130), (7, 64)])
@pytest.mark.parametrize("beta", [0.0, 1.0, -0.5])
def test_tall_skinny_products_of_the_iteration(eng, o, v, beta, monkeypatch):
    """The products of t1 with a four-index array that stream their large operand (tall x skinny, csrc/tall.hip): the label forms of
    src/ccsd.f90:1165-1191, :1275-1290, :1700, :1255-1272 at extents past the kernel's thresholds, both orientations of C (lanes along
    the tall or along the skinny index), summed index contiguous or strided, accumulating or not -- against numpy and against
    the gather kernel on the same operands (AFESP_TALL is read once: the gather kernel is forced through its tile arguments)."""
    monkeypatch.setenv("AFESP_TALL_MIN", "8192")   # (the default threshold is 2^17 rows)
    rng = np.random.default_rng(100 * o + v)
    n_ia = -(-9000 // v)                        # enough (i, a) pairs for 8192 rows (b, i, a) of the tall index
    oi, va = (n_ia, 1) if n_ia <= 40 else (40, -(-n_ia // 40))
    t1 = _rand(rng, o, v)
    g = _rand(rng, v, v, oi, va)               # <eb|ia>-shaped: (v, v, o', v')
    cases = [
        ("je", t1, "ebia", g, "jbia", (o, v, oi, va)),     # t(j,e) <eb|ia>: summed index fastest in the tall operand, C lanes along j
        ("beia", g, "je", t1, "bjia", (v, o, oi, va)),     # <be|ia> t(j,e): summed index second, C lanes along b (the tall side)
        ("ie", t1, "baje", np.asfortranarray(g.transpose(0, 1, 2, 3)), "ijab", None),   # filled in below
    ]
    h = _rand(rng, v, va, oi, v)               # <ab|je>-shaped (b, a, j, e): summed index slowest
    cases[2] = ("ie", t1, "baje", h, "ijab", (o, oi, va, v))
    before = eng.launch_counts()["tall"]     # (per context: include/afesp.h, afesp_launch_counts)
    for la, A, lb, B, lc, shape in cases:
        C0 = _rand(rng, *shape)
        ref = beta * C0 + 1.5 * np.einsum(f"{la},{lb}->{lc}", A, B)
        got = eng.contract(1.5, A, la, B, lb, beta, C0.copy(order="F"), lc)
        assert np.max(np.abs(got - ref)) < 1e-11, (la, lb, lc)
        gather = eng.contract(1.5, A, la, B, lb, beta, C0.copy(order="F"), lc, force_tm=1, force_tn=1)
        assert np.max(np.abs(got - gather)) < 1e-11, (la, lb, lc)
    # (o = 1: C(i,j,a,b) then runs along j first, across the fastest index of <ab|je> -- that one stays with the gather kernel)
    taken = eng.launch_counts()["tall"] - before
    assert taken == (3 if o > 1 else 2), "the streamed kernel did not take these products"
    # a product that reads its tall operand across its fastest index stays with the gather kernel
    m = _rand(rng, o, v, oi * va, v)           # <mb|ie>-shaped (m, b, i, e)
    got = eng.contract(1.0, t1, "je", m, "mbie", 0.0, np.zeros((oi * va, o, o, v), order="F"), "ijmb")
    assert np.max(np.abs(got - np.einsum("je,mbie->ijmb", t1, m))) < 1e-11
    assert eng.launch_counts()["tall"] - before == taken


@pytest.mark.parametrize("order", ["".join(p) for p in itertools.permutations("1234")])
def test_omp_reshape_all_24_orders(eng, order):
    rng = np.random.default_rng(11)
    x = _rand(rng, 3, 4, 5, 6)
    dims = (orc.i64 * 4)(*x.shape)
    oshape = tuple(x.shape[int(c) - 1] for c in order)
    ref = np.zeros(x.size)
    orc.lib().orc_permute4(dims, order.encode(), np.ascontiguousarray(x.ravel(order="F")), ref, 0, 0.0)
    got = eng.omp_reshape(x, order)
    assert np.array_equal(got.ravel(order="F"), ref)
    # crib from SURVEY.md 8(a): '3124' means out(k,i,j,l) = in(i,j,k,l)
    if order == "3124":
        assert got[2, 1, 3, 4] == x[1, 3, 2, 4]
    # beta form
    y0 = _rand(rng, *oshape)
    ref2 = np.ascontiguousarray(y0.ravel(order="F")).copy()
    orc.lib().orc_permute4(dims, order.encode(), np.ascontiguousarray(x.ravel(order="F")), ref2, 1, 0.5)
    got2 = eng.omp_reshape(x, order, out_arr=y0, beta=0.5)
    assert np.array_equal(got2.ravel(order="F"), ref2)


@pytest.mark.parametrize("order", ["".join(p) for p in itertools.permutations("1234")])
def test_omp_reshape_transposing_orders_through_the_tiled_kernel(eng, order):
    """Extents past the thresholds of the LDS-tiled permute (unit-stride index changes, both extents >= 8, >= 2^16 elements),
    not multiples of the 32 x 32 tile; bit-exact, with and without beta."""
    rng = np.random.default_rng(12)
    x = _rand(rng, 9, 35, 41, 10)
    dims = (orc.i64 * 4)(*x.shape)
    oshape = tuple(x.shape[int(c) - 1] for c in order)
    ref = np.zeros(x.size)
    orc.lib().orc_permute4(dims, order.encode(), np.ascontiguousarray(x.ravel(order="F")), ref, 0, 0.0)
    got = eng.omp_reshape(x, order)
    assert np.array_equal(got.ravel(order="F"), ref)
    y0 = _rand(rng, *oshape)
    ref2 = np.ascontiguousarray(y0.ravel(order="F")).copy()
    orc.lib().orc_permute4(dims, order.encode(), np.ascontiguousarray(x.ravel(order="F")), ref2, 1, 0.5)
    got2 = eng.omp_reshape(x, order, out_arr=y0, beta=0.5)
    assert np.array_equal(got2.ravel(order="F"), ref2)
