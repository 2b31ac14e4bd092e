// ccsd.hip -- spin-free CCSD (Piecuch et al., CPC 149 (2002) 71) on the device.
//
// Follows the fixed-point path of the reference exactly -- MP1 start, Jacobi update, DIIS from iteration 1,
// the same convergence rule (src/ccsd.f90:279-402) -- but every contraction site of
// update_restricted_intermediates (src/ccsd.f90:1040-1312) and update_amplitudes_restricted (:1538-1732)
// is a single gather-GEMM on the resident tensors: no reshape temporaries, and the integrals are never
// antisymmetrised in place (the reference mutates and restores v_oovv/v_vvov/v_oovo every iteration,
// :1089,:1101-1126; here the three "2x - x^T" companions are built once at init).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "ccsd.h"
#include "comm.h"
#include "fused.h"

namespace afesp {

static inline int64_t tri64(int64_t i, int64_t j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// A state of the same extents can be initialised again where it lies: every buffer, table and plan keeps its address, so the programs
// compiled for it (fused.h: recording and compiling them is 0.7-0.8 ms, a third of a small molecule's whole solve) stay valid -- the
// next point of a scan over geometries (utils/els_wrapper.py re-enters with the same shape) starts at the steady iteration time.
// AFESP_CC_REINIT=0: always from scratch.
bool ccsd_can_reinit(const CCState& s, int o, int v, int diis_nerr)
{
    if (!knobs().cc_reinit) return false;
    return s.ready && s.o == o && s.v == v && s.nerr == diis_nerr && s.pp_sym == pp_sym_pays(o, v);
}

void ccsd_init(Context& cx, CCState& s, int o, int v, const double* eri_mo_dev, const double* e_host, int diis_nerr)
{
    if (o <= 0 || v <= 0) throw Error(1, "ccsd_init: need at least one occupied and one virtual orbital");
    const bool again = ccsd_can_reinit(s, o, v, diis_nerr);
    if (again) {
        cx.quiesce();
        s.ready = false;
        if (s.eri_own) { cx.release(s.eri_own); s.eri_own = nullptr; }
        ring_reinit(s);
        s.cs_packed = s.partials_live = false;
        s.tail_pending = false;
        s.nact = 0; s.it = 0;
        s.hist_plain = 0;
    } else {
        ccsd_free(cx, s);
    }
    s.o = o; s.v = v;
    const int64_t O = o, V = v, n = o + v, ov = O * V, o2v2 = O * O * V * V;
    // (a state initialised again keeps what it has: `T` / `A` allocate only for a new one)
    auto T = [&](Tensor& t, std::initializer_list<int64_t> dims) { if (!again) t = cx.tensor(dims); };
    auto A = [&](double*& ptr, int64_t nd) { if (!again) ptr = cx.alloc(nd); };
    A(s.e, n);
    AFESP_HIP(hipMemcpyAsync(s.e, e_host, sizeof(double) * n, hipMemcpyHostToDevice, cx.stream));
    cx.sync();
    T(s.v_oovv, {O, O, V, V}); T(s.v_ovov, {O, V, O, V}); T(s.v_vvov, {V, V, O, V});
    T(s.v_oovo, {O, O, V, O}); T(s.v_oooo, {O, O, O, O});
    s.pp_sym = pp_sym_pays(O, V);
    s.eri_src = eri_mo_dev;
    if (!s.pp_sym) T(s.v_vvvv, {V, V, V, V});   // the plain ladder reads <ef|ab>; the pair form is built from the packed array
    else if (again && s.v_vvvv.d) { cx.release(s.v_vvvv.d); s.v_vvvv = Tensor(); }   // (formed on request for the last system)
    T(s.w_oovv, {O, O, V, V}); T(s.w_oovo, {O, O, V, O});   // (w_vvov: on first use, ccsd_need_w_vvov)
    // ccsd.f90:496-512: <pq|rs> = (pr|qs), virtual offsets removed
    k_slice_phys(cx, s.v_oovv.d, eri_mo_dev, o, o, v, v, 0, 0, o, o);
    k_slice_phys(cx, s.v_ovov.d, eri_mo_dev, o, v, o, v, 0, o, 0, o);
    k_slice_phys(cx, s.v_vvov.d, eri_mo_dev, v, v, o, v, o, o, 0, o);
    k_slice_phys(cx, s.v_oovo.d, eri_mo_dev, o, o, v, o, 0, 0, o, 0);
    k_slice_phys(cx, s.v_oooo.d, eri_mo_dev, o, o, o, o, 0, 0, 0, 0);
    if (!s.pp_sym) k_slice_phys(cx, s.v_vvvv.d, eri_mo_dev, v, v, v, v, o, o, o, o);
    k_antisym_pair(cx, s.w_oovv.d, s.v_oovv.d, O, O, V, V, 1);   // 2<ij|ab> - <ij|ba>   (ccsd.f90:1089)
    k_antisym_pair(cx, s.w_oovo.d, s.v_oovo.d, O, O, V, O, 0);   // 2<ij|ak> - <ji|ak>   (ccsd.f90:1121)
    // nothing writes these again while the state lives: contract() keeps the re-laid-out copies it makes of them
    const int64_t fid = ++cx.amp_clock;
    for (Tensor* t : {&s.v_oovv, &s.v_ovov, &s.v_vvov, &s.v_oovo, &s.v_oooo, &s.w_oovv, &s.w_oovo}) t->frozen = fid;
    s.frozen_id = fid;
    if (again && s.w_vvov.d) {   // (the companion made on first use: it follows the new integrals)
        k_antisym_pair(cx, s.w_vvov.d, s.v_vvov.d, s.v, s.v, s.o, s.v, 0);
        s.w_vvov.frozen = fid;
    }
    T(s.D1, {O, V}); T(s.D2, {O, O, V, V});
    k_denominators(cx, s.D1.d, s.D2.d, s.e, o, v);
    s.nvec = ov + o2v2;
    if (!again) {
        s.amp = cx.alloc(s.nvec);
        s.t1 = view(s.amp, {O, V}); s.t2 = view(s.amp + ov, {O, O, V, V});
        double* res = cx.alloc(s.nvec);
        s.r1 = view(res, {O, V}); s.r2 = view(res + ov, {O, O, V, V});
    } else {
        AFESP_HIP(hipMemsetAsync(s.t1.d, 0, sizeof(double) * ov, cx.stream));          // t1 = 0 (ccsd.f90:520)
        AFESP_HIP(hipMemsetAsync(s.t2_old.d, 0, sizeof(double) * o2v2, cx.stream));
    }
    T(s.t2_old, {O, O, V, V});
    T(s.I_vo, {V, O}); T(s.I_vv, {V, V}); T(s.I_oo_p, {O, O}); T(s.I_oo, {O, O});
    T(s.c, {O, O, V, V}); T(s.asym, {O, O, V, V}); T(s.x_voov, {V, O, O, V});
    T(s.I_oooo, {O, O, O, O}); T(s.I_ovov, {O, V, O, V}); T(s.I_voov, {V, O, O, V});
    T(s.I_ooov_p, {O, O, O, V}); T(s.z_ooov, {O, O, O, V});
    // pp-ladder (ccsd.f90:1669) over the symmetry-unique column pairs a <= b
    {
        const int64_t np = V * (V + 1) / 2, K2 = V * V, N2 = O * O;
        A(s.pp, N2 * np + O * O * V * V + O * V);   // [PP | r2_sh | r1_sh]: the one exchange buffer of a split iteration
        s.r2_sh = s.pp + N2 * np;
        s.r1_sh = s.r2_sh + O * O * V * V;
        std::vector<int64_t> tab;
        if (!s.pp_sym) {
            // rows p of <ef|ab> viewed as [ef x ab], all (i,j) columns
            tab.reserve((size_t)(np + 2 * K2 + N2));
            for (int64_t b = 0; b < V; ++b)
                for (int64_t a = 0; a <= b; ++a) tab.push_back((a + V * b) * K2);     // offAm[p]: column (a,b) of v_vvvv
            for (int64_t k = 0; k < K2; ++k) tab.push_back(k);                        // offAk[(e,f)]
            for (int64_t k = 0; k < K2; ++k) tab.push_back(N2 * k);                   // offBk[(e,f)] in c(i,j,e,f)
            for (int64_t x = 0; x < N2; ++x) tab.push_back(x);                        // offBn = offCn[(i,j)]
            for (int64_t pz = 0; pz < np; ++pz) tab.push_back(N2 * pz);               // offCm[p] in PP(i,j,p)
        } else {
            // symmetric + antisymmetric halves (see ccsd_pp_ladder): dense operands with even leading dimensions, so the
            // 16-byte staging applies whatever the parity of o and v
            auto even = [](int64_t x) { return (x + 1) & ~(int64_t)1; };
            const int64_t npa = V * (V - 1) / 2, ks = even(np), ka = even(npa), ns = even(O * (O + 1) / 2), na = even(O * (O - 1) / 2);
            s.pp_ks = ks; s.pp_ka = ka; s.pp_ns = ns; s.pp_na = na;
            // A row of V+- (and of W+-, which shares the row tables) is [ V(ef, .) : ks | c+-(mn, .) : ns ], and c+- has ns more rows
            // I+-(., mn): the hole-hole ladder (ccsd.f90:1673) as ns / na more summation steps of the same two products (ccsd_pp_ladder)
            const int64_t lds = ks + ns, lda = ka + na, kx = lds;
            s.pp_lds = lds; s.pp_lda = lda;
            // (the products of t2 with <ef|ia> for I_ooov_p use the same tables and the same P+- buffers: rows m = (i,a))
            const int64_t nm = std::max(np, O * V);
            s.pp_nm = nm;
            A(s.pp_vs, lds * np); A(s.pp_cs, ns * kx); A(s.pp_ps, ns * nm);
            A(s.ov_ws, lds * O * V);
            A(s.pp_ts, ns * ks);
            A(s.oo_vs, ns * ks); A(s.oo_xs, ns * ns);
            A(s.r1x, O * V);
            if (npa > 0 && na > 0) {
                A(s.pp_va, lda * npa); A(s.pp_ca, na * kx); A(s.pp_pa, na * nm);
                A(s.ov_wa, lda * O * V);
                A(s.pp_ta, na * ka);
                A(s.oo_va, na * ka); A(s.oo_xa, na * na);
            }
            k_vvvv_sympack_packed(cx, s.pp_vs, s.pp_va, eri_mo_dev, o, v, lds, lda);
            k_vvx_sympack(cx, s.ov_ws, s.ov_wa, s.v_vvov.d, v, O * V, lds, lda);
            k_c_sympack(cx, s.oo_vs, s.oo_va, s.v_oovv.d, s.o, s.v, ns, na, true);   // 1/2 (<ij|ef> +- <ij|fe>), 1/4 on e == f
            // [ x (k or n) | lds*m | lda*m | ns*k | na*k | ns*m | na*m ]: entries beyond the antisymmetric extents unused
            s.pp_kn = kx;
            for (int64_t k = 0; k < s.pp_kn; ++k) tab.push_back(k);
            for (int64_t m = 0; m < nm; ++m) tab.push_back(lds * m);
            for (int64_t m = 0; m < nm; ++m) tab.push_back(lda * m);
            for (int64_t k = 0; k < kx; ++k) tab.push_back(ns * k);
            for (int64_t k = 0; k < kx; ++k) tab.push_back(na * k);
            for (int64_t m = 0; m < nm; ++m) tab.push_back(ns * m);
            for (int64_t m = 0; m < nm; ++m) tab.push_back(na * m);
        }
        if (!again) {   // (the tables depend on the extents only)
            s.pp_tab = cx.alloc_i64((int64_t)tab.size());
            AFESP_HIP(hipMemcpyAsync(s.pp_tab, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
            cx.sync();
        }
    }
    // ccsd.f90:520-521: t1 = 0, t2 = v_oovv / D
    k_div(cx, s.t2.d, s.v_oovv.d, s.D2.d, o2v2);
    // ccsd.f90:577-615
    if (!again) diis_alloc(cx, s, diis_nerr);
    s.energy = s.energy_old = s.rms = 0.0;
    s.amp_epoch = ++cx.amp_clock;   // (a clock of the context: a re-initialised state never repeats an epoch)
    s.cr_epoch = ++cx.amp_clock;
    s.ready = true;
    cx.sync();
}

void ccsd_need_vvvv(Context& cx, CCState& s)
{
    if (s.v_vvvv.d) return;
    if (!s.eri_src)
        throw Error(1, "the <ef|ab> slice is formed on request from the packed MO integrals, and those have been replaced since "
                       "afesp_ccsd_init: initialise the solver again");
    const int64_t V = s.v;
    s.v_vvvv = cx.tensor({V, V, V, V});
    k_slice_phys(cx, s.v_vvvv.d, s.eri_src, s.v, s.v, s.v, s.v, s.o, s.o, s.o, s.o);
}

// 2<ab|ic> - <ba|ic> (ccsd.f90:1101), the o v^3 companion of <ab|ic>: only the product form of I_vv's t1 term reads it (small systems,
// the rank-split iteration) -- the large-system path takes that term from the diagonals of two products (ccsd_intermediates) and never
// allocates it (1.3 GB at o = 20, v = 200; 15 GB at o = 40, v = 360)
static void ccsd_need_w_vvov(Context& cx, CCState& s)
{
    if (s.w_vvov.d) return;
    s.w_vvov = cx.tensor({(int64_t)s.v, (int64_t)s.v, (int64_t)s.o, (int64_t)s.v});
    k_antisym_pair(cx, s.w_vvov.d, s.v_vvov.d, s.v, s.v, s.o, s.v, 0);
    s.w_vvov.frozen = s.frozen_id;
}

void ccsd_free(Context& cx, CCState& s)
{
    if (!s.o) return;
    if (s.eri_own) cx.release(s.eri_own);
    double* bufs[] = {s.e, s.v_oovv.d, s.v_ovov.d, s.v_vvov.d, s.v_oovo.d, s.v_oooo.d, s.v_vvvv.d, s.w_oovv.d, s.w_vvov.d,
                      s.w_oovo.d, s.D1.d, s.D2.d, s.amp, s.r1.d, s.t2_old.d, s.I_vo.d, s.I_vv.d, s.I_oo_p.d, s.I_oo.d, s.c.d,
                      s.asym.d, s.x_voov.d, s.I_oooo.d, s.I_ovov.d, s.I_voov.d, s.I_ooov_p.d, s.z_ooov.d, s.amp_s, s.hist_t,
                      s.hist_e, s.coef, s.I_vovv_pp.d, s.I_ooov_pp.d, s.pp, (double*)s.pp_tab, s.pp_vs, s.pp_va, s.pp_cs,
                      s.pp_ca, s.pp_ps, s.pp_pa, s.bmat, s.ov_ws, s.ov_wa, s.pp_ts, s.pp_ta, s.oo_vs, s.oo_va, s.oo_xs, s.oo_xa, s.r1x};
    for (double* b : bufs) cx.release(b);
    // Contraction plans are keyed by shape and stay valid for the life of the context (the AO->MO plans carry tables of
    // n^2 npair entries: rebuilding them costs a quarter of a second at n = 220), so a new system of the same extents finds
    // them again.  Their offset tables are device memory, though: a context that has seen many different systems drops them
    // here, where no captured iteration refers to them.
    if (cx.plans.size() > 1024 || cx.plan_bytes > ((size_t)4 << 30)) cx.plan_clear();
    cx.drop_scratch();
    triples_plan_free(s);
    ring_free(cx, s);
    s = CCState();
}

void diis_alloc(Context& cx, DiisRing& s, int diis_nerr)
{
    s.nerr = diis_nerr; s.nact = 0; s.it = 0;
    if (diis_nerr >= 2) {
        if (diis_nerr > 15) throw Error(1, "ccsd_init: ccsd_diis_n_errmat > 15 is not supported");
        s.amp_s = cx.alloc(s.nvec);
        s.hist_t = cx.alloc(s.nvec * diis_nerr);
        s.hist_e = cx.alloc(s.nvec * diis_nerr);
        s.coef = cx.alloc(32);
        s.bmat = cx.alloc((int64_t)diis_nerr * diis_nerr);
    }
}

void diis_save(Context& cx, DiisRing& s)
{
    if (s.nerr >= 2) k_copy(cx, s.amp_s, s.amp, s.nvec);   // ccsd.f90:342-343
}
void ccsd_diis_save(Context& cx, CCState& s) { diis_save(cx, s); }

static bool lanes_pay(const CCState& s);
static void ccsd_oooo_pair_form(Context& cx, CCState& s);

// Lanes pay when a launch cannot fill the device anyway: o^2 v^2 up to 2^20 elements (H2O/cc-pVTZ: 70 225).
bool ccsd_uses_lanes(const CCState& s) { return lanes_pay(s); }

static bool lanes_pay(const CCState& s)
{
    const bool off = knobs().no_lanes;
    // (tuning knob AFESP_SMALL_MAX: the largest o^2 v^2 that still takes the small-system paths)
    const int64_t small_max = knobs().small_max;
    return !off && !s.sharded && s.t2.size() <= small_max;
}

// Rank split of one iteration ("next"): with a communicator of more than one rank the o^3 v^3 ring products and
// the pp-ladder -- 20 of the 28 ms of an iteration at o = 20, v = 200 -- are evaluated for this rank's slice of a virtual
// index only, into buffers that start from zero; ONE all-reduce of [PP | r2_sh] then gives every rank the same residual
// and every rank applies the same update: amplitudes, DIIS history and energies stay replicated and identical.  The slice
// index is external to both the intermediate and the product that consumes it, so a rank builds exactly the part of
// I_ovov / I_voov it needs itself.  Everything else of the iteration (8 ms) is replicated.
// The split is OPT-IN: afesp_ccsd_set_split(ctx, 1) or AFESP_CC_SHARD=1 (0 / AFESP_CC_SHARD=0: replicas, whatever the other
// says).  A communicator alone does not change the algorithm of the iteration -- the 192 MB in-place all-reduce on the engine
// stream is a different path from the scalar sum of (T), and a caller should have checked it on its nodes (bench.py does: one
// iteration both ways from the same amplitudes) before timing or trusting it.
void ccsd_refresh_sharding(Context& cx, CCState& s)
{
    const int env = knobs().cc_shard;
    const int world = cx.comm ? cx.comm->world : 1;
    s.sh_world = world;
    s.sh_rank = cx.comm ? cx.comm->rank : 0;
    const bool on = cx.cc_split_mode >= 0 ? cx.cc_split_mode == 1 : env == 1;
    s.sharded = world > 1 && on && env != 0;
    // measurement only (tools/split_slice_time.py): AFESP_CC_TIME_SLICE="rank,world" without a communicator -- the iteration does a
    // rank's share of the split form and skips the exchange; the amplitudes that come out are NOT the iteration's
    if (!cx.comm)
        if (knobs().cc_time_slice) { s.sharded = true; s.sh_rank = knobs().cc_time_rank; s.sh_world = knobs().cc_time_world; }
}

// This rank's slice [v0, v1) of a virtual index in a split iteration (the whole range otherwise).  (Cuts on even indices only -- so
// that every slice keeps the 16-byte staging of the gather kernel -- were measured at o = 20, v = 200, 8 ranks: slices of 24 / 26
// instead of 25, the 26-wide ones a tile row past 512, the slowest rank 6.1 ms against 6.0: not kept.)
static void slice_bounds(const CCState& s, int64_t* v0, int64_t* v1)
{
    if (!s.sharded) { *v0 = 0; *v1 = s.v; return; }
    *v0 = (int64_t)s.v * s.sh_rank / s.sh_world;
    *v1 = (int64_t)s.v * (s.sh_rank + 1) / s.sh_world;
}

static Tensor slice_axis(Tensor t, int axis, int64_t lo, int64_t hi)
{
    t.d += lo * t.stride[axis];
    t.dim[axis] = hi - lo;
    return t;
}

void ccsd_intermediates(Context& cx, CCState& s, bool save_for_diis)
{
    auto C = [&](double al, const Tensor& A, const char* la, const Tensor& B, const char* lb, double be, const Tensor& Cc,
                 const char* lc) { contract(cx, al, A, la, B, lb, be, Cc, lc); };
    ccsd_refresh_sharding(cx, s);
    s.amps_touched = false;
    s.r1x_valid = false;
    // this rank's slice [v0, v1) of the last (virtual) index of I_ovov / I_voov: the whole range unless the iteration is split
    int64_t v0, v1;
    slice_bounds(s, &v0, &v1);
    auto sl = [&](const Tensor& t, int axis) { return slice_axis(t, axis, v0, v1); };
    // asym_t2, c_oovv                                                    ccsd.f90:1063-1079
    // (large systems: together with the copies of the amplitudes that the ring products read, ring.hip)
    if (!(lanes_pay(s) || cx.rec) && ring_tg_applies(s) && ring_tg_pack(cx, s)) {}
    else k_asym_c(cx, s.asym.d, s.c.d, s.t1.d, s.t2.d, s.o, s.v);
    // Small systems are bound by the latency of ~100 dependent launches: the independent chains below then run on four
    // lanes (streams) side by side.  Each intermediate is built entirely on one lane; I_vo feeds I_oo (same lane), x_voov feeds
    // I_voov (same lane) and the last term of I_ooov_p (explicit event).
    const bool par = lanes_pay(s) && !cx.rec;   // (a recorded sequence is levelled by the compiler of fused.h instead)
    const bool fused = cx.rec != nullptr;
    // large systems: the three o^3 v^3 products of I_ovov / I_voov in one launch of the LDS-DMA GEMM (ring.hip) -- the small terms are
    // formed here as always, the products are skipped
    const bool ring = !par && !fused && ring_tg_applies(s);
    if (!ring) ring_invalidate(s);   // (after the pack above: `ring` is the condition it ran under)
    auto lane = [&](int i) { if (par) cx.use_lane(i); };
    if (par) cx.fork(6);
    lane(0);
    // I_vo(a,i) = (2<im|ae> - <im|ea>) t(m,e)                            ccsd.f90:1085-1092
    C(1.0, s.w_oovv, "miea", s.t1, "me", 0.0, s.I_vo, "ai");
    lane(1);
    // I_vv(b,a)                                                          ccsd.f90:1096-1113
    // Large systems (one stream, whole tensors): its t1 term (2<eb|ma> - <be|ma>) t(m,e) is the m = i diagonal of two products
    // over <eb|ia> that the iteration forms anyway -- y(j,b,i,a) = t(j,e) <eb|ia> (the last term of I_ovov) and x_voov -- so the
    // pass over the 2x - x^T companion of <eb|ia> (o v^3 elements: 0.34 of 25 ms at o = 20, v = 200) is not made: y is written
    // first, into the empty I_ovov, and k_ivv_diag picks both diagonals up below.
    // (split iteration: the same, per slice of a -- every reader of I_vv then takes this rank's columns only, ccsd_amplitudes)
    const bool ivv_diag = !par && !fused && s.o >= 2;
    if (!ivv_diag) {
        ccsd_need_w_vvov(cx, s);
        C(1.0, s.w_vvov, "ebma", s.t1, "me", 0.0, s.I_vv, "ba");
    }
    if (!s.sharded) C(-1.0, s.w_oovv, "mneb", s.c, "mnea", ivv_diag ? 0.0 : 1.0, s.I_vv, "ba");
    // (split iteration with o == 1: the product form of the t1 term above has written all of I_vv -- this rank's columns add to it)
    else if (v1 > v0) C(-1.0, s.w_oovv, "mneb", sl(s.c, 3), "mnea", ivv_diag ? 0.0 : 1.0, slice_axis(s.I_vv, 1, v0, v1), "ba");
    lane(0);
    // I_oo_p(j,i)                                                        ccsd.f90:1115-1132
    C(1.0, s.w_oovo, "miej", s.t1, "me", 0.0, s.I_oo_p, "ji");
    C(1.0, s.asym, "mjef", s.v_oovv, "mief", 1.0, s.I_oo_p, "ji");
    // I_oo(j,i) = I_oo_p + t(j,e) I_vo(e,i)                              ccsd.f90:1134-1137
    // (launch-fused path: a copy of a result costs a dependency level of its own, an o^2 x o v^2 product costs nothing -- I_oo gets
    // the two products of I_oo_p again, in the stage where I_vo is ready)
    if (fused) {
        C(1.0, s.w_oovo, "miej", s.t1, "me", 0.0, s.I_oo, "ji");
        C(1.0, s.asym, "mjef", s.v_oovv, "mief", 1.0, s.I_oo, "ji");
    } else {
        k_copy(cx, s.I_oo.d, s.I_oo_p.d, s.I_oo.size());
    }
    C(1.0, s.t1, "je", s.I_vo, "ei", 1.0, s.I_oo, "ji");
    lane(1);
    // I_oooo(k,l,i,j)                                                    ccsd.f90:1139-1156
    k_copy(cx, s.I_oooo.d, s.v_oooo.d, s.I_oooo.size());
    if (s.pp_sym && !par && !fused) ccsd_oooo_pair_form(cx, s);   // c(kl,ef) <ij|ef> over pair indices: an eighth of the multiply-adds
    else C(1.0, s.c, "klef", s.v_oovv, "ijef", 1.0, s.I_oooo, "klij");
    C(1.0, s.t1, "ke", s.v_oovo, "ilej", 1.0, s.I_oooo, "klij");
    C(1.0, s.t1, "le", s.v_oovo, "jkei", 1.0, s.I_oooo, "klij");
    lane(2);
    // I_ovov(j,b,i,a)                                                    ccsd.f90:1158-1191
    // (a split iteration only ever reads this rank's slice of I_ovov / I_voov: every term is built for the slice only -- the last
    // index is the slowest one, a slice is one contiguous range)
    const int64_t a_len = s.I_ovov.stride[3], a_off = v0 * a_len, a_cnt = (v1 - v0) * a_len;
    if (ivv_diag && s.sharded) {
        if (v1 > v0) C(1.0, s.t1, "je", sl(s.v_vvov, 3), "ebia", 0.0, sl(s.I_ovov, 3), "jbia");
    } else if (ivv_diag) {
        // y and x_voov (below) in one launch: both stream <eb|ia> against t1, one along each of its two leading indices -- the o v^3
        // integrals cross HBM once for the two (tall.h, tall_dual_kernel; contract_pair falls back to the two products)
        contract_pair(cx, {1.0, &s.t1, "je", &s.v_vvov, "ebia", 0.0, &s.I_ovov, "jbia"}, {1.0, &s.v_vvov, "beia", &s.t1, "je", 0.0, &s.x_voov, "bjia"});
    } else if (v1 > v0) {
        k_copy(cx, s.I_ovov.d + a_off, s.v_ovov.d + a_off, a_cnt);
        if (!ring) C(-0.5, s.v_oovv, "mibe", sl(s.c, 2), "mjae", 1.0, sl(s.I_ovov, 3), "jbia");   // (o^3 v^3)
        C(-1.0, s.v_oovo, "mibj", sl(s.t1, 1), "ma", 1.0, sl(s.I_ovov, 3), "jbia");
        C(1.0, s.t1, "je", sl(s.v_vvov, 3), "ebia", 1.0, sl(s.I_ovov, 3), "jbia");
    }
    lane(3);
    // x_voov(b,j,i,a) = <be|ia> t(j,e)                                   ccsd.f90:1275-1290
    // (split iteration: only this rank's slice of a is ever read -- by I_voov and by the x_voov term of I_ooov_p below)
    if (s.sharded) { if (v1 > v0) C(1.0, sl(s.v_vvov, 3), "beia", s.t1, "je", 0.0, sl(s.x_voov, 3), "bjia"); }
    else if (!ivv_diag) C(1.0, s.v_vvov, "beia", s.t1, "je", 0.0, s.x_voov, "bjia");
    if (ivv_diag && v1 > v0) {
        k_ivv_diag(cx, s.I_vv.d, s.I_ovov.d, s.x_voov.d, s.o, s.v, (int)v0, (int)v1);
        k_axpby(cx, s.I_ovov.d + a_off, 1.0, s.v_ovov.d + a_off, 1.0, a_cnt);
        if (!ring) C(-0.5, s.v_oovv, "mibe", sl(s.c, 2), "mjae", 1.0, sl(s.I_ovov, 3), "jbia");   // (o^3 v^3)
        C(-1.0, s.v_oovo, "mibj", sl(s.t1, 1), "ma", 1.0, sl(s.I_ovov, 3), "jbia");
    }
    const int x_voov_ready = par ? cx.mark() : 0;
    // I_voov(b,j,i,a)                                                    ccsd.f90:1193-1252
    if (v1 > v0) {
        permute_add(cx, 1.0, sl(s.v_oovv, 2), "jiab", 0.0, sl(s.I_voov, 3), "bjia");
        // (launch-fused path: x_voov's product a second time, straight into I_voov, instead of adding the finished x_voov a level later)
        if (fused) C(1.0, s.v_vvov, "beia", s.t1, "je", 1.0, s.I_voov, "bjia");
        else k_axpby(cx, s.I_voov.d + a_off, 1.0, s.x_voov.d + a_off, 1.0, a_cnt);
        if (!ring) {
            C(0.5, s.w_oovv, "imbe", sl(s.t2, 3), "mjea", 1.0, sl(s.I_voov, 3), "bjia");       // (o^3 v^3 each)
            C(-0.5, s.v_oovv, "imbe", sl(s.c, 2), "mjae", 1.0, sl(s.I_voov, 3), "bjia");
        }
        C(-1.0, s.v_oovo, "imbj", sl(s.t1, 1), "ma", 1.0, sl(s.I_voov, 3), "bjia");
    }
    if (ring) ring_tg_intermediates(cx, s);
    lane(4);
    // (the copy of the amplitudes the DIIS error vector is taken against, ccsd.f90:342-343: nothing writes them before the update
    // at the end of ccsd_amplitudes, so it rides on a lane instead of standing in front of the iteration)
    if (save_for_diis) diis_save(cx, s);
    // I_vovv_p(c,i,a,b) = <ab|ci> - t(m,a) <mi|cb> - t(m,b) <ma|ic>          ccsd.f90:1255-1272, :1296-1299
    // is only ever contracted with t(i,e) over its first index (:1700), so the o v^3 tensor is not formed: the first term
    // is contracted from v_vvov directly (ccsd_amplitudes) and the two t1-dressed terms go through o^3 v tensors,
    //   y_ooov(i,m,j,b) = t(i,e) <mj|eb>,   y_oovo(i,m,a,j) = t(i,e) <ma|je>,
    //   r2(ijab) -= t(m,a) y_ooov(i,m,j,b) + t(m,b) y_oovo(i,m,a,j)
    // (three passes over o v^3 elements less per iteration; ccsd_build_I_vovv_p forms the tensor itself on request).
    // The residual only ever enters as r2(ijab) + r2(jiba) (P(ia/jb), t2_update_kernel), so a term may be replaced by its image
    // under (i <-> j, a <-> b): the second one becomes -t(m,a) y_oovo(j,m,b,i), and with the I_ooov_p term of :1705-1715,
    // -t(m,a) I_ooov_p(i,j,m,b), all three are ONE product with
    //   z_ooov(i,j,m,b) = y_ooov(i,m,j,b) + y_oovo(j,m,b,i) + I_ooov_p(i,j,m,b)
    // -- two passes over the o^2 v^2 residual less (0.27 ms of 27 at o = 20, v = 200; two launches less for a small system).
    // (launch-fused path: z_ooov starts from the bare term of I_ooov_p and collects every product of it directly, below)
    if (fused) permute_add(cx, 1.0, s.v_oovo, "kjai", 0.0, s.z_ooov, "jkia");
    C(1.0, s.t1, "ie", s.v_oovv, "mjeb", fused ? 1.0 : 0.0, s.z_ooov, "ijmb");
    C(1.0, s.t1, "je", s.v_ovov, "mbie", 1.0, s.z_ooov, "ijmb");
    lane(5);
    // I_ooov_p(j,k,i,a)                                                  ccsd.f90:1302-1308
    permute_add(cx, 1.0, s.v_oovo, "kjai", 0.0, s.I_ooov_p, "jkia");
    if (s.sharded) {
        // split iteration: the t2 <ef|ia> term for this rank's slice of a only, into a buffer of its own -- ccsd_amplitudes
        // contracts that slice with t1 into the rank-partial residual (the term is 1.3 of the iteration's 24.5 ms at o = 20, v = 200;
        // I_ooov_p itself then lacks it on every rank)
        Tensor osh = view(cx.scratch("ooov_sh", s.I_ooov_p.size()), {(int64_t)s.o, (int64_t)s.o, (int64_t)s.o, (int64_t)s.v});
        if (v1 > v0) {
            if (s.pp_sym) {
                AFESP_HIP(hipMemsetAsync(osh.d + osh.stride[3] * v0, 0, sizeof(double) * osh.stride[3] * (v1 - v0), cx.stream));
                ccsd_ooov_pair_form(cx, s, osh.d, v0, v1);
            } else {
                C(1.0, s.t2, "jkef", sl(s.v_vvov, 3), "efia", 0.0, sl(osh, 3), "jkia");
            }
            if (par) cx.wait(x_voov_ready);
            C(1.0, s.t1, "je", sl(s.x_voov, 3), "ekia", 1.0, sl(osh, 3), "jkia");   // (the x_voov term too: its slice only)
        }
    } else if (s.pp_sym) {
        ccsd_ooov_pair_form(cx, s);   // t2(jk,ef) <ef|ia> over pair indices, as the pp-ladder
    } else {
        C(1.0, s.t2, "jkef", s.v_vvov, "efia", 1.0, s.I_ooov_p, "jkia");
    }
    if (par) cx.wait(x_voov_ready);
    if (!s.sharded) C(1.0, s.t1, "je", s.x_voov, "ekia", 1.0, s.I_ooov_p, "jkia");
    if (par) cx.join();
    if (fused) {
        // z_ooov = y_ooov + y_oovo + I_ooov_p with every term of I_ooov_p added to it directly (the consumer of z_ooov then waits for
        // x_voov only, not for a pass over the finished I_ooov_p)
        C(1.0, s.t2, "jkef", s.v_vvov, "efia", 1.0, s.z_ooov, "jkia");
        C(1.0, s.t1, "je", s.x_voov, "ekia", 1.0, s.z_ooov, "jkia");
    } else {
        k_axpby(cx, s.z_ooov.d, 1.0, s.I_ooov_p.d, 1.0, s.z_ooov.size());
    }
}

// Particle-particle ladder (src/ccsd.f90:1669), the O(o^2 v^4) term.  pp(ijab) = sum_ef c(ij,ef) <ef|ab> obeys
// pp(ijab) = pp(jiba), so only the v(v+1)/2 column pairs a <= b are needed; the packed result PP(ij,p) enters the amplitude
// update directly (t2_update_kernel) instead of being accumulated into r2.
//
// Large systems split it once more (Scuseria, Janssen, Schaefer 1988): with c+- = c(ijef) +- c(ijfe) and
// V+- = <ef|ab> +- <fe|ab>, both (anti)symmetric under e <-> f, under i <-> j resp. a <-> b,
//   pp(ijab) = sum_{e<=f}' 1/2 c+ V+  +  sum_{e<f} 1/2 c- V-  =  Ps(ij,ab) + Pa(ij,ab),    pp(jiab) = Ps - Pa   (i <= j, a <= b),
// two products over pair indices only -- [o(o+1)/2] x [v(v+1)/2]^2 and [o(o-1)/2] x [v(v-1)/2]^2, a quarter of the
// reference's dgemm.  V+- are built once (the integrals are immutable), c+- per iteration.
// columns the launcher's tiles cover for n of them: 32 / 64 / multiples of 128, or of 112 / 96 at 0.93 of the wide tile's rate (gett.hip)
static int tile_cols(int64_t n)
{
    if (n <= 32) return 32;
    if (n <= 64) return 64;
    const int64_t wide = (n + 127) / 128 * 128, n7 = (int64_t)((double)((n + 111) / 112 * 112) / 0.93), n6 = (int64_t)((double)((n + 95) / 96 * 96) / 0.93);
    return (int)std::min(wide, std::min(n7, n6));
}
bool pp_sym_pays(int64_t O, int64_t V)
{
    if (knobs().pp_sym >= 0) return knobs().pp_sym == 1;
    // the split pays when it still halves the work after padding the o-pair extents to whole column tiles
    const double plain = (double)tile_cols(O * O) * (double)(V * V) * (double)(V * (V + 1) / 2);
    const double split = (double)tile_cols(O * (O + 1) / 2) * (double)(V * (V + 1) / 2) * (double)(V * (V + 1) / 2) +
                         (double)tile_cols(O * (O - 1) / 2) * (double)(V * (V - 1) / 2) * (double)(V * (V - 1) / 2);
    return plain > 1.5 * split;
}

// Sharded iteration: a rank evaluates the rows (a <= b) of its range of b only -- contiguous row ranges of both pair
// products, [b0 (b0+1)/2, b1 (b1+1)/2) and [b0 (b0-1)/2, b1 (b1-1)/2), the boundaries chosen for equal row counts -- and
// leaves the rest of PP zero for the all-reduce (ccsd_amplitudes).
static void pp_b_range(const CCState& s, int64_t* b0, int64_t* b1)
{
    const int64_t V = s.v;
    if (!s.sharded) { *b0 = 0; *b1 = V; return; }
    auto cut = [&](int r) { return (int64_t)std::llround(std::sqrt((double)r / (double)s.sh_world) * (double)V); };
    *b0 = s.sh_rank == 0 ? 0 : std::min(V, cut(s.sh_rank));
    *b1 = s.sh_rank + 1 == s.sh_world ? V : std::min(V, cut(s.sh_rank + 1));
}

// I_oooo(k,l,i,j) += sum_ef c(kl,ef) <ij|ef> (ccsd.f90:1139-1156) over pair indices: X(klij) = Xs + Xa for k <= l, i <= j (and
// X(lkij) = X(klji) = Xs - Xa), Xs = sum_{e<=f} c+(kl,ef) vs(ij,ef), Xa = sum_{e<f} c-(kl,ef) va(ij,ef) with vs / va = 1/2 (<ij|ef> +- <ij|fe>)
// built once.  Both operands are pair-packed (128 MB in all at o = 20, v = 200 against 256 MB) and the multiply-adds an eighth of the
// plain product's: 0.36 -> 0.08 ms.  c+- stay packed for the pp-ladder (s.cs_packed).
static void ccsd_oooo_pair_form(Context& cx, CCState& s)
{
    const int64_t ks = s.pp_ks, ka = s.pp_ka, ns = s.pp_ns, na = s.pp_na, nm = s.pp_nm, kx = s.pp_lds;
    const int64_t* t = s.pp_tab;
    const int64_t* u = t + s.pp_kn;
    k_c_sympack(cx, s.pp_cs, s.pp_ca, s.c.d, s.o, s.v, ns, na);
    s.cs_packed = true;
    GettProblem gp;
    gp.alpha = 1.0; gp.beta = 0.0;
    gp.nbatch = 1; gp.batchA = gp.batchB = gp.batchC = nullptr;
    gp.a_kcontig = false; gp.b_kcontig = false;
    gp.wide = true;
    gp.offAm = gp.offBn = gp.offCm = t;
    gp.A = s.pp_cs; gp.B = s.oo_vs; gp.C = s.oo_xs;
    gp.offAk = gp.offBk = u + 2 * nm; gp.offCn = u + 2 * nm + 2 * kx;
    gp.M = gp.N = (int)ns; gp.K = (int)ks;
    AFESP_HIP(gett_launch(gp, cx.ws, cx.stream));
    if (s.pp_pa) {
        gp.A = s.pp_ca; gp.B = s.oo_va; gp.C = s.oo_xa;
        gp.offAk = gp.offBk = u + 2 * nm + kx; gp.offCn = u + 3 * nm + 2 * kx;
        gp.M = gp.N = (int)na; gp.K = (int)ka;
        AFESP_HIP(gett_launch(gp, cx.ws, cx.stream));
    }
    k_oooo_pair_expand_add(cx, s.I_oooo.d, s.oo_xs, s.pp_pa ? s.oo_xa : nullptr, s.o, ns, na);
}

void ccsd_pp_ladder(Context& cx, CCState& s)
{
    const int64_t O = s.o, V = s.v, np = V * (V + 1) / 2, K2 = V * V, N2 = O * O;
    int64_t b0, b1;
    pp_b_range(s, &b0, &b1);
    const int64_t p0 = b0 * (b0 + 1) / 2, p1 = b1 * (b1 + 1) / 2, q0 = b0 * (b0 - 1) / 2, q1 = b1 * (b1 - 1) / 2;
    GettProblem gp;
    gp.alpha = 1.0; gp.beta = 0.0;
    gp.nbatch = 1; gp.batchA = gp.batchB = gp.batchC = nullptr;
    gp.a_kcontig = true; gp.b_kcontig = false;
    if (!s.pp_sym) {
        gp.A = s.v_vvvv.d; gp.B = s.c.d; gp.C = s.pp;
        gp.offAm = s.pp_tab; gp.offAk = s.pp_tab + np; gp.offBk = s.pp_tab + np + K2; gp.offBn = s.pp_tab + np + 2 * K2;
        gp.offCm = s.pp_tab + np + 2 * K2 + N2; gp.offCn = gp.offBn;
        gp.offAm += p0; gp.offCm += p0;
        gp.M = (int)(p1 - p0); gp.N = (int)N2; gp.K = (int)K2;
        gp.wide = (O % 2 == 0) && (V % 2 == 0) && (np % 2 == 0);
        if (cx.rec) cx.rec->product(gp, V * V * V * V, N2 * K2, N2 * np);
        else if (p1 > p0) AFESP_HIP(gett_launch(gp, cx.ws, cx.stream));
        return;
    }
    const int64_t npa = V * (V - 1) / 2, ks = s.pp_ks, ka = s.pp_ka, ns = s.pp_ns, na = s.pp_na, nm = s.pp_nm, kx = s.pp_lds;
    const int64_t* t = s.pp_tab;
    const int64_t* u = t + s.pp_kn;   // [ lds*m | lda*m | ns*k | na*k | ns*m | na*m ]: nm, nm, kx, kx, nm, nm entries
    if (!s.cs_packed) k_c_sympack(cx, s.pp_cs, s.pp_ca, s.c.d, s.o, s.v, ns, na);   // (a large system's I_oooo has packed them already)
    s.cs_packed = false;
    // The hole-hole ladder 1/2 I_oooo(ijmn) c(mnab) (ccsd.f90:1673) has the symmetry of the pp-ladder and the same pair decomposition
    // over (m,n): hh(ij,ab) = sum_{m<=n} Is(ij,mn) c+(mn,ab) + sum_{m<n} Ia(ij,mn) c-(mn,ab) with Is / Ia = 1/2 (I(ijmn) +- I(ijnm)).  So it
    // rides in the two products below as ns / na more summation steps: c+-(mn, .) behind V+-(ef, .) in every row, Is / Ia behind
    // c+-(., ef) -- 2 x 210 x 20100 x 210 multiply-adds in a launch that is running anyway instead of a product over o^2 x o^2 x v^2
    // (0.29 ms at o = 20, v = 200) and a pass over the residual.
    const bool fold = !cx.rec && !s.sharded && p0 == 0 && p1 == np;
    if (fold) {
        k_rows_append(cx, s.pp_vs, s.pp_lds, ks, s.pp_cs, ns, np);
        if (s.pp_pa) k_rows_append(cx, s.pp_va, s.pp_lda, ka, s.pp_ca, na, npa);
        k_oooo_sympack(cx, s.pp_cs + ns * ks, s.pp_pa ? s.pp_ca + na * ka : nullptr, s.I_oooo.d, s.o, ns, na);
    }
    gp.wide = true;
    gp.offAk = t; gp.offBn = gp.offCn = t;
    gp.A = s.pp_vs; gp.B = s.pp_cs; gp.C = s.pp_ps;
    gp.offAm = u + p0; gp.offBk = u + 2 * nm; gp.offCm = u + 2 * nm + 2 * kx + p0;
    gp.M = (int)(p1 - p0); gp.N = (int)ns; gp.K = (int)(fold ? ks + ns : ks);
    // tuning knobs AFESP_PP_SPLIT (K slices of the two pair products, 0 = the launcher's own choice) and
    // AFESP_PP_TILES="tm,tn,split,tm,tn,split" (tile codes and K slices of the symmetric / the antisymmetric product)
    const int* pt = knobs().pp_tiles;
    if (cx.rec) cx.rec->product(gp, s.pp_lds * np, ns * ks, ns * np);
    else if (p1 > p0) AFESP_HIP(gett_launch(gp, cx.ws, cx.stream, pt[2], pt[0], pt[1]));
    if (s.pp_pa) {
        gp.A = s.pp_va; gp.B = s.pp_ca; gp.C = s.pp_pa;
        gp.offAm = u + nm + q0; gp.offBk = u + 2 * nm + kx; gp.offCm = u + 3 * nm + 2 * kx + q0;
        gp.M = (int)(q1 - q0); gp.N = (int)na; gp.K = (int)(fold ? ka + na : ka);
        if (cx.rec) cx.rec->product(gp, s.pp_lda * npa, na * ka, na * npa);
        else if (q1 > q0) AFESP_HIP(gett_launch(gp, cx.ws, cx.stream, pt[5], pt[3], pt[4]));
    }
    (void)npa;
    k_pp_expand(cx, s.pp, s.pp_ps, s.pp_pa, s.o, s.v, ns, na, p0, p1);
}

// I_ooov_p(j,k,i,a) += sum_ef t2(jk,ef) <ef|ia>  (ccsd.f90:1302-1308) in the pair form of the pp-ladder: t2 has the (anti)symmetry
// of c under j <-> k together with e <-> f, so with t+- = t2(jkef) +- t2(jkfe) and W+- = <ef|ia> +- <fe|ia> the product is
// Ts + Ta for j <= k and Ts - Ta for k < j, two products over pair indices (half the work).  W+- are built at init, the c+- and
// P+- buffers of the ladder hold t+- and the two results (the ladder runs later in the iteration).
void ccsd_ooov_pair_form(Context& cx, CCState& s, double* out, int64_t a0, int64_t a1)
{
    if (!out) { out = s.I_ooov_p.d; a0 = 0; a1 = s.v; }
    if (a1 <= a0) return;
    // rows m = (i,a), a the slow index: the range [a0, a1) of a is the contiguous row range [o a0, o a1)
    const int64_t O = s.o, V = s.v, np = V * (V + 1) / 2, npa = V * (V - 1) / 2, ks = s.pp_ks, ka = s.pp_ka, ns = s.pp_ns, na = s.pp_na,
                  nm = s.pp_nm, M = O * (a1 - a0), m0 = O * a0;
    const int64_t* t = s.pp_tab;
    const int64_t* u = t + s.pp_kn;
    k_c_sympack(cx, s.pp_ts, s.pp_ta, s.t2.d, s.o, s.v, ns, na);   // (buffers of its own: c+- stay packed from I_oooo to the ladder)
    const int64_t kx = s.pp_lds;   // rows of the "ns*k" / "na*k" tables
    GettProblem gp;
    gp.alpha = 1.0; gp.beta = 0.0;
    gp.nbatch = 1; gp.batchA = gp.batchB = gp.batchC = nullptr;
    gp.a_kcontig = true; gp.b_kcontig = false;
    gp.wide = true;
    gp.offAk = t; gp.offBn = gp.offCn = t;
    // few tiles (o v rows x o-pair columns), long K: slice K until the device is full
    auto slices = [&](int64_t n, int64_t k) {
        if (M >= 2048) return 0;   // (the launcher's own score: 256 x 112 / 256 x 96 tiles in as many slices as fill the device)
        const int64_t tiles = ((M + 255) / 256) * ((n + 127) / 128), ksteps = (k + 15) / 16;
        int64_t sp = (256 + tiles - 1) / tiles;
        while (sp > 1 && ksteps / sp < 32) --sp;
        return (int)std::max<int64_t>(1, std::min<int64_t>(sp, 32));
    };
    gp.A = s.ov_ws; gp.B = s.pp_ts; gp.C = s.pp_ps;
    gp.offAm = u + m0; gp.offBk = u + 2 * nm; gp.offCm = u + 2 * nm + 2 * kx + m0;
    gp.M = (int)M; gp.N = (int)ns; gp.K = (int)ks;
    if (cx.rec) cx.rec->product(gp, s.pp_lds * O * V, ns * ks, ns * O * V);
    else AFESP_HIP(gett_launch(gp, cx.ws, cx.stream, slices(ns, ks)));
    if (s.ov_wa) {
        gp.A = s.ov_wa; gp.B = s.pp_ta; gp.C = s.pp_pa;
        gp.offAm = u + nm + m0; gp.offBk = u + 2 * nm + kx; gp.offCm = u + 3 * nm + 2 * kx + m0;
        gp.M = (int)M; gp.N = (int)na; gp.K = (int)ka;
        if (cx.rec) cx.rec->product(gp, s.pp_lda * O * V, na * ka, na * O * V);
        else AFESP_HIP(gett_launch(gp, cx.ws, cx.stream, slices(na, ka)));
    }
    (void)np; (void)npa;
    k_pair_expand_add(cx, out + O * O * m0, s.pp_ps + ns * m0, s.ov_wa ? s.pp_pa + na * m0 : nullptr, s.o, M, ns, na);
    // The T1 equation's asym(m,i,e,f) <ef|ma> (src/ccsd.f90:1569-1631) is a trace of the product just formed: with X(j,k,i',a) =
    // sum_ef t2(jkef) <ef|i'a> it is sum_m [2 X(m,i,m,a) - X(i,m,m,a)] -- o^2 v sums of o terms out of the two results, instead of a
    // pass over the o v^3 integrals (0.34 ms at o = 20, v = 200) and a re-laid-out copy of asym_t2 (0.07).  Whole products only.
    if (out == s.I_ooov_p.d && M == O * V && !cx.rec) {
        k_ooov_r1_trace(cx, s.r1x, s.pp_ps, s.ov_wa ? s.pp_pa : nullptr, s.o, s.v, ns, na);
        s.r1x_valid = true;
    }
}

void ccsd_amplitudes(Context& cx, CCState& s, bool defer_update)
{
    auto C = [&](double al, const Tensor& A, const char* la, const Tensor& B, const char* lb, double be, const Tensor& Cc,
                 const char* lc) { contract(cx, al, A, la, B, lb, be, Cc, lc); };
    // lanes (small systems only, see ccsd_intermediates): T1 in two groups, the pp-ladder on its own lane, the other T2 terms
    // in three groups; all but the first group of each go into partial buffers that are added after the join
    ccsd_refresh_sharding(cx, s);
    const bool par = lanes_pay(s) && !cx.rec, sh = s.sharded;
    const bool ring = !par && !cx.rec && !sh && ring_live(s);   // the ring terms through the buffers ccsd_intermediates left (ring.hip)
    s.partials_live = par;
    ring_res_clear(s);
    int64_t v0, v1;
    slice_bounds(s, &v0, &v1);
    auto sl = [&](const Tensor& t, int axis) { return slice_axis(t, axis, v0, v1); };
    auto lane = [&](int i) { if (par) cx.use_lane(i); };
    Tensor r2b = s.r2, r2c = s.r2, r1b = s.r1;
    Tensor r2s = s.r2;              // receives the three ring products: the rank-partial buffer when the iteration is split
    if (sh) {
        r2s.d = s.r2_sh;
        const int64_t np = (int64_t)s.v * (s.v + 1) / 2;
        AFESP_HIP(hipMemsetAsync(s.pp, 0, sizeof(double) * ((int64_t)s.o * s.o * np + s.r2.size() + s.r1.size()), cx.stream));   // [PP | r2_sh | r1_sh]
    }
    if (par) {
        r2b.d = cx.scratch("r2_lane2", s.r2.size());
        r2c.d = cx.scratch("r2_lane3", s.r2.size());
        r1b.d = cx.scratch("r1_lane5", s.r1.size());
        cx.fork(6);
    }
    lane(0);
    // ---- T1, Eq. 43                                                    ccsd.f90:1569-1631
    // (split iteration: the two terms that read a sliced quantity -- I_vv, <ef|ma> -- go per slice of a into r1_sh, which rides behind
    // [PP | r2_sh] in the one exchange)
    if (!sh) C(1.0, s.t1, "ie", s.I_vv, "ea", 0.0, s.r1, "ia");
    C(-1.0, s.I_oo_p, "im", s.t1, "ma", sh ? 0.0 : 1.0, s.r1, "ia");
    C(1.0, s.asym, "miea", s.I_vo, "em", 1.0, s.r1, "ia");
    C(2.0, s.v_oovv, "miea", s.t1, "me", 1.0, s.r1, "ia");
    lane(5);
    if (par && cx.test_throw == 1) {   // test hook (afesp_test_inject): a failure in the middle of the laned update
        cx.test_throw = 0;
        throw Error(99, "injected failure (afesp_test_inject)");
    }
    C(-1.0, s.v_ovov, "maie", s.t1, "me", par ? 0.0 : 1.0, r1b, "ia");
    C(-1.0, s.v_oovo, "mien", s.asym, "mnea", 1.0, r1b, "ia");
    if (!sh && s.r1x_valid && !par && !cx.rec) k_axpby(cx, r1b.d, 1.0, s.r1x, 1.0, r1b.size());   // (formed with the t2 <ef|ia> product: ccsd_ooov_pair_form)
    else if (!sh) C(1.0, s.asym, "mief", s.v_vvov, "efma", 1.0, r1b, "ia");
    lane(1);
    // ---- T2, Eq. 44                                                    ccsd.f90:1637-1716
    // (large-system path: the streamed product over <ab|ej> opens the residual instead of accumulating into it -- an accumulating
    // launch of tall_kernel fetches the old values at every tile's end)
    const bool open_with_vvov = !par && !cx.rec && !sh;   // (split iteration: that product goes per slice of b into the partial residual, below)
    // ... and that product is x_voov again: <ba|je> = (bj|ae) = (bj|ea) = <be|ja> (real orbitals), so sum_e t(i,e) <ba|je> = x_voov(b,i,j,a),
    // which ccsd_intermediates formed from the same t1 -- a transposing copy of o^2 v^2 elements instead of a third pass over the
    // o v^3 integrals (0.32 -> 0.08 ms at o = 20, v = 200)
    // (ring.hip: two of the three ring terms open the residual -- the LDS-DMA GEMM stores, it does not accumulate -- and everything else adds to them)
    if (ring) ring_tg_residual(cx, s);                                              // :1680-1695 ring terms, all three
    // (amplitudes replaced from outside since the intermediates were formed: x_voov carries the OLD t1, the reference's update the new one)
    if (open_with_vvov && !s.amps_touched) permute_add(cx, 1.0, s.x_voov, "bija", ring ? 1.0 : 0.0, s.r2, "ijab");   // :1700, bare part: t(i,e) <ab|ej>
    else if (open_with_vvov) C(1.0, s.t1, "ie", s.v_vvov, "baje", ring ? 1.0 : 0.0, s.r2, "ijab");
    // (split iteration: every term of the T2 residual is evaluated for this rank's slice of b -- or of a, where that is the index the
    // sliced intermediate carries -- into the zeroed partial residual: the replicated residual has no terms of its own)
    if (!sh) {
        C(1.0, s.t2, "ijae", s.I_vv, "eb", open_with_vvov ? 1.0 : 0.0, s.r2, "ijab");   // :1647
        C(-1.0, s.t2, "miba", s.I_oo, "jm", 1.0, s.r2, "ijab");                // :1654-1664
    }
    lane(4);
    ccsd_pp_ladder(cx, s);                                                 // :1669  particle-particle ladder
    lane(1);
    // :1673  hole-hole ladder (the pair form of the pp-ladder carries it along: ccsd_pp_ladder)
    if (!sh && !(s.pp_sym && !cx.rec)) C(0.5, s.I_oooo, "ijmn", s.c, "mnab", 1.0, s.r2, "ijab");
    lane(2);
    if (ring) {
    } else if (!sh) {
        C(-1.0, s.t2, "mjae", s.I_ovov, "iemb", par ? 0.0 : 1.0, r2b, "ijab");   // :1680-1695 ring terms
        C(-1.0, s.I_ovov, "iema", s.t2, "mjeb", 1.0, r2b, "ijab");
    } else if (v1 > v0) {           // the slice of I_ovov / I_voov this rank built, into the zeroed partial residual
        C(-1.0, s.t2, "mjae", sl(s.I_ovov, 3), "iemb", 1.0, sl(r2s, 3), "ijab");
        C(-1.0, sl(s.I_ovov, 3), "iema", s.t2, "mjeb", 1.0, sl(r2s, 2), "ijab");
        C(1.0, s.asym, "miea", sl(s.I_voov, 3), "ejmb", 1.0, sl(r2s, 3), "ijab");
        // -t(m,a) [t2 <ef|mb>](i,j,m,b) for this rank's b (the part of z_ooov that ccsd_intermediates built per slice)
        Tensor osh = view(cx.scratch("ooov_sh", s.I_ooov_p.size()), {(int64_t)s.o, (int64_t)s.o, (int64_t)s.o, (int64_t)s.v});
        C(-1.0, s.t1, "ma", sl(osh, 3), "ijmb", 1.0, sl(r2s, 3), "ijab");
        C(1.0, s.t1, "ie", slice_axis(s.v_vvov, 0, v0, v1), "baje", 1.0, sl(r2s, 3), "ijab");   // :1700, bare part, this rank's b
        C(1.0, s.t2, "ijae", slice_axis(s.I_vv, 1, v0, v1), "eb", 1.0, sl(r2s, 3), "ijab");       // :1647
        C(-1.0, slice_axis(s.t2, 2, v0, v1), "miba", s.I_oo, "jm", 1.0, sl(r2s, 3), "ijab");      // :1654-1664
        C(0.5, s.I_oooo, "ijmn", sl(s.c, 3), "mnab", 1.0, sl(r2s, 3), "ijab");                    // :1673
        C(-1.0, s.t1, "ma", sl(s.z_ooov, 3), "ijmb", 1.0, sl(r2s, 3), "ijab");                    // :1705-1715
        Tensor r1s = s.r1;
        r1s.d = s.r1_sh;
        C(1.0, s.t1, "ie", slice_axis(s.I_vv, 1, v0, v1), "ea", 1.0, slice_axis(r1s, 1, v0, v1), "ia");
        C(1.0, s.asym, "mief", sl(s.v_vvov, 3), "efma", 1.0, slice_axis(r1s, 1, v0, v1), "ia");
    }
    lane(3);
    if (!sh && !ring) C(1.0, s.asym, "miea", s.I_voov, "ejmb", par ? 0.0 : 1.0, r2c, "ijab");
    if (!open_with_vvov && !sh) C(1.0, s.t1, "ie", s.v_vvov, "baje", 1.0, r2c, "ijab");   // :1700, bare part: t(i,e) <ab|ej>
    if (!sh) C(-1.0, s.t1, "ma", s.z_ooov, "ijmb", 1.0, r2c, "ijab");      // :1705-1715 and the t1-dressed parts of :1700 (ccsd_intermediates)
    if (par) cx.join();   // (the partial residuals r1b, r2b, r2c are added up by the update kernel below)
    if (sh) {
        // the one exchange of a split iteration: sum over ranks of [PP | r2_sh] (64 + 128 MB at o = 20, v = 200), in place
        const int64_t np = (int64_t)s.v * (s.v + 1) / 2;
        comm_allreduce_dev(cx, cx.comm, s.pp, (int64_t)s.o * s.o * np + s.r2.size() + s.r1.size());
        k_copy(cx, s.r2.d, s.r2_sh, s.r2.size());   // (the residual the update kernel and afesp_ccsd_get_tensor read)
        k_axpby(cx, s.r1.d, 1.0, s.r1_sh, 1.0, s.r1.size());
    }
    if (defer_update) return;
    // P(ia/jb), + v_oovv, Jacobi divide                                  ccsd.f90:1720-1728
    k_t2_update(cx, s.t2.d, s.r2.d, par ? r2b.d : nullptr, par ? r2c.d : nullptr, s.v_oovv.d, s.D2.d, s.pp, s.o, s.v, s.t1.d, s.r1.d,
                par ? r1b.d : nullptr, s.D1.d, ring ? ring_Y(s) : nullptr);
}

// The intermediate of ccsd.f90:1255-1272 as a tensor (tests / afesp_ccsd_get_tensor); the iteration never forms it.
void ccsd_build_I_vovv_p(Context& cx, CCState& s, const Tensor& out)
{
    permute_add(cx, 1.0, s.v_vvov, "baic", 0.0, out, "ciab");
    contract(cx, -1.0, s.v_oovv, "micb", s.t1, "ma", 1.0, out, "ciab");
    contract(cx, -1.0, s.v_ovov, "maic", s.t1, "mb", 1.0, out, "ciab");
}

// The tail of a launch-fused iteration (kernels.hip, cc_tail_kernel / cc_finalize_kernel): P(ia/jb) and the Jacobi division
// (ccsd.f90:1720-1728), the energy and rms sums (:1764-1782, :1803-1806) and -- speculatively, the caller decides afterwards -- the
// history push and the solve of update_diis_cc (:633-666) for the slot the next update would use.
void ccsd_tail_launch(Context& cx, CCState& s)
{
    auto launch = [](Context& c, CCState* st) {
        CCTail a;
        a.t2 = st->t2.d; a.t1 = st->t1.d; a.r2 = st->r2.d; a.r1 = st->r1.d; a.voovv = st->v_oovv.d; a.D2 = st->D2.d; a.D1 = st->D1.d;
        a.pp = st->pp; a.t2_old = st->t2_old.d; a.o = st->o; a.v = st->v;
        a.r2y = ring_res_live(*st) ? ring_Y(*st) : nullptr;   // (large systems: one ring term lies in a buffer of its own, ring.hip)
        a.half_hist = st->hist_plain == 0;
        if (st->hist_plain > 0) --st->hist_plain;
        a.nerr = st->nerr;
        a.ny = 0; a.slot = 0;
        a.ht = a.he = nullptr; a.amp_s = a.hist_e = nullptr; a.stride = st->nvec;
        a.coef = st->coef; a.bmat = st->bmat;
        if (st->nerr >= 2) {   // ccsd.f90:633-646, without touching the counters (diis_update advances them if it is called)
            int it = st->it + 1;
            if (it > st->nerr) it -= st->nerr;
            a.slot = it - 1;
            a.ny = std::min(st->nact + 1, st->nerr);
            a.ht = st->hist_t + (int64_t)a.slot * st->nvec;
            a.he = st->hist_e + (int64_t)a.slot * st->nvec;
            a.amp_s = st->amp_s; a.hist_e = st->hist_e;
        }
        st->tail_pending = st->nerr >= 2;
        st->tail_slot = a.slot; st->tail_n = a.ny;
        a.seq = ++c.res_seq;
        k_cc_tail(c, a);
    };
    if (cx.rec) {
        CCState* st = &s;
        const int64_t n2 = s.t2.size(), n1 = s.t1.size(), np = (int64_t)s.o * s.o * ((int64_t)s.v * (s.v + 1) / 2);
        std::vector<FusedRange> rd = {frange(s.r2.d, n2), frange(s.r1.d, n1), frange(s.v_oovv.d, n2), frange(s.D2.d, n2), frange(s.D1.d, n1),
                                      frange(s.pp, np), frange(s.t2_old.d, n2)};
        std::vector<FusedRange> wr = {frange(s.amp, s.nvec), frange(s.t2_old.d, n2), frange(cx.scal, 64 + 18 * 512)};
        if (s.nerr >= 2) {
            rd.push_back(frange(s.amp_s, s.nvec));
            rd.push_back(frange(s.hist_e, s.nvec * s.nerr));
            wr.push_back(frange(s.hist_t, s.nvec * s.nerr));
            wr.push_back(frange(s.hist_e, s.nvec * s.nerr));
            wr.push_back(frange(s.coef, 32));
            wr.push_back(frange(s.bmat, (int64_t)s.nerr * s.nerr));
        }
        cx.rec->opaque(rd, wr, [launch, st](Context& c) { launch(c, st); }, 2);
        return;
    }
    launch(cx, &s);
}

int ccsd_tail_read(Context& cx, CCState& s, double e_tol, double t_tol)
{
    // the finalize kernel writes [energy, rms, failure, sequence number] into coherent host memory, the number last: poll it for a
    // while (a stream synchronisation costs ~30 us of wake-up latency), then fall back to waiting for the stream
    volatile double* hr = cx.res_host;
    const double want = (double)cx.res_seq;
    bool seen = false;
    for (int spin = 0; spin < 200000; ++spin) {
        if (__atomic_load_n((const int64_t*)&cx.res_host[3], __ATOMIC_ACQUIRE) == *(const int64_t*)&want) { seen = true; break; }
        if ((spin & 1023) == 1023 && hipStreamQuery(cx.stream) != hipErrorNotReady) break;
    }
    if (!seen) {
        AFESP_HIP(hipStreamSynchronize(cx.stream));
        if (hr[3] != want) throw Error(2, "ccsd_tail_read: the iteration's results did not arrive");
    }
    for (int j = 0; j < s.tail_n; ++j)
        for (int i = 0; i < s.tail_n; ++i) s.tail_b[i + 16 * j] = hr[8 + i + 16 * j];
    s.energy_old = s.energy;        // ccsd.f90:1760
    s.energy = hr[0];
    s.rms = hr[1];                  // un-rooted, ccsd.f90:1806
    return (std::sqrt(s.rms) < t_tol && std::fabs(s.energy - s.energy_old) < e_tol) ? 1 : 0;   // ccsd.f90:1805
}

// The energy evaluation in two halves: the launches (part of the replayed iteration, capi.hip) and the host read.
void ccsd_energy_launch(Context& cx, CCState& s)
{
    k_cc_energy(cx, cx.scal, s.v_oovv.d, s.t1.d, s.t2.d, s.t2_old.d, s.o, s.v);
}
int ccsd_energy(Context& cx, CCState& s, double e_tol, double t_tol)
{
    ccsd_energy_launch(cx, s);
    return ccsd_energy_read(cx, s, e_tol, t_tol);
}
int ccsd_energy_read(Context& cx, CCState& s, double e_tol, double t_tol)
{
    double* h = host_scalars(cx, DIIS_FLAG_SLOT + 1);
    diis_check_flag(cx, h);
    s.energy_old = s.energy;        // ccsd.f90:1760
    s.energy = h[0];
    s.rms = h[1];                   // un-rooted, ccsd.f90:1806
    return (std::sqrt(h[1]) < t_tol && std::fabs(s.energy - s.energy_old) < e_tol) ? 1 : 0;   // ccsd.f90:1805
}

// build_cr_ccsd_t_intermediates, src/ccsd.f90:2338-2551.  Data flow as in the reference: t1/t2 are the converged amplitudes,
// I_vo and asym_t2 are what the last update_restricted_intermediates left behind (:2374-2378).  The three terms of
// I_ooov_pp that the reference sums over `e = 1, nocc` although e is a virtual index (:2535) are summed over the first
// min(o,v) virtuals here too -- the bundled CR goldens contain that bound.
void ccsd_cr_intermediates(Context& cx, CCState& s)
{
    if (!s.ready) throw Error(1, "ccsd_cr_intermediates: no CCSD state");
    const int64_t O = s.o, V = s.v;
    auto C = [&](double al, const Tensor& A, const char* la, const Tensor& B, const char* lb, double be, const Tensor& Cc,
                 const char* lc) { contract(cx, al, A, la, B, lb, be, Cc, lc); };
    if (!s.have_cr) {
        s.I_vovv_pp = cx.tensor({V, O, V, V});
        s.I_ooov_pp = cx.tensor({O, O, O, V});
        s.have_cr = true;
    }
    Tensor xvp = view(cx.scratch("cr_xvp", V * V * V * O), {V, V, V, O}), xv = view(cx.scratch("cr_xv", V * V * V * O), {V, V, V, O});
    Tensor xovov_p = view(cx.scratch("cr_a", O * V * O * V), {O, V, O, V}), xvoov_p = view(cx.scratch("cr_b", O * V * O * V), {V, O, O, V});
    Tensor xovov_pp = view(cx.scratch("cr_c", O * V * O * V), {O, V, O, V}), xvoov_pp = view(cx.scratch("cr_d", O * V * O * V), {V, O, O, V});
    Tensor xovoo = view(cx.scratch("cr_e", O * V * O * O), {O, V, O, O});
    // x_vvvo_p(b,c,a,i) = <cb|ia> - 1/2 t(m,a) <mi|bc>  (:2429);   x_vvvo = x_vvvo_p - 1/2 t(m,a) <mi|bc>  (:2465)
    permute_add(cx, 1.0, s.v_vvov, "cbia", 0.0, xvp, "bcai");
    C(-0.5, s.t1, "ma", s.v_oovv, "mibc", 1.0, xvp, "bcai");
    k_copy(cx, xv.d, xvp.d, xv.size());
    C(-0.5, s.t1, "ma", s.v_oovv, "mibc", 1.0, xv, "bcai");
    // x_ovov_p / x_ovov_pp (j,b,i,a)  (:2441, :2489)
    k_copy(cx, xovov_p.d, s.v_ovov.d, xovov_p.size());
    C(-0.5, s.v_oovo, "mibj", s.t1, "ma", 1.0, xovov_p, "jbia");
    C(1.0, s.t1, "je", xvp, "beai", 1.0, xovov_p, "jbia");
    k_copy(cx, xovov_pp.d, s.v_ovov.d, xovov_pp.size());
    C(-1.0, s.v_oovo, "mibj", s.t1, "ma", 1.0, xovov_pp, "jbia");
    C(0.5, s.t1, "je", xv, "beai", 1.0, xovov_pp, "jbia");
    // x_voov_p / x_voov_pp (b,j,i,a)  (:2453, :2501)
    permute_add(cx, 1.0, s.v_oovv, "ijba", 0.0, xvoov_p, "bjia");
    C(-0.5, s.v_oovo, "imbj", s.t1, "ma", 1.0, xvoov_p, "bjia");
    C(1.0, xvp, "ebai", s.t1, "je", 1.0, xvoov_p, "bjia");
    permute_add(cx, 1.0, s.v_oovv, "ijba", 0.0, xvoov_pp, "bjia");
    C(-1.0, s.v_oovo, "imbj", s.t1, "ma", 1.0, xvoov_pp, "bjia");
    C(0.5, xv, "ebai", s.t1, "je", 1.0, xvoov_pp, "bjia");
    // x_ovoo(k,a,i,j) = <ji|ak> + t(k,e) <ij|ea>  (:2477)
    permute_add(cx, 1.0, s.v_oovo, "jiak", 0.0, xovoo, "kaij");
    C(1.0, s.t1, "ke", s.v_oovv, "ijea", 1.0, xovoo, "kaij");
    // I_vovv_pp(c,i,a,b)  (:2513-2520)
    permute_add(cx, 1.0, s.v_vvov, "baic", 0.0, s.I_vovv_pp, "ciab");
    ccsd_need_vvvv(cx, s);
    C(1.0, s.v_vvvv, "ecba", s.t1, "ie", 1.0, s.I_vovv_pp, "ciab");
    C(-1.0, xovov_p, "icma", s.t1, "mb", 1.0, s.I_vovv_pp, "ciab");
    C(-1.0, s.t1, "ma", xvoov_p, "cimb", 1.0, s.I_vovv_pp, "ciab");
    C(-1.0, s.I_vo, "cm", s.t2, "miab", 1.0, s.I_vovv_pp, "ciab");
    C(1.0, s.t2, "mnba", xovoo, "icmn", 1.0, s.I_vovv_pp, "ciab");
    C(1.0, xv, "ceam", s.asym, "imbe", 1.0, s.I_vovv_pp, "ciab");
    C(-1.0, xv, "ecam", s.t2, "mieb", 1.0, s.I_vovv_pp, "ciab");
    C(-1.0, s.t2, "miae", xv, "ecbm", 1.0, s.I_vovv_pp, "ciab");
    // I_ooov_pp(j,k,i,a)  (:2532-2539)
    permute_add(cx, 1.0, s.v_oovo, "kjai", 0.0, s.I_ooov_pp, "jkia");
    C(-1.0, s.v_oooo, "mikj", s.t1, "ma", 1.0, s.I_ooov_pp, "jkia");
    C(1.0, xovov_pp, "jeia", s.t1, "ke", 1.0, s.I_ooov_pp, "jkia");
    C(1.0, s.t1, "je", xvoov_pp, "ekia", 1.0, s.I_ooov_pp, "jkia");
    C(1.0, s.t2, "kjef", xv, "efai", 1.0, s.I_ooov_pp, "jkia");
    const int64_t eb = std::min(O, V);
    auto clip = [](Tensor t, int axis, int64_t n) { t.dim[axis] = n; return t; };
    Tensor xo_e = clip(xovoo, 1, eb), as_e = clip(s.asym, 2, eb), t2_e2 = clip(s.t2, 2, eb), t2_e3 = clip(s.t2, 3, eb);
    C(1.0, xo_e, "jeim", as_e, "mkea", 1.0, s.I_ooov_pp, "jkia");
    C(-1.0, xo_e, "jemi", t2_e2, "mkea", 1.0, s.I_ooov_pp, "jkia");
    C(-1.0, t2_e3, "mjae", xo_e, "kemi", 1.0, s.I_ooov_pp, "jkia");
}

void ccsd_diis_update(Context& cx, CCState& s) { diis_update(cx, s); }

void diis_update(Context& cx, DiisRing& s)
{
    if (s.nerr < 2) return;
    if (s.tail_pending) {
        // the launch-fused tail of this iteration has pushed the history and solved for the coefficients already (ccsd_tail_launch)
        s.tail_pending = false;
        s.it = s.tail_slot + 1;
        s.nact = s.tail_n;
        // [B -1; -1 0] c = (0,...,0,-1) (ccsd.f90:653-666; the reference calls dsysv, linalg.fpp:38-56), Gaussian elimination with
        // partial pivoting on the host: at most 17 x 17
        const int n = s.nact, N = n + 1;
        double A[17][18];
        for (int i = 0; i < N; ++i)
            for (int j = 0; j <= N; ++j)
                A[i][j] = (i < n && j < n) ? s.tail_b[i + 16 * j] : (j == N) ? (i == n ? -1.0 : 0.0) : (i == n && j == n) ? 0.0 : -1.0;
        for (int k = 0; k < N; ++k) {
            int p = k;
            for (int i = k + 1; i < N; ++i)
                if (std::fabs(A[i][k]) > std::fabs(A[p][k])) p = i;
            if (A[p][k] == 0.0) throw Error(4, "ccsd::update_diis_cc: Linear solve failed!");   // ccsd.f90:666
            if (p != k)
                for (int j = 0; j <= N; ++j) std::swap(A[k][j], A[p][j]);
            for (int i = k + 1; i < N; ++i) {
                const double f = A[i][k] / A[k][k];
                for (int j = k; j <= N; ++j) A[i][j] -= f * A[k][j];
            }
        }
        double c[17];
        for (int k = N - 1; k >= 0; --k) {
            double r = A[k][N];
            for (int j = k + 1; j < N; ++j) r -= A[k][j] * c[j];
            c[k] = r / A[k][k];
        }
        k_lincomb_vals(cx, s.amp, s.hist_t, s.nvec, c, n, s.nvec);
        return;
    }
    // ccsd.f90:633-646
    s.it += 1;
    if (s.it > s.nerr) s.it -= s.nerr;
    if (s.nact < s.nerr) s.nact += 1;
    const int slot = s.it - 1, n = s.nact;
    double* ht = s.hist_t + (int64_t)slot * s.nvec;
    double* he = s.hist_e + (int64_t)slot * s.nvec;
    // ccsd.f90:653-673: only row/column `slot` of B changes; history, error vector and its dot products come out of one pass, the
    // solve (which sums that pass's partial sums itself) and the extrapolation follow on the device with no host round trip:
    // three launches queued behind the amplitude update (seven until round 3)
    k_diis_push(cx, ht, he, s.amp, s.amp_s, s.hist_e, s.nvec, n, slot, s.nvec);
    k_diis_solve(cx, s.coef, s.bmat, cx.scal + DIIS_FLAG_SLOT, n, s.nerr, slot);
    k_lincomb(cx, s.amp, s.hist_t, s.nvec, s.coef, n, s.nvec);
}

void diis_check_flag(Context& cx, const double* host_scal)
{
    if (host_scal[DIIS_FLAG_SLOT] != 0.0) {
        AFESP_HIP(hipMemsetAsync(cx.scal + DIIS_FLAG_SLOT, 0, sizeof(double), cx.stream));
        throw Error(4, "ccsd::update_diis_cc: Linear solve failed!");   // ccsd.f90:666
    }
}

}  // namespace afesp
