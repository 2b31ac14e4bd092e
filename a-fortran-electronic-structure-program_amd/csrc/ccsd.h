// ccsd.h -- device-resident state of the spin-free CCSD solver and the (T) correction.
#pragma once
#include "afesp_internal.h"

namespace afesp {

// DIIS ring (ccsd.f90:38-67, shared by the spin-free and the spin-orbital solver): vectors are [t1 ; t2]
// concatenated, length nvec
struct DiisRing {
    int nerr = 0, nact = 0, it = 0;
    int64_t nvec = 0;
    double *amp = nullptr;      // [t1 ; t2] contiguous, t1 = amp, t2 = amp + o*v
    double *amp_s = nullptr;    // amplitudes saved at the top of the iteration (t1_s/t2_s)
    double *hist_t = nullptr, *hist_e = nullptr;   // nerr * nvec each
    double *coef = nullptr;     // device coefficients
    double *bmat = nullptr;     // error overlap matrix on the device (nerr x nerr, full)
    // launch-fused tail (ccsd_tail_launch): the history push and the solve of the NEXT update have already run, for slot / count
    // tail_slot / tail_n; diis_update then only advances the counters and extrapolates.
    bool tail_pending = false;
    int tail_slot = 0, tail_n = 0;
    double tail_b[256];          // the error overlap matrix of that update as the finalize kernel left it in host memory (tail_n x tail_n, ld 16)
};
void diis_alloc(Context& cx, DiisRing& r, int diis_nerr);   // r.nvec and r.amp set by the caller (init_diis_cc_t, :577-615)
void diis_save(Context& cx, DiisRing& r);                   // ccsd.f90:342-343
void diis_update(Context& cx, DiisRing& r);                 // update_diis_cc, ccsd.f90:617-676

struct CCState : DiisRing {
    int o = 0, v = 0;
    bool ready = false;
    double* e = nullptr;   // orbital energies on device, length o+v
    // integral slices (ccsd.f90:507-512) and their immutable "2x - x^T" companions
    Tensor v_oovv, v_ovov, v_vvov, v_oovo, v_oooo, v_vvvv, w_oovv, w_vvov, w_oovo;
    Tensor D1, D2, t1, t2, t2_old, r1, r2;
    double* pp = nullptr;          // packed particle-particle ladder PP(i,j,p), p over a <= b; the sharded ring terms' partial
                                   // residual r2_sh (o^2 v^2) sits right behind it: one all-reduce covers both
    double* r2_sh = nullptr;
    double* r1_sh = nullptr;       // ... and the rank-partial r1 (o v) behind that
    // rank split of the iteration's large products (afesp_comm_init with world > 1; ccsd_refresh_sharding)
    bool sharded = false;
    int sh_rank = 0, sh_world = 1;
    int64_t* pp_tab = nullptr;     // offset tables of the pp-ladder GEMM: [Am | k | Bk | n]
    // symmetric / antisymmetric form (large systems): V+-(ef,ab) built at init, c+-(ij,ef) and the two products per iteration
    bool pp_sym = false;
    double *pp_vs = nullptr, *pp_va = nullptr, *pp_cs = nullptr, *pp_ca = nullptr, *pp_ps = nullptr, *pp_pa = nullptr;
    int64_t pp_ks = 0, pp_ka = 0, pp_ns = 0, pp_na = 0, pp_kn = 0;   // even leading dimensions of those operands
    // rows of V+- / W+- are [ V(ef, .) | c+-(mn, .) ]: the hole-hole ladder (ccsd.f90:1673) rides in the pp-ladder's products as ns / na
    // more summation steps (ccsd_pp_ladder); pp_lds / pp_lda = ks + ns / ka + na are the row strides
    int64_t pp_lds = 0, pp_lda = 0;
    double *pp_ts = nullptr, *pp_ta = nullptr;      // t2+-(jk,ef) of the I_ooov_p product (c+- stay packed from I_oooo to the ladder)
    double *oo_vs = nullptr, *oo_va = nullptr;      // 1/2 (<ij|ef> +- <ij|fe>) over pairs (frozen): I_oooo's c <ij|ef> term in pair form
    double *oo_xs = nullptr, *oo_xa = nullptr;      // its two results, (kl) x (ij) pairs
    bool cs_packed = false;                         // pp_cs / pp_ca hold c+- of the current amplitudes (ccsd_intermediates)
    double* r1x = nullptr;                          // asym(m,i,e,f) <ef|ma> as a trace of the pair-form t2 <ef|ia> product (ccsd_ooov_pair_form)
    bool r1x_valid = false;
    // iterations for which the DIIS history may still hold an error vector without the amplitudes' symmetry e(i,j,a,b) = e(j,i,b,a):
    // afesp_ccsd_set_amplitudes accepts any t2, and the error vector of the iteration that starts from it lives for nerr iterations.
    // While > 0 the large-system tail sums the DIIS overlaps over every element instead of a <= b (kernels.hip, cc_tail_kernel).
    int hist_plain = 0;
    bool amps_touched = true;                       // t1 / t2 were replaced from outside since the last ccsd_intermediates (afesp_ccsd_set_amplitudes)
    int64_t pp_nm = 0;                                     // rows the row tables cover: max(v(v+1)/2, o v)
    double *ov_ws = nullptr, *ov_wa = nullptr;             // the same split of <ef|ia> (v_vvov) for I_ooov_p, built at init
    Tensor I_vo, I_vv, I_oo_p, I_oo, c, asym, x_voov, I_oooo, I_ovov, I_voov, I_ooov_p;
    Tensor z_ooov;                 // I_ooov_p plus the t1-dressed pieces that stand in for I_vovv_p (ccsd_intermediates)
    double energy = 0.0, energy_old = 0.0, rms = 0.0;
    void* tplan = nullptr;      // cached (T) launch plan (triples.hip)
    Tensor I_vovv_pp, I_ooov_pp;   // completely renormalised moments (ccsd.f90:2338-2551), built on request
    // v_vvvv is only formed at init when the plain ladder reads it (small systems); otherwise on request (ccsd_need_vvvv) from
    // the packed MO integrals: eri_src points at them (the context's resident array, or eri_own when the host handed them in and
    // the state keeps its device copy); NULL once they have been replaced
    const double* eri_src = nullptr;
    double* eri_own = nullptr;
    bool have_cr = false;
    // bumped by every entry point that may change t1 / t2 (resp. the CR intermediates): what is derived from them -- the (T)
    // operand copies, triples.hip -- is rebuilt only when these have moved on
    int64_t amp_epoch = 0, cr_epoch = 0;
    void* ring = nullptr;       // the ring products' operands and descriptors on the LDS-DMA GEMM (ring.hip), large systems
    bool partials_live = false; // the last amplitudes call left terms of r2 / r1 in the laned partial buffers (afesp_ccsd_get_tensor)
    int64_t frozen_id = 0;   // what ccsd_init stamped the immutable integral slices with (contract() keeps re-laid-out copies of those)
};
void triples_plan_free(CCState& s);

// ring.hip: the six o^3 v^3 ring products of a large system's iteration as two launches of the LDS-DMA GEMM
bool ring_tg_applies(const CCState& s);
bool ring_tg_pack(Context& cx, CCState& s);            // asym_t2, c and the amplitudes' [K | row] copies in one pass (false: run k_asym_c)
void ring_tg_intermediates(Context& cx, CCState& s);   // I_ovov' / I_voov' from the small terms left in I_ovov / I_voov
void ring_tg_residual(Context& cx, CCState& s);        // the three ring terms: two OPEN r2 (i,j,a,b), one into ring_Y (j,i,a,b)
void ring_tg_materialize(Context& cx, CCState& s, const Tensor& I_ovov_out, const Tensor& I_voov_out);   // the reference's layout (tests)
bool ring_live(const CCState& s);        // the intermediates of the current iteration live in the ring buffers
bool ring_res_live(const CCState& s);    // ... and a ring term of the current residual in ring_Y
void ring_res_clear(CCState& s);
void ring_invalidate(CCState& s);
void ring_reinit(CCState& s);            // the state is initialised again where it lies: the integrals' copies follow
const double* ring_Y(const CCState& s);
void ring_free(Context& cx, CCState& s);

// eri_mo_dev: packed chemist MO integrals ON DEVICE (length neri(o+v)); e_host: orbital energies (host)
void ccsd_init(Context& cx, CCState& s, int o, int v, const double* eri_mo_dev, const double* e_host, int diis_nerr);
// whether ccsd_init would initialise the state again where it lies (same extents: addresses, plans and compiled programs stay valid)
bool ccsd_can_reinit(const CCState& s, int o, int v, int diis_nerr);
void ccsd_need_vvvv(Context& cx, CCState& s);          // forms <ef|ab> (v^4) if the state does not hold it yet
void ccsd_refresh_sharding(Context& cx, CCState& s);   // call before an iteration: picks up the context's communicator
bool ccsd_uses_lanes(const CCState& s);   // small systems: the iteration's chains run on parallel lanes (ccsd.hip)
void ccsd_diis_save(Context& cx, CCState& s);
void ccsd_intermediates(Context& cx, CCState& s, bool save_for_diis = false);   // save_for_diis: ccsd_diis_save rides along
void ccsd_amplitudes(Context& cx, CCState& s, bool defer_update = false);   // defer_update: everything but the final division (ccsd_tail_launch does it)
// amplitude update + energy / rms sums + DIIS push and solve in two launches, results into pinned host memory (small systems);
// ccsd_tail_read waits for them (polling, no stream synchronisation) and applies the convergence rule
void ccsd_tail_launch(Context& cx, CCState& s);
int ccsd_tail_read(Context& cx, CCState& s, double e_tol, double t_tol);
void ccsd_pp_ladder(Context& cx, CCState& s);
// out = nullptr: into I_ooov_p, all rows; otherwise the rows (i, a) with a in [a0, a1) into out(j,k,i,a) (the split iteration's slice)
void ccsd_ooov_pair_form(Context& cx, CCState& s, double* out = nullptr, int64_t a0 = 0, int64_t a1 = 0);
void ccsd_build_I_vovv_p(Context& cx, CCState& s, const Tensor& out);   // out(c,i,a,b), dense v x o x v x v
bool pp_sym_pays(int64_t o, int64_t v);   // whether ccsd_init chooses the split form (AFESP_PP_SYM=0/1 overrides)
// updates s.energy / s.energy_old / s.rms (un-rooted, as ccsd.f90:1806); returns 1 if converged
int ccsd_energy(Context& cx, CCState& s, double e_tol, double t_tol);
void ccsd_energy_launch(Context& cx, CCState& s);                               // the two kernels only
int ccsd_energy_read(Context& cx, CCState& s, double e_tol, double t_tol);        // host read of what they left
void ccsd_diis_update(Context& cx, CCState& s);
void ccsd_free(Context& cx, CCState& s);

// (T): out[0]=E[T] out[1]=E(T) out[2]=D[T] out[3]=D(T) contributions of the unordered triples with
// flat index in [t_begin, t_end) of the i<=j<=k enumeration; D base term (ccsd.f90:2243) added iff t_begin==0.
int64_t triples_count(int o);
// cost-balanced shard boundaries of the flat triple list: bounds[0..world]
void triples_shard_bounds(int o, int v, bool cr, int world, int64_t* bounds);
// occupied block size of the flat triple order on this device (ranks must agree on it)
int triples_block_size(int o, int v, bool cr);
hipError_t triples_read_orbit_stamps(unsigned long long* out, int n);   // diagnostic builds only (triples_orbit.h)
// cr = true: also out[4] = sum t_bar.M3, out[5] = out[4] + sum z_bar.M3 (needs ccsd_cr_intermediates)
// want_d = false: only out[0], out[1] (what plain CCSD(T)/[T] need; the reference skips y and the D sums there too)
void ccsd_triples(Context& cx, CCState& s, int64_t t_begin, int64_t t_end, double* out_host, bool cr = false, bool want_d = true);
// build_cr_ccsd_t_intermediates (ccsd.f90:2338-2551) on the converged amplitudes
void ccsd_cr_intermediates(Context& cx, CCState& s);

}  // namespace afesp
