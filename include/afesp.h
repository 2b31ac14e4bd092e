/*
 * afesp.h -- C-ABI of libafesp_hip.so: the MI355X (gfx950) coupled-cluster engine behind the AFESP hot path.
 *
 * The reference (brianz98/A-Fortran-Electronic-Structure-Program) has no FFI: the path is entered by three
 * use-associated calls in src/main.F90:98,105,112.  This header is the boundary a Fortran host binds with
 * ISO_C_BINDING (interface block + binding in INTEGRATION.md): flat column-major fp64 arrays plus extents, exactly
 * what the reference already did for its own accelerator variant (do_ccsd_t_spinorb_acc, src/ccsd.f90:1924-1938).
 *
 * Conventions
 *   - every array is Fortran column-major, fp64, index order as declared in the reference
 *     (t1(o,v), t2(o,o,v,v): first index fastest; virtual indices 1..v with the o offset removed, src/ccsd.f90:429,438)
 *   - packed ERI arrays use the reference's 8-fold order (src/integrals.f90:187-210): ij = i(i-1)/2+j (i>=j, 1-based),
 *     ijkl = ij(ij-1)/2+kl (ij>=kl); length npair(npair+1)/2, npair = n(n+1)/2
 *   - canon_coeff is (MO, AO): row = MO, column = AO (src/hf.f90:102,127)
 *   - host pointers are borrowed for the duration of the call; device memory is owned by the context
 *   - every function returns 0 on success; non-zero -> afesp_last_error(ctx) (the Fortran host maps it to error(),
 *     src/error_handling.f90:7-20).  Without a usable GPU afesp_ctx_create fails: there is no CPU fallback.
 *   - extents are int64_t / int (the reference's int32 packed-index limit n<=99, src/integrals.f90:21, is lifted)
 *
 * Environment variables.  The library needs none.  Every AFESP_* variable it understands is defined, with its default, in ONE
 * place -- csrc/knobs.h -- and follows the environment at the granularity of one call of this header (re-read at the top of every
 * entry point; never in the middle of one).  Three kinds:
 *   TEST-ONLY path selectors (they choose which kernels evaluate a quantity; results agree to ~1e-13 but summation orders differ --
 *     tests use them to send a small system down a large system's path or to hold two forms of one product against each other; not
 *     for production use):  AFESP_SMALL_MAX, AFESP_NO_LANES, AFESP_FUSED, AFESP_FUSED_LANES, AFESP_PP_SYM, AFESP_RING_TG,
 *     AFESP_RING_TG_MIN, AFESP_RING_PACK, AFESP_LARGE_TAIL, AFESP_TALL, AFESP_TALL_MIN, AFESP_TALL_DUAL, AFESP_GETT_SK, AFESP_T_GEMM,
 *     AFESP_T_ONE_POOL, AFESP_CC_REINIT, AFESP_CC_SHARD, AFESP_CC_TIME_SLICE, AFESP_AO2MO_TG, AFESP_AO2MO_PAIR, AFESP_AO2MO_MIXED, AFESP_AO2MO_PAD,
 *     AFESP_AO2MO_BLOCKED, AFESP_MP2_PACKED, AFESP_NO_GRAPH, AFESP_GRAPH_AFTER, AFESP_NO_PRELOAD, AFESP_PRELOAD_LANES,
 *     AFESP_PRELOAD_GETT, AFESP_PLAN_VERIFY
 *   tuning (tile / slice / pool sizes, scheduling):  AFESP_PP_SPLIT, AFESP_PP_TILES, AFESP_REPACK_MIN, AFESP_PLAN_DEVICE_FROM,
 *     AFESP_FUSED_BIG_FLOP, AFESP_FUSED_ITEMS, AFESP_FUSED_MIN_STEPS, AFESP_FUSED_MAX_MFMA, AFESP_FUSED_NB, AFESP_SPLIT_BELOW,
 *     AFESP_SPLIT_MIN_STEPS, AFESP_TG_PATCH, AFESP_TG_GRID, AFESP_TG_PRIO_SHIFT, AFESP_TG_DYNAMIC, AFESP_T_BLOCK, AFESP_T_POOL_GIB,
 *     AFESP_T_SPLIT_TILES
 *   diagnostics (printing, measurement; no effect on results):  AFESP_TG_DBG, AFESP_GRAPH_DEBUG, AFESP_PRELOAD_DEBUG,
 *     AFESP_FUSED_DEBUG, AFESP_FUSED_PER_OP, AFESP_GETT_DEBUG, AFESP_T_DEBUG, AFESP_CONTRACT_TRACE, AFESP_STAMPS_GROUPED
 */
#ifndef AFESP_H
#define AFESP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct afesp_ctx afesp_ctx;

/* Context = one GPU (HIP device `device`), one stream, resident tensors.  One context per process per GPU. */
int afesp_ctx_create(int device, afesp_ctx** out);
void afesp_ctx_destroy(afesp_ctx* ctx);
const char* afesp_last_error(const afesp_ctx* ctx);
int afesp_version(void);
int64_t afesp_neri(int64_t nbasis); /* packed length, src/integrals.f90:175-176 */

/* Replaces `call do_mp2_spatial(sys, int_store)` (src/main.F90:98, src/mp2.f90:261-449).
 *   in : nbasis n, nocc o, canon_coeff[n*n] (MO,AO), canon_levels[n], eri_packed[neri] (AO basis; may be NULL after
 *        afesp_read_eri_text -- the AO integrals are then already on the device)
 *   out: eri_mo_packed[neri] (may be NULL: the MO integrals then stay on the device only), *e_mp2
 * The transformed integrals stay resident in the context for afesp_ccsd_init(..., eri_mo_packed = NULL). */
int afesp_ao2mo_mp2(afesp_ctx* ctx, int64_t nbasis, int64_t nocc, const double* canon_coeff, const double* canon_levels,
                    const double* eri_packed, double* eri_mo_packed, double* e_mp2);

/* Replaces init_cc + init_diis_cc_t (src/ccsd.f90:313-316, :404-615).
 *   eri_mo_packed: packed MO integrals from the host, or NULL to use the ones afesp_ao2mo_mp2 left on the device.
 *   diis_n_errmat: sys%ccsd_diis_n_errmat (<2 switches DIIS off, src/ccsd.f90:593-595). */
int afesp_ccsd_init(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, const double* eri_mo_packed, const double* canon_levels,
                    int diis_n_errmat);

/* One pass of the loop body src/ccsd.f90:340-360: save t for DIIS, update_restricted_intermediates,
 * update_amplitudes_restricted, update_cc_energy.  *rms_sq is the UN-rooted sum of (dT2)^2 the reference stores and
 * prints (src/ccsd.f90:1806); *converged follows src/ccsd.f90:1805. */
int afesp_ccsd_iterate(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged);
/* update_cc_energy alone on the current amplitudes (the "MP1" line, src/ccsd.f90:325). */
int afesp_ccsd_energy(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged);
/* update_diis_cc (src/ccsd.f90:395, :617-676). */
int afesp_ccsd_diis(afesp_ctx* ctx);
/* The whole solver loop src/ccsd.f90:325-396.  iter_energy / iter_rms_sq (length maxiter+1, may be NULL) receive the
 * iteration table incl. entry 0 = "MP1".  *niter = iterations taken, or -1 if not converged within maxiter. */
int afesp_ccsd_solve(afesp_ctx* ctx, int maxiter, double e_tol, double t_tol, double* iter_energy, double* iter_rms_sq,
                     int* niter);
/* Converged amplitudes (what move_alloc hands to int_store_cc, src/ccsd.f90:386-387). */
int afesp_ccsd_get_amplitudes(afesp_ctx* ctx, double* t1, double* t2);
/* t2 must carry the symmetry of closed-shell amplitudes, t2(i,j,a,b) = t2(j,i,b,a) -- every set the solver itself produces does, to the
 * bit; the residual is formed as r2 + its image and the pp-ladder over pair indices.  (A large system sums its DIIS overlaps over a <= b
 * only; for the nerr iterations in which the error vector of a handed-in set is part of the history it sums every element, so a set that
 * is symmetric up to rounding costs nothing in accuracy.) */
int afesp_ccsd_set_amplitudes(afesp_ctx* ctx, const double* t1, const double* t2);
/* Named device tensor -> host (tests / debugging).  Names: v_oovv v_ovov v_vvov v_oovo v_oooo v_vvvv I_vo I_vv I_oo_p
 * I_oo c_oovv asym_t2 x_voov I_oooo I_ovov I_voov I_vovv_p I_ooov_p r1 r2 D1 D2 t1 t2
 * (v_vvvv and I_vovv_p are not kept by a large system's iteration and are formed by this call; v_vvvv needs the packed MO
 * integrals the solver was initialised from: status 1 if afesp_ao2mo_mp2 has replaced them since afesp_ccsd_init;
 * r2 is the T2 residual before P(ia/jb) up to terms held as their images under (i <-> j, a <-> b): r2(ijab) + r2(jiba) is what
 * equals the same sum of the reference's tmp_t2, src/ccsd.f90:1720-1728) */
int afesp_ccsd_get_tensor(afesp_ctx* ctx, const char* name, double* out, int64_t capacity);
/* intermediates / amplitude equations separately (src/ccsd.f90:350,357), for term-by-term parity tests.  The two calls are ONE update
 * of one set of amplitudes, as in the reference's loop: terms are regrouped between them (the t1-dressed parts of I_vovv_p, the bare
 * t(i,e) <ab|ej> term), so amplitudes replaced in between give a residual that is neither the old nor the new one. */
int afesp_ccsd_update_intermediates(afesp_ctx* ctx);
int afesp_ccsd_update_amplitudes(afesp_ctx* ctx);

/* Replaces `call do_ccsd_t_spatial(...)` (src/main.F90:112, src/ccsd.f90:2018-2293) on the amplitudes resident in ctx.
 * The (i<=j<=k) triples are numbered 0..afesp_ccsd_t_ntriples-1; [t_begin, t_end) selects this rank's shard (the sums
 * of all shards are what one RCCL all-reduce combines; the D base term 1+2|t1|^2+asym.c, src/ccsd.f90:2243, is added by
 * the shard that holds triple 0).
 *   out[0] = E[T]  (src/ccsd.f90:2218-2219)     out[1] = E(T) incl. the z term (:2220 as in R/CR mode = the correct (T))
 *   out[2] = D[T]  (:2230-2231, :2243)          out[3] = D(T) (:2232) */
int64_t afesp_ccsd_t_ntriples(int64_t nocc);
int afesp_ccsd_t(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double out[4]);
/* Shard boundaries for `world` ranks: rank r evaluates [bounds[r], bounds[r+1]) (bounds has world+1 entries, bounds[0] = 0,
 * bounds[world] = afesp_ccsd_t_ntriples).  The reference splits the (i,j,k) loop `collapse(3) schedule(static,10)` over its
 * threads (src/ccsd.f90:2091); here the triples are evaluated block triple by block triple of the occupied index, and a block
 * triple cut by a shard end costs more per triple, so equal counts are not equal times: the boundaries equalise a cost
 * estimate instead.  Identical on every rank; cr != 0 for shards of afesp_ccsd_t_cr (its pool holds two sets of blocks).
 * Any other partition of the list is valid too -- the sums of the shards add up to the whole. */
int afesp_ccsd_t_shard_bounds(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, int cr, int world, int64_t* bounds);
/* The same for the plain CCSD(T)_spatial / CCSD[T]_spatial types, which need neither y nor the D sums (the reference skips
 * them there as well, src/ccsd.f90:2181-2185, :2228-2247): out[0] = E[T], out[1] = E(T).  Cheaper: the z term is evaluated
 * once per element instead of at its six permutations (csrc/triples_orbit.h). */
int afesp_ccsd_t_plain(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double out[2]);

/* Completely renormalised CCSD[T]/(T).  afesp_ccsd_cr_intermediates replaces
 * build_cr_ccsd_t_intermediates (src/ccsd.f90:381, :2338-2551) and must be called on the converged amplitudes, before any
 * further afesp_ccsd_iterate.  afesp_ccsd_t_cr = afesp_ccsd_t plus the generalised-moment sums:
 *   out[4] = sum t_bar.M3 (src/ccsd.f90:2223-2224)   out[5] = out[4] + sum z_bar.M3 (:2225)
 * so that E_CR[T] = out[4]/out[2] and E_CR(T) = out[5]/out[3] (src/ccsd.f90:2268-2272).  Same sharding contract. */
int afesp_ccsd_cr_intermediates(afesp_ctx* ctx);
int afesp_ccsd_t_cr(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double out[6]);

/* Input/output side of the path.
 * afesp_read_eri_text replaces the two-body loop of read_integrals_in (src/integrals.f90:146-161): parses `eri.dat`
 * ("i j a b value", 1-based) straight into the 8-fold packed array and leaves it ON THE DEVICE, so that a following
 * afesp_ao2mo_mp2(..., eri_packed = NULL, ...) transforms without another host pass; eri_packed (may be NULL) also
 * receives the packed host copy the SCF needs, *nread the number of lines.
 * afesp_write_fcidump replaces write_fcidump (src/mp2.f90:451-487) from the MO integrals resident after
 * afesp_ao2mo_mp2: same line format (I3,I3,I3,I3,ES17.9), same 1e-7 threshold, same (header-less) content. */
int afesp_read_eri_text(afesp_ctx* ctx, const char* path, int64_t nbasis, double* eri_packed, int64_t* nread);
/* The same residency from an array the caller already holds (int_store%eri). */
int afesp_set_eri(afesp_ctx* ctx, int64_t nbasis, const double* eri_packed);
/* Replaces build_fock (src/hf.f90:349-385), the O(n^4) step of every SCF iteration, on the resident packed AO
 * integrals: fock(i,j) = core_hamil(i,j) + sum_kl density(k,l) [2 (ij|kl) - (ik|jl)]; n x n column-major host arrays. */
int afesp_build_fock(afesp_ctx* ctx, int64_t nbasis, const double* density, const double* core_hamil, double* fock);
int afesp_write_fcidump(afesp_ctx* ctx, const char* path, int64_t nbasis, int64_t* nwritten);

/* Spin-orbital path: replaces `call do_ccsd_spinorb(sys, int_store, int_store_cc)` (src/main.F90:67,
 * src/ccsd.f90:71-277) and `call do_ccsd_t_spinorb(...)` (src/main.F90:79, src/ccsd.f90:1812-1922).
 * Spin orbitals are interleaved alpha,beta (src/ccsd.f90:108-143); the spin-orbital extents are the reference's
 * (src/geometry.f90:44-45): nocc = nel, nvirt = 2*nbasis - nel.  Arrays: t1(nocc,nvirt), t2(nocc,nocc,nvirt,nvirt).
 *   afesp_ccsd_so_init    = antisymmetrised integrals + slices (:108-207), init_cc(.not.restricted), init_diis_cc_t.
 *                           eri_mo_packed NULL = the MO integrals afesp_ao2mo_mp2 left on the device; canon_levels has
 *                           nbasis entries (spatial).  flags bit 0 (AFESP_SO_FOO_AS_PUBLISHED): put the tau~ term of F_mi
 *                           where Stanton's Eq. 4 has it; by default it lands transposed, as src/ccsd.f90:789-794 codes it
 *                           (the reference's shipped ref_out predates that dgemm and needs the flag to be reproduced).
 *   afesp_ccsd_so_energy  = update_cc_energy, unrestricted branch (:1783-1806); same outputs as afesp_ccsd_energy.
 *   afesp_ccsd_so_iterate = build_tau, build_F, build_W, update_amplitudes, update_cc_energy (:229-251 loop body).
 *   afesp_ccsd_so_diis    = update_diis_cc (:274).
 *   afesp_ccsd_so_t       = E_T of src/ccsd.f90:1910 restricted to the triples i<j<k numbered [t_begin, t_end) of
 *                           afesp_ccsd_so_t_ntriples(nocc) (the summand is antisymmetric in i,j,k; shards add up). */
#define AFESP_SO_FOO_AS_PUBLISHED 1
int afesp_ccsd_so_init(afesp_ctx* ctx, int64_t nbasis, int64_t nel, const double* eri_mo_packed, const double* canon_levels,
                       int diis_n_errmat, int flags);
int afesp_ccsd_so_energy(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged);
int afesp_ccsd_so_iterate(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged);
int afesp_ccsd_so_diis(afesp_ctx* ctx);
int afesp_ccsd_so_get_amplitudes(afesp_ctx* ctx, double* t1, double* t2);
int afesp_ccsd_so_set_amplitudes(afesp_ctx* ctx, const double* t1, const double* t2);
/* name: F_vv F_oo F_ov W_oooo (stored i,j,m,n) W_vvvv (stored e,f,a,b) W_ovvo tau tau_tilde oovv vvvv t1 t2 */
int afesp_ccsd_so_get_tensor(afesp_ctx* ctx, const char* name, double* out, int64_t capacity);
int64_t afesp_ccsd_so_t_ntriples(int64_t nocc);
int afesp_ccsd_so_t(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double* e_t);

/* ---- Multi-GPU: one process per GPU, each with its own context.  The reference has no distributed layer;
 * its (T) loop ends in an OpenMP `reduction(+: ...)` over threads (src/ccsd.f90:2091, entered from src/main.F90:112).  Here
 * every rank evaluates its shard [bounds[r], bounds[r+1]) of the triple list (afesp_ccsd_t_shard_bounds) and that
 * reduction becomes ONE sum over the ranks of the 2...6 partial scalars: afesp_allreduce_sum.
 *   transport AFESP_COMM_RCCL: ncclAllReduce(sum, fp64) on the context's stream (xGMI between the GPUs of a node); librccl is
 *             opened on the first call, a single-rank run never needs it.
 *   transport AFESP_COMM_HOST: ranks of one node add through a file-backed shared segment in a fixed rank order.  For
 *             rehearsing the rank logic where ranks SHARE a GPU (RCCL refuses two ranks on one device); host memory only.
 *   bootstrap_path: a file name in a directory every rank sees, unique to this job (the launcher makes it): rank 0 publishes
 *             the RCCL unique id there / it backs the shared segment; removed once every rank has joined.  With
 *             unique_id != NULL (128 bytes from afesp_comm_unique_id on rank 0, distributed by the caller -- bench.py
 *             broadcasts it through torch.distributed) no file is used.
 * A communicator alone does not change how the CCSD iteration is evaluated: splitting it over the ranks is a separate, opt-in
 * switch (afesp_ccsd_set_split below).  Once that is on, every rank must make the same sequence of afesp_ccsd_* calls. */
#define AFESP_COMM_RCCL 0
#define AFESP_COMM_HOST 1
int afesp_device_count(void);
int afesp_comm_unique_id(char id_out[128]);
int afesp_comm_init(afesp_ctx* ctx, int rank, int world, int transport, const char* bootstrap_path, const char* unique_id);
int afesp_comm_destroy(afesp_ctx* ctx);
/* in-place sum over the ranks of n host doubles (every rank passes the same n) */
int afesp_allreduce_sum(afesp_ctx* ctx, double* inout, int64_t n);
/* The CCSD iteration split over the ranks of the communicator: the o^3 v^3 ring products, the pp-ladder and every other term that
 * carries a virtual index which can be sliced (the whole T2 residual, the <eb|ia> products, I_vv, two T1 terms) are evaluated for
 * the rank's slice of that index; one all-reduce of [PP | partial T2 residual | partial T1 residual] per iteration; amplitudes,
 * DIIS history and energies stay replicated and identical on every rank (replaces nothing in the reference: its iteration is
 * one process, src/ccsd.f90:340-395).  While the split is on, afesp_ccsd_get_tensor returns sliced intermediates as a rank built
 * them (I_vv, I_ovov, I_voov, x_voov: the rank's slice; I_ooov_p without its t2 <ef|ia> and x_voov terms).
 * Opt-in: mode 1 = split, 0 = replicas, -1 = as the environment says (AFESP_CC_SHARD=1 splits; default replicas;
 * AFESP_CC_SHARD=0 keeps replicas whatever mode says).  *split of afesp_ccsd_is_split = what the next iteration will do. */
int afesp_ccsd_set_split(afesp_ctx* ctx, int mode);
int afesp_ccsd_is_split(afesp_ctx* ctx, int* split);
/* Small systems (o^2 v^2 <= 2^20 amplitudes, one rank): the iteration of afesp_ccsd_iterate / afesp_ccsd_solve -- every contraction
 * site of update_restricted_intermediates and update_amplitudes_restricted (src/ccsd.f90:1040-1312, :1538-1732), update_cc_energy
 * (:1764-1806) and the first half of update_diis_cc (:633-663) -- runs as a compiled sequence of ~11 launches, one grouped launch
 * per dependency level (csrc/fused.h), instead of ~75 launches call by call.  mode 1 = on, 0 = off (the call-by-call path on
 * parallel streams, graph-replayed after AFESP_GRAPH_AFTER iterations), -1 = as the environment says (AFESP_FUSED=0 switches it
 * off; default on).  *launches of afesp_ccsd_iteration_launches = kernel launches of one compiled iteration (0: not compiled,
 * or not eligible). */
int afesp_ccsd_set_fused(afesp_ctx* ctx, int mode);
int afesp_ccsd_iteration_launches(afesp_ctx* ctx, int* launches);
/* Occupied block size of the (T) triple enumeration on this rank's device (it depends on the device memory size and on the
 * AFESP_T_POOL_GIB / AFESP_T_SPLIT_TILES environment): ranks whose values differ would enumerate different flat orders, so
 * callers compare it across ranks before sharding (bench.py and els_amd put it into their first all-reduce). */
int afesp_ccsd_t_block_size(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, int cr, int* block_size);

/* The context's device arena (csrc/contract.hip): out = {allocations that reached the driver, requests served from blocks the
 * context had given back, idle bytes, live bytes}. */
int afesp_arena_stats(afesp_ctx* ctx, double out[4]);

/* Test hook: what = 1 makes the next laned (small-system) amplitude update throw once, from a lane other than the main one
 * -- the failure mode of a capture that dies half-way (tests/test_gpu_cc.py). */
int afesp_test_inject(afesp_ctx* ctx, int what);
/* Test / diagnostic hook, per context: launches so far of {the streamed tall x skinny kernel, the gather kernel through the operator
 * layer's planner, the LDS-DMA GEMM with 128-row tiles, the LDS-DMA GEMM with 96-row tiles where the rows end} -- tests check with it
 * that a product took the kernel meant for its shape (tests/test_gpu_operators.py, tests/test_gpu_cc.py). */
int afesp_launch_counts(afesp_ctx* ctx, uint64_t out[4]);
/* Test / diagnostic hook, per process: launch sites that have resolved their kernel function under the process-wide first-use lock so
 * far (csrc/first_use.h: every first use of a kernel function -- by a launch, an occupancy query or the start-up thread's preload -- is
 * made under one lock, so two host threads never first-touch a translation unit or a function at the same time). */
uint64_t afesp_first_use_count(void);
/* Test hook, host logic only (no device needed): 1 if a system of these extents takes the grouped ring launches of the LDS-DMA GEMM
 * (csrc/ring.hip: from o v = 3584 on, and only while the 32-bit row byte offsets of an operand reach every row, 8 Kc o v < 4 GiB),
 * 0 if its six o^3 v^3 ring products stay on the gather kernel. */
int afesp_test_ring_path(int64_t nocc, int64_t nvirt);
/* Diagnostic builds only: n > 0: per (workgroup, wave) cycle sums of the GEMM kernel's last launch (tools/stamp_probe.py);
 * n < 0: the first -n phase sums of the (T) orbit kernel since the last call (tools/orbit_stamps.py).  Zeros in a shipped build. */
int afesp_debug_stamps(unsigned long long* out, int n);

/* Operator layer (src/linalg.fpp), exported for parity tests against the oracle.
 * afesp_gemm    = dgemm_wrapper (src/linalg.fpp:58-89): C(m x n) = alpha op(A) op(B) + beta C, host arrays.
 * afesp_permute4 = omp_reshape (src/linalg.fpp:99-156): out(perm) = beta*out + in; has_beta=0 zeroes `out` first. */
int afesp_gemm(afesp_ctx* ctx, char transA, char transB, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
               const double* B, double beta, double* C);
int afesp_permute4(afesp_ctx* ctx, const int64_t dims[4], const char order[4], const double* in, double* out, int has_beta,
                   double beta);
/* General labelled contraction on host arrays (tests): C[lc] = alpha sum A[la] B[lb] + beta C[lc], dense col-major. */
int afesp_contract(afesp_ctx* ctx, double alpha, const double* A, const char* la, const int64_t* dimsA, const double* B,
                   const char* lb, const int64_t* dimsB, double beta, double* C, const char* lc, const int64_t* dimsC,
                   int force_split, int force_tm, int force_tn);

/* ---- device-resident entry points used by bench.py (inputs generated in HBM; nothing crosses PCIe in the timed region)
 * Fill the context with the SURVEY.md 8(d) synthetic system: identity C, ladder orbital energies, hashed ERIs of
 * magnitude `scale` carrying the 8-fold symmetry, written straight into the physicist slices. */
int afesp_synthetic_init(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, double scale, uint64_t seed, int diis_n_errmat);
/* Hashed packed AO integrals of magnitude `scale` left resident on the device, as afesp_read_eri_text leaves a file's:
 * afesp_ao2mo_mp2(eri_packed = NULL) then transforms them (AO->MO timing at sizes with no bundled eri.dat). */
int afesp_synthetic_ao(afesp_ctx* ctx, int64_t nbasis, double scale, uint64_t seed);
/* Floating-point operations of one particle-particle ladder (src/ccsd.f90:1669) as this context evaluates it. */
int afesp_ccsd_pp_ladder_flop(afesp_ctx* ctx, double* flop);
/* ... and of one whole CCSD iteration (src/ccsd.f90:340-395; SURVEY.md 8(d) sum with the two pair-form products as executed). */
int afesp_ccsd_iteration_flop(afesp_ctx* ctx, double* flop);
/* Kernel-only timing helpers: average HIP-event milliseconds per launch over `reps` launches on the context stream. */
int afesp_time_pp_ladder(afesp_ctx* ctx, int reps, double* ms_per_launch);
/* y = a x + b y on n doubles (8 B per lane, 24 n bytes of HBM traffic per launch): PMC calibration / achievable-bandwidth probe. */
int afesp_bench_stream(afesp_ctx* ctx, int64_t n, int reps, double* ms_per_launch);
/* Same for an arbitrary labelled contraction on hashed device operands (dense column-major extents). */
int afesp_bench_contract(afesp_ctx* ctx, const char* la, const int64_t* dimsA, const char* lb, const int64_t* dimsB,
                         const char* lc, const int64_t* dimsC, int reps, double* ms_per_launch);
/* HIP-event timing of the (T) launches on the context stream.  Returns the totals accumulated since the previous call
 * (out = {GEMM ms, GEMM launches, GEMM flop, orbit-kernel ms, orbit launches, orbit algorithmic bytes, GEMM flop including the
 * zero padding its tiles execute, GEMM kernel: 1 LDS-DMA kernel / 0 gather kernel}), clears them and switches the
 * instrumentation on/off for the following afesp_ccsd_t calls. */
int afesp_profile(afesp_ctx* ctx, int enable, double out[8]);
/* Process-wide tuning overrides of the GEMM launcher (0 = heuristic): tile-walk group, tile shape (1|2|4), split-K. */
int afesp_set_tuning(int group_m, int force_tm, int force_tn, int force_split);

#ifdef __cplusplus
}
#endif
#endif
