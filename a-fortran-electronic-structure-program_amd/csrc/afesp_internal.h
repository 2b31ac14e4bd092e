// afesp_internal.h -- device context, tensor views and the contraction planner (C++ side, not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "gett.h"
#include "tgemm.h"

namespace afesp {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define AFESP_HIP(expr)                                                                                    \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            throw ::afesp::Error(2, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                                        ":" + std::to_string(__LINE__) + ")");                            \
    } while (0)

// Dense or strided view of device memory, Fortran index order (dim[0] fastest when dense).
struct Tensor {
    double* d = nullptr;
    int rank = 0;
    int64_t dim[6] = {0, 0, 0, 0, 0, 0};
    int64_t stride[6] = {0, 0, 0, 0, 0, 0};
    int64_t frozen = 0;   // non-zero: an id under which the contents never change (MO integrals): re-laid-out copies are kept (contract)
    int64_t size() const
    {
        int64_t s = 1;
        for (int i = 0; i < rank; ++i) s *= dim[i];
        return s;
    }
};

struct Plan {
    int64_t *offAm, *offAk, *offBk, *offBn, *offCm, *offCn;
    int M, N, K;
    bool swapped, a_kc, b_kc, wide;
    bool a_mu = false, b_nu = false;   // consecutive m (n) are consecutive in A (B): its leading free label has unit stride there
    int repack = 0;                 // 1/2: caller's first/second operand is re-laid-out into scratch before the product
    int64_t repack_stride[6] = {0, 0, 0, 0, 0, 0};
    // the copy in the plan's scratch buffer is the re-laid-out tensor (frozen id, address) as of this scratch epoch: not made again
    int64_t repack_of = 0, repack_epoch = -1;
    const double* repack_src = nullptr;
};

struct Comm;
struct Recorder;   // fused.h

// Device arena owned by the context.  On this runtime a hipMalloc of a GB or more is normally 0.3 ms but now and then takes
// 0.1-5 s, and a calculation frees and re-allocates tens of GB between its stages (AO->MO temporaries,
// the CCSD tensors, the (T) pool).  Blocks given back therefore stay with the context and are handed out again -- best fit,
// at most a quarter larger than asked for -- so that only the first use of a size ever reaches the driver; when the driver
// is out of memory the idle blocks are returned to it and the request is retried.
struct Arena {
    std::map<void*, size_t> live;            // blocks handed out
    std::multimap<size_t, void*> idle;       // blocks given back, by size
    size_t idle_bytes = 0, driver_calls = 0, reuse_hits = 0;
    void* get(size_t bytes);
    void* take_largest(size_t min_bytes, size_t max_bytes, size_t* got);   // the largest idle block within the limits (or null), whole
    void put(void* p);
    void trim();                             // hipFree every idle block
    void destroy();                          // ... and every live one
};

struct Context {
    int device = 0;
    Comm* comm = nullptr;                 // rank-to-rank sums (comm.h); null = single rank
    int cc_split_mode = -1;               // afesp_ccsd_set_split: 1 split the CCSD iteration over the ranks, 0 replicas, -1 environment (default off)
    int test_throw = 0;                   // test hook (afesp_test_inject): the next laned amplitude update throws
    int fused_mode = -1;                  // afesp_ccsd_set_fused: 1 launch-fused small-system path on, 0 off, -1 environment (default on)
    // Recording (fused.h): the offset tables of the plans made meanwhile are not uploaded one by one -- their host images wait here
    // (destination, image) and go to the device in as few copies as their destinations are contiguous (plan_alloc hands out
    // consecutive pieces of a slab: normally ONE copy for the 43 plans of an iteration).  plan_uploads_issue: the copies, no wait;
    // the images stay alive until plan_uploads_done (after a stream synchronisation).
    struct PendingTables { int64_t* dst; std::vector<int64_t> img; };
    std::vector<PendingTables> pending_host;
    std::vector<std::vector<int64_t>> pending_merged;
    void plan_uploads_issue();
    void plan_uploads_done() { pending_host.clear(); pending_merged.clear(); }
    Recorder* rec = nullptr;              // set while a call sequence is being recorded for the launch-fused path (fused.h): nothing is launched
    hipStream_t stream = nullptr;
    Arena arena;                          // every device allocation of the context goes through it
    std::vector<void*> owned;             // everything freed at destroy
    GettWorkspace ws{nullptr, 0};
    TgLaunchState tg;                     // ticket counters / grid size of this context's tgemm_kernel launches (tgemm.h)
    unsigned long long n_tall = 0, n_gett = 0;   // launches of the streamed tall x skinny kernel / the gather kernel by contract() (afesp_launch_counts)
    bool in_repack = false;               // set while contract() runs on a re-laid-out operand
    GettProblem* capture = nullptr;       // set by contract_pair: the next product is handed back instead of being launched
    // Lanes: extra streams (each with its own split-K workspace) on which independent chains of small launches run side
    // by side.  `stream` / `ws` above always denote the lane in use; lane 0 is the context's main stream.
    struct Lane {
        hipStream_t stream = nullptr;
        GettWorkspace ws{nullptr, 0};
        hipEvent_t done = nullptr;
    };
    std::vector<Lane> lanes;
    size_t lane_ws_bytes = (size_t)8 << 20;   // split-K workspace of a lane (lanes made so far are re-equipped when this grows)
    // Lanes made ahead of time on the context's start-up thread (afesp_ctx_create): five streams are five hardware queues,
    // 10-25 ms that a small molecule would otherwise pay inside its first iteration.  fork() adopts them.
    struct Prepared {
        std::vector<Lane> lanes;
        void* ws_block = nullptr;
        size_t ws_bytes = 0;
    };
    Prepared prepared;
    std::thread startup;                  // makes `prepared` and preloads the kernels' code; joined by fork() and at destruction
    static void prepare_lanes(Prepared& out, int nlanes);   // no access to the context: safe beside the caller's thread
    hipEvent_t fork_ev = nullptr;
    int cur_lane = 0;
    void fork(int nlanes);                // every lane waits for what the main stream has queued so far
    void use_lane(int i);
    void join();                          // the main stream waits for every lane; back on lane 0
    int mark();                           // event after what the lane in use has queued so far ...
    void wait(int mark_id);               // ... which the lane in use now waits for
    std::vector<hipEvent_t> marks;
    int marks_used = 0;
    double* scal = nullptr;               // small device scratch for reductions (64 doubles)
    double* scal_host = nullptr;          // pinned mirror
    // results a kernel writes straight into host memory (coherent, mapped): [energy, rms, -, sequence number, ..., B(i,j) at 8 + i + 16 j]
    // -- the sequence number is written last; res_seq is the number the next such launch will write (k_cc_tail)
    double* res_host = nullptr;
    double* res_dev = nullptr;
    int64_t res_seq = 0;
    int64_t pub_seq = 0;                  // sequence number of host_scalars' publishing area (res_host + 264)
    std::map<std::string, Plan> plans;
    size_t plan_bytes = 0;                // device bytes of their offset tables
    // plan tables come out of slabs (a small system builds ~45 plans in its first iteration: one device allocation each was
    // a third of that iteration); large tables get blocks of their own
    std::vector<void*> plan_blocks;
    int64_t* plan_slab = nullptr;
    size_t plan_slab_left = 0;
    int64_t* plan_alloc(size_t n);        // n int64, 16-byte aligned, not zeroed
    void plan_clear();                    // drops every plan and its tables
    int64_t plan_epoch = 0;               // bumped by plan_clear: compiled programs that point into the tables go stale (fused.h)
    std::map<std::string, std::pair<void*, size_t>> cache;   // named scratch buffers kept across calls (not zeroed)
    std::string last_error;
    // optional HIP-event timing of the (T) launches (bench.py roofline): enabled by afesp_profile
    bool prof = false;
    double prof_gemm_ms = 0.0, prof_gemm_flop = 0.0, prof_orbit_ms = 0.0, prof_orbit_bytes = 0.0;
    double prof_gemm_flop_padded = 0.0;   // ... including the zero padding the tiles execute (tile edges, K steps)
    int prof_gemm_kind = 0;               // 1: the LDS-DMA kernel (tgemm.h), 0: the grouped gather kernel (gett.h)
    int64_t prof_gemm_launches = 0, prof_orbit_launches = 0;

    double* alloc(int64_t n);             // zero-initialised doubles
    double* alloc_raw(int64_t n);         // uninitialised (large temporaries that are fully overwritten)
    int64_t* alloc_i64(int64_t n);
    void release(void* p);                // early free of an `owned` buffer
    double* scratch(const std::string& name, int64_t ndoubles);   // cached, uninitialised, grows on demand
    void drop_scratch();
    void drop_scratch(const std::string& prefix);   // only the cached buffers whose name starts with `prefix`
    int64_t scratch_epoch = 0;            // bumped whenever cached scratch buffers are freed (captured graphs go stale)
    // what the (T) operand copies in the cached buffers t_vt / t_tt / t_vs / t_ts (/ t_vt2 / t_tt2) were built from: state, its
    // amplitude epoch, the scratch epoch, whether ts is there, the CR epoch (-1: not built)
    int64_t amp_clock = 0;                // source of CCState::amp_epoch / cr_epoch
    const void* t_ops_owner = nullptr;
    int64_t t_ops_amp = -1, t_ops_scratch = -1, t_ops_cr = -1;
    bool t_ops_ts = false;
    Tensor tensor(std::initializer_list<int64_t> dims);
    void sync();
    void quiesce();                       // every lane and the main stream idle (before memory returns to the arena, after an error)
    ~Context();
};

Tensor view(double* d, std::initializer_list<int64_t> dims);

// C[lc] = alpha * sum_K A[la] * B[lb] + beta * C[lc].  One label character per tensor index; labels shared by
// A and B and absent from C are summed.  Optional batching: nbatch problems whose operand bases are shifted by
// the device arrays bA/bB/bC (element offsets; null = no shift).
void contract(Context& cx, double alpha, const Tensor& A, const char* la, const Tensor& B, const char* lb, double beta,
              const Tensor& C, const char* lc, int nbatch = 1, const int64_t* bA = nullptr, const int64_t* bB = nullptr,
              const int64_t* bC = nullptr, int force_split = 0, int force_tm = 0, int force_tn = 0);

// Two products that stream ONE tall array against ONE skinny matrix, enumerated alike in both label strings (t(j,e) <eb|ia> and
// <be|ia> t(j,e)): one launch in which the array crosses HBM once where the pair qualifies (tall.h, tall_dual_kernel), otherwise
// the two calls of contract() one after the other.
struct ContractCall {
    double alpha;
    const Tensor* A;
    const char* la;
    const Tensor* B;
    const char* lb;
    double beta;
    const Tensor* C;
    const char* lc;
};
void contract_pair(Context& cx, const ContractCall& c1, const ContractCall& c2);

// out[lo] = beta * out[lo] + alpha * in[li]  (li is a permutation of lo)
void permute_add(Context& cx, double alpha, const Tensor& in, const char* li, double beta, const Tensor& out,
                 const char* lo);

// ---- elementwise / reduction kernels (kernels.hip)
void k_fill(Context& cx, double* x, int64_t n, double val);
void k_copy(Context& cx, double* dst, const double* src, int64_t n);
void k_axpby(Context& cx, double* y, double a, const double* x, double b, int64_t n);   // y = a x + b y
void k_ivv_diag(Context& cx, double* ivv, const double* y, const double* x, int o, int v, int a0, int a1);   // columns a in [a0, a1):   // I_vv(b,a) += 2 sum_m y(m,b,m,a) - sum_m x(b,m,m,a)
void k_div(Context& cx, double* out, const double* num, const double* den, int64_t n);
void k_antisym_pair(Context& cx, double* out, const double* in, int64_t d0, int64_t d1, int64_t d2, int64_t d3,
                    int which);   // which=0: 2x - x(swap idx 0,1)   which=1: 2x - x(swap idx 2,3)
void k_asym_c(Context& cx, double* asym, double* c, const double* t1, const double* t2, int o, int v);
// t2 = (P(ia/jb)[r2 + r2b + r2c] + pp + v_oovv) / D2 and t1 = (r1 + r1b) / D1 (ccsd.f90:1720-1728); r2b, r2c, r1b may be null
// r2y: a partial residual held with i and j exchanged, r2y(j,i,a,b) (ring.hip), or null
void k_t2_update(Context& cx, double* t2, const double* r2, const double* r2b, const double* r2c, const double* v_oovv, const double* D2,
                 const double* pp, int o, int v, double* t1, const double* r1, const double* r1b, const double* D1, const double* r2y = nullptr);
void k_add_swapped(Context& cx, double* out, const double* y, int o, int v);   // out(i,j,a,b) += y(j,i,a,b)
void k_r2_full(Context& cx, double* out, const double* r2, const double* pp, int o, int v);
void k_denominators(Context& cx, double* D1, double* D2, const double* e, int o, int v);
// symmetric / antisymmetric operands of the pp-ladder (pairs x <= y indexed y(y+1)/2 + x, pairs x < y indexed y(y-1)/2 + x)
void k_vvvv_sympack_packed(Context& cx, double* vs, double* va, const double* packed, int o, int v, int64_t ks, int64_t ka);
// half: 1/2 (x(ijef) +- x(ijfe)), 1/4 on e == f (the weights of k_vvvv_sympack_packed: the integral side of a pair-form product)
void k_c_sympack(Context& cx, double* cs, double* ca, const double* c, int o, int v, int64_t ns, int64_t na, bool half = false);
// I_oooo(k,l,i,j) += Xs(kl,ij) +- Xa(kl,ij) (+ where k < l and i < j order alike); xs / xa: (kl) x (ij) pairs, leading dimensions ns / na
void k_oooo_pair_expand_add(Context& cx, double* I, const double* xs, const double* xa, int o, int64_t ns, int64_t na);
// Is(ij,mn) = 1/2 (I(ijmn) + I(ijnm)) (1/4 on m == n) over i <= j, m <= n;  Ia(ij,mn) = 1/2 (I(ijmn) - I(ijnm)) over i < j, m < n
void k_oooo_sympack(Context& cx, double* is, double* ia, const double* I, int o, int64_t ns, int64_t na);
// dst[k0 + r + ld * col] = src[r + ns * col], r < ns, col < ncol: c+-(mn, .) behind V+-(ef, .) in every row of the ladder's operand
void k_rows_append(Context& cx, double* dst, int64_t ld, int64_t k0, const double* src, int64_t ns, int64_t ncol);
void k_pp_expand(Context& cx, double* pp, const double* ps, const double* pa, int o, int v, int64_t ns, int64_t na,
                 int64_t p0 = 0, int64_t p1 = -1);   // rows [p0, p1) of PP only (a rank's share); default: all
void k_vvx_sympack(Context& cx, double* ws, double* wa, const double* x, int v, int64_t ncol, int64_t ks, int64_t ka);
void k_pair_expand_add(Context& cx, double* out, const double* ps, const double* pa, int o, int64_t ncol, int64_t ns, int64_t na);
// r1x(i,a) = sum_m [2 X(m,i,m,a) - X(i,m,m,a)] of X(j,k,i',a) = Ts(jk; i'a) +- Ta(jk; i'a) (the pair-form product of t2 with <ef|ia>, columns
// (i',a) = i' + o a): the term asym(m,i,e,f) <ef|ma> of the T1 equation (src/ccsd.f90:1569-1631) as a trace of that product
void k_ooov_r1_trace(Context& cx, double* r1x, const double* ps, const double* pa, int o, int v, int64_t ns, int64_t na);
// out[0] = sum (2 v(ijab) - v(ijba)) (t2 + t1 t1), out[1] = sum (t2 - t2_old)^2 ; then t2_old = t2
void k_cc_energy(Context& cx, double* out2, const double* v_oovv, const double* t1, const double* t2, double* t2_old,
                 int o, int v);
void k_mp2_energy(Context& cx, double* out1, const double* v_oovv, const double* D2, int o, int v);
double k_mp2_packed(Context& cx, const double* eri_packed, const double* e_host, int o, int v);   // the same from the packed MO integrals (device), one launch, result on the host; e_host: the n orbital energies
void k_dots(Context& cx, double* out, const double* x, const double* ybase, int64_t ystride, int ny, int64_t n,
            bool accumulate);   // out[j] (+)= <x, ybase + j*ystride>
void k_lincomb(Context& cx, double* out, const double* xbase, int64_t xstride, const double* coef_dev, int nx,
               int64_t n);      // out = sum_j coef[j] * x_j
void k_sub(Context& cx, double* out, const double* a, const double* b, int64_t n);
// DIIS extrapolation coefficients on the device: bmat (nerr x nerr) gets row/column `slot` from dots[0..n), coef[0..n) out
void k_diis_solve(Context& cx, double* coef, double* bmat, double* flag, int n, int nerr, int slot);   // sums k_diis_push's partials itself
void k_diis_push(Context& cx, double* ht, double* he, const double* amp, const double* amp_s, const double* hist_e, int64_t stride, int ny,
                 int slot, int64_t n);
// The tail of a small system's iteration in two launches (kernels.hip: cc_tail_kernel, cc_finalize_kernel): amplitude update, energy /
// rms sums and -- when ny > 0 -- the DIIS history push and solve for slot `slot` with ny active vectors; energy and rms land in
// cx.scal[0..1] and, with the DIIS failure flag and `seq`, in cx.res_host
struct CCTail {
    double *t2, *t1;
    const double *r2, *r1, *voovv, *D2, *D1, *pp;
    double* t2_old;
    int o, v;
    double *ht, *he;
    const double *amp_s, *hist_e;
    int64_t stride;
    int ny, slot, nerr;
    double *coef, *bmat;
    int64_t seq;
    const double* r2y = nullptr;   // a partial residual held with i and j exchanged (ring.hip), or null
    bool half_hist = true;         // (with r2y) the DIIS overlaps summed over a <= b only: every history vector has e(i,j,a,b) = e(j,i,b,a)
};
void k_cc_tail(Context& cx, const CCTail& a);
// out = sum_j coef[j] x_j with the coefficients handed over by value (the host solved for them)
void k_lincomb_vals(Context& cx, double* out, const double* xbase, int64_t xstride, const double* coef_host, int nx, int64_t n);
constexpr int DIIS_FLAG_SLOT = 48;   // cx.scal[48]: set by diis_solve_kernel when the solve fails, read with the energies
void diis_check_flag(Context& cx, const double* host_scal);   // throws the reference's error (ccsd.f90:666) if it is set
// pair-symmetric AO->MO: u(i,j,KL) from the packed array; out(k,l,PQ) = in(q,p,tri(k,l)); packed[tri(PQ,RS)] = full(s,r,PQ)
// (ld: leading dimension of the squared-up arrays, 0 = n; the LDS-DMA transforms pad it to whole K steps -- kernels.hip, pair_square_kernel)
void k_unpack_half(Context& cx, double* u, const double* packed, int n, int64_t c_begin = 0, int64_t c_end = -1, int ld = 0);   // slab of (kl) pairs
void k_pair_transpose(Context& cx, double* out, const double* in, int n, int ld = 0);
void k_pad_rows_zero(Context& cx, double* x, int n, int ld, int64_t ncol);   // x(n .. ld - 1, c) = 0 for every column c
// out(:,:,S) = C in(:,:,S) C^T for npairs symmetric n x n blocks, n <= 64: both quarter transforms of a pair index in one launch
void k_pair_xform(Context& cx, double* out, const double* in, const double* C, int n, int64_t npairs, int mode = 0);   // modes: kernels.hip
void k_square_transpose(Context& cx, double* out, const double* in, int64_t n);                                        // out(y, x) = in(x, y)
void k_pair_square_packed(Context& cx, double* out, const double* g, int n, int64_t c_begin, int64_t c_end);   // out(k,l,P) = g(P, tri(k,l))
void k_tri_pack(Context& cx, double* g, const double* half, int n, int64_t k_begin, int64_t k_end);           // g(PQ,K) = half(q,p,K)
void k_pack_pairs(Context& cx, double* packed, const double* full, int n, int64_t p_begin = 0, int64_t p_end = -1, int ld = 0);
// out(p,q,r,s) = packed[ index( (p+b0)(r+b2) | (q+b1)(s+b3) ) ]  physicist <pq|rs> from packed chemist (pr|qs)
void k_slice_phys(Context& cx, double* out, const double* packed, int d0, int d1, int d2, int d3, int b0, int b1, int b2,
                  int b3);
// Fock matrix from the half-unpacked integrals u(x,y,P) (k_unpack_half); work holds k_build_fock_work(n) doubles
void k_build_fock(Context& cx, double* fock, const double* hcore, const double* dens, const double* u, double* work, int n, int ld = 0);
int64_t k_build_fock_work(int n);
double* host_scalars(Context& cx, int n);
double* host_scalars_slot(Context& cx, double* seq);          // a kernel of the caller publishes itself (contract.hip); nullptr: use host_scalars
double* host_scalars_wait(Context& cx, int n, double seq);     // polls for that sequence number, returns the n values on the host

// Code-object preload: the runtime loads a translation unit's device code on the first use of one of its kernels (55-70 ms for
// the GEMM instantiations alone).  Each unit names one of its kernels here; afesp_ctx_create asks for their attributes on a
// background thread, so the load runs beside the caller's host work (input parsing, the SCF set-up) instead of inside the
// first CCSD iteration.
void preload_gett();
void preload_contract();
void preload_kernels();
void preload_small_path_kernels();   // kernels.hip: per-kernel first-use resolution of what a small system launches
void preload_ccsd_so();
void preload_fused();
void preload_triples();   // copies cx.scal[0..n) to pinned host memory and synchronises

}  // namespace afesp
