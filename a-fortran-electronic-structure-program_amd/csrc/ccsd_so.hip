// ccsd_so.hip -- spin-orbital CCSD (Stanton, Gauss, Watts, Bartlett 1991 as coded in the reference's src/ccsd.f90).
// Every contraction the reference writes as reshape + dgemm or as a loop nest is one label-driven contract() here;
// the index letters are the reference's.
#include "ccsd_so.h"
#include "fused.h"

#include <cmath>

namespace afesp {
namespace {

constexpr int TB = 256;
inline unsigned blocks_for(int64_t n) { return (unsigned)std::min<int64_t>((n + TB - 1) / TB, 65536); }
#define SO_STRIDE(X_, N_) for (int64_t X_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; X_ < (N_); X_ += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ int64_t tri(int64_t i, int64_t j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// ccsd.f90:108-143,193-207: out(p,q,r,s) = <p+b0 q+b1 || r+b2 s+b3> over spin orbitals x = 2 X + spin, from the packed
// chemist MO integrals: <pq|rs> = (PR|QS) [sp = sr][sq = ss]
__global__ void slice_asym_kernel(double* out, const double* packed, int d0, int d1, int d2, int d3, int b0, int b1, int b2, int b3)
{
    const int64_t n = (int64_t)d0 * d1 * d2 * d3;
    SO_STRIDE(x, n)
    {
        const int p = (int)(x % d0) + b0;
        int64_t y = x / d0;
        const int q = (int)(y % d1) + b1;
        y /= d1;
        const int r = (int)(y % d2) + b2, s = (int)(y / d2) + b3;
        const int P = p >> 1, Q = q >> 1, R = r >> 1, S = s >> 1;
        double val = 0.0;
        if ((p & 1) == (r & 1) && (q & 1) == (s & 1)) val += packed[tri(tri(P, R), tri(Q, S))];
        if ((p & 1) == (s & 1) && (q & 1) == (r & 1)) val -= packed[tri(tri(P, S), tri(Q, R))];
        out[x] = val;
    }
}

// ccsd.f90:437-448
}  // namespace
void preload_ccsd_so()
{
    first_use_touch(reinterpret_cast<const void*>(slice_asym_kernel));
    (void)hipGetLastError();
}
namespace {
__global__ void so_denominators_kernel(double* D1, double* D2, const double* e, int o, int v)
{
    const int64_t n2 = (int64_t)o * o * v * v, n1 = (int64_t)o * v;
    const int os = o / 2;
    SO_STRIDE(x, n2)
    {
        const int i = (int)(x % o), j = (int)((x / o) % o), a = (int)((x / ((int64_t)o * o)) % v), b = (int)(x / ((int64_t)o * o * v));
        D2[x] = e[i / 2] + e[j / 2] - e[a / 2 + os] - e[b / 2 + os];
        if (x < n1) D1[x] = e[(int)(x % o) / 2] - e[(int)(x / o) / 2 + os];
    }
}

// ccsd.f90:678-714
__global__ void so_tau_kernel(double* tau, double* tau_t, const double* t1, const double* t2, int o, int v, double* t1_w)
{
    const int64_t n2 = (int64_t)o * o * v * v;
    SO_STRIDE(x, n2)
    {
        if (x < (int64_t)o * v) t1_w[x] = t1[x];   // the t1 these intermediates belong to (so_build_W_vvvv forms W_abef from it on request)
        const int i = (int)(x % o), j = (int)((x / o) % o), a = (int)((x / ((int64_t)o * o)) % v), b = (int)(x / ((int64_t)o * o * v));
        const double y = t1[i + o * a] * t1[j + o * b] - t1[i + o * b] * t1[j + o * a];
        const double tt = t2[x] + 0.5 * y;
        tau_t[x] = tt;
        tau[x] = tt + 0.5 * y;
    }
}

// ccsd.f90:877-887: scratch(n,f,j,b) = 1/2 t2(j,n,f,b) + t1(j,f) t1(n,b)
__global__ void so_ring_operand_kernel(double* out, const double* t1, const double* t2, int o, int v)
{
    const int64_t n2 = (int64_t)o * v * o * v;
    SO_STRIDE(x, n2)
    {
        const int n = (int)(x % o), f = (int)((x / o) % v), j = (int)((x / ((int64_t)o * v)) % o), b = (int)(x / ((int64_t)o * v * o));
        out[x] = 0.5 * t2[j + (int64_t)o * (n + (int64_t)o * (f + (int64_t)v * b))] + t1[j + o * f] * t1[n + o * b];
    }
}

// ccsd.f90:1783-1806 (unrestricted branch): out[0] = 1/4 sum <ij||ab> (t2 + 2 t1 t1), out[1] = sum (t2 - t2_old)^2;
// t2_old <- t2.  One block-partial per block, summed by the caller's k_final_sum.
__global__ void so_energy_kernel(double* partial, const double* oovv, const double* t1, const double* t2, double* t2_old, int o, int v)
{
    __shared__ double red[2][TB / 64];
    const int64_t n2 = (int64_t)o * o * v * v;
    double e = 0.0, r = 0.0;
    SO_STRIDE(x, n2)
    {
        const int i = (int)(x % o), j = (int)((x / o) % o), a = (int)((x / ((int64_t)o * o)) % v), b = (int)(x / ((int64_t)o * o * v));
        const double t = t2[x];
        e += 0.25 * oovv[x] * (t + 2.0 * t1[i + o * a] * t1[j + o * b]);
        const double d = t - t2_old[x];
        r += d * d;
        t2_old[x] = t;
    }
    for (int off = 32; off > 0; off >>= 1) {
        e += __shfl_down(e, off, 64);
        r += __shfl_down(r, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = e;
        red[1][threadIdx.x >> 6] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double se = 0.0, sr = 0.0;
        for (int w = 0; w < TB / 64; ++w) {
            se += red[0][w];
            sr += red[1][w];
        }
        partial[blockIdx.x] = se;
        partial[gridDim.x + blockIdx.x] = sr;
    }
}

__global__ void so_sum2_kernel(double* out, const double* partial, int nblk)
{
    __shared__ double red[TB];
    const double* p = partial + (int64_t)blockIdx.x * nblk;
    double s = 0.0;
    for (int x = threadIdx.x; x < nblk; x += TB) s += p[x];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = TB / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// ccsd.f90:969-1034 assembled in one pass: r2 holds the two ladder terms; AB carries P(ij)P(ab), A carries P(ij), Bm
// carries P(ab):  t2 = [<ij||ab> + r2 + P(ij)P(ab) AB + P(ij) A + P(ab) Bm] / D
__global__ void so_t2_assemble_kernel(double* t2, const double* r2, const double* oovv, const double* AB, const double* A, const double* Bm,
                                      const double* D2, int o, int v)
{
    const int64_t n2 = (int64_t)o * o * v * v;
    SO_STRIDE(x, n2)
    {
        const int i = (int)(x % o), j = (int)((x / o) % o), a = (int)((x / ((int64_t)o * o)) % v), b = (int)(x / ((int64_t)o * o * v));
        const int64_t ji = j + (int64_t)o * (i + (int64_t)o * (a + (int64_t)v * b));
        const int64_t ba = i + (int64_t)o * (j + (int64_t)o * (b + (int64_t)v * a));
        const int64_t jiba = j + (int64_t)o * (i + (int64_t)o * (b + (int64_t)v * a));
        const double val = oovv[x] + r2[x] + (AB[x] - AB[ji] - AB[ba] + AB[jiba]) + (A[x] - A[ji]) + (Bm[x] - Bm[ba]);
        t2[x] = val / D2[x];
    }
}

// F_oo helper: G(m,i) = F_oo(m,i) + 1/2 Y(i,m)
__global__ void so_g_kernel(double* G, const double* F_oo, const double* Y, int o)
{
    SO_STRIDE(x, (int64_t)o * o)
    {
        const int m = (int)(x % o), i = (int)(x / o);
        G[x] = F_oo[x] + 0.5 * Y[i + o * m];
    }
}

// pairs x < y are numbered y(y-1)/2 + x
__device__ __forceinline__ void unpair_lt(int64_t p, int& lo, int& hi)
{
    int64_t h = (int64_t)((1.0 + sqrt(1.0 + 8.0 * (double)p)) * 0.5);
    while (h * (h - 1) / 2 > p) --h;
    while ((h + 1) * h / 2 <= p) ++h;
    hi = (int)h;
    lo = (int)(p - h * (h - 1) / 2);
}
// va(ef, ab) = <ab||ef>, e < f, a < b, leading dimension ka
__global__ void so_vvvv_asympack_kernel(double* va, const double* vvvv, int v, int64_t ka)
{
    const int64_t V = v, np = V * (V - 1) / 2;
    SO_STRIDE(x, np * np)
    {
        int e, f, a, b;
        unpair_lt(x % np, e, f);
        unpair_lt(x / np, a, b);
        va[x % np + ka * (x / np)] = vvvv[a + V * (b + V * (e + V * f))];
    }
}
// ta(ij, ef) = tau(i,j,e,f), i < j, e < f, leading dimension na
__global__ void so_tau_asympack_kernel(double* ta, const double* tau, int o, int v, int64_t na)
{
    const int64_t O = o, V = v, npo = O * (O - 1) / 2, npv = V * (V - 1) / 2;
    SO_STRIDE(x, npo * npv)
    {
        int i, j, e, f;
        unpair_lt(x % npo, i, j);
        unpair_lt(x / npo, e, f);
        ta[x % npo + na * (x / npo)] = tau[i + O * (j + O * (e + V * f))];
    }
}
// r2(i,j,a,b) = +- pa(ij, ab): + for (i < j, a < b) and (j < i, b < a), - when exactly one pair is exchanged, 0 on the diagonals
__global__ void so_ladder_expand_kernel(double* r2, const double* pa, int o, int v, int64_t na)
{
    const int64_t n2 = (int64_t)o * o * v * v;
    SO_STRIDE(x, n2)
    {
        const int i = (int)(x % o), j = (int)((x / o) % o), a = (int)((x / ((int64_t)o * o)) % v), b = (int)(x / ((int64_t)o * o * v));
        double val = 0.0;
        if (i != j && a != b) {
            const int il = min(i, j), ih = max(i, j), al = min(a, b), ah = max(a, b);
            const double p = pa[(int64_t)ih * (ih - 1) / 2 + il + na * ((int64_t)ah * (ah - 1) / 2 + al)];
            val = ((i < j) == (a < b)) ? p : -p;
        }
        r2[x] = val;
    }
}

#define SO_LAUNCH(kernel, n, ...)                                                         \
    do {                                                                                  \
        if ((n) > 0) {                                                                    \
            AFESP_KLAUNCH(kernel, dim3(blocks_for(n)), dim3(TB), 0, cx.stream, __VA_ARGS__); \
            AFESP_HIP(hipGetLastError());                                                 \
        }                                                                                 \
    } while (0)

// ... or, while a call sequence is being recorded for the levelled path (fused.h), noted with the memory it reads and writes
#define SO_KERNEL(READS, WRITES, kernel, n, ...)                                                                         \
    do {                                                                                                                 \
        if (cx.rec) {                                                                                                    \
            if ((n) > 0)                                                                                                 \
                cx.rec->opaque(READS, WRITES, [=](Context& c_) {                                                         \
                    AFESP_KLAUNCH(kernel, dim3(blocks_for(n)), dim3(TB), 0, c_.stream, __VA_ARGS__);                \
                    AFESP_HIP(hipGetLastError());                                                                        \
                });                                                                                                      \
        } else {                                                                                                         \
            SO_LAUNCH(kernel, n, __VA_ARGS__);                                                                           \
        }                                                                                                                \
    } while (0)
#define FR(p, n) frange(p, n)
typedef std::vector<FusedRange> Ranges;

}  // namespace

void so_init(Context& cx, SOState& s, int nbasis, int nel, const double* eri_mo_dev, const double* e_host, int diis_nerr,
             bool foo_as_published)
{
    if (nbasis <= 0 || nel <= 0 || (nel & 1) || nel >= 2 * nbasis)
        throw Error(1, "ccsd_so_init: need an even electron count with at least one occupied and one virtual spatial orbital");
    so_free(cx, s);
    const int o = nel, v = 2 * nbasis - nel;
    s.o = o; s.v = v; s.n = nbasis;
    s.foo_as_published = foo_as_published;
    const int64_t O = o, V = v, ov = O * V, o2v2 = O * O * V * V;
    s.e = cx.alloc(nbasis);
    AFESP_HIP(hipMemcpyAsync(s.e, e_host, sizeof(double) * nbasis, hipMemcpyHostToDevice, cx.stream));
    cx.sync();
    s.oooo = cx.tensor({O, O, O, O}); s.ooov = cx.tensor({O, O, O, V}); s.ovoo = cx.tensor({O, V, O, O});
    s.oovo = cx.tensor({O, O, V, O}); s.oovv = cx.tensor({O, O, V, V}); s.ovvo = cx.tensor({O, V, V, O});
    s.ovvv = cx.tensor({O, V, V, V}); s.vovv = cx.tensor({V, O, V, V}); s.vvvv = cx.tensor({V, V, V, V});
    auto slice = [&](const Tensor& t, int b0, int b1, int b2, int b3) {
        SO_LAUNCH(slice_asym_kernel, t.size(), t.d, eri_mo_dev, (int)t.dim[0], (int)t.dim[1], (int)t.dim[2], (int)t.dim[3], b0, b1, b2, b3);
    };
    slice(s.oooo, 0, 0, 0, 0); slice(s.ooov, 0, 0, 0, o); slice(s.ovoo, 0, o, 0, 0); slice(s.oovo, 0, 0, o, 0);
    slice(s.oovv, 0, 0, o, o); slice(s.ovvo, 0, o, o, 0); slice(s.ovvv, 0, o, o, o); slice(s.vovv, o, 0, o, o);
    slice(s.vvvv, o, o, o, o);
    s.D1 = cx.tensor({O, V}); s.D2 = cx.tensor({O, O, V, V});
    SO_LAUNCH(so_denominators_kernel, o2v2, s.D1.d, s.D2.d, s.e, o, v);
    s.nvec = ov + o2v2;
    s.amp = cx.alloc(s.nvec);
    s.t1 = view(s.amp, {O, V}); s.t2 = view(s.amp + ov, {O, O, V, V});
    double* res = cx.alloc(s.nvec);
    s.r1 = view(res, {O, V}); s.r2 = view(res + ov, {O, O, V, V});
    s.t2_old = cx.tensor({O, O, V, V});
    s.F_vv = cx.tensor({V, V}); s.F_oo = cx.tensor({O, O}); s.F_ov = cx.tensor({O, V});
    s.W_oooo = cx.tensor({O, O, O, O}); s.W_ovvo = cx.tensor({O, V, V, O});   // (W_vvvv: so_build_W_vvvv, on request only)
    {
        // the bare part of 1/2 tau W_abef over antisymmetric pairs (so_ladder)
        auto even = [](int64_t x) { return (x + 1) & ~(int64_t)1; };
        const int64_t npv = V * (V - 1) / 2, npo = O * (O - 1) / 2, ka = even(std::max<int64_t>(npv, 1)), na = even(std::max<int64_t>(npo, 1));
        s.lad_ka = ka; s.lad_na = na;
        if (npv > 0 && npo > 0) {
            s.va = cx.alloc(ka * npv); s.ta = cx.alloc(na * ka); s.pa = cx.alloc(na * npv);
            SO_LAUNCH(so_vvvv_asympack_kernel, npv * npv, s.va, s.vvvv.d, v, ka);
            // [ k (also n) | ka*m | na*k | na*m ]
            std::vector<int64_t> tab;
            const int64_t kn = std::max(ka, na);
            for (int64_t k = 0; k < kn; ++k) tab.push_back(k);
            for (int64_t m = 0; m < npv; ++m) tab.push_back(ka * m);
            for (int64_t k = 0; k < ka; ++k) tab.push_back(na * k);
            for (int64_t m = 0; m < npv; ++m) tab.push_back(na * m);
            s.lad_tab = cx.alloc_i64((int64_t)tab.size());
            AFESP_HIP(hipMemcpyAsync(s.lad_tab, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
            cx.sync();
        }
    }
    s.tau = cx.tensor({O, O, V, V}); s.tau_t = cx.tensor({O, O, V, V}); s.t1_w = cx.tensor({O, V});
    k_div(cx, s.t2.d, s.oovv.d, s.D2.d, o2v2);   // ccsd.f90:516 (t1 = 0 from the zero-filled allocation, :472)
    diis_alloc(cx, s, diis_nerr);
    s.energy = s.energy_old = s.rms = 0.0;
    s.ready = true;
    cx.sync();
}

void so_free(Context& cx, SOState& s)
{
    if (!s.o) return;
    double* bufs[] = {s.e, s.oooo.d, s.ooov.d, s.ovoo.d, s.oovo.d, s.oovv.d, s.ovvo.d, s.ovvv.d, s.vovv.d, s.vvvv.d, s.D1.d,
                      s.D2.d, s.amp, s.r1.d, s.t2_old.d, s.F_vv.d, s.F_oo.d, s.F_ov.d, s.W_oooo.d, s.W_vvvv.d, s.W_ovvo.d,
                      s.tau.d, s.tau_t.d, s.amp_s, s.hist_t, s.hist_e, s.coef, s.bmat, s.va, s.ta, s.pa, (double*)s.lad_tab, s.t1_w.d};
    for (double* b : bufs) cx.release(b);
    cx.drop_scratch();
    so_triples_plan_free(s);
    s = SOState();
}

void so_intermediates(Context& cx, SOState& s)
{
    auto C = [&](double al, const Tensor& A, const char* la, const Tensor& B, const char* lb, double be, const Tensor& Cc,
                 const char* lc) { contract(cx, al, A, la, B, lb, be, Cc, lc); };
    auto P = [&](double al, const Tensor& in, const char* li, double be, const Tensor& out, const char* lo) {
        permute_add(cx, al, in, li, be, out, lo);
    };
    const int o = s.o, v = s.v;
    const int64_t O = o, V = v;
    SO_KERNEL((Ranges{FR(s.t1.d, O * V), FR(s.t2.d, s.t2.size())}), (Ranges{FR(s.tau.d, s.t2.size()), FR(s.tau_t.d, s.t2.size()), FR(s.t1_w.d, O * V)}),
              so_tau_kernel, s.t2.size(), s.tau.d, s.tau_t.d, s.t1.d, s.t2.d, o, v, s.t1_w.d);
    // ---- build_F, ccsd.f90:716-797
    C(1.0, s.ovvv, "mafe", s.t1, "mf", 0.0, s.F_vv, "ae");             // :749-759
    C(0.5, s.tau_t, "mnaf", s.oovv, "mnfe", 1.0, s.F_vv, "ae");        // :783-786 (tmp_4_1(a,m,n,f) = tau~(m,n,a,f))
    C(-1.0, s.ooov, "nmie", s.t1, "ne", 0.0, s.F_oo, "mi");            // :757-767
    // :789-794: dgemm('N','N',nocc,nocc,...,tau_tilde,tmp_4_1,F_oo) yields C(i,m) = 1/2 sum tau~(i,n,e,f) <mn||ef>
    // and adds it to F_oo(i,m) although F_oo is read as (m,i) everywhere; kept as coded unless the context asks for
    // Stanton's Eq. 4 order, which is what the reference's shipped ref_out (2022) was computed with.
    C(0.5, s.tau_t, "inef", s.oovv, "mnef", 1.0, s.F_oo, s.foo_as_published ? "mi" : "im");
    C(1.0, s.oovv, "mnef", s.t1, "nf", 0.0, s.F_ov, "me");             // :770-780
    // ---- build_W, ccsd.f90:799-905
    // Eq. 6, stored W(i,j,m,n) (:842-846)
    Tensor sc = view(cx.scratch("so_sc_oooo", O * O * O * O), {O, O, O, O});
    C(1.0, s.ooov, "mnie", s.t1, "je", 0.0, sc, "mnij");               // :826
    P(1.0, s.oooo, "mnij", 0.0, s.W_oooo, "ijmn");
    P(1.0, sc, "mnij", 1.0, s.W_oooo, "ijmn");
    P(-1.0, sc, "mnji", 1.0, s.W_oooo, "ijmn");                        // :827-828
    C(0.5, s.oovv, "mnef", s.tau, "ijef", 1.0, s.W_oooo, "ijmn");      // :832-836
    // Eq. 7, W_abef (:852-861), is not stored: so_amplitudes contracts tau with its three terms one by one (so_ladder)
    // Eq. 8 (:866-902)
    k_copy(cx, s.W_ovvo.d, s.ovvo.d, s.ovvo.size());
    C(1.0, s.ovvv, "mbef", s.t1, "jf", 1.0, s.W_ovvo, "mbej");         // :867
    C(1.0, s.t1, "nb", s.oovo, "nmej", 1.0, s.W_ovvo, "mbej");         // :871-877
    Tensor ro = view(cx.scratch("so_ring_operand", O * V * O * V), {O, V, O, V});
    SO_KERNEL((Ranges{FR(s.t1.d, O * V), FR(s.t2.d, s.t2.size())}), (Ranges{FR(ro.d, ro.size())}), so_ring_operand_kernel, ro.size(), ro.d, s.t1.d,
              s.t2.d, o, v);
    C(-1.0, s.oovv, "mnef", ro, "nfjb", 1.0, s.W_ovvo, "mbej");        // :883-901
}

// W_abef = <ab||ef> - P(ab) t(m,b) <ma||ef> as a tensor W(e,f,a,b) (ccsd.f90:852-861): tests / afesp_ccsd_so_get_tensor only
void so_build_W_vvvv(Context& cx, SOState& s)
{
    const int64_t V = s.v;
    if (!s.W_vvvv.d) s.W_vvvv = cx.tensor({V, V, V, V});
    Tensor sv = view(cx.scratch("so_sc_vvvv", V * V * V * V), {V, V, V, V});
    contract(cx, 1.0, s.t1_w, "mb", s.ovvv, "maef", 0.0, sv, "baef");             // :853 (t1 as of the last so_intermediates)
    permute_add(cx, 1.0, s.vvvv, "abef", 0.0, s.W_vvvv, "efab");
    permute_add(cx, 1.0, sv, "baef", 1.0, s.W_vvvv, "efab");                      // + reshape_scratch(a,b,e,f) = scratch(b,a,e,f)
    permute_add(cx, -1.0, sv, "abef", 1.0, s.W_vvvv, "efab");                     // - scratch(a,b,e,f)
}

// r2(ijab) = 1/2 sum_ef tau(ijef) W(efab) (ccsd.f90:1021-1024) without forming W_abef -- at the H2O/cc-pVTZ shape (v = 106 spin
// orbitals) building it was three permuting passes over v^4 = 1 GB plus a v^4 scratch, 2.6 of the 4.3 ms of an iteration, and the
// product read it once more.  With W(efab) = <ab||ef> + t(m,b) <ma||ef> - t(m,a) <mb||ef>:
//   bare part   sum_{e<f} tau(ijef) <ab||ef> for i < j, a < b only -- tau and the integrals are antisymmetric in each pair -- one
//               product over pair indices, an eighth of the o^2 v^4 multiply-adds, against va(ef, ab) built once (so_init);
//   t1 parts    1/2 sum_m [ t(m,b) Z(ijma) - t(m,a) Z(ijmb) ],  Z(ijma) = sum_ef tau(ijef) <ma||ef>  (o^3 v^3 and two K = o products).
static void so_ladder(Context& cx, SOState& s)
{
    const int o = s.o, v = s.v;
    const int64_t O = o, V = v, npv = V * (V - 1) / 2, npo = O * (O - 1) / 2, ka = s.lad_ka, na = s.lad_na;
    if (npv == 0 || npo == 0) {
        k_fill(cx, s.r2.d, s.r2.size(), 0.0);
    } else {
        SO_KERNEL((Ranges{FR(s.tau.d, s.tau.size())}), (Ranges{FR(s.ta, na * ka)}), so_tau_asympack_kernel, npo * npv, s.ta, s.tau.d, o, v, na);
        const int64_t* t = s.lad_tab;
        const int64_t kn = std::max(ka, na);
        GettProblem gp;
        gp.alpha = 1.0; gp.beta = 0.0;
        gp.nbatch = 1; gp.batchA = gp.batchB = gp.batchC = nullptr;
        gp.a_kcontig = true; gp.b_kcontig = false;
        gp.wide = true;
        gp.A = s.va; gp.B = s.ta; gp.C = s.pa;
        gp.offAk = t; gp.offBn = gp.offCn = t;
        gp.offAm = t + kn; gp.offBk = t + kn + npv; gp.offCm = t + kn + npv + ka;
        gp.M = (int)npv; gp.N = (int)na; gp.K = (int)ka;
        if (cx.rec) cx.rec->product(gp, ka * npv, na * ka, na * npv);
        else AFESP_HIP(gett_launch(gp, cx.ws, cx.stream));
        SO_KERNEL((Ranges{FR(s.pa, na * npv)}), (Ranges{FR(s.r2.d, s.r2.size())}), so_ladder_expand_kernel, s.r2.size(), s.r2.d, s.pa, o, v, na);
    }
    Tensor Z = view(cx.scratch("so_Z", O * O * O * V), {O, O, O, V});
    contract(cx, 1.0, s.tau, "ijef", s.ovvv, "maef", 0.0, Z, "ijma");
    contract(cx, 0.5, Z, "ijma", s.t1, "mb", 1.0, s.r2, "ijab");
    contract(cx, -0.5, Z, "ijmb", s.t1, "ma", 1.0, s.r2, "ijab");
}

void so_amplitudes(Context& cx, SOState& s)
{
    auto C = [&](double al, const Tensor& A, const char* la, const Tensor& B, const char* lb, double be, const Tensor& Cc,
                 const char* lc) { contract(cx, al, A, la, B, lb, be, Cc, lc); };
    const int o = s.o, v = s.v;
    const int64_t O = o, V = v, o2v2 = O * O * V * V;
    // ---- T1, ccsd.f90:931-957
    C(1.0, s.t1, "ie", s.F_vv, "ae", 0.0, s.r1, "ia");
    C(-1.0, s.F_oo, "mi", s.t1, "ma", 1.0, s.r1, "ia");
    C(1.0, s.t1, "me", s.ovvo, "maei", 1.0, s.r1, "ia");
    C(1.0, s.t2, "miea", s.F_ov, "me", 1.0, s.r1, "ia");
    C(0.5, s.t2, "mife", s.ovvv, "mafe", 1.0, s.r1, "ia");
    C(-0.5, s.t2, "mnea", s.oovo, "mnei", 1.0, s.r1, "ia");
    // ---- T2, ccsd.f90:959-1034
    Tensor AB = view(cx.scratch("so_AB", o2v2), {O, O, V, V}), A = view(cx.scratch("so_A", o2v2), {O, O, V, V});
    Tensor Bm = view(cx.scratch("so_B", o2v2), {O, O, V, V});
    Tensor Q = view(cx.scratch("so_Q", O * V * O * O), {O, V, O, O});
    Tensor X = view(cx.scratch("so_X", V * V), {V, V}), Y = view(cx.scratch("so_Y", O * O), {O, O});
    Tensor G = view(cx.scratch("so_G", O * O), {O, O});
    // P(ij)P(ab) [ t_imae W_mbej - t_ie t_ma <mb||ej> ]  (:973-994)
    C(1.0, s.ovvo, "mbej", s.t1, "ie", 0.0, Q, "mbij");
    C(-1.0, s.t1, "ma", Q, "mbij", 0.0, AB, "ijab");
    C(1.0, s.t2, "miea", s.W_ovvo, "mbej", 1.0, AB, "ijab");
    // P(ab) [ t_ijae (F_be - 1/2 t_mb F_me) ]  (:996-1003)  and  -P(ab) t_ma <mb||ij> = -P(ab) Z(ijab)  (:1012-1016)
    k_copy(cx, X.d, s.F_vv.d, V * V);
    C(-0.5, s.t1, "mb", s.F_ov, "me", 1.0, X, "be");
    C(1.0, s.t2, "ijae", X, "be", 0.0, Bm, "ijab");
    C(-1.0, s.oovo, "ijam", s.t1, "mb", 1.0, Bm, "ijab");
    // -P(ij) [ t_imab (F_mj + 1/2 t_je F_me) ]  (:1004-1007,:1017-1020)  and  P(ij) t_ie <ej||ab>  (:1008-1011)
    C(1.0, s.t1, "je", s.F_ov, "me", 0.0, Y, "jm");
    SO_KERNEL((Ranges{FR(s.F_oo.d, O * O), FR(Y.d, O * O)}), (Ranges{FR(G.d, O * O)}), so_g_kernel, O * O, G.d, s.F_oo.d, Y.d, o);
    C(-1.0, G, "mi", s.t2, "mjab", 0.0, A, "ijab");
    C(1.0, s.t1, "ie", s.vovv, "ejab", 1.0, A, "ijab");
    // 1/2 tau_mnab W_mnij + 1/2 tau_ijef W_abef  (:1021-1024)
    so_ladder(cx, s);
    C(0.5, s.W_oooo, "ijmn", s.tau, "mnab", 1.0, s.r2, "ijab");
    // :1027-1028 (t1 first: r1 / D_ia)
    k_div(cx, s.t1.d, s.r1.d, s.D1.d, O * V);
    SO_KERNEL((Ranges{FR(s.r2.d, o2v2), FR(s.oovv.d, o2v2), FR(AB.d, o2v2), FR(A.d, o2v2), FR(Bm.d, o2v2), FR(s.D2.d, o2v2)}), (Ranges{FR(s.t2.d, o2v2)}),
              so_t2_assemble_kernel, o2v2, s.t2.d, s.r2.d, s.oovv.d, AB.d, A.d, Bm.d, s.D2.d, o, v);
}

int so_energy(Context& cx, SOState& s, double e_tol, double t_tol)
{
    const int64_t n2 = s.t2.size();
    const int nblk = (int)std::min<int64_t>((n2 + TB - 1) / TB, 1024);
    double* partial = cx.scratch("so_energy_partial", 2 * 1024);
    AFESP_KLAUNCH(so_energy_kernel, dim3(nblk), dim3(TB), 0, cx.stream, partial, s.oovv.d, s.t1.d, s.t2.d, s.t2_old.d, s.o, s.v);
    AFESP_HIP(hipGetLastError());
    AFESP_KLAUNCH(so_sum2_kernel, dim3(2), dim3(TB), 0, cx.stream, cx.scal, partial, nblk);
    AFESP_HIP(hipGetLastError());
    double* h = host_scalars(cx, DIIS_FLAG_SLOT + 1);
    diis_check_flag(cx, h);
    s.energy_old = s.energy;
    s.energy = h[0];
    s.rms = h[1];
    return (std::sqrt(h[1]) < t_tol && std::fabs(s.energy - s.energy_old) < e_tol) ? 1 : 0;
}

}  // namespace afesp
