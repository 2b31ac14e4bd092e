// kernels.hip -- HBM-bound elementwise / reduction / index kernels of the coupled-cluster path.
// All tensors are Fortran column-major (first index fastest); lanes run along the fastest index.
#include <algorithm>

#include "afesp_internal.h"
#include "fused.h"

namespace afesp {

namespace {
constexpr int TB = 256;
inline unsigned grid_for(int64_t n, int cap = 4096) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + TB - 1) / TB, cap)); }
#define GRID_STRIDE(IDX_, n) for (int64_t IDX_ = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; IDX_ < (n); IDX_ += (int64_t)gridDim.x * blockDim.x)

// Sum over the 64 lanes of a wave, the same value in every lane, in a fixed order (bit-reproducible).  Four butterfly steps inside
// each row of 16 lanes as DPP moves (one VALU instruction per 32-bit half; a __shfl is a ds_bpermute plus ~8 instructions of lane
// arithmetic -- eighteen sums of a block reduction were ~2000 instructions per wave), then the four row sums through v_readlane.
template <int CTRL>
__device__ __forceinline__ double dpp_permuted(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int from)   // `from` uniform: the value lands in scalar registers
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), from), hi = __builtin_amdgcn_readlane(__double2hiint(v), from);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_permuted<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_permuted<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_permuted<0x141>(v);   // row_half_mirror
    v += dpp_permuted<0x140>(v);   // row_mirror: every lane of a row holds the row's sum
    return ((lane_value(v, 0) + lane_value(v, 16)) + lane_value(v, 32)) + lane_value(v, 48);
}
// block-wide sum of up to NV values; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sm /* [NV*4] */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        v[q] = wave_sum(v[q]);
        if (lane == 0) sm[q * 4 + w] = v[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = sm[q * 4] + sm[q * 4 + 1] + sm[q * 4 + 2] + sm[q * 4 + 3];
    }
}

__global__ void fill_kernel(double* x, int64_t n, double val) { GRID_STRIDE(i, n) x[i] = val; }
__global__ void axpby_kernel(double* y, double a, const double* x, double b, int64_t n)
{
    GRID_STRIDE(i, n) y[i] = a * x[i] + (b != 0.0 ? b * y[i] : 0.0);
}
__global__ void div_kernel(double* out, const double* num, const double* den, int64_t n) { GRID_STRIDE(i, n) out[i] = num[i] / den[i]; }
__global__ void sub_kernel(double* out, const double* a, const double* b, int64_t n) { GRID_STRIDE(i, n) out[i] = a[i] - b[i]; }

// out = 2 x - x(with index pair swapped).  which=0: pair (0,1); which=1: pair (2,3).   (linalg.fpp:158-274, immutable form)
__global__ void antisym_pair_kernel(double* out, const double* in, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int which)
{
    const int64_t n = d0 * d1 * d2 * d3;
    GRID_STRIDE(x, n)
    {
        int64_t i0 = x % d0, r = x / d0, i1 = r % d1;
        r /= d1;
        int64_t i2 = r % d2, i3 = r / d2;
        int64_t y = which == 0 ? i1 + d0 * (i0 + d1 * (i2 + d2 * i3)) : i0 + d0 * (i1 + d1 * (i3 + d2 * i2));
        out[x] = 2.0 * in[x] - in[y];
    }
}

// asym_t2 = 2 t2 - t2(jiab) (ccsd.f90:1063-1064);  c = t2 + t1 t1 (ccsd.f90:1071-1079)
__global__ void asym_c_kernel(double* asym, double* c, const double* t1, const double* t2, int o, int v)
{
    const int64_t n = (int64_t)o * o * v * v;
    GRID_STRIDE(x, n)
    {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        double t = t2[x];
        asym[x] = 2.0 * t - t2[j + (int64_t)o * (i + (int64_t)o * (a + (int64_t)v * b))];
        c[x] = t + t1[i + o * a] * t1[j + o * b];
    }
}

// t2 = (r2(ijab) + r2(jiba) + v_oovv) / D2   (ccsd.f90:1720-1728).  The particle-particle ladder pp(ijab) =
// sum_ef c(ij,ef) <ef|ab> satisfies pp(ijab) = pp(jiba), so it is computed for a <= b only and stored packed as
// PP(i,j,p), p = b(b+1)/2 + a; its contribution to r2(ijab) + r2(jiba) is 2 * 1/2 * pp = PP (ccsd.f90:1669).
__global__ void t2_update_kernel(double* t2, const double* r2, const double* r2b, const double* r2c, const double* voovv, const double* D2,
                                 const double* pp, int o, int v, double* t1, const double* r1, const double* r1b, const double* D1,
                                 const double* r2y)
{
    // (r2b, r2c, r1b: the partial residuals of the laned iteration, null otherwise; t1 = (r1 + r1b) / D1 rides along)
    const int64_t n = (int64_t)o * o * v * v, n1 = (int64_t)o * v;
    GRID_STRIDE(x, n)
    {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        int64_t y = j + (int64_t)o * (i + (int64_t)o * (b + (int64_t)v * a));
        const int64_t lad = (a <= b) ? i + (int64_t)o * (j + (int64_t)o * ((int64_t)b * (b + 1) / 2 + a))
                                     : j + (int64_t)o * (i + (int64_t)o * ((int64_t)a * (a + 1) / 2 + b));
        double rx = r2[x], ry = r2[y];
        if (r2b) { rx += r2b[x]; ry += r2b[y]; }
        if (r2c) { rx += r2c[x]; ry += r2c[y]; }
        if (r2y) {   // (held with i and j exchanged: the same 8 o^2 bytes of memory as x resp. y)
            rx += r2y[j + (int64_t)o * (i + (int64_t)o * (a + (int64_t)v * b))];
            ry += r2y[i + (int64_t)o * (j + (int64_t)o * (b + (int64_t)v * a))];
        }
        t2[x] = (rx + ry + pp[lad] + voovv[x]) / D2[x];
        if (x < n1) t1[x] = (r1[x] + (r1b ? r1b[x] : 0.0)) / D1[x];
    }
}

// ccsd.f90:436-445
__global__ void denominators_kernel(double* D1, double* D2, const double* e, int o, int v)
{
    const int64_t n = (int64_t)o * o * v * v;
    GRID_STRIDE(x, n)
    {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        D2[x] = e[i] + e[j] - e[a + o] - e[b + o];
        if (j == 0 && b == 0) D1[i + o * a] = e[i] - e[a + o];
    }
}

// ccsd.f90:1764-1782: two sums, then t2_old = t2 (:1804).  Deterministic: fixed grid, per-block partials, ordered final sum.
constexpr int RED_BLOCKS = 512;
__global__ __launch_bounds__(TB) void cc_energy_kernel(double* partial, const double* voovv, const double* t1, const double* t2,
                                                       double* t2_old, int o, int v)
{
    __shared__ double sm[8];
    const int64_t n = (int64_t)o * o * v * v;
    double acc[2] = {0.0, 0.0};
    GRID_STRIDE(x, n)
    {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        double t = t2[x];
        double vx = voovv[i + (int64_t)o * (j + (int64_t)o * (b + (int64_t)v * a))];
        acc[0] += (2.0 * voovv[x] - vx) * (t + t1[i + o * a] * t1[j + o * b]);
        double d = t - t2_old[x];
        acc[1] += d * d;
        t2_old[x] = t;
    }
    block_sum<2>(acc, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = acc[0];
        partial[RED_BLOCKS + blockIdx.x] = acc[1];
    }
}
// mp2.f90:418-440 on the <ij|ab> slice
__global__ __launch_bounds__(TB) void mp2_energy_kernel(double* partial, const double* voovv, const double* D2, int o, int v)
{
    __shared__ double sm[4];
    const int64_t n = (int64_t)o * o * v * v;
    double acc[1] = {0.0};
    GRID_STRIDE(x, n)
    {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        double vx = voovv[i + (int64_t)o * (j + (int64_t)o * (b + (int64_t)v * a))];
        acc[0] += voovv[x] * (2.0 * voovv[x] - vx) / D2[x];
    }
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
// The same sum straight from the packed MO integrals (<ij|ab> = (ia|jb): mp2.f90:418-440 with the slice and the denominators formed on
// the fly), the ordered sum of the block partials by whichever block finishes last, and the result into the host's publishing
// area (contract.hip, host_scalars_slot): ONE launch where a small system's MP2 energy took five (slice, denominators, sum, final
// sum, publication: 29 of the 163 us of an AO->MO + MP2 call at n = 58).
struct Mp2Levels { double e[256]; };   // orbital energies by value (kernel arguments) when they fit: no upload in front of the launch
template <bool BYVAL>
__global__ __launch_bounds__(TB) void mp2_packed_kernel(double* partial, unsigned* counter, double* scal, double* pub, double seq,
                                                        const double* __restrict__ eri, const double* __restrict__ e_dev, Mp2Levels lv, int o, int v)
{
    __shared__ double sm[4];
    __shared__ bool last;
    __shared__ double e[BYVAL ? 256 : 1];
    if (BYVAL) {
        if ((int)threadIdx.x < o + v) e[threadIdx.x] = lv.e[threadIdx.x];
        __syncthreads();
    }
    const double* ep = BYVAL ? e : e_dev;
    const int64_t n = (int64_t)o * o * v * v;
    auto tri2 = [](int64_t p, int64_t q) { return p >= q ? p * (p + 1) / 2 + q : q * (q + 1) / 2 + p; };
    double acc[1] = {0.0};
    GRID_STRIDE(x, n)
    {
        const int i = (int)(x % o);
        int64_t r = x / o;
        const int j = (int)(r % o);
        r /= o;
        const int a = (int)(r % v), b = (int)(r / v);
        const int64_t ia = tri2(o + a, i), jb = tri2(o + b, j), ib = tri2(o + b, i), ja = tri2(o + a, j);
        const double vx = eri[tri2(ia, jb)], vex = eri[tri2(ib, ja)];
        acc[0] += vx * (2.0 * vx - vex) / (ep[i] + ep[j] - ep[o + a] - ep[o + b]);
    }
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = acc[0];
        __threadfence();
        last = atomicInc(counter, gridDim.x - 1) == gridDim.x - 1;   // (wraps to zero: ready for the next call)
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double tot[1] = {0.0};
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) tot[0] += __builtin_nontemporal_load(&partial[b]);
    block_sum<1>(tot, sm);
    if (threadIdx.x == 0) {
        scal[0] = tot[0];
        if (pub) {
            pub[0] = tot[0];
            __threadfence_system();
            __hip_atomic_store(&pub[64], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// out[q] (+)= sum_b partial[q*nblk + b], fixed order
__global__ void final_sum_kernel(double* out, const double* partial, int nblk, int nq, int accumulate)
{
    __shared__ double sm[4];
    for (int q = 0; q < nq; ++q) {
        double acc[1] = {0.0};
        for (int b = threadIdx.x; b < nblk; b += blockDim.x) acc[0] += partial[q * nblk + b];
        block_sum<1>(acc, sm);
        if (threadIdx.x == 0) out[q] = (accumulate ? out[q] : 0.0) + acc[0];
        __syncthreads();
    }
}

// partial[j*RED_BLOCKS + blk] = block's share of <x, y_j>
__global__ __launch_bounds__(TB) void dots_kernel(double* partial, const double* x, const double* ybase, int64_t ystride, int64_t n)
{
    __shared__ double sm[4];
    const double* y = ybase + (int64_t)blockIdx.y * ystride;
    double acc[1] = {0.0};
    GRID_STRIDE(i, n) acc[0] += x[i] * y[i];
    block_sum<1>(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.y * RED_BLOCKS + blockIdx.x] = acc[0];
}
// The first half of a DIIS update in one pass (ccsd.f90:633-673): the new amplitudes go into the history (ht), their difference
// to the previous ones into the error history (he = row `slot` of hist_e), and the block's shares of <he, hist_e_j>, j < ny,
// into partial[j*RED_BLOCKS + blk] (same partition and order as dots_kernel).
__global__ __launch_bounds__(TB) void diis_push_kernel(double* partial, double* ht, double* he, const double* amp, const double* amp_s,
                                                       const double* hist_e, int64_t stride, int ny, int slot, int64_t n)
{
    __shared__ double sm[16 * 4];
    double acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
    GRID_STRIDE(i, n)
    {
        const double a = amp[i], e = a - amp_s[i];
        ht[i] = a;
        he[i] = e;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < ny) acc[j] += e * (j == slot ? e : hist_e[(int64_t)j * stride + i]);
    }
    block_sum<16>(acc, sm);
    if (threadIdx.x == 0)
        for (int j = 0; j < ny; ++j) partial[j * RED_BLOCKS + blockIdx.x] = acc[j];
}
__global__ void lincomb_kernel(double* out, const double* xbase, int64_t xstride, const double* coef, int nx, int64_t n)
{
    GRID_STRIDE(i, n)
    {
        double s = 0.0;
        for (int j = 0; j < nx; ++j) s += coef[j] * xbase[j * xstride + i];
        out[i] = s;
    }
}

__device__ __forceinline__ int64_t tri(int64_t i, int64_t j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
// pair index of x <= y: y(y+1)/2 + x; inverse of it
__device__ __forceinline__ void unpair(int64_t p, int& lo, int& hi)
{
    int64_t h = (int64_t)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (h * (h + 1) / 2 > p) --h;
    while ((h + 1) * (h + 2) / 2 <= p) ++h;
    hi = (int)h;
    lo = (int)(p - h * (h + 1) / 2);
}

// ccsd.f90:496-512: out(p,q,r,s) = <p+b0 q+b1 | r+b2 s+b3> = (pr|qs) read from the packed chemist array
__global__ void slice_phys_kernel(double* out, const double* packed, int d0, int d1, int d2, int d3, int b0, int b1, int b2, int b3)
{
    const int64_t n = (int64_t)d0 * d1 * d2 * d3;
    GRID_STRIDE(x, n)
    {
        int p = (int)(x % d0);
        int64_t r_ = x / d0;
        int q = (int)(r_ % d1);
        r_ /= d1;
        int r = (int)(r_ % d2), s = (int)(r_ / d2);
        out[x] = packed[tri(tri(p + b0, r + b2), tri(q + b1, s + b3))];
    }
}
// build_fock (hf.f90:349-385): F(i,j) = H(i,j) + sum_kl D(k,l) [2 (ij|kl) - (ik|jl)].  Both sums run over the half-unpacked
// integrals u(x,y,P) = (xy|ab), P = tri(a,b), a >= b -- the first stage of the AO->MO transform, built once per SCF -- so
// every unique pair slab is read once per sum, in 8-byte-per-lane coalesced rows, and every partial sum is combined in a
// fixed order (bit-reproducible run to run):
//   J(x,y)  = sum_P u(x,y,P) dv(P),  dv(P) = D(a,b) + D(b,a)  (D(a,a) on a == b)           fock_j_kernel, FOCK_CHUNKS partial sums
//   K(x,a) += sum_y u(x,y,P) D(y,b),   K(x,b) += sum_y u(x,y,P) D(y,a)  (a != b)              fock_k_kernel, one workgroup per slab
constexpr int FOCK_CHUNKS = 64;
__global__ void fock_dv_kernel(double* dv, const double* dens, int n)
{
    const int64_t np = (int64_t)n * (n + 1) / 2;
    GRID_STRIDE(p, np)
    {
        int b, a;
        unpair(p, b, a);
        dv[p] = a == b ? dens[a + (int64_t)n * a] : dens[a + (int64_t)n * b] + dens[b + (int64_t)n * a];
    }
}
// (ld: leading dimension of u(x, y, P), n or -- where afesp_ao2mo_mp2 will run its transforms on the LDS-DMA GEMM -- n rounded up to
// whole K steps, afesp_internal.h: ao2mo_ld)
__global__ __launch_bounds__(256) void fock_j_kernel(double* jpart, const double* u, const double* dv, int n, int ld)
{
    const int64_t n2 = (int64_t)n * n, np = (int64_t)n * (n + 1) / 2, pl = (int64_t)ld * n;
    const int64_t xy = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (np + FOCK_CHUNKS - 1) / FOCK_CHUNKS, p0 = blockIdx.y * per, p1 = p0 + per < np ? p0 + per : np;
    if (xy >= n2) return;
    const int64_t at = xy % n + (int64_t)ld * (xy / n);
    double acc = 0.0;
#pragma unroll 8
    for (int64_t p = p0; p < p1; ++p) acc += u[at + pl * p] * dv[p];
    jpart[(int64_t)blockIdx.y * n2 + xy] = acc;
}
__global__ __launch_bounds__(256) void fock_k_kernel(double* kp1, double* kp2, const double* u, const double* dens, int n, int ld)
{
    extern __shared__ double dcol[];   // D(:,b) then D(:,a)
    const int64_t n2 = (int64_t)ld * n, p = blockIdx.x;
    int b, a;
    unpair(p, b, a);
    for (int y = threadIdx.x; y < n; y += 256) {
        dcol[y] = dens[y + (int64_t)n * b];
        dcol[n + y] = dens[y + (int64_t)n * a];
    }
    __syncthreads();
    const double* m = u + n2 * p;
    for (int x = threadIdx.x; x < n; x += 256) {
        double w1 = 0.0, w2 = 0.0;
#pragma unroll 8
        for (int y = 0; y < n; ++y) {
            const double v = m[x + (int64_t)ld * y];
            w1 += v * dcol[y];
            w2 += v * dcol[n + y];
        }
        kp1[p * n + x] = w1;
        kp2[p * n + x] = w2;
    }
}
// F(x,a) = H(x,a) + 2 sum_c jpart[c](x,a) - sum_{b<=a} kp1[tri(a,b)](x) - sum_{b>a} kp2[tri(b,a)](x)
__global__ void fock_reduce_kernel(double* fock, const double* hcore, const double* jpart, const double* kp1, const double* kp2, int n)
{
    const int64_t n2 = (int64_t)n * n;
    GRID_STRIDE(xa, n2)
    {
        const int x = (int)(xa % n), a = (int)(xa / n);
        double j = 0.0, k = 0.0;
        for (int c = 0; c < FOCK_CHUNKS; ++c) j += jpart[(int64_t)c * n2 + xa];
        for (int b = 0; b <= a; ++b) k += kp1[((int64_t)a * (a + 1) / 2 + b) * n + x];
        for (int b = a + 1; b < n; ++b) k += kp2[((int64_t)b * (b + 1) / 2 + a) * n + x];
        fock[xa] = hcore[xa] + 2.0 * j - k;
    }
}
}  // namespace

#define LAUNCH(kernel, grid, ...)                                               \
    do {                                                                        \
        AFESP_KLAUNCH(kernel, grid, dim3(TB), 0, cx.stream, __VA_ARGS__);  \
        AFESP_HIP(hipGetLastError());                                           \
    } while (0)

void preload_kernels()
{
    first_use_touch(reinterpret_cast<const void*>(fill_kernel));
    (void)hipGetLastError();
}

void k_fill(Context& cx, double* x, int64_t n, double val)
{
    if (cx.rec) {
        if (n > 0) cx.rec->opaque({}, {frange(x, n)}, [=](Context& c_) { k_fill(c_, x, n, val); });
        return;
    }
    if (n > 0) LAUNCH(fill_kernel, dim3(grid_for(n)), x, n, val);
}
void k_copy(Context& cx, double* dst, const double* src, int64_t n)
{
    if (cx.rec) {   // launch-fused path (fused.h): one of the copies of an elementwise stage
        const int64_t one = 1;
        if (n > 0) cx.rec->elementwise(dst, src, 1, &n, &one, &one, 1.0, 0.0);
        return;
    }
    if (n > 0) AFESP_HIP(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToDevice, cx.stream));
}
void k_axpby(Context& cx, double* y, double a, const double* x, double b, int64_t n)
{
    if (cx.rec) {
        const int64_t one = 1;
        if (n > 0) cx.rec->elementwise(y, x, 1, &n, &one, &one, a, b);
        return;
    }
    if (n > 0) LAUNCH(axpby_kernel, dim3(grid_for(n)), y, a, x, b, n);
}
void k_div(Context& cx, double* out, const double* num, const double* den, int64_t n)
{
    if (cx.rec) {
        if (n > 0) cx.rec->opaque({frange(num, n), frange(den, n)}, {frange(out, n)}, [=](Context& c_) { k_div(c_, out, num, den, n); });
        return;
    }
    if (n > 0) LAUNCH(div_kernel, dim3(grid_for(n)), out, num, den, n);
}
// I_vv(b,a) += 2 sum_m y(m,b,m,a) - sum_m x(b,m,m,a): the t1 term of I_vv (ccsd.f90:1096-1101) from the two products over
// <eb|ia> the iteration forms anyway, y(j,b,i,a) = t(j,e) <eb|ia> and x(b,j,i,a) = <be|ia> t(j,e) (ccsd.hip)
__global__ __launch_bounds__(256) void ivv_diag_kernel(double* ivv, const double* y, const double* x, int o, int v, int a0, int a1)
{
    const int64_t idx0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx0 >= (int64_t)v * (a1 - a0)) return;
    const int64_t b = idx0 % v, a = a0 + idx0 / v, idx = b + (int64_t)v * a;
    double sy = 0.0, sx = 0.0;
    for (int m = 0; m < o; ++m) {
        sy += y[m + (int64_t)o * (b + (int64_t)v * (m + (int64_t)o * a))];
        sx += x[b + (int64_t)v * (m + (int64_t)o * (m + (int64_t)o * a))];
    }
    ivv[idx] += 2.0 * sy - sx;
}
void k_ivv_diag(Context& cx, double* ivv, const double* y, const double* x, int o, int v, int a0, int a1)
{
    if (cx.rec) throw Error(2, "k_ivv_diag: not part of a recorded sequence");
    if (a1 > a0) LAUNCH(ivv_diag_kernel, dim3(grid_for((int64_t)v * (a1 - a0))), ivv, y, x, o, v, a0, a1);
}
void k_sub(Context& cx, double* out, const double* a, const double* b, int64_t n) { if (n > 0) LAUNCH(sub_kernel, dim3(grid_for(n)), out, a, b, n); }
void k_antisym_pair(Context& cx, double* out, const double* in, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int which)
{
    int64_t n = d0 * d1 * d2 * d3;
    if (n > 0) LAUNCH(antisym_pair_kernel, dim3(grid_for(n)), out, in, d0, d1, d2, d3, which);
}
void k_asym_c(Context& cx, double* asym, double* c, const double* t1, const double* t2, int o, int v)
{
    if (cx.rec) {
        const int64_t n2 = (int64_t)o * o * v * v;
        cx.rec->opaque({frange(t1, (int64_t)o * v), frange(t2, n2)}, {frange(asym, n2), frange(c, n2)},
                       [=](Context& c_) { k_asym_c(c_, asym, c, t1, t2, o, v); });
        return;
    }
    LAUNCH(asym_c_kernel, dim3(grid_for((int64_t)o * o * v * v)), asym, c, t1, t2, o, v);
}
__global__ void add_swapped_kernel(double* out, const double* y, int o, int64_t n)
{
    GRID_STRIDE(x, n)
    {
        const int i = (int)(x % o), j = (int)((x / o) % o);
        out[x] += y[x + (int64_t)(j - i) * (1 - o)];   // (j + o i) - (i + o j) = (j - i)(1 - o)
    }
}
void k_add_swapped(Context& cx, double* out, const double* y, int o, int v)
{
    const int64_t n = (int64_t)o * o * v * v;
    LAUNCH(add_swapped_kernel, dim3(grid_for(n)), out, y, o, n);
}
void k_t2_update(Context& cx, double* t2, const double* r2, const double* r2b, const double* r2c, const double* v_oovv, const double* D2,
                 const double* pp, int o, int v, double* t1, const double* r1, const double* r1b, const double* D1, const double* r2y)
{
    if (cx.rec) {
        if (r2y) throw Error(2, "k_t2_update: the exchanged partial residual is not part of a recorded sequence");
        const int64_t n2 = (int64_t)o * o * v * v, n1 = (int64_t)o * v, np = (int64_t)o * o * ((int64_t)v * (v + 1) / 2);
        std::vector<FusedRange> rd = {frange(r2, n2), frange(v_oovv, n2), frange(D2, n2), frange(pp, np), frange(r1, n1), frange(D1, n1)};
        if (r2b) rd.push_back(frange(r2b, n2));
        if (r2c) rd.push_back(frange(r2c, n2));
        if (r1b) rd.push_back(frange(r1b, n1));
        cx.rec->opaque(rd, {frange(t2, n2), frange(t1, n1)},
                       [=](Context& c_) { k_t2_update(c_, t2, r2, r2b, r2c, v_oovv, D2, pp, o, v, t1, r1, r1b, D1); });
        return;
    }
    LAUNCH(t2_update_kernel, dim3(grid_for((int64_t)o * o * v * v)), t2, r2, r2b, r2c, v_oovv, D2, pp, o, v, t1, r1, r1b, D1, r2y);
}
// r2_full(ijab) = r2(ijab) + 1/2 pp(ijab): the residual before P(ia/jb) (tests / get_tensor) -- the reference's tmp_t2 up to terms that are
// held as their images under (i <-> j, a <-> b) (ccsd.hip, z_ooov): r2_full(ijab) + r2_full(jiba) is what equals the reference's
__global__ void r2_full_kernel(double* out, const double* r2, const double* pp, int o, int v)
{
    const int64_t n = (int64_t)o * o * v * v;
    GRID_STRIDE(x, n)
    {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        const int64_t lad = (a <= b) ? i + (int64_t)o * (j + (int64_t)o * ((int64_t)b * (b + 1) / 2 + a))
                                     : j + (int64_t)o * (i + (int64_t)o * ((int64_t)a * (a + 1) / 2 + b));
        out[x] = r2[x] + 0.5 * pp[lad];
    }
}
void k_r2_full(Context& cx, double* out, const double* r2, const double* pp, int o, int v)
{
    LAUNCH(r2_full_kernel, dim3(grid_for((int64_t)o * o * v * v)), out, r2, pp, o, v);
}
// ---- symmetric / antisymmetric form of the particle-particle ladder (ccsd.hip, ccsd_pp_ladder)
// Vs(ef,ab) = 1/2 (<ef|ab> + <fe|ab>) (x 1/2 on e == f) over e <= f, a <= b;  Va(ef,ab) = 1/2 (<ef|ab> - <fe|ab>) over
// e < f, a < b.  Columns (a,b) have leading dimensions ks / ka; built once per calculation.
// Read straight from the packed chemist MO integrals, <ef|ab> = (ea|fb) (ccsd.f90:496-512): a large system never forms the v^4
// slice <ef|ab> itself (12.8 GB and a 13 ms gather at v = 200; ccsd_need_vvvv builds it on request).
__global__ void vvvv_sympack_packed_kernel(double* vs, double* va, const double* packed, int o, int v, int64_t ks, int64_t ka)
{
    const int64_t V = v, nps = V * (V + 1) / 2, n = V * V * nps;
    GRID_STRIDE(x, n)
    {
        const int e = (int)(x % V), f = (int)((x / V) % V);
        if (e > f) continue;
        const int64_t mp = x / (V * V);
        int a, b;
        unpair(mp, a, b);
        const double p = packed[tri(tri(o + e, o + a), tri(o + f, o + b))], q = packed[tri(tri(o + f, o + a), tri(o + e, o + b))];
        vs[(int64_t)f * (f + 1) / 2 + e + ks * mp] = (e == f ? 0.25 : 0.5) * (p + q);
        if (va && e < f && a < b) va[(int64_t)f * (f - 1) / 2 + e + ka * ((int64_t)b * (b - 1) / 2 + a)] = 0.5 * (p - q);
    }
}
// cs(ij,ef) = c(ijef) + c(ijfe) over i <= j, e <= f;  ca(ij,ef) = c(ijef) - c(ijfe) over i < j, e < f; leading dimensions ns / na
__global__ void c_sympack_kernel(double* cs, double* ca, const double* c, int o, int v, int64_t ns, int64_t na, int half)
{
    const int64_t O = o, V = v, n = O * O * V * V;
    GRID_STRIDE(x, n)
    {
        const int i = (int)(x % O);
        int64_t r = x / O;
        const int j = (int)(r % O);
        r /= O;
        const int e = (int)(r % V), f = (int)(r / V);
        if (i > j || e > f) continue;
        const double p = c[x], q = c[i + O * (j + O * (f + V * e))];
        const double ws = half ? (e == f ? 0.25 : 0.5) : 1.0, wa = half ? 0.5 : 1.0;
        cs[(int64_t)j * (j + 1) / 2 + i + ns * ((int64_t)f * (f + 1) / 2 + e)] = ws * (p + q);
        if (ca && i < j && e < f) ca[(int64_t)j * (j - 1) / 2 + i + na * ((int64_t)f * (f - 1) / 2 + e)] = wa * (p - q);
    }
}
__global__ void oooo_pair_expand_add_kernel(double* I, const double* xs, const double* xa, int o, int64_t ns, int64_t na)
{
    const int64_t O = o, n = O * O * O * O;
    GRID_STRIDE(x, n)
    {
        const int k = (int)(x % O), l = (int)((x / O) % O), i = (int)((x / (O * O)) % O), j = (int)(x / (O * O * O));
        const int kl_lo = k < l ? k : l, kl_hi = k < l ? l : k, ij_lo = i < j ? i : j, ij_hi = i < j ? j : i;
        double val = xs[(int64_t)kl_hi * (kl_hi + 1) / 2 + kl_lo + ns * ((int64_t)ij_hi * (ij_hi + 1) / 2 + ij_lo)];
        if (xa && k != l && i != j) {
            const double w = xa[(int64_t)kl_hi * (kl_hi - 1) / 2 + kl_lo + na * ((int64_t)ij_hi * (ij_hi - 1) / 2 + ij_lo)];
            val += ((k < l) == (i < j)) ? w : -w;
        }
        I[x] += val;
    }
}
__global__ void oooo_sympack_kernel(double* is, double* ia, const double* I, int o, int64_t ns, int64_t na)
{
    const int64_t O = o, n = O * O * O * O;
    GRID_STRIDE(x, n)
    {
        const int i = (int)(x % O), j = (int)((x / O) % O), m = (int)((x / (O * O)) % O), nn = (int)(x / (O * O * O));
        if (i > j || m > nn) continue;
        const double p = I[x], q = I[i + O * (j + O * (nn + O * m))];
        is[(int64_t)j * (j + 1) / 2 + i + ns * ((int64_t)nn * (nn + 1) / 2 + m)] = (m == nn ? 0.25 : 0.5) * (p + q);
        if (ia && i < j && m < nn) ia[(int64_t)j * (j - 1) / 2 + i + na * ((int64_t)nn * (nn - 1) / 2 + m)] = 0.5 * (p - q);
    }
}
__global__ void rows_append_kernel(double* dst, int64_t ld, int64_t k0, const double* src, int64_t ns, int64_t n)
{
    GRID_STRIDE(x, n)
    {
        const int64_t r = x % ns, col = x / ns;
        dst[k0 + r + ld * col] = src[x];
    }
}
// PP(i,j,p) = Ps(ij,p) +/- Pa(ij,p): + for i < j, - for i > j (p over a <= b; Pa vanishes on i == j and on a == b)
__global__ void pp_expand_kernel(double* pp, const double* ps, const double* pa, int o, int v, int64_t ns, int64_t na, int64_t p0, int64_t p1)
{
    const int64_t O = o, n = O * O * (p1 - p0), x0 = O * O * p0;
    GRID_STRIDE(xr, n)
    {
        const int64_t x = x0 + xr;
        const int i = (int)(x % O), j = (int)((x / O) % O);
        const int64_t p = x / (O * O);
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        double val = ps[(int64_t)hi * (hi + 1) / 2 + lo + ns * p];
        if (i != j) {
            int a, b;
            unpair(p, a, b);
            if (a != b) {
                const double w = pa[(int64_t)hi * (hi - 1) / 2 + lo + na * ((int64_t)b * (b - 1) / 2 + a)];
                val += i < j ? w : -w;
            }
        }
        pp[x] = val;
    }
}
// Same split of the leading pair (e,f) for a tensor x(e,f,col) whose columns carry no pair structure (v_vvov: col = (i,a)):
// ws(ef,col) = 1/2 (x(e,f,col) + x(f,e,col)) (x 1/2 on e == f) over e <= f,  wa(ef,col) = 1/2 (x(e,f,col) - x(f,e,col)) over e < f
__global__ void vvx_sympack_kernel(double* ws, double* wa, const double* x, int v, int64_t ncol, int64_t ks, int64_t ka)
{
    const int64_t V = v, n = V * V * ncol;
    GRID_STRIDE(t, n)
    {
        const int e = (int)(t % V), f = (int)((t / V) % V);
        if (e > f) continue;
        const int64_t col = t / (V * V);
        const double p = x[t], q = x[f + V * e + V * V * col];
        ws[(int64_t)f * (f + 1) / 2 + e + ks * col] = (e == f ? 0.25 : 0.5) * (p + q);
        if (wa && e < f) wa[(int64_t)f * (f - 1) / 2 + e + ka * col] = 0.5 * (p - q);
    }
}
// out(j,k,col) += Ts(jk,col) +/- Ta(jk,col): + for j < k, - for j > k  (the pair-form product of t2 with v_vvov, ccsd.hip)
__global__ void pair_expand_add_kernel(double* out, const double* ps, const double* pa, int o, int64_t ncol, int64_t ns, int64_t na)
{
    const int64_t O = o, n = O * O * ncol;
    GRID_STRIDE(x, n)
    {
        const int j = (int)(x % O), k = (int)((x / O) % O);
        const int64_t col = x / (O * O);
        const int lo = j < k ? j : k, hi = j < k ? k : j;
        double val = ps[(int64_t)hi * (hi + 1) / 2 + lo + ns * col];
        if (j != k && pa) {
            const double w = pa[(int64_t)hi * (hi - 1) / 2 + lo + na * col];
            val += j < k ? w : -w;
        }
        out[x] += val;
    }
}
__global__ void ooov_r1_trace_kernel(double* r1x, const double* ps, const double* pa, int o, int v, int64_t ns, int64_t na)
{
    const int64_t n = (int64_t)o * v;
    GRID_STRIDE(x, n)
    {
        const int i = (int)(x % o), a = (int)(x / o);
        double s = 0.0;
        for (int m = 0; m < o; ++m) {
            const int lo = m < i ? m : i, hi = m < i ? i : m;
            const int64_t col = m + (int64_t)o * a;
            // 2 X(m,i,m,a) - X(i,m,m,a) = Ts + 3 sgn Ta, sgn = +1 for m < i, -1 for m > i (X(j,k,.) = Ts + Ta for j < k, Ts - Ta for j > k)
            s += ps[(int64_t)hi * (hi + 1) / 2 + lo + ns * col];
            if (pa && m != i) {
                const double w = pa[(int64_t)hi * (hi - 1) / 2 + lo + na * col];
                s += m < i ? 3.0 * w : -3.0 * w;
            }
        }
        r1x[x] = s;
    }
}
void k_ooov_r1_trace(Context& cx, double* r1x, const double* ps, const double* pa, int o, int v, int64_t ns, int64_t na)
{
    if (cx.rec) throw Error(2, "k_ooov_r1_trace: not part of a recorded sequence");
    LAUNCH(ooov_r1_trace_kernel, dim3(grid_for((int64_t)o * v)), r1x, ps, pa, o, v, ns, na);
}
void k_vvx_sympack(Context& cx, double* ws, double* wa, const double* x, int v, int64_t ncol, int64_t ks, int64_t ka)
{
    LAUNCH(vvx_sympack_kernel, dim3(grid_for((int64_t)v * v * ncol, 65536)), ws, wa, x, v, ncol, ks, ka);
}
void k_pair_expand_add(Context& cx, double* out, const double* ps, const double* pa, int o, int64_t ncol, int64_t ns, int64_t na)
{
    if (cx.rec) {
        std::vector<FusedRange> rd = {frange(out, (int64_t)o * o * ncol), frange(ps, ns * ncol)};
        if (pa) rd.push_back(frange(pa, na * ncol));
        cx.rec->opaque(rd, {frange(out, (int64_t)o * o * ncol)}, [=](Context& c_) { k_pair_expand_add(c_, out, ps, pa, o, ncol, ns, na); });
        return;
    }
    LAUNCH(pair_expand_add_kernel, dim3(grid_for((int64_t)o * o * ncol)), out, ps, pa, o, ncol, ns, na);
}
void k_vvvv_sympack_packed(Context& cx, double* vs, double* va, const double* packed, int o, int v, int64_t ks, int64_t ka)
{
    LAUNCH(vvvv_sympack_packed_kernel, dim3(grid_for((int64_t)v * v * ((int64_t)v * (v + 1) / 2), 65536)), vs, va, packed, o, v, ks, ka);
}
void k_oooo_pair_expand_add(Context& cx, double* I, const double* xs, const double* xa, int o, int64_t ns, int64_t na)
{
    if (cx.rec) throw Error(2, "k_oooo_pair_expand_add: not part of a recorded sequence");
    LAUNCH(oooo_pair_expand_add_kernel, dim3(grid_for((int64_t)o * o * o * o)), I, xs, xa, o, ns, na);
}
void k_oooo_sympack(Context& cx, double* is, double* ia, const double* I, int o, int64_t ns, int64_t na)
{
    if (cx.rec) throw Error(2, "k_oooo_sympack: not part of a recorded sequence");
    LAUNCH(oooo_sympack_kernel, dim3(grid_for((int64_t)o * o * o * o)), is, ia, I, o, ns, na);
}
void k_rows_append(Context& cx, double* dst, int64_t ld, int64_t k0, const double* src, int64_t ns, int64_t ncol)
{
    if (cx.rec) throw Error(2, "k_rows_append: not part of a recorded sequence");
    if (ns * ncol > 0) LAUNCH(rows_append_kernel, dim3(grid_for(ns * ncol)), dst, ld, k0, src, ns, ns * ncol);
}
void k_c_sympack(Context& cx, double* cs, double* ca, const double* c, int o, int v, int64_t ns, int64_t na, bool half)
{
    if (cx.rec) {
        if (half) throw Error(2, "k_c_sympack: the weighted form is not part of a recorded sequence");
        const int64_t np = (int64_t)v * (v + 1) / 2, npa = (int64_t)v * (v - 1) / 2, ks = (np + 1) & ~(int64_t)1, ka = (npa + 1) & ~(int64_t)1;
        std::vector<FusedRange> wr = {frange(cs, ns * ks)};
        if (ca) wr.push_back(frange(ca, na * ka));
        cx.rec->opaque({frange(c, (int64_t)o * o * v * v)}, wr, [=](Context& c_) { k_c_sympack(c_, cs, ca, c, o, v, ns, na); });
        return;
    }
    LAUNCH(c_sympack_kernel, dim3(grid_for((int64_t)o * o * v * v)), cs, ca, c, o, v, ns, na, half ? 1 : 0);
}
void k_pp_expand(Context& cx, double* pp, const double* ps, const double* pa, int o, int v, int64_t ns, int64_t na, int64_t p0, int64_t p1)
{
    if (p1 < 0) p1 = (int64_t)v * (v + 1) / 2;
    if (p1 <= p0) return;
    if (cx.rec) {
        std::vector<FusedRange> rd = {frange(ps, ns * p1)};
        if (pa) rd.push_back(frange(pa, na * std::max<int64_t>(1, p1)));
        cx.rec->opaque(rd, {frange(pp + (int64_t)o * o * p0, (int64_t)o * o * (p1 - p0))},
                       [=](Context& c_) { k_pp_expand(c_, pp, ps, pa, o, v, ns, na, p0, p1); });
        return;
    }
    LAUNCH(pp_expand_kernel, dim3(grid_for((int64_t)o * o * (p1 - p0))), pp, ps, pa, o, v, ns, na, p0, p1);
}
// update_diis_cc (ccsd.f90:653-673) without leaving the device: the new row/column `slot` of the error overlap matrix comes
// from `dots`, and the (n+1) x (n+1) system [B -1; -1 0] c = (0,...,0,-1) is solved by Gaussian elimination with partial
// pivoting (the reference calls dsysv, linalg.fpp:38-56; the matrix is at most 16 x 16) by ONE WAVE: lane j holds column j of
// the augmented matrix [A | rhs] in registers (all row indices are compile-time constants: the loops are fully unrolled), pivot
// column and multipliers travel by lane broadcasts.  The coefficients stay in HBM for lincomb_kernel.  flag[0] is set to 1
// when a pivot vanishes (reported by the next energy evaluation).
constexpr int DIIS_MAXN = 17;
// (wave 0 of the block; dots[0..n) in LDS; returns true when a pivot vanishes)
__device__ __forceinline__ bool diis_solve_wave(double* coef, double* bmat, const double* dots, int n, int nerr, int slot)
{
    const int lane = threadIdx.x, N = n + 1;   // columns 0..n of A, column N = right-hand side
    double col[DIIS_MAXN];
#pragma unroll
    for (int i = 0; i < DIIS_MAXN; ++i) {
        double v = 0.0;
        if (lane < n && i < n) v = i == slot ? dots[lane] : lane == slot ? dots[i] : bmat[i + nerr * lane];
        else if (lane < n && i == n) v = -1.0;
        else if (lane == n && i < n) v = -1.0;
        else if (lane == N && i == n) v = -1.0;
        col[i] = v;
    }
    if (lane < n) bmat[slot + nerr * lane] = bmat[lane + nerr * slot] = dots[lane];
    // Column k of the matrix lives in lane k, and k is a compile-time constant in the unrolled loops below: its elements are read
    // with v_readlane (a few cycles, the value uniform in scalar registers) -- a __shfl is a ds_bpermute, ~100 cycles of latency,
    // and the elimination is a chain of ~N^2 of them (the single-wave solve took 18-20 us with shuffles).
    bool singular = false;
#pragma unroll
    for (int k = 0; k < DIIS_MAXN; ++k) {
        if (k < N && !singular) {
            double colk[DIIS_MAXN];   // column k from row k down, before the row exchange
            double big = -1.0;
            int p = k;
#pragma unroll
            for (int i = k; i < DIIS_MAXN; ++i) {
                if (i < N) {   // (uniform: rows beyond the system cost nothing)
                    colk[i] = lane_value(col[i], k);
                    const double v = fabs(colk[i]);
                    if (v > big) { big = v; p = i; }
                }
            }
            singular = big == 0.0;
            double pivot = colk[k];
            if (p != k) {
                const double t = col[k];
#pragma unroll
                for (int i = k + 1; i < DIIS_MAXN; ++i)
                    if (i == p) { col[k] = col[i]; col[i] = t; pivot = colk[i]; colk[i] = colk[k]; }
            }
            const double rkk = singular ? 0.0 : 1.0 / pivot;
#pragma unroll
            for (int i = k + 1; i < DIIS_MAXN; ++i) {
                if (i < N) {
                    const double f = colk[i] * rkk;
                    col[i] -= f * col[k];
                }
            }
        }
    }
    if (singular) return true;
    // back substitution, column by column, on the right-hand side held uniformly: x_k = r_k / U(k,k), r_i -= U(i,k) x_k for i < k
    double r[DIIS_MAXN];
#pragma unroll
    for (int i = 0; i < DIIS_MAXN; ++i) r[i] = i < N ? lane_value(col[i], N) : 0.0;
    double xj = 0.0;
#pragma unroll
    for (int k = DIIS_MAXN - 1; k >= 0; --k) {
        if (k < N) {
            const double xk = r[k] / lane_value(col[k], k);
#pragma unroll
            for (int i = 0; i < k; ++i) r[i] -= lane_value(col[i], k) * xk;
            if (lane == k) xj = xk;
        }
    }
    if (lane < n) coef[lane] = xj;
    return false;
}
__global__ __launch_bounds__(TB) void diis_solve_kernel(double* coef, double* bmat, const double* partial, double* flag, int n, int nerr, int slot)
{
    // the new row of B first: dots[q] = sum_b partial[q*RED_BLOCKS + b], one wave per q (fixed order), all q side by side
    __shared__ double dots[DIIS_MAXN];
    for (int q = threadIdx.x >> 6; q < n; q += TB / 64) {
        double acc = 0.0;
        for (int b = threadIdx.x & 63; b < RED_BLOCKS; b += 64) acc += partial[q * RED_BLOCKS + b];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) dots[q] = acc;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (diis_solve_wave(coef, bmat, dots, n, nerr, slot) && threadIdx.x == 0) flag[0] = 1.0;
}

// ---- the tail of a small system's iteration in two launches (launch-fused path, fused.h) instead of six and a blocking copy:
// cc_tail_kernel      P(ia/jb) + Jacobi division (t2_update_kernel), the energy and rms sums (cc_energy_kernel) and the first half of the
//                     DIIS update (diis_push_kernel) in ONE pass over the amplitudes: the new t1 enters the energy as r1 / D1, the very
//                     quotient that is stored, so nothing waits for another thread's store;
// cc_finalize_kernel  the ordered sums of the partials, the DIIS solve (speculative: the caller decides afterwards whether it
//                     extrapolates), and the results written straight into pinned host memory, sequence number last -- the host
//                     polls that word instead of paying a copy and a stream synchronisation (~40 us) per iteration.
struct TailArgs {
    double *t2, *t1;
    const double *r2, *r1, *voovv, *D2, *D1, *pp;
    double* t2_old;
    int o, v;
    double *ht, *he;               // DIIS history rows of this iteration ([t1 ; t2] vectors); unused when ny == 0
    const double *amp_s, *hist_e;
    int64_t stride;
    int ny, slot;
    const double* r2y;             // YSW: a partial residual held with i and j exchanged (large systems, ring.hip)
    int half;                      // YSW: the DIIS overlaps over a <= b only (every history vector is symmetric, CCState::hist_plain == 0)
};
// NYT = the history rows a thread reads per element (ny rounded up to 0 / 4 / 8 / 16): rows past ny are clamped duplicates
template <int NYT, bool YSW = false>
__global__ __launch_bounds__(TB) void cc_tail_kernel(double* partial, TailArgs p)
{
    __shared__ double sm[18 * 4];
    const int o = p.o, v = p.v, ny = p.ny, slot = p.slot;
    const int64_t n = (int64_t)o * o * v * v, n1 = (int64_t)o * v;
    double acc[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) acc[j] = 0.0;
    const int nyc = ny ? ny - 1 : 0;
    // t2 part.  Every load of an element is issued before anything is computed from one (a load behind a use costs a memory
    // round trip of its own).
    GRID_STRIDE(x, n)
    {
        const int i = (int)(x % o);
        int64_t r = x / o;
        const int j = (int)(r % o);
        r /= o;
        const int a = (int)(r % v), b = (int)(r / v);
        const int64_t y = j + (int64_t)o * (i + (int64_t)o * (b + (int64_t)v * a));
        const int64_t lad = (a <= b) ? i + (int64_t)o * (j + (int64_t)o * ((int64_t)b * (b + 1) / 2 + a))
                                     : j + (int64_t)o * (i + (int64_t)o * ((int64_t)a * (a + 1) / 2 + b));
        double r2x = p.r2[x], r2y = p.r2[y];
        const double ppv = p.pp[lad], vx0 = p.voovv[x], d2 = p.D2[x];
        if (YSW) {   // (i and j exchanged: the same 8 o^2 bytes of memory as x resp. y)
            r2x += p.r2y[j + (int64_t)o * (i + (int64_t)o * (a + (int64_t)v * b))];
            r2y += p.r2y[i + (int64_t)o * (j + (int64_t)o * (b + (int64_t)v * a))];
        }
        const double ria = p.r1[i + o * a], dia = p.D1[i + o * a], rjb = p.r1[j + o * b], djb = p.D1[j + o * b];
        const double vx = p.voovv[i + (int64_t)o * (j + (int64_t)o * (b + (int64_t)v * a))];
        const double told = p.t2_old[x];
        double as2 = 0.0, h[NYT > 0 ? NYT : 1];
        // (large systems, YSW: the error vectors have the symmetry of the amplitudes, e(i,j,a,b) = e(j,i,b,a) to the bit -- both are the same
        // sums of the same numbers -- so their overlaps are summed over a <= b only, the elements a < b twice: b is the slowest index, the
        // elements a <= b of a history vector are contiguous runs, and half of its bytes are not read (0.5 GB of 2.2 at eight vectors))
        // (only while every vector of the history is one the solver itself made: an amplitude set handed in from outside need not have
        // the symmetry, and its error vector stays in the history for nerr iterations -- p.half is 0 for those, CCState::hist_plain)
        const bool half = YSW && p.half;
        const bool hw = !half || a <= b;
        if (NYT > 0) {
            as2 = p.amp_s[n1 + x];
            if (hw) {
#pragma unroll
                for (int q = 0; q < NYT; ++q) h[q] = p.hist_e[(int64_t)min(q, nyc) * p.stride + n1 + x];
            }
        }
        const double t = (r2x + r2y + ppv + vx0) / d2;
        const double tia = ria / dia, tjb = rjb / djb;
        acc[16] += (2.0 * vx0 - vx) * (t + tia * tjb);
        const double d = t - told;
        acc[17] += d * d;
        p.t2_old[x] = t;
        p.t2[x] = t;
        if (NYT > 0) {
            const double e = t - as2;
            p.ht[n1 + x] = t;
            p.he[n1 + x] = e;
            if (hw) {
                const double ew = (half && a < b) ? 2.0 * e : e;
#pragma unroll
                for (int q = 0; q < NYT; ++q) acc[q] += q < ny ? ew * (q == slot ? e : h[q]) : 0.0;
            }
        }
    }
    // t1 part: o v elements, the first blocks only
    GRID_STRIDE(x, n1)
    {
        const double r1x = p.r1[x], d1x = p.D1[x];
        double as1 = 0.0, h[NYT > 0 ? NYT : 1];
        if (NYT > 0) {
            as1 = p.amp_s[x];
#pragma unroll
            for (int q = 0; q < NYT; ++q) h[q] = p.hist_e[(int64_t)min(q, nyc) * p.stride + x];
        }
        const double t1v = r1x / d1x;
        p.t1[x] = t1v;
        if (NYT > 0) {
            const double e = t1v - as1;
            p.ht[x] = t1v;
            p.he[x] = e;
#pragma unroll
            for (int q = 0; q < NYT; ++q) acc[q] += q < ny ? e * (q == slot ? e : h[q]) : 0.0;
        }
    }
    block_sum<18>(acc, sm);
    if (threadIdx.x == 0) {
        for (int q = 0; q < ny; ++q) partial[q * RED_BLOCKS + blockIdx.x] = acc[q];
        partial[16 * RED_BLOCKS + blockIdx.x] = acc[16];
        partial[17 * RED_BLOCKS + blockIdx.x] = acc[17];
    }
}
__global__ __launch_bounds__(TB) void cc_finalize_kernel(double* out2, double* host_res, double seq, double* bmat, const double* partial, int nblk,
                                                         int n, int nerr, int slot)
{
    __shared__ double dots[DIIS_MAXN + 2];
    for (int q = threadIdx.x >> 6; q < n + 2; q += TB / 64) {
        const int row = q < n ? q : 16 + (q - n);
        double acc = 0.0;
        for (int b = threadIdx.x & 63; b < nblk; b += 64) acc += partial[row * RED_BLOCKS + b];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) dots[q] = acc;
    }
    __syncthreads();
    // row / column `slot` of the error overlap matrix (ccsd.f90:653-663) on the device, and the whole matrix for the host, which
    // solves the <= 17 x 17 system itself (as the reference does, linalg.fpp:38-56) while this stream goes on
    if ((int)threadIdx.x < n) bmat[slot + nerr * threadIdx.x] = bmat[threadIdx.x + nerr * slot] = dots[threadIdx.x];
    __syncthreads();
    for (int x = threadIdx.x; x < n * n; x += TB) {
        const int i = x % n, j = x / n;
        host_res[8 + i + 16 * j] = (i == slot) ? dots[j] : (j == slot) ? dots[i] : bmat[i + nerr * j];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = dots[n];
        out2[1] = dots[n + 1];
        host_res[0] = dots[n];
        host_res[1] = dots[n + 1];
        __threadfence_system();
        __hip_atomic_store(&host_res[3], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
struct Coefs {
    double c[16];
};
__global__ void lincomb_vals_kernel(double* out, const double* xbase, int64_t xstride, Coefs coef, int nx, int64_t n)
{
    GRID_STRIDE(i, n)
    {
        double s = 0.0;
        for (int j = 0; j < nx; ++j) s += coef.c[j] * xbase[j * xstride + i];
        out[i] = s;
    }
}
void k_lincomb_vals(Context& cx, double* out, const double* xbase, int64_t xstride, const double* coef_host, int nx, int64_t n)
{
    if (nx > 16) throw Error(3, "k_lincomb_vals: too many vectors");
    Coefs c;
    for (int j = 0; j < 16; ++j) c.c[j] = j < nx ? coef_host[j] : 0.0;
    LAUNCH(lincomb_vals_kernel, dim3(grid_for(n)), out, xbase, xstride, c, nx, n);
}
void k_diis_solve(Context& cx, double* coef, double* bmat, double* flag, int n, int nerr, int slot)
{
    AFESP_KLAUNCH(diis_solve_kernel, dim3(1), dim3(TB), 0, cx.stream, coef, bmat, cx.scal + 64, flag, n, nerr, slot);
    AFESP_HIP(hipGetLastError());
}
void k_diis_push(Context& cx, double* ht, double* he, const double* amp, const double* amp_s, const double* hist_e, int64_t stride, int ny,
                 int slot, int64_t n)
{
    if (ny > 16) throw Error(3, "k_diis_push: too many vectors");
    LAUNCH(diis_push_kernel, dim3(RED_BLOCKS), cx.scal + 64, ht, he, amp, amp_s, hist_e, stride, ny, slot, n);
}
void k_denominators(Context& cx, double* D1, double* D2, const double* e, int o, int v)
{
    LAUNCH(denominators_kernel, dim3(grid_for((int64_t)o * o * v * v)), D1, D2, e, o, v);
}
// partial sums live at cx.scal + 64 (2*RED_BLOCKS doubles reserved by the context)
static double* partials(Context& cx) { return cx.scal + 64; }

void k_cc_energy(Context& cx, double* out2, const double* v_oovv, const double* t1, const double* t2, double* t2_old, int o, int v)
{
    if (cx.rec) {
        const int64_t n2 = (int64_t)o * o * v * v;
        cx.rec->opaque({frange(v_oovv, n2), frange(t1, (int64_t)o * v), frange(t2, n2), frange(t2_old, n2)},
                       {frange(t2_old, n2), frange(out2, 2), frange(partials(cx), 2 * RED_BLOCKS)},
                       [=](Context& c_) { k_cc_energy(c_, out2, v_oovv, t1, t2, t2_old, o, v); }, 2);
        return;
    }
    LAUNCH(cc_energy_kernel, dim3(RED_BLOCKS), partials(cx), v_oovv, t1, t2, t2_old, o, v);
    LAUNCH(final_sum_kernel, dim3(1), out2, partials(cx), RED_BLOCKS, 2, 0);
}
void k_cc_tail(Context& cx, const CCTail& a)
{
    if (a.ny > 16) throw Error(3, "k_cc_tail: too many DIIS vectors");
    TailArgs p;
    p.t2 = a.t2; p.t1 = a.t1; p.r2 = a.r2; p.r1 = a.r1; p.voovv = a.voovv; p.D2 = a.D2; p.D1 = a.D1; p.pp = a.pp; p.t2_old = a.t2_old;
    p.o = a.o; p.v = a.v; p.ht = a.ht; p.he = a.he; p.amp_s = a.amp_s; p.hist_e = a.hist_e; p.stride = a.stride; p.ny = a.ny; p.slot = a.slot;
    p.r2y = a.r2y;
    p.half = a.half_hist ? 1 : 0;
    const int nblk = (int)grid_for((int64_t)a.o * a.o * a.v * a.v, RED_BLOCKS);   // (only blocks that have elements write partials)
    if (a.r2y) {
        if (a.ny == 0) LAUNCH((cc_tail_kernel<0, true>), dim3(nblk), partials(cx), p);
        else if (a.ny <= 4) LAUNCH((cc_tail_kernel<4, true>), dim3(nblk), partials(cx), p);
        else if (a.ny <= 8) LAUNCH((cc_tail_kernel<8, true>), dim3(nblk), partials(cx), p);
        else LAUNCH((cc_tail_kernel<16, true>), dim3(nblk), partials(cx), p);
    } else
    if (a.ny == 0) LAUNCH(cc_tail_kernel<0>, dim3(nblk), partials(cx), p);
    else if (a.ny <= 4) LAUNCH(cc_tail_kernel<4>, dim3(nblk), partials(cx), p);
    else if (a.ny <= 8) LAUNCH(cc_tail_kernel<8>, dim3(nblk), partials(cx), p);
    else LAUNCH(cc_tail_kernel<16>, dim3(nblk), partials(cx), p);
    LAUNCH(cc_finalize_kernel, dim3(1), cx.scal, cx.res_dev, (double)a.seq, a.bmat, partials(cx), nblk, a.ny, a.nerr, a.slot);
}
// E(MP2) of the packed MO integrals on the host; e_dev: the n orbital energies on the device
double k_mp2_packed(Context& cx, const double* eri_packed, const double* e_host, int o, int v)
{
    double seq = 0.0;
    double* pub = host_scalars_slot(cx, &seq);
    unsigned* counter = reinterpret_cast<unsigned*>(cx.scal + 56);   // (zero between launches: atomicInc wraps)
    const int nblk = (int)grid_for((int64_t)o * o * v * v, RED_BLOCKS);
    Mp2Levels lv;
    if (o + v <= 256) {
        for (int q = 0; q < o + v; ++q) lv.e[q] = e_host[q];
        LAUNCH(mp2_packed_kernel<true>, dim3(nblk), partials(cx), counter, cx.scal, pub, seq, eri_packed, (const double*)nullptr, lv, o, v);
    } else {
        double* e_dev = cx.scratch("ao2mo_e", o + v);
        AFESP_HIP(hipMemcpyAsync(e_dev, e_host, sizeof(double) * (o + v), hipMemcpyHostToDevice, cx.stream));
        LAUNCH(mp2_packed_kernel<false>, dim3(nblk), partials(cx), counter, cx.scal, pub, seq, eri_packed, e_dev, lv, o, v);
    }
    return pub ? host_scalars_wait(cx, 1, seq)[0] : host_scalars(cx, 1)[0];
}
void k_mp2_energy(Context& cx, double* out1, const double* v_oovv, const double* D2, int o, int v)
{
    LAUNCH(mp2_energy_kernel, dim3(RED_BLOCKS), partials(cx), v_oovv, D2, o, v);
    LAUNCH(final_sum_kernel, dim3(1), out1, partials(cx), RED_BLOCKS, 1, 0);
}
void k_dots(Context& cx, double* out, const double* x, const double* ybase, int64_t ystride, int ny, int64_t n, bool accumulate)
{
    if (ny <= 0) return;
    if (ny > 16) throw Error(3, "k_dots: too many vectors");
    LAUNCH(dots_kernel, dim3(RED_BLOCKS, ny), partials(cx), x, ybase, ystride, n);
    LAUNCH(final_sum_kernel, dim3(1), out, partials(cx), RED_BLOCKS, ny, accumulate ? 1 : 0);
}
void k_lincomb(Context& cx, double* out, const double* xbase, int64_t xstride, const double* coef_dev, int nx, int64_t n)
{
    if (n > 0) LAUNCH(lincomb_kernel, dim3(grid_for(n)), out, xbase, xstride, coef_dev, nx, n);
}
// ---- pair-symmetric AO->MO (capi.hip, afesp_ao2mo_mp2): the three layout steps between the quarter transforms
// Both steps have the shape  out(x,y,C) = src(C, tri(x,y)):  a pair index is squared up into the two leading (fastest)
// indices of the result while the other pair index C moves from fastest (in src) to slowest.  A workgroup stages a
// 16 x 16 x 16 tile through LDS so that both the reads (16 consecutive C, or 16 consecutive members of the packed pair)
// and the writes (16 consecutive x) are 128-byte runs; the tile and its mirror image (x and y exchanged) come from one read.
//   MODE 0  unpack_half:     out(i,j,KL) = packed[tri(tri(i,j), KL)]          (ij|kl) with ij squared up, for every pair KL
//   MODE 1  pair_transpose:  out(k,l,PQ) = in(q,p,tri(k,l)), PQ = tri(p,q)    (pq|kl) -> (kl|PQ), kl squared up, p >= q
//   MODE 2  out(k,l,P) = g(P, tri(k,l)), g a plain [np x np] array         the same from the pair-packed half-transformed integrals
// The C blocks [c_begin, c_end) of the result are produced (c_begin a multiple of 16), at out(x,y,C - c_begin): the blocked
// transform of afesp_ao2mo_mp2 works on slabs of C.
// ld >= n: leading dimension of `out` (and of MODE 1's source): the LDS-DMA transforms of a basis size that is no multiple of 16 keep
// their temporaries with columns of ld = 16 ceil(n / 16) doubles, so that every column -- a 128-byte line per K step of the GEMM,
// a 128-byte run of this kernel -- starts on a line (round 6; n = 220: 39.2 -> 36 ms per transform).
template <int MODE>
__global__ __launch_bounds__(256) void pair_square_kernel(double* out, const double* src, int n, int64_t c_begin, int64_t c_end, int ld)
{
    constexpr int T = 16, TP = T + 1, SC = T * TP + 3;   // rows padded: the mirrored tile is read out of LDS along y
    __shared__ double tile[T * SC];
    const int64_t N = n, np = N * (N + 1) / 2, L = ld;
    const int nb = (n + T - 1) / T, nbp = nb * (nb + 1) / 2;
    // a workgroup owns the tile pair (x-block xb >= y-block yb) of one C block: src(C, tri(x,y)) is read once and written
    // to out(x,y,C) and to its mirror image out(y,x,C)
    const int64_t cb = (int64_t)blockIdx.x / nbp;
    int yb, xb;
    unpair((int64_t)blockIdx.x % nbp, yb, xb);
    const int x0 = xb * T, y0 = yb * T;
    const int64_t c0 = c_begin + cb * T;
    // which tile direction is contiguous in src: C (dir 0) or y (dir 1: the whole tile lies in rows C of the packed triangle,
    // where the members x >= y of a pair run along y)
    int dir = 0;
    if (MODE == 0 && tri(min(x0 + T, n) - 1, min(y0 + T, n) - 1) <= c0) dir = 1;
    const int lane = threadIdx.x % T, row = threadIdx.x / T;
    int64_t pq_off = 0;
    if (MODE == 1 && c0 + lane < c_end) {
        int q, p;
        unpair(c0 + lane, q, p);
        pq_off = q + L * p;
    }
#pragma unroll 4
    for (int it = 0; it < T; ++it) {
        const int c = dir == 0 ? lane : row, yi = dir == 0 ? it : lane, xi = dir == 0 ? row : it;
        const int X = x0 + xi, Y = y0 + yi;
        if (X < n && Y < n && c0 + c < c_end)
            tile[c * SC + yi * TP + xi] = MODE == 0 ? src[tri(tri(X, Y), c0 + c)]
                                        : MODE == 1 ? src[pq_off + L * N * tri(X, Y)] : src[(c0 + c) + np * tri(X, Y)];
    }
    __syncthreads();
    const int64_t cr = c0 - c_begin;   // position of the block in the slab
    {
        const int X = x0 + lane, Y = y0 + row;          // out(x,y,C): lanes along x
        if (X < n && Y < n) {
#pragma unroll 4
            for (int c = 0; c < T; ++c)
                if (c0 + c < c_end) out[X + L * Y + L * N * (cr + c)] = tile[c * SC + row * TP + lane];
        }
    }
    if (xb != yb) {
        const int X = y0 + lane, Y = x0 + row;          // the mirror image out(y,x,C): lanes along y
        if (X < n && Y < n) {
#pragma unroll 4
            for (int c = 0; c < T; ++c)
                if (c0 + c < c_end) out[X + L * Y + L * N * (cr + c)] = tile[c * SC + lane * TP + row];
        }
    }
}
// packed[tri(PQ,RS)] = full(s,r,PQ - p_begin) for RS = tri(r,s) <= PQ, PQ in [p_begin, p_end)  (mp2.f90:388-410 on the
// pair-packed result; the whole range in one call, or slab by slab)
__global__ void pack_pairs_kernel(double* packed, const double* full, int n, int64_t p_begin, int64_t p_end, int ld)
{
    const int64_t N = n, np = N * (N + 1) / 2, tot = np * (p_end - p_begin), L = ld;
    GRID_STRIDE(x, tot)
    {
        const int64_t rs = x % np, pq = p_begin + x / np;
        if (rs > pq) continue;
        int s_, r_;
        unpair(rs, s_, r_);
        packed[pq * (pq + 1) / 2 + rs] = full[s_ + L * r_ + L * N * (pq - p_begin)];
    }
}
// g(PQ, K) = half(q, p, K - k_begin), PQ = tri(p,q) over p >= q, K in [k_begin, k_end): the half-transformed integrals of a slab
// of (kl) pairs, pair-packed in (pq), into the [np x np] array the second pair of transforms gathers from
__global__ void tri_pack_kernel(double* g, const double* half, int n, int64_t k_begin, int64_t k_end)
{
    const int64_t N = n, np = N * (N + 1) / 2, tot = np * (k_end - k_begin);
    GRID_STRIDE(x, tot)
    {
        const int64_t pq = x % np, k = x / np;
        int q, p;
        unpair(pq, q, p);
        g[pq + np * (k_begin + k)] = half[q + N * p + N * N * k];
    }
}
static unsigned pair_square_grid(int n, int64_t c_begin, int64_t c_end)
{
    const int64_t nb = (n + 15) / 16;
    return (unsigned)(nb * (nb + 1) / 2 * ((c_end - c_begin + 15) / 16));
}
void k_unpack_half(Context& cx, double* u, const double* packed, int n, int64_t c_begin, int64_t c_end, int ld)
{
    const int64_t np = (int64_t)n * (n + 1) / 2;
    if (c_end < 0) c_end = np;
    if (c_end > c_begin) LAUNCH(pair_square_kernel<0>, dim3(pair_square_grid(n, c_begin, c_end)), u, packed, n, c_begin, c_end, ld > 0 ? ld : n);
}
// zeroes rows [n, ld) of every column of x(ld, ncol): the padding of the LDS-DMA transforms' temporaries (read as K padding, times zero)
__global__ __launch_bounds__(256) void pad_rows_zero_kernel(double* x, int n, int ld, int64_t ncol)
{
    const int w = ld - n;
    GRID_STRIDE(i, ncol * w) x[(i / w) * ld + n + (i % w)] = 0.0;
}
void k_pad_rows_zero(Context& cx, double* x, int n, int ld, int64_t ncol)
{
    if (ld > n && ncol > 0) LAUNCH(pad_rows_zero_kernel, dim3(grid_for(ncol * (ld - n), 65536)), x, n, ld, ncol);
}
// ---- both quarter transforms of a pair index in ONE kernel, for bases of up to 64 functions (mp2.f90:321-348 resp. :357-385):
//   out(:, :, S) = C in(:, :, S) C^T   for every pair S, in(:, :, S) symmetric
// A workgroup owns one S: the n x n block goes to LDS once (zero-padded to 64 x 64), T1 = C U is formed by the four waves
// (32 x 32 quadrants, 2 x 2 accumulators of v_mfma_f64_16x16x4_f64), written back over U in the layout the second product reads
// its A fragments in, and T2 = T1 C^T leaves through its transpose -- T2 is symmetric -- so that the lanes of a store run along the
// fastest index.  The coefficient fragments come straight from memory (C is 27 KB at n = 58: L1 / L2 resident), all of them
// requested before the first product starts.  No intermediate touches HBM: the two gather-GEMM launches per pair, their K-slice
// reductions and 2 x 8 n^2 npair bytes of traffic become one launch.
constexpr int PX = 64, PXS = 66;   // padded extent, LDS row stride
// MODE 0: in = u(a, b, S) squares (n x n per pair), out = squares            (the transform between the layout kernels)
// MODE 1: in = the 8-fold packed AO integrals, block S gathered through the packed index; out = pair columns g(PQ, S), p >= q
// MODE 2: in = pair columns h(KL, S);  out = the packed MO integrals, run S: packed[S (S + 1) / 2 + RS], RS <= S
// -- with modes 1 and 2 and one transposition of the npair x npair matrix between them the whole AO->MO transform of a small basis
// is three launches and moves 8 (2 neri + 4 npair^2) bytes: no squared-up copy of the integrals exists at any point.
template <int MODE>
__global__ __launch_bounds__(256, 2) void pair_xform_kernel(double* __restrict__ out, const double* __restrict__ in, const double* __restrict__ C,
                                                            int n)
{
    typedef double v4d_t __attribute__((ext_vector_type(4)));
    __shared__ double S[PX * PXS];
    const int64_t nn = (int64_t)n * n, np = (int64_t)n * (n + 1) / 2, blk = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, lm = lane & 15, lk = lane >> 4;
    // coefficient fragments: as A operand of the first product (row p, k = i) and as B operand of the second (column q, k = j)
    double ca[16][2], cb[16][2];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = min(4 * s + lk, n - 1);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            ca[s][f] = C[min(32 * wm + 16 * f + lm, n - 1) + (int64_t)n * k];
            cb[s][f] = C[min(32 * wn + 16 * f + lm, n - 1) + (int64_t)n * k];
        }
    }
    // U into LDS, zero-padded; (a, b) and (b, a) are the same number, so the lanes run along the fastest index on both sides
    // (clamped addresses, the padding zeroed afterwards: a load under a condition would be waited for on its own, sixteen round
    // trips instead of one)
    double ur[PX * PX / 256];
#pragma unroll
    for (int r = 0; r < PX * PX / 256; ++r) {
        const int e = t + 256 * r, a = min(e & 63, n - 1), b = min(e >> 6, n - 1);
        if (MODE == 0) ur[r] = in[nn * blk + a + (int64_t)n * b];
        else if (MODE == 1) ur[r] = in[tri(tri(a, b), blk)];
        else ur[r] = in[np * blk + tri(a, b)];
    }
#pragma unroll
    for (int r = 0; r < PX * PX / 256; ++r) {
        const int e = t + 256 * r, a = e & 63, b = e >> 6;
        S[b * PXS + a] = (a < n && b < n) ? ur[r] : 0.0;
    }
    __syncthreads();
    v4d_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d_t){0.0, 0.0, 0.0, 0.0};
    // T1(p, j) = sum_i C(p, i) U(i, j)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        double bf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = S[(4 * s + lk) * PXS + 32 * wn + 16 * j + lm];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[s][i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // every wave has read U
    // T1 over U, k-major for the second product's A fragments: S[j][p]  (C/D layout: column = lane & 15, row = (lane >> 4) + 4 r)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(32 * wn + 16 * j + lm) * PXS + 32 * wm + 16 * i + lk + 4 * r] = acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (v4d_t){0.0, 0.0, 0.0, 0.0};
    // T2(p, q) = sum_j T1(p, j) C(q, j); the k >= n rows of T1 are zero (U's padding), so the clamped coefficient rows do no harm
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        double af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = S[(4 * s + lk) * PXS + 32 * wm + 16 * i + lm];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], cb[s][j], acc[i][j], 0, 0, 0);
    }
    // T2 is symmetric: (row, col) of an accumulator is written as element (col, row) so that the lanes of a store run along the
    // fastest index of the destination -- squares: out(q, p); pair columns / packed runs: the pair (p, q) for q <= p
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 32 * wm + 16 * i + lk + 4 * r;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = 32 * wn + 16 * j + lm;
                if (MODE == 0) {
                    if (p < n && q < n) out[nn * blk + q + (int64_t)n * p] = acc[i][j][r];
                } else if (MODE == 1) {
                    if (p < n && q <= p) out[np * blk + (int64_t)p * (p + 1) / 2 + q] = acc[i][j][r];
                } else {
                    const int64_t rs = (int64_t)p * (p + 1) / 2 + q;
                    if (p < n && q <= p && rs <= blk) out[blk * (blk + 1) / 2 + rs] = acc[i][j][r];
                }
            }
        }
}
// out(y, x) = in(x, y), n x n: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void square_transpose_kernel(double* __restrict__ out, const double* __restrict__ in, int64_t n)
{
    __shared__ double tile[32][33];
    const int64_t tiles = (n + 31) / 32, bx = blockIdx.x % tiles, by = blockIdx.x / tiles;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t x = bx * 32 + tx, y = by * 32 + ty + 8 * r;
        if (x < n && y < n) tile[ty + 8 * r][tx] = in[x + n * y];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t y = by * 32 + tx, x = bx * 32 + ty + 8 * r;
        if (x < n && y < n) out[y + n * x] = tile[tx][ty + 8 * r];
    }
}
void k_pair_xform(Context& cx, double* out, const double* in, const double* C, int n, int64_t npairs, int mode)
{
    if (n > PX) throw Error(3, "k_pair_xform: more than 64 basis functions");
    if (npairs <= 0) return;
    if (mode == 0) AFESP_KLAUNCH(pair_xform_kernel<0>, dim3((unsigned)npairs), dim3(256), 0, cx.stream, out, in, C, n);
    else if (mode == 1) AFESP_KLAUNCH(pair_xform_kernel<1>, dim3((unsigned)npairs), dim3(256), 0, cx.stream, out, in, C, n);
    else AFESP_KLAUNCH(pair_xform_kernel<2>, dim3((unsigned)npairs), dim3(256), 0, cx.stream, out, in, C, n);
    AFESP_HIP(hipGetLastError());
}
void k_square_transpose(Context& cx, double* out, const double* in, int64_t n)
{
    const int64_t tiles = (n + 31) / 32;
    if (n > 0) LAUNCH(square_transpose_kernel, dim3((unsigned)(tiles * tiles)), out, in, n);
}
void k_pair_transpose(Context& cx, double* out, const double* in, int n, int ld)
{
    const int64_t np = (int64_t)n * (n + 1) / 2;
    LAUNCH(pair_square_kernel<1>, dim3(pair_square_grid(n, 0, np)), out, in, n, (int64_t)0, np, ld > 0 ? ld : n);
}
void k_pair_square_packed(Context& cx, double* out, const double* g, int n, int64_t c_begin, int64_t c_end)
{
    if (c_end > c_begin) LAUNCH(pair_square_kernel<2>, dim3(pair_square_grid(n, c_begin, c_end)), out, g, n, c_begin, c_end, n);
}
void k_tri_pack(Context& cx, double* g, const double* half, int n, int64_t k_begin, int64_t k_end)
{
    const int64_t np = (int64_t)n * (n + 1) / 2;
    if (k_end > k_begin) LAUNCH(tri_pack_kernel, dim3(grid_for(np * (k_end - k_begin), 65536)), g, half, n, k_begin, k_end);
}
void k_pack_pairs(Context& cx, double* packed, const double* full, int n, int64_t p_begin, int64_t p_end, int ld)
{
    const int64_t np = (int64_t)n * (n + 1) / 2;
    if (p_end < 0) p_end = np;
    if (p_end > p_begin) LAUNCH(pack_pairs_kernel, dim3(grid_for(np * (p_end - p_begin), 65536)), packed, full, n, p_begin, p_end, ld > 0 ? ld : n);
}
void k_slice_phys(Context& cx, double* out, const double* packed, int d0, int d1, int d2, int d3, int b0, int b1, int b2, int b3)
{
    int64_t n = (int64_t)d0 * d1 * d2 * d3;
    if (n > 0) LAUNCH(slice_phys_kernel, dim3(grid_for(n, 65536)), out, packed, d0, d1, d2, d3, b0, b1, b2, b3);
}

void k_build_fock(Context& cx, double* fock, const double* hcore, const double* dens, const double* u, double* work, int n, int ld)
{
    if (ld <= 0) ld = n;
    // work: [ dv (npair) | jpart (FOCK_CHUNKS n^2) | kp1 (npair n) | kp2 (npair n) ]
    const int64_t n2 = (int64_t)n * n, np = (int64_t)n * (n + 1) / 2;
    double *dv = work, *jpart = dv + np, *kp1 = jpart + FOCK_CHUNKS * n2, *kp2 = kp1 + np * n;
    LAUNCH(fock_dv_kernel, dim3(grid_for(np)), dv, dens, n);
    LAUNCH(fock_j_kernel, dim3((unsigned)((n2 + 255) / 256), FOCK_CHUNKS), jpart, u, dv, n, ld);
    AFESP_KLAUNCH(fock_k_kernel, dim3((unsigned)np), dim3(256), 2 * n * sizeof(double), cx.stream, kp1, kp2, u, dens, n, ld);
    AFESP_HIP(hipGetLastError());
    LAUNCH(fock_reduce_kernel, dim3(grid_for(n2)), fock, hcore, jpart, kp1, kp2, n);
}
int64_t k_build_fock_work(int n)
{
    const int64_t n2 = (int64_t)n * n, np = (int64_t)n * (n + 1) / 2;
    return np + FOCK_CHUNKS * n2 + 2 * np * n;
}

}  // namespace afesp

namespace afesp {
// the kernels of a small system's AO->MO transform, iteration tail and set-up: the runtime resolves a kernel function on its first
// use (0.1 - 0.7 ms each) -- three milliseconds of a process's first iteration otherwise
void preload_small_path_kernels()
{
    const void* fns[] = {reinterpret_cast<const void*>(asym_c_kernel), reinterpret_cast<const void*>(c_sympack_kernel),
                         reinterpret_cast<const void*>(denominators_kernel), reinterpret_cast<const void*>(mp2_energy_kernel),
                         reinterpret_cast<const void*>(mp2_packed_kernel<true>),
                         reinterpret_cast<const void*>(cc_energy_kernel), reinterpret_cast<const void*>(final_sum_kernel),
                         reinterpret_cast<const void*>(cc_tail_kernel<0>), reinterpret_cast<const void*>(cc_tail_kernel<4>),
                         reinterpret_cast<const void*>(cc_tail_kernel<8>), reinterpret_cast<const void*>(cc_finalize_kernel),
                         reinterpret_cast<const void*>(lincomb_vals_kernel), reinterpret_cast<const void*>(slice_phys_kernel),
                         reinterpret_cast<const void*>(antisym_pair_kernel), reinterpret_cast<const void*>(pp_expand_kernel),
                         reinterpret_cast<const void*>(pair_expand_add_kernel), reinterpret_cast<const void*>(pair_xform_kernel<1>),
                         reinterpret_cast<const void*>(pair_xform_kernel<2>), reinterpret_cast<const void*>(square_transpose_kernel)};
    for (const void* f : fns) first_use_touch(f);   // (each under the process-wide first-use lock, first_use.h)
}
}  // namespace afesp
