// triples.hip -- perturbative triples (T) of the spin-free formulation, src/ccsd.f90:2018-2293.
//
// The reference visits all o^3 ordered (i,j,k) and builds W^{ijk}(abc) from twelve strided dot products per
// element (:2168-2173).  Writing X^{pqr}(a,b,c) = sum_d t2(p,q,a,d) <cb|rd> - sum_l t2(l,p,b,a) <rq|cl> (the first
// pair of terms), the six pairs are X evaluated at the six simultaneous permutations of (ijk)/(abc):
//     W^{ijk}(abc) = X^{ijk}(abc) + X^{jik}(bac) + X^{kji}(cba) + X^{ikj}(acb) + X^{jki}(bca) + X^{kij}(cab)
// so W^{pi(ijk)}(pi(abc)) = W^{ijk}(abc), and the same holds for D, z and y.  The energy functional summed over
// ALL ordered triples with the reference's x_bar = 4/3 x(abc) - 2 x(acb) + 2/3 x(cab) (:2314-2318) therefore equals
// the sum with the symmetrised x_bar = [4 x(abc) + x(bca) + x(cab) - 2 x(acb) - 2 x(bac) - 2 x(cba)]/3 (the weight
// of a permutation class is what matters once all (ijk) are summed), and that per-triple functional is invariant
// under permuting (ijk).  Here: only i<=j<=k is visited, with multiplicity 6/3/1; each distinct ordered X is ONE
// (v x v) . (v x v^2) MFMA GEMM plus one (v^2 x o) . (o x v) GEMM, batched over a chunk of triples.
#include <algorithm>
#include <cmath>

#include "ccsd.h"

namespace afesp {

int64_t triples_count(int o) { return (int64_t)o * (o + 1) * (o + 2) / 6; }

struct TripleMeta {
    int i, j, k, pad;
    double mult;
    int64_t xoff[6];   // element offsets of X^{ijk}, X^{jik}, X^{kji}, X^{ikj}, X^{jki}, X^{kij} in the X pool
    int64_t woff;
};

// W(a,b,c) = X0(a,b,c) + X1(b,a,c) + X2(c,b,a) + X3(a,c,b) + X4(b,c,a) + X5(c,a,b)      ccsd.f90:2168-2173
__global__ __launch_bounds__(256) void triples_w_kernel(double* __restrict__ Wpool, const double* __restrict__ Xpool,
                                                        const TripleMeta* __restrict__ meta, int v)
{
    const TripleMeta m = meta[blockIdx.y];
    const int64_t v3 = (int64_t)v * v * v;
    double* W = Wpool + m.woff;
    const double* X0 = Xpool + m.xoff[0]; const double* X1 = Xpool + m.xoff[1]; const double* X2 = Xpool + m.xoff[2];
    const double* X3 = Xpool + m.xoff[3]; const double* X4 = Xpool + m.xoff[4]; const double* X5 = Xpool + m.xoff[5];
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < v3; x += (int64_t)gridDim.x * blockDim.x) {
        const int a = (int)(x % v), b = (int)((x / v) % v), c = (int)(x / ((int64_t)v * v));
#define AT(p, q, r) ((p) + (int64_t)v * ((q) + (int64_t)v * (r)))
        W[x] = X0[x] + X1[AT(b, a, c)] + X2[AT(c, b, a)] + X3[AT(a, c, b)] + X4[AT(b, c, a)] + X5[AT(c, a, b)];
    }
}

struct TriplesIn {
    const double* e;
    const double* t1;
    const double* t2;
    const double* v_oovv;
    int o, v;
};

// Per element: D (ccsd.f90:2175), z (:2178-2179), y (:2183-2184), symmetrised bars, four sums (:2218-2233).
__global__ __launch_bounds__(256) void triples_e_kernel(double* __restrict__ partial, const double* __restrict__ Wpool,
                                                        const TripleMeta* __restrict__ meta, TriplesIn in, int nblk_total)
{
    __shared__ double sm[16];
    const TripleMeta m = meta[blockIdx.y];
    const int o = in.o, v = in.v;
    const int64_t v3 = (int64_t)v * v * v;
    const double* W = Wpool + m.woff;
    const int i = m.i, j = m.j, k = m.k;
    const double eo = in.e[i] + in.e[j] + in.e[k];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#define T1(p, x) in.t1[(p) + o * (x)]
#define T2(p, q, x, y) in.t2[(p) + (int64_t)o * ((q) + (int64_t)o * ((x) + (int64_t)v * (y)))]
#define VO(p, q, x, y) in.v_oovv[(p) + (int64_t)o * ((q) + (int64_t)o * ((x) + (int64_t)v * (y)))]
#define ZZ(x, y, z) (T1(i, x) * VO(j, k, y, z) + T1(j, y) * VO(i, k, x, z) + T1(k, z) * VO(i, j, x, y))
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < v3; x += (int64_t)gridDim.x * blockDim.x) {
        const int a = (int)(x % v), b = (int)((x / v) % v), c = (int)(x / ((int64_t)v * v));
        const double D = eo - in.e[a + o] - in.e[b + o] - in.e[c + o];
        const double w = W[x];
        const double wb = (4.0 * w + W[AT(b, c, a)] + W[AT(c, a, b)] - 2.0 * (W[AT(a, c, b)] + W[AT(b, a, c)] + W[AT(c, b, a)])) / 3.0;
        const double zb = (4.0 * ZZ(a, b, c) + ZZ(b, c, a) + ZZ(c, a, b) - 2.0 * (ZZ(a, c, b) + ZZ(b, a, c) + ZZ(c, b, a))) / 3.0;
        const double y = T1(i, a) * T1(j, b) * T1(k, c) + T1(i, a) * T2(j, k, b, c) + T1(j, b) * T2(i, k, a, c) + T1(k, c) * T2(i, j, a, b);
        const double tbar = wb / D, zbar = zb / D;
        acc[0] += tbar * w;
        acc[1] += zbar * w;
        acc[2] += tbar * y;
        acc[3] += zbar * y;
    }
    // block reduction (wave shuffle, then 4 waves through LDS)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double s = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) sm[q * 4 + wv] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        partial[(int64_t)q * nblk_total + blk] = m.mult * (sm[q * 4] + sm[q * 4 + 1] + sm[q * 4 + 2] + sm[q * 4 + 3]);
    }
}

// out[q] += sum_b partial[q][b] in a fixed order
__global__ __launch_bounds__(256) void triples_sum_kernel(double* out, const double* partial, int nblk)
{
    __shared__ double sm[4];
    const int q = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += partial[(int64_t)q * nblk + b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[q] += sm[0] + sm[1] + sm[2] + sm[3];
}

// ccsd.f90:2243: 1 + 2 sum t1^2 + sum asym_t2 * c_oovv
__global__ __launch_bounds__(256) void triples_dbase_kernel(double* out, TriplesIn in)
{
    __shared__ double sm[4];
    const int o = in.o, v = in.v;
    const int64_t n = (int64_t)o * o * v * v;
    double s = 0.0;
    for (int64_t x = threadIdx.x; x < n; x += blockDim.x) {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        double t = in.t2[x];
        s += (2.0 * t - T2(j, i, a, b)) * (t + T1(i, a) * T1(j, b));
    }
    for (int x = threadIdx.x; x < o * v; x += blockDim.x) s += 2.0 * in.t1[x] * in.t1[x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2] += 1.0 + sm[0] + sm[1] + sm[2] + sm[3];   // D(T) = out[2] + out[3], so the base term enters once
    }
}

void ccsd_triples(Context& cx, CCState& s, int64_t t_begin, int64_t t_end, double* out_host)
{
    if (!s.ready) throw Error(1, "ccsd_triples: no converged CCSD state in this context");
    const int o = s.o, v = s.v;
    const int64_t O = o, V = v, v2 = V * V, v3 = V * V * V;
    const int64_t ntot = triples_count(o);
    t_begin = std::max<int64_t>(0, t_begin);
    t_end = std::min<int64_t>(ntot, t_end);
    // operand layouts with the summed index first (the reference does the same, ccsd.f90:2056-2066)
    Tensor t2r = cx.tensor({V, V, O, O});   // t2r(d,a,j,i) = t2(i,j,a,d)
    Tensor t2h = cx.tensor({O, V, V, O});   // t2h(l,a,b,i) = t2(l,i,b,a)
    Tensor vr = cx.tensor({V, V, V, O});    // vr(d,b,c,k)  = <cb|kd> = v_vvov(c,b,k,d)
    Tensor orr = cx.tensor({O, V, O, O});   // or(l,c,j,k)  = <kj|cl> = v_oovo(k,j,c,l)
    permute_add(cx, 1.0, s.t2, "ijad", 0.0, t2r, "daji");
    permute_add(cx, 1.0, s.t2, "liba", 0.0, t2h, "labi");
    permute_add(cx, 1.0, s.v_vvov, "cbkd", 0.0, vr, "dbck");
    permute_add(cx, 1.0, s.v_oovo, "kjcl", 0.0, orr, "lcjk");
    k_fill(cx, cx.scal, 4, 0.0);
    TriplesIn in{s.e, s.t1.d, s.t2.d, s.v_oovv.d, o, v};

    // chunk size from a memory budget: 6 X blocks + 1 W block of v^3 doubles per triple
    const int64_t per = 7 * v3 * (int64_t)sizeof(double);
    int64_t nb = std::max<int64_t>(1, ((int64_t)6 << 30) / per);
    nb = std::min<int64_t>(nb, 2048);
    nb = std::min<int64_t>(nb, std::max<int64_t>(1, t_end - t_begin));
    double* Xpool = cx.alloc(6 * nb * v3);
    double* Wpool = cx.alloc(nb * v3);
    TripleMeta* meta_d = (TripleMeta*)cx.alloc((int64_t)(nb * sizeof(TripleMeta) / sizeof(double) + 1));
    int64_t* boff = cx.alloc_i64(6 * nb * 5);   // bA_p, bB_p, bC, bA_h, bB_h
    const int eblocks = (int)std::max<int64_t>(1, std::min<int64_t>((v3 + 255) / 256, 256));
    double* partial = cx.alloc(4 * nb * eblocks);

    // enumerate i<=j<=k in a fixed order and walk the requested range chunk by chunk
    std::vector<TripleMeta> meta;
    std::vector<int64_t> hA, hB, hC, hAh, hBh;
    int64_t flat = 0;
    auto flush = [&]() {
        if (meta.empty()) return;
        const int nx = (int)hC.size(), nt = (int)meta.size();
        std::vector<int64_t> pack;
        pack.reserve((size_t)nx * 5);
        pack.insert(pack.end(), hA.begin(), hA.end());
        pack.insert(pack.end(), hB.begin(), hB.end());
        pack.insert(pack.end(), hC.begin(), hC.end());
        pack.insert(pack.end(), hAh.begin(), hAh.end());
        pack.insert(pack.end(), hBh.begin(), hBh.end());
        AFESP_HIP(hipMemcpyAsync(boff, pack.data(), pack.size() * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
        AFESP_HIP(hipMemcpyAsync(meta_d, meta.data(), meta.size() * sizeof(TripleMeta), hipMemcpyHostToDevice, cx.stream));
        cx.sync();
        // X(a,b,c) = sum_d t2r(d,a | q,p) vr(d,b,c | r)                     particle half of ccsd.f90:2168
        Tensor Ablk = view(t2r.d, {V, V}), Bblk = view(vr.d, {V, V, V}), Xblk = view(Xpool, {V, V, V});
        contract(cx, 1.0, Ablk, "da", Bblk, "dbc", 0.0, Xblk, "abc", nx, boff, boff + nx, boff + 2 * nx);
        // X(a,b,c) -= sum_l t2h(l,a,b | p) or(l,c | q,r)                      hole half
        Tensor Ah = view(t2h.d, {O, V, V}), Bh = view(orr.d, {O, V});
        contract(cx, -1.0, Ah, "lab", Bh, "lc", 1.0, Xblk, "abc", nx, boff + 3 * nx, boff + 4 * nx, boff + 2 * nx);
        hipLaunchKernelGGL(triples_w_kernel, dim3(eblocks, nt), dim3(256), 0, cx.stream, Wpool, Xpool, meta_d, v);
        AFESP_HIP(hipGetLastError());
        hipLaunchKernelGGL(triples_e_kernel, dim3(eblocks, nt), dim3(256), 0, cx.stream, partial, Wpool, meta_d, in, eblocks * nt);
        AFESP_HIP(hipGetLastError());
        hipLaunchKernelGGL(triples_sum_kernel, dim3(4), dim3(256), 0, cx.stream, cx.scal, partial, eblocks * nt);
        AFESP_HIP(hipGetLastError());
        cx.sync();
        meta.clear(); hA.clear(); hB.clear(); hC.clear(); hAh.clear(); hBh.clear();
    };
    for (int i = 0; i < o && flat < t_end; ++i)
        for (int j = i; j < o && flat < t_end; ++j)
            for (int k = j; k < o && flat < t_end; ++k, ++flat) {
                if (flat < t_begin) continue;
                TripleMeta m;
                m.i = i; m.j = j; m.k = k; m.pad = 0;
                m.mult = (i == j && j == k) ? 1.0 : (i == j || j == k) ? 3.0 : 6.0;
                m.woff = (int64_t)meta.size() * v3;
                const int P[6][3] = {{i, j, k}, {j, i, k}, {k, j, i}, {i, k, j}, {j, k, i}, {k, i, j}};
                for (int q = 0; q < 6; ++q) {
                    int found = -1;
                    for (int r = 0; r < q; ++r)
                        if (P[r][0] == P[q][0] && P[r][1] == P[q][1] && P[r][2] == P[q][2]) { found = r; break; }
                    if (found >= 0) { m.xoff[q] = m.xoff[found]; continue; }
                    const int p_ = P[q][0], q_ = P[q][1], r_ = P[q][2];
                    m.xoff[q] = (int64_t)hC.size() * v3;
                    hC.push_back(m.xoff[q]);
                    hA.push_back(v2 * (q_ + O * p_));          // t2r(:,:,q,p)
                    hB.push_back(v3 * r_);                      // vr(:,:,:,r)
                    hAh.push_back(O * v2 * p_);                 // t2h(:,:,:,p)
                    hBh.push_back(O * V * (q_ + O * r_));       // or(:,:,q,r)
                }
                meta.push_back(m);
                if ((int64_t)meta.size() == nb) flush();
            }
    flush();
    if (t_begin == 0) {
        hipLaunchKernelGGL(triples_dbase_kernel, dim3(1), dim3(256), 0, cx.stream, cx.scal, in);
        AFESP_HIP(hipGetLastError());
    }
    double* h = host_scalars(cx, 4);
    out_host[0] = h[0];            // E[T]
    out_host[1] = h[0] + h[1];     // E(T)            ccsd.f90:2220
    out_host[2] = h[2];            // D[T]
    out_host[3] = h[2] + h[3];     // D(T)            ccsd.f90:2232
    cx.release(Xpool); cx.release(Wpool); cx.release(meta_d); cx.release(boff); cx.release(partial);
    cx.release(t2r.d); cx.release(t2h.d); cx.release(vr.d); cx.release(orr.d);
}

}  // namespace afesp
