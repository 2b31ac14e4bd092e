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
// under permuting (ijk).  Here: only i<=j<=k is visited, with multiplicity 6/3/1.
// The hole halves can be re-dealt among the six terms (only their sum matters): pairing the particle half of term 1 with
// the hole half of term 3 gives
//     X^{ijk}(a,b,c) = sum_d t2(i,j,a,d) <cb|kd>  -  sum_l t2(l,k,b,c) <ij|al>
// whose two halves share the row index (b,c; k) and the column index (a; i,j): ONE GEMM over the concatenated summation
// index kappa = d (+) l of length v+o, with no read-modify-write pass for the hole term.  The spin-free path then fuses
// the six X into three products of twice that length and launches them grouped (plan_fused below); the spin-orbital
// path launches its three blocks per slab (plan_for).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "ccsd.h"
#include "ccsd_so.h"
#include "tgemm.h"
#include "triples_orbit.h"

namespace afesp {

int64_t triples_count(int o) { return (int64_t)o * (o + 1) * (o + 2) / 6; }

// out[q] += sum_b partial[q][b] in a fixed order.  gridDim.y > 1: block (q, y) sums slice y of the nblk values into
// out[q * gridDim.y + y] (first stage of sum_partials below; a million partials per chunk at o=20, v=200 took 0.7 ms in
// one block per quantity).
__global__ __launch_bounds__(256) void triples_sum_kernel(double* out, const double* partial, int nblk)
{
    __shared__ double sm[4];
    const int q = blockIdx.x;
    const int per = (nblk + (int)gridDim.y - 1) / (int)gridDim.y;
    const int lo = (int)blockIdx.y * per, hi = min(nblk, lo + per);
    double s = 0.0;
    for (int b = lo + threadIdx.x; b < hi; b += blockDim.x) s += partial[(int64_t)q * nblk + b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (gridDim.y > 1) out[q * gridDim.y + blockIdx.y] = sm[0] + sm[1] + sm[2] + sm[3];
        else out[q] += sm[0] + sm[1] + sm[2] + sm[3];
    }
}

// Small systems (one chunk, few partials): the ordered sums of all nq quantities and their publication to the host in ONE block --
// scal[q] = dst[q] = sum_b partial[q][b], then the sequence number (host_scalars_slot, contract.hip).  Replaces a fill, a sum and a
// publishing launch (three of the eight launches of a plain (T) at o = 5, v = 53).
__global__ __launch_bounds__(256) void triples_sum_publish_kernel(double* __restrict__ scal, double* __restrict__ dst, double seq,
                                                                  const double* __restrict__ partial, int nq, int nblk)
{
    __shared__ double sm[6][4];
    for (int q = 0; q < nq; ++q) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblk; b += blockDim.x) s += partial[(int64_t)q * nblk + b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < 6) {
        const double val = (int)threadIdx.x < nq ? sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3] : 0.0;
        scal[threadIdx.x] = val;
        dst[threadIdx.x] = val;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&dst[64], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// diagnostic builds (-DAFESP_ORBIT_STAMPS): reads and clears the orbit kernel's phase sums (triples_orbit.h)
hipError_t triples_read_orbit_stamps(unsigned long long* out, int n)
{
#ifdef AFESP_ORBIT_STAMPS
    std::vector<unsigned long long> buf((size_t)ORB_SLOTS * 16, 0ull);
    hipError_t e = hipMemcpyFromSymbol(buf.data(), HIP_SYMBOL(g_orbit_stamp), sizeof(unsigned long long) * buf.size());
    if (e != hipSuccess) return e;
    for (int i = 0; i < n && i < 16; ++i) {
        out[i] = 0;
        for (int sl = 0; sl < ORB_SLOTS; ++sl) out[i] += buf[(size_t)sl * 16 + i];
    }
    std::fill(buf.begin(), buf.end(), 0ull);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_orbit_stamp), buf.data(), sizeof(unsigned long long) * buf.size());
#else
    for (int i = 0; i < n && i < 16; ++i) out[i] = 0;
    return hipSuccess;
#endif
}


// out[q] += sum of partial[q][0..nblk) for q < nq, two stages when there are many partials; `tmp` holds nq * 128 doubles
static void sum_partials(Context& cx, double* out, const double* partial, int nq, int nblk, double* tmp)
{
    if (nblk > 8192) {
        AFESP_KLAUNCH(triples_sum_kernel, dim3(nq, 128), dim3(256), 0, cx.stream, tmp, partial, nblk);
        AFESP_HIP(hipGetLastError());
        AFESP_KLAUNCH(triples_sum_kernel, dim3(nq), dim3(256), 0, cx.stream, out, tmp, 128);
    } else {
        AFESP_KLAUNCH(triples_sum_kernel, dim3(nq), dim3(256), 0, cx.stream, out, partial, nblk);
    }
    AFESP_HIP(hipGetLastError());
}

// ccsd.f90:2243: 1 + 2 sum t1^2 + sum asym_t2 * c_oovv.  One partial per block; the ordered sum is done by triples_sum_kernel.
__global__ __launch_bounds__(256) void triples_dbase_kernel(double* partial, TriplesIn in, int nblk_total)
{
    __shared__ double sm[4];
    const int o = in.o, v = in.v;
    const int64_t n = (int64_t)o * o * v * v;
    double s = 0.0;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
        int i = (int)(x % o);
        int64_t r = x / o;
        int j = (int)(r % o);
        r /= o;
        int a = (int)(r % v), b = (int)(r / v);
        double tv = in.t2[x];
        double tx = in.t2[j + (int64_t)o * (i + (int64_t)o * (a + (int64_t)v * b))];
        s += (2.0 * tv - tx) * (tv + in.t1[i + o * a] * in.t1[j + o * b]);
        if (j == 0 && b == 0) s += 2.0 * in.t1[i + o * a] * in.t1[i + o * a];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int q = 0; q < 4; ++q) partial[(int64_t)q * nblk_total + blockIdx.x] = 0.0;
        partial[(int64_t)2 * nblk_total + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3] + (blockIdx.x == 0 ? 1.0 : 0.0);
    }
}

// The concatenated (T) operands in one pass each (instead of a memset and four or two permutations):
//   vt (kappa,b,c,k) and vtT(kappa,c,b,k):  kappa < v: X(b,c,k,kappa) through the strides sx[] = (b,c,k,d);
//                                           v <= kappa < v+o: t2(kappa-v, k, b, c);  else 0 (K padding)
__global__ __launch_bounds__(256) void triples_build_vt_kernel(double* vt, double* vtT, const double* X, int64_t sb, int64_t sc, int64_t sk,
                                                             int64_t sd, const double* t2, int o, int v, int Kc)
{
    const int64_t n = (int64_t)Kc * v * v * o;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
        const int kap = (int)(x % Kc);
        int64_t r = x / Kc;
        const int b = (int)(r % v);
        r /= v;
        const int c = (int)(r % v), k = (int)(r / v);
        double val = 0.0;
        if (kap < v) val = X[sb * b + sc * c + sk * k + sd * kap];
        else if (kap < v + o) val = t2[(kap - v) + (int64_t)o * (k + (int64_t)o * (b + (int64_t)v * c))];
        vt[x] = val;
        vtT[kap + (int64_t)Kc * (c + (int64_t)v * (b + (int64_t)v * k))] = val;
    }
}
// Same, for a source that is unit-stride along c (the plain (T): X = v_vvov(c,b,k,d)): a workgroup moves a 32 x 32 tile of the
// (kappa, c) plane of one (b,k) through LDS, so the source is read along c and both results are written along kappa (the
// thread-per-element kernel fetched 16 bytes from HBM per byte it needed: 3.4 ms at o=20, v=200).
__global__ __launch_bounds__(256) void triples_build_vt_tiled_kernel(double* vt, double* vtT, const double* X, int64_t sb, int64_t sk, int64_t sd,
                                                                   const double* t2, int o, int v, int Kc)
{
    __shared__ double tile[32][33];
    const int nkt = (Kc + 31) / 32, nct = (v + 31) / 32;
    int64_t blk = blockIdx.x;
    const int kap0 = (int)(blk % nkt) * 32;
    blk /= nkt;
    const int c0 = (int)(blk % nct) * 32;
    blk /= nct;
    const int b = (int)(blk % v), k = (int)(blk / v);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kap = kap0 + ty + 8 * r, c = c0 + tx;
        double val = 0.0;
        if (c < v) {
            if (kap < v) val = X[sb * b + c + sk * k + sd * kap];
            else if (kap < v + o) val = t2[(kap - v) + (int64_t)o * (k + (int64_t)o * (b + (int64_t)v * c))];
        }
        tile[ty + 8 * r][tx] = val;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kap = kap0 + tx, c = c0 + ty + 8 * r;
        if (kap < Kc && c < v) {
            const double val = tile[tx][ty + 8 * r];
            vt[kap + (int64_t)Kc * (b + (int64_t)v * (c + (int64_t)v * k))] = val;
            vtT[kap + (int64_t)Kc * (c + (int64_t)v * (b + (int64_t)v * k))] = val;
        }
    }
}
//   tt(kappa,a,j,i):  kappa < v: t2(i,j,a,kappa);  v <= kappa < v+o: -Y(i,j,a,kappa-v) through sy[] = (i,j,a,l);  else 0
__global__ __launch_bounds__(256) void triples_build_tt_kernel(double* tt, const double* t2, const double* Y, int64_t si, int64_t sj, int64_t sa,
                                                             int64_t sl, int o, int v, int Kc)
{
    const int64_t n = (int64_t)Kc * v * o * o;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t)gridDim.x * blockDim.x) {
        const int kap = (int)(x % Kc);
        int64_t r = x / Kc;
        const int a = (int)(r % v);
        r /= v;
        const int j = (int)(r % o), i = (int)(r / o);
        double val = 0.0;
        if (kap < v) val = t2[i + (int64_t)o * (j + (int64_t)o * (a + (int64_t)v * kap))];
        else if (kap < v + o) val = -Y[si * i + sj * j + sa * a + sl * (kap - v)];
        tt[x] = val;
    }
}

// Everything about a (T) evaluation that depends only on (o, v, triple range): chunking, per-chunk GEMM column tables,
// per-triple metadata, the cube-orbit list.  Built once, kept in device memory, reused by every later call.
struct TriplesPlan {
    int o = 0, v = 0;
    int64_t t_begin = -1, t_end = -1, nb = 0;
    int norb = 0;
    bool cr = false;
    int mode = 0;   // 0 spin-free (fused pairs of terms, plan_fused), 1 spin-orbital (i<j<k, three blocks, plan_for)
    int sblock = 0; // fused scheme: occupied block size of the triple enumeration
    int64_t epoch = -1;   // Context::scratch_epoch when the device tables were placed (they live in cached scratch buffers)
    struct Group { int r; int64_t start, N; int q = 0; int64_t koffA = 0, koffB = 0; };   // q, koff*: fused scheme only
    struct Chunk { int nt; int64_t meta_off, tab_off, ntab; std::vector<Group> groups;
                   int64_t gdesc_off = 0; int total_tiles = 0, max_ntiles = 0;      // fused scheme: GettGroup array (device)
                   // fused scheme, groups with q == r: Y^{p;qq}(x;y,z) = X(x;y,z) + X(x;z,y), only X is computed (half the
                   // summation length, a launch of its own) and the orbit kernel adds the transpose (TripleMeta::pad)
                   int64_t gdesc_diag_off = 0; int ngroups_diag = 0, total_tiles_diag = 0, max_ntiles_diag = 0, ngroups_off = 0;
                   int64_t ncol_off = 0, ncol_diag = 0; bool split_diag = false;
                   // LDS-DMA kernel (tgemm.h): all groups of the chunk in one launch, the q == r groups with half the K steps
                   int64_t tg_off = 0, tab32_off = 0; int tg_ngroups = 0, tg_tiles = 0, tg_max_ntiles = 0; };
    std::vector<Chunk> chunks;
    int64_t* tables = nullptr;     // [kappa | Am | Cm | per chunk: offBn, offCn]
    int64_t off_k = 0, off_Am = 0, off_Cm = 0;
    TripleMeta* meta = nullptr;
    int* orbits = nullptr;
    GettGroup* gdesc = nullptr;    // fused scheme: group descriptors of all chunks
    bool use_tg = false;           // fused scheme: the products run on tgemm_kernel (tgemm.h) instead of the grouped gett_kernel
    TgGroup* tgdesc = nullptr;
    uint32_t* tables32 = nullptr;  // [rowA (v^2) | per chunk: colB]  byte offsets
    bool c_pairs = false;          // columns (x, x + 1), x even, of a block are adjacent in the cube-blocked layout: v even (tgemm.h)
    double* pool0 = nullptr;       // fused scheme, not completely renormalised: base the block offsets refer to (assemble_pool)
};

// The block pool of the fused (T) as a list of device blocks.  A block of the pool is only ever addressed as base + offset
// (GEMM column tables, TripleMeta::xoff), and the address space is flat, so the pool need not be ONE allocation: blocks the
// context's arena holds idle -- the two AO->MO temporaries, 2 x 9.4 GB at config 5 -- are taken whole and filled with as many
// blocks as fit, and only the rest is asked for anew (5 GB instead of 24: on this runtime a fresh allocation out of recycled
// device memory costs 60-75 ms per GB).  Returns the element offset of every block relative to *pool0.
static std::vector<int64_t> assemble_pool(Context& cx, int64_t nblocks, int64_t vp3, double** pool0)
{
    const size_t bb = (size_t)vp3 * sizeof(double);
    struct Piece { char* base; int64_t n; };
    std::vector<Piece> have;
    int64_t cap = 0;
    for (int i = 0;; ++i) {
        auto it = cx.cache.find("t_xpool" + std::to_string(i));
        if (it == cx.cache.end()) break;
        have.push_back({(char*)it->second.first, (int64_t)(it->second.second / bb)});
        cap += have.back().n;
    }
    if (cap < nblocks) {
        if (!have.empty()) cx.drop_scratch("t_xpool");
        have.clear();
        int64_t need = nblocks;
        for (int i = 0; need > 0; ++i) {
            const std::string name = "t_xpool" + std::to_string(i);
            size_t got = 0;
            // an idle block that is just what is still needed; else the largest one that can be filled completely (and holds a
            // good part of it); else a new one for the rest
            void* q = cx.arena.take_largest((size_t)need * bb, (size_t)need * bb + (size_t)need * bb / 4, &got);
            if (!q) q = cx.arena.take_largest((size_t)std::min<int64_t>(need, 8) * bb, (size_t)need * bb, &got);
            int64_t k;
            if (q) {
                k = std::min<int64_t>(need, (int64_t)(got / bb));
                cx.cache[name] = {q, got};
            } else {
                k = need;
                q = cx.scratch(name, k * vp3);
            }
            have.push_back({(char*)q, k});
            need -= k;
        }
    }
    *pool0 = (double*)have[0].base;
    std::vector<int64_t> off;
    off.reserve((size_t)nblocks);
    for (const Piece& pc : have)
        for (int64_t b = 0; b < pc.n && (int64_t)off.size() < nblocks; ++b)
            off.push_back((int64_t)((pc.base - have[0].base) / (ptrdiff_t)sizeof(double)) + b * vp3);
    return off;
}

static bool fused_use_tg(int o, int v);

// Plan of the spin-orbital (T): i<j<k, three blocks per triple; one tgemm_kernel launch per chunk (or, under AFESP_T_GEMM=gett, one
// gather-GEMM launch per integral slab and chunk).
static TriplesPlan* plan_for(Context& cx, void*& slot, int o, int v, int64_t t_begin, int64_t t_end)
{
    TriplesPlan* p = (TriplesPlan*)slot;
    if (p && p->o == o && p->v == v && p->t_begin == t_begin && p->t_end == t_end && p->mode == 1 && p->epoch == cx.scratch_epoch &&
        p->use_tg == (fused_use_tg(o, v) && (((int64_t)v + o + 15) / 16 * 16) >= 2 * TG_BK))
        return p;
    delete p;
    p = new TriplesPlan();
    slot = p;
    const int64_t O = o, V = v, Kc = (V + O + 15) / 16 * 16;   // padded, see ccsd_triples
    p->o = o; p->v = v; p->t_begin = t_begin; p->t_end = t_end; p->cr = false; p->mode = 1;
    // the products of this path on the LDS-DMA kernel too (tgemm.h): K-contiguous operands, one run of the summation index per group
    p->use_tg = fused_use_tg(o, v) && Kc >= 2 * TG_BK;
    p->c_pairs = (v % 2) == 0;
    std::vector<uint32_t> tab32;
    if (p->use_tg)
        for (int64_t m = 0; m < V * V; ++m) tab32.push_back((uint32_t)(8 * Kc * m));
    // chunk size: 6 X blocks of (padded) v^3 doubles per triple; W never leaves LDS
    const int64_t nt8 = (V + TT - 1) / TT, vp3 = nt8 * nt8 * nt8 * CUBE;   // a block is stored cube by cube, edges padded to 8
    const int64_t per = 3 * vp3 * (int64_t)sizeof(double);
    // pool budget: a quarter of the device memory, at most 64 GiB (MI355X: 288 GB -> 64 GiB, ~165 triples per chunk at
    // v = 200): the more triples share an integral slab, the wider each GEMM and the smaller its ragged last round
    size_t mem_free = 0, mem_total = 0;
    AFESP_HIP(hipMemGetInfo(&mem_free, &mem_total));
    const int64_t budget = std::min<int64_t>((int64_t)64 << 30, (int64_t)(mem_total / 4));
    int64_t nb = std::max<int64_t>(1, budget / per);
    nb = std::min<int64_t>(nb, 4096);
    nb = std::min<int64_t>(nb, std::max<int64_t>(1, t_end - t_begin));
    p->nb = nb;
    std::vector<int64_t> tab;
    p->off_k = 0;
    for (int64_t x = 0; x < Kc; ++x) tab.push_back(x);
    p->off_Am = (int64_t)tab.size();
    for (int64_t c = 0; c < V; ++c)
        for (int64_t b = 0; b < V; ++b) tab.push_back(Kc * (b + V * c));      // rows (b,c) of vt(:,b,c,k)
    p->off_Cm = (int64_t)tab.size();
    for (int64_t c = 0; c < V; ++c)
        for (int64_t b = 0; b < V; ++b)                                       // rows (b,c) of X(a,b,c), cube-blocked
            tab.push_back(CUBE * nt8 * (b / TT) + TT * (b % TT) + CUBE * nt8 * nt8 * (c / TT) + TT * TT * (c % TT));
    std::vector<TripleMeta> metas;
    struct Ord { int p, q, r; int64_t buf; };
    std::vector<Ord> ords;
    std::vector<TripleMeta> cur;
    auto flush = [&]() {
        if (cur.empty()) return;
        TriplesPlan::Chunk ch;
        ch.nt = (int)cur.size();
        ch.meta_off = (int64_t)metas.size();
        metas.insert(metas.end(), cur.begin(), cur.end());
        // group the chunk's ordered triples by their last index r (the occupied index carried by vt)
        std::vector<int64_t> hBn, hCn;
        for (int r = 0; r < o; ++r) {
            const int64_t start = (int64_t)hBn.size();
            for (const Ord& od : ords)
                if (od.r == r)
                    for (int64_t a = 0; a < V; ++a) {
                        hBn.push_back(Kc * (a + V * (od.q + O * od.p)));   // tt(:, a, q, p)
                        hCn.push_back(CUBE * (a / TT) + a % TT + vp3 * od.buf);
                    }
            const int64_t N = (int64_t)hBn.size() - start;
            if (N > 0) ch.groups.push_back({r, start, N});
        }
        ch.ntab = (int64_t)hBn.size();
        if (p->use_tg) {
            ch.tab32_off = (int64_t)tab32.size();
            for (int64_t x : hBn) tab32.push_back((uint32_t)(8 * x));
        }
        ch.tab_off = (int64_t)tab.size();
        tab.insert(tab.end(), hBn.begin(), hBn.end());
        tab.insert(tab.end(), hCn.begin(), hCn.end());
        p->chunks.push_back(std::move(ch));
        cur.clear(); ords.clear();
    };
    int64_t flat = 0;
    const int step = 1;   // strictly increasing: the summand is antisymmetric in i,j,k
    for (int i = 0; i < o && flat < t_end; ++i)
        for (int j = i + step; j < o && flat < t_end; ++j)
            for (int k = j + step; k < o && flat < t_end; ++k, ++flat) {
                if (flat < t_begin) continue;
                TripleMeta m;
                m.i = i; m.j = j; m.k = k; m.pad = 0;
                m.mult = 1.0;
                m.woff = 0;
                // ordered (p,q,r): the amplitude operand carries the pair (p,q), the integral operand carries r.
                // spin-orbital: Y^{i;jk}, Y^{j;ik}, Y^{k;ij} (see so_triples)
                const int P3[6][3] = {{j, k, i}, {i, k, j}, {i, j, k}, {j, k, i}, {j, k, i}, {j, k, i}};
                const int (*P)[3] = P3;
                for (int q = 0; q < 6; ++q) {
                    int found = -1;
                    for (int r = 0; r < q; ++r)
                        if (P[r][0] == P[q][0] && P[r][1] == P[q][1] && P[r][2] == P[q][2]) { found = r; break; }
                    if (found >= 0) { m.xoff[q] = m.xoff[found]; continue; }
                    const int64_t buf = (int64_t)ords.size();
                    m.xoff[q] = buf * vp3;
                    ords.push_back({P[q][0], P[q][1], P[q][2], buf});
                }
                cur.push_back(m);
                if ((int64_t)cur.size() == nb) flush();
            }
    flush();
    // orbits of 8x8x8 cubes under index permutation: tile triples A <= B <= C
    std::vector<int> orb;
    for (int A = 0; A < nt8; ++A)
        for (int B = A; B < nt8; ++B)
            for (int C = B; C < nt8; ++C) orb.push_back(A | (B << 10) | (C << 20));
    p->norb = (int)orb.size();
    // (names of their own: a spin-free plan cached in the same context keeps pointing at its tables)
    tab.resize(tab.size() + 128, 0);   // (the LDS-DMA kernel reads its tables in whole 1-KiB pieces, tgemm.h)
    p->tables = (int64_t*)cx.scratch("so_tables", (int64_t)tab.size());
    p->meta = (TripleMeta*)cx.scratch("so_meta", (int64_t)(metas.size() * sizeof(TripleMeta) / sizeof(double) + 1));
    p->orbits = (int*)cx.scratch("so_orbits", (int64_t)orb.size() / 2 + 1);
    AFESP_HIP(hipMemcpyAsync(p->tables, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
    std::vector<TgGroup> tg;
    if (p->use_tg) {
        tab32.resize(tab32.size() + 256, 0u);
        p->tables32 = (uint32_t*)cx.scratch("so_tables32", (int64_t)(tab32.size() / 2 + 1));
        AFESP_HIP(hipMemcpyAsync(p->tables32, tab32.data(), tab32.size() * sizeof(uint32_t), hipMemcpyHostToDevice, cx.stream));
        const int64_t v2 = V * V;
        const int mt = (int)((v2 + TG_BM - 1) / TG_BM);
        for (TriplesPlan::Chunk& ch : p->chunks) {
            const int64_t* tabs = p->tables + ch.tab_off;
            ch.tg_off = (int64_t)tg.size();
            int mx = 0, tile = 0;
            for (const TriplesPlan::Group& g : ch.groups) mx = std::max(mx, (int)((g.N + TG_BN - 1) / TG_BN));
            const int gm = tgemm_group_m((int)v2, mx);
            for (const TriplesPlan::Group& g : ch.groups) {
                TgGroup d{};
                d.a1 = d.a2 = Kc * v2 * g.r;            // vt(:, ., ., r)
                d.colB = p->tables32 + ch.tab32_off + g.start;
                d.offCn = tabs + ch.ntab + g.start;
                d.N = (int)g.N;
                d.ntiles = (int)((g.N + TG_BN - 1) / TG_BN);
                d.tile_start = tile;
                d.nk1 = d.nk = (int)(Kc / TG_BK);
                d.inv_width = tgemm_inverse(gm * d.ntiles);
                tile += mt * d.ntiles;
                tg.push_back(d);
            }
            TgGroup end{};
            end.tile_start = tile;
            tg.push_back(end);
            ch.tg_ngroups = (int)ch.groups.size();
            ch.tg_tiles = tile;
            ch.tg_max_ntiles = mx;
        }
        p->tgdesc = (TgGroup*)cx.scratch("so_tgdesc", (int64_t)(tg.size() * sizeof(TgGroup) / sizeof(double) + 1));
        AFESP_HIP(hipMemcpyAsync(p->tgdesc, tg.data(), tg.size() * sizeof(TgGroup), hipMemcpyHostToDevice, cx.stream));
    }
    AFESP_HIP(hipMemcpyAsync(p->meta, metas.data(), metas.size() * sizeof(TripleMeta), hipMemcpyHostToDevice, cx.stream));
    AFESP_HIP(hipMemcpyAsync(p->orbits, orb.data(), orb.size() * sizeof(int), hipMemcpyHostToDevice, cx.stream));
    cx.sync();   // the host vectors die here
    p->epoch = cx.scratch_epoch;
    return p;
}

// ---------------------------------------------------------------------------------------------------- fused scheme
// The six terms of W pair up by the index that comes from the amplitude operand ("single"):
//   Y^{p;qr}(x;y,z) = X^{pqr}(x;y,z) + X^{prq}(x;z,y)
//                   = sum_kappa [ tt(kappa;x,q,p) vt(kappa;y,z,r) + tt(kappa;x,r,p) vt(kappa;z,y,q) ]
// i.e. ONE GEMM over a summation index of length 2(v+o) whose second half reads the slab of q with (y,z) transposed
// (a second copy vtT(kappa;y,z,q) = vt(kappa;z,y,q) right behind vt) and the amplitude column of the pair (r,p) instead
// of (q,p) -- both are constant shifts inside a group of columns that share the unordered pair {q,r}, so they go into
// the group's K-offset tables.  Then
//   W^{ijk}(a,b,c) = Y^{i;jk}(a;b,c) + Y^{j;ik}(b;a,c) + Y^{k;ij}(c;a,b):
// half as many blocks written by the GEMMs and read by the orbit kernel, and twice the K per tile.
// A GEMM needs many columns per {q,r}: the triples are therefore enumerated by blocks of `s` occupied indices --
// block triple (I<=J<=K) major, i<=j<=k inside -- so that one chunk (= one block triple) has up to s columns sets
// per pair.  The flat index [t_begin, t_end) the ranks shard refers to THIS order.
static int fused_block_size(int o, int v, bool cr, int64_t budget_bytes)
{
    (void)cr;   // the budget is per pool (the completely renormalised variant has two)
    const int64_t nt8 = (v + TT - 1) / TT, vp3 = nt8 * nt8 * nt8 * CUBE;
    const int64_t per_block = vp3 * (int64_t)sizeof(double);
    int smax = 1;
    while (smax < o && (int64_t)3 * (smax + 1) * (smax + 1) * (smax + 1) * per_block <= budget_bytes) ++smax;
    // What a larger block buys is wider column groups (N = s v columns per occupied pair, in 128-column tiles) and longer
    // launches; from about 900 columns on neither matters any more (config 5: s = 5, 7 -> 506.8, 506.0 ms; s = 4 -> 549), while
    // the pool grows with s^3 -- and every byte allocated for the first time costs on this runtime.  So: the
    // smallest size with >= 900 columns per group, or one of the next two if it fills its column tiles more than 2 % better.
    if (knobs().t_block > 0) return std::max(1, std::min(smax, knobs().t_block));   // tuning knob AFESP_T_BLOCK
    const int smin = std::min(smax, std::max(1, (900 + v - 1) / v));
    auto fill = [&](int sz) { return (double)((int64_t)v * sz) / (double)((((int64_t)v * sz + 127) / 128) * 128); };
    int best = smin;
    for (int sz = smin + 1; sz <= std::min(smax, smin + 2); ++sz)
        if (fill(sz) > fill(best) + 0.02) best = sz;
    return best;
}

static int64_t device_pool_budget()
{
    size_t mem_free = 0, mem_total = 0;
    AFESP_HIP(hipMemGetInfo(&mem_free, &mem_total));
    if (knobs().t_pool_gib >= 0) return knobs().t_pool_gib << 30;   // tuning knob AFESP_T_POOL_GIB
    // upper limit of one pool: a quarter of the device, 64 GiB at most (fused_block_size normally stays well below it: 24 GB at
    // config 5)
    return std::min<int64_t>((int64_t)64 << 30, (int64_t)(mem_total / 4));
}

// Which kernel runs the fused products: the LDS-DMA kernel (tgemm.h) whenever its 32-bit byte offsets reach every row of vt
// and every column of tt; AFESP_T_GEMM=gett forces the grouped gather kernel (A/B runs, tests of both).
static bool fused_use_tg(int o, int v)
{
    if (knobs().t_gemm_gett) return false;
    const int64_t O = o, V = v, Kc = (V + O + 15) / 16 * 16;
    return 8 * Kc * V * V < ((int64_t)1 << 32) && 8 * Kc * V * O * O < ((int64_t)1 << 32);
}

static TriplesPlan* plan_fused(Context& cx, void*& slot, int o, int v, int64_t t_begin, int64_t t_end, bool cr)
{
    TriplesPlan* p = (TriplesPlan*)slot;
    const bool use_tg = fused_use_tg(o, v);
    if (p && p->o == o && p->v == v && p->t_begin == t_begin && p->t_end == t_end && p->cr == cr && p->mode == 0 &&
        p->epoch == cx.scratch_epoch && p->use_tg == use_tg)
        return p;
    delete p;
    p = new TriplesPlan();
    slot = p;
    const int64_t O = o, V = v, v2 = V * V, Kc = (V + O + 15) / 16 * 16;
    const int64_t nt8 = (V + TT - 1) / TT, vp3 = nt8 * nt8 * nt8 * CUBE;
    p->o = o; p->v = v; p->t_begin = t_begin; p->t_end = t_end; p->cr = cr; p->mode = 0; p->use_tg = use_tg;
    // a group's columns are (x, block) with x fastest: column 2k is an even x iff v is even; rows and blocks start at multiples of 8
    p->c_pairs = (v % 2) == 0;
    const int nk1 = (int)(Kc / TG_BK);
    std::vector<uint32_t> tab32;
    if (use_tg)
        for (int64_t m = 0; m < v2; ++m) tab32.push_back((uint32_t)(8 * Kc * m));   // rows (b,c) of vt / vtT, bytes
    const int sb = fused_block_size(o, v, cr, device_pool_budget());
    p->sblock = sb;
    const int nbk = (o + sb - 1) / sb;
    std::vector<int64_t> tab;
    p->off_k = 0;   // unused in this scheme (every group has its own K tables)
    p->off_Am = (int64_t)tab.size();
    for (int64_t c = 0; c < V; ++c)
        for (int64_t b = 0; b < V; ++b) tab.push_back(Kc * (b + V * c));      // rows (b,c) of vt(:,b,c,r) and vtT(:,b,c,q)
    p->off_Cm = (int64_t)tab.size();
    for (int64_t c = 0; c < V; ++c)
        for (int64_t b = 0; b < V; ++b)
            tab.push_back(CUBE * nt8 * (b / TT) + TT * (b % TT) + CUBE * nt8 * nt8 * (c / TT) + TT * TT * (c % TT));
    std::vector<TripleMeta> metas;
    int64_t flat = 0, max_blocks = 1;
    int tm, tn, BM, BN;
    gett_grouped_tile((int)v2, true, &tm, &tn, &BM, &BN);
    const int mtiles = (int)((v2 + BM - 1) / BM);
    const int64_t split_min_tiles = knobs().t_split_tiles;   // test / tuning knob AFESP_T_SPLIT_TILES, read when a plan is built
    // first pass: the largest number of distinct blocks Y^{p;qr} of one block triple inside the requested range = the pool size
    {
        int64_t fl = 0;
        for (int I = 0; I < nbk; ++I)
            for (int J = I; J < nbk; ++J)
                for (int K = J; K < nbk; ++K) {
                    std::unordered_map<int64_t, int> seen;
                    for (int i = I * sb; i < std::min(o, (I + 1) * sb); ++i)
                        for (int j = std::max(i, J * sb); j < std::min(o, (J + 1) * sb); ++j)
                            for (int k = std::max(j, K * sb); k < std::min(o, (K + 1) * sb); ++k, ++fl) {
                                if (fl < t_begin || fl >= t_end) continue;
                                const int pq[3][3] = {{i, j, k}, {j, i, k}, {k, i, j}};
                                for (auto& b : pq) seen.emplace(((int64_t)b[0] * o + std::min(b[1], b[2])) * o + std::max(b[1], b[2]), 0);
                            }
                    max_blocks = std::max<int64_t>(max_blocks, (int64_t)seen.size());
                }
    }
    // where block b of the pool lives (element offset from the pool base): one allocation for the completely renormalised
    // variant (its second pool mirrors the first), pieces of idle memory otherwise
    // The plain evaluation keeps its pool as the pieces "t_xpool0..", the completely renormalised one as "t_xpool" + "t_mpool":
    // whichever this plan does not use goes back to the arena (a plain (T) followed by a CR one held three pools, ~72 GB at
    // o = 20, v = 200, and more where the arena has nothing idle left to trim).
    if (cr) {
        if (cx.cache.count("t_xpool0")) cx.drop_scratch("t_xpool");
    } else {
        cx.drop_scratch("t_mpool");
        if (cx.cache.count("t_xpool")) {
            // (exactly that name: the prefix would take the pieces along)
            cx.quiesce();
            cx.arena.put(cx.cache["t_xpool"].first);
            cx.cache.erase("t_xpool");
            ++cx.scratch_epoch;
        }
    }
    std::vector<int64_t> blk_off;
    if (cr || knobs().t_one_pool) {
        for (int64_t b = 0; b < max_blocks; ++b) blk_off.push_back(b * vp3);
        p->pool0 = nullptr;
    } else {
        blk_off = assemble_pool(cx, max_blocks, vp3, &p->pool0);
    }
    for (int I = 0; I < nbk; ++I)
        for (int J = I; J < nbk; ++J)
            for (int K = J; K < nbk; ++K) {
                // triples of this block triple inside the requested range
                std::vector<TripleMeta> cur;
                struct Blk { int p, q, r; int64_t buf; };
                std::vector<Blk> blks;
                std::unordered_map<int64_t, int64_t> seen;
                auto block_of = [&](int pp, int qq, int rr) {
                    if (qq > rr) std::swap(qq, rr);
                    const int64_t key = ((int64_t)pp * o + qq) * o + rr;
                    auto it = seen.find(key);
                    if (it != seen.end()) return it->second;
                    blks.push_back({pp, qq, rr, (int64_t)blks.size()});
                    seen.emplace(key, blks.back().buf);
                    return blks.back().buf;
                };
                for (int i = I * sb; i < std::min(o, (I + 1) * sb); ++i)
                    for (int j = std::max(i, J * sb); j < std::min(o, (J + 1) * sb); ++j)
                        for (int k = std::max(j, K * sb); k < std::min(o, (K + 1) * sb); ++k, ++flat) {
                            if (flat < t_begin || flat >= t_end) continue;
                            TripleMeta m;
                            m.i = i; m.j = j; m.k = k;
                            // blocks Y^{i;jk}, Y^{j;ik}, Y^{k;ij} whose pair coincides hold only X (see Chunk)
                            m.pad = (j == k ? 1 : 0) | (i == k ? 2 : 0) | (i == j ? 4 : 0);
                            m.mult = (i == j && j == k) ? 1.0 : (i == j || j == k) ? 3.0 : 6.0;
                            m.woff = 0;
                            for (int q = 0; q < 6; ++q) m.xoff[q] = 0;
                            m.xoff[0] = blk_off[(size_t)block_of(i, j, k)];   // Y^{i;jk}(a;b,c)
                            m.xoff[1] = blk_off[(size_t)block_of(j, i, k)];   // Y^{j;ik}(b;a,c)
                            m.xoff[5] = blk_off[(size_t)block_of(k, i, j)];   // Y^{k;ij}(c;a,b)
                            cur.push_back(m);
                        }
                if (cur.empty()) continue;
                TriplesPlan::Chunk ch;
                ch.nt = (int)cur.size();
                ch.meta_off = (int64_t)metas.size();
                metas.insert(metas.end(), cur.begin(), cur.end());
                if ((int64_t)blks.size() > max_blocks) throw Error(2, "triples plan: block count of a chunk exceeds the pool");
                // groups of columns that share {q,r}
                std::vector<int64_t> hBn, hCn, hK;
                std::stable_sort(blks.begin(), blks.end(), [](const Blk& x, const Blk& y) { return x.q != y.q ? x.q < y.q : x.r < y.r; });
                for (size_t b0 = 0; b0 < blks.size();) {
                    const int q = blks[b0].q, r = blks[b0].r;
                    size_t b1 = b0;
                    while (b1 < blks.size() && blks[b1].q == q && blks[b1].r == r) ++b1;
                    const int64_t start = (int64_t)hBn.size();
                    for (size_t bi = b0; bi < b1; ++bi)
                        for (int64_t x = 0; x < V; ++x) {
                            hBn.push_back(Kc * (x + V * (q + O * blks[bi].p)));           // tt(:, x, q, p); second half: (r, p)
                            hCn.push_back(CUBE * (x / TT) + x % TT + blk_off[(size_t)blks[bi].buf]);
                        }
                    b0 = b1;
                    TriplesPlan::Group g;
                    g.r = r; g.q = q; g.start = start; g.N = (int64_t)hBn.size() - start;
                    g.koffA = (int64_t)hK.size();
                    for (int64_t x = 0; x < Kc; ++x) hK.push_back(x);
                    for (int64_t x = 0; x < Kc; ++x) hK.push_back(Kc * v2 * O + Kc * v2 * ((int64_t)q - r) + x);   // vtT(:, ., ., q)
                    g.koffB = (int64_t)hK.size();
                    for (int64_t x = 0; x < Kc; ++x) hK.push_back(x);
                    for (int64_t x = 0; x < Kc; ++x) hK.push_back(Kc * V * ((int64_t)r - q) + x);                  // tt(:, x, r, p)
                    ch.groups.push_back(g);
                }
                // the q == r groups get a launch of their own (half the summation length) when it fills the device a few
                // times over; a small system keeps them in the one launch, computed in full
                int64_t diag_tiles = 0;
                for (const TriplesPlan::Group& g : ch.groups)
                    if (g.q == g.r) diag_tiles += mtiles * ((g.N + BN - 1) / BN);
                ch.split_diag = diag_tiles >= split_min_tiles;
                // (the LDS-DMA kernel takes a K-step count per group: the q == r groups always run over half the summation
                // index, in the same launch -- unless that would be a single step, which its pipeline does not take)
                if (use_tg) ch.split_diag = nk1 >= 2;
                if (!ch.split_diag)
                    for (int t = 0; t < ch.nt; ++t) metas[ch.meta_off + t].pad = 0;
                ch.ntab = (int64_t)hBn.size();
                if (use_tg) {
                    ch.tab32_off = (int64_t)tab32.size();
                    for (int64_t x : hBn) tab32.push_back((uint32_t)(8 * x));
                }
                ch.tab_off = (int64_t)tab.size();
                tab.insert(tab.end(), hBn.begin(), hBn.end());
                tab.insert(tab.end(), hCn.begin(), hCn.end());
                const int64_t kbase = (int64_t)tab.size() - ch.tab_off;
                tab.insert(tab.end(), hK.begin(), hK.end());
                for (TriplesPlan::Group& g : ch.groups) { g.koffA += kbase; g.koffB += kbase; }
                p->chunks.push_back(std::move(ch));
            }
    p->nb = max_blocks;   // blocks (not triples) the pool must hold
    std::vector<int> orb;
    for (int A = 0; A < nt8; ++A)
        for (int B = A; B < nt8; ++B)
            for (int C = B; C < nt8; ++C) orb.push_back(A | (B << 10) | (C << 20));
    p->norb = (int)orb.size();
    if (metas.empty()) metas.push_back(TripleMeta());
    tab.resize(tab.size() + 128, 0);   // (as tab32 below)
    p->tables = (int64_t*)cx.scratch("t_tables", (int64_t)tab.size());
    p->meta = (TripleMeta*)cx.scratch("t_meta", (int64_t)(metas.size() * sizeof(TripleMeta) / sizeof(double) + 1));
    p->orbits = (int*)cx.scratch("t_orbits", (int64_t)orb.size() / 2 + 1);
    AFESP_HIP(hipMemcpyAsync(p->tables, tab.data(), tab.size() * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
    AFESP_HIP(hipMemcpyAsync(p->meta, metas.data(), metas.size() * sizeof(TripleMeta), hipMemcpyHostToDevice, cx.stream));
    AFESP_HIP(hipMemcpyAsync(p->orbits, orb.data(), orb.size() * sizeof(int), hipMemcpyHostToDevice, cx.stream));
    // group descriptors of the grouped GEMM launches (one launch per chunk): tile counts for the tile shape the launcher
    // will use, table pointers into the uploaded tables
    std::vector<GettGroup> gd;
    for (TriplesPlan::Chunk& ch : p->chunks) {
        const int64_t* tabs = p->tables + ch.tab_off;
      for (int diag = 0; diag < 2; ++diag) {
        (diag ? ch.gdesc_diag_off : ch.gdesc_off) = (int64_t)gd.size();
        int tile = 0, ng = 0, mx = 0;
        int64_t ncol = 0;
        for (const TriplesPlan::Group& g : ch.groups) {
            if ((ch.split_diag && g.q == g.r) != (diag == 1)) continue;
            GettGroup d;
            d.a_off = Kc * v2 * g.r;
            d.offAk = tabs + g.koffA;   // second half: vtT(x, ., ., q)
            d.offBk = tabs + g.koffB;   // second half: tt(x, ., r, p)
            d.offBn = tabs + g.start;
            d.offCn = tabs + ch.ntab + g.start;
            d.N = (int)g.N;
            d.ntiles = (int)((g.N + BN - 1) / BN);
            d.tile_start = tile;
            d.pad = 0;
            tile += mtiles * d.ntiles;
            mx = std::max(mx, d.ntiles);
            ncol += g.N;
            ++ng;
            gd.push_back(d);
        }
        GettGroup end{};
        end.tile_start = tile;
        gd.push_back(end);
        if (diag) { ch.total_tiles_diag = tile; ch.ngroups_diag = ng; ch.max_ntiles_diag = mx; ch.ncol_diag = ncol; }
        else { ch.total_tiles = tile; ch.ngroups_off = ng; ch.max_ntiles = mx; ch.ncol_off = ncol; }
      }
    }
    p->gdesc = (GettGroup*)cx.scratch("t_gdesc", (int64_t)(gd.size() * sizeof(GettGroup) / sizeof(double) + 1));
    AFESP_HIP(hipMemcpyAsync(p->gdesc, gd.data(), gd.size() * sizeof(GettGroup), hipMemcpyHostToDevice, cx.stream));
    std::vector<TgGroup> tg;
    if (use_tg) {
        tab32.resize(tab32.size() + 256, 0u);   // the kernel's table transfers read whole 1-KiB pieces (tgemm.h)
        p->tables32 = (uint32_t*)cx.scratch("t_tables32", (int64_t)(tab32.size() / 2 + 1));
        AFESP_HIP(hipMemcpyAsync(p->tables32, tab32.data(), tab32.size() * sizeof(uint32_t), hipMemcpyHostToDevice, cx.stream));
        const int mt = (int)((v2 + TG_BM - 1) / TG_BM);
        for (TriplesPlan::Chunk& ch : p->chunks) {
            const int64_t* tabs = p->tables + ch.tab_off;
            ch.tg_off = (int64_t)tg.size();
            int mx = 0;
            for (const TriplesPlan::Group& g : ch.groups) mx = std::max(mx, (int)((g.N + TG_BN - 1) / TG_BN));
            const int gm = tgemm_group_m((int)v2, mx);
            int tile = 0;
            // Order: the groups over the whole summation index first, the coinciding-pair groups (half the K steps) behind them.
            // Every tile of a round of the persistent grid then takes the same time -- but for the one round where the two sections
            // meet -- and the workgroups of an XCD keep walking their patch of C in step, i.e. keep finding each other's operand
            // lines in that XCD's L2 (mixed in group order, short and long tiles put the workgroups out of step for good:
            // 56 GB fetched past L2 per launch instead of 13).
            for (int pass = 0; pass < 2; ++pass)
            for (const TriplesPlan::Group& g : ch.groups) {
                const bool half = ch.split_diag && g.q == g.r;
                if (half != (pass == 1)) continue;
                TgGroup d;
                d.a1 = Kc * v2 * g.r;                   // vt(:, ., ., r)
                d.a2 = Kc * v2 * O + Kc * v2 * g.q;     // vtT(:, ., ., q)
                d.b1 = 0;
                d.b2 = Kc * V * ((int64_t)g.r - g.q);   // tt(:, x, r, p) instead of tt(:, x, q, p)
                d.c0 = 0;
                d.colB = p->tables32 + ch.tab32_off + g.start;
                d.offCn = tabs + ch.ntab + g.start;
                d.N = (int)g.N;
                d.ntiles = (int)((g.N + TG_BN - 1) / TG_BN);
                d.tile_start = tile;
                d.nk1 = nk1;
                d.nk = half ? nk1 : 2 * nk1;
                d.inv_width = tgemm_inverse(gm * d.ntiles);
                if ((int64_t)mt * d.ntiles * gm * d.ntiles >= ((int64_t)1 << 32)) throw Error(2, "triples plan: tile walk out of range");
                tile += mt * d.ntiles;
                tg.push_back(d);
            }
            TgGroup end{};
            end.tile_start = tile;
            tg.push_back(end);
            ch.tg_ngroups = (int)ch.groups.size();
            ch.tg_tiles = tile;
            ch.tg_max_ntiles = mx;
        }
        p->tgdesc = (TgGroup*)cx.scratch("t_tgdesc", (int64_t)(tg.size() * sizeof(TgGroup) / sizeof(double) + 1));
        AFESP_HIP(hipMemcpyAsync(p->tgdesc, tg.data(), tg.size() * sizeof(TgGroup), hipMemcpyHostToDevice, cx.stream));
    }
    cx.sync();   // the host vectors die here
    p->epoch = cx.scratch_epoch;
    return p;
}

// Estimated device time (seconds, nominal rates) of the fused evaluation of the flat range [b, e): the grouped GEMM tiles
// of every block triple the range touches (a block triple cut by a range end keeps its pair groups but with fewer
// columns each, so cost is not additive in the range) plus the orbit kernel's reads.  Only ratios matter: it balances
// the shards of triples_shard_bounds.
static double fused_range_cost(int o, int v, int sb, int64_t b, int64_t e, int BM, int BN, int64_t split_min_tiles)
{
    const int64_t V = v, O = o, v2 = V * V, Kc = (V + O + 15) / 16 * 16;
    const int64_t nt8 = (V + TT - 1) / TT, vp3 = nt8 * nt8 * nt8 * CUBE;
    const int64_t mtiles = (v2 + BM - 1) / BM;
    const int nbk = (o + sb - 1) / sb;
    const double half_tile_s = 2.0 * BM * BN * (double)Kc / 58e12;   // one tile over half the summation index, whole device
    const double triple_s = 3.0 * (double)vp3 * 8.0 / 3.9e12;       // orbit kernel: three blocks read once
    double cost = 0.0;
    int64_t flat = 0;
    std::unordered_map<int64_t, int> npair;   // (q,r) -> number of blocks (distinct p)
    std::unordered_map<int64_t, char> seen;
    for (int I = 0; I < nbk && flat < e; ++I)
        for (int J = I; J < nbk && flat < e; ++J)
            for (int K = J; K < nbk && flat < e; ++K) {
                npair.clear();
                seen.clear();
                int64_t nt = 0;
                auto need = [&](int pp, int qq, int rr) {
                    if (qq > rr) std::swap(qq, rr);
                    if (seen.emplace(((int64_t)pp * o + qq) * o + rr, 1).second) ++npair[(int64_t)qq * o + rr];
                };
                for (int i = I * sb; i < std::min(o, (I + 1) * sb); ++i)
                    for (int j = std::max(i, J * sb); j < std::min(o, (J + 1) * sb); ++j)
                        for (int k = std::max(j, K * sb); k < std::min(o, (K + 1) * sb); ++k, ++flat) {
                            if (flat < b || flat >= e) continue;
                            ++nt;
                            need(i, j, k);
                            need(j, i, k);
                            need(k, i, j);
                        }
                if (nt == 0) continue;
                int64_t off = 0, diag = 0;
                for (const auto& g : npair) {
                    const int64_t tiles = mtiles * ((V * g.second + BN - 1) / BN);
                    (g.first / o == g.first % o ? diag : off) += tiles;
                }
                const double halves = 2.0 * (double)off + (diag >= split_min_tiles ? 1.0 : 2.0) * (double)diag;
                // (a launch of fewer tiles than CUs still takes one tile's time; ~40 us of launches per block triple)
                cost += std::max(halves, 512.0) * half_tile_s + (double)nt * triple_s + 40e-6;
            }
    return cost;
}

// Shard boundaries of the flat triple list for `world` ranks, balanced by fused_range_cost: rank r evaluates
// [bounds[r], bounds[r+1]).  Every rank computes the same boundaries (they depend on o, v, the device memory size and
// the tuning environment only).
void triples_shard_bounds(int o, int v, bool cr, int world, int64_t* bounds)
{
    const int64_t V = v, nt = triples_count(o);
    int tm, tn, BM, BN;
    gett_grouped_tile((int)(V * V), true, &tm, &tn, &BM, &BN);
    const int sb = fused_block_size(o, v, cr, device_pool_budget());
    const int64_t split_min_tiles = knobs().t_split_tiles;
    auto cost = [&](int64_t b, int64_t e) { return fused_range_cost(o, v, sb, b, e, BM, BN, split_min_tiles); };
    bounds[0] = 0;
    // A small system is launch-bound: a single triple already costs about as much as the whole list (the estimate's floor of
    // one round of tiles per launch), so there is nothing for the cost to balance -- equal counts then (F2/cc-pVDZ on two
    // ranks would otherwise give rank 0 the one triple (0,0,0)).
    if (cost(0, nt) < (double)world * cost(0, 1)) {
        for (int r = 1; r <= world; ++r) bounds[r] = nt * r / world;
        return;
    }
    for (int r = 0; r < world; ++r) {
        const int64_t b = bounds[r];
        if (r == world - 1 || b >= nt) { bounds[r + 1] = nt; continue; }
        const double target = cost(b, nt) / (double)(world - r);
        int64_t lo = b, hi = nt;   // smallest end whose cost reaches the target
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            if (cost(b, mid) >= target) hi = mid; else lo = mid + 1;
        }
        if (lo > b + 1 && target - cost(b, lo - 1) < cost(b, lo) - target) --lo;   // the nearer of the two
        bounds[r + 1] = lo;
    }
}

int triples_block_size(int o, int v, bool cr) { return fused_block_size(o, v, cr, device_pool_budget()); }

void triples_plan_free(CCState& s)
{
    delete (TriplesPlan*)s.tplan;
    s.tplan = nullptr;
}

void ccsd_triples(Context& cx, CCState& s, int64_t t_begin, int64_t t_end, double* out_host, bool cr, bool want_d)
{
    if (cr) want_d = true;
    if (cr && !s.have_cr) throw Error(1, "ccsd_triples: completely renormalised mode needs ccsd_cr_intermediates first");
    if (!s.ready) throw Error(1, "ccsd_triples: no converged CCSD state in this context");
    const int o = s.o, v = s.v;
    const int64_t O = o, V = v, v2 = V * V, v3 = V * V * V;
    const int64_t nt8 = (V + TT - 1) / TT, vp3 = nt8 * nt8 * nt8 * CUBE;   // stored size of one X block (cube-blocked)
    // the summed extent v+o is padded with zero rows to whole K steps of the GEMM: a ragged last step costs ~8 % of a
    // 14-step tile (measured: K=220 49 TF, K=224 54 TF at M=v^2=40000)
    const int64_t Kc = (V + O + 15) / 16 * 16;
    t_begin = std::max<int64_t>(0, t_begin);
    t_end = std::min<int64_t>(triples_count(o), t_end);
    TriplesPlan* p = plan_fused(cx, s.tplan, s.o, s.v, t_begin, t_end, cr);
    if (knobs().t_debug)
        fprintf(stderr, "afesp (T): o %d v %d block %d chunks %zu kernel %s\n", o, v, p->sblock, p->chunks.size(), p->use_tg ? "tgemm" : "gett");
    // concatenated operands, summed index kappa = [d ; l] first (the reference also moves the summed index first, :2056-2066)
    //   vt(kappa,b,c,k): kappa<v: <cb|kd> = v_vvov(c,b,k,d);  kappa=v+l: t2(l,k,b,c)
    //   tt(kappa,a,j,i): kappa<v: t2(i,j,a,d);                kappa=v+l: -<ij|al> = -v_oovo(i,j,a,l)
    //   vtT(kappa,b,c,k) = vt(kappa,c,b,k) sits right behind vt (slabs o..2o-1): second half of the fused summation index
    Tensor vt = view(cx.scratch("t_vt", 2 * Kc * v2 * O), {Kc, V, V, O}), tt = view(cx.scratch("t_tt", Kc * V * O * O), {Kc, V, O, O});
    Tensor vtT = vt;
    vtT.d = vt.d + Kc * v2 * O;
    auto blocks = [](int64_t n) { return dim3((unsigned)std::min<int64_t>((n + 255) / 256, 65536)); };
    // The operand copies depend on the amplitudes only (and the CR ones on the CR intermediates): a call on unchanged amplitudes
    // -- the next shard of the same (T), the CR evaluation after the plain one, every timed repetition -- finds them in the
    // cached buffers and skips the 3-4 ms of rebuilding them (every shard of a multi-GPU (T) used to pay that).
    const bool ops_valid = cx.t_ops_owner == (const void*)&s && cx.t_ops_amp == s.amp_epoch && cx.t_ops_scratch == cx.scratch_epoch;
    if (!ops_valid) { cx.t_ops_ts = false; cx.t_ops_cr = -1; }
    // v_vvov(c,b,k,d): strides of (b,c,k,d) = (V, 1, V^2, V^2 O);  v_oovo(i,j,a,l): (1, O, O^2, O^2 V)
    if (!ops_valid) {
        const int64_t nblk = ((Kc + 31) / 32) * ((V + 31) / 32) * V * O;
        if (nblk < ((int64_t)1 << 31))
            AFESP_KLAUNCH(triples_build_vt_tiled_kernel, dim3((unsigned)nblk), dim3(256), 0, cx.stream, vt.d, vtT.d, s.v_vvov.d, V, v2,
                               v2 * O, s.t2.d, o, v, (int)Kc);
        else
            AFESP_KLAUNCH(triples_build_vt_kernel, blocks(Kc * v2 * O), dim3(256), 0, cx.stream, vt.d, vtT.d, s.v_vvov.d, V,
                               (int64_t)1, v2, v2 * O, s.t2.d, o, v, (int)Kc);
        AFESP_HIP(hipGetLastError());
        AFESP_KLAUNCH(triples_build_tt_kernel, blocks(Kc * V * O * O), dim3(256), 0, cx.stream, tt.d, s.t2.d, s.v_oovo.d, (int64_t)1, O,
                           O * O, O * O * V, o, v, (int)Kc);
        AFESP_HIP(hipGetLastError());
    }
    Tensor vs = view(cx.scratch("t_vs", v2 * O * O), {V, V, O, O});    // vs(x,y,p,q)  = v_oovv(p,q,x,y)
    Tensor ts = view(cx.scratch("t_ts", v2 * O * O), {V, V, O, O});    // ts(x,y,p,q)  = t2(p,q,x,y)
    if (!ops_valid) permute_add(cx, 1.0, s.v_oovv, "pqxy", 0.0, vs, "xypq");
    if (want_d && !cx.t_ops_ts) {                                      // only y needs the t2 patches
        permute_add(cx, 1.0, s.t2, "pqxy", 0.0, ts, "xypq");
        cx.t_ops_ts = true;
    }
    // completely renormalised mode: the same GEMMs with I_vovv_pp / -I_ooov_pp in place of <cb|kd> / -<ij|al>
    //   vt2(kappa,b,c,k): kappa<v: I_vovv_pp(d,k,b,c);  kappa=v+l: t2(l,k,b,c)
    //   tt2(kappa,a,j,i): kappa<v: t2(i,j,a,d);         kappa=v+l: -I_ooov_pp(j,i,l,a)      (ccsd.f90:2188-2193)
    Tensor vt2, tt2;
    double* Mpool = nullptr;
    if (cr) {
        vt2 = view(cx.scratch("t_vt2", 2 * Kc * v2 * O), {Kc, V, V, O});
        tt2 = view(cx.scratch("t_tt2", Kc * V * O * O), {Kc, V, O, O});
        Tensor vt2T = vt2;
        vt2T.d = vt2.d + Kc * v2 * O;
        if (cx.t_ops_cr != s.cr_epoch) {
            // I_vovv_pp(d,k,b,c): strides of (b,c,k,d) = (V O, V^2 O, V, 1);  I_ooov_pp(j,i,l,a): (i,j,a,l) = (O, 1, O^3, O^2)
            AFESP_KLAUNCH(triples_build_vt_kernel, blocks(Kc * v2 * O), dim3(256), 0, cx.stream, vt2.d, vt2T.d, s.I_vovv_pp.d, V * O,
                               v2 * O, V, (int64_t)1, s.t2.d, o, v, (int)Kc);
            AFESP_HIP(hipGetLastError());
            AFESP_KLAUNCH(triples_build_tt_kernel, blocks(Kc * V * O * O), dim3(256), 0, cx.stream, tt2.d, s.t2.d, s.I_ooov_pp.d, O,
                               (int64_t)1, O * O * O, O * O, o, v, (int)Kc);
            AFESP_HIP(hipGetLastError());
            cx.t_ops_cr = s.cr_epoch;
        }
        Mpool = cx.scratch("t_mpool", p->nb * vp3);
    }
    // (the buffers above may have been allocated just now, which does not move the scratch epoch; dropping any of them does)
    cx.t_ops_owner = (const void*)&s;
    cx.t_ops_amp = s.amp_epoch;
    cx.t_ops_scratch = cx.scratch_epoch;
    const int nq = cr ? 6 : want_d ? 4 : 2;
    // one chunk, few partials, no base term behind it: the sums and their way to the host are one launch (triples_sum_publish_kernel)
    const bool dbase = t_begin == 0 && want_d;
    double pub_seq = 0.0;
    double* pub = (p->chunks.size() == 1 && !dbase && (int64_t)p->norb * p->chunks[0].nt <= 8192) ? host_scalars_slot(cx, &pub_seq) : nullptr;
    if (!pub) k_fill(cx, cx.scal, 6, 0.0);
    TriplesIn in{s.e, s.t1.d, vs.d, ts.d, s.t2.d, o, v};
    double* Xpool = p->pool0 ? p->pool0 : cx.scratch("t_xpool", p->nb * vp3);
    int64_t max_nt = 1;
    for (const TriplesPlan::Chunk& ch : p->chunks) max_nt = std::max<int64_t>(max_nt, ch.nt);
    double* partial = cx.scratch("t_partial", 6 * std::max<int64_t>((int64_t)p->norb * max_nt, 512));
    double* sum_tmp = cx.scratch("t_sum_tmp", 6 * 128);
    // one stream, no host round trip until the four sums are read back: chunk c+1's GEMMs overwrite the X pool only
    // after chunk c's orbit kernel has consumed it (stream order)
    std::vector<hipEvent_t> evs;
    auto stamp = [&]() {
        if (!cx.prof) return;
        hipEvent_t e;
        AFESP_HIP(hipEventCreate(&e));
        AFESP_HIP(hipEventRecord(e, cx.stream));
        evs.push_back(e);
    };
    for (const TriplesPlan::Chunk& ch : p->chunks) {
        stamp();
        {
            // all column groups of the chunk in ONE persistent launch (gett_launch_grouped): no ragged last round and no
            // launch gap per group
            GettProblem gp;
            gp.A = vt.d;
            gp.B = tt.d;
            gp.C = Xpool;
            gp.offAm = p->tables + p->off_Am;
            gp.offAk = gp.offBk = gp.offBn = gp.offCn = nullptr;       // per group
            gp.offCm = p->tables + p->off_Cm;
            gp.M = (int)v2; gp.N = 0; gp.K = (int)(2 * Kc);             // [slab r ; transposed slab q] x [pair (q,p) ; pair (r,p)]
            gp.alpha = 1.0; gp.beta = 0.0;
            gp.nbatch = 1; gp.batchA = gp.batchB = gp.batchC = nullptr;
            gp.a_kcontig = gp.b_kcontig = true;
            gp.wide = true;   // both operands are contiguous along kappa and Kc is a multiple of 16: every row, column and K
                              // offset and every group shift is even, whatever the parity of v
            GettProblem gm = gp;
            gm.A = vt2.d;
            gm.B = tt2.d;
            gm.C = Mpool;
            if (p->use_tg) {
                TgProblem tp{vt.d, tt.d, Xpool, p->tables32, p->tables + p->off_Cm, (int)v2, p->c_pairs, (int)((V + O - (Kc - TG_BK) + 3) / 4)};
                AFESP_HIP(tgemm_launch(tp, p->tgdesc + ch.tg_off, ch.tg_ngroups, ch.tg_tiles, ch.tg_max_ntiles, cx.stream, cx.tg));
                if (cr) {
                    TgProblem tm{vt2.d, tt2.d, Mpool, p->tables32, p->tables + p->off_Cm, (int)v2, p->c_pairs, (int)((V + O - (Kc - TG_BK) + 3) / 4)};
                    AFESP_HIP(tgemm_launch(tm, p->tgdesc + ch.tg_off, ch.tg_ngroups, ch.tg_tiles, ch.tg_max_ntiles, cx.stream, cx.tg));
                }
            } else {
            if (ch.ngroups_off > 0) {
                const GettGroup* gd = p->gdesc + ch.gdesc_off;
                AFESP_HIP(gett_launch_grouped(gp, gd, ch.ngroups_off, ch.total_tiles, ch.max_ntiles, cx.stream));
                if (cr) AFESP_HIP(gett_launch_grouped(gm, gd, ch.ngroups_off, ch.total_tiles, ch.max_ntiles, cx.stream));
            }
            if (ch.ngroups_diag > 0) {   // pairs q == r: the first half of the summation index only
                const GettGroup* gd = p->gdesc + ch.gdesc_diag_off;
                gp.K = gm.K = (int)Kc;
                AFESP_HIP(gett_launch_grouped(gp, gd, ch.ngroups_diag, ch.total_tiles_diag, ch.max_ntiles_diag, cx.stream));
                if (cr) AFESP_HIP(gett_launch_grouped(gm, gd, ch.ngroups_diag, ch.total_tiles_diag, ch.max_ntiles_diag, cx.stream));
            }
            }
            if (cx.prof) {
                cx.prof_gemm_kind = p->use_tg ? 1 : 0;
                // what the tiles execute, zero padding included: whole tiles along both edges, whole K steps -- the LDS-DMA
                // kernel skips the last quarter of a run's last step when it is all padding (tgemm.h, ktail4)
                for (const TriplesPlan::Group& g : ch.groups) {
                    const bool half = ch.split_diag && g.q == g.r;
                    if (p->use_tg) {
                        const double kexec = (double)(Kc - TG_BK) + ((V + O - (Kc - TG_BK) + 3) / 4 < 4 ? 12.0 : 16.0);
                        cx.prof_gemm_flop_padded += 2.0 * (double)((v2 + TG_BM - 1) / TG_BM * TG_BM) * (double)((g.N + TG_BN - 1) / TG_BN * TG_BN) *
                                                    kexec * (half ? 1.0 : 2.0);
                    } else {
                        int tm_, tn_, BM_, BN_;
                        gett_grouped_tile((int)v2, true, &tm_, &tn_, &BM_, &BN_);
                        cx.prof_gemm_flop_padded += 2.0 * (double)((v2 + BM_ - 1) / BM_ * BM_) * (double)((g.N + BN_ - 1) / BN_ * BN_) *
                                                    (double)Kc * (half ? 1.0 : 2.0);
                    }
                }
                cx.prof_gemm_launches += p->use_tg ? 1 : (ch.ngroups_off > 0) + (ch.ngroups_diag > 0);
                cx.prof_gemm_flop += 2.0 * (double)gp.M * (double)(2 * ch.ncol_off + ch.ncol_diag) * (double)(V + O);
            }
        }
        stamp();
        if (cr)
            AFESP_KLAUNCH((triples_orbit_kernel<true, true>), dim3(p->norb, ch.nt), dim3(256), 0, cx.stream, partial, Xpool, Mpool,
                               p->meta + ch.meta_off, p->orbits, in, p->norb * ch.nt);
        else if (!want_d)
            AFESP_KLAUNCH((triples_orbit_kernel<false, true, false>), dim3(p->norb, ch.nt), dim3(256), 0, cx.stream, partial, Xpool,
                               Mpool, p->meta + ch.meta_off, p->orbits, in, p->norb * ch.nt);
        else
            AFESP_KLAUNCH((triples_orbit_kernel<false, true>), dim3(p->norb, ch.nt), dim3(256), 0, cx.stream, partial, Xpool, Mpool,
                               p->meta + ch.meta_off, p->orbits, in, p->norb * ch.nt);
        AFESP_HIP(hipGetLastError());
        stamp();
        if (cx.prof) {
            cx.prof_orbit_launches += 1;
            cx.prof_orbit_bytes += 8.0 * 3.0 * (double)v3 * ch.nt;
        }
        if (pub) {
            AFESP_KLAUNCH(triples_sum_publish_kernel, dim3(1), dim3(256), 0, cx.stream, cx.scal, pub, pub_seq, partial, nq, p->norb * ch.nt);
            AFESP_HIP(hipGetLastError());
        } else {
            sum_partials(cx, cx.scal, partial, nq, p->norb * ch.nt, sum_tmp);
        }
    }
    if (dbase) {
        AFESP_KLAUNCH(triples_dbase_kernel, dim3(256), dim3(256), 0, cx.stream, partial, in, 256);
        AFESP_HIP(hipGetLastError());
        AFESP_KLAUNCH(triples_sum_kernel, dim3(4), dim3(256), 0, cx.stream, cx.scal, partial, 256);
        AFESP_HIP(hipGetLastError());
    }
    double* h = pub ? host_scalars_wait(cx, 6, pub_seq) : host_scalars(cx, 6);
    // (the published sums can be seen before the runtime has marked the stamps in front of them complete: hipEventElapsedTime then says
    // "device not ready" -- seen with six ranks sharing one GPU in a rehearsal, round 5)
    if (!evs.empty()) AFESP_HIP(hipEventSynchronize(evs.back()));
    for (size_t q = 0; q + 2 < evs.size(); q += 3) {
        float a = 0.f, b = 0.f;
        AFESP_HIP(hipEventElapsedTime(&a, evs[q], evs[q + 1]));
        AFESP_HIP(hipEventElapsedTime(&b, evs[q + 1], evs[q + 2]));
        cx.prof_gemm_ms += a;
        cx.prof_orbit_ms += b;
    }
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    out_host[0] = h[0];            // E[T]
    out_host[1] = h[0] + h[1];     // E(T)            ccsd.f90:2220
    if (!want_d) return;
    out_host[2] = h[2];            // D[T]
    out_host[3] = h[2] + h[3];     // D(T)            ccsd.f90:2232
    if (cr) {
        out_host[4] = h[4];            // sum t_bar.M3     ccsd.f90:2223-2224
        out_host[5] = h[4] + h[5];     // + sum z_bar.M3   ccsd.f90:2225
    }
}

// ------------------------------------------------------------------------------------------------ spin-orbital (T)
// do_ccsd_t_spinorb, ccsd.f90:1812-1922.  With
//   Y^{p;qr}(a,b,c) = sum_f <fp||bc> t2(q,r,a,f) - sum_m t2(m,p,c,b) <ma||qr>
// the connected numerator of :1877-1884 is Y^{i;jk} - Y^{j;ik} - Y^{k;ji} = Y^{i;jk} - Y^{j;ik} + Y^{k;ij}; each Y is one
// GEMM over the concatenated index kappa = f (+) m with rows (b,c) and columns (a; pair), exactly the spin-free layout:
//   vt(kappa,b,c,p):  kappa<v: vovv(f,p,b,c);   kappa=v+m: t2(m,p,c,b)
//   tt(kappa,a,r,q):  kappa<v: t2(q,r,a,f);     kappa=v+m: -ovoo(m,a,q,r)
// The summand of :1910 is antisymmetric in (i,j,k): only i<j<k is visited (weight 6/36, in the kernel).
int64_t so_triples_count(int o) { return (int64_t)o * (o - 1) * (o - 2) / 6; }

void so_triples_plan_free(SOState& s)
{
    delete (TriplesPlan*)s.tplan;
    s.tplan = nullptr;
}

// canon_levels_spinorb (ccsd.f90:451-454): spin orbital x carries the energy of spatial orbital x / 2
__global__ void so_levels_kernel(double* e_so, const double* e, int n2)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n2) e_so[x] = e[x / 2];
}

double so_triples(Context& cx, SOState& s, int64_t t_begin, int64_t t_end)
{
    if (!s.ready) throw Error(1, "ccsd_so_triples: no converged spin-orbital CCSD state in this context");
    const int o = s.o, v = s.v;
    const int64_t O = o, V = v, v2 = V * V;
    const int64_t nt8 = (V + TT - 1) / TT, vp3 = nt8 * nt8 * nt8 * CUBE;
    const int64_t Kc = (V + O + 15) / 16 * 16;
    t_begin = std::max<int64_t>(0, t_begin);
    t_end = std::min<int64_t>(so_triples_count(o), t_end);
    k_fill(cx, cx.scal, 1, 0.0);
    if (t_end <= t_begin) return 0.0;
    TriplesPlan* p = plan_for(cx, s.tplan, o, v, t_begin, t_end);
    Tensor vt = view(cx.scratch("t_vt", Kc * v2 * O), {Kc, V, V, O}), tt = view(cx.scratch("t_tt", Kc * V * O * O), {Kc, V, O, O});
    Tensor vs = view(cx.scratch("t_vs", v2 * O * O), {V, V, O, O});    // vs(x,y,p,q) = <pq||xy>
    double* e_so = cx.scratch("t_eso", O + V);
    // The operand copies depend on the amplitudes only: a call on unchanged amplitudes -- the next shard of the same (T), a timed
    // repetition -- finds them in the cached buffers (as the spin-free path does) instead of five permuting passes, two memsets and a
    // round trip of the orbital energies through the host with two stream synchronisations (round 4: every call).
    const bool ops_valid = cx.t_ops_owner == (const void*)&s && cx.t_ops_amp == s.amp_epoch && cx.t_ops_scratch == cx.scratch_epoch;
    if (!ops_valid) {
    auto sub = [&](const Tensor& full, int64_t row0, int64_t nrows) {
        Tensor t = full;
        t.d = full.d + row0;
        t.dim[0] = nrows;
        return t;
    };
    if (Kc != V + O) {
        AFESP_HIP(hipMemsetAsync(vt.d, 0, sizeof(double) * Kc * v2 * O, cx.stream));
        AFESP_HIP(hipMemsetAsync(tt.d, 0, sizeof(double) * Kc * V * O * O, cx.stream));
    }
    permute_add(cx, 1.0, s.vovv, "fpbc", 0.0, sub(vt, 0, V), "fbcp");
    permute_add(cx, 1.0, s.t2, "mpcb", 0.0, sub(vt, V, O), "mbcp");
    permute_add(cx, 1.0, s.t2, "qraf", 0.0, sub(tt, 0, V), "farq");
    permute_add(cx, -1.0, s.ovoo, "maqr", 0.0, sub(tt, V, O), "marq");
    permute_add(cx, 1.0, s.oovv, "pqxy", 0.0, vs, "xypq");
    AFESP_KLAUNCH(so_levels_kernel, dim3((unsigned)((O + V + 255) / 256)), dim3(256), 0, cx.stream, e_so, s.e, (int)(O + V));
    AFESP_HIP(hipGetLastError());
    }
    cx.t_ops_owner = (const void*)&s;
    cx.t_ops_amp = s.amp_epoch;
    cx.t_ops_scratch = cx.scratch_epoch;
    cx.t_ops_ts = false;
    cx.t_ops_cr = -1;
    TriplesIn in{e_so, s.t1.d, vs.d, nullptr, s.t2.d, o, v};
    double* Xpool = cx.scratch("t_xpool", 3 * p->nb * vp3);
    double* partial = cx.scratch("t_partial", std::max<int64_t>((int64_t)p->norb * p->nb, 512));
    for (const TriplesPlan::Chunk& ch : p->chunks) {
        const int64_t* tabs = p->tables + ch.tab_off;
        if (p->use_tg) {
            TgProblem tp{vt.d, tt.d, Xpool, p->tables32, p->tables + p->off_Cm, (int)v2, p->c_pairs, (int)((V + O - (Kc - TG_BK) + 3) / 4)};
            AFESP_HIP(tgemm_launch(tp, p->tgdesc + ch.tg_off, ch.tg_ngroups, ch.tg_tiles, ch.tg_max_ntiles, cx.stream, cx.tg));
        } else
        for (const TriplesPlan::Group& g : ch.groups) {
            GettProblem gp;
            gp.A = vt.d + Kc * v2 * g.r;
            gp.B = tt.d;
            gp.C = Xpool;
            gp.offAm = p->tables + p->off_Am; gp.offAk = p->tables + p->off_k; gp.offBk = p->tables + p->off_k;
            gp.offBn = tabs + g.start;
            gp.offCm = p->tables + p->off_Cm; gp.offCn = tabs + ch.ntab + g.start;
            gp.M = (int)v2; gp.N = (int)g.N; gp.K = (int)Kc;
            gp.alpha = 1.0; gp.beta = 0.0;
            gp.nbatch = 1; gp.batchA = gp.batchB = gp.batchC = nullptr;
            gp.a_kcontig = gp.b_kcontig = true;
            gp.wide = true;   // (as in the spin-free plan: K-contiguous operands, Kc a multiple of 16)
            AFESP_HIP(gett_launch(gp, cx.ws, cx.stream));
        }
        AFESP_KLAUNCH(triples_so_orbit_kernel, dim3(p->norb, ch.nt), dim3(256), 0, cx.stream, partial, Xpool,
                           p->meta + ch.meta_off, p->orbits, in, p->norb * ch.nt);
        AFESP_HIP(hipGetLastError());
        // (two stages, 128 blocks first: one block walking the 67 200 partials of the H2O/cc-pVTZ shape took 0.10 ms of a 2.8 ms evaluation)
        sum_partials(cx, cx.scal, partial, 1, p->norb * ch.nt, cx.scratch("t_sum_tmp", 6 * 128));
    }
    double* h = host_scalars(cx, 1);
    return h[0];
}

void preload_triples()
{
    first_use_touch(reinterpret_cast<const void*>(triples_sum_kernel));
    first_use_touch(reinterpret_cast<const void*>(triples_sum_publish_kernel));
    first_use_touch(reinterpret_cast<const void*>(triples_build_vt_tiled_kernel));
    first_use_touch(reinterpret_cast<const void*>(triples_build_tt_kernel));
    first_use_touch(reinterpret_cast<const void*>(triples_dbase_kernel));
    first_use_touch(reinterpret_cast<const void*>(triples_orbit_kernel<false, true, false>));
    first_use_touch(reinterpret_cast<const void*>(triples_orbit_kernel<false, true, true>));
    first_use_touch(reinterpret_cast<const void*>(triples_orbit_kernel<true, true, true>));
    (void)hipGetLastError();
    preload_tgemm();
}

}  // namespace afesp
