// tall.hip -- streaming fp64 product for tall x skinny contractions on v_mfma_f64_16x16x4_f64 (gfx950).  See tall.h.
//
// Roles: X = the tall operand (T rows, 10^4 ... 10^6), Y = the skinny one (S <= 32 columns), K a few hundred.
//   lane l = (t = l & 15, k = l >> 4) of a wave loads X(t0 + t, k0 + k) -- exactly the register image of the MFMA's A operand
//   (rows t) or B operand (columns t) for one K step of 4, so the element goes from HBM into the fragment register and nowhere else.
//   Y (zero-padded to 16 or 32 columns and to whole chunks of K) and the K-offset table of X live in LDS; one ds_read_b64 per MFMA.
//   TALL_N = false: C's lanes run along s (C's fastest index belongs to the skinny side): D = X-block (16 x 4) . Y-block (4 x 16)
//   TALL_N = true : C's lanes run along t:                                                D = Y-block^T (16 x 4) . X-block^T (4 x 16)
// A wave owns tiles (16 values of t) w, w + nwaves, ... and runs their K range as one stream of chunks of CH loads; two chunks are
// in flight (register double buffer), all loads unconditional from clamped -- always valid -- addresses, so that hipcc counts
// them instead of draining the queue (cf. fused.hip).  Bound: HBM (8 (T K + T S) bytes, the MFMAs at S = 32 need a third of that time).
#include "tall.h"

#include <algorithm>
#include <cstdlib>

namespace afesp {

typedef double tall_v4d __attribute__((ext_vector_type(4)));

struct TallArgs {
    const double* X;
    const double* Y;
    double* C;
    const int64_t *offXt, *offXk, *offYk, *offYs, *offCt, *offCs;
    int T, S, K;
    int nch;      // chunks of CH loads per tile: nch * CH * 4 >= K
    int ntiles;   // ceil(T / 16)
    double alpha, beta;
};

constexpr int TALL_EP = 16;   // tiles of a wave whose row offsets sit in LDS at a time

// One wave's stream: tiles w, w + nwaves, ... of problem `a` (its K-offset table ktab and the image Yl of the skinny operand in LDS,
// rX = this wave's 2 * TALL_EP * 16 table entries)
template <bool TALL_N, int NS, int CH>
__device__ __forceinline__ void tall_stream(const TallArgs& a, const double* Yl, const int64_t* ktab, int64_t* rX, int w, int nwaves, int lane)
{
    constexpr int SP = NS == 1 ? 16 : 48;
    int64_t* rC = rX + TALL_EP * 16;
    const int lt = lane & 15, lk = lane >> 4;
    if (w >= a.ntiles) return;
    const int ntl = (a.ntiles - w + nwaves - 1) / nwaves;   // tiles of this wave: w, w + nwaves, ...
    const int64_t* kt = ktab + lk;
    // column offsets of C: they depend on the lane only
    int64_t cs[NS][TALL_N ? 4 : 1];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int r = 0; r < (TALL_N ? 4 : 1); ++r) cs[j][r] = a.offCs[min(TALL_N ? 16 * j + lk + 4 * r : 16 * j + lt, a.S - 1)];

    tall_v4d acc[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) acc[j] = (tall_v4d){0.0, 0.0, 0.0, 0.0};

    for (int e0 = 0; e0 < ntl; e0 += TALL_EP) {
        const int ne = min(TALL_EP, ntl - e0);
        // (the stream is empty here: the previous epoch's last tile has been stored)
        for (int idx = lane; idx < TALL_EP * 16; idx += 64) {
            const int tile = w + (e0 + min(idx >> 4, ne - 1)) * nwaves;
            const int row = min(tile * 16 + (idx & 15), a.T - 1);
            rX[idx] = a.offXt[row];
            rC[idx] = a.offCt[row];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int G = ne * a.nch;
        int tl_f = 0, ch_f = 0;   // gather cursor: local tile and chunk of the next fetch
        auto fetch = [&](double (&buf)[CH]) {
            const int64_t row = rX[tl_f * 16 + lt];
#pragma unroll
            for (int i = 0; i < CH; ++i) buf[i] = a.X[row + kt[4 * (ch_f * CH + i)]];
            const bool wrap = ch_f + 1 == a.nch;
            ch_f = wrap ? 0 : ch_f + 1;
            tl_f = wrap ? min(tl_f + 1, TALL_EP - 1) : tl_f;   // (behind the epoch's end: a valid address; the data is not used)
        };
        int tl_c = 0, ch_c = 0;   // compute cursor
        auto store_tile = [&]() {
            const int t0 = (w + (e0 + tl_c) * nwaves) * 16;
            // TALL_N = false: register r of accumulator j is C(t0 + lk + 4 r, 16 j + lt); true: C(t0 + lt, 16 j + lk + 4 r)
            if (!TALL_N) {
                int64_t ct[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) ct[r] = rC[tl_c * 16 + lk + 4 * r];
                double old[4][NS];
                if (a.beta != 0.0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < NS; ++j) old[r][j] = a.C[ct[r] + cs[j][0]];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (t0 + lk + 4 * r >= a.T) continue;
#pragma unroll
                    for (int j = 0; j < NS; ++j) {
                        if (16 * j + lt >= a.S) continue;
                        double val = a.alpha * acc[j][r];
                        if (a.beta != 0.0) val += a.beta * old[r][j];
                        a.C[ct[r] + cs[j][0]] = val;
                    }
                }
            } else {
                const int64_t ct = rC[tl_c * 16 + lt];
                double old[NS][4];
                if (a.beta != 0.0) {
#pragma unroll
                    for (int j = 0; j < NS; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) old[j][r] = a.C[ct + cs[j][TALL_N ? r : 0]];
                }
                if (t0 + lt < a.T) {
#pragma unroll
                    for (int j = 0; j < NS; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (16 * j + lk + 4 * r >= a.S) continue;
                            double val = a.alpha * acc[j][r];
                            if (a.beta != 0.0) val += a.beta * old[j][r];
                            a.C[ct + cs[j][TALL_N ? r : 0]] = val;
                        }
                }
            }
        };
        auto compute = [&](const double (&buf)[CH]) {
            const double* yr = Yl + (size_t)(4 * ch_c * CH + lk) * SP + lt;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const double y = yr[(size_t)4 * i * SP + 16 * j];
                    acc[j] = TALL_N ? __builtin_amdgcn_mfma_f64_16x16x4f64(y, buf[i], acc[j], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f64_16x16x4f64(buf[i], y, acc[j], 0, 0, 0);
                }
            }
            if (++ch_c == a.nch) {
                store_tile();
#pragma unroll
                for (int j = 0; j < NS; ++j) acc[j] = (tall_v4d){0.0, 0.0, 0.0, 0.0};
                ch_c = 0;
                ++tl_c;
            }
        };
        double b0[CH], b1[CH];
        fetch(b0);
        for (int g = 0; g < G; g += 2) {
            fetch(b1);
            compute(b0);
            fetch(b0);
            if (g + 1 < G) compute(b1);
        }
    }
}

template <bool TALL_N, int NS, int CH>
__global__ __launch_bounds__(512) void tall_kernel(TallArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double tall_lds[];
    constexpr int SP = NS == 1 ? 16 : 48;   // row stride of the Y image: 32 dwords mod 64, a ds_read_b64 of (4 k) x (16 s) is conflict-free
    const int KR = a.nch * CH * 4;          // rows of the image: K padded to whole chunks
    double* Yl = tall_lds;
    int64_t* ktab = reinterpret_cast<int64_t*>(tall_lds + (size_t)KR * SP);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // per wave: the X row offsets and the C row offsets of its next TALL_EP tiles (read back by other lanes of the same wave only)
    int64_t* rX = ktab + KR + (size_t)wave * (2 * TALL_EP * 16);
    for (int idx = threadIdx.x; idx < KR * 16 * NS; idx += blockDim.x) {
        const int k = idx / (16 * NS), s = idx % (16 * NS);
        Yl[k * SP + s] = (k < a.K && s < a.S) ? a.Y[a.offYk[k] + a.offYs[s]] : 0.0;
    }
    for (int k = threadIdx.x; k < KR; k += blockDim.x) ktab[k] = a.offXk[k < a.K ? k : a.K - 1];
    __syncthreads();
    tall_stream<TALL_N, NS, CH>(a, Yl, ktab, rX, (int)blockIdx.x * 8 + wave, (int)gridDim.x * 8, lane);
}

// Two products that stream the SAME tall array against the SAME skinny matrix, one along each of the array's two leading indices
// (y(j; b,i,a) = sum_e t(j,e) <eb|ia> and x(b; j,i,a) = sum_e <be|ia> t(j,e), src/ccsd.f90:1165-1191, :1275-1290): waves 0-3 of a
// workgroup run problem a (C's lanes along s), waves 4-7 problem b (C's lanes along t), tile for tile side by side -- tile n of either
// is a strip of the same v x v slab, and the workgroups of an XCD take runs of consecutive tiles, so that whichever wave comes second
// finds the slab in that XCD's L2: the array crosses HBM once instead of twice.  One image of the skinny matrix serves both.
template <int NS, int CH>
__global__ __launch_bounds__(512) void tall_dual_kernel(TallArgs a, TallArgs b)
{
    extern __shared__ __attribute__((aligned(16))) double tall_lds[];
    constexpr int SP = NS == 1 ? 16 : 48;
    const int KR = a.nch * CH * 4;
    double* Yl = tall_lds;
    int64_t* ktab = reinterpret_cast<int64_t*>(tall_lds + (size_t)KR * SP);   // [a's | b's]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t* rX = ktab + 2 * KR + (size_t)wave * (2 * TALL_EP * 16);
    for (int idx = threadIdx.x; idx < KR * 16 * NS; idx += blockDim.x) {
        const int k = idx / (16 * NS), s = idx % (16 * NS);
        Yl[k * SP + s] = (k < a.K && s < a.S) ? a.Y[a.offYk[k] + a.offYs[s]] : 0.0;
    }
    for (int k = threadIdx.x; k < KR; k += blockDim.x) {
        ktab[k] = a.offXk[k < a.K ? k : a.K - 1];
        ktab[KR + k] = b.offXk[k < b.K ? k : b.K - 1];
    }
    __syncthreads();
    // workgroup b of the launch runs on XCD b % 8: virtual workgroup numbers that are consecutive inside an XCD
    const int g = (int)gridDim.x, vb = (g & 7) ? (int)blockIdx.x : ((int)blockIdx.x & 7) * (g >> 3) + ((int)blockIdx.x >> 3);
    const int w = vb * 4 + (wave & 3), nw = g * 4;
    if (wave < 4) tall_stream<false, NS, CH>(a, Yl, ktab, rX, w, nw, lane);
    else tall_stream<true, NS, CH>(b, Yl, ktab + KR, rX, w, nw, lane);
}

// chunk length and count for nl = ceil(K / 4) loads per tile: the fewest padded loads, the longer chunk on a tie
static void tall_chunks(int K, int* ch, int* nch)
{
    const int nl = (K + 3) / 4;
    int best = 0, bn = 0, bw = 1 << 30;
    for (int c = 4; c <= 16; ++c) {
        const int n = (nl + c - 1) / c;
        if (n * c <= bw) { bw = n * c; best = c; bn = n; }
    }
    *ch = best;
    *nch = bn;
}

static size_t tall_lds_bytes(int K, int S)
{
    int ch, nch;
    tall_chunks(K, &ch, &nch);
    const size_t KR = (size_t)nch * ch * 4, SP = S <= 16 ? 16 : 48;
    return (KR * (SP + 1) + (size_t)8 * 2 * TALL_EP * 16) * sizeof(double);
}

bool tall_eligible(const GettProblem& p)
{
    const bool off = !knobs().tall;
    if (off || p.nbatch != 1 || p.batchA || p.batchB || p.batchC) return false;
    const int64_t S = std::min(p.M, p.N), T = std::max(p.M, p.N);
    // The lanes of a load are 16 values of t in C's order: they touch whole lines when the tall operand is contiguous along k or
    // along that order of t.  A product that transposes its tall operand (t(j,e) <mb|ie> -> (i,j,m,b): 166 us here against 78
    // through the gather kernel's LDS image) stays with gett_kernel; below ~10^5 rows the two are level (AFESP_TALL_MIN).
    const int64_t tmin = knobs().tall_min;
    const bool tall_n = p.N > p.M;
    const bool coalesced = tall_n ? (p.b_kcontig || p.b_nunit) : (p.a_kcontig || p.a_munit);
    if (S < 1 || S > 32 || T < tmin || T < 64 * S || p.K < 16 || !coalesced) return false;
    return tall_lds_bytes(p.K, (int)S) <= (size_t)150 * 1024;
}

template <bool TALL_N, int NS, int CH>
static hipError_t tall_launch_one(const TallArgs& a, size_t lds, int grid, hipStream_t st)
{
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(tall_kernel<TALL_N, NS, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    AFESP_KLAUNCH((tall_kernel<TALL_N, NS, CH>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}

template <bool TALL_N, int NS>
static hipError_t tall_launch_ch(int ch, const TallArgs& a, size_t lds, int grid, hipStream_t st)
{
    switch (ch) {
#define AFESP_TALL_CASE(c) case c: return tall_launch_one<TALL_N, NS, c>(a, lds, grid, st);
        AFESP_TALL_CASE(4) AFESP_TALL_CASE(5) AFESP_TALL_CASE(6) AFESP_TALL_CASE(7) AFESP_TALL_CASE(8) AFESP_TALL_CASE(9) AFESP_TALL_CASE(10)
        AFESP_TALL_CASE(11) AFESP_TALL_CASE(12) AFESP_TALL_CASE(13) AFESP_TALL_CASE(14) AFESP_TALL_CASE(15) AFESP_TALL_CASE(16)
#undef AFESP_TALL_CASE
    }
    return hipErrorInvalidValue;
}

static TallArgs tall_args(const GettProblem& p, int* ch)
{
    const bool tall_n = p.N > p.M;   // C's lanes (n) run along the tall index
    TallArgs a;
    a.X = tall_n ? p.B : p.A;
    a.Y = tall_n ? p.A : p.B;
    a.C = p.C;
    a.offXt = tall_n ? p.offBn : p.offAm;
    a.offXk = tall_n ? p.offBk : p.offAk;
    a.offYk = tall_n ? p.offAk : p.offBk;
    a.offYs = tall_n ? p.offAm : p.offBn;
    a.offCt = tall_n ? p.offCn : p.offCm;
    a.offCs = tall_n ? p.offCm : p.offCn;
    a.T = tall_n ? p.N : p.M;
    a.S = tall_n ? p.M : p.N;
    a.K = p.K;
    a.alpha = p.alpha;
    a.beta = p.beta;
    tall_chunks(p.K, ch, &a.nch);
    a.ntiles = (a.T + 15) / 16;
    return a;
}
static int tall_cus()
{
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    return cus;
}

hipError_t tall_launch(const GettProblem& p, hipStream_t stream)
{
    if (!tall_eligible(p)) return hipErrorInvalidValue;
    const bool tall_n = p.N > p.M;
    int ch;
    const TallArgs a = tall_args(p, &ch);
    const int cus = tall_cus();
    const size_t lds = tall_lds_bytes(p.K, a.S);
    // (a second workgroup per CU where the LDS images and -- one 16-column fragment -- the registers allow it)
    const int per_cu = (a.S <= 16 && lds <= (size_t)76 * 1024) ? 2 : 1;
    const int grid = std::max(1, std::min(cus * per_cu, (a.ntiles + 7) / 8));
    if (a.S <= 16) return tall_n ? tall_launch_ch<true, 1>(ch, a, lds, grid, stream) : tall_launch_ch<false, 1>(ch, a, lds, grid, stream);
    return tall_n ? tall_launch_ch<true, 2>(ch, a, lds, grid, stream) : tall_launch_ch<false, 2>(ch, a, lds, grid, stream);
}

// ---- two products over one tall array in one launch (tall_dual_kernel)
static size_t tall_dual_lds_bytes(int K, int S)
{
    int ch, nch;
    tall_chunks(K, &ch, &nch);
    return tall_lds_bytes(K, S) + (size_t)nch * ch * 4 * sizeof(int64_t);   // (the second problem's K-offset table)
}

bool tall_dual_eligible(const GettProblem& p1, const GettProblem& p2)
{
    const bool off = !knobs().tall_dual;
    if (off || !tall_eligible(p1) || !tall_eligible(p2)) return false;
    const bool n1 = p1.N > p1.M, n2 = p2.N > p2.M;
    if (n1 == n2) return false;   // one product with C's lanes along the skinny index, one along the tall one
    int c1, c2;
    const TallArgs a = tall_args(p1, &c1), b = tall_args(p2, &c2);
    // the same tall array, the same skinny matrix (the CALLER vouches that both products enumerate its two indices alike: one image
    // of it serves both), the same extents
    if (a.X != b.X || a.Y != b.Y || a.T != b.T || a.S != b.S || a.K != b.K || c1 != c2 || a.nch != b.nch) return false;
    return tall_dual_lds_bytes(p1.K, a.S) <= (size_t)150 * 1024;
}

template <int NS, int CH>
static hipError_t tall_dual_launch_one(const TallArgs& a, const TallArgs& b, size_t lds, int grid, hipStream_t st)
{
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(tall_dual_kernel<NS, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    AFESP_KLAUNCH((tall_dual_kernel<NS, CH>), dim3(grid), dim3(512), lds, st, a, b);
    return hipGetLastError();
}

hipError_t tall_launch_dual(const GettProblem& p1, const GettProblem& p2, hipStream_t stream)
{
    if (!tall_dual_eligible(p1, p2)) return hipErrorInvalidValue;
    const bool first_n = p1.N > p1.M;   // the kernel's problem a has C's lanes along s, b along t
    int ch;
    const TallArgs a = tall_args(first_n ? p2 : p1, &ch), b = tall_args(first_n ? p1 : p2, &ch);
    const size_t lds = tall_dual_lds_bytes(p1.K, a.S);
    int grid = std::max(1, std::min(tall_cus(), (a.ntiles + 3) / 4));
    if (grid >= 8) grid &= ~7;   // (whole eighths: every XCD the same run of tiles)
    switch (ch) {
#define AFESP_TALL_CASE(c) case c: return a.S <= 16 ? tall_dual_launch_one<1, c>(a, b, lds, grid, stream) : tall_dual_launch_one<2, c>(a, b, lds, grid, stream);
        AFESP_TALL_CASE(4) AFESP_TALL_CASE(5) AFESP_TALL_CASE(6) AFESP_TALL_CASE(7) AFESP_TALL_CASE(8) AFESP_TALL_CASE(9) AFESP_TALL_CASE(10)
        AFESP_TALL_CASE(11) AFESP_TALL_CASE(12) AFESP_TALL_CASE(13) AFESP_TALL_CASE(14) AFESP_TALL_CASE(15) AFESP_TALL_CASE(16)
#undef AFESP_TALL_CASE
    }
    return hipErrorInvalidValue;
}

void preload_tall()
{
    first_use_touch(reinterpret_cast<const void*>(tall_kernel<false, 2, 13>));
    (void)hipGetLastError();
}

}  // namespace afesp
