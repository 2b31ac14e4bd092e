// gett.hip -- separable-offset fp64 GEMM on v_mfma_f64_16x16x4_f64 (gfx950).  See gett.h.
//
// Work decomposition (one workgroup = 256 threads = 4 waves in a 2x2 grid):
//   block tile  BM x BN = (32*TM) x (32*TN),  K step BK = 16
//   wave tile   (16*TM) x (16*TN) = TM x TN accumulators of v_mfma_f64_16x16x4_f64
// Per K step a wave issues 4*TM*TN MFMAs (64 cycles each on gfx950: fp64 MFMA runs at 32 FLOP/clk/SIMD)
// against 4*(TM+TN) ds_read_b64, so the matrix pipe, not LDS, is the limiter; the global gather
// for tile t+1 is issued before the MFMAs of tile t and parked in registers (one barrier per K step).
//
// LDS images (padding chosen so that ds_read_b64 fragment reads AND the staging ds_write_b64 are
// bank-conflict free; bank = (byte/4) mod 64 for b64 reads, mod 32 for writes):
//   operand contiguous along m (or n):  s[k][mn], row stride BMN+16 doubles  (stride = 32 mod 64 dwords)
//   operand contiguous along k:         s[mn][k], row stride BK+2  doubles  (36 dwords)
// MFMA operand lane map (f64 16x16x4): lane l supplies A[m = l&15][k = l>>4], B[k = l>>4][n = l&15];
// result register r of lane l is C[m = (l>>4) + 4r][n = l&15]  (cdna_hip_programming.md section 3).
#include "gett.h"

#include <cstdio>
#include <cstdlib>

namespace afesp {

typedef double v4d __attribute__((ext_vector_type(4)));

// Diagnostic builds only (AFESP_GETT_VARIANT bit 64): per (workgroup, wave) cycle sums -- [0] whole stream, [1] parked at the
// step barrier (incl. the LDS-write drain in front of it), [2] the LDS-write block (incl. its wait for the gathered data),
// [3] number of steps.  Read with gett_read_stamps; never part of a timed or shipped build.
static __device__ unsigned long long g_gett_stamp[256 * 8 * 4];   // (one per translation unit: gett.hip, gett_grouped.hip)

constexpr int BK = 16;

#ifndef AFESP_GETT_VARIANT
#define AFESP_GETT_VARIANT 0
#endif
#define AFESP_GETT_VARIANT_ (AFESP_GETT_VARIANT)

typedef double v2d __attribute__((ext_vector_type(2)));

// W = elements moved per load/store instruction along the operand's contiguous direction (1: 8 B, 2: 16 B).
// W = 2 halves the global-load, address-arithmetic and ds_write instruction counts -- on gfx950 the f64 MFMA shares its
// issue/datapath with VALU and VMEM address work, so every instruction removed from the loop is matrix-pipe time won
// (measured: the loop without its loads and LDS stores runs at 71 TF against 49 TF with them).
template <int BMN, bool KC, int NT, int W>
struct TileImg {
    static constexpr int LD = KC ? (BK + 2) : (BMN + 16);
    static constexpr int SIZE = KC ? BMN * LD : BK * LD;
    static constexpr int PER = BMN * BK / (NT * W);   // W-element groups staged per thread per K step
    static constexpr int KG = BK / W, MG = BMN / W;
    __device__ static __forceinline__ int at(int mn, int k) { return KC ? mn * LD + k : k * LD + mn; }
    // staging map: first (mn,k) of the r-th group thread t handles
    __device__ static __forceinline__ int mn_of(int t, int r) { return KC ? (t / KG) + (NT / KG) * r : (t % MG) * W; }
    __device__ static __forceinline__ int k_of(int t, int r) { return KC ? (t % KG) * W : t / MG + (NT / MG) * r; }
};

// Per-thread gather state of one operand.  The K-offsets of step t+2, the data of step t+1 and the MFMAs of step t
// are in flight together: offsets are fetched one step ahead of the data they address, so no load is waited for
// before the MFMAs of the current step have been issued.
// Rows/columns beyond M/N read a clamped (valid) address and produce garbage only in rows/columns of C that are never
// stored; only the K tail has to be zeroed, which stash() does for the single partial step.
template <int BMN, bool KC, int NT, int W>
struct Stager {
    using T = TileImg<BMN, KC, NT, W>;
    static constexpr int NROW = KC ? T::PER : 1;
    static constexpr int NKO = KC ? 1 : T::PER;
    int64_t rowoff[NROW];
    int64_t ko[NKO];
    const double* base;
    const int64_t* offK;
    int K;

    __device__ __forceinline__ void init(const double* X, const int64_t* offMN, const int64_t* offK_, int mn0, int MN,
                                         int K_, int t)
    {
        base = X;
        offK = offK_;
        K = K_;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            int mn = mn0 + T::mn_of(t, r);
            rowoff[r] = offMN[mn < MN ? mn : MN - (KC ? 1 : W)];   // (!KC, W = 2: MN is even, so MN - 2 is the last group)
        }
    }
    // row (or column) offsets of another tile, same operand
    __device__ __forceinline__ void rows(const int64_t* offMN, int mn0, int MN, int t)
    {
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            int mn = mn0 + T::mn_of(t, r);
            rowoff[r] = offMN[mn < MN ? mn : MN - (KC ? 1 : W)];
        }
    }
    // offsets of the K step starting at k0 (clamped: always a valid table entry)
    __device__ __forceinline__ void fetch_ko(int k0, int t)
    {
#pragma unroll
        for (int r = 0; r < NKO; ++r) {
            int k = k0 + T::k_of(t, r);
            ko[r] = offK[k < K ? k : K - W];
        }
    }
    __device__ __forceinline__ void fetch(double (&reg)[T::PER][W]) const
    {
#pragma unroll
        for (int r = 0; r < T::PER; ++r) {
            const double* src = base + rowoff[KC ? r : 0] + ko[KC ? 0 : r];
            if (W == 2) {
                const v2d x = *reinterpret_cast<const v2d*>(src);   // 16-B aligned by the planner's eligibility test
                reg[r][0] = x[0];
                reg[r][W - 1] = x[1];
            } else {
                reg[r][0] = *src;
            }
        }
    }
    __device__ __forceinline__ void stash(double* s, const double (&reg)[T::PER][W], int t, int k0, int kend, bool tail) const
    {
#pragma unroll
        for (int r = 0; r < T::PER; ++r) {
            const int mn = T::mn_of(t, r), k = T::k_of(t, r);
            double x0 = reg[r][0], x1 = reg[r][W - 1];
            if (tail) {
                x0 = (k0 + k < kend) ? x0 : 0.0;
                x1 = (k0 + k + (KC ? W - 1 : 0) < kend) ? x1 : 0.0;
            }
            if (W == 2) *reinterpret_cast<v2d*>(&s[T::at(mn, k)]) = (v2d){x0, x1};
            else s[T::at(mn, k)] = x0;
        }
    }
};

// Values loaded from the group descriptors are the same in every lane; saying so keeps them in scalar registers (as
// vector registers they pushed the grouped kernel from 13 to 31 spilled registers: 57 -> 50 TF).
__device__ __forceinline__ int uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t uniform_i64(int64_t x)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)x), hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
    return ((int64_t)hi << 32) | (unsigned)lo;
}
// (the round trip through the global address space tells the compiler what a pointer read from memory cannot: loads
// through it are global_load, not flat_load -- a flat load also counts on lgkmcnt, so every wait for LDS fragments would
// wait for the table fetch as well)
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p)
{
    typedef const T __attribute__((address_space(1)))* gptr;
    return (const T*)reinterpret_cast<gptr>(uniform_i64(reinterpret_cast<int64_t>(p)));
}

struct GettKernelArgs {
    GettProblem p;
    int ksplit;   // grid.y
    int kchunk;   // K range per split, multiple of BK
    double* ws;   // partial sums [z][split][M][N] when ksplit > 1
    int mtiles, ntiles;
    int gm;       // m-tiles per group (tile walk order)
    int dbg;      // measurement only: bit 0 skips the epilogue stores
    const GettGroup* groups;   // grouped launch (gett.h) or nullptr
    int total_tiles;
    int sk_units;              // stream-K instantiation (SK): K steps per workgroup
    int sk_nk;                 // ... and the steps a tile counts for in the sequence (>= its own: the steps behind K multiply zeros)
};

// XCD-aware bijective remap (cdna_hip_programming.md T1): blocks b, b+8, b+16... share an XCD (and its L2);
// give them consecutive tile ids so that neighbouring tiles -- which share an A or B panel -- hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// GRP = true: grouped launch (gett.h); a separate instantiation so that the plain kernels keep their register allocation
// RAG = false: the K range of every workgroup is whole K steps, which compiles the K-tail masking (8 v_cndmask per
// 16-byte LDS store, 48 of the ~65 VALU instructions of a K step of the 256x128 tile) out of the loop
// TS >= TN: column fragments STAGED per wave (the B image is 16 WN TS columns wide) while TN of them are multiplied -- tiles of 112
// or 96 columns (TN = 7 / 6, TS = 8, one wave column) for the pair products whose 210 / 190 columns fill 82 % / 74 % of two
// 128-column tiles: the staging map needs a power of two, the accumulators do not.
// SK = true: stream-K.  The tiles' K steps form ONE sequence (tile after tile); workgroup w runs steps [w U, (w + 1) U) of it, U =
// sk_units -- the tail of one tile and the head of the next (U < steps per tile: the launcher's condition) -- and every piece goes to
// the partial-sum slab of its rank among the pieces of its tile (gett_reduce_kernel adds them up; the launcher zeroes the slabs that
// only some tiles reach).  For products of a few hundred long tiles -- the pp-ladder's pair products at o = 20, v = 200: 158 tiles of
// 1270 steps are 1.85 rounds of 256 workgroups in three K slices, 7 % of the device idle -- every workgroup gets the same 784 steps.
template <int WM, int WN, int TM, int TN, bool AKC, bool BKC, int W, bool GRP = false, bool RAG = !GRP, int TS = TN, bool SK = false>
// Occupancy bound of the 4-wave tiles: two waves per SIMD, i.e. a budget of 256 registers.  With more than that hipcc keeps
// the accumulators in AGPRs, and on gfx950 v_mfma_f64_16x16x4_f64 with AGPR accumulators issues every 138 cycles instead of
// every 64 (tools/mfma_peak.hip: 34.7 against 77.7 TFLOP/s, one wave per SIMD, sixteen independent accumulators).
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && TM * TN < 16 && !(AFESP_GETT_VARIANT_ & 2048)) ? 2 : 1) void gett_kernel(GettKernelArgs a)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN, BNS = 16 * WN * TS;
    static_assert(TS == TN || (TS > TN && WN == 1), "a narrower tile than its image: one wave column only");
    using TA = TileImg<BM, AKC, NT, W>;
    using TB = TileImg<BNS, BKC, NT, W>;
    __shared__ double lds[2 * (TA::SIZE + TB::SIZE)];
    constexpr int STAGE = TA::SIZE + TB::SIZE;   // buffer b: A image at lds + b*STAGE, B image right behind it

    const GettProblem& p = a.p;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int z = blockIdx.z, split = blockIdx.y;
    const int kbeg = split * a.kchunk;
    const int kend = min(p.K, kbeg + a.kchunk);
    const double* Ab = p.A + (p.batchA ? p.batchA[z] : 0);
    const double* Bb = p.B + (p.batchB ? p.batchB[z] : 0);

    // Persistent workgroups: block b owns tiles b, b + gridDim.x, ... and runs their K steps as ONE stream -- the
    // gathers of the next tile's first steps are issued under the MFMAs of this tile's last steps and the C stores of
    // a finished tile drain under the next tile's MFMAs.  (Measured at M=40000, N=8192, K=224 with one tile per
    // workgroup: 57 us of MFMA per tile, 7 us of exposed first-load latency and 4 us of store drain.)
    // Tile order: a group is `gm` m-tiles x all n-tiles walked m-fastest, so that the ~32 workgroups co-resident on one
    // XCD form a near-square patch of C and both operand panels are re-used out of that XCD's L2; within a round of
    // gridDim.x tiles the XCD remap gives each XCD consecutive ids.
    // Grouped launch: the tile ids of all groups are concatenated; `cursor` is the group of the previous lookup (ids only
    // grow along a workgroup's stream).
    const int ntiles_all = GRP ? a.total_tiles : a.mtiles * a.ntiles;
    // (stream-K: a tile is sk_nk steps of the sequence -- its own, padded so that workgroups eight apart, i.e. on one XCD, start at
    // the same step of their tiles and read the narrow operand's lines together; the steps behind K are masked like a K tail)
    const int nk = SK ? a.sk_nk : (kend - kbeg + BK - 1) / BK;
    const int nk_real = (kend - kbeg + BK - 1) / BK;
    // stream-K: this workgroup's steps [sk_g0, sk_g1) of the sequence, its first tile and the step inside it where it starts
    const int sk_g0 = SK ? (int)blockIdx.x * a.sk_units : 0, sk_g1 = SK ? min(sk_g0 + a.sk_units, ntiles_all * nk) : 0;
    if (SK && sk_g0 >= sk_g1) return;
    const int sk_first = SK ? sk_g0 / nk : 0, sk_kt0 = SK ? sk_g0 - sk_first * nk : 0;
    const int ntl = SK ? (sk_g1 - 1) / nk - sk_first + 1 : (ntiles_all - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    int ctile = 0;   // tiles of this workgroup's stream finished so far
    auto origin = [&](int j, int& m0, int& n0, int& cursor) {
        const int r0 = j * (int)gridDim.x;
        int tile = SK ? sk_first + j : r0 + xcd_remap(blockIdx.x, min((int)gridDim.x, ntiles_all - r0));
        int nt = a.ntiles;
        if (GRP) {
            while (tile >= uniform_i32(a.groups[cursor + 1].tile_start)) ++cursor;
            tile -= uniform_i32(a.groups[cursor].tile_start);
            nt = uniform_i32(a.groups[cursor].ntiles);
        }
        const int width = a.gm * nt, grp = tile / width, first = grp * a.gm;
        const int gsz = min(a.mtiles - first, a.gm), rem = tile - grp * width;
        m0 = (first + rem % gsz) * BM;
        n0 = (rem / gsz) * BN;
    };
    int fgrp = 0, cgrp = 0;   // group of the tile being fetched / being finished

    v4d acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // epilogue: lane l, register r of accumulator (i,j) is C[m0 + wm*16*TM + 16i + (l>>4) + 4r][n0 + ... + (l&15)]
    auto store_tile = [&](int m0, int n0) {
        if (a.dbg & 1) return;
        const int nl = n0 + wn * 16 * TN + (lane & 15);
        const int ml = m0 + wm * 16 * TM + (lane >> 4);
        const int64_t* offCn = GRP ? uniform_ptr(a.groups[cgrp].offCn) : p.offCn;
        const int Ncur = GRP ? uniform_i32(a.groups[cgrp].N) : p.N;
        if (!SK && a.ksplit == 1) {
            double* Cb = p.C + (p.batchC ? p.batchC[z] : 0);
            int64_t cn[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int n = nl + 16 * j;
                cn[j] = offCn[n < Ncur ? n : Ncur - 1];
            }
            // (beta != 0: the 4 TN old values of a 16-row block are requested together, from clamped -- always valid -- addresses,
            // before any of them is needed: element by element, each load stood between a wait and a store that might alias it,
            // 16 TM dependent round trips to memory per tile -- most of the time of a short accumulating product)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int64_t cm[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cm[r] = p.offCm[min(ml + 16 * i + 4 * r, p.M - 1)];
                double old[4][TN];
                if (p.beta != 0.0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < TN; ++j) old[r][j] = Cb[cm[r] + cn[j]];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ml + 16 * i + 4 * r >= p.M) continue;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if (nl + 16 * j >= Ncur) continue;
                        double val = p.alpha * acc[i][j][r];
                        if (p.beta != 0.0) val += p.beta * old[r][j];
                        Cb[cm[r] + cn[j]] = val;
                    }
                }
            }
        } else {
            // (stream-K: the rank of this workgroup among the pieces of the tile -- the first piece starts in workgroup (tile nk) / U)
            const int part = SK ? (int)blockIdx.x - (int)(((int64_t)(sk_first + ctile) * nk) / a.sk_units) : z * a.ksplit + split;
            double* slab = a.ws + (int64_t)part * (int64_t)p.M * p.N;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = ml + 16 * i + 4 * r;
                    if (m >= p.M) continue;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        int n = nl + 16 * j;
                        if (n < p.N) slab[(int64_t)m * p.N + n] = acc[i][j][r];
                    }
                }
        }
    };

    if (nk <= 0) {   // empty contraction: C = beta*C
        for (int j = 0; j < ntl; ++j) {
            int m0, n0;
            origin(j, m0, n0, cgrp);
            store_tile(m0, n0);
        }
        return;
    }
    // the last step of every tile is partial (grouped launches take whole K steps only, gett_launch_grouped)
    const bool ragged = RAG ? ((kend - kbeg) % BK) != 0 : false;
    const int G = SK ? sk_g1 - sk_g0 : ntl * nk;     // K steps in this workgroup's stream

    // Gather cursors.  Data fetches run two stream steps ahead of the MFMAs, their K offsets three steps ahead; when the
    // data cursor leaves a tile the row/column offsets of the next tile are reloaded in place (they are dead until the
    // next fetch, three MFMA groups later).
    Stager<BM, AKC, NT, W> stA;
    Stager<BNS, BKC, NT, W> stB;
    int fkt = sk_kt0, ftile = 0, kok = sk_kt0;
    {
        int m0, n0;
        origin(0, m0, n0, fgrp);
        if (GRP) {
            const GettGroup& g = a.groups[fgrp];
            stA.init(Ab + uniform_i64(g.a_off), p.offAm, uniform_ptr(g.offAk), m0, p.M, p.K, t);
            stB.init(Bb, uniform_ptr(g.offBn), uniform_ptr(g.offBk), n0, uniform_i32(g.N), p.K, t);
        } else {
            stA.init(Ab, p.offAm, p.offAk, m0, p.M, p.K, t);
            stB.init(Bb, p.offBn, p.offBk, n0, p.N, p.K, t);
        }
    }
    auto advance_fetch = [&]() {
        if (++fkt == nk) {
            fkt = 0;
            if (++ftile < ntl) {
                int m0, n0;
                origin(ftile, m0, n0, fgrp);
                stA.rows(p.offAm, m0, p.M, t);
                if (GRP) {
                    const GettGroup& g = a.groups[fgrp];
                    stA.base = Ab + uniform_i64(g.a_off);
                    stA.offK = uniform_ptr(g.offAk);
                    stB.offK = uniform_ptr(g.offBk);
                    stB.rows(uniform_ptr(g.offBn), n0, uniform_i32(g.N), t);
                } else {
                    stB.rows(p.offBn, n0, p.N, t);
                }
            }
        }
    };
    auto next_ko = [&]() {
        stA.fetch_ko(kbeg + kok * BK, t);
        stB.fetch_ko(kbeg + kok * BK, t);
        kok = (kok + 1 == nk) ? 0 : kok + 1;
    };

    // Register ring of depth 2: while the MFMAs of step g run, the data of steps g+1 (set P) and g+2 (set Q) and the
    // offsets of step g+3 are in flight; set P is written to LDS after the MFMAs.  Two K steps (~8k cycles of MFMA at
    // TM=TN=4) cover the loaded HBM latency with a single wave per SIMD.
    double ra0[TA::PER][W], rb0[TB::PER][W], ra1[TA::PER][W], rb1[TB::PER][W];
    next_ko();
    stA.fetch(ra0);
    stB.fetch(rb0);
    advance_fetch();
    next_ko();
    stA.stash(lds, ra0, t, kbeg + sk_kt0 * BK, kend, SK ? sk_kt0 >= nk_real - 1 : ragged && sk_kt0 == nk - 1);
    stB.stash(lds + TA::SIZE, rb0, t, kbeg + sk_kt0 * BK, kend, SK ? sk_kt0 >= nk_real - 1 : ragged && sk_kt0 == nk - 1);
    stA.fetch(ra1);                       // step 1 (stale but valid addresses if it does not exist)
    stB.fetch(rb1);
    advance_fetch();
    next_ko();
    __syncthreads();

    const int fa = wm * 16 * TM + (lane & 15), fb = wn * 16 * TN + (lane & 15), fk = lane >> 4;
    // Fragment registers are double-buffered too: the ds_reads of K sub-step s+1 are issued before the MFMAs of
    // sub-step s, and the reads of the NEXT step's sub-step 0 are issued right after the barrier, underneath the MFMAs
    // of this step's last sub-step -- so no MFMA ever waits for LDS latency except in the prologue.
    double af0[TM], bf0[TN], af1[TM], bf1[TN];
    auto frag = [&](double (&af)[TM], double (&bf)[TN], const double* cA, const double* cB, int s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = cA[TA::at(fa + 16 * i, 4 * s + fk)];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = cB[TB::at(fb + 16 * j, 4 * s + fk)];
    };
    auto mfma = [&](const double (&af)[TM], const double (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    };
    frag(af0, bf0, lds, lds + TA::SIZE, 0);
    // One stream step.  On entry set 0 holds sub-step 0 of LDS buffer `cur`.  Register set (pa,pb) holds the gathered
    // data of step g+1 and is written to the other LDS buffer; set (qa,qb) receives step g+2.  The single barrier sits
    // between sub-steps 2 and 3: every read of buffer `cur` has been issued before it and every write of buffer cur^1 is
    // complete, so after it the next step's first fragments can be read while sub-step 3 still multiplies.
    // Two waves share each SIMD in the 8-wave tiles, and the younger half (waves 4-7) loses VALU arbitration to the older
    // one at equal priority (MI355X_MICROARCH.md, two waves per SIMD): one static raise, no per-cluster flips
    // (A/B in one session, o=20 v=200: ring 61.3 -> 62.0 TF, pp-ladder 49.5 -> 50.0 TF; flips around every MFMA
    // cluster instead: -1 %).
    if (AFESP_GETT_VARIANT_ & 4) {   // (a real raise for waves 4-7 only: the guard must be provably wave-uniform)
        if (NT == 512 && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_setprio(1);
    } else if (NT == 512 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    int kt = sk_kt0;
// Schedule variants of one stream step (compile-time, A/B-measured in one GPU session: tools/ab_gemm.py):
//   EARLY   the data of step g+1 is written to LDS right after the barrier that freed its buffer, so the whole step's
//           MFMAs cover the ds_write completion the next barrier waits for; otherwise ("late") just before sub-step 2
//   PIN     keep sub-step 2's MFMAs in front of the barrier (hipcc otherwise sinks them behind it, and the matrix pipe
//           then idles from the first ds_write until the last wave has passed the barrier)
#ifndef AFESP_GETT_VARIANT
#define AFESP_GETT_VARIANT 0
#endif
    constexpr int VARIANT = AFESP_GETT_VARIANT;
    constexpr bool PIN = (VARIANT & 1) != 0;
    constexpr bool STAGGER = (VARIANT & 8) ? true : (VARIANT & 512) ? false : GRP;   // default: the grouped kernels only (A/B below)
    unsigned long long stamp_bar = 0, stamp_stash = 0, stamp_n = 0, stamp_n2 = 0;
    const unsigned long long stamp_t0 = (VARIANT & 64) ? __builtin_amdgcn_s_memtime() : 0;
#define AFESP_GETT_STASH(pa, pb)                                                                \
        if (st) {                                                                               \
            unsigned long long s0_ = 0;                                                         \
            if (VARIANT & 64) { __builtin_amdgcn_sched_barrier(0); s0_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } \
            if (VARIANT & 128) { __builtin_amdgcn_s_waitcnt(0x0070); __builtin_amdgcn_sched_barrier(0); stamp_n2 += __builtin_amdgcn_s_memtime() - s0_; __builtin_amdgcn_sched_barrier(0); } \
            stA.stash(lds + (cur ^ 1) * STAGE, pa, t, kbeg + ktn * BK, kend, tail);             \
            stB.stash(lds + (cur ^ 1) * STAGE + TA::SIZE, pb, t, kbeg + ktn * BK, kend, tail);  \
            if (VARIANT & 64) { __builtin_amdgcn_sched_barrier(0); stamp_stash += __builtin_amdgcn_s_memtime() - s0_; __builtin_amdgcn_sched_barrier(0); } \
        }
#define AFESP_GETT_STEP(qa, qb, pa, pb, EARLY) AFESP_GETT_STEP_(qa, qb, pa, pb, EARLY, false)
#define AFESP_GETT_STEP_(qa, qb, pa, pb, EARLY, STEADY)                                         \
    {                                                                                           \
        const int cur = g & 1;                                                                  \
        const double* cA = lds + cur * STAGE;                                                   \
        const double* cB = cA + TA::SIZE;                                                       \
        const bool ld = (STEADY) || g + 2 < G, st = (STEADY) || g + 1 < G;                      \
        const int ktn = (kt + 1 == nk) ? 0 : kt + 1;                                            \
        const bool tail = SK ? ktn >= nk_real - 1 : ragged && (ktn == nk - 1);                  \
        if (EARLY) { AFESP_GETT_STASH(pa, pb) }                                                 \
        frag(af1, bf1, cA, cB, 1);                                                              \
        if (ld) stA.fetch(qa);                                                                  \
        mfma(af0, bf0);                                                                         \
        frag(af0, bf0, cA, cB, 2);                                                              \
        if (ld) {                                                                               \
            stB.fetch(qb);                                                                      \
            advance_fetch();                                                                    \
        }                                                                                       \
        mfma(af1, bf1);                                                                         \
        frag(af1, bf1, cA, cB, 3);                                                              \
        if (ld) next_ko();                                                                      \
        if (!(EARLY)) { AFESP_GETT_STASH(pa, pb) }                                              \
        mfma(af0, bf0);                                                                         \
        if (PIN) __builtin_amdgcn_sched_barrier(0);                                             \
        if (VARIANT & 64) {                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                  \
            const unsigned long long b0_ = __builtin_amdgcn_s_memtime();                        \
            __syncthreads();                                                                    \
            stamp_bar += __builtin_amdgcn_s_memtime() - b0_;                                    \
            ++stamp_n;                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                  \
        } else                                                                                  \
        __syncthreads();                                                                        \
        if (st) frag(af0, bf0, lds + (cur ^ 1) * STAGE, lds + (cur ^ 1) * STAGE + TA::SIZE, 0); \
        mfma(af1, bf1);                                                                         \
        if (ktn == 0) {                                                                         \
            int m0, n0;                                                                         \
            origin(ctile, m0, n0, cgrp);                                                        \
            store_tile(m0, n0);                                                                 \
            ++ctile;                                                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0}; \
        }                                                                                       \
        kt = ktn;                                                                               \
    }
#define AFESP_GETT_LOOP(EARLY)                                                                  \
    {                                                                                           \
        int g = 0;                                                                              \
        if (!GRP || (VARIANT & 4096))   /* (the grouped kernel spills 49 registers with this second body: A/B 518 -> 591 ms) */ \
            for (; g + 3 < G; g += 2) {   /* steady state: every step gathers and writes, no flags (+1 % on the plain tiles) */ \
                AFESP_GETT_STEP_(ra0, rb0, ra1, rb1, EARLY, true)                               \
                ++g;                                                                            \
                AFESP_GETT_STEP_(ra1, rb1, ra0, rb0, EARLY, true)                               \
                --g;                                                                            \
            }                                                                                   \
        for (; g + 1 < G; g += 2) {                                                             \
            AFESP_GETT_STEP(ra0, rb0, ra1, rb1, EARLY)                                          \
            ++g;                                                                                \
            AFESP_GETT_STEP(ra1, rb1, ra0, rb0, EARLY)                                          \
            --g;                                                                                \
        }                                                                                       \
        if (g < G) AFESP_GETT_STEP(ra0, rb0, ra1, rb1, EARLY)                                   \
    }
    // Stagger (8-wave tiles): the two waves of a SIMD run the same program between the same barriers, so left alone they
    // reach their LDS writes, their waits and the barrier together and the matrix pipe has nothing to do meanwhile.
    // Waves 4-7 therefore write early and waves 0-3 late: one half's non-matrix phase lies under the other half's MFMAs.
    if (STAGGER && NT == 512) {
        // one loop body, the position of the LDS writes chosen by a wave-uniform flag (two copies of the whole loop made
        // hipcc spill 150-200 registers)
        const bool early_half = (VARIANT & 16) ? __builtin_amdgcn_readfirstlane(wave) < 4 : __builtin_amdgcn_readfirstlane(wave) >= 4;
        AFESP_GETT_LOOP(early_half)
    } else if (VARIANT & 2) {
        AFESP_GETT_LOOP(true)
    } else {
        AFESP_GETT_LOOP(false)
    }
#undef AFESP_GETT_LOOP
#undef AFESP_GETT_STASH
    if (SK && kt != 0) {   // the stream ended inside a tile: its piece
        int m0, n0;
        origin(ctile, m0, n0, cgrp);
        store_tile(m0, n0);
    }
    if ((VARIANT & 64) && lane == 0 && blockIdx.x < 256 && wave < 8 && blockIdx.y == 0 && blockIdx.z == 0) {
        unsigned long long* d = g_gett_stamp + ((int)blockIdx.x * 8 + wave) * 4;
        d[0] = __builtin_amdgcn_s_memtime() - stamp_t0;
        d[1] = stamp_bar;
        d[2] = stamp_stash;
        d[3] = (VARIANT & 128) ? (stamp_n | ((stamp_n2 / (stamp_n ? stamp_n : 1)) << 20)) : stamp_n;
    }
#undef AFESP_GETT_STEP
}

#ifndef AFESP_GETT_GROUPED_TU
// Deterministic split-K combine: fixed summation order over the split index.  The slab loads of one element are
// independent (issued eight at a time), only the adds are ordered; 32-bit index arithmetic whenever M*N allows it (the
// 64-bit division alone was a third of this kernel's 10 us on the 25 x 2809 outputs of the o=5, v=53 iteration).
__global__ __launch_bounds__(256) void gett_reduce_kernel(GettKernelArgs a)
{
    const GettProblem& p = a.p;
    const int64_t mn = (int64_t)p.M * p.N;
    const int z = blockIdx.y;
    double* Cb = p.C + (p.batchC ? p.batchC[z] : 0);
    const double* W = a.ws + (int64_t)z * a.ksplit * mn;
    const bool small = mn <= 0x7fffffff;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < mn; x += (int64_t)gridDim.x * blockDim.x) {
        int m, n;
        if (small) {
            m = (int)((unsigned)x / (unsigned)p.N);
            n = (int)((unsigned)x - (unsigned)m * (unsigned)p.N);
        } else {
            m = (int)(x / p.N);
            n = (int)(x % p.N);
        }
        double* dst = Cb + p.offCm[m] + p.offCn[n];
        double s = 0.0;
        int k = 0;
        for (; k + 8 <= a.ksplit; k += 8) {
            double w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = W[(int64_t)(k + q) * mn + x];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += w[q];
        }
        for (; k < a.ksplit; ++k) s += W[(int64_t)k * mn + x];
        double val = p.alpha * s;
        if (p.beta != 0.0) val += p.beta * *dst;
        *dst = val;
    }
}

int g_dbg = 0;

// Every instantiation the launchers can pick (the lists in gett_launch / gett_launch_grouped): the runtime resolves a kernel
// function on its first use (~0.7 ms each), which adds up to 30 ms in the first iteration of a small molecule.
template <int WM, int WN, int TM, int TN, int W>
static void preload_cfg()
{
    first_use_touch(reinterpret_cast<const void*>(gett_kernel<WM, WN, TM, TN, true, true, W>));
    first_use_touch(reinterpret_cast<const void*>(gett_kernel<WM, WN, TM, TN, true, false, W>));
    first_use_touch(reinterpret_cast<const void*>(gett_kernel<WM, WN, TM, TN, false, true, W>));
    first_use_touch(reinterpret_cast<const void*>(gett_kernel<WM, WN, TM, TN, false, false, W>));
}
void preload_gett()
{
    first_use_touch(reinterpret_cast<const void*>(gett_reduce_kernel));
    // the small tiles first: they are what a small molecule launches within milliseconds of the context's creation
    preload_cfg<2, 2, 1, 1, 1>(); preload_cfg<2, 2, 1, 2, 1>(); preload_cfg<2, 2, 1, 4, 1>();
    preload_cfg<2, 2, 2, 1, 1>(); preload_cfg<2, 2, 2, 2, 1>(); preload_cfg<2, 2, 2, 2, 2>(); preload_cfg<2, 2, 2, 4, 1>(); preload_cfg<2, 2, 2, 4, 2>();
    preload_cfg<2, 2, 4, 1, 1>(); preload_cfg<2, 2, 4, 2, 1>(); preload_cfg<2, 2, 4, 2, 2>();
    preload_cfg<2, 4, 4, 2, 1>(); preload_cfg<2, 4, 4, 2, 2>();
    preload_gett_grouped();   // (gett_grouped.hip)
    preload_cfg<4, 2, 4, 4, 1>(); preload_cfg<4, 2, 4, 4, 2>(); preload_cfg<2, 4, 4, 4, 1>(); preload_cfg<2, 4, 4, 4, 2>();
    (void)hipGetLastError();
}

hipError_t gett_read_stamps(unsigned long long* out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gett_stamp), sizeof(unsigned long long) * (size_t)(n < 256 * 8 * 4 ? n : 256 * 8 * 4));
}

#else
extern int g_dbg;
// the grouped instantiations (the lists in gett_launch_grouped)
void preload_gett_grouped()
{
    first_use_touch(reinterpret_cast<const void*>(gett_kernel<2, 4, 4, 2, true, true, 2, true>));
    first_use_touch(reinterpret_cast<const void*>(gett_kernel<2, 4, 4, 2, true, true, 1, true>));
    first_use_touch(reinterpret_cast<const void*>(gett_kernel<4, 2, 4, 4, true, true, 2, true>));
    (void)hipGetLastError();
}
// diagnostic builds: the stamps of the grouped kernels live in this translation unit
hipError_t gett_read_stamps_grouped(unsigned long long* out, int n)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gett_stamp), sizeof(unsigned long long) * (size_t)(n < 256 * 8 * 4 ? n : 256 * 8 * 4));
}
#endif

// Resident workgroups the device holds for one kernel instantiation (CUs x occupancy), found once per instantiation.
template <typename Kern>
static int resident_blocks(Kern kern, int threads)
{
    int dev = 0, cus = 256, occ = 1;
    if (hipGetDevice(&dev) != hipSuccess) return cus;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    {   // (the occupancy query resolves the function: a first use like any other, first_use.h)
        std::lock_guard<std::mutex> lk(first_use_mutex());
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, 0) != hipSuccess || occ <= 0) occ = 1;
    }
    return cus * occ;
}

template <int WM, int WN, int TM, int TN, bool AK, bool BK_, int W, bool GRP = false, bool RAG = !GRP, int TS = TN, bool SK = false>
static void launch_one(const GettKernelArgs& a, dim3 grid, hipStream_t st)
{
    if (SK) {   // (stream-K: the launcher has sized the grid)
        AFESP_KLAUNCH((gett_kernel<WM, WN, TM, TN, AK, BK_, W, GRP, RAG, TS, SK>), grid, dim3(64 * WM * WN), 0, st, a);
        return;
    }
    static const int cap = resident_blocks(gett_kernel<WM, WN, TM, TN, AK, BK_, W, GRP, RAG, TS>, 64 * WM * WN);
    // persistent grid: no more workgroups than the device holds at once, the rest of the tiles are walked in-kernel
    int per = cap / (int)(grid.y * grid.z);
    if (per < 1) per = 1;
    if (g_dbg & 2) per = 1 << 30;   // measurement only: one tile per workgroup
    if (g_dbg & 4) per = per / 4 > 0 ? per / 4 : 1;   // measurement only: a quarter of the device (tools/burst_probe.py)
    if ((int)grid.x > per) grid.x = (unsigned)per;
    AFESP_KLAUNCH((gett_kernel<WM, WN, TM, TN, AK, BK_, W, GRP, RAG, TS>), grid, dim3(64 * WM * WN), 0, st, a);
}

template <int WM, int WN, int TM, int TN, int W, int TS = TN>
static void launch_cfg(const GettKernelArgs& a, dim3 grid, hipStream_t st)
{
    const bool ak = a.p.a_kcontig, bk = a.p.b_kcontig;
    // (RAG = false for the plain 8-wave tiles with whole K steps was A/B-measured at -2...3 % on the ring, pp-ladder and
    // K=224/3520 products although it removes 48 VALU instructions per step and spills nothing: the plain kernels keep the
    // masking; in the grouped kernel the same change gains 4 %)
    if (ak && bk) launch_one<WM, WN, TM, TN, true, true, W, false, true, TS>(a, grid, st);
    else if (ak) launch_one<WM, WN, TM, TN, true, false, W, false, true, TS>(a, grid, st);
    else if (bk) launch_one<WM, WN, TM, TN, false, true, W, false, true, TS>(a, grid, st);
    else launch_one<WM, WN, TM, TN, false, false, W, false, true, TS>(a, grid, st);
}

// (stream-K instantiations: the 256 x 112 / 256 x 96 tiles with 16-byte staging only -- the pair products of the pp-ladder)
template <int WM, int WN, int TM, int TN, int W, int TS>
static void launch_cfg_sk(const GettKernelArgs& a, dim3 grid, hipStream_t st)
{
    const bool ak = a.p.a_kcontig, bk = a.p.b_kcontig;
    if (ak && bk) launch_one<WM, WN, TM, TN, true, true, W, false, true, TS, true>(a, grid, st);
    else if (ak) launch_one<WM, WN, TM, TN, true, false, W, false, true, TS, true>(a, grid, st);
    else if (bk) launch_one<WM, WN, TM, TN, false, true, W, false, true, TS, true>(a, grid, st);
    else launch_one<WM, WN, TM, TN, false, false, W, false, true, TS, true>(a, grid, st);
}

#ifdef AFESP_GETT_GROUPED_TU
extern int g_group_m, g_allow_wide;
#else
int g_group_m = 0;   // >0 overrides the tile-walk group size (tuning knob, see afesp_set_tuning)
int g_force_tm = 0, g_force_tn = 0, g_force_split = 0, g_allow_wide = 1;
// K is sliced when a product has fewer tiles than this (tuning knob AFESP_SPLIT_BELOW)
#define g_split_below (knobs().split_below)
// ... into slices of at least this many K steps (tuning knob AFESP_SPLIT_MIN_STEPS)
#define g_split_min_steps (knobs().split_min_steps)

// Block tile extent (rows or columns) for a requested code: 1 -> 32, 2 -> 64, 4 -> 128; 8 = 128 with 8 waves.
static int pick_t(int extent)
{
    if (extent <= 32) return 1;
    if (extent <= 64) return 2;
    return 4;
}

hipError_t gett_launch(const GettProblem& p, const GettWorkspace& ws, hipStream_t stream, int force_split, int force_tm,
                       int force_tn)
{
    if (p.M <= 0 || p.N <= 0 || p.nbatch <= 0) return hipSuccess;
    GettKernelArgs a;
    a.p = p;
    if (!force_tm) force_tm = g_force_tm;
    if (!force_tn) force_tn = g_force_tn;
    if (!force_split) force_split = g_force_split;
    // 16-byte staging pairs two elements along an operand's contiguous direction: K for a K-contiguous operand (K even),
    // otherwise its row / column index (M or N even)
    const bool wide = p.wide && g_allow_wide && (p.a_kcontig || p.M % 2 == 0) && (p.b_kcontig || p.N % 2 == 0) && (p.K % 2 == 0) &&
                      p.nbatch == 1;
    int tm = force_tm ? force_tm : pick_t(p.M), tn = force_tn ? force_tn : pick_t(p.N);
    // tall problems with 16-byte staging: the 256x128 tile (8 waves, 4x4 MFMA grid per wave) halves the staging
    // instructions per MFMA once more (62 TF against 55 TF for 128x128 on the o^3 v^3 ring contraction at o=20, v=200)
    const int ksteps = (p.K + BK - 1) / BK;
    int wq_split = 0;
    if (!force_tm && !force_tn && tm == 4 && tn == 4 && wide && p.M >= 2048) {
        // Wave quantisation: 256 CUs take one workgroup each, so a grid of 316 tiles runs as 256 + 60.  Score the two tile
        // shapes with 1..4 K slices by (relative tile speed) x (fill of the last round) and keep the best.
        auto score = [&](int bm, int bn, int s, double rate, int slots) {
            const int64_t work = (int64_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.nbatch * s;
            const int64_t rounds = (work + slots - 1) / slots;
            const double waste = (double)((p.M + bm - 1) / bm * bm) * ((p.N + bn - 1) / bn * bn) / ((double)p.M * p.N);
            return rate * (double)work / (double)(rounds * slots) / waste * (s > 1 ? 1.0 - 0.012 * s : 1.0);   // (a slab to write and to sum per slice)
        };
        // (K slices offered: 1 ... 4, and up to 16 for products of few tiles and a long summation index -- a rank's row range of the
        // pp-ladder in a split iteration is 20 tiles of 1257 K steps: four slices left two thirds of the device idle)
        static const int slice_opts[] = {1, 2, 3, 4, 6, 8, 12, 16};
        double best = 0.0;
        for (int big = 0; big < 2; ++big)
            for (int s : slice_opts) {
                if (s > 1 && (force_split > 0 || ksteps / s < 32)) continue;
                const double sc = score(big ? 256 : 128, 128, s, big ? 1.0 : 0.89, 256);
                if (sc > best * 1.02) { best = sc; tm = big ? 16 : 4; tn = big ? 8 : 4; wq_split = s; }
            }
        // A few columns past a multiple of 128 leave most of the last column tile empty (the antisymmetric pair product of the
        // pp-ladder at o = 20: 190 columns): the 4-wave 128 x 64 tile (two workgroups per CU) then wins although it runs at 0.82
        // of the big tile's rate (3.41 -> 3.11 ms there).  A short product is mostly tile prologue and epilogue, where the small tile
        // loses nothing (I_oooo(ijmn) c(mnab) -> r2, K = 400, 400 columns: 321 -> 274 us); in between it is not offered.
        if (ksteps >= 256 || ksteps < 64)
            for (int s : slice_opts) {
                if (s > 1 && (force_split > 0 || ksteps / s < 32)) continue;
                const double sc = score(128, 64, s, ksteps >= 256 ? 0.82 : 0.97, 512);
                if (sc > best * 1.02) { best = sc; tm = 4; tn = 2; wq_split = s; }
            }
        // 256 x 112 / 256 x 96 (eight waves one above the other, 2 x 7 / 2 x 6 accumulators each under a 128-column image): the
        // same staging for 7/8 resp. 6/8 of the MFMAs, i.e. 0.93 of the big tile's rate on what it multiplies -- worth it where
        // the columns end just past a multiple of 112 or 96 (the pair products at o = 20: 210 = 2 x 112 - 14, 190 = 2 x 96 - 2;
        // tools/ladder_tiles.py: 6.87 -> 6.21 ms per ladder)
        for (int cols = 112; cols >= 96; cols -= 16)
            for (int s : slice_opts) {
                if (s > 1 && (force_split > 0 || ksteps / s < 32)) continue;
                const double sc = score(256, cols, s, 0.93, 256);
                if (sc > best * 1.02) { best = sc; tm = 16; tn = cols / 16; wq_split = s; }
            }
    }
    if (tn == 6 || tn == 7) tm = 16;   // (the narrow tiles exist under 256 rows only)
    const int BM = tm == 16 ? 256 : tm == 8 ? 128 : 32 * tm, BN = tn == 16 ? 256 : tn == 8 ? 128 : tn == 7 ? 112 : tn == 6 ? 96 : 32 * tn;
    a.mtiles = (p.M + BM - 1) / BM;
    a.ntiles = (p.N + BN - 1) / BN;
    const int64_t tiles = (int64_t)a.mtiles * a.ntiles * p.nbatch;
    int split = 1;
    if (force_split > 0) {
        split = force_split;
    } else if (wq_split > 1) {
        split = wq_split;
    } else if (tiles < g_split_below && ksteps >= 2 * g_split_min_steps) {
        // too few tiles to fill 256 CUs: slice K, at least 4 K steps per slice, aim for ~2 blocks per CU
        // (8 per CU for the 32 x 32 tile of a long-K product was measured: 108 -> 170 us for 20 x 20 results over 800 000 summed indices)
        split = (int)((512 + tiles - 1) / tiles);
        if (split > ksteps / g_split_min_steps) split = ksteps / g_split_min_steps;
    } else if (tiles < 1024 && ksteps >= 64 && (int64_t)p.M * p.N * 16 <= ((int64_t)p.M + p.N) * p.K) {
        // a long product that mostly streams an operand (its output is small change beside it): a workgroup per CU has too few
        // loads in flight for the HBM rate -- w(e,b,m,a) t(m,e) -> I_vv(b,a) at o = 20, v = 200: 313 tiles 0.42 ms, in four K slices 0.27
        split = (int)((1024 + tiles - 1) / tiles);
        if (split > ksteps / 8) split = ksteps / 8;
    }
    if (split > ksteps) split = ksteps > 0 ? ksteps : 1;
    const int64_t need = (int64_t)split * p.nbatch * p.M * p.N * (int64_t)sizeof(double);
    if (split > 1 && (ws.ptr == nullptr || (size_t)need > ws.bytes)) {
        split = (ws.ptr == nullptr) ? 1 : (int)(ws.bytes / ((size_t)p.nbatch * p.M * p.N * sizeof(double)));
        if (split < 1) split = 1;
    }
    int steps_per = (ksteps + split - 1) / split;
    if (steps_per < 1) steps_per = 1;
    a.kchunk = steps_per * BK;
    a.ksplit = (ksteps + steps_per - 1) / steps_per;
    if (a.ksplit < 1) a.ksplit = 1;
    a.ws = ws.ptr;
    a.dbg = g_dbg;
    a.groups = nullptr;
    a.total_tiles = 0;
    a.sk_units = 0;
    a.sk_nk = 0;
    a.gm = g_group_m > 0 ? g_group_m : (a.ntiles >= 8 ? 4 : a.ntiles >= 4 ? 8 : a.ntiles >= 2 ? 16 : 32);
    if (a.gm > a.mtiles) a.gm = a.mtiles;
    dim3 grid((unsigned)(a.mtiles * a.ntiles), (unsigned)a.ksplit, (unsigned)p.nbatch);
    // Stream-K (gett_kernel, SK) for the 256 x 112 / 96 tiles where the slices chosen above leave a round of the device partly idle:
    // every workgroup the same U consecutive K steps of the tiles' sequence, U below a tile's length; the pieces of a tile -- ceil(steps / U)
    // of them, one more where the sequence is cut inside its first U steps -- meet in the slabs of the split-K workspace.
    const bool sk_off = !knobs().gett_sk;
    bool sk = false;
    if (!sk_off && tm == 16 && (tn == 6 || tn == 7) && wide && p.nbatch == 1 && force_split == 0 && a.ksplit > 1 && ksteps >= 64 && ws.ptr) {
        const int wgs = 256;   // (one 8-wave workgroup per CU)
        const int64_t items = tiles * a.ksplit, rounds = (items + wgs - 1) / wgs;
        // Workgroups w, w + 8, ... share an XCD and its L2.  With U = j nk' / 8 (nk' = a tile's steps rounded up to a multiple of
        // eight, j = ceil(tiles / 32)) they all start at the same step of their tiles and walk the narrow operand's K range together:
        // eight K positions in flight instead of 256 (U = total / 256 exactly: FETCH_SIZE of a ladder product 7.2 -> 10 GB, and only 3 of
        // the 7 % that balance promises arrive).  The price: j / 8 >= tiles / 256, the last workgroups of the grid run short or not at all.
        const int nkp = (ksteps + 7) / 8 * 8, j8 = (int)((tiles + 31) / 32);
        const int U = j8 * (nkp / 8);
        const int64_t total = tiles * nkp;
        const int parts = (nkp + U - 1) / U + 1;
        if ((double)items / (double)(rounds * wgs) < 0.97 && (double)tiles / (32.0 * j8) > (double)items / (double)(rounds * wgs) + 0.02 && j8 < 8 &&
            U >= 32 && parts <= 8 && (size_t)parts * p.M * p.N * sizeof(double) <= ws.bytes && total < ((int64_t)1 << 31)) {
            sk = true;
            a.sk_units = U;
            a.sk_nk = nkp;
            a.kchunk = nkp * BK;
            a.ksplit = parts;
            grid = dim3((unsigned)((total + U - 1) / U), 1, 1);
            // the slabs that only the tiles cut early reach: zero (every tile writes the first ceil(steps / U) of them)
            const int always = (nkp + U - 1) / U;
            hipError_t me = hipMemsetAsync(ws.ptr + (size_t)always * p.M * p.N, 0, (size_t)(parts - always) * p.M * p.N * sizeof(double), stream);
            if (me != hipSuccess) return me;
        }
    }
    const bool gett_debug = knobs().gett_debug;   // every launch: extents, tile codes, K slices
    if (gett_debug)
        fprintf(stderr, "gett_launch M %d N %d K %d batch %d akc %d bkc %d wide %d -> tm %d tn %d tiles %d x %d split %d (steps per slice %d)\n", p.M, p.N, p.K,
                p.nbatch, (int)p.a_kcontig, (int)p.b_kcontig, (int)wide, tm, tn, a.mtiles, a.ntiles, a.ksplit, a.kchunk / BK);
    // tile code (tm,tn) -> wave grid x per-wave MFMA grid.  (4,4) is the 8-wave 128x128 tile: two waves per SIMD share
    // the matrix pipe, so one wave's gather/LDS phases are covered by the other's MFMAs.
#define AFESP_CFG(TM_, TN_, WM_, WN_, PM_, PN_) \
    if (tm == TM_ && tn == TN_) {                                                           \
        if (wide && TM_ >= 2 && TN_ >= 2) launch_cfg<WM_, WN_, PM_, PN_, (TM_ >= 2 && TN_ >= 2) ? 2 : 1>(a, grid, stream); \
        else launch_cfg<WM_, WN_, PM_, PN_, 1>(a, grid, stream);                            \
    }
    AFESP_CFG(1, 1, 2, 2, 1, 1) AFESP_CFG(1, 2, 2, 2, 1, 2) AFESP_CFG(1, 4, 2, 2, 1, 4)
    AFESP_CFG(2, 1, 2, 2, 2, 1) AFESP_CFG(2, 2, 2, 2, 2, 2) AFESP_CFG(2, 4, 2, 2, 2, 4)
    AFESP_CFG(4, 1, 2, 2, 4, 1) AFESP_CFG(4, 2, 2, 2, 4, 2) AFESP_CFG(4, 4, 2, 4, 4, 2)
    AFESP_CFG(8, 8, 2, 2, 4, 4) AFESP_CFG(16, 8, 4, 2, 4, 4) AFESP_CFG(8, 16, 2, 4, 4, 4)
#undef AFESP_CFG
    // 256 x 112 / 256 x 96: eight waves one above the other, 2 x 7 / 2 x 6 accumulators each, a 128-column image
    if (sk) {
        if (tn == 7) launch_cfg_sk<8, 1, 2, 7, 2, 8>(a, grid, stream);
        else launch_cfg_sk<8, 1, 2, 6, 2, 8>(a, grid, stream);
    } else {
        if (tm == 16 && tn == 7) { if (wide) launch_cfg<8, 1, 2, 7, 2, 8>(a, grid, stream); else launch_cfg<8, 1, 2, 7, 1, 8>(a, grid, stream); }
        if (tm == 16 && tn == 6) { if (wide) launch_cfg<8, 1, 2, 6, 2, 8>(a, grid, stream); else launch_cfg<8, 1, 2, 6, 1, 8>(a, grid, stream); }
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return err;
    if (a.ksplit > 1) {
        int64_t mn = (int64_t)p.M * p.N;
        unsigned gx = (unsigned)((mn + 255) / 256 > 2048 ? 2048 : (mn + 255) / 256);
        AFESP_KLAUNCH(gett_reduce_kernel, dim3(gx, (unsigned)p.nbatch), dim3(256), 0, stream, a);
        err = hipGetLastError();
    }
    return err;
}

void gett_grouped_tile(int M, bool wide, int* tm, int* tn, int* BM, int* BN)
{
    if (wide && M >= 2048) { *tm = 16; *tn = 8; *BM = 256; *BN = 128; }
    else { *tm = 4; *tn = 4; *BM = 128; *BN = 128; }
}

#endif   // !AFESP_GETT_GROUPED_TU

#ifdef AFESP_GETT_GROUPED_TU
hipError_t gett_launch_grouped(const GettProblem& p, const GettGroup* dev_groups, int ngroups, int total_tiles, int max_ntiles,
                               hipStream_t stream)
{
    if (p.M <= 0 || ngroups <= 0 || total_tiles <= 0 || p.K <= 0) return hipSuccess;
    if (p.nbatch != 1 || !p.a_kcontig || !p.b_kcontig || p.K % BK != 0) return hipErrorInvalidValue;
    GettKernelArgs a;
    a.p = p;
    const bool wide = p.wide && g_allow_wide && (p.K % 2 == 0);   // (both operands are K-contiguous here: M, N of any parity)
    int tm, tn, BM, BN;
    gett_grouped_tile(p.M, wide, &tm, &tn, &BM, &BN);
    a.mtiles = (p.M + BM - 1) / BM;
    a.ntiles = max_ntiles;
    a.ksplit = 1;
    a.kchunk = (p.K + BK - 1) / BK * BK;
    a.ws = nullptr;
    a.dbg = g_dbg;
    a.groups = dev_groups;
    a.total_tiles = total_tiles;
    a.sk_units = 0;
    a.sk_nk = 0;
    // a patch of gm m-tiles x all n-tiles of a group should be the ~32 tiles one XCD works on in a round (its L2 then
    // serves every operand panel of the patch once)
    a.gm = g_group_m > 0 ? g_group_m : std::max(1, (32 + max_ntiles / 2) / std::max(1, max_ntiles));
    if (a.gm > a.mtiles) a.gm = a.mtiles;
    dim3 grid((unsigned)total_tiles, 1, 1);
    if (tm == 16) {
        if (wide) {
            launch_one<4, 2, 4, 4, true, true, 2, true>(a, grid, stream);
        }
        else return hipErrorInvalidValue;
    } else {
        if (wide) launch_one<2, 4, 4, 2, true, true, 2, true>(a, grid, stream);
        else launch_one<2, 4, 4, 2, true, true, 1, true>(a, grid, stream);
    }
    return hipGetLastError();
}
#endif   // AFESP_GETT_GROUPED_TU

}  // namespace afesp
