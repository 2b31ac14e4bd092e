// triples_orbit.h -- device side of the (T) combine: W assembly, bars and the four sums in one pass over the X blocks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace afesp {

struct TripleMeta {
    int i, j, k, pad;   // pad: fused scheme, bit t set = block t of (Y^{i;jk}, Y^{j;ik}, Y^{k;ij}) holds X only (triples.hip)
    double mult;       // number of distinct ordered permutations of (i,j,k): 6, 3 or 1
    int64_t xoff[6];   // element offsets of X^{ijk}, X^{jik}, X^{kji}, X^{ikj}, X^{jki}, X^{kij} in the X pool
    int64_t woff;
};

struct TriplesIn {
    const double* e;
    const double* t1;
    const double* voovv_s;   // voovv_s(x,y,p,q) = v_oovv(p,q,x,y): contiguous v x v slice per occupied pair
    const double* t2_s;      // t2_s(x,y,p,q)    = t2(p,q,x,y)
    const double* t2;        // natural layout, for the D base term only
    int o, v;
};

// Diagnostic builds only (-DAFESP_ORBIT_STAMPS): cycles of wave 0 of a sample of workgroups (one slot per linear block index modulo
// ORB_SLOTS, plain stores -- atomics on one address serialise the whole launch), per phase -- [0] prologue up to the first barrier,
// [1+2s] barrier + park + barrier of term s (includes the wait for its HBM data), [2+2s] its permuted reads, [7] energy phase,
// [8] reduction and store, [9] = 1.  Read (summed over the slots) with afesp_debug_stamps(out, -10).
#ifdef AFESP_ORBIT_STAMPS
constexpr int ORB_SLOTS = 4096;
__device__ unsigned long long g_orbit_stamp[ORB_SLOTS * 16];
#define ORB_STAMP(k)                                                                                                          \
    {                                                                                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                         \
        if (threadIdx.x == 0) g_orbit_stamp[((blockIdx.y * gridDim.x + blockIdx.x) % ORB_SLOTS) * 16 + (k)] = now_ - orb_last_; \
        orb_last_ = now_;                                                                                                     \
    }
#else
#define ORB_STAMP(k) {}
#endif

// terms requested ahead of the one being permuted (A/B in one session, tools/build_orbit_variant.sh with
// EXTRA=-DAFESP_ORBIT_DEPTH=n, config 5: plain 52.5 / 51.8 / 55.0 ms for n = 1 / 2 / 3, completely renormalised 184 / 174 / 169 ms)
#ifndef AFESP_ORBIT_DEPTH
#define AFESP_ORBIT_DEPTH 0   // 0: two, three for the completely renormalised variant
#endif
constexpr int TT = 8;                 // cube edge
constexpr int CUBE = TT * TT * TT;    // 512 elements
constexpr int PATCH = TT * TT;

// The six simultaneous index permutations of ccsd.f90:2168-2173 in the order of the six term pairs:
// (abc) (bac) (cba) (acb) (bca) (cab); sig(s,d) = which of (a,b,c) sits in position d of term s.
__host__ __device__ constexpr int sig(int s, int d)
{
    return s == 0 ? d : s == 1 ? (d == 0 ? 1 : d == 1 ? 0 : 2) : s == 2 ? 2 - d : s == 3 ? (d == 0 ? 0 : d == 1 ? 2 : 1)
                      : s == 4 ? (d + 1) % 3 : (d + 2) % 3;
}
// 1/x for the energy denominators (sums of orbital-energy differences: finite, far from the ends of the exponent range):
// v_rcp_f64 and two Newton steps, 5 instructions against the ~25 of an IEEE division with its scaling and fix-up -- the orbit
// kernels were bound by instruction issue, not by HBM.  Relative error ~1e-16.
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}
// XOR-swizzled cube image: any one coordinate may run along the lanes without bank conflicts
__device__ __forceinline__ int cidx(int p0, int p1, int p2) { return ((p0 ^ p1 ^ p2) & 7) | (p1 << 3) | (p2 << 6); }

// Fused "orbit" kernel.  The virtual index space is cut into 8x8x8 cubes; a workgroup owns the orbit of one cube
// under the six permutations -- the smallest set closed under every index permutation the formulas use:
//   W(a,b,c) = sum_s X_s(sigma_s(a,b,c))                                              ccsd.f90:2168-2173
//   t_bar    = [4W(abc) + W(bca) + W(cab) - 2W(acb) - 2W(bac) - 2W(cba)] / 3D         symmetrised :2314-2318
// Every X element is read from HBM exactly once (whole 4 KiB cubes, 16 bytes per lane) and W never leaves the registers.
// CR = true additionally assembles the completely-renormalised moment M3 (ccsd.f90:2186-2194) from a second pool of
// blocks (same offsets) and accumulates sum t_bar.M3, sum z_bar.M3 (ccsd.f90:2222-2226); M3 is only needed at (a,b,c)
// itself, so it stays in registers.
// FUSED = true: the pool holds the three pair-summed blocks Y^{i;jk}, Y^{j;ik}, Y^{k;ij} (triples.hip, plan_fused) at
// xoff[0], xoff[1], xoff[5]; they enter W at the permutations (abc), (bac), (cab) -- terms 0, 1, 5 of the six.
__host__ __device__ constexpr int orbit_term(bool fused, int idx) { return fused ? (idx == 0 ? 0 : idx == 1 ? 1 : 5) : idx; }

// Index of the composed permutation: sig(compose(q,s), d) == sig(q, sig(s, d)), i.e. sigma_s applied to the q-th image of an
// element is its compose(q,s)-th image.
__host__ __device__ constexpr int compose(int q, int s)
{
    for (int r = 0; r < 6; ++r)
        if (sig(r, 0) == sig(q, sig(s, 0)) && sig(r, 1) == sig(q, sig(s, 1)) && sig(r, 2) == sig(q, sig(s, 2))) return r;
    return 0;
}

// A THREAD owns the orbit of an element: with e = (a,b,c) in the base cube at local coordinates (l0,l1,l2), its q-th image
// x^q = (e_sig(q,0), e_sig(q,1), e_sig(q,2)) lies in cube q at the permuted local coordinates, so the W assembly reads the
// staged cubes at permuted coordinates and leaves W(x^0..x^5) in six registers (two orbits per thread).  Everything the
// energy formulas combine -- W, Z and y at the six images, one common denominator -- is then in the thread's registers:
// W is never written back to LDS, D and its reciprocal are evaluated once per orbit instead of once per element, and the
// t1 rows / V patches are read 27 times per orbit instead of 6 x 6.  (Round 1 gave a thread the SAME local coordinates in all
// six cubes: 72 LDS reads of W per thread, and the kernel was bound by instruction issue at 4.15 TB/s.)
// A cube orbit whose tiles coincide has 6/|H| distinct cubes (H = the stabiliser of the tile triple): every element of the
// distinct cubes is then visited by exactly |H| (thread, image) pairs, so the workgroup's sums are divided by |H|.
// WANT_D = false (plain CCSD(T)/[T], which the reference also evaluates without y and the D sums, ccsd.f90:2181-2185 and
// :2228-2247): only E[T] and the z term.  The symmetriser P in z_bar = P Z / D is self-adjoint and D is symmetric, and a
// thread sums over a set of elements closed under every permutation, so  sum z_bar W = sum Z (P W)/D = sum Z t_bar.
template <bool CR, bool FUSED, bool WANT_D = true>
__global__ __launch_bounds__(256, CR ? 2 : WANT_D ? 3 : 4) void triples_orbit_kernel(double* __restrict__ partial, const double* __restrict__ Xpool,
                                                            const double* __restrict__ Mpool,
                                                            const TripleMeta* __restrict__ meta,
                                                            const int* __restrict__ orbits, TriplesIn in, int nblk_total)
{
    __shared__ __attribute__((aligned(16))) double stage[6 * CUBE];   // the six cubes of one term
    // patches: vp[pair][sx][sy][lx + 8 ly] = V_pair(x in tile[sx], y in tile[sy]); pairs (j,k), (i,k), (i,j); tp the same of t2
    __shared__ double vp[27 * PATCH];
    __shared__ double tp[WANT_D ? 27 * PATCH : 1];
    __shared__ double t1r[96];                 // t1r[occ][slot][l] = t1(occ, tile[slot]*8 + l); then evl[slot][l] = e(o + tile[slot]*8 + l)
    __shared__ int srcq[6][6];                 // srcq[s][q]: which cube of the orbit is sigma_s applied to cube q
    __shared__ double red[24];
#ifdef AFESP_ORBIT_STAMPS
    unsigned long long orb_last_ = __builtin_amdgcn_s_memtime();
#endif
    const TripleMeta m = meta[blockIdx.y];
    const int o = in.o, v = in.v, t = threadIdx.x;
    const int packed = orbits[blockIdx.x];
    const int tile[3] = {packed & 1023, (packed >> 10) & 1023, (packed >> 20) & 1023};
    if (t < 36) {
        const int s = t / 6, q = t % 6;
        int want[3];
        for (int d = 0; d < 3; ++d) want[d] = tile[sig(q, sig(s, d))];   // position d of the source cube
        int found = 0;
        for (int r = 5; r >= 0; --r)
            if (tile[sig(r, 0)] == want[0] && tile[sig(r, 1)] == want[1] && tile[sig(r, 2)] == want[2]) found = r;
        srcq[s][q] = found;
    }
    const int64_t vv = (int64_t)v * v;
    // local coordinates of this thread's two base elements (half = 0, 1)
    const int l0 = t & 7, l1 = (t >> 3) & 7, l2h[2] = {t >> 6, (t >> 6) + 4};
    // The X blocks are stored cube by cube (triples.hip: element (a,b,c) of a block sits at
    // 512*(a/8 + nt8*(b/8) + nt8^2*(c/8)) + a%8 + 8*(b%8) + 64*(c%8)), so one cube is 4 KiB of contiguous HBM: thread t
    // fetches elements 2t, 2t+1 of each of the six cubes with one 16-byte load and parks them in the swizzled image.
    const int nt8 = (v + TT - 1) / TT;
    const int p0 = (2 * t) & 7, p1 = (t >> 2) & 7, p2 = t >> 5;
    const int sbase = (((p0 ^ p1 ^ p2) & 7) & ~1) | (p1 << 3) | (p2 << 6);
    const bool flip = ((p1 ^ p2) & 1) != 0;   // the XOR swizzle swaps the two elements of the pair
    typedef double v2d_t __attribute__((ext_vector_type(2)));
    auto load_term = [&](const double* X, v2d_t (&xin)[6]) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int T0 = tile[sig(q, 0)], T1 = tile[sig(q, 1)], T2 = tile[sig(q, 2)];
            const v2d_t x = *reinterpret_cast<const v2d_t*>(X + (int64_t)CUBE * (T0 + (int64_t)nt8 * (T1 + (int64_t)nt8 * T2)) + 2 * t);
            const int g0 = T0 * TT + p0;
            const bool rows = (T1 * TT + p1 < v) && (T2 * TT + p2 < v);   // the padding of a block is never written
            xin[q][0] = (rows && g0 < v) ? x[0] : 0.0;
            xin[q][1] = (rows && g0 + 1 < v) ? x[1] : 0.0;
        }
    };
    auto park_term = [&](const v2d_t (&xin)[6]) {
#pragma unroll
        for (int q = 0; q < 6; ++q)
            *reinterpret_cast<v2d_t*>(&stage[q * CUBE + sbase]) = flip ? (v2d_t){xin[q][1], xin[q][0]} : xin[q];
    };
    // DEPTH terms are requested ahead of the one being permuted out of LDS (6 loads of 16 bytes per thread and term); the next
    // one follows as soon as a term's registers are parked
    constexpr int NT1 = FUSED ? 3 : 6;
    constexpr int NTERM = CR ? 2 * NT1 : NT1;
    constexpr int DEPTH = AFESP_ORBIT_DEPTH > 0 ? (AFESP_ORBIT_DEPTH < NT1 ? AFESP_ORBIT_DEPTH : NT1) : CR ? 3 : 2;
    v2d_t xin[DEPTH][6];
#pragma unroll
    for (int s2 = 0; s2 < DEPTH; ++s2) load_term(Xpool + m.xoff[orbit_term(FUSED, s2)], xin[s2]);
    // the patches, t1 rows and virtual orbital energies of the three tiles travel under the first term.  Wave g stages patches
    // g, g+4, ... of the 18 (three occupied pairs x six ordered pairs of tile slots) the energy phase reads: the patch index is
    // wave-uniform, a lane owns one element
    const int occ[3] = {m.i, m.j, m.k};
    const int pairp[3] = {m.j, m.i, m.i}, pairq[3] = {m.k, m.k, m.j};
    {
        const int g = __builtin_amdgcn_readfirstlane(t >> 6), loc = t & 63;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int P = g + 4 * i;
            if (P < 18) {
                const int pr = P / 6, pi = P % 6;
                const int sx = pi < 2 ? 0 : pi < 4 ? 1 : 2, sy = pi == 0 ? 1 : pi == 1 ? 2 : pi == 2 ? 0 : pi == 3 ? 2 : pi == 4 ? 0 : 1;
                const int gx = tile[sx] * TT + (loc & 7), gy = tile[sy] * TT + (loc >> 3);
                const bool ok = gx < v && gy < v;
                const int64_t off = (ok ? gx + (int64_t)v * gy : 0) + vv * (pairp[pr] + (int64_t)o * pairq[pr]);
                const int dst = (pr * 9 + sx * 3 + sy) * PATCH + loc;
                const double a = in.voovv_s[off];
                vp[dst] = ok ? a : 0.0;
                if (WANT_D) {
                    const double b = in.t2_s[off];
                    tp[dst] = ok ? b : 0.0;
                }
            }
        }
    }
    if (t < 72) {
        const int oc = t / 24, sl = (t / 8) % 3, l = t & 7, g = tile[sl] * TT + l;
        t1r[t] = g < v ? in.t1[occ[oc] + o * g] : 0.0;
    } else if (t < 96) {
        const int sl = (t - 72) / 8, l = t & 7, g = tile[sl] * TT + l;
        t1r[t] = in.e[(g < v ? g : 0) + o];
    }
    const double* evl = t1r + 72;
    double wreg[12];                 // wreg[2q + h] = W at the q-th image of base element h
    double mreg[CR ? 12 : 1];
#pragma unroll
    for (int r = 0; r < 12; ++r) wreg[r] = 0.0;
#pragma unroll
    for (int r = 0; r < (CR ? 12 : 1); ++r) mreg[r] = 0.0;
#pragma unroll
    for (int s2 = 0; s2 < NTERM; ++s2) {
        const int s = orbit_term(FUSED, s2 % NT1);
        // a block whose occupied pair coincides (Y^{p;qq}) holds X(x;y,z) only: Y = X + X with (y,z) exchanged, i.e. the same
        // staged cubes read through the permutation s followed by that exchange: (abc)->(acb), (bac)->(bca), (cab)->(cba)
        const int sT = s == 0 ? 3 : s == 1 ? 4 : 2;
        const bool sym = FUSED && ((m.pad >> (s2 % NT1)) & 1);
        if (s2 == 0) ORB_STAMP(0)
        __syncthreads();   // the previous term's readers are done with `stage` (also publishes srcq and the patches on the first pass)
        park_term(xin[s2 % DEPTH]);
        if (s2 + DEPTH < NTERM)
            load_term((s2 + DEPTH < NT1 ? Xpool : Mpool) + m.xoff[orbit_term(FUSED, (s2 + DEPTH) % NT1)], xin[s2 % DEPTH]);
        __syncthreads();
        if (s2 < 3) ORB_STAMP(1 + 2 * s2)
        // W(x^q) = sum_s X_s(sigma_s x^q); sigma_s x^q is the compose(q,s)-th image of the base element: cube srcq[s][q],
        // local coordinates l[sig(q, sig(s, .))].  (`sym` is uniform over the workgroup: one branch around the twelve reads, not
        // one per read -- a branch per read made every read wait for the one before.)
#define ORB_READ(sg) stage[srcq[sg][q] * CUBE + cidx(l[sig(q, sig(sg, 0))], l[sig(q, sig(sg, 1))], l[sig(q, sig(sg, 2))])]
        double xr[12];
        if (FUSED && sym) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const int q = r >> 1;
                const int l[3] = {l0, l1, l2h[r & 1]};
                xr[r] = ORB_READ(s) + ORB_READ(sT);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
                const int q = r >> 1;
                const int l[3] = {l0, l1, l2h[r & 1]};
                xr[r] = ORB_READ(s);
            }
        }
#undef ORB_READ
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            if (s2 < NT1) wreg[r] += xr[r];
            else mreg[CR ? r : 0] += xr[r];
        }
        if (s2 < 3) ORB_STAMP(2 + 2 * s2)
    }
    const double eo = in.e[m.i] + in.e[m.j] + in.e[m.k];
    constexpr int NQ = CR ? 6 : WANT_D ? 4 : 2;
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // the second orbit's LDS addresses are made to "depend" on the first orbit's sums (an empty asm): its ~30 reads then stay
        // behind the first orbit's arithmetic instead of being hoisted in front of it and spilled
        int l2v = l2h[h];
        if (h == 1) asm volatile("" : "+v"(l2v) : "v"(acc[0]), "v"(acc[NQ - 1]));
        const int l[3] = {l0, l1, l2v};
        const bool live = tile[0] * TT + l[0] < v && tile[1] * TT + l[1] < v && tile[2] * TT + l[2] < v;
        const double D = eo - evl[l[0]] - evl[TT + l[1]] - evl[2 * TT + l[2]];   // the same for the six images
        const double rD = rcp_nr(3.0 * D);       // (padding reads valid orbital energies: D is finite and non-zero there too)
        const double r3D = live ? rD : 0.0;
        double t1v[3][3], Z[6], Y[WANT_D ? 6 : 1];
#pragma unroll
        for (int oc = 0; oc < 3; ++oc)
#pragma unroll
            for (int c = 0; c < 3; ++c) t1v[oc][c] = t1r[oc * 24 + c * TT + l[c]];
        {
            double V[3][3][3];
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int cx = 0; cx < 3; ++cx)
#pragma unroll
                    for (int cy = 0; cy < 3; ++cy)
                        if (cx != cy) V[pr][cx][cy] = vp[(pr * 9 + cx * 3 + cy) * PATCH + l[cx] + TT * l[cy]];
            // Z(x,y,z) = t1(i,x) V_jk(y,z) + t1(j,y) V_ik(x,z) + t1(k,z) V_ij(x,y)      ccsd.f90:2178-2179 (numerator)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int s0 = sig(q, 0), s1 = sig(q, 1), s2 = sig(q, 2);
                Z[q] = t1v[0][s0] * V[0][s1][s2] + t1v[1][s1] * V[1][s0][s2] + t1v[2][s2] * V[2][s0][s1];
            }
        }
        if (WANT_D) {
            // (the t2 patches are read once the V patches are dead: same device)
            int lt[3] = {l[0], l[1], l[2]};
            asm volatile("" : "+v"(lt[0]), "+v"(lt[1]), "+v"(lt[2]) : "v"(Z[0]), "v"(Z[1]), "v"(Z[2]), "v"(Z[3]), "v"(Z[4]), "v"(Z[5]));
            double T2[3][3][3];
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int cx = 0; cx < 3; ++cx)
#pragma unroll
                    for (int cy = 0; cy < 3; ++cy)
                        if (cx != cy) T2[pr][cx][cy] = tp[(pr * 9 + cx * 3 + cy) * PATCH + lt[cx] + TT * lt[cy]];
            // y (ccsd.f90:2183-2184)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int s0 = sig(q, 0), s1 = sig(q, 1), s2 = sig(q, 2);
                Y[q] = t1v[0][s0] * t1v[1][s1] * t1v[2][s2] + t1v[0][s0] * T2[0][s1][s2] + t1v[1][s1] * T2[1][s0][s2] +
                       t1v[2][s2] * T2[2][s0][s1];
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const double w = wreg[2 * q + h];
            // 3 x the bar of ccsd.f90:2314-2318 at the q-th image, over 3D
#define BAR3(A) (4.0 * A(q) + A(compose(q, 4)) + A(compose(q, 5)) - 2.0 * (A(compose(q, 3)) + A(compose(q, 1)) + A(compose(q, 2))))
#define W_AT(r) wreg[2 * (r) + h]
#define Z_AT(r) Z[r]
            const double tbar = BAR3(W_AT) * r3D;
            acc[0] += tbar * w;
            if (!WANT_D) {
                acc[1] += tbar * Z[q];
                continue;
            }
            const double zbar = BAR3(Z_AT) * r3D;
#undef BAR3
#undef W_AT
#undef Z_AT
            const double y = Y[WANT_D ? q : 0];
            acc[1] += zbar * w;
            acc[WANT_D ? 2 : 0] += tbar * y;
            acc[WANT_D ? 3 : 0] += zbar * y;
            if (CR) {
                const double mm = mreg[CR ? 2 * q + h : 0];
                acc[NQ - 2] += tbar * mm;
                acc[NQ - 1] += zbar * mm;
            }
        }
    }
    ORB_STAMP(7)
    // |H|: how many of the six permutations leave the tile triple where it is
    const int nH = (tile[0] == tile[1] && tile[1] == tile[2]) ? 6 : (tile[0] == tile[1] || tile[1] == tile[2] || tile[0] == tile[2]) ? 2 : 1;
    const int lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double sdl = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sdl += __shfl_down(sdl, off, 64);
        if (lane == 0) red[q * 4 + wv] = sdl;
    }
    __syncthreads();
    if (t < NQ) {
        const int blk = blockIdx.y * gridDim.x + blockIdx.x;
        partial[(int64_t)t * nblk_total + blk] = m.mult * (red[t * 4] + red[t * 4 + 1] + red[t * 4 + 2] + red[t * 4 + 3]) / (double)nH;
    }
    ORB_STAMP(8)
#ifdef AFESP_ORBIT_STAMPS
    if (t == 0) g_orbit_stamp[((blockIdx.y * gridDim.x + blockIdx.x) % ORB_SLOTS) * 16 + 9] = 1ull;
#endif
}

// Spin-orbital (T), ccsd.f90:1812-1922.  For i<j<k the three GEMM blocks Y^{i;jk}, Y^{j;ik}, Y^{k;ij} (triples.hip,
// so_triples) give the connected numerator before P(a/bc):  R = Y^{i;jk} - Y^{j;ik} + Y^{k;ij}  (all three in the same
// (a,b,c) orientation, so R is assembled in registers straight from the 16-byte cube loads); then
//   t3c = R(abc) - R(bac) - R(cba)                                         :1894-1896 (reshape orders (2,1,3), (3,2,1))
//   t3d = P(a/bc)[t1(i,a)<jk||bc> - t1(j,a)<ik||bc> - t1(k,a)<ji||bc>]     :1873-1874, :1890-1892
//   E_T += t3c (t3c + t3d) / D / 6      (the reference sums all ordered (i,j,k) with 1/36; the summand is antisymmetric)
// One workgroup = the orbit of one 8x8x8 cube under index permutation, as in the spin-free kernel.
__global__ __launch_bounds__(256, 3) void triples_so_orbit_kernel(double* __restrict__ partial, const double* __restrict__ Xpool,
                                                                 const TripleMeta* __restrict__ meta,
                                                                 const int* __restrict__ orbits, TriplesIn in, int nblk_total)
{
    __shared__ __attribute__((aligned(16))) double wl[6 * CUBE];   // R on the six cubes of the orbit
    __shared__ double vp[27 * PATCH];                              // <pq||xy> patches for pairs (j,k), (i,k), (i,j)
    __shared__ double t1r[96];                                     // t1r[occ][slot][l] = t1(occ, tile[slot]*8 + l); then evl
    __shared__ int srcq[6][6];
    __shared__ double red[4];
    const TripleMeta m = meta[blockIdx.y];
    const int o = in.o, v = in.v, t = threadIdx.x;
    const int packed = orbits[blockIdx.x];
    const int tile[3] = {packed & 1023, (packed >> 10) & 1023, (packed >> 20) & 1023};
    if (t < 36) {
        const int s = t / 6, q = t % 6;
        int want[3];
        for (int d = 0; d < 3; ++d) want[d] = tile[sig(q, sig(s, d))];
        int found = 0;
        for (int r = 5; r >= 0; --r)
            if (tile[sig(r, 0)] == want[0] && tile[sig(r, 1)] == want[1] && tile[sig(r, 2)] == want[2]) found = r;
        srcq[s][q] = found;
    }
    const int64_t vv = (int64_t)v * v;
    const int nt8 = (v + TT - 1) / TT;
    const int p0 = (2 * t) & 7, p1 = (t >> 2) & 7, p2 = t >> 5;
    const int sbase = (((p0 ^ p1 ^ p2) & 7) & ~1) | (p1 << 3) | (p2 << 6);
    const bool flip = ((p1 ^ p2) & 1) != 0;
    typedef double v2d_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const int T0 = tile[sig(q, 0)], T1 = tile[sig(q, 1)], T2 = tile[sig(q, 2)];
        const int64_t off = (int64_t)CUBE * (T0 + (int64_t)nt8 * (T1 + (int64_t)nt8 * T2)) + 2 * t;
        const v2d_t x0 = *reinterpret_cast<const v2d_t*>(Xpool + m.xoff[0] + off);
        const v2d_t x1 = *reinterpret_cast<const v2d_t*>(Xpool + m.xoff[1] + off);
        const v2d_t x2 = *reinterpret_cast<const v2d_t*>(Xpool + m.xoff[2] + off);
        const int g0 = T0 * TT + p0;
        const bool rows = (T1 * TT + p1 < v) && (T2 * TT + p2 < v);
        const double r0 = (rows && g0 < v) ? x0[0] - x1[0] + x2[0] : 0.0;
        const double r1 = (rows && g0 + 1 < v) ? x0[1] - x1[1] + x2[1] : 0.0;
        *reinterpret_cast<v2d_t*>(&wl[q * CUBE + sbase]) = flip ? (v2d_t){r1, r0} : (v2d_t){r0, r1};
    }
    const int occ[3] = {m.i, m.j, m.k};
    const int pairp[3] = {m.j, m.i, m.i}, pairq[3] = {m.k, m.k, m.j};
    for (int el = t; el < 27 * PATCH; el += 256) {
        const int pr = el / (9 * PATCH), rest = el % (9 * PATCH), sx = rest / (3 * PATCH), sy = (rest / PATCH) % 3, loc = rest % PATCH;
        const int gx = tile[sx] * TT + (loc & 7), gy = tile[sy] * TT + (loc >> 3);
        const bool ok = gx < v && gy < v;
        const int64_t off = ok ? gx + (int64_t)v * gy + vv * (pairp[pr] + (int64_t)o * pairq[pr]) : 0;
        const double a = in.voovv_s[off];
        vp[el] = ok ? a : 0.0;
    }
    if (t < 72) {
        const int oc = t / 24, sl = (t / 8) % 3, l = t & 7, g = tile[sl] * TT + l;
        t1r[t] = g < v ? in.t1[occ[oc] + o * g] : 0.0;
    } else if (t < 96) {
        const int sl = (t - 72) / 8, l = t & 7, g = tile[sl] * TT + l;
        t1r[t] = in.e[(g < v ? g : 0) + o];
    }
    const double* evl = t1r + 72;   // evl[slot][l] = e(o + tile[slot]*8 + l)
    __syncthreads();
    const int l0 = t & 7, l1 = (t >> 3) & 7, l2h[2] = {t >> 6, (t >> 6) + 4};
    const double eo = in.e[m.i] + in.e[m.j] + in.e[m.k];
    double acc = 0.0;
    // a thread owns the orbit of two base elements (see triples_orbit_kernel): the q-th image lies in cube srcq[0][q] at the
    // permuted local coordinates; D, the t1 rows and the V patches are read once per orbit
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int l2v = l2h[h];
        if (h == 1) asm volatile("" : "+v"(l2v) : "v"(acc));   // the second orbit's reads stay behind the first orbit's arithmetic
        const int l[3] = {l0, l1, l2v};
        const bool live = tile[0] * TT + l[0] < v && tile[1] * TT + l[1] < v && tile[2] * TT + l[2] < v;
        const double D = eo - evl[l[0]] - evl[TT + l[1]] - evl[2 * TT + l[2]];
        const double rD = rcp_nr(D);
        double R[6], t1v[3][3], V[3][3][3], RAW[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) R[q] = wl[srcq[0][q] * CUBE + cidx(l[sig(q, 0)], l[sig(q, 1)], l[sig(q, 2)])];
#pragma unroll
        for (int oc = 0; oc < 3; ++oc)
#pragma unroll
            for (int c = 0; c < 3; ++c) t1v[oc][c] = t1r[oc * 24 + c * TT + l[c]];
#pragma unroll
        for (int pr = 0; pr < 3; ++pr)
#pragma unroll
            for (int cx = 0; cx < 3; ++cx)
#pragma unroll
                for (int cy = 0; cy < 3; ++cy)
                    if (cx != cy) V[pr][cx][cy] = vp[(pr * 9 + cx * 3 + cy) * PATCH + l[cx] + TT * l[cy]];
        // RAW(x,y,z) = t1(i,x)<jk||yz> - t1(j,x)<ik||yz> + t1(k,x)<ij||yz> at the q-th image          :1873-1874
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int s0 = sig(q, 0), s1 = sig(q, 1), s2 = sig(q, 2);
            RAW[q] = t1v[0][s0] * V[0][s1][s2] - t1v[1][s0] * V[1][s1][s2] + t1v[2][s0] * V[2][s1][s2];
        }
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const double t3c = R[q] - R[compose(q, 1)] - R[compose(q, 2)];          // :1894-1896
            const double t3d = RAW[q] - RAW[compose(q, 1)] - RAW[compose(q, 2)];    // :1890-1892
            sum += t3c * (t3c + t3d);
        }
        acc += live ? sum * rD : 0.0;
    }
    // |H|: how many of the six permutations leave the tile triple where it is (each element is visited |H| times)
    const int nH = (tile[0] == tile[1] && tile[1] == tile[2]) ? 6 : (tile[0] == tile[1] || tile[1] == tile[2] || tile[0] == tile[2]) ? 2 : 1;
    const int lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (t == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) / (6.0 * nH);
}

}  // namespace afesp
