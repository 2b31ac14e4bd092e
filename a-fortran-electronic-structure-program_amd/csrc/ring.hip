// ring.hip -- the six o^3 v^3 "ring" products of a large system's CCSD iteration on the LDS-DMA GEMM (tgemm.h).
//
// update_restricted_intermediates forms I_ovov with one such product and I_voov with two (src/ccsd.f90:1170-1182, :1207-1227),
// update_amplitudes_restricted contracts them with the amplitudes in three more (:1680-1695): at o = 20, v = 200 half of an
// iteration.  On the gather kernel (gett.h) each is a launch of its own -- 512 tiles of 256 x 128 = exactly two rounds at 0.95
// tile fill, a ramp and a straggler round per launch, 0.76 of the fp64 MFMA peak.  Here they are TWO launches of tgemm_kernel:
//
//   L1   rows (j,a), columns (i,b), K = (m,e):
//        group 1  [ t2'(ja;me) | c'(ja;me) ] x [ 1/2 w(imbe) | -1/2 v(imbe) ]   ->  I_voov'        (two runs of K: :1207-1227)
//        group 2    c'(ja;me)                x   1/2 v(mibe)                    -> -I_ovov'        (:1170-1182)
//   L2   rows (j,b), columns (i,a):
//        group 1  [ t2'(jb;me) | I_voov'(jb;me) ] x [ -I_ovov'(ia;me) | asym'(ia;me) ]  ->  R(i,j,a,b)   (:1688-1695, two terms)
//        group 2    -I_ovov'(ib;me)               x   t2x(ja;me)                        ->  Y(j,i,a,b)   (:1680-1687)
//
// Every operand is a dense matrix [K fastest | free pair] with K = (m,e) padded to whole K steps -- what the kernel's LDS-DMA
// staging asks for.  The integrals' copies (with the factors folded in) are made once per state, the amplitudes' copies once
// per iteration (o^2 v^2 elements each).  L1 writes the two intermediates directly in the layout L2 reads them in; their small
// terms (<ia|jb>, the t1 products) are formed in the reference's layout as before and added by one transposing pass each.
// The third ring term has its row / column pairs the other way round: it lands in a buffer of its own with i and j exchanged,
// and the amplitude update -- which only ever needs r2(ijab) + r2(jiba) -- reads it that way.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "ccsd.h"

namespace afesp {

struct RingTg {
    int64_t ov = 0, Kc = 0, each = 0;   // rows = columns = summation length o v, padded K, doubles per buffer
    double* slab = nullptr;             // ten buffers of `each` doubles
    double *cp, *t2p, *t2x, *asp, *nIo, *Ivo, *F1, *F2, *F3, *Y;
    uint32_t* rc32 = nullptr;           // byte offset of row / column x of an operand: 8 Kc x   (ov + 256 entries)
    int64_t *cm1 = nullptr, *cn1 = nullptr, *cm2 = nullptr, *cn2 = nullptr;   // C offsets of L1 / L2 (ov + 128 entries each)
    int64_t* tab_block = nullptr;
    TgGroup* groups = nullptr;          // device: L1's three descriptors, then L2's
    int tiles = 0, mx = 0;
    bool pairs2 = false;
    bool frozen_built = false;
    bool packed = false;                // the amplitudes' four copies are those of the current t1 / t2 (ring_tg_pack)
    bool live = false;                  // I_ovov' / I_voov' hold the current intermediates (the reference-layout tensors only their small terms)
    bool res_live = false;              // R / Y hold ring terms of the current residual
};

__global__ __launch_bounds__(256) void ring_tables_kernel(uint32_t* rc32, int64_t* cm1, int64_t* cn1, int64_t* cm2, int64_t* cn2, int o, int v,
                                                          int64_t Kc)
{
    const int64_t ov = (int64_t)o * v;
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < ov + 256; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = x < ov ? x : ov - 1;   // (the kernel fetches its tables in whole pieces: the padding repeats the last entry)
        rc32[x] = (uint32_t)(8 * Kc * y);
        if (x < ov + 128) {
            const int64_t p = y % o, q = y / o;
            cm1[x] = Kc * y;
            cn1[x] = y;
            cm2[x] = (int64_t)o * p + (int64_t)o * o * v * q;
            cn2[x] = p + (int64_t)o * o * q;
        }
    }
}

// AFESP_RING_TG=0: the gather kernel's six launches at every size.  From o v = 3584 on (28 row tiles; AFESP_RING_TG_MIN; read per
// call: tests send small systems down this path): a launch needs about four 128 x 128 tiles per workgroup slot to beat the gather
// kernel, whose tile shapes adapt to the extents (iteration, gather kernel / this path: o = 16, v = 160 8.23 / 8.16 ms, o = 18, v = 180
// 14.6 / 15.0, o = 20, v = 200 23.9 / 22.6; profiles/r05_mid_sweep.txt).  Row offsets inside an operand are 32-bit byte offsets:
// 8 Kc o v < 4 GiB.
bool ring_tg_applies(const CCState& s)
{
    if (!knobs().ring_tg) return false;
    const int64_t min_ov = knobs().ring_tg_min;
    const int64_t ov = (int64_t)s.o * s.v, Kc = (ov + TG_BK - 1) / TG_BK * TG_BK;
    return !s.sharded && ov >= min_ov && ov >= 2 * TG_BK && 8 * Kc * ov < ((int64_t)1 << 32) - 4096;
}

// asym_t2 = 2 t2 - t2(jiab), c = t2 + t1 t1 (ccsd.f90:1063-1079) AND the four [K | row] copies the two launches read, from ONE pass
// over t2: a workgroup stages the o^2 x P x Q block t2(:,:,p0..,q0..) in LDS (reads: runs of o^2 doubles) and writes it out in the
// three orders -- the reference's (asym, c), [(m,p) | (j,q)] (t2', asym': runs of o P doubles) and [(m,q) | (j,p)] (t2x, c': o Q).
// 8 (1 + 6) o^2 v^2 bytes instead of 8 (3 + 8) for k_asym_c and four permuting copies.
struct RingPackArgs {
    const double *t1, *t2;
    double *asym, *c, *cp, *t2p, *t2x, *asp;
    int o, v, P, Q;
    int64_t Kc;
    unsigned inv_o, inv_oo;   // tgemm_inverse(o), tgemm_inverse(o * o)
};
__device__ __forceinline__ unsigned ring_div(unsigned x, unsigned inv) { return inv ? __umulhi(x, inv) : x; }

__global__ __launch_bounds__(256) void ring_pack_kernel(RingPackArgs a)
{
    extern __shared__ double tile[];   // t2(m, j, p0 + pp, q0 + qq) at (m + o j) + o^2 (pp + P qq)
    const int o = a.o, v = a.v, oo = o * o, P = a.P, Q = a.Q;
    const int tiles_p = (v + P - 1) / P;
    const int p0 = ((int)blockIdx.x % tiles_p) * P, q0 = ((int)blockIdx.x / tiles_p) * Q;
    const int np = min(P, v - p0), nq = min(Q, v - q0), n = oo * np * nq;
    const unsigned inv_np = np == 1 ? 0u : (unsigned)(((uint64_t)1 << 32) / (unsigned)np + 1);
    const int64_t O = o, V = v;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int r = (int)ring_div((unsigned)idx, a.inv_oo), mj = idx - r * oo;
        const int qq = (int)ring_div((unsigned)r, inv_np), pp = r - qq * np;
        tile[mj + oo * (pp + P * qq)] = a.t2[mj + O * O * ((p0 + pp) + V * (q0 + qq))];
    }
    __syncthreads();
    // the reference's layout: asym(m,j,p,q), c(m,j,p,q)
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int r = (int)ring_div((unsigned)idx, a.inv_oo), mj = idx - r * oo;
        const int qq = (int)ring_div((unsigned)r, inv_np), pp = r - qq * np;
        const int j = (int)ring_div((unsigned)mj, a.inv_o), m = mj - j * o;
        const int blk = oo * (pp + P * qq);
        const double t = tile[mj + blk], tt = tile[j + o * m + blk];
        const int64_t x = mj + O * O * ((p0 + pp) + V * (q0 + qq));
        a.asym[x] = 2.0 * t - tt;
        a.c[x] = t + a.t1[m + o * (p0 + pp)] * a.t1[j + o * (q0 + qq)];
    }
    // [(m,p) | (j,q)]: t2'(jq; mp) = t2(m,j,p,q), asym'(jq; mp) = asym(m,j,p,q)
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        int r = (int)ring_div((unsigned)idx, a.inv_o);
        const int m = idx - r * o;
        int r2 = (int)ring_div((unsigned)r, inv_np);
        const int pp = r - r2 * np;
        const int qq = (int)ring_div((unsigned)r2, a.inv_o), j = r2 - qq * o;
        const int blk = oo * (pp + P * qq);
        const double t = tile[m + o * j + blk], tt = tile[j + o * m + blk];
        const int64_t out = m + O * (p0 + pp) + a.Kc * (j + O * (q0 + qq));
        a.t2p[out] = t;
        a.asp[out] = 2.0 * t - tt;
    }
    // [(m,q) | (j,p)]: t2x(jp; mq) = t2(m,j,p,q), c'(jp; mq) = c(m,j,p,q)
    const unsigned inv_nq = nq == 1 ? 0u : (unsigned)(((uint64_t)1 << 32) / (unsigned)nq + 1);
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        int r = (int)ring_div((unsigned)idx, a.inv_o);
        const int m = idx - r * o;
        int r2 = (int)ring_div((unsigned)r, inv_nq);
        const int qq = r - r2 * nq;
        const int pp = (int)ring_div((unsigned)r2, a.inv_o), j = r2 - pp * o;
        const double t = tile[m + o * j + oo * (pp + P * qq)];
        const int64_t out = m + O * (q0 + qq) + a.Kc * (j + O * (p0 + pp));
        a.t2x[out] = t;
        a.cp[out] = t + a.t1[m + o * (p0 + pp)] * a.t1[j + o * (q0 + qq)];
    }
}

static RingTg* ring_get(Context& cx, CCState& s)
{
    if (s.ring) return (RingTg*)s.ring;
    RingTg* r = new RingTg();
    s.ring = r;
    const int64_t O = s.o, V = s.v;
    r->ov = O * V;
    r->Kc = (r->ov + TG_BK - 1) / TG_BK * TG_BK;
    r->each = r->Kc * r->ov + 2 * TG_BK;   // (+ slack: a K step of the last row may be fetched whole)
    r->each = (r->each + 15) / 16 * 16;
    r->slab = cx.alloc(10 * r->each);      // zeroed: the K padding of every operand stays zero for ever
    double** bufs[10] = {&r->cp, &r->t2p, &r->t2x, &r->asp, &r->nIo, &r->Ivo, &r->F1, &r->F2, &r->F3, &r->Y};
    for (int q = 0; q < 10; ++q) *bufs[q] = r->slab + q * r->each;
    const int64_t n32 = r->ov + 256, n64 = r->ov + 128;
    int64_t* tab = cx.alloc_i64(4 * n64 + (n32 + 1) / 2 + 8);
    r->cm1 = tab; r->cn1 = tab + n64; r->cm2 = tab + 2 * n64; r->cn2 = tab + 3 * n64;
    r->rc32 = (uint32_t*)(tab + 4 * n64);
    r->tab_block = tab;
    AFESP_KLAUNCH(ring_tables_kernel, dim3((unsigned)std::min<int64_t>((n32 + 255) / 256, 4096)), dim3(256), 0, cx.stream, r->rc32, r->cm1,
                       r->cn1, r->cm2, r->cn2, s.o, s.v, r->Kc);
    AFESP_HIP(hipGetLastError());
    // descriptors: offsets are relative to the slab, so they are made once
    const int M = (int)r->ov, mt = (M + TG_BM - 1) / TG_BM, nt = (M + TG_BN - 1) / TG_BN;
    const int nk1 = (int)(r->Kc / TG_BK), gm = tgemm_group_m(M, nt);
    r->mx = nt;
    r->tiles = 2 * mt * nt;
    auto off = [&](const double* p) { return (int64_t)(p - r->slab); };
    TgGroup g[6] = {};
    auto fill = [&](TgGroup& d, const double* a1, const double* a2, const double* b1, const double* b2, const double* c, const int64_t* cn,
                    int tile0, int nk) {
        d.a1 = off(a1); d.a2 = off(a2); d.b1 = off(b1); d.b2 = off(b2); d.c0 = off(c);
        d.colB = r->rc32; d.offCn = cn;
        d.N = M; d.ntiles = nt; d.tile_start = tile0; d.nk1 = nk1; d.nk = nk;
        d.inv_width = tgemm_inverse(gm * nt);
    };
    // (the long tiles first: what is dealt last decides how ragged the end of the launch is)
    fill(g[0], r->t2p, r->cp, r->F2, r->F3, r->Ivo, r->cn1, 0, 2 * nk1);
    fill(g[1], r->cp, r->cp, r->F1, r->F1, r->nIo, r->cn1, mt * nt, nk1);
    g[2].tile_start = 2 * mt * nt;
    // (the two terms with rows (j,b), columns (i,a) open the residual itself: s.r2 lives as long as the state)
    fill(g[3], r->t2p, r->Ivo, r->nIo, r->asp, s.r2.d, r->cn2, 0, 2 * nk1);
    fill(g[4], r->nIo, r->nIo, r->t2x, r->t2x, r->Y, r->cn2, mt * nt, nk1);
    g[5].tile_start = 2 * mt * nt;
    r->groups = (TgGroup*)cx.alloc((int64_t)(6 * sizeof(TgGroup) / (sizeof(double)) + 1));
    AFESP_HIP(hipMemcpyAsync(r->groups, g, sizeof(g), hipMemcpyHostToDevice, cx.stream));
    AFESP_HIP(hipStreamSynchronize(cx.stream));   // g is a temporary
    r->pairs2 = (O % 2 == 0);
    return r;
}

void ring_free(Context& cx, CCState& s)
{
    RingTg* r = (RingTg*)s.ring;
    if (!r) return;
    cx.release(r->slab);
    cx.release(r->tab_block);
    cx.release(r->groups);
    delete r;
    s.ring = nullptr;
}

bool ring_live(const CCState& s) { return s.ring && ((RingTg*)s.ring)->live; }
bool ring_res_live(const CCState& s) { return s.ring && ((RingTg*)s.ring)->res_live; }
void ring_res_clear(CCState& s) { if (s.ring) ((RingTg*)s.ring)->res_live = false; }
const double* ring_Y(const CCState& s) { return ((RingTg*)s.ring)->Y; }

// a view [K = (m,e) | row pair (p,x)] of one of the slab's buffers, indexed with the labels of the tensor it is copied from
static Tensor kview(const RingTg* r, double* buf, const CCState& s, int pos_m, int pos_p, int pos_x, int pos_e)
{
    Tensor t;
    t.d = buf;
    t.rank = 4;
    const int64_t O = s.o, V = s.v;
    t.dim[pos_m] = O; t.stride[pos_m] = 1;
    t.dim[pos_e] = V; t.stride[pos_e] = O;
    t.dim[pos_p] = O; t.stride[pos_p] = r->Kc;
    t.dim[pos_x] = V; t.stride[pos_x] = r->Kc * O;
    return t;
}

// asym_t2, c and the amplitudes' four copies in one pass (instead of k_asym_c + four permuting copies); false: not available for these
// extents (the o^2 block of one (p,q) does not fit the LDS) -- the caller runs k_asym_c
bool ring_tg_pack(Context& cx, CCState& s)
{
    const bool off = !knobs().ring_pack;
    const int64_t oo = (int64_t)s.o * s.o;
    if (off || oo * 8 > 65536 || oo >= 65536 / 4) return false;
    RingTg* r = ring_get(cx, s);
    RingPackArgs a;
    a.t1 = s.t1.d; a.t2 = s.t2.d; a.asym = s.asym.d; a.c = s.c.d;
    a.cp = r->cp; a.t2p = r->t2p; a.t2x = r->t2x; a.asp = r->asp;
    a.o = s.o; a.v = s.v; a.Kc = r->Kc;
    a.P = a.Q = oo * 16 * 8 <= 65536 ? 4 : oo * 4 * 8 <= 65536 ? 2 : 1;
    a.inv_o = tgemm_inverse(s.o);
    a.inv_oo = tgemm_inverse((int)oo);
    const int tiles = ((s.v + a.P - 1) / a.P) * ((s.v + a.Q - 1) / a.Q);
    AFESP_KLAUNCH(ring_pack_kernel, dim3((unsigned)tiles), dim3(256), (size_t)(oo * a.P * a.Q * 8), cx.stream, a);
    AFESP_HIP(hipGetLastError());
    r->packed = true;
    return true;
}

// I_ovov' and I_voov' from the small terms that ccsd_intermediates has left in I_ovov / I_voov (reference layout)
void ring_tg_intermediates(Context& cx, CCState& s)
{
    RingTg* r = ring_get(cx, s);
    if (!r->frozen_built) {
        // the integrals' copies, factors folded in: 1/2 <mi|be>, 1/2 (2<im|be> - <im|eb>), -1/2 <im|be> as [(m,e) | (i,b)]
        permute_add(cx, 0.5, s.v_oovv, "mibe", 0.0, kview(r, r->F1, s, 0, 1, 2, 3), "mibe");
        permute_add(cx, 0.5, s.w_oovv, "imbe", 0.0, kview(r, r->F2, s, 1, 0, 2, 3), "imbe");
        permute_add(cx, -0.5, s.v_oovv, "imbe", 0.0, kview(r, r->F3, s, 1, 0, 2, 3), "imbe");
        r->frozen_built = true;
    }
    // the amplitudes' copies: c'(ja;me) = c(m,j,a,e), t2'(ja;me) = t2(m,j,e,a), t2x(ja;me) = t2(m,j,a,e), asym'(ia;me) = asym(m,i,e,a)
    if (!r->packed) {   // (ring_tg_pack has not made them with asym_t2 and c)
        permute_add(cx, 1.0, s.c, "mjae", 0.0, kview(r, r->cp, s, 0, 1, 2, 3), "mjae");
        permute_add(cx, 1.0, s.t2, "mjea", 0.0, kview(r, r->t2p, s, 0, 1, 3, 2), "mjea");
        permute_add(cx, 1.0, s.t2, "mjae", 0.0, kview(r, r->t2x, s, 0, 1, 2, 3), "mjae");
        permute_add(cx, 1.0, s.asym, "miea", 0.0, kview(r, r->asp, s, 0, 1, 3, 2), "miea");
    }
    r->packed = false;
    const int ktail4 = (int)((r->ov - (r->Kc - TG_BK) + 3) / 4);
    TgProblem p{r->slab, r->slab, r->slab, r->rc32, r->cm1, (int)r->ov, true, ktail4};
    p.tag = 1;
    AFESP_HIP(tgemm_launch(p, r->groups, 2, r->tiles, r->mx, cx.stream, cx.tg));
    // + the small terms: -I_ovov'[(i,b) | (j,a)] -= I_ovov(j,b,i,a),  I_voov'[(i,b) | (j,a)] += I_voov(b,j,i,a)
    permute_add(cx, -1.0, s.I_ovov, "jbia", 1.0, kview(r, r->nIo, s, 2, 0, 3, 1), "jbia");
    permute_add(cx, 1.0, s.I_voov, "bjia", 1.0, kview(r, r->Ivo, s, 2, 1, 3, 0), "bjia");
    r->live = true;
}

// the three ring terms of the T2 residual: two into r2(i,j,a,b) itself (stored, not added: call it first), one into Y (i and j exchanged)
void ring_tg_residual(Context& cx, CCState& s)
{
    RingTg* r = (RingTg*)s.ring;
    if (!r || !r->live) throw Error(2, "ring_tg_residual: the intermediates of this iteration were not formed on this path");
    const int ktail4 = (int)((r->ov - (r->Kc - TG_BK) + 3) / 4);
    TgProblem p{r->slab, r->slab, r->slab, r->rc32, r->cm2, (int)r->ov, r->pairs2, ktail4};
    p.tag = 1;
    AFESP_HIP(tgemm_launch(p, r->groups + 3, 2, r->tiles, r->mx, cx.stream, cx.tg));
    r->res_live = true;
}

// tests / afesp_ccsd_get_tensor: the two intermediates in the reference's layout
void ring_tg_materialize(Context& cx, CCState& s, const Tensor& I_ovov_out, const Tensor& I_voov_out)
{
    RingTg* r = (RingTg*)s.ring;
    permute_add(cx, -1.0, kview(r, r->nIo, s, 2, 0, 3, 1), "jbia", 0.0, I_ovov_out, "jbia");
    permute_add(cx, 1.0, kview(r, r->Ivo, s, 2, 1, 3, 0), "bjia", 0.0, I_voov_out, "bjia");
}

void ring_reinit(CCState& s)
{
    ring_invalidate(s);
    if (s.ring) ((RingTg*)s.ring)->frozen_built = false;
}

void ring_invalidate(CCState& s)
{
    if (s.ring) { ((RingTg*)s.ring)->live = false; ((RingTg*)s.ring)->res_live = false; ((RingTg*)s.ring)->packed = false; }
}

}  // namespace afesp
