// tgemm.hip -- LDS-DMA staged grouped fp64 GEMM on v_mfma_f64_16x16x4_f64 (gfx950).  See tgemm.h.
//
// Workgroup = 4 waves (2 x 2), tile 128 x 128, K step 16; a wave owns 64 x 64 = 4 x 4 accumulators (128 VGPRs; on gfx950 the
// f64 MFMA issues every 64 cycles with VGPR accumulators and every 138 with AGPR ones, so the bound of two waves per SIMD --
// 256 registers -- is also what keeps hipcc from moving them).  Two workgroups per CU: each SIMD holds one wave of each.
//
// LDS: two stages of [A image 16 KiB | B image 16 KiB], then two slots of C offsets (128 row + 128 column int64 each).
//   image row = one row (column) of the operand, 16 doubles = 128 B = eight 16-byte chunks; chunk c of row r is stored at
//   chunk position c ^ ((r >> 1) & 7): a wave-wide ds_read_b128 of one chunk column of 16 rows is then bank-conflict free.
//   A global_load_lds_dwordx4 writes 1 KiB = 8 image rows, lane l -> row l >> 3, position l & 7; the swizzle is applied to
//   the SOURCE address (lane l fetches chunk (l & 7) ^ swizzle(row)).
// Fragments: lane (m = l & 15, f = l >> 4) reads chunk f + 4h of its row in half h = 0, 1 of a step: two consecutive k.  The
// first doubles of the four f feed one MFMA (k = 2(f + 4h)), the second doubles the next (k + 1); A and B use the same map, so
// every k of the step is summed exactly once.
//
// One stream step (stage `cur` holds this step's data, F0 its first-half fragments):
//     read F1(cur) | 32 MFMA on F0 | s_waitcnt vmcnt(0) lgkmcnt(0), s_barrier | read F0(cur^1) of the next step |
//     DMA of step + 2 into stage cur | 32 MFMA on F1 | (last step of a tile: store C, clear the accumulators)
// The barrier in the middle of the step says: every wave has its fragments of stage `cur` in registers (the stage may be
// overwritten) and every wave's DMA of the next step has landed (it was issued one whole step earlier).
// All vector-memory instructions of the loop are issued and waited for by hand (asm), so hipcc's own wait insertion never
// sees a transfer in flight; the tables of the next tile are ordinary loads issued right after a DMA batch and first used
// right after the following barrier's vmcnt(0).
#include "tgemm.h"

#include <algorithm>
#include <cstdlib>

namespace afesp {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TG_STAGE = 32768;             // bytes per stage
constexpr int TG_BIMG = 16384;              // B image inside a stage
constexpr int TG_SIDE = 2 * TG_STAGE;       // C-offset slots
constexpr int TG_LDS = TG_SIDE + 2 * 2048;
// marks a row / column beyond M / N in the C-offset slots (a valid offset may well be negative: the (T) block pool is pieces of
// memory addressed relative to the first one)
constexpr int64_t TG_NONE = INT64_MIN;

struct TgArgs {
    TgProblem p;
    const TgGroup* groups;
    int mtiles, gm, total_tiles;
    unsigned inv_gm, inv_gl;   // reciprocals (tgemm_inverse) of gm and of the size of the last, partial group of m-tiles
    int dbg;                   // measurement only (AFESP_TG_DBG): 1 no C stores, 2 no DMA after the prologue, 4 no vmcnt wait at the barrier
};

__device__ __forceinline__ int tg_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t tg_uni64(int64_t x)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)x), hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
    return ((int64_t)hi << 32) | (unsigned)lo;
}
template <typename T>
__device__ __forceinline__ const T* tg_uniptr(const T* p)
{
    typedef const T __attribute__((address_space(1)))* gptr;
    return (const T*)reinterpret_cast<gptr>(tg_uni64(reinterpret_cast<int64_t>(p)));
}
__device__ __forceinline__ int tg_xcd_remap(int b, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__global__ __launch_bounds__(256, 2) void tgemm_kernel(TgArgs a)
{
    __shared__ __attribute__((aligned(1024))) unsigned char lds[TG_LDS];
    const TgProblem& p = a.p;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = tg_uni(t >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const unsigned lds0 = (unsigned)tg_uni((int)(unsigned)(uintptr_t)lds);   // LDS byte address of the array (low half of the flat address)

    // ---- tile walk: as gett_kernel (groups of gm m-tiles x all n-tiles of a column group, XCD remap inside a round)
    // (the descriptors are read through the constant address space: scalar loads, no vector-memory wait in the stream)
    typedef const TgGroup __attribute__((address_space(4)))* cgptr;
    const cgptr G = (cgptr)(uintptr_t)a.groups;
    const int nwg = (int)gridDim.x;
    const int ntl = (a.total_tiles - (int)blockIdx.x + nwg - 1) / nwg;
    int cursor = 0;
    auto origin = [&](int j, int& m0, int& n0) {
        const int r0 = j * nwg;
        int tile = r0 + tg_xcd_remap((int)blockIdx.x, min(nwg, a.total_tiles - r0));
        while (tile >= G[cursor + 1].tile_start) ++cursor;
        tile -= G[cursor].tile_start;
        const int nt = G[cursor].ntiles;
        // (divisions by multiplication with reciprocals made on the host: scalar instructions only -- a division by a run-time
        // number goes through the vector unit, and a vector register it overwrites may still be the target of a table load,
        // which would put a vmcnt(0) behind the DMA batch just issued)
        const unsigned iw = G[cursor].inv_width;
        const int width = a.gm * nt, grp = iw ? (int)__umulhi((unsigned)tile, iw) : tile, first = grp * a.gm;
        const int gsz = min(a.mtiles - first, a.gm), rem = tile - grp * width;
        const unsigned ig = gsz == a.gm ? a.inv_gm : a.inv_gl;
        const int col = ig ? (int)__umulhi((unsigned)rem, ig) : rem;
        m0 = (first + rem - col * gsz) * TG_BM;
        n0 = col * TG_BN;
    };

    // ---- DMA lanes: wave w, instruction q writes image rows 8 (4w + q) + (l >> 3); source chunk = (l & 7) ^ swizzle(row)
    const int drow = 32 * wave + (lane >> 3);                                   // + 8 q
    const unsigned dch0 = (unsigned)(((lane & 7) ^ (lane >> 4)) * 16);          // q even
    const unsigned dch1 = (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) * 16);    // q odd
    // ---- fragment lanes
    const int fm = lane & 15, ff = lane >> 4, fs = fm >> 1;
    unsigned rdA0 = (unsigned)((wm * 64 + fm) * 128 + ((ff ^ fs) & 7) * 16);
    unsigned rdA1 = (unsigned)((wm * 64 + fm) * 128 + (((ff + 4) ^ fs) & 7) * 16);
    unsigned rdB0 = (unsigned)(TG_BIMG + (wn * 64 + fm) * 128 + ((ff ^ fs) & 7) * 16);
    unsigned rdB1 = (unsigned)(TG_BIMG + (wn * 64 + fm) * 128 + (((ff + 4) ^ fs) & 7) * 16);

    // ---- fetch cursor (two steps ahead of the MFMAs) and the tile entered but not yet committed
    int fj = 0, fkt = 0, fnk = 0, fnk1 = 0;
    const char *fa1 = nullptr, *fa2 = nullptr, *fb1 = nullptr, *fb2 = nullptr;
    bool ffull = false;
    unsigned voffA[4], voffB[4];
    int pnk = 0, pnk1 = 0;
    const char *pa1 = nullptr, *pa2 = nullptr, *pb1 = nullptr, *pb2 = nullptr;
    bool pfull = false, pending = false;
    unsigned nvA[4], nvB[4];
    int64_t nc = 0;
    bool nc_valid = false;

    auto enter_tile = [&](int j) {
        int m0, n0;
        origin(j, m0, n0);
        const cgptr g = G + cursor;
        const int N = g->N;
        pnk = g->nk;
        pnk1 = g->nk1;
        pa1 = (const char*)(p.A + g->a1);
        pa2 = (const char*)(p.A + g->a2);
        pb1 = (const char*)p.B;
        pb2 = (const char*)(p.B + g->b2);
        pfull = (m0 + TG_BM <= p.M) && (n0 + TG_BN <= N);
        typedef const uint32_t __attribute__((address_space(1)))* u32g;
        typedef const int64_t __attribute__((address_space(1)))* i64g;
        const uint32_t* colB = (const uint32_t*)(u32g)(uintptr_t)g->colB;
        const int64_t* offCn = (const int64_t*)(i64g)(uintptr_t)g->offCn;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = m0 + drow + 8 * q, n = n0 + drow + 8 * q;
            nvA[q] = p.rowA[m < p.M ? m : p.M - 1];
            nvB[q] = colB[n < N ? n : N - 1];
        }
        // (validity is applied when the value is used, so that nothing here waits for a load)
        {
            const int m = m0 + t, n = n0 + t - 128;
            const int64_t* src = t < 128 ? p.offCm + (m < p.M ? m : p.M - 1) : offCn + (n < N ? n : N - 1);
            nc = *src;
            nc_valid = t < 128 ? m < p.M : n < N;
        }
        pending = true;
    };
    auto commit_tile = [&](int j) {
        // (every asm below is ordered after the barrier statement: the loads above are first needed here)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("" : "+v"(nvA[q]), "+v"(nvB[q]));
            voffA[q] = nvA[q] + ((q & 1) ? dch1 : dch0);
            voffB[q] = nvB[q] + ((q & 1) ? dch1 : dch0);
        }
        asm volatile("" : "+v"(nc));
        *reinterpret_cast<int64_t*>(lds + TG_SIDE + (j & 1) * 2048 + t * 8) = nc_valid ? nc : TG_NONE;
        fnk = pnk; fnk1 = pnk1; fa1 = pa1; fa2 = pa2; fb1 = pb1; fb2 = pb2; ffull = pfull;
        pending = false;
    };
    // eight 1-KiB transfers of this wave: A image rows 32w .. 32w+31, B image rows 32w .. 32w+31 of stage `stage`
    auto dma = [&](int stage) {
        // (readfirstlane: the values are uniform, but hipcc is free to compute a select of them in vector registers, and the
        // asm below needs scalar ones; the s_nop 4 that opens it covers the VALU-write -> VMEM-read hazard of such a register)
        const int k = tg_uni(fkt), k1 = tg_uni(fnk1);
        const char* ap = (const char*)tg_uni64((int64_t)((k < k1) ? fa1 + (int64_t)k * 128 : fa2 + (int64_t)(k - k1) * 128));
        const char* bp = (const char*)tg_uni64((int64_t)((k < k1) ? fb1 + (int64_t)k * 128 : fb2 + (int64_t)(k - k1) * 128));
        const unsigned dst = (unsigned)tg_uni((int)(lds0 + (unsigned)(stage * TG_STAGE + wave * 4096)));
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 m0, %[d]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a0], %[ap]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a1], %[ap]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a2], %[ap]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a3], %[ap]\n\t"
            "s_add_u32 m0, m0, 0x3400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b0], %[bp]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b1], %[bp]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b2], %[bp]\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b3], %[bp]"
            :
            : [a0] "v"(voffA[0]), [a1] "v"(voffA[1]), [a2] "v"(voffA[2]), [a3] "v"(voffA[3]), [b0] "v"(voffB[0]),
              [b1] "v"(voffB[1]), [b2] "v"(voffB[2]), [b3] "v"(voffB[3]), [ap] "s"(ap), [bp] "s"(bp), [d] "s"(dst)
            : "memory", "scc");
    };
    // One fetch step: the DMA batch of the cursor's step into `stage`.  When it is the last step of its tile the next tile is
    // entered FIRST: its table loads are ordinary loads, and whatever wait hipcc attaches to them (a register it reuses may
    // still be the target of the previous tile's loads as far as its path-insensitive bookkeeping knows) must not come behind
    // the batch -- in front of it nothing is in flight, the step's barrier has just drained the queue.
    int cj_started = 0;   // (0 during the prologue)
    auto fetch_step = [&](int stage) {
        const bool last = fkt + 1 == fnk;
        if (last && fj + 1 < ntl) enter_tile(fj + 1);
        if (!(a.dbg & 2) || cj_started == 0) dma(stage);
        fkt = last ? 0 : fkt + 1;   // (written as selects: an if/else of increments made hipcc keep the two counters in scratch memory)
        fj += last ? 1 : 0;
    };

    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    v2d fa0[4], fb0[4], fa1_[4], fb1_[4];
    auto frag = [&](v2d (&fa)[4], v2d (&fb)[4], unsigned ra, unsigned rb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v2d*>(lds + ra + i * 2048);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const v2d*>(lds + rb + j * 2048);
    };
    auto mfma = [&](const v2d (&fa)[4], const v2d (&fb)[4]) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
    };
    // C(m0 + 64 wm + 16 i + (l >> 4) + 4 r, n0 + 64 wn + 16 j + (l & 15)) = acc[i][j][r]; offsets from the tile's slot
    auto store_tile = [&](int slot, bool full) {
        const int64_t* sc = reinterpret_cast<const int64_t*>(lds + TG_SIDE + slot * 2048);
        int64_t cn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cn[j] = sc[128 + wn * 64 + 16 * j + fm];
        if (full) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t cm = sc[wm * 64 + 16 * i + 4 * r + ff];
#pragma unroll
                    for (int j = 0; j < 4; ++j) p.C[cm + cn[j]] = acc[i][j][r];
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t cm = sc[wm * 64 + 16 * i + 4 * r + ff];
                    if (cm == TG_NONE) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cn[j] != TG_NONE) p.C[cm + cn[j]] = acc[i][j][r];
                }
        }
    };

    // ---- prologue: tile 0, steps 0 and 1 (every group has nk >= 2)
    enter_tile(0);
    commit_tile(0);
    int kt = 0, cj = 0, cur = 0, nk_cur = fnk;
    bool full_cur = ffull;
    fetch_step(0);
    fetch_step(1);
    cj_started = 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    frag(fa0, fb0, rdA0, rdB0);

    // ---- stream of steps
    for (;;) {
        frag(fa1_, fb1_, rdA1, rdB1);
        __builtin_amdgcn_sched_barrier(0);
        mfma(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        if (a.dbg & 4) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // the stage just read is free; the other one holds the next step
        const int freed = cur;
        cur ^= 1;
        rdA0 ^= TG_STAGE; rdA1 ^= TG_STAGE; rdB0 ^= TG_STAGE; rdB1 ^= TG_STAGE;
        frag(fa0, fb0, rdA0, rdB0);
        if (pending) commit_tile(fj);
        if (fj < ntl) {
            fetch_step(freed);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma(fa1_, fb1_);
        __builtin_amdgcn_sched_barrier(0);
        if (++kt == nk_cur) {
            if (!(a.dbg & 1)) store_tile(cj & 1, full_cur);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
            kt = 0;
            if (++cj == ntl) break;
            // every group has nk >= 2: the next tile was committed one step ago at the latest, and the one after it
            // (entered in this very step if the next tile has two steps) is not committed before the next step
            nk_cur = fnk;
            full_cur = ffull;
        }
    }
}

// floor(x / d) == umulhi(x, tgemm_inverse(d)) for x * d < 2^32 (tile ids inside a group and patch widths are far below);
// 0 stands for d = 1 (no 32-bit factor reproduces x itself)
unsigned tgemm_inverse(int d) { return d <= 1 ? 0u : (unsigned)(((uint64_t)1 << 32) / (unsigned)d + 1); }
// a patch of gm m-tiles x all n-tiles of a group = the ~64 tiles one XCD works on in a round
int tgemm_group_m(int M, int max_ntiles)
{
    const int mtiles = (M + TG_BM - 1) / TG_BM;
    return std::min(mtiles, std::max(1, (64 + max_ntiles / 2) / std::max(1, max_ntiles)));
}

void preload_tgemm()
{
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(tgemm_kernel));
    (void)hipGetLastError();
}

hipError_t tgemm_launch(const TgProblem& p, const TgGroup* dev_groups, int ngroups, int total_tiles, int max_ntiles, hipStream_t stream)
{
    if (p.M <= 0 || ngroups <= 0 || total_tiles <= 0) return hipSuccess;
    static int cap = 0;
    if (cap == 0) {
        int dev = 0, cus = 256, occ = 2;
        if (hipGetDevice(&dev) == hipSuccess) {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tgemm_kernel, 256, 0) != hipSuccess || occ <= 0) occ = 2;
        }
        cap = cus * occ;
    }
    TgArgs a;
    a.p = p;
    a.groups = dev_groups;
    a.mtiles = (p.M + TG_BM - 1) / TG_BM;
    a.total_tiles = total_tiles;
    static const int dbg_env = getenv("AFESP_TG_DBG") ? atoi(getenv("AFESP_TG_DBG")) : 0;
    a.dbg = dbg_env;
    a.gm = tgemm_group_m(p.M, max_ntiles);
    a.inv_gm = tgemm_inverse(a.gm);
    a.inv_gl = tgemm_inverse(std::max(1, a.mtiles % a.gm));
    static const int grid_env = getenv("AFESP_TG_GRID") ? atoi(getenv("AFESP_TG_GRID")) : 0;   // diagnostic: fewer workgroups, longer tile streams
    const unsigned grid = (unsigned)std::min(total_tiles, grid_env > 0 ? grid_env : cap);
    hipLaunchKernelGGL(tgemm_kernel, dim3(grid), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace afesp
