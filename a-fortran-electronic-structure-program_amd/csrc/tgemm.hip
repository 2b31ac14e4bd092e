// tgemm.hip -- LDS-DMA staged grouped fp64 GEMM on v_mfma_f64_16x16x4_f64 (gfx950).  See tgemm.h.
//
// Workgroup = 4 waves (2 x 2), tile 128 x 128, K step 16; a wave owns 64 x 64 = 4 x 4 accumulators (128 VGPRs; on gfx950 the
// f64 MFMA issues every 64 cycles with VGPR accumulators and every 138 with AGPR ones, so the bound of two waves per SIMD --
// 256 registers -- is also what keeps hipcc from moving them).  Two workgroups per CU: each SIMD holds one wave of each.
//
// LDS: two stages of [A image 16 KiB | B image 16 KiB], then two slots of a tile's tables (C row / column offsets, A row /
// B column byte offsets), which arrive by LDS-DMA as well.
//   image row = one row (column) of the operand, 16 doubles = 128 B = eight 16-byte chunks; chunk c of row r is stored at
//   chunk position c ^ ((r >> 1) & 7): a wave-wide ds_read_b128 of one chunk column of 16 rows is then bank-conflict free.
//   A global_load_lds_dwordx4 writes 1 KiB = 8 image rows, lane l -> row l >> 3, position l & 7 (LDS address = M0 +
//   instruction offset + 16 l, M0 any LDS byte address: tools/glds_probe.hip); the swizzle is applied to the SOURCE address
//   (lane l fetches chunk (l & 7) ^ swizzle(row)).
// Fragments: lane (m = l & 15, f = l >> 4) reads chunk f + 4h of its row in half h = 0, 1 of a step: two consecutive k.  The
// first doubles of the four f feed one MFMA (k = 2(f + 4h)), the second doubles the next (k + 1); A and B use the same map, so
// every k of the step is summed exactly once.
// Columns: row 16 j + f' of the B image holds tile column col(j, f') = 32 (j >> 1) + 2 f' + (j & 1), so that a lane's
// accumulators (i, 2jj) and (i, 2jj + 1) hold two ADJACENT columns of C: where those are adjacent in memory too
// (TgProblem::c_pairs) the tile is stored with 32 global_store_dwordx4 per wave instead of 64 dwordx2 -- the store tail is
// bound by the number of store instructions, ~270 cycles each for a wave whatever their width (tools/tgemm_check.hip big).
//
// K tails: the last step of each of the two runs of the summation index carries only TgProblem::ktail4 groups of four valid k
// (the rest is zero padding).  The second MFMA pass of that step's second half is skipped; with three groups the second half's
// fragments are read as single doubles, lane f taking k = 8 + f, so that the first pass holds all of them (with fewer the second
// half is all zeros).
//
// One stream step (stage `cur` holds this step's data, F0 its first-half fragments):
//     32 MFMA on F0, F1(cur) requested in their first gaps | s_waitcnt vmcnt(0) lgkmcnt(0), s_barrier |
//     32 MFMA on F1, in their gaps: F0(cur^1) of the next step requested, the DMA of step + 2 into stage cur issued |
//     (last step of a tile: store C, clear the accumulators)
// The barrier in the middle of the step says: every wave has its fragments of stage `cur` in registers (the stage may be
// overwritten) and every wave's DMA of the next step has landed (it was issued one whole step earlier).
// All vector-memory loads of the kernel are LDS-DMA transfers issued and waited for by hand (asm) -- the tables of the next
// tile too -- so hipcc's own wait insertion never sees one in flight and never drains the queue.
#include "tgemm.h"

#include <algorithm>
#include <cstdlib>

namespace afesp {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TG_STAGE = 32768;             // bytes per stage
constexpr int TG_BIMG = 16384;              // B image inside a stage
constexpr int TG_SIDE = 2 * TG_STAGE;       // table slots of the tiles being fetched / computed, 4 KiB each:
constexpr int TG_SLOT = 4096;               //   [C row offsets 128 x int64 | C column offsets 128 x int64 | rowA 256 x u32 | colB 256 x u32]
constexpr int TG_NSLOT = 3;                 // (three: with two-step tiles the tile entered now is two behind the one whose epilogue still reads its slot)
constexpr int TG_TKT = TG_SIDE + TG_NSLOT * TG_SLOT;   // two ticket words (dynamic tile assignment)
constexpr int TG_LDS = TG_TKT + 64;

struct TgArgs {
    TgProblem p;
    const TgGroup* groups;
    int mtiles, gm, total_tiles;
    unsigned inv_gm, inv_gl;   // reciprocals (tgemm_inverse) of gm and of the size of the last, partial group of m-tiles
    int dbg;                   // measurement only (AFESP_TG_DBG): 1 no C stores
    // Dynamic tile assignment (large launches).  The matrix pipe of a SIMD goes to the older of its two waves, so of the two
    // workgroups of a CU one runs at up to twice the other's speed: with tiles dealt statically the workgroups of an XCD end up
    // hundreds of microseconds apart (their common operand lines are long gone from the L2 when the slow ones ask for them) and
    // the fast ones idle at the end of the launch.  Instead each XCD has a ticket counter (tickets[xcd], zeroed before the
    // launch): ticket k of XCD x is tile (k / chunk) * grid + chunk * x + k % chunk -- the tiles of that XCD's patches in order --
    // and a workgroup draws its next ticket one tile ahead.  nullptr: tiles dealt statically.
    unsigned* tickets;
    int chunk;                 // grid / 8
    unsigned inv_chunk;
    // Time-sliced priority.  At equal priority the older wave of a SIMD wins the matrix pipe, i.e. the workgroup dispatched first
    // on a CU runs at up to twice the speed of the second (tools/tgemm_check.hip, stamps: the first half of the grid ends its 8th
    // tile 0.07 ms after the earliest, the second half 0.26 ms; with s_setprio 3 on the second half it is the other way round).
    // So the halves of the grid take turns: in slices of 2^prio_shift x 10 ns of the device-wide clock (s_memrealtime) one half
    // runs at priority 1, then the other -- no communication, and the two workgroups of a CU stay level.  0: off.
    int prio_shift;
    // MIXED instantiation only: rows per m-tile, 128 or 96.  A tile with at most 96 rows left (every tile when bm = 96, the last one of
    // bm = 128 when M mod 128 <= 96) runs as a 96-row tile: a wave then owns 48 x 64 = 3 x 4 accumulators, its workgroup's A image has
    // 96 rows (wave w stages rows 24 w .., fragments of wave row wm start at row 48 wm) and a quarter of the MFMAs is not issued.
    int bm;
};

#ifdef TG_STAMPS
// Diagnostic builds only (tools/tgemm_check.hip -DTG_STAMPS): per (workgroup, wave) cycle sums -- [0] the whole stream, [1] the
// step barriers (wait + barrier), [2] the tile epilogues, [3] steps, [4] tiles, [5] the longest single barrier
__device__ unsigned long long g_tg_stamp[512 * 4 * 8];
#endif

__device__ __forceinline__ int tg_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int64_t tg_uni64(int64_t x)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)x), hi = __builtin_amdgcn_readfirstlane((int)(x >> 32));
    return ((int64_t)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ int tg_xcd_remap(int b, int nwg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <bool MIXED>
__device__ __forceinline__ void tgemm_body(const TgArgs& a)
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
    const bool dyn = a.tickets != nullptr;
    const int xcd = (int)blockIdx.x & 7;
    unsigned* tslot = reinterpret_cast<unsigned*>(lds + TG_TKT);
    // tile number n of this workgroup's stream -> tile id, or -1 behind the last one
    auto tile_static = [&](int n) { return n < ntl ? n * nwg + tg_xcd_remap((int)blockIdx.x, min(nwg, a.total_tiles - n * nwg)) : -1; };
    auto tile_ticket = [&](unsigned k) {
        const unsigned q = a.inv_chunk ? __umulhi(k, a.inv_chunk) : k;
        const int64_t tile = (int64_t)q * nwg + (int64_t)a.chunk * xcd + (k - q * (unsigned)a.chunk);
        return tile < a.total_tiles ? (int)tile : -1;
    };
    int cursor = 0;
    auto origin = [&](int tile, int& m0, int& n0) {
        // (tile ids only grow along a workgroup's stream, statically dealt or drawn)
        while (tile >= G[cursor + 1].tile_start) ++cursor;
        tile -= G[cursor].tile_start;
        const int nt = G[cursor].ntiles;
        // (divisions by multiplication with reciprocals made on the host: scalar instructions only)
        const unsigned iw = G[cursor].inv_width;
        const int width = a.gm * nt, grp = iw ? (int)__umulhi((unsigned)tile, iw) : tile, first = grp * a.gm;
        const int gsz = min(a.mtiles - first, a.gm), rem = tile - grp * width;
        const unsigned ig = gsz == a.gm ? a.inv_gm : a.inv_gl;
        const int col = ig ? (int)__umulhi((unsigned)rem, ig) : rem;
        m0 = (first + rem - col * gsz) * (MIXED ? a.bm : TG_BM);
        n0 = col * TG_BN;
    };

    // ---- DMA lanes: wave w, instruction q writes image rows 8 (4w + q) + (l >> 3); source chunk = (l & 7) ^ swizzle(row)
    const int drow = 32 * wave + (lane >> 3);                                   // + 8 q
    const unsigned dch0 = (unsigned)(((lane & 7) ^ (lane >> 4)) * 16);          // q even
    const unsigned dch1 = (unsigned)(((lane & 7) ^ (4 + (lane >> 4))) * 16);    // q odd
    const unsigned lane4 = (unsigned)(lane * 4);
    // ---- fragment lanes
    const int fm = lane & 15, ff = lane >> 4, fs = fm >> 1;
    unsigned rdA0 = (unsigned)((wm * 64 + fm) * 128 + ((ff ^ fs) & 7) * 16);
    unsigned rdA1 = (unsigned)((wm * 64 + fm) * 128 + (((ff + 4) ^ fs) & 7) * 16);
    unsigned rdB0 = (unsigned)(TG_BIMG + (wn * 64 + fm) * 128 + ((ff ^ fs) & 7) * 16);
    unsigned rdB1 = (unsigned)(TG_BIMG + (wn * 64 + fm) * 128 + (((ff + 4) ^ fs) & 7) * 16);
    // (K tail of three groups: lane f reads the single double k = 8 + f = element f & 1 of chunk 4 + (f >> 1))
    unsigned rdA1t = (unsigned)((wm * 64 + fm) * 128 + (((4 + (ff >> 1)) ^ fs) & 7) * 16 + (ff & 1) * 8);
    unsigned rdB1t = (unsigned)(TG_BIMG + (wn * 64 + fm) * 128 + (((4 + (ff >> 1)) ^ fs) & 7) * 16 + (ff & 1) * 8);

    // ---- fetch cursor (two steps ahead of the MFMAs) and the tile entered but not yet committed
    int entered = 0, fkt = 0, fnk = 0, fnk1 = 0;   // tiles entered so far; the cursor's step inside the last of them
    bool f_live = true;                             // the cursor points at a tile (false behind the last one)
    unsigned tk = 0;                                // the ticket drawn for the tile after the next (lane 0 of wave 0)
    bool tk_pending = false;
    int fmrem = 0, fnrem = 0;   // rows / columns of C from the tile's origin to M / N
    const char *fa1 = nullptr, *fa2 = nullptr, *fb1 = nullptr, *fb2 = nullptr;
    unsigned voffA[4], voffB[4];
    int pnk = 0, pnk1 = 0, pmrem = 0, pnrem = 0;
    int pni = 4, fni = 4, cni = 4;   // (MIXED) fragment rows per wave of the tile entered / fetched / computed: 4, or 3 for a 96-row tile
    int64_t pc0 = 0, fc0 = 0;
    const char *pa1 = nullptr, *pa2 = nullptr, *pb1 = nullptr, *pb2 = nullptr;
    bool pending = false;

    // A tile is entered one step before its first transfer: its group's scalars are read (scalar loads) and its four tables go
    // to table slot j % 3, one per wave, 1 KiB each (the row tables are read 128 entries beyond the tile: the host pads them).
    auto enter_tile = [&](int tile) {
        const int j = entered;
        int m0, n0;
        origin(tile, m0, n0);
        const cgptr g = G + cursor;
        const int N = g->N;
        pnk = g->nk;
        pnk1 = g->nk1;
        pa1 = (const char*)(p.A + g->a1);
        pa2 = (const char*)(p.A + g->a2);
        pb1 = (const char*)(p.B + g->b1);
        pc0 = g->c0;
        pb2 = (const char*)(p.B + g->b2);
        pmrem = p.M - m0;
        pnrem = N - n0;
        if (MIXED) pni = (a.bm == 96 || pmrem <= 96) ? 3 : 4;
        const char* src = wave == 0   ? (const char*)(p.offCm + m0)
                          : wave == 1 ? (const char*)((const int64_t*)(uintptr_t)g->offCn + n0)
                          : wave == 2 ? (const char*)(p.rowA + m0)
                                      : (const char*)((const uint32_t*)(uintptr_t)g->colB + n0);
        src = (const char*)tg_uni64((int64_t)src);
        const unsigned dst = (unsigned)tg_uni((int)(lds0 + (unsigned)(TG_SIDE + (j % TG_NSLOT) * TG_SLOT + wave * 1024)));
        // (the instruction offset moves the source and the LDS address alike; s_nop 4: a scalar register fresh from a
        // readfirstlane must not be read by a vector-memory instruction in the next five states)
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
                     "global_load_lds_dword %[v], %[p]\n\tglobal_load_lds_dword %[v], %[p] offset:256\n\t"
                     "global_load_lds_dword %[v], %[p] offset:512\n\tglobal_load_lds_dword %[v], %[p] offset:768"
                     :
                     : [v] "v"(lane4), [p] "s"(src), [d] "s"(dst)
                     : "memory");
        pending = true;
        ++entered;
    };
    // ... and committed after the next barrier (its tables are in LDS): row / column byte offsets of this thread's transfers
    auto commit_tile = [&](int j) {
        const unsigned* sr = reinterpret_cast<const unsigned*>(lds + TG_SIDE + (j % TG_NSLOT) * TG_SLOT + 2048);
        const int lm = min(MIXED ? 32 * pni : TG_BM, pmrem) - 1, ln = min(TG_BN, pnrem) - 1;   // rows / columns beyond M / N fetch the last valid one
        // (96-row tile: wave w stages A rows 24 w + 8 q .., so the swizzle parity of a transfer follows w + q; its fourth transfer repeats
        // the next wave's first one -- same bytes to the same place -- and wave 3's lands behind the image)
        const int arow = MIXED ? 8 * pni * wave + (lane >> 3) : drow;
        const int apar = (MIXED && pni == 3) ? (wave & 1) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = drow + 8 * q, ra = arow + 8 * q;
            const int c = 32 * (r >> 5) + 2 * (r & 15) + ((r >> 4) & 1);   // tile column of B image row r (header: Columns)
            voffA[q] = sr[min(ra, lm)] + (((q + apar) & 1) ? dch1 : dch0);
            voffB[q] = sr[256 + min(c, ln)] + ((q & 1) ? dch1 : dch0);
        }
        fni = pni;
        fnk = pnk; fnk1 = pnk1; fa1 = pa1; fa2 = pa2; fb1 = pb1; fb2 = pb2; fmrem = pmrem; fnrem = pnrem; fc0 = pc0;
        pending = false;
    };
    // The eight 1-KiB transfers of this wave for the cursor's step (A image rows 32w .. 32w+31, then B image rows 32w .. 32w+31 of
    // the stage) are issued one by one between MFMAs (TG_DMA below); dma_setup computes their scalar operands.
    // (readfirstlane: the values are uniform, but hipcc is free to compute a select of them in vector registers, and the asm
    // needs scalar ones.  Every piece opens with s_nop 4: a scalar register written by the vector unit -- a readfirstlane, or the
    // v_readlane with which hipcc reloads a SPILLED scalar register right in front of the statement -- must not be read by a
    // vector-memory instruction in the next five states, and hipcc pads no hazard of an asm statement)
    const char *dma_ap = nullptr, *dma_bp = nullptr;
    unsigned dma_dst = 0, dma_dsta = 0;
    auto dma_setup = [&](int stage) {
        const int k = tg_uni(fkt), k1 = tg_uni(fnk1);
        dma_ap = (const char*)tg_uni64((int64_t)((k < k1) ? fa1 + (int64_t)k * 128 : fa2 + (int64_t)(k - k1) * 128));
        dma_bp = (const char*)tg_uni64((int64_t)((k < k1) ? fb1 + (int64_t)k * 128 : fb2 + (int64_t)(k - k1) * 128));
        dma_dst = (unsigned)tg_uni((int)(lds0 + (unsigned)(stage * TG_STAGE + wave * 4096)));
        dma_dsta = MIXED ? (unsigned)tg_uni((int)(lds0 + (unsigned)(stage * TG_STAGE + wave * 1024 * fni))) : dma_dst;
    };
#define TG_DMA(Q)                                                                                                        \
    asm volatile("s_nop %c[nop]\n\ts_add_u32 m0, %[d], %[off]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[p]"         \
                 :                                                                                                       \
                 : [v] "v"((Q) < 4 ? voffA[(Q) & 3] : voffB[(Q) & 3]), [p] "s"((Q) < 4 ? dma_ap : dma_bp), [d] "s"((Q) < 4 ? dma_dsta : dma_dst), \
                   [off] "i"(((Q) < 4 ? 0 : TG_BIMG) + ((Q) & 3) * 1024), [nop] "i"(4)                    \
                 : "memory", "scc")
    // The cursor moves on one step.  When the step is the last one of its tile the next tile is entered BEFORE the step's own
    // transfers are issued (its table transfers are then the older ones; everything is waited for at the step's barrier).
    // the tile after the last one entered: dealt statically, or the ticket drawn for it one tile ago -- and the next draw
    auto acquire = [&]() {
        if (!dyn) return tile_static(entered);
        const unsigned k = (unsigned)tg_uni((int)tslot[entered & 1]);
        if (t == 0)   // (one lane; returns the counter's previous value; waited for at the step's barrier like every transfer)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(tk) : "v"(0u), "v"(1u), "s"(a.tickets + xcd) : "memory");
        tk_pending = true;
        return tile_ticket(k);
    };
    auto fetch_begin = [&](int stage) {
        const bool last = f_live && fkt + 1 == fnk;
        bool more = true;
        if (last) {
            const int nt = acquire();
            more = nt >= 0;
            if (more) enter_tile(nt);
        }
        dma_setup(stage);
        fkt = last ? 0 : f_live ? fkt + 1 : fkt;   // (written as selects: an if/else of increments made hipcc keep the counters in scratch memory)
        f_live = f_live && more;
    };

    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    v2d fa0[4], fb0[4], fa1_[4], fb1_[4];
    // acc[i][j][r] = C(m0 + 64 wm + 16 i + (l >> 4) + 4 r, n0 + 64 wn + col(j, l & 15)); offsets from the tile's slot
    auto store_tile = [&](int slot, int mrem, int nrem, int64_t c0, int ni) {
        double* const Cg = p.C + c0;
        const int64_t* sc = reinterpret_cast<const int64_t*>(lds + TG_SIDE + slot * TG_SLOT);
        const int wrow = MIXED ? wm * 16 * ni : wm * 64;   // first tile row of this wave
        if (mrem >= (MIXED ? 32 * ni : TG_BM) && nrem >= TG_BN && p.c_pairs) {
            int64_t cn[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) cn[jj] = sc[128 + wn * 64 + 32 * jj + 2 * fm];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MIXED && i >= ni) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t cm = sc[wrow + 16 * i + 4 * r + ff];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)   // (scalar base + 32-bit offset instead of 64-bit addresses: same time, measured)
                        *reinterpret_cast<v2d*>(Cg + cm + cn[jj]) = (v2d){acc[i][2 * jj][r], acc[i][2 * jj + 1][r]};
                }
            }
        } else if (p.c_pairs) {
            // a tile on the edge of C: still 16 bytes per lane where both columns of the pair exist -- single doubles leave every
            // 32-byte sector half written until the other parity's instruction comes, and the L2 fetches such sectors from HBM first
            // (AO->MO at n = 220, whose second row tile is partial: 55 GB fetched per transform, 44 GB with the stores off)
            int nl[2];
            int64_t cn[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                nl[jj] = wn * 64 + 32 * jj + 2 * fm;
                cn[jj] = sc[128 + nl[jj]];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ml = wrow + 16 * i + 4 * r + ff;
                    if (ml >= mrem || (MIXED && i >= ni)) continue;
                    const int64_t cm = sc[ml];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        if (nl[jj] + 1 < nrem) *reinterpret_cast<v2d*>(Cg + cm + cn[jj]) = (v2d){acc[i][2 * jj][r], acc[i][2 * jj + 1][r]};
                        else if (nl[jj] < nrem) Cg[cm + cn[jj]] = acc[i][2 * jj][r];
                    }
                }
        } else {
            int64_t cn[4];
            int nl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                nl[j] = wn * 64 + 32 * (j >> 1) + 2 * fm + (j & 1);
                cn[j] = sc[128 + nl[j]];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ml = wrow + 16 * i + 4 * r + ff;
                    if (ml >= mrem || (MIXED && i >= ni)) continue;
                    const int64_t cm = sc[ml];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (nl[j] < nrem) Cg[cm + cn[j]] = acc[i][j][r];
                }
        }
    };

    // MFMA number x of a half step: k element x >> 4 of the fragment pairs, accumulator ((x >> 2) & 3, x & 3)
#define TG_MF(FA, FB, X)                                                                                                 \
    acc[((X) >> 2) & 3][(X) & 3] =                                                                                       \
        __builtin_amdgcn_mfma_f64_16x16x4f64(FA[((X) >> 2) & 3][(X) >> 4], FB[(X) & 3][(X) >> 4], acc[((X) >> 2) & 3][(X) & 3], 0, 0, 0)
#define TG_SB __builtin_amdgcn_sched_barrier(0)
    // fragment number q of a half step: q < 4 the A rows 16 q .., else the B columns 16 (q - 4) ..
#define TG_FRAG(FA, FB, RA, RB, Q)                                                                                       \
    if ((Q) < 4) FA[(Q) & 3] = *reinterpret_cast<const v2d*>(lds + RA + ((Q) & 3) * 2048);                               \
    else FB[(Q) & 3] = *reinterpret_cast<const v2d*>(lds + RB + ((Q) & 3) * 2048)

    // ---- prologue: tile 0, steps 0 and 1 (every group has nk >= 2)
    int tile0;
    if (dyn) {
        if (t == 0) {   // the tickets of this workgroup's first two tiles
            tslot[0] = atomicAdd(a.tickets + xcd, 1u);
            tslot[1] = atomicAdd(a.tickets + xcd, 1u);
        }
        __syncthreads();
        tile0 = tile_ticket((unsigned)tg_uni((int)tslot[0]));
        if (tile0 < 0) return;   // (more workgroups than this XCD has tiles)
    } else {
        tile0 = tile_static(0);
    }
    enter_tile(tile0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    commit_tile(0);
    int kt = 0, cj = 0, cur = 0, nk_cur = fnk, nk1_cur = fnk1, mrem_cur = fmrem, nrem_cur = fnrem;
    int64_t c0_cur = fc0;
    if (MIXED) {
        cni = fni;
        if (cni == 3) { rdA0 -= (unsigned)(wm * 2048); rdA1 -= (unsigned)(wm * 2048); rdA1t -= (unsigned)(wm * 2048); }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        fetch_begin(st);
        TG_DMA(0); TG_DMA(1); TG_DMA(2); TG_DMA(3); TG_DMA(4); TG_DMA(5); TG_DMA(6); TG_DMA(7);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) { TG_FRAG(fa0, fb0, rdA0, rdB0, q); }
#ifdef TG_STAMPS
    unsigned long long st_bar = 0, st_epi = 0, st_steps = 0, st_tiles = 0, st_max = 0, st_t8 = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif

    // ---- stream of steps.  Everything that is not an MFMA is issued in the gaps between MFMAs (a 64-cycle MFMA leaves its wave
    // free to issue other instructions meanwhile): a wave whose partner on the SIMD is parked -- at its barrier, in its tile
    // epilogue -- then keeps the matrix pipe busy by itself.  sched_barrier pins the order written here.
    const unsigned my_half = (int)blockIdx.x >= nwg / 2 ? 1u : 0u;
    bool prio_hi = false;
    for (;;) {
        const unsigned long long rt = a.prio_shift ? __builtin_amdgcn_s_memrealtime() : 0ull;   // (consumed behind the barrier)
        // K tail of this step (header): t3 = its second half has one pass of single-double fragments, t2 = no second half
        const bool ktl = (kt + 1 == nk1_cur || kt + 1 == nk_cur) && p.ktail4 < 4;
        const bool t3 = ktl && p.ktail4 == 3;   // (fewer groups: the second half is all zeros, its first pass then runs on them)
        // first half: the second half's fragments are requested in the first eight gaps
#define TG_A(Q) TG_FRAG(fa1_, fb1_, rdA1, rdB1, Q); TG_SB; TG_MF(fa0, fb0, Q); TG_SB;
        TG_A(0) TG_A(1) TG_A(2) TG_A(3) TG_A(4) TG_A(5) TG_A(6) TG_A(7)
#undef TG_A
        // (a scalar branch between two MFMAs costs the matrix pipe tens of cycles -- sixteen of them in these gaps took the kernel
        // from 67 to 55 TFLOP/s -- so the rare cases get ONE branch each per step)
        if (t3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fa1_[q][0] = *reinterpret_cast<const double*>(lds + rdA1t + q * 2048);
                fb1_[q][0] = *reinterpret_cast<const double*>(lds + rdB1t + q * 2048);
            }
        }
        TG_SB;
        const bool ni4 = !MIXED || cni == 4;   // (a 96-row tile has no fourth row of accumulators: ONE branch per half step)
        if (!MIXED) {
#pragma unroll
            for (int x = 8; x < 32; ++x) TG_MF(fa0, fb0, x);
        } else {
            // (the fourth accumulator row last, under the one branch)
#pragma unroll
            for (int x = 8; x < 12; ++x) TG_MF(fa0, fb0, x);
#pragma unroll
            for (int x = 16; x < 28; ++x) TG_MF(fa0, fb0, x);
            TG_SB;
            if (ni4) {
#pragma unroll
                for (int x = 12; x < 16; ++x) TG_MF(fa0, fb0, x);
#pragma unroll
                for (int x = 28; x < 32; ++x) TG_MF(fa0, fb0, x);
            }
        }
        TG_SB;
#ifdef TG_STAMPS
        const unsigned long long st_b0 = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef TG_STAMPS
        {
            const unsigned long long d = __builtin_amdgcn_s_memtime() - st_b0;
            st_bar += d;
            st_max = d > st_max ? d : st_max;
            ++st_steps;
        }
#endif
        TG_SB;
        if (a.prio_shift) {
            const bool hi = (((unsigned)(rt >> a.prio_shift)) & 1u) == my_half;
            if (hi != prio_hi) {
                if (hi) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
                prio_hi = hi;
            }
        }
        // the stage just read is free; the other one holds the next step, whose first-half fragments are requested in the first
        // eight gaps of the second half
        const int freed = cur;
        cur ^= 1;
        rdA0 ^= TG_STAGE; rdA1 ^= TG_STAGE; rdB0 ^= TG_STAGE; rdB1 ^= TG_STAGE; rdA1t ^= TG_STAGE; rdB1t ^= TG_STAGE;
        if (MIXED) {
            // the fragments requested next are the NEXT step's: behind a tile's last step they lie in an image of the next tile's height
            // (that tile was committed a step ago at the latest, the one after it is not before the next step: fni is the next tile's)
            if (kt + 1 == nk_cur && fni != cni) {
                const unsigned d = (unsigned)(wm * 2048 * (fni - cni));
                rdA0 += d; rdA1 += d; rdA1t += d;
            }
        }
#define TG_B(Q) TG_FRAG(fa0, fb0, rdA0, rdB0, Q); TG_SB; TG_MF(fa1_, fb1_, Q); TG_SB;
        TG_B(0) TG_B(1) TG_B(2) TG_B(3) TG_B(4) TG_B(5) TG_B(6) TG_B(7)
#undef TG_B
        if (tk_pending) {   // the ticket drawn in the previous step has landed (this step's barrier): publish it to the other waves
            asm volatile("" : "+v"(tk));
            if (t == 0) tslot[entered & 1] = tk;
            tk_pending = false;
        }
        if (pending) commit_tile(entered - 1);
        // (behind the last step of the stream the transfers are issued all the same -- the first step of the last tile once more,
        // into a stage nobody reads again: cheaper than eight branches per step; they are waited for before the kernel ends)
        fetch_begin(freed);
        TG_SB;
#define TG_C(Q) TG_DMA(Q); TG_SB; TG_MF(fa1_, fb1_, 8 + (Q)); TG_SB;
        if (!MIXED) {
            TG_C(0) TG_C(1) TG_C(2) TG_C(3) TG_C(4) TG_C(5) TG_C(6) TG_C(7)
            if (!ktl) {
#pragma unroll
                for (int x = 16; x < 32; ++x) TG_MF(fa1_, fb1_, x);
            }
        } else {
            // (the transfers ride in the gaps of the MFMAs every tile issues: two per gap)
            TG_DMA(0); TG_SB; TG_DMA(1); TG_SB; TG_MF(fa1_, fb1_, 8); TG_SB;
            TG_DMA(2); TG_SB; TG_DMA(3); TG_SB; TG_MF(fa1_, fb1_, 9); TG_SB;
            TG_DMA(4); TG_SB; TG_DMA(5); TG_SB; TG_MF(fa1_, fb1_, 10); TG_SB;
            TG_DMA(6); TG_SB; TG_DMA(7); TG_SB; TG_MF(fa1_, fb1_, 11); TG_SB;
            if (ni4) {
#pragma unroll
                for (int x = 12; x < 16; ++x) TG_MF(fa1_, fb1_, x);
            }
            TG_SB;
            if (!ktl) {
#pragma unroll
                for (int x = 16; x < 28; ++x) TG_MF(fa1_, fb1_, x);
                TG_SB;
                if (ni4) {
#pragma unroll
                    for (int x = 28; x < 32; ++x) TG_MF(fa1_, fb1_, x);
                }
            }
        }
#undef TG_C
        TG_SB;
        if (++kt == nk_cur) {
#ifdef TG_STAMPS
            const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
#endif
            if (!(a.dbg & 1)) store_tile(cj % TG_NSLOT, mrem_cur, nrem_cur, c0_cur, cni);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
#ifdef TG_STAMPS
            st_epi += __builtin_amdgcn_s_memtime() - st_e0;
            ++st_tiles;
            if (st_tiles == 8) st_t8 = __builtin_amdgcn_s_memrealtime();   // (100 MHz) when this workgroup finished its 8th tile
#endif
            kt = 0;
            if (++cj == entered) break;   // (the tile after this one, if there is one, was entered steps ago)
            // every group has nk >= 2: the next tile was committed one step ago at the latest, and the one after it
            // (entered in this very step if the next tile has two steps) is not committed before the next step
            nk_cur = fnk;
            nk1_cur = fnk1;
            mrem_cur = fmrem;
            nrem_cur = fnrem;
            c0_cur = fc0;
            if (MIXED) cni = fni;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no transfer may land in LDS that belongs to the next workgroup
#ifdef TG_STAMPS
    if (lane == 0 && blockIdx.x < 512) {
        unsigned long long* d = g_tg_stamp + ((size_t)blockIdx.x * 4 + wave) * 8;
        d[0] = __builtin_amdgcn_s_memtime() - st_t0;
        d[1] = st_bar; d[2] = st_epi; d[3] = st_steps; d[4] = st_tiles; d[5] = st_max;
        d[6] = st_t8;
        d[7] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID[15:0]: wave, simd, pipe, cu, sh, se
    }
#endif
#undef TG_MF
#undef TG_SB
#undef TG_FRAG
#undef TG_DMA
}

__global__ __launch_bounds__(256, 2) void tgemm_kernel(TgArgs a) { tgemm_body<false>(a); }
// the same body under the names of its other users (TgProblem::tag)
__global__ __launch_bounds__(256, 2) void tgemm_ring_kernel(TgArgs a) { tgemm_body<false>(a); }
__global__ __launch_bounds__(256, 2) void tgemm_xform_kernel(TgArgs a) { tgemm_body<false>(a); }
// ... with 96-row tiles where the rows end (TgArgs::bm): the AO->MO transforms (220 rows = 128 + 96 instead of two 128-row tiles)
__global__ __launch_bounds__(256, 2) void tgemm_mixed_kernel(TgArgs a) { tgemm_body<true>(a); }

// floor(x / d) == umulhi(x, tgemm_inverse(d)) for x * d < 2^32 (tile ids inside a group and patch widths are far below);
// 0 stands for d = 1 (no 32-bit factor reproduces x itself)
unsigned tgemm_inverse(int d) { return d <= 1 ? 0u : (unsigned)(((uint64_t)1 << 32) / (unsigned)d + 1); }
// a patch of gm m-tiles x all n-tiles of a group = the ~64 tiles one XCD works on in a round
int tgemm_group_m(int M, int max_ntiles, int bm)
{
    const int mtiles = (M + bm - 1) / bm;
    if (mtiles <= 4) return mtiles;   // few rows: the m-tiles of a column tile side by side (they share its B lines)
    const int patch = knobs().tg_patch;   // tuning knob AFESP_TG_PATCH: tiles per patch
    return std::min(mtiles, std::max(1, (patch + max_ntiles / 2) / std::max(1, max_ntiles)));
}

void preload_tgemm()
{
    first_use_touch(reinterpret_cast<const void*>(tgemm_kernel));
    first_use_touch(reinterpret_cast<const void*>(tgemm_mixed_kernel));
    first_use_touch(reinterpret_cast<const void*>(tgemm_ring_kernel));
    first_use_touch(reinterpret_cast<const void*>(tgemm_xform_kernel));
    (void)hipGetLastError();
}

void tgemm_state_free(TgLaunchState& st)
{
    if (st.tickets) (void)hipFree(st.tickets);
    st.tickets = nullptr;
}

hipError_t tgemm_launch(const TgProblem& p, const TgGroup* dev_groups, int ngroups, int total_tiles, int max_ntiles, hipStream_t stream,
                        TgLaunchState& st, int bm)
{
    if (p.M <= 0 || ngroups <= 0 || total_tiles <= 0) return hipSuccess;
    if (bm != 0 && bm != 96 && bm != TG_BM) return hipErrorInvalidValue;
    const bool mixed = bm != 0;
    int& cap = mixed ? st.cap_mixed : st.cap;
    if (cap == 0) {
        int dev = 0, cus = 256, occ = 2;
        if (hipGetDevice(&dev) == hipSuccess) {
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
            std::lock_guard<std::mutex> lk(first_use_mutex());   // (the occupancy query resolves the function: a first use, first_use.h)
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mixed ? tgemm_mixed_kernel : tgemm_kernel, 256, 0) != hipSuccess || occ <= 0) occ = 2;
        }
        cap = cus * occ;
    }
    TgArgs a;
    a.p = p;
    a.groups = dev_groups;
    a.bm = mixed ? bm : TG_BM;
    a.mtiles = (p.M + a.bm - 1) / a.bm;
    a.total_tiles = total_tiles;
    a.dbg = knobs().tg_dbg;
    a.gm = tgemm_group_m(p.M, max_ntiles, a.bm);
    a.inv_gm = tgemm_inverse(a.gm);
    a.inv_gl = tgemm_inverse(std::max(1, a.mtiles % a.gm));
    const int grid_env = knobs().tg_grid;   // diagnostic AFESP_TG_GRID: fewer workgroups, longer tile streams
    const unsigned grid = (unsigned)std::min(total_tiles, grid_env > 0 ? grid_env : cap);
    // tickets: launches of many rounds of tiles whose grid splits evenly over the XCDs (AFESP_TG_DYNAMIC=0: always static)
    const int prio_env = knobs().tg_prio_shift;   // AFESP_TG_PRIO_SHIFT: 2^11 x 10 ns = 20 us slices
    a.prio_shift = (int64_t)total_tiles >= (int64_t)2 * grid ? prio_env : 0;
    const bool dyn_env = knobs().tg_dynamic != 0;
    const bool dyn_force = knobs().tg_dynamic == 2;   // AFESP_TG_DYNAMIC=2 (tests): also for small launches
    a.tickets = nullptr;
    a.chunk = (int)(grid / 8);
    a.inv_chunk = tgemm_inverse(a.chunk);
    // (tickets for launches of fewer rounds whose tiles differ in length -- the ring products' groups over one and over two runs of
    // K -- were measured in round 5: o = 16 ... 18, 1.6 ... 2.6 tiles per workgroup: 2-4 % slower than dealt statically)
    if (dyn_env && grid % 8 == 0 && ((int64_t)total_tiles >= (int64_t)4 * grid || dyn_force)) {
        if (!st.tickets && hipMalloc((void**)&st.tickets, 64) != hipSuccess) return hipErrorOutOfMemory;
        const hipError_t me = hipMemsetAsync(st.tickets, 0, 64, stream);
        if (me != hipSuccess) return me;
        a.tickets = st.tickets;
    }
    if (mixed) AFESP_KLAUNCH(tgemm_mixed_kernel, dim3(grid), dim3(256), 0, stream, a);
    else if (p.tag == 1) AFESP_KLAUNCH(tgemm_ring_kernel, dim3(grid), dim3(256), 0, stream, a);
    else if (p.tag == 2) AFESP_KLAUNCH(tgemm_xform_kernel, dim3(grid), dim3(256), 0, stream, a);
    else AFESP_KLAUNCH(tgemm_kernel, dim3(grid), dim3(256), 0, stream, a);
    ++(mixed ? st.launches_mixed : st.launches);
    return hipGetLastError();
}

}  // namespace afesp
