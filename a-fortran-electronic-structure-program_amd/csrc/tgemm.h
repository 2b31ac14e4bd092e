// tgemm.h -- grouped fp64 GEMM for operands that are both contiguous along the summation index, staged by LDS-DMA.
//
// The (T) products of src/ccsd.f90:2168-2173 (csrc/triples.hip, plan_fused) are 85 % of a config-5 step.  Both of their
// operands -- vt(kappa; b,c,r) and tt(kappa; a,q,p) -- are contiguous along the summation index, every K step of 16 is one
// 128-byte line per row, and the summation index of a group of columns is at most two contiguous runs (the slab of r, then
// the transposed slab of q).  That is all this kernel supports, and what it buys over the general gather kernel (gett.h):
//   * operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write, no address VALU in the
//     loop: a row is SGPR base + 32-bit VGPR byte offset, the K step moves the SGPR base);
//   * two independent 4-wave workgroups per CU on 128 x 128 tiles instead of one 8-wave workgroup on 256 x 128: the barrier
//     skew, the tile epilogue and the DMA waits of one workgroup lie under the other one's MFMAs;
//   * fragments are read as ds_read_b128 pairs of k (the k order inside a step is remapped identically for A and B).
#pragma once
#include <hip/hip_runtime.h>
#include "first_use.h"
#include <stdint.h>

namespace afesp {

// One group of columns: its A panel (two runs of the summation index), the shift of its B columns in the second run, its
// column tables.  Device array of ngroups + 1 entries; the last one only carries tile_start.
struct TgGroup {
    int64_t a1, a2;          // element offsets (from TgProblem::A) of the A panel in K steps [0, nk1) and [nk1, nk)
    int64_t b1, b2;          // element shifts of every B column in K steps [0, nk1) and [nk1, nk)
    int64_t c0;              // element shift of every C element of the group (slabs of one tensor share their column tables)
    const uint32_t* colB;    // [N] byte offsets of the columns of B (from TgProblem::B)
    const int64_t* offCn;    // [N] element offsets of the columns of C
    int N, ntiles, tile_start, nk1, nk;
    unsigned inv_width;      // tgemm_inverse(tgemm_group_m(M, max_ntiles) * ntiles)
};

struct TgProblem {
    const double* A;
    const double* B;
    double* C;
    const uint32_t* rowA;    // [M] byte offsets of the rows of A inside a panel
    const int64_t* offCm;    // [M] element offsets of the rows of C
    int M;
    // columns 2k and 2k + 1 of every group are adjacent in C (offCn[2k + 1] == offCn[2k] + 1) and offCm[m] + offCn[2k] is even,
    // C 16-byte aligned: whole tiles are then stored 16 bytes per lane
    bool c_pairs = false;
    // groups of four valid k in the last K step of each of a group's two runs (1..4; the rest of that step is zero padding)
    int ktail4 = 4;
    // which entry point of the (one) kernel body launches it: 0 tgemm_kernel (the (T) products), 1 tgemm_ring_kernel (the CCSD
    // iteration's ring products), 2 tgemm_xform_kernel (AO->MO quarter transforms on 128-row tiles) -- names of their own so that a
    // kernel trace's per-kernel averages mean something (a (T) launch is 21 ms, a ring launch 5.6, a transform 1-5)
    int tag = 0;
};

constexpr int TG_BM = 128, TG_BN = 128, TG_BK = 16;

// Every group needs nk >= 2 and 1 <= nk1 <= nk; tables as above.  max_ntiles = the largest ntiles of a group.
// The kernel fetches a tile's tables in 1-KiB pieces: rowA and every colB must be readable up to 255 entries, offCm and every
// offCn up to 127 entries beyond their last element (pad the buffers; what is read there is never used).
// What a launcher keeps between launches: the per-XCD ticket counters (64 bytes of device memory) and the number of workgroups
// the device holds at once.  One per context (Context::tg) -- two contexts that launch at the same time on one device each
// draw from their own counters; launches of ONE state must be ordered on one stream (the counters are reset in stream order).
struct TgLaunchState {
    unsigned* tickets = nullptr;
    int cap = 0, cap_mixed = 0;
    unsigned long long launches = 0, launches_mixed = 0;   // tgemm_kernel / tgemm_mixed_kernel launches through this state (afesp_launch_counts)
};
void tgemm_state_free(TgLaunchState& st);
// bm = 0: 128-row tiles throughout (tgemm_kernel).  bm = 128 / 96: m-tiles of that many rows, and a tile with at most 96 rows left runs
// as a 96-row tile -- three of a wave's four accumulator rows, three quarters of the MFMAs (tgemm_mixed_kernel): M = 220 is 128 + 96
// rows instead of two 128-row tiles; the callers count their tiles with the same bm (and pass it to tgemm_group_m).
hipError_t tgemm_launch(const TgProblem& p, const TgGroup* dev_groups, int ngroups, int total_tiles, int max_ntiles, hipStream_t stream,
                        TgLaunchState& st, int bm = 0);
void preload_tgemm();
unsigned tgemm_inverse(int d);
int tgemm_group_m(int M, int max_ntiles, int bm = TG_BM);

}  // namespace afesp
