// gett.h -- separable-offset ("gather") fp64 GEMM on MFMA for gfx950.
//
//   C[offCm[m] + offCn[n]] = alpha * sum_k A[offAm[m] + offAk[k]] * B[offBk[k] + offBn[n]] + beta * C[...]
//
// Every tensor contraction of the coupled-cluster path is a GEMM whose row, column
// and summation indices are each a *group* of tensor indices.  Because the address of a tensor element is a
// sum of per-index strides, grouping indices keeps the address separable into a row part and a column part.
// The reference permutes tensors in memory (omp_reshape, src/linalg.fpp:99-156) until a dense dgemm
// (src/linalg.fpp:58-89) applies; here the permutation lives in three pairs of small offset tables and the
// operands are gathered straight into LDS, so no permuted copy ever touches HBM.
#pragma once
#include <hip/hip_runtime.h>
#include "first_use.h"
#include <stdint.h>

namespace afesp {

struct GettProblem {
    const double* A;
    const double* B;
    double* C;
    const int64_t* offAm;  // [M]
    const int64_t* offAk;  // [K]
    const int64_t* offBk;  // [K]
    const int64_t* offBn;  // [N]
    const int64_t* offCm;  // [M]
    const int64_t* offCn;  // [N]
    int M, N, K;
    double alpha, beta;
    // batching: grid.z problems, operand bases shifted by batch{A,B,C}[z] elements (device arrays or null)
    int nbatch;
    const int64_t* batchA;
    const int64_t* batchB;
    const int64_t* batchC;
    // layout hints: true = consecutive k are (mostly) consecutive in memory, false = consecutive m (or n) are
    bool a_kcontig, b_kcontig;
    // every operand offset is even and the contiguous direction of both operands advances in unit-stride pairs:
    // 16-byte loads/stores are legal (set by the planner after checking the tables)
    bool wide = false;
    // consecutive m are consecutive elements of A / consecutive n of B (the planner's view of the leading free label; tall.h)
    bool a_munit = false, b_nunit = false;
};

// Grouped launch: several products that share M, K, B, C, the row tables and the K tables, but differ in the A panel, the
// column tables and the column count, walked as ONE persistent tile stream (no ragged last round and no launch gap per
// product).  Each group may have its own K-offset tables as long as their first K step (16 entries) is the same in all
// groups -- the offsets of a tile's first step are fetched before the gather cursor has switched groups.  Device array
// of ngroups + 1 entries; the last one only carries tile_start.
struct GettGroup {
    int64_t a_off;                 // A panel of this group: p.A + a_off
    const int64_t* offAk;
    const int64_t* offBk;
    const int64_t* offBn;
    const int64_t* offCn;
    int N, ntiles, tile_start, pad;
};
// Tile shape the grouped launch uses for (M, wide): codes as in gett_launch; BM/BN in elements.
void gett_grouped_tile(int M, bool wide, int* tm, int* tn, int* BM, int* BN);
hipError_t gett_launch_grouped(const GettProblem& p, const GettGroup* dev_groups, int ngroups, int total_tiles, int max_ntiles,
                               hipStream_t stream);

// Workspace for split-K partial sums.  The launcher picks the tile shape and the split count itself.
struct GettWorkspace {
    double* ptr;
    size_t bytes;
};

// Returns hipSuccess or the launch error.  `force_split` > 0 overrides the heuristic (tests).
hipError_t gett_launch(const GettProblem& p, const GettWorkspace& ws, hipStream_t stream, int force_split = 0,
                       int force_tm = 0, int force_tn = 0);

hipError_t gett_read_stamps(unsigned long long* out, int n);           // diagnostic builds (gett.hip)
hipError_t gett_read_stamps_grouped(unsigned long long* out, int n);   // ... of the grouped kernels (gett_grouped.hip)
void preload_gett_grouped();

extern int g_group_m, g_force_tm, g_force_tn, g_force_split, g_allow_wide, g_dbg;

}  // namespace afesp
