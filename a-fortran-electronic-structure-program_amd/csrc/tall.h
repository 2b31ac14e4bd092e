// tall.h -- streaming fp64 product for "tall x skinny" contractions (one extent of C at most 32, the other in the 10^4 ... 10^6).
//
//   C[offCm[m] + offCn[n]] = alpha * sum_k A[offAm[m] + offAk[k]] * B[offBk[k] + offBn[n]] + beta * C[...]      (GettProblem, gett.h)
//
// with min(M, N) <= 32 and K of a few hundred: every product of the coupled-cluster iteration that pairs t1 (o x v) with a
// four-index array -- t(j,e) <eb|ia>, <be|ia> t(j,e), t(i,e) <ab|ej>, t(i,e) <mj|eb>, ... (src/ccsd.f90:1165-1191, :1275-1290,
// :1700).  They stream the large operand once and are bound by HBM, not by the matrix pipe; gett_kernel's LDS-staged tiles with
// a barrier per 16 k keep too few bytes in flight for that (3.3 - 3.8 TB/s at o = 20, v = 200).  Here a WAVE owns 16 rows of the
// tall index for the whole of K: its operand elements go from memory straight into MFMA fragment registers (lane (t, k) of
// v_mfma_f64_16x16x4_f64 loads X(t0 + t, k0 + k): no LDS, no barrier in the stream, two chunks of up to 16 loads in flight per
// wave), the skinny operand sits in LDS for the lifetime of the workgroup.
// Operands must be finite wherever the offset tables reach: padded k of the last chunk are masked through the zeroed image of the
// skinny operand, the tall operand is read from a clamped (valid) address and multiplied as loaded -- an Inf / NaN there would make
// the masked product a NaN.
#pragma once
#include "gett.h"

namespace afesp {

bool tall_eligible(const GettProblem& p);                          // shape test; batched problems are not taken
hipError_t tall_launch(const GettProblem& p, hipStream_t stream);   // precondition: tall_eligible(p)
// Two products that stream the same tall array against the same skinny matrix, one along each of the array's two leading indices, in ONE
// launch: the array crosses HBM once (tall.hip, tall_dual_kernel).  The caller vouches that both problems enumerate the skinny matrix
// alike (offYk / offYs of both describe the same image); everything else is checked.
bool tall_dual_eligible(const GettProblem& p1, const GettProblem& p2);
hipError_t tall_launch_dual(const GettProblem& p1, const GettProblem& p2, hipStream_t stream);   // precondition: tall_dual_eligible
void preload_tall();

}  // namespace afesp
