// fused.h -- launch-fused evaluation of a small system's contraction sequences (one grouped launch per dependency level).
//
// A CCSD iteration of the systems the reference ships (o = 5...9, v = 19...53: src/ccsd.f90:1040-1312, :1538-1732) is ~45 products of
// 10^5...10^8 flop each: microseconds of matrix-pipe time behind ~75 kernel launches, K-slice reductions and stream hops.  What
// bounds it is the NUMBER of dependent launches, so the sequence is not launched call by call.  It is RECORDED once -- every
// contract() / gett product, permuting copy and elementwise kernel of the unchanged solver code lands in a Recorder with the
// memory it reads and writes -- then levelled by data dependence and compiled into a handful of launches:
//
//   stage 0 (elementwise)  every copy / permuted copy whose inputs are ready, as ONE launch (fused_ew_kernel) + the opaque kernels
//   stage 1 (products)     every product whose operands are ready, all tiles and all K slices of all of them as ONE launch
//                          (fused_gemm_kernel: a wave per (product, tile, K slice); partial sums go to slabs in C's own layout)
//   stage 2 (elementwise)  the slabs of every result of stage 1 summed in a fixed order (C = beta C + sum), same launch as the
//                          copies that have become ready
//   stage 3 (products) ...
//
// Operands must be FINITE everywhere the tables can reach, also beyond a K slice's end: the last block of a slice is masked on the A
// fragment only (`valid ? a : 0`), B is multiplied as loaded from its clamped address -- an Inf / NaN there would turn the masked
// zero into a NaN.  Every tensor of the solver states is (they are zero-initialised and only ever hold finite results); a caller of
// Recorder::product with hand-built tables has to keep that.
// Products that accumulate into one result in the same stage share its slabs: one pass sums them all.  The compiled program
// (descriptor tables in device memory) is replayed every iteration; nothing is decided on the host between launches.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "afesp_internal.h"

namespace afesp {

struct FusedRange {
    const char* lo;
    const char* hi;
};
inline FusedRange frange(const void* p, int64_t ndoubles)
{
    return FusedRange{(const char*)p, (const char*)p + 8 * ndoubles};
}

struct Recorder {
    enum Kind { PRODUCT = 0, ELEMENTWISE = 1, OPAQUE = 2 };
    struct Op {
        int kind = OPAQUE;
        std::vector<FusedRange> reads, writes;
        int stage = -1;
        // PRODUCT
        GettProblem g{};
        int64_t c_span = 0;
        // ELEMENTWISE: out[oo(x)] = beta out[oo(x)] + alpha in[io(x)]
        double* out = nullptr;
        const double* in = nullptr;
        int rank = 0;
        int64_t dim[6] = {0, 0, 0, 0, 0, 0}, so[6] = {0, 0, 0, 0, 0, 0}, si[6] = {0, 0, 0, 0, 0, 0};
        double alpha = 1.0, beta = 0.0;
        // OPAQUE: a kernel of the solver launched as it is (on the context's stream at replay time)
        std::function<void(Context&)> fn;
        int nlaunch = 1;           // kernels that function launches (afesp_ccsd_iteration_launches)
        bool heavy = false;        // a product with a tiled launch of its own: the heavy ones of a stage run side by side on the context's lanes
    };
    std::vector<Op> ops;
    bool failed = false;
    bool uses_scratch = false;    // a recorded call took a cached scratch buffer of the context: the program dies with the scratch epoch
    std::string why;

    // C must be dense over [C, C + c_span) (every offCm[m] + offCn[n] distinct, all of the span covered)
    void product(const GettProblem& g, int64_t a_span, int64_t b_span, int64_t c_span);
    void elementwise(double* out, const double* in, int rank, const int64_t* dim, const int64_t* so, const int64_t* si, double alpha,
                     double beta);
    void opaque(std::vector<FusedRange> reads, std::vector<FusedRange> writes, std::function<void(Context&)> fn, int nlaunch = 1);
    void fail(const std::string& w)
    {
        if (!failed) why = w;
        failed = true;
    }
};

struct FusedProgram;
// Levels and compiles what was recorded; null (and r.why says why) when something in it cannot be fused.
FusedProgram* fused_compile(Context& cx, Recorder& r);
void fused_run(Context& cx, const FusedProgram* p);   // launches only: may be captured into a graph
void fused_free(Context& cx, FusedProgram* p);
int fused_launches(const FusedProgram* p);           // kernels per replay
int64_t fused_epoch(const FusedProgram* p);          // Context::scratch_epoch it was compiled under, -2: refers to no scratch buffer
int64_t fused_plan_epoch(const FusedProgram* p);     // Context::plan_epoch it was compiled under (its items point into the plans' tables)
void preload_fused();

// A program slot of a solver state: record `body` on first use (it runs the ordinary solver code with cx.rec set), replay afterwards.
struct FusedSlot {
    FusedProgram* prog = nullptr;
    bool disabled = false;        // recording failed once: the direct path from then on
    std::string why;
};
template <typename Body>
inline bool fused_exec(Context& cx, FusedSlot& slot, Body body);   // false: not fused, caller runs the body itself
void fused_slot_reset(Context& cx, FusedSlot& slot);
bool fused_enabled(const Context& cx);                             // afesp_ccsd_set_fused, else AFESP_FUSED=0 switches the whole mechanism off

template <typename Body>
inline bool fused_exec(Context& cx, FusedSlot& slot, Body body)
{
    if (slot.disabled || !fused_enabled(cx)) return false;
    if (slot.prog && fused_epoch(slot.prog) != -2 && fused_epoch(slot.prog) != cx.scratch_epoch) fused_slot_reset(cx, slot);
    // (Context::plan_clear has dropped the offset tables the program's items point into -- another solver state of the context was
    // re-initialised after many different systems: compile again)
    if (slot.prog && fused_plan_epoch(slot.prog) != cx.plan_epoch) fused_slot_reset(cx, slot);
    if (!slot.prog) {
        const bool dbg = knobs().fused_debug;
        const auto t0 = std::chrono::steady_clock::now();
        Recorder rec;
        cx.rec = &rec;
        try {
            body();
        } catch (...) {
            cx.rec = nullptr;
            // (the plans made so far stay in the context: their tables must still reach the device)
            try { cx.plan_uploads_issue(); } catch (...) {}
            (void)hipStreamSynchronize(cx.stream);
            cx.plan_uploads_done();
            throw;
        }
        cx.rec = nullptr;
        const auto t1 = std::chrono::steady_clock::now();
        try {
            slot.prog = fused_compile(cx, rec);
        } catch (...) {
            slot.disabled = true;      // (not compiled again every iteration: the direct path from now on)
            slot.why = "fused_compile threw";
            throw;
        }
        if (dbg)
            fprintf(stderr, "afesp fused: recorded in %.2f ms, compiled in %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
        if (!slot.prog) {
            slot.disabled = true;
            slot.why = rec.why;
            if (knobs().fused_debug) fprintf(stderr, "afesp: not fused: %s\n", rec.why.c_str());
            return false;
        }
    }
    fused_run(cx, slot.prog);
    return true;
}

}  // namespace afesp
