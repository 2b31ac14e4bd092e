// fused.hip -- the recorder, the levelling compiler and the two kernels of the launch-fused small-system path.  See fused.h.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "fused.h"

namespace afesp {

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ device descriptors
struct FItem {                      // one wave's work: a tile of a product over a range of the summation index -- everything the
    const double* A;                // wave needs in ONE 128-byte scalar read (a second, dependent read of a product descriptor is
    const double* B;                // a memory round trip, and round trips are what a small product costs)
    double* P;                      // this K slice's slab, in C's own layout
    const int64_t *offAm, *offBn, *offCm, *offCn;
    const uint32_t* kofs;           // (A, B) byte offsets of the summation index, interleaved, padded past K (build_kofs_kernel)
    int M, N, K, m0, n0, k0, k1, tile;
    double alpha;
    int64_t pad[3];
};
static_assert(sizeof(FItem) == 128, "FItem is read as two s_load_dwordx16");
struct EwOp {                       // out[oo(x)] = beta out[oo(x)] + alpha sum_{s < nsum} in[io(x) + s sstride]
    double* out;
    const double* in;
    int rank, nsum;
    int gshift, pad;                // 2^gshift threads share an element's sum (many slabs, few elements)
    int64_t dim[6], so[6], si[6];
    int64_t n, sstride;
    double alpha, beta;
};
struct EwBlk {
    int op, pad;
    int64_t x0;
};
constexpr int EW_PER_BLOCK = 512;   // thread-elements a workgroup of 256 threads handles (two passes)
constexpr int KOFS_PAD = 256;       // entries behind K that repeat the last offset: the pipeline reads ahead without clamping

// the planner's two K tables (element offsets, int64) as one table of 32-bit byte offsets, (A, B) interleaved: all products of a
// program in one launch, blockIdx.y = product (one launch per product was 0.1 ms of host time for the 43 of an iteration)
struct KofsJob {
    uint32_t* dst;
    const int64_t *offAk, *offBk;
    int K, Kpad;
};
__global__ __launch_bounds__(256) void build_kofs_kernel(const KofsJob* __restrict__ jobs)
{
    const KofsJob j = jobs[blockIdx.y];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < j.Kpad; k += gridDim.x * blockDim.x) {
        const int kk = min(k, j.K - 1);
        j.dst[2 * k] = (uint32_t)(j.offAk[kk] << 3);
        j.dst[2 * k + 1] = (uint32_t)(j.offBk[kk] << 3);
    }
}

// ------------------------------------------------------------------ product kernel
// A wave owns (16 TMF) x (16 TNF) results over its K range; operands go straight from memory into the MFMA fragments -- no LDS, no
// barrier.  For products of this size a memory round trip (~1-2 us) is what counts, so a wave keeps NB - 1 blocks of sixteen k
// of operand loads in flight and the K offsets another NB - 1 blocks ahead of those: block j's MFMAs run while the data of blocks
// j+1 ... j+NB-1 and the offsets of blocks j+NB ... j+2NB-2 are on their way (the hardware counts 63 loads per wave at most:
// NB = 3 for the four-fragment tiles).  Inside a block lane (x = l & 15, q = l >> 4) takes k = 4 q + u for MFMA u -- any
// assignment of a block's sixteen k to (MFMA, lane group) is a valid order of summation -- so its four K offsets are ONE 32-byte
// read of the interleaved table, and an operand that is contiguous along k is read in whole 128-byte lines.  All loads are
// unconditional (hipcc drains every load in flight at the join of a branch that contains one): blocks past the end read the
// table's padding and multiply zeros.
template <int TMF, int TNF, int NB>
__device__ __forceinline__ void fused_tile(const FItem& it, const int lane)
{
    constexpr int U = 4;
    typedef const char __attribute__((address_space(1)))* gcp;
    typedef const uint32_t __attribute__((address_space(1)))* gup;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef const u4 __attribute__((address_space(1)))* gu4p;
    const gcp A = (gcp)it.A;
    const gcp B = (gcp)it.B;
    const int lm = lane & 15, lk = lane >> 4;
    uint32_t aoff[TMF], boff[TNF], cn[TNF], cm[TMF][4];
#pragma unroll
    for (int i = 0; i < TMF; ++i) aoff[i] = ((gup)it.offAm)[2 * min(it.m0 + 16 * i + lm, it.M - 1)] << 3;
#pragma unroll
    for (int j = 0; j < TNF; ++j) boff[j] = ((gup)it.offBn)[2 * min(it.n0 + 16 * j + lm, it.N - 1)] << 3;
    const int kend = it.k1, k0 = it.k0;
    const int nblk = (kend - k0 + 4 * U - 1) / (4 * U);
    const gu4p kofs = (gu4p)(it.kofs + 2 * (k0 + 4 * lk));   // this lane's four (A, B) pairs of block 0; a block is 32 entries on
    u4 ko[NB][2];                                              // [buffer][(ka0 kb0 ka1 kb1), (ka2 kb2 ka3 kb3)]
    double a[NB][U][TMF], b[NB][U][TNF];
    auto load_off = [&](int j, int buf) {
        ko[buf][0] = kofs[8 * j];
        ko[buf][1] = kofs[8 * j + 1];
    };
    auto load_dat = [&](int j, int buf) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t ka = ko[buf][u >> 1][2 * (u & 1)], kb = ko[buf][u >> 1][2 * (u & 1) + 1];
#if defined(AFESP_FUSED_DIAG) && (AFESP_FUSED_DIAG & 2)   // diagnostic build: no operand loads
#pragma unroll
            for (int i = 0; i < TMF; ++i) a[buf][u][i] = 1e-300 * (double)(ka + aoff[i]);   // (tiny: the iteration stays finite)
#pragma unroll
            for (int jj = 0; jj < TNF; ++jj) b[buf][u][jj] = 1e-300 * (double)(kb + boff[jj]);
#else
#pragma unroll
            for (int i = 0; i < TMF; ++i) a[buf][u][i] = *(const double __attribute__((address_space(1)))*)(A + (aoff[i] + ka));
#pragma unroll
            for (int jj = 0; jj < TNF; ++jj) b[buf][u][jj] = *(const double __attribute__((address_space(1)))*)(B + (kb + boff[jj]));
#endif
        }
    };
    // prologue: offsets of blocks 0 ... NB-2, then (one round trip later) their data and the offsets of the next NB-1 blocks; the
    // C offsets ride along with the first batch so that the stores at the end wait for nothing
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) load_off(j, j);
#pragma unroll
    for (int j = 0; j < TNF; ++j) cn[j] = ((gup)it.offCn)[2 * min(it.n0 + 16 * j + lm, it.N - 1)] << 3;
#pragma unroll
    for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) cm[i][r] = ((gup)it.offCm)[2 * min(it.m0 + 16 * i + lk + 4 * r, it.M - 1)] << 3;
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) load_dat(j, j);
#pragma unroll
    for (int j = NB - 1; j < 2 * NB - 2; ++j) load_off(j, j % NB);
    v4d acc[TMF][TNF];
#pragma unroll
    for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int j = 0; j < TNF; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int jb = 0; jb < nblk; jb += NB) {
#pragma unroll
        for (int d = 0; d < NB; ++d) {
            const int j = jb + d;
            load_dat(j + NB - 1, (d + NB - 1) % NB);
            load_off(j + 2 * NB - 2, (d + NB - 2) % NB);
            __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise gathers all loads of the unrolled round in front of all its MFMAs)
            // (the k beyond the item's range -- the tail of a product's last slice -- are zeroed here, where the values are consumed:
            // a select next to the load would wait for the load)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool valid = k0 + 4 * U * j + 4 * lk + u < kend;
#pragma unroll
                for (int i = 0; i < TMF; ++i) {
                    const double av = valid ? a[d][u][i] : 0.0;
#pragma unroll
                    for (int jj = 0; jj < TNF; ++jj) {
#if defined(AFESP_FUSED_DIAG) && (AFESP_FUSED_DIAG & 1)   // diagnostic build: no MFMAs (the loads stay alive through one add)
                        acc[i][jj][0] += av + b[d][u][jj];
#else
                        acc[i][jj] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[d][u][jj], acc[i][jj], 0, 0, 0);
#endif
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 r
    char* __restrict__ P = (char*)it.P;
#pragma unroll
    for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = it.m0 + 16 * i + lk + 4 * r;
#pragma unroll
            for (int j = 0; j < TNF; ++j) {
                const int col = it.n0 + 16 * j + lm;
                if (row < it.M && col < it.N) *(double*)(P + (cm[i][r] + cn[j])) = it.alpha * acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256, 2) void fused_gemm_kernel(const FItem* __restrict__ items, int nitems)
{
    const int w = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (w >= nitems) return;
    const FItem it = items[w];
    const int lane = threadIdx.x & 63;
    switch (it.tile) {
        case 0: fused_tile<1, 1, 4>(it, lane); break;
        case 1: fused_tile<2, 2, 3>(it, lane); break;
        case 2: fused_tile<4, 1, 3>(it, lane); break;
        case 3: fused_tile<1, 4, 3>(it, lane); break;
        case 4: fused_tile<1, 1, 2>(it, lane); break;
        case 5: fused_tile<2, 2, 2>(it, lane); break;
        case 6: fused_tile<4, 1, 2>(it, lane); break;
        default: fused_tile<1, 4, 2>(it, lane); break;
    }
}

// ------------------------------------------------------------------ elementwise kernel
__global__ __launch_bounds__(256) void fused_ew_kernel(const EwOp* __restrict__ ops, const EwBlk* __restrict__ blks)
{
    const EwBlk b = blks[blockIdx.x];
    const EwOp& op = ops[b.op];
    const int rank = op.rank, nsum = op.nsum, gshift = op.gshift, G = 1 << gshift;
    const int64_t n = op.n, sstride = op.sstride;
    const double alpha = op.alpha, beta = op.beta;
    double* __restrict__ out = op.out;
    const double* __restrict__ in = op.in;
#pragma unroll
    for (int t = 0; t < EW_PER_BLOCK / 256; ++t) {
        const int tid = t * 256 + threadIdx.x;
        const int64_t x = b.x0 + (tid >> gshift);
        const int g = tid & (G - 1);
        const bool live = x < n;          // (whole groups of G lanes are live or not: the shuffles below stay inside a group)
        int64_t oo = x, io = x;
        if (rank > 1 && live) {
            int64_t r = x;
            oo = io = 0;
            for (int q = 0; q < rank; ++q) {
                const int64_t i = r % op.dim[q];
                r /= op.dim[q];
                oo += i * op.so[q];
                io += i * op.si[q];
            }
        }
        // slabs g, g + G, g + 2G, ... in this order, four loads in flight; then the G partial sums in a fixed tree: the same bits
        // on every run
        double s = 0.0;
        if (live) {
            int q = g;
            for (; q + 3 * G < nsum; q += 4 * G) {
                const double l0 = in[io + (int64_t)q * sstride], l1 = in[io + (int64_t)(q + G) * sstride],
                             l2 = in[io + (int64_t)(q + 2 * G) * sstride], l3 = in[io + (int64_t)(q + 3 * G) * sstride];
                s += l0; s += l1; s += l2; s += l3;
            }
            for (; q < nsum; q += G) s += in[io + (int64_t)q * sstride];
        }
        for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (live && g == 0) {
            double val = alpha * s;
            if (beta != 0.0) val += beta * out[oo];
            out[oo] = val;
        }
    }
}

void preload_fused()
{
    first_use_touch(reinterpret_cast<const void*>(fused_gemm_kernel));
    first_use_touch(reinterpret_cast<const void*>(fused_ew_kernel));
    first_use_touch(reinterpret_cast<const void*>(build_kofs_kernel));
    (void)hipGetLastError();
}

// ------------------------------------------------------------------ recorder
void Recorder::product(const GettProblem& g, int64_t a_span, int64_t b_span, int64_t c_span)
{
    if (g.nbatch != 1 || g.batchA || g.batchB || g.batchC) return fail("batched product");
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return fail("empty product");
    // A product with real work in it keeps the tiled kernel of its own (gett.h: LDS-staged 16-byte gathers, a launch of tens of
    // microseconds anyway); it takes its place in the levelled sequence as a kernel launched as it is.
    const double big = knobs().fused_big_flop;
    if (2.0 * g.M * (double)g.N * g.K >= big || a_span >= ((int64_t)1 << 28) || b_span >= ((int64_t)1 << 28) || c_span >= ((int64_t)1 << 28)) {
        std::vector<FusedRange> rd = {frange(g.A, a_span), frange(g.B, b_span)};
        if (g.beta != 0.0) rd.push_back(frange(g.C, c_span));
        const GettProblem gp = g;
        opaque(rd, {frange(g.C, c_span)}, [gp](Context& c) { AFESP_HIP(gett_launch(gp, c.ws, c.stream)); }, 2);
        ops.back().heavy = true;
        return;
    }
    if ((int64_t)g.M * g.N != c_span) return fail("product into a strided view");
    Op o;
    o.kind = PRODUCT;
    o.g = g;
    o.c_span = c_span;
    o.reads = {frange(g.A, a_span), frange(g.B, b_span)};
    if (g.beta != 0.0) o.reads.push_back(frange(g.C, c_span));
    o.writes = {frange(g.C, c_span)};
    ops.push_back(std::move(o));
}

void Recorder::elementwise(double* out, const double* in, int rank, const int64_t* dim, const int64_t* so, const int64_t* si, double alpha,
                           double beta)
{
    if (rank > 6) return fail("elementwise rank > 6");
    Op o;
    o.kind = ELEMENTWISE;
    o.out = out;
    o.in = in;
    o.alpha = alpha;
    o.beta = beta;
    // contiguous runs merge (a plain copy becomes rank 1)
    int64_t ospan = 1, ispan = 1;
    for (int q = 0; q < rank; ++q) {
        if (dim[q] == 1) continue;
        ospan += (dim[q] - 1) * so[q];
        ispan += (dim[q] - 1) * si[q];
        if (o.rank > 0 && so[q] == o.so[o.rank - 1] * o.dim[o.rank - 1] && si[q] == o.si[o.rank - 1] * o.dim[o.rank - 1]) {
            o.dim[o.rank - 1] *= dim[q];
        } else {
            o.dim[o.rank] = dim[q]; o.so[o.rank] = so[q]; o.si[o.rank] = si[q];
            ++o.rank;
        }
    }
    if (o.rank == 0) { o.rank = 1; o.dim[0] = 1; o.so[0] = o.si[0] = 1; }
    if (o.rank == 1 && (o.so[0] != 1 || o.si[0] != 1)) { o.dim[1] = 1; o.so[1] = o.si[1] = 0; o.rank = 2; }   // (the kernel's rank-1 case is the unit-stride one)
    o.reads = {frange(in, ispan)};
    if (beta != 0.0) o.reads.push_back(frange(out, ospan));
    o.writes = {frange(out, ospan)};
    ops.push_back(std::move(o));
}

void Recorder::opaque(std::vector<FusedRange> reads, std::vector<FusedRange> writes, std::function<void(Context&)> fn, int nlaunch)
{
    Op o;
    o.kind = OPAQUE;
    o.reads = std::move(reads);
    o.writes = std::move(writes);
    o.fn = std::move(fn);
    o.nlaunch = nlaunch;
    ops.push_back(std::move(o));
}

// ------------------------------------------------------------------ program
struct FusedProgram {
    struct Stage {
        int kind = 0;   // 0 elementwise (+ opaque kernels), 1 products
        const FItem* items = nullptr;
        int nitems = 0;
        const EwOp* ewops = nullptr;
        const EwBlk* blks = nullptr;
        int nblk = 0;
        std::vector<std::function<void(Context&)>> opaque;
        std::vector<std::function<void(Context&)>> heavy;   // opaque products with a tiled launch of their own (independent inside the stage)
        std::vector<std::pair<int, int>> per_op;   // diagnostic (AFESP_FUSED_PER_OP=1): (first item, count) of each product, launched alone
    };
    std::vector<Stage> stages;
    void* desc = nullptr;
    double* slabs = nullptr;
    uint32_t* ktab = nullptr;
    int64_t epoch = 0;
    int64_t plan_epoch = 0;      // Context::plan_epoch it was compiled under: its items point into the plans' offset tables
    int launches = 0;
};

namespace {

bool overlap(const std::vector<FusedRange>& a, const std::vector<FusedRange>& b)
{
    for (auto& x : a)
        for (auto& y : b)
            if (x.lo < y.hi && y.lo < x.hi) return true;
    return false;
}

struct TileKind {
    int code, tm, tn;   // wave tile = 16 tm x 16 tn
};
TileKind pick_tile(int M, int N)
{
    if (M <= 16 && N <= 16) return {0, 1, 1};
    if (N <= 16) return {2, 4, 1};
    if (M <= 16) return {3, 1, 4};
    return {1, 2, 2};
}

}  // namespace

bool fused_enabled(const Context& cx)
{
    const bool on = knobs().fused;
    return cx.fused_mode >= 0 ? cx.fused_mode == 1 : on;
}

FusedProgram* fused_compile(Context& cx, Recorder& r)
{
    // The plan tables built while recording are still on their way to the device (cx.pending_host holds their host images): they
    // share the ONE wait of this function -- the one behind the upload of the program's own descriptors -- or, on a way out
    // without a program, the guard's.
    struct PendingGuard {
        Context& cx;
        bool synced = false;
        ~PendingGuard()
        {
            if (cx.pending_host.empty()) return;
            if (!synced) (void)hipStreamSynchronize(cx.stream);
            cx.plan_uploads_done();
        }
    } pending{cx};
    cx.plan_uploads_issue();   // (whatever becomes of the program: the plans stay in the context and need their tables)
    if (r.failed) return nullptr;
    auto& ops = r.ops;
    const bool debug = knobs().fused_debug;
    // ---- levels: products on odd stages (their sums are complete one stage later), everything else on even ones
    int nstage = 0;
    for (size_t x = 0; x < ops.size(); ++x) {
        Recorder::Op& X = ops[x];
        int smin = 0;
        for (size_t y = 0; y < x; ++y) {
            const Recorder::Op& Y = ops[y];
            if (!(overlap(X.reads, Y.writes) || overlap(X.writes, Y.writes) || overlap(X.writes, Y.reads))) continue;
            const int ydone = Y.kind == Recorder::PRODUCT ? Y.stage + 1 : Y.stage;
            // a product that adds itself (beta = 1) to the result of an earlier product may share its stage and its sum, as long
            // as it does not read that result as an operand; and two products that both only ADD themselves to one result
            // commute -- neither waits for the other (whoever reads the result waits for both)
            const bool joins = X.kind == Recorder::PRODUCT && Y.kind == Recorder::PRODUCT && X.g.C == Y.g.C && X.c_span == Y.c_span &&
                               X.g.beta == 1.0 && !overlap({X.reads[0], X.reads[1]}, Y.writes) &&
                               !overlap({Y.reads[0], Y.reads[1]}, X.writes);
            if (joins && Y.g.beta == 1.0) continue;
            smin = std::max(smin, joins ? Y.stage : ydone + 1);
        }
        const int parity = X.kind == Recorder::PRODUCT ? 1 : 0;
        if ((smin & 1) != parity) ++smin;
        X.stage = smin;
        nstage = std::max(nstage, smin + (parity ? 2 : 1));
    }
    // ---- groups of products that sum into one result in one stage
    struct Group {
        double* C;
        int stage;
        int64_t span, pstride;
        double beta;
        std::vector<size_t> members;
        int nslab = 0;
        int64_t slab0 = 0;   // offset (doubles) of the first slab
    };
    std::vector<Group> groups;
    std::map<std::pair<const void*, int>, size_t> gidx;
    for (size_t x = 0; x < ops.size(); ++x) {
        if (ops[x].kind != Recorder::PRODUCT) continue;
        auto key = std::make_pair((const void*)ops[x].g.C, ops[x].stage);
        auto it = gidx.find(key);
        if (it == gidx.end()) {
            Group g;
            g.C = ops[x].g.C; g.stage = ops[x].stage; g.span = ops[x].c_span; g.pstride = (ops[x].c_span + 1) & ~(int64_t)1;
            g.beta = ops[x].g.beta;
            gidx[key] = groups.size();
            groups.push_back(g);
            it = gidx.find(key);
        } else if (ops[x].g.beta != 1.0 || groups[it->second].span != ops[x].c_span) {
            r.fail("two products overwrite one result in one stage");
            return nullptr;
        }
        groups[it->second].members.push_back(x);
    }
    // ---- tiles and K slices, stage by stage.  An item is a few microseconds of one wave: at most 32 K steps of a four-MFMA tile
    // (128 of a single-MFMA one), fewer when the stage has little work for the ~2000 waves the device holds.
    int ncu = 256;
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cx.device) == hipSuccess && v > 0) ncu = v;
    }
    // (tuning knobs AFESP_FUSED_ITEMS / _MIN_STEPS / _MAX_MFMA / _NB, knobs.h)
    const int64_t target_items = knobs().fused_items > 0 ? knobs().fused_items : (int64_t)ncu * 8;
    const int64_t min_steps = knobs().fused_min_steps, max_mfma = knobs().fused_max_mfma;
    const int nb_code = knobs().fused_nb == 2 ? 4 : 0;
    const bool per_op = knobs().fused_per_op;
    std::vector<int> slices(ops.size(), 0), tilecode(ops.size(), 0);
    std::vector<int64_t> slicelen(ops.size(), 0);
    std::vector<std::vector<size_t>> stage_ops((size_t)nstage);   // product ops of a stage
    for (int s = 1; s < nstage; s += 2) {
        double work = 0.0;   // MFMAs over all tiles
        for (size_t x = 0; x < ops.size(); ++x) {
            if (ops[x].kind != Recorder::PRODUCT || ops[x].stage != s) continue;
            const TileKind t = pick_tile(ops[x].g.M, ops[x].g.N);
            // (an item runs whole rounds of NB blocks of four K steps: the two-deep variants for products with a short summation index)
            tilecode[x] = t.code + ((ops[x].g.K + 15) / 16 < 3 ? 4 : nb_code);
            const int64_t tiles = (int64_t)((ops[x].g.M + 16 * t.tm - 1) / (16 * t.tm)) * ((ops[x].g.N + 16 * t.tn - 1) / (16 * t.tn));
            work += (double)tiles * ((ops[x].g.K + 3) / 4) * (t.tm * t.tn);
            stage_ops[(size_t)s].push_back(x);
        }
        const int64_t budget = std::min<int64_t>(max_mfma, (int64_t)(work / (double)target_items + 0.999));   // MFMAs per item
        for (size_t x : stage_ops[(size_t)s]) {
            const GettProblem& g = ops[x].g;
            const TileKind t = pick_tile(g.M, g.N);
            const int64_t ksteps = (g.K + 3) / 4;
            const int64_t round = 4 * (tilecode[x] >= 4 ? 2 : (t.code >= 1 ? 3 : 4));   // K steps of a pipeline round of this tile's variant
            int64_t len = std::max<int64_t>(min_steps, budget / (t.tm * t.tn));
            len = (len + round - 1) / round * round;
            int64_t ns = (ksteps + len - 1) / len;
            if (ns > 512) {
                ns = 512;
                len = ((ksteps + ns - 1) / ns + round - 1) / round * round;
                ns = (ksteps + len - 1) / len;
            }
            slices[x] = (int)ns;
            slicelen[x] = len;
        }
    }
    // ---- slabs
    int64_t slab_total = 0;
    std::vector<int> slab_of(ops.size(), 0);   // first slab of a product inside its group
    for (Group& g : groups) {
        g.slab0 = slab_total;
        for (size_t x : g.members) {
            slab_of[x] = g.nslab;
            g.nslab += slices[x];
        }
        slab_total += g.pstride * g.nslab;
    }
    if (slab_total > ((int64_t)1 << 28)) {   // 2 GiB of partial sums: not a small system
        r.fail("partial-sum slabs too large");
        return nullptr;
    }
    // (anything below may throw -- an allocation, a copy, the wait: the program and its blocks then go back, and fused_exec marks the slot)
    struct Holder {
        Context& cx;
        FusedProgram* p;
        ~Holder() { if (p) fused_free(cx, p); }
    } hold{cx, new FusedProgram()};
    FusedProgram* P = hold.p;
    P->epoch = r.uses_scratch ? cx.scratch_epoch : -2;
    P->plan_epoch = cx.plan_epoch;
    // the K-offset tables of the products in the form the kernel reads (32-bit byte offsets, interleaved, padded)
    std::vector<int64_t> ktab_at(ops.size(), 0);
    int64_t ktab_total = 0;
    for (size_t x = 0; x < ops.size(); ++x) {
        if (ops[x].kind != Recorder::PRODUCT) continue;
        ktab_at[x] = ktab_total;
        ktab_total += 2 * ((int64_t)ops[x].g.K + KOFS_PAD);
        ktab_total = (ktab_total + 7) & ~(int64_t)7;   // every table starts on a 32-byte boundary
    }
    // (slabs and tables in one block: every first allocation of a context is a driver call)
    const int64_t slab_len = (std::max<int64_t>(slab_total, 1) + 3) & ~(int64_t)3;
    P->slabs = cx.alloc_raw(slab_len + ktab_total / 2 + 4);
    P->ktab = (uint32_t*)(P->slabs + slab_len);
    std::vector<KofsJob> kjobs;
    int kmax = 1;
    for (size_t x = 0; x < ops.size(); ++x) {
        if (ops[x].kind != Recorder::PRODUCT) continue;
        const int Kpad = ops[x].g.K + KOFS_PAD;
        kjobs.push_back(KofsJob{P->ktab + ktab_at[x], ops[x].g.offAk, ops[x].g.offBk, ops[x].g.K, Kpad});
        kmax = std::max(kmax, Kpad);
    }
    // ---- descriptors: one host image, one upload
    std::vector<char> img;
    auto put = [&](const void* p, size_t bytes) {
        const size_t at = (img.size() + 127) & ~(size_t)127;
        img.resize(at + bytes);
        memcpy(img.data() + at, p, bytes);
        return at;
    };
    struct StageAt {
        size_t items = 0, ewops = 0, blks = 0;
    };
    std::vector<StageAt> at((size_t)nstage);
    P->stages.resize((size_t)nstage);
    for (int s = 0; s < nstage; ++s) {
        FusedProgram::Stage& st = P->stages[(size_t)s];
        st.kind = s & 1;
        if (st.kind == 1) {
            std::vector<FItem> items;
            for (size_t x : stage_ops[(size_t)s]) {
                const GettProblem& g = ops[x].g;
                const Group& grp = groups[gidx[std::make_pair((const void*)g.C, s)]];
                const TileKind t = pick_tile(g.M, g.N);
                const size_t first = items.size();
                for (int m0 = 0; m0 < g.M; m0 += 16 * t.tm)
                    for (int n0 = 0; n0 < g.N; n0 += 16 * t.tn)
                        for (int q = 0; q < slices[x]; ++q) {
                            FItem it;
                            it.A = g.A; it.B = g.B;
                            it.P = P->slabs + grp.slab0 + grp.pstride * (slab_of[x] + q);
                            it.offAm = g.offAm; it.offBn = g.offBn; it.offCm = g.offCm; it.offCn = g.offCn;
                            it.kofs = P->ktab + ktab_at[x];
                            it.M = g.M; it.N = g.N; it.K = g.K; it.m0 = m0; it.n0 = n0;
                            it.k0 = (int)(4 * slicelen[x] * q); it.k1 = (int)std::min<int64_t>(g.K, 4 * slicelen[x] * (q + 1));
                            it.tile = tilecode[x];
                            it.alpha = g.alpha;
                            it.pad[0] = it.pad[1] = it.pad[2] = 0;
                            items.push_back(it);
                        }
                if (per_op) st.per_op.push_back({(int)first, (int)(items.size() - first)});
            }
            // the longest items first: the tail of the launch is made of short ones
            if (!per_op)
                std::stable_sort(items.begin(), items.end(), [](const FItem& a, const FItem& b) {
                    auto cost = [](const FItem& i) { return (int64_t)(i.k1 - i.k0) * (((i.tile & 3) == 0) ? 1 : 4); };
                    return cost(a) > cost(b);
                });
            st.nitems = (int)items.size();
            if (st.nitems) {
                at[(size_t)s].items = put(items.data(), items.size() * sizeof(FItem));
                ++P->launches;
            }
        } else {
            std::vector<EwOp> eops;
            for (const Group& g : groups) {   // the sums of the products of the stage before
                if (g.stage + 1 != s) continue;
                EwOp e{};
                e.out = g.C; e.in = P->slabs + g.slab0;
                e.rank = 1; e.nsum = g.nslab;
                while (e.gshift < 6 && (g.nslab >> e.gshift) > 8) ++e.gshift;
                e.dim[0] = g.span; e.so[0] = e.si[0] = 1;
                e.n = g.span; e.sstride = g.pstride;
                e.alpha = 1.0; e.beta = g.beta;
                eops.push_back(e);
            }
            for (size_t x = 0; x < ops.size(); ++x) {
                const Recorder::Op& o = ops[x];
                if (o.stage != s) continue;
                if (o.kind == Recorder::OPAQUE) {
                    (o.heavy ? st.heavy : st.opaque).push_back(o.fn);
                    P->launches += o.nlaunch;
                } else if (o.kind == Recorder::ELEMENTWISE) {
                    EwOp e{};
                    e.out = o.out; e.in = o.in;
                    e.rank = o.rank; e.nsum = 1;
                    e.n = 1;
                    for (int q = 0; q < o.rank; ++q) {
                        e.dim[q] = o.dim[q]; e.so[q] = o.so[q]; e.si[q] = o.si[q];
                        e.n *= o.dim[q];
                    }
                    e.sstride = 0;
                    e.alpha = o.alpha; e.beta = o.beta;
                    eops.push_back(e);
                }
            }
            std::vector<EwBlk> blks;
            for (size_t q = 0; q < eops.size(); ++q) {
                const int64_t per = EW_PER_BLOCK >> eops[q].gshift;
                for (int64_t x0 = 0; x0 < eops[q].n; x0 += per) blks.push_back(EwBlk{(int)q, 0, x0});
            }
            st.nblk = (int)blks.size();
            if (st.nblk) {
                at[(size_t)s].ewops = put(eops.data(), eops.size() * sizeof(EwOp));
                at[(size_t)s].blks = put(blks.data(), blks.size() * sizeof(EwBlk));
                ++P->launches;
            }
        }
    }
    const size_t kjobs_at = kjobs.empty() ? 0 : put(kjobs.data(), kjobs.size() * sizeof(KofsJob));
    P->desc = cx.alloc_raw((int64_t)(img.size() / 8 + 2));
    AFESP_HIP(hipMemcpyAsync(P->desc, img.data(), img.size(), hipMemcpyHostToDevice, cx.stream));
    if (!kjobs.empty()) {
        AFESP_KLAUNCH(build_kofs_kernel, dim3((unsigned)std::min(16, (kmax + 255) / 256), (unsigned)kjobs.size()), dim3(256), 0, cx.stream,
                           (const KofsJob*)((const char*)P->desc + kjobs_at));
        AFESP_HIP(hipGetLastError());
    }
    AFESP_HIP(hipStreamSynchronize(cx.stream));   // img is a temporary
    pending.synced = true;
    for (int s = 0; s < nstage; ++s) {
        FusedProgram::Stage& st = P->stages[(size_t)s];
        const char* base = (const char*)P->desc;
        if (st.kind == 1 && st.nitems) {
            st.items = (const FItem*)(base + at[(size_t)s].items);
        } else if (st.kind == 0 && st.nblk) {
            st.ewops = (const EwOp*)(base + at[(size_t)s].ewops);
            st.blks = (const EwBlk*)(base + at[(size_t)s].blks);
        }
    }
    if (debug) {
        fprintf(stderr, "afesp fused program: %zu recorded calls -> %d stages, %d launches, %.1f MB of slabs\n", ops.size(), nstage, P->launches,
                slab_total * 8e-6);
        for (int s = 0; s < nstage; ++s) {
            const FusedProgram::Stage& st = P->stages[(size_t)s];
            if (st.kind == 1) {
                fprintf(stderr, "  stage %d: %zu products, %d wave items\n", s, stage_ops[(size_t)s].size(), st.nitems);
                for (size_t x : stage_ops[(size_t)s])
                    fprintf(stderr, "      M %6d N %6d K %6d  tile %d  slices %2d  alpha %+.2f beta %.0f -> %p\n", ops[x].g.M, ops[x].g.N, ops[x].g.K,
                            tilecode[x], slices[x], ops[x].g.alpha, ops[x].g.beta, (void*)ops[x].g.C);
            } else {
                fprintf(stderr, "  stage %d: %d elementwise blocks, %zu opaque kernels, %zu products on their own tiled launch\n", s, st.nblk, st.opaque.size(),
                        st.heavy.size());
            }
        }
    }
    hold.p = nullptr;
    return P;
}

void fused_run(Context& cx, const FusedProgram* P)
{
    for (const FusedProgram::Stage& st : P->stages) {
        if (st.kind == 1) {
            if (!st.nitems) continue;
            if (!st.per_op.empty()) {
                for (auto& r : st.per_op) {
                    AFESP_KLAUNCH(fused_gemm_kernel, dim3((unsigned)((r.second + 3) / 4)), dim3(256), 0, cx.stream, st.items + r.first, r.second);
                    AFESP_HIP(hipGetLastError());
                }
                continue;
            }
            AFESP_KLAUNCH(fused_gemm_kernel, dim3((unsigned)((st.nitems + 3) / 4)), dim3(256), 0, cx.stream, st.items, st.nitems);
            AFESP_HIP(hipGetLastError());
        } else {
            if (st.nblk) {
                AFESP_KLAUNCH(fused_ew_kernel, dim3((unsigned)st.nblk), dim3(256), 0, cx.stream, st.ewops, st.blks);
                AFESP_HIP(hipGetLastError());
            }
            for (auto& fn : st.opaque) fn(cx);
            // The products that keep a tiled launch of their own are independent inside a stage, and at the sizes where they occur
            // (o ~ 10, v ~ 100) none of them fills the device: they run side by side on the context's lanes (each lane with a
            // split-K workspace of its own), the main stream waits for all of them.  AFESP_FUSED_LANES=0: one after the other.
            const bool lanes_on = knobs().fused_lanes;
            if (st.heavy.size() >= 2 && lanes_on) {
                const int nl = (int)std::min<size_t>(st.heavy.size(), 4);
                cx.lane_ws_bytes = std::max(cx.lane_ws_bytes, (size_t)64 << 20);
                cx.fork(nl);
                for (size_t i = 0; i < st.heavy.size(); ++i) {
                    cx.use_lane((int)(i % (size_t)nl));
                    st.heavy[i](cx);
                }
                cx.join();
            } else {
                for (auto& fn : st.heavy) fn(cx);
            }
        }
    }
}

void fused_free(Context& cx, FusedProgram* P)
{
    if (!P) return;
    cx.release(P->slabs);   // (the K tables sit behind the slabs in the same block)
    cx.release(P->desc);
    delete P;
}

int fused_launches(const FusedProgram* P) { return P ? P->launches : 0; }
int64_t fused_epoch(const FusedProgram* P) { return P ? P->epoch : -1; }
int64_t fused_plan_epoch(const FusedProgram* P) { return P ? P->plan_epoch : -1; }

void fused_slot_reset(Context& cx, FusedSlot& slot)
{
    if (slot.prog) fused_free(cx, slot.prog);
    slot.prog = nullptr;
    slot.disabled = false;
    slot.why.clear();
}

}  // namespace afesp
