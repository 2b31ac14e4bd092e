// capi.hip -- extern "C" boundary (include/afesp.h), the AO->MO transform, and the synthetic-input generators.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/afesp.h"
#include "ccsd.h"
#include "ccsd_so.h"
#include "tgemm.h"
#include "comm.h"
#include "fused.h"
#include "tall.h"

using namespace afesp;

struct afesp_ctx {
    Context cx;
    CCState cc;
    SOState so;
    double* eri_mo_dev = nullptr;   // packed MO integrals left on the device by afesp_ao2mo_mp2
    int64_t eri_mo_n = 0;           // nbasis they belong to
    double* eri_ao_dev = nullptr;   // packed AO integrals uploaded by afesp_read_eri_text
    // With the chains of a small-system iteration spread over lanes, issuing ~110 launches from the host (~4 us each) is
    // what is left; from the second call on the iteration is therefore replayed as a hipGraph (captured across the
    // lanes).  Only used where lanes are (small systems); AFESP_NO_GRAPH=1 keeps plain launches.
    struct GraphSlot {
        hipGraphExec_t exec = nullptr;
        int64_t epoch = -1;      // Context::scratch_epoch at capture
        int calls = 0;
        bool disabled = false;
        void reset()
        {
            if (exec) (void)hipGraphExecDestroy(exec);
            exec = nullptr;
            calls = 0;
            disabled = knobs().no_graph;
        }
    } graph_cc;
    int64_t eri_ao_n = 0;
    // the launch-fused path of a small system (fused.h): the recorded and levelled call sequences of the spin-free solver --
    // intermediates alone, amplitudes alone (the term-by-term entry points) and the whole iteration
    FusedSlot fused_int, fused_amp, fused_iter;
    FusedSlot fused_so;   // ... and the spin-orbital iteration (build_tau / F / W + update_amplitudes)
    void cc_programs_reset()
    {
        graph_cc.reset();
        fused_slot_reset(cx, fused_int);
        fused_slot_reset(cx, fused_amp);
        fused_slot_reset(cx, fused_iter);
    }
    void so_programs_reset() { fused_slot_reset(cx, fused_so); }
    // the LDS-DMA transforms' temporaries whose padding rows are known to be zero (afesp_ao2mo_mp2): buffers, extents, scratch epoch
    const double *pad_a = nullptr, *pad_b = nullptr;
    int64_t pad_n = 0, pad_ld = 0, pad_epoch = -1;
    int64_t half_n = 0, half_ld = 0, half_epoch = -1;   // scratch "ao2mo_a" holds the half-unpacked AO integrals of this basis size / leading dimension / epoch
};

namespace {

// The code-object preload (afesp_ctx_create) runs ONCE per process and device: the first context of a device starts the start-up
// thread, later ones start none.  What makes it safe beside the caller's own launches -- and two callers' launches beside each other --
// is not this claim but first_use.h: every first use of a kernel function, by the preload lists and by every launch site alike, is
// made under ONE process-wide lock (the round-5 abort "Cannot find Symbol with name ...slice_phys_kernel..." was the start-up thread
// and afesp_synthetic_init resolving that one function at the same moment).  Entry points therefore no longer wait for the preload
// (round 5 made them: the first Fock builds of els_amd run beside it again).
struct PreloadClaim {
    std::mutex mu;
    unsigned long long claimed = 0;   // bit per device (a device id >= 64 is never claimed: its kernels load on first use, under the lock)
    bool claim(int dev)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= 64 || (claimed >> dev) & 1ull) return false;
        claimed |= 1ull << dev;
        return true;
    }
};
PreloadClaim g_preload;

template <class F>
int guarded(afesp_ctx* c, F&& f)
{
    if (!c) return 1;
    first_use_tls_device() = c->cx.device;   // (the launch sites' per-device flags, first_use.h)
    knobs_refresh();                          // (every AFESP_* variable follows the environment call by call, knobs.h)
    // A body that threw may have forked lanes without joining them: before the caller can free or re-initialise anything, every
    // lane is idle and lane 0 is the one in use again.
    auto settle = [&]() {
        Context& cx = c->cx;
        if (!cx.lanes.empty()) {
            cx.quiesce();
            cx.use_lane(0);
            cx.marks_used = 0;
        }
    };
    try {
        f();
        return 0;
    } catch (const Error& e) {
        c->cx.last_error = e.what();
        settle();
        return e.code ? e.code : 1;
    } catch (const std::exception& e) {
        c->cx.last_error = e.what();
        settle();
        return 1;
    }
}

int64_t neri_of(int64_t n)
{
    int64_t np = n * (n + 1) / 2;
    return np * (np + 1) / 2;
}

// splitmix64 -> uniform in [0,1)
__device__ __forceinline__ double hash_uniform(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
__global__ void synth_packed_kernel(double* packed, int64_t n, double scale, uint64_t seed)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x)
        packed[k] = scale * (2.0 * hash_uniform(seed + (uint64_t)k) - 1.0);
}

}  // namespace

// Runs `body` (launches on the context's lanes, no host synchronisation) directly for the first AFESP_GRAPH_AFTER calls,
// then captures it into a graph once and replays the graph afterwards.  Nothing executes during capture, so a failed
// capture simply falls back to running the body.
template <typename Body>
static void replay(afesp_ctx* ctx, afesp_ctx::GraphSlot& g, bool eligible, Body body)
{
    Context& cx = ctx->cx;
    if (g.exec && g.epoch != cx.scratch_epoch) {   // a scratch buffer the graph refers to may have been freed since
        (void)hipGraphExecDestroy(g.exec);
        g.exec = nullptr;
        g.calls = 0;
    }
    if (g.exec) {
        AFESP_HIP(hipGraphLaunch(g.exec, cx.stream));
        return;
    }
    if (!eligible || g.disabled) {
        body();
        return;
    }
    // Capturing and instantiating the ~110-node graph costs ~10 ms; a replay saves ~0.1 ms over the laned launches.  A real
    // molecule converges in 15-30 iterations, so the capture waits until a context has iterated long enough for it to pay
    // (AFESP_GRAPH_AFTER, default 40 calls).
    const int graph_after = knobs().graph_after;
    if (g.calls == 0 || g.epoch != cx.scratch_epoch || g.calls < graph_after) {
        // first call, or cached scratch buffers were dropped since the last one: whatever the body (re)builds or allocates is
        // done here, outside any capture
        body();
        g.calls = (g.epoch != cx.scratch_epoch) ? 1 : g.calls + 1;
        g.epoch = cx.scratch_epoch;
        return;
    }
    // The capture is opened on the origin stream (lane 0).  A body that throws half-way leaves another lane selected and
    // events outstanding: both are put back BEFORE the capture is ended, and the capture is ended on the origin stream --
    // ending it on a lane's stream would leave lane 0 capturing for ever, and the direct run below would execute nothing.
    cx.use_lane(0);
    hipStream_t origin = cx.stream;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(origin, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        g.disabled = true;
        body();
        return;
    }
    bool ok = true;
    try {
        body();
    } catch (...) {
        ok = false;
    }
    cx.use_lane(0);
    cx.marks_used = 0;
    const hipError_t e = hipStreamEndCapture(origin, &graph);
    if (ok && e == hipSuccess && graph && hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) == hipSuccess) {
        (void)hipGraphDestroy(graph);
        g.epoch = cx.scratch_epoch;
        AFESP_HIP(hipGraphLaunch(g.exec, cx.stream));
        return;
    }
    if (knobs().graph_debug) fprintf(stderr, "afesp: graph capture failed (body ok %d, end capture %d)\n", (int)ok, (int)e);
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    g.exec = nullptr;
    g.disabled = true;
    // a failed capture (e.g. lanes left unjoined by the throw) has been invalidated by EndCapture; make sure of it
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(origin, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        throw Error(2, "afesp: the stream is still capturing after a failed graph capture");
    }
    for (size_t i = 1; i < cx.lanes.size(); ++i) {   // lanes that were pulled into the capture are out of it as well
        hipStreamCaptureStatus ls = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(cx.lanes[i].stream, &ls) != hipSuccess || ls != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            throw Error(2, "afesp: a lane is still capturing after a failed graph capture");
        }
    }
    body();
}

// One iteration up to the energy kernels (no host synchronisation): the launch-fused program of a small system (fused.h; recorded
// from the very calls below on first use), the call-by-call sequence otherwise.
static bool ccsd_iteration_body(afesp_ctx* ctx)   // true: the launch-fused program ran (read with ccsd_tail_read)
{
    if (ccsd_uses_lanes(ctx->cc) &&
        fused_exec(ctx->cx, ctx->fused_iter, [&] {
            ccsd_intermediates(ctx->cx, ctx->cc, true);
            ccsd_amplitudes(ctx->cx, ctx->cc, true);
            ccsd_tail_launch(ctx->cx, ctx->cc);
        }))
        return true;
    // Large systems (one stream, whole-tensor products): the same two-kernel tail -- P(ia/jb) + division, the energy / rms sums and the
    // DIIS history push in ONE pass over the residual instead of three (update, energy, push: 26 against 23 passes over o^2 v^2 elements at
    // eight history vectors, and no host wait between the energy and the push); the <= 17 x 17 system is then solved on the host.
    // AFESP_LARGE_TAIL=0: the three kernels.
    if (!ccsd_uses_lanes(ctx->cc) && knobs().large_tail) {
        ccsd_intermediates(ctx->cx, ctx->cc, true);
        ccsd_amplitudes(ctx->cx, ctx->cc, true);
        ccsd_tail_launch(ctx->cx, ctx->cc);
        return true;
    }
    ctx->cc.tail_pending = false;
    replay(ctx, ctx->graph_cc, ccsd_uses_lanes(ctx->cc), [&] {
        ccsd_intermediates(ctx->cx, ctx->cc, true);
        ccsd_amplitudes(ctx->cx, ctx->cc);
        ccsd_energy_launch(ctx->cx, ctx->cc);
    });
    return false;
}

extern "C" {

int afesp_version(void) { return 1; }
int64_t afesp_neri(int64_t nbasis) { return neri_of(nbasis); }

int afesp_ctx_create(int device, afesp_ctx** out)
{
    if (!out) return 1;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return 10;   // no GPU: fail loudly, no CPU fallback
    if (device < 0 || device >= ndev) return 11;
    if (hipSetDevice(device) != hipSuccess) return 12;
    afesp_ctx* c = new afesp_ctx();
    c->cx.device = device;
    int rc = guarded(c, [&] {
        AFESP_HIP(hipStreamCreate(&c->cx.stream));
        c->cx.scal = c->cx.alloc(64 + 18 * 512);
        AFESP_HIP(hipHostMalloc((void**)&c->cx.scal_host, sizeof(double) * 64, hipHostMallocDefault));
        AFESP_HIP(hipHostMalloc((void**)&c->cx.res_host, sizeof(double) * (8 + 256 + 72), hipHostMallocCoherent | hipHostMallocMapped));
        memset(c->cx.res_host, 0, sizeof(double) * (8 + 256 + 72));
        AFESP_HIP(hipHostGetDevicePointer((void**)&c->cx.res_dev, c->cx.res_host, 0));
        c->cx.ws.bytes = (size_t)256 << 20;   // split-K slabs
        c->cx.ws.ptr = c->cx.alloc((int64_t)(c->cx.ws.bytes / sizeof(double)));
        c->cx.sync();
        // The device code of a translation unit is loaded on the first use of one of its kernels -- 55-70 ms in all, which a
        // small molecule would pay inside its first CCSD iteration.  Ask for it now, on a thread of its own: the caller goes
        // on with its host work (parsing eri.dat, the SCF set-up) meanwhile.  AFESP_NO_PRELOAD=1 switches it off.
        if (!knobs().no_preload) {
            // one start-up thread per process and device (PreloadClaim); a later context of the device starts none
            if (g_preload.claim(device)) c->cx.startup = std::thread([device, c] {
                first_use_tls_device() = device;
                if (hipSetDevice(device) != hipSuccess) return;
                // (the parallel streams of the call-by-call iteration: small systems run the launch-fused iteration on ONE stream since
                // round 4, so the 10-25 ms of queue creation are only spent ahead of time on request; fork() makes them when needed)
                if (knobs().preload_lanes) Context::prepare_lanes(c->cx.prepared, 6);   // (the context itself is not touched: fork() adopts them)
                {
                    // the runtime's own first-use set-up (staging buffers of pageable copies, its fill / copy kernels): ~8 ms that
                    // the first plan upload of a process would otherwise pay inside the first CCSD iteration
                    void* d = nullptr;
                    hipStream_t st = nullptr;
                    std::vector<double> h(4096, 1.0);
                    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipMalloc(&d, h.size() * sizeof(double)) == hipSuccess) {
                        (void)hipMemsetAsync(d, 0, h.size() * sizeof(double), st);
                        (void)hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, st);
                        (void)hipMemcpyAsync((char*)d + 8192, d, 8192, hipMemcpyDeviceToDevice, st);
                        (void)hipMemcpyAsync(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost, st);
                        (void)hipStreamSynchronize(st);
                    }
                    if (d) (void)hipFree(d);
                    if (st) (void)hipStreamDestroy(st);
                    (void)hipGetLastError();
                }
                const bool dbg = knobs().preload_debug;   // time per translation unit on stderr
                auto timed = [dbg](const char* what, void (*fn)()) {
                    const auto t0 = std::chrono::steady_clock::now();
                    fn();
                    if (dbg) fprintf(stderr, "afesp preload %-10s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                };
                // (the gather kernel's module -- 10 ms to load, a hundred instantiations -- is not asked for: a small system never
                // launches it, and whatever loads here holds the runtime's lock against the caller's own first launches, e.g. the
                // Fock builds of the SCF that els_amd starts at once; a large system loads it with its first product.
                // AFESP_PRELOAD_GETT=1 restores it.)
                timed("kernels", preload_kernels);
                timed("fused", preload_fused);
                timed("contract", preload_contract);
                timed("small path", preload_small_path_kernels);
                timed("triples", preload_triples);
                timed("ccsd_so", preload_ccsd_so);
                if (knobs().preload_gett) timed("gett", preload_gett);
            });
        }
    });
    if (rc) {
        if (c->cx.startup.joinable()) c->cx.startup.join();
        delete c;
        return rc;
    }
    *out = c;
    return 0;
}

void afesp_ctx_destroy(afesp_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->cx.device);
    if (ctx->cx.startup.joinable()) ctx->cx.startup.join();
    ctx->cc_programs_reset();
    ctx->so_programs_reset();
    comm_destroy(ctx->cx.comm);
    ctx->cx.comm = nullptr;
    triples_plan_free(ctx->cc);
    so_triples_plan_free(ctx->so);
    ring_free(ctx->cx, ctx->cc);   // (the host-side descriptor of the ring launches; its device blocks go with the context)
    delete ctx;
}

const char* afesp_last_error(const afesp_ctx* ctx) { return ctx ? ctx->cx.last_error.c_str() : "null context"; }

// Which form afesp_ao2mo_mp2 takes for basis size n, and the leading dimension of the squared-up temporaries that goes with it (the
// half-unpacked AO integrals afesp_build_fock leaves for it included): the transforms on the LDS-DMA GEMM keep columns of
// 16 ceil(n / 16) doubles -- every column then starts on a 128-byte line, for the GEMM's K steps and for the layout kernels' runs alike
// (n = 220: 39.2 -> 36 ms per transform; n = 224 ran FASTER than n = 220 before, profiles/r06_ao2mo_alignment_scan.txt) -- every other
// form keeps n.
static bool ao2mo_blocked(int64_t n)
{
    const int64_t np = n * (n + 1) / 2;
    return knobs().ao2mo_blocked >= 0 ? knobs().ao2mo_blocked == 1 : n * n * np >= ((int64_t)1 << 31);
}
static bool ao2mo_use_tg(int64_t n)
{
    // the LDS-DMA GEMM: even n, and from n = 96 on (its tile has 128 rows: below that most of a tile is padding and the transform is
    // launch-bound anyway); AFESP_AO2MO_TG=0 / 1: never / for every even n >= 16 (tests, A/B runs)
    return !ao2mo_blocked(n) && n % 2 == 0 && n >= 16 && (knobs().ao2mo_tg >= 0 ? knobs().ao2mo_tg == 1 : n >= 96);
}
static int64_t ao2mo_ld(int64_t n) { return ao2mo_use_tg(n) && knobs().ao2mo_pad ? (n + 15) / 16 * 16 : n; }

// ---- a quarter transform on the LDS-DMA GEMM (tgemm.h): out(x2, m, S) = sum_x1 C(m, x1) in(x1, x2, S)
// The transformed index is the fastest one of `in`, so every column (x2, S) of the product is a contiguous run of n doubles: both
// operands are contiguous along the summation index (C goes in as a zero-padded transpose), which is all that kernel asks for.
// The result comes out with x2 fastest and the new index second -- the layout the NEXT quarter transform wants for its input
// (and the one the old path produced after two of them: (p,q,K), (r,s,P)).  Needs an even n (16-byte chunks, pairs of columns).
__global__ __launch_bounds__(256) void ao2mo_ct_kernel(double* ct, const double* c, int n, int Kc)
{
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < n * Kc; x += gridDim.x * blockDim.x) {
        const int m = x / Kc, k = x % Kc;
        ct[x] = k < n ? c[m + n * k] : 0.0;
    }
}
// rowA[m] = byte offset of row m of the padded transpose; colB[c] = byte offset of column c = x2 + n Sloc of a slab of `in`;
// offCm[m] = ld m; offCn[c] = x2 + ld n Sloc (elements); the pads behind them (tgemm.h) are zero.
// ld: the temporaries' columns are ld doubles long (n of them data, the rest zero): ld = Kc puts every column on a 128-byte line
__global__ __launch_bounds__(256) void ao2mo_tables_kernel(uint32_t* rowA, uint32_t* colB, int64_t* offCm, int64_t* offCn, int n, int Kc, int64_t ncol, int64_t ld)
{
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < ncol + 256; x += (int64_t)gridDim.x * blockDim.x) {
        if (x < n + 256) rowA[x] = x < n ? (uint32_t)(8 * Kc * x) : 0u;
        if (x < n + 128) offCm[x] = x < n ? ld * x : 0;
        colB[x] = x < ncol ? (uint32_t)(8 * ld * x) : 0u;
        if (x < ncol + 128) offCn[x] = x < ncol ? (x % n) + ld * n * (x / n) : 0;
    }
}

// The second pair of transforms is only needed where the packed result has an entry: (rs|PQ) for RS <= PQ, i.e. r <= p(PQ).  The
// last transform therefore runs over the columns (r, PQ) with r <= p only -- p + 1 of them per pair PQ = tri(p, q), rounded up to
// an even count (pairs of columns are stored together) -- about half of all: colB / offCn list them pair by pair, relative to
// the pair's slab (cstart[PQ] = first column of the pair).
__global__ __launch_bounds__(256) void ao2mo_tables_tri_kernel(uint32_t* colB, int64_t* offCn, const int64_t* cstart, int n, int64_t np, int64_t sl, int64_t ld)
{
    for (int64_t P = blockIdx.x; P < np; P += gridDim.x) {
        const int64_t c0 = cstart[P], cnt = cstart[P + 1] - c0, rel = P % sl;
        for (int64_t r = threadIdx.x; r < cnt; r += blockDim.x) {
            colB[c0 + r] = (uint32_t)(8 * ld * (r + (int64_t)n * rel));
            offCn[c0 + r] = r + ld * n * rel;
        }
    }
}

// columns (x2, S) with x2 < TG_BM only (the pair transposition behind the second transform reads its result (x2, m, S) for
// x2 <= m only: the rows m < 128 are needed for these columns only), relative to a slab: column c = x2 + 128 Sloc
__global__ __launch_bounds__(256) void ao2mo_tables_lo_kernel(uint32_t* colB, int64_t* offCn, int n, int cnt, int64_t ncol, int64_t ld)
{
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < ncol + 256; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x2 = x % cnt, sloc = x / cnt;
        colB[x] = x < ncol ? (uint32_t)(8 * ld * (x2 + (int64_t)n * sloc)) : 0u;
        if (x < ncol + 128) offCn[x] = x < ncol ? x2 + ld * n * sloc : 0;
    }
}

namespace {
struct Ao2moTg {
    int64_t n = 0, Kc = 0, sl = 0;   // basis size, padded summation length, (S) pairs per slab (one TgGroup each)
    int64_t ld = 0;                  // leading dimension of the temporaries: Kc (ao2mo_ld), or n
    double* ct = nullptr;
    uint32_t *rowA = nullptr, *colB = nullptr;
    int64_t *offCm = nullptr, *offCn = nullptr;
    TgGroup* groups = nullptr;       // room for the descriptors of every transform of one call (no host synchronisation between them)
    int64_t groups_cap = 0, groups_used = 0;
    std::vector<std::vector<TgGroup>> host;   // ... whose host copies live until the call's final synchronisation
    // columns (r, PQ), r <= p(PQ) only (ao2mo_tables_tri_kernel)
    uint32_t* colB_tri = nullptr;
    int64_t* offCn_tri = nullptr;
    std::vector<int64_t> cstart;     // [np + 1], host copy
    int64_t p_split = 0;             // first pair PQ with p >= TG_BM (the pairs below it need the first 128 rows only)
    uint32_t* colB_lo = nullptr;     // columns x2 < TG_BM (ao2mo_tables_lo_kernel)
    int64_t* offCn_lo = nullptr;
};

// tables and the padded transpose of the coefficient matrix for basis size n (cached scratch: rebuilt per call, microseconds)
static Ao2moTg ao2mo_tg_prepare(Context& cx, const double* Cm, int64_t n, int64_t np, int64_t ld)
{
    Ao2moTg t;
    t.n = n;
    t.ld = ld;
    t.Kc = (n + 15) / 16 * 16;
    // a slab's columns are addressed with 32-bit byte offsets: n * sl columns of n doubles each below 4 GiB
    t.sl = std::min<int64_t>(np, std::min<int64_t>(8192, (((int64_t)1 << 32) - 4096) / (8 * t.ld * n)));
    const int64_t ncol = n * t.sl;
    t.ct = cx.scratch("ao2mo_ct", n * t.Kc);
    t.rowA = (uint32_t*)cx.scratch("ao2mo_t32", (n + 256 + ncol + 256) / 2 + 2);
    t.colB = t.rowA + n + 256;
    t.offCm = (int64_t*)cx.scratch("ao2mo_t64", n + 128 + ncol + 128 + 2);
    t.offCn = t.offCm + n + 128;
    AFESP_KLAUNCH(ao2mo_ct_kernel, dim3((unsigned)((n * t.Kc + 255) / 256)), dim3(256), 0, cx.stream, t.ct, Cm, (int)n, (int)t.Kc);
    AFESP_HIP(hipGetLastError());
    AFESP_KLAUNCH(ao2mo_tables_kernel, dim3((unsigned)std::min<int64_t>((ncol + 256 + 255) / 256, 65536)), dim3(256), 0, cx.stream, t.rowA,
                       t.colB, t.offCm, t.offCn, (int)n, (int)t.Kc, ncol, t.ld);
    AFESP_HIP(hipGetLastError());
    const int64_t ng = (np + t.sl - 1) / t.sl;
    t.groups_cap = 8 * (ng + 2);
    t.groups = (TgGroup*)cx.scratch("ao2mo_tg", (int64_t)(t.groups_cap * sizeof(TgGroup) / sizeof(double) + 1));
    // the triangular column list of the last transform
    t.cstart.assign((size_t)np + 1, 0);
    for (int64_t pp = 0, P = 0; pp < n; ++pp)
        for (int64_t q = 0; q <= pp; ++q, ++P) t.cstart[(size_t)P + 1] = t.cstart[(size_t)P] + ((pp + 2) & ~(int64_t)1);
    t.p_split = std::min<int64_t>(np, (int64_t)TG_BM * (TG_BM + 1) / 2);
    const int64_t ctot = t.cstart[(size_t)np];
    t.colB_tri = (uint32_t*)cx.scratch("ao2mo_t32t", (ctot + 256) / 2 + 2);
    t.offCn_tri = (int64_t*)cx.scratch("ao2mo_t64t", ctot + 128 + 2);
    int64_t* cs_dev = (int64_t*)cx.scratch("ao2mo_cs", np + 2);
    AFESP_HIP(hipMemcpyAsync(cs_dev, t.cstart.data(), (size_t)(np + 1) * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
    AFESP_HIP(hipMemsetAsync(t.colB_tri + ctot, 0, 256 * sizeof(uint32_t), cx.stream));
    AFESP_HIP(hipMemsetAsync(t.offCn_tri + ctot, 0, 128 * sizeof(int64_t), cx.stream));
    AFESP_KLAUNCH(ao2mo_tables_tri_kernel, dim3((unsigned)std::min<int64_t>(np, 65536)), dim3(256), 0, cx.stream, t.colB_tri, t.offCn_tri,
                       cs_dev, (int)n, np, t.sl, t.ld);
    AFESP_HIP(hipGetLastError());
    if (n > TG_BM) {
        const int64_t nlo = (int64_t)TG_BM * t.sl;
        t.colB_lo = (uint32_t*)cx.scratch("ao2mo_t32h", (nlo + 256) / 2 + 2);
        t.offCn_lo = (int64_t*)cx.scratch("ao2mo_t64h", nlo + 128 + 2);
        AFESP_KLAUNCH(ao2mo_tables_lo_kernel, dim3((unsigned)std::min<int64_t>((nlo + 256 + 255) / 256, 65536)), dim3(256), 0, cx.stream,
                           t.colB_lo, t.offCn_lo, (int)n, (int)TG_BM, nlo, t.ld);
        AFESP_HIP(hipGetLastError());
    }
    return t;
}

// one quarter transform over the pairs S in [s_begin, s_end) of `in` (n x n x np), rows row0 <= m < row0 + M of the result only;
// cols: 0 every column (x2, S), 1 only x2 <= p(S), 2 only x2 < 128 (the rest of `out` is left untouched)
static void ao2mo_tg_xform(Context& cx, Ao2moTg& t, const double* in, double* out, int64_t s_begin, int64_t s_end, int64_t M, int cols,
                    int64_t row0 = 0)
{
    if (s_end <= s_begin || M <= 0) return;
    const bool tri = cols == 1, lo = cols == 2;
    const int64_t nlo = TG_BM;
    const int64_t n = t.n, g_lo = s_begin / t.sl, g_hi = (s_end - 1) / t.sl;
    const int mt = (int)((M + TG_BM - 1) / TG_BM);
    t.host.emplace_back();
    std::vector<TgGroup>& hv = t.host.back();
    int mx = 0, tile = 0;
    auto ncols = [&](int64_t s0, int64_t s1) { return tri ? t.cstart[(size_t)s1] - t.cstart[(size_t)s0] : (lo ? nlo : n) * (s1 - s0); };
    for (int64_t g = g_lo; g <= g_hi; ++g) {
        const int64_t s0 = std::max(s_begin, g * t.sl), s1 = std::min(s_end, (g + 1) * t.sl);
        mx = std::max(mx, (int)((ncols(s0, s1) + TG_BN - 1) / TG_BN));
    }
    const int gm = tgemm_group_m((int)M, mx);
    for (int64_t g = g_lo; g <= g_hi; ++g) {
        const int64_t s0 = std::max(s_begin, g * t.sl), s1 = std::min(s_end, (g + 1) * t.sl);
        TgGroup d{};
        d.a1 = d.a2 = 0;
        d.b1 = d.b2 = t.ld * n * g * t.sl;       // (the tables are relative to the slab's first pair; columns are ld doubles long)
        d.c0 = t.ld * n * g * t.sl;
        d.colB = tri ? t.colB_tri + t.cstart[(size_t)s0] : lo ? t.colB_lo + nlo * (s0 - g * t.sl) : t.colB + n * (s0 - g * t.sl);
        d.offCn = tri ? t.offCn_tri + t.cstart[(size_t)s0] : lo ? t.offCn_lo + nlo * (s0 - g * t.sl) : t.offCn + n * (s0 - g * t.sl);
        d.N = (int)ncols(s0, s1);
        d.ntiles = (d.N + TG_BN - 1) / TG_BN;
        d.tile_start = tile;
        d.nk1 = d.nk = (int)(t.Kc / TG_BK);
        d.inv_width = tgemm_inverse(gm * d.ntiles);
        if ((int64_t)mt * d.ntiles * gm * d.ntiles >= ((int64_t)1 << 32)) throw Error(2, "ao2mo: tile walk out of range");
        tile += mt * d.ntiles;
        hv.push_back(d);
    }
    TgGroup end{};
    end.tile_start = tile;
    hv.push_back(end);
    const int ng = (int)hv.size() - 1;
    if (t.groups_used + (int64_t)hv.size() > t.groups_cap) throw Error(2, "ao2mo: descriptor buffer too small");
    TgGroup* dev = t.groups + t.groups_used;
    t.groups_used += (int64_t)hv.size();
    AFESP_HIP(hipMemcpyAsync(dev, hv.data(), hv.size() * sizeof(TgGroup), hipMemcpyHostToDevice, cx.stream));
    TgProblem p{t.ct, in, out, t.rowA + row0, t.offCm + row0, (int)M, true, (int)((n - (t.Kc - TG_BK) + 3) / 4)};
    p.tag = 2;
    // (rows that end at most 96 past a multiple of 128 -- n = 220: 92 -- take a 96-row last tile: three quarters of its MFMAs, tgemm.h)
    const int bm = (knobs().ao2mo_mixed && M % TG_BM != 0 && M % TG_BM <= 96) ? TG_BM : 0;
    AFESP_HIP(tgemm_launch(p, dev, ng, tile, mx, cx.stream, cx.tg, bm));
}
}  // namespace

// src/mp2.f90:261-449.  Four quarter transforms as MFMA GEMMs; each pass contracts the leading AO index with C(MO,AO) and
// the planner writes the result with the new MO index in place.
int afesp_ao2mo_mp2(afesp_ctx* ctx, int64_t nbasis, int64_t nocc, const double* canon_coeff, const double* canon_levels,
                    const double* eri_packed, double* eri_mo_packed, double* e_mp2)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        const int64_t n = nbasis, o = nocc, v = n - o, ne = neri_of(n);
        if (n <= 0 || o <= 0 || v <= 0 || n > 1024) throw Error(1, "afesp_ao2mo_mp2: bad extents");
        if (!eri_packed && (!ctx->eri_ao_dev || ctx->eri_ao_n != n))
            throw Error(1, "afesp_ao2mo_mp2: eri_packed is NULL and no AO integrals were read onto the device for this basis size");
        // upload buffer, then the packed MO integrals; a transform of the same basis size overwrites the previous result
        cx.drop_scratch("t_");   // the (T) pool of a previous system holds the blocks the two temporaries below were
        double* packed = ctx->eri_mo_dev;
        if (ctx->cc.eri_src == packed) ctx->cc.eri_src = nullptr;   // a solver state initialised from them can no longer form <ef|ab>
        if (!packed || ctx->eri_mo_n != n) {
            if (packed) cx.release(packed);
            ctx->eri_mo_dev = nullptr;
            packed = cx.alloc_raw(ne);
        }
        const double* ao = ctx->eri_ao_dev;  // NULL source: transformed where afesp_read_eri_text / afesp_set_eri left them
        if (eri_packed) {
            AFESP_HIP(hipMemcpyAsync(packed, eri_packed, sizeof(double) * ne, hipMemcpyHostToDevice, cx.stream));
            ao = packed;
        }
        Tensor Cm = view(cx.scratch("ao2mo_c", n * n), {n, n});
        AFESP_HIP(hipMemcpyAsync(Cm.d, canon_coeff, sizeof(double) * n * n, hipMemcpyHostToDevice, cx.stream));
        // Pair symmetry: (ij|kl) is transformed for the n(n+1)/2 pairs k >= l only, the half-transformed (pq|kl) kept for
        // p >= q only -- 4 n^5 flop and two buffers of n^2 x npair instead of 8 n^5 and two of n^4.
        const int64_t np = n * (n + 1) / 2;
        const int64_t L = ao2mo_ld(n);   // leading dimension of the squared-up temporaries (and of what afesp_build_fock left)
        const bool have_u = ao == ctx->eri_ao_dev && ctx->half_n == n && ctx->half_ld == L && ctx->half_epoch == cx.scratch_epoch;   // afesp_build_fock left (ij|KL)
        // slab by slab from temporaries of 16 GiB each (n >= 256): at n = 220 the blocked form is 5 % slower (58.9 against 55.8 ms:
        // one more pass over the half-transformed integrals) for 12.5 GB less -- it is there for the sizes where 2 n^2 npair
        // doubles no longer fit beside the rest (n = 400: 2 x 103 GB)
        const bool blocked = ao2mo_blocked(n);
        if (!blocked) {
            // Small bases: the whole tensor at once, nine launches.  The two temporaries are cached scratch: a second transform in
            // the same context reuses them, the next afesp_ccsd_init / afesp_ccsd_so_init gives them back.
            // (16 doubles of slack behind each: the LDS-DMA GEMM reads whole 16-element K steps, i.e. up to Kc - n elements past a
            // column's end -- the next column's, finite, times the zero padding of C -- and past the tensor's end behind the last one)
            // (the LDS-DMA form: columns of L = 16 ceil(n / 16) doubles, ao2mo_ld)
            Tensor Ta = view(cx.scratch("ao2mo_a", L * n * np + 16), {n, n, np}), Tb = view(cx.scratch("ao2mo_b", L * n * np + 16), {n, n, np});
            const bool use_tg = ao2mo_use_tg(n);
            // up to 64 basis functions: the LDS-resident pair transform (AFESP_AO2MO_PAIR=0: the gather-GEMM form)
            const bool pair_path = n <= 64 && !use_tg && knobs().ao2mo_pair;
            if (!have_u && !pair_path) k_unpack_half(cx, Ta.d, ao, (int)n, 0, -1, (int)L);   // (ij|KL), ij squared up
            ctx->half_n = 0;                                             // the transform overwrites it
            if (use_tg) {
                AFESP_HIP(hipMemsetAsync(Ta.d + L * n * np, 0, 16 * sizeof(double), cx.stream));
                AFESP_HIP(hipMemsetAsync(Tb.d + L * n * np, 0, 16 * sizeof(double), cx.stream));
                // (rows n .. L - 1 of every column are K padding of the products -- read, times the zero padding of C: finite, so zero.
                // No kernel of this form or of afesp_build_fock writes them, so they are zeroed once per (buffers, n, L): 0.2 ms each)
                if (ctx->pad_a != Ta.d || ctx->pad_b != Tb.d || ctx->pad_n != n || ctx->pad_ld != L || ctx->pad_epoch != cx.scratch_epoch) {
                    k_pad_rows_zero(cx, Ta.d, (int)n, (int)L, n * np);
                    k_pad_rows_zero(cx, Tb.d, (int)n, (int)L, n * np);
                    ctx->pad_a = Ta.d; ctx->pad_b = Tb.d; ctx->pad_n = n; ctx->pad_ld = L; ctx->pad_epoch = cx.scratch_epoch;
                }
                Ao2moTg tg = ao2mo_tg_prepare(cx, Cm.d, n, np, L);
                ao2mo_tg_xform(cx, tg, Ta.d, Tb.d, 0, np, n, 0);         // (ij|K) -> (j p|K)        mp2.f90:321-333
                // (jp|K) -> (x2 m|K), mp2.f90:338-348: the transposition below reads x2 <= m only -- the rows m < 128 are computed
                // for the columns x2 < 128 only
                if (n > TG_BM) {
                    ao2mo_tg_xform(cx, tg, Tb.d, Ta.d, 0, np, n - TG_BM, 0, TG_BM);
                    ao2mo_tg_xform(cx, tg, Tb.d, Ta.d, 0, np, TG_BM, 2);
                } else {
                    ao2mo_tg_xform(cx, tg, Tb.d, Ta.d, 0, np, n, 0);
                }
                k_pair_transpose(cx, Tb.d, Ta.d, (int)n, (int)L);       // (kl|PQ), kl squared up, p >= q
                // Second pair: only (rs|PQ) with RS <= PQ is packed (mp2.f90:388-410), i.e. r <= p and s <= r.  Rows beyond the
                // first 128 are therefore skipped for the pairs with p < 128, and the last transform runs over the columns
                // (r, PQ) with r <= p only -- 2.6 n^5 flop in 128-row tiles instead of 4 (the reference: 8).
                const int64_t ps = tg.p_split, m_lo = std::min<int64_t>(n, TG_BM);
                ao2mo_tg_xform(cx, tg, Tb.d, Ta.d, 0, ps, m_lo, 0);      // (kl|P) -> (l r|P)        mp2.f90:357-367
                ao2mo_tg_xform(cx, tg, Tb.d, Ta.d, ps, np, n, 0);
                ao2mo_tg_xform(cx, tg, Ta.d, Tb.d, 0, ps, m_lo, 1);      // (lr|P) -> (r s|P)        mp2.f90:375-385
                ao2mo_tg_xform(cx, tg, Ta.d, Tb.d, ps, np, n, 1);
                k_pack_pairs(cx, packed, Tb.d, (int)n, 0, -1, (int)L);   // mp2.f90:388-410
                cx.sync();                                               // (the descriptors' host copies die with tg)
            } else if (pair_path) {
                ctx->pad_n = 0;   // (this form writes the temporaries densely: whatever padding rows another form had zeroed are data now)
                // up to 64 basis functions (every bundled input, the H2O/cc-pVTZ shape): both quarter transforms of a pair index in one
                // kernel with the n x n block resident in LDS -- five launches for the whole transform (AFESP_AO2MO_PAIR=0: the
                // gather-GEMM form below)
                // straight from the packed AO integrals to pair columns, one transposition of the npair x npair matrix, straight into
                // the packed MO integrals: three launches, no squared-up copy (Ta / Tb hold the two npair^2 matrices)
                k_pair_xform(cx, Tb.d, ao, Cm.d, (int)n, np, 1);             // (ij|K) -> g(PQ, K)       mp2.f90:321-348
                k_square_transpose(cx, Ta.d, Tb.d, np);                      // g(K, PQ)
                k_pair_xform(cx, packed, Ta.d, Cm.d, (int)n, np, 2);         // (kl|P) -> (rs|P), RS <= P  mp2.f90:357-410
            } else {
                ctx->pad_n = 0;
                contract(cx, 1.0, Cm, "pi", Ta, "ijK", 0.0, Tb, "pjK");      // mp2.f90:321-333
                contract(cx, 1.0, Cm, "qj", Tb, "pjK", 0.0, Ta, "pqK");      // mp2.f90:338-348
                k_pair_transpose(cx, Tb.d, Ta.d, (int)n);                    // (kl|PQ), kl squared up, p >= q
                contract(cx, 1.0, Cm, "rk", Tb, "klP", 0.0, Ta, "rlP");      // mp2.f90:357-367
                contract(cx, 1.0, Cm, "sl", Ta, "rlP", 0.0, Tb, "rsP");      // mp2.f90:375-385
                k_pack_pairs(cx, packed, Tb.d, (int)n);                  // mp2.f90:388-410
            }
        } else {
            // Large bases: slab by slab.  The first pair of transforms acts on every (kl) pair separately and the second on every
            // (pq) pair, so only the half-transformed integrals have to exist as a whole -- pair-packed, g(PQ,K), np^2 doubles
            // (4.7 GB at n = 220) -- and the n^2 npair temporaries (2 x 9.4 GB) shrink to two slabs of S pairs.  S is chosen so
            // that a slab's column tiles fill whole rounds of the persistent GEMM grid.
            ctx->pad_n = 0;
            int64_t S = std::max<int64_t>(16, ((int64_t)256 * 128 * 14 / n) / 16 * 16);
            if (S > np) S = (np + 15) / 16 * 16;
            double* g = cx.scratch("ao2mo_g", np * np);
            double* sa = have_u ? nullptr : cx.scratch("ao2mo_a", n * n * S);   // (with (ij|KL) left by the Fock build: its slabs, in place)
            double* sb = cx.scratch("ao2mo_b", n * n * S);
            double* u = have_u ? cx.scratch("ao2mo_a", n * n * np) : nullptr;
            ctx->half_n = 0;                                             // the transform overwrites it
            for (int64_t k0 = 0; k0 < np; k0 += S) {
                const int64_t k1 = std::min(np, k0 + S), len = k1 - k0;
                double* a_s = have_u ? u + n * n * k0 : sa;
                if (!have_u) k_unpack_half(cx, a_s, ao, (int)n, k0, k1);                     // (ij|K), ij squared up, K in the slab
                Tensor Ta = view(a_s, {n, n, len}), Tb = view(sb, {n, n, len});
                contract(cx, 1.0, Cm, "pi", Ta, "ijK", 0.0, Tb, "pjK");                      // mp2.f90:321-333
                contract(cx, 1.0, Cm, "qj", Tb, "pjK", 0.0, Ta, "pqK");                      // mp2.f90:338-348
                k_tri_pack(cx, g, a_s, (int)n, k0, k1);                                      // g(PQ,K), p >= q
            }
            if (have_u) { sa = u; }                                      // (dead now: its first slab serves the second pair)
            for (int64_t p0 = 0; p0 < np; p0 += S) {
                const int64_t p1 = std::min(np, p0 + S), len = p1 - p0;
                k_pair_square_packed(cx, sb, g, (int)n, p0, p1);                             // (kl|P), kl squared up, P in the slab
                Tensor Ta = view(sa, {n, n, len}), Tb = view(sb, {n, n, len});
                contract(cx, 1.0, Cm, "rk", Tb, "klP", 0.0, Ta, "rlP");                      // mp2.f90:357-367
                contract(cx, 1.0, Cm, "sl", Ta, "rlP", 0.0, Tb, "rsP");                      // mp2.f90:375-385
                k_pack_pairs(cx, packed, sb, (int)n, p0, p1);                                // mp2.f90:388-410
            }
        }
        ctx->eri_mo_dev = packed;
        ctx->eri_mo_n = n;
        // MP2 energy on the <ij|ab> slice (mp2.f90:418-440)
        double emp2 = 0.0;
        if (o * o * v * v <= ((int64_t)1 << 22) && knobs().mp2_packed) {   // (AFESP_MP2_PACKED=0: the five-launch form at every size, A/B runs)
            // small systems: one launch, straight from the packed array (the slice and the denominators are formed on the fly)
            emp2 = k_mp2_packed(cx, packed, canon_levels, (int)o, (int)v);
        } else {
            double* e_dev = cx.scratch("ao2mo_e", n);
            AFESP_HIP(hipMemcpyAsync(e_dev, canon_levels, sizeof(double) * n, hipMemcpyHostToDevice, cx.stream));
            Tensor voovv = view(cx.scratch("ao2mo_v", o * o * v * v), {o, o, v, v}), D1 = view(cx.scratch("ao2mo_d1", o * v), {o, v}),
                   D2 = view(cx.scratch("ao2mo_d2", o * o * v * v), {o, o, v, v});
            k_slice_phys(cx, voovv.d, packed, (int)o, (int)o, (int)v, (int)v, 0, 0, (int)o, (int)o);
            k_denominators(cx, D1.d, D2.d, e_dev, (int)o, (int)v);
            k_mp2_energy(cx, cx.scal, voovv.d, D2.d, (int)o, (int)v);
            emp2 = host_scalars(cx, 1)[0];
        }
        if (e_mp2) *e_mp2 = emp2;
        if (eri_mo_packed) {
            AFESP_HIP(hipMemcpyAsync(eri_mo_packed, packed, sizeof(double) * ne, hipMemcpyDeviceToHost, cx.stream));
            cx.sync();
        }
    });
}

int afesp_ccsd_init(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, const double* eri_mo_packed, const double* canon_levels,
                    int diis_n_errmat)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        const int64_t n = nocc + nvirt;
        if (nocc <= 0 || nvirt <= 0 || n > 1024) throw Error(1, "afesp_ccsd_init: bad extents");
        const double* src = ctx->eri_mo_dev;
        double* tmp = nullptr;
        if (eri_mo_packed) {
            // (a same-shape state is about to be initialised where it lies: the packed integrals it kept for ccsd_need_vvvv go back
            // to the arena BEFORE their successor is asked for -- a geometry scan never holds two packed arrays)
            if (ccsd_can_reinit(ctx->cc, (int)nocc, (int)nvirt, diis_n_errmat) && ctx->cc.eri_own) {
                cx.quiesce();
                if (ctx->cc.eri_src == ctx->cc.eri_own) ctx->cc.eri_src = nullptr;
                cx.release(ctx->cc.eri_own);
                ctx->cc.eri_own = nullptr;
            }
            tmp = cx.alloc(neri_of(n));
            AFESP_HIP(hipMemcpyAsync(tmp, eri_mo_packed, sizeof(double) * neri_of(n), hipMemcpyHostToDevice, cx.stream));
            src = tmp;
        } else if (!src || ctx->eri_mo_n != n) {
            throw Error(1, "afesp_ccsd_init: no MO integrals resident for this basis size (call afesp_ao2mo_mp2 first)");
        }
        // (a state of the same extents is initialised again where it lies: its compiled programs stay)
        if (!ccsd_can_reinit(ctx->cc, (int)nocc, (int)nvirt, diis_n_errmat)) {
            ctx->cc_programs_reset();
            cx.drop_scratch("ao2mo_");   // the AO->MO temporaries
        } else {
            ctx->graph_cc.reset();
        }
        ccsd_init(cx, ctx->cc, (int)nocc, (int)nvirt, src, canon_levels, diis_n_errmat);
        if (tmp) {
            // a large system forms <ef|ab> on request only (ccsd_need_vvvv): its state keeps the device copy of the integrals
            if (ctx->cc.v_vvvv.d) { cx.release(tmp); ctx->cc.eri_src = nullptr; }
            else ctx->cc.eri_own = tmp;
        }
    });
}

int afesp_ccsd_energy(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_energy: call afesp_ccsd_init first");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        int conv = ccsd_energy(ctx->cx, ctx->cc, e_tol, t_tol);
        if (energy) *energy = ctx->cc.energy;
        if (rms_sq) *rms_sq = ctx->cc.rms;
        if (converged) *converged = conv;
    });
}

int afesp_ccsd_update_intermediates(afesp_ctx* ctx)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready) throw Error(1, "call afesp_ccsd_init first");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_refresh_sharding(ctx->cx, ctx->cc);
        if (!(ccsd_uses_lanes(ctx->cc) && fused_exec(ctx->cx, ctx->fused_int, [&] { ccsd_intermediates(ctx->cx, ctx->cc); })))
            ccsd_intermediates(ctx->cx, ctx->cc);
        ctx->cx.sync();
    });
}
int afesp_ccsd_update_amplitudes(afesp_ctx* ctx)
{
    return guarded(ctx, [&] {
        ctx->cc.amp_epoch = ++ctx->cx.amp_clock;   // (the amplitudes may change: derived copies go stale)
        if (!ctx->cc.ready) throw Error(1, "call afesp_ccsd_init first");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_refresh_sharding(ctx->cx, ctx->cc);
        ctx->cc.tail_pending = false;
        if (!(ccsd_uses_lanes(ctx->cc) && fused_exec(ctx->cx, ctx->fused_amp, [&] { ccsd_amplitudes(ctx->cx, ctx->cc); })))
            ccsd_amplitudes(ctx->cx, ctx->cc);
        ctx->cx.sync();
    });
}

int afesp_ccsd_iterate(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged)
{
    return guarded(ctx, [&] {
        ctx->cc.amp_epoch = ++ctx->cx.amp_clock;   // (the amplitudes may change: derived copies go stale)
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_iterate: call afesp_ccsd_init first");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_refresh_sharding(ctx->cx, ctx->cc);
        const bool fused = ccsd_iteration_body(ctx);
        int conv = fused ? ccsd_tail_read(ctx->cx, ctx->cc, e_tol, t_tol) : ccsd_energy_read(ctx->cx, ctx->cc, e_tol, t_tol);
        if (energy) *energy = ctx->cc.energy;
        if (rms_sq) *rms_sq = ctx->cc.rms;
        if (converged) *converged = conv;
    });
}

int afesp_ccsd_diis(afesp_ctx* ctx)
{
    return guarded(ctx, [&] {
        ctx->cc.amp_epoch = ++ctx->cx.amp_clock;   // (the amplitudes may change: derived copies go stale)
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_diis: call afesp_ccsd_init first");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_diis_update(ctx->cx, ctx->cc);
    });
}

int afesp_ccsd_solve(afesp_ctx* ctx, int maxiter, double e_tol, double t_tol, double* iter_energy, double* iter_rms_sq, int* niter)
{
    return guarded(ctx, [&] {
        ctx->cc.amp_epoch = ++ctx->cx.amp_clock;   // (the amplitudes may change: derived copies go stale)
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_solve: call afesp_ccsd_init first");
        Context& cx = ctx->cx;
        CCState& s = ctx->cc;
        AFESP_HIP(hipSetDevice(cx.device));
        // ccsd.f90:314-315, :325
        s.energy = s.energy_old = 0.0;
        k_fill(cx, s.t2_old.d, s.t2_old.size(), 0.0);
        ccsd_energy(cx, s, e_tol, t_tol);
        if (iter_energy) iter_energy[0] = s.energy;
        if (iter_rms_sq) iter_rms_sq[0] = s.rms;
        int result = -1;
        ccsd_refresh_sharding(cx, s);
        for (int it = 1; it <= maxiter; ++it) {
            const bool fused = ccsd_iteration_body(ctx);
            int conv = fused ? ccsd_tail_read(cx, s, e_tol, t_tol) : ccsd_energy_read(cx, s, e_tol, t_tol);
            if (iter_energy) iter_energy[it] = s.energy;
            if (iter_rms_sq) iter_rms_sq[it] = s.rms;
            if (conv) {
                result = it;
                break;
            }
            ccsd_diis_update(cx, s);
        }
        if (result < 0 && maxiter > 0) diis_check_flag(cx, host_scalars(cx, DIIS_FLAG_SLOT + 1));   // a solve that failed after the last energy read
        if (niter) *niter = result;
    });
}

int afesp_ccsd_get_amplitudes(afesp_ctx* ctx, double* t1, double* t2)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_get_amplitudes: no CCSD state");
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (t1) AFESP_HIP(hipMemcpyAsync(t1, ctx->cc.t1.d, sizeof(double) * ctx->cc.t1.size(), hipMemcpyDeviceToHost, cx.stream));
        if (t2) AFESP_HIP(hipMemcpyAsync(t2, ctx->cc.t2.d, sizeof(double) * ctx->cc.t2.size(), hipMemcpyDeviceToHost, cx.stream));
        diis_check_flag(cx, host_scalars(cx, DIIS_FLAG_SLOT + 1));   // afesp_ccsd_diis does not wait for its solve: a failure surfaces here at the latest
    });
}

int afesp_ccsd_set_amplitudes(afesp_ctx* ctx, const double* t1, const double* t2)
{
    return guarded(ctx, [&] {
        ctx->cc.amp_epoch = ++ctx->cx.amp_clock;   // (the amplitudes may change: derived copies go stale)
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_set_amplitudes: no CCSD state");
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        // (a large system holds I_ovov / I_voov and copies of the OLD amplitudes in the layout of its ring launches (ring.hip): an
        // afesp_ccsd_update_amplitudes that follows without new intermediates reads the reference-layout tensors and the new amplitudes)
        if (ring_live(ctx->cc)) {
            ring_tg_materialize(cx, ctx->cc, ctx->cc.I_ovov, ctx->cc.I_voov);
            ring_invalidate(ctx->cc);
        }
        ctx->cc.amps_touched = true;
        if (t2) ctx->cc.hist_plain = ctx->cc.nerr + 1;   // (its error vector may lack the amplitudes' symmetry: full DIIS sums until it has left the history)
        if (t1) AFESP_HIP(hipMemcpyAsync(ctx->cc.t1.d, t1, sizeof(double) * ctx->cc.t1.size(), hipMemcpyHostToDevice, cx.stream));
        if (t2) AFESP_HIP(hipMemcpyAsync(ctx->cc.t2.d, t2, sizeof(double) * ctx->cc.t2.size(), hipMemcpyHostToDevice, cx.stream));
        cx.sync();
    });
}

int afesp_ccsd_get_tensor(afesp_ctx* ctx, const char* name, double* out, int64_t capacity)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_get_tensor: no CCSD state");
        CCState& s = ctx->cc;
        struct { const char* n; const Tensor* t; } tab[] = {
            {"v_oovv", &s.v_oovv}, {"v_ovov", &s.v_ovov}, {"v_vvov", &s.v_vvov}, {"v_oovo", &s.v_oovo}, {"v_oooo", &s.v_oooo},
            {"v_vvvv", &s.v_vvvv}, {"I_vo", &s.I_vo}, {"I_vv", &s.I_vv}, {"I_oo_p", &s.I_oo_p}, {"I_oo", &s.I_oo},
            {"c_oovv", &s.c}, {"asym_t2", &s.asym}, {"x_voov", &s.x_voov}, {"I_oooo", &s.I_oooo}, {"I_ovov", &s.I_ovov},
            {"I_voov", &s.I_voov}, {"I_ooov_p", &s.I_ooov_p}, {"r1", &s.r1}, {"r2", &s.r2},
            {"D1", &s.D1}, {"D2", &s.D2}, {"t1", &s.t1}, {"t2", &s.t2}};
        if (!strcmp(name, "I_vovv_p")) {   // not formed by the iteration (ccsd.hip): built from the current t1 on request
            Context& cx = ctx->cx;
            AFESP_HIP(hipSetDevice(cx.device));
            const int64_t O = s.o, V = s.v;
            if (V * O * V * V > capacity) throw Error(1, "afesp_ccsd_get_tensor: buffer too small for I_vovv_p");
            Tensor t = view(cx.scratch("I_vovv_p", V * O * V * V), {V, O, V, V});
            ccsd_build_I_vovv_p(cx, s, t);
            AFESP_HIP(hipMemcpyAsync(out, t.d, sizeof(double) * t.size(), hipMemcpyDeviceToHost, cx.stream));
            cx.sync();
            return;
        }
        if (!strcmp(name, "v_vvvv")) {
            AFESP_HIP(hipSetDevice(ctx->cx.device));
            ccsd_need_vvvv(ctx->cx, s);
        }
        for (auto& e : tab)
            if (!strcmp(e.n, name)) {
                if (e.t->size() > capacity) throw Error(1, std::string("afesp_ccsd_get_tensor: buffer too small for ") + name);
                AFESP_HIP(hipSetDevice(ctx->cx.device));
                const double* src = e.t->d;
                // the residuals of a laned iteration lie in partial buffers that the update kernel adds up (ccsd_amplitudes)
                // (only what the LAST amplitudes call left there: a launch-fused or large-system call after a laned one has none)
                auto add_partial = [&](double* dst, const char* buf) {
                    auto it = ctx->cx.cache.find(buf);
                    if (s.partials_live && it != ctx->cx.cache.end()) k_axpby(ctx->cx, dst, 1.0, (const double*)it->second.first, 1.0, e.t->size());
                };
                if (!strcmp(name, "r2")) {   // the reference's tmp_t2 before P(ia/jb) includes 1/2 pp; it is kept packed here
                    double* full = ctx->cx.scratch("r2_full", e.t->size());
                    k_r2_full(ctx->cx, full, s.r2.d, s.pp, s.o, s.v);
                    add_partial(full, "r2_lane2");
                    add_partial(full, "r2_lane3");
                    if (ring_res_live(s)) k_add_swapped(ctx->cx, full, ring_Y(s), s.o, s.v);   // a ring term of a large system's residual (ring.hip)
                    src = full;
                } else if (ring_live(s) && (!strcmp(name, "I_ovov") || !strcmp(name, "I_voov"))) {
                    // a large system's iteration holds these two in the layout its ring products read (ring.hip): turned back on request
                    const int64_t O = s.o, V = s.v;
                    Tensor io = view(ctx->cx.scratch("ring_I_ovov", e.t->size()), {O, V, O, V}), iv = view(ctx->cx.scratch("ring_I_voov", e.t->size()), {V, O, O, V});
                    ring_tg_materialize(ctx->cx, s, io, iv);
                    src = !strcmp(name, "I_ovov") ? io.d : iv.d;
                } else if (!strcmp(name, "r1")) {
                    double* full = ctx->cx.scratch("r1_full", e.t->size());
                    k_copy(ctx->cx, full, s.r1.d, e.t->size());
                    add_partial(full, "r1_lane5");
                    src = full;
                }
                AFESP_HIP(hipMemcpyAsync(out, src, sizeof(double) * e.t->size(), hipMemcpyDeviceToHost, ctx->cx.stream));
                ctx->cx.sync();
                return;
            }
        throw Error(1, std::string("afesp_ccsd_get_tensor: unknown tensor ") + name);
    });
}

int64_t afesp_ccsd_t_ntriples(int64_t nocc) { return triples_count((int)nocc); }

int afesp_ccsd_t_shard_bounds(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, int cr, int world, int64_t* bounds)
{
    return guarded(ctx, [&] {
        if (nocc < 1 || nvirt < 1 || world < 1 || !bounds) throw Error(1, "afesp_ccsd_t_shard_bounds: bad arguments");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        triples_shard_bounds((int)nocc, (int)nvirt, cr != 0, world, bounds);
    });
}

int afesp_ccsd_t(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double out[4])
{
    return guarded(ctx, [&] {
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_triples(ctx->cx, ctx->cc, t_begin, t_end, out);
    });
}

int afesp_ccsd_t_plain(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double out[2])
{
    return guarded(ctx, [&] {
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_triples(ctx->cx, ctx->cc, t_begin, t_end, out, false, false);
    });
}

int afesp_ccsd_cr_intermediates(afesp_ctx* ctx)
{
    return guarded(ctx, [&] {
        ctx->cc.cr_epoch = ++ctx->cx.amp_clock;
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_cr_intermediates(ctx->cx, ctx->cc);
        ctx->cx.sync();
    });
}

int afesp_ccsd_t_cr(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double out[6])
{
    return guarded(ctx, [&] {
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ccsd_triples(ctx->cx, ctx->cc, t_begin, t_end, out, true);
    });
}

// ---------------------------------------------------------------- input / output side of the path
// read_integrals_in, two-body part (src/integrals.f90:146-161): lines "i j a b value", 1-based, any blank separation, in
// any order; a later line for the same packed slot overwrites an earlier one; slots never mentioned stay 0.
int afesp_read_eri_text(afesp_ctx* ctx, const char* path, int64_t nbasis, double* eri_packed, int64_t* nread)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (nbasis <= 0 || nbasis > 1024 || !path) throw Error(1, "afesp_read_eri_text: bad arguments");
        const int64_t ne = neri_of(nbasis);
        FILE* f = fopen(path, "rb");
        if (!f) throw Error(2, std::string("afesp_read_eri_text: cannot open ") + path);
        std::vector<double> host((size_t)ne, 0.0);
        std::vector<char> buf((size_t)(8 << 20) + 1);
        size_t keep = 0;
        int64_t lines = 0;
        bool bad = false;
        auto tri = [](int64_t i, int64_t j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; };
        for (;;) {
            const size_t got = fread(buf.data() + keep, 1, buf.size() - 1 - keep, f);
            const size_t have = keep + got;
            if (have == 0) break;
            buf[have] = 0;
            // parse whole lines only; the tail (an incomplete line) is carried into the next block
            size_t end = have;
            if (got > 0) {
                while (end > 0 && buf[end - 1] != '\n') --end;
                if (end == 0 && have == buf.size() - 1) { bad = true; break; }   // a "line" longer than the buffer
                if (end == 0) end = 0;
            }
            const size_t stop = got > 0 ? end : have;
            char* p = buf.data();
            char* const lim = buf.data() + stop;
            const char saved = *lim;
            *lim = 0;
            while (p < lim) {
                while (p < lim && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) ++p;
                if (p >= lim) break;
                // list-directed input (src/integrals.f90:150 `read (ir, *) i, j, a, b, val`): fields are separated by blanks
                // and/or one comma, a real may carry a Fortran D exponent ("1.0D-05"); whatever follows the fifth field on
                // the record is ignored, as the reference's read does
                auto sep = [&](char*& c) {
                    while (c < lim && (*c == ' ' || *c == '\t' || *c == '\r')) ++c;
                    if (c < lim && *c == ',') ++c;
                    while (c < lim && (*c == ' ' || *c == '\t' || *c == '\r')) ++c;
                };
                char* q;
                long idx[4];
                bool ok = true;
                for (int k = 0; k < 4 && ok; ++k) {
                    if (*p == '\n') { ok = false; break; }
                    idx[k] = strtol(p, &q, 10);
                    ok = (q != p) && (q >= lim || *q == ' ' || *q == '\t' || *q == ',' || *q == '\r');
                    p = q;
                    if (ok) sep(p);
                }
                double val = 0.0;
                if (ok && *p != '\n') {
                    char tok[64];
                    size_t len = 0;
                    while (p + len < lim && len < sizeof(tok) - 1 && p[len] != ' ' && p[len] != '\t' && p[len] != ',' && p[len] != '\r' &&
                           p[len] != '\n') {
                        const char ch = p[len];
                        tok[len] = (ch == 'D' || ch == 'd') ? 'E' : ch;
                        ++len;
                    }
                    tok[len] = 0;
                    char* tq = nullptr;
                    val = strtod(tok, &tq);
                    ok = len > 0 && tq == tok + len;   // the whole token is the number
                    p += len;
                } else {
                    ok = false;
                }
                if (!ok) { bad = true; break; }
                for (int k = 0; k < 4; ++k)
                    if (idx[k] < 1 || idx[k] > nbasis) ok = false;
                if (!ok) { bad = true; break; }
                host[(size_t)tri(tri(idx[0] - 1, idx[1] - 1), tri(idx[2] - 1, idx[3] - 1))] = val;
                ++lines;
                while (p < lim && *p != '\n') ++p;   // ignore anything else on the line
            }
            *lim = saved;
            if (bad || got == 0) break;
            keep = have - stop;
            memmove(buf.data(), buf.data() + stop, keep);
        }
        fclose(f);
        if (bad) throw Error(2, std::string("afesp_read_eri_text: malformed line or index outside 1..nbasis in ") + path);
        if (ctx->eri_ao_dev) cx.release(ctx->eri_ao_dev);
        ctx->eri_ao_dev = cx.alloc(ne);
        ctx->eri_ao_n = nbasis;
        ctx->half_n = 0;
        AFESP_HIP(hipMemcpyAsync(ctx->eri_ao_dev, host.data(), sizeof(double) * ne, hipMemcpyHostToDevice, cx.stream));
        cx.sync();
        if (eri_packed) memcpy(eri_packed, host.data(), sizeof(double) * ne);
        if (nread) *nread = lines;
    });
}

// Packed AO integrals from a host array (for callers that already hold int_store%eri), same residency as the reader's.
int afesp_set_eri(afesp_ctx* ctx, int64_t nbasis, const double* eri_packed)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (nbasis <= 0 || nbasis > 1024 || !eri_packed) throw Error(1, "afesp_set_eri: bad arguments");
        const int64_t ne = neri_of(nbasis);
        if (ctx->eri_ao_dev) cx.release(ctx->eri_ao_dev);
        ctx->eri_ao_dev = cx.alloc(ne);
        ctx->eri_ao_n = nbasis;
        ctx->half_n = 0;
        AFESP_HIP(hipMemcpyAsync(ctx->eri_ao_dev, eri_packed, sizeof(double) * ne, hipMemcpyHostToDevice, cx.stream));
        cx.sync();
    });
}

// build_fock (src/hf.f90:349-385) on the resident packed AO integrals
int afesp_build_fock(afesp_ctx* ctx, int64_t nbasis, const double* density, const double* core_hamil, double* fock)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (!ctx->eri_ao_dev || ctx->eri_ao_n != nbasis || !density || !core_hamil || !fock)
            throw Error(1, "afesp_build_fock: no AO integrals resident for this basis size (afesp_read_eri_text / afesp_set_eri)");
        const int64_t n2 = nbasis * nbasis;
        double* buf = cx.scratch("fock_io", 3 * n2);
        AFESP_HIP(hipMemcpyAsync(buf, density, sizeof(double) * n2, hipMemcpyHostToDevice, cx.stream));
        AFESP_HIP(hipMemcpyAsync(buf + n2, core_hamil, sizeof(double) * n2, hipMemcpyHostToDevice, cx.stream));
        // the half-unpacked integrals (ij|KL) live in the scratch buffer the AO->MO transform starts from ("ao2mo_a"): built on
        // the first Fock build of an SCF, reused by every later one and by afesp_ao2mo_mp2
        const int64_t np = nbasis * (nbasis + 1) / 2, L = ao2mo_ld(nbasis);   // (columns as afesp_ao2mo_mp2 will want them)
        double* u = cx.scratch("ao2mo_a", L * nbasis * np + 16);   // (+16: the size afesp_ao2mo_mp2 asks for, so that it finds this very buffer)
        if (ctx->half_n != nbasis || ctx->half_ld != L || ctx->half_epoch != cx.scratch_epoch) {
            if (ctx->pad_n != nbasis || ctx->pad_ld != L) ctx->pad_n = 0;   // (another layout lands in the buffer the transforms share)
            k_unpack_half(cx, u, ctx->eri_ao_dev, (int)nbasis, 0, -1, (int)L);
            ctx->half_n = nbasis;
            ctx->half_ld = L;
            ctx->half_epoch = cx.scratch_epoch;
        }
        double* work = cx.scratch("fock_work", k_build_fock_work((int)nbasis));
        ctx->half_epoch = cx.scratch_epoch;   // (growing fock_work moves the epoch, not u)
        k_build_fock(cx, buf + 2 * n2, buf + n2, buf, u, work, (int)nbasis, (int)L);
        AFESP_HIP(hipMemcpyAsync(fock, buf + 2 * n2, sizeof(double) * n2, hipMemcpyDeviceToHost, cx.stream));
        cx.sync();
    });
}

// write_fcidump (src/mp2.f90:451-487): the packed MO integrals in canonical order, one line "p q r s value" in format
// (I3,I3,I3,I3,ES17.9) for every |value| > 1e-7 (no header, no one-electron part -- as the reference writes it).
int afesp_write_fcidump(afesp_ctx* ctx, const char* path, int64_t nbasis, int64_t* nwritten)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (!ctx->eri_mo_dev || ctx->eri_mo_n != nbasis || !path)
            throw Error(1, "afesp_write_fcidump: no MO integrals resident for this basis size (call afesp_ao2mo_mp2 first)");
        const int64_t ne = neri_of(nbasis);
        std::vector<double> host((size_t)ne);
        AFESP_HIP(hipMemcpyAsync(host.data(), ctx->eri_mo_dev, sizeof(double) * ne, hipMemcpyDeviceToHost, cx.stream));
        cx.sync();
        FILE* f = fopen(path, "w");
        if (!f) throw Error(2, std::string("afesp_write_fcidump: cannot open ") + path);
        int64_t pqrs = 0, lines = 0;
        for (int64_t p = 1; p <= nbasis; ++p)
            for (int64_t q = 1; q <= p; ++q)
                for (int64_t r = 1; r <= p; ++r) {
                    const int64_t s_up = (p == r) ? q : r;
                    for (int64_t s = 1; s <= s_up; ++s) {
                        const double x = host[(size_t)pqrs++];
                        if (std::fabs(x) > 1e-7) {
                            fprintf(f, "%3d%3d%3d%3d%17.9E\n", (int)p, (int)q, (int)r, (int)s, x);
                            ++lines;
                        }
                    }
                }
        fclose(f);
        if (nwritten) *nwritten = lines;
    });
}

// ---------------------------------------------------------------- spin-orbital path
int afesp_ccsd_so_init(afesp_ctx* ctx, int64_t nbasis, int64_t nel, const double* eri_mo_packed, const double* canon_levels,
                       int diis_n_errmat, int flags)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (nbasis <= 0 || nbasis > 512 || nel <= 0 || nel >= 2 * nbasis) throw Error(1, "afesp_ccsd_so_init: bad extents");
        const double* src = ctx->eri_mo_dev;
        double* tmp = nullptr;
        if (eri_mo_packed) {
            tmp = cx.alloc(neri_of(nbasis));
            AFESP_HIP(hipMemcpyAsync(tmp, eri_mo_packed, sizeof(double) * neri_of(nbasis), hipMemcpyHostToDevice, cx.stream));
            src = tmp;
        } else if (!src || ctx->eri_mo_n != nbasis) {
            throw Error(1, "afesp_ccsd_so_init: no MO integrals resident for this basis size (call afesp_ao2mo_mp2 first)");
        }
        cx.drop_scratch("ao2mo_");   // the AO->MO temporaries
        ctx->so_programs_reset();
        so_init(cx, ctx->so, (int)nbasis, (int)nel, src, canon_levels, diis_n_errmat, (flags & AFESP_SO_FOO_AS_PUBLISHED) != 0);
        ctx->so.amp_epoch = ++cx.amp_clock;
        if (tmp) cx.release(tmp);
    });
}

int afesp_ccsd_so_energy(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged)
{
    return guarded(ctx, [&] {
        if (!ctx->so.ready) throw Error(1, "afesp_ccsd_so_energy: call afesp_ccsd_so_init first");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        int conv = so_energy(ctx->cx, ctx->so, e_tol, t_tol);
        if (energy) *energy = ctx->so.energy;
        if (rms_sq) *rms_sq = ctx->so.rms;
        if (converged) *converged = conv;
    });
}

int afesp_ccsd_so_iterate(afesp_ctx* ctx, double e_tol, double t_tol, double* energy, double* rms_sq, int* converged)
{
    return guarded(ctx, [&] {
        if (!ctx->so.ready) throw Error(1, "afesp_ccsd_so_iterate: call afesp_ccsd_so_init first");
        ctx->so.amp_epoch = ++ctx->cx.amp_clock;   // (the amplitudes may change: derived copies go stale)
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        // (the levelled sequence of fused.h where the system is small enough for its products to be launch-bound: the big ones
        // keep their own kernels inside it)
        auto body = [&] {
            diis_save(ctx->cx, ctx->so);
            so_intermediates(ctx->cx, ctx->so);
            so_amplitudes(ctx->cx, ctx->so);
        };
        if (!(ctx->so.t2.size() <= ((int64_t)1 << 22) && fused_exec(ctx->cx, ctx->fused_so, body))) body();
        int conv = so_energy(ctx->cx, ctx->so, e_tol, t_tol);
        if (energy) *energy = ctx->so.energy;
        if (rms_sq) *rms_sq = ctx->so.rms;
        if (converged) *converged = conv;
    });
}

int afesp_ccsd_so_diis(afesp_ctx* ctx)
{
    return guarded(ctx, [&] {
        if (!ctx->so.ready) throw Error(1, "afesp_ccsd_so_diis: call afesp_ccsd_so_init first");
        ctx->so.amp_epoch = ++ctx->cx.amp_clock;
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        diis_update(ctx->cx, ctx->so);
    });
}

int afesp_ccsd_so_get_amplitudes(afesp_ctx* ctx, double* t1, double* t2)
{
    return guarded(ctx, [&] {
        if (!ctx->so.ready) throw Error(1, "afesp_ccsd_so_get_amplitudes: no spin-orbital CCSD state");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        SOState& s = ctx->so;
        if (t1) AFESP_HIP(hipMemcpyAsync(t1, s.t1.d, sizeof(double) * s.t1.size(), hipMemcpyDeviceToHost, ctx->cx.stream));
        if (t2) AFESP_HIP(hipMemcpyAsync(t2, s.t2.d, sizeof(double) * s.t2.size(), hipMemcpyDeviceToHost, ctx->cx.stream));
        ctx->cx.sync();
    });
}

int afesp_ccsd_so_set_amplitudes(afesp_ctx* ctx, const double* t1, const double* t2)
{
    return guarded(ctx, [&] {
        if (!ctx->so.ready) throw Error(1, "afesp_ccsd_so_set_amplitudes: no spin-orbital CCSD state");
        ctx->so.amp_epoch = ++ctx->cx.amp_clock;
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        SOState& s = ctx->so;
        if (t1) AFESP_HIP(hipMemcpyAsync(s.t1.d, t1, sizeof(double) * s.t1.size(), hipMemcpyHostToDevice, ctx->cx.stream));
        if (t2) AFESP_HIP(hipMemcpyAsync(s.t2.d, t2, sizeof(double) * s.t2.size(), hipMemcpyHostToDevice, ctx->cx.stream));
        ctx->cx.sync();
    });
}

int afesp_ccsd_so_get_tensor(afesp_ctx* ctx, const char* name, double* out, int64_t capacity)
{
    return guarded(ctx, [&] {
        if (!ctx->so.ready) throw Error(1, "afesp_ccsd_so_get_tensor: no spin-orbital CCSD state");
        SOState& s = ctx->so;
        if (!strcmp(name, "W_vvvv")) {   // not formed by the iteration (so_ladder): built from the current t1 on request
            AFESP_HIP(hipSetDevice(ctx->cx.device));
            so_build_W_vvvv(ctx->cx, s);
        }
        struct { const char* n; const Tensor* t; } tab[] = {
            {"F_vv", &s.F_vv}, {"F_oo", &s.F_oo}, {"F_ov", &s.F_ov}, {"W_oooo", &s.W_oooo}, {"W_vvvv", &s.W_vvvv},
            {"W_ovvo", &s.W_ovvo}, {"tau", &s.tau}, {"tau_tilde", &s.tau_t}, {"oovv", &s.oovv}, {"vvvv", &s.vvvv},
            {"t1", &s.t1}, {"t2", &s.t2}};
        for (auto& e : tab)
            if (!strcmp(e.n, name)) {
                if (e.t->size() > capacity) throw Error(1, std::string("afesp_ccsd_so_get_tensor: buffer too small for ") + name);
                AFESP_HIP(hipSetDevice(ctx->cx.device));
                AFESP_HIP(hipMemcpyAsync(out, e.t->d, sizeof(double) * e.t->size(), hipMemcpyDeviceToHost, ctx->cx.stream));
                ctx->cx.sync();
                return;
            }
        throw Error(1, std::string("afesp_ccsd_so_get_tensor: unknown tensor ") + name);
    });
}

int64_t afesp_ccsd_so_t_ntriples(int64_t nocc) { return so_triples_count((int)nocc); }

int afesp_ccsd_so_t(afesp_ctx* ctx, int64_t t_begin, int64_t t_end, double* e_t)
{
    return guarded(ctx, [&] {
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        const double e = so_triples(ctx->cx, ctx->so, t_begin, t_end);
        if (e_t) *e_t = e;
    });
}

// ---------------------------------------------------------------- operator layer on host arrays
int afesp_contract(afesp_ctx* ctx, double alpha, const double* A, const char* la, const int64_t* dimsA, const double* B,
                   const char* lb, const int64_t* dimsB, double beta, double* C, const char* lc, const int64_t* dimsC,
                   int force_split, int force_tm, int force_tn)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        auto mk = [&](const char* l, const int64_t* dims) {
            Tensor t;
            t.rank = (int)strlen(l);
            if (t.rank > 6) throw Error(1, "afesp_contract: rank > 6");
            int64_t s = 1;
            for (int i = 0; i < t.rank; ++i) {
                t.dim[i] = dims[i];
                t.stride[i] = s;
                s *= dims[i];
            }
            t.d = cx.alloc(s);
            return t;
        };
        Tensor tA = mk(la, dimsA), tB = mk(lb, dimsB), tC = mk(lc, dimsC);
        AFESP_HIP(hipMemcpyAsync(tA.d, A, sizeof(double) * tA.size(), hipMemcpyHostToDevice, cx.stream));
        AFESP_HIP(hipMemcpyAsync(tB.d, B, sizeof(double) * tB.size(), hipMemcpyHostToDevice, cx.stream));
        AFESP_HIP(hipMemcpyAsync(tC.d, C, sizeof(double) * tC.size(), hipMemcpyHostToDevice, cx.stream));
        contract(cx, alpha, tA, la, tB, lb, beta, tC, lc, 1, nullptr, nullptr, nullptr, force_split, force_tm, force_tn);
        AFESP_HIP(hipMemcpyAsync(C, tC.d, sizeof(double) * tC.size(), hipMemcpyDeviceToHost, cx.stream));
        cx.sync();
        cx.release(tA.d); cx.release(tB.d); cx.release(tC.d);
    });
}

int afesp_gemm(afesp_ctx* ctx, char transA, char transB, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
               const double* B, double beta, double* C)
{
    // dgemm_wrapper (linalg.fpp:58-89): leading dimensions follow from the logical shapes
    const bool ta = (transA == 'T' || transA == 't'), tb = (transB == 'T' || transB == 't');
    int64_t dA[2] = {ta ? k : m, ta ? m : k}, dB[2] = {tb ? n : k, tb ? k : n}, dC[2] = {m, n};
    return afesp_contract(ctx, alpha, A, ta ? "km" : "mk", dA, B, tb ? "nk" : "kn", dB, beta, C, "mn", dC, 0, 0, 0);
}

int afesp_permute4(afesp_ctx* ctx, const int64_t dims[4], const char order[4], const double* in, double* out, int has_beta, double beta)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        // omp_reshape (linalg.fpp:136-147): character d of `order` names the input index in output position d
        const char names[5] = "ijkl";
        char lo[5] = {0, 0, 0, 0, 0};
        int64_t od[4];
        for (int d = 0; d < 4; ++d) {
            int p = order[d] - '1';
            if (p < 0 || p > 3) throw Error(1, "afesp_permute4: bad order string");
            lo[d] = names[p];
            od[d] = dims[p];
        }
        Tensor tin = cx.tensor({dims[0], dims[1], dims[2], dims[3]}), tout = cx.tensor({od[0], od[1], od[2], od[3]});
        AFESP_HIP(hipMemcpyAsync(tin.d, in, sizeof(double) * tin.size(), hipMemcpyHostToDevice, cx.stream));
        if (has_beta) AFESP_HIP(hipMemcpyAsync(tout.d, out, sizeof(double) * tout.size(), hipMemcpyHostToDevice, cx.stream));
        permute_add(cx, 1.0, tin, names, has_beta ? beta : 0.0, tout, lo);
        AFESP_HIP(hipMemcpyAsync(out, tout.d, sizeof(double) * tout.size(), hipMemcpyDeviceToHost, cx.stream));
        cx.sync();
        cx.release(tin.d); cx.release(tout.d);
    });
}

// ---------------------------------------------------------------- synthetic inputs generated in HBM
int afesp_synthetic_init(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, double scale, uint64_t seed, int diis_n_errmat)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        const int64_t n = nocc + nvirt, ne = neri_of(n);
        if (nocc <= 0 || nvirt <= 0 || n > 1024) throw Error(1, "afesp_synthetic_init: bad extents");
        std::vector<double> e((size_t)n);
        for (int64_t i = 0; i < nocc; ++i) e[i] = -2.0 + (nocc > 1 ? (double)i / (double)(nocc - 1) : 0.0);
        for (int64_t a = 0; a < nvirt; ++a) e[nocc + a] = 1.0 + (nvirt > 1 ? 2.0 * (double)a / (double)(nvirt - 1) : 0.0);
        double* packed = cx.alloc(ne);
        AFESP_KLAUNCH(synth_packed_kernel, dim3(4096), dim3(256), 0, cx.stream, packed, ne, scale, seed);
        AFESP_HIP(hipGetLastError());
        if (!ccsd_can_reinit(ctx->cc, (int)nocc, (int)nvirt, diis_n_errmat)) ctx->cc_programs_reset();
        else ctx->graph_cc.reset();
        ccsd_init(cx, ctx->cc, (int)nocc, (int)nvirt, packed, e.data(), diis_n_errmat);
        if (ctx->cc.v_vvvv.d) { cx.release(packed); ctx->cc.eri_src = nullptr; }
        else ctx->cc.eri_own = packed;   // kept for ccsd_need_vvvv
    });
}

// Hashed packed AO integrals left on the device as afesp_read_eri_text would leave them (AO->MO timing: bench.py)
int afesp_synthetic_ao(afesp_ctx* ctx, int64_t nbasis, double scale, uint64_t seed)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (nbasis <= 0 || nbasis > 1024) throw Error(1, "afesp_synthetic_ao: bad extents");
        const int64_t ne = neri_of(nbasis);
        if (ctx->eri_ao_dev) cx.release(ctx->eri_ao_dev);
        ctx->eri_ao_dev = cx.alloc(ne);
        ctx->eri_ao_n = nbasis;
        ctx->half_n = 0;
        AFESP_KLAUNCH(synth_packed_kernel, dim3(4096), dim3(256), 0, cx.stream, ctx->eri_ao_dev, ne, scale, seed);
        AFESP_HIP(hipGetLastError());
        cx.sync();
    });
}

// Floating-point operations of one particle-particle ladder as this context evaluates it (plain a <= b form or the
// symmetric/antisymmetric pair form, ccsd.hip)
int afesp_ccsd_pp_ladder_flop(afesp_ctx* ctx, double* flop)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_pp_ladder_flop: no CCSD state");
        const double O = ctx->cc.o, V = ctx->cc.v, ps = V * (V + 1) / 2, pa = V * (V - 1) / 2;
        if (flop) *flop = ctx->cc.pp_sym ? 2.0 * (O * (O + 1) / 2 * ps * ps + O * (O - 1) / 2 * pa * pa) : 2.0 * O * O * V * V * ps;
    });
}

// Floating-point operations of one CCSD iteration as this context evaluates it: SURVEY.md 8(d)'s sum over the contraction sites,
// with the pp-ladder and the t2 x <ef|ia> product counted in the form they are executed (plain, a <= b, or over pair indices)
int afesp_ccsd_iteration_flop(afesp_ctx* ctx, double* flop)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready) throw Error(1, "afesp_ccsd_iteration_flop: no CCSD state");
        const double O = ctx->cc.o, V = ctx->cc.v, ps = V * (V + 1) / 2, pa = V * (V - 1) / 2, os = O * (O + 1) / 2, oa = O * (O - 1) / 2;
        const bool sym = ctx->cc.pp_sym;
        const double pp = sym ? 2.0 * (os * ps * ps + oa * pa * pa) : 2.0 * O * O * V * V * ps;
        const double ooov = sym ? 2.0 * O * V * (os * ps + oa * pa) : 2.0 * O * O * O * V * V * V;
        const double o3v3 = O * O * O * V * V * V;
        // large-system path (round 5): c <ij|ef> -> I_oooo and the hole-hole ladder over pair indices (the latter inside the pp-ladder's
        // products), and the bare t(i,e) <ab|ej> term as a copy of x_voov instead of a third o^2 v^3 product
        const bool large = !ccsd_uses_lanes(ctx->cc);
        const double oooo = (sym && large) ? 4.0 * (os * os * ps + oa * oa * pa) : 2.0 * O * O * O * O * V * V;
        // (... and asym(m,i,e,f) <ef|ma> -> r1 as a trace of the pair-form t2 <ef|ia> product: one more o^2 v^3 product that is not executed)
        const double o2v3 = large ? (sym ? 14.0 : 16.0) : 18.0;
        if (flop)
            *flop = pp + ooov + 12.0 * o3v3 + oooo + 2.0 * O * O * O * O * V + o2v3 * O * O * V * V * V +
                    2.0 * O * V * V * V + 14.0 * O * O * O * V * V;
    });
}

int afesp_bench_stream(afesp_ctx* ctx, int64_t n, int reps, double* ms_per_launch)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        double* x = cx.scratch("bench_x", n);
        double* y = cx.scratch("bench_y", n);
        k_fill(cx, x, n, 1.0);
        k_fill(cx, y, n, 2.0);
        hipEvent_t a, b;
        AFESP_HIP(hipEventCreate(&a));
        AFESP_HIP(hipEventCreate(&b));
        k_axpby(cx, y, 0.5, x, 0.25, n);
        AFESP_HIP(hipEventRecord(a, cx.stream));
        for (int r = 0; r < reps; ++r) k_axpby(cx, y, 0.5, x, 0.25, n);
        AFESP_HIP(hipEventRecord(b, cx.stream));
        AFESP_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        AFESP_HIP(hipEventElapsedTime(&ms, a, b));
        if (ms_per_launch) *ms_per_launch = (double)ms / (reps > 0 ? reps : 1);
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    });
}

int afesp_profile(afesp_ctx* ctx, int enable, double out[8])
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        if (out) {
            out[0] = cx.prof_gemm_ms; out[1] = (double)cx.prof_gemm_launches; out[2] = cx.prof_gemm_flop;
            out[3] = cx.prof_orbit_ms; out[4] = (double)cx.prof_orbit_launches; out[5] = cx.prof_orbit_bytes;
            out[6] = cx.prof_gemm_flop_padded; out[7] = (double)cx.prof_gemm_kind;
        }
        cx.prof = enable != 0;
        cx.prof_gemm_ms = cx.prof_gemm_flop = cx.prof_gemm_flop_padded = cx.prof_orbit_ms = cx.prof_orbit_bytes = 0.0;
        cx.prof_gemm_launches = cx.prof_orbit_launches = 0;
    });
}

int afesp_set_tuning(int group_m, int force_tm, int force_tn, int force_split)
{
    g_allow_wide = !(group_m & 0x10000); g_dbg = (group_m >> 17) & 7; group_m &= 0xffff;
    g_group_m = group_m; g_force_tm = force_tm; g_force_tn = force_tn; g_force_split = force_split;
    return 0;
}

// Device-only timing of one labelled contraction on hashed operands (tuning / roofline measurements).
int afesp_bench_contract(afesp_ctx* ctx, const char* la, const int64_t* dimsA, const char* lb, const int64_t* dimsB,
                         const char* lc, const int64_t* dimsC, int reps, double* ms_per_launch)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        auto mk = [&](const char* l, const int64_t* dims, uint64_t seed) {
            Tensor t;
            t.rank = (int)strlen(l);
            int64_t s = 1;
            for (int i = 0; i < t.rank; ++i) { t.dim[i] = dims[i]; t.stride[i] = s; s *= dims[i]; }
            t.d = cx.alloc(s);
            AFESP_KLAUNCH(synth_packed_kernel, dim3(4096), dim3(256), 0, cx.stream, t.d, s, 1.0, seed);
            return t;
        };
        Tensor tA = mk(la, dimsA, 1), tB = mk(lb, dimsB, 2), tC = mk(lc, dimsC, 3);
        hipEvent_t a, b;
        AFESP_HIP(hipEventCreate(&a));
        AFESP_HIP(hipEventCreate(&b));
        contract(cx, 1.0, tA, la, tB, lb, 0.0, tC, lc);
        AFESP_HIP(hipEventRecord(a, cx.stream));
        for (int r = 0; r < reps; ++r) contract(cx, 1.0, tA, la, tB, lb, 0.0, tC, lc);
        AFESP_HIP(hipEventRecord(b, cx.stream));
        AFESP_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        AFESP_HIP(hipEventElapsedTime(&ms, a, b));
        if (ms_per_launch) *ms_per_launch = (double)ms / (reps > 0 ? reps : 1);
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        cx.release(tA.d); cx.release(tB.d); cx.release(tC.d);
    });
}

int afesp_time_pp_ladder(afesp_ctx* ctx, int reps, double* ms_per_launch)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready) throw Error(1, "afesp_time_pp_ladder: no CCSD state");
        Context& cx = ctx->cx;
        CCState& s = ctx->cc;
        AFESP_HIP(hipSetDevice(cx.device));
        hipEvent_t a, b;
        AFESP_HIP(hipEventCreate(&a));
        AFESP_HIP(hipEventCreate(&b));
        ccsd_pp_ladder(cx, s);   // warm
        AFESP_HIP(hipEventRecord(a, cx.stream));
        for (int r = 0; r < reps; ++r) ccsd_pp_ladder(cx, s);
        AFESP_HIP(hipEventRecord(b, cx.stream));
        AFESP_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        AFESP_HIP(hipEventElapsedTime(&ms, a, b));
        if (ms_per_launch) *ms_per_launch = (double)ms / (reps > 0 ? reps : 1);
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
    });
}

// ---------------------------------------------------------------- multi-GPU: the sum over ranks (comm.h)
int afesp_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int afesp_comm_unique_id(char id_out[128])
{
    if (!id_out) return 1;
    try {
        comm_unique_id(id_out);
        return 0;
    } catch (const Error& e) {
        return e.code ? e.code : 1;
    } catch (...) {
        return 1;
    }
}

int afesp_comm_init(afesp_ctx* ctx, int rank, int world, int transport, const char* bootstrap_path, const char* unique_id)
{
    return guarded(ctx, [&] {
        Context& cx = ctx->cx;
        AFESP_HIP(hipSetDevice(cx.device));
        if (cx.comm) throw Error(1, "afesp_comm_init: this context already has a communicator");
        cx.comm = comm_create(cx, rank, world, transport, bootstrap_path, unique_id);
        ctx->cc_programs_reset();   // a captured iteration does not contain the rank split
    });
}

int afesp_comm_destroy(afesp_ctx* ctx)
{
    return guarded(ctx, [&] {
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        ctx->cx.sync();
        comm_destroy(ctx->cx.comm);
        ctx->cx.comm = nullptr;
        ctx->cc_programs_reset();
    });
}

int afesp_allreduce_sum(afesp_ctx* ctx, double* inout, int64_t n)
{
    return guarded(ctx, [&] {
        if (n < 0 || (n > 0 && !inout)) throw Error(1, "afesp_allreduce_sum: bad arguments");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        if (!ctx->cx.comm) return;   // single rank: the sum over one rank
        comm_allreduce_host(ctx->cx, ctx->cx.comm, inout, n);
    });
}

int afesp_ccsd_is_split(afesp_ctx* ctx, int* split)
{
    return guarded(ctx, [&] {
        if (!ctx->cc.ready || !split) throw Error(1, "afesp_ccsd_is_split: no CCSD state");
        ccsd_refresh_sharding(ctx->cx, ctx->cc);
        *split = ctx->cc.sharded ? 1 : 0;
    });
}

int afesp_ccsd_set_fused(afesp_ctx* ctx, int mode)
{
    return guarded(ctx, [&] {
        if (mode < -1 || mode > 1) throw Error(1, "afesp_ccsd_set_fused: mode is -1 (environment), 0 (call by call) or 1 (launch-fused)");
        ctx->cx.fused_mode = mode;
    });
}

int afesp_ccsd_iteration_launches(afesp_ctx* ctx, int* launches)
{
    return guarded(ctx, [&] {
        if (!launches) throw Error(1, "afesp_ccsd_iteration_launches: null argument");
        *launches = fused_launches(ctx->fused_iter.prog);
    });
}

int afesp_ccsd_set_split(afesp_ctx* ctx, int mode)
{
    return guarded(ctx, [&] {
        if (mode < -1 || mode > 1) throw Error(1, "afesp_ccsd_set_split: mode is -1 (environment), 0 (replicas) or 1 (split)");
        ctx->cx.cc_split_mode = mode;
    });
}

int afesp_ccsd_t_block_size(afesp_ctx* ctx, int64_t nocc, int64_t nvirt, int cr, int* block_size)
{
    return guarded(ctx, [&] {
        if (nocc < 1 || nvirt < 1 || !block_size) throw Error(1, "afesp_ccsd_t_block_size: bad arguments");
        AFESP_HIP(hipSetDevice(ctx->cx.device));
        *block_size = triples_block_size((int)nocc, (int)nvirt, cr != 0);
    });
}

// diagnostic builds of the GEMM kernel (AFESP_GETT_VARIANT bit 64): the per-wave cycle stamps of the last launch
int afesp_debug_stamps(unsigned long long* out, int n)
{
    knobs_refresh();
    if (n < 0) return triples_read_orbit_stamps(out, -n) == hipSuccess ? 0 : 1;   // the (T) orbit kernel's phase sums
    if (knobs().stamps_grouped) return gett_read_stamps_grouped(out, n) == hipSuccess ? 0 : 1;
    return gett_read_stamps(out, n) == hipSuccess ? 0 : 1;
}

// which kernel took the products of this context so far (tests: a shape that should stream did, the LDS-DMA GEMM ran with 96-row tiles)
int afesp_test_ring_path(int64_t nocc, int64_t nvirt)
{
    knobs_refresh();
    CCState s;
    s.o = (int)nocc;
    s.v = (int)nvirt;
    return ring_tg_applies(s) ? 1 : 0;
}

uint64_t afesp_first_use_count(void) { return first_use_count().load(std::memory_order_relaxed); }

int afesp_launch_counts(afesp_ctx* ctx, uint64_t out[4])
{
    return guarded(ctx, [&] {
        out[0] = ctx->cx.n_tall; out[1] = ctx->cx.n_gett; out[2] = ctx->cx.tg.launches; out[3] = ctx->cx.tg.launches_mixed;
    });
}

// device arena of the context: {driver allocations so far, requests served from idle blocks, idle bytes, live bytes}
int afesp_arena_stats(afesp_ctx* ctx, double out[4])
{
    return guarded(ctx, [&] {
        const Arena& a = ctx->cx.arena;
        size_t live = 0;
        for (auto& kv : a.live) live += kv.second;
        out[0] = (double)a.driver_calls; out[1] = (double)a.reuse_hits; out[2] = (double)a.idle_bytes; out[3] = (double)live;
    });
}

int afesp_test_inject(afesp_ctx* ctx, int what)
{
    return guarded(ctx, [&] { ctx->cx.test_throw = what; });
}

}  // extern "C"
