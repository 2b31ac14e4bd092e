// contract.hip -- device context and the label-driven contraction planner on top of the gather-GEMM.
//
// A contraction site of the reference is "omp_reshape the operands until dgemm_wrapper applies"
// (src/linalg.fpp:58-156; sites listed in SURVEY.md section 2a).  Here a site is one contract() call that
// names the tensor indices; the planner groups them into row / column / summation groups, orders each group for
// coalesced access, builds the six offset tables once (cached per shape) and launches the MFMA kernel.
#include <algorithm>
#include <cstring>

#include "afesp_internal.h"
#include "fused.h"
#include "tall.h"

namespace afesp {

// ------------------------------------------------------------------ arena
static size_t arena_round(size_t bytes)
{
    if (bytes < 256) bytes = 256;
    const size_t g = bytes >= ((size_t)1 << 20) ? ((size_t)2 << 20) : 256;   // 2 MiB steps for large blocks: sizes that recur, recur exactly
    return (bytes + g - 1) / g * g;
}
void* Arena::get(size_t bytes)
{
    bytes = arena_round(bytes);
    auto it = idle.lower_bound(bytes);
    if (it != idle.end() && it->first <= bytes + std::max(bytes / 4, (size_t)1 << 20)) {
        void* p = it->second;
        live[p] = it->first;
        idle_bytes -= it->first;
        idle.erase(it);
        ++reuse_hits;
        return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    ++driver_calls;
    if (e != hipSuccess) {   // out of device memory: give the idle blocks back and try once more
        (void)hipGetLastError();
        trim();
        e = hipMalloc(&p, bytes);
        ++driver_calls;
    }
    if (e != hipSuccess) throw Error(2, std::string("device allocation of ") + std::to_string(bytes >> 20) + " MiB failed: " + hipGetErrorString(e));
    live[p] = bytes;
    return p;
}
void* Arena::take_largest(size_t min_bytes, size_t max_bytes, size_t* got)
{
    for (auto it = idle.rbegin(); it != idle.rend(); ++it) {
        if (it->first > max_bytes) continue;
        if (it->first < min_bytes) break;
        void* p = it->second;
        *got = it->first;
        live[p] = it->first;
        idle_bytes -= it->first;
        idle.erase(std::next(it).base());
        ++reuse_hits;
        return p;
    }
    return nullptr;
}
void Arena::put(void* p)
{
    auto it = live.find(p);
    if (it == live.end()) return;
    idle.emplace(it->second, p);
    idle_bytes += it->second;
    live.erase(it);
}
void Arena::trim()
{
    for (auto& kv : idle) (void)hipFree(kv.second);
    idle.clear();
    idle_bytes = 0;
}
void Arena::destroy()
{
    trim();
    for (auto& kv : live) (void)hipFree(kv.first);
    live.clear();
}

// ------------------------------------------------------------------ context
double* Context::alloc(int64_t n)
{
    const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(double);
    void* p = arena.get(bytes);
    AFESP_HIP(hipMemsetAsync(p, 0, bytes, stream));
    owned.push_back(p);
    return (double*)p;
}
double* Context::alloc_raw(int64_t n)
{
    void* p = arena.get((size_t)(n > 0 ? n : 1) * sizeof(double));
    owned.push_back(p);
    return (double*)p;
}
int64_t* Context::alloc_i64(int64_t n) { return (int64_t*)alloc(n); }
double* Context::scratch(const std::string& name, int64_t n)
{
    if (rec) rec->uses_scratch = true;
    const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(double);
    auto it = cache.find(name);
    if (it != cache.end() && it->second.second >= bytes) return (double*)it->second.first;
    if (it != cache.end()) {
        ++scratch_epoch;
        AFESP_HIP(hipStreamSynchronize(stream));
        arena.put(it->second.first);
        cache.erase(it);
    }
    void* p = arena.get(bytes);
    cache[name] = {p, bytes};
    return (double*)p;
}
// Blocks given back to the arena are handed out again at once (and alloc() memsets them asynchronously), so nothing queued on
// ANY lane may still touch them: a laned body that threw has forked its lanes without joining them.  These are cold paths.
void Context::quiesce()
{
    for (Lane& l : lanes)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (stream) (void)hipStreamSynchronize(stream);
    (void)hipGetLastError();
}
void Context::drop_scratch()
{
    ++scratch_epoch;
    quiesce();
    for (auto& kv : cache) arena.put(kv.second.first);
    cache.clear();
}
void Context::drop_scratch(const std::string& prefix)
{
    bool any = false;
    for (auto it = cache.begin(); it != cache.end();) {
        if (it->first.compare(0, prefix.size(), prefix) == 0) {
            if (!any) quiesce();
            any = true;
            arena.put(it->second.first);
            it = cache.erase(it);
        } else {
            ++it;
        }
    }
    if (any) ++scratch_epoch;
}
void Context::release(void* p)
{
    if (!p) return;
    auto it = std::find(owned.begin(), owned.end(), p);
    if (it != owned.end()) {
        owned.erase(it);
        quiesce();
        arena.put(p);
    }
}
int64_t* Context::plan_alloc(size_t n)
{
    n = (n + 1) & ~(size_t)1;
    constexpr size_t SLAB = (size_t)1 << 19;   // 4 MiB of int64
    if (n > SLAB / 4) {
        void* p = arena.get(n * sizeof(int64_t));
        plan_blocks.push_back(p);
        return (int64_t*)p;
    }
    if (plan_slab_left < n) {
        plan_slab = (int64_t*)arena.get(SLAB * sizeof(int64_t));
        plan_blocks.push_back(plan_slab);
        plan_slab_left = SLAB;
    }
    int64_t* p = plan_slab;
    plan_slab += n;
    plan_slab_left -= n;
    return p;
}
void Context::plan_uploads_issue()
{
    std::vector<PendingTables*> v;
    for (auto& t : pending_host)
        if (t.dst && !t.img.empty()) v.push_back(&t);
    std::sort(v.begin(), v.end(), [](const PendingTables* a, const PendingTables* b) { return a->dst < b->dst; });
    for (size_t i = 0; i < v.size();) {
        size_t j = i + 1;
        int64_t* end = v[i]->dst + v[i]->img.size();
        while (j < v.size() && v[j]->dst == end) { end += v[j]->img.size(); ++j; }
        if (j == i + 1) {
            AFESP_HIP(hipMemcpyAsync(v[i]->dst, v[i]->img.data(), v[i]->img.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        } else {
            std::vector<int64_t> merged;
            merged.reserve((size_t)(end - v[i]->dst));
            for (size_t k = i; k < j; ++k) merged.insert(merged.end(), v[k]->img.begin(), v[k]->img.end());
            pending_merged.push_back(std::move(merged));
            AFESP_HIP(hipMemcpyAsync(v[i]->dst, pending_merged.back().data(), pending_merged.back().size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        }
        for (size_t k = i; k < j; ++k) v[k]->dst = nullptr;   // issued
        i = j;
    }
}
void Context::plan_clear()
{
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : plan_blocks) arena.put(p);
    plan_blocks.clear();
    plan_slab = nullptr;
    plan_slab_left = 0;
    plans.clear();
    plan_bytes = 0;
    ++plan_epoch;
}

Tensor Context::tensor(std::initializer_list<int64_t> dims)
{
    int64_t n = 1;
    for (auto d : dims) n *= d;
    return view(alloc(n), dims);
}
void Context::sync() { AFESP_HIP(hipStreamSynchronize(stream)); }

void Context::fork(int nlanes)
{
    if (cur_lane != 0) throw Error(1, "Context::fork: already forked");
    if (startup.joinable()) startup.join();
    if (lanes.empty()) {
        lanes.resize(1);
        lanes[0].stream = stream;
        lanes[0].ws = ws;
        AFESP_HIP(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
    }
    if (lanes.size() == 1 && !prepared.lanes.empty()) {   // made on the start-up thread (joined by the caller of fork)
        if (prepared.ws_block) {
            arena.live[prepared.ws_block] = prepared.ws_bytes;
            owned.push_back(prepared.ws_block);
        }
        for (Lane& l : prepared.lanes) lanes.push_back(l);
        prepared.lanes.clear();
        prepared.ws_block = nullptr;
    }
    for (size_t i = 1; i < lanes.size(); ++i)
        if (lanes[i].ws.bytes < lane_ws_bytes) {   // (the old workspace stays with the context: a lane may still be reading it)
            void* p = arena.get(lane_ws_bytes);
            owned.push_back(p);
            lanes[i].ws.ptr = (double*)p;
            lanes[i].ws.bytes = lane_ws_bytes;
        }
    while ((int)lanes.size() < nlanes) {
        Lane l;
        AFESP_HIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        AFESP_HIP(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
        l.ws.bytes = lane_ws_bytes;   // (8 MiB unless a caller asked for more: lanes mostly carry small problems)
        void* p = arena.get(l.ws.bytes);
        owned.push_back(p);
        l.ws.ptr = (double*)p;
        lanes.push_back(l);
    }
    AFESP_HIP(hipEventRecord(fork_ev, lanes[0].stream));
    for (size_t i = 1; i < lanes.size(); ++i) AFESP_HIP(hipStreamWaitEvent(lanes[i].stream, fork_ev, 0));
}

void Context::prepare_lanes(Prepared& out, int nlanes)
{
    const size_t each = (size_t)8 << 20;
    void* block = nullptr;
    if (nlanes < 2 || hipMalloc(&block, each * (size_t)(nlanes - 1)) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    out.ws_block = block;
    out.ws_bytes = each * (size_t)(nlanes - 1);
    for (int i = 1; i < nlanes; ++i) {
        Lane l;
        if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            break;   // fork() makes the rest
        }
        l.ws.ptr = (double*)((char*)block + each * (size_t)(i - 1));
        l.ws.bytes = each;
        out.lanes.push_back(l);
    }
}

void Context::use_lane(int i)
{
    if (lanes.empty()) return;
    cur_lane = i % (int)lanes.size();
    stream = lanes[(size_t)cur_lane].stream;
    ws = lanes[(size_t)cur_lane].ws;
}

int Context::mark()
{
    if (marks_used == (int)marks.size()) {
        hipEvent_t e;
        AFESP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        marks.push_back(e);
    }
    AFESP_HIP(hipEventRecord(marks[(size_t)marks_used], stream));
    return marks_used++;
}

void Context::wait(int mark_id) { AFESP_HIP(hipStreamWaitEvent(stream, marks[(size_t)mark_id], 0)); }

void Context::join()
{
    marks_used = 0;
    if (lanes.empty()) return;
    for (size_t i = 1; i < lanes.size(); ++i) {
        AFESP_HIP(hipEventRecord(lanes[i].done, lanes[i].stream));
        AFESP_HIP(hipStreamWaitEvent(lanes[0].stream, lanes[i].done, 0));
    }
    use_lane(0);
}
Context::~Context()
{
    if (startup.joinable()) startup.join();
    use_lane(0);
    for (size_t i = 1; i < lanes.size(); ++i) {
        (void)hipStreamSynchronize(lanes[i].stream);
        (void)hipStreamDestroy(lanes[i].stream);
        (void)hipEventDestroy(lanes[i].done);
    }
    for (Lane& l : prepared.lanes) {
        (void)hipStreamDestroy(l.stream);
        (void)hipEventDestroy(l.done);
    }
    if (prepared.ws_block) (void)hipFree(prepared.ws_block);
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    for (hipEvent_t e : marks) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamSynchronize(stream);
    arena.destroy();   // every block the context ever obtained: owned, cached scratch and idle ones
    if (scal_host) (void)hipHostFree(scal_host);
    if (res_host) (void)hipHostFree(res_host);
    tgemm_state_free(tg);
    if (stream) (void)hipStreamDestroy(stream);
}

Tensor view(double* d, std::initializer_list<int64_t> dims)
{
    Tensor t;
    t.d = d;
    t.rank = (int)dims.size();
    int64_t s = 1;
    int i = 0;
    for (auto x : dims) {
        t.dim[i] = x;
        t.stride[i] = s;
        s *= x;
        ++i;
    }
    return t;
}

// cx.scal[0..n) on the host.  A copy + stream synchronisation costs 30-40 us of wake-up latency on this runtime -- a fifth of a small
// system's (T), as much as five of its kernels; so a one-block kernel publishes the values into coherent pinned memory, sequence
// number last, and the host polls that word (falling back to waiting for the stream if it does not appear).
__global__ void publish_scalars_kernel(double* __restrict__ dst, const double* __restrict__ src, int n, double seq)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&dst[64], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

double* host_scalars(Context& cx, int n)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (n > 64 || !cx.res_dev || hipStreamIsCapturing(cx.stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        AFESP_HIP(hipMemcpyAsync(cx.scal_host, cx.scal, sizeof(double) * n, hipMemcpyDeviceToHost, cx.stream));
        cx.sync();
        return cx.scal_host;
    }
    constexpr int PUB = 8 + 256;   // the publishing area of res_host: 64 values and their sequence number
    const double want = (double)++cx.pub_seq;
    AFESP_KLAUNCH(publish_scalars_kernel, dim3(1), dim3(64), 0, cx.stream, cx.res_dev + PUB, cx.scal, n, want);
    AFESP_HIP(hipGetLastError());
    return host_scalars_wait(cx, n, want);
}

// where a kernel of the caller's own may publish up to 64 values itself (values first, then the sequence number *seq into slot 64
// with release semantics at system scope, as publish_scalars_kernel does); nullptr: not available, use host_scalars
double* host_scalars_slot(Context& cx, double* seq)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (!cx.res_dev || hipStreamIsCapturing(cx.stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    *seq = (double)++cx.pub_seq;
    return cx.res_dev + 8 + 256;
}

double* host_scalars_wait(Context& cx, int n, double want)
{
    constexpr int PUB = 8 + 256;
    bool seen = false;
    for (int spin = 0; spin < 400000; ++spin) {
        if (__atomic_load_n((const int64_t*)&cx.res_host[PUB + 64], __ATOMIC_ACQUIRE) == *(const int64_t*)&want) { seen = true; break; }
        if ((spin & 1023) == 1023 && hipStreamQuery(cx.stream) != hipErrorNotReady) break;
    }
    if (!seen) {
        AFESP_HIP(hipStreamSynchronize(cx.stream));
        if (cx.res_host[PUB + 64] != want) throw Error(2, "host_scalars: the published values did not arrive");
    }
    for (int q = 0; q < n; ++q) cx.scal_host[q] = cx.res_host[PUB + q];
    return cx.scal_host;
}

// ------------------------------------------------------------------ planner
namespace {

struct Lab {
    char c;
    int64_t dim;
    int64_t sa, sb, sc;   // strides in A, B, C (0 if absent)
};

int64_t stride_of(const Tensor& t, const char* l, char c)
{
    for (int i = 0; i < t.rank; ++i)
        if (l[i] == c) return t.stride[i];
    return -1;
}

char min_stride_label(const Tensor& t, const char* l)
{
    char best = 0;
    int64_t bs = INT64_MAX;
    for (int i = 0; i < t.rank; ++i)
        if (t.dim[i] > 1 && t.stride[i] < bs) {
            bs = t.stride[i];
            best = l[i];
        }
    if (!best && t.rank > 0) best = l[0];
    return best;
}

std::vector<int64_t> table(const std::vector<Lab>& g, int which)
{
    int64_t n = 1;
    for (auto& l : g) n *= l.dim;
    std::vector<int64_t> t((size_t)n);
    std::vector<int64_t> idx(g.size(), 0);
    for (int64_t x = 0; x < n; ++x) {
        int64_t off = 0;
        for (size_t q = 0; q < g.size(); ++q) off += idx[q] * (which == 0 ? g[q].sa : which == 1 ? g[q].sb : g[q].sc);
        t[(size_t)x] = off;
        for (size_t q = 0; q < g.size(); ++q) {
            if (++idx[q] < g[q].dim) break;
            idx[q] = 0;
        }
    }
    return t;
}

// The same table written by the device: for the big enumerations (the AO->MO transforms walk 220 x 24310 = 5.3 M columns: six
// host loops and 170 MB of uploads made the first transform of a process 309 ms against 54 ms for the next one).
struct TabArgs {
    int nd;
    int64_t dim[8], stride[8];
    int64_t n;
};

__global__ __launch_bounds__(256) void plan_table_kernel(int64_t* __restrict__ out, TabArgs a)
{
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < a.n; x += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = x, off = 0;
        for (int q = 0; q < a.nd; ++q) {
            off += (r % a.dim[q]) * a.stride[q];
            r /= a.dim[q];
        }
        out[x] = off;
    }
}

int64_t stride_in(const Lab& l, int which) { return which == 0 ? l.sa : which == 1 ? l.sb : l.sc; }

int64_t table_size(const std::vector<Lab>& g)
{
    int64_t n = 1;
    for (auto& l : g) n *= l.dim;
    return n;
}

// What pairs() / evens() below find by scanning a table, from the strides alone.  Labels of extent 1 contribute nothing; labels
// that continue each other (stride of the next = extent x stride of this one) enumerate like one label.
void table_flags(const std::vector<Lab>& g, int which, bool* pairs, bool* evens)
{
    std::vector<std::pair<int64_t, int64_t>> ds;   // (extent, stride), fastest first, merged
    for (auto& l : g) {
        if (l.dim == 1) continue;
        const int64_t st = stride_in(l, which);
        if (!ds.empty() && st == ds.back().first * ds.back().second) ds.back().first *= l.dim;
        else ds.push_back({l.dim, st});
    }
    *evens = true;
    for (auto& d : ds)
        if (d.second & 1) *evens = false;
    *pairs = !ds.empty() && ds[0].first % 2 == 0 && ds[0].second == 1;
    for (size_t q = 1; q < ds.size(); ++q)
        if (ds[q].second & 1) *pairs = false;
}

int64_t* upload(Context& cx, const std::vector<int64_t>& h)
{
    int64_t* d = cx.plan_alloc(h.size());
    AFESP_HIP(hipMemcpyAsync(d, h.data(), h.size() * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
    AFESP_HIP(hipStreamSynchronize(cx.stream));   // h is a temporary
    return d;
}

void sort_by(std::vector<Lab>& g, int which)
{
    std::stable_sort(g.begin(), g.end(), [which](const Lab& x, const Lab& y) {
        int64_t a = which == 0 ? x.sa : which == 1 ? x.sb : x.sc, b = which == 0 ? y.sa : which == 1 ? y.sb : y.sc;
        return a < b;
    });
}

bool has(const std::vector<Lab>& g, char c)
{
    for (auto& l : g)
        if (l.c == c) return true;
    return false;
}

}  // namespace

void contract(Context& cx, double alpha, const Tensor& A0, const char* la0, const Tensor& B0, const char* lb0, double beta,
              const Tensor& C, const char* lc, int nbatch, const int64_t* bA0, const int64_t* bB0, const int64_t* bC,
              int force_split, int force_tm, int force_tn)
{
    if ((int)strlen(la0) != A0.rank || (int)strlen(lb0) != B0.rank || (int)strlen(lc) != C.rank)
        throw Error(3, std::string("contract: label/rank mismatch ") + la0 + "," + lb0 + "->" + lc);
    std::string key = std::string(la0) + "," + lb0 + ">" + lc;
    auto sig = [&key](const Tensor& t) {
        for (int i = 0; i < t.rank; ++i) key += ":" + std::to_string(t.dim[i]) + "/" + std::to_string(t.stride[i]);
        key += ";";
    };
    sig(A0); sig(B0); sig(C);
    // the re-layout decision below depends on these as well: a shape first seen unbatched must not hand its plan to a batched call
    key += (nbatch > 1 || bA0 || bB0) ? "|b" : "|u";
    const bool norepack = cx.in_repack || cx.rec != nullptr;   // (a recorded product of the launch-fused path is small: never re-laid-out)
    key += norepack ? "r" : "-";
    auto it = cx.plans.find(key);
    if (it == cx.plans.end()) {
        // orientation: the kernel's column index n runs along lanes in the epilogue -> put C's fastest label in N
        char cfast = min_stride_label(C, lc);
        bool swapped = stride_of(A0, la0, cfast) >= 0;
        const Tensor& A = swapped ? B0 : A0;
        const Tensor& B = swapped ? A0 : B0;
        const char* la = swapped ? lb0 : la0;
        const char* lb = swapped ? la0 : lb0;
        std::vector<Lab> M, N, K;
        for (int i = 0; i < A.rank; ++i) {
            Lab l{la[i], A.dim[i], A.stride[i], stride_of(B, lb, la[i]), stride_of(C, lc, la[i])};
            bool inB = l.sb >= 0, inC = l.sc >= 0;
            if (inB == inC) throw Error(3, "contract: label '" + std::string(1, l.c) + "' must be in exactly two tensors: " + key);
            if (inB && B.dim[std::strchr(lb, l.c) - lb] != l.dim) throw Error(3, "contract: extent mismatch in " + key);
            if (inC && C.dim[std::strchr(lc, l.c) - lc] != l.dim) throw Error(3, "contract: extent mismatch in " + key);
            (inC ? M : K).push_back(l);
        }
        for (int i = 0; i < B.rank; ++i) {
            Lab l{lb[i], B.dim[i], stride_of(A, la, lb[i]), B.stride[i], stride_of(C, lc, lb[i])};
            if (l.sa >= 0) continue;   // K label, already taken from A
            if (l.sc < 0) throw Error(3, "contract: label '" + std::string(1, l.c) + "' only in one operand: " + key);
            if (C.dim[std::strchr(lc, l.c) - lc] != l.dim) throw Error(3, "contract: extent mismatch in " + key);
            N.push_back(l);
        }
        if ((int)(M.size() + N.size()) != C.rank) throw Error(3, "contract: output label not produced: " + key);
        char afast = min_stride_label(A, la), bfast = min_stride_label(B, lb);
        sort_by(N, 2);
        sort_by(M, has(M, afast) ? 0 : 2);
        // both operands want their own fastest label first in K: the larger one is the one that streams from HBM
        if (has(K, afast) && has(K, bfast)) sort_by(K, B.size() > 4 * A.size() ? 1 : 0);
        else if (has(K, afast)) sort_by(K, 0);
        else if (has(K, bfast)) sort_by(K, 1);
        else sort_by(K, 0);
        Plan p;
        p.swapped = swapped;
        // An operand whose unit-stride label leads neither its free nor its contracted enumeration is read 8 bytes per
        // 64-byte line on every K step.  When the product is large enough to pay for it, re-lay it out once into scratch
        // (free labels in C order first, then K) and plan again on the copy.
        {
            auto lead = [](const std::vector<Lab>& g) {
                for (auto& l : g)
                    if (l.dim > 1) return l.c;
                return (char)0;
            };
            const bool a_ok = lead(K) == afast || lead(M) == afast || A.size() <= 4096;
            const bool b_ok = lead(K) == bfast || lead(N) == bfast || B.size() <= 4096;
            int64_t Md = 1, Nd = 1;
            for (auto& l : M) Md *= l.dim;
            for (auto& l : N) Nd *= l.dim;
            const int64_t limit = (int64_t)1 << 28;
            int which = 0;   // 1 = A, 2 = B (kernel roles)
            // (... or when the other operand is so much larger that the copy is small change beside streaming it:
            // asym(m,i,e,f) <ef|ma> -> r1(i,a) at o = 20, v = 200 reads 1.28 GB of integrals against 128 MB of amplitudes that it
            // would otherwise gather 8 bytes at a time -- 0.76 -> 0.45 ms)
            if (nbatch == 1 && !bA0 && !bB0 && !norepack) {
                // (the other operand's free extent from which the copy pays: 256 -- a rank's 500-row slice of a ring product in a split
                // iteration runs 0.52 ms from the operand as it lies, 0.29 + 0.07 from its copy; tuning knob AFESP_REPACK_MIN)
                const int64_t rp_min = knobs().repack_min;
                // (... or when the summation index is so long that both operands simply stream: gathered 8 bytes per line, a rank's
                // slice of asym(m,i,e,f) <ef|ma> -> r1(i,a) ran at 0.9 TB/s -- 316 us for 288 MB)
                int64_t Kd = 1;
                for (auto& l : K) Kd *= l.dim;
                const bool longk = Kd >= 65536;
                if (!b_ok && (Md >= rp_min || A.size() >= 8 * B.size() || (longk && B.size() <= 2 * A.size())) && B.size() <= limit) which = 2;
                else if (!a_ok && (Nd >= rp_min || B.size() >= 8 * A.size() || (longk && A.size() <= 2 * B.size())) && A.size() <= limit) which = 1;
            }
            if (which) {
                const Tensor& T = which == 1 ? A : B;
                const char* lt = which == 1 ? la : lb;
                std::vector<Lab> F = which == 1 ? M : N;
                sort_by(F, 2);
                int64_t run = 1;
                auto place = [&](const std::vector<Lab>& g) {
                    for (auto& l : g) {
                        p.repack_stride[std::strchr(lt, l.c) - lt] = run;
                        run *= l.dim;
                    }
                };
                place(F);
                place(K);
                (void)T;
                p.repack = (which == 1) != swapped ? 1 : 2;
                p.offAm = p.offAk = p.offBk = p.offBn = p.offCm = p.offCn = nullptr;
                p.M = p.N = p.K = 0;
                p.a_kc = p.b_kc = p.wide = false;
                it = cx.plans.emplace(key, p).first;
            }
        }
        if (it == cx.plans.end()) {
        p.a_kc = !K.empty() && K[0].c == afast && K[0].sa == 1;
        p.b_kc = !K.empty() && K[0].c == bfast && K[0].sb == 1;
        {
            auto lead_unit = [](const std::vector<Lab>& g, int which) {
                for (auto& l : g)
                    if (l.dim > 1) return (which == 0 ? l.sa : l.sb) == 1;
                return true;
            };
            p.a_mu = lead_unit(M, 0);
            p.b_nu = lead_unit(N, 1);
        }
        const std::vector<Lab>* grp[6] = {&M, &K, &K, &N, &M, &N};
        const int whichs[6] = {0, 0, 1, 1, 2, 2};   // offAm, offAk, offBk, offBn, offCm, offCn
        int64_t nn[6];
        for (int q = 0; q < 6; ++q) nn[q] = table_size(*grp[q]);
        if (nn[0] > INT32_MAX || nn[3] > INT32_MAX || nn[1] > INT32_MAX) throw Error(3, "contract: extent too large");
        p.M = (int)nn[0]; p.N = (int)nn[3]; p.K = (int)nn[1];
        // 16-byte staging is legal when every offset is even and the contiguous direction advances in unit-stride pairs
        bool pr[6], ev[6];
        for (int q = 0; q < 6; ++q) table_flags(*grp[q], whichs[q], &pr[q], &ev[q]);
        p.wide = (p.a_kc ? (pr[1] && ev[0]) : (pr[0] && ev[1])) && (p.b_kc ? (pr[2] && ev[3]) : (pr[3] && ev[2]));
        // the six tables in one allocation (a plan per contraction site: ~45 of them in a CCSD iteration), each starting on a
        // 16-byte boundary; small ones are enumerated here and copied, big ones are written by the device
        {
            const bool verify = knobs().plan_verify;   // tests: both builders, and the flags, must agree
            const int64_t device_from = knobs().plan_device_from;
            int64_t** dst[6] = {&p.offAm, &p.offAk, &p.offBk, &p.offBn, &p.offCm, &p.offCn};
            size_t start[6], total = 0;
            for (int q = 0; q < 6; ++q) {
                start[q] = total;
                total += (size_t)nn[q] + ((size_t)nn[q] & 1);
            }
            int64_t* base = cx.plan_alloc(total);
            cx.plan_bytes += total * sizeof(int64_t);
            std::vector<int64_t> small;   // the host-built tables, one copy
            std::vector<std::pair<size_t, size_t>> runs;   // (start in `small`, start in the allocation) of each
            for (int q = 0; q < 6; ++q) {
                *dst[q] = base + start[q];
                if (nn[q] >= device_from) {
                    TabArgs ta;
                    ta.nd = 0;
                    for (auto& l : *grp[q]) {
                        if (ta.nd == 8) throw Error(3, "contract: too many labels in one group: " + key);
                        ta.dim[ta.nd] = l.dim;
                        ta.stride[ta.nd++] = stride_in(l, whichs[q]);
                    }
                    ta.n = nn[q];
                    const unsigned blocks = (unsigned)std::min<int64_t>((nn[q] + 255) / 256, 8192);
                    AFESP_KLAUNCH(plan_table_kernel, dim3(blocks), dim3(256), 0, cx.stream, base + start[q], ta);
                    AFESP_HIP(hipGetLastError());
                } else {
                    const std::vector<int64_t> t = table(*grp[q], whichs[q]);
                    runs.push_back({small.size(), start[q]});
                    small.insert(small.end(), t.begin(), t.end());
                }
            }
            if (runs.size() == 6) {
                // all six host-built (every plan of a small system): one image of the allocation, ONE copy -- six copies per plan
                // were 0.8 of the 1.0 ms that recording the 43 calls of a small iteration took
                std::vector<int64_t> image(total, 0);
                for (size_t r = 0; r < 6; ++r) {
                    const size_t len = (r + 1 < 6 ? runs[r + 1].first : small.size()) - runs[r].first;
                    std::copy(small.begin() + runs[r].first, small.begin() + runs[r].first + len, image.begin() + runs[r].second);
                }
                small.swap(image);
                if (cx.rec && !verify) {   // recording: the image waits for plan_uploads_issue (fused_compile, or the way out of a failed recording)
                    cx.pending_host.push_back({base, std::move(small)});
                    small.clear();
                } else if (total) {
                    AFESP_HIP(hipMemcpyAsync(base, small.data(), total * sizeof(int64_t), hipMemcpyHostToDevice, cx.stream));
                }
            } else {
                for (size_t r = 0; r < runs.size(); ++r) {
                    const size_t len = (r + 1 < runs.size() ? runs[r + 1].first : small.size()) - runs[r].first;
                    if (len)
                        AFESP_HIP(hipMemcpyAsync(base + runs[r].second, small.data() + runs[r].first, len * sizeof(int64_t), hipMemcpyHostToDevice,
                                                 cx.stream));
                }
            }
            if (cx.rec && !verify) {   // recording: ONE wait for all the uploads (fused_compile)
                if (!small.empty()) cx.pending_host.push_back({nullptr, std::move(small)});   // (copies already issued from it: kept alive only)
            } else {
                AFESP_HIP(hipStreamSynchronize(cx.stream));                      // `small` is a temporary
            }
            if (verify) {
                auto pairs = [](const std::vector<int64_t>& t) {
                    if (t.size() % 2) return false;
                    for (size_t x = 0; x + 1 < t.size(); x += 2)
                        if ((t[x] & 1) || t[x + 1] != t[x] + 1) return false;
                    return true;
                };
                auto evens = [](const std::vector<int64_t>& t) {
                    for (int64_t x : t)
                        if (x & 1) return false;
                    return true;
                };
                for (int q = 0; q < 6; ++q) {
                    const std::vector<int64_t> ref = table(*grp[q], whichs[q]);
                    std::vector<int64_t> got(ref.size());
                    AFESP_HIP(hipMemcpy(got.data(), base + start[q], ref.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
                    if (got != ref) throw Error(2, "contract: device-built offset table differs from the host enumeration: " + key);
                    if (pairs(ref) != pr[q] || evens(ref) != ev[q])
                        throw Error(2, "contract: analytic alignment flags differ from the scanned table: " + key);
                }
            }
        }
        it = cx.plans.emplace(key, p).first;
        }
    }
    Plan& p = it->second;
    if (p.repack) {
        const Tensor& src = p.repack == 1 ? A0 : B0;
        Tensor packed = src;
        packed.frozen = 0;
        for (int i = 0; i < src.rank; ++i) packed.stride[i] = p.repack_stride[i];
        packed.d = cx.scratch("repack:" + std::to_string(cx.cur_lane) + ":" + key, src.size());
        const char* ls = p.repack == 1 ? la0 : lb0;
        // (an immutable operand -- the MO integrals of a solver state -- is re-laid-out once, not once per iteration)
        const bool have = src.frozen != 0 && p.repack_of == src.frozen && p.repack_src == src.d && p.repack_epoch == cx.scratch_epoch;
        if (!have) {
            permute_add(cx, 1.0, src, ls, 0.0, packed, ls);
            p.repack_of = src.frozen;
            p.repack_src = src.d;
            p.repack_epoch = cx.scratch_epoch;
        }
        cx.in_repack = true;
        try {
            contract(cx, alpha, p.repack == 1 ? packed : A0, la0, p.repack == 2 ? packed : B0, lb0, beta, C, lc, nbatch, bA0, bB0,
                     bC, force_split, force_tm, force_tn);
        } catch (...) {
            cx.in_repack = false;
            throw;
        }
        cx.in_repack = false;
        return;
    }
    GettProblem g;
    g.A = p.swapped ? B0.d : A0.d;
    g.B = p.swapped ? A0.d : B0.d;
    g.C = C.d;
    g.offAm = p.offAm; g.offAk = p.offAk; g.offBk = p.offBk; g.offBn = p.offBn; g.offCm = p.offCm; g.offCn = p.offCn;
    g.M = p.M; g.N = p.N; g.K = p.K;
    g.alpha = alpha; g.beta = beta;
    g.nbatch = nbatch;
    g.batchA = p.swapped ? bB0 : bA0;
    g.batchB = p.swapped ? bA0 : bB0;
    g.batchC = bC;
    g.a_kcontig = p.a_kc; g.b_kcontig = p.b_kc;
    g.a_munit = p.a_mu; g.b_nunit = p.b_nu;
    g.wide = p.wide && ((uintptr_t)g.A % 16 == 0) && ((uintptr_t)g.B % 16 == 0);
    if (cx.capture && !cx.rec) {   // contract_pair: the planned product goes back to it
        *cx.capture = g;
        cx.capture = nullptr;
        return;
    }
    if (cx.rec) {   // launch-fused path: the product joins the recording instead of being launched
        auto span = [](const Tensor& t) {
            int64_t s = 1;
            for (int i = 0; i < t.rank; ++i) s += (t.dim[i] - 1) * t.stride[i];
            return s;
        };
        cx.rec->product(g, span(p.swapped ? B0 : A0), span(p.swapped ? A0 : B0), span(C));
        return;
    }
    // AFESP_CONTRACT_TRACE=1 (tools/contract_trace.py): every product alone on the device, its labels, extents and time on stderr
    const bool trace = knobs().contract_trace;
    // tall x skinny (one extent of C at most 32, K a few hundred: the products of t1 with a four-index array): streamed, tall.h
    const bool tall = !force_split && !force_tm && !force_tn && tall_eligible(g);
    ++(tall ? cx.n_tall : cx.n_gett);
    auto launch = [&]() { return tall ? tall_launch(g, cx.stream) : gett_launch(g, cx.ws, cx.stream, force_split, force_tm, force_tn); };
    if (!trace) {
        AFESP_HIP(launch());
        return;
    }
    hipEvent_t e0, e1;
    AFESP_HIP(hipEventCreate(&e0));
    AFESP_HIP(hipEventCreate(&e1));
    AFESP_HIP(hipDeviceSynchronize());
    AFESP_HIP(hipEventRecord(e0, cx.stream));
    AFESP_HIP(launch());
    AFESP_HIP(hipEventRecord(e1, cx.stream));
    AFESP_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    AFESP_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double fl = 2.0 * g.M * (double)g.N * g.K * nbatch, by = 8.0 * (A0.size() + B0.size() + C.size());
    fprintf(stderr, "contract %-6s,%-6s>%-6s M %7d N %7d K %7d akc %d bkc %d wide %d %9.1f us %6.2f TF %7.1f GB/s%s\n", la0, lb0, lc, g.M, g.N,
            g.K, (int)g.a_kcontig, (int)g.b_kcontig, (int)g.wide, ms * 1e3, fl / ms * 1e-9, by / ms * 1e-6, cx.in_repack ? "  (repacked)" : tall ? "  (tall)" : "");
}

void contract_pair(Context& cx, const ContractCall& c1, const ContractCall& c2)
{
    const bool trace = knobs().contract_trace;
    auto plain = [&](const ContractCall& c) { contract(cx, c.alpha, *c.A, c.la, *c.B, c.lb, c.beta, *c.C, c.lc); };
    if (cx.rec || trace) {
        plain(c1);
        plain(c2);
        return;
    }
    GettProblem g[2];
    const ContractCall* cc[2] = {&c1, &c2};
    for (int i = 0; i < 2; ++i) {
        g[i].M = 0;
        cx.capture = &g[i];
        try {
            plain(*cc[i]);
        } catch (...) {
            cx.capture = nullptr;
            throw;
        }
        if (cx.capture) {   // (nothing was planned: an empty product)
            cx.capture = nullptr;
            g[i].M = 0;
        }
    }
    auto one = [&](const GettProblem& p) {
        if (p.M <= 0 || p.N <= 0) return;
        const bool tall = tall_eligible(p);
        ++(tall ? cx.n_tall : cx.n_gett);
        AFESP_HIP(tall ? tall_launch(p, cx.stream) : gett_launch(p, cx.ws, cx.stream, 0, 0, 0));
    };
    if (g[0].M > 0 && g[1].M > 0 && tall_dual_eligible(g[0], g[1])) {
        ++cx.n_tall;
        AFESP_HIP(tall_launch_dual(g[0], g[1], cx.stream));
    } else {
        one(g[0]);
        one(g[1]);
    }
}

// ------------------------------------------------------------------ permute_add
struct PermArgs {
    int rank;
    int64_t dim[6], so[6], si[6];
    int64_t n;
    double alpha, beta;
};

__global__ __launch_bounds__(256) void permute_add_kernel(double* __restrict__ out, const double* __restrict__ in, PermArgs a)
{
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < a.n; x += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = x, oo = 0, io = 0;
        for (int q = 0; q < a.rank; ++q) {
            int64_t i = r % a.dim[q];
            r /= a.dim[q];
            oo += i * a.so[q];
            io += i * a.si[q];
        }
        double val = a.alpha * in[io];
        if (a.beta != 0.0) val += a.beta * out[oo];
        out[oo] = val;
    }
}

// Transposing case: the unit-stride index of `in` (label fi) differs from that of `out` (label fo).  A workgroup moves a
// 32 x 32 tile of the (fo, fi) plane through LDS, so reads run along fi and writes along fo; the other indices are
// enumerated by the block index.
struct PermTileArgs {
    int nother;
    int64_t dim[4], so[4], si[4];
    int64_t d_o, d_i, in_stride_o, out_stride_i, tiles_o, tiles_i;
    double alpha, beta;
};

__global__ __launch_bounds__(256) void permute_add_tiled_kernel(double* __restrict__ out, const double* __restrict__ in, PermTileArgs a)
{
    __shared__ double tile[32][33];
    int64_t b = blockIdx.x;
    const int64_t o0 = (b % a.tiles_o) * 32;
    b /= a.tiles_o;
    const int64_t i0 = (b % a.tiles_i) * 32;
    b /= a.tiles_i;
    int64_t bo = 0, bi = 0;
    for (int q = 0; q < a.nother; ++q) {
        const int64_t i = b % a.dim[q];
        b /= a.dim[q];
        bo += i * a.so[q];
        bi += i * a.si[q];
    }
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ro = ty + 8 * r;
        if (o0 + ro < a.d_o && i0 + tx < a.d_i) tile[ro][tx] = in[bi + (o0 + ro) * a.in_stride_o + (i0 + tx)];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ri = ty + 8 * r;
        if (i0 + ri < a.d_i && o0 + tx < a.d_o) {
            const int64_t addr = bo + (o0 + tx) + (i0 + ri) * a.out_stride_i;
            double val = a.alpha * tile[tx][ri];
            if (a.beta != 0.0) val += a.beta * out[addr];
            out[addr] = val;
        }
    }
}

void preload_contract()
{
    first_use_touch(reinterpret_cast<const void*>(permute_add_kernel));
    first_use_touch(reinterpret_cast<const void*>(plan_table_kernel));
    first_use_touch(reinterpret_cast<const void*>(permute_add_tiled_kernel));
    first_use_touch(reinterpret_cast<const void*>(publish_scalars_kernel));
    (void)hipGetLastError();
}

void permute_add(Context& cx, double alpha, const Tensor& in, const char* li, double beta, const Tensor& out, const char* lo)
{
    if ((int)strlen(li) != in.rank || (int)strlen(lo) != out.rank || in.rank != out.rank || in.rank > 6)
        throw Error(3, "permute_add: rank mismatch");
    // iterate in ascending output stride so that consecutive threads write consecutive addresses
    std::vector<int> ord(out.rank);
    for (int i = 0; i < out.rank; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return out.stride[x] < out.stride[y]; });
    PermArgs a;
    a.rank = out.rank;
    a.n = 1;
    for (int q = 0; q < out.rank; ++q) {
        int i = ord[q];
        const char* p = std::strchr(li, lo[i]);
        if (!p) throw Error(3, "permute_add: labels are not a permutation");
        int j = (int)(p - li);
        if (in.dim[j] != out.dim[i]) throw Error(3, "permute_add: extent mismatch");
        a.dim[q] = out.dim[i];
        a.so[q] = out.stride[i];
        a.si[q] = in.stride[j];
        a.n *= out.dim[i];
    }
    a.alpha = alpha;
    a.beta = beta;
    if (a.n == 0) return;
    if (cx.rec) {
        cx.rec->elementwise(out.d, in.d, a.rank, a.dim, a.so, a.si, alpha, beta);
        return;
    }
    // q_o / q_i: positions (in the enumeration above) of the indices that are unit-stride in out / in
    int q_o = -1, q_i = -1;
    for (int q = 0; q < a.rank; ++q) {
        if (a.so[q] == 1 && a.dim[q] > 1 && q_o < 0) q_o = q;
        if (a.si[q] == 1 && a.dim[q] > 1 && q_i < 0) q_i = q;
    }
    if (q_o >= 0 && q_i >= 0 && q_o != q_i && a.rank <= 6 && a.dim[q_o] >= 8 && a.dim[q_i] >= 8 && a.n >= (1 << 16)) {
        PermTileArgs t;
        t.nother = 0;
        int64_t blocks = 1;
        for (int q = 0; q < a.rank; ++q) {
            if (q == q_o || q == q_i) continue;
            t.dim[t.nother] = a.dim[q]; t.so[t.nother] = a.so[q]; t.si[t.nother] = a.si[q];
            blocks *= a.dim[q];
            ++t.nother;
        }
        t.d_o = a.dim[q_o]; t.d_i = a.dim[q_i];
        t.in_stride_o = a.si[q_o]; t.out_stride_i = a.so[q_i];
        t.tiles_o = (t.d_o + 31) / 32; t.tiles_i = (t.d_i + 31) / 32;
        t.alpha = alpha; t.beta = beta;
        blocks *= t.tiles_o * t.tiles_i;
        if (blocks < ((int64_t)1 << 31)) {
            AFESP_KLAUNCH(permute_add_tiled_kernel, dim3((unsigned)blocks), dim3(256), 0, cx.stream, out.d, in.d, t);
            AFESP_HIP(hipGetLastError());
            return;
        }
    }
    unsigned grid = (unsigned)std::min<int64_t>((a.n + 255) / 256, 4096);
    AFESP_KLAUNCH(permute_add_kernel, dim3(grid), dim3(256), 0, cx.stream, out.d, in.d, a);
    AFESP_HIP(hipGetLastError());
}

}  // namespace afesp
