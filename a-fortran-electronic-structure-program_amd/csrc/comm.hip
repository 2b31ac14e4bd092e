// comm.hip -- rank-to-rank sums (comm.h): RCCL all-reduce on the context stream, or a shared host segment for ranks that
// share a GPU.
#include "comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>

namespace afesp {

namespace {

// ---- RCCL, opened on first use (torch ships a librccl.so.1 of its own: whichever the process has loaded is reused)
struct NcclId { char internal[128]; };
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

Rccl& rccl()
{
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            x.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (x.h) break;
        }
        if (!x.h) return x;
        x.GetUniqueId = (int (*)(NcclId*))dlsym(x.h, "ncclGetUniqueId");
        x.CommInitRank = (int (*)(void**, int, NcclId, int))dlsym(x.h, "ncclCommInitRank");
        x.CommDestroy = (int (*)(void*))dlsym(x.h, "ncclCommDestroy");
        x.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(x.h, "ncclAllReduce");
        x.GetErrorString = (const char* (*)(int))dlsym(x.h, "ncclGetErrorString");
        if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.AllReduce) x.h = nullptr;
        return x;
    }();
    if (!r.h) throw Error(20, "afesp_comm: librccl.so.1 could not be opened (RCCL transport needs ROCm's RCCL on the library path)");
    return r;
}

void nccl_check(int rc, const char* what)
{
    if (rc == 0) return;
    Rccl& r = rccl();
    throw Error(21, std::string("afesp_comm: ") + what + " failed: " + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
}
constexpr int NCCL_FLOAT64 = 8, NCCL_SUM = 0;   // rccl.h: ncclDouble, ncclSum

// ---- HOST transport: one file-backed segment, [header | slot of rank 0 | slot of rank 1 | ...]
struct SegHeader {
    std::atomic<uint32_t> magic;
    uint32_t world;
    int64_t slot_doubles;
    std::atomic<uint32_t> count;
    std::atomic<uint32_t> gen;
};
constexpr uint32_t SEG_MAGIC = 0xAFE59C0Du;
constexpr size_t SEG_HEADER_BYTES = 4096;
constexpr int64_t SLOT_DOUBLES = (int64_t)1 << 20;   // 8 MiB per rank
constexpr double WAIT_LIMIT_S = 300.0;

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// waits for `pred` with back-off; every wait on another rank is bounded
template <class P>
void wait_for(P pred, const char* what)
{
    const double t0 = now_s();
    for (int spins = 0; !pred(); ++spins) {
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(spins > 20000 ? 1000 : 20));
        if ((spins & 1023) == 0 && now_s() - t0 > WAIT_LIMIT_S)
            throw Error(22, std::string("afesp_comm: timed out waiting for the other ranks (") + what + ")");
    }
}

void seg_barrier(Comm* c)
{
    SegHeader* h = (SegHeader*)c->seg;
    const uint32_t g = h->gen.load(std::memory_order_acquire);
    if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->world) {
        h->count.store(0, std::memory_order_relaxed);
        h->gen.store(g + 1, std::memory_order_release);
    } else {
        wait_for([&] { return h->gen.load(std::memory_order_acquire) != g; }, "barrier");
    }
}

double* slot(Comm* c, int r) { return (double*)((char*)c->seg + SEG_HEADER_BYTES) + (int64_t)r * c->slot_doubles; }

void read_exact(const std::string& path, void* buf, size_t n, const char* what)
{
    wait_for([&] {
        struct stat st;
        return stat(path.c_str(), &st) == 0 && (size_t)st.st_size >= n;
    }, what);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f || fread(buf, 1, n, f) != n) {
        if (f) fclose(f);
        throw Error(23, "afesp_comm: cannot read bootstrap file " + path);
    }
    fclose(f);
}

void write_then_rename(const std::string& path, const void* buf, size_t n)
{
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(buf, 1, n, f) != n) {
        if (f) fclose(f);
        throw Error(23, "afesp_comm: cannot write bootstrap file " + tmp);
    }
    fclose(f);
    if (rename(tmp.c_str(), path.c_str()) != 0) throw Error(23, "afesp_comm: cannot publish bootstrap file " + path);
}

void ensure_pinned(Comm* c, int64_t n)
{
    if (c->pinned_doubles >= n) return;
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->pinned = nullptr;
    AFESP_HIP(hipHostMalloc((void**)&c->pinned, sizeof(double) * (size_t)n, hipHostMallocDefault));
    c->pinned_doubles = n;
}

}  // namespace

void comm_unique_id(char id[128])
{
    NcclId x;
    nccl_check(rccl().GetUniqueId(&x), "ncclGetUniqueId");
    memcpy(id, x.internal, 128);
}

Comm* comm_create(Context& cx, int rank, int world, int transport, const char* bootstrap_path, const char* unique_id)
{
    if (world < 1 || rank < 0 || rank >= world) throw Error(1, "afesp_comm_init: bad rank / world");
    if (transport != 0 && transport != 1) throw Error(1, "afesp_comm_init: unknown transport");
    Comm* c = new Comm();
    c->rank = rank; c->world = world; c->transport = transport;
    try {
        if (transport == 0) {
            NcclId id;
            if (unique_id) {
                memcpy(id.internal, unique_id, 128);
            } else if (world == 1) {
                nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
            } else {
                if (!bootstrap_path || !bootstrap_path[0])
                    throw Error(1, "afesp_comm_init: RCCL transport with world > 1 needs a unique id or a bootstrap path");
                c->path = bootstrap_path;
                if (rank == 0) {
                    nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
                    write_then_rename(c->path, id.internal, 128);
                } else {
                    read_exact(c->path, id.internal, 128, "RCCL unique id");
                }
            }
            AFESP_HIP(hipSetDevice(cx.device));
            nccl_check(rccl().CommInitRank(&c->nccl_comm, world, id, rank), "ncclCommInitRank");
            if (rank == 0 && !c->path.empty()) (void)unlink(c->path.c_str());   // every rank has joined: the id is spent
            c->dev_small = cx.alloc(64);
            ensure_pinned(c, 64);
        } else {
            c->slot_doubles = SLOT_DOUBLES;
            c->seg_bytes = SEG_HEADER_BYTES + (size_t)world * (size_t)SLOT_DOUBLES * sizeof(double);
            if (world > 1) {
                if (!bootstrap_path || !bootstrap_path[0]) throw Error(1, "afesp_comm_init: HOST transport needs a bootstrap path");
                c->path = bootstrap_path;
                if (rank == 0) {
                    const std::string tmp = c->path + ".tmp";
                    c->fd = open(tmp.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
                    if (c->fd < 0 || ftruncate(c->fd, (off_t)c->seg_bytes) != 0) throw Error(23, "afesp_comm: cannot create " + tmp);
                    c->seg = mmap(nullptr, c->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
                    if (c->seg == MAP_FAILED) { c->seg = nullptr; throw Error(23, "afesp_comm: mmap failed for " + tmp); }
                    SegHeader* h = new (c->seg) SegHeader();
                    h->world = (uint32_t)world;
                    h->slot_doubles = SLOT_DOUBLES;
                    h->count.store(0);
                    h->gen.store(0);
                    h->magic.store(SEG_MAGIC, std::memory_order_release);
                    if (rename(tmp.c_str(), c->path.c_str()) != 0) throw Error(23, "afesp_comm: cannot publish " + c->path);
                } else {
                    wait_for([&] {
                        struct stat st;
                        return stat(c->path.c_str(), &st) == 0 && (size_t)st.st_size == c->seg_bytes;
                    }, "shared segment");
                    c->fd = open(c->path.c_str(), O_RDWR);
                    if (c->fd < 0) throw Error(23, "afesp_comm: cannot open " + c->path);
                    c->seg = mmap(nullptr, c->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
                    if (c->seg == MAP_FAILED) { c->seg = nullptr; throw Error(23, "afesp_comm: mmap failed for " + c->path); }
                    SegHeader* h = (SegHeader*)c->seg;
                    wait_for([&] { return h->magic.load(std::memory_order_acquire) == SEG_MAGIC; }, "segment header");
                    if (h->world != (uint32_t)world) throw Error(1, "afesp_comm_init: the ranks disagree about the world size");
                }
                seg_barrier(c);
                if (rank == 0) (void)unlink(c->path.c_str());   // the mappings keep the segment alive
            }
            ensure_pinned(c, 1 << 16);
        }
    } catch (...) {
        comm_destroy(c);
        throw;
    }
    return c;
}

void comm_destroy(Comm* c)
{
    if (!c) return;
    if (c->nccl_comm) (void)rccl().CommDestroy(c->nccl_comm);
    if (c->seg) (void)munmap(c->seg, c->seg_bytes);
    if (c->fd >= 0) (void)close(c->fd);
    if (c->pinned) (void)hipHostFree(c->pinned);
    delete c;
}

void comm_barrier(Context& cx, Comm* c)
{
    if (!c || c->world == 1) return;
    double one = 1.0;
    comm_allreduce_host(cx, c, &one, 1);
}

void comm_allreduce_host(Context& cx, Comm* c, double* host, int64_t n)
{
    if (!c || n <= 0) return;
    if (c->transport == 0) {
        // few scalars: through a small device buffer on the context stream
        for (int64_t x0 = 0; x0 < n; x0 += 64) {
            const int64_t len = std::min<int64_t>(64, n - x0);
            memcpy(c->pinned, host + x0, sizeof(double) * (size_t)len);
            AFESP_HIP(hipMemcpyAsync(c->dev_small, c->pinned, sizeof(double) * (size_t)len, hipMemcpyHostToDevice, cx.stream));
            nccl_check(rccl().AllReduce(c->dev_small, c->dev_small, (size_t)len, NCCL_FLOAT64, NCCL_SUM, c->nccl_comm, cx.stream),
                       "ncclAllReduce");
            AFESP_HIP(hipMemcpyAsync(c->pinned, c->dev_small, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, cx.stream));
            cx.sync();
            memcpy(host + x0, c->pinned, sizeof(double) * (size_t)len);
        }
        return;
    }
    if (c->world == 1) return;
    for (int64_t x0 = 0; x0 < n; x0 += c->slot_doubles) {
        const int64_t len = std::min<int64_t>(c->slot_doubles, n - x0);
        memcpy(slot(c, c->rank), host + x0, sizeof(double) * (size_t)len);
        seg_barrier(c);
        for (int64_t x = 0; x < len; ++x) {
            double s = 0.0;
            for (int r = 0; r < c->world; ++r) s += slot(c, r)[x];   // fixed order: identical on every rank
            host[x0 + x] = s;
        }
        seg_barrier(c);
    }
}

void comm_allreduce_dev(Context& cx, Comm* c, double* dev, int64_t n)
{
    if (!c || n <= 0) return;
    if (c->transport == 0) {
        if (c->world == 1) return;
        nccl_check(rccl().AllReduce(dev, dev, (size_t)n, NCCL_FLOAT64, NCCL_SUM, c->nccl_comm, cx.stream), "ncclAllReduce");
        return;
    }
    if (c->world == 1) return;
    const int64_t chunk = (int64_t)1 << 22;
    ensure_pinned(c, std::min<int64_t>(chunk, n));
    for (int64_t x0 = 0; x0 < n; x0 += chunk) {
        const int64_t len = std::min<int64_t>(chunk, n - x0);
        AFESP_HIP(hipMemcpyAsync(c->pinned, dev + x0, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, cx.stream));
        cx.sync();
        comm_allreduce_host(cx, c, c->pinned, len);
        AFESP_HIP(hipMemcpyAsync(dev + x0, c->pinned, sizeof(double) * (size_t)len, hipMemcpyHostToDevice, cx.stream));
        cx.sync();
    }
}

}  // namespace afesp
