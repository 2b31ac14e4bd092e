// first_use.h -- every FIRST use of a kernel function in this process happens under one lock.
//
// The HIP runtime loads a translation unit's device code and resolves a kernel function lazily, on the first launch (or attribute
// query) that names it.  Two host threads that first-use the same function at the same time -- two contexts on two threads, a
// context's start-up thread and its caller -- can end in the runtime's abort "Cannot find Symbol with name: ..." (hip_global.cpp;
// round 5: seen inside afesp_synthetic_init, symbol slice_phys_kernel of kernels.hip, while the start-up thread asked for the
// attributes of the same kernel).  Closed by construction here, for every kernel of every unit, not by a list:
//   * AFESP_KLAUNCH -- the only way this library launches a kernel -- keeps one flag per launch site (per template instantiation)
//     and device; while the flag is clear the site takes first_use_mutex(), asks for the function's attributes (which loads the
//     unit's code object and resolves the function) and sets the flag.  Afterwards a launch costs one relaxed load more.
//   * first_use_touch -- what the start-up thread's preload lists call -- resolves under the same lock.
// So any two first uses (of one function, or of two functions of one code object) are ordered, whoever makes them; a launch whose
// site flag is set names a function that was resolved under the lock before.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "knobs.h"

#include <atomic>
#include <mutex>

namespace afesp {

inline std::mutex& first_use_mutex()
{
    static std::mutex m;
    return m;
}
// the device the calling thread works on: set by the C-ABI entry points and the start-up thread (-1: ask the runtime)
inline int& first_use_tls_device()
{
    static thread_local int dev = -1;
    return dev;
}
inline uint64_t first_use_bit()
{
    int d = first_use_tls_device();
    if (d < 0 && hipGetDevice(&d) != hipSuccess) d = 0;
    return (uint64_t)1 << (d & 63);
}
inline void first_use_touch(const void* fn)
{
    std::lock_guard<std::mutex> lk(first_use_mutex());
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, fn);
    (void)hipGetLastError();
}
inline void first_use_resolve(const void* fn, std::atomic<uint64_t>& seen, uint64_t bit)
{
    first_use_touch(fn);
    seen.fetch_or(bit, std::memory_order_release);
}
// how many launch sites have resolved their function so far (tests: the lock was taken where it had to be)
inline std::atomic<uint64_t>& first_use_count()
{
    static std::atomic<uint64_t> n{0};
    return n;
}

}  // namespace afesp

// (a kernel name with template commas goes in parenthesised, as for hipLaunchKernelGGL)
#define AFESP_KLAUNCH(kernel, grid, block, shmem, stream, ...)                                                          \
    do {                                                                                                                \
        static std::atomic<uint64_t> afesp_seen_{0};                                                                    \
        const uint64_t afesp_bit_ = ::afesp::first_use_bit();                                                           \
        if (!(afesp_seen_.load(std::memory_order_acquire) & afesp_bit_)) {                                              \
            ::afesp::first_use_resolve(reinterpret_cast<const void*>(kernel), afesp_seen_, afesp_bit_);                 \
            ::afesp::first_use_count().fetch_add(1, std::memory_order_relaxed);                                         \
        }                                                                                                               \
        kernel<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                                                    \
    } while (0)
