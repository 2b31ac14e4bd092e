// grid_barrier_probe.hip -- what one device-wide barrier costs on MI355X, against the ~4.5 us a kernel boundary costs: the number
// behind the choice "one launch per dependency level" (csrc/fused.hip) instead of one cooperative kernel with barriers between levels.
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe_bin && tools/grid_barrier_probe_bin
// A sense-reversing barrier on one agent-scope counter: every workgroup's thread 0 adds 1 (release), the last arrival flips the
// sense word, the others spin on it (acquire).  All workgroups are co-resident (grid <= workgroups the device holds at once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void barrier_loop(unsigned* count, unsigned* sense, int nbar, unsigned long long* cycles)
{
    const unsigned nwg = gridDim.x;
    unsigned my = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int b = 0; b < nbar; ++b) {
        __syncthreads();
        if (threadIdx.x == 0) {
            my ^= 1u;
            if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1 == nwg) {
                __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sense, my, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(sense, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != my) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = __builtin_readcyclecounter() - t0;
}
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

int main()
{
    unsigned *count, *sense;
    unsigned long long* cyc;
    hipMalloc(&count, 4); hipMalloc(&sense, 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int nbar = 200;
    for (int grid : {64, 256, 512, 1024, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(count, 0, 4); hipMemset(sense, 0, 4);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(barrier_loop, dim3(grid), dim3(256), 0, 0, count, sense, nbar, cyc);
            hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("grid %5d workgroups of 256: %d barriers in %.1f us -> %.2f us per barrier\n", grid, nbar, ms * 1e3, ms * 1e3 / nbar);
        }
    }
    // kernel boundaries on one stream for comparison: n empty kernels back to back
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < nbar; ++k) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, (int*)nullptr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("%d empty kernels (256 x 256) back to back: %.2f us per kernel boundary\n", nbar, ms * 1e3 / nbar);
    }
    return 0;
}
