// cumask_probe.hip -- what a stream restricted to R compute units (hipExtStreamCreateWithCUMask) reads from HBM, and whether the
// mask bits are dealt round-robin over the 8 XCDs (bit k -> XCD k % 8).  build: hipcc -O3 --offload-arch=gfx950 -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_kernel(const double2* __restrict__ src, int64_t n, double* out, unsigned* where)
{
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) out[0] = acc;
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        // HW_ID: cu_id bits 11:8, sh_id 12, se_id 15:13 (gfx9)
        atomicAdd(where + (xcc & 7) * 64 + ((hw >> 8) & 63), 1u);
    }
}

int main(int argc, char** argv)
{
    const int64_t bytes = (int64_t)4 << 30, n = bytes / 16;
    double2* buf; double* out; unsigned* where;
    CK(hipMalloc((void**)&buf, bytes)); CK(hipMalloc((void**)&out, 8)); CK(hipMalloc((void**)&where, 8 * 64 * 4));
    CK(hipMemset(buf, 0, bytes));
    const int Rs[] = {256, 8, 16, 24, 32, 64, 128, 240};
    for (int R : Rs) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // the LAST R bits (the GEMM keeps the first 256 - R)
        for (int b = 256 - R; b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
        hipStream_t st;
        CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        CK(hipMemsetAsync(where, 0, 8 * 64 * 4, st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int grid = R * 8;
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, buf, n, out, where);   // warm
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, st, buf, n, out, where);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned> h(8 * 64);
        CK(hipMemcpy(h.data(), where, 8 * 64 * 4, hipMemcpyDeviceToHost));
        int per_xcd[8] = {0}, cus = 0;
        for (int x = 0; x < 8; ++x) for (int c = 0; c < 64; ++c) if (h[x * 64 + c]) { ++per_xcd[x]; ++cus; }
        printf("R %3d  %8.1f GB/s  %6.1f GB/s per CU   distinct (xcd, hw cu) slots %3d, per XCD:", R, bytes / ms * 1e-6, bytes / ms * 1e-6 / R, cus);
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcd[x]);
        printf("\n");
        CK(hipStreamDestroy(st));
    }
    return 0;
}
