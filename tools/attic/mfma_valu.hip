// How much VALU work can be co-issued with back-to-back v_mfma_f64_16x16x4_f64 before the matrix pipe starves?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NV, int KIND>
__global__ __launch_bounds__(256, 2) void k(double* out, unsigned long long* clk, int iters)
{
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 0.5, b = 1.0 - threadIdx.x * 1e-4;
    unsigned x[8]; unsigned long long y[8]; double z[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x + i; y[i] = threadIdx.x * 3ull + i; z[i] = i * 0.5 + threadIdx.x; }
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                if (KIND == 0) x[q & 7] = x[q & 7] * 3u + (unsigned)it;              // 32-bit int VALU
                else if (KIND == 1) y[q & 7] = (y[q & 7] << 3) + y[(q + 1) & 7];     // 64-bit int VALU (v_lshl_add_u64)
                else z[q & 7] = z[q & 7] > 1e9 ? 0.0 : z[q & 7] + 1.0;               // f64 add + cndmask
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i] + (double)y[i] + z[i];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NV, int KIND>
void run(int blocks, int iters)
{
    double* d; unsigned long long* c;
    CK(hipMalloc(&d, sizeof(double) * blocks * 256));
    CK(hipMalloc(&c, sizeof(unsigned long long) * blocks * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<NV, KIND><<<blocks, 256>>>(d, c, 10);
    CK(hipEventRecord(e0));
    k<NV, KIND><<<blocks, 256>>>(d, c, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double fl = (double)blocks * 4 * iters * 4 * 2048.0;
    const char* kn[] = {"int32", "int64", "f64add+sel"};
    printf("waves/SIMD=%d  %2d x %-10s per MFMA: %7.2f TFLOP/s\n", blocks / 256, NV, kn[KIND], fl / ms / 1e9);
    CK(hipFree(d)); CK(hipFree(c));
}
int main()
{
    run<0, 0>(256, 40000); run<0, 0>(512, 40000);
    run<2, 0>(256, 40000); run<4, 0>(256, 40000); run<8, 0>(256, 40000); run<16, 0>(256, 40000);
    run<2, 0>(512, 40000); run<4, 0>(512, 40000); run<8, 0>(512, 40000); run<16, 0>(512, 40000);
    run<2, 1>(512, 40000); run<4, 1>(512, 40000); run<8, 1>(512, 40000);
    run<2, 2>(512, 40000); run<4, 2>(512, 40000); run<8, 2>(512, 40000);
    return 0;
}
