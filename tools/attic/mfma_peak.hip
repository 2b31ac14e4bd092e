// Practical fp64 MFMA ceiling on this device: register-resident v_mfma_f64_16x16x4_f64 loop, with in-kernel clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256, (NACC <= 8 ? 2 : 1)) void k(double* out, unsigned long long* clk, int iters)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 0.5, b = 1.0 - threadIdx.x * 1e-4;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void k2(double* out, unsigned long long* clk, int iters)
{
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 0.5, b = 1.0 - threadIdx.x * 1e-4;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void k4(double* out, unsigned long long* clk, int iters)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = threadIdx.x * 1e-3 + 0.5, b = 1.0 - threadIdx.x * 1e-4;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
void run4(int blocks, int iters, const char* tag)
{
    double* d; unsigned long long* c;
    CK(hipMalloc(&d, sizeof(double) * blocks * 256));
    CK(hipMalloc(&c, sizeof(unsigned long long) * blocks * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k4<NACC><<<blocks, 256>>>(d, c, 10);
    CK(hipEventRecord(e0));
    k4<NACC><<<blocks, 256>>>(d, c, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost));
    double fl = (double)blocks * 4 * iters * NACC * 512.0;
    printf("MFMA 4x4x4f64 %-12s blocks=%4d nacc=%2d: %8.3f ms %7.2f TFLOP/s  clock %.3f GHz  %.1f cycles per MFMA per wave\n", tag, blocks, NACC,
           ms, fl / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / ((double)iters * NACC));
    CK(hipFree(d)); CK(hipFree(c));
}
template <int NACC>
__global__ __launch_bounds__(256) void kv(double* out, unsigned long long* clk, int iters)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i * 0.25;
    double a = threadIdx.x * 1e-9 + 0.999999, b = 1e-7 - threadIdx.x * 1e-12;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
void runv(int blocks, int iters, const char* tag)
{
    double* d; unsigned long long* c;
    CK(hipMalloc(&d, sizeof(double) * blocks * 256));
    CK(hipMalloc(&c, sizeof(unsigned long long) * blocks * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kv<NACC><<<blocks, 256>>>(d, c, 10);
    CK(hipEventRecord(e0));
    kv<NACC><<<blocks, 256>>>(d, c, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost));
    double fl = (double)blocks * 256 * iters * NACC * 2.0;
    printf("VALU fma_f64 %-12s blocks=%4d nacc=%2d: %8.3f ms %7.2f TFLOP/s  clock %.3f GHz  %.2f cycles per v_fma_f64 per wave\n", tag, blocks, NACC,
           ms, fl / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / ((double)iters * NACC));
    CK(hipFree(d)); CK(hipFree(c));
}
template <int NACC, bool TWO = false>
void run(int blocks, int iters, const char* tag)
{
    double* d; unsigned long long* c;
    CK(hipMalloc(&d, sizeof(double) * blocks * 256));
    CK(hipMalloc(&c, sizeof(unsigned long long) * blocks * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (TWO) k2<NACC><<<blocks, 256>>>(d, c, 10); else k<NACC><<<blocks, 256>>>(d, c, 10);
    CK(hipEventRecord(e0));
    if (TWO) k2<NACC><<<blocks, 256>>>(d, c, iters); else k<NACC><<<blocks, 256>>>(d, c, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost));
    double fl = (double)blocks * 4 * iters * NACC * 2048.0;
    double ghz = (double)h[0] / ((double)h[1] * 10.0) ;   // memrealtime = 100 MHz -> ns*... cycles per 10ns tick
    double cyc_per_mfma_wave = (double)h[0] / ((double)iters * NACC);
    printf("%-16s blocks=%4d nacc=%2d: %8.3f ms %7.2f TFLOP/s  clock %.3f GHz  %.1f shader-cycles per MFMA per wave\n", tag, blocks, NACC,
           ms, fl / ms / 1e9, ghz, cyc_per_mfma_wave);
    CK(hipFree(d)); CK(hipFree(c));
}
int main()
{
    run<16>(256, 20000, "1 wave/SIMD");
    run<8>(256, 20000, "1 wave/SIMD n8");
    run<16, true>(256, 20000, "1w, 256-reg");
    run<16, true>(512, 20000, "2w, 256-reg");
    run<12, true>(512, 20000, "2w, 256-reg");
    run<16>(512, 20000, "2 waves/SIMD");
    run<8>(512, 20000, "2 waves/SIMD");
    run<4>(1024, 40000, "4 waves/SIMD");
    run<2>(2048, 40000, "8 waves/SIMD");
    run<1>(256, 100000, "dependent chain");
    run<16>(1, 20000, "one CU only");
    run4<16>(256, 100000, "1 wave/SIMD");
    run4<16>(512, 100000, "2 waves/SIMD");
    run4<16>(1024, 100000, "4 waves/SIMD");
    runv<16>(256, 400000, "1 wave/SIMD");
    runv<16>(512, 400000, "2 waves/SIMD");
    runv<16>(1024, 400000, "4 waves/SIMD");
    runv<16>(2048, 200000, "8 waves/SIMD");
    return 0;
}
