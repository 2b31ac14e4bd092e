// Does LDS / global traffic issued between v_mfma_f64_16x16x4_f64 slow the matrix pipe?  (2 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
// MODE 0: nothing; 1: NL x ds_read_b64 per MFMA (results feed the next MFMA operands); 2: NL x ds_read_b128 per 2 MFMA;
// 3: NL x global_load_dwordx2 per MFMA (L2-resident); 4: NL x ds_write_b64 per MFMA
template <int MODE, int NL>
__global__ __launch_bounds__(256, 2) void k(double* out, const double* g, int iters)
{
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3 + 0.5, b = 1.0 - threadIdx.x * 1e-4;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < NL; ++q) { double x = lds[(idx + 64 * q + 256 * i) & 4095]; if (q & 1) a = x; else b = x; }
            } else if (MODE == 2) {
                if (i & 1) {
#pragma unroll
                    for (int q = 0; q < NL; ++q) { v2d x = *(const v2d*)&lds[(2 * idx + 512 * q) & 4094]; a = x[0]; b = x[1]; }
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < NL; ++q) { double x = g[(idx + 256 * q + 1024 * i + 4096 * (it & 15)) & 65535]; if (q & 1) a = x; else b = x; }
            } else if (MODE == 4) {
#pragma unroll
                for (int q = 0; q < NL; ++q) lds[(idx + 256 * q + 1024 * i) & 4095] = a;
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + a + b + lds[threadIdx.x];
}
template <int MODE, int NL>
void run(const char* tag)
{
    const int blocks = 512, iters = 20000;
    double *d, *g;
    CK(hipMalloc(&d, sizeof(double) * blocks * 256));
    CK(hipMalloc(&g, sizeof(double) * 65536));
    CK(hipMemset(g, 0, sizeof(double) * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k<MODE, NL><<<blocks, 256>>>(d, g, 10);
    CK(hipEventRecord(e0));
    k<MODE, NL><<<blocks, 256>>>(d, g, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %7.2f TFLOP/s\n", tag, (double)blocks * 4 * iters * 4 * 2048.0 / ms / 1e9);
    CK(hipFree(d)); CK(hipFree(g));
}
int main()
{
    run<0, 0>("mfma only");
    run<1, 1>("1 ds_read_b64 / mfma");
    run<1, 2>("2 ds_read_b64 / mfma");
    run<1, 4>("4 ds_read_b64 / mfma");
    run<2, 1>("1 ds_read_b128 / 2 mfma");
    run<2, 2>("2 ds_read_b128 / 2 mfma");
    run<3, 1>("1 global_load_b64 / mfma");
    run<3, 2>("2 global_load_b64 / mfma");
    run<4, 1>("1 ds_write_b64 / mfma");
    run<4, 2>("2 ds_write_b64 / mfma");
    return 0;
}
