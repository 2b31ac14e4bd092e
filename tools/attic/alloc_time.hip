// alloc_time.hip -- hipMalloc / first touch / hipFree of one large buffer, three rounds.
// build: hipcc --offload-arch=gfx950 -O2 tools/alloc_time.hip -o tools/alloc_time ; run: tools/alloc_time <GB>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(double* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
int main(int argc, char** argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 47.6;
    size_t bytes = (size_t)(gb * 1e9);
    for (int r = 0; r < 3; ++r) {
        void* p = nullptr;
        double t0 = now();
        hipError_t e = hipMalloc(&p, bytes);
        double t1 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (double*)p, bytes / 8);
        (void)hipDeviceSynchronize();
        double t2 = now();
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (double*)p, bytes / 8);
        (void)hipDeviceSynchronize();
        double t3 = now();
        (void)hipFree(p);
        double t4 = now();
        printf("%.1f GB: malloc %.1f ms (err %d)  first touch %.1f ms  second touch %.1f ms  free %.1f ms\n", gb, (t1 - t0) * 1e3, (int)e,
               (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    }
    return 0;
}
