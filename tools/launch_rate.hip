// launch_rate.hip -- how many kernel launches per second the host gets into 1..8 streams from as many threads (is a small system's
// iteration, bound by the launch rate of one thread, worth feeding from several?).  build: hipcc -O3 --offload-arch=gfx950 -pthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(double* x, int n) { if (threadIdx.x < n) x[threadIdx.x + 64 * blockIdx.x] += 1.0; }
int main()
{
    double* buf; hipMalloc(&buf, 1 << 20);
    const int per = 4000;
    for (int nt : {1, 2, 3, 4, 6, 8}) {
        std::vector<hipStream_t> st(nt);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        // warm
        for (auto& s : st) { hipLaunchKernelGGL(tiny, dim3(4), dim3(64), 0, s, buf, 64); hipStreamSynchronize(s); }
        std::atomic<int> go{0};
        std::vector<std::thread> th;
        std::vector<double> host_us(nt);
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                while (!go.load()) {}
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < per; ++i) hipLaunchKernelGGL(tiny, dim3(4), dim3(64), 0, st[t], buf + 4096 * t, 64);
                host_us[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                hipStreamSynchronize(st[t]);
            });
        auto w0 = std::chrono::steady_clock::now();
        go.store(1);
        for (auto& x : th) x.join();
        const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
        double h = 0; for (double x : host_us) h += x / nt;
        printf("%d thread(s) x %d launches: host %.2f us per launch per thread, wall %.2f us per launch overall (%.2f M launches/s)\n", nt, per,
               h / per, wall / (per * nt), per * nt / wall);
        for (auto& s : st) hipStreamDestroy(s);
    }
    // one thread feeding 6 streams round-robin
    {
        const int nt = 6;
        std::vector<hipStream_t> st(nt);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < per * nt; ++i) hipLaunchKernelGGL(tiny, dim3(4), dim3(64), 0, st[i % nt], buf + 4096 * (i % nt), 64);
        const double host = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        for (auto& s : st) hipStreamSynchronize(s);
        const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("1 thread -> 6 streams round-robin: host %.2f us per launch, wall %.2f us per launch\n", host / (per * nt), wall / (per * nt));
    }
    return 0;
}
