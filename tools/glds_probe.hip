// glds_probe.hip -- where does global_load_lds_dwordx4 put its bytes?  (diagnostic; hipcc --offload-arch=gfx950 -o tools/glds_probe_bin)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(const unsigned* src, unsigned* dump, unsigned m0val, int use_off)
{
    __shared__ __attribute__((aligned(1024))) unsigned lds[36864];   // 144 KiB
    const int t = threadIdx.x;
    for (int i = t; i < 36864; i += 256) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)lds);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    if (wave == 1) {
        const unsigned voff = (unsigned)((t & 63) * 16);
        const unsigned d = lds0 + m0val;
        if (use_off)
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:2048" :: "v"(voff), "s"(d), "s"(src) : "memory");
        else
            asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(voff), "s"(d), "s"(src) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = t; i < 36864; i += 256) dump[i] = lds[i];
    if (t == 0) dump[36864] = lds0;
}
int main()
{
    unsigned *src, *dump;
    std::vector<unsigned> h(4096), out(36865);
    for (int i = 0; i < 4096; ++i) h[i] = 0x10000u + i;
    hipMalloc(&src, 4096 * 4); hipMalloc(&dump, 36865 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const unsigned m0s[] = {0, 64512, 65536, 70000 / 16 * 16, 100000 / 16 * 16, 147456 - 1024};
    for (int use_off = 0; use_off < 2; ++use_off)
        for (unsigned m0 : m0s) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, src, dump, m0, use_off);
            hipDeviceSynchronize();
            hipMemcpy(out.data(), dump, 36865 * 4, hipMemcpyDeviceToHost);
            int first = -1, last = -1, n = 0;
            for (int i = 0; i < 36864; ++i) if (out[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; ++n; }
            printf("use_off=%d m0=%u lds0=%u: %d dwords changed, first byte %d last byte %d", use_off, m0, out[36864], n, first * 4, last * 4 + 3);
            if (first >= 0) printf("  first vals %x %x %x %x %x", out[first], out[first + 1], out[first + 2], out[first + 3], out[first + 4]);
            printf("\n");
        }
    return 0;
}
