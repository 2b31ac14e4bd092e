// store_probe.hip -- what a GEMM tile's C epilogue costs next to its MFMAs (round 6).
// Each wave runs `tiles` tiles of `steps` x 64 v_mfma_f64_16x16x4_f64 on 16 accumulators (two 4-wave workgroups per CU, as
// tgemm_kernel) and writes its 64 x 64 part of a 128 x 128 tile in the cube-blocked layout of the (T) blocks:
//   MODE 0  no stores                       MODE 1  32 global_store_dwordx4 behind the last MFMA (today's epilogue)
//   MODE 2  the last step's MFMAs ordered accumulator pair by pair, the pair's four stores between the next pair's MFMAs
//   MODE 3  64 global_atomic_add_f64 behind the last MFMA            MODE 4  as 2 with atomics (8 per pair)
//   MODE 5  as 1, every store instruction one contiguous KiB          MODE 6  as 2, spread over the last TWO steps' worth of gaps
// hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/store_probe_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(I, J) acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[I][J], 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(double* C, int64_t cmask, unsigned long long* clk, int tiles, int steps)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave & 1, wn = wave >> 1;
    const int fm = lane & 15, ff = lane >> 4;
    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1e-3 * (1 + (lane & 7)), b = 1e-3 * (2 + (lane >> 3));
    // element offset of (row, col) of the tile: col = a (fastest, 8 per cube), row = b (8 per cube), cubes of 512
    auto off = [&](int row, int col) { return (int64_t)512 * (col >> 3) + (col & 7) + 8 * (row & 7) + (int64_t)512 * 25 * (row >> 3); };
    int64_t cn[2], cm[16];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) cn[jj] = off(0, wn * 64 + 32 * jj + 2 * fm);
#pragma unroll
    for (int x = 0; x < 16; ++x) cm[x] = off(wm * 64 + 16 * (x >> 2) + 4 * (x & 3) + ff, 0);
    if (MODE == 5) {   // one contiguous KiB per instruction
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) cn[jj] = (int64_t)jj * 128 + 2 * lane;
#pragma unroll
        for (int x = 0; x < 16; ++x) cm[x] = (int64_t)(wave * 16 + x) * 256;
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long epi = 0;
    for (int tl = 0; tl < tiles; ++tl) {
        double* const Cg = C + ((((int64_t)blockIdx.x * tiles + tl) * ((int64_t)1 << 22)) & cmask);   // a tile's region: every tile somewhere else
        const int full = (MODE == 2 || MODE == 4) ? steps - 1 : (MODE == 6 ? steps - 2 : steps);
        for (int s = 0; s < full; ++s) {
#pragma unroll
            for (int x = 0; x < 64; ++x) MF((x >> 2) & 3, x & 3);
            SB;
        }
#define ST(I, JJ, R) *reinterpret_cast<v2d*>(Cg + cm[4 * (I) + (R)] + cn[JJ]) = (v2d){acc[I][2 * (JJ)][R], acc[I][2 * (JJ) + 1][R]}
#define AT(I, J, R) unsafeAtomicAdd(Cg + cm[4 * (I) + (R)] + cn[(J) >> 1] + ((J) & 1), acc[I][J][R])
        const unsigned long long e0 = __builtin_amdgcn_s_memtime();
        if (MODE == 1 || MODE == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) ST(i, jj, r);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) AT(i, j, r);
        } else if (MODE == 2 || MODE == 4 || MODE == 6) {
            // last step(s): accumulator row p gets its MFMAs (four accumulators in turn), the stores of row p - 1 ride between them
            constexpr int NM = MODE == 6 ? 8 : 4;   // MFMAs per accumulator in the spread region
#pragma unroll
            for (int p = 0; p <= 4; ++p) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {   // eight slots of NM / 2 MFMAs; slot q carries store (jj = q & 1, r = q >> 1) of the previous row
                    if (p < 4) {
#pragma unroll
                        for (int m = 0; m < NM / 2; ++m) { MF(p, (q * (NM / 2) + m) & 3); SB; }
                    }
                    if (p > 0) {
                        if (MODE == 4) { AT(p - 1, 2 * (q & 1), q >> 1); SB; AT(p - 1, 2 * (q & 1) + 1, q >> 1); SB; }
                        else { ST(p - 1, q & 1, q >> 1); SB; }
                    }
                }
            }
        }
        epi += __builtin_amdgcn_s_memtime() - e0;
        SB;
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678) C[0] = s;
    if (lane == 0) {
        unsigned long long* d = clk + ((size_t)blockIdx.x * 4 + wave) * 4;
        d[0] = c1 - c0; d[1] = r1 - r0; d[2] = epi;
    }
}

template <int MODE>
static void run(double* C, int64_t cmask, unsigned long long* clk, int blocks, int tiles, int steps, const char* tag)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<MODE><<<blocks, 256>>>(C, cmask, clk, 2, steps);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    double epi = 0, cyc = 0, ghz = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        probe<MODE><<<blocks, 256>>>(C, cmask, clk, tiles, steps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * blocks * 16);
            CK(hipMemcpy(h, clk, sizeof(unsigned long long) * blocks * 16, hipMemcpyDeviceToHost));
            epi = cyc = 0;
            for (int i = 0; i < blocks * 4; ++i) { epi += (double)h[4 * i + 2]; cyc += (double)h[4 * i]; }
            ghz = (double)h[0] / ((double)h[1] * 10.0);
            epi /= (double)blocks * 4 * tiles; cyc /= (double)blocks * 4 * tiles;
            free(h);
        }
    }
    const double fl = (double)blocks * 4 * tiles * steps * 64 * 2048.0;
    printf("%-44s blocks %4d tiles %3d steps %2d: %8.3f ms %6.2f TFLOP/s  %.3f GHz  %8.0f cycles per tile and wave, %6.0f in the epilogue region\n", tag, blocks,
           tiles, steps, best, fl / best / 1e9, ghz, cyc, epi);
}

int main(int argc, char** argv)
{
    const int tiles = argc > 1 ? atoi(argv[1]) : 24, steps = argc > 2 ? atoi(argv[2]) : 28;
    const int64_t bytes = (int64_t)8 << 30;
    double* C; unsigned long long* clk;
    CK(hipMalloc(&C, bytes + ((int64_t)64 << 20)));
    CK(hipMemset(C, 0, bytes + ((int64_t)64 << 20)));
    CK(hipMalloc(&clk, sizeof(unsigned long long) * 1024 * 16));
    const int64_t cmask = bytes / 8 - 1;
    for (int blocks : {512, 256}) {
        run<0>(C, cmask, clk, blocks, tiles, steps, "0 no stores");
        run<1>(C, cmask, clk, blocks, tiles, steps, "1 32 stores x4 behind the last MFMA");
        run<5>(C, cmask, clk, blocks, tiles, steps, "5 same, one contiguous KiB per instruction");
        run<2>(C, cmask, clk, blocks, tiles, steps, "2 stores between the last step's MFMAs");
        run<6>(C, cmask, clk, blocks, tiles, steps, "6 stores between the last two steps' MFMAs");
        run<3>(C, cmask, clk, blocks, tiles, steps, "3 64 atomic adds behind the last MFMA");
        run<4>(C, cmask, clk, blocks, tiles, steps, "4 atomic adds between the last step's MFMAs");
        run<0>(C, cmask, clk, blocks, tiles, steps, "0 no stores (again)");
    }
    return 0;
}
