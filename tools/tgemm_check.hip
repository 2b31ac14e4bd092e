// tgemm_check.hip -- tgemm_kernel on a random grouped problem against a host loop, or (-DTG_STAMPS, `big`) where its waves spend
// their cycles on config-5-like extents (diagnostic).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/tgemm_check.hip -o tools/tgemm_check_bin   [-DTG_STAMPS -o tools/tgemm_stamps_bin]
// usage: tgemm_check_bin [M Kc N0 N1]        tgemm_stamps_bin big [M Kc N ngroups]
#include "../a-fortran-electronic-structure-program_amd/csrc/tgemm.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace afesp;
static TgLaunchState g_tg;   // (one launcher state for this stand-alone program)
// TG_BM=128 / 96: the instantiation with 96-row tiles where the rows end (tgemm.h); unset: 128-row tiles throughout
static const int g_bm = getenv("TG_BM") ? atoi(getenv("TG_BM")) : 0;
static const int g_bme = g_bm ? g_bm : 128;

static int run_big(int argc, char** argv)
{
    const int M = argc > 2 ? atoi(argv[2]) : 40000, Kc = argc > 3 ? atoi(argv[3]) : 224, N = argc > 4 ? atoi(argv[4]) : 1000;
    const int ng = argc > 5 ? atoi(argv[5]) : 12;
    const int nk1 = Kc / 16, mt = (M + g_bme - 1) / g_bme, nt = (N + 127) / 128;
    const size_t na = (size_t)2 * M * Kc, nb = (size_t)2 * ng * N * Kc, nc = (size_t)M * N * ng;
    double *dA, *dB, *dC; uint32_t* d32; int64_t* d64; TgGroup* dg;
    hipMalloc(&dA, na * 8); hipMalloc(&dB, nb * 8); hipMalloc(&dC, nc * 8);
    hipMalloc(&d32, ((size_t)M + (size_t)ng * N + 256) * 4); hipMalloc(&d64, ((size_t)M + (size_t)ng * N + 128) * 8); hipMalloc(&dg, (ng + 1) * sizeof(TgGroup));
    std::vector<double> h(1 << 20);
    for (auto& x : h) x = (rand() % 2001 - 1000) / 1000.0;
    for (size_t o = 0; o < na; o += h.size()) hipMemcpy(dA + o, h.data(), std::min(h.size(), na - o) * 8, hipMemcpyHostToDevice);
    for (size_t o = 0; o < nb; o += h.size()) hipMemcpy(dB + o, h.data(), std::min(h.size(), nb - o) * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> t32((size_t)M + (size_t)ng * N);
    std::vector<int64_t> t64((size_t)M + (size_t)ng * N);
    // C: pairs of columns interleaved (tgemm.h c_pairs): C(m, n) at 2 m + (n & 1) + 2 M (n >> 1)
    const bool csmall = getenv("TG_CSMALL") != nullptr;   // diagnostic: every tile's stores land in the same 128 KiB (no HBM write traffic)
    for (int m = 0; m < M; ++m) { t32[m] = (uint32_t)((size_t)8 * Kc * m); t64[m] = csmall ? 2 * (m % 128) : 2 * m; }
    for (int n = 0; n < ng * N; ++n) { t32[(size_t)M + n] = (uint32_t)((size_t)8 * Kc * n); t64[(size_t)M + n] = csmall ? (n & 1) + (int64_t)2 * 128 * ((n >> 1) % 64) : (n & 1) + (int64_t)2 * M * (n >> 1); }
    hipMemcpy(d32, t32.data(), t32.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d64, t64.data(), t64.size() * 8, hipMemcpyHostToDevice);
    std::vector<TgGroup> g(ng + 1);
    const int gm = tgemm_group_m(M, nt, g_bme);
    int tile = 0;
    for (int q = 0; q < ng; ++q) {
        g[q].a1 = 0; g[q].a2 = (int64_t)M * Kc; g[q].b1 = 0; g[q].b2 = (int64_t)ng * N * Kc; g[q].c0 = 0;
        g[q].colB = d32 + M + (size_t)q * N; g[q].offCn = d64 + M + (size_t)q * N;
        g[q].N = N; g[q].ntiles = nt; g[q].tile_start = tile; g[q].nk1 = nk1; g[q].nk = (q % 6 == 5) ? nk1 : 2 * nk1;
        g[q].inv_width = tgemm_inverse(gm * nt);
        tile += mt * nt;
    }
    g[ng].tile_start = tile;
    hipMemcpy(dg, g.data(), (ng + 1) * sizeof(TgGroup), hipMemcpyHostToDevice);
    TgProblem p{dA, dB, dC, d32, d64, M, getenv("TG_NOPAIRS") == nullptr, getenv("TG_KT4") ? atoi(getenv("TG_KT4")) : 4};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double flop = 0.0;
    for (int q = 0; q < ng; ++q) flop += 2.0 * M * N * 16.0 * g[q].nk;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipError_t e = tgemm_launch(p, dg, ng, tile, nt, 0, g_tg, g_bm);
        hipEventRecord(e1, 0);
        hipError_t e2 = hipDeviceSynchronize();
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %s sync %s: %d tiles, %.3f ms, %.2f TF (executed, unpadded)\n", hipGetErrorString(e), hipGetErrorString(e2), tile, ms, flop / ms / 1e9);
    }
#ifdef TG_STAMPS
    std::vector<unsigned long long> st(512 * 4 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_tg_stamp), st.size() * 8);
    double tot = 0, bar = 0, epi = 0, steps = 0, tiles = 0, mx = 0; int n = 0;
    for (int b = 0; b < 512; ++b) for (int w = 0; w < 4; ++w) {
        const unsigned long long* d = &st[((size_t)b * 4 + w) * 8];
        if (!d[3]) continue;
        tot += d[0]; bar += d[1]; epi += d[2]; steps += d[3]; tiles += d[4]; mx = std::max(mx, (double)d[5]); ++n;
    }
    printf("stamps over %d waves: cycles per step %.0f (MFMA floor 8192 at two waves per SIMD), of which barrier %.0f; per tile: epilogue %.0f; longest barrier %.0f\n",
           n, tot / steps, bar / steps, epi / tiles, mx);
    { double lo = 0, hi = 0; int nl = 0, nh = 0; unsigned long long base = ~0ull;
      for (int b = 0; b < 512; ++b) { const unsigned long long t8 = st[((size_t)b * 4) * 8 + 6]; if (t8) base = std::min(base, t8); }
      for (int b = 0; b < 512; ++b) { const unsigned long long t8 = st[((size_t)b * 4) * 8 + 6]; if (!t8) continue; if (b < 256) { lo += t8 - base; ++nl; } else { hi += t8 - base; ++nh; } }
      printf("  mean time of the 8th tile's end after the earliest: workgroups 0..255 %.0f, 256..511 %.0f (x 10 ns)\n", nl ? lo / nl : 0.0, nh ? hi / nh : 0.0); }
    // when did the workgroups of an XCD (b & 7) finish their 8th tile?  spread in units of 10 ns; a K step is ~370 units
    for (int x = 0; x < 8; ++x) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int b = x; b < 512; b += 8) { const unsigned long long t8 = st[((size_t)b * 4) * 8 + 6]; if (t8) { lo = std::min(lo, t8); hi = std::max(hi, t8); } }
        printf("  XCD %d: workgroups finish their 8th tile within %llu x 10 ns of each other\n", x, hi > lo ? hi - lo : 0ull);
    }
    // distribution of per-wave barrier share
    for (int w = 0; w < 4; ++w) {
        double b2 = 0, s2 = 0;
        for (int b = 0; b < 512; ++b) { const unsigned long long* d = &st[((size_t)b * 4 + w) * 8]; b2 += d[1]; s2 += d[3]; }
        printf("  wave %d: barrier cycles per step %.0f\n", w, s2 ? b2 / s2 : 0.0);
    }
#endif
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 1 && !strcmp(argv[1], "big")) return run_big(argc, argv);
    const int M = argc > 1 ? atoi(argv[1]) : 300, Kc = argc > 2 ? atoi(argv[2]) : 48;
    const int Ns[2] = {argc > 3 ? atoi(argv[3]) : 200, argc > 4 ? atoi(argv[4]) : 70};
    const int nk1 = Kc / 16, ncol = Ns[0] + Ns[1];
    std::vector<double> A((size_t)2 * M * Kc), B((size_t)2 * ncol * Kc), C((size_t)M * ncol, -7.0), R((size_t)M * ncol, 0.0);
    srand(3);
    for (auto& x : A) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto& x : B) x = (rand() % 2001 - 1000) / 1000.0;
    // valid summation length Kv <= Kc of each run (TG_KV): zero padding behind it, as the (T) operands have
    const int Kv = getenv("TG_KV") ? atoi(getenv("TG_KV")) : Kc;
    for (size_t r = 0; r < A.size() / Kc; ++r) for (int k = Kv; k < Kc; ++k) A[r * Kc + k] = 0.0;
    for (size_t r = 0; r < B.size() / Kc; ++r) for (int k = Kv; k < Kc; ++k) B[r * Kc + k] = 0.0;
    std::vector<uint32_t> rowA(M), colB(ncol);
    std::vector<int64_t> offCm(M), offCn(ncol);
    for (int m = 0; m < M; ++m) { rowA[m] = (uint32_t)(8 * Kc * m); offCm[m] = m; }
    // C layout: plain column-major, or (TG_PAIRS=1; needs even N0, N1) pairs of columns interleaved: C(m, n) at 2 m + (n & 1) + 2 M (n >> 1)
    const bool pairs = getenv("TG_PAIRS") != nullptr;
    if (pairs) for (int m = 0; m < M; ++m) offCm[m] = 2 * m;
    for (int n = 0; n < ncol; ++n) { colB[n] = (uint32_t)(8 * Kc * ((n * 7) % ncol)); offCn[n] = pairs ? (n & 1) + (int64_t)2 * M * (n >> 1) : (int64_t)M * n; }   // 7 coprime to ncol assumed
    const int mt = (M + g_bme - 1) / g_bme;
    uint32_t* d32; int64_t* d64; double *dA, *dB, *dC; TgGroup* dg;
    hipMalloc(&d32, (M + ncol + 256) * 4); hipMalloc(&d64, (M + ncol + 128) * 8);
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dg, 3 * sizeof(TgGroup));
    hipMemcpy(d32, rowA.data(), M * 4, hipMemcpyHostToDevice); hipMemcpy(d32 + M, colB.data(), ncol * 4, hipMemcpyHostToDevice);
    hipMemcpy(d64, offCm.data(), M * 8, hipMemcpyHostToDevice); hipMemcpy(d64 + M, offCn.data(), ncol * 8, hipMemcpyHostToDevice);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    TgGroup g[3] = {};
    int mx = 0, tile = 0;
    for (int q = 0; q < 2; ++q) mx = std::max(mx, (Ns[q] + 127) / 128);
    const int gm = tgemm_group_m(M, mx, g_bme);
    for (int q = 0; q < 2; ++q) {
        g[q].a1 = 0; g[q].a2 = (int64_t)M * Kc; g[q].b1 = 0; g[q].b2 = (int64_t)ncol * Kc; g[q].c0 = 0;
        g[q].colB = d32 + M + (q ? Ns[0] : 0); g[q].offCn = d64 + M + (q ? Ns[0] : 0);
        g[q].N = Ns[q]; g[q].ntiles = (Ns[q] + 127) / 128; g[q].tile_start = tile; g[q].nk1 = nk1; g[q].nk = q ? (nk1 >= 2 ? nk1 : 2 * nk1) : 2 * nk1;
        g[q].inv_width = tgemm_inverse(gm * g[q].ntiles);
        tile += mt * g[q].ntiles;
    }
    g[2].tile_start = tile;
    hipMemcpy(dg, g, sizeof(g), hipMemcpyHostToDevice);
    // (offCn[n] = M n: columns are adjacent only when M == 1; pairs are exercised with TG_PAIRS=1, which lays C out column-pair-major)
    TgProblem p{dA, dB, dC, d32, d64, M, pairs, (Kv - (Kc - 16) + 3) / 4};
    hipError_t e = tgemm_launch(p, dg, 2, tile, mx, 0, g_tg, g_bm);
    hipError_t e2 = hipDeviceSynchronize();
    printf("launch %s sync %s tiles %d gm %d\n", hipGetErrorString(e), hipGetErrorString(e2), tile, gm);
    hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
    for (int q = 0, n0 = 0; q < 2; n0 += Ns[q], ++q)
        for (int n = n0; n < n0 + Ns[q]; ++n)
            for (int m = 0; m < M; ++m) {
                double s = 0.0;
                const double* b1 = B.data() + colB[n] / 8;
                for (int k = 0; k < Kc; ++k) s += A[(size_t)m * Kc + k] * b1[k];
                if (g[q].nk == 2 * nk1)
                    for (int k = 0; k < Kc; ++k) s += A[(size_t)M * Kc + (size_t)m * Kc + k] * b1[(size_t)ncol * Kc + k];
                R[(size_t)(offCm[m] + offCn[n])] = s;
            }
    double mxe = 0.0; long bad = 0, untouched = 0;
    for (size_t i = 0; i < C.size(); ++i) { double d = fabs(C[i] - R[i]); if (d > mxe) mxe = d; if (d > 1e-10) ++bad; if (C[i] == -7.0) ++untouched; }
    printf("max err %.3e, bad %ld of %zu, untouched %ld\n", mxe, bad, C.size(), untouched);
    if (bad) {
        printf("16x16 sub-block map (rows m/16, cols n/16): '.' ok, 'x' wrong, 'u' untouched\n");
        for (int mb = 0; mb < (M + 15) / 16; ++mb) {
            for (int nb = 0; nb < (ncol + 15) / 16; ++nb) {
                int w = 0, u = 0;
                for (int m = mb * 16; m < std::min(M, mb * 16 + 16); ++m)
                    for (int n = nb * 16; n < std::min(ncol, nb * 16 + 16); ++n) {
                        size_t i = (size_t)(offCm[m] + offCn[n]);
                        if (C[i] == -7.0) ++u; else if (fabs(C[i] - R[i]) > 1e-10) ++w;
                    }
                putchar(u ? 'u' : w ? 'x' : '.');
            }
            putchar('\n');
        }
    }
    return bad ? 1 : 0;
}
