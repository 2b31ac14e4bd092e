/* afesp_oracle_blas.c -- TEST / BENCH INFRASTRUCTURE ONLY (never linked into or called by the product).
 *
 * The (T) loop and the pp-ladder of the reference in the reference's own SHAPE: one dgemm per permuted term and OpenMP over
 * the (i,j,k) blocks with a scalar reduction (src/ccsd.f90:2056-2066 operands, :2091 reduction, :2152-2237 loop body,
 * :1669 ladder dgemm), so that the CPU baseline of bench.py is BLAS-backed like the reference's CPU path is.  The BLAS is
 * the ILP64 OpenBLAS that numpy bundles (symbols scipy_cblas_dgemm64_, scipy_openblas_set_num_threads64_); it is opened at
 * run time from the path the caller hands in (tests/orc.py finds it next to numpy) -- nothing is linked at build time.
 * The elementwise part is the loop of afesp_oracle.c (orc_ccsd_t_impl), which is pinned against the reference's bundled
 * outputs; tests/test_oracle_golden.py checks this file against it. */
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;
typedef void (*dgemm_fn)(int order, int ta, int tb, i64 m, i64 n, i64 k, double alpha, const double *A, i64 lda, const double *B,
                         i64 ldb, double beta, double *C, i64 ldc);
typedef void (*setthr_fn)(int);
static dgemm_fn g_dgemm;
static setthr_fn g_setthr;
enum { COL = 102, NOT = 111, TRN = 112 };

/* 0 on success; 1: library not found; 2: symbols missing */
int orcb_load(const char *openblas_path)
{
    if (g_dgemm) return 0;
    void *h = dlopen(openblas_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return 1;
    g_dgemm = (dgemm_fn)dlsym(h, "scipy_cblas_dgemm64_");
    g_setthr = (setthr_fn)dlsym(h, "scipy_openblas_set_num_threads64_");
    if (!g_dgemm || !g_setthr) {
        g_dgemm = NULL;
        return 2;
    }
    return 0;
}

/* ccsd.f90:2152-2237: all ordered (i,j,k) with flat index in [t_begin, t_end); out[0..3] = E[T], E(T), D[T], D(T) as
 * orc_ccsd_t (the base term of the D sums is added by the caller holding t_begin == 0).  Arrays are the column-major ones
 * of afesp_oracle.c: t1(o,v), t2(o,o,v,v), v_vvov(v,v,o,v), v_oovo(o,o,v,o), v_oovv(o,o,v,v). */
int orcb_ccsd_t(i64 o, i64 v, const double *e, const double *t1, const double *t2, const double *v_vvov, const double *v_oovo,
                const double *v_oovv, i64 t_begin, i64 t_end, double *out)
{
    if (!g_dgemm) return 1;
    g_setthr(1);   /* the reference's layout: threads over (i,j,k), a serial dgemm inside each */
    const i64 v2 = v * v, v3 = v * v * v;
    double eT = 0.0, eTT = 0.0, dT = 0.0, dTT = 0.0;
    int failed = 0;
#define t1_(i, a) t1[(i) + o * (a)]
#define t2_(i, j, a, b) t2[(i) + o * ((j) + o * ((a) + v * (b)))]
#define oovo_(i, j, a, k) v_oovo[(i) + o * ((j) + o * ((a) + v * (k)))]
#define oovv_(i, j, a, b) v_oovv[(i) + o * ((j) + o * ((a) + v * (b)))]
#pragma omp parallel reduction(+ : eT, eTT, dT, dTT)
    {
        double *X = (double *)malloc(sizeof(double) * v3), *W = (double *)malloc(sizeof(double) * v3);
        double *T3 = (double *)malloc(sizeof(double) * v3), *Z = (double *)malloc(sizeof(double) * v3);
        double *Y = (double *)malloc(sizeof(double) * v3);
        double *TA = (double *)malloc(sizeof(double) * v2), *G = (double *)malloc(sizeof(double) * v * o);
        if (!X || !W || !T3 || !Z || !Y || !TA || !G) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp barrier
        if (!failed) {
#pragma omp for schedule(dynamic, 1)
            for (i64 ijk = t_begin; ijk < t_end; ++ijk) {
                const i64 i = ijk / (o * o), j = (ijk / o) % o, k = ijk % o;
                memset(W, 0, sizeof(double) * v3);
                /* particle terms (:2168-2173): X^{pqr}(x;y,z) = sum_d t2(p,q,x,d) <zy|rd>, as Xt[(z,y),x] = V_r . T_pq^T with
                 * V_r[(z,y),d] = v_vvov(z,y,r,d) used where it lies (leading dimension v^2 o) */
                const i64 P[6][3] = {{i, j, k}, {j, i, k}, {k, j, i}, {i, k, j}, {j, k, i}, {k, i, j}};
                for (int s = 0; s < 6; ++s) {
                    const i64 p = P[s][0], q = P[s][1], r = P[s][2];
                    for (i64 d = 0; d < v; ++d)
                        for (i64 x = 0; x < v; ++x) TA[x + v * d] = t2_(p, q, x, d);
                    g_dgemm(COL, NOT, TRN, v2, v, v, 1.0, v_vvov + v2 * r, v2 * o, TA, v, 0.0, X, v2);
                    /* X[z + v y + v^2 x] enters W(a,b,c) at (x;y,z) = (a;b,c) (b;a,c) (c;b,a) (a;c,b) (b;c,a) (c;a,b) */
                    for (i64 c = 0; c < v; ++c)
                        for (i64 b = 0; b < v; ++b)
                            for (i64 a = 0; a < v; ++a) {
                                const i64 x = s == 0 || s == 3 ? a : s == 1 || s == 4 ? b : c;
                                const i64 y = s == 0 ? b : s == 1 ? a : s == 2 ? b : s == 3 ? c : s == 4 ? c : a;
                                const i64 z = s == 0 ? c : s == 1 ? c : s == 2 ? a : s == 3 ? b : s == 4 ? a : b;
                                W[a + v * (b + v * c)] += X[z + v * y + v2 * x];
                            }
                }
                /* hole terms: H^{p;qr}(x,y;z) = sum_l t2(l,p,x,y) <qr|zl> = T_p^T . G^T, T_p[l,(x,y)] = t2(l,p,x,y) where it lies
                 * (leading dimension o^2), G[z,l] = v_oovo(q,r,z,l) copied */
                const i64 Hh[6][3] = {{i, k, j}, {j, k, i}, {k, i, j}, {i, j, k}, {j, i, k}, {k, j, i}};
                for (int s = 0; s < 6; ++s) {
                    const i64 p = Hh[s][0], q = Hh[s][1], r = Hh[s][2];
                    for (i64 l = 0; l < o; ++l)
                        for (i64 z = 0; z < v; ++z) G[z + v * l] = oovo_(q, r, z, l);
                    g_dgemm(COL, TRN, TRN, v2, v, o, 1.0, t2 + o * p, o * o, G, v, 0.0, X, v2);
                    /* H[x + v y + v^2 z] leaves W(a,b,c) at (x,y;z) = (b,a;c) (a,b;c) (b,c;a) (c,a;b) (c,b;a) (a,c;b) */
                    for (i64 c = 0; c < v; ++c)
                        for (i64 b = 0; b < v; ++b)
                            for (i64 a = 0; a < v; ++a) {
                                const i64 x = s == 0 ? b : s == 1 ? a : s == 2 ? b : s == 3 ? c : s == 4 ? c : a;
                                const i64 y = s == 0 ? a : s == 1 ? b : s == 2 ? c : s == 3 ? a : s == 4 ? b : c;
                                const i64 z = s == 0 ? c : s == 1 ? c : s == 2 ? a : s == 3 ? b : s == 4 ? a : b;
                                W[a + v * (b + v * c)] -= X[x + v * y + v2 * z];
                            }
                }
                const double eo = e[i] + e[j] + e[k];
                for (i64 c = 0; c < v; ++c)
                    for (i64 b = 0; b < v; ++b)
                        for (i64 a = 0; a < v; ++a) {
                            const i64 x = a + v * (b + v * c);
                            const double D = eo - e[a + o] - e[b + o] - e[c + o];
                            T3[x] = W[x] / D;                                                             /* :2175 */
                            Z[x] = (t1_(i, a) * oovv_(j, k, b, c) + t1_(j, b) * oovv_(i, k, a, c) + t1_(k, c) * oovv_(i, j, a, b)) / D;
                            Y[x] = t1_(i, a) * t1_(j, b) * t1_(k, c) + t1_(i, a) * t2_(j, k, b, c) + t1_(j, b) * t2_(i, k, a, c) +
                                   t1_(k, c) * t2_(i, j, a, b);                                            /* :2183-2184 */
                        }
                double s_tw = 0.0, s_zw = 0.0, s_ty = 0.0, s_zy = 0.0;
                for (i64 c = 0; c < v; ++c)
                    for (i64 b = 0; b < v; ++b)
                        for (i64 a = 0; a < v; ++a) {   /* x_bar = 4/3 x(abc) - 2 x(acb) + 2/3 x(cab), :2314-2318 */
                            const i64 x = a + v * (b + v * c), xacb = a + v * (c + v * b), xcab = c + v * (a + v * b);
                            const double tb = 4.0 * T3[x] / 3.0 - 2.0 * T3[xacb] + 2.0 * T3[xcab] / 3.0;
                            const double zb = 4.0 * Z[x] / 3.0 - 2.0 * Z[xacb] + 2.0 * Z[xcab] / 3.0;
                            s_tw += tb * W[x]; s_zw += zb * W[x]; s_ty += tb * Y[x]; s_zy += zb * Y[x];
                        }
                eT += s_tw; eTT += s_tw + s_zw; dT += s_ty; dTT += s_ty + s_zy;
            }
        }
        free(X); free(W); free(T3); free(Z); free(Y); free(TA); free(G);
    }
    if (failed) return 3;
    if (t_begin == 0) {   /* :2243 */
        double base = 1.0;
        for (i64 x = 0; x < o * v; ++x) base += 2.0 * t1[x] * t1[x];
        for (i64 b = 0; b < v; ++b) for (i64 a = 0; a < v; ++a) for (i64 j = 0; j < o; ++j) for (i64 i = 0; i < o; ++i)
            base += (2.0 * t2_(i, j, a, b) - t2_(j, i, a, b)) * (t2_(i, j, a, b) + t1_(i, a) * t1_(j, b));
        dT += base; dTT += base;
    }
    out[0] = eT; out[1] = eTT; out[2] = dT; out[3] = dTT;
    return 0;
}

/* C(m,n) = alpha A(m,k) B(k,n) + beta C, column-major, on `threads` OpenBLAS threads: the pp-ladder of ccsd.f90:1669 is one
 * such call (m = o^2, k = v^2, n = v^2 or a slab of its columns). */
int orcb_gemm(i64 m, i64 n, i64 k, double alpha, const double *A, const double *B, double beta, double *Cm, int threads)
{
    if (!g_dgemm) return 1;
    g_setthr(threads);
    g_dgemm(COL, NOT, NOT, m, n, k, alpha, A, m, B, k, beta, Cm, m);
    return 0;
}

/* ---- the loop sites of the reference's CCSD iteration (bench.py cpu_baseline: they are NOT dgemm calls in the reference, and an
 * iteration's CPU time at config 5 is theirs, not the ladder's).  Same loop nests, same OpenMP clauses, restricted to a slab of
 * the outermost index so that a bounded sample can be timed and scaled; arrays are the column-major ones of afesp_oracle.c. */

/* ccsd.f90:1170-1182: I_ovov(j,b,i,a) -= 0.5 v_oovv(m,i,b,e) c_oovv(m,j,a,e) for a in [a0, a1) */
int orcb_ring_I_ovov(i64 o, i64 v, const double *v_oovv, const double *c_oovv, double *I_ovov, i64 a0, i64 a1)
{
#define oovv4(X, p, q, r, s) X[(p) + o * ((q) + o * ((r) + v * (s)))]
#define ovov4(X, p, q, r, s) X[(p) + o * ((q) + v * ((r) + o * (s)))]
#pragma omp parallel for collapse(2) schedule(static, 10)
    for (i64 a = a0; a < a1; ++a)
        for (i64 i = 0; i < o; ++i)
            for (i64 b = 0; b < v; ++b)
                for (i64 j = 0; j < o; ++j)
                    for (i64 e = 0; e < v; ++e)
                        for (i64 m = 0; m < o; ++m)
                            ovov4(I_ovov, j, b, i, a) -= 0.5 * oovv4(v_oovv, m, i, b, e) * oovv4(c_oovv, m, j, a, e);
    return 0;
}

/* ccsd.f90:1680-1695 (Eq. 44, terms 6-8: "seems hopeless, use OMP"):
 * tmp_t2(i,j,a,b) += sum_{e,m} [ -t2(m,j,a,e) I_ovov(i,e,m,b) - I_ovov(i,e,m,a) t2(m,j,e,b) + asym_t2(m,i,e,a) I_voov(e,j,m,b) ]
 * for b in [b0, b1);  I_voov(e,j,m,b) is v x o x o x v */
int orcb_ring_t2(i64 o, i64 v, const double *t2, const double *asym_t2, const double *I_ovov, const double *I_voov, double *tmp_t2,
                 i64 b0, i64 b1)
{
#define voov4(X, p, q, r, s) X[(p) + v * ((q) + o * ((r) + o * (s)))]
#pragma omp parallel for collapse(3) schedule(static, 10)
    for (i64 b = b0; b < b1; ++b)
        for (i64 a = 0; a < v; ++a)
            for (i64 j = 0; j < o; ++j)
                for (i64 i = 0; i < o; ++i) {
                    double tmp = 0.0;
                    for (i64 e = 0; e < v; ++e)
                        for (i64 m = 0; m < o; ++m)
                            tmp = tmp - oovv4(t2, m, j, a, e) * ovov4(I_ovov, i, e, m, b) - ovov4(I_ovov, i, e, m, a) * oovv4(t2, m, j, e, b) +
                                  oovv4(asym_t2, m, i, e, a) * voov4(I_voov, e, j, m, b);
                    oovv4(tmp_t2, i, j, a, b) += tmp;
                }
    return 0;
}
