/* afesp_oracle_so.c -- TEST INFRASTRUCTURE ONLY (see afesp_oracle.c): CPU restatement of the reference's
 * spin-orbital CCSD / CCSD(T) path (Stanton, Gauss, Watts, Bartlett 1991 as coded in src/ccsd.f90).
 *
 *   do_ccsd_spinorb      ccsd.f90:71-277     antisymmetrised spin-orbital integrals, slices, driver
 *   init_cc (.not.restricted)  :437-448,:516 denominators in 2x2 spin blocks, t2 = <ij||ab>/D
 *   build_tau            ccsd.f90:678-714
 *   build_F              ccsd.f90:716-797    (as coded: the tau~ part of F_oo is accumulated transposed, see below)
 *   build_W              ccsd.f90:799-905    W_oooo stored (i,j,m,n), W_vvvv stored (e,f,a,b), W_ovvo (m,b,e,j)
 *   update_amplitudes    ccsd.f90:907-1038
 *   update_cc_energy     ccsd.f90:1783-1806  (unrestricted branch)
 *   do_ccsd_t_spinorb    ccsd.f90:1812-1922
 * DIIS is the same update_diis_cc (ccsd.f90:617-676) on the spin-orbital t1/t2.
 *
 * Everything is written as the plain loops the BLAS calls and reshapes of the reference amount to; index order and
 * layouts are the reference's (Fortran column-major, first index fastest, spin orbitals interleaved alpha,beta).
 * Pinning: sample_data/h2o-cc-pvdz/1.80_104.45/ref_out (a spin-orbital run shipped with the reference, Feb 2022):
 * 19 iteration energies and the final CCSD energy, reproduced to 5e-13 with foo_as_published = 1 (see so_F) --
 * tests/test_oracle_golden.py.  No shipped output exercises do_ccsd_t_spinorb with inputs that are also shipped; the
 * (T) restatement is checked against the spin-free (T) of the same molecule (both are the same quantity). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
i64 orc_eri_index(i64 i, i64 j, i64 k, i64 l);
int orc_linsolve(int n, double *A, double *b);

typedef struct {
    i64 o, v, n;   /* spin-orbital occupied / virtual counts (geometry.f90:44-45), spatial basis size */
    double *e;     /* spatial orbital energies, length n */
    double *oooo, *ooov, *ovoo, *oovo, *oovv, *ovvo, *ovvv, *vovv, *vvvv;
    double *D1, *D2, *t1, *t2, *t2_old, *r1, *r2;
    double *F_vv, *F_oo, *F_ov, *W_oooo, *W_vvvv, *W_ovvo, *tau, *tau_t;
    double energy, energy_old, rms;
    int nerr, nact, it;
    double *d_t1, *d_e1, *d_t2, *d_e2, *t1_s, *t2_s;
    int foo_as_published;   /* see so_F */
} orc_so;

#define O (s->o)
#define V (s->v)
#define IX4(a, b, c, d, n1, n2, n3) ((a) + (n1) * ((b) + (n2) * ((c) + (n3) * (d))))
#define T1(i, a) s->t1[(i) + O * (a)]
#define T2(i, j, a, b) s->t2[IX4(i, j, a, b, O, O, V)]
#define TAU(i, j, a, b) s->tau[IX4(i, j, a, b, O, O, V)]
#define TAUT(i, j, a, b) s->tau_t[IX4(i, j, a, b, O, O, V)]
#define OOOO(i, j, k, l) s->oooo[IX4(i, j, k, l, O, O, O)]
#define OOOV(i, j, k, a) s->ooov[IX4(i, j, k, a, O, O, O)]
#define OVOO(i, a, j, k) s->ovoo[IX4(i, a, j, k, O, V, O)]
#define OOVO(i, j, a, k) s->oovo[IX4(i, j, a, k, O, O, V)]
#define OOVV(i, j, a, b) s->oovv[IX4(i, j, a, b, O, O, V)]
#define OVVO(i, a, b, j) s->ovvo[IX4(i, a, b, j, O, V, V)]
#define OVVV(i, a, b, c) s->ovvv[IX4(i, a, b, c, O, V, V)]
#define VOVV(a, i, b, c) s->vovv[IX4(a, i, b, c, V, O, V)]
#define VVVV(a, b, c, d) s->vvvv[IX4(a, b, c, d, V, V, V)]
#define FVV(a, e) s->F_vv[(a) + V * (e)]
#define FOO(m, i) s->F_oo[(m) + O * (i)]
#define FOV(m, e) s->F_ov[(m) + O * (e)]
#define WOOOO(i, j, m, n) s->W_oooo[IX4(i, j, m, n, O, O, O)]
#define WVVVV(e, f, a, b) s->W_vvvv[IX4(e, f, a, b, V, V, V)]
#define WOVVO(m, b, e, j) s->W_ovvo[IX4(m, b, e, j, O, V, V)]

static double *dalloc(i64 n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* <pq||rs> over spin orbitals p = 2P + spin (ccsd.f90:108-143) */
static double asym_so(const double *eri, i64 p, i64 q, i64 r, i64 s_)
{
    const i64 P = p >> 1, Q = q >> 1, R = r >> 1, S = s_ >> 1;
    const int sp = (int)(p & 1), sq = (int)(q & 1), sr = (int)(r & 1), ss = (int)(s_ & 1);
    double x = 0.0;
    if (sp == sr && sq == ss) x += eri[orc_eri_index(P, R, Q, S)];
    if (sp == ss && sq == sr) x -= eri[orc_eri_index(P, S, Q, R)];
    return x;
}

orc_so *orc_so_create(i64 n, i64 nel, const double *eri_mo, const double *e, int diis_nerr)
{
    orc_so *s = (orc_so *)calloc(1, sizeof(orc_so));
    s->n = n; s->o = nel; s->v = 2 * n - nel;
    const i64 o = O, v = V;
    s->e = dalloc(n);
    memcpy(s->e, e, sizeof(double) * n);
    s->oooo = dalloc(o * o * o * o); s->ooov = dalloc(o * o * o * v); s->ovoo = dalloc(o * v * o * o);
    s->oovo = dalloc(o * o * v * o); s->oovv = dalloc(o * o * v * v); s->ovvo = dalloc(o * v * v * o);
    s->ovvv = dalloc(o * v * v * v); s->vovv = dalloc(v * o * v * v); s->vvvv = dalloc(v * v * v * v);
#define FILL(arr, n0, n1, n2, n3, off0, off1, off2, off3)                                                \
    _Pragma("omp parallel for") for (i64 d = 0; d < n3; ++d) for (i64 c = 0; c < n2; ++c)                   \
        for (i64 b = 0; b < n1; ++b) for (i64 a = 0; a < n0; ++a)                                           \
            s->arr[IX4(a, b, c, d, n0, n1, n2)] = asym_so(eri_mo, a + off0, b + off1, c + off2, d + off3);
    FILL(oooo, o, o, o, o, 0, 0, 0, 0)
    FILL(ooov, o, o, o, v, 0, 0, 0, o)
    FILL(ovoo, o, v, o, o, 0, o, 0, 0)
    FILL(oovo, o, o, v, o, 0, 0, o, 0)
    FILL(oovv, o, o, v, v, 0, 0, o, o)
    FILL(ovvo, o, v, v, o, 0, o, o, 0)
    FILL(ovvv, o, v, v, v, 0, o, o, o)
    FILL(vovv, v, o, v, v, o, 0, o, o)
    FILL(vvvv, v, v, v, v, o, o, o, o)
#undef FILL
    s->D1 = dalloc(o * v); s->D2 = dalloc(o * o * v * v);
    s->t1 = dalloc(o * v); s->t2 = dalloc(o * o * v * v); s->t2_old = dalloc(o * o * v * v);
    s->r1 = dalloc(o * v); s->r2 = dalloc(o * o * v * v);
    /* ccsd.f90:437-448: e(i) + e(j) - e(a + nocc/2) - e(b + nocc/2) on spatial labels, copied to the 2x2 spin blocks */
    const i64 os = o / 2;
    for (i64 a = 0; a < v; ++a) for (i64 i = 0; i < o; ++i) s->D1[i + o * a] = e[i / 2] - e[a / 2 + os];
    for (i64 b = 0; b < v; ++b) for (i64 a = 0; a < v; ++a) for (i64 j = 0; j < o; ++j) for (i64 i = 0; i < o; ++i)
        s->D2[IX4(i, j, a, b, o, o, v)] = e[i / 2] + e[j / 2] - e[a / 2 + os] - e[b / 2 + os];
    for (i64 x = 0; x < o * o * v * v; ++x) s->t2[x] = s->oovv[x] / s->D2[x];   /* :516 */
    s->F_vv = dalloc(v * v); s->F_oo = dalloc(o * o); s->F_ov = dalloc(o * v);
    s->W_oooo = dalloc(o * o * o * o); s->W_vvvv = dalloc(v * v * v * v); s->W_ovvo = dalloc(o * v * v * o);
    s->tau = dalloc(o * o * v * v); s->tau_t = dalloc(o * o * v * v);
    s->nerr = diis_nerr; s->nact = 0; s->it = 0;
    if (diis_nerr >= 2) {
        s->d_t1 = dalloc(o * v * diis_nerr); s->d_e1 = dalloc(o * v * diis_nerr);
        s->d_t2 = dalloc(o * o * v * v * diis_nerr); s->d_e2 = dalloc(o * o * v * v * diis_nerr);
        s->t1_s = dalloc(o * v); s->t2_s = dalloc(o * o * v * v);
    }
    return s;
}

void orc_so_destroy(orc_so *s)
{
    if (!s) return;
    double *all[] = {s->e, s->oooo, s->ooov, s->ovoo, s->oovo, s->oovv, s->ovvo, s->ovvv, s->vovv, s->vvvv, s->D1, s->D2, s->t1,
                     s->t2, s->t2_old, s->r1, s->r2, s->F_vv, s->F_oo, s->F_ov, s->W_oooo, s->W_vvvv, s->W_ovvo, s->tau, s->tau_t,
                     s->d_t1, s->d_e1, s->d_t2, s->d_e2, s->t1_s, s->t2_s};
    for (size_t i = 0; i < sizeof(all) / sizeof(all[0]); ++i) free(all[i]);
    free(s);
}

/* ccsd.f90:678-714 */
static void so_tau(orc_so *s)
{
#pragma omp parallel for collapse(2)
    for (i64 b = 0; b < V; ++b) for (i64 a = 0; a < V; ++a) for (i64 j = 0; j < O; ++j) for (i64 i = 0; i < O; ++i) {
        const double x = T1(i, a) * T1(j, b) - T1(i, b) * T1(j, a);
        TAUT(i, j, a, b) = T2(i, j, a, b) + 0.5 * x;
        TAU(i, j, a, b) = TAUT(i, j, a, b) + 0.5 * x;
    }
}

/* ccsd.f90:716-797.  NB the second F_oo term (:791-794): dgemm('N','N',nocc,nocc,...,tau_tilde,tmp_4_1,F_oo) produces
 * C(i,m) = 1/2 sum tau~(i,n,e,f) <mn||ef> and adds it to F_oo(i,m), while the loop above it (:757-767) filled F_oo(m,i)
 * and every consumer reads F_oo(m,i).  Default: restated as coded.  With foo_as_published the term lands in F_oo(m,i)
 * (Stanton Eq. 4); that is what the reference's shipped spin-orbital run ref_out (Feb 2022) was produced with -- all 19
 * iteration energies agree to 5e-13 -- and it is the only difference between that run and the current source found. */
static void so_F(orc_so *s)
{
    memset(s->F_vv, 0, sizeof(double) * V * V);
    memset(s->F_oo, 0, sizeof(double) * O * O);
    memset(s->F_ov, 0, sizeof(double) * O * V);
#pragma omp parallel for collapse(2)
    for (i64 a = 0; a < V; ++a) for (i64 e = 0; e < V; ++e) {
        double x = 0.0;
        for (i64 f = 0; f < V; ++f) for (i64 m = 0; m < O; ++m) x += T1(m, f) * OVVV(m, a, f, e);
        for (i64 f = 0; f < V; ++f) for (i64 n = 0; n < O; ++n) for (i64 m = 0; m < O; ++m)
            x += 0.5 * TAUT(m, n, a, f) * OOVV(m, n, f, e);
        FVV(a, e) = x;
    }
    for (i64 m = 0; m < O; ++m) for (i64 i = 0; i < O; ++i) {
        double x = 0.0;
        for (i64 e = 0; e < V; ++e) for (i64 n = 0; n < O; ++n) x -= T1(n, e) * OOOV(n, m, i, e);
        FOO(m, i) = x;
    }
    for (i64 i = 0; i < O; ++i) for (i64 m = 0; m < O; ++m) {
        double x = 0.0;
        for (i64 f = 0; f < V; ++f) for (i64 e = 0; e < V; ++e) for (i64 n = 0; n < O; ++n)
            x += TAUT(i, n, e, f) * OOVV(m, n, e, f);
        if (s->foo_as_published) FOO(m, i) += 0.5 * x;
        else FOO(i, m) += 0.5 * x;
    }
    for (i64 e = 0; e < V; ++e) for (i64 m = 0; m < O; ++m) {
        double x = 0.0;
        for (i64 f = 0; f < V; ++f) for (i64 n = 0; n < O; ++n) x += T1(n, f) * OOVV(m, n, e, f);
        FOV(m, e) = x;
    }
}

/* ccsd.f90:799-905 */
static void so_W(orc_so *s)
{
    const i64 o = O, v = V;
    /* Eq. 6, stored (i,j,m,n) */
    double *sc = dalloc(o * o * o * o);
#pragma omp parallel for collapse(2)
    for (i64 j = 0; j < o; ++j) for (i64 i = 0; i < o; ++i) for (i64 n = 0; n < o; ++n) for (i64 m = 0; m < o; ++m) {
        double x = 0.0;
        for (i64 e = 0; e < v; ++e) x += OOOV(m, n, i, e) * T1(j, e);
        sc[IX4(m, n, i, j, o, o, o)] = x;
    }
#pragma omp parallel for collapse(2)
    for (i64 j = 0; j < o; ++j) for (i64 i = 0; i < o; ++i) for (i64 n = 0; n < o; ++n) for (i64 m = 0; m < o; ++m) {
        double x = OOOO(m, n, i, j) + sc[IX4(m, n, i, j, o, o, o)] - sc[IX4(m, n, j, i, o, o, o)];
        for (i64 f = 0; f < v; ++f) for (i64 e = 0; e < v; ++e) x += 0.5 * OOVV(m, n, e, f) * TAU(i, j, e, f);
        WOOOO(i, j, m, n) = x;
    }
    free(sc);
    /* Eq. 7, stored (e,f,a,b) */
    sc = dalloc(v * v * v * v);
#pragma omp parallel for collapse(2)
    for (i64 f = 0; f < v; ++f) for (i64 e = 0; e < v; ++e) for (i64 a = 0; a < v; ++a) for (i64 b = 0; b < v; ++b) {
        double x = 0.0;
        for (i64 m = 0; m < o; ++m) x += T1(m, b) * OVVV(m, a, e, f);
        sc[IX4(b, a, e, f, v, v, v)] = x;
    }
#pragma omp parallel for collapse(2)
    for (i64 f = 0; f < v; ++f) for (i64 e = 0; e < v; ++e) for (i64 b = 0; b < v; ++b) for (i64 a = 0; a < v; ++a)
        WVVVV(e, f, a, b) = VVVV(a, b, e, f) + sc[IX4(b, a, e, f, v, v, v)] - sc[IX4(a, b, e, f, v, v, v)];
    free(sc);
    /* Eq. 8 */
#pragma omp parallel for collapse(2)
    for (i64 j = 0; j < o; ++j) for (i64 e = 0; e < v; ++e) for (i64 b = 0; b < v; ++b) for (i64 m = 0; m < o; ++m) {
        double x = OVVO(m, b, e, j);
        for (i64 f = 0; f < v; ++f) x += OVVV(m, b, e, f) * T1(j, f);
        for (i64 n = 0; n < o; ++n) x += T1(n, b) * OOVO(n, m, e, j);
        for (i64 f = 0; f < v; ++f) for (i64 n = 0; n < o; ++n)
            x -= OOVV(m, n, e, f) * (0.5 * T2(j, n, f, b) + T1(j, f) * T1(n, b));
        WOVVO(m, b, e, j) = x;
    }
}

/* ccsd.f90:907-1038 */
static void so_amplitudes(orc_so *s)
{
    const i64 o = O, v = V;
#define R1(i, a) s->r1[(i) + o * (a)]
#define R2(i, j, a, b) s->r2[IX4(i, j, a, b, o, o, v)]
#pragma omp parallel for collapse(2)
    for (i64 a = 0; a < v; ++a) for (i64 i = 0; i < o; ++i) {
        double x = 0.0;
        for (i64 e = 0; e < v; ++e) x += T1(i, e) * FVV(a, e);
        for (i64 m = 0; m < o; ++m) x -= FOO(m, i) * T1(m, a);
        for (i64 e = 0; e < v; ++e) for (i64 m = 0; m < o; ++m) {
            x += T1(m, e) * OVVO(m, a, e, i) + T2(m, i, e, a) * FOV(m, e);
            for (i64 f = 0; f < v; ++f) x += 0.5 * T2(m, i, f, e) * OVVV(m, a, f, e);
            for (i64 n = 0; n < o; ++n) x -= 0.5 * T2(m, n, e, a) * OOVO(m, n, e, i);
        }
        R1(i, a) = x / s->D1[i + o * a];
    }
    double *ts = dalloc(o * o * v * v), *X = dalloc(v * v), *Y = dalloc(o * o), *Z = dalloc(o * o * v * v);
#define TS(i, j, a, b) ts[IX4(i, j, a, b, o, o, v)]
#pragma omp parallel for collapse(2)
    for (i64 b = 0; b < v; ++b) for (i64 a = 0; a < v; ++a) for (i64 j = 0; j < o; ++j) for (i64 i = 0; i < o; ++i) {
        double x = 0.0;
        for (i64 e = 0; e < v; ++e) for (i64 m = 0; m < o; ++m)
            x += -T1(i, e) * T1(m, a) * OVVO(m, b, e, j) + T2(m, i, e, a) * WOVVO(m, b, e, j);
        TS(i, j, a, b) = x;
    }
    for (i64 b = 0; b < v; ++b) for (i64 e = 0; e < v; ++e) {   /* X(b,e) = sum_m t1(m,b) F_ov(m,e) */
        double x = 0.0;
        for (i64 m = 0; m < o; ++m) x += T1(m, b) * FOV(m, e);
        X[b + v * e] = x;
    }
    for (i64 j = 0; j < o; ++j) for (i64 m = 0; m < o; ++m) {   /* Y(j,m) = sum_e t1(j,e) F_ov(m,e) */
        double x = 0.0;
        for (i64 e = 0; e < v; ++e) x += T1(j, e) * FOV(m, e);
        Y[j + o * m] = x;
    }
#pragma omp parallel for collapse(2)
    for (i64 y = 0; y < v; ++y) for (i64 x_ = 0; x_ < v; ++x_) for (i64 j = 0; j < o; ++j) for (i64 i = 0; i < o; ++i) {
        double x = 0.0;
        for (i64 m = 0; m < o; ++m) x += OOVO(i, j, x_, m) * T1(m, y);
        Z[IX4(i, j, x_, y, o, o, v)] = x;
    }
#pragma omp parallel for collapse(2)
    for (i64 b = 0; b < v; ++b) for (i64 a = 0; a < v; ++a) for (i64 j = 0; j < o; ++j) for (i64 i = 0; i < o; ++i) {
        double x = OOVV(i, j, a, b) + TS(i, j, a, b) - TS(j, i, a, b) - TS(i, j, b, a) + TS(j, i, b, a);
        for (i64 e = 0; e < v; ++e) {
            x += T2(i, j, a, e) * FVV(b, e) - T2(i, j, b, e) * FVV(a, e);                       /* P_ab t_ijae F_be */
            x -= 0.5 * (T2(i, j, a, e) * X[b + v * e] - T2(i, j, b, e) * X[a + v * e]);         /* -1/2 P_ab t_ijae t_mb F_me */
            x += T1(i, e) * VOVV(e, j, a, b) - T1(j, e) * VOVV(e, i, a, b);                     /* P_ij t_ie <ej||ab> */
        }
        for (i64 m = 0; m < o; ++m) {
            x -= 0.5 * (Y[i + o * m] * T2(m, j, a, b) - Y[j + o * m] * T2(m, i, a, b));         /* -1/2 P_ij t_je F_me t_imab */
            x += -FOO(m, i) * T2(m, j, a, b) + FOO(m, j) * T2(m, i, a, b);                      /* -P_ij t_imab F_mj */
        }
        x += Z[IX4(i, j, b, a, o, o, v)] - Z[IX4(i, j, a, b, o, o, v)];                          /* -P_ab t_ma <mb||ij> */
        for (i64 n = 0; n < o; ++n) for (i64 m = 0; m < o; ++m) x += 0.5 * WOOOO(i, j, m, n) * TAU(m, n, a, b);
        for (i64 f = 0; f < v; ++f) for (i64 e = 0; e < v; ++e) x += 0.5 * TAU(i, j, e, f) * WVVVV(e, f, a, b);
        R2(i, j, a, b) = x / s->D2[IX4(i, j, a, b, o, o, v)];
    }
    free(ts); free(X); free(Y); free(Z);
    memcpy(s->t1, s->r1, sizeof(double) * o * v);
    memcpy(s->t2, s->r2, sizeof(double) * o * o * v * v);
#undef R1
#undef R2
#undef TS
}

/* ccsd.f90:1783-1806 */
int orc_so_energy(orc_so *s, double e_tol, double t_tol)
{
    double ecc = 0.0, rms = 0.0;
    s->energy_old = s->energy;
    for (i64 b = 0; b < V; ++b) for (i64 a = 0; a < V; ++a) for (i64 j = 0; j < O; ++j) for (i64 i = 0; i < O; ++i) {
        ecc += 0.25 * OOVV(i, j, a, b) * (T2(i, j, a, b) + 2.0 * T1(i, a) * T1(j, b));
        const double d = T2(i, j, a, b) - s->t2_old[IX4(i, j, a, b, O, O, V)];
        rms += d * d;
    }
    s->energy = ecc;
    memcpy(s->t2_old, s->t2, sizeof(double) * O * O * V * V);
    s->rms = rms;
    return (sqrt(rms) < t_tol && fabs(s->energy - s->energy_old) < e_tol) ? 1 : 0;
}

void orc_so_iterate(orc_so *s)
{
    so_tau(s);
    so_F(s);
    so_W(s);
    so_amplitudes(s);
}

static double ddot(i64 n, const double *x, const double *y)
{
    double r = 0.0;
    for (i64 i = 0; i < n; ++i) r += x[i] * y[i];
    return r;
}

void orc_so_diis_save(orc_so *s)
{
    if (s->nerr < 2) return;
    memcpy(s->t1_s, s->t1, sizeof(double) * O * V);
    memcpy(s->t2_s, s->t2, sizeof(double) * O * O * V * V);
}

/* ccsd.f90:617-676 */
int orc_so_diis_update(orc_so *s)
{
    if (s->nerr < 2) return 0;
    const i64 n1 = O * V, n2 = O * O * V * V;
    s->it += 1;
    if (s->it > s->nerr) s->it -= s->nerr;
    if (s->nact < s->nerr) s->nact += 1;
    const int slot = s->it - 1, n = s->nact, N = n + 1;
    memcpy(s->d_t1 + n1 * slot, s->t1, sizeof(double) * n1);
    memcpy(s->d_t2 + n2 * slot, s->t2, sizeof(double) * n2);
    for (i64 x = 0; x < n1; ++x) s->d_e1[n1 * slot + x] = s->t1[x] - s->t1_s[x];
    for (i64 x = 0; x < n2; ++x) s->d_e2[n2 * slot + x] = s->t2[x] - s->t2_s[x];
    double *B = dalloc(N * N), *c = dalloc(N);
    for (int j = 0; j < N; ++j) B[n + N * j] = -1.0;
    B[n + N * n] = 0.0;
    c[n] = -1.0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j)
        B[i + N * j] = ddot(n1, s->d_e1 + n1 * i, s->d_e1 + n1 * j) + ddot(n2, s->d_e2 + n2 * i, s->d_e2 + n2 * j);
    int ierr = orc_linsolve(N, B, c);
    if (!ierr) {
        memset(s->t1, 0, sizeof(double) * n1);
        memset(s->t2, 0, sizeof(double) * n2);
        for (int i = 0; i < n; ++i) {
            for (i64 x = 0; x < n1; ++x) s->t1[x] += c[i] * s->d_t1[n1 * i + x];
            for (i64 x = 0; x < n2; ++x) s->t2[x] += c[i] * s->d_t2[n2 * i + x];
        }
    }
    free(B); free(c);
    return ierr;
}

/* ccsd.f90:229-275 driver; entry 0 of the tables is the "MP1" line */
int orc_so_solve(orc_so *s, int maxiter, double e_tol, double t_tol, double *iter_energy, double *iter_rms)
{
    s->energy = 0.0; s->energy_old = 0.0;
    memset(s->t2_old, 0, sizeof(double) * O * O * V * V);
    orc_so_energy(s, e_tol, t_tol);
    if (iter_energy) iter_energy[0] = s->energy;
    if (iter_rms) iter_rms[0] = s->rms;
    for (int it = 1; it <= maxiter; ++it) {
        orc_so_diis_save(s);
        orc_so_iterate(s);
        int conv = orc_so_energy(s, e_tol, t_tol);
        if (iter_energy) iter_energy[it] = s->energy;
        if (iter_rms) iter_rms[it] = s->rms;
        if (conv) return it;
        if (orc_so_diis_update(s)) return -2;
    }
    return -1;
}

void orc_so_set_foo_as_published(orc_so *s, int on) { s->foo_as_published = on; }
double orc_so_get_energy(const orc_so *s) { return s->energy; }
double orc_so_get_rms(const orc_so *s) { return s->rms; }
i64 orc_so_nocc(const orc_so *s) { return s->o; }
i64 orc_so_nvirt(const orc_so *s) { return s->v; }
double *orc_so_t1(orc_so *s) { return s->t1; }
double *orc_so_t2(orc_so *s) { return s->t2; }
/* 0 F_vv, 1 F_oo, 2 F_ov, 3 W_oooo, 4 W_vvvv, 5 W_ovvo, 6 tau, 7 tau_tilde, 8 oovv, 9 vvvv */
double *orc_so_field(orc_so *s, int which)
{
    double *f[] = {s->F_vv, s->F_oo, s->F_ov, s->W_oooo, s->W_vvvv, s->W_ovvo, s->tau, s->tau_t, s->oovv, s->vvvv};
    return (which >= 0 && which < 10) ? f[which] : NULL;
}

/* ccsd.f90:1812-1922: E_T = sum_{ijk} sum_{abc} t3c (t3c/D + t3d) / 36, all (i,j,k), P(i/jk) explicit, P(a/bc) by the
 * two transposed copies (reshape orders (2,1,3) and (3,2,1): tmp(b,a,c) and tmp(c,b,a)). */
double orc_so_triples(orc_so *s)
{
    const i64 o = O, v = V, os = o / 2, v3 = v * v * v;
    double e_t = 0.0;
#define ESO(p) s->e[(p) / 2]
#pragma omp parallel reduction(+ : e_t)
    {
        double *d = dalloc(v3), *c = dalloc(v3), *cd = dalloc(v3);
#pragma omp for collapse(3) schedule(dynamic)
        for (i64 i = 0; i < o; ++i) for (i64 j = 0; j < o; ++j) for (i64 k = 0; k < o; ++k) {
            for (i64 a = 0; a < v; ++a) for (i64 cc = 0; cc < v; ++cc) for (i64 b = 0; b < v; ++b) {
                const double D = ESO(i) + ESO(j) + ESO(k) - s->e[a / 2 + os] - s->e[b / 2 + os] - s->e[cc / 2 + os];
                /* vvoo(b,c,j,k) = oovv(j,k,b,c) (:263) */
                d[a + v * (b + v * cc)] = (T1(i, a) * OOVV(j, k, b, cc) - T1(j, a) * OOVV(i, k, b, cc) - T1(k, a) * OOVV(j, i, b, cc)) / D;
                double x = 0.0;
                /* t2_reshape(f,a,k,j) = t2(j,k,a,f) (:1841) */
                for (i64 f = 0; f < v; ++f)
                    x += VOVV(f, i, b, cc) * T2(j, k, a, f) - VOVV(f, j, b, cc) * T2(i, k, a, f) - VOVV(f, k, b, cc) * T2(j, i, a, f);
                for (i64 m = 0; m < o; ++m)
                    x += -T2(m, i, cc, b) * OVOO(m, a, j, k) + T2(m, j, cc, b) * OVOO(m, a, i, k) + T2(m, k, cc, b) * OVOO(m, a, j, i);
                c[a + v * (b + v * cc)] = x;
                cd[a + v * (b + v * cc)] = x / D;
            }
            double sum = 0.0;
            for (i64 cc = 0; cc < v; ++cc) for (i64 b = 0; b < v; ++b) for (i64 a = 0; a < v; ++a) {
#define P3(arr) (arr[a + v * (b + v * cc)] - arr[b + v * (a + v * cc)] - arr[cc + v * (b + v * a)])
                sum += P3(c) * (P3(cd) + P3(d));
#undef P3
            }
            e_t += sum / 36.0;
        }
        free(d); free(c); free(cd);
    }
#undef ESO
    return e_t;
}
