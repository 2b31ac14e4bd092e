// ccsd_so.h -- device-resident state of the spin-orbital CCSD solver and its (T) correction
// (reference: do_ccsd_spinorb ccsd.f90:71-277, build_tau/F/W :678-905, update_amplitudes :907-1038,
//  do_ccsd_t_spinorb :1812-1922).  Spin orbitals are interleaved alpha,beta as in ccsd.f90:108-143; o and v are
// spin-orbital counts (geometry.f90:44-45: nocc = nel, nvirt = 2 nbasis - nel).
#pragma once
#include "ccsd.h"

namespace afesp {

struct SOState : DiisRing {
    int o = 0, v = 0, n = 0;
    bool ready = false;
    bool foo_as_published = false;   // see so_build_F
    double* e = nullptr;             // spatial orbital energies on device, length n
    // antisymmetrised slices <pq||rs> (ccsd.f90:193-207)
    Tensor oooo, ooov, ovoo, oovo, oovv, ovvo, ovvv, vovv, vvvv;
    Tensor D1, D2, t1, t2, t2_old, r1, r2;
    Tensor F_vv, F_oo, F_ov, W_oooo, W_vvvv, W_ovvo, tau, tau_t;
    Tensor t1_w;                     // t1 as the last so_intermediates saw it
    double energy = 0.0, energy_old = 0.0, rms = 0.0;
    void* tplan = nullptr;           // cached (T) launch plan (triples_so.hip)
    int64_t amp_epoch = 0;           // bumped by every entry point that may change t1 / t2: the (T) operand copies are rebuilt only then
    // 1/2 tau_ijef W_abef (ccsd.f90:1021-1024) without W_abef (so_ladder): the bare part over antisymmetric pairs -- va(ef, ab) =
    // <ab||ef> for e < f, a < b built once, ta(ij, ef) = tau for i < j, e < f and the product pa(ij, ab) per iteration
    double *va = nullptr, *ta = nullptr, *pa = nullptr;
    int64_t* lad_tab = nullptr;
    int64_t lad_ka = 0, lad_na = 0;  // even leading dimensions: v(v-1)/2 and o(o-1)/2 rounded up
};

// eri_mo_dev: packed chemist MO integrals on the device (length neri(nbasis)); e_host: spatial orbital energies (host)
void so_init(Context& cx, SOState& s, int nbasis, int nel, const double* eri_mo_dev, const double* e_host, int diis_nerr,
             bool foo_as_published);
void so_free(Context& cx, SOState& s);
void so_intermediates(Context& cx, SOState& s);   // build_tau, build_F, build_W
void so_amplitudes(Context& cx, SOState& s);      // update_amplitudes
void so_build_W_vvvv(Context& cx, SOState& s);    // W_abef itself (ccsd.f90:852-861), on request: the iteration never forms it
int so_energy(Context& cx, SOState& s, double e_tol, double t_tol);
// (T): contribution of the triples i<j<k with flat index in [t_begin, t_end) to E_T (ccsd.f90:1910)
int64_t so_triples_count(int o);
double so_triples(Context& cx, SOState& s, int64_t t_begin, int64_t t_end);
void so_triples_plan_free(SOState& s);

}  // namespace afesp
