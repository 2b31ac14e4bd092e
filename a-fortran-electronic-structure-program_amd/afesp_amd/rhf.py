"""Restricted Hartree-Fock, host side (numpy).  Not on the accelerated path (O(n^4), <0.1 s at every
config) but it produces the two inputs of the path -- `canon_coeff` (MO x AO!) and
`canon_levels` -- so it follows the reference's iteration exactly: src/hf.f90:21-151 (driver),
:193-236 (DIIS), :320-338 (energy / convergence), :349-385 (Fock build).
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .inputs import Integrals, SystemIn, eri_index


@dataclasses.dataclass
class RHFResult:
    converged: bool
    e_hf: float                 # electronic energy (no nuclear repulsion), hf.f90:125
    canon_coeff: np.ndarray     # (MO, AO)  hf.f90:102,127
    canon_levels: np.ndarray
    fock_ao: np.ndarray
    iters: list


def unpack_eri(n: int, packed: np.ndarray) -> np.ndarray:
    idx = np.arange(n)
    i, j, k, l = np.meshgrid(idx, idx, idx, idx, indexing="ij")
    return packed[eri_index(i, j, k, l)]


def do_rhf(sysin: SystemIn, ints: Integrals, scf_guess: np.ndarray | None = None) -> RHFResult:
    n, nocc = ints.nbasis, ints.nel // 2
    S, H = ints.ovlp, ints.core_hamil
    V = unpack_eri(n, ints.eri)
    # X = S^-1/2 = U s^-1/2 U^T   (hf.f90:49-66)
    s, U = np.linalg.eigh(S)
    X = U @ np.diag(1.0 / np.sqrt(s)) @ U.T
    fock = (scf_guess if (sysin.scf_read_guess and scf_guess is not None) else H).copy()
    nerr = sysin.scf_diis_n_errmat
    use_diis = nerr >= 2
    dF = np.zeros((nerr, n, n)) if use_diis else None
    dE = np.zeros((nerr, n, n)) if use_diis else None
    d_iter = d_active = 0
    energy = energy_old = 0.0
    dens_old = np.zeros((n, n))
    iters = []
    for it in range(1, sysin.scf_maxiter + 1):
        w, A = np.linalg.eigh(X.T @ fock @ X)
        C = (X @ A).T                                  # rows = MOs (hf.f90:102)
        dens = C[:nocc].T @ C[:nocc]
        energy_old, energy = energy, float(np.sum(dens * (H + fock)))
        rms = float(np.sqrt(np.sum((dens - dens_old) ** 2)))
        dens_old = dens
        iters.append((it, energy, energy - energy_old, rms))
        if rms < sysin.scf_d_tol and abs(energy - energy_old) < sysin.scf_e_tol:
            return RHFResult(True, energy, C, w, fock, iters)
        # hf.f90:349-385
        fock = H + 2.0 * np.einsum("ijkl,kl->ij", V, dens) - np.einsum("ikjl,kl->ij", V, dens)
        if use_diis:                                   # hf.f90:193-236
            d_iter += 1
            if d_iter > nerr:
                d_iter -= nerr
            if d_active < nerr:
                d_active += 1
            dF[d_iter - 1] = fock
            dE[d_iter - 1] = fock @ dens @ S - S @ dens @ fock
            m = d_active
            if m > 1:
                B = np.zeros((m + 1, m + 1))
                B[:m, :m] = np.einsum("iab,jab->ij", dE[:m], dE[:m])
                B[m, :m] = B[:m, m] = -1.0
                rhs = np.zeros(m + 1)
                rhs[m] = -1.0
                c = np.linalg.solve(B, rhs)
                fock = np.einsum("i,iab->ab", c[:m], dF[:m])
    return RHFResult(False, energy, C, w, fock, iters)
