"""Readers for the reference's on-disk inputs and stdout goldens.

File formats follow the reference's readers (citations into /root/reference):
  els.in namelist ............ src/system.f90:81-167 (defaults = system_t, :43-67)
  s.dat / t.dat / v.dat ...... src/integrals.f90:94-137 (1-based "i j value", lower triangle)
  eri.dat .................... src/integrals.f90:146-161 ("i j k l value", 8-fold unique entries)
  geom.dat ................... src/geometry.f90:23-46, e_nuc :74-95
  guess_in.dat ............... src/hf.f90:153-170
  stdout energy table ........ src/main.F90:123-175 (what utils/els_wrapper.py:104-127 greps)
"""
from __future__ import annotations

import dataclasses
import os
import re

import numpy as np


@dataclasses.dataclass
class SystemIn:
    """Mirror of the user-facing part of system_t (src/system.f90:10-69)."""
    calc_type: str = "CCSD(T)_spatial"
    scf_e_tol: float = 1e-6
    scf_d_tol: float = 1e-6
    scf_diis_n_errmat: int = 6
    ccsd_e_tol: float = 1e-6
    ccsd_t_tol: float = 1e-6
    ccsd_diis_n_errmat: int = 8
    scf_maxiter: int = 50
    ccsd_maxiter: int = 50
    write_fcidump: bool = False
    scf_read_guess: bool = False
    scf_write_guess: bool = False
    # derived by the calc_type switch (src/system.f90:116-165)
    level: str = "CCSD(T)"        # one of RHF, MP2, CCSD, CCSD(T)
    restricted: bool = True
    ccsd_t_paren: bool = False
    ccsd_t_renorm: bool = False
    ccsd_t_comp_renorm: bool = False


_CALC_TYPES = {
    # name: (level, restricted, paren, renorm, comp_renorm)   src/system.f90:116-165
    "RHF": ("RHF", True, False, False, False),
    "UHF": ("RHF", False, False, False, False),
    "MP2_spinorb": ("MP2", False, False, False, False),
    "MP2_spatial": ("MP2", True, False, False, False),
    "CCSD_spinorb": ("CCSD", False, False, False, False),
    "CCSD_spatial": ("CCSD", True, False, False, False),
    "CCSD(T)_spinorb": ("CCSD(T)", False, False, False, False),
    "CCSD(T)_spatial": ("CCSD(T)", True, True, False, False),
    "CCSD[T]_spatial": ("CCSD(T)", True, False, False, False),
    "RCCSD(T)_spatial": ("CCSD(T)", True, True, True, False),
    "RCCSD[T]_spatial": ("CCSD(T)", True, False, True, False),
    "CRCCSD(T)_spatial": ("CCSD(T)", True, True, False, True),
    "CRCCSD[T]_spatial": ("CCSD(T)", True, False, False, True),
}


def _parse_value(text: str):
    t = text.strip().rstrip(",").strip()
    if t.lower() in (".true.", "t", "true"):
        return True
    if t.lower() in (".false.", "f", "false"):
        return False
    if t and t[0] in "\"'":
        return t.strip("\"'")
    try:
        return int(t)
    except ValueError:
        return float(t.lower().replace("d", "e"))


def read_els_in(path: str) -> SystemIn:
    """Parse the &elsinput namelist.  Keys that are absent keep the system_t defaults
    (the reference leaves them uninitialised -- SURVEY.md section 5 hazard)."""
    sysin = SystemIn()
    body = open(path).read()
    m = re.search(r"&elsinput(.*?)^\s*/", body, re.S | re.M | re.I)
    if not m:
        raise ValueError("invalid input file format!")   # system.f90:111
    for key, val in re.findall(r"(\w+)\s*=\s*(\"[^\"]*\"|'[^']*'|[^,\n]+)", m.group(1)):
        key = key.lower()
        if not hasattr(sysin, key):
            raise ValueError("invalid input file format!")
        setattr(sysin, key, _parse_value(val))
    if sysin.calc_type not in _CALC_TYPES:
        raise ValueError("Unrecognised calculation type!")   # system.f90:163
    (sysin.level, sysin.restricted, sysin.ccsd_t_paren, sysin.ccsd_t_renorm,
     sysin.ccsd_t_comp_renorm) = _CALC_TYPES[sysin.calc_type]
    return sysin


def eri_index(i, j, k, l):
    """0-based packed index of (ij|kl): src/integrals.f90:196-210 composed twice."""
    def tri(a, b):
        a, b = np.maximum(a, b), np.minimum(a, b)
        return a * (a + 1) // 2 + b
    return tri(tri(i, j), tri(k, l))


def npair(n: int) -> int:
    return n * (n + 1) // 2


def neri(n: int) -> int:
    npr = npair(n)
    return npr * (npr + 1) // 2


@dataclasses.dataclass
class Integrals:
    nbasis: int
    ovlp: np.ndarray
    ke: np.ndarray
    ele_nuc: np.ndarray
    core_hamil: np.ndarray
    eri: np.ndarray          # packed, length neri(nbasis)
    e_nuc: float = 0.0
    nel: int = 0
    natoms: int = 0


def _read_two_index(path: str, n: int | None = None):
    dat = np.loadtxt(path, ndmin=2)
    i = dat[:, 0].astype(np.int64) - 1
    j = dat[:, 1].astype(np.int64) - 1
    if n is None:
        n = int(max(i.max(), j.max())) + 1      # integrals.f90:82-91
    mat = np.zeros((n, n))
    mat[i, j] = dat[:, 2]
    mat[j, i] = dat[:, 2]
    return mat, n


def read_integrals(directory: str) -> Integrals:
    """s.dat, t.dat, v.dat, eri.dat, geom.dat from `directory` (hard-coded names, integrals.f90:69-73)."""
    ovlp, n = _read_two_index(os.path.join(directory, "s.dat"))
    ke, _ = _read_two_index(os.path.join(directory, "t.dat"), n)
    en, _ = _read_two_index(os.path.join(directory, "v.dat"), n)
    dat = np.loadtxt(os.path.join(directory, "eri.dat"), ndmin=2)
    idx = dat[:, :4].astype(np.int64) - 1
    eri = np.zeros(neri(n))
    eri[eri_index(idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3])] = dat[:, 4]
    ints = Integrals(n, ovlp, ke, en, ke + en, eri)
    # geometry.f90:23-46, :74-95
    with open(os.path.join(directory, "geom.dat")) as fh:
        natoms = int(fh.readline().split()[0])
        charges, coords = [], []
        for _ in range(natoms):
            w = fh.readline().split()
            charges.append(int(float(w[0])))
            coords.append([float(x) for x in w[1:4]])
    coords = np.array(coords)
    e_nuc = 0.0
    for b in range(1, natoms):
        for a in range(b):
            e_nuc += charges[a] * charges[b] / np.linalg.norm(coords[a] - coords[b])
    ints.e_nuc, ints.nel, ints.natoms = e_nuc, int(sum(charges)), natoms
    return ints


def read_scf_guess(path: str, n: int) -> np.ndarray:
    dat = np.loadtxt(path, ndmin=2)
    g = np.zeros((n, n))
    g[dat[:, 0].astype(int) - 1, dat[:, 1].astype(int) - 1] = dat[:, 2]
    return g


_ENERGY_LINES = {
    "RHF energy": "rhf_total",
    "MP2 correlation energy": "mp2_corr",
    "CCSD correlation energy": "ccsd_corr",
    "CCSD[T] correlation energy": "ccsd_bt_corr",
    "CCSD(T) correlation energy": "ccsd_pt_corr",
    "R-CCSD[T] correlation energy": "r_ccsd_bt_corr",
    "R-CCSD(T) correlation energy": "r_ccsd_pt_corr",
    "CR-CCSD[T] correlation energy": "cr_ccsd_bt_corr",
    "CR-CCSD(T) correlation energy": "cr_ccsd_pt_corr",
    "T1 diagnostic": "t1_diag",
    "D[T]": "d_bt",
    "D(T)": "d_pt",
    "Nuclear repulsion": "e_nuc",
    "Total energy": "total",
}


def parse_els_out(path: str) -> dict:
    """Pull the machine-readable numbers out of a reference stdout capture: the final energy table
    (main.F90:123-175), the SCF and CCSD iteration tables (hf.f90:110-113, ccsd.f90:326-331,362-363)
    and the orbital energies (hf.f90:119-122)."""
    out: dict = {"scf_iters": [], "cc_iters": [], "orbital_energies": {}}
    section = None
    final = False
    for line in open(path):
        s = line.strip()
        if s.startswith("Restricted Hartree-Fock"):
            section = "scf"
        elif s == "CCSD":
            section = "cc"
        elif s.startswith("Final energy breakdown"):
            final = True
            section = None
        if final:
            m = re.match(r"(.+?):\s+(-?\d+\.\d+)\s*$", s)
            if m and m.group(1).strip() in _ENERGY_LINES:
                out[_ENERGY_LINES[m.group(1).strip()]] = float(m.group(2))
            continue
        if section == "scf":
            w = s.split()
            if len(w) == 5 and w[0].isdigit():
                out["scf_iters"].append((int(w[0]), float(w[1]), float(w[2]), float(w[3])))
            elif len(w) == 2 and w[0].isdigit() and re.match(r"-?\d+\.\d+$", w[1]):
                out["orbital_energies"][int(w[0])] = float(w[1])
        elif section == "cc":
            w = s.split()
            if len(w) == 4 and w[0] == "MP1":
                out["cc_iters"].append((0, float(w[1]), float(w[2]), float(w[3])))
            elif len(w) == 5 and w[0].isdigit():
                out["cc_iters"].append((int(w[0]), float(w[1]), float(w[2]), float(w[3])))
            m = re.match(r"Final CCSD Energy \(Hartree\):\s+(-?\d+\.\d+)", s)
            if m:
                out["final_ccsd"] = float(m.group(1))
        m = re.match(r"MP2 correlation energy \(Hartree\):\s+(-?\d+\.\d+)", s)
        if m:
            out["mp2_line"] = float(m.group(1))
    return out
