"""Shared helpers: run the reference-compatible host pipeline up to the inputs of the hot path."""
from __future__ import annotations

import functools
import os

import numpy as np

from afesp_amd import inputs, rhf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Energies measured on the reference itself and recorded in SURVEY.md section 8(c) (H2O has no bundled
# spatial-path stdout; N2/F2 values are identical to their bundled els.out).
SURVEY_GOLD = {
    "h2o-cc-pvdz": dict(rhf_total=-75.5763632810, mp2_corr=-0.2885875159, ccsd_corr=-0.3116057309,
                        ccsd_bt_corr=-0.3302565754, ccsd_pt_corr=-0.3292222294, t1_diag=0.0301634237,
                        r_ccsd_pt_corr=-0.3249210039, d_bt=1.3218573700, d_pt=1.3230294579),
    "n2-cc-pvdz": dict(rhf_total=-108.3305827541, mp2_corr=-0.8459445164, ccsd_corr=-0.5813264819,
                       ccsd_bt_corr=-0.6993574848, ccsd_pt_corr=-0.6848274031, t1_diag=0.0323534020,
                       r_ccsd_pt_corr=-0.6352432081, d_bt=1.9149923969, d_pt=1.9196440233),
    "f2-cc-pvdz": dict(rhf_total=-198.6159545893, mp2_corr=-0.4373493658, ccsd_corr=-0.4503407126,
                       ccsd_bt_corr=-0.4709791761, ccsd_pt_corr=-0.4699908833, t1_diag=0.0137594955,
                       r_ccsd_pt_corr=-0.4666671896, d_bt=1.2033195175, d_pt=1.2035769057),
}


@functools.lru_cache(maxsize=None)
def load(name: str):
    """-> (SystemIn, Integrals, RHFResult, golden dict or {})"""
    d = os.path.join(GOLDEN, name)
    si = inputs.read_els_in(os.path.join(d, "els.in"))
    ints = inputs.read_integrals(d)
    guess = inputs.read_scf_guess(os.path.join(d, "guess_in.dat"), ints.nbasis) if si.scf_read_guess else None
    res = rhf.do_rhf(si, ints, guess)
    out = os.path.join(d, "els.out")
    gold = inputs.parse_els_out(out) if os.path.exists(out) else {}
    return si, ints, res, gold


def lcg_uniform(count: int, seed: int = 12345) -> np.ndarray:
    """SURVEY.md 8(d) synthetic recipe: x <- (6364136223846793005 x + 1442695040888963407) mod 2^63,
    u = (x >> 11) / 2^52."""
    out = np.empty(count)
    x = seed
    a, c, m = 6364136223846793005, 1442695040888963407, (1 << 63) - 1
    for k in range(count):
        x = (a * x + c) & m
        out[k] = (x >> 11) / float(1 << 52)
    return out


def synthetic_system(o: int, v: int, scale: float = 0.02, seed: int = 12345):
    """Identity C, ladder orbital energies, LCG-uniform packed ERIs (SURVEY.md 8(d) config-2 recipe)."""
    n = o + v
    e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
    eri = scale * (2.0 * lcg_uniform(inputs.neri(n), seed) - 1.0)
    return n, e, eri
