#!/usr/bin/env python3
"""Decomposes the short-K tile cost of the 256x128 persistent GEMM: time against rounds of 256 tiles and K steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
M = 4096
res = {}
for K in (224, 448, 896):
    for r in (1, 2, 4, 8, 16):
        N = 128 * 16 * r
        ms = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=7)
        res[(K, r)] = ms * 1e3
        print(f"K={K:4d} rounds={r:2d}: {ms*1e3:8.1f} us  ({2.0*M*N*K/ms/1e9:5.1f} TF)", flush=True)
for K in (224, 448, 896):
    per = (res[(K, 16)] - res[(K, 8)]) / 8
    print(f"K={K}: steady-state per tile {per:.1f} us = {per / (K // 16):.2f} us per K step; first round {res[(K, 1)]:.1f} us")
s = ((res[(896, 16)] - res[(896, 8)]) - (res[(224, 16)] - res[(224, 8)])) / 8 / (56 - 14)
print(f"per K step (from K=896 vs 224): {s:.3f} us; per-tile constant: {(res[(224, 16)] - res[(224, 8)]) / 8 - 14 * s:.2f} us")
eng.close()
