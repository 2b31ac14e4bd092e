"""(T)-shaped product (M = v^2, N = a few thousand columns) against K: how much of the short-K loss is per-tile overhead."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
eng = Engine(0)
M, N = 40000, 8192
for K in (112, 220, 224, 440, 880, 1760, 3520):
    ms = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=5)
    tiles = ((M + 255) // 256) * (N // 128)
    per_tile_us = ms * 1e3 / (tiles / 256.0)
    print(f"K={K:5d}: {ms:8.3f} ms {2.0*M*N*K/ms/1e9:6.2f} TF  per-tile-round {per_tile_us:8.2f} us", flush=True)
eng.close()
eng = Engine(0)
eng.set_tuning(1 << 17, 0, 0, 0)
for K in (224, 880):
    ms = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=5)
    tiles = ((M + 255) // 256) * (N // 128)
    print(f"no epilogue K={K:5d}: {ms:8.3f} ms per-tile-round {ms * 1e3 / (tiles / 256.0):8.2f} us", flush=True)
eng.set_tuning(0, 0, 0, 0)
eng.close()
