#!/usr/bin/env python3
"""Where a wave of the 8-wave GEMM kernel spends its cycles (diagnostic build: tools/build_variants.sh with variant bit 64).
usage: AFESP_LIBRARY=build/ab/libafesp_v64.so stamp_probe.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine


def main():
    eng = Engine(0)
    M, N = 40000, 8192
    for K in (448, 3520):
        ms = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=2)
        buf = (C.c_ulonglong * (256 * 8 * 4))()
        eng.L.afesp_debug_stamps(buf, 256 * 8 * 4)
        a = np.array(buf, dtype=np.float64).reshape(256, 8, 4)
        a = a[a[:, :, 3].min(axis=1) > 0]
        raw = a[..., 3].astype(np.uint64)
        a[..., 3] = (raw & np.uint64((1 << 20) - 1)).astype(np.float64)
        dw = (raw >> np.uint64(20)).astype(np.float64) * a[..., 3]   # variant bit 128: cycles waiting for the gathered data; ping-pong: C01 phase
        tot, bar, st, n = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
        print(f"K={K}: {2.0*M*N*K/ms/1e9:.1f} TF  workgroups {a.shape[0]}  steps/wave {n.mean():.0f}  cycles/step {np.mean(tot/n):.0f} "
              f"(MFMA floor 8192)  barrier {np.mean(bar/n):.0f}  stash {np.mean(st/n):.0f}")
        for w in range(8):
            print(f"   wave {w}: cycles/step {np.mean(tot[:, w]/n[:, w]):.0f} barrier {np.mean(bar[:, w]/n[:, w]):.0f} stash {np.mean(st[:, w]/n[:, w]):.0f} (data wait {np.mean(dw[:, w]/n[:, w]):.0f})")
    eng.close()


if __name__ == "__main__":
    main()
