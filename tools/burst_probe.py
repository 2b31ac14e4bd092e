#!/usr/bin/env python3
"""Is the tile epilogue bound by the device-wide write burst?  Same per-workgroup tile stream on all CUs and on a quarter."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine


def main():
    eng = Engine(0)
    M = 4096
    for K in (224, 896):
        for quarter in (0, 1):
            eng.set_tuning((4 << 17) if quarter else 0, 0, 0, 0)
            wgs = 64 if quarter else 256
            res = {}
            for r in (8, 16):
                N = 128 * 16 * r // (4 if quarter else 1)      # r tiles per workgroup either way
                res[r] = eng.bench_contract("km", (K, M), "kn", (K, N), "nm", (N, M), reps=7) * 1e3
            per = (res[16] - res[8]) / 8
            print(f"K={K} workgroups={wgs}: steady-state per tile {per:.1f} us ({per - (K // 16) * 4.0:.1f} us beyond {K // 16} x 4.0 us)")
    eng.set_tuning(0, 0, 0, 0)
    eng.close()


if __name__ == "__main__":
    main()
