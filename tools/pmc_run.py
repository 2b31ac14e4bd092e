#!/usr/bin/env python3
"""Workload for the PMC passes: a calibration stream (known bytes) followed by pp-ladder launches at a given shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
o, v = int(sys.argv[1]), int(sys.argv[2])
eng = Engine(0)
n = 1 << 28                                   # 2 GiB per vector: axpby moves 24 n bytes = 6.44e9 B per launch
ms = eng.bench_stream(n, 3)
print(f"stream axpby n={n}: {ms:.3f} ms/launch  {24*n/ms/1e6:.1f} GB/s")
eng.synthetic_init(o, v, 0.005, 12345, 8)
eng.ccsd_energy()
ms = eng.time_pp_ladder(5)
print(f"pp-ladder o={o} v={v}: {ms:.4f} ms/launch  {2*o*o*v**4/ms/1e9:.2f} TFLOP/s  {8*(v**4+2*o*o*v*v)/ms/1e6:.1f} GB/s algorithmic")
eng.close()
