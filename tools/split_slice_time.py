#!/usr/bin/env python3
"""What one rank of a split CCSD iteration does, timed on one GPU without the exchange (AFESP_CC_TIME_SLICE="rank,world": the
library then evaluates that rank's share and skips the all-reduce -- the numbers that come out are not an iteration's).
usage: split_slice_time.py [o v]   -> one line per world in {1, 2, 4, 8}, slowest / fastest rank"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from afesp_amd.capi import Engine
    o, v = int(sys.argv[2]), int(sys.argv[3])
    with Engine(0) as eng:
        eng.synthetic_init(o, v, 0.0005, 12345, 8)
        eng.ccsd_energy()
        for _ in range(2): eng.ccsd_iterate()
        t0 = time.perf_counter()
        for _ in range(4): eng.ccsd_iterate()
        print("MS %.3f" % ((time.perf_counter() - t0) / 4 * 1e3))
    sys.exit(0)
o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
for world in (1, 2, 4, 8):
    times = []
    for rank in sorted({0, world // 2, world - 1}):
        env = dict(os.environ)
        if world > 1: env["AFESP_CC_TIME_SLICE"] = f"{rank},{world}"
        out = subprocess.run([sys.executable, __file__, "child", str(o), str(v)], env=env, capture_output=True, text=True)
        if out.returncode: print(out.stdout, out.stderr); sys.exit(1)
        times.append(float([l for l in out.stdout.splitlines() if l.startswith("MS")][0].split()[1]))
    print(f"o={o} v={v} world {world}: a rank's iteration without the exchange {min(times):.2f} - {max(times):.2f} ms (ranks 0, w/2, w-1)", flush=True)
