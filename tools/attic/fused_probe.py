#!/usr/bin/env python3
"""Launch-fused small-system path (csrc/fused.hip) against the call-by-call path: energies of a short solve in both modes (two
processes: AFESP_FUSED is read once) and the iteration time.  usage: fused_probe.py [o v]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from afesp_amd.capi import Engine
    o, v = int(sys.argv[2]), int(sys.argv[3])
    with Engine(0) as eng:
        eng.synthetic_init(o, v, 0.02, 12345, 8)
        nit, en, rm = eng.do_ccsd_spatial(12, 1e-12, 1e-12)
        print("E", " ".join("%.14f" % x for x in en[:13]))
        eng.synthetic_init(o, v, 0.02, 12345, 8)
        eng.ccsd_energy()
        for _ in range(5): eng.ccsd_iterate(); eng.ccsd_diis()
        t0 = time.perf_counter()
        for _ in range(30): eng.ccsd_iterate(); eng.ccsd_diis()
        print("T unreplayed %.1f us" % ((time.perf_counter() - t0) / 30 * 1e6))
        for _ in range(20): eng.ccsd_iterate(); eng.ccsd_diis()
        t0 = time.perf_counter()
        for _ in range(50): eng.ccsd_iterate(); eng.ccsd_diis()
        print("T replayed %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
    sys.exit(0)
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(5, 53), (7, 21), (9, 19), (4, 10)]
for (o, v) in shapes:
    res = {}
    for mode in ("1", "0"):
        env = dict(os.environ, AFESP_FUSED=mode)
        out = subprocess.run([sys.executable, __file__, "child", str(o), str(v)], env=env, capture_output=True, text=True)
        if out.returncode: print(out.stdout, out.stderr); sys.exit(1)
        res[mode] = out.stdout.splitlines()
        if mode == "1" and os.environ.get("AFESP_FUSED_DEBUG"): print(out.stderr)
    e1 = [float(x) for x in res["1"][0].split()[1:]]; e0 = [float(x) for x in res["0"][0].split()[1:]]
    print("o=%d v=%d max |dE| fused vs call-by-call over 12 iterations: %.2e" % (o, v, max(abs(a - b) for a, b in zip(e1, e0))))
    print("   fused:        ", " | ".join(res["1"][1:]))
    print("   call-by-call: ", " | ".join(res["0"][1:]), flush=True)
