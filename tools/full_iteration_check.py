#!/usr/bin/env python3
"""One CCSD amplitude update at config 5 (o=20, v=200) on the device against the loop-form CPU restatement (oracle/afesp_oracle.c,
pinned to the reference's bundled outputs) from the SAME integrals and the SAME non-trivial amplitudes: every intermediate, both
residuals and the updated t1 / t2, element by element.  The restatement needs several minutes on 16 threads at this size, which is
why this is a tool (result kept under profiles/) and not a test.  usage: full_iteration_check.py [o v]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 16)))
import orc
from afesp_amd import inputs
from afesp_amd.capi import Engine


def hash_uniform(k, seed):   # numpy twin of the device generator (csrc/capi.hip, splitmix64)
    with np.errstate(over="ignore"):
        x = (k.astype(np.uint64) + np.uint64(seed)) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / 9007199254740992.0


def main():
    o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
    n, scale, seed = o + v, 0.005, 12345
    e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
    ne = inputs.neri(n)
    eri = np.empty(ne)
    for a in range(0, ne, 1 << 24):   # in slabs: the hash works on uint64 temporaries
        b = min(ne, a + (1 << 24))
        eri[a:b] = scale * (2.0 * hash_uniform(np.arange(a, b, dtype=np.uint64), seed) - 1.0)
    t0 = time.perf_counter()
    cc = orc.OracleCC(o, v, eri, e, 8)
    print("oracle state built in %.1f s" % (time.perf_counter() - t0), flush=True)
    with Engine(0) as eng:
        eng.synthetic_init(o, v, scale, seed, 8)
        assert np.array_equal(eng.tensor("v_oovv"), cc.field("v_oovv")), "the host twin of the integral generator disagrees with the device"
        eng.ccsd_energy(); eng.ccsd_iterate()           # t1 != 0 from here on
        t1, t2 = eng.amplitudes()
        cc.t1[...] = t1
        cc.t2[...] = t2
        t0 = time.perf_counter(); eng.update_intermediates()
        names = ("I_vo", "I_vv", "I_oo_p", "I_oo", "c_oovv", "asym_t2", "x_voov", "I_oooo", "I_ovov", "I_voov", "I_vovv_p", "I_ooov_p")
        dev = {name: eng.tensor(name) for name in names}   # (I_vovv_p is formed on request from the CURRENT t1: before the update)
        eng.update_amplitudes(); g1, g2 = eng.amplitudes()
        print("device: intermediates + amplitudes (+ downloads) %.2f s" % (time.perf_counter() - t0), flush=True)
        t0 = time.perf_counter(); cc.L.orc_cc_intermediates(cc.h); print("oracle intermediates %.1f s" % (time.perf_counter() - t0), flush=True)
        worst = 0.0
        for name in names:
            ref = cc.field(name)
            d = np.max(np.abs(dev[name] - ref)) / max(1.0, np.max(np.abs(ref)))
            worst = max(worst, d)
            print("  %-9s max rel diff %.2e   (max |ref| %.3e)" % (name, d, np.max(np.abs(ref))), flush=True)
        del dev
        t0 = time.perf_counter(); cc.L.orc_cc_amplitudes(cc.h); print("oracle amplitudes %.1f s" % (time.perf_counter() - t0), flush=True)
        for name, got in (("r1", eng.tensor("r1")), ("r2", eng.tensor("r2")), ("t1", g1), ("t2", g2)):
            ref = cc.field(name) if name in ("r1", "r2") else (cc.t1 if name == "t1" else cc.t2)
            if name == "r2":   # compared as it enters the amplitudes, under P(ia/jb): the engine holds some terms as their images (csrc/ccsd.hip, z_ooov)
                got, ref = got + got.transpose(1, 0, 3, 2), ref + ref.transpose(1, 0, 3, 2)
            d = np.max(np.abs(got - ref)) / max(1.0, np.max(np.abs(ref)))
            worst = max(worst, d)
            print("  %-9s max rel diff %.2e   (max |ref| %.3e)" % (name, d, np.max(np.abs(ref))), flush=True)
    print("worst %.2e  %s" % (worst, "ok" if worst < 1e-10 else "FAILED"))
    return 0 if worst < 1e-10 else 1


if __name__ == "__main__":
    sys.exit(main())
