#!/usr/bin/env python3
"""Stage times of post-SCF calculations through the C-ABI on synthetic AO integrals (default o=20, v=200): context, AO->MO + MP2,
CCSD initialisation, iterations, (T) -- first calls included, as a user of the library sees them.  The calculation is repeated
`rounds` times in ONE context (argv[4], default 1): from the second round on every device block comes out of the context's arena
(csrc/contract.hip) and no stage should show the multi-second hipMalloc outliers of DESIGN.md section 4.4.
usage: pipeline_time.py [o v [iterations [rounds]]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine


def main():
    o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20, 200)
    nit = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    n = o + v
    q, _ = np.linalg.qr(np.random.default_rng(7).standard_normal((n, n)))
    e = np.concatenate([-2.0 + np.arange(o) / max(o - 1, 1), 1.0 + 2.0 * np.arange(v) / max(v - 1, 1)])
    rng = np.random.default_rng(3); d = rng.standard_normal((n, n)); d = d + d.T
    t0 = time.perf_counter(); eng = Engine(0); print("%-60s %.1f ms" % ("context", (time.perf_counter() - t0) * 1e3))
    table = []
    for rd in range(rounds):
        T = {}
        calls = [eng.arena_stats()["driver_calls"]]
        def stage(name, t0):
            T[name] = time.perf_counter() - t0
            now = eng.arena_stats()["driver_calls"]
            if now != calls[0]: T[name + " [driver allocations]"] = (now - calls[0]) * 1e-3   # (printed as a count)
            calls[0] = now
        t0 = time.perf_counter(); eng.synthetic_ao(n, 0.0005, 777); stage("AO integrals generated on the device", t0)
        t0 = time.perf_counter(); eng.build_fock(n, d, d); stage("first Fock build (squares the integrals up)", t0)
        t0 = time.perf_counter(); eng.build_fock(n, d, d); stage("second Fock build", t0)
        t0 = time.perf_counter(); emp2, _ = eng.do_mp2_spatial(n, o, q, e, None, want_eri_mo=False); stage("AO->MO + MP2", t0)
        t0 = time.perf_counter(); eng.ccsd_init(o, v, e, None, 8); eng.ccsd_energy(); stage("ccsd_init + MP1 energy", t0)
        per = []
        for it in range(nit):
            t0 = time.perf_counter(); r = eng.ccsd_iterate(); eng.ccsd_diis(); per.append(time.perf_counter() - t0)
        T["CCSD iteration, first"] = per[0]
        T["CCSD iteration, second"] = per[1]
        T["CCSD iteration, median of the rest"] = float(np.median(per[2:]))
        t0 = time.perf_counter(); et = eng.do_ccsd_t_spatial_plain(); stage("(T), first call", t0)
        t0 = time.perf_counter(); et = eng.do_ccsd_t_spatial_plain(); stage("(T), second call", t0)
        table.append(T)
    eng_stats = eng.arena_stats()
    eng.close()
    keys = []
    for T in table:
        keys += [k for k in T if k not in keys]
    for k in keys:
        print("%-72s %s" % (k, " | ".join("%8.1f" % (T.get(k, 0.0) * 1e3) for T in table) + ("" if k.endswith("]") else "  ms")))
    print("arena:", eng_stats)
    if rounds > 2:
        worst = max((max(T[k] for T in table[1:]) - min(T[k] for T in table[1:]), k) for k in table[0] if not k.endswith("]"))
        print("largest spread of one stage over rounds 2..%d: %.1f ms (%s)" % (rounds, worst[0] * 1e3, worst[1]))
    print("E(MP2) %.10f  last CCSD energy %.10f  E[T] %.10f" % (emp2, r[0], et[0]))


if __name__ == "__main__":
    main()
