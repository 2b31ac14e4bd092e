#!/usr/bin/env python3
"""Past config 5 (default o=30, v=300: v^4 = 65 GB; also o=40, v=360) through the DEFAULT large-system path -- ring slab and grouped
ring launches, pair forms, K-stacked launches, stream-K ladder tiles, two-kernel tail, 96-row tiles -- one CCSD iteration and (T) checked
against restatements, not only against themselves:
  * every intermediate of the iteration (I_vo, I_vv, I_oo_p, I_oo, I_oooo, I_ovov, I_voov, x_voov, I_ooov_p) and r1 element by element
    against tests/np_cc.py (the numpy restatement that tests/test_oracle_golden.py pins to the loop form) on the tensors the device holds;
  * the T2 residual (symmetrised, as it enters the update) and the UPDATED t2 on sampled column pairs (a, b) -- pp-ladder included: <ef|ab>
    for those pairs is regenerated on the host from the hashed integrals;
  * t2(i,j,a,b) = t2(j,i,b,a) over the whole updated tensor;
  * (T): all sorted triples of the first occupied block against the dgemm-per-term restatement (oracle/afesp_oracle_blas.c), shards add up.
A guard against 32-bit overflows in plans, tables and row offsets (csrc/ring.hip: 8 Kc o v < 4 GiB).  usage: big_system_check.py o v [npairs]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from afesp_amd.capi import Engine
import np_cc
import orc


def hash_uniform(k, seed):
    with np.errstate(over="ignore"):
        x = (k.astype(np.uint64) + np.uint64(seed)) + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / 9007199254740992.0


def tri(i, j):
    hi, lo = np.maximum(i, j), np.minimum(i, j)
    return hi * (hi + 1) // 2 + lo


def main():
    o, v = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30, 300)
    npairs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    scale, seed = 0.002, 12345
    worst = {}

    def check(name, got, ref, tol=1e-10):
        err = float(np.max(np.abs(got - ref)) / float(np.max(np.abs(ref))))   # relative to the tensor's largest element
        worst[name] = err
        print(f"  {name:10s} max rel err {err:.2e}  (max |ref| {np.max(np.abs(ref)):.3e})", flush=True)
        assert err < tol, (name, err)

    eng = Engine(0)
    t0 = time.perf_counter(); eng.synthetic_init(o, v, scale, seed, 4); print("init %.1f s" % (time.perf_counter() - t0), flush=True)
    print("MP1", eng.ccsd_energy())
    t0 = time.perf_counter(); r = eng.ccsd_iterate(); eng.ccsd_diis(); print("iter 0", r, "%.3f s" % (time.perf_counter() - t0), flush=True)
    # the iteration under test: from amplitudes with t1 != 0 (the first one starts from t1 = 0)
    t1, t2 = eng.amplitudes()
    t0 = time.perf_counter(); r = eng.ccsd_iterate(); print("iter 1", r, "%.3f s" % (time.perf_counter() - t0), flush=True)
    counts = eng.launch_counts()
    print("launch counts [tall, gett, tgemm, tgemm mixed]:", counts, flush=True)
    oovv, ovov, vvov, oovo, oooo = (eng.tensor(k) for k in ("v_oovv", "v_ovov", "v_vvov", "v_oovo", "v_oooo"))
    t0 = time.perf_counter()
    I = np_cc.intermediates(t1, t2, oovv, ovov, vvov, oovo, oooo)
    print("numpy intermediates %.1f s" % (time.perf_counter() - t0), flush=True)
    for name in ("I_vo", "I_vv", "I_oo_p", "I_oo", "I_oooo", "I_ovov", "I_voov", "x_voov", "I_ooov_p"):
        check(name, eng.tensor(name), I[name])
    check("r1", eng.tensor("r1"), np_cc.r1(t1, I, oovv, ovov, vvov, oovo))
    # sampled column pairs: corners, a diagonal one, random ones
    rng = np.random.default_rng(7)
    pairs = [(0, 0), (v - 1, v - 1), (0, v - 1), (v // 2, v // 2 + 1)] + [tuple(int(x) for x in rng.integers(0, v, 2)) for _ in range(npairs)]
    r2d, D2, t1n_t2n = eng.tensor("r2"), eng.tensor("D2"), eng.amplitudes()
    t2n = t1n_t2n[1]
    ee, ff = np.meshgrid(np.arange(v), np.arange(v), indexing="ij")

    def vvvv(a, b):   # <ef|ab> = (ea|fb) of the hashed packed MO integrals, virtual offsets added (csrc/capi.hip, synth_packed_kernel)
        idx = tri(tri(ee + o, np.full_like(ee, a + o)), tri(ff + o, np.full_like(ff, b + o)))
        return scale * (2.0 * hash_uniform(idx, seed) - 1.0)

    e_r2 = e_t2 = 0.0
    for (a, b) in pairs:
        rab = np_cc.r2_cols(t1, t2, I, oovv, ovov, vvov, vvvv(a, b), a, b)
        rba = np_cc.r2_cols(t1, t2, I, oovv, ovov, vvov, vvvv(b, a), b, a)
        sym_ref = rab + rba.T
        sym_dev = r2d[:, :, a, b] + r2d[:, :, b, a].T   # (a term may sit in its image under (i<->j, a<->b): only the sum is defined)
        e_r2 = max(e_r2, float(np.max(np.abs(sym_dev - sym_ref)) / float(np.max(np.abs(sym_ref)))))
        tn = np_cc.new_t2_cols(rab, rba, oovv, D2, a, b)
        e_t2 = max(e_t2, float(np.max(np.abs(t2n[:, :, a, b] - tn)) / float(np.max(np.abs(tn)))))
    print(f"  r2 + image on {len(pairs)} column pairs: max rel err {e_r2:.2e};  updated t2 there: {e_t2:.2e}", flush=True)
    assert e_r2 < 1e-10 and e_t2 < 1e-10
    sym = float(np.max(np.abs(t2n - t2n.transpose(1, 0, 3, 2))))
    print(f"  t2(ijab) - t2(jiba) over the whole updated tensor: {sym:.2e}", flush=True)
    assert sym < 1e-13
    del I, r2d, D2, ovov, oooo
    # ---- (T)
    nt = eng.ntriples()
    t0 = time.perf_counter(); full = eng.do_ccsd_t_spatial(); dt = time.perf_counter() - t0
    fl = nt * 12.0 * v**3 * (v + o)
    print("(T) %.3f s  %.1f TFLOP/s (symmetric count)" % (dt, fl / dt / 1e12), full, flush=True)
    cuts = [0, nt // 3, nt // 2 + 7, nt]
    parts = sum(eng.do_ccsd_t_spatial(a, b) for a, b in zip(cuts[:-1], cuts[1:]))
    print("shards", parts, "max rel diff %.2e" % np.max(np.abs(parts - full) / np.abs(full)), flush=True)
    assert np.all(np.isfinite(full)) and np.max(np.abs(parts - full) / np.abs(full)) < 1e-10
    L = orc.blas_lib()
    if L is not None:
        # the engine enumerates block triples of sb occupied orbitals: its first sb(sb+1)(sb+2)/6 sorted triples i <= j <= k < sb are the
        # sb^3 ordered triples of the reference's loop over those orbitals
        sb = eng.t_block_size()
        got = eng.do_ccsd_t_spatial(0, sb * (sb + 1) * (sb + 2) // 6)
        f = lambda a: np.ascontiguousarray(a.ravel(order="F"))
        e = np.concatenate([-2.0 + np.arange(o) / (o - 1), 1.0 + 2.0 * np.arange(v) / (v - 1)])
        args = (o, v, e, f(t1n_t2n[0]), f(t2n), f(vvov), f(oovo), f(oovv))
        ref = np.zeros(4)
        t0 = time.perf_counter()
        for i in range(sb):
            for j in range(sb):
                out = np.zeros(4)
                lo = (i * o + j) * o
                assert L.orcb_ccsd_t(*args, lo, lo + sb, out) == 0
                ref += out
        err = float(np.max(np.abs(got - ref)) / max(1.0, float(np.max(np.abs(ref)))))
        print(f"(T) first block (s = {sb}: {sb**3} ordered triples on the host, %.1f s): device {got}  restatement {ref}  max rel err {err:.2e}"
              % (time.perf_counter() - t0), flush=True)
        assert err < 1e-10
    else:
        print("(T) block check skipped: numpy's OpenBLAS not found")
    eng.close()
    print("worst:", {k: "%.1e" % x for k, x in worst.items()})
    print("ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
