#!/usr/bin/env python3
"""Time of each rank's (T) shard at config 5 for N = 1, 2, 4, 8 ranks, run one after the other on one GPU: the slowest
shard bounds the multi-GPU (T) time (the shards are independent; only 4 doubles are reduced)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
from afesp_amd.capi import Engine
from afesp_amd.dist import shard_range
o, v = 20, 200
eng = Engine(0)
eng.synthetic_init(o, v, 0.005, 12345, 8)
eng.ccsd_energy(); eng.ccsd_iterate(); eng.ccsd_diis()
nt = eng.ntriples()
eng.do_ccsd_t_spatial_plain(0, nt)
whole = None
for N in (1, 2, 4, 8):
    ts = []
    bounds = eng.shard_bounds(N) if len(sys.argv) < 2 else None   # any argument: equal counts instead
    for r in range(N):
        lo, hi = (bounds[r], bounds[r + 1]) if bounds else shard_range(nt, r, N)
        eng.do_ccsd_t_spatial_plain(lo, hi)               # plan + warm
        t0 = time.perf_counter(); eng.do_ccsd_t_spatial_plain(lo, hi); ts.append(time.perf_counter() - t0)
    if whole is None:
        whole = ts[0]
    print(f"N={N}: shard times (ms) " + " ".join("%.0f" % (x * 1e3) for x in ts) + f"  max {max(ts)*1e3:.0f} ms  sum/max/N = {sum(ts)/max(ts)/N:.2f}  "
          f"(T) of the slowest rank vs one GPU: {whole / max(ts):.2f}x")
eng.close()
