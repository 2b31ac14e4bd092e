"""world_size-2 gloo run of the multi-rank (T) layout: shard ranges + the single all-reduce (CPU, no GPU needed).
The per-rank partial sums come from the oracle evaluated on each rank's slice."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from afesp_amd import dist as adist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "a-fortran-electronic-structure-program_amd"))
    import torch.distributed as dist
    import molecules
    import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o, v = 4, 7
    n, e, eri = molecules.synthetic_system(o, v, scale=0.05)
    cc = orc.OracleCC(o, v, eri, e, 8)
    cc.solve(40, 1e-8, 1e-9)
    lo, hi = adist.shard_range(o ** 3, rank, world)
    part = cc.triples(e, lo, hi)
    total = adist.allreduce_scalars(part)
    full = cc.triples(e)
    ret[rank] = (lo, hi, float(np.max(np.abs(total - full))))
    dist.destroy_process_group()


def test_two_rank_triples_allreduce_matches_single_rank():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    assert ret[0][0] == 0 and ret[0][1] == ret[1][0] and ret[1][1] == 4 ** 3
    assert ret[0][2] < 1e-12 and ret[1][2] < 1e-12


def test_shard_ranges_partition_every_world_size():
    for n in (1, 35, 84, 165, 1540):
        for w in (1, 2, 3, 4, 8):
            cuts = [adist.shard_range(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
