"""Multi-rank runs of the product on the GPU box: ranks are separate processes with their own context, the (T) triples are
split over them and summed by afesp_allreduce_sum -- the sum that replaces the reference's OpenMP reduction (src/ccsd.f90:2091).
A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so two-rank cases use the host-segment
transport; the RCCL transport is exercised with a world of one rank (library loading, communicator, stream use)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import molecules

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "a-fortran-electronic-structure-program_amd", "host")


def _run_ranks(tmp_path, world, name, transport="host", env=None):
    boot = str(tmp_path / "bootstrap")
    procs = []
    for r in range(world):
        out = tmp_path / f"rank{r}.json"
        procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_worker.py"), str(r), str(world), transport,
                                        boot, name, str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                       env=dict(os.environ, **(env or {}))), out))
    res = []
    try:
        for p, out in procs:
            log, _ = p.communicate(timeout=600)
            assert p.returncode == 0, log
            res.append(json.load(open(out)))
    finally:
        # a rank that failed (or timed out) leaves the others waiting for it in a collective: none may outlive the test
        for p, _ in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return res


def _gpus():
    from afesp_amd import capi
    return capi.device_count()


@pytest.mark.parametrize("name,world", [("f2-cc-pvdz", 2), ("n2-cc-pvdz", 3)])
def test_engine_ranks_shard_triples_and_reduce(tmp_path, name, world):
    """BASELINE config 4 in miniature: F2 (and N2) with the triples split over ranks of the Engine."""
    res = _run_ranks(tmp_path, world, name)
    g = molecules.SURVEY_GOLD[name]
    _, _, _, gold = molecules.load(name)
    for r in res:
        assert r["ones"] == [float(world), float(sum(range(world)))]
        assert r["bounds"] == res[0]["bounds"] and r["bounds"][0] == 0
        assert r["sbs"][1] * world == r["sbs"][0] ** 2                     # every rank enumerates the same block order
        assert r["total"] == res[0]["total"] and r["cr_total"] == res[0]["cr_total"]   # fixed-order sum: identical on every rank
        ec, t = r["e_ccsd"], r["total"]
        assert abs(ec - g["ccsd_corr"]) < 1e-8
        assert abs(ec + t[0] - g["ccsd_bt_corr"]) < 1e-8 and abs(ec + t[1] - g["ccsd_pt_corr"]) < 1e-8   # north_star: 1e-8 Eh
        assert abs(t[2] - g["d_bt"]) < 1e-8 and abs(t[3] - g["d_pt"]) < 1e-8
        c = r["cr_total"]
        assert abs(ec + c[4] / c[2] - gold["cr_ccsd_bt_corr"]) < 1e-8 and abs(ec + c[5] / c[3] - gold["cr_ccsd_pt_corr"]) < 1e-8
    # the shards are real shards: every rank evaluated a different, non-empty range and the partial sums add up
    parts = np.array([r["part"] for r in res])
    assert all(b1 > b0 for b0, b1 in zip(res[0]["bounds"][:-1], res[0]["bounds"][1:]))
    assert np.max(np.abs(parts.sum(axis=0) - np.array(res[0]["total"]))) < 1e-13
    assert np.all(np.abs(parts[:, 0]) < abs(res[0]["total"][0]))


@pytest.mark.parametrize("name,world,pp_sym", [("f2-cc-pvdz", 2, "0"), ("n2-cc-pvdz", 3, "1")])
def test_ccsd_iteration_split_over_ranks(tmp_path, name, world, pp_sym):
    """the iteration's o^3 v^3 ring products and the pp-ladder (both of its forms) evaluated slice by
    slice on the ranks, one all-reduce of [PP | partial residual] per iteration -- every rank must walk the reference's own
    iteration table (els.out, 12 decimals) and land on the replica path's energies."""
    res = _run_ranks(tmp_path, world, name, env={"AFESP_CC_SHARD": "1", "AFESP_PP_SYM": pp_sym})
    g = molecules.SURVEY_GOLD[name]
    _, _, _, gold = molecules.load(name)
    for r in res:
        assert r["split"] is True
        assert r["nit"] == gold["cc_iters"][-1][0]
        for (git, ge, gde, grms) in gold["cc_iters"]:
            assert abs(r["en"][git] - ge) < 1e-10, (git, r["en"][git], ge)
        assert r["en"] == res[0]["en"]                                          # replicated state: bit-identical on every rank
        assert abs(r["e_ccsd"] - g["ccsd_corr"]) < 1e-8
        assert abs(r["e_ccsd"] + r["total"][1] - g["ccsd_pt_corr"]) < 1e-8
    plain = _run_ranks(tmp_path, world, name, env={"AFESP_CC_SHARD": "0", "AFESP_PP_SYM": pp_sym})
    assert plain[0]["split"] is False
    assert np.max(np.abs(np.array(plain[0]["en"]) - np.array(res[0]["en"]))) < 1e-11   # split == replicas


@pytest.mark.parametrize("name,world", [("f2-cc-pvdz", 2)])
def test_engine_ranks_over_rccl(tmp_path, name, world):
    """The same two-rank run over the RCCL transport (bootstrap file for the unique id, ncclCommInitRank, ncclAllReduce on the
    engine stream), one GPU per rank: needs a box with at least two GPUs, skipped on the one-GPU test box."""
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs for RCCL ranks (this box has {_gpus()})")
    res = _run_ranks(tmp_path, world, name, transport="rccl")
    g = molecules.SURVEY_GOLD[name]
    for r in res:
        assert r["ones"] == [float(world), float(sum(range(world)))]
        assert r["total"] == res[0]["total"]
        assert abs(r["e_ccsd"] + r["total"][1] - g["ccsd_pt_corr"]) < 1e-8
    split = _run_ranks(tmp_path, world, name, transport="rccl", env={"AFESP_CC_SHARD": "1"})
    _, _, _, gold = molecules.load(name)
    for r in split:
        assert r["split"] is True
        for (git, ge, gde, grms) in gold["cc_iters"]:
            assert abs(r["en"][git] - ge) < 1e-10
        assert r["en"] == split[0]["en"]


def test_bench_two_gpus_over_rccl():
    """`python bench.py --gpus 2` (nccl backend = RCCL) on a box with two GPUs: the product's all-reduce carries every leg, the
    line records the per-rank shard times and whether the iteration was split.  Skipped on the one-GPU test box."""
    if _gpus() < 2:
        pytest.skip(f"needs 2 GPUs (this box has {_gpus()})")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "n2", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout + res.stderr
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    assert "ncclAllReduce" in line["t_allreduce"]
    assert len(line["per_rank"]["t_shard_ms"]) == 2
    for leg in line["real_molecules_same_run"].values():
        assert "ncclAllReduce" in leg["t_allreduce"] and leg["max_abs_error_vs_reference_Eh"] < 1e-8


def test_split_is_opt_in(tmp_path):
    """A communicator alone does not change the algorithm of the CCSD iteration: without AFESP_CC_SHARD=1 (or
    afesp_ccsd_set_split) two ranks run replicas."""
    res = _run_ranks(tmp_path, 2, "f2-cc-pvdz")
    assert all(r["split"] is False for r in res)


def test_rccl_transport_with_one_rank():
    """The RCCL path of the boundary on the one GPU of this box: librccl is opened, a communicator of one rank is created and an
    all-reduce runs on the engine's stream (the sum over one rank is the identity)."""
    from afesp_amd import capi
    with capi.Engine(0) as eng:
        assert len(eng.comm_unique_id()) == 128
        eng.comm_init(0, 1, capi.COMM_RCCL)
        x = np.array([1.5, -2.25, 3.0e-9])
        assert np.array_equal(eng.allreduce_sum(x), x)
        big = np.arange(200.0)
        assert np.array_equal(eng.allreduce_sum(big), big)
        eng.comm_destroy()
        assert np.array_equal(eng.allreduce_sum(x), x)          # no communicator: one rank


def _host_case(tmp_path, name, calc_type):
    src = os.path.join(molecules.GOLDEN, name)
    for f in ("s.dat", "t.dat", "v.dat", "eri.dat", "geom.dat", "guess_in.dat"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), tmp_path)
    text = open(os.path.join(src, "els.in")).read()
    (tmp_path / "els.in").write_text(text.replace("CRCCSD(T)_spatial", calc_type))


@pytest.mark.parametrize("calc_type", ["CCSD(T)_spatial", "CRCCSD(T)_spatial"])
def test_fortran_host_rank_mode_f2(tmp_path, calc_type):
    """BASELINE config 4 through the Fortran host: `els_mgpu.sh 2 host` = two els_amd ranks, F2/cc-pVDZ, (T) sharded, rank 0
    prints the reference's table; energies against the reference's own els.out."""
    from afesp_amd import inputs
    if not os.path.exists(os.path.join(HOST_DIR, "els_amd")):
        pytest.skip("els_amd not built")
    _host_case(tmp_path, "f2-cc-pvdz", calc_type)
    res = subprocess.run([os.path.join(HOST_DIR, "els_mgpu.sh"), "2", "host"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Ranks: 2, transport host" in res.stdout
    (tmp_path / "o").write_text(res.stdout)
    got = inputs.parse_els_out(str(tmp_path / "o"))
    gold = inputs.parse_els_out(os.path.join(molecules.GOLDEN, "f2-cc-pvdz", "els.out"))
    keys = ["rhf_total", "mp2_corr", "ccsd_corr", "ccsd_bt_corr", "ccsd_pt_corr"]
    if calc_type.startswith("CR"):
        keys += ["r_ccsd_bt_corr", "r_ccsd_pt_corr", "cr_ccsd_bt_corr", "cr_ccsd_pt_corr", "d_bt", "d_pt"]
    for k in keys:
        ref = gold[k] if k in gold else molecules.SURVEY_GOLD["f2-cc-pvdz"][k]
        assert abs(got[k] - ref) < 1e-8, (k, got[k], ref)


def test_bench_gpus_flag_spawns_ranks(tmp_path):
    """`python bench.py --gpus 2 --backend gloo` on a one-GPU box: the parent starts two ranks before touching the GPU; the line
    says n_gpus 2 and the ranks counted by the all-reduce of ones are 2."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "n2",
                          "--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    assert "afesp_allreduce_sum" in line["t_allreduce"]
    assert line["value"] > 0 and line["value_survey_count"] > 0
    # the record of a multi-rank run can be audited per rank
    assert len(line["per_rank"]["t_shard_ms"]) == 2 and len(line["per_rank"]["allreduce_us_incl_wait_for_slowest_rank"]) == 2
    assert line["t_shard_ms_min"] <= line["t_shard_ms_max"] and sum(line["per_rank"]["t_shard_triples"]) == line["config"]["triples"]
    assert line["ccsd_split"] is False and "launch-bound" in line["ccsd_split_check"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "n2", "--steps", "3", "--warmup", "1", "--no-extra",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout + one.stderr
    ref = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    assert ref["n_gpus"] == 1 and np.max(np.abs(np.array(ref["e_t"]) - np.array(line["e_t"]))) < 1e-12
    # a launcher whose world differs from --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extra"], capture_output=True, text=True,
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert bad.returncode != 0 and "refusing" in (bad.stdout + bad.stderr)
