"""CPU-side checks: C-ABI exports, header/library agreement, input parsing, the Fortran host's RHF."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import molecules
from afesp_amd import capi, inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = capi.load_library()
    header = open(os.path.join(ROOT, "include", "afesp.h")).read()
    declared = set(re.findall(r"\b(afesp_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(capi.EXPORTS) <= declared
    assert lib.afesp_neri(28) == 82621


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.AfespError):
        capi.Engine(0)


def test_namelist_defaults_and_calc_types(tmp_path):
    p = tmp_path / "els.in"
    p.write_text('&elsinput\ncalc_type="RCCSD[T]_spatial",\nccsd_maxiter = 7\n/\n')
    si = inputs.read_els_in(str(p))
    assert si.ccsd_maxiter == 7 and si.scf_read_guess is False and si.ccsd_diis_n_errmat == 8
    assert si.ccsd_t_renorm and not si.ccsd_t_paren and si.level == "CCSD(T)"
    p.write_text('&elsinput\ncalc_type="nonsense"\n/\n')
    with pytest.raises(ValueError):
        inputs.read_els_in(str(p))


def test_packed_reader_matches_eri_index_rule():
    si, ints, res, gold = molecules.load("h2o-cc-pvdz")
    n = ints.nbasis
    assert ints.eri.size == inputs.neri(n)
    # first data line of eri.dat is (1 1|1 1)
    first = float(open(os.path.join(molecules.GOLDEN, "h2o-cc-pvdz", "eri.dat")).readline().split()[4])
    assert ints.eri[0] == first
    assert inputs.eri_index(3, 1, 2, 0) == inputs.eri_index(0, 2, 1, 3)


def test_fortran_host_rhf_on_cpu(tmp_path):
    exe = os.path.join(ROOT, "a-fortran-electronic-structure-program_amd", "host", "els_amd")
    if not os.path.exists(exe):
        pytest.skip("els_amd not built")
    src = os.path.join(molecules.GOLDEN, "f2-cc-pvdz")
    for f in ("s.dat", "t.dat", "v.dat", "eri.dat", "geom.dat"):
        shutil.copy(os.path.join(src, f), tmp_path)
    (tmp_path / "els.in").write_text(open(os.path.join(src, "els.in")).read().replace("CRCCSD(T)_spatial", "RHF"))
    res = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    (tmp_path / "o").write_text(res.stdout)
    got = inputs.parse_els_out(str(tmp_path / "o"))
    gold = inputs.parse_els_out(os.path.join(src, "els.out"))
    assert abs(got["rhf_total"] - gold["rhf_total"]) < 2e-9
    assert len(got["scf_iters"]) == len(gold["scf_iters"])
    for a, b in zip(got["scf_iters"], gold["scf_iters"]):
        assert abs(a[1] - b[1]) < 5e-9


def test_bench_script_defines_everything_it_calls():
    """bench.py only runs on a GPU box; here: it compiles and every plain name it loads resolves (a builtin, a module-level
    definition or import, or a name bound inside the function that uses it)."""
    import ast
    import builtins
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    top = set(dir(builtins))
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            top.add(node.name)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            top.update((a.asname or a.name).split(".")[0] for a in node.names)
        elif isinstance(node, ast.Assign):
            top.update(t.id for t in node.targets if isinstance(t, ast.Name))
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        bound = {a.arg for a in fn.args.args + fn.args.kwonlyargs}
        for sub in ast.walk(fn):
            if isinstance(sub, ast.Name) and isinstance(sub.ctx, (ast.Store, ast.Del)):
                bound.add(sub.id)
            elif isinstance(sub, (ast.FunctionDef, ast.Lambda)):
                if isinstance(sub, ast.FunctionDef):
                    bound.add(sub.name)
                bound.update(a.arg for a in sub.args.args)
            elif isinstance(sub, (ast.Import, ast.ImportFrom)):
                bound.update((a.asname or a.name).split(".")[0] for a in sub.names)
            elif isinstance(sub, ast.ExceptHandler) and sub.name:
                bound.add(sub.name)
        missing = {sub.id for sub in ast.walk(fn) if isinstance(sub, ast.Name) and isinstance(sub.ctx, ast.Load)} - bound - top
        assert not missing, (fn.name, sorted(missing))


def test_bench_flop_counts_match_the_survey_table():
    """F_it (reference formulation, full pp-ladder dgemm) and F_T for the BASELINE configurations."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ref_iter = lambda o, v: bench.flops_iter(o, v, 2 * o**2 * v**4)
    for (o, v), f_it, f_t in [((5, 19), 2.278e7, None), ((5, 53), 7.320e8, None), ((7, 21), 7.616e7, 1.067e9),
                              ((9, 19), 1.099e8, 1.680e9), ((20, 200), 2.251e12, 1.690e14)]:
        assert abs(ref_iter(o, v) - f_it) < 6e-4 * f_it, (o, v, ref_iter(o, v))
        if f_t:
            assert abs(bench.flops_t_ref(o, v) - f_t) < 6e-4 * f_t
    assert abs(bench.flops_t_sym(20, 200) - 3.25e13) < 2e-3 * 3.25e13
    # the two cheaper evaluations of the ladder are counted as what they execute
    assert bench.flops_iter(20, 200) < ref_iter(20, 200)


def test_multi_rank_launcher_ends_the_job_when_a_rank_fails(tmp_path):
    """host/els_mgpu.sh: a failing rank ends the job -- the other ranks are terminated, the exit status is non-zero, nothing
    is left running (here every rank fails at once: there is no eri.dat)."""
    import shutil
    import subprocess
    import time
    host = os.path.join(ROOT, "a-fortran-electronic-structure-program_amd", "host")
    if not os.path.exists(os.path.join(host, "els_amd")):
        pytest.skip("els_amd not built")
    src = os.path.join(ROOT, "tests", "golden", "f2-cc-pvdz")
    for f in ("s.dat", "t.dat", "v.dat", "geom.dat", "els.in"):
        shutil.copy(os.path.join(src, f), tmp_path)
    t0 = time.time()
    res = subprocess.run([os.path.join(host, "els_mgpu.sh"), "3", "host"], cwd=tmp_path, capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, AFESP_JOB_TIMEOUT="60"))
    assert res.returncode != 0
    assert "a rank failed" in res.stderr
    assert time.time() - t0 < 60


def test_ring_launches_give_way_to_the_gather_kernel_at_the_row_offset_bound(monkeypatch):
    """csrc/ring.hip addresses a row of an operand by a 32-bit byte offset: the grouped ring launches of the LDS-DMA GEMM apply from
    o v = 3584 on and only while 8 Kc o v < 4 GiB (Kc = o v rounded up to whole K steps of 16); past that -- and below the lower bound --
    the six ring products stay on the gather kernel, whose offsets are 64-bit.  Host logic, no device: faked extents either side of
    both bounds (o v = 23 128 is the last product of o = 56 that fits, 23 184 the first that does not)."""
    from afesp_amd import capi
    L = capi.load_library()
    monkeypatch.delenv("AFESP_RING_TG", raising=False)
    monkeypatch.delenv("AFESP_RING_TG_MIN", raising=False)
    assert L.afesp_test_ring_path(20, 200) == 1 and L.afesp_test_ring_path(40, 360) == 1
    assert L.afesp_test_ring_path(16, 160) == 0                     # o v = 2560 < 3584
    assert L.afesp_test_ring_path(56, 413) == 1                     # 8 * 23136 * 23128 = 4.2807e9 < 2^32 - 4096
    assert L.afesp_test_ring_path(56, 414) == 0                     # 8 * 23184 * 23184 = 4.3000e9: the gather kernel
    assert L.afesp_test_ring_path(64, 512) == 0 and L.afesp_test_ring_path(100, 1000) == 0
    for o, v in ((56, 413), (56, 414), (30, 300), (72, 321), (73, 317)):
        ov = o * v
        kc = (ov + 15) // 16 * 16
        assert L.afesp_test_ring_path(o, v) == (1 if ov >= 3584 and 8 * kc * ov < 2**32 - 4096 else 0), (o, v)
    monkeypatch.setenv("AFESP_RING_TG", "0")
    assert L.afesp_test_ring_path(20, 200) == 0


def test_every_environment_variable_is_defined_in_one_place_and_listed_in_the_header():
    """csrc/knobs.h is the only file of the library that reads the environment; the AFESP_* names it parses are exactly the ones
    include/afesp.h lists (test-only path selectors, tuning, diagnostics) -- plus AFESP_SO_FOO_AS_PUBLISHED / AFESP_LIBRARY, which belong
    to the hosts, not to the library."""
    import re
    csrc = os.path.join(ROOT, "a-fortran-electronic-structure-program_amd", "csrc")
    readers = []
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")) and fn != "knobs.h":
            if re.search(r"\bgetenv\s*\(", open(os.path.join(csrc, fn)).read()):
                readers.append(fn)
    assert readers == [], readers
    parsed = set(re.findall(r'"(AFESP_[A-Z0-9_]+)"', open(os.path.join(csrc, "knobs.h")).read()))
    header = open(os.path.join(ROOT, "include", "afesp.h")).read()
    block = header[header.index("Environment variables."):header.index("#ifndef AFESP_H")]
    listed = set(re.findall(r"AFESP_[A-Z0-9_]+", block)) - {"AFESP_H"}
    assert parsed == listed, (sorted(parsed - listed), sorted(listed - parsed))
