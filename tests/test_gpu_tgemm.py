"""The LDS-DMA GEMM kernel (csrc/tgemm.hip) on its own: tools/tgemm_check.hip runs it on small random grouped problems against a
host loop -- what the (T) and AO->MO tests reach only through whole calculations: every K tail, tiles of two K steps, partial row and
column tiles, pair stores, long tile streams per workgroup and tiles drawn from the ticket counters."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "tgemm_check.hip")
KERNEL = os.path.join(ROOT, "a-fortran-electronic-structure-program_amd", "csrc", "tgemm.hip")


@pytest.fixture(scope="module")
def check_bin(tmp_path_factory):
    """the stand-alone check program: the prebuilt one if it is newer than the sources, else compiled here (hipcc, ~20 s)"""
    pre = os.path.join(ROOT, "tools", "tgemm_check_bin")
    if os.path.exists(pre) and os.path.getmtime(pre) >= max(os.path.getmtime(SRC), os.path.getmtime(KERNEL)):
        return pre
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = str(tmp_path_factory.mktemp("tgemm") / "tgemm_check_bin")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", SRC, "-o", out], check=True, capture_output=True, timeout=600)
    return out


def _run(binary, shape, **env):
    res = subprocess.run([binary] + [str(x) for x in shape], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, **{k: str(v) for k, v in env.items()}))
    assert res.returncode == 0 and "bad 0 of" in res.stdout and "untouched 0" in res.stdout, (shape, env, res.stdout[-600:], res.stderr[-300:])


@pytest.mark.parametrize("kv", [48, 47, 44, 41, 40, 37, 36, 33])
def test_k_tails(check_bin, kv):
    """valid summation length 48 ... 33 of a padded 48 (ktail4 = 4, 4, 3, 3, 2, 2, 1, 1), plain and pair stores"""
    _run(check_bin, (300, 48, 200, 70), TG_KV=kv)
    _run(check_bin, (300, 48, 200, 70), TG_KV=kv, TG_PAIRS=1, AFESP_TG_GRID=2)


@pytest.mark.parametrize("shape", [(300, 48, 200, 70), (361, 32, 96, 20), (900, 48, 150, 150), (1000, 32, 300, 130), (40, 16, 10, 10),
                                   (3000, 64, 300, 130), (512, 64, 256, 128)])
def test_shapes_streams_and_tickets(check_bin, shape):
    """partial tiles in both directions, groups of two K steps beside longer ones, one tile per workgroup / long streams / tiles
    drawn from the per-XCD ticket counters (forced for these small launches)"""
    _run(check_bin, shape)
    _run(check_bin, shape, AFESP_TG_GRID=3)
    pairs = {"TG_PAIRS": 1} if shape[2] % 2 == 0 and shape[3] % 2 == 0 else {}
    for grid in (8, 16, 64):
        _run(check_bin, shape, AFESP_TG_DYNAMIC=2, AFESP_TG_GRID=grid, **pairs)


@pytest.mark.parametrize("bm", [128, 96])
@pytest.mark.parametrize("shape", [(220, 48, 200, 70), (92, 32, 300, 130), (300, 48, 200, 70), (210, 64, 150, 150), (190, 48, 96, 20), (900, 48, 150, 150),
                                   (224, 32, 256, 128), (97, 32, 130, 40), (40, 16, 10, 10)])
def test_tiles_of_96_rows_where_the_rows_end(check_bin, shape, bm):
    """The instantiation with 96-row tiles (TgArgs::bm; the AO->MO transforms' 220 rows = 128 + 96): m-tiles of 128 rows whose last one
    runs 96 rows high when at most 96 are left, and m-tiles of 96 rows throughout -- tile streams that change height from tile to
    tile, one tile per workgroup, tiles drawn from the ticket counters, K tails and pair stores."""
    _run(check_bin, shape, TG_BM=bm)
    _run(check_bin, shape, TG_BM=bm, AFESP_TG_GRID=3)
    _run(check_bin, shape, TG_BM=bm, AFESP_TG_GRID=2, TG_KV=shape[1] - 5)
    if shape[2] % 2 == 0 and shape[3] % 2 == 0:
        _run(check_bin, shape, TG_BM=bm, TG_PAIRS=1, AFESP_TG_DYNAMIC=2, AFESP_TG_GRID=8)
