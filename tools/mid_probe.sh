cd /root/repo
AFESP_FUSED_DEBUG=1 python - <<'PY' 2>&1 | grep -v "^      M" | head -30
import sys
sys.path.insert(0, "a-fortran-electronic-structure-program_amd")
from afesp_amd.capi import Engine
with Engine(0) as e:
    e.synthetic_init(10, 100, 0.02, 1, 8); e.ccsd_energy()
    for _ in range(3): e.ccsd_iterate(); e.ccsd_diis()
PY
