#!/bin/bash
# kernel_regs.sh VARIANT -- device asm of gett.hip (UNIT=gett_grouped: the grouped kernels, add their flags through EXTRA_FLAGS) for one schedule variant (/tmp/gett_vN.s) + VGPR / spill counts of the 8-wave kernels
V=${1:-0}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -S --cuda-device-only -DAFESP_GETT_VARIANT=$V ${EXTRA_FLAGS:-} \
    "$HERE/a-fortran-electronic-structure-program_amd/csrc/${UNIT:-gett}.hip" -o /tmp/gett_v$V.s 2>&1 | grep -v "hip-link"
python3 - /tmp/gett_v$V.s <<'PY'
import re,sys
s=open(sys.argv[1]).read()
md=s[s.index('amdhsa.kernels'):]
for m in re.finditer(r"\.name:\s+(_ZN5afesp11gett_kernelILi4ELi2ELi4ELi4E\S+)\n.*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)",md,re.S):
    print(sys.argv[1], m.group(1)[25:62], "vgpr", m.group(2), "spill", m.group(3))
PY
