#!/bin/bash
# build_fused_diag.sh -- libraries whose fused_gemm_kernel (csrc/fused.hip) leaves out the MFMAs (diag1), the operand loads (diag2) or
# both (diag3): what a product stage costs without them (select with AFESP_LIBRARY=build/ab/libafesp_fdiagN.so; results are wrong)
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$HERE/a-fortran-electronic-structure-program_amd/csrc"
OUT="$HERE/build/ab"
mkdir -p "$OUT"
make -C "$CS" -j8 > /dev/null
for v in 1 2 3; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DAFESP_FUSED_DIAG=$v -c "$CS/fused.hip" -o "$OUT/fused_d$v.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libafesp_fdiag$v.so" "$OUT/fused_d$v.o" \
      "$CS"/{gett,gett_grouped,tgemm,contract,kernels,ccsd,ccsd_so,triples,comm,capi}.o -ldl -lpthread && echo "built fdiag$v" ) &
done
wait
