#!/bin/bash
# build_variants.sh V1 V2 ... -- builds of libafesp_hip.so that differ only in the GEMM schedule variant (gett.hip and
# gett_grouped.hip, AFESP_GETT_VARIANT; EXTRA_FLAGS for both, GROUPED_EXTRA replaces the grouped unit's own flags) for A/B runs in one GPU session (tools/ab_gemm.py, tools/ab_triples.py).  Output: build/ab/libafesp_vN.so
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$HERE/a-fortran-electronic-structure-program_amd/csrc"
OUT="$HERE/build/ab"
mkdir -p "$OUT"
make -C "$CS" -j8 > /dev/null
GROUPED_FLAGS="-mllvm -amdgpu-sched-strategy=max-memory-clause"   # as in csrc/Makefile
for v in "$@"; do
  ( F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -DAFESP_GETT_VARIANT=$v ${EXTRA_FLAGS:-}"
    /opt/rocm/bin/hipcc $F -c "$CS/gett.hip" -o "$OUT/gett_v$v.o" &&
    /opt/rocm/bin/hipcc $F ${GROUPED_EXTRA:-$GROUPED_FLAGS} -c "$CS/gett_grouped.hip" -o "$OUT/gett_grouped_v$v.o" &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libafesp_v$v.so" "$OUT/gett_v$v.o" "$OUT/gett_grouped_v$v.o" \
      "$CS"/{tgemm,contract,kernels,ccsd,ccsd_so,triples,comm,capi}.o -ldl -lpthread && echo "built v$v" ) &
done
wait
