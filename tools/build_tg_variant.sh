#!/bin/bash
# build_tg_variant.sh NAME [git-rev] -- libafesp_hip.so with csrc/tgemm.hip (+ tgemm.h, triples.hip) taken from a git revision (default:
# the working tree) for A/B runs of the LDS-DMA GEMM in one GPU session (AFESP_LIBRARY=tools/ab/libafesp_NAME.so).
set -e
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
CS="$HERE/a-fortran-electronic-structure-program_amd/csrc"
OUT="$HERE/tools/ab"
NAME="$1"; REV="$2"
mkdir -p "$OUT/src_$NAME"
make -C "$CS" -j8 > /dev/null
cp "$CS"/*.h "$CS"/*.hip "$OUT/src_$NAME/"
mkdir -p "$OUT/include" && cp "$HERE/include/afesp.h" "$OUT/include/" 2>/dev/null || true
if [ -n "$REV" ]; then
  for f in tgemm.hip tgemm.h triples.hip; do git -C "$HERE" show "$REV:a-fortran-electronic-structure-program_amd/csrc/$f" > "$OUT/src_$NAME/$f"; done
fi
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -I$HERE/include ${EXTRA_FLAGS:-}"
( cd "$OUT/src_$NAME" && sed -i 's#"../../include/afesp.h"#"afesp.h"#' *.hip *.h 2>/dev/null; /opt/rocm/bin/hipcc $F -c tgemm.hip -o tgemm.o && /opt/rocm/bin/hipcc $F -c triples.hip -o triples.o )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libafesp_$NAME.so" "$OUT/src_$NAME/tgemm.o" "$OUT/src_$NAME/triples.o" \
  "$CS"/{gett,gett_grouped,contract,kernels,ccsd,ccsd_so,comm,capi}.o -ldl -lpthread && echo "built $NAME"
