#!/bin/bash
# fresh_process.sh [molecule] [calc_type] [runs] -- wall time of els_amd in a fresh process on a bundled molecule, with and without
# the background code preload (AFESP_NO_PRELOAD=1), from the host's own "Total execution time" and stage lines.
MOL=${1:-n2-cc-pvdz}; CALC=${2:-CCSD(T)_spatial}; RUNS=${3:-3}
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
D=$(mktemp -d /tmp/afesp_fresh.XXXXXX); cd "$D"
cp "$R"/tests/golden/$MOL/{s.dat,t.dat,v.dat,eri.dat,geom.dat,els.in} . ; cp "$R"/tests/golden/$MOL/guess_in.dat . 2>/dev/null
sed -i "s/CRCCSD(T)_spatial/$CALC/" els.in
for mode in 1 0; do
  for i in $(seq $RUNS); do
    AFESP_NO_PRELOAD=$mode "$R"/a-fortran-electronic-structure-program_amd/host/els_amd > out.txt 2> err.txt || { cat err.txt; exit 1; }
    echo "no_preload=$mode $(grep -E 'Time taken for restricted (Hartree|CCSD:)|Total execution' out.txt | sed 's/Time taken for restricted //; s/  */ /g' | tr '\n' ';')"
  done
done
rm -rf "$D"
