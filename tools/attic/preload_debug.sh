R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
D=$(mktemp -d /tmp/afesp_fresh.XXXXXX); cd "$D"
cp "$R"/tests/golden/n2-cc-pvdz/{s.dat,t.dat,v.dat,eri.dat,geom.dat,els.in} . ; cp "$R"/tests/golden/n2-cc-pvdz/guess_in.dat . 2>/dev/null
sed -i "s/CRCCSD(T)_spatial/CCSD(T)_spatial/" els.in
for i in 1 2 3; do AFESP_PRELOAD_DEBUG=1 "$R"/a-fortran-electronic-structure-program_amd/host/els_amd > out.txt 2> err.txt; grep -E "preload" err.txt | tr '\n' ';'; echo; grep -E "Time taken for restricted (Hartree|CCSD:)|Total execution" out.txt | sed 's/Time taken for restricted //; s/  */ /g' | tr '\n' ';'; echo; done
