﻿!mod$ v1 sum:79a3ab83e6cb0cf0
!need$ 107b15495db65b68 n host_linalg
!need$ d25cf8cf498cc32f n host_support
!need$ 8178ccce6aab5cfe n host_config
!need$ aa745800e0550814 n host_inputs
module host_scf
use host_support,only:dp
use host_support,only:i8
use host_support,only:out
use host_support,only:err
use host_support,only:fail
use host_support,only:seconds
use host_config,only:level_rhf
use host_config,only:level_mp2
use host_config,only:level_ccsd
use host_config,only:level_ccsd_t
use host_config,only:run_config
use host_config,only:read_config
use host_inputs,only:molecule
use host_inputs,only:pair
use host_inputs,only:eri_slot
use host_inputs,only:read_two_index
use host_inputs,only:read_molecule
use host_linalg,only:sym_eig
use host_linalg,only:solve
contains
subroutine rhf(cfg,mol,e_hf,coeff,levels,converged)
type(run_config),intent(in)::cfg
type(molecule),intent(in)::mol
real(8),intent(out)::e_hf
real(8),allocatable,intent(out)::coeff(:,:)
real(8),allocatable,intent(out)::levels(:)
logical(4),intent(out)::converged
end
end
