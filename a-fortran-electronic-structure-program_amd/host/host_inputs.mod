﻿!mod$ v1 sum:aa745800e0550814
!need$ d25cf8cf498cc32f n host_support
module host_inputs
use host_support,only:dp
use host_support,only:i8
use host_support,only:out
use host_support,only:err
use host_support,only:fail
use host_support,only:seconds
type::molecule
integer(4)::nbasis=0_4
integer(4)::natoms=0_4
integer(4)::nel=0_4
integer(4)::nocc=0_4
integer(4)::nvirt=0_4
real(8)::e_nuc=0._8
real(8),allocatable::ovlp(:,:)
real(8),allocatable::hcore(:,:)
real(8),allocatable::eri(:)
end type
contains
pure function pair(i,j) result(ij)
integer(4),intent(in)::i
integer(4),intent(in)::j
integer(8)::ij
end
pure function eri_slot(i,j,k,l) result(ijkl)
integer(4),intent(in)::i
integer(4),intent(in)::j
integer(4),intent(in)::k
integer(4),intent(in)::l
integer(8)::ijkl
end
subroutine read_two_index(file,mat,n)
character(*,1),intent(in)::file
real(8),allocatable,intent(inout)::mat(:,:)
integer(4),intent(inout)::n
end
subroutine read_molecule(mol)
type(molecule),intent(out)::mol
end
end
