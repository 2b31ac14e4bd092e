﻿!mod$ v1 sum:8178ccce6aab5cfe
!need$ d25cf8cf498cc32f n host_support
module host_config
use host_support,only:dp
use host_support,only:i8
use host_support,only:out
use host_support,only:err
use host_support,only:fail
use host_support,only:seconds
integer(4),parameter::level_rhf=0_4
integer(4),parameter::level_mp2=1_4
integer(4),parameter::level_ccsd=2_4
integer(4),parameter::level_ccsd_t=3_4
type::run_config
character(40_4,1)::calc_type="CCSD(T)_spatial                         "
real(8)::scf_e_tol=9.99999999999999954748111825886258685613938723690807819366455078125e-7_8
real(8)::scf_d_tol=9.99999999999999954748111825886258685613938723690807819366455078125e-7_8
real(8)::ccsd_e_tol=9.99999999999999954748111825886258685613938723690807819366455078125e-7_8
real(8)::ccsd_t_tol=9.99999999999999954748111825886258685613938723690807819366455078125e-7_8
integer(4)::scf_diis_n_errmat=6_4
integer(4)::ccsd_diis_n_errmat=8_4
integer(4)::scf_maxiter=50_4
integer(4)::ccsd_maxiter=50_4
logical(4)::write_fcidump=.false._4
logical(4)::scf_read_guess=.false._4
logical(4)::scf_write_guess=.false._4
integer(4)::level=3_4
logical(4)::paren=.false._4
logical(4)::renorm=.false._4
end type
contains
subroutine read_config(cfg)
type(run_config),intent(out)::cfg
end
end
