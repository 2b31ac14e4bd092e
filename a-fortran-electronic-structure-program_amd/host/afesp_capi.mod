﻿!mod$ v1 sum:199029fe129bfbc2
!need$ 0bde2ac47243ead2 i iso_c_binding
!need$ bb381bf46e508468 i __fortran_builtins
module afesp_capi
use,intrinsic::iso_c_binding,only:c_associated
use,intrinsic::iso_c_binding,only:c_funloc
use,intrinsic::iso_c_binding,only:c_funptr
use,intrinsic::iso_c_binding,only:c_f_pointer
use,intrinsic::iso_c_binding,only:c_loc
use,intrinsic::iso_c_binding,only:c_null_funptr
use,intrinsic::iso_c_binding,only:c_null_ptr
use,intrinsic::iso_c_binding,only:c_ptr
use,intrinsic::iso_c_binding,only:c_sizeof
use,intrinsic::iso_c_binding,only:operator(==)
use,intrinsic::iso_c_binding,only:operator(/=)
use,intrinsic::iso_c_binding,only:c_int8_t
use,intrinsic::iso_c_binding,only:c_int16_t
use,intrinsic::iso_c_binding,only:c_int32_t
use,intrinsic::iso_c_binding,only:c_int64_t
use,intrinsic::iso_c_binding,only:c_int128_t
use,intrinsic::iso_c_binding,only:c_int
use,intrinsic::iso_c_binding,only:c_short
use,intrinsic::iso_c_binding,only:c_long
use,intrinsic::iso_c_binding,only:c_long_long
use,intrinsic::iso_c_binding,only:c_signed_char
use,intrinsic::iso_c_binding,only:c_size_t
use,intrinsic::iso_c_binding,only:c_intmax_t
use,intrinsic::iso_c_binding,only:c_intptr_t
use,intrinsic::iso_c_binding,only:c_ptrdiff_t
use,intrinsic::iso_c_binding,only:c_int_least8_t
use,intrinsic::iso_c_binding,only:c_int_fast8_t
use,intrinsic::iso_c_binding,only:c_int_least16_t
use,intrinsic::iso_c_binding,only:c_int_fast16_t
use,intrinsic::iso_c_binding,only:c_int_least32_t
use,intrinsic::iso_c_binding,only:c_int_fast32_t
use,intrinsic::iso_c_binding,only:c_int_least64_t
use,intrinsic::iso_c_binding,only:c_int_fast64_t
use,intrinsic::iso_c_binding,only:c_int_least128_t
use,intrinsic::iso_c_binding,only:c_int_fast128_t
use,intrinsic::iso_c_binding,only:c_float
use,intrinsic::iso_c_binding,only:c_double
use,intrinsic::iso_c_binding,only:c_long_double
use,intrinsic::iso_c_binding,only:c_float_complex
use,intrinsic::iso_c_binding,only:c_double_complex
use,intrinsic::iso_c_binding,only:c_long_double_complex
use,intrinsic::iso_c_binding,only:c_bool
use,intrinsic::iso_c_binding,only:c_char
use,intrinsic::iso_c_binding,only:c_null_char
use,intrinsic::iso_c_binding,only:c_alert
use,intrinsic::iso_c_binding,only:c_backspace
use,intrinsic::iso_c_binding,only:c_form_feed
use,intrinsic::iso_c_binding,only:c_new_line
use,intrinsic::iso_c_binding,only:c_carriage_return
use,intrinsic::iso_c_binding,only:c_horizontal_tab
use,intrinsic::iso_c_binding,only:c_vertical_tab
use,intrinsic::iso_c_binding,only:c_float128
use,intrinsic::iso_c_binding,only:c_float128_complex
use,intrinsic::iso_c_binding,only:c_uint8_t
use,intrinsic::iso_c_binding,only:c_uint16_t
use,intrinsic::iso_c_binding,only:c_uint32_t
use,intrinsic::iso_c_binding,only:c_uint64_t
use,intrinsic::iso_c_binding,only:c_uint128_t
use,intrinsic::iso_c_binding,only:c_unsigned_char
use,intrinsic::iso_c_binding,only:c_unsigned_short
use,intrinsic::iso_c_binding,only:c_unsigned
use,intrinsic::iso_c_binding,only:c_unsigned_long
use,intrinsic::iso_c_binding,only:c_unsigned_long_long
use,intrinsic::iso_c_binding,only:c_uintmax_t
use,intrinsic::iso_c_binding,only:c_uint_fast8_t
use,intrinsic::iso_c_binding,only:c_uint_fast16_t
use,intrinsic::iso_c_binding,only:c_uint_fast32_t
use,intrinsic::iso_c_binding,only:c_uint_fast64_t
use,intrinsic::iso_c_binding,only:c_uint_fast128_t
use,intrinsic::iso_c_binding,only:c_uint_least8_t
use,intrinsic::iso_c_binding,only:c_uint_least16_t
use,intrinsic::iso_c_binding,only:c_uint_least32_t
use,intrinsic::iso_c_binding,only:c_uint_least64_t
use,intrinsic::iso_c_binding,only:c_uint_least128_t
use,intrinsic::iso_c_binding,only:c_f_procpointer
use,intrinsic::__fortran_builtins,only:iso_c_binding$__fortran_builtins$c_associated_c_ptr=>c_associated_c_ptr
private::c_associated
private::c_funloc
private::c_funptr
private::c_f_pointer
private::c_loc
private::c_null_funptr
private::c_null_ptr
private::c_ptr
private::c_sizeof
private::operator(==)
private::operator(/=)
private::c_int8_t
private::c_int16_t
private::c_int32_t
private::c_int64_t
private::c_int128_t
private::c_int
private::c_short
private::c_long
private::c_long_long
private::c_signed_char
private::c_size_t
private::c_intmax_t
private::c_intptr_t
private::c_ptrdiff_t
private::c_int_least8_t
private::c_int_fast8_t
private::c_int_least16_t
private::c_int_fast16_t
private::c_int_least32_t
private::c_int_fast32_t
private::c_int_least64_t
private::c_int_fast64_t
private::c_int_least128_t
private::c_int_fast128_t
private::c_float
private::c_double
private::c_long_double
private::c_float_complex
private::c_double_complex
private::c_long_double_complex
private::c_bool
private::c_char
private::c_null_char
private::c_alert
private::c_backspace
private::c_form_feed
private::c_new_line
private::c_carriage_return
private::c_horizontal_tab
private::c_vertical_tab
private::c_float128
private::c_float128_complex
private::c_uint8_t
private::c_uint16_t
private::c_uint32_t
private::c_uint64_t
private::c_uint128_t
private::c_unsigned_char
private::c_unsigned_short
private::c_unsigned
private::c_unsigned_long
private::c_unsigned_long_long
private::c_uintmax_t
private::c_uint_fast8_t
private::c_uint_fast16_t
private::c_uint_fast32_t
private::c_uint_fast64_t
private::c_uint_fast128_t
private::c_uint_least8_t
private::c_uint_least16_t
private::c_uint_least32_t
private::c_uint_least64_t
private::c_uint_least128_t
private::c_f_procpointer
private::iso_c_binding$__fortran_builtins$c_associated_c_ptr
interface
function afesp_ctx_create(device,ctx) bind(c,name="afesp_ctx_create") result(rc)
import::c_ptr
integer(4),value::device
type(c_ptr),intent(out)::ctx
integer(4)::rc
end
end interface
interface
subroutine afesp_ctx_destroy(ctx) bind(c,name="afesp_ctx_destroy")
import::c_ptr
type(c_ptr),value::ctx
end
end interface
interface
function afesp_last_error(ctx) bind(c,name="afesp_last_error") result(msg)
import::c_ptr
type(c_ptr),value::ctx
type(c_ptr)::msg
end
end interface
interface
function afesp_neri(nbasis) bind(c,name="afesp_neri") result(n)
integer(8),value::nbasis
integer(8)::n
end
end interface
interface
function afesp_ao2mo_mp2(ctx,nbasis,nocc,canon_coeff,canon_levels,eri_packed,eri_mo_packed,e_mp2) bind(c,name="afesp_ao2mo_mp2") result(rc)
import::c_ptr
type(c_ptr),value::ctx
integer(8),value::nbasis
integer(8),value::nocc
real(8),intent(in)::canon_coeff(1_8:*)
real(8),intent(in)::canon_levels(1_8:*)
real(8),intent(in)::eri_packed(1_8:*)
type(c_ptr),value::eri_mo_packed
real(8),intent(out)::e_mp2
integer(4)::rc
end
end interface
interface
function afesp_ccsd_init(ctx,nocc,nvirt,eri_mo_packed,canon_levels,diis_n_errmat) bind(c,name="afesp_ccsd_init") result(rc)
import::c_ptr
type(c_ptr),value::ctx
integer(8),value::nocc
integer(8),value::nvirt
type(c_ptr),value::eri_mo_packed
real(8),intent(in)::canon_levels(1_8:*)
integer(4),value::diis_n_errmat
integer(4)::rc
end
end interface
interface
function afesp_ccsd_energy(ctx,e_tol,t_tol,energy,rms_sq,converged) bind(c,name="afesp_ccsd_energy") result(rc)
import::c_ptr
type(c_ptr),value::ctx
real(8),value::e_tol
real(8),value::t_tol
real(8),intent(out)::energy
real(8),intent(out)::rms_sq
integer(4),intent(out)::converged
integer(4)::rc
end
end interface
interface
function afesp_ccsd_iterate(ctx,e_tol,t_tol,energy,rms_sq,converged) bind(c,name="afesp_ccsd_iterate") result(rc)
import::c_ptr
type(c_ptr),value::ctx
real(8),value::e_tol
real(8),value::t_tol
real(8),intent(out)::energy
real(8),intent(out)::rms_sq
integer(4),intent(out)::converged
integer(4)::rc
end
end interface
interface
function afesp_ccsd_diis(ctx) bind(c,name="afesp_ccsd_diis") result(rc)
import::c_ptr
type(c_ptr),value::ctx
integer(4)::rc
end
end interface
interface
function afesp_ccsd_get_amplitudes(ctx,t1,t2) bind(c,name="afesp_ccsd_get_amplitudes") result(rc)
import::c_ptr
type(c_ptr),value::ctx
real(8),intent(out)::t1(1_8:*)
real(8),intent(out)::t2(1_8:*)
integer(4)::rc
end
end interface
interface
function afesp_ccsd_t_ntriples(nocc) bind(c,name="afesp_ccsd_t_ntriples") result(n)
integer(8),value::nocc
integer(8)::n
end
end interface
interface
function afesp_ccsd_t(ctx,t_begin,t_end,out) bind(c,name="afesp_ccsd_t") result(rc)
import::c_ptr
type(c_ptr),value::ctx
integer(8),value::t_begin
integer(8),value::t_end
real(8),intent(out)::out(1_8:4_8)
integer(4)::rc
end
end interface
contains
function afesp_error_text(ctx) result(text)
type(c_ptr),intent(in)::ctx
character(:,1),allocatable::text
end
end
