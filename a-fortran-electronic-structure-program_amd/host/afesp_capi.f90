!> ISO_C_BINDING view of include/afesp.h -- the thin C-ABI between the Fortran host and the HIP engine.
!> Every routine returns a status (0 = ok); the host maps non-zero to its error() (reference: src/error_handling.f90:7-20).
module afesp_capi
   use, intrinsic :: iso_c_binding
   implicit none
   private
   public :: afesp_ctx_create, afesp_ctx_destroy, afesp_last_error, afesp_ao2mo_mp2, afesp_ccsd_init, afesp_ccsd_energy, &
             afesp_ccsd_iterate, afesp_ccsd_diis, afesp_ccsd_get_amplitudes, afesp_ccsd_t, afesp_ccsd_t_ntriples, &
             afesp_neri, afesp_error_text, afesp_ccsd_cr_intermediates, afesp_ccsd_t_cr, afesp_ccsd_so_init, &
             afesp_ccsd_so_energy, afesp_ccsd_so_iterate, afesp_ccsd_so_diis, afesp_ccsd_so_t, afesp_ccsd_so_t_ntriples, &
             afesp_read_eri_text, afesp_write_fcidump, afesp_build_fock, afesp_ccsd_t_plain, afesp_device_count, &
             afesp_comm_init, afesp_comm_destroy, afesp_allreduce_sum, afesp_ccsd_t_shard_bounds, afesp_ccsd_t_block_size, &
             AFESP_COMM_RCCL, AFESP_COMM_HOST

   integer(c_int), parameter :: AFESP_COMM_RCCL = 0, AFESP_COMM_HOST = 1

   interface
      function afesp_ctx_create(device, ctx) bind(C, name='afesp_ctx_create') result(rc)
         import :: c_int, c_ptr
         integer(c_int), value :: device
         type(c_ptr), intent(out) :: ctx
         integer(c_int) :: rc
      end function
      subroutine afesp_ctx_destroy(ctx) bind(C, name='afesp_ctx_destroy')
         import :: c_ptr
         type(c_ptr), value :: ctx
      end subroutine
      function afesp_last_error(ctx) bind(C, name='afesp_last_error') result(msg)
         import :: c_ptr
         type(c_ptr), value :: ctx
         type(c_ptr) :: msg
      end function
      function afesp_neri(nbasis) bind(C, name='afesp_neri') result(n)
         import :: c_int64_t
         integer(c_int64_t), value :: nbasis
         integer(c_int64_t) :: n
      end function
      !> replaces `call do_mp2_spatial(sys, int_store)` (reference src/main.F90:98)
      function afesp_ao2mo_mp2(ctx, nbasis, nocc, canon_coeff, canon_levels, eri_packed, eri_mo_packed, e_mp2) &
         bind(C, name='afesp_ao2mo_mp2') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: nbasis, nocc
         real(c_double), intent(in) :: canon_coeff(*), canon_levels(*)
         type(c_ptr), value :: eri_packed             ! c_loc(int_store%eri), or c_null_ptr after afesp_read_eri_text
         type(c_ptr), value :: eri_mo_packed          ! c_null_ptr keeps the MO integrals on the device only
         real(c_double), intent(out) :: e_mp2
         integer(c_int) :: rc
      end function
      !> replaces init_cc + init_diis_cc_t (reference src/ccsd.f90:313-316)
      function afesp_ccsd_init(ctx, nocc, nvirt, eri_mo_packed, canon_levels, diis_n_errmat) &
         bind(C, name='afesp_ccsd_init') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: nocc, nvirt
         type(c_ptr), value :: eri_mo_packed
         real(c_double), intent(in) :: canon_levels(*)
         integer(c_int), value :: diis_n_errmat
         integer(c_int) :: rc
      end function
      !> update_cc_energy on the current amplitudes (reference src/ccsd.f90:325)
      function afesp_ccsd_energy(ctx, e_tol, t_tol, energy, rms_sq, converged) bind(C, name='afesp_ccsd_energy') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: ctx
         real(c_double), value :: e_tol, t_tol
         real(c_double), intent(out) :: energy, rms_sq
         integer(c_int), intent(out) :: converged
         integer(c_int) :: rc
      end function
      !> one pass of the loop body (reference src/ccsd.f90:340-360)
      function afesp_ccsd_iterate(ctx, e_tol, t_tol, energy, rms_sq, converged) bind(C, name='afesp_ccsd_iterate') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: ctx
         real(c_double), value :: e_tol, t_tol
         real(c_double), intent(out) :: energy, rms_sq
         integer(c_int), intent(out) :: converged
         integer(c_int) :: rc
      end function
      !> update_diis_cc (reference src/ccsd.f90:395)
      function afesp_ccsd_diis(ctx) bind(C, name='afesp_ccsd_diis') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int) :: rc
      end function
      function afesp_ccsd_get_amplitudes(ctx, t1, t2) bind(C, name='afesp_ccsd_get_amplitudes') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: ctx
         real(c_double), intent(out) :: t1(*), t2(*)
         integer(c_int) :: rc
      end function
      function afesp_ccsd_t_ntriples(nocc) bind(C, name='afesp_ccsd_t_ntriples') result(n)
         import :: c_int64_t
         integer(c_int64_t), value :: nocc
         integer(c_int64_t) :: n
      end function
      !> replaces `call do_ccsd_t_spatial(...)` (reference src/main.F90:112); out = E[T], E(T), D[T], D(T)
      function afesp_ccsd_t(ctx, t_begin, t_end, out) bind(C, name='afesp_ccsd_t') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: t_begin, t_end
         real(c_double), intent(out) :: out(4)
         integer(c_int) :: rc
      end function
      !> the same for the plain CCSD(T)/CCSD[T] types: out = E[T], E(T) (no y, no D sums -- reference src/ccsd.f90:2181-2185)
      function afesp_ccsd_t_plain(ctx, t_begin, t_end, out) bind(C, name='afesp_ccsd_t_plain') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: t_begin, t_end
         real(c_double), intent(out) :: out(2)
         integer(c_int) :: rc
      end function
      !> replaces build_cr_ccsd_t_intermediates (reference src/ccsd.f90:381)
      function afesp_ccsd_cr_intermediates(ctx) bind(C, name='afesp_ccsd_cr_intermediates') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int) :: rc
      end function
      !> (T) with the completely renormalised moment sums: out = E[T], E(T), D[T], D(T), sum t_bar.M3, sum (t_bar+z_bar).M3
      function afesp_ccsd_t_cr(ctx, t_begin, t_end, out) bind(C, name='afesp_ccsd_t_cr') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: t_begin, t_end
         real(c_double), intent(out) :: out(6)
         integer(c_int) :: rc
      end function
      !> replaces the two-body loop of read_integrals_in (reference src/integrals.f90:146-161); the packed AO integrals
      !> also stay on the device for afesp_ao2mo_mp2(..., eri_packed = c_null_ptr, ...)
      function afesp_read_eri_text(ctx, path, nbasis, eri_packed, nread) bind(C, name='afesp_read_eri_text') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr, c_char
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(in) :: path(*)
         integer(c_int64_t), value :: nbasis
         real(c_double), intent(out) :: eri_packed(*)
         integer(c_int64_t), intent(out) :: nread
         integer(c_int) :: rc
      end function
      !> replaces build_fock (reference src/hf.f90:349-385) on the packed AO integrals resident after afesp_read_eri_text
      function afesp_build_fock(ctx, nbasis, density, core_hamil, fock) bind(C, name='afesp_build_fock') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: nbasis
         real(c_double), intent(in) :: density(*), core_hamil(*)
         real(c_double), intent(out) :: fock(*)
         integer(c_int) :: rc
      end function
      !> replaces write_fcidump (reference src/mp2.f90:451-487)
      function afesp_write_fcidump(ctx, path, nbasis, nwritten) bind(C, name='afesp_write_fcidump') result(rc)
         import :: c_int, c_int64_t, c_ptr, c_char
         type(c_ptr), value :: ctx
         character(kind=c_char), intent(in) :: path(*)
         integer(c_int64_t), value :: nbasis
         integer(c_int64_t), intent(out) :: nwritten
         integer(c_int) :: rc
      end function
      !> replaces the integral/slice/init part of do_ccsd_spinorb (reference src/ccsd.f90:100-215); flags bit 0 = Stanton's
      !> index order for the tau~ term of F_mi (see include/afesp.h)
      function afesp_ccsd_so_init(ctx, nbasis, nel, eri_mo_packed, canon_levels, diis_n_errmat, flags) &
         bind(C, name='afesp_ccsd_so_init') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: nbasis, nel
         type(c_ptr), value :: eri_mo_packed
         real(c_double), intent(in) :: canon_levels(*)
         integer(c_int), value :: diis_n_errmat, flags
         integer(c_int) :: rc
      end function
      !> update_cc_energy, unrestricted branch (reference src/ccsd.f90:217)
      function afesp_ccsd_so_energy(ctx, e_tol, t_tol, energy, rms_sq, converged) bind(C, name='afesp_ccsd_so_energy') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: ctx
         real(c_double), value :: e_tol, t_tol
         real(c_double), intent(out) :: energy, rms_sq
         integer(c_int), intent(out) :: converged
         integer(c_int) :: rc
      end function
      !> build_tau, build_F, build_W, update_amplitudes, update_cc_energy (reference src/ccsd.f90:238-245)
      function afesp_ccsd_so_iterate(ctx, e_tol, t_tol, energy, rms_sq, converged) bind(C, name='afesp_ccsd_so_iterate') result(rc)
         import :: c_int, c_double, c_ptr
         type(c_ptr), value :: ctx
         real(c_double), value :: e_tol, t_tol
         real(c_double), intent(out) :: energy, rms_sq
         integer(c_int), intent(out) :: converged
         integer(c_int) :: rc
      end function
      function afesp_ccsd_so_diis(ctx) bind(C, name='afesp_ccsd_so_diis') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int) :: rc
      end function
      function afesp_ccsd_so_t_ntriples(nocc) bind(C, name='afesp_ccsd_so_t_ntriples') result(n)
         import :: c_int64_t
         integer(c_int64_t), value :: nocc
         integer(c_int64_t) :: n
      end function
      !> replaces `call do_ccsd_t_spinorb(...)` (reference src/main.F90:79): e_t = E_T of src/ccsd.f90:1910
      function afesp_ccsd_so_t(ctx, t_begin, t_end, e_t) bind(C, name='afesp_ccsd_so_t') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: t_begin, t_end
         real(c_double), intent(out) :: e_t
         integer(c_int) :: rc
      end function
      !> ---- multi-GPU: one process per GPU; the OpenMP reduction of the reference's (T) loop (src/ccsd.f90:2091) becomes a
      !> sum over ranks
      function afesp_device_count() bind(C, name='afesp_device_count') result(n)
         import :: c_int
         integer(c_int) :: n
      end function
      function afesp_comm_init(ctx, rank, world, transport, bootstrap_path, unique_id) bind(C, name='afesp_comm_init') result(rc)
         import :: c_int, c_ptr, c_char
         type(c_ptr), value :: ctx
         integer(c_int), value :: rank, world, transport
         character(kind=c_char), intent(in) :: bootstrap_path(*)
         type(c_ptr), value :: unique_id              ! c_null_ptr: the id travels through bootstrap_path
         integer(c_int) :: rc
      end function
      function afesp_comm_destroy(ctx) bind(C, name='afesp_comm_destroy') result(rc)
         import :: c_int, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int) :: rc
      end function
      function afesp_allreduce_sum(ctx, inout, n) bind(C, name='afesp_allreduce_sum') result(rc)
         import :: c_int, c_int64_t, c_double, c_ptr
         type(c_ptr), value :: ctx
         real(c_double), intent(inout) :: inout(*)
         integer(c_int64_t), value :: n
         integer(c_int) :: rc
      end function
      function afesp_ccsd_t_shard_bounds(ctx, nocc, nvirt, cr, world, bounds) bind(C, name='afesp_ccsd_t_shard_bounds') result(rc)
         import :: c_int, c_int64_t, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: nocc, nvirt
         integer(c_int), value :: cr, world
         integer(c_int64_t), intent(out) :: bounds(*)
         integer(c_int) :: rc
      end function
      function afesp_ccsd_t_block_size(ctx, nocc, nvirt, cr, block_size) bind(C, name='afesp_ccsd_t_block_size') result(rc)
         import :: c_int, c_int64_t, c_ptr
         type(c_ptr), value :: ctx
         integer(c_int64_t), value :: nocc, nvirt
         integer(c_int), value :: cr
         integer(c_int), intent(out) :: block_size
         integer(c_int) :: rc
      end function
   end interface

contains

   !> Copy the engine's last error message into a Fortran string.
   function afesp_error_text(ctx) result(text)
      type(c_ptr), intent(in) :: ctx
      character(len=:), allocatable :: text
      type(c_ptr) :: p
      character(kind=c_char), pointer :: chars(:)
      integer :: n
      p = afesp_last_error(ctx)
      text = ''
      if (.not. c_associated(p)) return
      call c_f_pointer(p, chars, [4096])
      n = 0
      do while (n < 4096)
         if (chars(n + 1) == c_null_char) exit
         n = n + 1
      end do
      allocate (character(len=n) :: text)
      block
         integer :: i
         do i = 1, n
            text(i:i) = chars(i)
         end do
      end block
   end function

end module afesp_capi
