﻿!mod$ v1 sum:d25cf8cf498cc32f
!need$ f1de5abe9bfe2168 i iso_fortran_env
module host_support
use,intrinsic::iso_fortran_env,only:dp=>real64
use,intrinsic::iso_fortran_env,only:i8=>int64
use,intrinsic::iso_fortran_env,only:out=>output_unit
use,intrinsic::iso_fortran_env,only:err=>error_unit
contains
subroutine fail(where,why)
character(*,1),intent(in)::where
character(*,1),intent(in)::why
end
function seconds() result(t)
real(8)::t
end
end
