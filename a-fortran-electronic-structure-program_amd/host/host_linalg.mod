﻿!mod$ v1 sum:107b15495db65b68
!need$ d25cf8cf498cc32f n host_support
module host_linalg
use host_support,only:dp
use host_support,only:i8
use host_support,only:out
use host_support,only:err
use host_support,only:fail
use host_support,only:seconds
contains
subroutine sym_eig(a_in,w,v)
real(8),intent(in)::a_in(:,:)
real(8),intent(out)::w(:)
real(8),intent(out)::v(:,:)
end
subroutine solve(a,b,info)
real(8),intent(inout)::a(:,:)
real(8),intent(inout)::b(:)
integer(4),intent(out)::info
end
end
