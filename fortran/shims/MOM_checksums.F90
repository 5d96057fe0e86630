!> Drop-in for the checksum routines of src/framework/MOM_checksums.F90 that the dynamical core's debugging path and the restart
!! files use: hchksum :387/:1413, uchksum :1005/:1782, vchksum :1209/:1986, Bchksum = qchksum :688/:1586, the pairs hchksum_pair
!! :268/:326, uvchksum :879/:942, Bchksum_pair :558/:625 (each 2-d and 3-d, as the reference's generic interfaces) and MOM_checksums_init :2653 -- same names and
!! argument lists, served by mom6x_chksum (SURVEY 8f-4): the statistics (reproducing mean, min, max) and the
!! bit counts of the shifted domains are formed on the device from the array where it lives (the resident copy, or an upload), and
!! the two lines are written here with the reference's formats (chk_sum_msg1/5/_NSEW/_W/_S :2563-2640).
!! Not carried: rotated grids (HI%turns /= 0), chksum0 / zchksum / the 1-d..3-d `chksum` of unstaggered arrays (host arithmetic on
!! small arrays: keep the reference's), and field_checksum (:2440-2550): MOM_restart hands it the computational-domain SECTION of a field, a
!! temporary the device has never seen -- a host whose fields are resident takes the `checksum` attribute from mom6x_field_chksum
!! instead (mom6_amd/restart.py does; INTEGRATION.md section 4).
module MOM_checksums
use, intrinsic :: iso_c_binding
use, intrinsic :: iso_fortran_env, only : error_unit
use mom6x_c_api
use mom6x_host
use mom6x_shim_ctx
use MOM_coms,          only : PE_here, root_PE
use MOM_error_handler, only : MOM_error, FATAL
use MOM_file_parser,   only : log_version, param_file_type
use MOM_hor_index,     only : hor_index_type
implicit none ; private

public :: hchksum, uchksum, vchksum, Bchksum, qchksum, hchksum_pair, uvchksum, Bchksum_pair, MOM_checksums_init

interface hchksum ; module procedure chksum_h_2d, chksum_h_3d ; end interface
interface uchksum ; module procedure chksum_u_2d, chksum_u_3d ; end interface
interface vchksum ; module procedure chksum_v_2d, chksum_v_3d ; end interface
interface Bchksum ; module procedure chksum_B_2d, chksum_B_3d ; end interface
interface qchksum ; module procedure chksum_B_2d, chksum_B_3d ; end interface
interface hchksum_pair ; module procedure chksum_pair_h_2d, chksum_pair_h_3d ; end interface
interface uvchksum ; module procedure chksum_uv_2d, chksum_uv_3d ; end interface
interface Bchksum_pair ; module procedure chksum_pair_B_2d, chksum_pair_B_3d ; end interface

integer, parameter :: CHK_CORNERS = 1, CHK_NSEW = 2, CHK_W = 3, CHK_S = 4      !< enum mom6x_chksum_kind

contains

!> One array: mom6x_chksum on its device copy, then the two lines of the reference
subroutine chk(a, stg, nlev, rank, pt, mesg, haloshift, symmetric, omit_corners, scale, logunit, unscale)
  real(c_double), target, intent(in) :: a(*)
  integer,          intent(in) :: stg, nlev, rank
  character(len=*), intent(in) :: pt, mesg
  integer, optional, intent(in) :: haloshift, logunit
  logical, optional, intent(in) :: symmetric, omit_corners
  real,    optional, intent(in) :: scale, unscale
  type(mom6x_chksum_result) :: r
  real(c_double), target :: scaling
  type(c_ptr) :: p_scale
  integer(c_int) :: rc
  integer :: hshift, iounit
  logical :: sym, omit
  hshift = 0 ; if (present(haloshift)) hshift = haloshift
  sym = .false. ; if (present(symmetric)) sym = symmetric
  omit = .false. ; if (present(omit_corners)) omit = omit_corners
  p_scale = c_null_ptr
  if (present(unscale)) then ; scaling = unscale ; p_scale = c_loc(scaling)      ! (unscale takes precedence :1424-1426)
  elseif (present(scale)) then ; scaling = scale ; p_scale = c_loc(scaling) ; endif
  iounit = error_unit ; if (present(logunit)) iounit = logunit
  rc = mom6x_chksum(shim_ctx_current(), shim_up3(36, a, stg, nlev), int(nlev, c_int), int(rank, c_int), int(stg, c_int), &
                    int(hshift, c_int), merge(1_c_int, 0_c_int, sym), merge(1_c_int, 0_c_int, omit), p_scale, r)
  call shim_check(rc, "chksum: "//trim(mesg))      ! (a NaN in the computational domain: 'NaN detected: <mesg>', as :1457)
  if (PE_here() /= root_PE()) return
  write(iounit, '(A,3(A,ES25.16,1X),A)') pt, " mean=", r%mean, "min=", r%amin, "max=", r%amax, trim(mesg)      ! chk_sum_msg3
  select case (r%bc_kind)
    case (CHK_CORNERS) ; write(iounit, '(A,5(A,I10,1X),A)') pt, " c=", r%bc0, "sw=", r%bc(1), "se=", r%bc(2), "nw=", r%bc(3), "ne=", r%bc(4), trim(mesg)
    case (CHK_NSEW)    ; write(iounit, '(A,5(A,I10,1X),A)') pt, " c=", r%bc0, "N=", r%bc(1), "S=", r%bc(2), "E=", r%bc(3), "W=", r%bc(4), trim(mesg)
    case (CHK_W)       ; write(iounit, '(A,2(A,I10,1X),A)') pt, " c=", r%bc0, "W=", r%bc(1), trim(mesg)
    case (CHK_S)       ; write(iounit, '(A,2(A,I10,1X),A)') pt, " c=", r%bc0, "S=", r%bc(1), trim(mesg)
    case default       ; write(iounit, '(a,1(a,i10,1x),a)') pt, " c=", r%bc0, trim(mesg)
  end select
end subroutine chk

subroutine no_turns(HI)
  type(hor_index_type), intent(in) :: HI
  if (.not.HI%symmetric) call MOM_error(FATAL, "MOM_checksums: non-symmetric memory is not carried by the MI355X path.")
end subroutine no_turns

subroutine chksum_h_2d(array_m, mesg, HI_m, haloshift, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%isd:,HI_m%jsd:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_H, 1, 2, "h-point:", mesg, haloshift=haloshift, omit_corners=omit_corners, scale=scale, logunit=logunit, unscale=unscale)
end subroutine chksum_h_2d

subroutine chksum_h_3d(array_m, mesg, HI_m, haloshift, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%isd:,HI_m%jsd:,:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_H, size(array_m, 3), 3, "h-point:", mesg, haloshift=haloshift, omit_corners=omit_corners, scale=scale, logunit=logunit, unscale=unscale)
end subroutine chksum_h_3d

subroutine chksum_u_2d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%IsdB:,HI_m%jsd:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_U, 1, 2, "u-point:", mesg, haloshift, symmetric, omit_corners, scale, logunit, unscale)
end subroutine chksum_u_2d

subroutine chksum_u_3d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%IsdB:,HI_m%jsd:,:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_U, size(array_m, 3), 3, "u-point:", mesg, haloshift, symmetric, omit_corners, scale, logunit, unscale)
end subroutine chksum_u_3d

subroutine chksum_v_2d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%isd:,HI_m%JsdB:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_V, 1, 2, "v-point:", mesg, haloshift, symmetric, omit_corners, scale, logunit, unscale)
end subroutine chksum_v_2d

subroutine chksum_v_3d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%isd:,HI_m%JsdB:,:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_V, size(array_m, 3), 3, "v-point:", mesg, haloshift, symmetric, omit_corners, scale, logunit, unscale)
end subroutine chksum_v_3d

subroutine chksum_B_2d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%IsdB:,HI_m%JsdB:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_Q, 1, 2, "B-point:", mesg, haloshift, symmetric, omit_corners, scale, logunit, unscale)
end subroutine chksum_B_2d

subroutine chksum_B_3d(array_m, mesg, HI_m, haloshift, symmetric, omit_corners, scale, logunit, unscale)
  type(hor_index_type), target, intent(in) :: HI_m
  real, dimension(HI_m%IsdB:,HI_m%JsdB:,:), target, intent(in) :: array_m
  character(len=*), intent(in) :: mesg
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; real, optional, intent(in) :: unscale
  call no_turns(HI_m)
  call chk(array_m, STG_Q, size(array_m, 3), 3, "B-point:", mesg, haloshift, symmetric, omit_corners, scale, logunit, unscale)
end subroutine chksum_B_3d

! ---- the pairs: two calls with the reference's message prefixes (:313-320, :927-936, :605-616); without haloshift the reference
!      passes no omit_corners either; scalar_pair only matters on a rotated grid
subroutine chksum_pair_h_2d(mesg, arrayA, arrayB, HI, haloshift, omit_corners, scale, logunit, scalar_pair, unscale)
  character(len=*), intent(in) :: mesg ; type(hor_index_type), target, intent(in) :: HI
  real, dimension(HI%isd:,HI%jsd:), target, intent(in) :: arrayA, arrayB
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; logical, optional, intent(in) :: scalar_pair
  real, optional, intent(in) :: unscale
  if (present(haloshift)) then
    call chksum_h_2d(arrayA, 'x '//mesg, HI, haloshift, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_h_2d(arrayB, 'y '//mesg, HI, haloshift, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
  else
    call chksum_h_2d(arrayA, 'x '//mesg, HI, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_h_2d(arrayB, 'y '//mesg, HI, scale=scale, logunit=logunit, unscale=unscale)
  endif
end subroutine chksum_pair_h_2d

subroutine chksum_pair_h_3d(mesg, arrayA, arrayB, HI, haloshift, omit_corners, scale, logunit, scalar_pair, unscale)
  character(len=*), intent(in) :: mesg ; type(hor_index_type), target, intent(in) :: HI
  real, dimension(HI%isd:,HI%jsd:,:), target, intent(in) :: arrayA, arrayB
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; logical, optional, intent(in) :: scalar_pair
  real, optional, intent(in) :: unscale
  if (present(haloshift)) then
    call chksum_h_3d(arrayA, 'x '//mesg, HI, haloshift, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_h_3d(arrayB, 'y '//mesg, HI, haloshift, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
  else
    call chksum_h_3d(arrayA, 'x '//mesg, HI, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_h_3d(arrayB, 'y '//mesg, HI, scale=scale, logunit=logunit, unscale=unscale)
  endif
end subroutine chksum_pair_h_3d

subroutine chksum_uv_2d(mesg, arrayU, arrayV, HI, haloshift, symmetric, omit_corners, scale, logunit, scalar_pair, unscale)
  character(len=*), intent(in) :: mesg ; type(hor_index_type), target, intent(in) :: HI
  real, dimension(HI%IsdB:,HI%jsd:), target, intent(in) :: arrayU
  real, dimension(HI%isd:,HI%JsdB:), target, intent(in) :: arrayV
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; logical, optional, intent(in) :: scalar_pair
  real, optional, intent(in) :: unscale
  if (present(haloshift)) then
    call chksum_u_2d(arrayU, 'u '//mesg, HI, haloshift, symmetric, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_v_2d(arrayV, 'v '//mesg, HI, haloshift, symmetric, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
  else
    call chksum_u_2d(arrayU, 'u '//mesg, HI, symmetric=symmetric, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_v_2d(arrayV, 'v '//mesg, HI, symmetric=symmetric, scale=scale, logunit=logunit, unscale=unscale)
  endif
end subroutine chksum_uv_2d

subroutine chksum_uv_3d(mesg, arrayU, arrayV, HI, haloshift, symmetric, omit_corners, scale, logunit, scalar_pair, unscale)
  character(len=*), intent(in) :: mesg ; type(hor_index_type), target, intent(in) :: HI
  real, dimension(HI%IsdB:,HI%jsd:,:), target, intent(in) :: arrayU
  real, dimension(HI%isd:,HI%JsdB:,:), target, intent(in) :: arrayV
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; logical, optional, intent(in) :: scalar_pair
  real, optional, intent(in) :: unscale
  if (present(haloshift)) then
    call chksum_u_3d(arrayU, 'u '//mesg, HI, haloshift, symmetric, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_v_3d(arrayV, 'v '//mesg, HI, haloshift, symmetric, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
  else
    call chksum_u_3d(arrayU, 'u '//mesg, HI, symmetric=symmetric, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_v_3d(arrayV, 'v '//mesg, HI, symmetric=symmetric, scale=scale, logunit=logunit, unscale=unscale)
  endif
end subroutine chksum_uv_3d

subroutine chksum_pair_B_2d(mesg, arrayA, arrayB, HI, haloshift, symmetric, omit_corners, scale, logunit, scalar_pair, unscale)
  character(len=*), intent(in) :: mesg ; type(hor_index_type), target, intent(in) :: HI
  real, dimension(HI%IsdB:,HI%JsdB:), target, intent(in) :: arrayA, arrayB
  logical, optional, intent(in) :: symmetric ; integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; logical, optional, intent(in) :: scalar_pair
  real, optional, intent(in) :: unscale
  logical :: sym
  sym = .false. ; if (present(symmetric)) sym = symmetric
  if (present(haloshift)) then
    call chksum_B_2d(arrayA, 'x '//mesg, HI, haloshift, symmetric=sym, omit_corners=omit_corners, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_B_2d(arrayB, 'y '//mesg, HI, haloshift, symmetric=sym, omit_corners=omit_corners, scale=scale, logunit=logunit, unscale=unscale)
  else
    call chksum_B_2d(arrayA, 'x '//mesg, HI, symmetric=sym, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_B_2d(arrayB, 'y '//mesg, HI, symmetric=sym, scale=scale, logunit=logunit, unscale=unscale)
  endif
end subroutine chksum_pair_B_2d

subroutine chksum_pair_B_3d(mesg, arrayA, arrayB, HI, haloshift, symmetric, omit_corners, scale, logunit, scalar_pair, unscale)
  character(len=*), intent(in) :: mesg ; type(hor_index_type), target, intent(in) :: HI
  real, dimension(HI%IsdB:,HI%JsdB:,:), target, intent(in) :: arrayA, arrayB
  integer, optional, intent(in) :: haloshift ; logical, optional, intent(in) :: symmetric, omit_corners
  real, optional, intent(in) :: scale ; integer, optional, intent(in) :: logunit ; logical, optional, intent(in) :: scalar_pair
  real, optional, intent(in) :: unscale
  if (present(haloshift)) then
    call chksum_B_3d(arrayA, 'x '//mesg, HI, haloshift, symmetric, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_B_3d(arrayB, 'y '//mesg, HI, haloshift, symmetric, omit_corners, scale=scale, logunit=logunit, unscale=unscale)
  else
    call chksum_B_3d(arrayA, 'x '//mesg, HI, symmetric=symmetric, scale=scale, logunit=logunit, unscale=unscale)
    call chksum_B_3d(arrayB, 'y '//mesg, HI, symmetric=symmetric, scale=scale, logunit=logunit, unscale=unscale)
  endif
end subroutine chksum_pair_B_3d

!> MOM_checksums_init (:2653)
subroutine MOM_checksums_init(param_file)
  type(param_file_type), intent(in) :: param_file
  call log_version(param_file, "MOM_checksums", "mom6x", "")
end subroutine MOM_checksums_init

end module MOM_checksums
