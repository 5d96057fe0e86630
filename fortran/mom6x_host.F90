!> Host side of the MI355X dynamical core for a Fortran caller, written against PLAIN Fortran arrays with MOM6's
!! symmetric-memory extents -- no MOM6 / FMS module is used, so this file compiles on its own (amdflang) and is
!! exercised on the GPU by fortran/drive_double_gyre.F90.  The shim modules that carry the reference's module and
!! procedure names (fortran/shims/*.F90; they `use MOM_grid` etc. and compile inside a MOM6 tree) are thin wrappers
!! around what is here:
!!   * mom6x_pack_plane    one metric array of ocean_grid_type (G%dxT, G%mask2dCu, ...) -> its plane of the pitched
!!                         metric block mom6x_ctx_create takes (replaces the `G` argument of every reference routine);
!!   * dyn_state_type      the prognostic arrays of step_MOM_dyn_split_RK2 (MOM_dynamics_split_RK2.F90:294-296) as
!!                         device arrays, with the bookkeeping of which side is current;
!!   * dyn_state_upload / dyn_state_download
!!                         host -> device at the start of a run (or after the host changed the state, e.g. a host-side
!!                         ALE step), device -> host ONLY where the host reads it next (save_restart, post_data, the
!!                         thermodynamics if it stays on the host) -- never once per step;
!!   * dyn_step            mom6x_step_dyn_split_RK2 on the resident state (the wind stress is the only per-step upload).
module mom6x_host
  use, intrinsic :: iso_c_binding
  use mom6x_c_api
  implicit none ; private

  public :: mom6x_pack_plane, dyn_state_type, dyn_state_init, dyn_state_upload, dyn_state_download, dyn_step, dyn_state_end
  public :: mom6x_message, stagger_extent

  !> Staggering codes of mom6x_upload / mom6x_download (include/mom6x.h)
  integer, parameter, public :: STG_H = 0, STG_U = 1, STG_V = 2, STG_Q = 3
  !> Metric planes (enum of include/mom6x.h, 0-based)
  integer, parameter, public :: G_mask2dT = 0, G_mask2dCu = 1, G_mask2dCv = 2, G_mask2dBu = 3, G_dxT = 4, G_dyT = 5, &
      G_IdxT = 6, G_IdyT = 7, G_dxCu = 8, G_dyCu = 9, G_IdxCu = 10, G_IdyCu = 11, G_dxCv = 12, G_dyCv = 13, G_IdxCv = 14, &
      G_IdyCv = 15, G_dxBu = 16, G_dyBu = 17, G_IdxBu = 18, G_IdyBu = 19, G_areaT = 20, G_IareaT = 21, G_areaBu = 22, &
      G_IareaBu = 23, G_areaCu = 24, G_areaCv = 25, G_IareaCu = 26, G_IareaCv = 27, G_dy_Cu = 28, G_dx_Cv = 29, G_bathyT = 30, &
      G_CoriolisBu = 31, G_Coriolis2Bu = 32, G_COUNT = 33

  !> The arrays step_MOM_dyn_split_RK2 works on, resident in HBM
  type :: dyn_state_type
    type(c_ptr) :: ctx = c_null_ptr
    type(c_ptr) :: u = c_null_ptr, v = c_null_ptr, h = c_null_ptr, uh = c_null_ptr, vh = c_null_ptr
    type(c_ptr) :: uhtr = c_null_ptr, vhtr = c_null_ptr, eta_av = c_null_ptr, taux = c_null_ptr, tauy = c_null_ptr
    integer :: nk = 0
    logical :: on_device = .false.    !< the device copy is the current one
    logical :: host_current = .true.  !< the host arrays equal the device copy
  end type dyn_state_type

contains

  !> Extents of a Fortran array with MOM6 symmetric memory: h (SZI_,SZJ_), u (SZIB_,SZJ_), v (SZI_,SZJB_), q (SZIB_,SZJB_)
  subroutine stagger_extent(d, stagger, nx, ny)
    type(mom6x_dims), intent(in) :: d ; integer, intent(in) :: stagger ; integer, intent(out) :: nx, ny
    nx = d%ni + 2*d%halo ; ny = d%nj + 2*d%halo
    if (stagger == STG_U .or. stagger == STG_Q) nx = nx + 1
    if (stagger == STG_V .or. stagger == STG_Q) ny = ny + 1
  end subroutine stagger_extent

  !> Copy one 2-d metric array (extents of its staggering) into plane `m` of the pitched metric block.
  !! block has G_COUNT*d%slab elements; flat(i,j) = (i+ioff) + (j+joff)*pitch with the computational domain at i,j = 0.
  subroutine mom6x_pack_plane(d, block, m, arr, stagger)
    type(mom6x_dims), intent(in)    :: d
    real(c_double),   intent(inout) :: block(0:)
    integer,          intent(in)    :: m, stagger
    real(c_double),   intent(in)    :: arr(:,:)
    integer :: nx, ny, i, j, i0, j0
    integer(c_size_t) :: base
    call stagger_extent(d, stagger, nx, ny)
    if (size(arr,1) /= nx .or. size(arr,2) /= ny) error stop "mom6x_pack_plane: array extents do not match the staggering"
    i0 = -d%halo ; j0 = -d%halo                       ! local index of arr(1,1)
    if (stagger == STG_U .or. stagger == STG_Q) i0 = i0 - 1
    if (stagger == STG_V .or. stagger == STG_Q) j0 = j0 - 1
    base = int(m, c_size_t) * int(d%slab, c_size_t)
    do j = 1, ny ; do i = 1, nx
      block(base + int((i - 1 + i0 + d%ioff) + (j - 1 + j0 + d%joff) * d%pitch, c_size_t)) = arr(i, j)
    enddo ; enddo
  end subroutine mom6x_pack_plane

  subroutine dyn_state_init(S, ctx, d)
    type(dyn_state_type), intent(inout) :: S ; type(c_ptr), intent(in) :: ctx ; type(mom6x_dims), intent(in) :: d
    integer(c_size_t) :: n2, n3 ; integer(c_int) :: rc
    S%ctx = ctx ; S%nk = d%nk
    n2 = int(d%slab, c_size_t) ; n3 = n2 * int(d%nk, c_size_t)
    rc = mom6x_dev_alloc(ctx, S%u, n3) ; call chk(rc) ; rc = mom6x_dev_alloc(ctx, S%v, n3) ; call chk(rc)
    rc = mom6x_dev_alloc(ctx, S%h, n3) ; call chk(rc) ; rc = mom6x_dev_alloc(ctx, S%uh, n3) ; call chk(rc)
    rc = mom6x_dev_alloc(ctx, S%vh, n3) ; call chk(rc) ; rc = mom6x_dev_alloc(ctx, S%uhtr, n3) ; call chk(rc)
    rc = mom6x_dev_alloc(ctx, S%vhtr, n3) ; call chk(rc) ; rc = mom6x_dev_alloc(ctx, S%eta_av, n2) ; call chk(rc)
    rc = mom6x_dev_alloc(ctx, S%taux, n2) ; call chk(rc) ; rc = mom6x_dev_alloc(ctx, S%tauy, n2) ; call chk(rc)
  end subroutine dyn_state_init

  !> Host arrays -> device (start of a run, or after the host changed the state)
  subroutine dyn_state_upload(S, u, v, h, uh, vh, uhtr, vhtr)
    type(dyn_state_type), intent(inout) :: S
    real(c_double), intent(in) :: u(*), v(*), h(*), uh(*), vh(*), uhtr(*), vhtr(*)
    integer(c_int) :: rc
    rc = mom6x_upload(S%ctx, S%u, u, STG_U, S%nk) ; call chk(rc) ; rc = mom6x_upload(S%ctx, S%v, v, STG_V, S%nk) ; call chk(rc)
    rc = mom6x_upload(S%ctx, S%h, h, STG_H, S%nk) ; call chk(rc)
    rc = mom6x_upload(S%ctx, S%uh, uh, STG_U, S%nk) ; call chk(rc) ; rc = mom6x_upload(S%ctx, S%vh, vh, STG_V, S%nk) ; call chk(rc)
    rc = mom6x_upload(S%ctx, S%uhtr, uhtr, STG_U, S%nk) ; call chk(rc)
    rc = mom6x_upload(S%ctx, S%vhtr, vhtr, STG_V, S%nk) ; call chk(rc)
    S%on_device = .true. ; S%host_current = .true.
  end subroutine dyn_state_upload

  !> Device -> host arrays, only if the host copy is stale (call where the host reads the state next)
  subroutine dyn_state_download(S, u, v, h, uh, vh, uhtr, vhtr, eta_av)
    type(dyn_state_type), intent(inout) :: S
    real(c_double), intent(inout) :: u(*), v(*), h(*), uh(*), vh(*), uhtr(*), vhtr(*), eta_av(*)
    integer(c_int) :: rc
    if (S%host_current) return
    rc = mom6x_download(S%ctx, u, S%u, STG_U, S%nk) ; call chk(rc) ; rc = mom6x_download(S%ctx, v, S%v, STG_V, S%nk) ; call chk(rc)
    rc = mom6x_download(S%ctx, h, S%h, STG_H, S%nk) ; call chk(rc)
    rc = mom6x_download(S%ctx, uh, S%uh, STG_U, S%nk) ; call chk(rc) ; rc = mom6x_download(S%ctx, vh, S%vh, STG_V, S%nk) ; call chk(rc)
    rc = mom6x_download(S%ctx, uhtr, S%uhtr, STG_U, S%nk) ; call chk(rc)
    rc = mom6x_download(S%ctx, vhtr, S%vhtr, STG_V, S%nk) ; call chk(rc)
    rc = mom6x_download(S%ctx, eta_av, S%eta_av, STG_H, 1) ; call chk(rc)
    S%host_current = .true.
  end subroutine dyn_state_download

  !> One baroclinic step on the resident state (step_MOM_dyn_split_RK2, RK2.F90:294); forces%taux, forces%tauy are uploaded.
  subroutine dyn_step(S, taux, tauy, dt, calc_dtbt)
    type(dyn_state_type), intent(inout) :: S
    real(c_double), intent(in) :: taux(*), tauy(*) ; real(c_double), intent(in) :: dt ; logical, intent(in) :: calc_dtbt
    integer(c_int) :: rc
    if (.not.S%on_device) error stop "dyn_step: upload the state first"
    rc = mom6x_upload(S%ctx, S%taux, taux, STG_U, 1) ; call chk(rc) ; rc = mom6x_upload(S%ctx, S%tauy, tauy, STG_V, 1) ; call chk(rc)
    rc = mom6x_step_dyn_split_RK2(S%ctx, S%u, S%v, S%h, S%uh, S%vh, S%uhtr, S%vhtr, S%eta_av, S%taux, S%tauy, dt, &
                                  merge(1_c_int, 0_c_int, calc_dtbt), c_null_ptr)
    call chk(rc)
    S%host_current = .false.
  end subroutine dyn_step

  subroutine dyn_state_end(S)
    type(dyn_state_type), intent(inout) :: S
    integer(c_int) :: rc
    rc = mom6x_dev_free(S%ctx, S%u) ; rc = mom6x_dev_free(S%ctx, S%v) ; rc = mom6x_dev_free(S%ctx, S%h)
    rc = mom6x_dev_free(S%ctx, S%uh) ; rc = mom6x_dev_free(S%ctx, S%vh) ; rc = mom6x_dev_free(S%ctx, S%uhtr)
    rc = mom6x_dev_free(S%ctx, S%vhtr) ; rc = mom6x_dev_free(S%ctx, S%eta_av) ; rc = mom6x_dev_free(S%ctx, S%taux)
    rc = mom6x_dev_free(S%ctx, S%tauy)
    S%on_device = .false.
  end subroutine dyn_state_end

  !> mom6x_last_error() as a Fortran string (what a shim hands to MOM_error(FATAL, ...))
  function mom6x_message() result(msg)
    character(len=480) :: msg
    character(kind=c_char), pointer :: p(:)
    integer :: n
    msg = "" ; call c_f_pointer(mom6x_last_error(), p, [480])
    do n = 1, 480 ; if (p(n) == c_null_char) exit ; msg(n:n) = p(n) ; enddo
  end function mom6x_message

  subroutine chk(rc)
    integer(c_int), intent(in) :: rc
    if (rc /= 0) then
      print '(a)', "mom6x error: "//trim(mom6x_message())
      error stop 2
    endif
  end subroutine chk
end module mom6x_host
