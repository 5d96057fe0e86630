!> mom6x_shim_ctx -- the one device context (tile) of this process, shared by every shim module.
!!
!! MOM6 runs one tile per PE; the MI355X build runs one PE per GPU, so a process has exactly one mom6x_ctx.  The first shim
!! module whose *_init is called (normally MOM_dynamics_split_RK2 -> continuity_init) creates it from G, GV through
!! shim_ctx(); the others find it there.  Creation does what MOM_domains_init + MOM_grid_init do for the host:
!!   * the layout of the tile in HBM (mom6x_dims_init) and its place in the global domain (G%idg_offset, G%Domain%niglobal);
!!   * the metric block (one plane per array of ocean_grid_type, mom6x_pack_plane);
!!   * with more than one PE: the device is (PE mod devices on the node), the RCCL communicator of the LAYOUT is formed
!!     from a unique id made on the root PE and broadcast with MOM_coms' own broadcast, and mom6x_comm_init attaches the
!!     8-neighbour halo plan (pass_var / pass_vector of the reference become packed ncclSend / ncclRecv groups).
!! The module also owns a small pool of scratch device arrays for the shims that serve HOST callers (upload, compute,
!! download), so that repeated calls do not allocate -- and a registry of RESIDENT host arrays: an array the host has handed
!! over with shim_resident_add lives in HBM from then on; every shim that is given that array (recognised by its address)
!! works on the device copy directly, without an upload before and a download after the call, until the host asks for the
!! values (shim_resident_sync_host) or takes the array back (shim_resident_drop).  shim_transfer_count counts the arrays that
!! did cross PCIe.
module mom6x_shim_ctx
use, intrinsic :: iso_c_binding
use mom6x_c_api
use mom6x_host
use MOM_coms,          only : PE_here, root_PE, num_PEs, broadcast
use MOM_error_handler, only : MOM_error, FATAL, WARNING
use MOM_file_parser,   only : get_param, param_file_type
use MOM_grid,          only : ocean_grid_type
use MOM_verticalGrid,  only : verticalGrid_type
implicit none ; private

public :: shim_ctx, shim_ctx_is_up, shim_ctx_current, shim_ctx_end, shim_dims, shim_check, shim_buf, shim_nk, shim_set_domain_flags
public :: shim_up2, shim_up3, shim_down2, shim_down3, shim_out2, shim_out3
public :: shim_resident_add, shim_resident_drop, shim_resident_sync_host, shim_resident_host_changed, shim_resident_dev
public :: shim_transfer_count

type(c_ptr), save :: the_ctx = c_null_ptr
type(mom6x_dims), save :: the_dims
logical, save :: reentrant(2) = (/ .false., .false. /), flags_known = .false.
integer, parameter :: NBUF = 40
type(c_ptr), save :: bufs(NBUF) = c_null_ptr       !< scratch device arrays, by slot
integer(c_size_t), save :: buf_len(NBUF) = 0
!> The resident host arrays: the address of the host array, its copy in HBM, its staggering and number of levels
integer, parameter :: NRES = 64
integer(c_intptr_t), save :: res_addr(NRES) = 0
type(c_ptr), save :: res_dev(NRES) = c_null_ptr
integer, save :: res_stg(NRES) = 0, res_nlev(NRES) = 0, n_res = 0
integer(c_long_long), save :: n_transfers = 0      !< host <-> device array transfers made by the shims (uploads + downloads)

contains

!> REENTRANT_X / REENTRANT_Y as MOM_domains_init reads them (MOM_domains.F90): ocean_grid_type does not carry them in a
!! form a module outside the infra layer can read, so the first *_init that has the param_file hands them over.
subroutine shim_set_domain_flags(param_file)
  type(param_file_type), intent(in) :: param_file
  logical :: tripolar
  if (flags_known) return
  call get_param(param_file, "MOM_domains", "REENTRANT_X", reentrant(1), default=.true., do_not_log=.true.)
  call get_param(param_file, "MOM_domains", "REENTRANT_Y", reentrant(2), default=.false., do_not_log=.true.)
  call get_param(param_file, "MOM_domains", "TRIPOLAR_N", tripolar, default=.false., do_not_log=.true.)
  if (tripolar) call MOM_error(FATAL, "mom6x_shim_ctx: TRIPOLAR_N (a folded northern edge) is not carried by the MI355X path.")
  flags_known = .true.
end subroutine shim_set_domain_flags

logical function shim_ctx_is_up()
  shim_ctx_is_up = c_associated(the_ctx)
end function shim_ctx_is_up

!> The context, for a module whose entry points see no ocean_grid_type (MOM_checksums): it must exist already
function shim_ctx_current() result(ctx)
  type(c_ptr) :: ctx
  if (.not.c_associated(the_ctx)) call MOM_error(FATAL, "mom6x_shim_ctx: no device context yet (call a shim's *_init first).")
  ctx = the_ctx
end function shim_ctx_current

!> The context of this PE's tile; created on the first call.
function shim_ctx(G, GV) result(ctx)
  type(ocean_grid_type),   intent(in) :: G
  type(verticalGrid_type), intent(in) :: GV
  type(c_ptr) :: ctx
  type(mom6x_vgrid) :: gvx
  real(c_double), allocatable :: block(:)
  integer(c_int) :: rc, device, ndev
  integer :: halo

  if (c_associated(the_ctx)) then ; ctx = the_ctx ; return ; endif
  if (mom6x_abi_version() /= MOM6X_ABI_BUILT_FOR) call MOM_error(FATAL, &
      "mom6x_shim_ctx: libmom6x.so and fortran/mom6x_c_api.F90 were built for different versions of include/mom6x.h.")
  if (.not.flags_known) call MOM_error(FATAL, "mom6x_shim_ctx: shim_set_domain_flags must be called before the first shim_ctx.")
  if (.not.G%symmetric) call MOM_error(FATAL, "mom6x_shim_ctx: the MI355X path needs a SYMMETRIC_MEMORY_ build.")
  halo = G%isc - G%isd
  if (halo /= G%jsc - G%jsd) call MOM_error(FATAL, "mom6x_shim_ctx: NIHALO and NJHALO must be equal.")
  rc = mom6x_dims_init(the_dims, int(G%iec-G%isc+1, c_int), int(G%jec-G%jsc+1, c_int), int(GV%ke, c_int), int(halo, c_int))
  call shim_check(rc, "mom6x_dims_init")
  the_dims%i_glob0 = G%idg_offset ; the_dims%j_glob0 = G%jdg_offset
  the_dims%ni_glob = G%Domain%niglobal ; the_dims%nj_glob = G%Domain%njglobal
  the_dims%reentrant_x = merge(1_c_int, 0_c_int, reentrant(1)) ; the_dims%reentrant_y = merge(1_c_int, 0_c_int, reentrant(2))

  allocate(block(0:int(G_COUNT, c_size_t)*int(the_dims%slab, c_size_t)-1), source=0.0_c_double)
  call mom6x_pack_plane(the_dims, block, G_mask2dT, G%mask2dT, STG_H) ; call mom6x_pack_plane(the_dims, block, G_mask2dCu, G%mask2dCu, STG_U)
  call mom6x_pack_plane(the_dims, block, G_mask2dCv, G%mask2dCv, STG_V) ; call mom6x_pack_plane(the_dims, block, G_mask2dBu, G%mask2dBu, STG_Q)
  call mom6x_pack_plane(the_dims, block, G_dxT, G%dxT, STG_H)   ; call mom6x_pack_plane(the_dims, block, G_dyT, G%dyT, STG_H)
  call mom6x_pack_plane(the_dims, block, G_IdxT, G%IdxT, STG_H) ; call mom6x_pack_plane(the_dims, block, G_IdyT, G%IdyT, STG_H)
  call mom6x_pack_plane(the_dims, block, G_dxCu, G%dxCu, STG_U) ; call mom6x_pack_plane(the_dims, block, G_dyCu, G%dyCu, STG_U)
  call mom6x_pack_plane(the_dims, block, G_IdxCu, G%IdxCu, STG_U) ; call mom6x_pack_plane(the_dims, block, G_IdyCu, G%IdyCu, STG_U)
  call mom6x_pack_plane(the_dims, block, G_dxCv, G%dxCv, STG_V) ; call mom6x_pack_plane(the_dims, block, G_dyCv, G%dyCv, STG_V)
  call mom6x_pack_plane(the_dims, block, G_IdxCv, G%IdxCv, STG_V) ; call mom6x_pack_plane(the_dims, block, G_IdyCv, G%IdyCv, STG_V)
  call mom6x_pack_plane(the_dims, block, G_dxBu, G%dxBu, STG_Q) ; call mom6x_pack_plane(the_dims, block, G_dyBu, G%dyBu, STG_Q)
  call mom6x_pack_plane(the_dims, block, G_IdxBu, G%IdxBu, STG_Q) ; call mom6x_pack_plane(the_dims, block, G_IdyBu, G%IdyBu, STG_Q)
  call mom6x_pack_plane(the_dims, block, G_areaT, G%areaT, STG_H) ; call mom6x_pack_plane(the_dims, block, G_IareaT, G%IareaT, STG_H)
  call mom6x_pack_plane(the_dims, block, G_areaBu, G%areaBu, STG_Q) ; call mom6x_pack_plane(the_dims, block, G_IareaBu, G%IareaBu, STG_Q)
  call mom6x_pack_plane(the_dims, block, G_areaCu, G%areaCu, STG_U) ; call mom6x_pack_plane(the_dims, block, G_areaCv, G%areaCv, STG_V)
  call mom6x_pack_plane(the_dims, block, G_IareaCu, G%IareaCu, STG_U) ; call mom6x_pack_plane(the_dims, block, G_IareaCv, G%IareaCv, STG_V)
  call mom6x_pack_plane(the_dims, block, G_dy_Cu, G%dy_Cu, STG_U) ; call mom6x_pack_plane(the_dims, block, G_dx_Cv, G%dx_Cv, STG_V)
  call mom6x_pack_plane(the_dims, block, G_bathyT, G%bathyT, STG_H)
  call mom6x_pack_plane(the_dims, block, G_CoriolisBu, G%CoriolisBu, STG_Q) ; call mom6x_pack_plane(the_dims, block, G_Coriolis2Bu, G%Coriolis2Bu, STG_Q)

  if (.not.GV%Boussinesq) call MOM_error(FATAL, "mom6x_shim_ctx: the non-Boussinesq mode is not carried by the MI355X path.")
  gvx%g_Earth = GV%g_Earth ; gvx%Rho0 = GV%Rho0 ; gvx%Angstrom_H = GV%Angstrom_H ; gvx%H_subroundoff = GV%H_subroundoff
  gvx%dZ_subroundoff = GV%dZ_subroundoff ; gvx%H_to_Z = GV%H_to_Z ; gvx%Z_to_H = GV%Z_to_H ; gvx%H_to_RZ = GV%H_to_RZ
  gvx%RZ_to_H = GV%RZ_to_H ; gvx%Boussinesq = 1_c_int

  ! One PE per GPU: the PEs of a node take the node's devices in turn.
  ndev = mom6x_device_count()
  if (ndev <= 0) call MOM_error(FATAL, "mom6x_shim_ctx: no HIP device is visible ("//trim(mom6x_message())//").")
  device = int(mod(PE_here() - root_PE(), int(ndev)), c_int)
  rc = mom6x_ctx_create(the_ctx, the_dims, device, block, gvx, int(G%first_direction, c_int))
  call shim_check(rc, "mom6x_ctx_create")
  deallocate(block)
  if (num_PEs() > 1) call attach_layout(G)
  ctx = the_ctx
end function shim_ctx

!> More than one PE: the RCCL communicator over the PEs of the LAYOUT and this tile's halo plan.
subroutine attach_layout(G)
  type(ocean_grid_type), intent(in) :: G
  character(kind=c_char) :: id(128)
  integer :: idi(128), npx, npy, px, py, rank, n
  integer(c_int) :: rc
  npx = G%Domain%layout(1) ; npy = G%Domain%layout(2)
  if (npx * npy /= num_PEs()) call MOM_error(FATAL, "mom6x_shim_ctx: masked (land-only) PEs of the LAYOUT are not carried: "//&
      "LAYOUT(1)*LAYOUT(2) must equal the number of PEs.")
  ! The PE list of an FMS 2-d decomposition runs along i first (mpp_define_layout / mpp_define_domains).
  rank = PE_here() - root_PE() ; px = mod(rank, npx) ; py = rank / npx
  if ((px == 0) .neqv. (G%idg_offset == 0)) call MOM_error(FATAL, "mom6x_shim_ctx: the PE order of the LAYOUT is not i-fastest.")
  if ((py == 0) .neqv. (G%jdg_offset == 0)) call MOM_error(FATAL, "mom6x_shim_ctx: the PE order of the LAYOUT is not i-fastest.")
  idi(:) = 0
  if (PE_here() == root_PE()) then
    rc = mom6x_comm_unique_id(id) ; call shim_check(rc, "mom6x_comm_unique_id")
    do n = 1, 128 ; idi(n) = ichar(id(n)) ; enddo
  endif
  call broadcast(idi, 128, root_PE())
  do n = 1, 128 ; id(n) = char(idi(n), kind=c_char) ; enddo
  rc = mom6x_comm_init(the_ctx, int(npx, c_int), int(npy, c_int), int(px, c_int), int(py, c_int), id, 0_c_int)
  call shim_check(rc, "mom6x_comm_init")
end subroutine attach_layout

function shim_dims() result(d)
  type(mom6x_dims) :: d
  d = the_dims
end function shim_dims

integer(c_int) function shim_nk()
  shim_nk = the_dims%nk
end function shim_nk

!> A nonzero return code of the library is the reference's MOM_error(FATAL, ...) with the library's message.
subroutine shim_check(rc, where)
  integer(c_int), intent(in) :: rc ; character(len=*), intent(in) :: where
  if (rc /= 0) call MOM_error(FATAL, trim(where)//": "//trim(mom6x_message()))
end subroutine shim_check

!> Scratch device array number `slot` with room for nlev planes (allocated on first use, grown when needed).
function shim_buf(slot, nlev) result(p)
  integer, intent(in) :: slot, nlev
  type(c_ptr) :: p
  integer(c_size_t) :: n
  integer(c_int) :: rc
  if (slot < 1 .or. slot > NBUF) call MOM_error(FATAL, "mom6x_shim_ctx: scratch slot out of range.")
  n = int(the_dims%slab, c_size_t) * int(max(nlev, 1), c_size_t)
  if (c_associated(bufs(slot)) .and. buf_len(slot) < n) then
    rc = mom6x_dev_free(the_ctx, bufs(slot)) ; bufs(slot) = c_null_ptr
  endif
  if (.not.c_associated(bufs(slot))) then
    rc = mom6x_dev_alloc(the_ctx, bufs(slot), n) ; call shim_check(rc, "mom6x_dev_alloc") ; buf_len(slot) = n
  endif
  p = bufs(slot)
end function shim_buf

!> Where the resident array with the address of `a` is in the registry (0: it is not resident)
integer function res_find(a)
  real(c_double), target, intent(in) :: a(*)
  integer(c_intptr_t) :: addr ; integer :: n
  res_find = 0
  if (n_res == 0) return
  addr = transfer(c_loc(a), addr)
  do n = 1, n_res ; if (res_addr(n) == addr) then ; res_find = n ; return ; endif ; enddo
end function res_find

!> Hand a host array over to the device: uploaded once, every shim then works on the copy in HBM (no transfer per call).
subroutine shim_resident_add(a, stagger, nlev)
  real(c_double), target, intent(in) :: a(*) ; integer, intent(in) :: stagger, nlev
  integer(c_int) :: rc ; integer :: n
  if (.not.c_associated(the_ctx)) call MOM_error(FATAL, "shim_resident_add: no device context yet (call a shim's *_init first).")
  n = res_find(a)
  if (n == 0) then
    if (n_res >= NRES) call MOM_error(FATAL, "shim_resident_add: the registry of resident arrays is full.")
    n_res = n_res + 1 ; n = n_res
    res_addr(n) = transfer(c_loc(a), res_addr(n)) ; res_stg(n) = stagger ; res_nlev(n) = nlev
    rc = mom6x_dev_alloc(the_ctx, res_dev(n), int(the_dims%slab, c_size_t) * int(max(nlev, 1), c_size_t)) ; call shim_check(rc, "mom6x_dev_alloc")
  endif
  rc = mom6x_upload(the_ctx, res_dev(n), a, int(stagger, c_int), int(nlev, c_int)) ; call shim_check(rc, "mom6x_upload")
  n_transfers = n_transfers + 1
end subroutine shim_resident_add

!> The host has changed a resident array itself (initialisation, a host-side parameterisation): upload it again.
subroutine shim_resident_host_changed(a)
  real(c_double), target, intent(in) :: a(*)
  integer :: n
  n = res_find(a)
  if (n == 0) call MOM_error(FATAL, "shim_resident_host_changed: the array is not resident.")
  call shim_resident_add(a, res_stg(n), res_nlev(n))
end subroutine shim_resident_host_changed

!> The host wants the values (diagnostics, save_restart): download, the array stays resident.
subroutine shim_resident_sync_host(a)
  real(c_double), target, intent(inout) :: a(*)
  integer(c_int) :: rc ; integer :: n
  n = res_find(a)
  if (n == 0) return
  rc = mom6x_download(the_ctx, a, res_dev(n), int(res_stg(n), c_int), int(res_nlev(n), c_int)) ; call shim_check(rc, "mom6x_download")
  n_transfers = n_transfers + 1
end subroutine shim_resident_sync_host

!> Take the array back: downloaded (unless told not to) and forgotten.
subroutine shim_resident_drop(a, download)
  real(c_double), target, intent(inout) :: a(*) ; logical, optional, intent(in) :: download
  integer(c_int) :: rc ; integer :: n
  logical :: down
  n = res_find(a)
  if (n == 0) return
  down = .true. ; if (present(download)) down = download
  if (down) call shim_resident_sync_host(a)
  rc = mom6x_dev_free(the_ctx, res_dev(n))
  res_addr(n) = res_addr(n_res) ; res_dev(n) = res_dev(n_res) ; res_stg(n) = res_stg(n_res) ; res_nlev(n) = res_nlev(n_res)
  res_addr(n_res) = 0 ; res_dev(n_res) = c_null_ptr ; n_res = n_res - 1
end subroutine shim_resident_drop

!> The device copy of a resident array (c_null_ptr: not resident): for a host that calls the C ABI itself
function shim_resident_dev(a) result(p)
  real(c_double), target, intent(in) :: a(*)
  type(c_ptr) :: p ; integer :: n
  n = res_find(a)
  p = c_null_ptr ; if (n > 0) p = res_dev(n)
end function shim_resident_dev

!> Arrays the shims have moved between host and device since the last reset
integer(c_long_long) function shim_transfer_count(reset)
  logical, optional, intent(in) :: reset
  shim_transfer_count = n_transfers
  if (present(reset)) then ; if (reset) n_transfers = 0 ; endif
end function shim_transfer_count

!> Host array (MOM6 symmetric-memory extents of its staggering) -> its resident copy, or uploaded into the scratch slot;
!! returns the device pointer.
function shim_up3(slot, a, stagger, nlev) result(p)
  integer, intent(in) :: slot, stagger, nlev ; real(c_double), target, intent(in) :: a(*)
  type(c_ptr) :: p ; integer(c_int) :: rc ; integer :: n
  n = res_find(a)
  if (n > 0) then ; p = res_dev(n) ; return ; endif
  p = shim_buf(slot, nlev)
  rc = mom6x_upload(the_ctx, p, a, int(stagger, c_int), int(nlev, c_int)) ; call shim_check(rc, "mom6x_upload")
  n_transfers = n_transfers + 1
end function shim_up3

!> Where a shim should have the device write the RESULT that goes to the host array `a`: the resident copy, or the scratch slot
!! (from which shim_down3 brings it to the host).
function shim_out3(slot, a, nlev) result(p)
  integer, intent(in) :: slot, nlev ; real(c_double), target, intent(in) :: a(*)
  type(c_ptr) :: p ; integer :: n
  n = res_find(a)
  if (n > 0) then ; p = res_dev(n) ; else ; p = shim_buf(slot, nlev) ; endif
end function shim_out3
function shim_out2(slot, a) result(p)
  integer, intent(in) :: slot ; real(c_double), target, intent(in) :: a(*)
  type(c_ptr) :: p
  p = shim_out3(slot, a, 1)
end function shim_out2

function shim_up2(slot, a, stagger) result(p)
  integer, intent(in) :: slot, stagger ; real(c_double), target, intent(in) :: a(*)
  type(c_ptr) :: p
  p = shim_up3(slot, a, stagger, 1)
end function shim_up2

!> A result on the device -> the host array; a resident host array keeps it in HBM (a copy on the device if the shim had it
!! written somewhere else).
subroutine shim_down3(a, p, stagger, nlev)
  real(c_double), target, intent(inout) :: a(*) ; type(c_ptr), intent(in) :: p ; integer, intent(in) :: stagger, nlev
  integer(c_int) :: rc ; integer :: n
  n = res_find(a)
  if (n > 0) then
    if (c_associated(p, res_dev(n))) return      ! (an in-out argument: the device has worked on the resident copy itself)
    rc = mom6x_dev_copy(the_ctx, res_dev(n), p, int(the_dims%slab, c_size_t) * int(max(nlev, 1), c_size_t)) ; call shim_check(rc, "mom6x_dev_copy")
    return
  endif
  rc = mom6x_download(the_ctx, a, p, int(stagger, c_int), int(nlev, c_int)) ; call shim_check(rc, "mom6x_download")
  n_transfers = n_transfers + 1
end subroutine shim_down3

subroutine shim_down2(a, p, stagger)
  real(c_double), target, intent(inout) :: a(*) ; type(c_ptr), intent(in) :: p ; integer, intent(in) :: stagger
  call shim_down3(a, p, stagger, 1)
end subroutine shim_down2

!> The last user (end_dyn_split_RK2) destroys the context.
subroutine shim_ctx_end()
  integer(c_int) :: rc ; integer :: n
  if (.not.c_associated(the_ctx)) return
  do n = 1, NBUF
    if (c_associated(bufs(n))) then ; rc = mom6x_dev_free(the_ctx, bufs(n)) ; bufs(n) = c_null_ptr ; buf_len(n) = 0 ; endif
  enddo
  do n = 1, n_res ; rc = mom6x_dev_free(the_ctx, res_dev(n)) ; res_dev(n) = c_null_ptr ; res_addr(n) = 0 ; enddo
  n_res = 0
  rc = mom6x_ctx_destroy(the_ctx) ; the_ctx = c_null_ptr
end subroutine shim_ctx_end

end module mom6x_shim_ctx
