!> A Fortran host driving the MI355X dynamical core through ISO_C_BINDING: reads a case (grid metrics in MOM6's
!! symmetric-memory extents, parameters, state, forcing) written by tests/test_fortran_gpu.py, packs the metric block,
!! creates the context, initialises the modules in the order MOM.F90 does, uploads the state ONCE, runs nsteps of
!! step_MOM_dyn_split_RK2 on the resident state, downloads, and compares every prognostic array bit for bit with the
!! expected values of the case file (tests/golden/rk2_double_gyre_strong_drag_3steps: the oracle's).  Depends on nothing
!! but fortran/mom6x_c_api.F90, fortran/mom6x_host.F90 and libmom6x.so.   Usage: drive_double_gyre <case.bin>
program drive_double_gyre
  use, intrinsic :: iso_c_binding
  use, intrinsic :: iso_fortran_env, only : int64
  use mom6x_c_api
  use mom6x_host
  implicit none
  character(len=512) :: path
  integer :: un, ni, nj, nk, halo, nsteps, first_direction, nmet, rx, ry, magic, m, stg, nx, ny, n, nbad
  integer(c_int) :: rc
  real(c_double) :: dt
  type(mom6x_dims) :: d
  type(mom6x_vgrid) :: GV
  type(mom6x_continuity_params) :: cont
  type(mom6x_barotropic_params) :: bt
  type(mom6x_coriolis_params) :: cor
  type(mom6x_pgf_params) :: pgf
  type(mom6x_rk2_params) :: rk2
  real(c_double), allocatable :: Rlay(:), g_prime(:), block(:), plane(:,:)
  real(c_double), allocatable, target :: u(:,:,:), v(:,:,:), h(:,:,:), uh(:,:,:), vh(:,:,:), uhtr(:,:,:), vhtr(:,:,:), eta_av(:,:)
  real(c_double), allocatable :: a_u(:,:,:), a_v(:,:,:), h_u(:,:,:), h_v(:,:,:), Ray_u(:,:,:), Ray_v(:,:,:), taux(:,:), tauy(:,:)
  type(c_ptr) :: ctx, d_au, d_av, d_hu, d_hv, d_ru, d_rv
  type(dyn_state_type) :: S

  call get_command_argument(1, path)
  if (len_trim(path) == 0) error stop "usage: drive_double_gyre <case.bin>"
  open(newunit=un, file=trim(path), access="stream", form="unformatted", status="old", action="read")
  read(un) magic, ni, nj, nk, halo, nsteps, first_direction, nmet, rx, ry
  if (magic /= 1297042742) error stop "not a case file"
  if (nmet /= G_COUNT) error stop "the case file has a different number of metric planes than this build"
  read(un) dt
  read(un) GV ; read(un) cont ; read(un) bt ; read(un) cor ; read(un) pgf ; read(un) rk2
  allocate(Rlay(nk), g_prime(nk)) ; read(un) Rlay ; read(un) g_prime

  rc = mom6x_dims_init(d, int(ni, c_int), int(nj, c_int), int(nk, c_int), int(halo, c_int))
  if (rc /= 0) error stop "mom6x_dims_init"
  d%reentrant_x = rx ; d%reentrant_y = ry
  allocate(block(0:int(G_COUNT, c_size_t) * int(d%slab, c_size_t) - 1)) ; block = 0.0d0
  do m = 0, nmet - 1          ! what a shim does with G%mask2dT, G%dxT, ... : one array per plane, natural extents
    read(un) stg
    call stagger_extent(d, stg, nx, ny)
    allocate(plane(nx, ny)) ; read(un) plane
    call mom6x_pack_plane(d, block, m, plane, stg)
    deallocate(plane)
  enddo

  call alloc3(u, STG_U, nk) ; call alloc3(v, STG_V, nk) ; call alloc3(h, STG_H, nk)
  call alloc3(uh, STG_U, nk) ; call alloc3(vh, STG_V, nk) ; call alloc3(uhtr, STG_U, nk) ; call alloc3(vhtr, STG_V, nk)
  call alloc3(a_u, STG_U, nk + 1) ; call alloc3(a_v, STG_V, nk + 1) ; call alloc3(h_u, STG_U, nk) ; call alloc3(h_v, STG_V, nk)
  call alloc3(Ray_u, STG_U, nk) ; call alloc3(Ray_v, STG_V, nk)
  call alloc2(eta_av, STG_H) ; call alloc2(taux, STG_U) ; call alloc2(tauy, STG_V)
  read(un) u ; read(un) v ; read(un) h
  read(un) a_u ; read(un) a_v ; read(un) h_u ; read(un) h_v ; read(un) Ray_u ; read(un) Ray_v
  read(un) taux ; read(un) tauy
  uh = 0.0d0 ; vh = 0.0d0 ; uhtr = 0.0d0 ; vhtr = 0.0d0 ; eta_av = 0.0d0

  ! ---- initialisation, in the order of initialize_MOM (MOM.F90) / initialize_dyn_split_RK2 (RK2.F90:1346) ----------
  rc = mom6x_ctx_create(ctx, d, 0_c_int, block, GV, int(first_direction, c_int)) ; call must(rc, "mom6x_ctx_create")
  rc = mom6x_continuity_init(ctx, cont) ; call must(rc, "continuity_init")
  rc = mom6x_barotropic_init(ctx, bt) ; call must(rc, "barotropic_init")
  rc = mom6x_CoriolisAdv_init(ctx, cor) ; call must(rc, "CoriolisAdv_init")
  rc = mom6x_PressureForce_init(ctx, pgf, Rlay, g_prime) ; call must(rc, "PressureForce_init")
  rc = mom6x_initialize_dyn_split_RK2(ctx, rk2) ; call must(rc, "initialize_dyn_split_RK2")
  ! the vertvisc coefficients of this case are given (CS%a_u, CS%a_v, CS%h_u, CS%h_v of vertvisc_coef; visc%Ray_u, visc%Ray_v)
  rc = mom6x_dev_alloc(ctx, d_au, int(d%slab, c_size_t) * (nk + 1)) ; rc = mom6x_dev_alloc(ctx, d_av, int(d%slab, c_size_t) * (nk + 1))
  rc = mom6x_dev_alloc(ctx, d_hu, int(d%slab, c_size_t) * nk) ; rc = mom6x_dev_alloc(ctx, d_hv, int(d%slab, c_size_t) * nk)
  rc = mom6x_upload(ctx, d_au, a_u, STG_U, int(nk + 1, c_int)) ; call must(rc, "upload a_u")
  rc = mom6x_upload(ctx, d_av, a_v, STG_V, int(nk + 1, c_int)) ; call must(rc, "upload a_v")
  rc = mom6x_upload(ctx, d_hu, h_u, STG_U, int(nk, c_int)) ; rc = mom6x_upload(ctx, d_hv, h_v, STG_V, int(nk, c_int))
  rc = mom6x_dev_alloc(ctx, d_ru, int(d%slab, c_size_t) * nk) ; rc = mom6x_dev_alloc(ctx, d_rv, int(d%slab, c_size_t) * nk)
  rc = mom6x_upload(ctx, d_ru, Ray_u, STG_U, int(nk, c_int)) ; rc = mom6x_upload(ctx, d_rv, Ray_v, STG_V, int(nk, c_int))
  rc = mom6x_vertvisc_set_coef(ctx, d_au, d_av, d_hu, d_hv, d_ru, d_rv) ; call must(rc, "vertvisc_set_coef")

  call dyn_state_init(S, ctx, d)
  call dyn_state_upload(S, u, v, h, uh, vh, uhtr, vhtr)              ! ONCE: the state stays in HBM
  rc = mom6x_dyn_split_RK2_new_run(ctx, S%u, S%v, S%h, S%uh, S%vh, dt) ; call must(rc, "dyn_split_RK2_new_run")
  do n = 1, nsteps
    call dyn_step(S, taux, tauy, dt, n == 1)
  enddo
  call dyn_state_download(S, u, v, h, uh, vh, uhtr, vhtr, eta_av)   ! where the host reads it: here, to compare

  ! ---- compare with the expected computational-domain values, bit for bit -------------------------------------------
  nbad = 0
  call compare3("u", u, STG_U) ; call compare3("v", v, STG_V) ; call compare3("h", h, STG_H)
  call compare3("uh", uh, STG_U) ; call compare3("vh", vh, STG_V) ; call compare3("uhtr", uhtr, STG_U) ; call compare3("vhtr", vhtr, STG_V)
  call compare2("eta_av", eta_av, STG_H)
  close(un)
  call dyn_state_end(S)
  rc = mom6x_ctx_destroy(ctx)
  if (nbad == 0) then
    print '(a,i0,a)', "drive_double_gyre: PASS (", nsteps, " steps of step_MOM_dyn_split_RK2 from Fortran, 8 fields bit-identical)"
  else
    print '(a,i0,a)', "drive_double_gyre: FAIL (", nbad, " fields differ)"
    error stop 1
  endif

contains
  subroutine alloc3(a, stg, nl)
    real(c_double), allocatable, intent(out) :: a(:,:,:) ; integer, intent(in) :: stg, nl
    integer :: nx, ny
    call stagger_extent(d, stg, nx, ny) ; allocate(a(nx, ny, nl))
  end subroutine alloc3
  subroutine alloc2(a, stg)
    real(c_double), allocatable, intent(out) :: a(:,:) ; integer, intent(in) :: stg
    integer :: nx, ny
    call stagger_extent(d, stg, nx, ny) ; allocate(a(nx, ny))
  end subroutine alloc2
  subroutine must(rc, what)
    integer(c_int), intent(in) :: rc ; character(len=*), intent(in) :: what
    if (rc /= 0) then
      print '(a)', what//": "//trim(mom6x_message()) ; error stop 2
    endif
  end subroutine must
  !> The computational domain of a staggering inside the symmetric-memory array (u: I = isc-1..iec, v: J = jsc-1..jec)
  subroutine cdom(stg, i0, i1, j0, j1)
    integer, intent(in) :: stg ; integer, intent(out) :: i0, i1, j0, j1
    i0 = halo + 1 ; i1 = halo + ni ; j0 = halo + 1 ; j1 = halo + nj
    if (stg == STG_U .or. stg == STG_Q) i1 = i1 + 1     ! (the array starts one face further west)
    if (stg == STG_V .or. stg == STG_Q) j1 = j1 + 1
  end subroutine cdom
  subroutine compare3(name, a, stg)
    character(len=*), intent(in) :: name ; real(c_double), intent(in) :: a(:,:,:) ; integer, intent(in) :: stg
    real(c_double), allocatable :: want(:,:,:)
    integer :: i0, i1, j0, j1, ndiff
    call cdom(stg, i0, i1, j0, j1)
    allocate(want(i1 - i0 + 1, j1 - j0 + 1, size(a, 3))) ; read(un) want
    ndiff = count(transfer(reshape(a(i0:i1, j0:j1, :), [size(want)]), [1_int64]) /= transfer(reshape(want, [size(want)]), [1_int64]))
    call report(name, ndiff, maxval(abs(a(i0:i1, j0:j1, :) - want)), maxval(abs(want)))
  end subroutine compare3
  subroutine compare2(name, a, stg)
    character(len=*), intent(in) :: name ; real(c_double), intent(in) :: a(:,:) ; integer, intent(in) :: stg
    real(c_double), allocatable :: want(:,:)
    integer :: i0, i1, j0, j1, ndiff
    call cdom(stg, i0, i1, j0, j1)
    allocate(want(i1 - i0 + 1, j1 - j0 + 1)) ; read(un) want
    ndiff = count(transfer(reshape(a(i0:i1, j0:j1), [size(want)]), [1_int64]) /= transfer(reshape(want, [size(want)]), [1_int64]))
    call report(name, ndiff, maxval(abs(a(i0:i1, j0:j1) - want)), maxval(abs(want)))
  end subroutine compare2
  subroutine report(name, ndiff, err, scale)
    character(len=*), intent(in) :: name ; integer, intent(in) :: ndiff ; real(c_double), intent(in) :: err, scale
    if (ndiff == 0) then
      print '(a8,a,es10.3)', name, ": bit-identical, max |value| ", scale
    else
      print '(a8,a,i0,a,es10.3,a,es10.3)', name, ": ", ndiff, " values differ, max |diff| ", err, " of ", scale
      nbad = nbad + 1
    endif
  end subroutine report
end program drive_double_gyre
