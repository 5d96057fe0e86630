!> A Fortran host driving the MI355X dynamical core through ISO_C_BINDING: reads a case (grid metrics in MOM6's
!! symmetric-memory extents, parameters, state, forcing) written by tests/test_fortran_gpu.py, packs the metric block,
!! creates the context, initialises the modules in the order MOM.F90 does, uploads the state ONCE, runs nsteps of
!! step_MOM_dyn_split_RK2 on the resident state, downloads, and compares every prognostic array bit for bit with the
!! expected values of the case file (tests/golden/rk2_double_gyre_strong_drag_3steps: the oracle's); then the same run
!! interrupted before its last step -- restart set to the host through mom6x_download, context destroyed, a new context
!! restored from the saved set, last step -- must give the same bits (the reference's test.restart).  Depends on nothing
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
  real(c_double), allocatable :: u0(:,:,:), v0(:,:,:), h0(:,:,:), ua(:,:,:), va(:,:,:), ha(:,:,:), uha(:,:,:), vha(:,:,:), uhtra(:,:,:), vhtra(:,:,:), etaa(:,:)
  real(c_double), allocatable :: r_eta(:,:), r_uav(:,:,:), r_vav(:,:,:), r_CAu(:,:,:), r_CAv(:,:,:), r_diffu(:,:,:), r_diffv(:,:,:), r_ubtav(:,:), r_vbtav(:,:)
  real(c_double), target :: r_dtbt

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

  allocate(u0, source=u) ; allocate(v0, source=v) ; allocate(h0, source=h)

  ! ---- (1) the uninterrupted run: initialisation in the order of initialize_MOM (MOM.F90) / initialize_dyn_split_RK2
  !      (RK2.F90:1346), the state uploaded ONCE, nsteps steps on the resident state, one download where the host reads it
  call setup_model()
  call dyn_state_upload(S, u, v, h, uh, vh, uhtr, vhtr)
  rc = mom6x_dyn_split_RK2_new_run(ctx, S%u, S%v, S%h, S%uh, S%vh, dt) ; call must(rc, "dyn_split_RK2_new_run")
  do n = 1, nsteps
    call dyn_step(S, taux, tauy, dt, n == 1)
  enddo
  call dyn_state_download(S, u, v, h, uh, vh, uhtr, vhtr, eta_av)

  ! compare with the expected computational-domain values of the case file, bit for bit
  nbad = 0
  call compare3("u", u, STG_U) ; call compare3("v", v, STG_V) ; call compare3("h", h, STG_H)
  call compare3("uh", uh, STG_U) ; call compare3("vh", vh, STG_V) ; call compare3("uhtr", uhtr, STG_U) ; call compare3("vhtr", vhtr, STG_V)
  call compare2("eta_av", eta_av, STG_H)
  close(un)
  call end_model()
  allocate(ua, source=u) ; allocate(va, source=v) ; allocate(ha, source=h) ; allocate(uha, source=uh) ; allocate(vha, source=vh)
  allocate(uhtra, source=uhtr) ; allocate(vhtra, source=vhtr) ; allocate(etaa, source=eta_av)

  ! ---- (2) the same run interrupted after nsteps - 1 steps: the restart set of register_restarts_dyn_split_RK2 (:1222-1290)
  !      and register_barotropic_restarts (MOM_barotropic.F90:6279-6296) comes to the host through mom6x_download, the
  !      context is destroyed, a new one is initialised from the saved set, and the last step is taken: every prognostic
  !      array must equal the uninterrupted run's bit for bit (the reference's test.restart, .testing/Makefile).
  u = u0 ; v = v0 ; h = h0 ; uh = 0.0d0 ; vh = 0.0d0 ; uhtr = 0.0d0 ; vhtr = 0.0d0 ; eta_av = 0.0d0
  call setup_model()
  call dyn_state_upload(S, u, v, h, uh, vh, uhtr, vhtr)
  rc = mom6x_dyn_split_RK2_new_run(ctx, S%u, S%v, S%h, S%uh, S%vh, dt) ; call must(rc, "dyn_split_RK2_new_run")
  do n = 1, nsteps - 1
    call dyn_step(S, taux, tauy, dt, n == 1)
  enddo
  call dyn_state_download(S, u, v, h, uh, vh, uhtr, vhtr, eta_av)
  call alloc2(r_eta, STG_H) ; call alloc3(r_uav, STG_U, nk) ; call alloc3(r_vav, STG_V, nk) ; call alloc3(r_CAu, STG_U, nk)
  call alloc3(r_CAv, STG_V, nk) ; call alloc3(r_diffu, STG_U, nk) ; call alloc3(r_diffv, STG_V, nk)
  call alloc2(r_ubtav, STG_U) ; call alloc2(r_vbtav, STG_V)
  rc = mom6x_download(ctx, r_eta, mom6x_rk2_field(ctx, 16_c_int), STG_H, 1_c_int) ; call must(rc, "save sfc")
  rc = mom6x_download(ctx, r_uav, mom6x_rk2_field(ctx, 12_c_int), STG_U, int(nk, c_int)) ; call must(rc, "save u2")
  rc = mom6x_download(ctx, r_vav, mom6x_rk2_field(ctx, 13_c_int), STG_V, int(nk, c_int)) ; call must(rc, "save v2")
  rc = mom6x_download(ctx, r_CAu, mom6x_rk2_field(ctx, 2_c_int), STG_U, int(nk, c_int)) ; call must(rc, "save CAu")
  rc = mom6x_download(ctx, r_CAv, mom6x_rk2_field(ctx, 3_c_int), STG_V, int(nk, c_int)) ; call must(rc, "save CAv")
  rc = mom6x_download(ctx, r_diffu, mom6x_rk2_field(ctx, 6_c_int), STG_U, int(nk, c_int)) ; call must(rc, "save diffu")
  rc = mom6x_download(ctx, r_diffv, mom6x_rk2_field(ctx, 7_c_int), STG_V, int(nk, c_int)) ; call must(rc, "save diffv")
  rc = mom6x_download(ctx, r_ubtav, mom6x_barotropic_field(ctx, 0_c_int), STG_U, 1_c_int) ; call must(rc, "save ubtav")
  rc = mom6x_download(ctx, r_vbtav, mom6x_barotropic_field(ctx, 1_c_int), STG_V, 1_c_int) ; call must(rc, "save vbtav")
  rc = mom6x_barotropic_dtbt(ctx, c_loc(r_dtbt), c_null_ptr) ; call must(rc, "save DTBT")
  call end_model()

  call setup_model()                                   ! a new context: nothing of the old one survives
  call dyn_state_upload(S, u, v, h, uh, vh, uhtr, vhtr)
  rc = mom6x_upload(ctx, mom6x_rk2_field(ctx, 16_c_int), r_eta, STG_H, 1_c_int) ; call must(rc, "restore sfc")
  rc = mom6x_upload(ctx, mom6x_rk2_field(ctx, 12_c_int), r_uav, STG_U, int(nk, c_int)) ; call must(rc, "restore u2")
  rc = mom6x_upload(ctx, mom6x_rk2_field(ctx, 13_c_int), r_vav, STG_V, int(nk, c_int)) ; call must(rc, "restore v2")
  rc = mom6x_upload(ctx, mom6x_rk2_field(ctx, 2_c_int), r_CAu, STG_U, int(nk, c_int)) ; call must(rc, "restore CAu")
  rc = mom6x_upload(ctx, mom6x_rk2_field(ctx, 3_c_int), r_CAv, STG_V, int(nk, c_int)) ; call must(rc, "restore CAv")
  rc = mom6x_upload(ctx, mom6x_rk2_field(ctx, 6_c_int), r_diffu, STG_U, int(nk, c_int)) ; call must(rc, "restore diffu")
  rc = mom6x_upload(ctx, mom6x_rk2_field(ctx, 7_c_int), r_diffv, STG_V, int(nk, c_int)) ; call must(rc, "restore diffv")
  rc = mom6x_upload(ctx, mom6x_barotropic_field(ctx, 0_c_int), r_ubtav, STG_U, 1_c_int) ; call must(rc, "restore ubtav")
  rc = mom6x_upload(ctx, mom6x_barotropic_field(ctx, 1_c_int), r_vbtav, STG_V, 1_c_int) ; call must(rc, "restore vbtav")
  rc = mom6x_barotropic_dtbt(ctx, c_null_ptr, c_loc(r_dtbt)) ; call must(rc, "restore DTBT")
  rc = mom6x_rk2_set_CAu_pred_stored(ctx, 1_c_int) ; call must(rc, "CAu_pred stored")       ! query_initialized(CAu) :1616
  call dyn_step(S, taux, tauy, dt, .false.)
  call dyn_state_download(S, u, v, h, uh, vh, uhtr, vhtr, eta_av)
  call end_model()
  call same3("restart: u", u, ua) ; call same3("restart: v", v, va) ; call same3("restart: h", h, ha)
  call same3("restart: uh", uh, uha) ; call same3("restart: vh", vh, vha)
  call same3("restart: uhtr", uhtr, uhtra) ; call same3("restart: vhtr", vhtr, vhtra)
  call same3("restart: eta_av", reshape(eta_av, [size(eta_av, 1), size(eta_av, 2), 1]), reshape(etaa, [size(etaa, 1), size(etaa, 2), 1]))

  if (nbad == 0) then
    print '(a,i0,a)', "drive_double_gyre: PASS (", nsteps, " steps of step_MOM_dyn_split_RK2 from Fortran, 8 fields bit-identical; restarted run identical)"
  else
    print '(a,i0,a)', "drive_double_gyre: FAIL (", nbad, " fields differ)"
    error stop 1
  endif

contains
  !> A context with every module of the dynamical core initialised and the vertvisc coefficient set of the case in place
  subroutine setup_model()
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
  end subroutine setup_model
  subroutine end_model()
    call dyn_state_end(S)
    rc = mom6x_ctx_destroy(ctx) ; ctx = c_null_ptr
  end subroutine end_model
  subroutine same3(name, a, b)
    character(len=*), intent(in) :: name ; real(c_double), intent(in) :: a(:,:,:), b(:,:,:)
    integer :: i0, i1, j0, j1, ndiff
    ! the halos are not part of a restart file: compare the computational domain (one more face for u / v arrays)
    i0 = halo + 1 ; j0 = halo + 1 ; i1 = size(a, 1) - halo ; j1 = size(a, 2) - halo
    ndiff = count(transfer(reshape(a(i0:i1, j0:j1, :), [size(a(i0:i1, j0:j1, :))]), [1_int64]) /= &
                  transfer(reshape(b(i0:i1, j0:j1, :), [size(a(i0:i1, j0:j1, :))]), [1_int64]))
    call report(name, ndiff, maxval(abs(a(i0:i1, j0:j1, :) - b(i0:i1, j0:j1, :))), maxval(abs(b(i0:i1, j0:j1, :))))
  end subroutine same3
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
      print '(a,a,es10.3)', name, ": bit-identical, max |value| ", scale
    else
      print '(a,a,i0,a,es10.3,a,es10.3)', name, ": ", ndiff, " values differ, max |diff| ", err, " of ", scale
      nbad = nbad + 1
    endif
  end subroutine report
end program drive_double_gyre
