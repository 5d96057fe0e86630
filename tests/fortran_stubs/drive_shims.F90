!> tests/fortran_stubs/drive_shims.F90 -- a miniature of what MOM.F90 does with the dynamical core, driving the SHIM MODULES of
!! fortran/shims/ (module MOM_dynamics_split_RK2 and, through it, MOM_continuity_PPM, MOM_barotropic, MOM_CoriolisAdv,
!! MOM_PressureForce, MOM_vert_friction; then MOM_tracer_advect and mom6x_diabatic_solvers directly) on the GPU, against the
!! interface stand-ins of mom_stubs.F90.  TEST INFRASTRUCTURE.  The case (grid, parameters as a MOM_input-like table, state,
!! set_viscous_BBL fields, forcing, expected final state) is written by tests/test_fortran_gpu.py.
!!
!!   run A   register_restarts -> initialize (new run) -> nsteps x step_MOM_dyn_split_RK2            == expected, bit for bit
!!   run B   the same for `save_after` steps, then what save_restart does with the registry -> a file; end_dyn_split_RK2
!!   run C   register_restarts -> restore_state from the file -> initialize (restart branch: mirrors uploaded, DTBT kept,
!!           CAu_pred marked as stored) -> the remaining steps                                       == expected, bit for bit
!!   run D   the same file read as an older one without CAu, CAv (initialize_dyn_split_RK2 :1620-1640 re-forms them): close to
!!           the uninterrupted run, nothing left at its zero fill
!!   then    one advect_tracer call (uhr_out / vhr_out honoured) and one tracer_vertdiff call through their shims: a uniform
!!           tracer stays uniform, the leftover transports are finite and smaller than the transports.
!! Exit code 0 and the word PASS only if everything holds.  Usage: drive_shims <case.bin> <scratch restart file> [resident]
program drive_shims
  use, intrinsic :: iso_c_binding
  use MOM_coms, only : stub_npes
  use MOM_cpu_clock, only : stub_clock_calls, stub_clock_names, stub_nclocks
  use MOM_diag_mediator, only : diag_ctrl
  use MOM_domains, only : MOM_domain_type
  use MOM_dynamics_split_RK2
  use MOM_ALE, only : ALE_CS, ALE_init, ALE_end, ALE_regrid, ALE_remap_tracers, ALE_remap_set_h_vel, ALE_remap_velocities, &
                      ALE_update_regrid_weights, ALE_remap_init_conds, pre_ALE_adjustments
  use MOM_checksums, only : hchksum, uvchksum, Bchksum, hchksum_pair, MOM_checksums_init
  use MOM_hor_visc, only : hor_visc_CS, hor_visc_init, hor_visc_end, horizontal_viscosity, hor_visc_vel_stencil
  use MOM_file_parser, only : param_file_type, stub_set_param
  use MOM_forcing_type, only : mech_forcing
  use MOM_get_input, only : directories
  use MOM_grid, only : ocean_grid_type
  use MOM_harmonic_analysis, only : harmonic_analysis_CS
  use MOM_hor_index, only : hor_index_type
  use MOM_MEKE_types, only : MEKE_type
  use MOM_lateral_mixing_coeffs, only : VarMix_CS
  use MOM_open_boundary, only : ocean_OBC_type, update_OBC_CS
  use MOM_porous_barriers, only : porous_barrier_type
  use MOM_restart, only : MOM_restart_CS, register_restart_field, stub_save_restart, stub_restore_state, stub_restart_names
  use MOM_set_visc, only : set_visc_CS
  use MOM_stochastics, only : stochastic_CS
  use MOM_thickness_diffuse, only : thickness_diffuse_CS
  use MOM_time_manager, only : time_type
  use MOM_tracer_advect, only : advect_tracer, tracer_advect_init, tracer_advect_end, tracer_advect_CS
  use MOM_tracer_registry, only : tracer_registry_type
  use MOM_unit_scaling, only : unit_scale_type
  use MOM_variables, only : thermo_var_ptrs, vertvisc_type, ocean_internal_state, accel_diag_ptrs, cont_diag_ptrs
  use MOM_verticalGrid, only : verticalGrid_type
  use mom6x_diabatic_solvers, only : tracer_vertdiff
  use MOM_barotropic, only : barotropic_CS, barotropic_init, register_barotropic_restarts, btstep, btcalc, bt_mass_source, barotropic_end, set_dtbt
  use MOM_CoriolisAdv, only : CoriolisAdv_CS, CoriolisAdv_init, CorAdCalc, CoriolisAdv_end
  use MOM_variables, only : BT_cont_type
  use mom6x_shim_ctx, only : shim_resident_add, shim_resident_sync_host, shim_resident_drop, shim_transfer_count
  implicit none
  character(len=512) :: path, rpath, mode
  integer :: un, ni, nj, nk, halo, nsteps, save_after, first_direction, nmet, nparams, magic, m, n, nbad
  real :: dt, gvs(9)
  character(len=48) :: pname ; character(len=64) :: pvalue
  type(ocean_grid_type), target :: G
  type(MOM_domain_type), target :: Dom
  type(hor_index_type) :: HI
  type(verticalGrid_type) :: GV
  type(unit_scale_type) :: US
  type(param_file_type) :: PF
  type(time_type), target :: Time
  type(diag_ctrl), target :: diag
  type(thermo_var_ptrs) :: tv
  type(vertvisc_type) :: visc
  type(mech_forcing) :: forces
  type(accel_diag_ptrs), target :: ADp
  type(cont_diag_ptrs), target :: CDp
  type(ocean_internal_state) :: MIS
  type(VarMix_CS) :: VarMix ; type(MEKE_type) :: MEKE ; type(thickness_diffuse_CS) :: TD ; type(porous_barrier_type) :: pbv
  type(stochastic_CS) :: STOCH ; type(set_visc_CS), target :: set_visc ; type(directories) :: dirs
  type(ocean_OBC_type), pointer :: OBC => NULL() ; type(update_OBC_CS), pointer :: update_OBC => NULL()
  type(ALE_CS), pointer :: ALE_CSp => NULL() ; type(harmonic_analysis_CS), pointer :: HA_CSp => NULL()
  type(MOM_dyn_split_RK2_CS), pointer :: CS => NULL()
  type(MOM_restart_CS) :: RCS
  real, pointer :: p_surf_begin(:,:) => NULL(), p_surf_end(:,:) => NULL()
  real, allocatable, target :: u(:,:,:), v(:,:,:), h(:,:,:), uh(:,:,:), vh(:,:,:), uhtr(:,:,:), vhtr(:,:,:), eta_av(:,:), eta(:,:)
  real, allocatable, target :: u0(:,:,:), v0(:,:,:), h0(:,:,:), taux(:,:), tauy(:,:)
  real, allocatable :: xu(:,:,:), xv(:,:,:), xh(:,:,:), xuh(:,:,:), xvh(:,:,:), xuhtr(:,:,:), xvhtr(:,:,:), xeta(:,:)
  real, allocatable :: dCAu(:,:,:), dCAv(:,:,:), dPFu(:,:,:), dPFv(:,:,:), ddiffu(:,:,:), ddiffv(:,:,:), dubt(:,:,:), dvbt(:,:,:)
  real, allocatable :: dpbce(:,:,:), duav(:,:,:), dvav(:,:,:)
  real, allocatable, target :: T0(:,:,:), Trho(:,:,:), Srho(:,:,:)
  real, allocatable :: yh(:,:,:), yT(:,:,:), yS(:,:,:), yhn(:,:,:), ydz(:,:,:)
  real, allocatable :: xdiffu(:,:,:), xdiffv(:,:,:), xhn(:,:,:), xdz(:,:,:), xT(:,:,:), xur(:,:,:), xvr(:,:,:)
  real :: max_depth
  integer :: magic2
  integer, target :: ntrunc
  integer :: cont_stencil
  logical :: calc_dtbt, resident

  call get_command_argument(1, path) ; call get_command_argument(2, rpath) ; call get_command_argument(3, mode)
  if (len_trim(path) == 0 .or. len_trim(rpath) == 0) error stop "usage: drive_shims <case.bin> <scratch restart file> [resident]"
  resident = (trim(mode) == "resident")
  open(newunit=un, file=trim(path), access="stream", form="unformatted", status="old", action="read")
  read(un) magic, ni, nj, nk, halo, nsteps, save_after, first_direction, nmet, nparams
  if (magic /= 1297042743) error stop "not a shim case file"
  read(un) dt ; read(un) gvs
  ! ---- verticalGrid_type, unit_scale_type (all scaling factors 1) ---------------------------------------------------------
  GV%ke = nk ; GV%g_Earth = gvs(1) ; GV%Rho0 = gvs(2) ; GV%Angstrom_H = gvs(3) ; GV%Angstrom_Z = gvs(3) ; GV%Angstrom_m = gvs(3)
  GV%H_subroundoff = gvs(4) ; GV%dZ_subroundoff = gvs(5) ; GV%H_to_Z = gvs(6) ; GV%Z_to_H = gvs(7) ; GV%H_to_RZ = gvs(8) ; GV%RZ_to_H = gvs(9)
  allocate(GV%Rlay(nk), GV%g_prime(nk)) ; read(un) GV%Rlay ; read(un) GV%g_prime
  do m = 1, nparams ; read(un) pname, pvalue ; call stub_set_param(PF, trim(pname), trim(pvalue)) ; enddo
  if (resident) call stub_set_param(PF, "MOM6X_RESIDENT_STATE", "True")
  ! ---- ocean_grid_type with MOM6's symmetric-memory index conventions (isc = halo + 1, IsdB = isd - 1) ---------------------
  G%isc = halo + 1 ; G%iec = halo + ni ; G%jsc = halo + 1 ; G%jec = halo + nj
  G%isd = 1 ; G%ied = ni + 2*halo ; G%jsd = 1 ; G%jed = nj + 2*halo
  G%IscB = G%isc - 1 ; G%IecB = G%iec ; G%JscB = G%jsc - 1 ; G%JecB = G%jec
  G%IsdB = G%isd - 1 ; G%IedB = G%ied ; G%JsdB = G%jsd - 1 ; G%JedB = G%jed ; G%ke = nk
  G%first_direction = first_direction ; G%symmetric = .true.
  Dom%niglobal = ni ; Dom%njglobal = nj ; Dom%nihalo = halo ; Dom%njhalo = halo ; Dom%layout = (/ 1, 1 /) ; G%Domain => Dom
  HI%isc = G%isc ; HI%iec = G%iec ; HI%jsc = G%jsc ; HI%jec = G%jec ; HI%isd = G%isd ; HI%ied = G%ied ; HI%jsd = G%jsd ; HI%jed = G%jed
  HI%IscB = G%IscB ; HI%IecB = G%IecB ; HI%JscB = G%JscB ; HI%JecB = G%JecB
  HI%IsdB = G%IsdB ; HI%IedB = G%IedB ; HI%JsdB = G%JsdB ; HI%JedB = G%JedB ; G%HI = HI
  if (nmet /= 33) error stop "the case file has a different number of metric planes than this driver reads"
  call rd2(G%mask2dT, 0) ; call rd2(G%mask2dCu, 1) ; call rd2(G%mask2dCv, 2) ; call rd2(G%mask2dBu, 3)
  call rd2(G%dxT, 0) ; call rd2(G%dyT, 0) ; call rd2(G%IdxT, 0) ; call rd2(G%IdyT, 0)
  call rd2(G%dxCu, 1) ; call rd2(G%dyCu, 1) ; call rd2(G%IdxCu, 1) ; call rd2(G%IdyCu, 1)
  call rd2(G%dxCv, 2) ; call rd2(G%dyCv, 2) ; call rd2(G%IdxCv, 2) ; call rd2(G%IdyCv, 2)
  call rd2(G%dxBu, 3) ; call rd2(G%dyBu, 3) ; call rd2(G%IdxBu, 3) ; call rd2(G%IdyBu, 3)
  call rd2(G%areaT, 0) ; call rd2(G%IareaT, 0) ; call rd2(G%areaBu, 3) ; call rd2(G%IareaBu, 3)
  call rd2(G%areaCu, 1) ; call rd2(G%areaCv, 2) ; call rd2(G%IareaCu, 1) ; call rd2(G%IareaCv, 2)
  call rd2(G%dy_Cu, 1) ; call rd2(G%dx_Cv, 2) ; call rd2(G%bathyT, 0) ; call rd2(G%CoriolisBu, 3) ; call rd2(G%Coriolis2Bu, 3)
  ! ---- state, set_viscous_BBL outputs, forcing, expected results ---------------------------------------------------------------
  call rd3(u0, 1, nk) ; call rd3(v0, 2, nk) ; call rd3(h0, 0, nk)
  call rd2(visc%Kv_bbl_u, 1) ; call rd2(visc%Kv_bbl_v, 2) ; call rd2(visc%bbl_thick_u, 1) ; call rd2(visc%bbl_thick_v, 2)
  call rd3(visc%Kv_shear, 0, nk + 1)
  call rd2(taux, 1) ; call rd2(tauy, 2)
  forces%taux => taux ; forces%tauy => tauy
  allocate(xu(ni+1,nj,nk), xv(ni,nj+1,nk), xh(ni,nj,nk), xuh(ni+1,nj,nk), xvh(ni,nj+1,nk), xuhtr(ni+1,nj,nk), xvhtr(ni,nj+1,nk), xeta(ni,nj))
  read(un) xu ; read(un) xv ; read(un) xh ; read(un) xuh ; read(un) xvh ; read(un) xuhtr ; read(un) xvhtr ; read(un) xeta
  allocate(dCAu(ni+1,nj,nk), dCAv(ni,nj+1,nk), dPFu(ni+1,nj,nk), dPFv(ni,nj+1,nk), ddiffu(ni+1,nj,nk), ddiffv(ni,nj+1,nk), &
           dubt(ni+1,nj,nk), dvbt(ni,nj+1,nk), dpbce(ni,nj,nk), duav(ni+1,nj,nk), dvav(ni,nj+1,nk))
  read(un) dCAu ; read(un) dCAv ; read(un) dPFu ; read(un) dPFv ; read(un) ddiffu ; read(un) ddiffv ; read(un) dubt ; read(un) dvbt
  read(un) dpbce ; read(un) duav ; read(un) dvav
  ! ---- the standalone calls of MOM_hor_visc and MOM_ALE: a tracer, and the oracle's results for the INITIAL state -------------
  read(un) magic2 ; if (magic2 /= 1297042744) error stop "the case file has no hor_visc / ALE section"
  read(un) max_depth
  call rd3(T0, 0, nk)
  allocate(xdiffu(ni+1,nj,nk), xdiffv(ni,nj+1,nk), xhn(ni,nj,nk), xdz(ni,nj,nk+1), xT(ni,nj,nk), xur(ni+1,nj,nk), xvr(ni,nj+1,nk))
  read(un) xdiffu ; read(un) xdiffv ; read(un) xhn ; read(un) xdz ; read(un) xT ; read(un) xur ; read(un) xvr
  ! the RHO coordinate: T, S of a column with some static instability, the oracle's convective_adjustment and regridding of it
  call rd3(Trho, 0, nk) ; call rd3(Srho, 0, nk)
  allocate(yh(ni,nj,nk), yT(ni,nj,nk), yS(ni,nj,nk), yhn(ni,nj,nk), ydz(ni,nj,nk+1))
  read(un) yh ; read(un) yT ; read(un) yS ; read(un) yhn ; read(un) ydz
  close(un)
  call al3(u, 1) ; call al3(v, 2) ; call al3(h, 0) ; call al3(uh, 1) ; call al3(vh, 2) ; call al3(uhtr, 1) ; call al3(vhtr, 2)
  allocate(eta_av(G%isd:G%ied,G%jsd:G%jed), eta(G%isd:G%ied,G%jsd:G%jed))
  nbad = 0

  ! =================================== run A: uninterrupted ====================================================================
  call fresh_state()
  call start_model(restore=.false.)
  do n = 1, nsteps ; call one_step(n) ; enddo
  call finish_host_view()
  call compare_all("A (uninterrupted)")
  call stop_model()

  ! =================================== run B: save_after steps, save_restart, end ==============================================
  call fresh_state()
  call start_model(restore=.false.)
  do n = 1, save_after ; call one_step(n) ; enddo
  call finish_host_view()
  if (resident) call refresh_host_mirrors(CS, G, GV)      ! what MOM.F90 does before save_restart in the resident mode
  print '(a)', "registered restart variables:"//trim(stub_restart_names(RCS))
  call stub_save_restart(RCS, trim(rpath))
  call stop_model()

  ! =================================== run C: restore_state, the remaining steps ==============================================
  u = 0.0 ; v = 0.0 ; h = GV%Angstrom_H ; uh = 0.0 ; vh = 0.0 ; eta_av = 0.0 ; eta = 0.0
  ! uhtr, vhtr are not restart variables: MOM.F90 zeroes them after every tracer step; here they carry on from run B
  call start_model(restore=.true.)
  ! (DTBT is a fraction here: barotropic_init keeps the file's DTBT and, by the reference's rule MOM_barotropic.F90:5970, still
  !  asks for one set_dtbt -- which returns the same value, pbce of the layered pressure force being independent of the state)
  if (.not.calc_dtbt) then ; print '(a)', "FAIL: calc_dtbt must be true unless DTBT is fixed AND in the restart file" ; nbad = nbad + 1 ; endif
  do n = save_after + 1, nsteps ; call one_step(n) ; enddo
  call finish_host_view()
  call compare_all("C (restarted)")
  call compare_diag_pointers()
  call check_clocks()
  call tracer_checks()
  call resident_submodule_checks()
  call hor_visc_and_ALE_checks()
  call checksum_checks()
  call stop_model()

  ! =================================== run D: the same file read as one WITHOUT CAu, CAv =======================================
  ! initialize_dyn_split_RK2 :1620-1640 then forms h_av with one continuity call on u2, v2 and CAu_pred, CAv_pred with CorAdCalc:
  ! not the uninterrupted run bit for bit (the stored accelerations came from the end-of-step transports), but the same flow
  ! to the accuracy of one step's time-centring.  (Round 3 left h_av at zero on this path.)
  u = 0.0 ; v = 0.0 ; h = GV%Angstrom_H ; uh = 0.0 ; vh = 0.0 ; eta_av = 0.0 ; eta = 0.0 ; uhtr = 0.0 ; vhtr = 0.0
  call start_model(restore=.true., skip="CAu,CAv")
  do n = save_after + 1, nsteps ; call one_step(n) ; enddo
  call finish_host_view()
  call near3("D (restart file without CAu, CAv) u", u(G%IscB:G%IecB,G%jsc:G%jec,:), xu, 1.0e-5)
  call near3("D (restart file without CAu, CAv) v", v(G%isc:G%iec,G%JscB:G%JecB,:), xv, 1.0e-5)
  call near3("D (restart file without CAu, CAv) h", h(G%isc:G%iec,G%jsc:G%jec,:), xh, 1.0e-3)
  call stop_model()

  if (nbad == 0) then
    print '(a)', "PASS"
  else
    print '(a,i0)', "FAILED checks: ", nbad ; error stop 1
  endif

contains

  subroutine fresh_state()
    u = u0 ; v = v0 ; h = h0 ; uh = 0.0 ; vh = 0.0 ; uhtr = 0.0 ; vhtr = 0.0 ; eta_av = 0.0 ; eta = 0.0
  end subroutine fresh_state

  !> MOM.F90's initialisation order for this module: the restart registrations (u, v, h are MOM.F90's own), restore_state on
  !! a restarted run, then initialize_dyn_split_RK2.
  subroutine start_model(restore, skip)
    logical, intent(in) :: restore
    character(len=*), optional, intent(in) :: skip
    type(MOM_restart_CS) :: fresh
    RCS = fresh
    call register_restart_field(u, "u", .true., RCS) ; call register_restart_field(v, "v", .true., RCS)
    call register_restart_field(h, "h", .true., RCS)
    call register_restarts_dyn_split_RK2(HI, GV, US, PF, CS, RCS, uh, vh)
    if (restore .and. present(skip)) then ; call stub_restore_state(RCS, trim(rpath), skip)
    elseif (restore) then ; call stub_restore_state(RCS, trim(rpath)) ; endif
    call initialize_dyn_split_RK2(u, v, h, tv, uh, vh, eta, Time, G, GV, US, PF, diag, CS, HA_CSp, RCS, dt, ADp, CDp, MIS, &
                                  VarMix, MEKE, TD, OBC, update_OBC, ALE_CSp, set_visc, visc, dirs, ntrunc, pbv, calc_dtbt, cont_stencil)
    if (cont_stencil /= 3) then ; print '(a,i0)', "FAIL: continuity_stencil = ", cont_stencil ; nbad = nbad + 1 ; endif
    if (resident) call dyn_split_RK2_host_changed(CS)     ! the host's uhtr, vhtr (carried over a restart here) are newer
  end subroutine start_model

  subroutine one_step(n)
    integer, intent(in) :: n
    logical :: cd
    cd = calc_dtbt .and. (n == 1 .or. n == save_after + 1)     ! MOM.F90:1297-1301: only the first step after initialisation
    call step_MOM_dyn_split_RK2(u, v, h, tv, visc, Time, dt, forces, p_surf_begin, p_surf_end, uh, vh, uhtr, vhtr, eta_av, &
                                G, GV, US, CS, cd, VarMix, MEKE, TD, pbv, STOCH)
    calc_dtbt = .false.
  end subroutine one_step

  !> In the resident mode the host asks for the state where it reads it (here: before comparing / saving)
  subroutine finish_host_view()
    if (resident) call dyn_split_RK2_sync_host(CS, u, v, h, uh, vh, uhtr, vhtr, eta_av)
  end subroutine finish_host_view

  subroutine stop_model()
    call end_dyn_split_RK2(CS)
    if (associated(CS)) then ; print '(a)', "FAIL: end_dyn_split_RK2 left CS associated" ; nbad = nbad + 1 ; endif
  end subroutine stop_model

  subroutine compare_all(tag)
    character(len=*), intent(in) :: tag
    call cmp3(tag//" u", u(G%IscB:G%IecB,G%jsc:G%jec,:), xu) ; call cmp3(tag//" v", v(G%isc:G%iec,G%JscB:G%JecB,:), xv)
    call cmp3(tag//" h", h(G%isc:G%iec,G%jsc:G%jec,:), xh)
    call cmp3(tag//" uh", uh(G%IscB:G%IecB,G%jsc:G%jec,:), xuh) ; call cmp3(tag//" vh", vh(G%isc:G%iec,G%JscB:G%JecB,:), xvh)
    call cmp3(tag//" uhtr", uhtr(G%IscB:G%IecB,G%jsc:G%jec,:), xuhtr) ; call cmp3(tag//" vhtr", vhtr(G%isc:G%iec,G%JscB:G%JecB,:), xvhtr)
    call cmp3(tag//" eta_av", reshape(eta_av(G%isc:G%iec,G%jsc:G%jec), (/ ni, nj, 1 /)), reshape(xeta, (/ ni, nj, 1 /)))
  end subroutine compare_all

  subroutine cmp3(name, a, b)
    character(len=*), intent(in) :: name ; real, intent(in) :: a(:,:,:), b(:,:,:)
    integer :: nd
    nd = count(transfer(a, 1_8, size(a)) /= transfer(b, 1_8, size(b)))      ! bit patterns, zeros of opposite sign included
    if (nd == 0) then
      print '(a)', trim(name)//": bit-identical"
    else
      print '(a,i0,a,es10.3)', trim(name)//": ", nd, " values differ, max |diff| = ", maxval(abs(a - b)) ; nbad = nbad + 1
    endif
  end subroutine cmp3

  !> What initialize_dyn_split_RK2 hands to MOM_diagnostics (Accel_diag) and MOM.F90 (MIS) by pointer, RK2.F90:1512-1534:
  !! associated, and after the last step equal to the oracle's arrays.  In the resident mode the host asks for them
  !! (refresh_host_mirrors), as before save_restart.
  subroutine compare_diag_pointers()
    if (resident) call refresh_host_mirrors(CS, G, GV)
    if (.not.(associated(ADp%CAu) .and. associated(ADp%CAv) .and. associated(ADp%PFu) .and. associated(ADp%PFv) .and. &
              associated(ADp%diffu) .and. associated(ADp%diffv) .and. associated(ADp%u_accel_bt) .and. associated(ADp%v_accel_bt) .and. &
              associated(MIS%pbce) .and. associated(MIS%u_av) .and. associated(MIS%v_av) .and. associated(MIS%CAu))) then
      print '(a)', "FAIL: initialize_dyn_split_RK2 left Accel_diag / MIS pointers unassociated" ; nbad = nbad + 1 ; return
    endif
    call cmp3("Accel_diag%CAu", ADp%CAu(G%IscB:G%IecB,G%jsc:G%jec,:), dCAu) ; call cmp3("Accel_diag%CAv", ADp%CAv(G%isc:G%iec,G%JscB:G%JecB,:), dCAv)
    call cmp3("Accel_diag%PFu", ADp%PFu(G%IscB:G%IecB,G%jsc:G%jec,:), dPFu) ; call cmp3("Accel_diag%PFv", ADp%PFv(G%isc:G%iec,G%JscB:G%JecB,:), dPFv)
    call cmp3("Accel_diag%diffu", ADp%diffu(G%IscB:G%IecB,G%jsc:G%jec,:), ddiffu) ; call cmp3("Accel_diag%diffv", ADp%diffv(G%isc:G%iec,G%JscB:G%JecB,:), ddiffv)
    call cmp3("Accel_diag%u_accel_bt", ADp%u_accel_bt(G%IscB:G%IecB,G%jsc:G%jec,:), dubt)
    call cmp3("Accel_diag%v_accel_bt", ADp%v_accel_bt(G%isc:G%iec,G%JscB:G%JecB,:), dvbt)
    call cmp3("MIS%pbce", MIS%pbce(G%isc:G%iec,G%jsc:G%jec,:), dpbce)
    call cmp3("MIS%u_av", MIS%u_av(G%IscB:G%IecB,G%jsc:G%jec,:), duav) ; call cmp3("MIS%v_av", MIS%v_av(G%isc:G%iec,G%JscB:G%JecB,:), dvav)
  end subroutine compare_diag_pointers

  subroutine near3(name, a, b, tol)
    character(len=*), intent(in) :: name ; real, intent(in) :: a(:,:,:), b(:,:,:), tol
    if (all(a == a) .and. maxval(abs(a - b)) <= tol) then
      print '(a,es10.3)', trim(name)//": max |diff| from the uninterrupted run = ", maxval(abs(a - b))
    else
      print '(a,es10.3)', "FAIL "//trim(name)//": max |diff| = ", maxval(abs(a - b)) ; nbad = nbad + 1
    endif
  end subroutine near3

  !> The shims keep the reference's cpu clocks: every step passed through the device clock and the transfer clock
  subroutine check_clocks()
    integer :: c, hits
    hits = 0
    do c = 1, stub_nclocks
      if (index(stub_clock_names(c), "Ocean dynamics on the device") > 0 .and. stub_clock_calls(c) > 0) hits = hits + 1
    enddo
    if (hits == 0) then ; print '(a)', "FAIL: the step never passed through its cpu clock" ; nbad = nbad + 1 ; endif
  end subroutine check_clocks

  !> advect_tracer and tracer_vertdiff through their shims (host arrays in and out)
  subroutine tracer_checks()
    type(tracer_advect_CS), pointer :: TA => NULL()
    type(tracer_registry_type), pointer :: Reg => NULL()
    real, allocatable, target :: tr1(:,:,:), tr2(:,:,:)
    real, allocatable :: uhr(:,:,:), vhr(:,:,:), ea(:,:,:), eb(:,:,:)
    integer :: i, j, k
    real :: lo, hi
    allocate(Reg) ; Reg%ntr = 2
    call al3(tr1, 0) ; call al3(tr2, 0) ; call al3(uhr, 1) ; call al3(vhr, 2)
    tr1 = 35.0
    do k = 1, nk ; do j = G%jsd, G%jed ; do i = G%isd, G%ied
      tr2(i,j,k) = 10.0 + 5.0 * sin(0.3 * i) * cos(0.2 * j) + 0.1 * k
    enddo ; enddo ; enddo
    lo = minval(tr2) ; hi = maxval(tr2)
    Reg%Tr(1)%t => tr1 ; Reg%Tr(2)%t => tr2 ; Reg%Tr(2)%advect_scheme = 2     ! ADVECT_PPM for the second one
    call stub_set_param(PF, "TRACER_ADVECTION_SCHEME", "PPM:H3")
    call tracer_advect_init(Time, G, US, PF, diag, TA)
    uhr = -1.0e30 ; vhr = -1.0e30
    call advect_tracer(h, uhtr, vhtr, OBC, dt * nsteps, G, GV, US, TA, Reg, uhr_out=uhr, vhr_out=vhr)
    if (maxval(abs(tr1(G%isc:G%iec,G%jsc:G%jec,:) - 35.0)) > 1.0e-11) then
      print '(a,es10.3)', "FAIL: a uniform tracer did not stay uniform: ", maxval(abs(tr1(G%isc:G%iec,G%jsc:G%jec,:) - 35.0)) ; nbad = nbad + 1
    endif
    if (minval(tr2(G%isc:G%iec,G%jsc:G%jec,:)) < lo - 1.0e-9 .or. maxval(tr2(G%isc:G%iec,G%jsc:G%jec,:)) > hi + 1.0e-9) then
      print '(a)', "FAIL: advect_tracer left the initial bounds" ; nbad = nbad + 1
    endif
    ! uhr_out / vhr_out are intent(out) in the reference: they must have been written (the sentinel is gone), and what the
    ! iteration could not use is no larger than what it was given
    if (any(uhr(G%IscB:G%IecB,G%jsc:G%jec,:) < -1.0e29) .or. any(vhr(G%isc:G%iec,G%JscB:G%JecB,:) < -1.0e29)) then
      print '(a)', "FAIL: uhr_out / vhr_out were not written" ; nbad = nbad + 1
    elseif (maxval(abs(uhr(G%IscB:G%IecB,G%jsc:G%jec,:))) > maxval(abs(uhtr)) + 1.0e-6) then
      print '(a)', "FAIL: leftover transports exceed the transports" ; nbad = nbad + 1
    else
      print '(a,es10.3)', "advect_tracer through the shim: uniform tracer exact, bounds kept, max |uhr_out| = ", maxval(abs(uhr(G%IscB:G%IecB,G%jsc:G%jec,:)))
    endif
    call al3(ea, 0) ; call al3(eb, 0)
    ea = 0.5 ; eb = 0.5 ; ea(:,:,1) = 0.0 ; eb(:,:,nk) = 0.0
    call tracer_vertdiff(h, ea, eb, dt, tr1, G, GV)
    if (maxval(abs(tr1(G%isc:G%iec,G%jsc:G%jec,:) - 35.0)) > 1.0e-11) then
      print '(a)', "FAIL: tracer_vertdiff changed a uniform tracer" ; nbad = nbad + 1
    else
      print '(a)', "tracer_vertdiff through the shim: uniform tracer exact"
    endif
    call tracer_advect_end(TA)
    deallocate(Reg)
  end subroutine tracer_checks

  !> CorAdCalc and btstep through their shims, first with plain host arrays (every argument crosses PCIe on every call), then with
  !! the same arrays handed over to the device (shim_resident_add): the calls must then move NOTHING -- counted -- and give the
  !! same bits.
  subroutine resident_submodule_checks()
    type(CoriolisAdv_CS) :: CorCS
    type(barotropic_CS) :: BTCS
    type(MOM_restart_CS) :: RCS2
    type(BT_cont_type), pointer :: no_BT_cont => NULL()
    type(accel_diag_ptrs), pointer :: no_ADp => NULL()
    real, dimension(:,:), pointer :: no2 => NULL()
    real, dimension(:,:,:), pointer :: no3 => NULL()
    real, allocatable, target :: CAu(:,:,:), CAv(:,:,:), CAu2(:,:,:), CAv2(:,:,:), bcu(:,:,:), bcv(:,:,:), pbce(:,:,:), vru(:,:,:), vrv(:,:,:)
    real, allocatable, target :: alu(:,:,:), alv(:,:,:), alu2(:,:,:), alv2(:,:,:), eta_in(:,:), eta_o(:,:), eta_o2(:,:), spv(:,:)
    real, allocatable, target :: uhb(:,:), vhb(:,:), uhb2(:,:), vhb2(:,:)
    logical :: cd
    integer :: k, rep
    integer(c_long_long) :: n_plain, n_res
    call al3(CAu, 1) ; call al3(CAv, 2) ; call al3(CAu2, 1) ; call al3(CAv2, 2) ; call al3(bcu, 1) ; call al3(bcv, 2) ; call al3(pbce, 0)
    call al3(vru, 1) ; call al3(vrv, 2) ; call al3(alu, 1) ; call al3(alv, 2) ; call al3(alu2, 1) ; call al3(alv2, 2)
    allocate(eta_in(G%isd:G%ied,G%jsd:G%jed), eta_o(G%isd:G%ied,G%jsd:G%jed), eta_o2(G%isd:G%ied,G%jsd:G%jed), spv(G%isd:G%ied,G%jsd:G%jed), source=0.0)
    allocate(uhb(G%IsdB:G%IedB,G%jsd:G%jed), uhb2(G%IsdB:G%IedB,G%jsd:G%jed), vhb(G%isd:G%ied,G%JsdB:G%JedB), vhb2(G%isd:G%ied,G%JsdB:G%JedB), source=0.0)
    eta_in = (sum(h, 3) - G%bathyT * GV%Z_to_H) * G%mask2dT
    do k = 1, nk
      bcu(:,:,k) = 1.0e-6 * u(:,:,k) ; bcv(:,:,k) = 1.0e-6 * v(:,:,k) ; pbce(:,:,k) = GV%g_Earth * GV%H_to_Z * (1.0 + 1.0e-3 * (k - 1))
      vru(:,:,k) = 0.9 * G%mask2dCu ; vrv(:,:,k) = 0.9 * G%mask2dCv
    enddo
    call CoriolisAdv_init(Time, G, GV, US, PF, diag, ADp, CorCS)
    call register_barotropic_restarts(HI, GV, US, PF, BTCS, RCS2)
    call barotropic_init(u, v, h, Time, G, GV, US, PF, diag, BTCS, RCS2, cd, no_BT_cont, OBC)
    ! ---- plain host arrays
    n_plain = shim_transfer_count(reset=.true.)
    call CorAdCalc(u, v, h, uh, vh, CAu, CAv, OBC, ADp, G, GV, US, CorCS, pbv)
    call btcalc(h, G, GV, BTCS, may_use_default=.true.)
    call set_dtbt(G, GV, US, BTCS, pbce=pbce)
    call bt_mass_source(h, eta_in, .true., G, GV, BTCS)
    call btstep(u, v, eta_in, dt, bcu, bcv, forces, pbce, eta_in, u, v, alu, alv, eta_o, uhb, vhb, G, GV, US, BTCS, vru, vrv, spv, &
                no_ADp, OBC, no_BT_cont, no2, no2, no2, no3, no3, no3, no3)
    n_plain = shim_transfer_count(reset=.true.)
    ! ---- the same arrays resident in HBM (results into a second set)
    call shim_resident_add(u, 1, nk) ; call shim_resident_add(v, 2, nk) ; call shim_resident_add(h, 0, nk)
    call shim_resident_add(uh, 1, nk) ; call shim_resident_add(vh, 2, nk) ; call shim_resident_add(CAu2, 1, nk) ; call shim_resident_add(CAv2, 2, nk)
    call shim_resident_add(eta_in, 0, 1) ; call shim_resident_add(bcu, 1, nk) ; call shim_resident_add(bcv, 2, nk)
    call shim_resident_add(taux, 1, 1) ; call shim_resident_add(tauy, 2, 1) ; call shim_resident_add(pbce, 0, nk)
    call shim_resident_add(vru, 1, nk) ; call shim_resident_add(vrv, 2, nk) ; call shim_resident_add(alu2, 1, nk) ; call shim_resident_add(alv2, 2, nk)
    call shim_resident_add(eta_o2, 0, 1) ; call shim_resident_add(uhb2, 1, 1) ; call shim_resident_add(vhb2, 2, 1)
    n_res = shim_transfer_count(reset=.true.)
    do rep = 1, 2
      call CorAdCalc(u, v, h, uh, vh, CAu2, CAv2, OBC, ADp, G, GV, US, CorCS, pbv)
      call btcalc(h, G, GV, BTCS, may_use_default=.true.)
      call bt_mass_source(h, eta_in, .true., G, GV, BTCS)
      call btstep(u, v, eta_in, dt, bcu, bcv, forces, pbce, eta_in, u, v, alu2, alv2, eta_o2, uhb2, vhb2, G, GV, US, BTCS, vru, vrv, spv, &
                  no_ADp, OBC, no_BT_cont, no2, no2, no2, no3, no3, no3, no3)
    enddo
    n_res = shim_transfer_count(reset=.true.)
    print '(a,i0,a,i0)', "sub-module shims: arrays across PCIe per CorAdCalc + btcalc + bt_mass_source + btstep: plain ", n_plain, ", resident ", n_res
    if (n_plain < 20) then ; print '(a)', "FAIL: the plain calls should have moved their arguments" ; nbad = nbad + 1 ; endif
    if (n_res /= 0) then ; print '(a)', "FAIL: calls on resident arrays moved data between host and device" ; nbad = nbad + 1 ; endif
    call shim_resident_sync_host(CAu2) ; call shim_resident_sync_host(CAv2) ; call shim_resident_sync_host(alu2) ; call shim_resident_sync_host(alv2)
    call shim_resident_sync_host(eta_o2) ; call shim_resident_sync_host(uhb2) ; call shim_resident_sync_host(vhb2)
    call cmp3("resident CorAdCalc CAu", CAu2(G%IscB:G%IecB,G%jsc:G%jec,:), CAu(G%IscB:G%IecB,G%jsc:G%jec,:))
    call cmp3("resident CorAdCalc CAv", CAv2(G%isc:G%iec,G%JscB:G%JecB,:), CAv(G%isc:G%iec,G%JscB:G%JecB,:))
    call cmp3("resident btstep accel_layer_u", alu2(G%IscB:G%IecB,G%jsc:G%jec,:), alu(G%IscB:G%IecB,G%jsc:G%jec,:))
    call cmp3("resident btstep accel_layer_v", alv2(G%isc:G%iec,G%JscB:G%JecB,:), alv(G%isc:G%iec,G%JscB:G%JecB,:))
    call cmp3("resident btstep eta_out", reshape(eta_o2(G%isc:G%iec,G%jsc:G%jec), (/ ni, nj, 1 /)), reshape(eta_o(G%isc:G%iec,G%jsc:G%jec), (/ ni, nj, 1 /)))
    call cmp3("resident btstep uhbtav", reshape(uhb2(G%IscB:G%IecB,G%jsc:G%jec), (/ ni + 1, nj, 1 /)), reshape(uhb(G%IscB:G%IecB,G%jsc:G%jec), (/ ni + 1, nj, 1 /)))
    call cmp3("resident btstep vhbtav", reshape(vhb2(G%isc:G%iec,G%JscB:G%JecB), (/ ni, nj + 1, 1 /)), reshape(vhb(G%isc:G%iec,G%JscB:G%JecB), (/ ni, nj + 1, 1 /)))
    if (maxval(abs(alu)) == 0.0 .or. maxval(abs(CAu)) == 0.0) then ; print '(a)', "FAIL: the shims returned zeros" ; nbad = nbad + 1 ; endif
    ! the host takes its arrays back (the state ones without a download: nothing wrote them)
    call shim_resident_drop(u, download=.false.) ; call shim_resident_drop(v, download=.false.) ; call shim_resident_drop(h, download=.false.)
    call shim_resident_drop(uh, download=.false.) ; call shim_resident_drop(vh, download=.false.)
    call shim_resident_drop(taux, download=.false.) ; call shim_resident_drop(tauy, download=.false.)
    call shim_resident_drop(CAu2) ; call shim_resident_drop(CAv2) ; call shim_resident_drop(alu2) ; call shim_resident_drop(alv2)
    call shim_resident_drop(eta_in, download=.false.) ; call shim_resident_drop(bcu, download=.false.) ; call shim_resident_drop(bcv, download=.false.)
    call shim_resident_drop(pbce, download=.false.) ; call shim_resident_drop(vru, download=.false.) ; call shim_resident_drop(vrv, download=.false.)
    call shim_resident_drop(eta_o2) ; call shim_resident_drop(uhb2) ; call shim_resident_drop(vhb2)
    call barotropic_end(BTCS) ; call CoriolisAdv_end(CorCS)
  end subroutine resident_submodule_checks

  !> horizontal_viscosity through MOM_hor_visc and one ALE step (ALE_regrid, ALE_remap_tracers, ALE_remap_set_h_vel x 2,
  !! ALE_remap_velocities) through MOM_ALE on the INITIAL state, host arrays in and out, the parameters from the MOM_input table:
  !! every result equals the oracle's bit for bit.
  subroutine hor_visc_and_ALE_checks()
    type(hor_visc_CS) :: HV
    type(ALE_CS), pointer :: ALE => NULL()
    type(tracer_registry_type), pointer :: Reg => NULL()
    real, allocatable, target :: du(:,:,:), dv(:,:,:), hh(:,:,:), hn(:,:,:), dz(:,:,:), Tr(:,:,:), ur(:,:,:), vr(:,:,:)
    real, allocatable :: huo(:,:,:), hvo(:,:,:), hun(:,:,:), hvn(:,:,:)
    call hor_visc_init(Time, G, GV, US, PF, diag, HV, ADp=ADp)
    if (hor_visc_vel_stencil(HV) /= 2) then ; print '(a)', "FAIL: hor_visc_vel_stencil" ; nbad = nbad + 1 ; endif
    call al3(du, 1) ; call al3(dv, 2) ; call al3(hh, 0) ; hh = h0
    call horizontal_viscosity(u0, v0, hh, uh, vh, du, dv, MEKE, VarMix, G, GV, US, HV, tv, dt)
    call cmp3("horizontal_viscosity (MOM_hor_visc) diffu", du(G%IscB:G%IecB,G%jsc:G%jec,:), xdiffu)
    call cmp3("horizontal_viscosity (MOM_hor_visc) diffv", dv(G%isc:G%iec,G%JscB:G%JecB,:), xdiffv)
    if (maxval(abs(xdiffu)) <= 0.0) then ; print '(a)', "FAIL: the lateral friction of the case is zero" ; nbad = nbad + 1 ; endif
    call hor_visc_end(HV)

    call stub_set_param(PF, "REGRIDDING_COORDINATE_MODE", "ZSTAR") ; call stub_set_param(PF, "REMAPPING_SCHEME", "PLM")
    call ALE_init(PF, GV, US, max_depth, ALE)
    if (.not.ALE_remap_init_conds(ALE)) then ; print '(a)', "FAIL: ALE_remap_init_conds" ; nbad = nbad + 1 ; endif
    call ALE_update_regrid_weights(dt, ALE)
    call al3(hn, 0) ; allocate(dz(G%isd:G%ied,G%jsd:G%jed,nk+1), source=0.0)
    call ALE_regrid(G, GV, US, h0, hn, dz, tv, ALE)
    call cmp3("ALE_regrid (MOM_ALE) h_new", hn(G%isc:G%iec,G%jsc:G%jec,:), xhn)
    call cmp3("ALE_regrid (MOM_ALE) dzRegrid", dz(G%isc:G%iec,G%jsc:G%jec,:), xdz)
    allocate(Reg) ; Reg%ntr = 1 ; call al3(Tr, 0) ; Tr = T0 ; Reg%Tr(1)%t => Tr
    call ALE_remap_tracers(ALE, G, GV, h0, hn, Reg)
    call cmp3("ALE_remap_tracers (MOM_ALE)", Tr(G%isc:G%iec,G%jsc:G%jec,:), xT)
    if (maxval(abs(xT - T0(G%isc:G%iec,G%jsc:G%jec,:))) <= 0.0) then ; print '(a)', "FAIL: the remapping of the case moves nothing" ; nbad = nbad + 1 ; endif
    call al3(huo, 1) ; call al3(hvo, 2) ; call al3(hun, 1) ; call al3(hvn, 2) ; call al3(ur, 1) ; call al3(vr, 2)
    call ALE_remap_set_h_vel(ALE, G, GV, h0, huo, hvo, OBC) ; call ALE_remap_set_h_vel(ALE, G, GV, hn, hun, hvn, OBC)
    ur = u0 ; vr = v0
    call ALE_remap_velocities(ALE, G, GV, huo, hvo, hun, hvn, ur, vr)
    call cmp3("ALE_remap_velocities (MOM_ALE) u", ur(G%IscB:G%IecB,G%jsc:G%jec,:), xur)
    call cmp3("ALE_remap_velocities (MOM_ALE) v", vr(G%isc:G%iec,G%JscB:G%JecB,:), xvr)
    ! ---- the same remapping on arrays the host has handed over (shim_resident_add): nothing crosses PCIe, the same bits
    block
      real, allocatable, target :: hr(:,:,:), hnr(:,:,:), Trr(:,:,:), urr(:,:,:), vrr(:,:,:), huor(:,:,:), hvor(:,:,:), hunr(:,:,:), hvnr(:,:,:)
      integer(c_long_long) :: n_res
      call al3(hr, 0) ; call al3(hnr, 0) ; call al3(Trr, 0) ; call al3(urr, 1) ; call al3(vrr, 2)
      call al3(huor, 1) ; call al3(hvor, 2) ; call al3(hunr, 1) ; call al3(hvnr, 2)
      hr = h0 ; hnr = hn ; Trr = T0 ; urr = u0 ; vrr = v0
      call shim_resident_add(hr, 0, nk) ; call shim_resident_add(hnr, 0, nk) ; call shim_resident_add(Trr, 0, nk)
      call shim_resident_add(urr, 1, nk) ; call shim_resident_add(vrr, 2, nk) ; call shim_resident_add(huor, 1, nk)
      call shim_resident_add(hvor, 2, nk) ; call shim_resident_add(hunr, 1, nk) ; call shim_resident_add(hvnr, 2, nk)
      n_res = shim_transfer_count(reset=.true.)
      Reg%Tr(1)%t => Trr
      call ALE_remap_tracers(ALE, G, GV, hr, hnr, Reg)
      call ALE_remap_set_h_vel(ALE, G, GV, hr, huor, hvor, OBC) ; call ALE_remap_set_h_vel(ALE, G, GV, hnr, hunr, hvnr, OBC)
      call ALE_remap_velocities(ALE, G, GV, huor, hvor, hunr, hvnr, urr, vrr)
      n_res = shim_transfer_count(reset=.true.)
      print '(a,i0)', "MOM_ALE on resident arrays: arrays across PCIe per ALE_remap_tracers + 2 x ALE_remap_set_h_vel + ALE_remap_velocities: ", n_res
      if (n_res /= 0) then ; print '(a)', "FAIL: the resident calls of MOM_ALE moved arrays" ; nbad = nbad + 1 ; endif
      call shim_resident_sync_host(Trr) ; call shim_resident_sync_host(urr) ; call shim_resident_sync_host(vrr)
      call cmp3("resident ALE_remap_tracers", Trr(G%isc:G%iec,G%jsc:G%jec,:), xT)
      call cmp3("resident ALE_remap_velocities u", urr(G%IscB:G%IecB,G%jsc:G%jec,:), xur)
      call cmp3("resident ALE_remap_velocities v", vrr(G%isc:G%iec,G%JscB:G%JecB,:), xvr)
      call shim_resident_drop(hr, download=.false.) ; call shim_resident_drop(hnr, download=.false.) ; call shim_resident_drop(Trr, download=.false.)
      call shim_resident_drop(urr, download=.false.) ; call shim_resident_drop(vrr, download=.false.) ; call shim_resident_drop(huor, download=.false.)
      call shim_resident_drop(hvor, download=.false.) ; call shim_resident_drop(hunr, download=.false.) ; call shim_resident_drop(hvnr, download=.false.)
    end block
    call ALE_end(ALE)
    if (associated(ALE)) then ; print '(a)', "FAIL: ALE_end left CS associated" ; nbad = nbad + 1 ; endif
    deallocate(Reg)
    call ALE_rho_checks()
  end subroutine hor_visc_and_ALE_checks

  !> The density coordinate through MOM_ALE: ALE_init reads REGRIDDING_COORDINATE_MODE = RHO (target densities from GV%Rlay, the
  !! equation of state from the table), pre_ALE_adjustments makes the columns statically stable, ALE_regrid builds the grid: five
  !! arrays equal to the oracle's bit for bit.
  subroutine ALE_rho_checks()
    type(ALE_CS), pointer :: ALE => NULL()
    type(tracer_registry_type), pointer :: Reg => NULL()
    type(thermo_var_ptrs) :: tv2
    real, allocatable, target :: hh(:,:,:), TT(:,:,:), SS(:,:,:), hn(:,:,:), dz(:,:,:)
    call stub_set_param(PF, "ENABLE_THERMODYNAMICS", "True") ; call stub_set_param(PF, "EQN_OF_STATE", "LINEAR")
    call stub_set_param(PF, "REGRIDDING_COORDINATE_MODE", "RHO") ; call stub_set_param(PF, "REGRIDDING_ANSWER_DATE", "20190101")
    call ALE_init(PF, GV, US, max_depth, ALE)
    call ALE_update_regrid_weights(dt, ALE)
    call al3(hh, 0) ; call al3(TT, 0) ; call al3(SS, 0) ; hh = h0 ; TT = Trho ; SS = Srho
    tv2%T => TT ; tv2%S => SS
    call pre_ALE_adjustments(G, GV, US, hh, tv2, Reg, ALE)
    call cmp3("pre_ALE_adjustments (MOM_ALE, RHO) h", hh(G%isc:G%iec,G%jsc:G%jec,:), yh)
    call cmp3("pre_ALE_adjustments (MOM_ALE, RHO) T", TT(G%isc:G%iec,G%jsc:G%jec,:), yT)
    call cmp3("pre_ALE_adjustments (MOM_ALE, RHO) S", SS(G%isc:G%iec,G%jsc:G%jec,:), yS)
    if (maxval(abs(yT - Trho(G%isc:G%iec,G%jsc:G%jec,:))) <= 0.0) then ; print '(a)', "FAIL: no column of the case is statically unstable" ; nbad = nbad + 1 ; endif
    call al3(hn, 0) ; allocate(dz(G%isd:G%ied,G%jsd:G%jed,nk+1), source=0.0)
    call ALE_regrid(G, GV, US, hh, hn, dz, tv2, ALE)
    call cmp3("ALE_regrid (MOM_ALE, RHO) h_new", hn(G%isc:G%iec,G%jsc:G%jec,:), yhn)
    call cmp3("ALE_regrid (MOM_ALE, RHO) dzRegrid", dz(G%isc:G%iec,G%jsc:G%jec,:), ydz)
    call ALE_end(ALE)
    call stub_set_param(PF, "ENABLE_THERMODYNAMICS", "False") ; call stub_set_param(PF, "REGRIDDING_COORDINATE_MODE", "ZSTAR")
  end subroutine ALE_rho_checks

  !> The debugging checksums through MOM_checksums on the INITIAL state, written to standard output (logunit = 6): tests/test_fortran_gpu.py
  !! compares the lines with the ones the Python host formats from the same device routine (whose numbers are held to the oracle).
  subroutine checksum_checks()
    call MOM_checksums_init(PF)
    call hchksum(h0, "h0 [MOM_checksums]", HI, haloshift=1, logunit=6)
    call uvchksum("uv0 [MOM_checksums]", u0, v0, HI, haloshift=1, symmetric=.true., logunit=6)
    call Bchksum(G%CoriolisBu, "f [MOM_checksums]", HI, haloshift=0, symmetric=.true., logunit=6)
    call hchksum(h0, "h0 x2 [MOM_checksums]", HI, unscale=2.0, logunit=6)
    call hchksum_pair("hT [MOM_checksums]", h0, T0, HI, haloshift=2, omit_corners=.true., logunit=6)
  end subroutine checksum_checks

  subroutine rd2(a, stg)
    real, allocatable, intent(inout) :: a(:,:) ; integer, intent(in) :: stg
    integer :: s
    if (stg == 0) allocate(a(G%isd:G%ied,G%jsd:G%jed))
    if (stg == 1) allocate(a(G%IsdB:G%IedB,G%jsd:G%jed))
    if (stg == 2) allocate(a(G%isd:G%ied,G%JsdB:G%JedB))
    if (stg == 3) allocate(a(G%IsdB:G%IedB,G%JsdB:G%JedB))
    read(un) s
    if (s /= stg) error stop "drive_shims: staggering of an array in the case file is not what the driver expects"
    read(un) a
  end subroutine rd2

  subroutine rd3(a, stg, nl)
    real, allocatable, intent(inout) :: a(:,:,:) ; integer, intent(in) :: stg, nl
    integer :: s
    if (stg == 0) allocate(a(G%isd:G%ied,G%jsd:G%jed,nl))
    if (stg == 1) allocate(a(G%IsdB:G%IedB,G%jsd:G%jed,nl))
    if (stg == 2) allocate(a(G%isd:G%ied,G%JsdB:G%JedB,nl))
    read(un) s
    if (s /= stg) error stop "drive_shims: staggering of an array in the case file is not what the driver expects"
    read(un) a
  end subroutine rd3

  subroutine al3(a, stg)
    real, allocatable, intent(inout) :: a(:,:,:) ; integer, intent(in) :: stg
    if (stg == 0) allocate(a(G%isd:G%ied,G%jsd:G%jed,nk), source=0.0)
    if (stg == 1) allocate(a(G%IsdB:G%IedB,G%jsd:G%jed,nk), source=0.0)
    if (stg == 2) allocate(a(G%isd:G%ied,G%JsdB:G%JedB,nk), source=0.0)
  end subroutine al3
end program drive_shims
