!> ISO_C_BINDING interfaces to the MI355X-native dynamical core (include/mom6x.h).
!!
!! This module is what a MOM6 build adds to bind the reference's Fortran host to the HIP library
!! (libmom6x.so).  It follows the only in-tree precedent for C interop, src/framework/posix.F90:52-229.
!! Every derived type mirrors a C struct of include/mom6x.h field for field (bind(C)); every interface
!! names the C entry point and, in its comment, the reference procedure it stands in for.
module mom6x_c_api
  use, intrinsic :: iso_c_binding
  implicit none ; private

  public :: mom6x_dims, mom6x_vgrid, mom6x_continuity_params, mom6x_BT_cont, mom6x_barotropic_params
  public :: mom6x_coriolis_params, mom6x_pgf_params, mom6x_eos_params, mom6x_rk2_params, mom6x_rk2_hooks
  public :: mom6x_PressureForce_set_tv, mom6x_vertvisc_params, mom6x_vertvisc_init, mom6x_vertvisc_set_visc, mom6x_vertvisc_coef
  public :: mom6x_hor_visc_params, mom6x_hor_visc_init, mom6x_horizontal_viscosity, mom6x_vertvisc_set_direct_stress
  public :: mom6x_remapping_params, mom6x_ALE_remap_tracers, mom6x_ALE_remap_set_h_vel, mom6x_ALE_remap_velocities
  public :: mom6x_ALE_remap_velocities_conserve_ke, mom6x_ALE_remap_velocities_from_h, mom6x_comm_overlap_btstep
  public :: mom6x_remapping_core_h, mom6x_regrid_zstar_params, mom6x_ALE_regrid_zstar
  public :: mom6x_regrid_rho_params, mom6x_ALE_regrid_rho, mom6x_ALE_regrid_hycom1, mom6x_ALE_convective_adjustment
  public :: mom6x_chksum_result, mom6x_sum_output_params, mom6x_energy_sums, mom6x_reproducing_sum_3d, mom6x_reproducing_sum_2d
  public :: mom6x_remap_dyn_split_RK2_aux_vars
  public :: mom6x_chksum, mom6x_field_chksum, mom6x_sum_output_init, mom6x_depth_list, mom6x_write_energy, mom6x_barotropic_dtbt
  public :: mom6x_btstep_warnings
  public :: mom6x_dims_init, mom6x_ctx_create, mom6x_ctx_destroy, mom6x_ctx_sync, mom6x_last_error
  public :: mom6x_dev_alloc, mom6x_dev_free, mom6x_dev_copy, mom6x_upload, mom6x_download, mom6x_struct_size
  public :: mom6x_continuity_init, mom6x_continuity_PPM, mom6x_barotropic_init, mom6x_btcalc, mom6x_btcalc_strict
  public :: mom6x_bt_mass_source, mom6x_set_dtbt, mom6x_set_dtbt_pbce, mom6x_set_dtbt_pbce_eta, mom6x_btstep
  public :: mom6x_CoriolisAdv_init, mom6x_CorAdCalc, mom6x_PressureForce_init, mom6x_PressureForce
  public :: mom6x_vertvisc_set_coef, mom6x_vertvisc, mom6x_vertvisc_remnant
  public :: mom6x_initialize_dyn_split_RK2, mom6x_dyn_split_RK2_new_run, mom6x_dyn_split_RK2_restart_fills, mom6x_rk2_field, mom6x_rk2_set_CAu_pred_stored
  public :: mom6x_step_dyn_split_RK2, mom6x_comm_unique_id, mom6x_comm_init, mom6x_pass_fields
  public :: mom6x_transport, mom6x_comm_set_transport
  public :: mom6x_abi_version, mom6x_device_count, MOM6X_ABI_BUILT_FOR, mom6x_barotropic_field
  public :: MOM6X_RK2_HAVE_ETA, MOM6X_RK2_HAVE_DIFFU, MOM6X_RK2_HAVE_U2, MOM6X_RK2_HAVE_CAU, MOM6X_RK2_HAVE_UH, MOM6X_RK2_HAVE_H2
  public :: mom6x_tracer_advect_init, mom6x_advect_tracer, mom6x_triDiagTS, mom6x_triDiagTS_Eulerian
  public :: mom6x_tracer_vertdiff, mom6x_tracer_vertdiff_Eulerian, mom6x_diabatic_is_trivial
  public :: mom6x_tracer_vertdiff_sink, mom6x_tracer_vertdiff_Eulerian_sink

  !> include/mom6x.h MOM6X_ABI_VERSION this module mirrors; a host compares it with mom6x_abi_version() at start-up
  integer(c_int), parameter :: MOM6X_ABI_BUILT_FOR = 6
  !> mom6x_dyn_split_RK2_restart_fills: the restart variables the host has uploaded (include/mom6x.h MOM6X_RK2_HAVE_*)
  integer(c_int), parameter :: MOM6X_RK2_HAVE_ETA = 1, MOM6X_RK2_HAVE_DIFFU = 2, MOM6X_RK2_HAVE_U2 = 4, MOM6X_RK2_HAVE_CAU = 8, &
                               MOM6X_RK2_HAVE_UH = 16, MOM6X_RK2_HAVE_H2 = 32

  !> mom6x_dims: hor_index_type extents (MOM_hor_index.F90:14-44) + the device layout
  type, bind(C) :: mom6x_dims
    integer(c_int) :: ni, nj, nk, halo, ioff, joff, pitch, slab
    integer(c_int) :: i_glob0, j_glob0, ni_glob, nj_glob, reentrant_x, reentrant_y
  end type mom6x_dims

  !> mom6x_vgrid: the scalars of verticalGrid_type (MOM_verticalGrid.F90:25-90)
  type, bind(C) :: mom6x_vgrid
    real(c_double) :: g_Earth, Rho0, Angstrom_H, H_subroundoff, dZ_subroundoff
    real(c_double) :: H_to_Z, Z_to_H, H_to_RZ, RZ_to_H
    integer(c_int) :: Boussinesq
  end type mom6x_vgrid

  !> continuity_PPM_CS (MOM_continuity_PPM.F90:35-68)
  type, bind(C) :: mom6x_continuity_params
    integer(c_int) :: upwind_1st, monotonic, simple_2nd
    real(c_double) :: tol_eta, tol_vel, CFL_limit_adjust
    integer(c_int) :: aggress_adjust, vol_CFL, better_iter, use_visc_rem_max, marginal_faces
    integer(c_int) :: sum_order   !< 0: the reference's sequential k sums (bit-identical); 1: 16-lane tree (default of the device); 2: the tree + fused multiply-adds at fixed sites
  end type mom6x_continuity_params

  !> BT_cont_type (MOM_variables.F90:315-350): device pointers
  type, bind(C) :: mom6x_BT_cont
    type(c_ptr) :: FA_u_EE, FA_u_E0, FA_u_W0, FA_u_WW, uBT_WW, uBT_EE
    type(c_ptr) :: FA_v_NN, FA_v_N0, FA_v_S0, FA_v_SS, vBT_SS, vBT_NN
    type(c_ptr) :: h_u, h_v
  end type mom6x_BT_cont

  !> barotropic_CS parameters (MOM_barotropic.F90:108-330)
  type, bind(C) :: mom6x_barotropic_params
    real(c_double) :: bebt, dtbt, dt_bt_filter
    integer(c_int) :: BT_project_velocity, Sadourny, strong_drag, wt_uv_bug, use_old_coriolis_bracket_bug
    integer(c_int) :: visc_rem_u_uh0, clip_velocity
    real(c_double) :: CFL_trunc, vel_underflow, G_extra, BT_Coriolis_scale, maxCFL_BT_cont
    integer(c_int) :: bound_BT_corr, BT_cont_bounds
    real(c_double) :: dtbt_fraction, Z_ref
    integer(c_int) :: use_wide_halos, BTHALO, min_stencil   !< BT_USE_WIDE_HALOS (T), BTHALO (0), BT_WIDE_HALO_MIN_STENCIL (0)
    integer(c_int) :: nonlinear_continuity, nonlin_cont_update_period   !< NONLINEAR_BT_CONTINUITY (F), NONLIN_BT_CONT_UPDATE_PERIOD (1): read by btstep without a BT_cont_type
    integer(c_int) :: bt_thick_scheme   !< BT_THICK_SCHEME: 0 FROM_BT_CONT, 1 HYBRID, 2 HARMONIC, 3 ARITHMETIC (MOM6X_BT_THICK_*)
    real(c_double) :: maxvel            !< MAXVEL (3e8): eta_cor_bound of BOUND_BT_CORRECTION without the BT_cont bounds
  end type mom6x_barotropic_params

  type, bind(C) :: mom6x_coriolis_params   !< CoriolisAdv_CS (MOM_CoriolisAdv.F90:29-100)
    integer(c_int) :: Coriolis_Scheme, KE_Scheme, bound_Coriolis, no_slip, Coriolis_En_Dis, PV_Adv_Scheme
    real(c_double) :: F_eff_max_blend, wt_lin_blend
  end type mom6x_coriolis_params

  type, bind(C) :: mom6x_pgf_params        !< PressureForce_FV_CS (MOM_PressureForce_FV.F90:40-110)
    real(c_double) :: rho_ref
    integer(c_int) :: rho_ref_bug
    real(c_double) :: Z_ref
  end type mom6x_pgf_params

  type, bind(C) :: mom6x_vertvisc_params   !< vertvisc_CS (MOM_vert_friction.F90:39-180)
    real(c_double) :: Kv, Kvml_invZ2, Hmix, Hbbl, harm_BL_val, Kv_extra_bbl
    integer(c_int) :: harmonic_visc, bottomdraglaw, answer_date
  end type mom6x_vertvisc_params

  type, bind(C) :: mom6x_hor_visc_params   !< hor_visc_CS (MOM_hor_visc.F90:36-259), the members the device path reads
    integer(c_int) :: Laplacian, biharmonic
    real(c_double) :: Kh, Kh_bg_min, Kh_vel_scale
    integer(c_int) :: Smagorinsky_Kh
    real(c_double) :: Smag_Lap_const
    integer(c_int) :: bound_Kh, better_bound_Kh, add_LES_viscosity
    real(c_double) :: Ah, Ah_vel_scale, Ah_time_scale
    integer(c_int) :: Smagorinsky_Ah
    real(c_double) :: Smag_bi_const
    integer(c_int) :: bound_Ah, better_bound_Ah, bound_Coriolis
    real(c_double) :: bound_Cor_vel
    integer(c_int) :: use_land_mask
    real(c_double) :: bound_coef
    integer(c_int) :: no_slip, backscatter_underbound
    real(c_double) :: dt
    integer(c_int) :: Leith_Kh
    real(c_double) :: Leith_Lap_const
    integer(c_int) :: Leith_Ah
    real(c_double) :: Leith_bi_const
    integer(c_int) :: modified_Leith, use_beta_in_Leith
  end type mom6x_hor_visc_params

  type, bind(C) :: mom6x_remapping_params   !< remapping_CS (MOM_remapping.F90:47-84), the members the device path reads
    integer(c_int) :: scheme                !< 0 PCM, 2 PLM, 4 PPM_H4, 5 PPM_IH4 (the module's REMAPPING_* parameters :86-96)
    integer(c_int) :: boundary_extrapolation, force_bounds_in_subcell, force_bounds_in_target
    integer(c_int) :: om4_remap_via_sub_cells, answer_date
    real(c_double) :: h_neglect, h_neglect_edge
  end type mom6x_remapping_params

  type, bind(C) :: mom6x_regrid_zstar_params   !< the members of regridding_CS (MOM_regridding.F90:40-140) the z* branch reads
    real(c_double) :: min_thickness, old_grid_weight, depth_of_time_filter_shallow, depth_of_time_filter_deep, Z_ref
  end type mom6x_regrid_zstar_params

  !> what REGRIDDING_RHO / REGRIDDING_HYCOM1 read of regridding_CS on top of the z* members (INTERPOLATION_SCHEME: 0 P1M_H2,
  !! 3 PLM, 5 PPM_H4)
  type, bind(C) :: mom6x_regrid_rho_params
    type(mom6x_regrid_zstar_params) :: f
    integer(c_int) :: interp_scheme, boundary_extrapolation
    real(c_double) :: ref_pressure, compressibility_fraction
    integer(c_int) :: integrate_downward_for_e
  end type mom6x_regrid_rho_params

  !> a halo transport of the host's own (nine c_funptr with RCCL's meaning, see include/mom6x.h), e.g. over GPU-aware MPI
  type, bind(C) :: mom6x_transport
    type(c_funptr) :: get_unique_id, comm_init_rank, comm_destroy, send, recv, group_start, group_end, all_reduce, error_string
  end type mom6x_transport

  type, bind(C) :: mom6x_chksum_result         !< the numbers of the two lines of chksum_{h,u,v,B}_{2d,3d} (MOM_checksums.F90)
    real(c_double) :: mean, amin, amax
    integer(c_int) :: bc0, bc(4), nbc, bc_kind
  end type mom6x_chksum_result

  type, bind(C) :: mom6x_sum_output_params     !< the members of Sum_output_CS the sums of write_energy read
    integer(c_int) :: do_APE_calc, use_temperature
    real(c_double) :: dt_in_T, D_list_min_inc, Z_ref, C_p
  end type mom6x_sum_output_params

  type, bind(C) :: mom6x_energy_sums           !< write_energy (MOM_sum_output.F90:321): the global sums
    real(c_double) :: mass_tot, KE_tot, PE_tot, max_CFL(2)
    integer(c_int64_t) :: mass_EFP(6), salt_EFP(6), heat_EFP(6)
  end type mom6x_energy_sums

  type, bind(C) :: mom6x_eos_params        !< tv%eqn_of_state (MOM_EOS.F90:99-150) + EOS-only switches of PressureForce_FV_CS
    integer(c_int) :: form                 !< 1 EOS_LINEAR, 2 EOS_WRIGHT, 3 EOS_WRIGHT_FULL, 4 EOS_WRIGHT_REDUCED, 5 EOS_UNESCO, 6 EOS_ROQUET_RHO, 7 EOS_JACKETT06, 8 EOS_ROQUET_SPV
    real(c_double) :: Rho_T0_S0, dRho_dT, dRho_dS, dRho_dp
    integer(c_int) :: MassWghtInterp, use_SSH_in_Z0p
    integer(c_int) :: Recon_Scheme, boundary_extrap, MassWghtInterpVanOnly   !< ALE: PLM reconstruction of T, S for the pressure force
    real(c_double) :: h_nonvanished
    integer(c_int) :: EOS_quadrature       !< EOS_QUADRATURE: int_density_dz_generic_pcm instead of the analytic integrals
  end type mom6x_eos_params

  type, bind(C) :: mom6x_rk2_params        !< MOM_dyn_split_RK2_CS (MOM_dynamics_split_RK2.F90:85-273)
    real(c_double) :: be, begw
    integer(c_int) :: split_bottom_stress, BT_use_layer_fluxes, store_CAu, visc_rem_dt_bug, remap_aux
    integer(c_int) :: no_BT_cont = 0   !< 1: USE_BT_CONT_TYPE = False
  end type mom6x_rk2_params

  type, bind(C) :: mom6x_rk2_hooks         !< host callbacks for the un-ported callees (SURVEY 8f)
    type(c_ptr)    :: user
    type(c_funptr) :: vertvisc_coef
    type(c_funptr) :: horizontal_viscosity
  end type mom6x_rk2_hooks

  interface
    integer(c_int) function mom6x_abi_version() bind(C, name="mom6x_abi_version")
      import :: c_int
    end function
    integer(c_int) function mom6x_device_count() bind(C, name="mom6x_device_count")
      import :: c_int
    end function
    !> ubtav (0), vbtav (1), ... of barotropic_CS: the arrays register_barotropic_restarts (MOM_barotropic.F90:6253) registers
    type(c_ptr) function mom6x_barotropic_field(ctx, which) bind(C, name="mom6x_barotropic_field")
      import :: c_ptr, c_int ; type(c_ptr), value :: ctx ; integer(c_int), value :: which
    end function
    integer(c_int) function mom6x_struct_size(which) bind(C, name="mom6x_struct_size")
      import :: c_int ; integer(c_int), value :: which
    end function
    integer(c_int) function mom6x_dims_init(d, ni, nj, nk, halo) bind(C, name="mom6x_dims_init")
      import :: c_int, mom6x_dims
      type(mom6x_dims), intent(out) :: d ; integer(c_int), value :: ni, nj, nk, halo
    end function
    !> replaces the `G`, `GV` arguments of every routine: metrics are uploaded once
    integer(c_int) function mom6x_ctx_create(ctx, dims, device, metrics_host, GV, first_direction) &
        bind(C, name="mom6x_ctx_create")
      import :: c_int, c_ptr, c_double, mom6x_dims, mom6x_vgrid
      type(c_ptr), intent(out) :: ctx ; type(mom6x_dims), intent(in) :: dims ; integer(c_int), value :: device
      real(c_double), intent(in) :: metrics_host(*) ; type(mom6x_vgrid), intent(in) :: GV
      integer(c_int), value :: first_direction
    end function
    integer(c_int) function mom6x_ctx_destroy(ctx) bind(C, name="mom6x_ctx_destroy")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx
    end function
    integer(c_int) function mom6x_ctx_sync(ctx) bind(C, name="mom6x_ctx_sync")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx
    end function
    type(c_ptr) function mom6x_last_error() bind(C, name="mom6x_last_error")
      import :: c_ptr
    end function
    integer(c_int) function mom6x_dev_alloc(ctx, p, n) bind(C, name="mom6x_dev_alloc")
      import :: c_int, c_ptr, c_size_t ; type(c_ptr), value :: ctx ; type(c_ptr), intent(out) :: p
      integer(c_size_t), value :: n
    end function
    integer(c_int) function mom6x_dev_free(ctx, p) bind(C, name="mom6x_dev_free")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx, p
    end function
    integer(c_int) function mom6x_dev_copy(ctx, dst, src, n) bind(C, name="mom6x_dev_copy")
      import :: c_int, c_ptr, c_size_t ; type(c_ptr), value :: ctx, dst, src ; integer(c_size_t), value :: n
    end function
    !> Fortran array with MOM6 symmetric-memory extents -> pitched device array (stagger 0 h, 1 u, 2 v, 3 q)
    integer(c_int) function mom6x_upload(ctx, dev, host_f, stagger, nk) bind(C, name="mom6x_upload")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, dev ; real(c_double), intent(in) :: host_f(*)
      integer(c_int), value :: stagger, nk
    end function
    integer(c_int) function mom6x_download(ctx, host_f, dev, stagger, nk) bind(C, name="mom6x_download")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, dev ; real(c_double), intent(inout) :: host_f(*)
      integer(c_int), value :: stagger, nk
    end function

    !> continuity_PPM_init, MOM_continuity_PPM.F90:2674
    integer(c_int) function mom6x_continuity_init(ctx, p) bind(C, name="mom6x_continuity_init")
      import :: c_int, c_ptr, mom6x_continuity_params ; type(c_ptr), value :: ctx
      type(mom6x_continuity_params), intent(in) :: p
    end function
    !> continuity_PPM, MOM_continuity_PPM.F90:86 (optional arguments: c_null_ptr when absent)
    integer(c_int) function mom6x_continuity_PPM(ctx, u, v, hin, h, uh, vh, dt, uhbt, vhbt, visc_rem_u, visc_rem_v, &
        u_cor, v_cor, BT_cont, du_cor, dv_cor) bind(C, name="mom6x_continuity_PPM")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, u, v, hin, h, uh, vh ; real(c_double), value :: dt
      type(c_ptr), value :: uhbt, vhbt, visc_rem_u, visc_rem_v, u_cor, v_cor, BT_cont, du_cor, dv_cor
    end function

    !> barotropic_init, MOM_barotropic.F90:5301
    integer(c_int) function mom6x_barotropic_init(ctx, p) bind(C, name="mom6x_barotropic_init")
      import :: c_int, c_ptr, mom6x_barotropic_params ; type(c_ptr), value :: ctx
      type(mom6x_barotropic_params), intent(in) :: p
    end function
    !> btcalc, MOM_barotropic.F90:4360
    integer(c_int) function mom6x_btcalc_strict(ctx, h, h_u, h_v) bind(C, name="mom6x_btcalc_strict")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx, h, h_u, h_v
    end function
    integer(c_int) function mom6x_btcalc(ctx, h, h_u, h_v) bind(C, name="mom6x_btcalc")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx, h, h_u, h_v
    end function
    !> bt_mass_source, MOM_barotropic.F90:5243
    integer(c_int) function mom6x_bt_mass_source(ctx, h, eta, set_cor) bind(C, name="mom6x_bt_mass_source")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx, h, eta ; integer(c_int), value :: set_cor
    end function
    !> set_dtbt, MOM_barotropic.F90:3509
    integer(c_int) function mom6x_set_dtbt(ctx, pbce, gtot_est, SSH_add, dtbt_out) bind(C, name="mom6x_set_dtbt")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, pbce ; real(c_double), value :: gtot_est, SSH_add
      real(c_double), intent(out) :: dtbt_out
    end function
    integer(c_int) function mom6x_set_dtbt_pbce_eta(ctx, pbce, eta, SSH_add, dtbt_out) bind(C, name="mom6x_set_dtbt_pbce_eta")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, pbce, eta ; real(c_double), value :: SSH_add
      real(c_double), intent(out) :: dtbt_out
    end function
    integer(c_int) function mom6x_set_dtbt_pbce(ctx, pbce, dtbt_out) bind(C, name="mom6x_set_dtbt_pbce")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, pbce ; real(c_double), intent(out) :: dtbt_out
    end function
    !> btstep, MOM_barotropic.F90:455
    integer(c_int) function mom6x_btstep(ctx, U_in, V_in, eta_in, dt, bc_accel_u, bc_accel_v, taux, tauy, pbce, &
        eta_PF_in, U_Cor, V_Cor, accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v, &
        BT_cont, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav) bind(C, name="mom6x_btstep")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, U_in, V_in, eta_in ; real(c_double), value :: dt
      type(c_ptr), value :: bc_accel_u, bc_accel_v, taux, tauy, pbce, eta_PF_in, U_Cor, V_Cor
      type(c_ptr), value :: accel_layer_u, accel_layer_v, eta_out, uhbtav, vhbtav, visc_rem_u, visc_rem_v
      type(c_ptr), value :: BT_cont, taux_bot, tauy_bot, uh0, vh0, u_uh0, v_vh0, etaav
    end function

    !> CoriolisAdv_init :1054 / CorAdCalc :125 (MOM_CoriolisAdv.F90)
    integer(c_int) function mom6x_CoriolisAdv_init(ctx, p) bind(C, name="mom6x_CoriolisAdv_init")
      import :: c_int, c_ptr, mom6x_coriolis_params ; type(c_ptr), value :: ctx
      type(mom6x_coriolis_params), intent(in) :: p
    end function
    integer(c_int) function mom6x_CorAdCalc(ctx, u, v, h, uh, vh, CAu, CAv) bind(C, name="mom6x_CorAdCalc")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx, u, v, h, uh, vh, CAu, CAv
    end function
    !> PressureForce_init :85 / PressureForce :41 (MOM_PressureForce.F90 -> PressureForce_FV_Bouss)
    integer(c_int) function mom6x_PressureForce_init(ctx, p, Rlay, g_prime) bind(C, name="mom6x_PressureForce_init")
      import :: c_int, c_ptr, c_double, mom6x_pgf_params ; type(c_ptr), value :: ctx
      type(mom6x_pgf_params), intent(in) :: p ; real(c_double), intent(in) :: Rlay(*), g_prime(*)
    end function
    integer(c_int) function mom6x_PressureForce(ctx, h, PFu, PFv, pbce, eta) bind(C, name="mom6x_PressureForce")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx, h, PFu, PFv, pbce, eta
    end function
    !> vertvisc :557 / vertvisc_remnant :1229 (MOM_vert_friction.F90); coefficients from vertvisc_coef :1357
    !> tv%T, tv%S (device), tv%eqn_of_state for the use_EOS branch of PressureForce_FV_Bouss (FV.F90:1206)
    integer(c_int) function mom6x_PressureForce_set_tv(ctx, T, S, eos) bind(C, name="mom6x_PressureForce_set_tv")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, T, S, eos
    end function
    !> vertvisc_init :3135 / the vertvisc_type inputs / vertvisc_coef :1357 on the device
    integer(c_int) function mom6x_vertvisc_init(ctx, p) bind(C, name="mom6x_vertvisc_init")
      import :: c_ptr, c_int, mom6x_vertvisc_params
      type(c_ptr), value :: ctx ; type(mom6x_vertvisc_params), intent(in) :: p
    end function
    integer(c_int) function mom6x_vertvisc_set_visc(ctx, Kv_bbl_u, Kv_bbl_v, bbl_thick_u, bbl_thick_v, Kv_shear, Ray_u, Ray_v) &
        bind(C, name="mom6x_vertvisc_set_visc")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, Kv_bbl_u, Kv_bbl_v, bbl_thick_u, bbl_thick_v, Kv_shear, Ray_u, Ray_v
    end function
    !> ALE_remap_tracers (MOM_ALE.F90:760): fields = array of nfields device pointers (Reg%Tr(m)%t)
    integer(c_int) function mom6x_ALE_remap_tracers(ctx, p, h_old, h_new, fields, nfields) bind(C, name="mom6x_ALE_remap_tracers")
      import :: c_ptr, c_int, mom6x_remapping_params
      type(c_ptr), value :: ctx, h_old, h_new, fields ; type(mom6x_remapping_params), intent(in) :: p
      integer(c_int), value :: nfields
    end function
    !> ALE_remap_set_h_vel (MOM_ALE.F90:882)
    integer(c_int) function mom6x_ALE_remap_set_h_vel(ctx, h_new, h_u, h_v) bind(C, name="mom6x_ALE_remap_set_h_vel")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, h_new, h_u, h_v
    end function
    !> ALE_remap_velocities (MOM_ALE.F90:1089)
    integer(c_int) function mom6x_ALE_remap_velocities(ctx, p, h_old_u, h_old_v, h_new_u, h_new_v, u, v) &
        bind(C, name="mom6x_ALE_remap_velocities")
      import :: c_ptr, c_int, mom6x_remapping_params
      type(c_ptr), value :: ctx, h_old_u, h_old_v, h_new_u, h_new_v, u, v ; type(mom6x_remapping_params), intent(in) :: p
    end function
    !> ... with REMAP_VEL_CONSERVE_KE and allow_preserve_variance (MOM_ALE.F90:1166-1195)
    integer(c_int) function mom6x_ALE_remap_velocities_conserve_ke(ctx, p, h_old_u, h_old_v, h_new_u, h_new_v, u, v) &
        bind(C, name="mom6x_ALE_remap_velocities_conserve_ke")
      import :: c_ptr, c_int, mom6x_remapping_params
      type(c_ptr), value :: ctx, h_old_u, h_old_v, h_new_u, h_new_v, u, v ; type(mom6x_remapping_params), intent(in) :: p
    end function
    !> ALE_remap_set_h_vel (old grid), ALE_remap_set_h_vel (new grid), ALE_remap_velocities as one call, from the cells' thicknesses
    integer(c_int) function mom6x_ALE_remap_velocities_from_h(ctx, p, h_old, h_new, u, v) bind(C, name="mom6x_ALE_remap_velocities_from_h")
      import :: c_ptr, c_int, mom6x_remapping_params
      type(c_ptr), value :: ctx, h_old, h_new, u, v ; type(mom6x_remapping_params), intent(in) :: p
    end function
    !> ALE_regrid (MOM_ALE.F90:518) for REGRIDDING_ZSTAR; coordinateResolution: nk host values
    integer(c_int) function mom6x_ALE_regrid_zstar(ctx, p, coordinateResolution, h, h_new, dzRegrid) bind(C, name="mom6x_ALE_regrid_zstar")
      import :: c_ptr, c_int, c_double, mom6x_regrid_zstar_params
      type(c_ptr), value :: ctx, h, h_new, dzRegrid ; type(mom6x_regrid_zstar_params), intent(in) :: p
      real(c_double), intent(in) :: coordinateResolution(*)
    end function
    !> regridding_main (MOM_regridding.F90:862) for REGRIDDING_RHO: target_density = nk+1 host values
    integer(c_int) function mom6x_ALE_regrid_rho(ctx, p, eos, target_density, h, T, S, h_new, dzRegrid) bind(C, name="mom6x_ALE_regrid_rho")
      import :: c_ptr, c_int, c_double, mom6x_regrid_rho_params, mom6x_eos_params
      type(c_ptr), value :: ctx, h, T, S, h_new, dzRegrid
      type(mom6x_regrid_rho_params), intent(in) :: p ; type(mom6x_eos_params), intent(in) :: eos
      real(c_double), intent(in) :: target_density(*)
    end function
    !> regridding_main for REGRIDDING_HYCOM1; max_interface_depths / max_layer_thickness: c_loc of host arrays or c_null_ptr
    integer(c_int) function mom6x_ALE_regrid_hycom1(ctx, p, eos, coordinateResolution, target_density, max_interface_depths, &
                                                    max_layer_thickness, h, T, S, h_new, dzRegrid) bind(C, name="mom6x_ALE_regrid_hycom1")
      import :: c_ptr, c_int, c_double, mom6x_regrid_rho_params, mom6x_eos_params
      type(c_ptr), value :: ctx, max_interface_depths, max_layer_thickness, h, T, S, h_new, dzRegrid
      type(mom6x_regrid_rho_params), intent(in) :: p ; type(mom6x_eos_params), intent(in) :: eos
      real(c_double), intent(in) :: coordinateResolution(*), target_density(*)
    end function
    !> convective_adjustment (MOM_regridding.F90:1905): h, T, S (device) reordered in place
    integer(c_int) function mom6x_ALE_convective_adjustment(ctx, eos, h, T, S) bind(C, name="mom6x_ALE_convective_adjustment")
      import :: c_ptr, c_int, mom6x_eos_params
      type(c_ptr), value :: ctx, h, T, S ; type(mom6x_eos_params), intent(in) :: eos
    end function
    !> remapping_core_h (MOM_remapping.F90:234) for ncol packed columns
    integer(c_int) function mom6x_remapping_core_h(ctx, p, ncol, n0, h0, u0, n1, h1, u1) bind(C, name="mom6x_remapping_core_h")
      import :: c_ptr, c_int, mom6x_remapping_params
      type(c_ptr), value :: ctx, h0, u0, h1, u1 ; type(mom6x_remapping_params), intent(in) :: p
      integer(c_int), value :: ncol, n0, n1
    end function
    !> remap_dyn_split_RK2_aux_vars (MOM_dynamics_split_RK2.F90:1302)
    integer(c_int) function mom6x_remap_dyn_split_RK2_aux_vars(ctx, p, h_old_u, h_old_v, h_new_u, h_new_v) &
        bind(C, name="mom6x_remap_dyn_split_RK2_aux_vars")
      import :: c_ptr, c_int, mom6x_remapping_params
      type(c_ptr), value :: ctx, h_old_u, h_old_v, h_new_u, h_new_v ; type(mom6x_remapping_params), intent(in) :: p
    end function
    !> reproducing_sum_3d (MOM_coms.F90:349); sums, EFP_sum, EFP_lay_sums, err: c_null_ptr when absent
    integer(c_int) function mom6x_reproducing_sum_3d(ctx, array, nk, is, ie, js, je, unscale, only_on_PE, sum, sums, EFP_sum, &
        EFP_lay_sums, err) bind(C, name="mom6x_reproducing_sum_3d")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, array, sums, EFP_sum, EFP_lay_sums, err
      integer(c_int), value :: nk, is, ie, js, je, only_on_PE ; real(c_double), value :: unscale ; real(c_double), intent(out) :: sum
    end function
    !> reproducing_sum_2d (MOM_coms.F90:235)
    integer(c_int) function mom6x_reproducing_sum_2d(ctx, array, is, ie, js, je, unscale, only_on_PE, sum, EFP_sum, err) &
        bind(C, name="mom6x_reproducing_sum_2d")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, array, EFP_sum, err
      integer(c_int), value :: is, ie, js, je, only_on_PE ; real(c_double), value :: unscale ; real(c_double), intent(out) :: sum
    end function
    !> chksum_{h,u,v,B}_{2d,3d} (MOM_checksums.F90); scale: c_null_ptr when absent
    integer(c_int) function mom6x_chksum(ctx, array, nk, rank, stagger, haloshift, symmetric, omit_corners, scale, res) &
        bind(C, name="mom6x_chksum")
      import :: c_ptr, c_int, mom6x_chksum_result
      type(c_ptr), value :: ctx, array, scale ; integer(c_int), value :: nk, rank, stagger, haloshift, symmetric, omit_corners
      type(mom6x_chksum_result), intent(out) :: res
    end function
    !> the restart `checksum` attribute (MOM_restart.F90:1741) of a device-resident field
    integer(c_int) function mom6x_field_chksum(ctx, array, nk, is, ie, js, je, unscale, chksum) bind(C, name="mom6x_field_chksum")
      import :: c_ptr, c_int, c_double, c_int64_t
      type(c_ptr), value :: ctx, array ; integer(c_int), value :: nk, is, ie, js, je ; real(c_double), value :: unscale
      integer(c_int64_t), intent(out) :: chksum
    end function
    !> MOM_sum_output_init (MOM_sum_output.F90:147) + depth_list_setup (:1161); g_prime: GV%g_prime(1:nk)
    integer(c_int) function mom6x_sum_output_init(ctx, p, g_prime) bind(C, name="mom6x_sum_output_init")
      import :: c_ptr, c_int, c_double, mom6x_sum_output_params
      type(c_ptr), value :: ctx ; type(mom6x_sum_output_params), intent(in) :: p ; real(c_double), intent(in) :: g_prime(*)
    end function
    integer(c_int) function mom6x_depth_list(ctx, listsize, depth, area, vol_below) bind(C, name="mom6x_depth_list")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, depth, area, vol_below ; integer(c_int), intent(out) :: listsize
    end function
    !> the sums of write_energy (MOM_sum_output.F90:321); T, S: c_null_ptr without ENABLE_THERMODYNAMICS
    integer(c_int) function mom6x_write_energy(ctx, u, v, h, T, S, sums, mass_lay, KE, PE, Z_0APE) bind(C, name="mom6x_write_energy")
      import :: c_ptr, c_int, c_double, mom6x_energy_sums
      type(c_ptr), value :: ctx, u, v, h, T, S ; type(mom6x_energy_sums), intent(out) :: sums
      real(c_double), intent(out) :: mass_lay(*), KE(*), PE(*), Z_0APE(*)
    end function
    !> CS%dtbt, the restart scalar DTBT (MOM_barotropic.F90:6290): get / set are c_null_ptr when not wanted
    integer(c_int) function mom6x_barotropic_dtbt(ctx, get, set) bind(C, name="mom6x_barotropic_dtbt")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, get, set
    end function
    !> the count of btstep's "eta has dropped below bathyT" warnings (MOM_barotropic.F90:2738-2745) and the first offender
    integer(c_int) function mom6x_btstep_warnings(ctx, reset, count, info) bind(C, name="mom6x_btstep_warnings")
      import :: c_ptr, c_int, c_long_long, c_double
      type(c_ptr), value :: ctx ; integer(c_int), value :: reset
      integer(c_long_long), intent(out) :: count ; real(c_double), intent(out) :: info(4)
    end function
    !> DIRECT_STRESS / HMIX_STRESS (MOM_vert_friction.F90:3208, :707); h = vertvisc's thickness argument (device pointer)
    integer(c_int) function mom6x_vertvisc_set_direct_stress(ctx, Hmix_stress, h) bind(C, name="mom6x_vertvisc_set_direct_stress")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, h ; real(c_double), value :: Hmix_stress
    end function
    !> hor_visc_init (MOM_hor_visc.F90:2322): the 2-D viscosity planes are made on the device from the metric block
    integer(c_int) function mom6x_hor_visc_init(ctx, p) bind(C, name="mom6x_hor_visc_init")
      import :: c_ptr, c_int, mom6x_hor_visc_params
      type(c_ptr), value :: ctx ; type(mom6x_hor_visc_params), intent(in) :: p
    end function
    !> horizontal_viscosity(u, v, h, uh, vh, diffu, diffv, ...) (MOM_hor_visc.F90:266); uh, vh only feed FrictWork
    integer(c_int) function mom6x_horizontal_viscosity(ctx, u, v, h, diffu, diffv) bind(C, name="mom6x_horizontal_viscosity")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, u, v, h, diffu, diffv
    end function
    integer(c_int) function mom6x_vertvisc_coef(ctx, u, v, h, dt) bind(C, name="mom6x_vertvisc_coef")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, u, v, h ; real(c_double), value :: dt
    end function
    integer(c_int) function mom6x_vertvisc_set_coef(ctx, a_u, a_v, h_u, h_v, Ray_u, Ray_v) &
        bind(C, name="mom6x_vertvisc_set_coef")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx, a_u, a_v, h_u, h_v, Ray_u, Ray_v
    end function
    integer(c_int) function mom6x_vertvisc(ctx, u, v, taux, tauy, dt, taux_bot, tauy_bot) bind(C, name="mom6x_vertvisc")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, u, v, taux, tauy ; real(c_double), value :: dt
      type(c_ptr), value :: taux_bot, tauy_bot
    end function
    integer(c_int) function mom6x_vertvisc_remnant(ctx, visc_rem_u, visc_rem_v, dt) bind(C, name="mom6x_vertvisc_remnant")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, visc_rem_u, visc_rem_v ; real(c_double), value :: dt
    end function

    !> initialize_dyn_split_RK2 :1346 / step_MOM_dyn_split_RK2 :294 (MOM_dynamics_split_RK2.F90)
    integer(c_int) function mom6x_initialize_dyn_split_RK2(ctx, p) bind(C, name="mom6x_initialize_dyn_split_RK2")
      import :: c_int, c_ptr, mom6x_rk2_params ; type(c_ptr), value :: ctx ; type(mom6x_rk2_params), intent(in) :: p
    end function
    integer(c_int) function mom6x_dyn_split_RK2_new_run(ctx, u, v, h, uh, vh, dt) bind(C, name="mom6x_dyn_split_RK2_new_run")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, u, v, h, uh, vh ; real(c_double), value :: dt
    end function
    !> initialize_dyn_split_RK2 :1577-1668 for a restarted run: have = the MOM6X_RK2_HAVE_* bits of the uploaded restart variables
    integer(c_int) function mom6x_dyn_split_RK2_restart_fills(ctx, u, v, h, uh, vh, dt, have) bind(C, name="mom6x_dyn_split_RK2_restart_fills")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, u, v, h, uh, vh ; real(c_double), value :: dt ; integer(c_int), value :: have
    end function
    !> a restarted run that read CAu_pred, CAv_pred from the file: the first step must not recompute them (RK2.F90:1616)
    integer(c_int) function mom6x_rk2_set_CAu_pred_stored(ctx, stored) bind(C, name="mom6x_rk2_set_CAu_pred_stored")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx ; integer(c_int), value :: stored
    end function
    type(c_ptr) function mom6x_rk2_field(ctx, which) bind(C, name="mom6x_rk2_field")
      import :: c_ptr, c_int ; type(c_ptr), value :: ctx ; integer(c_int), value :: which
    end function
    integer(c_int) function mom6x_step_dyn_split_RK2(ctx, u_inst, v_inst, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy, &
        dt, calc_dtbt, hooks) bind(C, name="mom6x_step_dyn_split_RK2")
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: ctx, u_inst, v_inst, h, uh, vh, uhtr, vhtr, eta_av, taux, tauy
      real(c_double), value :: dt ; integer(c_int), value :: calc_dtbt ; type(c_ptr), value :: hooks
    end function

    !> MOM_domains: LAYOUT and halo updates over RCCL
    !> communicators made afterwards use the host's transport; c_null_ptr: back to RCCL
    integer(c_int) function mom6x_comm_set_transport(t) bind(C, name="mom6x_comm_set_transport")
      import :: c_ptr, c_int
      type(c_ptr), value :: t
    end function
    integer(c_int) function mom6x_comm_unique_id(id128) bind(C, name="mom6x_comm_unique_id")
      import :: c_int, c_char ; character(kind=c_char), intent(out) :: id128(128)
    end function
    integer(c_int) function mom6x_comm_init(ctx, npx, npy, px, py, id128, force_nccl_self) bind(C, name="mom6x_comm_init")
      import :: c_int, c_ptr, c_char ; type(c_ptr), value :: ctx ; integer(c_int), value :: npx, npy, px, py
      character(kind=c_char), intent(in) :: id128(128) ; integer(c_int), value :: force_nccl_self
    end function
    !> btstep's own group pass (MOM_barotropic.F90:2505-2512) overlapped with the own-points half of the next sub-step (off by default)
    integer(c_int) function mom6x_comm_overlap_btstep(ctx, on) bind(C, name="mom6x_comm_overlap_btstep")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx ; integer(c_int), value :: on
    end function
    integer(c_int) function mom6x_pass_fields(ctx, fields, staggers, nks, n) bind(C, name="mom6x_pass_fields")
      import :: c_int, c_ptr ; type(c_ptr), value :: ctx ; type(c_ptr), intent(in) :: fields(*)
      integer(c_int), intent(in) :: staggers(*), nks(*) ; integer(c_int), value :: n
    end function
    ! ---- MOM_tracer_advect / MOM_diabatic_aux / MOM_tracer_diabatic ------------------------------------
    integer(c_int) function mom6x_tracer_advect_init(ctx, dt_dyn, default_scheme, useHuynhStencilBug) &
        bind(C, name="mom6x_tracer_advect_init")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx ; real(c_double), value :: dt_dyn ; integer(c_int), value :: default_scheme, useHuynhStencilBug
    end function
    !> advect_tracer (MOM_tracer_advect.F90:53): tracers = array of ntr device pointers (Reg%Tr(m)%t)
    integer(c_int) function mom6x_advect_tracer(ctx, h_end, uhtr, vhtr, dt, tracers, schemes, ntr, x_first_in, max_iter_in, &
        uhr_out, vhr_out, iters_out) bind(C, name="mom6x_advect_tracer")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, h_end, uhtr, vhtr, tracers, schemes, uhr_out, vhr_out, iters_out
      real(c_double), value :: dt ; integer(c_int), value :: ntr, x_first_in, max_iter_in
    end function
    integer(c_int) function mom6x_triDiagTS(ctx, is, ie, js, je, hold, ea, eb, T, S) bind(C, name="mom6x_triDiagTS")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, hold, ea, eb, T, S ; integer(c_int), value :: is, ie, js, je
    end function
    integer(c_int) function mom6x_triDiagTS_Eulerian(ctx, is, ie, js, je, hold, ent, T, S) &
        bind(C, name="mom6x_triDiagTS_Eulerian")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx, hold, ent, T, S ; integer(c_int), value :: is, ie, js, je
    end function
    integer(c_int) function mom6x_tracer_vertdiff(ctx, h_old, ea, eb, dt, tr, sfc_flux, btm_flux, convert_flux) &
        bind(C, name="mom6x_tracer_vertdiff")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, h_old, ea, eb, tr, sfc_flux, btm_flux ; real(c_double), value :: dt
      integer(c_int), value :: convert_flux
    end function
    integer(c_int) function mom6x_tracer_vertdiff_Eulerian(ctx, h_old, ent, dt, tr, sfc_flux, btm_flux, convert_flux) &
        bind(C, name="mom6x_tracer_vertdiff_Eulerian")
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: ctx, h_old, ent, tr, sfc_flux, btm_flux ; real(c_double), value :: dt
      integer(c_int), value :: convert_flux
    end function
    !> tracer_vertdiff / _Eulerian with sink_rate (MOM_tracer_diabatic.F90:123-179 / :315-380); btm_reservoir may be c_null_ptr
    integer(c_int) function mom6x_tracer_vertdiff_sink(ctx, h_old, ea, eb, dt, tr, sfc_flux, btm_flux, btm_reservoir, sink_rate, convert_flux) &
        bind(C, name="mom6x_tracer_vertdiff_sink")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, h_old, ea, eb, tr, sfc_flux, btm_flux, btm_reservoir
      real(c_double), value :: dt, sink_rate ; integer(c_int), value :: convert_flux
    end function
    integer(c_int) function mom6x_tracer_vertdiff_Eulerian_sink(ctx, h_old, ent, dt, tr, sfc_flux, btm_flux, btm_reservoir, sink_rate, convert_flux) &
        bind(C, name="mom6x_tracer_vertdiff_Eulerian_sink")
      import :: c_int, c_ptr, c_double ; type(c_ptr), value :: ctx, h_old, ent, tr, sfc_flux, btm_flux, btm_reservoir
      real(c_double), value :: dt, sink_rate ; integer(c_int), value :: convert_flux
    end function
    integer(c_int) function mom6x_diabatic_is_trivial(ctx) bind(C, name="mom6x_diabatic_is_trivial")
      import :: c_ptr, c_int
      type(c_ptr), value :: ctx
    end function
  end interface

end module mom6x_c_api
