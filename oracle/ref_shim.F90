!> ORACLE SUPPORT (test infrastructure only): bind(C) doors into the REAL reference modules that compile from their own
!! source files without FMS / netCDF (src/ALE/PLM_functions.F90, src/ALE/PCM_functions.F90: no `use` statements at all;
!! src/ALE/Recon1d_PPM_H4_2019.F90 with Recon1d_type.F90 and src/framework/numerical_testing_type.F90;
!! src/equation_of_state/MOM_EOS_UNESCO.F90 with MOM_EOS_base_type.F90).
!! oracle/Makefile compiles those two files where they lie under /root/reference together with this file into
!! oracle/_ref/libref_ale.so; tests/test_remap_cpu.py then holds orc_remap.c's PLM / PCM reconstructions to the
!! reference's own code, bit for bit, on random columns.  Nothing of the reference is copied here.
module ref_shim
  use, intrinsic :: iso_c_binding
  use PLM_functions, only : PLM_reconstruction, PLM_boundary_extrapolation, PLM_slope_wa, PLM_monotonized_slope, &
                            PLM_extrapolate_slope
  use PCM_functions, only : PCM_reconstruction
  use Recon1d_PPM_H4_2019, only : PPM_H4_2019
  use MOM_EOS_UNESCO, only : UNESCO_EOS
  use MOM_EOS_Roquet_rho, only : Roquet_rho_EOS
  use MOM_EOS_Jackett06, only : Jackett06_EOS
  use MOM_EOS_Roquet_SpV, only : Roquet_SpV_EOS
  implicit none
contains
  !> src/equation_of_state/MOM_EOS_UNESCO.F90 (with MOM_EOS_base_type.F90: no other `use`): the elemental density :95, density
  !! anomaly :133 and the T, S derivatives :244 of n points -- oracle/orc_dyn.c's unesco_* are held to these bit for bit.
  !> src/equation_of_state/MOM_EOS_Roquet_SpV.F90 likewise (density :318, anomaly :335, derivatives :427).
  subroutine ref_ROQUET_SPV(n, T, S, p, rho_ref, rho, rho_anom, drho_dT, drho_dS) bind(C, name="ref_ROQUET_SPV")
    integer(c_int), value :: n
    real(c_double), intent(in) :: T(n), S(n), p(n)
    real(c_double), value :: rho_ref
    real(c_double), intent(out) :: rho(n), rho_anom(n), drho_dT(n), drho_dS(n)
    type(Roquet_SpV_EOS) :: eos
    integer :: i
    do i=1,n
      rho(i) = eos%density_elem(T(i), S(i), p(i))
      rho_anom(i) = eos%density_anomaly_elem(T(i), S(i), p(i), rho_ref)
      call eos%calculate_density_derivs_elem(T(i), S(i), p(i), drho_dT(i), drho_dS(i))
    enddo
  end subroutine
  !> src/equation_of_state/MOM_EOS_Jackett06.F90 likewise (density :77, anomaly :111, derivatives :216).
  subroutine ref_JACKETT06(n, T, S, p, rho_ref, rho, rho_anom, drho_dT, drho_dS) bind(C, name="ref_JACKETT06")
    integer(c_int), value :: n
    real(c_double), intent(in) :: T(n), S(n), p(n)
    real(c_double), value :: rho_ref
    real(c_double), intent(out) :: rho(n), rho_anom(n), drho_dT(n), drho_dS(n)
    type(Jackett06_EOS) :: eos
    integer :: i
    do i=1,n
      rho(i) = eos%density_elem(T(i), S(i), p(i))
      rho_anom(i) = eos%density_anomaly_elem(T(i), S(i), p(i), rho_ref)
      call eos%calculate_density_derivs_elem(T(i), S(i), p(i), drho_dT(i), drho_dS(i))
    enddo
  end subroutine
  !> src/equation_of_state/MOM_EOS_Roquet_rho.F90 likewise (density :192, anomaly :248, derivatives :340).
  subroutine ref_ROQUET_RHO(n, T, S, p, rho_ref, rho, rho_anom, drho_dT, drho_dS) bind(C, name="ref_ROQUET_RHO")
    integer(c_int), value :: n
    real(c_double), intent(in) :: T(n), S(n), p(n)
    real(c_double), value :: rho_ref
    real(c_double), intent(out) :: rho(n), rho_anom(n), drho_dT(n), drho_dS(n)
    type(Roquet_rho_EOS) :: eos
    integer :: i
    do i=1,n
      rho(i) = eos%density_elem(T(i), S(i), p(i))
      rho_anom(i) = eos%density_anomaly_elem(T(i), S(i), p(i), rho_ref)
      call eos%calculate_density_derivs_elem(T(i), S(i), p(i), drho_dT(i), drho_dS(i))
    enddo
  end subroutine
  subroutine ref_UNESCO(n, T, S, p, rho_ref, rho, rho_anom, drho_dT, drho_dS) bind(C, name="ref_UNESCO")
    integer(c_int), value :: n
    real(c_double), intent(in) :: T(n), S(n), p(n)
    real(c_double), value :: rho_ref
    real(c_double), intent(out) :: rho(n), rho_anom(n), drho_dT(n), drho_dS(n)
    type(UNESCO_EOS) :: eos
    integer :: i
    do i=1,n
      rho(i) = eos%density_elem(T(i), S(i), p(i))
      rho_anom(i) = eos%density_anomaly_elem(T(i), S(i), p(i), rho_ref)
      call eos%calculate_density_derivs_elem(T(i), S(i), p(i), drho_dT(i), drho_dS(i))
    enddo
  end subroutine
  !> The reference's class-based PPM with explicit 4th-order edge values, 2019 expressions (Recon1d_PPM_H4_2019.F90:86):
  !! the same algorithm as build_reconstructions_1d's REMAPPING_PPM_H4 branch without boundary extrapolation.
  subroutine ref_PPM_H4_2019(n, h, u, h_neglect, ul, ur) bind(C, name="ref_PPM_H4_2019")
    integer(c_int), value :: n
    real(c_double), intent(in) :: h(n), u(n)
    real(c_double), value :: h_neglect
    real(c_double), intent(out) :: ul(n), ur(n)
    type(PPM_H4_2019) :: r
    call r%init(int(n), h_neglect=h_neglect)
    call r%reconstruct(h, u)
    ul(:) = r%ul(:) ; ur(:) = r%ur(:)
    call r%destroy()
  end subroutine
  !> Recon1d_type.F90:173 remap_to_sub_grid with the PPM_H4_2019 reconstruction: sub-cell averages and integrals (with the
  !! thickest-sub-cell conservation fix) on a given intersection of the source and target grids.
  subroutine ref_PPM_H4_2019_to_sub_grid(n0, h0, u0, h_neglect, n1, h_sub, isrc_start, isrc_end, isrc_max, isub_src, &
                                         u_sub, uh_sub) bind(C, name="ref_PPM_H4_2019_to_sub_grid")
    integer(c_int), value :: n0, n1
    real(c_double), intent(in) :: h0(n0), u0(n0), h_sub(n0+n1+1)
    real(c_double), value :: h_neglect
    integer(c_int), intent(in) :: isrc_start(n0), isrc_end(n0), isrc_max(n0), isub_src(n0+n1+1)
    real(c_double), intent(out) :: u_sub(n0+n1+1), uh_sub(n0+n1+1)
    type(PPM_H4_2019) :: r
    real(c_double) :: err
    call r%init(int(n0), h_neglect=h_neglect)
    call r%reconstruct(h0, u0)
    call r%remap_to_sub_grid(h0, u0, int(n1), h_sub, isrc_start, isrc_end, isrc_max, isub_src, u_sub, uh_sub, err)
    call r%destroy()
  end subroutine
  subroutine ref_PLM_reconstruction(n, h, u, edges, coefs, h_neglect, extrapolate) bind(C, name="ref_PLM_reconstruction")
    integer(c_int), value :: n, extrapolate
    real(c_double), intent(in) :: h(n), u(n)
    real(c_double), intent(inout) :: edges(n,2), coefs(n,2)
    real(c_double), value :: h_neglect
    call PLM_reconstruction(int(n), h, u, edges, coefs, h_neglect)
    if (extrapolate /= 0) call PLM_boundary_extrapolation(int(n), h, u, edges, coefs, h_neglect)
  end subroutine
  subroutine ref_PCM_reconstruction(n, u, edges, coefs) bind(C, name="ref_PCM_reconstruction")
    integer(c_int), value :: n
    real(c_double), intent(in) :: u(n)
    real(c_double), intent(inout) :: edges(n,2), coefs(n,1)
    call PCM_reconstruction(int(n), u, edges, coefs)
  end subroutine
  real(c_double) function ref_PLM_slope_wa(h_l, h_c, h_r, h_neglect, u_l, u_c, u_r) bind(C, name="ref_PLM_slope_wa")
    real(c_double), value :: h_l, h_c, h_r, h_neglect, u_l, u_c, u_r
    ref_PLM_slope_wa = PLM_slope_wa(h_l, h_c, h_r, h_neglect, u_l, u_c, u_r)
  end function
  real(c_double) function ref_PLM_monotonized_slope(u_l, u_c, u_r, s_l, s_c, s_r) bind(C, name="ref_PLM_monotonized_slope")
    real(c_double), value :: u_l, u_c, u_r, s_l, s_c, s_r
    ref_PLM_monotonized_slope = PLM_monotonized_slope(u_l, u_c, u_r, s_l, s_c, s_r)
  end function
  real(c_double) function ref_PLM_extrapolate_slope(h_l, h_c, h_neglect, u_l, u_c) bind(C, name="ref_PLM_extrapolate_slope")
    real(c_double), value :: h_l, h_c, h_neglect, u_l, u_c
    ref_PLM_extrapolate_slope = PLM_extrapolate_slope(h_l, h_c, h_neglect, u_l, u_c)
  end function
end module ref_shim
