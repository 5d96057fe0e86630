!> Start-up check a Fortran host performs: the bind(C) mirrors have the sizes the C library was built with.
program check_abi
  use, intrinsic :: iso_c_binding
  use mom6x_c_api
  implicit none
  type(mom6x_dims) :: d
  type(mom6x_vgrid) :: gv
  type(mom6x_continuity_params) :: cp
  type(mom6x_BT_cont) :: btc
  type(mom6x_barotropic_params) :: bp
  type(mom6x_coriolis_params) :: co
  type(mom6x_pgf_params) :: pg
  type(mom6x_rk2_params) :: rk
  type(mom6x_rk2_hooks) :: hk
  type(mom6x_eos_params) :: eo
  type(mom6x_vertvisc_params) :: vv
  type(mom6x_hor_visc_params) :: hv
  type(mom6x_remapping_params) :: rm
  type(mom6x_regrid_zstar_params) :: rz
  type(mom6x_chksum_result) :: cr
  type(mom6x_sum_output_params) :: sp
  type(mom6x_energy_sums) :: es
  type(mom6x_regrid_rho_params) :: rr
  integer :: nbad, rc
  nbad = 0
  call chk(0, int(c_sizeof(d)), "mom6x_dims")
  call chk(1, int(c_sizeof(gv)), "mom6x_vgrid")
  call chk(2, int(c_sizeof(cp)), "mom6x_continuity_params")
  call chk(3, int(c_sizeof(btc)), "mom6x_BT_cont")
  call chk(4, int(c_sizeof(bp)), "mom6x_barotropic_params")
  call chk(5, int(c_sizeof(co)), "mom6x_coriolis_params")
  call chk(6, int(c_sizeof(pg)), "mom6x_pgf_params")
  call chk(7, int(c_sizeof(rk)), "mom6x_rk2_params")
  call chk(8, int(c_sizeof(hk)), "mom6x_rk2_hooks")
  call chk(9, int(c_sizeof(eo)), "mom6x_eos_params")
  call chk(10, int(c_sizeof(vv)), "mom6x_vertvisc_params")
  call chk(11, int(c_sizeof(hv)), "mom6x_hor_visc_params")
  call chk(12, int(c_sizeof(rm)), "mom6x_remapping_params")
  call chk(13, int(c_sizeof(rz)), "mom6x_regrid_zstar_params")
  call chk(14, int(c_sizeof(cr)), "mom6x_chksum_result")
  call chk(15, int(c_sizeof(sp)), "mom6x_sum_output_params")
  call chk(16, int(c_sizeof(es)), "mom6x_energy_sums")
  call chk(17, int(c_sizeof(rr)), "mom6x_regrid_rho_params")
  if (mom6x_abi_version() /= MOM6X_ABI_BUILT_FOR) then
    print '(a,i0,a,i0)', "ABI version: library ", mom6x_abi_version(), ", fortran/mom6x_c_api.F90 ", MOM6X_ABI_BUILT_FOR ; nbad = nbad + 1
  endif
  rc = mom6x_dims_init(d, 1440, 1080, 75, 4)
  if (rc /= 0 .or. d%pitch /= 1472 .or. d%ioff /= 16) then
    print *, "mom6x_dims_init mismatch", rc, d%pitch, d%ioff ; nbad = nbad + 1
  endif
  if (nbad == 0) then
    print '(a)', "fortran ABI check OK"
  else
    print '(a,i0)', "fortran ABI check FAILED: ", nbad
    stop 1
  endif
contains
  subroutine chk(which, fsize, name)
    integer, intent(in) :: which, fsize
    character(len=*), intent(in) :: name
    if (mom6x_struct_size(int(which, c_int)) /= fsize) then
      print *, "size mismatch for ", name, mom6x_struct_size(int(which, c_int)), fsize
      nbad = nbad + 1
    endif
  end subroutine chk
end program check_abi
