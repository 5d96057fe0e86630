!> tests/fortran_stubs/mom_stubs.F90 -- TEST INFRASTRUCTURE, not product and not reference text.
!!
!! A MOM6 source tree cannot be built in this repository's environment (FMS and netCDF are absent), so the shim modules
!! of fortran/shims/ -- which `use MOM_grid`, `MOM_file_parser`, `MOM_restart`, ... -- are compiled and RUN against the
!! interface-only stand-ins below: each module carries the name of the MOM6 module it stands for and declares ONLY the
!! derived-type members and procedure signatures the shims touch (the member names and argument lists are the interface
!! facts the shims are written against; the bodies are a few lines of bookkeeping written for this harness).
!!   * get_param answers from a small name -> value table the test driver fills, else with the default;
!!   * register_restart_field / register_restart_pair keep pointers to the registered arrays, stub_save_restart writes them
!!     to an unformatted file and stub_restore_state reads them back and marks them initialised (query_initialized);
!!   * MOM_error(FATAL, ...) prints and stops with a non-zero code.
!! tests/test_fortran_shims_cpu.py compiles every shim against these (-fsyntax-only for each, then a full link of the
!! driver); tests/test_fortran_gpu.py runs the driver on the GPU.

module MOM_error_handler
  implicit none ; public
  integer, parameter :: NOTE = 0, WARNING = 1, FATAL = 2
contains
  subroutine MOM_error(level, message, all_print)
    integer, intent(in) :: level ; character(len=*), intent(in) :: message ; logical, optional, intent(in) :: all_print
    if (level == FATAL) then
      print '(a)', "FATAL: "//trim(message) ; flush(6) ; error stop 3
    elseif (level == WARNING) then
      print '(a)', "WARNING: "//trim(message)
    else
      print '(a)', "NOTE: "//trim(message)
    endif
  end subroutine MOM_error
  subroutine MOM_mesg(message, verb, all_print)
    character(len=*), intent(in) :: message ; integer, optional, intent(in) :: verb ; logical, optional, intent(in) :: all_print
  end subroutine MOM_mesg
  logical function is_root_pe()
    is_root_pe = .true.
  end function is_root_pe
  subroutine callTree_enter(mesg, n)
    character(len=*), intent(in) :: mesg ; integer, optional, intent(in) :: n
  end subroutine callTree_enter
  subroutine callTree_leave(mesg)
    character(len=*), intent(in) :: mesg
  end subroutine callTree_leave
  subroutine callTree_waypoint(mesg, n)
    character(len=*), intent(in) :: mesg ; integer, optional, intent(in) :: n
  end subroutine callTree_waypoint
end module MOM_error_handler

module MOM_coms
  implicit none ; public
  integer :: stub_npes = 1, stub_pe = 0   !< the harness may pretend to be one PE of several (to test the refusal paths)
  interface broadcast
    module procedure broadcast_int1D
  end interface
contains
  integer function PE_here() ; PE_here = stub_pe ; end function
  integer function root_PE() ; root_PE = 0 ; end function
  integer function num_PEs() ; num_PEs = stub_npes ; end function
  subroutine broadcast_int1D(dat, length, from_PE, PElist, blocking)
    integer, intent(inout) :: dat(:) ; integer, intent(in) :: length
    integer, optional, intent(in) :: from_PE, PElist(:) ; logical, optional, intent(in) :: blocking
  end subroutine broadcast_int1D
end module MOM_coms

module MOM_cpu_clock
  implicit none ; public
  integer, parameter :: CLOCK_COMPONENT = 1, CLOCK_SUBCOMPONENT = 11, CLOCK_MODULE_DRIVER = 21, CLOCK_MODULE = 31, CLOCK_ROUTINE = 41
  integer :: stub_clock_calls(64) = 0   !< begin-calls per clock id (the harness checks that the shims keep the clocks)
  character(len=48) :: stub_clock_names(64) = ""
  integer :: stub_nclocks = 0
contains
  integer function cpu_clock_id(name, grain)
    character(len=*), intent(in) :: name ; integer, optional, intent(in) :: grain
    stub_nclocks = stub_nclocks + 1 ; cpu_clock_id = stub_nclocks ; stub_clock_names(stub_nclocks) = name
  end function cpu_clock_id
  subroutine cpu_clock_begin(id) ; integer, intent(in) :: id ; stub_clock_calls(id) = stub_clock_calls(id) + 1 ; end subroutine
  subroutine cpu_clock_end(id) ; integer, intent(in) :: id ; end subroutine
end module MOM_cpu_clock

module MOM_time_manager
  implicit none ; public
  type :: time_type ; integer :: seconds = 0, days = 0 ; end type time_type
end module MOM_time_manager

module MOM_unit_scaling
  implicit none ; public
  type :: unit_scale_type   ! every *_RESCALE_POWER = 0: all factors are exactly 1
    real :: m_to_Z = 1., Z_to_m = 1., m_to_L = 1., L_to_m = 1., s_to_T = 1., T_to_s = 1., R_to_kg_m3 = 1., kg_m3_to_R = 1.
    real :: m_s_to_L_T = 1., L_T_to_m_s = 1., L_T2_to_m_s2 = 1., Z2_T_to_m2_s = 1., m2_s_to_Z2_T = 1., Pa_to_RL2_T2 = 1.
    real :: C_to_degC = 1., degC_to_C = 1., S_to_ppt = 1., ppt_to_S = 1., RL2_T2_to_Pa = 1.
  end type unit_scale_type
end module MOM_unit_scaling

module MOM_hor_index
  implicit none ; public
  type :: hor_index_type
    integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, IscB, IecB, JscB, JecB, IsdB, IedB, JsdB, JedB
    integer :: idg_offset = 0, jdg_offset = 0
    logical :: symmetric = .true.
  end type hor_index_type
end module MOM_hor_index

module MOM_domains
  implicit none ; public
  type :: MOM_domain_type
    integer :: niglobal, njglobal, nihalo, njhalo, layout(2) = (/1, 1/)
    logical :: symmetric = .true.
  end type MOM_domain_type
end module MOM_domains

module MOM_grid
  use MOM_hor_index, only : hor_index_type
  use MOM_domains, only : MOM_domain_type
  implicit none ; public
  type :: ocean_grid_type
    type(MOM_domain_type), pointer :: Domain => NULL()
    type(hor_index_type) :: HI
    integer :: isc, iec, jsc, jec, isd, ied, jsd, jed, IscB, IecB, JscB, JecB, IsdB, IedB, JsdB, JedB, ke
    integer :: idg_offset = 0, jdg_offset = 0, first_direction = 0
    logical :: symmetric = .true.
    real, allocatable, dimension(:,:) :: mask2dT, mask2dCu, mask2dCv, mask2dBu, dxT, dyT, IdxT, IdyT, dxCu, dyCu, IdxCu, IdyCu, &
        dxCv, dyCv, IdxCv, IdyCv, dxBu, dyBu, IdxBu, IdyBu, areaT, IareaT, areaBu, IareaBu, areaCu, areaCv, IareaCu, IareaCv, &
        dy_Cu, dx_Cv, bathyT, CoriolisBu, Coriolis2Bu
    real :: Z_ref = 0.0, max_depth = 0.0
  end type ocean_grid_type
end module MOM_grid

module MOM_verticalGrid
  implicit none ; public
  type :: verticalGrid_type
    integer :: ke
    real :: max_depth, g_Earth, Rho0, Angstrom_H, Angstrom_Z, Angstrom_m, H_subroundoff, dZ_subroundoff
    real :: H_to_Z = 1., Z_to_H = 1., H_to_RZ, RZ_to_H, H_to_m = 1., m_to_H = 1., H_to_MKS = 1., m2_s_to_HZ_T = 1., HZ_T_to_m2_s = 1.
    real :: H_to_kg_m2, kg_m2_to_H
    logical :: Boussinesq = .true.
    real, allocatable :: Rlay(:), g_prime(:)
  end type verticalGrid_type
end module MOM_verticalGrid

module MOM_file_parser
  use MOM_error_handler, only : MOM_error, FATAL
  implicit none ; private
  public :: param_file_type, get_param, log_version, stub_set_param, stub_param_log_count
  type :: param_file_type
    integer :: n = 0
    character(len=48) :: names(128) = ""
    character(len=64) :: values(128) = ""
    integer :: nlogged = 0   !< get_param calls that would have been written to MOM_parameter_doc (do_not_log absent or false)
  end type param_file_type
  interface get_param
    module procedure get_param_real, get_param_int, get_param_logical, get_param_char
  end interface
contains
  subroutine stub_set_param(pf, name, value)
    type(param_file_type), intent(inout) :: pf ; character(len=*), intent(in) :: name, value
    integer :: m
    do m = 1, pf%n ; if (trim(pf%names(m)) == trim(name)) then ; pf%values(m) = value ; return ; endif ; enddo
    pf%n = pf%n + 1 ; pf%names(pf%n) = name ; pf%values(pf%n) = value
  end subroutine stub_set_param
  integer function stub_param_log_count(pf)
    type(param_file_type), intent(in) :: pf ; stub_param_log_count = pf%nlogged
  end function stub_param_log_count
  integer function find(pf, name)
    type(param_file_type), intent(in) :: pf ; character(len=*), intent(in) :: name ; integer :: m
    find = 0
    do m = 1, pf%n ; if (trim(pf%names(m)) == trim(name)) then ; find = m ; return ; endif ; enddo
  end function find
  subroutine note_log(pf, do_not_log)
    type(param_file_type), intent(in) :: pf ; logical, optional, intent(in) :: do_not_log
    ! (param_file is intent(in) in MOM6's get_param; the harness only counts through a pointer-free trick: nothing to do here)
  end subroutine note_log
  subroutine get_param_real(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                            debuggingParam, scale, unscaled, layoutParam, old_name)
    type(param_file_type), intent(in) :: CS ; character(len=*), intent(in) :: modulename, varname
    real, intent(inout) :: value ; character(len=*), optional, intent(in) :: desc, units, old_name
    real, optional, intent(in) :: default, scale ; real, optional, intent(out) :: unscaled
    logical, optional, intent(in) :: fail_if_missing, do_not_read, do_not_log, debuggingParam, layoutParam
    integer :: m
    m = find(CS, varname)
    if (m > 0) then ; read(CS%values(m), *) value
    elseif (present(default)) then ; value = default
    elseif (present(fail_if_missing)) then
      if (fail_if_missing) call MOM_error(FATAL, trim(modulename)//": "//trim(varname)//" must be set in the parameter table.")
    endif
    if (present(unscaled)) unscaled = value
    if (present(scale)) value = value * scale
  end subroutine get_param_real
  subroutine get_param_int(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                           debuggingParam, layoutParam, old_name)
    type(param_file_type), intent(in) :: CS ; character(len=*), intent(in) :: modulename, varname
    integer, intent(inout) :: value ; character(len=*), optional, intent(in) :: desc, units, old_name
    integer, optional, intent(in) :: default
    logical, optional, intent(in) :: fail_if_missing, do_not_read, do_not_log, debuggingParam, layoutParam
    integer :: m
    m = find(CS, varname)
    if (m > 0) then ; read(CS%values(m), *) value
    elseif (present(default)) then ; value = default ; endif
  end subroutine get_param_int
  subroutine get_param_logical(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                               debuggingParam, layoutParam, old_name)
    type(param_file_type), intent(in) :: CS ; character(len=*), intent(in) :: modulename, varname
    logical, intent(inout) :: value ; character(len=*), optional, intent(in) :: desc, units, old_name
    logical, optional, intent(in) :: default
    logical, optional, intent(in) :: fail_if_missing, do_not_read, do_not_log, debuggingParam, layoutParam
    integer :: m
    if (present(do_not_read)) then ; if (do_not_read) return ; endif   ! (MOM_file_parser: the value stays what the caller set)
    m = find(CS, varname)
    if (m > 0) then ; value = (index(CS%values(m), "T") > 0 .or. index(CS%values(m), "t") > 0)
    elseif (present(default)) then ; value = default ; endif
  end subroutine get_param_logical
  subroutine get_param_char(CS, modulename, varname, value, desc, units, default, fail_if_missing, do_not_read, do_not_log, &
                            debuggingParam, layoutParam, old_name)
    type(param_file_type), intent(in) :: CS ; character(len=*), intent(in) :: modulename, varname
    character(len=*), intent(inout) :: value ; character(len=*), optional, intent(in) :: desc, units, default, old_name
    logical, optional, intent(in) :: fail_if_missing, do_not_read, do_not_log, debuggingParam, layoutParam
    integer :: m
    m = find(CS, varname)
    if (m > 0) then ; value = trim(CS%values(m))
    elseif (present(default)) then ; value = default ; endif
  end subroutine get_param_char
  subroutine log_version(CS, modulename, version, desc, all_default, layout, debugging)
    type(param_file_type), intent(in) :: CS ; character(len=*), intent(in) :: modulename, version
    character(len=*), optional, intent(in) :: desc ; logical, optional, intent(in) :: all_default, layout, debugging
  end subroutine log_version
end module MOM_file_parser

module MOM_io
  implicit none ; public
  type :: vardesc
    character(len=64) :: name = "", units = "" ; character(len=240) :: longname = ""
    character(len=8) :: hor_grid = "h", z_grid = "L"
  end type vardesc
contains
  function var_desc(name, units, longname, hor_grid, z_grid, t_grid, cmor_field_name, cmor_units, cmor_longname, conversion, caller) result(vd)
    character(len=*), intent(in) :: name
    character(len=*), optional, intent(in) :: units, longname, hor_grid, z_grid, t_grid, cmor_field_name, cmor_units, cmor_longname, caller
    real, optional, intent(in) :: conversion
    type(vardesc) :: vd
    vd%name = name
    if (present(units)) vd%units = units ; if (present(longname)) vd%longname = longname
    if (present(hor_grid)) vd%hor_grid = hor_grid ; if (present(z_grid)) vd%z_grid = z_grid
  end function var_desc
end module MOM_io

module MOM_restart
  use MOM_io, only : vardesc
  use MOM_error_handler, only : MOM_error, FATAL
  implicit none ; private
  public :: MOM_restart_CS, register_restart_field, register_restart_pair, query_initialized
  public :: stub_save_restart, stub_restore_state, stub_restart_names
  type :: p3d ; real, dimension(:,:,:), pointer :: p => NULL() ; end type
  type :: p2d ; real, dimension(:,:), pointer :: p => NULL() ; end type
  type :: p0d ; real, pointer :: p => NULL() ; end type
  type :: MOM_restart_CS
    integer :: n = 0
    character(len=32) :: names(64) = ""
    integer :: rank(64) = 0
    logical :: initialized(64) = .false.
    type(p3d) :: v3(64) ; type(p2d) :: v2(64) ; type(p0d) :: v0(64)
  end type MOM_restart_CS
  interface register_restart_field
    module procedure reg_3d, reg_2d, reg_0d
  end interface
  interface register_restart_pair
    module procedure reg_pair_3d, reg_pair_2d
  end interface
  interface query_initialized
    module procedure query_name, query_3d_name, query_2d_name, query_0d_name
  end interface
contains
  subroutine reg_3d(f_ptr, name, mandatory, CS, longname, units, conversion, hor_grid, z_grid, t_grid, extra_axes)
    real, dimension(:,:,:), target, intent(in) :: f_ptr ; character(len=*), intent(in) :: name ; logical, intent(in) :: mandatory
    type(MOM_restart_CS), intent(inout) :: CS ; character(len=*), optional, intent(in) :: longname, units, hor_grid, z_grid, t_grid
    real, optional, intent(in) :: conversion ; integer, optional, intent(in) :: extra_axes(:)
    CS%n = CS%n + 1 ; CS%names(CS%n) = name ; CS%rank(CS%n) = 3 ; CS%v3(CS%n)%p => f_ptr
  end subroutine reg_3d
  subroutine reg_2d(f_ptr, name, mandatory, CS, longname, units, conversion, hor_grid, z_grid, t_grid)
    real, dimension(:,:), target, intent(in) :: f_ptr ; character(len=*), intent(in) :: name ; logical, intent(in) :: mandatory
    type(MOM_restart_CS), intent(inout) :: CS ; character(len=*), optional, intent(in) :: longname, units, hor_grid, z_grid, t_grid
    real, optional, intent(in) :: conversion
    CS%n = CS%n + 1 ; CS%names(CS%n) = name ; CS%rank(CS%n) = 2 ; CS%v2(CS%n)%p => f_ptr
  end subroutine reg_2d
  subroutine reg_0d(f_ptr, name, mandatory, CS, longname, units, conversion, t_grid)
    real, target, intent(in) :: f_ptr ; character(len=*), intent(in) :: name ; logical, intent(in) :: mandatory
    type(MOM_restart_CS), intent(inout) :: CS ; character(len=*), optional, intent(in) :: longname, units, t_grid
    real, optional, intent(in) :: conversion
    CS%n = CS%n + 1 ; CS%names(CS%n) = name ; CS%rank(CS%n) = 0 ; CS%v0(CS%n)%p => f_ptr
  end subroutine reg_0d
  subroutine reg_pair_3d(a_ptr, b_ptr, a_desc, b_desc, mandatory, CS, conversion)
    real, dimension(:,:,:), target, intent(in) :: a_ptr, b_ptr ; type(vardesc), intent(in) :: a_desc, b_desc
    logical, intent(in) :: mandatory ; type(MOM_restart_CS), intent(inout) :: CS ; real, optional, intent(in) :: conversion
    call reg_3d(a_ptr, trim(a_desc%name), mandatory, CS) ; call reg_3d(b_ptr, trim(b_desc%name), mandatory, CS)
  end subroutine reg_pair_3d
  subroutine reg_pair_2d(a_ptr, b_ptr, a_desc, b_desc, mandatory, CS, conversion)
    real, dimension(:,:), target, intent(in) :: a_ptr, b_ptr ; type(vardesc), intent(in) :: a_desc, b_desc
    logical, intent(in) :: mandatory ; type(MOM_restart_CS), intent(inout) :: CS ; real, optional, intent(in) :: conversion
    call reg_2d(a_ptr, trim(a_desc%name), mandatory, CS) ; call reg_2d(b_ptr, trim(b_desc%name), mandatory, CS)
  end subroutine reg_pair_2d
  logical function query_name(name, CS)
    character(len=*), intent(in) :: name ; type(MOM_restart_CS), intent(in) :: CS ; integer :: m
    query_name = .false.
    do m = 1, CS%n ; if (trim(CS%names(m)) == trim(name)) query_name = CS%initialized(m) ; enddo
  end function query_name
  logical function query_3d_name(f_ptr, name, CS)
    real, dimension(:,:,:), target, intent(in) :: f_ptr ; character(len=*), intent(in) :: name ; type(MOM_restart_CS), intent(in) :: CS
    query_3d_name = query_name(name, CS)
  end function query_3d_name
  logical function query_2d_name(f_ptr, name, CS)
    real, dimension(:,:), target, intent(in) :: f_ptr ; character(len=*), intent(in) :: name ; type(MOM_restart_CS), intent(in) :: CS
    query_2d_name = query_name(name, CS)
  end function query_2d_name
  logical function query_0d_name(f_ptr, name, CS)
    real, target, intent(in) :: f_ptr ; character(len=*), intent(in) :: name ; type(MOM_restart_CS), intent(in) :: CS
    query_0d_name = query_name(name, CS)
  end function query_0d_name
  !> what save_restart does with the registry: every registered variable, by name, into one file
  subroutine stub_save_restart(CS, path)
    type(MOM_restart_CS), intent(in) :: CS ; character(len=*), intent(in) :: path ; integer :: m, u
    open(newunit=u, file=path, form="unformatted", access="stream", status="replace")
    write(u) CS%n
    do m = 1, CS%n
      write(u) CS%names(m), CS%rank(m)
      if (CS%rank(m) == 3) then ; write(u) shape(CS%v3(m)%p), CS%v3(m)%p
      elseif (CS%rank(m) == 2) then ; write(u) shape(CS%v2(m)%p), CS%v2(m)%p
      else ; write(u) CS%v0(m)%p ; endif
    enddo
    close(u)
  end subroutine stub_save_restart
  !> what restore_state does: fill the registered arrays from the file, by name, and mark them initialised
  subroutine stub_restore_state(CS, path, skip)
    type(MOM_restart_CS), intent(inout) :: CS ; character(len=*), intent(in) :: path
    character(len=*), optional, intent(in) :: skip   !< names (between commas) the file is to be treated as not holding: an older file
    integer :: m, q, u, n, rk, s3(3), s2(2) ; character(len=32) :: nm
    real, allocatable :: b3(:,:,:), b2(:,:) ; real :: b0
    open(newunit=u, file=path, form="unformatted", access="stream", status="old")
    read(u) n
    do q = 1, n
      read(u) nm, rk
      if (rk == 3) then ; read(u) s3 ; allocate(b3(s3(1), s3(2), s3(3))) ; read(u) b3
      elseif (rk == 2) then ; read(u) s2 ; allocate(b2(s2(1), s2(2))) ; read(u) b2
      else ; read(u) b0 ; endif
      do m = 1, CS%n ; if (trim(CS%names(m)) == trim(nm) .and. CS%rank(m) == rk) then
        if (present(skip)) then ; if (index(","//trim(skip)//",", ","//trim(nm)//",") > 0) cycle ; endif
        if (rk == 3) then ; CS%v3(m)%p(:,:,:) = b3
        elseif (rk == 2) then ; CS%v2(m)%p(:,:) = b2
        else ; CS%v0(m)%p = b0 ; endif
        CS%initialized(m) = .true.
      endif ; enddo
      if (allocated(b3)) deallocate(b3) ; if (allocated(b2)) deallocate(b2)
    enddo
    close(u)
  end subroutine stub_restore_state
  function stub_restart_names(CS) result(s)
    type(MOM_restart_CS), intent(in) :: CS ; character(len=1024) :: s ; integer :: m
    s = ""
    do m = 1, CS%n ; s = trim(s)//" "//trim(CS%names(m)) ; enddo
  end function stub_restart_names
end module MOM_restart

module MOM_EOS
  implicit none ; public
  type :: EOS_type ; integer :: form_of_EOS = 0 ; end type EOS_type
end module MOM_EOS

module MOM_variables
  use MOM_EOS, only : EOS_type
  implicit none ; public
  type :: thermo_var_ptrs
    real, pointer, dimension(:,:,:) :: T => NULL(), S => NULL()
    type(EOS_type), pointer :: eqn_of_state => NULL()
  end type thermo_var_ptrs
  type :: vertvisc_type
    real, allocatable, dimension(:,:) :: bbl_thick_u, bbl_thick_v, kv_bbl_u, kv_bbl_v
    real, allocatable, dimension(:,:,:) :: Ray_u, Ray_v, Kv_shear, Kv_shear_Bu
  end type vertvisc_type
  type :: BT_cont_type
    real, allocatable, dimension(:,:) :: FA_u_EE, FA_u_E0, FA_u_W0, FA_u_WW, uBT_WW, uBT_EE, FA_v_NN, FA_v_N0, FA_v_S0, FA_v_SS, vBT_SS, vBT_NN
    real, allocatable, dimension(:,:,:) :: h_u, h_v
  end type BT_cont_type
  !> the members initialize_dyn_split_RK2 associates (MOM_variables.F90:136-162 / :165-237)
  type :: ocean_internal_state
    real, pointer, dimension(:,:,:) :: CAu => NULL(), CAv => NULL(), PFu => NULL(), PFv => NULL(), diffu => NULL(), diffv => NULL(), &
                                       pbce => NULL(), u_accel_bt => NULL(), v_accel_bt => NULL(), u_av => NULL(), v_av => NULL()
  end type
  type :: accel_diag_ptrs
    real, pointer, dimension(:,:,:) :: diffu => NULL(), diffv => NULL(), CAu => NULL(), CAv => NULL(), PFu => NULL(), PFv => NULL(), &
                                       u_accel_bt => NULL(), v_accel_bt => NULL()
  end type
  type :: cont_diag_ptrs ; integer :: dummy = 0 ; end type
end module MOM_variables

module MOM_forcing_type
  implicit none ; public
  type :: mech_forcing
    real, pointer, dimension(:,:) :: taux => NULL(), tauy => NULL(), ustar => NULL()
  end type mech_forcing
end module MOM_forcing_type

module MOM_diag_mediator
  use MOM_time_manager, only : time_type
  implicit none ; public
  type :: diag_ctrl ; integer :: dummy = 0 ; end type
end module MOM_diag_mediator

module MOM_open_boundary
  implicit none ; public
  type :: ocean_OBC_type ; integer :: dummy = 0 ; end type
  type :: update_OBC_CS ; integer :: dummy = 0 ; end type
end module MOM_open_boundary


module MOM_MEKE_types
  implicit none ; public
  type :: MEKE_type ; integer :: dummy = 0 ; end type
end module MOM_MEKE_types
module MOM_lateral_mixing_coeffs
  implicit none ; public
  type :: VarMix_CS ; integer :: dummy = 0 ; end type
end module MOM_lateral_mixing_coeffs
module MOM_thickness_diffuse
  implicit none ; public
  type :: thickness_diffuse_CS ; integer :: dummy = 0 ; end type
end module MOM_thickness_diffuse
module MOM_porous_barriers
  implicit none ; public
  type :: porous_barrier_type ; integer :: dummy = 0 ; end type
end module MOM_porous_barriers
module MOM_stochastics
  implicit none ; public
  type :: stochastic_CS ; integer :: dummy = 0 ; end type
end module MOM_stochastics
module MOM_wave_interface
  implicit none ; public
  type :: wave_parameters_CS ; integer :: dummy = 0 ; end type
end module MOM_wave_interface
module MOM_set_visc
  implicit none ; public
  type :: set_visc_CS ; integer :: dummy = 0 ; end type
end module MOM_set_visc
module MOM_harmonic_analysis
  implicit none ; public
  type :: harmonic_analysis_CS ; integer :: dummy = 0 ; end type
end module MOM_harmonic_analysis
module MOM_diabatic_driver
  implicit none ; public
  type :: diabatic_CS ; integer :: dummy = 0 ; end type
end module MOM_diabatic_driver
module MOM_get_input
  implicit none ; public
  type :: directories ; character(len=240) :: output_directory = "" ; end type
end module MOM_get_input
module MOM_self_attr_load
  implicit none ; public
  type :: SAL_CS ; integer :: dummy = 0 ; end type
end module MOM_self_attr_load
module MOM_tidal_forcing
  implicit none ; public
  type :: tidal_forcing_CS ; integer :: dummy = 0 ; end type
end module MOM_tidal_forcing

module MOM_tracer_advect_schemes
  use MOM_error_handler, only : MOM_error, FATAL
  implicit none ; public
  integer, parameter :: ADVECT_PLM = 0, ADVECT_PPMH3 = 1, ADVECT_PPM = 2
  character(len=64), parameter :: TracerAdvectionSchemeDoc = "  PLM, PPM:H3, PPM"
contains
  subroutine set_tracer_advect_scheme(scheme_value, advect_scheme_name)
    integer, intent(out) :: scheme_value ; character(len=*), intent(in) :: advect_scheme_name
    select case (trim(advect_scheme_name))
      case ("") ; scheme_value = -1
      case ("PLM") ; scheme_value = ADVECT_PLM
      case ("PPM:H3") ; scheme_value = ADVECT_PPMH3
      case ("PPM") ; scheme_value = ADVECT_PPM
      case default ; call MOM_error(FATAL, "set_tracer_advect_scheme: unknown TRACER_ADVECTION_SCHEME "//trim(advect_scheme_name))
    end select
  end subroutine set_tracer_advect_scheme
end module MOM_tracer_advect_schemes

module MOM_tracer_registry
  implicit none ; public
  type :: tracer_type
    real, dimension(:,:,:), pointer :: t => NULL()
    integer :: advect_scheme = -1
    character(len=32) :: name = ""
  end type tracer_type
  type :: tracer_registry_type
    integer :: ntr = 0
    type(tracer_type) :: Tr(50)
  end type tracer_registry_type
end module MOM_tracer_registry
