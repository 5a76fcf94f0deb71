! varden_boxlib_ext.f90 -- FBoxLib / VARDEN modules BEYOND the hot path's own that the reference's callers `use` and that are thin over the C-ABI.
!
! tools/check_reference_callers.py runs `flang -fsyntax-only` on the reference's own callers of the hot path (src/varden.f90, src/advance_timestep.f90,
! src/initialize.f90, src/regrid.f90, ... read IN PLACE from the reference tree, nothing copied) against the modules of varden_boxlib.f90 and of this
! file, and lists what is still missing (profiles/r06_reference_callers_syntax.txt; INTEGRATION.md section 2 reads it).  The modules here:
!   parallel            parallel_myproc / nprocs / IOProcessor / IOProcessorNode / wtime / barrier / reduce (MPI_MAX, MPI_MIN, MPI_SUM), one rank per GPU:
!                       the ranks of varden_amd_initialize; reductions across ranks go through vdn_comm_allreduce_max (RCCL)
!   BoxLib              boxlib_initialize / boxlib_finalize (src/main.f90:8,17): rank and device from the launcher's environment (RANK / LOCAL_RANK /
!                       WORLD_SIZE as torchrun and bench.py export them, else one rank)
!   bl_error_module     bl_error / bl_warn / bl_assert
!   bl_prof_module      type bl_prof_timer, build(bpt, name) / destroy(bpt) (src/advance_timestep.f90:58-60,99-101): the library opens the roctx ranges of
!                       these names itself (runtime.hip: Prof), so the host-side timers are empty here
!   bl_IO_module        unit_new
!   vort_module         make_vorticity(vort, comp, u, dx, bc) / make_magvel(magvel, comp, u)   (src/makevort.f90:16,58)
!   fillpatch_module, ml_prolongation_module   fillpatch(fine, crse, ...) / ml_nodal_prolongation(fine, crse, ir)   (src/regrid.f90:279-280, 317-344)
! probin_module (varden_boxlib.f90) carries every entry of src/_parameters with its default, the &PROBIN namelist, probin_init and probin_close.
! NOT here (listed by the report as the maintainer's remaining work): FBoxLib's host-side box calculus and I/O (list_box_module, box_util_module, fabio_module,
! plotfile_module, checkpoint / restart, bl_mem_stat, bl_timer), layouts built one level at a time (layout_build_ba, make_new_grids_module, tag_boxes_module --
! the library builds whole hierarchies: vdn_make_new_grids), and the modules INSIDE advance_timestep (pre_advance_module, scalar_advance_module, ...): the
! boundary is advance_timestep itself, their kernels are the vdn_k_* hooks of include/varden_amd.h.

module BoxLib
  use parallel
  use bl_error_module
  implicit none
contains
  ! src/main.f90:8: one rank per GPU; the rank, the world size and the device come from the launcher (torchrun / bench.py / mpirun export one of these sets)
  subroutine boxlib_initialize()
    integer :: rank, nranks
    rank = env_int('RANK', env_int('OMPI_COMM_WORLD_RANK', 0))
    nranks = env_int('WORLD_SIZE', env_int('OMPI_COMM_WORLD_SIZE', 1))
    call parallel_set_ranks(rank, nranks)
  end subroutine boxlib_initialize
  subroutine boxlib_finalize()
  end subroutine boxlib_finalize
  integer function boxlib_device()
    boxlib_device = env_int('LOCAL_RANK', env_int('OMPI_COMM_WORLD_LOCAL_RANK', 0))
  end function boxlib_device
  integer function env_int(name, dflt)
    character(len=*), intent(in) :: name
    integer, intent(in) :: dflt
    character(len=32) :: v
    integer :: st, ios
    env_int = dflt
    call get_environment_variable(name, v, status=st)
    if (st /= 0) return
    read(v, *, iostat=ios) env_int
    if (ios /= 0) env_int = dflt
  end function env_int
end module BoxLib

module vort_module
  use multifab_module
  use define_bc_module
  use varden_amd, only: vamd_make_vorticity => make_vorticity, vamd_make_magvel => make_magvel
  implicit none
contains
  ! make_vorticity(vort, comp, u, dx, bc)   (src/makevort.f90:16-22)
  subroutine make_vorticity(vort, comp, u, dx, bc)
    integer, intent(in) :: comp
    type(multifab), intent(inout) :: vort
    type(multifab), intent(inout) :: u
    real(dp_t), intent(in) :: dx(:)
    type(bc_level), intent(in) :: bc
    call vamd_make_vorticity(vort%v, comp, u%v, dx, as_vamd_tower(bc))
  end subroutine make_vorticity
  ! make_magvel(magvel, comp, u)   (src/makevort.f90:58-62)
  subroutine make_magvel(magvel, comp, u)
    integer, intent(in) :: comp
    type(multifab), intent(inout) :: magvel
    type(multifab), intent(inout) :: u
    call vamd_make_magvel(magvel%v, comp, u%v)
  end subroutine make_magvel
end module vort_module

module fillpatch_module
  use multifab_module
  use define_bc_module
  use varden_amd, only: vamd_fillpatch => fillpatch
  implicit none
contains
  ! fillpatch(fine, crse, ng, ir, bc_crse, bc_fine, icomp_fine, icomp_crse, bcomp, nc)   (FBoxLib; src/regrid.f90:317-330): every point of the new fine level
  ! from the coarse level (the caller then copies the old fine data over it).  The coarse and fine components coincide at every call site of the reference.
  subroutine fillpatch(fine, crse, ng, ir, bc_crse, bc_fine, icomp_fine, icomp_crse, bcomp, nc, no_final_physbc_input, lim_slope_input, lin_limit_input, &
                       fill_crse_input, stencil_width_input, fourth_order_input)
    type(multifab), intent(inout) :: fine, crse
    integer, intent(in) :: ng, ir(:), icomp_fine, icomp_crse, bcomp, nc
    type(bc_level), intent(in) :: bc_crse, bc_fine
    ! FBoxLib's options; the reference passes no_final_physbc_input = .true. only (the library's fillpatch applies no physical boundary: the caller does)
    logical, intent(in), optional :: no_final_physbc_input, lim_slope_input, lin_limit_input, fill_crse_input, fourth_order_input
    integer, intent(in), optional :: stencil_width_input
    if (any(ir /= 2)) error stop 'fillpatch: refinement ratio 2 only'
    if (icomp_fine /= icomp_crse) error stop 'fillpatch: the coarse and fine components must coincide'
    call vamd_fillpatch(fine%v, crse%v, icomp_fine, nc)
  end subroutine fillpatch
end module fillpatch_module

module ml_prolongation_module
  use multifab_module
  use varden_amd, only: vamd_nodal_prolongation => ml_nodal_prolongation
  implicit none
contains
  ! ml_nodal_prolongation(fine, crse, ir)   (FBoxLib; src/regrid.f90:342-344: the pressure of a new fine level)
  subroutine ml_nodal_prolongation(fine, crse, ir)
    type(multifab), intent(inout) :: fine, crse
    integer, intent(in) :: ir(:)
    if (any(ir /= 2)) error stop 'ml_nodal_prolongation: refinement ratio 2 only'
    call vamd_nodal_prolongation(fine%v, crse%v)
  end subroutine ml_nodal_prolongation
end module ml_prolongation_module
