! varden_amd_mod.f90 -- Fortran (ISO_C_BINDING) host side of the MI355X-native VARDEN hot path.
!
! This is the module a VARDEN maintainer `use`s instead of the reference's
!     advance_module (src/advance_timestep.f90), estdt_module (src/estdt.f90),
!     hgproject_module (src/hgproject.f90) and of the FBoxLib containers those take
!     (multifab_module, ml_layout_module, define_bc_module).
! Names, argument order and meaning follow the reference; every routine forwards to the C-ABI of
! include/varden_amd.h (libvarden_amd.so, hand-written HIP for gfx950).  All multifab data lives in
! HBM: `dataptr` returns the DEVICE address (type(c_ptr)); use multifab_copy_to_host /
! multifab_copy_from_host for host access.  A non-zero C return code becomes `error stop` with the
! library's message (the reference calls bl_error, e.g. src/hgproject.f90:502).
!
! Differences from the reference that a caller sees:
!   * boxes are 0-based index boxes (lo(3), hi(3)) as in BoxLib; components are 1-based here
!     (Fortran side) and converted to the C-ABI's 0-based components inside this module;
!   * only dm = 3 and single-level hierarchies are implemented in this round.
module varden_amd
  use iso_c_binding
  implicit none
  private

  integer, parameter, public :: dp_t = c_double

  ! bc_module constants
  integer, parameter, public :: PERIODIC = -1, INTERIOR = 0, INLET = 11, OUTLET = 12, SYMMETRY = 13, &
                                SLIP_WALL = 14, NO_SLIP_WALL = 15
  integer, parameter, public :: REFLECT_ODD = 20, REFLECT_EVEN = 21, FOEXTRAP = 22, EXT_DIR = 23, HOEXTRAP = 24
  integer, parameter, public :: BC_PER = -1, BC_INT = 0, BC_DIR = 1, BC_NEU = 2
  ! proj_parameters (src/proj_parameters.f90)
  integer, parameter, public :: initial_projection = 1, divu_iters = 2, pressure_iters = 3, regular_timestep = 4

  ! mirror of vdn_params (include/varden_amd.h) == the probin_module values the hot path reads
  type, bind(C), public :: vdn_params
     integer(c_int) :: dm, nscal, slope_order, use_minion, boussinesq, stencil_order, diffusion_type
     integer(c_int) :: verbose, mg_verbose, prob_type
     real(c_double) :: visc_coef, diff_coef, cflfac, max_dt_growth
     real(c_double) :: u_bc(2,3), v_bc(2,3), w_bc(2,3), rho_bc(2,3), trac_bc(2,3)   ! C [dir][side] == Fortran (side,dir)
     integer(c_int) :: mg_nu1, mg_nu2, mg_nub, mg_max_iter, hg_max_iter, hg_nu1, hg_nu2, hg_nub
     real(c_double) :: hg_omega, mac_rel_eps, hg_rel_eps
     integer(c_int) :: abort_on_max_iter, hg_fmg, mac_fmg
     real(c_double) :: hg_omega_pre1, hg_omega_pre2, hg_omega_fac1, hg_omega_fac2, hg_omega_fac3
     integer(c_int) :: mg_predict
  end type vdn_params

  type, bind(C), public :: vdn_box
     integer(c_int) :: lo(3), hi(3)
  end type vdn_box

  type, public :: ml_layout
     type(c_ptr) :: h = c_null_ptr
     integer :: nlevel = 0, dim = 3
  end type ml_layout

  type, public :: multifab
     type(c_ptr) :: h = c_null_ptr
     integer :: dim = 3, nc = 1, ng = 0
  end type multifab

  type, public :: bc_tower
     type(c_ptr) :: h = c_null_ptr
  end type bc_tower

  public :: varden_amd_initialize, varden_amd_finalize, probin_defaults, varden_amd_set_extruded_2d
  public :: ml_layout_build, ml_layout_destroy
  public :: bc_tower_build, bc_tower_destroy
  public :: multifab_build, multifab_build_edge, multifab_build_nodal, multifab_destroy, nfabs, get_box, dataptr, &
            setval, multifab_copy_c, norm_inf, norm_inf_c, multifab_fill_boundary, multifab_physbc, &
            multifab_copy_to_host, multifab_copy_from_host, multifab_fab_size
  public :: advance_timestep, estdt, hgproject, macproject
  public :: ml_cc_restriction, ml_edge_restriction, multifab_fill_ghost_cells, create_umac_grown, ml_restrict_and_fill
  public :: fillpatch, ml_nodal_prolongation, multifab_copy_layouts, make_new_grids, make_vorticity, make_magvel

  interface
     subroutine vdn_params_default(p) bind(C, name="vdn_params_default")
       import :: vdn_params
       type(vdn_params), intent(out) :: p
     end subroutine
     integer(c_int) function vdn_init(p, rank, nranks, device) bind(C, name="vdn_init")
       import :: vdn_params, c_int
       type(vdn_params), intent(in) :: p
       integer(c_int), value :: rank, nranks, device
     end function
     integer(c_int) function vdn_finalize() bind(C, name="vdn_finalize")
       import :: c_int
     end function
     integer(c_int) function vdn_set_extruded_2d(on) bind(C, name="vdn_set_extruded_2d")   ! a 2-D problem as its z-uniform 3-D copy: velpred_2d's hi-x OUTLET rule
       import :: c_int
       integer(c_int), value :: on
     end function
     type(c_ptr) function vdn_last_error() bind(C, name="vdn_last_error")
       import :: c_ptr
     end function
     integer(c_int) function vdn_layout_create(nlev, rr, pd, nboxes, boxes, owner, pmask, out) bind(C, name="vdn_layout_create")
       import :: c_int, c_ptr, vdn_box
       integer(c_int), value :: nlev
       integer(c_int), intent(in) :: rr(*), nboxes(*), owner(*), pmask(*)
       type(vdn_box), intent(in) :: pd(*), boxes(*)
       type(c_ptr), intent(out) :: out
     end function
     integer(c_int) function vdn_layout_destroy(la) bind(C, name="vdn_layout_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: la
     end function
     integer(c_int) function vdn_bc_tower_create(la, phys_bc, out) bind(C, name="vdn_bc_tower_create")
       import :: c_int, c_ptr
       type(c_ptr), value :: la
       integer(c_int), intent(in) :: phys_bc(*)
       type(c_ptr), intent(out) :: out
     end function
     integer(c_int) function vdn_bc_tower_destroy(b) bind(C, name="vdn_bc_tower_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: b
     end function
     integer(c_int) function vdn_multifab_create(la, lev, nc, ng, nodal, out) bind(C, name="vdn_multifab_create")
       import :: c_int, c_ptr
       type(c_ptr), value :: la
       integer(c_int), value :: lev, nc, ng
       integer(c_int), intent(in) :: nodal(3)
       type(c_ptr), intent(out) :: out
     end function
     integer(c_int) function vdn_multifab_destroy(mf) bind(C, name="vdn_multifab_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: mf
     end function
     integer(c_int) function vdn_multifab_nfabs(mf) bind(C, name="vdn_multifab_nfabs")
       import :: c_int, c_ptr
       type(c_ptr), value :: mf
     end function
     integer(c_int) function vdn_multifab_get_box(mf, i, bx) bind(C, name="vdn_multifab_get_box")
       import :: c_int, c_ptr, vdn_box
       type(c_ptr), value :: mf
       integer(c_int), value :: i
       type(vdn_box), intent(out) :: bx
     end function
     integer(c_long) function vdn_multifab_fab_size(mf, i) bind(C, name="vdn_multifab_fab_size")
       import :: c_int, c_long, c_ptr
       type(c_ptr), value :: mf
       integer(c_int), value :: i
     end function
     integer(c_int) function vdn_multifab_dataptr(mf, i, dev) bind(C, name="vdn_multifab_dataptr")
       import :: c_int, c_ptr
       type(c_ptr), value :: mf
       integer(c_int), value :: i
       type(c_ptr), intent(out) :: dev
     end function
     integer(c_int) function vdn_multifab_copy_to_host(mf, i, host) bind(C, name="vdn_multifab_copy_to_host")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: mf
       integer(c_int), value :: i
       real(c_double), intent(out) :: host(*)
     end function
     integer(c_int) function vdn_multifab_copy_from_host(mf, i, host) bind(C, name="vdn_multifab_copy_from_host")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: mf
       integer(c_int), value :: i
       real(c_double), intent(in) :: host(*)
     end function
     integer(c_int) function vdn_multifab_setval(mf, val, comp, nc, all) bind(C, name="vdn_multifab_setval")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: mf
       real(c_double), value :: val
       integer(c_int), value :: comp, nc, all
     end function
     integer(c_int) function vdn_multifab_copy_c(dst, dcomp, src, scomp, nc, ng) bind(C, name="vdn_multifab_copy_c")
       import :: c_int, c_ptr
       type(c_ptr), value :: dst, src
       integer(c_int), value :: dcomp, scomp, nc, ng
     end function
     integer(c_int) function vdn_multifab_norm_inf(mf, comp, nc, out) bind(C, name="vdn_multifab_norm_inf")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: mf
       integer(c_int), value :: comp, nc
       real(c_double), intent(out) :: out
     end function
     integer(c_int) function vdn_multifab_fill_boundary(mf) bind(C, name="vdn_multifab_fill_boundary")
       import :: c_int, c_ptr
       type(c_ptr), value :: mf
     end function
     integer(c_int) function vdn_multifab_physbc(mf, scomp, bccomp, nc, bct) bind(C, name="vdn_multifab_physbc")
       import :: c_int, c_ptr
       type(c_ptr), value :: mf, bct
       integer(c_int), value :: scomp, bccomp, nc
     end function
     integer(c_int) function vdn_advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, evf, esf, bct, dt, time, dx, &
                                                  press_comp, proj_type) bind(C, name="vdn_advance_timestep")
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: istep, press_comp, proj_type
       type(c_ptr), value :: mla, bct
       type(c_ptr), intent(in) :: sold(*), uold(*), snew(*), unew(*), gp(*), p(*), evf(*), esf(*)
       real(c_double), value :: dt, time
       real(c_double), intent(in) :: dx(*)
     end function
     integer(c_int) function vdn_estdt(lev, u, s, gp, evf, dx, dtold, dt) bind(C, name="vdn_estdt")
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: lev
       type(c_ptr), value :: u, s, gp, evf
       real(c_double), intent(in) :: dx(3)
       real(c_double), value :: dtold
       real(c_double), intent(out) :: dt
     end function
     integer(c_int) function vdn_hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, bct, press_comp) &
                                           bind(C, name="vdn_hgproject")
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: proj_type, press_comp
       type(c_ptr), value :: mla, bct
       type(c_ptr), intent(in) :: unew(*), uold(*), rhohalf(*), p(*), gp(*)
       real(c_double), intent(in) :: dx(*)
       real(c_double), value :: dt
     end function
     integer(c_int) function vdn_macproject(mla, umac, rho, mac_rhs, dx, bct, bc_comp) bind(C, name="vdn_macproject")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: mla, bct
       type(c_ptr), intent(in) :: umac(*), rho(*), mac_rhs(*)
       real(c_double), intent(in) :: dx(*)
       integer(c_int), value :: bc_comp
     end function
     integer(c_int) function vdn_ml_cc_restriction(crse, fine, icomp, nc) bind(C, name="vdn_ml_cc_restriction")
       import :: c_int, c_ptr
       type(c_ptr), value :: crse, fine
       integer(c_int), value :: icomp, nc
     end function
     integer(c_int) function vdn_ml_edge_restriction(crse, fine, dir) bind(C, name="vdn_ml_edge_restriction")
       import :: c_int, c_ptr
       type(c_ptr), value :: crse, fine
       integer(c_int), value :: dir
     end function
     integer(c_int) function vdn_multifab_fill_ghost_cells(fine, crse, icomp, nc) bind(C, name="vdn_multifab_fill_ghost_cells")
       import :: c_int, c_ptr
       type(c_ptr), value :: fine, crse
       integer(c_int), value :: icomp, nc
     end function
     integer(c_int) function vdn_create_umac_grown(fine, crse, dir) bind(C, name="vdn_create_umac_grown")
       import :: c_int, c_ptr
       type(c_ptr), value :: fine, crse
       integer(c_int), value :: dir
     end function
     integer(c_int) function vdn_ml_restrict_and_fill(nlev, mf, icomp, bcomp, nc, same_boundary, bct) bind(C, name="vdn_ml_restrict_and_fill")
       import :: c_int, c_ptr
       integer(c_int), value :: nlev, icomp, bcomp, nc, same_boundary
       type(c_ptr), intent(in) :: mf(*)
       type(c_ptr), value :: bct
     end function
     integer(c_int) function vdn_fillpatch(fine, crse, icomp, nc) bind(C, name="vdn_fillpatch")
       import :: c_int, c_ptr
       type(c_ptr), value :: fine, crse
       integer(c_int), value :: icomp, nc
     end function
     integer(c_int) function vdn_make_vorticity(vort, comp, u, dx, bct) bind(C, name="vdn_make_vorticity")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: vort, u, bct
       integer(c_int), value :: comp
       real(c_double), intent(in) :: dx(*)
     end function
     integer(c_int) function vdn_make_magvel(magvel, comp, u) bind(C, name="vdn_make_magvel")
       import :: c_int, c_ptr
       type(c_ptr), value :: magvel, u
       integer(c_int), value :: comp
     end function
     integer(c_int) function vdn_ml_nodal_prolongation(fine, crse) bind(C, name="vdn_ml_nodal_prolongation")
       import :: c_int, c_ptr
       type(c_ptr), value :: fine, crse
     end function
     integer(c_int) function vdn_multifab_copy_layouts(dst, dcomp, src, scomp, nc) bind(C, name="vdn_multifab_copy_layouts")
       import :: c_int, c_ptr
       type(c_ptr), value :: dst, src
       integer(c_int), value :: dcomp, scomp, nc
     end function
     integer(c_int) function vdn_make_new_grids(s, lev1, buf_wid, nest, min_eff, min_width, blocking, max_grid_size, maxboxes, boxes_out, &
                                                nboxes_out, ntagged) bind(C, name="vdn_make_new_grids")
       import :: c_int, c_ptr, c_double, c_long, vdn_box
       type(c_ptr), value :: s
       integer(c_int), value :: lev1, buf_wid, nest, min_width, blocking, max_grid_size, maxboxes
       real(c_double), value :: min_eff
       type(vdn_box), intent(out) :: boxes_out(*)
       integer(c_int), intent(out) :: nboxes_out
       integer(c_long), intent(out) :: ntagged
     end function
     integer(c_size_t) function c_strlen(s) bind(C, name="strlen")
       import :: c_ptr, c_size_t
       type(c_ptr), value :: s
     end function
  end interface

contains

  ! ---- error convention: non-zero return => error stop with the library's message ----------------
  subroutine chk(rc, where)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: where
    type(c_ptr) :: msg
    character(kind=c_char), pointer :: s(:)
    integer :: n, i
    character(len=512) :: buf
    if (rc == 0) return
    msg = vdn_last_error()
    n = int(c_strlen(msg))
    call c_f_pointer(msg, s, [n])
    buf = ' '
    do i = 1, min(n, 512)
       buf(i:i) = s(i)
    end do
    write(*,*) 'varden_amd: ', where, ': ', trim(buf)
    error stop 1
  end subroutine chk

  subroutine probin_defaults(p)
    type(vdn_params), intent(out) :: p
    call vdn_params_default(p)
  end subroutine probin_defaults

  subroutine varden_amd_initialize(p, rank, nranks, device)
    type(vdn_params), intent(in) :: p
    integer, intent(in) :: rank, nranks, device
    call chk(vdn_init(p, int(rank, c_int), int(nranks, c_int), int(device, c_int)), 'vdn_init')
  end subroutine varden_amd_initialize

  ! the 3-D kernels run a z-uniform copy of a 2-D problem: velpred_2d's hi-x OUTLET rule (include/varden_amd.h: vdn_set_extruded_2d); after varden_amd_initialize
  subroutine varden_amd_set_extruded_2d(on)
    logical, intent(in) :: on
    call chk(vdn_set_extruded_2d(merge(1_c_int, 0_c_int, on)), 'vdn_set_extruded_2d')
  end subroutine varden_amd_set_extruded_2d

  subroutine varden_amd_finalize()
    call chk(vdn_finalize(), 'vdn_finalize')
  end subroutine varden_amd_finalize

  ! ---- ml_layout -------------------------------------------------------------------------------------
  subroutine ml_layout_build(mla, nlev, rr, pd, nboxes, boxes, owner, pmask)
    type(ml_layout), intent(out) :: mla
    integer, intent(in) :: nlev, rr(:), nboxes(:), owner(:)
    type(vdn_box), intent(in) :: pd(:), boxes(:)
    logical, intent(in) :: pmask(3)
    integer(c_int) :: pm(3)
    pm = merge(1, 0, pmask)
    call chk(vdn_layout_create(int(nlev, c_int), int(rr, c_int), pd, int(nboxes, c_int), boxes, int(owner, c_int), pm, mla%h), &
             'ml_layout_build')
    mla%nlevel = nlev
    mla%dim = 3
  end subroutine ml_layout_build

  subroutine ml_layout_destroy(mla)
    type(ml_layout), intent(inout) :: mla
    call chk(vdn_layout_destroy(mla%h), 'ml_layout_destroy')
    mla%h = c_null_ptr
  end subroutine ml_layout_destroy

  ! ---- bc_tower (define_bc_tower.f90): phys_bc(dir, side) as in the reference ---------------------------
  subroutine bc_tower_build(bct, mla, phys_bc)
    type(bc_tower), intent(out) :: bct
    type(ml_layout), intent(in) :: mla
    integer, intent(in) :: phys_bc(3, 2)
    integer(c_int) :: flat(6)
    integer :: d, s
    do d = 1, 3
       do s = 1, 2
          flat((d - 1) * 2 + s) = phys_bc(d, s)
       end do
    end do
    call chk(vdn_bc_tower_create(mla%h, flat, bct%h), 'bc_tower_build')
  end subroutine bc_tower_build

  subroutine bc_tower_destroy(bct)
    type(bc_tower), intent(inout) :: bct
    call chk(vdn_bc_tower_destroy(bct%h), 'bc_tower_destroy')
    bct%h = c_null_ptr
  end subroutine bc_tower_destroy

  ! ---- multifab ---------------------------------------------------------------------------------------
  subroutine multifab_build(mf, mla, lev, nc, ng)
    type(multifab), intent(out) :: mf
    type(ml_layout), intent(in) :: mla
    integer, intent(in) :: lev, nc, ng
    integer(c_int) :: nodal(3)
    nodal = 0
    call chk(vdn_multifab_create(mla%h, int(lev - 1, c_int), int(nc, c_int), int(ng, c_int), nodal, mf%h), 'multifab_build')
    mf%nc = nc; mf%ng = ng
  end subroutine multifab_build

  subroutine multifab_build_edge(mf, mla, lev, nc, ng, dir)
    type(multifab), intent(out) :: mf
    type(ml_layout), intent(in) :: mla
    integer, intent(in) :: lev, nc, ng, dir
    integer(c_int) :: nodal(3)
    nodal = 0
    nodal(dir) = 1
    call chk(vdn_multifab_create(mla%h, int(lev - 1, c_int), int(nc, c_int), int(ng, c_int), nodal, mf%h), 'multifab_build_edge')
    mf%nc = nc; mf%ng = ng
  end subroutine multifab_build_edge

  subroutine multifab_build_nodal(mf, mla, lev, nc, ng)
    type(multifab), intent(out) :: mf
    type(ml_layout), intent(in) :: mla
    integer, intent(in) :: lev, nc, ng
    integer(c_int) :: nodal(3)
    nodal = 1
    call chk(vdn_multifab_create(mla%h, int(lev - 1, c_int), int(nc, c_int), int(ng, c_int), nodal, mf%h), 'multifab_build_nodal')
    mf%nc = nc; mf%ng = ng
  end subroutine multifab_build_nodal

  subroutine multifab_destroy(mf)
    type(multifab), intent(inout) :: mf
    call chk(vdn_multifab_destroy(mf%h), 'multifab_destroy')
    mf%h = c_null_ptr
  end subroutine multifab_destroy

  integer function nfabs(mf)
    type(multifab), intent(in) :: mf
    nfabs = vdn_multifab_nfabs(mf%h)
  end function nfabs

  function get_box(mf, i) result(bx)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    type(vdn_box) :: bx
    call chk(vdn_multifab_get_box(mf%h, int(i - 1, c_int), bx), 'get_box')
  end function get_box

  function dataptr(mf, i) result(dev)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    type(c_ptr) :: dev
    call chk(vdn_multifab_dataptr(mf%h, int(i - 1, c_int), dev), 'dataptr')
  end function dataptr

  integer(c_long) function multifab_fab_size(mf, i)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    multifab_fab_size = vdn_multifab_fab_size(mf%h, int(i - 1, c_int))
  end function multifab_fab_size

  subroutine multifab_copy_to_host(mf, i, host)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    real(dp_t), intent(out) :: host(*)
    call chk(vdn_multifab_copy_to_host(mf%h, int(i - 1, c_int), host), 'multifab_copy_to_host')
  end subroutine multifab_copy_to_host

  subroutine multifab_copy_from_host(mf, i, host)
    type(multifab), intent(inout) :: mf
    integer, intent(in) :: i
    real(dp_t), intent(in) :: host(*)
    call chk(vdn_multifab_copy_from_host(mf%h, int(i - 1, c_int), host), 'multifab_copy_from_host')
  end subroutine multifab_copy_from_host

  ! setval(mf, val, [comp, nc], [all])
  subroutine setval(mf, val, comp, nc, all)
    type(multifab), intent(inout) :: mf
    real(dp_t), intent(in) :: val
    integer, intent(in), optional :: comp, nc
    logical, intent(in), optional :: all
    integer :: c, n, a
    c = 1; if (present(comp)) c = comp
    n = mf%nc - c + 1; if (present(nc)) n = nc
    a = 0; if (present(all)) a = merge(1, 0, all)
    call chk(vdn_multifab_setval(mf%h, val, int(c - 1, c_int), int(n, c_int), int(a, c_int)), 'setval')
  end subroutine setval

  ! multifab_copy_c(dst, dcomp, src, scomp, nc, ng)
  subroutine multifab_copy_c(dst, dcomp, src, scomp, nc, ng)
    type(multifab), intent(inout) :: dst
    type(multifab), intent(in) :: src
    integer, intent(in) :: dcomp, scomp, nc
    integer, intent(in), optional :: ng
    integer :: g
    g = 0; if (present(ng)) g = ng
    call chk(vdn_multifab_copy_c(dst%h, int(dcomp - 1, c_int), src%h, int(scomp - 1, c_int), int(nc, c_int), int(g, c_int)), &
             'multifab_copy_c')
  end subroutine multifab_copy_c

  real(dp_t) function norm_inf(mf)
    type(multifab), intent(in) :: mf
    call chk(vdn_multifab_norm_inf(mf%h, 0_c_int, int(mf%nc, c_int), norm_inf), 'norm_inf')
  end function norm_inf
  ! norm_inf(mf, comp, nc)   (FBoxLib; src/advance_timestep.f90:187: the max norm of nc components from comp, 1-based)
  real(dp_t) function norm_inf_c(mf, comp, nc)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: comp, nc
    call chk(vdn_multifab_norm_inf(mf%h, int(comp - 1, c_int), int(nc, c_int), norm_inf_c), 'norm_inf')
  end function norm_inf_c

  subroutine multifab_fill_boundary(mf)
    type(multifab), intent(inout) :: mf
    call chk(vdn_multifab_fill_boundary(mf%h), 'multifab_fill_boundary')
  end subroutine multifab_fill_boundary

  ! multifab_physbc(s, start_scomp, start_bccomp, num_comp, the_bc_tower)   (src/multifab_physbc.f90:17)
  subroutine multifab_physbc(s, start_scomp, start_bccomp, num_comp, the_bc_tower)
    type(multifab), intent(inout) :: s
    integer, intent(in) :: start_scomp, start_bccomp, num_comp
    type(bc_tower), intent(in) :: the_bc_tower
    call chk(vdn_multifab_physbc(s%h, int(start_scomp - 1, c_int), int(start_bccomp - 1, c_int), int(num_comp, c_int), &
             the_bc_tower%h), 'multifab_physbc')
  end subroutine multifab_physbc

  ! ---- the hot path: same argument list as reference src/advance_timestep.f90:26-44 ---------------------
  subroutine advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, &
                              the_bc_tower, dt, time, dx, press_comp, proj_type)
    integer        , intent(in   ) :: istep
    type(ml_layout), intent(in   ) :: mla
    type(multifab) , intent(inout) :: sold(:), uold(:), snew(:), unew(:), gp(:), p(:)
    type(multifab) , intent(inout) :: ext_vel_force(:), ext_scal_force(:)
    real(dp_t)     , intent(in   ) :: dt, time, dx(:,:)
    type(bc_tower) , intent(in   ) :: the_bc_tower
    integer        , intent(in   ) :: press_comp, proj_type
    real(c_double) :: dxc(3 * size(dx, 1))
    integer :: n, d
    do n = 1, size(dx, 1)          ! dx(level, dir) -> C [level][dir]
       do d = 1, 3
          dxc((n - 1) * 3 + d) = dx(n, d)
       end do
    end do
    call chk(vdn_advance_timestep(int(istep, c_int), mla%h, handles(sold), handles(uold), handles(snew), handles(unew), &
                                  handles(gp), handles(p), handles(ext_vel_force), handles(ext_scal_force), the_bc_tower%h, &
                                  dt, time, dxc, int(press_comp, c_int), int(proj_type, c_int)), 'advance_timestep')
  end subroutine advance_timestep

  ! estdt(lev, u, s, gp, ext_vel_force, dx, dtold, dt)   (src/estdt.f90:15)
  subroutine estdt(lev, u, s, gp, ext_vel_force, dx, dtold, dt)
    integer        , intent(in ) :: lev
    type(multifab) , intent(in ) :: u, s, gp, ext_vel_force
    real(dp_t)     , intent(in ) :: dx(:), dtold
    real(dp_t)     , intent(out) :: dt
    real(c_double) :: dxc(3)
    dxc = dx(1:3)
    call chk(vdn_estdt(int(lev, c_int), u%h, s%h, gp%h, ext_vel_force%h, dxc, dtold, dt), 'estdt')
  end subroutine estdt

  ! hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, the_bc_tower, press_comp)  (src/hgproject.f90:17)
  subroutine hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, the_bc_tower, press_comp)
    integer        , intent(in   ) :: proj_type, press_comp
    type(ml_layout), intent(in   ) :: mla
    type(multifab) , intent(inout) :: unew(:), rhohalf(:), p(:), gp(:)
    type(multifab) , intent(in   ) :: uold(:)
    real(dp_t)     , intent(in   ) :: dx(:,:), dt
    type(bc_tower) , intent(in   ) :: the_bc_tower
    real(c_double) :: dxc(3 * size(dx, 1))
    integer :: n, d
    do n = 1, size(dx, 1)
       do d = 1, 3
          dxc((n - 1) * 3 + d) = dx(n, d)
       end do
    end do
    call chk(vdn_hgproject(int(proj_type, c_int), mla%h, handles(unew), handles(uold), handles(rhohalf), handles(p), &
                           handles(gp), dxc, dt, the_bc_tower%h, int(press_comp, c_int)), 'hgproject')
  end subroutine hgproject

  ! macproject(mla, umac, rho, dx, the_bc_tower, bc_comp, mac_rhs)  (src/macproject.f90:20); umac(n,d)
  subroutine macproject(mla, umac, rho, dx, the_bc_tower, bc_comp, mac_rhs)
    type(ml_layout), intent(in   ) :: mla
    type(multifab) , intent(inout) :: umac(:,:)
    type(multifab) , intent(in   ) :: rho(:), mac_rhs(:)
    real(dp_t)     , intent(in   ) :: dx(:,:)
    type(bc_tower) , intent(in   ) :: the_bc_tower
    integer        , intent(in   ) :: bc_comp
    real(c_double) :: dxc(3 * size(dx, 1))
    type(c_ptr) :: um(3 * size(umac, 1))
    integer :: n, d
    do n = 1, size(dx, 1)
       do d = 1, 3
          dxc((n - 1) * 3 + d) = dx(n, d)
          um((n - 1) * 3 + d) = umac(n, d)%h
       end do
    end do
    call chk(vdn_macproject(mla%h, um, handles(rho), handles(mac_rhs), dxc, the_bc_tower%h, int(bc_comp, c_int)), 'macproject')
  end subroutine macproject

  ! FBoxLib names; components 1-based as in the reference, rr is accepted for signature compatibility (ratio 2 only)
  subroutine ml_cc_restriction(crse, fine, rr)
    type(multifab), intent(inout) :: crse
    type(multifab), intent(in   ) :: fine
    integer       , intent(in   ) :: rr(:)
    call chk(vdn_ml_cc_restriction(crse%h, fine%h, 0_c_int, int(crse%nc, c_int)), 'ml_cc_restriction')
  end subroutine ml_cc_restriction
  subroutine ml_edge_restriction(crse, fine, rr, dir)
    type(multifab), intent(inout) :: crse
    type(multifab), intent(in   ) :: fine
    integer       , intent(in   ) :: rr(:), dir
    call chk(vdn_ml_edge_restriction(crse%h, fine%h, int(dir - 1, c_int)), 'ml_edge_restriction')
  end subroutine ml_edge_restriction
  subroutine multifab_fill_ghost_cells(fine, crse, icomp, nc)
    type(multifab), intent(inout) :: fine
    type(multifab), intent(in   ) :: crse
    integer       , intent(in   ) :: icomp, nc
    call chk(vdn_multifab_fill_ghost_cells(fine%h, crse%h, int(icomp - 1, c_int), int(nc, c_int)), 'multifab_fill_ghost_cells')
  end subroutine multifab_fill_ghost_cells
  subroutine create_umac_grown(fine, crse, dir)
    type(multifab), intent(inout) :: fine
    type(multifab), intent(in   ) :: crse
    integer       , intent(in   ) :: dir
    call chk(vdn_create_umac_grown(fine%h, crse%h, int(dir - 1, c_int)), 'create_umac_grown')
  end subroutine create_umac_grown
  subroutine ml_restrict_and_fill(nlevs, mf, the_bc_tower, icomp, bcomp, nc, same_boundary)
    integer       , intent(in   ) :: nlevs, icomp, bcomp, nc
    type(multifab), intent(inout) :: mf(:)
    type(bc_tower), intent(in   ) :: the_bc_tower
    logical       , intent(in   ) :: same_boundary
    call chk(vdn_ml_restrict_and_fill(int(nlevs, c_int), handles(mf), int(icomp - 1, c_int), int(bcomp - 1, c_int), int(nc, c_int), &
                                      merge(1_c_int, 0_c_int, same_boundary), the_bc_tower%h), 'ml_restrict_and_fill')
  end subroutine ml_restrict_and_fill

  ! fillpatch(fine, crse, 0, ...) / ml_nodal_prolongation / multifab_copy_c across box lists: src/regrid.f90:311-337
  subroutine fillpatch(fine, crse, icomp, nc)
    type(multifab), intent(inout) :: fine
    type(multifab), intent(in   ) :: crse
    integer       , intent(in   ) :: icomp, nc
    call chk(vdn_fillpatch(fine%h, crse%h, int(icomp - 1, c_int), int(nc, c_int)), 'fillpatch')
  end subroutine fillpatch
  ! vort_module (src/makevort.f90:16-91): the derived quantities of write_plotfile; bc is the tower, the level is u's
  subroutine make_vorticity(vort, comp, u, dx, the_bc_tower)
    type(multifab), intent(inout) :: vort, u
    integer       , intent(in   ) :: comp
    real(dp_t)    , intent(in   ) :: dx(:)
    type(bc_tower), intent(in   ) :: the_bc_tower
    real(c_double) :: d(3)
    d = 1.0_c_double; d(1:size(dx)) = dx
    call chk(vdn_make_vorticity(vort%h, int(comp - 1, c_int), u%h, d, the_bc_tower%h), 'make_vorticity')
  end subroutine make_vorticity
  subroutine make_magvel(magvel, comp, u)
    type(multifab), intent(inout) :: magvel, u
    integer       , intent(in   ) :: comp
    call chk(vdn_make_magvel(magvel%h, int(comp - 1, c_int), u%h), 'make_magvel')
  end subroutine make_magvel
  subroutine ml_nodal_prolongation(fine, crse)
    type(multifab), intent(inout) :: fine, crse
    call chk(vdn_ml_nodal_prolongation(fine%h, crse%h), 'ml_nodal_prolongation')
  end subroutine ml_nodal_prolongation
  subroutine multifab_copy_layouts(dst, dcomp, src, scomp, nc)
    type(multifab), intent(inout) :: dst
    type(multifab), intent(in   ) :: src
    integer       , intent(in   ) :: dcomp, scomp, nc
    call chk(vdn_multifab_copy_layouts(dst%h, int(dcomp - 1, c_int), src%h, int(scomp - 1, c_int), int(nc, c_int)), 'multifab_copy_c')
  end subroutine multifab_copy_layouts
  ! tag_boxes + make_new_grids (src/initialize.f90:247-248, src/regrid.f90:148-149): the boxes of level lev+1 from the state of level lev
  subroutine make_new_grids(new_grid, mf, lev, buf_wid, nest, max_grid_size, boxes, nboxes)
    logical       , intent(  out) :: new_grid
    type(multifab), intent(in   ) :: mf
    integer       , intent(in   ) :: lev, buf_wid, nest, max_grid_size
    type(vdn_box) , intent(  out) :: boxes(:)
    integer       , intent(  out) :: nboxes
    integer(c_int)  :: nb
    integer(c_long) :: nt
    call chk(vdn_make_new_grids(mf%h, int(lev, c_int), int(buf_wid, c_int), int(nest, c_int), 0.9_c_double, 4_c_int, 4_c_int, &
                                int(max_grid_size, c_int), int(size(boxes), c_int), boxes, nb, nt), 'make_new_grids')
    nboxes = nb
    new_grid = nb > 0
  end subroutine make_new_grids

  function handles(mfs) result(h)
    type(multifab), intent(in) :: mfs(:)
    type(c_ptr) :: h(size(mfs))
    integer :: n
    do n = 1, size(mfs)
       h(n) = mfs(n)%h
    end do
  end function handles

end module varden_amd
