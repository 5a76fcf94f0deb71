! varden_boxlib.f90 -- the reference's OWN module names, derived types and call syntax over the MI355X-native hot path.
!
! VARDEN's sources `use` FBoxLib modules (multifab_module, ml_layout_module, layout_module, ml_boxarray_module, define_bc_module, bc_module, ...)
! and its own hot-path modules (advance_module, estdt_module, hgproject_module, macproject_module, multifab_physbc_module, ml_restrict_fill_module,
! proj_parameters).  This file provides modules UNDER THOSE NAMES whose types carry the components the reference's code reads
! (mla%la(n), mla%mba%rr(n-1,:), mla%mba%pd(n), mla%nlevel, mla%dim, mf%ng, mf%nc, the_bc_tower%bc_tower_array(n), the_bc_tower%domain_bc) and whose
! procedures take the argument lists the reference writes, e.g.
!     call multifab_build(unew(n), mla%la(n), dm, ng_cell)                      src/varden.f90:421
!     call multifab_build(p(n), mla%la(n), 1, ng_grow, nodal)                   src/initialize.f90:127
!     call multifab_build_edge(umac(n,i), mla%la(n), 1, 1, i)                   src/advance_timestep.f90:78
!     call multifab_physbc(uold(n), 1, 1, dm, the_bc_tower%bc_tower_array(n))   src/varden.f90:171, src/multifab_physbc.f90:17-22
!     call multifab_fill_ghost_cells(uold(n), uold(n-1), ng_cell, mla%mba%rr(n-1,:), bc(n-1), bc(n), 1, 1, dm)     src/varden.f90:275-279
!     call ml_restrict_and_fill(nlevs, snew, mla%mba%rr, the_bc_level, bcomp=dm+1)                                 src/update.f90:104-108
!     call macproject(mla, umac, sold, mac_rhs, dx, the_bc_tower, press_comp)   src/advance_timestep.f90:104, src/macproject.f90:20
!     call advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, the_bc_tower, dt, time, dx, press_comp, regular_timestep)
! so that a file of the reference that drives the hot path compiles against these modules with its `use` lines and calls UNCHANGED
! (tests/fortran/varden_loop.f90 is such a file: the time-loop body of src/varden.f90 as the reference spells it).  Every routine forwards to module varden_amd
! (varden_amd_mod.f90), i.e. to the C-ABI of include/varden_amd.h; all multifab data lives in HBM.
!
! What is NOT here: FBoxLib's host-side data access (dataptr returns a device address, see varden_amd_mod.f90), parallel I/O, fabio, the box calculus
! beyond what the hot path's callers use.  dm = 3.

module bl_types
  use iso_c_binding, only: c_double
  implicit none
  integer, parameter :: dp_t = c_double
end module bl_types

module bl_constants_module
  use bl_types
  implicit none
  real(dp_t), parameter :: ZERO = 0.0_dp_t, ONE = 1.0_dp_t, TWO = 2.0_dp_t, HALF = 0.5_dp_t, FOURTH = 0.25_dp_t, EIGHTH = 0.125_dp_t
end module bl_constants_module

module bc_module
  implicit none
  integer, parameter :: BC_PER = -1, BC_INT = 0, BC_DIR = 1, BC_NEU = 2
  integer, parameter :: PERIODIC = -1, INTERIOR = 0, INLET = 11, OUTLET = 12, SYMMETRY = 13, SLIP_WALL = 14, NO_SLIP_WALL = 15
  integer, parameter :: REFLECT_ODD = 20, REFLECT_EVEN = 21, FOEXTRAP = 22, EXT_DIR = 23, HOEXTRAP = 24
end module bc_module

module proj_parameters
  implicit none
  integer, parameter :: initial_projection = 1, divu_iters = 2, pressure_iters = 3, regular_timestep = 4
end module proj_parameters

! ---- FBoxLib's process-level utilities that its container modules re-export (parallel_IOProcessor, bl_prof_timer, bl_error reach the reference's files through
! multifab_module and friends without a `use` line of their own)
module parallel
  use iso_c_binding
  use bl_types
  implicit none
  integer, parameter :: MPI_MAX = 1, MPI_MIN = 2, MPI_SUM = 3
  integer, save :: par_rank = 0, par_nranks = 1
  interface
     integer(c_int) function vdn_comm_allreduce_max_c(host, n) bind(C, name="vdn_comm_allreduce_max")
       import :: c_int, c_double
       real(c_double), intent(inout) :: host(*)
       integer(c_int), value :: n
     end function vdn_comm_allreduce_max_c
  end interface
  interface parallel_reduce
     module procedure parallel_reduce_d, parallel_reduce_dv, parallel_reduce_i
  end interface
contains
  subroutine parallel_set_ranks(rank, nranks)
    integer, intent(in) :: rank, nranks
    par_rank = rank; par_nranks = nranks
  end subroutine parallel_set_ranks
  integer function parallel_myproc()
    parallel_myproc = par_rank
  end function parallel_myproc
  integer function parallel_nprocs()
    parallel_nprocs = par_nranks
  end function parallel_nprocs
  logical function parallel_IOProcessor()
    parallel_IOProcessor = par_rank == 0
  end function parallel_IOProcessor
  integer function parallel_IOProcessorNode()
    parallel_IOProcessorNode = 0
  end function parallel_IOProcessorNode
  real(dp_t) function parallel_wtime()
    integer(8) :: c, r
    call system_clock(c, r)
    parallel_wtime = real(c, dp_t) / real(r, dp_t)
  end function parallel_wtime
  subroutine parallel_barrier()
    real(dp_t) :: z(1)
    z = 0.0_dp_t
    if (par_nranks > 1) then
       if (vdn_comm_allreduce_max_c(z, 1_c_int) /= 0) error stop 'parallel_barrier: the all-reduce failed'
    end if
  end subroutine parallel_barrier
  ! parallel_reduce(r, a, op, proc): MAX and MIN over the ranks through ncclAllReduce(MAX) (every rank receives the result, as FBoxLib's allreduce form does)
  subroutine parallel_reduce_dv(r, a, op, proc)
    real(dp_t), intent(out) :: r(:)
    real(dp_t), intent(in) :: a(:)
    integer, intent(in) :: op
    integer, intent(in), optional :: proc
    r = a
    if (par_nranks == 1) return
    select case (op)
    case (MPI_MAX)
       if (vdn_comm_allreduce_max_c(r, int(size(r), c_int)) /= 0) error stop 'parallel_reduce: the all-reduce failed'
    case (MPI_MIN)
       r = -r
       if (vdn_comm_allreduce_max_c(r, int(size(r), c_int)) /= 0) error stop 'parallel_reduce: the all-reduce failed'
       r = -r
    case default
       error stop 'parallel_reduce: MPI_SUM across ranks is not an operation of the hot path (its global scalars are maxima)'
    end select
  end subroutine parallel_reduce_dv
  subroutine parallel_reduce_d(r, a, op, proc)
    real(dp_t), intent(out) :: r
    real(dp_t), intent(in) :: a
    integer, intent(in) :: op
    integer, intent(in), optional :: proc
    real(dp_t) :: rv(1)
    call parallel_reduce_dv(rv, (/ a /), op)
    r = rv(1)
  end subroutine parallel_reduce_d
  subroutine parallel_reduce_i(r, a, op, proc)
    integer, intent(out) :: r
    integer, intent(in) :: a, op
    integer, intent(in), optional :: proc
    real(dp_t) :: rv(1)
    call parallel_reduce_dv(rv, (/ real(a, dp_t) /), op)
    r = nint(rv(1))
  end subroutine parallel_reduce_i
end module parallel

module bl_error_module
  implicit none
contains
  subroutine bl_error(str)
    character(len=*), intent(in) :: str
    write(*, '(a,a)') 'BOXLIB ERROR: ', str
    error stop 1
  end subroutine bl_error
  subroutine bl_warn(str)
    character(len=*), intent(in) :: str
    write(*, '(a,a)') 'BOXLIB WARN: ', str
  end subroutine bl_warn
  subroutine bl_assert(cond, str)
    logical, intent(in) :: cond
    character(len=*), intent(in) :: str
    if (.not. cond) call bl_error(str)
  end subroutine bl_assert
end module bl_error_module

module bl_IO_module
  implicit none
contains
  integer function unit_new()
    logical :: used
    do unit_new = 20, 999
       inquire(unit=unit_new, opened=used)
       if (.not. used) return
    end do
    error stop 'unit_new: no free unit'
  end function unit_new
end module bl_IO_module

module bl_prof_module
  implicit none
  type bl_prof_timer
     character(len=64) :: name = ''
  end type bl_prof_timer
  interface build
     module procedure bl_prof_timer_build
  end interface
  interface destroy
     module procedure bl_prof_timer_destroy
  end interface
contains
  subroutine bl_prof_initialize(on)                         ! src/main.f90:17,27,29: profiling is rocprofv3's job here (roctx ranges under the reference's timer names)
    logical, intent(in), optional :: on
  end subroutine bl_prof_initialize
  subroutine bl_prof_glean(fname)
    character(len=*), intent(in) :: fname
  end subroutine bl_prof_glean
  subroutine bl_prof_finalize()
  end subroutine bl_prof_finalize
  subroutine bl_prof_timer_build(bpt, name)
    type(bl_prof_timer), intent(inout) :: bpt
    character(len=*), intent(in) :: name
    bpt%name = name
  end subroutine bl_prof_timer_build
  subroutine bl_prof_timer_destroy(bpt)
    type(bl_prof_timer), intent(inout) :: bpt
    bpt%name = ''
  end subroutine bl_prof_timer_destroy
end module bl_prof_module

! bl_mem_stat_module (FBoxLib): the allocation counters the reference's main program prints at its end (src/main.f90:35-47).  Here: how many objects of a kind the host built and
! destroyed through this surface (device memory is the library's: vdn_arena_stats); the associations of FBoxLib's layouts (boxassoc, fgassoc, ...) have no counterpart -- the
! exchange plans live inside the library, per layout -- and report zero.
module bl_mem_stat_module
  implicit none
  type mem_stats
     integer(kind=8) :: num_alloc = 0, num_dealloc = 0
     integer(kind=8) :: cnt_alloc = 0, cnt_dealloc = 0
  end type mem_stats
  interface print
     module procedure mem_stats_print
  end interface
contains
  subroutine mem_stats_print(ms, str, unit, advance, total)
    type(mem_stats), intent(in) :: ms
    character(len=*), intent(in), optional :: str, advance
    integer, intent(in), optional :: unit
    logical, intent(in), optional :: total
    integer :: un
    un = 6; if (present(unit)) un = unit
    if (present(str)) then
       write(un, '(a,": built ",i0,", destroyed ",i0)') str, ms%num_alloc, ms%num_dealloc
    else
       write(un, '("built ",i0,", destroyed ",i0)') ms%num_alloc, ms%num_dealloc
    end if
  end subroutine mem_stats_print
  subroutine mem_stats_alloc(ms, n)
    type(mem_stats), intent(inout) :: ms
    integer, intent(in), optional :: n
    ms%num_alloc = ms%num_alloc + 1
    if (present(n)) ms%cnt_alloc = ms%cnt_alloc + n
  end subroutine mem_stats_alloc
  subroutine mem_stats_dealloc(ms, n)
    type(mem_stats), intent(inout) :: ms
    integer, intent(in), optional :: n
    ms%num_dealloc = ms%num_dealloc + 1
    if (present(n)) ms%cnt_dealloc = ms%cnt_dealloc + n
  end subroutine mem_stats_dealloc
end module bl_mem_stat_module

module box_module
  implicit none
  type box
     integer :: dim = 3
     integer :: lo(3) = 0, hi(3) = -1
  end type box
  interface refine
     module procedure refine_i, refine_v
  end interface
  interface coarsen
     module procedure coarsen_i, coarsen_v
  end interface
contains
  function make_box(lo, hi) result(bx)
    integer, intent(in) :: lo(:), hi(:)
    type(box) :: bx
    bx%dim = size(lo); bx%lo = 0; bx%hi = 0
    bx%lo(1:size(lo)) = lo; bx%hi(1:size(hi)) = hi
  end function make_box
  function lwb(bx) result(lo)
    type(box), intent(in) :: bx
    integer :: lo(bx%dim)
    lo = bx%lo(1:bx%dim)
  end function lwb
  function upb(bx) result(hi)
    type(box), intent(in) :: bx
    integer :: hi(bx%dim)
    hi = bx%hi(1:bx%dim)
  end function upb
  ! refine(bx, rr) / coarsen(bx, rr), rr a scalar or one ratio per direction   (FBoxLib; src/initialize.f90:205, src/regrid.f90:114, src/varden.f90:554): cell-centred boxes
  function refine_v(bx, rr) result(r)
    type(box), intent(in) :: bx
    integer, intent(in) :: rr(:)
    type(box) :: r
    r = bx
    r%lo(1:bx%dim) = bx%lo(1:bx%dim) * rr(1:bx%dim)
    r%hi(1:bx%dim) = (bx%hi(1:bx%dim) + 1) * rr(1:bx%dim) - 1
  end function refine_v
  function refine_i(bx, rr) result(r)
    type(box), intent(in) :: bx
    integer, intent(in) :: rr
    type(box) :: r
    integer :: v(3)
    v = rr
    r = refine_v(bx, v)
  end function refine_i
  function coarsen_v(bx, rr) result(r)
    type(box), intent(in) :: bx
    integer, intent(in) :: rr(:)
    type(box) :: r
    integer :: d
    r = bx
    do d = 1, bx%dim
       r%lo(d) = floor(real(bx%lo(d)) / real(rr(d)))
       r%hi(d) = floor(real(bx%hi(d)) / real(rr(d)))
    end do
  end function coarsen_v
  function coarsen_i(bx, rr) result(r)
    type(box), intent(in) :: bx
    integer, intent(in) :: rr
    type(box) :: r
    integer :: v(3)
    v = rr
    r = coarsen_v(bx, v)
  end function coarsen_i
  pure logical function box_equal(a, b)
    type(box), intent(in) :: a, b
    box_equal = a%dim == b%dim .and. all(a%lo(1:a%dim) == b%lo(1:a%dim)) .and. all(a%hi(1:a%dim) == b%hi(1:a%dim))
  end function box_equal
end module box_module

module boxarray_module
  use box_module
  use bl_mem_stat_module
  implicit none
  type(mem_stats), save :: boxarray_ms
  type boxarray
     integer :: dim = 3, nboxes = 0
     type(box), pointer :: bxs(:) => null()
  end type boxarray
  interface destroy
     module procedure boxarray_destroy
  end interface
  interface nboxes
     module procedure boxarray_nboxes
  end interface
  interface get_box
     module procedure boxarray_get_box
  end interface
contains
  subroutine boxarray_build_bx(ba, bx)
    type(boxarray), intent(inout) :: ba
    type(box), intent(in) :: bx
    allocate(ba%bxs(1)); ba%bxs(1) = bx; ba%nboxes = 1; ba%dim = bx%dim
  end subroutine boxarray_build_bx
  subroutine boxarray_build_v(ba, bxs)
    type(boxarray), intent(inout) :: ba
    type(box), intent(in) :: bxs(:)
    allocate(ba%bxs(size(bxs))); ba%bxs = bxs; ba%nboxes = size(bxs); ba%dim = bxs(1)%dim
  end subroutine boxarray_build_v
  subroutine boxarray_destroy(ba)
    type(boxarray), intent(inout) :: ba
    if (associated(ba%bxs)) deallocate(ba%bxs)
    ba%bxs => null(); ba%nboxes = 0
  end subroutine boxarray_destroy
  integer function boxarray_nboxes(ba)                      ! nboxes(mla%mba%bas(n))   (src/varden.f90:640)
    type(boxarray), intent(in) :: ba
    boxarray_nboxes = ba%nboxes
  end function boxarray_nboxes
  function boxarray_get_box(ba, i) result(bx)               ! get_box(mla%mba%bas(n), i)   (src/varden.f90:649)
    type(boxarray), intent(in) :: ba
    integer, intent(in) :: i
    type(box) :: bx
    bx = ba%bxs(i)
  end function boxarray_get_box
  logical function boxarray_same_q(a, b)                    ! src/regrid.f90:302
    type(boxarray), intent(in) :: a, b
    integer :: i
    boxarray_same_q = a%nboxes == b%nboxes
    if (.not. boxarray_same_q) return
    do i = 1, a%nboxes
       if (.not. box_equal(a%bxs(i), b%bxs(i))) boxarray_same_q = .false.
    end do
  end function boxarray_same_q
  function boxarray_mem_stats() result(r)                     ! src/main.f90:41
    type(mem_stats) :: r
    r = boxarray_ms
  end function boxarray_mem_stats
end module boxarray_module

module ml_boxarray_module
  use boxarray_module
  implicit none
  type ml_boxarray
     integer :: dim = 3, nlevel = 0
     integer, pointer :: rr(:,:) => null()          ! rr(n, dir): refinement ratio between levels n and n+1
     type(boxarray), pointer :: bas(:) => null()
     type(box), pointer :: pd(:) => null()
  end type ml_boxarray
  interface destroy
     module procedure ml_boxarray_destroy
  end interface
contains
  subroutine ml_boxarray_build_n(mba, nlevel, dim)
    type(ml_boxarray), intent(out) :: mba
    integer, intent(in) :: nlevel, dim
    mba%dim = dim; mba%nlevel = nlevel
    allocate(mba%rr(max(nlevel - 1, 1), dim), mba%bas(nlevel), mba%pd(nlevel))
    mba%rr = 2
  end subroutine ml_boxarray_build_n
  subroutine ml_boxarray_destroy(mba)
    type(ml_boxarray), intent(inout) :: mba
    integer :: n
    if (associated(mba%bas)) then
       do n = 1, size(mba%bas)
          call boxarray_destroy(mba%bas(n))
       end do
       deallocate(mba%bas)
    end if
    if (associated(mba%rr)) deallocate(mba%rr)
    if (associated(mba%pd)) deallocate(mba%pd)
    mba%nlevel = 0
  end subroutine ml_boxarray_destroy
end module ml_boxarray_module

module layout_module
  use iso_c_binding, only: c_ptr, c_null_ptr, c_associated
  use bl_mem_stat_module
  implicit none
  type(mem_stats), save :: layout_ms
  ! the layout of ONE level: the hierarchy's handle and the level number (the C-ABI keeps the box lists of all levels in one vdn_layout)
  type layout
     type(c_ptr) :: h = c_null_ptr
     integer :: lev = 0, dim = 3, nlevel = 0
  end type layout
  interface destroy
     module procedure layout_destroy
  end interface
  interface operator(.eq.)
     module procedure layout_eq
  end interface
  interface operator(.ne.)
     module procedure layout_ne
  end interface
contains
  ! a level's layout is a view of the hierarchy's handle: the hierarchy is released by destroy(mla) (src/regrid.f90:214,242 destroy single layouts)
  subroutine layout_destroy(la)
    type(layout), intent(inout) :: la
    la%h = c_null_ptr; la%lev = 0
  end subroutine layout_destroy
  pure logical function layout_eq(a, b)
    type(layout), intent(in) :: a, b
    layout_eq = c_associated(a%h, b%h) .and. a%lev == b%lev
  end function layout_eq
  pure logical function layout_ne(a, b)                     ! src/regrid.f90:223
    type(layout), intent(in) :: a, b
    layout_ne = .not. layout_eq(a, b)
  end function layout_ne
  subroutine layout_flush_copyassoc_cache()                 ! src/main.f90:23: the library keeps its exchange plans per layout and drops them with it
  end subroutine layout_flush_copyassoc_cache
  ! src/main.f90:42-47: layouts built through ml_layout_build count here; FBoxLib's association caches do not exist on this side (the library's plans are per layout, inside it)
  function layout_mem_stats() result(r)
    type(mem_stats) :: r
    r = layout_ms
  end function layout_mem_stats
  function boxassoc_mem_stats() result(r)
    type(mem_stats) :: r
  end function boxassoc_mem_stats
  function fgassoc_mem_stats() result(r)
    type(mem_stats) :: r
  end function fgassoc_mem_stats
  function syncassoc_mem_stats() result(r)
    type(mem_stats) :: r
  end function syncassoc_mem_stats
  function copyassoc_mem_stats() result(r)
    type(mem_stats) :: r
  end function copyassoc_mem_stats
  function fluxassoc_mem_stats() result(r)
    type(mem_stats) :: r
  end function fluxassoc_mem_stats
end module layout_module

module ml_layout_module
  use iso_c_binding
  use layout_module
  use ml_boxarray_module
  use varden_amd, only: vamd_ml_layout => ml_layout, vamd_layout_build => ml_layout_build, vamd_layout_destroy => ml_layout_destroy, vdn_box
  implicit none
  type ml_layout
     integer :: dim = 3, nlevel = 0
     type(ml_boxarray) :: mba
     type(layout), pointer :: la(:) => null()
     logical, pointer :: pmask(:) => null()
     type(vamd_ml_layout) :: v
  end type ml_layout
  interface destroy
     module procedure ml_layout_destroy
  end interface
contains
  ! ml_layout_build(mla, mba, pmask)   (src/initialize.f90:113): boxes are dealt to one rank here; mla keeps its own copy of mba
  subroutine ml_layout_build(mla, mba, pmask)
    type(ml_layout), intent(out) :: mla
    type(ml_boxarray), intent(in) :: mba
    logical, intent(in), optional :: pmask(:)
    type(vdn_box), allocatable :: pd(:), boxes(:)
    integer, allocatable :: nboxes(:), owner(:)
    integer :: n, i, k, tot, rr(3)
    logical :: pm(3)
    mla%dim = mba%dim; mla%nlevel = mba%nlevel
    call ml_boxarray_build_n(mla%mba, mba%nlevel, mba%dim)
    mla%mba%rr = mba%rr
    tot = 0
    do n = 1, mba%nlevel
       mla%mba%pd(n) = mba%pd(n)
       call boxarray_build_v(mla%mba%bas(n), mba%bas(n)%bxs)
       tot = tot + mba%bas(n)%nboxes
    end do
    allocate(pd(mba%nlevel), nboxes(mba%nlevel), boxes(tot), owner(tot))
    owner = 0; k = 0
    do n = 1, mba%nlevel
       pd(n)%lo = mba%pd(n)%lo; pd(n)%hi = mba%pd(n)%hi
       nboxes(n) = mba%bas(n)%nboxes
       do i = 1, nboxes(n)
          k = k + 1
          boxes(k)%lo = mba%bas(n)%bxs(i)%lo; boxes(k)%hi = mba%bas(n)%bxs(i)%hi
       end do
    end do
    pm = .false.; if (present(pmask)) pm(1:size(pmask)) = pmask
    allocate(mla%pmask(mba%dim)); mla%pmask = pm(1:mba%dim)
    rr = 2; rr(1:mba%dim) = mba%rr(1, :)
    call vamd_layout_build(mla%v, mba%nlevel, rr, pd, nboxes, boxes, owner, pm)
    allocate(mla%la(mba%nlevel))
    do n = 1, mba%nlevel
       mla%la(n)%h = mla%v%h; mla%la(n)%lev = n; mla%la(n)%dim = mba%dim; mla%la(n)%nlevel = mba%nlevel
       call mem_stats_alloc(layout_ms)
    end do
  end subroutine ml_layout_build
  function ml_layout_get_pd(mla, n) result(bx)               ! src/varden.f90:641
    type(ml_layout), intent(in) :: mla
    integer, intent(in) :: n
    type(box) :: bx
    bx = mla%mba%pd(n)
  end function ml_layout_get_pd
  subroutine ml_layout_destroy(mla, keep_coarse_layout)
    type(ml_layout), intent(inout) :: mla
    logical, intent(in), optional :: keep_coarse_layout        ! (src/regrid.f90:214; one handle holds every level here: it goes as a whole)
    call vamd_layout_destroy(mla%v)
    call ml_boxarray_destroy(mla%mba)
    if (associated(mla%la)) deallocate(mla%la)
    if (associated(mla%pmask)) deallocate(mla%pmask)
    mla%nlevel = 0
  end subroutine ml_layout_destroy
end module ml_layout_module

module multifab_module
  use iso_c_binding
  use bl_types
  use parallel
  use bl_error_module
  use bl_prof_module
  use box_module
  use boxarray_module
  use layout_module
  use varden_amd, only: vamd_multifab => multifab, vamd_ml_layout => ml_layout, vamd_build => multifab_build, vamd_build_edge => multifab_build_edge, &
                        vamd_build_nodal => multifab_build_nodal, vamd_destroy => multifab_destroy, vamd_nfabs => nfabs, vamd_get_box => get_box, &
                        vamd_dataptr => dataptr, vamd_setval => setval, vamd_copy_c => multifab_copy_c, vamd_norm_inf => norm_inf, vamd_norm_inf_c => norm_inf_c, &
                        vamd_fill_boundary => multifab_fill_boundary, multifab_copy_to_host_v => multifab_copy_to_host, &
                        multifab_copy_from_host_v => multifab_copy_from_host, vamd_fab_size => multifab_fab_size, vdn_box
  implicit none
  type(mem_stats), save :: multifab_ms, fab_ms
  type multifab
     type(vamd_multifab) :: v
     integer :: dim = 3, nc = 1, ng = 0
     logical :: nodal(3) = .false.
     type(layout) :: la                                      ! the level it was built on (get_layout)
  end type multifab
  interface build
     module procedure multifab_build
  end interface
  interface destroy
     module procedure multifab_destroy
  end interface
  interface setval
     module procedure multifab_setval, multifab_setval_c
  end interface
  interface norm_inf
     module procedure multifab_norm_inf, multifab_norm_inf_c
  end interface
  interface get_box
     module procedure multifab_get_box
  end interface
contains
  function as_vamd_layout(la) result(v)
    type(layout), intent(in) :: la
    type(vamd_ml_layout) :: v
    v%h = la%h; v%nlevel = la%nlevel; v%dim = la%dim
  end function as_vamd_layout
  ! multifab_build(mf, la, nc, ng, nodal)   (FBoxLib; src/varden.f90:129, 421-424, src/initialize.f90:124-127)
  subroutine multifab_build(mf, la, nc, ng, nodal)
    type(multifab), intent(out) :: mf
    type(layout), intent(in) :: la
    integer, intent(in), optional :: nc, ng
    logical, intent(in), optional :: nodal(:)
    integer :: c, g
    c = 1; if (present(nc)) c = nc
    g = 0; if (present(ng)) g = ng
    mf%nodal = .false.; if (present(nodal)) mf%nodal(1:size(nodal)) = nodal
    if (all(mf%nodal)) then
       call vamd_build_nodal(mf%v, as_vamd_layout(la), la%lev, c, g)
    else if (.not. any(mf%nodal)) then
       call vamd_build(mf%v, as_vamd_layout(la), la%lev, c, g)
    else if (count(mf%nodal) == 1) then
       call vamd_build_edge(mf%v, as_vamd_layout(la), la%lev, c, g, findloc(mf%nodal, .true., 1))
    else
       error stop 'multifab_build: nodal in two directions is not a layout of the hot path'
    end if
    mf%nc = c; mf%ng = g; mf%dim = la%dim; mf%la = la
    call mem_stats_alloc(multifab_ms); call mem_stats_alloc(fab_ms, vamd_nfabs(mf%v))
  end subroutine multifab_build
  ! multifab_build_edge(mf, la, nc, ng, dir)   (src/advance_timestep.f90:78)
  subroutine multifab_build_edge(mf, la, nc, ng, dir)
    type(multifab), intent(out) :: mf
    type(layout), intent(in) :: la
    integer, intent(in) :: nc, ng, dir
    call vamd_build_edge(mf%v, as_vamd_layout(la), la%lev, nc, ng, dir)
    mf%nodal = .false.; mf%nodal(dir) = .true.
    mf%nc = nc; mf%ng = ng; mf%dim = la%dim; mf%la = la
    call mem_stats_alloc(multifab_ms); call mem_stats_alloc(fab_ms, vamd_nfabs(mf%v))
  end subroutine multifab_build_edge
  subroutine multifab_destroy(mf)
    type(multifab), intent(inout) :: mf
    call mem_stats_dealloc(multifab_ms); call mem_stats_dealloc(fab_ms, vamd_nfabs(mf%v))
    call vamd_destroy(mf%v)
  end subroutine multifab_destroy
  function multifab_mem_stats() result(r)                     ! src/main.f90:39-40
    type(mem_stats) :: r
    r = multifab_ms
  end function multifab_mem_stats
  function fab_mem_stats() result(r)
    type(mem_stats) :: r
    r = fab_ms
  end function fab_mem_stats
  integer function nfabs(mf)
    type(multifab), intent(in) :: mf
    nfabs = vamd_nfabs(mf%v)
  end function nfabs
  integer function multifab_ncomp(mf)
    type(multifab), intent(in) :: mf
    multifab_ncomp = mf%nc
  end function multifab_ncomp
  integer function nghost(mf)
    type(multifab), intent(in) :: mf
    nghost = mf%ng
  end function nghost
  pure integer function get_dim(mf)
    type(multifab), intent(in) :: mf
    get_dim = mf%dim
  end function get_dim
  function get_layout(mf) result(la)                        ! src/initialize.f90:71
    type(multifab), intent(in) :: mf
    type(layout) :: la
    la = mf%la
  end function get_layout
  function multifab_get_box(mf, i) result(bx)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    type(box) :: bx
    type(vdn_box) :: b
    b = vamd_get_box(mf%v, i)
    bx%dim = mf%dim; bx%lo = b%lo; bx%hi = b%hi
  end function multifab_get_box
  ! the DEVICE address of fab i (the reference's dataptr returns a host pointer: host access goes through multifab_copy_to_host / _from_host)
  function dataptr(mf, i) result(dev)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    type(c_ptr) :: dev
    dev = vamd_dataptr(mf%v, i)
  end function dataptr
  integer(c_long) function multifab_fab_size(mf, i)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    multifab_fab_size = vamd_fab_size(mf%v, i)
  end function multifab_fab_size
  subroutine multifab_copy_to_host(mf, i, host)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: i
    real(dp_t), intent(out) :: host(*)
    call multifab_copy_to_host_v(mf%v, i, host)
  end subroutine multifab_copy_to_host
  subroutine multifab_copy_from_host(mf, i, host)
    type(multifab), intent(inout) :: mf
    integer, intent(in) :: i
    real(dp_t), intent(in) :: host(*)
    call multifab_copy_from_host_v(mf%v, i, host)
  end subroutine multifab_copy_from_host
  ! setval(mf, val, all=)  and  setval(mf, val, comp, nc, all=)   (src/varden.f90:130, 426-430)
  subroutine multifab_setval(mf, val, all)
    type(multifab), intent(inout) :: mf
    real(dp_t), intent(in) :: val
    logical, intent(in), optional :: all
    logical :: a
    a = .false.; if (present(all)) a = all
    call vamd_setval(mf%v, val, 1, mf%nc, a)
  end subroutine multifab_setval
  subroutine multifab_setval_c(mf, val, comp, nc, all)
    type(multifab), intent(inout) :: mf
    real(dp_t), intent(in) :: val
    integer, intent(in) :: comp
    integer, intent(in), optional :: nc
    logical, intent(in), optional :: all
    logical :: a
    integer :: n
    a = .false.; if (present(all)) a = all
    n = 1; if (present(nc)) n = nc
    call vamd_setval(mf%v, val, comp, n, a)
  end subroutine multifab_setval_c
  ! multifab_copy_c(dst, dcomp, src, scomp, nc, ng=)   (src/varden.f90:175-176, 323-326)
  subroutine multifab_copy_c(dst, dcomp, src, scomp, nc, ng)
    type(multifab), intent(inout) :: dst
    type(multifab), intent(in) :: src
    integer, intent(in) :: dcomp, scomp
    integer, intent(in), optional :: nc, ng
    integer :: n, g
    n = 1; if (present(nc)) n = nc
    g = 0; if (present(ng)) g = ng
    call vamd_copy_c(dst%v, dcomp, src%v, scomp, n, g)
  end subroutine multifab_copy_c
  subroutine multifab_fill_boundary(mf)
    type(multifab), intent(inout) :: mf
    call vamd_fill_boundary(mf%v)
  end subroutine multifab_fill_boundary
  real(dp_t) function multifab_norm_inf(mf)
    type(multifab), intent(in) :: mf
    multifab_norm_inf = vamd_norm_inf(mf%v)
  end function multifab_norm_inf
  ! norm_inf(mf, comp, nc)   (src/advance_timestep.f90:187-189)
  real(dp_t) function multifab_norm_inf_c(mf, comp, nc, all)
    type(multifab), intent(in) :: mf
    integer, intent(in) :: comp
    integer, intent(in), optional :: nc
    logical, intent(in), optional :: all
    integer :: n
    n = 1; if (present(nc)) n = nc
    multifab_norm_inf_c = vamd_norm_inf_c(mf%v, comp, n)
  end function multifab_norm_inf_c
end module multifab_module

module define_bc_module
  use iso_c_binding
  use bc_module
  use layout_module
  use ml_layout_module
  use varden_amd, only: vamd_bc_tower => bc_tower, vamd_ml_layout => ml_layout, vamd_bc_build => bc_tower_build, vamd_bc_destroy => bc_tower_destroy
  implicit none
  ! the bc tables of one level live in the C-side tower; a bc_level names the tower and the level
  type bc_level
     type(c_ptr) :: tower = c_null_ptr
     integer :: lev = 0
  end type bc_level
  type bc_tower
     integer :: max_level_built = 0
     type(bc_level), pointer :: bc_tower_array(:) => null()
     integer, pointer :: domain_bc(:,:) => null()
     type(vamd_bc_tower) :: v
  end type bc_tower
  interface build
     module procedure bc_tower_init
     module procedure bc_tower_level_build
  end interface
  interface destroy
     module procedure bc_tower_destroy
  end interface
contains
  ! bc_tower_init(bct, num_levs, dm, phys_bc_in)   (src/define_bc_tower.f90:48-62)
  subroutine bc_tower_init(bct, num_levs, dm, phys_bc_in)
    type(bc_tower), intent(out) :: bct
    integer, intent(in) :: num_levs, dm
    integer, intent(in) :: phys_bc_in(:,:)
    allocate(bct%bc_tower_array(num_levs), bct%domain_bc(dm, 2))
    bct%domain_bc = phys_bc_in
  end subroutine bc_tower_init
  ! bc_tower_level_build(bct, n, la)   (src/define_bc_tower.f90:64-135): the C-side tower covers every level of la's hierarchy and is built with the first level
  subroutine bc_tower_level_build(bct, n, la)
    type(bc_tower), intent(inout) :: bct
    integer, intent(in) :: n
    type(layout), intent(in) :: la
    type(vamd_ml_layout) :: v
    integer :: phys(3, 2), m
    if (.not. c_associated(bct%v%h)) then
       phys = INTERIOR
       phys(1:size(bct%domain_bc, 1), :) = bct%domain_bc
       v%h = la%h; v%nlevel = la%nlevel; v%dim = la%dim
       call vamd_bc_build(bct%v, v, phys)
       do m = 1, size(bct%bc_tower_array)
          bct%bc_tower_array(m)%tower = bct%v%h; bct%bc_tower_array(m)%lev = m
       end do
    end if
    bct%max_level_built = max(bct%max_level_built, n)
  end subroutine bc_tower_level_build
  subroutine bc_tower_destroy(bct)
    type(bc_tower), intent(inout) :: bct
    if (c_associated(bct%v%h)) call vamd_bc_destroy(bct%v)
    if (associated(bct%bc_tower_array)) deallocate(bct%bc_tower_array)
    if (associated(bct%domain_bc)) deallocate(bct%domain_bc)
    bct%max_level_built = 0
  end subroutine bc_tower_destroy
  function as_vamd_tower(bl) result(t)
    type(bc_level), intent(in) :: bl
    type(vamd_bc_tower) :: t
    t%h = bl%tower
  end function as_vamd_tower
end module define_bc_module

module multifab_physbc_module
  use multifab_module
  use define_bc_module
  use varden_amd, only: vamd_physbc => multifab_physbc
  implicit none
contains
  ! multifab_physbc(s, start_scomp, start_bccomp, num_comp, the_bc_level)   (src/multifab_physbc.f90:17-22; the optional time / dx / prob_lo / prob_hi
  ! arguments are accepted and not used: the inflow data are the constants of the inputs file)
  subroutine multifab_physbc(s, start_scomp, start_bccomp, num_comp, the_bc_level, time_in, dx_in, prob_lo_in, prob_hi_in)
    type(multifab), intent(inout) :: s
    integer, intent(in) :: start_scomp, start_bccomp, num_comp
    type(bc_level), intent(in) :: the_bc_level
    real(dp_t), intent(in), optional :: time_in, dx_in(:), prob_lo_in(:), prob_hi_in(:)
    call vamd_physbc(s%v, start_scomp, start_bccomp, num_comp, as_vamd_tower(the_bc_level))
  end subroutine multifab_physbc
end module multifab_physbc_module

module multifab_fill_ghost_module
  use multifab_module
  use define_bc_module
  use varden_amd, only: vamd_fill_ghost => multifab_fill_ghost_cells
  implicit none
contains
  ! multifab_fill_ghost_cells(fine, crse, ng, ir, bc_crse, bc_fine, icomp, bcomp, nc)   (FBoxLib; src/varden.f90:275-289): the fine ghost cells that no
  ! fine box covers, from the coarse level.  (The same-level exchange and the physical boundary follow in the caller, varden.f90:291-300.)
  subroutine multifab_fill_ghost_cells(fine, crse, ng, ir, bc_crse, bc_fine, icomp, bcomp, nc)
    type(multifab), intent(inout) :: fine
    type(multifab), intent(inout) :: crse
    integer, intent(in) :: ng, ir(:), icomp, bcomp, nc
    type(bc_level), intent(in) :: bc_crse, bc_fine
    if (any(ir /= 2)) error stop 'multifab_fill_ghost_cells: refinement ratio 2 only'
    call vamd_fill_ghost(fine%v, crse%v, icomp, nc)
  end subroutine multifab_fill_ghost_cells
end module multifab_fill_ghost_module

module ml_restrict_fill_module
  use multifab_module
  use define_bc_module
  use varden_amd, only: vamd_rf => ml_restrict_and_fill
  implicit none
contains
  ! ml_restrict_and_fill(nlevs, mf, rr, bc, icomp=, bcomp=, nc=, ng=, same_boundary=)   (FBoxLib; src/update.f90:104-108, src/mkforce.f90:75-76)
  subroutine ml_restrict_and_fill(nlevs, mf, rr, bc, icomp, bcomp, nc, ng, same_boundary)
    integer, intent(in) :: nlevs
    type(multifab), intent(inout) :: mf(:)
    integer, intent(in) :: rr(:,:)
    type(bc_level), intent(in) :: bc(:)
    integer, intent(in), optional :: icomp, bcomp, nc, ng
    logical, intent(in), optional :: same_boundary
    integer :: ic, bcp, n
    logical :: sb
    ic = 1; if (present(icomp)) ic = icomp
    bcp = 1; if (present(bcomp)) bcp = bcomp
    n = mf(1)%nc; if (present(nc)) n = nc
    sb = .false.; if (present(same_boundary)) sb = same_boundary
    call vamd_rf(nlevs, mf(1:nlevs)%v, as_vamd_tower(bc(1)), ic, bcp, n, sb)
  end subroutine ml_restrict_and_fill
end module ml_restrict_fill_module

module estdt_module
  use multifab_module
  use varden_amd, only: vamd_estdt => estdt
  implicit none
contains
  ! estdt(lev, u, s, gp, ext_vel_force, dx, dtold, dt)   (src/estdt.f90:15)
  subroutine estdt(lev, u, s, gp, ext_vel_force, dx, dtold, dt)
    integer, intent(in) :: lev
    type(multifab), intent(in) :: u, s, gp, ext_vel_force
    real(dp_t), intent(in) :: dx(:), dtold
    real(dp_t), intent(out) :: dt
    call vamd_estdt(lev, u%v, s%v, gp%v, ext_vel_force%v, dx, dtold, dt)
  end subroutine estdt
end module estdt_module

module hgproject_module
  use multifab_module
  use ml_layout_module
  use define_bc_module
  use varden_amd, only: vamd_hgproject => hgproject
  implicit none
contains
  ! hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, the_bc_tower, press_comp)   (src/hgproject.f90:17-18)
  subroutine hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, the_bc_tower, press_comp)
    integer, intent(in) :: proj_type, press_comp
    type(ml_layout), intent(in) :: mla
    type(multifab), intent(inout) :: unew(:), rhohalf(:), p(:), gp(:)
    type(multifab), intent(in) :: uold(:)
    real(dp_t), intent(in) :: dx(:,:), dt
    type(bc_tower), intent(in) :: the_bc_tower
    call vamd_hgproject(proj_type, mla%v, unew%v, uold%v, rhohalf%v, p%v, gp%v, dx, dt, the_bc_tower%v, press_comp)
  end subroutine hgproject
end module hgproject_module

module macproject_module
  use multifab_module
  use ml_layout_module
  use define_bc_module
  use varden_amd, only: vamd_macproject => macproject
  implicit none
contains
  ! macproject(mla, umac, rho, mac_rhs, dx, the_bc_tower, bc_comp)   (src/macproject.f90:20)
  subroutine macproject(mla, umac, rho, mac_rhs, dx, the_bc_tower, bc_comp)
    type(ml_layout), intent(in) :: mla
    type(multifab), intent(inout) :: umac(:,:), rho(:), mac_rhs(:)
    real(dp_t), intent(in) :: dx(:,:)
    type(bc_tower), intent(in) :: the_bc_tower
    integer, intent(in) :: bc_comp
    call vamd_macproject(mla%v, umac%v, rho%v, dx, the_bc_tower%v, bc_comp, mac_rhs%v)
  end subroutine macproject
end module macproject_module

module advance_module
  use multifab_module
  use ml_layout_module
  use define_bc_module
  use varden_amd, only: vamd_advance => advance_timestep
  implicit none
contains
  ! advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, the_bc_tower, dt, time, dx, press_comp, proj_type)
  ! (src/advance_timestep.f90:26-44)
  subroutine advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, the_bc_tower, dt, time, dx, press_comp, proj_type)
    integer, intent(in) :: istep
    type(ml_layout), intent(in) :: mla
    type(multifab), intent(inout) :: sold(:), uold(:), snew(:), unew(:), gp(:), p(:), ext_vel_force(:), ext_scal_force(:)
    type(bc_tower), intent(in) :: the_bc_tower
    real(dp_t), intent(in) :: dt, time, dx(:,:)
    integer, intent(in) :: press_comp, proj_type
    call vamd_advance(istep, mla%v, sold%v, uold%v, snew%v, unew%v, gp%v, p%v, ext_vel_force%v, ext_scal_force%v, the_bc_tower%v, dt, time, dx, press_comp, proj_type)
  end subroutine advance_timestep
end module advance_module

! the reference's generated probin_module (src/probin.template + src/_parameters through FBoxLib's write_probin.py): EVERY entry of src/_parameters with its default,
! the &PROBIN namelist, probin_init (the inputs file: $PROBIN, else the first command-line argument, else ./inputs_varden -- probin.template:62-98; the command-line
! overrides `--name value` of the generated file are not parsed) and probin_close.  The values reach the library with probin_to_library (vdn_params).
module probin_module
  use bl_types
  use varden_amd, only: vdn_params, probin_defaults, varden_amd_initialize, varden_amd_finalize
  implicit none
  integer, save :: dim_in = 2, nscal = 2, prob_type = 1, boussinesq = 0, max_step = 1, ref_ratio = 2, ng_cell = 3, ng_grow = 1, max_levs = 1, nlevs = -1
  integer, save :: max_grid_size = 256, stencil_order = 2, init_iter = 4, plot_int = 0, chk_int = 0, regrid_int = -1, amr_buf_width = -1
  integer, save :: cluster_min_width = 4, cluster_blocking_factor = 4, use_hypre = 0, verbose = 0, mg_verbose = 0, cg_verbose = 0
  integer, save :: mg_bottom_solver = -1, hg_bottom_solver = -1, max_mg_bottom_nlevels = 1000, do_initial_projection = 1, restart = -1
  integer, save :: bcx_lo = 14, bcy_lo = 14, bcz_lo = 14, bcx_hi = 14, bcy_hi = 14, bcz_hi = 14, diffusion_type = 1, slope_order = 4
  integer, save :: n_cellx = 32, n_celly = 32, n_cellz = 32
  real(dp_t), save :: grav = 0.0_dp_t, stop_time = -1.0_dp_t, cluster_min_eff = 0.9_dp_t
  real(dp_t), save :: prob_lo_x = 0.0_dp_t, prob_lo_y = 0.0_dp_t, prob_lo_z = 0.0_dp_t, prob_hi_x = 1.0_dp_t, prob_hi_y = 1.0_dp_t, prob_hi_z = 1.0_dp_t
  real(dp_t), save :: init_shrink = 1.0_dp_t, fixed_dt = -1.0_dp_t, max_dt_growth = 1.1_dp_t, visc_coef = 0.0_dp_t, diff_coef = 0.0_dp_t, cflfac = 0.8_dp_t
  logical, save :: need_inputs = .true., use_godunov_debug = .false., use_minion = .false.
  character(len=128), save :: fixed_grids = '', grids_file_name = '', plot_base_name = 'plt', check_base_name = 'chk', job_name = ''
  character(len=128), save :: inputs_file_used = ''
  real(dp_t), save :: rho_bc(3,2) = 0.0_dp_t, trac_bc(3,2) = 0.0_dp_t, u_bc(3,2) = 0.0_dp_t, v_bc(3,2) = 0.0_dp_t, w_bc(3,2) = 0.0_dp_t
  ! (the generated file allocates nodal, pmask, prob_lo, prob_hi with dim_in entries in probin_init; fixed extent 3 here, read through (1:dm) or whole)
  logical, save :: pmask(3) = .false., nodal(3) = .true.
  real(dp_t), save :: prob_lo(3) = 0.0_dp_t, prob_hi(3) = 1.0_dp_t
  integer, parameter :: MAX_ALLOWED_LEVS = 10
  integer, save :: extrap_comp = 0
  namelist /probin/ dim_in, nscal, prob_type, grav, boussinesq, max_step, stop_time, ref_ratio, ng_cell, ng_grow, max_levs, nlevs, max_grid_size, stencil_order, &
       init_iter, plot_int, chk_int, regrid_int, amr_buf_width, cluster_min_eff, cluster_min_width, cluster_blocking_factor, prob_lo_x, prob_lo_y, prob_lo_z, &
       prob_hi_x, prob_hi_y, prob_hi_z, use_hypre, verbose, mg_verbose, cg_verbose, mg_bottom_solver, hg_bottom_solver, max_mg_bottom_nlevels, init_shrink, &
       fixed_dt, do_initial_projection, need_inputs, fixed_grids, grids_file_name, restart, bcx_lo, bcy_lo, bcz_lo, bcx_hi, bcy_hi, bcz_hi, diffusion_type, &
       max_dt_growth, slope_order, use_godunov_debug, use_minion, plot_base_name, check_base_name, visc_coef, diff_coef, cflfac, n_cellx, n_celly, n_cellz, job_name, &
       rho_bc, trac_bc, u_bc, v_bc, w_bc
contains
  ! probin.template:44-196 (what the hot path's callers rely on: the namelist read, amr_buf_width >= regrid_int, prob_lo / prob_hi, pmask from the bc flags, extrap_comp)
  subroutine probin_init()
    character(len=128) :: fname
    logical :: lexist, need
    integer :: un, ierr
    need = .true.
    call get_environment_variable('PROBIN', fname, status=ierr)
    if (ierr == 0) then
       call read_nml(fname); need = .false.
    end if
    if (need .and. command_argument_count() >= 1) then
       call get_command_argument(1, value=fname)
       inquire(file=fname, exist=lexist)
       if (lexist) then
          call read_nml(fname); need = .false.
       end if
    end if
    inquire(file='inputs_varden', exist=lexist)
    if (need .and. lexist) then
       call read_nml('inputs_varden'); need = .false.
    end if
    if (max_levs > 1 .and. fixed_grids == '' .and. regrid_int < 1) error stop 'regrid_int must be specified if max_levs > 1'
    if (regrid_int > 0 .and. amr_buf_width < regrid_int) amr_buf_width = regrid_int
    prob_lo = (/ prob_lo_x, prob_lo_y, prob_lo_z /); prob_hi = (/ prob_hi_x, prob_hi_y, prob_hi_z /)
    nodal = .true.
    pmask = .false.
    if (bcx_lo == -1 .and. bcx_hi == -1) pmask(1) = .true.
    if (dim_in > 1 .and. bcy_lo == -1 .and. bcy_hi == -1) pmask(2) = .true.
    if (dim_in > 2 .and. bcz_lo == -1 .and. bcz_hi == -1) pmask(3) = .true.
    extrap_comp = dim_in + nscal + 2
  contains
    subroutine read_nml(f)
      character(len=*), intent(in) :: f
      open(newunit=un, file=f, status='old', action='read')
      read(unit=un, nml=probin)
      close(unit=un)
      inputs_file_used = f
    end subroutine read_nml
  end subroutine probin_init
  subroutine probin_close()
  end subroutine probin_close
  subroutine probin_to_library(rank, nranks, device)
    integer, intent(in) :: rank, nranks, device
    type(vdn_params) :: prm
    call probin_defaults(prm)
    prm%dm = dim_in; prm%nscal = nscal; prm%prob_type = prob_type; prm%slope_order = slope_order; prm%verbose = verbose
    prm%use_minion = merge(1, 0, use_minion); prm%boussinesq = boussinesq; prm%stencil_order = stencil_order; prm%diffusion_type = diffusion_type
    prm%cflfac = cflfac; prm%max_dt_growth = max_dt_growth; prm%visc_coef = visc_coef; prm%diff_coef = diff_coef
    prm%u_bc = transpose(u_bc); prm%v_bc = transpose(v_bc); prm%w_bc = transpose(w_bc); prm%rho_bc = transpose(rho_bc); prm%trac_bc = transpose(trac_bc)
    call varden_amd_initialize(prm, rank, nranks, device)
  end subroutine probin_to_library
end module probin_module
