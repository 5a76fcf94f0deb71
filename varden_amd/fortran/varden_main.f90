! varden_main.f90 -- the reference's executable flow (src/main.f90 -> src/varden.f90) in Fortran over the varden_amd module:
!   varden_main <inputs file> [max_step override]
! reads the run-time parameters from the &PROBIN namelist of an inputs file (src/_parameters / probin.template; exec/test/inputs_*),
! builds the grids (level 0 cut by max_grid_size; refined levels from tag_boxes + make_new_grids on the initial data, src/initialize.f90:152-342),
! runs the start-up sequence (initial projection varden.f90:126-138, ghost fills :165-178, first dt :186-199, pressure iterations :460-490) and
! the time loop (varden.f90:237-345: regrid every regrid_int steps :256-264 through src/regrid.f90:17-263 -- fillpatch, ml_nodal_prolongation,
! copies of the old data --, ghost fills, estdt over the levels, advance_timestep, new -> old) until max_step or stop_time.
! Scope: dim_in = 3, a cubic unit domain (every 3-D input of exec/test), prob_type 1 .. 4 (src/initdata.f90:190-306), one rank; dim_in = 2 (the four 2-D inputs of
! exec/test, all adaptive): the problem runs as its z-uniform copy -- n_cellx x n_celly x 16 cells of level 0, periodic along z, gravity along y, initdata_2d on every
! plane, velpred_2d's outlet rule (vdn_set_extruded_2d; DESIGN.md section 13): plane k = 0 is the 2-D answer.  Plot and checkpoint files stay with the
! Python mirror (varden_amd/plotfile.py).  The Python mirror of the same flow (varden_amd/inputs.py: run) sits on the same C-ABI: every step prints
! time, dt, max|u| and the boxes per level with 17 significant digits and tests/test_fortran_gpu.py compares the two step for step.
program varden_main
  use iso_c_binding
  use varden_amd
  implicit none
  integer, parameter :: MAXL = 4, MAXB = 8192
  ! ---- src/_parameters (the entries the path reads; defaults as there) ----
  integer :: dim_in = 2, nscal = 2, prob_type = 1, boussinesq = 0, max_step = 1, max_levs = 1, max_grid_size = 256, regrid_int = -1, amr_buf_width = -1
  integer :: n_cellx = 32, n_celly = 32, n_cellz = 32, init_iter = 4, do_initial_projection = 1, diffusion_type = 1, slope_order = 4
  integer :: use_minion = 0, stencil_order = 2, verbose = 0, mg_verbose = 0, plot_int = 0, chk_int = 0, restart = -1, ref_ratio = 2
  integer :: bcx_lo = 14, bcx_hi = 14, bcy_lo = 14, bcy_hi = 14, bcz_lo = 14, bcz_hi = 14, cluster_min_width = 4, cluster_blocking_factor = 4
  real(dp_t) :: grav = 0.d0, stop_time = -1.d0, prob_hi_x = 1.d0, prob_hi_y = 1.d0, prob_hi_z = 1.d0, prob_lo_x = 0.d0, prob_lo_y = 0.d0, prob_lo_z = 0.d0
  real(dp_t) :: init_shrink = 1.d0, cflfac = 0.8d0, max_dt_growth = 1.1d0, visc_coef = 0.d0, diff_coef = 0.d0, fixed_dt = -1.d0, cluster_min_eff = 0.9d0
  real(dp_t) :: u_bc(3,2) = 0.d0, v_bc(3,2) = 0.d0, w_bc(3,2) = 0.d0, rho_bc(3,2) = 1.d0, trac_bc(3,2) = 0.d0
  namelist /probin/ dim_in, nscal, prob_type, boussinesq, max_step, max_levs, max_grid_size, regrid_int, amr_buf_width, n_cellx, n_celly, n_cellz, &
       init_iter, do_initial_projection, diffusion_type, slope_order, use_minion, stencil_order, verbose, mg_verbose, plot_int, chk_int, restart, ref_ratio, &
       bcx_lo, bcx_hi, bcy_lo, bcy_hi, bcz_lo, bcz_hi, cluster_min_width, cluster_blocking_factor, grav, stop_time, prob_hi_x, prob_hi_y, prob_hi_z, &
       prob_lo_x, prob_lo_y, prob_lo_z, init_shrink, cflfac, max_dt_growth, visc_coef, diff_coef, fixed_dt, cluster_min_eff, u_bc, v_bc, w_bc, rho_bc, trac_bc

  ! one hierarchy: box lists, layout, boundary tower and the four multifabs regridding carries (regrid.f90:60-75)
  type hier
     integer :: nlev = 0
     integer :: nb(MAXL) = 0
     type(vdn_box), allocatable :: bx(:,:)          ! (box, level)
     type(ml_layout) :: mla
     type(bc_tower) :: bct
     type(multifab) :: uold(MAXL), sold(MAXL), gp(MAXL), p(MAXL)
  end type hier

  type(hier) :: H
  type(multifab) :: unew(MAXL), snew(MAXL), ext_vel_force(MAXL), ext_scal_force(MAXL), rhohalf(MAXL)
  type(vdn_params) :: prm
  character(len=256) :: fname, arg
  integer :: un, ios, n, dm, press_comp, istep, lev, abw, mgs, nsteps_arg, nregrids
  integer :: nn(3), gdir                                          ! cells of level 0 per direction; the component gravity acts on
  logical :: extruded = .false.
  integer, parameter :: NZ_EXT = 16                               ! cells of level 0 along z of the extruded copy of a 2-D problem
  integer :: phys_bc(3, 2)
  real(dp_t) :: dx(MAXL, 3), dt, dtold, dtlev, time, umax
  type(vdn_box), allocatable :: newb(:)
  integer :: nnew
  logical :: new_grid

  if (command_argument_count() < 1) stop 'usage: varden_main <inputs file> [max_step]'
  call get_command_argument(1, fname)
  open(newunit=un, file=trim(fname), status='old', action='read', iostat=ios)
  if (ios /= 0) stop 'varden_main: cannot open the inputs file'
  read(un, nml=probin, iostat=ios)
  if (ios /= 0) stop 'varden_main: cannot read the &PROBIN namelist'
  close(un)
  nsteps_arg = -1
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg, *) nsteps_arg
  end if
  if (nsteps_arg >= 0) max_step = nsteps_arg
  if (dim_in == 2) then                                         ! the z-uniform copy of the 2-D problem (see the header)
     if (prob_hi_x /= 1.d0 .or. abs(prob_hi_y / n_celly - prob_hi_x / n_cellx) > 1.d-15) stop 'varden_main: dim_in = 2: prob_hi_x = 1 and square cells'
     if (prob_type < 1 .or. prob_type > 3) stop 'varden_main: dim_in = 2: prob_type 1 .. 3 (src/initdata.f90:127-185)'
     extruded = .true.
     n_cellz = NZ_EXT; bcz_lo = PERIODIC; bcz_hi = PERIODIC
     dim_in = 3
  else
     if (dim_in /= 3) stop 'varden_main: dim_in = 2 or 3'
     if (n_cellx /= n_celly .or. n_cellx /= n_cellz .or. prob_hi_x /= 1.d0 .or. prob_hi_y /= 1.d0 .or. prob_hi_z /= 1.d0) &
          stop 'varden_main: cubic unit domain only'
  end if
  if (prob_type < 1 .or. prob_type > 4) stop 'varden_main: prob_type 1 .. 4 (src/initdata.f90)'
  if (max_levs > MAXL) stop 'varden_main: at most 4 levels'

  call probin_defaults(prm)
  prm%dm = dim_in; prm%nscal = nscal; prm%slope_order = slope_order; prm%use_minion = use_minion; prm%boussinesq = boussinesq
  prm%stencil_order = stencil_order; prm%diffusion_type = diffusion_type; prm%verbose = verbose; prm%prob_type = prob_type
  prm%visc_coef = visc_coef; prm%diff_coef = diff_coef; prm%cflfac = cflfac; prm%max_dt_growth = max_dt_growth
  prm%u_bc = transpose(u_bc); prm%v_bc = transpose(v_bc); prm%w_bc = transpose(w_bc); prm%rho_bc = transpose(rho_bc); prm%trac_bc = transpose(trac_bc)
  call varden_amd_initialize(prm, 0, 1, 0)
  if (extruded) call varden_amd_set_extruded_2d(.true.)
  n = n_cellx; dm = 3; press_comp = dm + nscal + 1
  nn = (/ n_cellx, n_celly, n_cellz /)
  gdir = merge(2, 3, extruded)
  mgs = max_grid_size
  abw = max(amr_buf_width, regrid_int, 1)                       ! probin.template:147-154
  phys_bc(1,:) = (/ bcx_lo, bcx_hi /); phys_bc(2,:) = (/ bcy_lo, bcy_hi /); phys_bc(3,:) = (/ bcz_lo, bcz_hi /)
  do lev = 1, MAXL
     dx(lev, :) = 1.d0 / (n * 2**(lev - 1))
  end do
  allocate(newb(MAXB))

  ! ---- grids: initialize_with_adaptive_grids (src/initialize.f90:152-342) ----
  call base_boxes(H)
  call alloc_state(H)
  call init_level(H, 1)
  do lev = 1, max_levs - 1
     call make_new_grids(new_grid, H%sold(lev), lev, abw, merge(0, 2, lev == 1), mgs, newb, nnew)      ! (tag_boxes reads valid cells only)
     if (.not. new_grid) exit
     call grow_hierarchy(H, newb, nnew, carry=.true.)
     call init_level(H, lev + 1)
  end do
  call make_temporaries()
  do lev = 1, H%nlev                                              ! the data of every level from initdata (initialize.f90:326-333)
     call init_level(H, lev)
  end do
  call fill_state_ghosts()

  ! ---- start-up (varden.f90:126-199, 460-490) ----
  time = 0.d0
  if (do_initial_projection > 0) then
     do lev = 1, H%nlev
        call multifab_build(rhohalf(lev), H%mla, lev, 1, 1)
        call setval(rhohalf(lev), 1.d0, all=.true.)
     end do
     call hgproject(initial_projection, H%mla, H%uold(1:H%nlev), H%uold(1:H%nlev), rhohalf(1:H%nlev), H%p(1:H%nlev), H%gp(1:H%nlev), &
                    dx(1:H%nlev,:), 1.d0, H%bct, press_comp)
     do lev = 1, H%nlev
        call multifab_destroy(rhohalf(lev))
        call setval(H%p(lev), 0.d0, all=.true.); call setval(H%gp(lev), 0.d0, all=.true.)
     end do
     call fill_state_ghosts()
  end if
  do lev = 1, H%nlev
     call multifab_copy_c(unew(lev), 1, H%uold(lev), 1, dm, 3); call multifab_copy_c(snew(lev), 1, H%sold(lev), 1, nscal, 3)
  end do
  dt = 1.d20
  do lev = 1, H%nlev
     call estdt(lev, H%uold(lev), H%sold(lev), H%gp(lev), ext_vel_force(lev), dx(lev,:), 1.d20, dtlev)
     dt = min(dt, dtlev)
  end do
  dt = limit_dt(dt * init_shrink, .true.)
  do istep = 1, init_iter
     call advance_timestep(istep, H%mla, H%sold(1:H%nlev), H%uold(1:H%nlev), snew(1:H%nlev), unew(1:H%nlev), H%gp(1:H%nlev), H%p(1:H%nlev), &
                           ext_vel_force(1:H%nlev), ext_scal_force(1:H%nlev), H%bct, dt, time, dx(1:H%nlev,:), press_comp, pressure_iters)
  end do

  ! ---- time loop (varden.f90:237-345) ----
  istep = 0; nregrids = 0
  do while (istep < max_step .and. (stop_time < 0.d0 .or. time < stop_time))
     istep = istep + 1
     if (max_levs > 1 .and. regrid_int > 0) then
        if (mod(istep - 1, regrid_int) == 0) call regrid()
     end if
     call fill_state_ghosts()
     if (istep > 1) then
        dtold = dt
        dt = 1.d20
        do lev = 1, H%nlev
           call estdt(lev, H%uold(lev), H%sold(lev), H%gp(lev), ext_vel_force(lev), dx(lev,:), dtold, dtlev)
           dt = min(dt, dtlev)
        end do
        dt = limit_dt(dt, .false.)
     end if
     call advance_timestep(istep, H%mla, H%sold(1:H%nlev), H%uold(1:H%nlev), snew(1:H%nlev), unew(1:H%nlev), H%gp(1:H%nlev), H%p(1:H%nlev), &
                           ext_vel_force(1:H%nlev), ext_scal_force(1:H%nlev), H%bct, dt, time, dx(1:H%nlev,:), press_comp, regular_timestep)
     umax = 0.d0
     do lev = 1, H%nlev
        call multifab_copy_c(H%uold(lev), 1, unew(lev), 1, dm); call multifab_copy_c(H%sold(lev), 1, snew(lev), 1, nscal)
        umax = max(umax, norm_inf(unew(lev)))
     end do
     time = time + dt
     write(*, '(a,i5,a,es25.17,a,es25.17,a,es25.17,a,i2,a,4i6)') ' step ', istep, '  time ', time, '  dt ', dt, '  |u|max ', umax, &
          '  levels ', H%nlev, '  boxes ', H%nb
  end do
  write(*, '(a,i4)') ' regrids: ', nregrids

  call free_temporaries()
  call free_state(H)
  call varden_amd_finalize()

contains

  ! fixed_dt and stop_time: varden.f90:196-199 (first step) and :318-326
  real(dp_t) function limit_dt(dtin, first)
    real(dp_t), intent(in) :: dtin
    logical, intent(in) :: first
    limit_dt = dtin
    if (fixed_dt > 0.d0) limit_dt = fixed_dt
    if (stop_time >= 0.d0) then
       if (time + limit_dt > stop_time) then
          if (first) then
             limit_dt = min(limit_dt, stop_time - time)
          else
             limit_dt = stop_time - time
          end if
       end if
    end if
  end function limit_dt

  ! level 0 cut by max_grid_size (boxarray_maxsize, initialize.f90:204-206): equal boxes, x fastest
  subroutine base_boxes(G)
    type(hier), intent(inout) :: G
    integer :: nd(3), bs(3), kx, ky, kz, q
    if (.not. allocated(G%bx)) allocate(G%bx(MAXB, MAXL))
    nd = max(1, (nn + mgs - 1) / mgs); bs = nn / nd
    q = 0
    do kz = 0, nd(3) - 1
       do ky = 0, nd(2) - 1
          do kx = 0, nd(1) - 1
             q = q + 1
             G%bx(q, 1)%lo = (/ kx * bs(1), ky * bs(2), kz * bs(3) /)
             G%bx(q, 1)%hi = (/ (kx + 1) * bs(1) - 1, (ky + 1) * bs(2) - 1, (kz + 1) * bs(3) - 1 /)
          end do
       end do
    end do
    G%nlev = 1; G%nb = 0; G%nb(1) = q
  end subroutine base_boxes

  ! layout, boundary tower and the carried multifabs of a hierarchy whose box lists are set (p = 0: regrid.f90:298)
  subroutine alloc_state(G)
    type(hier), intent(inout) :: G
    type(vdn_box) :: pd(MAXL)
    type(vdn_box), allocatable :: flat(:)
    integer, allocatable :: owner(:)
    integer :: l, q, tot, rr(3 * (MAXL - 1))          ! ref_ratio of every level pair, [lev][dir]
    tot = sum(G%nb(1:G%nlev))
    allocate(flat(tot), owner(tot))
    q = 0
    do l = 1, G%nlev
       pd(l)%lo = 0; pd(l)%hi = nn * 2**(l - 1) - 1
       flat(q + 1:q + G%nb(l)) = G%bx(1:G%nb(l), l)
       q = q + G%nb(l)
    end do
    owner = 0; rr = 2
    call ml_layout_build(G%mla, G%nlev, rr, pd(1:G%nlev), G%nb(1:G%nlev), flat, owner, &
                         (/ phys_bc(1,1) == PERIODIC, phys_bc(2,1) == PERIODIC, phys_bc(3,1) == PERIODIC /))
    call bc_tower_build(G%bct, G%mla, phys_bc)
    do l = 1, G%nlev
       call multifab_build(G%uold(l), G%mla, l, dm, 3); call multifab_build(G%sold(l), G%mla, l, nscal, 3)
       call multifab_build(G%gp(l), G%mla, l, dm, 1);   call multifab_build_nodal(G%p(l), G%mla, l, 1, 1)
       call setval(G%p(l), 0.d0, all=.true.)
    end do
  end subroutine alloc_state

  subroutine free_state(G)
    type(hier), intent(inout) :: G
    integer :: l
    do l = 1, G%nlev
       call multifab_destroy(G%uold(l)); call multifab_destroy(G%sold(l)); call multifab_destroy(G%gp(l)); call multifab_destroy(G%p(l))
    end do
    call bc_tower_destroy(G%bct); call ml_layout_destroy(G%mla)
  end subroutine free_state

  ! the hierarchy with one more level (box list newb): new multifabs, the data of the existing levels copied over when carry is set
  subroutine grow_hierarchy(G, nb_list, nbn, carry)
    type(hier), intent(inout) :: G
    type(vdn_box), intent(in) :: nb_list(:)
    integer, intent(in) :: nbn
    logical, intent(in) :: carry
    type(hier) :: N
    integer :: l
    allocate(N%bx(MAXB, MAXL))
    N%nlev = G%nlev + 1; N%nb = G%nb; N%bx = G%bx
    N%nb(N%nlev) = nbn; N%bx(1:nbn, N%nlev) = nb_list(1:nbn)
    call alloc_state(N)
    if (carry) then
       do l = 1, G%nlev
          call copy_level(N, l, G, l)
       end do
    end if
    call free_state(G)
    call move_hier(N, G)
  end subroutine grow_hierarchy

  subroutine move_hier(src, dst)
    type(hier), intent(inout) :: src, dst
    dst%nlev = src%nlev; dst%nb = src%nb
    if (.not. allocated(dst%bx)) allocate(dst%bx(MAXB, MAXL))
    dst%bx = src%bx
    dst%mla = src%mla; dst%bct = src%bct
    dst%uold = src%uold; dst%sold = src%sold; dst%gp = src%gp; dst%p = src%p
  end subroutine move_hier

  ! the four carried multifabs of level ld of D := those of level ls of S wherever the box lists overlap (multifab_copy_c across layouts)
  subroutine copy_level(D, ld, S, ls)
    type(hier), intent(inout) :: D
    type(hier), intent(in) :: S
    integer, intent(in) :: ld, ls
    call multifab_copy_layouts(D%uold(ld), 1, S%uold(ls), 1, dm); call multifab_copy_layouts(D%sold(ld), 1, S%sold(ls), 1, nscal)
    call multifab_copy_layouts(D%gp(ld), 1, S%gp(ls), 1, dm);     call multifab_copy_layouts(D%p(ld), 1, S%p(ls), 1, 1)
  end subroutine copy_level

  ! ghost cells of the levels 1 .. nl (what tagging and fillpatch read)
  subroutine fill_levels(G, nl, only_s)
    type(hier), intent(inout) :: G
    integer, intent(in) :: nl
    logical, intent(in), optional :: only_s
    logical :: os
    os = .false.; if (present(only_s)) os = only_s
    call ml_restrict_and_fill(nl, G%sold(1:nl), G%bct, 1, dm + 1, nscal, .false.)
    if (os) return
    call ml_restrict_and_fill(nl, G%uold(1:nl), G%bct, 1, 1, dm, .false.)
    call ml_restrict_and_fill(nl, G%gp(1:nl), G%bct, 1, press_comp + 1, dm, .true.)
  end subroutine fill_levels

  subroutine fill_state_ghosts()
    call fill_levels(H, H%nlev)
  end subroutine fill_state_ghosts

  ! initdata_3d (src/initdata.f90:190-306) on every box of level l; ghost cells at the background state.  prob_type 1: the bubble (u = 0, rho = tracer = the
  ! tanh blob, :212-238); 2: the same blob advected by u = (1, 0, 0) (:240-259); 3: the Rayleigh-Taylor interface (:195-200, 261-274; tracer 0);
  ! 4: the vortex tube (:276-306; coordinates measured from the BOX's low corner, as the reference writes it)
  subroutine init_level(G, l)
    type(hier), intent(inout) :: G
    integer, intent(in) :: l
    real(dp_t), allocatable :: s0(:,:,:,:), u0(:,:,:,:)
    integer :: b, i, j, k, lo(3), hi(3)
    real(dp_t) :: x, y, z, dist, r, xb, yb, zb, ryz
    real(dp_t), parameter :: pi = 3.141592653589793238462643383279502884d0
    do b = 1, G%nb(l)
       lo = G%bx(b, l)%lo; hi = G%bx(b, l)%hi
       allocate(u0(lo(1)-3:hi(1)+3, lo(2)-3:hi(2)+3, lo(3)-3:hi(3)+3, dm), s0(lo(1)-3:hi(1)+3, lo(2)-3:hi(2)+3, lo(3)-3:hi(3)+3, nscal))
       u0 = 0.d0; s0(:,:,:,1) = 1.d0; s0(:,:,:,2:) = 0.d0
       if (prob_type == 2) u0(:,:,:,1) = 1.d0
       do k = lo(3), hi(3)
          z = dx(l,3) * (k + 0.5d0)
          do j = lo(2), hi(2)
             y = dx(l,2) * (j + 0.5d0)
             do i = lo(1), hi(1)
                x = dx(l,1) * (i + 0.5d0)
                if (extruded) then                          ! initdata_2d (src/initdata.f90:127-185; densfact = 2) on every plane
                   select case (prob_type)
                   case (1, 2)
                      dist = sqrt((x - 0.5d0)**2 + (y - 0.5d0 * nn(2) / dble(nn(1)))**2)
                      r = 1.d0 + 0.5d0 * (2.d0 - 1.d0) * (1.d0 - tanh(30.d0 * (dist - 0.1d0)))
                      s0(i,j,k,1) = r
                      if (nscal > 1) s0(i,j,k,2) = r
                   case (3)
                      s0(i,j,k,1) = 1.d0 + 0.5d0 + 0.5d0 * tanh((y - 0.5d0 - pert(x)) / 0.01d0)
                   end select
                   cycle
                end if
                select case (prob_type)
                case (1, 2)
                   dist = sqrt((x - 0.5d0)**2 + (y - 0.5d0)**2 + (z - 0.5d0)**2)
                   r = 1.d0 + 0.5d0 * (10.d0 - 1.d0) * (1.d0 - tanh(30.d0 * (dist - 0.1d0)))
                   s0(i,j,k,1) = r
                   if (nscal > 1) s0(i,j,k,2) = r
                case (3)
                   s0(i,j,k,1) = 1.d0 + 0.5d0 + 0.5d0 * tanh((z - 0.5d0 - pert(x) - pert(y)) / 0.01d0)
                case (4)
                   xb = dx(l,1) * (i - lo(1) + 0.5d0) - 0.5d0; yb = dx(l,2) * (j - lo(2) + 0.5d0) - 0.5d0; zb = dx(l,3) * (k - lo(3) + 0.5d0) - 0.5d0
                   ryz = sqrt(yb * yb + zb * zb)
                   u0(i,j,k,1) = tanh((0.15d0 - ryz) / 0.0333d0)
                   u0(i,j,k,3) = 0.05d0 * exp(-15.d0 * (xb * xb + yb * yb))
                   s0(i,j,k,1) = 1.d0
                   if (nscal > 1) s0(i,j,k,2) = exp(-500.d0 * (0.15d0 - ryz)**2)
                end select
             end do
          end do
       end do
       call multifab_copy_from_host(G%uold(l), b, u0); call multifab_copy_from_host(G%sold(l), b, s0)
       deallocate(u0, s0)
    end do
  contains
    real(dp_t) function pert(t)
      real(dp_t), intent(in) :: t
      pert = 0.02d0 * sin(4.d0 * pi * t) + 0.01d0 * sin(8.d0 * pi * t)
    end function pert
  end subroutine init_level

  subroutine make_temporaries()
    integer :: l
    do l = 1, H%nlev
       call multifab_build(unew(l), H%mla, l, dm, 3); call multifab_build(snew(l), H%mla, l, nscal, 3)
       call multifab_build(ext_vel_force(l), H%mla, l, dm, 1); call multifab_build(ext_scal_force(l), H%mla, l, nscal, 1)
       call setval(ext_vel_force(l), grav, gdir, 1, all=.true.)        ! varden.f90:428-429 (the extruded copy of a 2-D problem: along y)
    end do
  end subroutine make_temporaries

  subroutine free_temporaries()
    integer :: l
    do l = 1, H%nlev
       call multifab_destroy(unew(l)); call multifab_destroy(snew(l)); call multifab_destroy(ext_vel_force(l)); call multifab_destroy(ext_scal_force(l))
    end do
  end subroutine free_temporaries

  ! src/regrid.f90:17-263: new grids level by level from the current state (tag_boxes + make_new_grids on the already regridded level below),
  ! build_and_fill_data :269-339 (interpolation from the coarser level, the old data of the level copied over it), then the temporaries
  subroutine regrid()
    type(hier) :: C, N
    integer :: l, ll, old_nlev
    logical :: ng
    old_nlev = H%nlev
    allocate(C%bx(MAXB, MAXL))
    C%nlev = 1; C%nb = 0; C%nb(1) = H%nb(1); C%bx(:, 1) = H%bx(:, 1)
    call alloc_state(C)
    call copy_level(C, 1, H, 1)
    l = 1
    do while (l < max_levs)
       call fill_levels(C, l)
       call make_new_grids(ng, C%sold(l), l, abw, merge(0, 2, l == 1), mgs, newb, nnew)
       if (.not. ng) exit
       allocate(N%bx(MAXB, MAXL))
       N%nlev = C%nlev + 1; N%nb = C%nb; N%bx = C%bx
       N%nb(N%nlev) = nnew; N%bx(1:nnew, N%nlev) = newb(1:nnew)
       call alloc_state(N)
       do ll = 1, l
          call copy_level(N, ll, C, ll)
       end do
       call fill_levels(N, l)
       call fillpatch(N%uold(l + 1), N%uold(l), 1, dm); call fillpatch(N%sold(l + 1), N%sold(l), 1, nscal)
       call fillpatch(N%gp(l + 1), N%gp(l), 1, dm);     call ml_nodal_prolongation(N%p(l + 1), N%p(l))
       if (old_nlev > l) call copy_level(N, l + 1, H, l + 1)
       call free_state(C)
       call move_hier(N, C)
       deallocate(N%bx)
       l = l + 1
    end do
    call free_temporaries()
    call free_state(H)
    call move_hier(C, H)
    call make_temporaries()
    call fill_state_ghosts()                                           ! regrid.f90:252-254
    do ll = 1, H%nlev
       call multifab_copy_c(unew(ll), 1, H%uold(ll), 1, dm, 3); call multifab_copy_c(snew(ll), 1, H%sold(ll), 1, nscal, 3)
       call multifab_fill_boundary(H%p(ll))
    end do
    nregrids = nregrids + 1
  end subroutine regrid

end program varden_main
