! varden_drv.f90 -- a thin Fortran driver over the varden_amd module: the flow of the reference's src/varden.f90
! (initial projection 126-138, ghost fills 165-178 / 273-300, estdt 186-199 / 302-318, initial pressure iterations
! 460-490, time loop 237-345) around the MI355X-native advance_timestep, with the reference's argument conventions
! (multifab arrays indexed by level, dx(level, dir), 1-based components).
! Problem: the 3-D density bubble of exec/test/inputs_bubble_3d (prob_type 1, src/initdata.f90:212-238), inviscid,
! no-slip walls.   usage:  varden_drv [n] [nsteps] [nlevs]
!   nlevs = 1: one box of n^3 cells;  nlevs = 2: that box plus one refined box over its central eighth (fine cells
!   n/2 .. 3n/2-1, ref_ratio 2) -- the multi-level conventions of src/varden.f90:273-300 (ml_restrict_and_fill on every
!   state multifab, dt = min over the levels of estdt).
! Every step prints time, dt and max|u| with 17 significant digits: tests/test_fortran_gpu.py compares them with the
! Python mirror of the same flow.
program varden_drv
  use iso_c_binding
  use varden_amd
  implicit none
  integer :: n, nsteps, nlevs, istep, i, j, k, lev, init_iter, dm, nscal, press_comp, count0, count1, rate, nf
  character(len=32) :: arg
  type(vdn_params) :: prm
  type(ml_layout) :: mla
  type(bc_tower) :: the_bc_tower
  type(multifab), allocatable :: uold(:), sold(:), unew(:), snew(:), gp(:), p(:), ext_vel_force(:), ext_scal_force(:), rhohalf(:)
  type(vdn_box), allocatable :: pd(:), boxes(:)
  integer, allocatable :: nboxes(:), owner(:)
  integer :: phys_bc(3, 2), rr(3)
  real(dp_t), allocatable :: dx(:,:)
  real(dp_t) :: dt, dtold, dtlev, time, x, y, z, dist, r, umax, cells
  real(dp_t), allocatable :: s0(:,:,:,:), u0(:,:,:,:)
  integer :: lo(3), hi(3)
  real(dp_t), parameter :: grav = -9.8d0, init_shrink = 0.1d0

  n = 64; nsteps = 5; nlevs = 1
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg, *) n
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg, *) nsteps
  end if
  if (command_argument_count() >= 3) then
     call get_command_argument(3, arg); read(arg, *) nlevs
  end if
  if (nlevs < 1 .or. nlevs > 2) stop 'varden_drv: nlevs must be 1 or 2'

  call probin_defaults(prm)
  prm%cflfac = 0.9d0
  call varden_amd_initialize(prm, 0, 1, 0)
  dm = 3; nscal = prm%nscal; press_comp = dm + nscal + 1; init_iter = 1

  allocate(pd(nlevs), boxes(nlevs), nboxes(nlevs), owner(nlevs), dx(nlevs, 3))
  allocate(uold(nlevs), sold(nlevs), unew(nlevs), snew(nlevs), gp(nlevs), p(nlevs), ext_vel_force(nlevs), ext_scal_force(nlevs), rhohalf(nlevs))
  do lev = 1, nlevs
     nf = n * 2**(lev - 1)
     pd(lev)%lo = 0; pd(lev)%hi = nf - 1
     dx(lev, :) = 1.d0 / nf
  end do
  boxes(1) = pd(1)
  if (nlevs == 2) then
     boxes(2)%lo = n / 2; boxes(2)%hi = 3 * n / 2 - 1
  end if
  rr = 2; nboxes = 1; owner = 0
  call ml_layout_build(mla, nlevs, rr, pd, nboxes, boxes, owner, (/ .false., .false., .false. /))
  phys_bc = NO_SLIP_WALL
  call bc_tower_build(the_bc_tower, mla, phys_bc)

  do lev = 1, nlevs
     call multifab_build(uold(lev), mla, lev, dm, 3);    call multifab_build(sold(lev), mla, lev, nscal, 3)
     call multifab_build(unew(lev), mla, lev, dm, 3);    call multifab_build(snew(lev), mla, lev, nscal, 3)
     call multifab_build(gp(lev), mla, lev, dm, 1);      call multifab_build_nodal(p(lev), mla, lev, 1, 1)
     call multifab_build(ext_vel_force(lev), mla, lev, dm, 1); call multifab_build(ext_scal_force(lev), mla, lev, nscal, 1)
     call setval(ext_vel_force(lev), grav, dm, 1, all=.true.)          ! varden.f90:428-429
     ! initdata_3d, prob_type 1 (initdata.f90:220-238) on the level's box, ghost cells at the background state
     lo = boxes(lev)%lo; hi = boxes(lev)%hi
     allocate(u0(lo(1)-3:hi(1)+3, lo(2)-3:hi(2)+3, lo(3)-3:hi(3)+3, dm), s0(lo(1)-3:hi(1)+3, lo(2)-3:hi(2)+3, lo(3)-3:hi(3)+3, nscal))
     u0 = 0.d0; s0(:,:,:,1) = 1.d0; s0(:,:,:,2) = 0.d0
     do k = lo(3), hi(3)
        z = dx(lev,3) * (k + 0.5d0)
        do j = lo(2), hi(2)
           y = dx(lev,2) * (j + 0.5d0)
           do i = lo(1), hi(1)
              x = dx(lev,1) * (i + 0.5d0)
              dist = sqrt((x - 0.5d0)**2 + (y - 0.5d0)**2 + (z - 0.5d0)**2)
              r = 1.d0 + 0.5d0 * (10.d0 - 1.d0) * (1.d0 - tanh(30.d0 * (dist - 0.1d0)))
              s0(i,j,k,1) = r; s0(i,j,k,2) = r
           end do
        end do
     end do
     call multifab_copy_from_host(uold(lev), 1, u0); call multifab_copy_from_host(sold(lev), 1, s0)
     deallocate(u0, s0)
  end do
  call fill_state_ghosts()

  ! initial projection with rhohalf = 1 (varden.f90:126-138)
  do lev = 1, nlevs
     call multifab_build(rhohalf(lev), mla, lev, 1, 1)
     call setval(rhohalf(lev), 1.d0, all=.true.)
  end do
  call hgproject(initial_projection, mla, uold, uold, rhohalf, p, gp, dx, 1.d0, the_bc_tower, press_comp)
  do lev = 1, nlevs
     call multifab_destroy(rhohalf(lev))
     call setval(p(lev), 0.d0, all=.true.); call setval(gp(lev), 0.d0, all=.true.)
  end do
  call fill_state_ghosts()
  do lev = 1, nlevs
     call multifab_copy_c(unew(lev), 1, uold(lev), 1, dm, 3); call multifab_copy_c(snew(lev), 1, sold(lev), 1, nscal, 3)
  end do

  time = 0.d0
  dt = 1.d20
  do lev = 1, nlevs                                                    ! varden.f90:186-199
     call estdt(lev, uold(lev), sold(lev), gp(lev), ext_vel_force(lev), dx(lev,:), 1.d20, dtlev)
     dt = min(dt, dtlev)
  end do
  dt = dt * init_shrink
  do istep = 1, init_iter                                              ! varden.f90:460-490
     call advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, the_bc_tower, &
                           dt, time, dx, press_comp, pressure_iters)
  end do

  call system_clock(count0, rate)
  do istep = 1, nsteps                                                 ! varden.f90:237-345
     call fill_state_ghosts()
     if (istep > 1) then
        dtold = dt
        dt = 1.d20
        do lev = 1, nlevs                                              ! varden.f90:302-318
           call estdt(lev, uold(lev), sold(lev), gp(lev), ext_vel_force(lev), dx(lev,:), dtold, dtlev)
           dt = min(dt, dtlev)
        end do
     end if
     call advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, the_bc_tower, &
                           dt, time, dx, press_comp, regular_timestep)
     umax = 0.d0
     do lev = 1, nlevs
        call multifab_copy_c(uold(lev), 1, unew(lev), 1, dm); call multifab_copy_c(sold(lev), 1, snew(lev), 1, nscal)
        umax = max(umax, norm_inf(unew(lev)))
     end do
     time = time + dt
     write(*, '(a,i5,a,es25.17,a,es25.17,a,es25.17)') ' step ', istep, '  time ', time, '  dt ', dt, '  |u|max ', umax
  end do
  call system_clock(count1)
  cells = dble(n)**3
  if (nlevs == 2) cells = 2.d0 * cells
  write(*, '(a,es12.4)') ' cells*steps/sec: ', cells * nsteps / (dble(count1 - count0) / rate)

  do lev = 1, nlevs
     call multifab_destroy(uold(lev)); call multifab_destroy(sold(lev)); call multifab_destroy(unew(lev)); call multifab_destroy(snew(lev))
     call multifab_destroy(gp(lev)); call multifab_destroy(p(lev)); call multifab_destroy(ext_vel_force(lev)); call multifab_destroy(ext_scal_force(lev))
  end do
  call bc_tower_destroy(the_bc_tower); call ml_layout_destroy(mla)
  call varden_amd_finalize()

contains

  subroutine fill_state_ghosts()
    if (nlevs == 1) then                                               ! varden.f90:291-300
       call multifab_fill_boundary(uold(1)); call multifab_fill_boundary(sold(1)); call multifab_fill_boundary(gp(1))
       call multifab_physbc(uold(1), 1, 1, dm, the_bc_tower)
       call multifab_physbc(sold(1), 1, dm + 1, nscal, the_bc_tower)
    else                                                               ! varden.f90:273-300 through ml_restrict_and_fill
       call ml_restrict_and_fill(nlevs, uold, the_bc_tower, 1, 1, dm, .false.)
       call ml_restrict_and_fill(nlevs, sold, the_bc_tower, 1, dm + 1, nscal, .false.)
       call ml_restrict_and_fill(nlevs, gp, the_bc_tower, 1, press_comp + 1, dm, .true.)     ! extrap_comp = press_comp + 1
    end if
  end subroutine fill_state_ghosts

end program varden_drv
