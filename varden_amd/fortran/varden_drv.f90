! varden_drv.f90 -- a thin Fortran driver over the varden_amd module: the single-level flow of the
! reference's src/varden.f90 (initial projection 126-138, ghost fills, estdt, initial pressure
! iterations 460-490, time loop 237-345) around the MI355X-native advance_timestep.
! Problem: the 3-D density bubble of exec/test/inputs_bubble_3d (prob_type 1, src/initdata.f90:212-238),
! inviscid, one box of n^3 cells.   usage:  varden_drv [n] [nsteps]
program varden_drv
  use iso_c_binding
  use varden_amd
  implicit none
  integer :: n, nsteps, istep, i, j, k, init_iter, dm, nscal, press_comp, count0, count1, rate
  character(len=32) :: arg
  type(vdn_params) :: prm
  type(ml_layout) :: mla
  type(bc_tower) :: the_bc_tower
  type(multifab) :: uold(1), sold(1), unew(1), snew(1), gp(1), p(1), ext_vel_force(1), ext_scal_force(1), rhohalf(1)
  type(vdn_box) :: pd(1), boxes(1)
  integer :: phys_bc(3, 2), rr(1), nboxes(1), owner(1)
  real(dp_t) :: dx(1, 3), dt, dtold, time, x, y, z, dist, r
  real(dp_t), allocatable :: s0(:,:,:,:), u0(:,:,:,:)
  real(dp_t), parameter :: grav = -9.8d0, init_shrink = 0.1d0

  n = 64; nsteps = 5
  if (command_argument_count() >= 1) then
     call get_command_argument(1, arg); read(arg, *) n
  end if
  if (command_argument_count() >= 2) then
     call get_command_argument(2, arg); read(arg, *) nsteps
  end if

  call probin_defaults(prm)
  prm%cflfac = 0.9d0
  call varden_amd_initialize(prm, 0, 1, 0)
  dm = 3; nscal = prm%nscal; press_comp = dm + nscal + 1; init_iter = 1

  pd(1)%lo = 0; pd(1)%hi = n - 1; boxes(1) = pd(1)
  rr = 2; nboxes = 1; owner = 0
  call ml_layout_build(mla, 1, rr, pd, nboxes, boxes, owner, (/ .false., .false., .false. /))
  phys_bc = NO_SLIP_WALL
  call bc_tower_build(the_bc_tower, mla, phys_bc)
  dx(1, :) = 1.d0 / n

  call multifab_build(uold(1), mla, 1, dm, 3);    call multifab_build(sold(1), mla, 1, nscal, 3)
  call multifab_build(unew(1), mla, 1, dm, 3);    call multifab_build(snew(1), mla, 1, nscal, 3)
  call multifab_build(gp(1), mla, 1, dm, 1);      call multifab_build_nodal(p(1), mla, 1, 1, 1)
  call multifab_build(ext_vel_force(1), mla, 1, dm, 1); call multifab_build(ext_scal_force(1), mla, 1, nscal, 1)
  call setval(ext_vel_force(1), grav, dm, 1, all=.true.)          ! varden.f90:428-429

  ! initdata_3d, prob_type 1 (initdata.f90:220-238)
  allocate(u0(-3:n+2, -3:n+2, -3:n+2, dm), s0(-3:n+2, -3:n+2, -3:n+2, nscal))
  u0 = 0.d0; s0(:,:,:,1) = 1.d0; s0(:,:,:,2) = 0.d0
  do k = 0, n - 1
     z = dx(1,3) * (k + 0.5d0)
     do j = 0, n - 1
        y = dx(1,2) * (j + 0.5d0)
        do i = 0, n - 1
           x = dx(1,1) * (i + 0.5d0)
           dist = sqrt((x - 0.5d0)**2 + (y - 0.5d0)**2 + (z - 0.5d0)**2)
           r = 1.d0 + 0.5d0 * (10.d0 - 1.d0) * (1.d0 - tanh(30.d0 * (dist - 0.1d0)))
           s0(i,j,k,1) = r; s0(i,j,k,2) = r
        end do
     end do
  end do
  call multifab_copy_from_host(uold(1), 1, u0); call multifab_copy_from_host(sold(1), 1, s0)
  deallocate(u0, s0)
  call fill_state_ghosts()

  ! initial projection with rhohalf = 1 (varden.f90:126-138)
  call multifab_build(rhohalf(1), mla, 1, 1, 1)
  call setval(rhohalf(1), 1.d0, all=.true.)
  call hgproject(initial_projection, mla, uold, uold, rhohalf, p, gp, dx, 1.d0, the_bc_tower, press_comp)
  call multifab_destroy(rhohalf(1))
  call setval(p(1), 0.d0, all=.true.); call setval(gp(1), 0.d0, all=.true.)
  call fill_state_ghosts()
  call multifab_copy_c(unew(1), 1, uold(1), 1, dm, 3); call multifab_copy_c(snew(1), 1, sold(1), 1, nscal, 3)

  time = 0.d0
  call estdt(1, uold(1), sold(1), gp(1), ext_vel_force(1), dx(1,:), 1.d20, dt)
  dt = dt * init_shrink
  do istep = 1, init_iter                                           ! varden.f90:460-490
     call advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, the_bc_tower, &
                           dt, time, dx, press_comp, pressure_iters)
  end do

  call system_clock(count0, rate)
  do istep = 1, nsteps                                              ! varden.f90:237-345
     call fill_state_ghosts()
     if (istep > 1) then
        dtold = dt
        call estdt(1, uold(1), sold(1), gp(1), ext_vel_force(1), dx(1,:), dtold, dt)
     end if
     call advance_timestep(istep, mla, sold, uold, snew, unew, gp, p, ext_vel_force, ext_scal_force, the_bc_tower, &
                           dt, time, dx, press_comp, regular_timestep)
     call multifab_copy_c(uold(1), 1, unew(1), 1, dm); call multifab_copy_c(sold(1), 1, snew(1), 1, nscal)
     time = time + dt
     write(*, '(a,i5,a,es14.6,a,es14.6,a,es12.4)') ' step ', istep, '  time ', time, '  dt ', dt, '  |u|max ', norm_inf(unew(1))
  end do
  call system_clock(count1)
  write(*, '(a,es12.4)') ' cells*steps/sec: ', dble(n)**3 * nsteps / (dble(count1 - count0) / rate)

  call multifab_destroy(uold(1)); call multifab_destroy(sold(1)); call multifab_destroy(unew(1)); call multifab_destroy(snew(1))
  call multifab_destroy(gp(1)); call multifab_destroy(p(1)); call multifab_destroy(ext_vel_force(1)); call multifab_destroy(ext_scal_force(1))
  call bc_tower_destroy(the_bc_tower); call ml_layout_destroy(mla)
  call varden_amd_finalize()

contains

  subroutine fill_state_ghosts()                                    ! varden.f90:291-300
    call multifab_fill_boundary(uold(1)); call multifab_fill_boundary(sold(1)); call multifab_fill_boundary(gp(1))
    call multifab_physbc(uold(1), 1, 1, dm, the_bc_tower)
    call multifab_physbc(sold(1), 1, dm + 1, nscal, the_bc_tower)
  end subroutine fill_state_ghosts

end program varden_drv
