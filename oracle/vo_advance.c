/* oracle/vo_advance.c -- advance_timestep orchestration, reference src/advance_timestep.f90:26-170,
 * advance_premac.f90:17-59, scalar_advance.f90:17-171, velocity_advance.f90:17-140 -- one level, one box.
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * ml_restrict_and_fill (FBoxLib, external) on one level = multifab_fill_boundary + multifab_physbc.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include "vo.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void fab_new(vo_fab *f, const int *lo, const int *hi, int ng, int face_dir, int nc, double val)
{
  int nd[3] = { 0, 0, 0 }; if (face_dir >= 0) nd[face_dir] = 1;
  vo_fab_init(f, NULL, lo, hi, ng, nd, nc);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}

/* fill_boundary + physbc with one bc component for all comps (same_boundary) or consecutive ones */
static void restrict_and_fill(vo_fab *f, int icomp, int bcomp, int nc, int same_boundary, const vo_bc *bc,
                              const int pmask[3], const vdn_params *prm)
{
  vo_fill_boundary(f, pmask);
  for (int c = 0; c < nc; c++) vo_physbc(f, icomp + c, same_boundary ? bcomp : bcomp + c, 1, bc, prm);
}

void vo_advance_timestep(vo_state *S, const double dx[3], double dt, const vo_bc *bc, const int pmask[3],
                         const vdn_params *prm, int proj_type, vo_mgstat st[2], double phase_sec[4])
{
  const int *lo = S->uold.lo, *hi = S->uold.hi;
  const int dm = 3, nscal = prm->nscal;
  vo_fab mac_rhs, rhohalf, umac[3], *ump[3], vel_force, scal_force, divu;
  vo_fab sedge[3], sflux[3], uedge[3], uflux[3], *sep[3], *sfp[3], *uep[3], *ufp[3];
  double t0;

  const int viscous = prm->visc_coef > 0.0, diffusive = prm->diff_coef > 0.0;
  vo_fab lapu, laps;

  /* advance_timestep.f90:65-80 */
  fab_new(&mac_rhs, lo, hi, 1, -1, 1, 0.0);
  fab_new(&rhohalf, lo, hi, 1, -1, dm, 0.0);
  for (int d = 0; d < 3; d++) { fab_new(&umac[d], lo, hi, 1, d, 1, 1.e20); ump[d] = &umac[d]; }

  /* lapu: advance_timestep.f90:85-93 */
  fab_new(&lapu, lo, hi, 0, -1, dm, 0.0);
  if (viscous) for (int c = 0; c < dm; c++) vo_explicit_diffusive_term(&lapu, &S->uold, c, c, dx, bc);

  /* advance_premac.f90:44-51: vel_force(visc_fac=1, s=sold) -> velpred (+ fill_boundary(umac), velpred.f90:108-112) */
  fab_new(&vel_force, lo, hi, 1, -1, dm, 0.0);
  vo_mkvelforce(&vel_force, &S->ext_vel_force, &S->gp, &S->sold, viscous ? &lapu : NULL, 1.0, prm);
  restrict_and_fill(&vel_force, 0, bc->extrap_comp, dm, 1, bc, pmask, prm);           /* mkforce.f90:75-76 */
  vo_velpred(&S->uold, ump, &vel_force, dx, dt, bc, prm);
  for (int d = 0; d < 3; d++) vo_fill_boundary(&umac[d], pmask);
  free(vel_force.p);

  /* MAC projection (advance_timestep.f90:100) */
  t0 = now();
  vo_macproject(ump, &S->sold, &mac_rhs, dx, bc, pmask, prm, &st[0]);
  if (phase_sec) phase_sec[2] = now() - t0;

  /* scalar_advance.f90:54-118 */
  t0 = now();
  {
    int is_cons[VO_MAXCOMP]; is_cons[0] = 1; for (int c = 1; c < nscal; c++) is_cons[c] = 0;
    fab_new(&scal_force, lo, hi, 1, -1, nscal, 0.0);
    fab_new(&divu, lo, hi, 1, -1, 1, 0.0);
    for (int d = 0; d < 3; d++) { fab_new(&sflux[d], lo, hi, 0, d, nscal, 0.0); fab_new(&sedge[d], lo, hi, 0, d, nscal, 0.0); sfp[d] = &sflux[d]; sep[d] = &sedge[d]; }
    fab_new(&laps, lo, hi, 0, -1, nscal, 0.0);
    if (diffusive) for (int c = 1; c < nscal; c++) vo_explicit_diffusive_term(&laps, &S->sold, c, dm + c, dx, bc);   /* scalar_advance.f90:80-89 */
    vo_mkscalforce(&scal_force, &S->ext_scal_force, diffusive ? &laps : NULL, 1.0, prm);
    restrict_and_fill(&scal_force, 0, bc->extrap_comp, nscal, 1, bc, pmask, prm);      /* mkforce.f90:283-284 */
    vo_mkflux(&S->sold, sep, sfp, ump, &scal_force, &divu, dx, dt, 0, is_cons, dm, bc, prm);
    vo_mkscalforce(&scal_force, &S->ext_scal_force, diffusive ? &laps : NULL, 0.0, prm);
    restrict_and_fill(&scal_force, 0, bc->extrap_comp, nscal, 1, bc, pmask, prm);
    vo_update(&S->sold, ump, sep, sfp, &scal_force, &S->snew, dx, dt, 0, is_cons);
    restrict_and_fill(&S->snew, 0, dm, nscal, 0, bc, pmask, prm);                      /* update.f90:106 */
    if (diffusive) {                                                                   /* scalar_advance.f90:144-162 */
      double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->diff_coef : dt * prm->diff_coef;
      vo_mgstat sst;
      for (int c = 1; c < nscal; c++) vo_diff_scalar_solve(&S->snew, &laps, dx, visc_mu, bc, pmask, prm, c, dm + c, &sst);
    }
    free(laps.p);
    free(scal_force.p); free(divu.p);
    for (int d = 0; d < 3; d++) { free(sflux[d].p); free(sedge[d].p); }
  }
  if (phase_sec) phase_sec[0] = now() - t0;

  /* make_at_halftime(rhohalf, sold, snew, 1, 1) (advance_timestep.f90:114; make_at_halftime.f90:64-65) */
  vo_make_at_halftime(&rhohalf, 0, &S->sold, &S->snew, 0);
  restrict_and_fill(&rhohalf, 0, dm + 0, 1, 0, bc, pmask, prm);
  if (prm->diffusion_type == 2) memset(lapu.p, 0, sizeof(double) * vo_size(&lapu));    /* advance_timestep.f90:116-120 */

  /* velocity_advance.f90:48-93 */
  t0 = now();
  {
    int is_cons[3] = { 0, 0, 0 };
    fab_new(&vel_force, lo, hi, 1, -1, dm, 0.0);
    for (int d = 0; d < 3; d++) { fab_new(&uflux[d], lo, hi, 0, d, dm, 0.0); fab_new(&uedge[d], lo, hi, 0, d, dm, 0.0); ufp[d] = &uflux[d]; uep[d] = &uedge[d]; }
    vo_mkvelforce(&vel_force, &S->ext_vel_force, &S->gp, &S->sold, viscous ? &lapu : NULL, 1.0, prm);
    restrict_and_fill(&vel_force, 0, bc->extrap_comp, dm, 1, bc, pmask, prm);
    vo_mkflux(&S->uold, uep, ufp, ump, &vel_force, &mac_rhs, dx, dt, 1, is_cons, 0, bc, prm);
    vo_mkvelforce(&vel_force, &S->ext_vel_force, &S->gp, &rhohalf, viscous ? &lapu : NULL, 0.0, prm);
    restrict_and_fill(&vel_force, 0, bc->extrap_comp, dm, 1, bc, pmask, prm);
    vo_update(&S->uold, ump, uep, ufp, &vel_force, &S->unew, dx, dt, 1, is_cons);
    restrict_and_fill(&S->unew, 0, 0, dm, 0, bc, pmask, prm);                          /* update.f90:104 */
    if (viscous) {                                                                     /* velocity_advance.f90:103-118 */
      double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->visc_coef : dt * prm->visc_coef;
      vo_mgstat vst;
      vo_visc_solve(&S->unew, &lapu, &rhohalf, &mac_rhs, dx, visc_mu, bc, pmask, prm, &vst);
    }
    free(vel_force.p);
    for (int d = 0; d < 3; d++) { free(uflux[d].p); free(uedge[d].p); }
  }
  if (phase_sec) phase_sec[1] = now() - t0;

  /* hgproject (advance_timestep.f90:133) */
  t0 = now();
  vo_hgproject(proj_type, &S->unew, &S->uold, &rhohalf, &S->p, &S->gp, dx, dt, bc, pmask, prm, &st[1]);
  if (phase_sec) phase_sec[3] = now() - t0;

  free(lapu.p);
  free(mac_rhs.p); free(rhohalf.p);
  for (int d = 0; d < 3; d++) free(umac[d].p);
}
