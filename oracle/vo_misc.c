/* oracle/vo_misc.c -- bc tables, ghost fills, forcing, update, rho-half, estdt, initdata.
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "vo.h"

/* ------------------------------------------------------------------------------------------
 * define_bc_tower.f90:158-252 (adv) and 254-340 (ell), for one box whose faces carry phys[][]
 * ---------------------------------------------------------------------------------------- */
void vo_bc_build(vo_bc *bc, const int phys[3][2], int dm, int nscal)
{
  int press = dm + nscal, extrap = press + 1;          /* 0-based */
  bc->press_comp = press; bc->extrap_comp = extrap;
  bc->ncomp_adv = dm + nscal + 2; bc->ncomp_ell = dm + nscal + 1;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    bc->phys[d][s] = phys[d][s];
    for (int c = 0; c < VO_MAXCOMP; c++) { bc->adv[d][s][c] = VDN_INTERIOR; bc->ell[d][s][c] = VDN_BC_INT; }
    int p = phys[d][s];
    int *a = bc->adv[d][s], *e = bc->ell[d][s];
    if (p == VDN_SLIP_WALL) {
      for (int c = 0; c < dm; c++) a[c] = VDN_HOEXTRAP;
      a[d] = VDN_EXT_DIR;
      for (int n = 0; n < nscal; n++) a[dm + n] = VDN_HOEXTRAP;
      a[press] = VDN_FOEXTRAP; a[extrap] = VDN_FOEXTRAP;
      for (int c = 0; c < dm; c++) e[c] = VDN_BC_NEU;
      e[d] = VDN_BC_DIR;
      for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
      e[press] = VDN_BC_NEU;
    } else if (p == VDN_NO_SLIP_WALL) {
      for (int c = 0; c < dm; c++) a[c] = VDN_EXT_DIR;
      for (int n = 0; n < nscal; n++) a[dm + n] = VDN_HOEXTRAP;
      a[press] = VDN_FOEXTRAP; a[extrap] = VDN_FOEXTRAP;
      for (int c = 0; c < dm; c++) e[c] = VDN_BC_DIR;
      for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
      e[press] = VDN_BC_NEU;
    } else if (p == VDN_INLET) {
      for (int c = 0; c < dm; c++) a[c] = VDN_EXT_DIR;
      for (int n = 0; n < nscal; n++) a[dm + n] = VDN_EXT_DIR;
      a[press] = VDN_FOEXTRAP; a[extrap] = VDN_FOEXTRAP;
      for (int c = 0; c < dm; c++) e[c] = VDN_BC_DIR;
      for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_DIR;
      e[press] = VDN_BC_NEU;
    } else if (p == VDN_OUTLET) {
      for (int c = 0; c < dm; c++) a[c] = VDN_FOEXTRAP;
      for (int n = 0; n < nscal; n++) a[dm + n] = VDN_FOEXTRAP;
      a[press] = VDN_EXT_DIR; a[extrap] = VDN_FOEXTRAP;
      for (int c = 0; c < dm; c++) e[c] = VDN_BC_NEU;
      for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
      e[press] = VDN_BC_DIR;
    } else if (p == VDN_SYMMETRY) {
      for (int c = 0; c < dm; c++) a[c] = VDN_REFLECT_EVEN;
      a[d] = VDN_REFLECT_ODD;
      for (int n = 0; n < nscal; n++) a[dm + n] = VDN_REFLECT_EVEN;
      a[press] = VDN_EXT_DIR; a[extrap] = VDN_REFLECT_EVEN;
      for (int c = 0; c < dm; c++) e[c] = VDN_BC_NEU;
      e[d] = VDN_BC_DIR;
      for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
      e[press] = VDN_BC_NEU;
    } else if (p == VDN_PERIODIC) {
      /* adv stays INTERIOR (define_bc_tower.f90:199-246 has no PERIODIC branch) */
      for (int c = 0; c < dm + nscal + 1; c++) e[c] = VDN_BC_PER;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * multifab_fill_boundary for ONE box covering the whole domain: periodic images only.
 * A point (cell, face or node) outside the valid index range of the fab in a periodic
 * direction is copied from the valid point shifted by the period.  [FBoxLib semantics,
 * external to the reference tree.]
 * ---------------------------------------------------------------------------------------- */
void vo_fill_boundary(vo_fab *f, const int pmask[3])
{
  int any = 0; for (int d = 0; d < 3; d++) if (pmask[d]) any = 1;
  if (!any || f->ng == 0) return;
  int per[3], vlo[3], vhi[3];
  for (int d = 0; d < 3; d++) { per[d] = f->hi[d] - f->lo[d] + 1; vlo[d] = f->lo[d]; vhi[d] = f->hi[d] + f->nd[d]; }
  int ng = f->ng;
  for (int c = 0; c < f->nc; c++)
  for (int k = vlo[2] - f->gz; k <= vhi[2] + f->gz; k++)
  for (int j = vlo[1] - ng; j <= vhi[1] + ng; j++)
  for (int i = vlo[0] - ng; i <= vhi[0] + ng; i++) {
    int idx[3] = { i, j, k }, src[3] = { i, j, k }, ghost = 0, ok = 1;
    for (int d = 0; d < 3; d++) {
      if (idx[d] < vlo[d]) { ghost = 1; if (pmask[d]) src[d] = idx[d] + per[d]; else ok = 0; }
      else if (idx[d] > vhi[d]) { ghost = 1; if (pmask[d]) src[d] = idx[d] - per[d]; else ok = 0; }
    }
    if (ghost && ok) VF(f, i, j, k, c) = VF(f, src[0], src[1], src[2], c);
  }
}

/* ------------------------------------------------------------------------------------------
 * multifab_physbc.f90:238-561 (physbc_3d), one component at a time.
 * ---------------------------------------------------------------------------------------- */
static double extdir_value(const vdn_params *prm, int icomp1, int d, int s, int *has, int dm)
{
  *has = 1;
  if (dm == 2) {                 /* multifab_physbc.f90:96-99: u, v, rho, tracer */
    switch (icomp1) { case 1: return prm->u_bc[d][s]; case 2: return prm->v_bc[d][s]; case 3: return prm->rho_bc[d][s]; case 4: return prm->trac_bc[d][s]; }
    *has = 0; return 0.0;
  }
  switch (icomp1) {              /* 1-based bc component, multifab_physbc.f90:282-287 */
    case 1: return prm->u_bc[d][s];
    case 2: return prm->v_bc[d][s];
    case 3: return prm->w_bc[d][s];
    case 4: return prm->rho_bc[d][s];
    case 5: return prm->trac_bc[d][s];
  }
  *has = 0; return 0.0;
}

static void physbc_one(vo_fab *f, int scomp, const int bc[3][2], int icomp1, const vdn_params *prm)
{
  int ng = f->ng; if (ng == 0) return;
  const int *lo = f->lo, *hi = f->hi;
  /* transverse ranges: x faces skip y/z ghosts where those sides are physical (254-276);
   * y faces use full x, restricted z; z faces use full x,y */
  int glo[3][3], ghi[3][3];   /* [face dir][transverse dir] */
  const int gd[3] = { ng, ng, f->gz };                 /* ghost width per direction (2-D fabs: none along z) */
  for (int d = 0; d < 3; d++) for (int t = 0; t < 3; t++) {
    int restricted = (t > d);                          /* x: y,z restricted; y: z restricted */
    int nlo = gd[t], nhi = gd[t];
    if (restricted) { nlo = (bc[t][0] == VDN_INTERIOR) ? gd[t] : 0; nhi = (bc[t][1] == VDN_INTERIOR) ? gd[t] : 0; }
    glo[d][t] = lo[t] - nlo; ghi[d][t] = hi[t] + nhi;
  }
  for (int d = 0; d < f->dm; d++) for (int s = 0; s < 2; s++) {
    int b = bc[d][s];
    if (b == VDN_INTERIOR) continue;
    int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    if (t1 > t2) { int t = t1; t1 = t2; t2 = t; }
    int r1lo, r1hi, r2lo, r2hi;
    if (b == VDN_EXT_DIR) { r1lo = lo[t1] - gd[t1]; r1hi = hi[t1] + gd[t1]; r2lo = lo[t2] - gd[t2]; r2hi = hi[t2] + gd[t2]; }
    else { r1lo = glo[d][t1]; r1hi = ghi[d][t1]; r2lo = glo[d][t2]; r2hi = ghi[d][t2]; }
    int has = 0; double ev = 0.0;
    if (b == VDN_EXT_DIR) { ev = extdir_value(prm, icomp1, d, s, &has, f->dm); if (!has) continue; }
    else if (b != VDN_FOEXTRAP && b != VDN_HOEXTRAP && b != VDN_REFLECT_EVEN && b != VDN_REFLECT_ODD) {
      fprintf(stderr, "vo_physbc: bc(%d,%d) = %d NOT YET SUPPORTED\n", d + 1, s + 1, b); abort();
    }
    int edge = s == 0 ? lo[d] : hi[d];     /* first interior cell */
    int in = s == 0 ? 1 : -1;              /* inward direction */
    for (int b2 = r2lo; b2 <= r2hi; b2++) for (int b1 = r1lo; b1 <= r1hi; b1++) {
      int q[3]; q[t1] = b1; q[t2] = b2;
      double v = 0.0;
      if (b == VDN_FOEXTRAP) { q[d] = edge; v = VF(f, q[0], q[1], q[2], scomp); }
      else if (b == VDN_HOEXTRAP) {
        double s0, s1, s2;
        q[d] = edge;          s0 = VF(f, q[0], q[1], q[2], scomp);
        q[d] = edge + in;     s1 = VF(f, q[0], q[1], q[2], scomp);
        q[d] = edge + 2 * in; s2 = VF(f, q[0], q[1], q[2], scomp);
        v = (15.0 * s0 - 10.0 * s1 + 3.0 * s2) * 0.125;
      } else if (b == VDN_EXT_DIR) v = ev;
      for (int g = 1; g <= ng; g++) {
        if (b == VDN_REFLECT_EVEN || b == VDN_REFLECT_ODD) {
          q[d] = edge + in * (g - 1);
          v = VF(f, q[0], q[1], q[2], scomp); if (b == VDN_REFLECT_ODD) v = -v;
        }
        q[d] = edge - in * g;
        VF(f, q[0], q[1], q[2], scomp) = v;
      }
    }
  }
}

void vo_physbc(vo_fab *f, int scomp, int bccomp, int nc, const vo_bc *bc, const vdn_params *prm)
{
  for (int c = 0; c < nc; c++) {
    int b[3][2];
    for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) b[d][s] = bc->adv[d][s][bccomp + c];
    physbc_one(f, scomp + c, b, bccomp + c + 1, prm);
  }
}

/* ------------------------------------------------------------------------------------------
 * mkforce.f90:144-236  mkvelforce_3d (lapu may be NULL == visc_coef*visc_fac*0)
 * ---------------------------------------------------------------------------------------- */
void vo_mkvelforce(vo_fab *vf, const vo_fab *ext, const vo_fab *gp, const vo_fab *s,
                   const vo_fab *lapu, double visc_fac, const vdn_params *prm)
{
  const int *lo = vf->lo, *hi = vf->hi;
  memset(vf->p, 0, sizeof(double) * vo_size(vf));           /* setval(vel_force,ZERO,all) mkforce.f90:52 */
  /* a cell (i,j,k) of the valid box, or of one of the six face halos; lapu taken from the
   * nearest valid cell (0th order extrapolation, mkforce.f90:186-234) */
  for (int k = lo[2] - 1; k <= hi[2] + 1; k++)
  for (int j = lo[1] - 1; j <= hi[1] + 1; j++)
  for (int i = lo[0] - 1; i <= hi[0] + 1; i++) {
    int out = (i < lo[0]) + (i > hi[0]) + (j < lo[1]) + (j > hi[1]) + (k < lo[2]) + (k > hi[2]);
    if (out > 1) continue;                                   /* faces only, no edges/corners */
    int ic = i < lo[0] ? lo[0] : (i > hi[0] ? hi[0] : i);
    int jc = j < lo[1] ? lo[1] : (j > hi[1] ? hi[1] : j);
    int kc = k < lo[2] ? lo[2] : (k > hi[2] ? hi[2] : k);
    for (int m = 0; m < 3; m++) {
      double l = lapu ? VF(lapu, ic, jc, kc, m) : 0.0;
      double lapu_local = prm->visc_coef * visc_fac * l;
      double e = VF(ext, i, j, k, m);
      if (out == 0 && prm->boussinesq == 1) e = VF(s, i, j, k, 1) * e;   /* valid cells only, 162-170 */
      VF(vf, i, j, k, m) = e + (lapu_local - VF(gp, i, j, k, m)) / VF(s, i, j, k, 0);
    }
  }
}

/* mkforce.f90:333-402  mkscalforce_3d: comps 2..nscal only */
void vo_mkscalforce(vo_fab *sf, const vo_fab *ext, const vo_fab *laps, double diff_fac, const vdn_params *prm)
{
  const int *lo = sf->lo, *hi = sf->hi;
  memset(sf->p, 0, sizeof(double) * vo_size(sf));
  for (int k = lo[2] - 1; k <= hi[2] + 1; k++)
  for (int j = lo[1] - 1; j <= hi[1] + 1; j++)
  for (int i = lo[0] - 1; i <= hi[0] + 1; i++) {
    int out = (i < lo[0]) + (i > hi[0]) + (j < lo[1]) + (j > hi[1]) + (k < lo[2]) + (k > hi[2]);
    if (out > 1) continue;
    int ic = i < lo[0] ? lo[0] : (i > hi[0] ? hi[0] : i);
    int jc = j < lo[1] ? lo[1] : (j > hi[1] ? hi[1] : j);
    int kc = k < lo[2] ? lo[2] : (k > hi[2] ? hi[2] : k);
    for (int m = 1; m < prm->nscal; m++) {
      double l = laps ? VF(laps, ic, jc, kc, m) : 0.0;
      double laps_local = prm->diff_coef * diff_fac * l;
      VF(sf, i, j, k, m) = VF(ext, i, j, k, m) + laps_local;
    }
  }
}

/* update.f90:186-278 */
void vo_update(const vo_fab *sold, vo_fab *umac[3], vo_fab *sedge[3], vo_fab *flux[3],
               const vo_fab *force, vo_fab *snew, const double dx[3], double dt, int is_vel,
               const int *is_cons)
{
  const int *lo = sold->lo, *hi = sold->hi;
  const vo_fab *um = umac[0], *vm = umac[1], *wm = umac[2];
  const vo_fab *sx = sedge[0], *sy = sedge[1], *sz = sedge[2];
  for (int comp = 0; comp < sold->nc; comp++) {
    int cons = (!is_vel) && is_cons[comp];
    #pragma omp parallel for
    for (int k = lo[2]; k <= hi[2]; k++)
    for (int j = lo[1]; j <= hi[1]; j++)
    for (int i = lo[0]; i <= hi[0]; i++) {
      if (cons) {
        double divsu = (VF(flux[0], i + 1, j, k, comp) - VF(flux[0], i, j, k, comp)) / dx[0]
                     + (VF(flux[1], i, j + 1, k, comp) - VF(flux[1], i, j, k, comp)) / dx[1]
                     + (VF(flux[2], i, j, k + 1, comp) - VF(flux[2], i, j, k, comp)) / dx[2];
        VF(snew, i, j, k, comp) = VF(sold, i, j, k, comp) - dt * divsu + dt * VF(force, i, j, k, comp);
      } else {
        double ubar = 0.5 * (VF(um, i, j, k, 0) + VF(um, i + 1, j, k, 0));
        double vbar = 0.5 * (VF(vm, i, j, k, 0) + VF(vm, i, j + 1, k, 0));
        double wbar = 0.5 * (VF(wm, i, j, k, 0) + VF(wm, i, j, k + 1, 0));
        double ugrads = ubar * (VF(sx, i + 1, j, k, comp) - VF(sx, i, j, k, comp)) / dx[0]
                      + vbar * (VF(sy, i, j + 1, k, comp) - VF(sy, i, j, k, comp)) / dx[1]
                      + wbar * (VF(sz, i, j, k + 1, comp) - VF(sz, i, j, k, comp)) / dx[2];
        VF(snew, i, j, k, comp) = VF(sold, i, j, k, comp) - dt * ugrads + dt * VF(force, i, j, k, comp);
      }
    }
  }
}

/* make_at_halftime.f90:95-115 */
void vo_make_at_halftime(vo_fab *rh, int oc, const vo_fab *so, const vo_fab *sn, int ic)
{
  const int *lo = rh->lo, *hi = rh->hi;
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
    VF(rh, i, j, k, oc) = 0.5 * (VF(so, i, j, k, ic) + VF(sn, i, j, k, ic));
}

/* estdt.f90:131-181 then 69-78 (single box, single rank) */
double vo_estdt(const vo_fab *vel, const vo_fab *s, const vo_fab *gp, const vo_fab *ext, const double dx[3],
                double dtold, const vdn_params *prm)
{
  const int *lo = vel->lo, *hi = vel->hi;
  double eps = (double)1.0e-8f;                     /* single-precision literal, estdt.f90:146 */
  double u = 0, v = 0, w = 0, fx = 0, fy = 0, fz = 0, dt = 1.e20;
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    u = fmax(u, fabs(VF(vel, i, j, k, 0)));
    v = fmax(v, fabs(VF(vel, i, j, k, 1)));
    w = fmax(w, fabs(VF(vel, i, j, k, 2)));
    double r = VF(s, i, j, k, 0);
    fx = fmax(fx, fabs(VF(gp, i, j, k, 0) / r - VF(ext, i, j, k, 0)));
    fy = fmax(fy, fabs(VF(gp, i, j, k, 1) / r - VF(ext, i, j, k, 1)));
    fz = fmax(fz, fabs(VF(gp, i, j, k, 2) / r - VF(ext, i, j, k, 2)));
  }
  if (u > eps) dt = fmin(dt, dx[0] / u);
  if (v > eps) dt = fmin(dt, dx[1] / v);
  if (w > eps) dt = fmin(dt, dx[2] / w);
  if (fx > eps) dt = fmin(dt, sqrt(2.0 * dx[0] / fx));
  if (fy > eps) dt = fmin(dt, sqrt(2.0 * dx[1] / fy));
  if (fz > eps) dt = fmin(dt, sqrt(2.0 * dx[2] / fz));
  if (dt == 1.e20) { dt = fmin(dx[0], dx[1]); dt = fmin(dt, dx[2]); }
  dt = dt * prm->cflfac;
  if (dtold > 0.0) dt = fmin(dt, prm->max_dt_growth * dtold);
  return dt;
}

/* initdata.f90:201-259: prob_type 1 (bubble, u=0) and 2 (advected blob, u=(1,0,0)) */
void vo_initdata(vo_fab *u, vo_fab *s, const double dx[3], int prob_type)
{
  const int *lo = u->lo, *hi = u->hi;
  const double xblob = 0.5, yblob = 0.5, zblob = 0.5, densfact = 10.0, blobrad = 0.1;
  memset(u->p, 0, sizeof(double) * vo_size(u));
  for (long n = 0; n < s->sc; n++) { s->p[n] = 1.0; s->p[s->sc + n] = 0.0; }
  if (prob_type == 2) for (long n = 0; n < u->sc; n++) u->p[n] = 1.0;
  for (int k = lo[2]; k <= hi[2]; k++) {
    double z = dx[2] * (k + 0.5);
    for (int j = lo[1]; j <= hi[1]; j++) {
      double y = dx[1] * (j + 0.5);
      for (int i = lo[0]; i <= hi[0]; i++) {
        double x = dx[0] * (i + 0.5);
        double dist = sqrt((x - xblob) * (x - xblob) + (y - yblob) * (y - yblob) + (z - zblob) * (z - zblob));
        double r = 1.0 + 0.5 * (densfact - 1.0) * (1.0 - tanh(30. * (dist - blobrad)));
        VF(s, i, j, k, 0) = r; VF(s, i, j, k, 1) = r;
      }
    }
  }
}
