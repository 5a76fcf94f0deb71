/* oracle/vo_2d.c -- the dm = 2 path of advance_timestep (BASELINE.json configs[0]: the reference's own
 * CPU-runnable case).  TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * Restated reference routines (one box, one level):
 *   velpred_2d            src/velpred.f90:125-524      (NB: all four CFL factors divide INSIDE max()/min())
 *   mkflux_2d             src/mkflux.f90:152-691
 *   update_2d             src/update.f90:113-184
 *   mkvelforce_2d         src/mkforce.f90:82-142 ;  mkscalforce_2d  src/mkforce.f90:290-331
 *   estdt_2d              src/estdt.f90:89-129
 *   divumac_2d / mk_mac_coeffs_2d / mkumac_2d   src/macproject.f90:226-248, 338-359, 538-576
 *   create_uvec_2d / mkgphi_2d / hg_update_2d   src/hgproject.f90:374-432, 517-541, 581-636
 *   initdata_2d           src/initdata.f90:127-194
 * A 2-D fab is one z-plane (vo_fab_init2d): p(lo1-ng:hi1+ng, lo2-ng:hi2+ng, nc), the BoxLib 2-D layout.
 * The elliptic solves reuse the multigrids of vo_macproject.c / vo_hgproject.c in their dm = 2 mode (5-point and
 * 9-point Q1 operators, no coarsening along z).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include "vo.h"

static inline double sgnpos(double a, double b, int c) { return c ? a : b; }       /* Fortran merge(a,b,c) */

static void fab2(vo_fab *f, const int *lo, const int *hi, int ng, int face_dir, int nc, double val)
{
  int nd[3] = { 0, 0, 0 }; if (face_dir >= 0) nd[face_dir] = 1;
  vo_fab_init2d(f, NULL, lo, hi, ng, nd, nc);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}
static void fab2_nodal(vo_fab *f, const int *lo, const int *hi, int ng, double val)
{
  int nd[3] = { 1, 1, 0 };
  vo_fab_init2d(f, NULL, lo, hi, ng, nd, 1);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}

/* boundary rule of a (left,right) pair on a domain face -- the same as oracle/vo_godunov.c:bc_pair (velpred.f90:270-307,
 * 349-383; mkflux.f90:311-371, 403-459); no 2-D quirk: hi-x OUTLET uses max (velpred.f90:305) */
static inline void bc_pair2(double *L, double *R, int phys, int side, int is_vel, int normal, double ghost)
{
  if (phys == VDN_INLET) { *L = ghost; *R = ghost; }
  else if (phys == VDN_SLIP_WALL) {
    if (is_vel && normal) { *L = 0.0; *R = 0.0; }
    else if (side == 0) *L = *R; else *R = *L;
  } else if (phys == VDN_NO_SLIP_WALL) {
    if (is_vel) { *L = 0.0; *R = 0.0; }
    else if (side == 0) *L = *R; else *R = *L;
  } else if (phys == VDN_OUTLET) {
    if (is_vel && normal) {
      double v = (side == 0) ? fmin(*R, 0.0) : fmax(*L, 0.0);
      *L = v; *R = v;
    } else if (side == 0) *L = *R; else *R = *L;
  }
}
static inline int fside(int f, int lo, int hi) { return f == lo ? 0 : (f == hi + 1 ? 1 : -1); }

/* ==========================================================================================
 * velpred_2d (velpred.f90:125-524), full-array form of the rolling kernel
 * ======================================================================================== */
void vo2_velpred(const vo_fab *u, vo_fab *umac[2], const vo_fab *force, const double dx[2], double dt, const vo_bc *bc, const vdn_params *prm)
{
  const int *lo = u->lo, *hi = u->hi;
  const int is = lo[0], ie = hi[0], js = lo[1], je = hi[1];
  const double dt2 = 0.5 * dt, dt4 = dt / 4.0, hx = dx[0], hy = dx[1];
  const int use_minion = prm->use_minion;
  vo_fab slx, sly;
  fab2(&slx, lo, hi, 1, -1, 2, 0.0); fab2(&sly, lo, hi, 1, -1, 2, 0.0);
  vo_slope(u, &slx, 0, 2, 0, bc, prm->slope_order);          /* velpred.f90:168-169 */
  vo_slope(u, &sly, 1, 2, 0, bc, prm->slope_order);
  /* eps (velpred.f90:216-227) */
  double umax = fabs(V2(u, is, js, 0));
  for (int j = js; j <= je; j++) for (int i = is; i <= ie; i++) { umax = fmax(umax, fabs(V2(u, i, j, 0))); umax = fmax(umax, fabs(V2(u, i, j, 1))); }
  const double eps = (umax == 0.0) ? 1.0e-8 : 1.0e-8 * umax;
  /* uimhx on x-faces is..ie+1, rows js-1..je+1;  uimhy on y-faces js..je+1, columns is-1..ie+1; both 2 comps.
   * ulx/urx (comp 1) and uly/ury (comp 2) are kept for the final states */
  vo_fab uimhx, uimhy, ulx, urx, uly, ury;
  fab2(&uimhx, lo, hi, 1, 0, 2, NAN); fab2(&uimhy, lo, hi, 1, 1, 2, NAN);
  fab2(&ulx, lo, hi, 1, 0, 1, NAN); fab2(&urx, lo, hi, 1, 0, 1, NAN); fab2(&uly, lo, hi, 1, 1, 1, NAN); fab2(&ury, lo, hi, 1, 1, 1, NAN);
  for (int j = js - 1; j <= je + 1; j++) for (int i = is; i <= ie + 1; i++) {        /* 1. velpred.f90:254-322 */
    double L[2], R[2];
    for (int c = 0; c < 2; c++) {
      L[c] = V2(u, i - 1, j, c) + (0.5 - dt2 * fmax(0.0, V2(u, i - 1, j, 0) / hx)) * V2(&slx, i - 1, j, c);
      R[c] = V2(u, i, j, c) - (0.5 + dt2 * fmin(0.0, V2(u, i, j, 0) / hx)) * V2(&slx, i, j, c);
      if (use_minion) { L[c] = L[c] + dt2 * V2(force, i - 1, j, c); R[c] = R[c] + dt2 * V2(force, i, j, c); }
    }
    int side = fside(i, is, ie);
    if (side >= 0) for (int c = 0; c < 2; c++) bc_pair2(&L[c], &R[c], bc->phys[0][side], side, 1, c == 0, side == 0 ? V2(u, is - 1, j, c) : V2(u, ie + 1, j, c));
    V2(&ulx, i, j, 0) = L[0]; V2(&urx, i, j, 0) = R[0];
    double uavg = 0.5 * (L[0] + R[0]);
    int test = ((L[0] <= 0.0 && R[0] >= 0.0) || (fabs(L[0] + R[0]) < eps));
    double un = sgnpos(L[0], R[0], uavg > 0.0); un = sgnpos(0.0, un, test);
    V2(&uimhx, i, j, 0) = un;
    double ut = sgnpos(L[1], R[1], un > 0.0); uavg = 0.5 * (L[1] + R[1]);
    V2(&uimhx, i, j, 1) = sgnpos(uavg, ut, fabs(un) < eps);
  }
  for (int j = js; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {        /* 2. velpred.f90:330-396 */
    double L[2], R[2];
    for (int c = 0; c < 2; c++) {
      L[c] = V2(u, i, j - 1, c) + (0.5 - dt2 * fmax(0.0, V2(u, i, j - 1, 1) / hy)) * V2(&sly, i, j - 1, c);
      R[c] = V2(u, i, j, c) - (0.5 + dt2 * fmin(0.0, V2(u, i, j, 1) / hy)) * V2(&sly, i, j, c);
      if (use_minion) { L[c] = L[c] + dt2 * V2(force, i, j - 1, c); R[c] = R[c] + dt2 * V2(force, i, j, c); }
    }
    int side = fside(j, js, je);
    if (side >= 0) for (int c = 0; c < 2; c++) bc_pair2(&L[c], &R[c], bc->phys[1][side], side, 1, c == 1, side == 0 ? V2(u, i, js - 1, c) : V2(u, i, je + 1, c));
    V2(&uly, i, j, 0) = L[1]; V2(&ury, i, j, 0) = R[1];
    double uavg = 0.5 * (L[1] + R[1]);
    int test = ((L[1] <= 0.0 && R[1] >= 0.0) || (fabs(L[1] + R[1]) < eps));
    double un = sgnpos(L[1], R[1], uavg > 0.0); un = sgnpos(0.0, un, test);
    V2(&uimhy, i, j, 1) = un;
    double ut = sgnpos(L[0], R[0], un > 0.0); uavg = 0.5 * (L[0] + R[0]);
    V2(&uimhy, i, j, 0) = sgnpos(uavg, ut, fabs(un) < eps);
  }
  for (int j = js; j <= je + 1; j++) for (int i = is; i <= ie; i++) {                /* 3. vmac, velpred.f90:402-443 */
    double l = V2(&uly, i, j, 0) - (dt4 / hx) * (V2(&uimhx, i + 1, j - 1, 0) + V2(&uimhx, i, j - 1, 0)) * (V2(&uimhx, i + 1, j - 1, 1) - V2(&uimhx, i, j - 1, 1));
    double r = V2(&ury, i, j, 0) - (dt4 / hx) * (V2(&uimhx, i + 1, j, 0) + V2(&uimhx, i, j, 0)) * (V2(&uimhx, i + 1, j, 1) - V2(&uimhx, i, j, 1));
    if (!use_minion) { l = l + dt2 * V2(force, i, j - 1, 1); r = r + dt2 * V2(force, i, j, 1); }
    double uavg = 0.5 * (l + r);
    int test = ((l <= 0.0 && r >= 0.0) || (fabs(l + r) < eps));
    double v = sgnpos(l, r, uavg > 0.0); v = sgnpos(0.0, v, test);
    int side = fside(j, js, je);
    if (side >= 0) {
      int ph = bc->phys[1][side];
      if (ph == VDN_SLIP_WALL || ph == VDN_NO_SLIP_WALL) v = 0.0;
      else if (ph == VDN_INLET) v = side == 0 ? V2(u, i, js - 1, 1) : V2(u, i, je + 1, 1);
      else if (ph == VDN_OUTLET) v = side == 0 ? fmin(r, 0.0) : fmax(l, 0.0);
    }
    V2(umac[1], i, j, 0) = v;
  }
  for (int j = js; j <= je; j++) for (int i = is; i <= ie + 1; i++) {                /* 4. umac, velpred.f90:455-496 */
    double l = V2(&ulx, i, j, 0) - (dt4 / hy) * (V2(&uimhy, i - 1, j + 1, 1) + V2(&uimhy, i - 1, j, 1)) * (V2(&uimhy, i - 1, j + 1, 0) - V2(&uimhy, i - 1, j, 0));
    double r = V2(&urx, i, j, 0) - (dt4 / hy) * (V2(&uimhy, i, j + 1, 1) + V2(&uimhy, i, j, 1)) * (V2(&uimhy, i, j + 1, 0) - V2(&uimhy, i, j, 0));
    if (!use_minion) { l = l + dt2 * V2(force, i - 1, j, 0); r = r + dt2 * V2(force, i, j, 0); }
    double uavg = 0.5 * (l + r);
    int test = ((l <= 0.0 && r >= 0.0) || (fabs(l + r) < eps));
    double v = sgnpos(l, r, uavg > 0.0); v = sgnpos(0.0, v, test);
    int side = fside(i, is, ie);
    if (side >= 0) {
      int ph = bc->phys[0][side];
      if (ph == VDN_SLIP_WALL || ph == VDN_NO_SLIP_WALL) v = 0.0;
      else if (ph == VDN_INLET) v = side == 0 ? V2(u, is - 1, j, 0) : V2(u, ie + 1, j, 0);
      else if (ph == VDN_OUTLET) v = side == 0 ? fmin(r, 0.0) : fmax(l, 0.0);
    }
    V2(umac[0], i, j, 0) = v;
  }
  free(slx.p); free(sly.p); free(uimhx.p); free(uimhy.p); free(ulx.p); free(urx.p); free(uly.p); free(ury.p);
}

/* ==========================================================================================
 * mkflux_2d (mkflux.f90:152-691), full-array form of the rolling kernel
 * ======================================================================================== */
void vo2_mkflux(const vo_fab *s, vo_fab *sedge[2], vo_fab *flux[2], vo_fab *umac[2], const vo_fab *force, const vo_fab *mac_rhs,
                const double dx[2], double dt, int is_vel, const int *is_cons, int bccomp, const vo_bc *bc, const vdn_params *prm)
{
  const int *lo = s->lo, *hi = s->hi;
  const int is = lo[0], ie = hi[0], js = lo[1], je = hi[1], ncomp = s->nc;
  const double dt2 = 0.5 * dt, dt4 = dt / 4.0, hx = dx[0], hy = dx[1];
  const int use_minion = prm->use_minion;
  const vo_fab *um = umac[0], *vm = umac[1];
  vo_fab slpx, slpy;
  fab2(&slpx, lo, hi, 1, -1, ncomp, 0.0); fab2(&slpy, lo, hi, 1, -1, ncomp, 0.0);
  vo_slope(s, &slpx, 0, ncomp, bccomp, bc, prm->slope_order);         /* mkflux.f90:201-205 */
  vo_slope(s, &slpy, 1, ncomp, bccomp, bc, prm->slope_order);
  double umax = fabs(V2(um, is, js, 0));                              /* mkflux.f90:248-264 */
  for (int j = js; j <= je; j++) for (int i = is; i <= ie + 1; i++) umax = fmax(umax, fabs(V2(um, i, j, 0)));
  for (int j = js; j <= je + 1; j++) for (int i = is; i <= ie; i++) umax = fmax(umax, fabs(V2(vm, i, j, 0)));
  const double eps = (umax == 0.0) ? 1.0e-8 : 1.0e-8 * umax;
  vo_fab slx, srx, sly, sry, simhx, simhy;
  fab2(&slx, lo, hi, 1, 0, 1, NAN); fab2(&srx, lo, hi, 1, 0, 1, NAN); fab2(&sly, lo, hi, 1, 1, 1, NAN); fab2(&sry, lo, hi, 1, 1, 1, NAN);
  fab2(&simhx, lo, hi, 1, 0, 1, NAN); fab2(&simhy, lo, hi, 1, 1, 1, NAN);
  for (int c = 0; c < ncomp; c++) {
    const int cons = is_cons[c];
    for (int j = js - 1; j <= je + 1; j++) for (int i = is; i <= ie + 1; i++) {      /* 1. mkflux.f90:296-377 */
      double L = V2(s, i - 1, j, c) + (0.5 - dt2 * V2(um, i, j, 0) / hx) * V2(&slpx, i - 1, j, c);
      double R = V2(s, i, j, c) - (0.5 + dt2 * V2(um, i, j, 0) / hx) * V2(&slpx, i, j, c);
      if (use_minion) { L = L + dt2 * V2(force, i - 1, j, c); R = R + dt2 * V2(force, i, j, c); }
      if (use_minion && cons) { L = L - dt2 * V2(s, i - 1, j, c) * V2(mac_rhs, i - 1, j, 0); R = R - dt2 * V2(s, i, j, c) * V2(mac_rhs, i, j, 0); }
      int side = fside(i, is, ie);
      if (side >= 0) bc_pair2(&L, &R, bc->phys[0][side], side, is_vel, c == 0, side == 0 ? V2(s, is - 1, j, c) : V2(s, ie + 1, j, c));
      V2(&slx, i, j, 0) = L; V2(&srx, i, j, 0) = R;
      double v = sgnpos(L, R, V2(um, i, j, 0) > 0.0), savg = 0.5 * (L + R);
      V2(&simhx, i, j, 0) = sgnpos(v, savg, fabs(V2(um, i, j, 0)) > eps);
    }
    for (int j = js; j <= je + 1; j++) for (int i = is - 1; i <= ie + 1; i++) {      /* 2. mkflux.f90:385-466 */
      double L = V2(s, i, j - 1, c) + (0.5 - dt2 * V2(vm, i, j, 0) / hy) * V2(&slpy, i, j - 1, c);
      double R = V2(s, i, j, c) - (0.5 + dt2 * V2(vm, i, j, 0) / hy) * V2(&slpy, i, j, c);
      if (use_minion) { L = L + dt2 * V2(force, i, j - 1, c); R = R + dt2 * V2(force, i, j, c); }
      if (use_minion && cons) { L = L - dt2 * V2(s, i, j - 1, c) * V2(mac_rhs, i, j - 1, 0); R = R - dt2 * V2(s, i, j, c) * V2(mac_rhs, i, j, 0); }
      int side = fside(j, js, je);
      if (side >= 0) bc_pair2(&L, &R, bc->phys[1][side], side, is_vel, c == 1, side == 0 ? V2(s, i, js - 1, c) : V2(s, i, je + 1, c));
      V2(&sly, i, j, 0) = L; V2(&sry, i, j, 0) = R;
      double v = sgnpos(L, R, V2(vm, i, j, 0) > 0.0), savg = 0.5 * (L + R);
      V2(&simhy, i, j, 0) = sgnpos(v, savg, fabs(V2(vm, i, j, 0)) > eps);
    }
    for (int j = js; j <= je + 1; j++) for (int i = is; i <= ie; i++) {              /* 3. sedgey, mkflux.f90:472-558 */
      double l, r;
      if (cons) {
        l = V2(&sly, i, j, 0) - (dt2 / hx) * (V2(&simhx, i + 1, j - 1, 0) * V2(um, i + 1, j - 1, 0) - V2(&simhx, i, j - 1, 0) * V2(um, i, j - 1, 0))
                              + (dt2 / hx) * V2(s, i, j - 1, c) * (V2(um, i + 1, j - 1, 0) - V2(um, i, j - 1, 0));
        r = V2(&sry, i, j, 0) - (dt2 / hx) * (V2(&simhx, i + 1, j, 0) * V2(um, i + 1, j, 0) - V2(&simhx, i, j, 0) * V2(um, i, j, 0))
                              + (dt2 / hx) * V2(s, i, j, c) * (V2(um, i + 1, j, 0) - V2(um, i, j, 0));
      } else {
        l = V2(&sly, i, j, 0) - (dt4 / hx) * (V2(um, i + 1, j - 1, 0) + V2(um, i, j - 1, 0)) * (V2(&simhx, i + 1, j - 1, 0) - V2(&simhx, i, j - 1, 0));
        r = V2(&sry, i, j, 0) - (dt4 / hx) * (V2(um, i + 1, j, 0) + V2(um, i, j, 0)) * (V2(&simhx, i + 1, j, 0) - V2(&simhx, i, j, 0));
      }
      if (!use_minion) { l = l + dt2 * V2(force, i, j - 1, c); r = r + dt2 * V2(force, i, j, c); }
      if (!use_minion && cons) { l = l - dt2 * V2(s, i, j - 1, c) * V2(mac_rhs, i, j - 1, 0); r = r - dt2 * V2(s, i, j, c) * V2(mac_rhs, i, j, 0); }
      double e = sgnpos(l, r, V2(vm, i, j, 0) > 0.0), savg = 0.5 * (l + r);
      e = sgnpos(e, savg, fabs(V2(vm, i, j, 0)) > eps);
      int side = fside(j, js, je);
      if (side >= 0) {
        int ph = bc->phys[1][side]; double in = side == 0 ? r : l;
        if (ph == VDN_INLET) e = side == 0 ? V2(s, i, js - 1, c) : V2(s, i, je + 1, c);
        else if (ph == VDN_SLIP_WALL) e = (is_vel && c == 1) ? 0.0 : in;
        else if (ph == VDN_NO_SLIP_WALL) e = is_vel ? 0.0 : in;
        else if (ph == VDN_OUTLET) e = (is_vel && c == 1) ? (side == 0 ? fmin(in, 0.0) : fmax(in, 0.0)) : in;
      }
      V2(sedge[1], i, j, c) = e;
      if (cons) V2(flux[1], i, j, c) = e * V2(vm, i, j, 0);
    }
    for (int j = js; j <= je; j++) for (int i = is; i <= ie + 1; i++) {              /* 4. sedgex, mkflux.f90:570-660 */
      double l, r;
      if (cons) {
        l = V2(&slx, i, j, 0) - (dt2 / hy) * (V2(&simhy, i - 1, j + 1, 0) * V2(vm, i - 1, j + 1, 0) - V2(&simhy, i - 1, j, 0) * V2(vm, i - 1, j, 0))
                              + (dt2 / hy) * V2(s, i - 1, j, c) * (V2(vm, i - 1, j + 1, 0) - V2(vm, i - 1, j, 0));
        r = V2(&srx, i, j, 0) - (dt2 / hy) * (V2(&simhy, i, j + 1, 0) * V2(vm, i, j + 1, 0) - V2(&simhy, i, j, 0) * V2(vm, i, j, 0))
                              + (dt2 / hy) * V2(s, i, j, c) * (V2(vm, i, j + 1, 0) - V2(vm, i, j, 0));
      } else {
        l = V2(&slx, i, j, 0) - (dt4 / hy) * (V2(vm, i - 1, j + 1, 0) + V2(vm, i - 1, j, 0)) * (V2(&simhy, i - 1, j + 1, 0) - V2(&simhy, i - 1, j, 0));
        r = V2(&srx, i, j, 0) - (dt4 / hy) * (V2(vm, i, j + 1, 0) + V2(vm, i, j, 0)) * (V2(&simhy, i, j + 1, 0) - V2(&simhy, i, j, 0));
      }
      if (!use_minion) { l = l + dt2 * V2(force, i - 1, j, c); r = r + dt2 * V2(force, i, j, c); }
      if (!use_minion && cons) { l = l - dt2 * V2(s, i - 1, j, c) * V2(mac_rhs, i - 1, j, 0); r = r - dt2 * V2(s, i, j, c) * V2(mac_rhs, i, j, 0); }
      double e = sgnpos(l, r, V2(um, i, j, 0) > 0.0), savg = 0.5 * (l + r);
      e = sgnpos(e, savg, fabs(V2(um, i, j, 0)) > eps);
      int side = fside(i, is, ie);
      if (side >= 0) {
        int ph = bc->phys[0][side]; double in = side == 0 ? r : l;
        if (ph == VDN_INLET) e = side == 0 ? V2(s, is - 1, j, c) : V2(s, ie + 1, j, c);
        else if (ph == VDN_SLIP_WALL) e = (is_vel && c == 0) ? 0.0 : in;
        else if (ph == VDN_NO_SLIP_WALL) e = is_vel ? 0.0 : in;
        else if (ph == VDN_OUTLET) e = (is_vel && c == 0) ? (side == 0 ? fmin(in, 0.0) : fmax(in, 0.0)) : in;
      }
      V2(sedge[0], i, j, c) = e;
      if (cons) V2(flux[0], i, j, c) = e * V2(um, i, j, 0);
    }
  }
  free(slpx.p); free(slpy.p); free(slx.p); free(srx.p); free(sly.p); free(sry.p); free(simhx.p); free(simhy.p);
}

/* update_2d (update.f90:113-184) */
void vo2_update(const vo_fab *sold, vo_fab *umac[2], vo_fab *sedge[2], vo_fab *flux[2], const vo_fab *force, vo_fab *snew,
                const double dx[2], double dt, int is_vel, const int *is_cons)
{
  const int *lo = sold->lo, *hi = sold->hi;
  for (int c = 0; c < sold->nc; c++) {
    int cons = (!is_vel) && is_cons[c];
    for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
      if (cons) {
        double divsu = (V2(flux[0], i + 1, j, c) - V2(flux[0], i, j, c)) / dx[0] + (V2(flux[1], i, j + 1, c) - V2(flux[1], i, j, c)) / dx[1];
        V2(snew, i, j, c) = V2(sold, i, j, c) - dt * divsu + dt * V2(force, i, j, c);
      } else {
        double ubar = 0.5 * (V2(umac[0], i, j, 0) + V2(umac[0], i + 1, j, 0));
        double vbar = 0.5 * (V2(umac[1], i, j, 0) + V2(umac[1], i, j + 1, 0));
        double ug = ubar * (V2(sedge[0], i + 1, j, c) - V2(sedge[0], i, j, c)) / dx[0] + vbar * (V2(sedge[1], i, j + 1, c) - V2(sedge[1], i, j, c)) / dx[1];
        V2(snew, i, j, c) = V2(sold, i, j, c) - dt * ug + dt * V2(force, i, j, c);
      }
    }
  }
}

/* mkvelforce_2d (mkforce.f90:82-142): valid cells, then the four one-cell edge halos (no corners) */
void vo2_mkvelforce(vo_fab *vf, const vo_fab *ext, const vo_fab *gp, const vo_fab *s, const vo_fab *lapu, double visc_fac, const vdn_params *prm)
{
  const int *lo = vf->lo, *hi = vf->hi;
  memset(vf->p, 0, sizeof(double) * vo_size(vf));
  for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++) {
    int out = (i < lo[0]) + (i > hi[0]) + (j < lo[1]) + (j > hi[1]);
    if (out > 1) continue;
    int ic = i < lo[0] ? lo[0] : (i > hi[0] ? hi[0] : i), jc = j < lo[1] ? lo[1] : (j > hi[1] ? hi[1] : j);
    for (int m = 0; m < 2; m++) {
      double l = lapu ? V2(lapu, ic, jc, m) : 0.0;
      double lapu_local = prm->visc_coef * visc_fac * l;
      double e = V2(ext, i, j, m);
      if (out == 0 && prm->boussinesq == 1) e = V2(s, i, j, 1) * e;
      V2(vf, i, j, m) = e + (lapu_local - V2(gp, i, j, m)) / V2(s, i, j, 0);
    }
  }
}
/* mkscalforce_2d (mkforce.f90:290-331) */
void vo2_mkscalforce(vo_fab *sf, const vo_fab *ext, const vo_fab *laps, double diff_fac, const vdn_params *prm)
{
  const int *lo = sf->lo, *hi = sf->hi;
  memset(sf->p, 0, sizeof(double) * vo_size(sf));
  for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++) {
    int out = (i < lo[0]) + (i > hi[0]) + (j < lo[1]) + (j > hi[1]);
    if (out > 1) continue;
    int ic = i < lo[0] ? lo[0] : (i > hi[0] ? hi[0] : i), jc = j < lo[1] ? lo[1] : (j > hi[1] ? hi[1] : j);
    for (int m = 1; m < prm->nscal; m++) {
      double l = laps ? V2(laps, ic, jc, m) : 0.0;
      V2(sf, i, j, m) = V2(ext, i, j, m) + prm->diff_coef * diff_fac * l;
    }
  }
}

/* estdt_2d (estdt.f90:89-129) + estdt.f90:69-78 */
double vo2_estdt(const vo_fab *vel, const vo_fab *s, const vo_fab *gp, const vo_fab *ext, const double dx[2], double dtold, const vdn_params *prm)
{
  const int *lo = vel->lo, *hi = vel->hi;
  double eps = (double)1.0e-8f;
  double u = 0, v = 0, fx = 0, fy = 0, dt = 1.e20;
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    u = fmax(u, fabs(V2(vel, i, j, 0))); v = fmax(v, fabs(V2(vel, i, j, 1)));
    double r = V2(s, i, j, 0);
    fx = fmax(fx, fabs(V2(gp, i, j, 0) / r - V2(ext, i, j, 0)));
    fy = fmax(fy, fabs(V2(gp, i, j, 1) / r - V2(ext, i, j, 1)));
  }
  if (u > eps) dt = fmin(dt, dx[0] / u);
  if (v > eps) dt = fmin(dt, dx[1] / v);
  if (fx > eps) dt = fmin(dt, sqrt(2.0 * dx[0] / fx));
  if (fy > eps) dt = fmin(dt, sqrt(2.0 * dx[1] / fy));
  if (dt == 1.e20) dt = fmin(dx[0], dx[1]);
  dt = dt * prm->cflfac;
  if (dtold > 0.0) dt = fmin(dt, prm->max_dt_growth * dtold);
  return dt;
}

/* initdata_2d (initdata.f90:127-171): prob_type 1 (bubble, densfact = 2) and 2 (advected blob) */
void vo2_initdata(vo_fab *u, vo_fab *s, const double dx[2], int prob_type)
{
  const int *lo = u->lo, *hi = u->hi;
  const double xblob = 0.5, yblob = 0.5, densfact = 2.0, blobrad = 0.1;
  memset(u->p, 0, sizeof(double) * vo_size(u));
  for (long n = 0; n < s->sc; n++) { s->p[n] = 1.0; s->p[s->sc + n] = 0.0; }
  if (prob_type == 2) for (long n = 0; n < u->sc; n++) u->p[n] = 1.0;
  for (int j = lo[1]; j <= hi[1]; j++) {
    double y = dx[1] * (j + 0.5);
    for (int i = lo[0]; i <= hi[0]; i++) {
      double x = dx[0] * (i + 0.5);
      double dist = sqrt((x - xblob) * (x - xblob) + (y - yblob) * (y - yblob));
      double r = 1.0 + 0.5 * (densfact - 1.0) * (1.0 - tanh(30. * (dist - blobrad)));
      V2(s, i, j, 0) = r; V2(s, i, j, 1) = r;
    }
  }
}

/* ==========================================================================================
 * MAC projection, dm = 2 (macproject.f90:20-133 with the 2-D kernels)
 * ======================================================================================== */
void vo2_macproject(vo_fab *umac[2], vo_fab *rho, const vo_fab *mac_rhs, const double dx[2], const vo_bc *bc, const int pmask[3],
                    const vdn_params *prm, vo_mgstat *st)
{
  const int *lo = rho->lo, *hi = rho->hi;
  vo_fab rh, phi, beta[2], *bp[3] = { &beta[0], &beta[1], NULL };
  fab2(&rh, lo, hi, 0, -1, 1, 0.0); fab2(&phi, lo, hi, 1, -1, 1, 0.0);
  fab2(&beta[0], lo, hi, 0, 0, 1, 0.0); fab2(&beta[1], lo, hi, 0, 1, 1, 0.0);
  int ellbc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ellbc[d][s] = d < 2 ? bc->ell[d][s][bc->press_comp] : VDN_BC_INT;
  double dxinv[2] = { 1.0 / dx[0], 1.0 / dx[1] };
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {        /* divumac_2d, then rh = mac_rhs - rh (190-196) */
    double d = (V2(umac[0], i + 1, j, 0) - V2(umac[0], i, j, 0)) * dxinv[0] + (V2(umac[1], i, j + 1, 0) - V2(umac[1], i, j, 0)) * dxinv[1];
    V2(&rh, i, j, 0) = d * -1.0 + V2(mac_rhs, i, j, 0);
  }
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0] + 1; i++) V2(&beta[0], i, j, 0) = 2.0 / (V2(rho, i, j, 0) + V2(rho, i - 1, j, 0));
  for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(&beta[1], i, j, 0) = 2.0 / (V2(rho, i, j, 0) + V2(rho, i, j - 1, 0));
  double dx3[3] = { dx[0], dx[1], 1.0 };
  vo_cc_solve_ab(&rh, &phi, NULL, bp, dx3, ellbc, prm->mac_rel_eps, -1.0, prm->mg_max_iter, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, 0, st);
  /* mkumac_2d with the solver's ghost closure (see vo_mkumac) */
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0] + 1; i++) {
    int side = fside(i, lo[0], hi[0]);
    if (side >= 0 && ellbc[0][side] == VDN_BC_NEU) continue;
    double g = (V2(&phi, i, j, 0) - V2(&phi, i - 1, j, 0)) / dx[0];
    V2(umac[0], i, j, 0) = V2(umac[0], i, j, 0) - V2(&beta[0], i, j, 0) * g;
  }
  for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    int side = fside(j, lo[1], hi[1]);
    if (side >= 0 && ellbc[1][side] == VDN_BC_NEU) continue;
    double g = (V2(&phi, i, j, 0) - V2(&phi, i, j - 1, 0)) / dx[1];
    V2(umac[1], i, j, 0) = V2(umac[1], i, j, 0) - V2(&beta[1], i, j, 0) * g;
  }
  vo_fill_boundary(umac[0], pmask); vo_fill_boundary(umac[1], pmask);
  free(rh.p); free(phi.p); free(beta[0].p); free(beta[1].p);
}

/* explicit diffusive term, dm = 2 (see vo_viscous.c) */
void vo2_explicit_diffusive_term(vo_fab *lap, const vo_fab *data, int comp, int bccomp, const double dx[2], const vo_bc *bc)
{
  const int *lo = data->lo, *hi = data->hi;
  double hi2[2] = { 1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]) };
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    const int q[2] = { i, j };
    const double p0 = V2(data, i, j, comp);
    double sum = 0.0;
    for (int d = 0; d < 2; d++) {
      int m[2] = { i, j }, p[2] = { i, j }; m[d] -= 1; p[d] += 1;
      double fm = p0 - V2(data, m[0], m[1], comp), fp = V2(data, p[0], p[1], comp) - p0;
      if (q[d] == lo[d]) { int e = bc->ell[d][0][bccomp]; if (e == VDN_BC_NEU) fm = 0.0; else if (e == VDN_BC_DIR) fm = 2.0 * fm; }
      if (q[d] == hi[d]) { int e = bc->ell[d][1][bccomp]; if (e == VDN_BC_NEU) fp = 0.0; else if (e == VDN_BC_DIR) fp = 2.0 * fp; }
      sum = sum + (fp - fm) * hi2[d];
    }
    V2(lap, i, j, comp) = sum;
  }
}
/* viscsolve.f90:19-306 / 308-515 with the 2-D right-hand sides (mkrhs_2d, viscsolve.f90:226-262) */
static void visc_solve2(vo_fab *unew, const vo_fab *lapu, const vo_fab *rho, const vo_fab *mac_rhs, const double dx[2], double mu,
                        const vo_bc *bc, const int pmask[3], const vdn_params *prm)
{
  const int *lo = unew->lo, *hi = unew->hi;
  vo_fab rh, phi, alpha, beta[2], *bp[3] = { &beta[0], &beta[1], NULL };
  fab2(&rh, lo, hi, 0, -1, 1, 0.0); fab2(&phi, lo, hi, 1, -1, 1, 0.0); fab2(&alpha, lo, hi, 0, -1, 1, 0.0);
  fab2(&beta[0], lo, hi, 0, 0, 1, mu); fab2(&beta[1], lo, hi, 0, 1, 1, mu);
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(&alpha, i, j, 0) = V2(rho, i, j, 0);
  const double third = 1.0 / 3.0;
  double dx3[3] = { dx[0], dx[1], 1.0 };
  for (int d = 0; d < 2; d++) {
    double visc_mu_dt = (prm->diffusion_type == 1) ? 2.0 * mu : mu;
    for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++) V2(&phi, i, j, 0) = V2(unew, i, j, d);
    for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
      double r = V2(unew, i, j, d) * V2(rho, i, j, 0);
      if (prm->diffusion_type == 1) r = r + mu * V2(lapu, i, j, d);
      int p[2] = { i, j }, m[2] = { i, j }; p[d] += 1; m[d] -= 1;
      r = r + third * visc_mu_dt * (V2(mac_rhs, p[0], p[1], 0) - V2(mac_rhs, m[0], m[1], 0)) / dx[d];
      V2(&rh, i, j, 0) = r;
    }
    int ellbc[3][2]; vo_mgstat st;
    for (int a = 0; a < 3; a++) for (int s = 0; s < 2; s++) ellbc[a][s] = a < 2 ? bc->ell[a][s][d] : VDN_BC_INT;
    vo_cc_solve_ab(&rh, &phi, &alpha, bp, dx3, ellbc, 1.e-12, -1.0, prm->mg_max_iter, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, 0, &st);
    for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(unew, i, j, d) = V2(&phi, i, j, 0);
  }
  vo_fill_boundary(unew, pmask);
  vo_physbc(unew, 0, 0, 2, bc, prm);
  free(rh.p); free(phi.p); free(alpha.p); free(beta[0].p); free(beta[1].p);
}
static void diff_scalar_solve2(vo_fab *snew, const vo_fab *laps, const double dx[2], double mu, const vo_bc *bc, const int pmask[3],
                               const vdn_params *prm, int icomp, int bccomp)
{
  const int *lo = snew->lo, *hi = snew->hi;
  vo_fab rh, phi, alpha, beta[2], *bp[3] = { &beta[0], &beta[1], NULL };
  fab2(&rh, lo, hi, 0, -1, 1, 0.0); fab2(&phi, lo, hi, 1, -1, 1, 0.0); fab2(&alpha, lo, hi, 0, -1, 1, 1.0);
  fab2(&beta[0], lo, hi, 0, 0, 1, mu); fab2(&beta[1], lo, hi, 0, 1, 1, mu);
  for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++) V2(&phi, i, j, 0) = V2(snew, i, j, icomp);
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    double r = V2(snew, i, j, icomp);
    if (prm->diffusion_type == 1) r = r + mu * V2(laps, i, j, icomp);
    V2(&rh, i, j, 0) = r;
  }
  int ellbc[3][2]; vo_mgstat st;
  for (int a = 0; a < 3; a++) for (int s = 0; s < 2; s++) ellbc[a][s] = a < 2 ? bc->ell[a][s][bccomp] : VDN_BC_INT;
  double dx3[3] = { dx[0], dx[1], 1.0 };
  vo_cc_solve_ab(&rh, &phi, &alpha, bp, dx3, ellbc, 1.e-12, -1.0, prm->mg_max_iter, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, 0, &st);
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(snew, i, j, icomp) = V2(&phi, i, j, 0);
  vo_fill_boundary(snew, pmask);
  vo_physbc(snew, icomp, bccomp, 1, bc, prm);
  free(rh.p); free(phi.p); free(alpha.p); free(beta[0].p); free(beta[1].p);
}

/* ==========================================================================================
 * HG projection, dm = 2 (hgproject.f90:17-178, hg_multigrid.f90:18-119 with the 2-D kernels)
 * ======================================================================================== */
void vo2_hgproject(int proj_type, vo_fab *unew, const vo_fab *uold, vo_fab *rhohalf, vo_fab *p, vo_fab *gp, const double dx[2], double dt,
                   const vo_bc *bc, const int pmask[3], const vdn_params *prm, vo_mgstat *st)
{
  const int *lo = unew->lo, *hi = unew->hi;
  const int ng = unew->ng;
  vo_fab rh, phi, gphi, coeffs;
  fab2_nodal(&rh, lo, hi, 1, 0.0); fab2_nodal(&phi, lo, hi, 1, 0.0);
  fab2(&gphi, lo, hi, 0, -1, 2, 0.0); fab2(&coeffs, lo, hi, 1, -1, 1, 0.0);
  int ellbc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ellbc[d][s] = d < 2 ? bc->ell[d][s][bc->press_comp] : VDN_BC_INT;
  double dtinv = 1.0 / dt;
  /* create_uvec_2d (hgproject.f90:374-432) */
  for (int d = 0; d < 2; d++) for (int s = 0; s < 2; s++) if (bc->phys[d][s] == VDN_INLET) {
    int rlo[2] = { lo[0] - 1, lo[1] - 1 }, rhi[2] = { hi[0] + 1, hi[1] + 1 };
    rlo[d] = rhi[d] = s ? hi[d] + 1 : lo[d] - 1;
    for (int m = 0; m < 2; m++) for (int j = rlo[1]; j <= rhi[1]; j++) for (int i = rlo[0]; i <= rhi[0]; i++) V2(gp, i, j, m) = 0.0;
  }
  if (proj_type == VDN_PRESSURE_ITERS) {
    for (int m = 0; m < 2; m++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
      V2(unew, i, j, m) = (V2(unew, i, j, m) - V2(uold, i, j, m)) * dtinv;
  } else if (proj_type == VDN_REGULAR_TIMESTEP) {
    for (int m = 0; m < 2; m++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
      V2(unew, i, j, m) = V2(unew, i, j, m) + dt * V2(gp, i, j, m) / V2(rhohalf, i, j, 0);
  }
  for (int d = 0; d < 2; d++) for (int s = 0; s < 2; s++)
    if (bc->phys[d][s] == VDN_SLIP_WALL || bc->phys[d][s] == VDN_NO_SLIP_WALL) {
      int rlo[2] = { lo[0] - ng, lo[1] - ng }, rhi[2] = { hi[0] + ng, hi[1] + ng };
      rlo[d] = rhi[d] = s ? hi[d] + 1 : lo[d] - 1;
      for (int m = 0; m < 2; m++) for (int j = rlo[1]; j <= rhi[1]; j++) for (int i = rlo[0]; i <= rhi[0]; i++) V2(unew, i, j, m) = 0.0;
    }
  vo_fill_boundary(unew, pmask);
  double rel = prm->hg_rel_eps > 0.0 ? prm->hg_rel_eps : 1.e-12;
  double abs_eps = -1.0;
  if (proj_type == VDN_INITIAL_PROJECTION && prm->prob_type == 4) abs_eps = 1.e-12;
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(&coeffs, i, j, 0) = 1.0 / V2(rhohalf, i, j, 0);
  vo_fill_boundary(&coeffs, pmask);
  double dx3[3] = { dx[0], dx[1], 1.0 };
  vo_nd_solve(&rh, &phi, &coeffs, unew, dx3, ellbc, pmask, rel, abs_eps, prm->hg_max_iter, prm->hg_nu1, prm->hg_nu2, prm->hg_nub, prm->hg_omega, 0, NULL, st);
  /* mkgphi_2d (hgproject.f90:517-541) */
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    V2(&gphi, i, j, 0) = 0.5 * (V2(&phi, i + 1, j, 0) + V2(&phi, i + 1, j + 1, 0) - V2(&phi, i, j, 0) - V2(&phi, i, j + 1, 0)) * (1.0 / dx[0]);
    V2(&gphi, i, j, 1) = 0.5 * (V2(&phi, i, j + 1, 0) + V2(&phi, i + 1, j + 1, 0) - V2(&phi, i, j, 0) - V2(&phi, i + 1, j, 0)) * (1.0 / dx[1]);
  }
  /* hg_update_2d (hgproject.f90:581-636) */
  for (int m = 0; m < 2; m++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    double v = V2(unew, i, j, m) - V2(&gphi, i, j, m) / V2(rhohalf, i, j, 0);
    if (proj_type == VDN_PRESSURE_ITERS) v = V2(uold, i, j, m) + dt * v;
    V2(unew, i, j, m) = v;
  }
  if (proj_type == VDN_INITIAL_PROJECTION || proj_type == VDN_DIVU_ITERS) {
    memset(gp->p, 0, sizeof(double) * vo_size(gp)); memset(p->p, 0, sizeof(double) * vo_size(p));
  } else if (proj_type == VDN_PRESSURE_ITERS) {
    for (int m = 0; m < 2; m++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(gp, i, j, m) = V2(gp, i, j, m) + V2(&gphi, i, j, m);
    for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0] + 1; i++) V2(p, i, j, 0) = V2(p, i, j, 0) + V2(&phi, i, j, 0);
  } else if (proj_type == VDN_REGULAR_TIMESTEP) {
    for (int m = 0; m < 2; m++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(gp, i, j, m) = dtinv * V2(&gphi, i, j, m);
    for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0] + 1; i++) V2(p, i, j, 0) = dtinv * V2(&phi, i, j, 0);
  }
  vo_fill_boundary(gp, pmask); vo_fill_boundary(p, pmask);
  free(rh.p); free(phi.p); free(gphi.p); free(coeffs.p);
}

/* ==========================================================================================
 * advance_timestep with dm = 2 (advance_timestep.f90:26-170 and callees) -- same orchestration as oracle/vo_advance.c
 * ======================================================================================== */
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static void restrict_and_fill2(vo_fab *f, int icomp, int bcomp, int nc, int same_boundary, const vo_bc *bc, const int pmask[3], const vdn_params *prm)
{
  vo_fill_boundary(f, pmask);
  for (int c = 0; c < nc; c++) vo_physbc(f, icomp + c, same_boundary ? bcomp : bcomp + c, 1, bc, prm);
}
void vo2_advance_timestep(vo_state *S, const double dx[2], double dt, const vo_bc *bc, const int pmask[3], const vdn_params *prm,
                          int proj_type, vo_mgstat st[2], double phase_sec[4])
{
  const int *lo = S->uold.lo, *hi = S->uold.hi;
  const int dm = 2, nscal = prm->nscal;
  const int viscous = prm->visc_coef > 0.0, diffusive = prm->diff_coef > 0.0;
  vo_fab mac_rhs, rhohalf, umac[2], *ump[2], vel_force, scal_force, divu, lapu, laps;
  vo_fab sedge[2], sflux[2], uedge[2], uflux[2], *sep[2], *sfp[2], *uep[2], *ufp[2];
  double t0;
  fab2(&mac_rhs, lo, hi, 1, -1, 1, 0.0); fab2(&rhohalf, lo, hi, 1, -1, dm, 0.0);
  for (int d = 0; d < 2; d++) { fab2(&umac[d], lo, hi, 1, d, 1, 1.e20); ump[d] = &umac[d]; }
  fab2(&lapu, lo, hi, 0, -1, dm, 0.0);
  if (viscous) for (int c = 0; c < dm; c++) vo2_explicit_diffusive_term(&lapu, &S->uold, c, c, dx, bc);
  fab2(&vel_force, lo, hi, 1, -1, dm, 0.0);
  vo2_mkvelforce(&vel_force, &S->ext_vel_force, &S->gp, &S->sold, viscous ? &lapu : NULL, 1.0, prm);
  restrict_and_fill2(&vel_force, 0, bc->extrap_comp, dm, 1, bc, pmask, prm);
  vo2_velpred(&S->uold, ump, &vel_force, dx, dt, bc, prm);
  for (int d = 0; d < 2; d++) vo_fill_boundary(&umac[d], pmask);
  free(vel_force.p);
  t0 = now();
  vo2_macproject(ump, &S->sold, &mac_rhs, dx, bc, pmask, prm, &st[0]);
  if (phase_sec) phase_sec[2] = now() - t0;
  t0 = now();
  {
    int is_cons[VO_MAXCOMP]; is_cons[0] = 1; for (int c = 1; c < nscal; c++) is_cons[c] = 0;
    fab2(&scal_force, lo, hi, 1, -1, nscal, 0.0); fab2(&divu, lo, hi, 1, -1, 1, 0.0);
    for (int d = 0; d < 2; d++) { fab2(&sflux[d], lo, hi, 0, d, nscal, 0.0); fab2(&sedge[d], lo, hi, 0, d, nscal, 0.0); sfp[d] = &sflux[d]; sep[d] = &sedge[d]; }
    fab2(&laps, lo, hi, 0, -1, nscal, 0.0);
    if (diffusive) for (int c = 1; c < nscal; c++) vo2_explicit_diffusive_term(&laps, &S->sold, c, dm + c, dx, bc);
    vo2_mkscalforce(&scal_force, &S->ext_scal_force, diffusive ? &laps : NULL, 1.0, prm);
    restrict_and_fill2(&scal_force, 0, bc->extrap_comp, nscal, 1, bc, pmask, prm);
    vo2_mkflux(&S->sold, sep, sfp, ump, &scal_force, &divu, dx, dt, 0, is_cons, dm, bc, prm);
    vo2_mkscalforce(&scal_force, &S->ext_scal_force, diffusive ? &laps : NULL, 0.0, prm);
    restrict_and_fill2(&scal_force, 0, bc->extrap_comp, nscal, 1, bc, pmask, prm);
    vo2_update(&S->sold, ump, sep, sfp, &scal_force, &S->snew, dx, dt, 0, is_cons);
    restrict_and_fill2(&S->snew, 0, dm, nscal, 0, bc, pmask, prm);
    if (diffusive) {
      double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->diff_coef : dt * prm->diff_coef;
      for (int c = 1; c < nscal; c++) diff_scalar_solve2(&S->snew, &laps, dx, visc_mu, bc, pmask, prm, c, dm + c);
    }
    free(laps.p); free(scal_force.p); free(divu.p);
    for (int d = 0; d < 2; d++) { free(sflux[d].p); free(sedge[d].p); }
  }
  if (phase_sec) phase_sec[0] = now() - t0;
  for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) V2(&rhohalf, i, j, 0) = 0.5 * (V2(&S->sold, i, j, 0) + V2(&S->snew, i, j, 0));
  restrict_and_fill2(&rhohalf, 0, dm + 0, 1, 0, bc, pmask, prm);
  if (prm->diffusion_type == 2) memset(lapu.p, 0, sizeof(double) * vo_size(&lapu));
  t0 = now();
  {
    int is_cons[2] = { 0, 0 };
    fab2(&vel_force, lo, hi, 1, -1, dm, 0.0);
    for (int d = 0; d < 2; d++) { fab2(&uflux[d], lo, hi, 0, d, dm, 0.0); fab2(&uedge[d], lo, hi, 0, d, dm, 0.0); ufp[d] = &uflux[d]; uep[d] = &uedge[d]; }
    vo2_mkvelforce(&vel_force, &S->ext_vel_force, &S->gp, &S->sold, viscous ? &lapu : NULL, 1.0, prm);
    restrict_and_fill2(&vel_force, 0, bc->extrap_comp, dm, 1, bc, pmask, prm);
    vo2_mkflux(&S->uold, uep, ufp, ump, &vel_force, &mac_rhs, dx, dt, 1, is_cons, 0, bc, prm);
    vo2_mkvelforce(&vel_force, &S->ext_vel_force, &S->gp, &rhohalf, viscous ? &lapu : NULL, 0.0, prm);
    restrict_and_fill2(&vel_force, 0, bc->extrap_comp, dm, 1, bc, pmask, prm);
    vo2_update(&S->uold, ump, uep, ufp, &vel_force, &S->unew, dx, dt, 1, is_cons);
    restrict_and_fill2(&S->unew, 0, 0, dm, 0, bc, pmask, prm);
    if (viscous) {
      double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->visc_coef : dt * prm->visc_coef;
      visc_solve2(&S->unew, &lapu, &rhohalf, &mac_rhs, dx, visc_mu, bc, pmask, prm);
    }
    free(vel_force.p);
    for (int d = 0; d < 2; d++) { free(uflux[d].p); free(uedge[d].p); }
  }
  if (phase_sec) phase_sec[1] = now() - t0;
  t0 = now();
  vo2_hgproject(proj_type, &S->unew, &S->uold, &rhohalf, &S->p, &S->gp, dx, dt, bc, pmask, prm, &st[1]);
  if (phase_sec) phase_sec[3] = now() - t0;
  free(lapu.p); free(mac_rhs.p); free(rhohalf.p);
  for (int d = 0; d < 2; d++) free(umac[d].p);
}
