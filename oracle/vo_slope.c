/* oracle/vo_slope.c -- limited slopes, reference src/slope.f90:148-588.
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * slopex_2d / slopey_2d / slopez_3d are the same 1-D operator applied along x, y, z
 * (the 3-D callers loop k-planes over the 2-D routines, velpred.f90:1848-1852,
 * mkflux.f90:1256-1260); this file applies it along direction `dir` on every line
 * of the grown box [lo-1,hi+1] in the two transverse directions.
 */
#include <math.h>
#include <stdlib.h>
#include "vo.h"

static inline double sgn1(double x) { return copysign(1.0, x); }   /* Fortran sign(one,x) */

/* The lines of slopes along y and z are strided in memory (a z-line of a 128^3 fab touches a new page every element: 40 ns per cell, half
 * of mkflux's time).  For dir != 0 the operator therefore runs on a whole x-row of lines at once: every array of the 1-D formulation gets a
 * second, unit-stride index `x`; per (line, position) the arithmetic is the same expression in the same order, hence the same bits. */
static void slope_rows(const vo_fab *s, vo_fab *sl, int dir, int nc, int bccomp, const vo_bc *bc, int slope_order)
{
  const int *lo = s->lo, *hi = s->hi;
  const int is = lo[dir], ie = hi[dir];
  const int o = 3 - dir;                        /* the transverse direction that is not x (dir is 1 or 2) */
  const double two3rd = 2.0 / 3.0, sixth = 1.0 / 6.0, third = 1.0 / 3.0, tenth = 0.1;
  const int nline = ie - is + 5;
  const int gx = 1, go = (o == 2 && s->dm == 2) ? 0 : 1;
  const int xlo = lo[0] - gx, xhi = hi[0] + gx, nx = xhi - xlo + 1;
  #pragma omp parallel
  {
  double *cen = (double *)malloc(sizeof(double) * (size_t)nline * nx * 4);
  double *lim = cen + (size_t)nline * nx, *flag = lim + (size_t)nline * nx, *fromm = flag + (size_t)nline * nx;
  #define RC(a, p, x) a[(size_t)((p) - (is - 2)) * nx + ((x) - xlo)]
  for (int comp = 0; comp < nc; comp++) {
    const int bclo = bc->adv[dir][0][bccomp + comp], bchi = bc->adv[dir][1][bccomp + comp];
    const int lo_special = (bclo == VDN_EXT_DIR || bclo == VDN_HOEXTRAP);
    const int hi_special = (bchi == VDN_EXT_DIR || bchi == VDN_HOEXTRAP);
    #pragma omp for
    for (int b = lo[o] - go; b <= hi[o] + go; b++) {
      int q[3]; q[o] = b;
      /* row pointers: S(p) + x and SL(p) + x */
      #define SROW(p) (q[dir] = (p), q[0] = xlo, &VF(s, q[0], q[1], q[2], comp))
      #define LROW(p) (q[dir] = (p), q[0] = xlo, &VF(sl, q[0], q[1], q[2], comp))
      if (slope_order == 2) {                /* slope.f90:177-219 */
        for (int p = is - 1; p <= ie + 1; p++) {
          const double *rp = SROW(p + 1), *r0 = SROW(p), *rm = SROW(p - 1); double *out = LROW(p);
          for (int x = 0; x < nx; x++) {
            double sp = rp[x], s0 = r0[x], sm = rm[x];
            double del = 0.5 * (sp - sm), dpls = 2.0 * (sp - s0), dmin = 2.0 * (s0 - sm);
            double slim = fmin(fabs(dpls), fabs(dmin));
            slim = (dpls * dmin > 0.0) ? slim : 0.0;
            out[x] = sgn1(del) * fmin(slim, fabs(del));
          }
        }
        if (lo_special) {
          const double *rp = SROW(is + 1), *r0 = SROW(is), *rm = SROW(is - 1); double *o1 = LROW(is - 1), *o0 = LROW(is);
          for (int x = 0; x < nx; x++) {
            o1[x] = 0.0;
            double sp = rp[x], s0 = r0[x], sm = rm[x];
            double del = (sp + 3.0 * s0 - 4.0 * sm) * third;
            double dpls = 2.0 * (sp - s0), dmin = 2.0 * (s0 - sm);
            double slim = fmin(fabs(dpls), fabs(dmin));
            slim = (dpls * dmin > 0.0) ? slim : 0.0;
            o0[x] = sgn1(del) * fmin(slim, fabs(del));
          }
        }
        if (hi_special) {
          const double *rp = SROW(ie + 1), *r0 = SROW(ie), *rm = SROW(ie - 1); double *o1 = LROW(ie + 1), *o0 = LROW(ie);
          for (int x = 0; x < nx; x++) {
            o1[x] = 0.0;
            double sp = rp[x], s0 = r0[x], sm = rm[x];
            double del = -(sm + 3.0 * s0 - 4.0 * sp) * third;
            double dpls = 2.0 * (s0 - sm), dmin = 2.0 * (sp - s0);
            double slim = fmin(fabs(dpls), fabs(dmin));
            slim = (dpls * dmin > 0.0) ? slim : 0.0;
            o0[x] = sgn1(del) * fmin(slim, fabs(del));
          }
        }
      } else {                               /* 4th order, slope.f90:221-284 */
        for (int p = is - 2; p <= ie + 2; p++) {
          const double *rp = SROW(p + 1), *r0 = SROW(p), *rm = SROW(p - 1);
          for (int x = 0; x < nx; x++) {
            double sp = rp[x], s0 = r0[x], sm = rm[x];
            double c = 0.5 * (sp - sm);
            double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
            double l = fmin(fabs(dmin), fabs(dpls));
            double lm = (dpls * dmin > 0.0) ? l : 0.0;
            double fl = sgn1(c);
            RC(cen, p, x + xlo) = c; RC(lim, p, x + xlo) = lm; RC(flag, p, x + xlo) = fl;
            RC(fromm, p, x + xlo) = fl * fmin(lm, fabs(c));
          }
        }
        for (int p = is - 1; p <= ie + 1; p++) {
          double *out = LROW(p);
          for (int x = 0; x < nx; x++) {
            double ds = 2.0 * two3rd * RC(cen, p, x + xlo) - sixth * (RC(fromm, p + 1, x + xlo) + RC(fromm, p - 1, x + xlo));
            out[x] = RC(flag, p, x + xlo) * fmin(fabs(ds), RC(lim, p, x + xlo));
          }
        }
        if (lo_special) {                    /* slope.f90:243-262 */
          const double *rm = SROW(is - 1), *r0 = SROW(is), *rp = SROW(is + 1), *rpp = SROW(is + 2);
          double *om = LROW(is - 1), *o0 = LROW(is), *o1 = LROW(is + 1);
          for (int x = 0; x < nx; x++) {
            om[x] = 0.0;
            double sm = rm[x], s0 = r0[x], sp = rp[x], spp = rpp[x];
            double del = -16.0 / 15.0 * sm + 0.5 * s0 + two3rd * sp - tenth * spp;
            double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
            double slim = fmin(fabs(dpls), fabs(dmin));
            slim = (dpls * dmin > 0.0) ? slim : 0.0;
            double v = sgn1(del) * fmin(slim, fabs(del));
            o0[x] = v;
            RC(fromm, is, x + xlo) = v;
            double ds = 2.0 * two3rd * RC(cen, is + 1, x + xlo) - sixth * (RC(fromm, is + 2, x + xlo) + RC(fromm, is, x + xlo));
            o1[x] = RC(flag, is + 1, x + xlo) * fmin(fabs(ds), RC(lim, is + 1, x + xlo));
          }
        }
        if (hi_special) {                    /* slope.f90:264-283 */
          const double *rp = SROW(ie + 1), *r0 = SROW(ie), *rm = SROW(ie - 1), *rmm = SROW(ie - 2);
          double *op = LROW(ie + 1), *o0 = LROW(ie), *o1 = LROW(ie - 1);
          for (int x = 0; x < nx; x++) {
            op[x] = 0.0;
            double sp = rp[x], s0 = r0[x], sm = rm[x], smm = rmm[x];
            double del = -(-16.0 / 15.0 * sp + 0.5 * s0 + two3rd * sm - tenth * smm);
            double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
            double slim = fmin(fabs(dpls), fabs(dmin));
            slim = (dpls * dmin > 0.0) ? slim : 0.0;
            double v = sgn1(del) * fmin(slim, fabs(del));
            o0[x] = v;
            RC(fromm, ie, x + xlo) = v;
            double ds = 2.0 * two3rd * RC(cen, ie - 1, x + xlo) - sixth * (RC(fromm, ie - 2, x + xlo) + RC(fromm, ie, x + xlo));
            o1[x] = RC(flag, ie - 1, x + xlo) * fmin(fabs(ds), RC(lim, ie - 1, x + xlo));
          }
        }
      }
      #undef SROW
      #undef LROW
    }
  }
  #undef RC
  free(cen);
  }
}

void vo_slope(const vo_fab *s, vo_fab *sl, int dir, int nc, int bccomp, const vo_bc *bc, int slope_order)
{
  const int *lo = s->lo, *hi = s->hi;
  int is = lo[dir], ie = hi[dir];
  int t1 = (dir + 1) % 3, t2 = (dir + 2) % 3;
  const double two3rd = 2.0 / 3.0, sixth = 1.0 / 6.0, third = 1.0 / 3.0, tenth = 0.1;
  int nline = ie - is + 5;

  if (slope_order == 0) {                    /* slope.f90:173-174 */
    for (long n = 0; n < vo_size(sl); n++) sl->p[n] = 0.0;
    return;
  }
  if (dir != 0 && s->dm == 3) { slope_rows(s, sl, dir, nc, bccomp, bc, slope_order); return; }
  #pragma omp parallel
  {
  double *cen = (double *)malloc(sizeof(double) * nline * 4);
  double *lim = cen + nline, *flag = lim + nline, *fromm = flag + nline;
  #define SC(a, i) a[(i) - (is - 2)]
  for (int comp = 0; comp < nc; comp++) {
    int bclo = bc->adv[dir][0][bccomp + comp], bchi = bc->adv[dir][1][bccomp + comp];
    int lo_special = (bclo == VDN_EXT_DIR || bclo == VDN_HOEXTRAP);
    int hi_special = (bchi == VDN_EXT_DIR || bchi == VDN_HOEXTRAP);
    #pragma omp for collapse(2)
    for (int b2 = lo[t2] - ((t2 == 2 && s->dm == 2) ? 0 : 1); b2 <= hi[t2] + ((t2 == 2 && s->dm == 2) ? 0 : 1); b2++)
    for (int b1 = lo[t1] - ((t1 == 2 && s->dm == 2) ? 0 : 1); b1 <= hi[t1] + ((t1 == 2 && s->dm == 2) ? 0 : 1); b1++) {
      int q[3]; q[t1] = b1; q[t2] = b2;
      #define S(i)  (q[dir] = (i), VF(s, q[0], q[1], q[2], comp))
      #define SL(i) (q[dir] = (i), &VF(sl, q[0], q[1], q[2], comp))
      if (slope_order == 2) {                /* slope.f90:177-219 */
        for (int i = is - 1; i <= ie + 1; i++) {
          double sp = S(i + 1), s0 = S(i), sm = S(i - 1);
          double del = 0.5 * (sp - sm), dpls = 2.0 * (sp - s0), dmin = 2.0 * (s0 - sm);
          double slim = fmin(fabs(dpls), fabs(dmin));
          slim = (dpls * dmin > 0.0) ? slim : 0.0;
          *SL(i) = sgn1(del) * fmin(slim, fabs(del));
        }
        if (lo_special) {
          *SL(is - 1) = 0.0;
          double sp = S(is + 1), s0 = S(is), sm = S(is - 1);
          double del = (sp + 3.0 * s0 - 4.0 * sm) * third;
          double dpls = 2.0 * (sp - s0), dmin = 2.0 * (s0 - sm);
          double slim = fmin(fabs(dpls), fabs(dmin));
          slim = (dpls * dmin > 0.0) ? slim : 0.0;
          *SL(is) = sgn1(del) * fmin(slim, fabs(del));
        }
        if (hi_special) {
          *SL(ie + 1) = 0.0;
          double sp = S(ie + 1), s0 = S(ie), sm = S(ie - 1);
          double del = -(sm + 3.0 * s0 - 4.0 * sp) * third;
          double dpls = 2.0 * (s0 - sm), dmin = 2.0 * (sp - s0);
          /* NB: x uses (dpls,dmin) = (2(s_ie - s_ie-1), 2(s_ie+1 - s_ie)) at slope.f90:209-210 while
           * y/z (369-370, 510-511) swap the names; min/product are symmetric so the value is equal */
          double slim = fmin(fabs(dpls), fabs(dmin));
          slim = (dpls * dmin > 0.0) ? slim : 0.0;
          *SL(ie) = sgn1(del) * fmin(slim, fabs(del));
        }
      } else {                               /* 4th order, slope.f90:221-284 */
        for (int i = is - 2; i <= ie + 2; i++) {
          double sp = S(i + 1), s0 = S(i), sm = S(i - 1);
          SC(cen, i) = 0.5 * (sp - sm);
          double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
          double l = fmin(fabs(dmin), fabs(dpls));
          SC(lim, i) = (dpls * dmin > 0.0) ? l : 0.0;
          SC(flag, i) = sgn1(SC(cen, i));
          SC(fromm, i) = SC(flag, i) * fmin(SC(lim, i), fabs(SC(cen, i)));
        }
        for (int i = is - 1; i <= ie + 1; i++) {
          double ds = 2.0 * two3rd * SC(cen, i) - sixth * (SC(fromm, i + 1) + SC(fromm, i - 1));
          *SL(i) = SC(flag, i) * fmin(fabs(ds), SC(lim, i));
        }
        if (lo_special) {                    /* slope.f90:243-262 */
          *SL(is - 1) = 0.0;
          double sm = S(is - 1), s0 = S(is), sp = S(is + 1), spp = S(is + 2);
          double del = -16.0 / 15.0 * sm + 0.5 * s0 + two3rd * sp - tenth * spp;
          double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
          double slim = fmin(fabs(dpls), fabs(dmin));
          slim = (dpls * dmin > 0.0) ? slim : 0.0;
          double v = sgn1(del) * fmin(slim, fabs(del));
          *SL(is) = v;
          SC(fromm, is) = v;
          double ds = 2.0 * two3rd * SC(cen, is + 1) - sixth * (SC(fromm, is + 2) + SC(fromm, is));
          *SL(is + 1) = SC(flag, is + 1) * fmin(fabs(ds), SC(lim, is + 1));
        }
        if (hi_special) {                    /* slope.f90:264-283 */
          *SL(ie + 1) = 0.0;
          double sp = S(ie + 1), s0 = S(ie), sm = S(ie - 1), smm = S(ie - 2);
          double del = -(-16.0 / 15.0 * sp + 0.5 * s0 + two3rd * sm - tenth * smm);
          double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
          double slim = fmin(fabs(dpls), fabs(dmin));
          slim = (dpls * dmin > 0.0) ? slim : 0.0;
          double v = sgn1(del) * fmin(slim, fabs(del));
          *SL(ie) = v;
          SC(fromm, ie) = v;
          double ds = 2.0 * two3rd * SC(cen, ie - 1) - sixth * (SC(fromm, ie - 2) + SC(fromm, ie));
          *SL(ie - 1) = SC(flag, ie - 1) * fmin(fabs(ds), SC(lim, ie - 1));
        }
      }
      #undef S
      #undef SL
    }
  }
  #undef SC
  free(cen);
  }
}
