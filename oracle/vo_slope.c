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
