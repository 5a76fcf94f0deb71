/* oracle/vo_amr.c -- two-level AMR pieces of the hot path (BASELINE.json configs[3]): the FBoxLib multi-level
 * operators the reference calls (ml_cc_restriction, ml_edge_restriction, multifab_fill_ghost_cells, create_umac_grown,
 * ml_restrict_and_fill, ml_cc_solve) and the multilevel macproject (src/macproject.f90:20-133) built on them.
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * None of these operators is in the reference tree (FBoxLib); the call sites fix WHAT they must do, the definitions
 * below are ours and are the ones the HIP path (varden_amd/csrc/amr.hip) implements:
 *   ml_cc_restriction     coarse cell = mean of its 8 fine cells                       (macproject.f90:204-206, hgproject.f90:355-357)
 *   ml_edge_restriction   coarse face = mean of the 4 fine faces that cover it         (velpred.f90:115-119, macproject.f90:330-333, 497-500)
 *   fill_ghost_cells      fine ghost cell = coarse parent + limited linear slopes      (macproject.f90:304-310; ml_restrict_and_fill)
 *                         (MC-limited central differences per direction, the limiter of slope.f90:181-187)
 *   create_umac_grown     fine ghost face = coarse face value (even index) or the mean of the two coarse faces around it
 *                         (odd index), piecewise constant across the face           (velpred.f90:102-107, macproject.f90:107-113)
 *   ml_cc_solve           composite solve: fine cells + uncovered coarse cells; coarse-fine ghost cells by quadratic
 *                         interpolation normal to the interface (8/15, 2/3, -1/5) of the transversely (central-slope)
 *                         interpolated coarse value and two fine cells; the coarse flux through an interface face is the
 *                         mean of the four fine fluxes.  Algorithm: FAC iteration -- composite residual, one V-cycle of
 *                         nu1 red-black sweeps on the fine level (homogeneous interface), one V-cycle of the single-level
 *                         multigrid on the whole coarse level, piecewise-constant prolongation, nu2 fine sweeps.
 * This round: ONE box per level, two levels, refinement ratio 2, the fine box properly nested (>= 1 coarse cell away from
 * any domain face it does not touch).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "vo.h"

static inline int fdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }       /* floor(a/2) */
static inline double sgn1(double x) { return copysign(1.0, x); }
static inline double mc_limited(double del, double sm, double s0, double sp)
{
  double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
  double slim = fmin(fabs(dpls), fabs(dmin));
  slim = (dpls * dmin > 0.0) ? slim : 0.0;
  return sgn1(del) * fmin(slim, fabs(del));
}
static inline int in_alloc(const vo_fab *f, int i, int j, int k)
{
  return i >= f->lo[0] - f->ng && i <= f->hi[0] + f->nd[0] + f->ng && j >= f->lo[1] - f->ng && j <= f->hi[1] + f->nd[1] + f->ng &&
         k >= f->lo[2] - f->gz && k <= f->hi[2] + f->nd[2] + f->gz;
}

void vo_ml_cc_restriction(vo_fab *crse, const vo_fab *fine, int icomp, int nc)
{
  for (int c = icomp; c < icomp + nc; c++)
  for (int K = fine->lo[2] / 2; K <= fine->hi[2] / 2; K++) for (int J = fine->lo[1] / 2; J <= fine->hi[1] / 2; J++) for (int I = fine->lo[0] / 2; I <= fine->hi[0] / 2; I++) {
    double s = 0.0;
    for (int kk = 0; kk < 2; kk++) for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) s = s + VF(fine, 2 * I + ii, 2 * J + jj, 2 * K + kk, c);
    VF(crse, I, J, K, c) = s * 0.125;
  }
}
void vo_ml_edge_restriction(vo_fab *crse, const vo_fab *fine, int dir)
{
  int lo[3], hi[3];
  for (int d = 0; d < 3; d++) { lo[d] = fine->lo[d] / 2; hi[d] = fine->hi[d] / 2; }
  hi[dir] += 1;
  const int t1 = (dir + 1) % 3, t2 = (dir + 2) % 3;
  for (int K = lo[2]; K <= hi[2]; K++) for (int J = lo[1]; J <= hi[1]; J++) for (int I = lo[0]; I <= hi[0]; I++) {
    int Q[3] = { I, J, K };
    double s = 0.0;
    for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) {
      int q[3]; q[dir] = 2 * Q[dir]; q[t1] = 2 * Q[t1] + a; q[t2] = 2 * Q[t2] + b;
      s = s + VF(fine, q[0], q[1], q[2], 0);
    }
    VF(crse, I, J, K, 0) = s * 0.25;
  }
}
/* every ghost cell of the fine fab whose parent lies inside the coarse fab's allocation */
void vo_fill_ghost_cells(vo_fab *fine, const vo_fab *crse, int icomp, int nc)
{
  const int ng = fine->ng;
  for (int c = icomp; c < icomp + nc; c++)
  for (int k = fine->lo[2] - ng; k <= fine->hi[2] + ng; k++) for (int j = fine->lo[1] - ng; j <= fine->hi[1] + ng; j++) for (int i = fine->lo[0] - ng; i <= fine->hi[0] + ng; i++) {
    if (i >= fine->lo[0] && i <= fine->hi[0] && j >= fine->lo[1] && j <= fine->hi[1] && k >= fine->lo[2] && k <= fine->hi[2]) continue;
    const int q[3] = { i, j, k }, P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
    if (!in_alloc(crse, P[0], P[1], P[2])) continue;
    const double c0 = VF(crse, P[0], P[1], P[2], c);
    double v = c0;
    for (int d = 0; d < 3; d++) {
      int m[3] = { P[0], P[1], P[2] }, p[3] = { P[0], P[1], P[2] }; m[d] -= 1; p[d] += 1;
      double sl = 0.0;
      if (in_alloc(crse, m[0], m[1], m[2]) && in_alloc(crse, p[0], p[1], p[2])) {
        const double cm = VF(crse, m[0], m[1], m[2], c), cp = VF(crse, p[0], p[1], p[2], c);
        sl = mc_limited(0.5 * (cp - cm), cm, c0, cp);
      }
      const double sg = (q[d] - 2 * P[d]) ? 0.25 : -0.25;
      v = v + sg * sl;
    }
    VF(fine, i, j, k, c) = v;
  }
}
/* ---- regridding -------------------------------------------------------------------------------------------
 * tag_boxes_3d (src/tag_boxes.f90:128-216; tag_boxes_2d :41-127 has the same thresholds): a cell of the level is tagged where the first
 * component of the state exceeds 1.01 (level 1), 1.1 (level 2), 1.5 (deeper levels) for prob_type 1 and 2, or lies strictly between
 * 1.2 and 1.8 for prob_type 3 (all levels).  tags: one byte per VALID cell of the fab, x fastest.  Returns -1 for any other prob_type
 * (bl_error('Unsupported prob_type'), :212). */
int vo_tag_boxes(const vo_fab *s, int lev, int prob_type, unsigned char *tags)
{
  const int nx = s->hi[0] - s->lo[0] + 1, ny = s->hi[1] - s->lo[1] + 1;
  if (!(prob_type == 1 || prob_type == 2 || prob_type == 3)) return -1;
  for (int k = s->lo[2]; k <= s->hi[2]; k++) for (int j = s->lo[1]; j <= s->hi[1]; j++) for (int i = s->lo[0]; i <= s->hi[0]; i++) {
    const double v = VF(s, i, j, k, 0);
    int t = 0;                                                  /* tagbox = .false. (:140) */
    if (prob_type == 1 || prob_type == 2) {
      if (lev == 1) t = v > 1.01;                               /* :147-155 */
      else if (lev == 2) t = v > 1.1;                           /* :158-166 */
      else t = v > 1.5;                                         /* :169-177 */
    } else t = (v > 1.2 && v < 1.8);                            /* :181-210, the same test on every level */
    tags[(size_t)(i - s->lo[0]) + (size_t)nx * ((size_t)(j - s->lo[1]) + (size_t)ny * (size_t)(k - s->lo[2]))] = (unsigned char)t;
  }
  return 0;
}
/* fillpatch(fine, crse, ng = 0, ...) as regrid.f90:311-325 uses it (FBoxLib routine, absent from the tree: OUR definition): every valid
 * cell of the new fine fab from the coarse one by the interpolation of vo_fill_ghost_cells -- parent value + MC-limited central
 * slopes times -1/4 / +1/4 per direction, zero slope where a coarse neighbour lies outside the coarse fab's allocation */
void vo_fillpatch(vo_fab *fine, const vo_fab *crse, int icomp, int nc)
{
  for (int c = icomp; c < icomp + nc; c++)
  for (int k = fine->lo[2]; k <= fine->hi[2]; k++) for (int j = fine->lo[1]; j <= fine->hi[1]; j++) for (int i = fine->lo[0]; i <= fine->hi[0]; i++) {
    const int q[3] = { i, j, k }, P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
    if (P[0] < crse->lo[0] || P[0] > crse->hi[0] || P[1] < crse->lo[1] || P[1] > crse->hi[1] || P[2] < crse->lo[2] || P[2] > crse->hi[2]) continue;
    const double c0 = VF(crse, P[0], P[1], P[2], c);
    double v = c0;
    for (int d = 0; d < 3; d++) {
      int m[3] = { P[0], P[1], P[2] }, p[3] = { P[0], P[1], P[2] }; m[d] -= 1; p[d] += 1;
      double sl = 0.0;
      if (in_alloc(crse, m[0], m[1], m[2]) && in_alloc(crse, p[0], p[1], p[2])) {
        const double cm = VF(crse, m[0], m[1], m[2], c), cp = VF(crse, p[0], p[1], p[2], c);
        sl = mc_limited(0.5 * (cp - cm), cm, c0, cp);
      }
      const double sg = (q[d] - 2 * P[d]) ? 0.25 : -0.25;
      v = v + sg * sl;
    }
    VF(fine, i, j, k, c) = v;
  }
}
/* ml_nodal_prolongation(fine, crse, rr) of regrid.f90:327 (FBoxLib routine: OUR definition): trilinear interpolation of the nodal pressure --
 * a fine node that coincides with a coarse node copies it, one on a coarse edge / face / cell centre takes the mean of the 2 / 4 / 8
 * coarse nodes around it (sum in z, y, x order, times the reciprocal of the count) */
void vo_nodal_prolongation(vo_fab *fine, const vo_fab *crse)
{
  for (int k = fine->lo[2]; k <= fine->hi[2] + 1; k++) for (int j = fine->lo[1]; j <= fine->hi[1] + 1; j++) for (int i = fine->lo[0]; i <= fine->hi[0] + 1; i++) {
    const int I = fdiv2(i), J = fdiv2(j), K = fdiv2(k), oi = i - 2 * I, oj = j - 2 * J, ok = k - 2 * K;
    if (I < crse->lo[0] || I > crse->hi[0] + 1 || J < crse->lo[1] || J > crse->hi[1] + 1 || K < crse->lo[2] || K > crse->hi[2] + 1) continue;
    double s = 0.0;
    for (int c = 0; c <= ok; c++) for (int b = 0; b <= oj; b++) for (int a = 0; a <= oi; a++) s = s + VF(crse, I + a, J + b, K + c, 0);
    VF(fine, i, j, k, 0) = s * (1.0 / (double)((1 + oi) * (1 + oj) * (1 + ok)));
  }
}
void vo_create_umac_grown(vo_fab *fine, const vo_fab *crse, int dir)
{
  const int ng = fine->ng;
  for (int k = fine->lo[2] - ng; k <= fine->hi[2] + fine->nd[2] + ng; k++) for (int j = fine->lo[1] - ng; j <= fine->hi[1] + fine->nd[1] + ng; j++)
  for (int i = fine->lo[0] - ng; i <= fine->hi[0] + fine->nd[0] + ng; i++) {
    if (i >= fine->lo[0] && i <= fine->hi[0] + fine->nd[0] && j >= fine->lo[1] && j <= fine->hi[1] + fine->nd[1] && k >= fine->lo[2] && k <= fine->hi[2] + fine->nd[2]) continue;
    const int q[3] = { i, j, k };
    int P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) }, P2[3];
    const int odd = q[dir] - 2 * P[dir];
    P2[0] = P[0]; P2[1] = P[1]; P2[2] = P[2]; P2[dir] += 1;
    if (!in_alloc(crse, P[0], P[1], P[2]) || (odd && !in_alloc(crse, P2[0], P2[1], P2[2]))) continue;
    VF(fine, i, j, k, 0) = odd ? 0.5 * (VF(crse, P[0], P[1], P[2], 0) + VF(crse, P2[0], P2[1], P2[2], 0)) : VF(crse, P[0], P[1], P[2], 0);
  }
}
/* fill_boundary of a level that is one box inside the domain pd: periodic wrap only where the box spans the domain */
static void level_fill_boundary(vo_fab *f, const int pmask[3], const int pdlo[3], const int pdhi[3])
{
  int pm[3];
  for (int d = 0; d < 3; d++) pm[d] = pmask[d] && f->lo[d] == pdlo[d] && f->hi[d] == pdhi[d];
  vo_fill_boundary(f, pm);
}
/* ml_restrict_and_fill for two levels: average down, coarse ghosts, fine ghosts (coarse interpolation, then same-level
 * periodic images, then the physical boundary) */
void vo_ml_restrict_and_fill(int nlev, vo_fab **mf, int icomp, int bcomp, int nc, int same_boundary, const vo_bc *bc, const int pmask[3],
                             const int *pd /* [lev][2][3] */, const vdn_params *prm)
{
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction(mf[n - 1], mf[n], icomp, nc);
  for (int n = 0; n < nlev; n++) {
    if (n > 0) vo_fill_ghost_cells(mf[n], mf[n - 1], icomp, nc);
    level_fill_boundary(mf[n], pmask, pd + 6 * n, pd + 6 * n + 3);
    for (int c = 0; c < nc; c++) vo_physbc(mf[n], icomp + c, same_boundary ? bcomp : bcomp + c, 1, &bc[n], prm);
  }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * composite cell-centred solve, two levels
 * ------------------------------------------------------------------------------------------------------------------- */
/* ghost layer of phi on one level: domain faces by the solver's closure (Neumann: phi_i, Dirichlet: -phi_i), periodic images */
static void phi_closure(vo_fab *phi, const int ellbc[3][2], const int pmask[3], const int pdlo[3], const int pdhi[3])
{
  const int *lo = phi->lo, *hi = phi->hi;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    if (ellbc[d][s] != VDN_BC_NEU && ellbc[d][s] != VDN_BC_DIR) continue;
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    for (int b2 = lo[t2]; b2 <= hi[t2]; b2++) for (int b1 = lo[t1]; b1 <= hi[t1]; b1++) {
      int q[3], g[3]; q[t1] = g[t1] = b1; q[t2] = g[t2] = b2; q[d] = s ? hi[d] : lo[d]; g[d] = s ? hi[d] + 1 : lo[d] - 1;
      const double v = VF(phi, q[0], q[1], q[2], 0);
      VF(phi, g[0], g[1], g[2], 0) = (ellbc[d][s] == VDN_BC_NEU) ? v : -v;
    }
  }
  level_fill_boundary(phi, pmask, pdlo, pdhi);
}
/* coarse-fine ghost cells of the fine phi on the faces of the fine box that are not domain faces */
static void cf_interp(vo_fab *pf, const vo_fab *pc, const int ellbc_f[3][2])
{
  const int *lo = pf->lo, *hi = pf->hi;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    if (ellbc_f[d][s] != VDN_BC_INT) continue;
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    const int ta = t1 < t2 ? t1 : t2, tb = t1 < t2 ? t2 : t1;
    for (int b2 = lo[tb]; b2 <= hi[tb]; b2++) for (int b1 = lo[ta]; b1 <= hi[ta]; b1++) {
      int g[3], f1[3], f2[3]; g[ta] = f1[ta] = f2[ta] = b1; g[tb] = f1[tb] = f2[tb] = b2;
      g[d] = s ? hi[d] + 1 : lo[d] - 1; f1[d] = s ? hi[d] : lo[d]; f2[d] = s ? hi[d] - 1 : lo[d] + 1;
      const int P[3] = { fdiv2(g[0]), fdiv2(g[1]), fdiv2(g[2]) };
      double pcs = VF(pc, P[0], P[1], P[2], 0);
      const int tt[2] = { ta, tb };
      for (int n = 0; n < 2; n++) {
        const int t = tt[n];
        int m[3] = { P[0], P[1], P[2] }, p[3] = { P[0], P[1], P[2] }; m[t] -= 1; p[t] += 1;
        const double sg = (g[t] - 2 * P[t]) ? 0.125 : -0.125;
        pcs = pcs + sg * (VF(pc, p[0], p[1], p[2], 0) - VF(pc, m[0], m[1], m[2], 0));
      }
      VF(pf, g[0], g[1], g[2], 0) = (8.0 / 15.0) * pcs + (2.0 / 3.0) * VF(pf, f1[0], f1[1], f1[2], 0) - 0.2 * VF(pf, f2[0], f2[1], f2[2], 0);
    }
  }
}
/* res = rh - A phi on the valid cells of one level, phi's ghost layer already filled; returns the max-norm over cells
 * where mask (may be NULL) is 0 */
static double plain_residual(const vo_fab *rh, const vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], vo_fab *res)
{
  const int *lo = rh->lo, *hi = rh->hi;
  const double hi2[3] = { 1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]), 1.0 / (dx[2] * dx[2]) };
  double nrm = 0.0;
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    const double p0 = VF(phi, i, j, k, 0);
    const double ax = (VF(beta[0], i + 1, j, k, 0) * (p0 - VF(phi, i + 1, j, k, 0)) + VF(beta[0], i, j, k, 0) * (p0 - VF(phi, i - 1, j, k, 0))) * hi2[0];
    const double ay = (VF(beta[1], i, j + 1, k, 0) * (p0 - VF(phi, i, j + 1, k, 0)) + VF(beta[1], i, j, k, 0) * (p0 - VF(phi, i, j - 1, k, 0))) * hi2[1];
    const double az = (VF(beta[2], i, j, k + 1, 0) * (p0 - VF(phi, i, j, k + 1, 0)) + VF(beta[2], i, j, k, 0) * (p0 - VF(phi, i, j, k - 1, 0))) * hi2[2];
    double Ap = ax + ay + az;
    if (alpha) Ap = Ap + VF(alpha, i, j, k, 0) * p0;
    const double r = VF(rh, i, j, k, 0) - Ap;
    VF(res, i, j, k, 0) = r;
    nrm = vo_nrm_acc(nrm, r);
  }
  return nrm;
}
/* Neumann domain faces carry no flux: the plain residual above would use beta*(phi_i - ghost) with ghost = phi_i = 0 flux, and
 * Dirichlet faces beta*(phi_i - (-phi_i)) = the 2b closure -- so the closure ghosts make the plain stencil exact. */

/* flux matching: replace, in the residual of the uncovered coarse cells next to the fine box, the coarse flux through each
 * interface face by the mean of the four fine fluxes */
static void reflux_residual(vo_fab *res_c, const vo_fab *phi_c, vo_fab *beta_c[3], const double dxc[3],
                            const vo_fab *phi_f, vo_fab *beta_f[3], const double dxf[3], const int ellbc_f[3][2])
{
  const int *flo = phi_f->lo, *fhi = phi_f->hi;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    if (ellbc_f[d][s] != VDN_BC_INT) continue;
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    const int ff = s ? fhi[d] + 1 : flo[d];              /* fine face index of the interface */
    const int Fc = ff / 2;                                /* coarse face index */
    for (int B2 = flo[t2] / 2; B2 <= fhi[t2] / 2; B2++) for (int B1 = flo[t1] / 2; B1 <= fhi[t1] / 2; B1++) {
      double sum = 0.0;
      for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) {
        int q[3], m[3]; q[d] = ff; q[t1] = 2 * B1 + a; q[t2] = 2 * B2 + b; m[0] = q[0]; m[1] = q[1]; m[2] = q[2]; m[d] -= 1;
        sum = sum + VF(beta_f[d], q[0], q[1], q[2], 0) * (VF(phi_f, q[0], q[1], q[2], 0) - VF(phi_f, m[0], m[1], m[2], 0)) / dxf[d];
      }
      const double Ff = sum * 0.25;
      int Q[3], M[3]; Q[d] = Fc; Q[t1] = B1; Q[t2] = B2; M[0] = Q[0]; M[1] = Q[1]; M[2] = Q[2]; M[d] -= 1;
      const double Fcrs = VF(beta_c[d], Q[0], Q[1], Q[2], 0) * (VF(phi_c, Q[0], Q[1], Q[2], 0) - VF(phi_c, M[0], M[1], M[2], 0)) / dxc[d];
      /* r = rh + (F_hi - F_lo)/h:  lo side of the fine box: the interface is the HI face of the uncovered cell M */
      if (s == 0) VF(res_c, M[0], M[1], M[2], 0) = VF(res_c, M[0], M[1], M[2], 0) + (Ff - Fcrs) / dxc[d];
      else        VF(res_c, Q[0], Q[1], Q[2], 0) = VF(res_c, Q[0], Q[1], Q[2], 0) - (Ff - Fcrs) / dxc[d];
    }
  }
}
static int covered(const vo_fab *fine, int I, int J, int K)
{
  return I >= fine->lo[0] / 2 && I <= fine->hi[0] / 2 && J >= fine->lo[1] / 2 && J <= fine->hi[1] / 2 && K >= fine->lo[2] / 2 && K <= fine->hi[2] / 2;
}
static void fab_like(vo_fab *f, const vo_fab *like, int ng, double val)
{
  vo_fab_init(f, NULL, like->lo, like->hi, ng, like->nd, 1);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}

#define VO_MAXLEV 4
typedef const int (*ellbc_t)[3][2];
static void fill_phi_ghosts(int nlev, vo_fab **phi, ellbc_t ellbc, const int pmask[3], const int *pd)
{
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction(phi[n - 1], phi[n], 0, 1);     /* keep coarser levels consistent under finer ones */
  for (int n = 0; n < nlev; n++) phi_closure(phi[n], ellbc[n], pmask, pd + 6 * n, pd + 6 * n + 3);
  for (int n = 1; n < nlev; n++) cf_interp(phi[n], phi[n - 1], ellbc[n]);
}
/* composite residual on every level; res[n] on cells covered by level n+1 = restriction of res[n+1]; returns the composite max-norm
 * (cells of each level that are not covered by the next finer one) */
static double composite_residual(int nlev, vo_fab **rh, vo_fab **phi, vo_fab **alpha, vo_fab **beta, const double *dx, ellbc_t ellbc, const int pmask[3], const int *pd, vo_fab **res)
{
  fill_phi_ghosts(nlev, phi, ellbc, pmask, pd);
  for (int n = 0; n < nlev; n++) (void)plain_residual(rh[n], phi[n], alpha ? alpha[n] : NULL, beta + 3 * n, dx + 3 * n, res[n]);
  for (int n = 1; n < nlev; n++) reflux_residual(res[n - 1], phi[n - 1], beta + 3 * (n - 1), dx + 3 * (n - 1), phi[n], beta + 3 * n, dx + 3 * n, ellbc[n]);
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction(res[n - 1], res[n], 0, 1);
  double nrm = 0.0;
  for (int n = 0; n < nlev; n++)
    for (int k = res[n]->lo[2]; k <= res[n]->hi[2]; k++) for (int j = res[n]->lo[1]; j <= res[n]->hi[1]; j++) for (int i = res[n]->lo[0]; i <= res[n]->hi[0]; i++)
      if (n == nlev - 1 || !covered(phi[n + 1], i, j, k)) nrm = vo_nrm_acc(nrm, VF(res[n], i, j, k, 0));
  return nrm;
}
/* fine += the prolongation of the correction `src` of the next coarser level (valid cells): piecewise constant into level 1,
 * LINEAR into the levels m >= 2 -- fine cell = (p0 + px + py + pz)/4 with p0 its parent and px, py, pz the parent's neighbours on the fine
 * cell's side; a neighbour that is not a cell of the source level (beyond the coarse-fine interface or a wall) counts as the parent itself,
 * one across a periodic boundary of a level that spans the domain is the periodic image.  (Round 3.  With the constant prolongation on every
 * hop three nested levels need 20 FAC iterations where two need 10; with the linear one into the levels >= 2, 11-12.  Into level 1 the
 * constant one is kept: the linear one costs two levels one or two iterations -- measured, base 32^3 and 64^3.) */
static void prolong_add(vo_fab *fine, const vo_fab *src, int m, const int pmask[3], const int *pd)
{
  const int *plo = pd + 6 * (m - 1), *phi_ = pd + 6 * (m - 1) + 3;         /* the source level's domain */
  int wrap[3];
  for (int d = 0; d < 3; d++) wrap[d] = pmask[d] && src->lo[d] == plo[d] && src->hi[d] == phi_[d];
  for (int k = fine->lo[2]; k <= fine->hi[2]; k++) for (int j = fine->lo[1]; j <= fine->hi[1]; j++) for (int i = fine->lo[0]; i <= fine->hi[0]; i++) {
    const int q[3] = { i / 2, j / 2, k / 2 };
    double v = VF(src, q[0], q[1], q[2], 0);
    if (m >= 2) {
      const int o[3] = { (i & 1) ? 1 : -1, (j & 1) ? 1 : -1, (k & 1) ? 1 : -1 };
      double pn[3];
      for (int d = 0; d < 3; d++) {
        int t[3] = { q[0], q[1], q[2] }; t[d] += o[d];
        if (t[d] < src->lo[d]) { if (wrap[d]) t[d] = src->hi[d]; else { pn[d] = v; continue; } }
        else if (t[d] > src->hi[d]) { if (wrap[d]) t[d] = src->lo[d]; else { pn[d] = v; continue; } }
        pn[d] = VF(src, t[0], t[1], t[2], 0);
      }
      v = 0.25 * (((v + pn[0]) + pn[1]) + pn[2]);
    }
    VF(fine, i, j, k, 0) = VF(fine, i, j, k, 0) + v;
  }
}
/* ghost layer of the correction e of level n as the operator of the composite residual sees it: closure at the domain faces, periodic
 * images, and beyond the coarse-fine interface the interpolation cf_interp from the correction ec of the next coarser level */
static void fill_e_ghosts(vo_fab *e, const vo_fab *ec, const int ellbc[3][2], const int pmask[3], const int *pdlo, const int *pdhi)
{
  phi_closure(e, ellbc, pmask, pdlo, pdhi);
  if (ec) cf_interp(e, ec, ellbc);
}
/* ... and as the relaxation wants it: zero beyond the domain faces (their closure is folded into the coefficients) */
static void zero_domain_ghosts(vo_fab *e, const int ellbc[3][2])
{
  const int *lo = e->lo, *hi = e->hi;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    if (ellbc[d][s] != VDN_BC_NEU && ellbc[d][s] != VDN_BC_DIR) continue;
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    for (int b2 = lo[t2]; b2 <= hi[t2]; b2++) for (int b1 = lo[t1]; b1 <= hi[t1]; b1++) {
      int g[3]; g[t1] = b1; g[t2] = b2; g[d] = s ? hi[d] + 1 : lo[d] - 1;
      VF(e, g[0], g[1], g[2], 0) = 0.0;
    }
  }
}

/* rh, phi: [lev];  beta: [lev*3 + d];  dx: [lev*3 + d];  ellbc[lev] per box;  pd: [lev][2][3].
 * One FAC iteration is a V-cycle over the levels in correction form (round 4; rounds 2-3 applied every level's correction to phi at once
 * and formed the composite residual again after each -- five residual passes over the finest of three levels per iteration where this
 * form makes two; on two levels the iterates are the same in exact arithmetic, on three the levels below see r_n - A_n e_n under a finer
 * level instead of the restriction of the finer level's new residual):
 *   composite residual r_n on every level / test;
 *   down, n = finest..1:  e_n = 0, nu1 red-black sweeps of A_n e_n = r_n (zero beyond the interface);  t = r_n - A_n e_n with the ghost cells of
 *        e_n as the composite operator fills them (closure, interpolation from e_{n-1} = 0);  r_{n-1} := restriction of t under level n, and
 *        its flux matching next to level n with the fluxes of e_n (the operator is linear: the change of the composite residual);
 *   level 0: ONE V-cycle of the single-level multigrid, A_0 e_0 = r_0;
 *   up, n = 1..finest:  e_n += P e_{n-1};  ghost cells of e_n beyond the interface interpolated from e_{n-1} and held;  nu2 sweeps of
 *        A_n e_n = r_n;
 *   phi_n += e_n on every level.
 * beta_base (may be NULL): the face coefficients the level-0 V-cycle takes instead of beta[0..2] -- the MAC projection hands over the
 * coefficients of level 0's own density, 2/(rho_i + rho_i-1) on every face, where beta carries the edge restriction of the finer level's
 * under it: the V-cycle is a preconditioner (the composite residual is formed with beta), the FAC counts stay or drop by one (measured),
 * and the HIP side can run its density-based kernels on that level. */
int vo_ml_cc_solve(int nlev, vo_fab **rh, vo_fab **phi, vo_fab **alpha, vo_fab **beta, const double *dx, const int ellbc[][3][2], const int pmask[3], const int *pd,
                   double rel_eps, int max_iter, const vdn_params *prm, vo_fab **beta_base, vo_mgstat *st)
{
  if (nlev < 2 || nlev > VO_MAXLEV) { fprintf(stderr, "vo_ml_cc_solve: 2..%d levels\n", VO_MAXLEV); abort(); }
  /* inhomogeneous Dirichlet data: the ghost cells of the incoming phi hold the boundary-FACE values (viscsolve.f90:270); the face term
   * 2b(phi_i - phi_b)/h^2 keeps its phi_i part in the operator (closure ghost = -phi_i) and its phi_b part goes to the right-hand side,
   * in the order x-lo, x-hi, y-lo, y-hi, z-lo, z-hi (as cc_load of the single-level solver) */
  for (int n = 0; n < nlev; n++) {
    const int *lo = rh[n]->lo, *hi = rh[n]->hi;
    const double hi2[3] = { 1.0 / (dx[3 * n] * dx[3 * n]), 1.0 / (dx[3 * n + 1] * dx[3 * n + 1]), 1.0 / (dx[3 * n + 2] * dx[3 * n + 2]) };
    for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
      double r = VF(rh[n], i, j, k, 0);
      if (i == lo[0] && ellbc[n][0][0] == VDN_BC_DIR) r = r + (2.0 * VF(beta[3 * n], i, j, k, 0)) * VF(phi[n], i - 1, j, k, 0) * hi2[0];
      if (i == hi[0] && ellbc[n][0][1] == VDN_BC_DIR) r = r + (2.0 * VF(beta[3 * n], i + 1, j, k, 0)) * VF(phi[n], i + 1, j, k, 0) * hi2[0];
      if (j == lo[1] && ellbc[n][1][0] == VDN_BC_DIR) r = r + (2.0 * VF(beta[3 * n + 1], i, j, k, 0)) * VF(phi[n], i, j - 1, k, 0) * hi2[1];
      if (j == hi[1] && ellbc[n][1][1] == VDN_BC_DIR) r = r + (2.0 * VF(beta[3 * n + 1], i, j + 1, k, 0)) * VF(phi[n], i, j + 1, k, 0) * hi2[1];
      if (k == lo[2] && ellbc[n][2][0] == VDN_BC_DIR) r = r + (2.0 * VF(beta[3 * n + 2], i, j, k, 0)) * VF(phi[n], i, j, k - 1, 0) * hi2[2];
      if (k == hi[2] && ellbc[n][2][1] == VDN_BC_DIR) r = r + (2.0 * VF(beta[3 * n + 2], i, j, k + 1, 0)) * VF(phi[n], i, j, k + 1, 0) * hi2[2];
      VF(rh[n], i, j, k, 0) = r;
    }
  }
  vo_fab res[VO_MAXLEV], e[VO_MAXLEV], t[VO_MAXLEV], *rp[VO_MAXLEV];
  for (int n = 0; n < nlev; n++) { fab_like(&res[n], rh[n], 0, 0.0); fab_like(&e[n], rh[n], 1, 0.0); fab_like(&t[n], rh[n], 0, 0.0); rp[n] = &res[n]; }
  /* norm of the right-hand side over the composite grid */
  double bnorm = 0.0;
  for (int n = 0; n < nlev; n++)
    for (int k = rh[n]->lo[2]; k <= rh[n]->hi[2]; k++) for (int j = rh[n]->lo[1]; j <= rh[n]->hi[1]; j++) for (int i = rh[n]->lo[0]; i <= rh[n]->hi[0]; i++)
      if (n == nlev - 1 || !covered(rh[n + 1], i, j, k)) bnorm = vo_nrm_acc(bnorm, VF(rh[n], i, j, k, 0));
  int it = 0, conv = 0; double rn = 0.0;
  if (bnorm == 0.0) conv = 1;
  while (!conv) {
    rn = composite_residual(nlev, rh, phi, alpha, beta, dx, ellbc, pmask, pd, rp);
    if (rn <= rel_eps * bnorm) { conv = 1; break; }
    if (it >= max_iter) break;
    for (int n = 0; n < nlev; n++) memset(e[n].p, 0, sizeof(double) * vo_size(&e[n]));
    for (int n = nlev - 1; n >= 1; n--) {               /* down: pre-relaxation, then the residual the next coarser level sees */
      vo_cc_smooth_ab(&res[n], &e[n], alpha ? alpha[n] : NULL, beta + 3 * n, dx + 3 * n, ellbc[n], prm->mg_nu1);
      fill_e_ghosts(&e[n], &e[n - 1], ellbc[n], pmask, pd + 6 * n, pd + 6 * n + 3);                   /* (e[n-1] = 0 here) */
      (void)plain_residual(&res[n], &e[n], alpha ? alpha[n] : NULL, beta + 3 * n, dx + 3 * n, &t[n]);
      reflux_residual(&res[n - 1], &e[n - 1], beta + 3 * (n - 1), dx + 3 * (n - 1), &e[n], beta + 3 * n, dx + 3 * n, ellbc[n]);
      vo_ml_cc_restriction(&res[n - 1], &t[n], 0, 1);
    }
    /* coarse correction: ONE V-cycle of the single-level multigrid on the whole coarse level */
    vo_mgstat cs;
    vo_cc_solve_ab(&res[0], &e[0], alpha ? alpha[0] : NULL, beta_base ? beta_base : beta, dx, ellbc[0], 0.0, -1.0, -1, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, 0, &cs);     /* (a nested-iteration start of the FIRST correction saves no FAC iteration here: measured, 10 -> 10) */
    fill_e_ghosts(&e[0], NULL, ellbc[0], pmask, pd, pd + 3);
    for (int n = 1; n < nlev; n++) {                    /* up: the coarser correction prolonged, post-relaxation with it beyond the interface */
      prolong_add(&e[n], &e[n - 1], n, pmask, pd);
      fill_e_ghosts(&e[n], &e[n - 1], ellbc[n], pmask, pd + 6 * n, pd + 6 * n + 3);
      zero_domain_ghosts(&e[n], ellbc[n]);
      vo_cc_smooth_ab_iface(&res[n], &e[n], alpha ? alpha[n] : NULL, beta + 3 * n, dx + 3 * n, ellbc[n], prm->mg_nu2);
      if (n < nlev - 1) fill_e_ghosts(&e[n], &e[n - 1], ellbc[n], pmask, pd + 6 * n, pd + 6 * n + 3);      /* what the next finer level's interface reads */
    }
    for (int n = 0; n < nlev; n++)
      for (int k = phi[n]->lo[2]; k <= phi[n]->hi[2]; k++) for (int j = phi[n]->lo[1]; j <= phi[n]->hi[1]; j++) for (int i = phi[n]->lo[0]; i <= phi[n]->hi[0]; i++)
        VF(phi[n], i, j, k, 0) = VF(phi[n], i, j, k, 0) + VF(&e[n], i, j, k, 0);
    it++;
  }
  fill_phi_ghosts(nlev, phi, ellbc, pmask, pd);       /* leave phi with consistent ghosts for mkumac */
  if (st) { st->cycles = it; st->res0 = bnorm; st->res = rn; }
  for (int n = 0; n < nlev; n++) { free(res[n].p); free(e[n].p); free(t[n].p); }
  return conv ? 0 : 1;
}

/* macproject.f90:20-133 on nlev levels.  umac: [lev*3 + d] (ng = 1), rho: [lev] (ghosts filled), mac_rhs: [lev] */
void vo_ml_macproject(int nlev, vo_fab **umac, vo_fab **rho, vo_fab **mac_rhs, const double *dx, const vo_bc *bc, const int pmask[3], const int *pd,
                      const vdn_params *prm, vo_mgstat *st)
{
  vo_fab rh[VO_MAXLEV], phi[VO_MAXLEV], beta[3 * VO_MAXLEV], *rhp[VO_MAXLEV], *php[VO_MAXLEV], *bp[3 * VO_MAXLEV];
  int ellbc[VO_MAXLEV][3][2];
  for (int n = 0; n < nlev; n++) {
    fab_like(&rh[n], rho[n], 0, 0.0); fab_like(&phi[n], rho[n], 1, 0.0); rhp[n] = &rh[n]; php[n] = &phi[n];
    for (int d = 0; d < 3; d++) {
      int nd[3] = { 0, 0, 0 }; nd[d] = 1;
      vo_fab_init(&beta[3 * n + d], NULL, rho[n]->lo, rho[n]->hi, 0, nd, 1);
      beta[3 * n + d].p = (double *)calloc(vo_size(&beta[3 * n + d]), sizeof(double)); bp[3 * n + d] = &beta[3 * n + d];
      for (int s = 0; s < 2; s++) ellbc[n][d][s] = bc[n].ell[d][s][bc[n].press_comp];
    }
  }
  /* divumac (macproject.f90:161-206): rh = mac_rhs - div(umac) on every level, then ml_cc_restriction */
  for (int n = 0; n < nlev; n++) {
    vo_divumac(umac + 3 * n, &rh[n], dx + 3 * n);
    for (int k = rh[n].lo[2]; k <= rh[n].hi[2]; k++) for (int j = rh[n].lo[1]; j <= rh[n].hi[1]; j++) for (int i = rh[n].lo[0]; i <= rh[n].hi[0]; i++)
      VF(&rh[n], i, j, k, 0) = VF(&rh[n], i, j, k, 0) * -1.0 + VF(mac_rhs[n], i, j, k, 0);
  }
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction(&rh[n - 1], &rh[n], 0, 1);
  /* mk_mac_coeffs (macproject.f90:296-334): rho's fine ghosts come from the caller's ml_restrict_and_fill; edge restriction */
  for (int n = 0; n < nlev; n++) vo_mk_mac_coeffs(rho[n], bp + 3 * n);
  vo_fab b0[3], *b0p[3];                                /* level 0's own coefficients, before the edge restriction overwrites the covered faces */
  for (int d = 0; d < 3; d++) { b0[d] = beta[d]; b0[d].p = (double *)malloc(sizeof(double) * vo_size(&beta[d])); memcpy(b0[d].p, beta[d].p, sizeof(double) * vo_size(&beta[d])); b0p[d] = &b0[d]; }
  for (int n = nlev - 1; n >= 1; n--) for (int d = 0; d < 3; d++) vo_ml_edge_restriction(bp[3 * (n - 1) + d], bp[3 * n + d], d);
  /* The level-0 V-cycle may run on level 0's own coefficients (round 3: the density-based kernels of the single-level solver) only where they are
   * close to the edge-restricted ones.  On averaged-down data 2/(rho_i + rho_i-1) is 1 / (mean rho), the restricted beta a mean of 1 / rho: with a sharp
   * density jump the own coefficients are softer by up to the density ratio, the correction overshoots and the FAC iteration slows down (10 : 1
   * one-cell jump: 45 iterations against 12) or diverges (100 : 1).  Round 4: own coefficients only if they agree with the restricted ones to 25 % on
   * every face (the smooth profiles of the reference's inputs at production resolution), else the restricted ones (the round-2 iteration). */
  double worst = 1.0;
  for (int d = 0; d < 3; d++) { const long sz = vo_size(&beta[d]); for (long q = 0; q < sz; q++) { const double a = beta[d].p[q], b = b0[d].p[q]; const double r1 = a / b, r2 = b / a; worst = vo_nrm_acc(worst, r1 > r2 ? r1 : r2); } }
  vo_ml_cc_solve(nlev, rhp, php, NULL, bp, dx, (const int (*)[3][2])ellbc, pmask, pd, prm->mac_rel_eps, prm->mg_max_iter, prm, worst <= 1.25 ? b0p : NULL, st);
  for (int d = 0; d < 3; d++) free(b0[d].p);
  /* mkumac on every level with the solver's ghost cells, then edge restriction and the ghost faces (macproject.f90:103-119) */
  for (int n = 0; n < nlev; n++) vo_mkumac(umac + 3 * n, &phi[n], bp + 3 * n, dx + 3 * n, ellbc[n]);
  for (int n = nlev - 1; n >= 1; n--) for (int d = 0; d < 3; d++) vo_ml_edge_restriction(umac[3 * (n - 1) + d], umac[3 * n + d], d);
  for (int d = 0; d < 3; d++) level_fill_boundary(umac[d], pmask, pd, pd + 3);
  for (int n = 1; n < nlev; n++) for (int d = 0; d < 3; d++) { vo_create_umac_grown(umac[3 * n + d], umac[3 * (n - 1) + d], d); level_fill_boundary(umac[3 * n + d], pmask, pd + 6 * n, pd + 6 * n + 3); }
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); for (int d = 0; d < 3; d++) free(beta[3 * n + d].p); }
}

/* visc_solve (viscsolve.f90:19-306) on nlev levels: per velocity component the composite solve of (rho - div mu grad) u = rhs.
 * unew: [lev] (ghost cells filled: they carry the wall values), lapu / rho / mac_rhs: [lev] */
void vo_ml_visc_solve(int nlev, vo_fab **unew, vo_fab **lapu, vo_fab **rho, vo_fab **mac_rhs, const double *dx, double mu, const vo_bc *bc,
                      const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st)
{
  vo_fab rh[VO_MAXLEV], phi[VO_MAXLEV], alpha[VO_MAXLEV], beta[3 * VO_MAXLEV], *rhp[VO_MAXLEV], *php[VO_MAXLEV], *alp[VO_MAXLEV], *bp[3 * VO_MAXLEV];
  int ellbc[VO_MAXLEV][3][2];
  for (int n = 0; n < nlev; n++) {
    fab_like(&rh[n], rho[n], 0, 0.0); fab_like(&phi[n], rho[n], 1, 0.0); fab_like(&alpha[n], rho[n], 0, 0.0);
    rhp[n] = &rh[n]; php[n] = &phi[n]; alp[n] = &alpha[n];
    for (int d = 0; d < 3; d++) {
      int nd[3] = { 0, 0, 0 }; nd[d] = 1;
      vo_fab_init(&beta[3 * n + d], NULL, rho[n]->lo, rho[n]->hi, 0, nd, 1);
      long sz = vo_size(&beta[3 * n + d]);
      beta[3 * n + d].p = (double *)malloc(sizeof(double) * sz);
      for (long q = 0; q < sz; q++) beta[3 * n + d].p[q] = mu;
      bp[3 * n + d] = &beta[3 * n + d];
    }
    for (int k = rho[n]->lo[2]; k <= rho[n]->hi[2]; k++) for (int j = rho[n]->lo[1]; j <= rho[n]->hi[1]; j++) for (int i = rho[n]->lo[0]; i <= rho[n]->hi[0]; i++)
      VF(&alpha[n], i, j, k, 0) = VF(rho[n], i, j, k, 0);
  }
  const double third = 1.0 / 3.0;
  const double visc_mu_dt = (prm->diffusion_type == 1) ? 2.0 * mu : mu;
  for (int d = 0; d < 3; d++) {
    for (int n = 0; n < nlev; n++) {
      const int *lo = unew[n]->lo, *hi = unew[n]->hi;
      for (int k = lo[2] - 1; k <= hi[2] + 1; k++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
        VF(&phi[n], i, j, k, 0) = VF(unew[n], i, j, k, d);
      for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
        double r = VF(unew[n], i, j, k, d) * VF(rho[n], i, j, k, 0);
        if (prm->diffusion_type == 1) r = r + mu * VF(lapu[n], i, j, k, d);
        int p[3] = { i, j, k }, m[3] = { i, j, k }; p[d] += 1; m[d] -= 1;
        r = r + third * visc_mu_dt * (VF(mac_rhs[n], p[0], p[1], p[2], 0) - VF(mac_rhs[n], m[0], m[1], m[2], 0)) / dx[3 * n + d];
        VF(&rh[n], i, j, k, 0) = r;
      }
      for (int a = 0; a < 3; a++) for (int s = 0; s < 2; s++) ellbc[n][a][s] = bc[n].ell[a][s][d];
    }
    vo_ml_cc_solve(nlev, rhp, php, alp, bp, dx, (const int (*)[3][2])ellbc, pmask, pd, 1.e-12, prm->mg_max_iter, prm, NULL, st);
    for (int n = 0; n < nlev; n++)
      for (int k = unew[n]->lo[2]; k <= unew[n]->hi[2]; k++) for (int j = unew[n]->lo[1]; j <= unew[n]->hi[1]; j++) for (int i = unew[n]->lo[0]; i <= unew[n]->hi[0]; i++)
        VF(unew[n], i, j, k, d) = VF(&phi[n], i, j, k, 0);
  }
  vo_ml_restrict_and_fill(nlev, unew, 0, 0, 3, 0, bc, pmask, pd, prm);           /* viscsolve.f90:106 */
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); free(alpha[n].p); for (int d = 0; d < 3; d++) free(beta[3 * n + d].p); }
}

/* diff_scalar_solve (viscsolve.f90:308-515) on nlev levels: (1 - div mu grad) s = s [+ mu laps], component icomp, bc component bccomp */
void vo_ml_diff_scalar_solve(int nlev, vo_fab **snew, vo_fab **laps, const double *dx, double mu, const vo_bc *bc, const int pmask[3], const int *pd,
                             const vdn_params *prm, int icomp, int bccomp, vo_mgstat *st)
{
  vo_fab rh[VO_MAXLEV], phi[VO_MAXLEV], alpha[VO_MAXLEV], beta[3 * VO_MAXLEV], *rhp[VO_MAXLEV], *php[VO_MAXLEV], *alp[VO_MAXLEV], *bp[3 * VO_MAXLEV];
  int ellbc[VO_MAXLEV][3][2];
  for (int n = 0; n < nlev; n++) {
    fab_like(&rh[n], snew[n], 0, 0.0); fab_like(&phi[n], snew[n], 1, 0.0); fab_like(&alpha[n], snew[n], 0, 1.0);
    rh[n].nc = phi[n].nc = alpha[n].nc = 1;
    rhp[n] = &rh[n]; php[n] = &phi[n]; alp[n] = &alpha[n];
    for (int d = 0; d < 3; d++) {
      int nd[3] = { 0, 0, 0 }; nd[d] = 1;
      vo_fab_init(&beta[3 * n + d], NULL, snew[n]->lo, snew[n]->hi, 0, nd, 1);
      long sz = vo_size(&beta[3 * n + d]);
      beta[3 * n + d].p = (double *)malloc(sizeof(double) * sz);
      for (long q = 0; q < sz; q++) beta[3 * n + d].p[q] = mu;
      bp[3 * n + d] = &beta[3 * n + d];
    }
    const int *lo = snew[n]->lo, *hi = snew[n]->hi;
    for (int k = lo[2] - 1; k <= hi[2] + 1; k++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
      VF(&phi[n], i, j, k, 0) = VF(snew[n], i, j, k, icomp);
    for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
      double r = VF(snew[n], i, j, k, icomp);
      if (prm->diffusion_type == 1) r = r + mu * VF(laps[n], i, j, k, icomp);
      VF(&rh[n], i, j, k, 0) = r;
    }
    for (int a = 0; a < 3; a++) for (int sd = 0; sd < 2; sd++) ellbc[n][a][sd] = bc[n].ell[a][sd][bccomp];
  }
  vo_ml_cc_solve(nlev, rhp, php, alp, bp, dx, (const int (*)[3][2])ellbc, pmask, pd, 1.e-12, prm->mg_max_iter, prm, NULL, st);
  for (int n = 0; n < nlev; n++)
    for (int k = snew[n]->lo[2]; k <= snew[n]->hi[2]; k++) for (int j = snew[n]->lo[1]; j <= snew[n]->hi[1]; j++) for (int i = snew[n]->lo[0]; i <= snew[n]->hi[0]; i++)
      VF(snew[n], i, j, k, icomp) = VF(&phi[n], i, j, k, 0);
  vo_ml_restrict_and_fill(nlev, snew, icomp, bccomp, 1, 0, bc, pmask, pd, prm);          /* viscsolve.f90:378-381 */
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); free(alpha[n].p); for (int d = 0; d < 3; d++) free(beta[3 * n + d].p); }
}

/* hgproject.f90:17-178 with nlevs > 1 (rel tolerance 1e-11 for two levels, 1e-10 for more: hgproject.f90:115-119) */
void vo_ml_hgproject(int nlev, int proj_type, vo_fab **unew, vo_fab **uold, vo_fab **rhohalf, vo_fab **p, vo_fab **gp, const double *dx, double dt,
                     const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st)
{
  vo_fab rh[VO_MAXLEV], phi[VO_MAXLEV], gphi[VO_MAXLEV], coeffs[VO_MAXLEV], *rhp[VO_MAXLEV], *php[VO_MAXLEV], *cfp[VO_MAXLEV];
  int ellbc[VO_MAXLEV][3][2];
  int nd1[3] = { 1, 1, 1 };
  for (int n = 0; n < nlev; n++) {
    const int *lo = unew[n]->lo, *hi = unew[n]->hi;
    vo_fab_init(&rh[n], NULL, lo, hi, 1, nd1, 1);    rh[n].p = (double *)calloc(vo_size(&rh[n]), sizeof(double));
    vo_fab_init(&phi[n], NULL, lo, hi, 1, nd1, 1);   phi[n].p = (double *)calloc(vo_size(&phi[n]), sizeof(double));
    vo_fab_init(&gphi[n], NULL, lo, hi, 0, NULL, 3); gphi[n].p = (double *)calloc(vo_size(&gphi[n]), sizeof(double));
    vo_fab_init(&coeffs[n], NULL, lo, hi, 1, NULL, 1); coeffs[n].p = (double *)calloc(vo_size(&coeffs[n]), sizeof(double));
    rhp[n] = &rh[n]; php[n] = &phi[n]; cfp[n] = &coeffs[n];
    for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ellbc[n][d][s] = bc[n].ell[d][s][bc[n].press_comp];
    vo_create_uvec(unew[n], uold[n], rhohalf[n], gp[n], dt, &bc[n], proj_type);
    level_fill_boundary(unew[n], pmask, pd + 6 * n, pd + 6 * n + 3);
    for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
      VF(&coeffs[n], i, j, k, 0) = 1.0 / VF(rhohalf[n], i, j, k, 0);
    level_fill_boundary(&coeffs[n], pmask, pd + 6 * n, pd + 6 * n + 3);
  }
  /* Under a finer level the coefficient of a level is the MEAN OF THE FINE sigma (what the multigrid's own coarsening takes, round 4), not
   * 1 / (mean of the fine rho) as coeffs = 1 / rhohalf gives on averaged-down data: the level's V-cycle and relaxation precondition the fine
   * equations, and with a sharp density jump the two differ by the density ratio (one heavy child among eight: 1/mean(rho) = 8/rho_heavy against
   * mean(1/rho) = 7/8) -- the softer operator overshoots and the FAC iteration diverged for one-cell jumps of 300 : 1 and more.  Smooth fields: the
   * same to O(h^2).  The composite equations themselves read sigma only on uncovered cells and are unchanged. */
  for (int n = nlev - 1; n >= 1; n--) { vo_ml_cc_restriction(&coeffs[n - 1], &coeffs[n], 0, 1); level_fill_boundary(&coeffs[n - 1], pmask, pd + 6 * (n - 1), pd + 6 * (n - 1) + 3); }
  double rel = prm->hg_rel_eps > 0.0 ? prm->hg_rel_eps : (nlev == 2 ? 1.e-11 : 1.e-10);
  double abs_eps = -1.0;
  if (proj_type == VDN_INITIAL_PROJECTION && prm->prob_type == 4) abs_eps = 1.e-12;
  vo_ml_nd_solve(nlev, rhp, php, cfp, unew, dx, (const int (*)[3][2])ellbc, pmask, rel, abs_eps, prm->hg_max_iter, prm, st);
  for (int n = 0; n < nlev; n++) {
    vo_mkgphi(&gphi[n], &phi[n], dx + 3 * n);
    vo_hg_update(proj_type, unew[n], uold[n], gp[n], &gphi[n], rhohalf[n], p[n], &phi[n], dt);
  }
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction(gp[n - 1], gp[n], 0, 3);              /* hgproject.f90:355-357 */
  for (int n = 0; n < nlev; n++) { level_fill_boundary(gp[n], pmask, pd + 6 * n, pd + 6 * n + 3); level_fill_boundary(p[n], pmask, pd + 6 * n, pd + 6 * n + 3); }
  vo_ml_restrict_and_fill(nlev, unew, 0, 0, 3, 0, bc, pmask, pd, prm);        /* hgproject.f90:364-366 */
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); free(gphi[n].p); free(coeffs[n].p); }
}

/* advance_timestep.f90:26-170 on nlev levels (inviscid): the orchestration of oracle/vo_advance.c with level loops, ml_restrict_and_fill
 * in place of fill_boundary + physbc, the velpred tail of velpred.f90:102-122 and the multilevel projections.  S: [lev] */
static void fab_new_l(vo_fab *f, const vo_fab *like, int ng, int face_dir, int nc, double val)
{
  int nd[3] = { 0, 0, 0 }; if (face_dir >= 0) nd[face_dir] = 1;
  vo_fab_init(f, NULL, like->lo, like->hi, ng, nd, nc);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}
void vo_ml_advance_timestep(int NL, vo_state *S, const double *dx, double dt, const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm,
                            int proj_type, vo_mgstat st[2])
{
  const int dm = 3, nscal = prm->nscal;
  vo_fab mac_rhs[VO_MAXLEV], rhohalf[VO_MAXLEV], umac[3 * VO_MAXLEV], vel_force[VO_MAXLEV], scal_force[VO_MAXLEV], divu[VO_MAXLEV], sedge[3 * VO_MAXLEV], sflux[3 * VO_MAXLEV], uedge[3 * VO_MAXLEV], uflux[3 * VO_MAXLEV];
  vo_fab *mrp[VO_MAXLEV], *rhp[VO_MAXLEV], *ump[3 * VO_MAXLEV], *vfp[VO_MAXLEV], *sfp2[VO_MAXLEV], *sep[3 * VO_MAXLEV], *sfp[3 * VO_MAXLEV], *uep[3 * VO_MAXLEV], *ufp[3 * VO_MAXLEV];
  vo_fab *uoldp[VO_MAXLEV], *soldp[VO_MAXLEV], *unewp[VO_MAXLEV], *snewp[VO_MAXLEV], *gpp[VO_MAXLEV], *pp[VO_MAXLEV];
  for (int n = 0; n < NL; n++) {
    uoldp[n] = &S[n].uold; soldp[n] = &S[n].sold; unewp[n] = &S[n].unew; snewp[n] = &S[n].snew; gpp[n] = &S[n].gp; pp[n] = &S[n].p;
    fab_new_l(&mac_rhs[n], &S[n].uold, 1, -1, 1, 0.0); mrp[n] = &mac_rhs[n];
    fab_new_l(&rhohalf[n], &S[n].uold, 1, -1, dm, 0.0); rhp[n] = &rhohalf[n];
    for (int d = 0; d < 3; d++) { fab_new_l(&umac[3 * n + d], &S[n].uold, 1, d, 1, 1.e20); ump[3 * n + d] = &umac[3 * n + d]; }
    fab_new_l(&vel_force[n], &S[n].uold, 1, -1, dm, 0.0); vfp[n] = &vel_force[n];
  }
  /* lapu (advance_timestep.f90:85-93; get_explicit_diffusive_term = cc_applyop per level on the filled ghost cells, then average down) */
  const int viscous = prm->visc_coef > 0.0;
  vo_fab lapu[VO_MAXLEV], *lap[VO_MAXLEV];
  for (int n = 0; n < NL; n++) {
    fab_new_l(&lapu[n], &S[n].uold, 0, -1, dm, 0.0); lap[n] = &lapu[n];
    if (viscous) for (int c = 0; c < dm; c++) vo_explicit_diffusive_term(&lapu[n], &S[n].uold, c, c, dx + 3 * n, &bc[n]);
  }
  if (viscous) for (int n = NL - 1; n >= 1; n--) vo_ml_cc_restriction(lap[n - 1], lap[n], 0, dm);
  /* advance_premac */
  for (int n = 0; n < NL; n++) vo_mkvelforce(&vel_force[n], &S[n].ext_vel_force, &S[n].gp, &S[n].sold, viscous ? &lapu[n] : NULL, 1.0, prm);
  vo_ml_restrict_and_fill(NL, vfp, 0, bc[0].extrap_comp, dm, 1, bc, pmask, pd, prm);
  for (int n = 0; n < NL; n++) vo_velpred(&S[n].uold, ump + 3 * n, &vel_force[n], dx + 3 * n, dt, &bc[n], prm);
  for (int d = 0; d < 3; d++) level_fill_boundary(&umac[d], pmask, pd, pd + 3);
  for (int n = 1; n < NL; n++) for (int d = 0; d < 3; d++) { vo_create_umac_grown(&umac[3 * n + d], &umac[3 * (n - 1) + d], d); level_fill_boundary(&umac[3 * n + d], pmask, pd + 6 * n, pd + 6 * n + 3); }
  for (int n = NL - 1; n >= 1; n--) for (int d = 0; d < 3; d++) vo_ml_edge_restriction(&umac[3 * (n - 1) + d], &umac[3 * n + d], d);
  /* MAC projection */
  vo_ml_macproject(NL, ump, soldp, mrp, dx, bc, pmask, pd, prm, &st[0]);
  /* scalar advance */
  {
    int is_cons[VO_MAXCOMP]; is_cons[0] = 1; for (int c = 1; c < nscal; c++) is_cons[c] = 0;
    const int diffusive = prm->diff_coef > 0.0;
    vo_fab laps[VO_MAXLEV], *lsp[VO_MAXLEV];
    for (int n = 0; n < NL; n++) {                                                         /* scalar_advance.f90:80-89, then average down */
      fab_new_l(&laps[n], &S[n].uold, 0, -1, nscal, 0.0); lsp[n] = &laps[n];
      if (diffusive) for (int c = 1; c < nscal; c++) vo_explicit_diffusive_term(&laps[n], &S[n].sold, c, dm + c, dx + 3 * n, &bc[n]);
    }
    if (diffusive) for (int n = NL - 1; n >= 1; n--) vo_ml_cc_restriction(lsp[n - 1], lsp[n], 1, nscal - 1);
    for (int n = 0; n < NL; n++) {
      fab_new_l(&scal_force[n], &S[n].uold, 1, -1, nscal, 0.0); sfp2[n] = &scal_force[n];
      fab_new_l(&divu[n], &S[n].uold, 1, -1, 1, 0.0);
      for (int d = 0; d < 3; d++) { fab_new_l(&sflux[3 * n + d], &S[n].uold, 0, d, nscal, 0.0); fab_new_l(&sedge[3 * n + d], &S[n].uold, 0, d, nscal, 0.0); sfp[3 * n + d] = &sflux[3 * n + d]; sep[3 * n + d] = &sedge[3 * n + d]; }
      vo_mkscalforce(&scal_force[n], &S[n].ext_scal_force, diffusive ? &laps[n] : NULL, 1.0, prm);
    }
    vo_ml_restrict_and_fill(NL, sfp2, 0, bc[0].extrap_comp, nscal, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) {
      vo_mkflux(&S[n].sold, sep + 3 * n, sfp + 3 * n, ump + 3 * n, &scal_force[n], &divu[n], dx + 3 * n, dt, 0, is_cons, dm, &bc[n], prm);
      vo_mkscalforce(&scal_force[n], &S[n].ext_scal_force, diffusive ? &laps[n] : NULL, 0.0, prm);
    }
    vo_ml_restrict_and_fill(NL, sfp2, 0, bc[0].extrap_comp, nscal, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) vo_update(&S[n].sold, ump + 3 * n, sep + 3 * n, sfp + 3 * n, &scal_force[n], &S[n].snew, dx + 3 * n, dt, 0, is_cons);
    vo_ml_restrict_and_fill(NL, snewp, 0, dm, nscal, 0, bc, pmask, pd, prm);
    if (diffusive) {                                                                   /* scalar_advance.f90:144-162 */
      const double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->diff_coef : dt * prm->diff_coef;
      vo_mgstat sst;
      for (int c = 1; c < nscal; c++) vo_ml_diff_scalar_solve(NL, snewp, lsp, dx, visc_mu, bc, pmask, pd, prm, c, dm + c, &sst);
    }
    for (int n = 0; n < NL; n++) free(laps[n].p);
    for (int n = 0; n < NL; n++) { free(scal_force[n].p); free(divu[n].p); for (int d = 0; d < 3; d++) { free(sflux[3 * n + d].p); free(sedge[3 * n + d].p); } }
  }
  for (int n = 0; n < NL; n++) vo_make_at_halftime(&rhohalf[n], 0, &S[n].sold, &S[n].snew, 0);
  vo_ml_restrict_and_fill(NL, rhp, 0, dm + 0, 1, 0, bc, pmask, pd, prm);
  if (viscous && prm->diffusion_type == 2) for (int n = 0; n < NL; n++) memset(lapu[n].p, 0, sizeof(double) * vo_size(&lapu[n]));    /* advance_timestep.f90:116-120 */
  /* velocity advance */
  {
    int is_cons[3] = { 0, 0, 0 };
    for (int n = 0; n < NL; n++) {
      for (int d = 0; d < 3; d++) { fab_new_l(&uflux[3 * n + d], &S[n].uold, 0, d, dm, 0.0); fab_new_l(&uedge[3 * n + d], &S[n].uold, 0, d, dm, 0.0); ufp[3 * n + d] = &uflux[3 * n + d]; uep[3 * n + d] = &uedge[3 * n + d]; }
      vo_mkvelforce(&vel_force[n], &S[n].ext_vel_force, &S[n].gp, &S[n].sold, viscous ? &lapu[n] : NULL, 1.0, prm);
    }
    vo_ml_restrict_and_fill(NL, vfp, 0, bc[0].extrap_comp, dm, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) {
      vo_mkflux(&S[n].uold, uep + 3 * n, ufp + 3 * n, ump + 3 * n, &vel_force[n], &mac_rhs[n], dx + 3 * n, dt, 1, is_cons, 0, &bc[n], prm);
      vo_mkvelforce(&vel_force[n], &S[n].ext_vel_force, &S[n].gp, &rhohalf[n], viscous ? &lapu[n] : NULL, 0.0, prm);
    }
    vo_ml_restrict_and_fill(NL, vfp, 0, bc[0].extrap_comp, dm, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) vo_update(&S[n].uold, ump + 3 * n, uep + 3 * n, ufp + 3 * n, &vel_force[n], &S[n].unew, dx + 3 * n, dt, 1, is_cons);
    vo_ml_restrict_and_fill(NL, unewp, 0, 0, dm, 0, bc, pmask, pd, prm);
    if (viscous) {                                                                     /* velocity_advance.f90:103-118 */
      const double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->visc_coef : dt * prm->visc_coef;
      vo_mgstat vst;
      vo_ml_visc_solve(NL, unewp, lap, rhp, mrp, dx, visc_mu, bc, pmask, pd, prm, &vst);
    }
    for (int n = 0; n < NL; n++) for (int d = 0; d < 3; d++) { free(uflux[3 * n + d].p); free(uedge[3 * n + d].p); }
  }
  vo_ml_hgproject(NL, proj_type, unewp, uoldp, rhp, pp, gpp, dx, dt, bc, pmask, pd, prm, &st[1]);
  for (int n = 0; n < NL; n++) { free(mac_rhs[n].p); free(rhohalf[n].p); free(vel_force[n].p); free(lapu[n].p); for (int d = 0; d < 3; d++) free(umac[3 * n + d].p); }
}
