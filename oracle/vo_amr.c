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
 * Refinement ratio 2, up to four levels, every refined level a LIST OF BOXES (round 5: level arrays with a cell mask, face fields box by box -- vo.h);
 * the levels properly nested (every box, coarsened and grown by two cells, inside the next coarser level or outside the domain).  Level 0 is one box.
 * Periodic sides (round 6): level 0 -- one box that spans the domain -- wraps; a refined level must stay clear of the periodic faces (its boxes, grown by the
 * ghost width, inside the domain in that direction: require_periodic_ok), so that every value it reads beyond its own cells is a coarse-fine interface value or a
 * cell of the level, never a periodic image.  (A refined level that reaches a periodic face is held on the HIP side by translation-invariance properties only.)
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

/* ---- levels as box lists (vo.h) ---------------------------------------------------------------------------------------------------------- */
void vo_level_build(vo_level *L, int nbox, const int *boxes)
{
  L->nbox = nbox; L->boxes = boxes; L->mg = 4;
  for (int d = 0; d < 3; d++) { L->blo[d] = boxes[d]; L->bhi[d] = boxes[3 + d]; }
  for (int b = 1; b < nbox; b++) for (int d = 0; d < 3; d++) {
    if (boxes[6 * b + d] < L->blo[d]) L->blo[d] = boxes[6 * b + d];
    if (boxes[6 * b + 3 + d] > L->bhi[d]) L->bhi[d] = boxes[6 * b + 3 + d];
  }
  const int g = L->mg;
  const size_t nx = (size_t)(L->bhi[0] - L->blo[0] + 1 + 2 * g), ny = (size_t)(L->bhi[1] - L->blo[1] + 1 + 2 * g), nz = (size_t)(L->bhi[2] - L->blo[2] + 1 + 2 * g);
  L->valid = (unsigned char *)calloc(nx * ny * nz, 1);
  for (int b = 0; b < nbox; b++) {
    const int *lo = boxes + 6 * b, *hi = lo + 3;
    for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
      unsigned char *v = &L->valid[(size_t)(i - L->blo[0] + g) + nx * ((size_t)(j - L->blo[1] + g) + ny * (size_t)(k - L->blo[2] + g))];
      if (*v) { fprintf(stderr, "vo_level_build: boxes %d overlaps another box\n", b); abort(); }
      *v = 1;
    }
  }
}
void vo_level_free(vo_level *L) { free(L->valid); L->valid = NULL; }
#define lv_valid vo_lv_valid
#define lv_multi vo_lv_multi

void vo_ml_cc_restriction(vo_fab *crse, const vo_fab *fine, int icomp, int nc) { vo_ml_cc_restriction_g(crse, fine, NULL, icomp, nc); }
void vo_ml_cc_restriction_g(vo_fab *crse, const vo_fab *fine, const vo_level *Lf, int icomp, int nc)
{
  for (int c = icomp; c < icomp + nc; c++)
  for (int K = fine->lo[2] / 2; K <= fine->hi[2] / 2; K++) for (int J = fine->lo[1] / 2; J <= fine->hi[1] / 2; J++) for (int I = fine->lo[0] / 2; I <= fine->hi[0] / 2; I++) {
    if (lv_multi(Lf) && !vo_valid(Lf, 2 * I, 2 * J, 2 * K)) continue;          /* boxes are coarse-aligned: all eight children or none */
    double s = 0.0;
    for (int kk = 0; kk < 2; kk++) for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 2; ii++) s = s + VF(fine, 2 * I + ii, 2 * J + jj, 2 * K + kk, c);
    VF(crse, I, J, K, c) = s * 0.125;
  }
}
void vo_ml_edge_restriction(vo_fab *crse, const vo_fab *fine, int dir) { vo_ml_edge_restriction_g(crse, fine, NULL, dir); }
/* level arrays of a single-valued face field (the solvers' face coefficients): a coarse face is covered where a fine cell on either side of it is */
void vo_ml_edge_restriction_g(vo_fab *crse, const vo_fab *fine, const vo_level *Lf, int dir)
{
  int lo[3], hi[3];
  for (int d = 0; d < 3; d++) { lo[d] = fine->lo[d] / 2; hi[d] = fine->hi[d] / 2; }
  hi[dir] += 1;
  const int t1 = (dir + 1) % 3, t2 = (dir + 2) % 3;
  for (int K = lo[2]; K <= hi[2]; K++) for (int J = lo[1]; J <= hi[1]; J++) for (int I = lo[0]; I <= hi[0]; I++) {
    int Q[3] = { I, J, K };
    if (lv_multi(Lf)) {
      int qa[3] = { 2 * I, 2 * J, 2 * K }, qb[3] = { 2 * I, 2 * J, 2 * K }; qb[dir] -= 1;
      if (!vo_valid(Lf, qa[0], qa[1], qa[2]) && !vo_valid(Lf, qb[0], qb[1], qb[2])) continue;
    }
    double s = 0.0;
    for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) {
      int q[3]; q[dir] = 2 * Q[dir]; q[t1] = 2 * Q[t1] + a; q[t2] = 2 * Q[t2] + b;
      s = s + VF(fine, q[0], q[1], q[2], 0);
    }
    VF(crse, I, J, K, 0) = s * 0.25;
  }
}
/* every cell of the fine level array that is not a cell of the level and whose parent lies inside the coarse array's allocation (a level of several
 * boxes: also the cells between the boxes -- each is a ghost cell of the boxes within ng cells of it, with the value BoxLib puts there) */
void vo_fill_ghost_cells(vo_fab *fine, const vo_fab *crse, int icomp, int nc) { vo_fill_ghost_cells_g(fine, crse, NULL, icomp, nc); }
void vo_fill_ghost_cells_g(vo_fab *fine, const vo_fab *crse, const vo_level *Lf, int icomp, int nc)
{
  const int ng = fine->ng;
  for (int c = icomp; c < icomp + nc; c++)
  for (int k = fine->lo[2] - ng; k <= fine->hi[2] + ng; k++) for (int j = fine->lo[1] - ng; j <= fine->hi[1] + ng; j++) for (int i = fine->lo[0] - ng; i <= fine->hi[0] + ng; i++) {
    if (lv_valid(Lf, fine, i, j, k)) continue;
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
/* periodic directions: level 0 spans the domain and wraps; every refined level keeps `margin` cells (its ghost width; 4 covers every caller) between its bounding
 * box and the periodic faces.  mf[n]: any cell-centred field of level n (its lo / hi are the level's bounding box) */
static void require_periodic_ok(int nlev, vo_fab *const *mf, const int pmask[3], const int *pd, const char *who)
{
  if (!(pmask[0] || pmask[1] || pmask[2])) return;
  for (int d = 0; d < 3; d++) {
    if (!pmask[d]) continue;
    if (mf[0]->lo[d] != pd[d] || mf[0]->hi[d] != pd[3 + d]) { fprintf(stderr, "%s: level 0 must span the domain in the periodic direction %d\n", who, d); abort(); }
    for (int n = 1; n < nlev; n++)
      if (mf[n]->lo[d] - 4 < pd[6 * n + d] || mf[n]->hi[d] + 4 > pd[6 * n + 3 + d]) {
        fprintf(stderr, "%s: level %d reaches the periodic faces of direction %d (cells %d..%d of %d..%d): not supported by the oracle\n", who, n, d, mf[n]->lo[d], mf[n]->hi[d], pd[6 * n + d], pd[6 * n + 3 + d]);
        abort();
      }
  }
}
#define LEV(lev, n) ((lev) ? (lev)[n] : NULL)
/* ml_restrict_and_fill: average down, then per level the coarse-fine ghost interpolation, the same-level periodic images, the physical boundary.
 * (On a level array the same-level exchange between boxes is the identity: a ghost cell inside another box IS that box's cell.) */
void vo_ml_restrict_and_fill(int nlev, vo_fab **mf, int icomp, int bcomp, int nc, int same_boundary, const vo_bc *bc, const int pmask[3],
                             const int *pd /* [lev][2][3] */, const vdn_params *prm)
{
  vo_ml_restrict_and_fill_g(nlev, NULL, mf, icomp, bcomp, nc, same_boundary, bc, pmask, pd, prm);
}
void vo_ml_restrict_and_fill_g(int nlev, const vo_level *const *lev, vo_fab **mf, int icomp, int bcomp, int nc, int same_boundary, const vo_bc *bc, const int pmask[3],
                               const int *pd, const vdn_params *prm)
{
  require_periodic_ok(nlev, mf, pmask, pd, "vo_ml_restrict_and_fill");
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction_g(mf[n - 1], mf[n], LEV(lev, n), icomp, nc);
  for (int n = 0; n < nlev; n++) {
    if (n > 0) vo_fill_ghost_cells_g(mf[n], mf[n - 1], LEV(lev, n), icomp, nc);
    level_fill_boundary(mf[n], pmask, pd + 6 * n, pd + 6 * n + 3);
    for (int c = 0; c < nc; c++) vo_physbc(mf[n], icomp + c, same_boundary ? bcomp : bcomp + c, 1, &bc[n], prm);
  }
}

/* ---------------------------------------------------------------------------------------------------------------------
 * composite cell-centred solve
 *
 * Levels are level arrays with a cell mask (vo.h).  The value of a field beyond a cell's face is: the neighbouring cell where that is a cell of the
 * level; the solver's closure where the face is a domain face (Neumann: the cell's own value, Dirichlet: minus it -- or, in the relaxation, zero with
 * the closure folded into the face coefficient: b := 0 / 2 b); else the coarse-fine interface value, held PER CELL AND DIRECTION (gh[2 d + side]): at a
 * re-entrant corner of a union of boxes the same ghost position is reached from two directions with two different interpolated values, exactly as
 * the ghost cells of two different boxes hold them in BoxLib.
 * ------------------------------------------------------------------------------------------------------------------- */
#define VO_MAXLEV 4
typedef const int (*ellbc_t)[3][2];
typedef struct clev {
  const vo_level *L;
  vo_fab *rh, *phi, *alpha, **beta;
  const double *dx; const int (*ell)[2]; const int *pdlo, *pdhi;
  vo_fab res, e, t;
  double *gphi[6], *ge[6];          /* interface values of phi / of the correction e next to each valid cell (indexed like res) */
  long ncell;
} clev;
static inline long cidx(const clev *M, int i, int j, int k) { return vo_idx(&M->res, i, j, k, 0); }
static inline int cvalid(const clev *M, int i, int j, int k) { return lv_valid(M->L, &M->res, i, j, k); }
/* (a periodic direction has no outside: the index wraps -- only level 0 ever asks, require_periodic_ok) */
static inline int is_per(const clev *M, int d) { return M->ell[d][0] == VDN_BC_PER; }
static inline int wrap1(const clev *M, int d, int q) { const int N = M->pdhi[d] - M->pdlo[d] + 1; return q < M->pdlo[d] ? q + N : (q > M->pdhi[d] ? q - N : q); }
static inline int in_domain1(const clev *M, int d, int q) { return q >= M->pdlo[d] && q <= M->pdhi[d]; }
/* value of F beyond face (d, s) of the valid cell (i,j,k) whose own value is p0, as the OPERATOR OF THE RESIDUAL reads it */
static inline double nb_res(const clev *M, const vo_fab *F, double *const G[6], int i, int j, int k, int d, int s, double p0)
{
  int q[3] = { i, j, k }; q[d] += s ? 1 : -1;
  if (!in_domain1(M, d, q[d])) { if (is_per(M, d)) q[d] = wrap1(M, d, q[d]); else return M->ell[d][s] == VDN_BC_NEU ? p0 : -p0; }
  if (cvalid(M, q[0], q[1], q[2])) return VF(F, q[0], q[1], q[2], 0);
  return G[2 * d + s][cidx(M, i, j, k)];
}
/* ... and as the RELAXATION reads it (zero beyond a domain face) */
static inline double nb_rlx(const clev *M, const vo_fab *F, double *const G[6], int i, int j, int k, int d, int s)
{
  int q[3] = { i, j, k }; q[d] += s ? 1 : -1;
  if (!in_domain1(M, d, q[d])) { if (is_per(M, d)) q[d] = wrap1(M, d, q[d]); else return 0.0; }
  if (cvalid(M, q[0], q[1], q[2])) return VF(F, q[0], q[1], q[2], 0);
  return G[2 * d + s][cidx(M, i, j, k)];
}
/* the coarse field pc at coarse cell Q as the interpolation reads it: a cell of the coarse level, or its closure beyond a domain face */
static inline double crse_val(const clev *C, const vo_fab *pc, const int Q[3])
{
  int q[3] = { Q[0], Q[1], Q[2] }; double w = 1.0;
  for (int d = 0; d < 3; d++) {
    if (is_per(C, d)) { q[d] = wrap1(C, d, q[d]); continue; }
    if (q[d] < C->pdlo[d]) { q[d] = C->pdlo[d]; if (C->ell[d][0] != VDN_BC_NEU) w = -w; }
    else if (q[d] > C->pdhi[d]) { q[d] = C->pdhi[d]; if (C->ell[d][1] != VDN_BC_NEU) w = -w; }
  }
  return w * VF(pc, q[0], q[1], q[2], 0);
}
/* interface values of the fine field pf (level F) from the coarse field pc (level C): quadratic interpolation normal to the interface (8/15, 2/3, -1/5)
 * of the coarse parent -- moved to the ghost cell's transverse position by central differences of the coarse field, +-1/8 each -- and the two fine
 * cells inside */
static void cf_interp(const clev *F, const clev *C, const vo_fab *pf, const vo_fab *pc, double *G[6])
{
  const vo_fab *r = &F->res;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    const int ta = t1 < t2 ? t1 : t2, tb = t1 < t2 ? t2 : t1;
    for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++) {
      if (!cvalid(F, i, j, k)) continue;
      int g[3] = { i, j, k }, f2[3] = { i, j, k }; g[d] += s ? 1 : -1; f2[d] += s ? -1 : 1;
      if (!in_domain1(F, d, g[d]) || cvalid(F, g[0], g[1], g[2])) continue;
      const int P[3] = { fdiv2(g[0]), fdiv2(g[1]), fdiv2(g[2]) };
      double pcs = crse_val(C, pc, P);
      const int tt[2] = { ta, tb };
      for (int n = 0; n < 2; n++) {
        const int t = tt[n];
        int m[3] = { P[0], P[1], P[2] }, p[3] = { P[0], P[1], P[2] }; m[t] -= 1; p[t] += 1;
        const double sg = (g[t] - 2 * P[t]) ? 0.125 : -0.125;
        pcs = pcs + sg * (crse_val(C, pc, p) - crse_val(C, pc, m));
      }
      G[2 * d + s][cidx(F, i, j, k)] = (8.0 / 15.0) * pcs + (2.0 / 3.0) * VF(pf, i, j, k, 0) - 0.2 * VF(pf, f2[0], f2[1], f2[2], 0);
    }
  }
}
/* out = rhs - A F on the valid cells (A's expression order: cc_apply of vo_macproject.c); returns the max-norm */
static double plain_residual(const clev *M, const vo_fab *rhs, const vo_fab *F, double *const G[6], vo_fab *out)
{
  const vo_fab *r = &M->res; const double *dx = M->dx;
  const double hi2[3] = { 1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]), 1.0 / (dx[2] * dx[2]) };
  double nrm = 0.0;
  for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++) {
    if (!cvalid(M, i, j, k)) continue;
    const double p0 = VF(F, i, j, k, 0);
    const double ax = (VF(M->beta[0], i + 1, j, k, 0) * (p0 - nb_res(M, F, G, i, j, k, 0, 1, p0)) + VF(M->beta[0], i, j, k, 0) * (p0 - nb_res(M, F, G, i, j, k, 0, 0, p0))) * hi2[0];
    const double ay = (VF(M->beta[1], i, j + 1, k, 0) * (p0 - nb_res(M, F, G, i, j, k, 1, 1, p0)) + VF(M->beta[1], i, j, k, 0) * (p0 - nb_res(M, F, G, i, j, k, 1, 0, p0))) * hi2[1];
    const double az = (VF(M->beta[2], i, j, k + 1, 0) * (p0 - nb_res(M, F, G, i, j, k, 2, 1, p0)) + VF(M->beta[2], i, j, k, 0) * (p0 - nb_res(M, F, G, i, j, k, 2, 0, p0))) * hi2[2];
    double Ap = ax + ay + az;
    if (M->alpha) Ap = Ap + VF(M->alpha, i, j, k, 0) * p0;
    const double rr = VF(rhs, i, j, k, 0) - Ap;
    VF(out, i, j, k, 0) = rr;
    nrm = vo_nrm_acc(nrm, rr);
  }
  return nrm;
}
/* nsweeps red-black Gauss-Seidel sweeps of A F = rhs on the valid cells; interface values G held (NULL: zero); colour by the level array's index
 * (the boxes of a refined level start at even indices: the colour of the global index) */
static void relax(const clev *M, const vo_fab *rhs, vo_fab *F, double *const G[6], int nsweeps)
{
  const vo_fab *r = &M->res; const double *dx = M->dx;
  const double hi2[3] = { 1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]), 1.0 / (dx[2] * dx[2]) };
  double *zero[6]; double *Z = NULL;
  if (!G) { Z = (double *)calloc((size_t)M->ncell, sizeof(double)); for (int q = 0; q < 6; q++) zero[q] = Z; G = zero; }
  for (int s = 0; s < nsweeps; s++) for (int color = 0; color < 2; color++) {
    #pragma omp parallel for collapse(2) schedule(static)
    for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++)
      for (int i = r->lo[0] + (((j - r->lo[1]) + (k - r->lo[2]) + color) & 1); i <= r->hi[0]; i += 2) {
        if (!cvalid(M, i, j, k)) continue;
        double b[3][2];
        for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) {
          int f[3] = { i, j, k }, q[3] = { i, j, k }; f[d] += sd; q[d] += sd ? 1 : -1;
          double v = VF(M->beta[d], f[0], f[1], f[2], 0);
          if (!in_domain1(M, d, q[d])) { if (M->ell[d][sd] == VDN_BC_NEU) v = 0.0; else if (M->ell[d][sd] == VDN_BC_DIR) v = 2.0 * v; }
          b[d][sd] = v;
        }
        const double p0 = VF(F, i, j, k, 0);
        const double ax = (b[0][1] * (p0 - nb_rlx(M, F, G, i, j, k, 0, 1)) + b[0][0] * (p0 - nb_rlx(M, F, G, i, j, k, 0, 0))) * hi2[0];
        const double ay = (b[1][1] * (p0 - nb_rlx(M, F, G, i, j, k, 1, 1)) + b[1][0] * (p0 - nb_rlx(M, F, G, i, j, k, 1, 0))) * hi2[1];
        const double az = (b[2][1] * (p0 - nb_rlx(M, F, G, i, j, k, 2, 1)) + b[2][0] * (p0 - nb_rlx(M, F, G, i, j, k, 2, 0))) * hi2[2];
        double Ap = ax + ay + az;
        double diag = (b[0][1] + b[0][0]) * hi2[0] + (b[1][1] + b[1][0]) * hi2[1] + (b[2][1] + b[2][0]) * hi2[2];
        if (M->alpha) { const double a0 = VF(M->alpha, i, j, k, 0); Ap = Ap + a0 * p0; diag = diag + a0; }
        if (diag != 0.0) VF(F, i, j, k, 0) = p0 + (VF(rhs, i, j, k, 0) - Ap) / diag;
      }
  }
  free(Z);
}
static inline int covered_by(const clev *F, int I, int J, int K) { return cvalid(F, 2 * I, 2 * J, 2 * K); }
/* flux matching: in the residual of the uncovered coarse cells next to the fine level, replace the coarse flux through each interface face by the mean
 * of the four fine fluxes.  Faces in the order x-lo, x-hi, y-lo, y-hi, z-lo, z-hi of the FINE level (lo: the fine level lies on the high side). */
static void reflux_residual(const clev *C, const clev *F, vo_fab *res_c, const vo_fab *phi_c, const vo_fab *phi_f, double *const Gf[6])
{
  const vo_fab *r = &C->res;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++) {
      /* (i,j,k): the UNCOVERED coarse cell; the covered one lies beyond its hi face (s = 0) or its lo face (s = 1) */
      if (!cvalid(C, i, j, k) || covered_by(F, i, j, k)) continue;
      int cv[3] = { i, j, k }; cv[d] += s ? -1 : 1;
      if (!in_domain1(C, d, cv[d]) || !cvalid(C, cv[0], cv[1], cv[2]) || !covered_by(F, cv[0], cv[1], cv[2])) continue;
      int Q[3] = { i, j, k }; if (s == 0) Q[d] += 1;            /* the coarse face, index of its high cell */
      int M_[3] = { Q[0], Q[1], Q[2] }; M_[d] -= 1;
      double sum = 0.0;
      for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) {
        int q[3], m[3]; q[d] = 2 * Q[d]; q[t1] = 2 * Q[t1] + a; q[t2] = 2 * Q[t2] + b; m[0] = q[0]; m[1] = q[1]; m[2] = q[2]; m[d] -= 1;
        /* s = 0: q is a fine cell, m lies beyond its lo face; s = 1: m is a fine cell, q beyond its hi face */
        const double vq = s == 0 ? VF(phi_f, q[0], q[1], q[2], 0) : Gf[2 * d + 1][cidx(F, m[0], m[1], m[2])];
        const double vm = s == 0 ? Gf[2 * d + 0][cidx(F, q[0], q[1], q[2])] : VF(phi_f, m[0], m[1], m[2], 0);
        sum = sum + VF(F->beta[d], q[0], q[1], q[2], 0) * (vq - vm) / F->dx[d];
      }
      const double Ff = sum * 0.25;
      const double Fcrs = VF(C->beta[d], Q[0], Q[1], Q[2], 0) * (VF(phi_c, Q[0], Q[1], Q[2], 0) - VF(phi_c, M_[0], M_[1], M_[2], 0)) / C->dx[d];
      if (s == 0) VF(res_c, M_[0], M_[1], M_[2], 0) = VF(res_c, M_[0], M_[1], M_[2], 0) + (Ff - Fcrs) / C->dx[d];
      else        VF(res_c, Q[0], Q[1], Q[2], 0) = VF(res_c, Q[0], Q[1], Q[2], 0) - (Ff - Fcrs) / C->dx[d];
    }
  }
}
static void fab_like(vo_fab *f, const vo_fab *like, int ng, double val)
{
  vo_fab_init(f, NULL, like->lo, like->hi, ng, like->nd, 1);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}
static void fill_phi_ghosts(int nlev, clev *M)
{
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction_g(M[n - 1].phi, M[n].phi, M[n].L, 0, 1);     /* keep coarser levels consistent under finer ones */
  for (int n = 1; n < nlev; n++) cf_interp(&M[n], &M[n - 1], M[n].phi, M[n - 1].phi, M[n].gphi);
}
/* composite residual on every level; res[n] on cells covered by level n+1 = restriction of res[n+1]; returns the composite max-norm
 * (cells of each level that are not covered by the next finer one) */
static double composite_residual(int nlev, clev *M)
{
  fill_phi_ghosts(nlev, M);
  for (int n = 0; n < nlev; n++) (void)plain_residual(&M[n], M[n].rh, M[n].phi, M[n].gphi, &M[n].res);
  for (int n = 1; n < nlev; n++) reflux_residual(&M[n - 1], &M[n], &M[n - 1].res, M[n - 1].phi, M[n].phi, M[n].gphi);
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction_g(&M[n - 1].res, &M[n].res, M[n].L, 0, 1);
  double nrm = 0.0;
  for (int n = 0; n < nlev; n++) {
    const vo_fab *r = &M[n].res;
    for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++)
      if (cvalid(&M[n], i, j, k) && (n == nlev - 1 || !covered_by(&M[n + 1], i, j, k))) nrm = vo_nrm_acc(nrm, VF(r, i, j, k, 0));
  }
  return nrm;
}
/* fine += the prolongation of the correction `src` of the next coarser level (valid cells): piecewise constant into level 1,
 * LINEAR into the levels m >= 2 -- fine cell = (p0 + px + py + pz)/4 with p0 its parent and px, py, pz the parent's neighbours on the fine
 * cell's side; a neighbour that is not a cell of the source level (beyond the coarse-fine interface or a wall) counts as the parent itself.
 * (Round 3.  With the constant prolongation on every hop three nested levels need 20 FAC iterations where two need 10; with the linear one into the
 * levels >= 2, 11-12.  Into level 1 the constant one is kept: the linear one costs two levels one or two iterations -- measured, base 32^3 and 64^3.) */
static void prolong_add(const clev *F, const clev *C, vo_fab *fine, const vo_fab *src, int m)
{
  const vo_fab *r = &F->res;
  for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++) {
    if (!cvalid(F, i, j, k)) continue;
    const int q[3] = { i / 2, j / 2, k / 2 };
    double v = VF(src, q[0], q[1], q[2], 0);
    if (m >= 2) {
      const int o[3] = { (i & 1) ? 1 : -1, (j & 1) ? 1 : -1, (k & 1) ? 1 : -1 };
      double pn[3];
      for (int d = 0; d < 3; d++) {
        int t[3] = { q[0], q[1], q[2] }; t[d] += o[d];
        pn[d] = (in_domain1(C, d, t[d]) && cvalid(C, t[0], t[1], t[2])) ? VF(src, t[0], t[1], t[2], 0) : v;
      }
      v = 0.25 * (((v + pn[0]) + pn[1]) + pn[2]);
    }
    VF(fine, i, j, k, 0) = VF(fine, i, j, k, 0) + v;
  }
}

/* rh, phi: [lev];  beta: [lev*3 + d];  dx: [lev*3 + d];  ellbc[lev]: of the level's bounding box;  pd: [lev][2][3].
 * One FAC iteration is a V-cycle over the levels in correction form (round 4; rounds 2-3 applied every level's correction to phi at once
 * and formed the composite residual again after each -- five residual passes over the finest of three levels per iteration where this
 * form makes two; on two levels the iterates are the same in exact arithmetic, on three the levels below see r_n - A_n e_n under a finer
 * level instead of the restriction of the finer level's new residual):
 *   composite residual r_n on every level / test;
 *   down, n = finest..1:  e_n = 0, nu1 red-black sweeps of A_n e_n = r_n (zero beyond the interface);  t = r_n - A_n e_n with the values of
 *        e_n beyond the interface as the composite operator takes them (interpolation from e_{n-1} = 0);  r_{n-1} := restriction of t under level n, and
 *        its flux matching next to level n with the fluxes of e_n (the operator is linear: the change of the composite residual);
 *   level 0: ONE V-cycle of the single-level multigrid, A_0 e_0 = r_0;
 *   up, n = 1..finest:  e_n += P e_{n-1};  the values of e_n beyond the interface interpolated from e_{n-1} and held;  nu2 sweeps of A_n e_n = r_n;
 *   phi_n += e_n on every level.
 * beta_base (may be NULL): the face coefficients the level-0 V-cycle takes instead of beta[0..2] -- the MAC projection hands over the
 * coefficients of level 0's own density, 2/(rho_i + rho_i-1) on every face, where beta carries the edge restriction of the finer level's
 * under it: the V-cycle is a preconditioner (the composite residual is formed with beta), the FAC counts stay or drop by one (measured),
 * and the HIP side can run its density-based kernels on that level. */
int vo_ml_cc_solve(int nlev, vo_fab **rh, vo_fab **phi, vo_fab **alpha, vo_fab **beta, const double *dx, const int ellbc[][3][2], const int pmask[3], const int *pd,
                   double rel_eps, int max_iter, const vdn_params *prm, vo_fab **beta_base, vo_mgstat *st)
{
  return vo_ml_cc_solve_g(nlev, NULL, rh, phi, alpha, beta, dx, ellbc, pmask, pd, rel_eps, max_iter, prm, beta_base, st, NULL);
}
int vo_ml_cc_solve_g(int nlev, const vo_level *const *lev, vo_fab **rh, vo_fab **phi, vo_fab **alpha, vo_fab **beta, const double *dx, const int ellbc[][3][2], const int pmask[3],
                     const int *pd, double rel_eps, int max_iter, const vdn_params *prm, vo_fab **beta_base, vo_mgstat *st, double **ghost)
{
  if (nlev < 2 || nlev > VO_MAXLEV) { fprintf(stderr, "vo_ml_cc_solve: 2..%d levels\n", VO_MAXLEV); abort(); }
  require_periodic_ok(nlev, rh, pmask, pd, "vo_ml_cc_solve");
  clev M[VO_MAXLEV];
  for (int n = 0; n < nlev; n++) {
    clev *m = &M[n];
    m->L = LEV(lev, n); m->rh = rh[n]; m->phi = phi[n]; m->alpha = alpha ? alpha[n] : NULL; m->beta = beta + 3 * n; m->dx = dx + 3 * n;
    m->ell = ellbc[n]; m->pdlo = pd + 6 * n; m->pdhi = pd + 6 * n + 3;
    fab_like(&m->res, rh[n], 0, 0.0); fab_like(&m->e, rh[n], 1, 0.0); fab_like(&m->t, rh[n], 0, 0.0);
    m->ncell = vo_size(&m->res);
    for (int q = 0; q < 6; q++) { m->gphi[q] = (double *)calloc((size_t)m->ncell, sizeof(double)); m->ge[q] = (double *)calloc((size_t)m->ncell, sizeof(double)); }
  }
  /* inhomogeneous Dirichlet data: the ghost cells of the incoming phi hold the boundary-FACE values (viscsolve.f90:270); the face term
   * 2b(phi_i - phi_b)/h^2 keeps its phi_i part in the operator (closure: -phi_i beyond the face) and its phi_b part goes to the right-hand side,
   * in the order x-lo, x-hi, y-lo, y-hi, z-lo, z-hi (as cc_load of the single-level solver) */
  for (int n = 0; n < nlev; n++) {
    const clev *m = &M[n]; const vo_fab *r = &m->res;
    const double hi2[3] = { 1.0 / (dx[3 * n] * dx[3 * n]), 1.0 / (dx[3 * n + 1] * dx[3 * n + 1]), 1.0 / (dx[3 * n + 2] * dx[3 * n + 2]) };
    for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++) {
      if (!cvalid(m, i, j, k)) continue;
      double v = VF(rh[n], i, j, k, 0);
      if (i == m->pdlo[0] && m->ell[0][0] == VDN_BC_DIR) v = v + (2.0 * VF(beta[3 * n], i, j, k, 0)) * VF(phi[n], i - 1, j, k, 0) * hi2[0];
      if (i == m->pdhi[0] && m->ell[0][1] == VDN_BC_DIR) v = v + (2.0 * VF(beta[3 * n], i + 1, j, k, 0)) * VF(phi[n], i + 1, j, k, 0) * hi2[0];
      if (j == m->pdlo[1] && m->ell[1][0] == VDN_BC_DIR) v = v + (2.0 * VF(beta[3 * n + 1], i, j, k, 0)) * VF(phi[n], i, j - 1, k, 0) * hi2[1];
      if (j == m->pdhi[1] && m->ell[1][1] == VDN_BC_DIR) v = v + (2.0 * VF(beta[3 * n + 1], i, j + 1, k, 0)) * VF(phi[n], i, j + 1, k, 0) * hi2[1];
      if (k == m->pdlo[2] && m->ell[2][0] == VDN_BC_DIR) v = v + (2.0 * VF(beta[3 * n + 2], i, j, k, 0)) * VF(phi[n], i, j, k - 1, 0) * hi2[2];
      if (k == m->pdhi[2] && m->ell[2][1] == VDN_BC_DIR) v = v + (2.0 * VF(beta[3 * n + 2], i, j, k + 1, 0)) * VF(phi[n], i, j, k + 1, 0) * hi2[2];
      VF(rh[n], i, j, k, 0) = v;
    }
  }
  /* norm of the right-hand side over the composite grid */
  double bnorm = 0.0;
  for (int n = 0; n < nlev; n++) {
    const vo_fab *r = &M[n].res;
    for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++)
      if (cvalid(&M[n], i, j, k) && (n == nlev - 1 || !covered_by(&M[n + 1], i, j, k))) bnorm = vo_nrm_acc(bnorm, VF(rh[n], i, j, k, 0));
  }
  /* solvability of a singular system (no Dirichlet face, no alpha): velpred's per-box dead band (velpred.f90:215-226) can leave the two copies of a face shared by two
   * boxes O(1e-9) apart, and div(umac) then does not sum to zero; a mean defect above 1e-11 of the norm is subtracted (varden_amd/csrc/amr.hip: composite_mean -- the
   * same rule; the summation order differs, the test tolerances cover it) */
  {
    int singular = alpha == NULL;
    for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) if (ellbc[0][d][sd] == VDN_BC_DIR) singular = 0;
    if (singular && bnorm > 0.0) {
      double total = 0.0;
      for (int n = 0; n < nlev; n++) {
        const vo_fab *r = &M[n].res; double lev = 0.0;
        for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++)
          if (cvalid(&M[n], i, j, k) && (n == nlev - 1 || !covered_by(&M[n + 1], i, j, k))) lev = lev + VF(rh[n], i, j, k, 0);
        total = total + lev * (dx[3 * n] * dx[3 * n + 1] * dx[3 * n + 2]);
      }
      double vol = dx[0] * dx[1] * dx[2];
      for (int d = 0; d < 3; d++) vol = vol * (double)(pd[3 + d] - pd[d] + 1);
      const double mean = total / vol;
      if (fabs(mean) > 1.e-11 * bnorm)
        for (int n = 0; n < nlev; n++) {
          const vo_fab *r = &M[n].res;
          for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++)
            if (cvalid(&M[n], i, j, k)) VF(rh[n], i, j, k, 0) = VF(rh[n], i, j, k, 0) - mean;
        }
    }
  }
  int it = 0, conv = 0; double rn = 0.0;
  if (bnorm == 0.0) conv = 1;
  while (!conv) {
    rn = composite_residual(nlev, M);
    if (rn <= rel_eps * bnorm) { conv = 1; break; }
    if (it >= max_iter) break;
    for (int n = 0; n < nlev; n++) memset(M[n].e.p, 0, sizeof(double) * vo_size(&M[n].e));
    for (int n = nlev - 1; n >= 1; n--) {               /* down: pre-relaxation, then the residual the next coarser level sees */
      relax(&M[n], &M[n].res, &M[n].e, NULL, prm->mg_nu1);
      cf_interp(&M[n], &M[n - 1], &M[n].e, &M[n - 1].e, M[n].ge);                                       /* (e[n-1] = 0 here) */
      (void)plain_residual(&M[n], &M[n].res, &M[n].e, M[n].ge, &M[n].t);
      reflux_residual(&M[n - 1], &M[n], &M[n - 1].res, &M[n - 1].e, &M[n].e, M[n].ge);
      vo_ml_cc_restriction_g(&M[n - 1].res, &M[n].t, M[n].L, 0, 1);
    }
    /* coarse correction: ONE V-cycle of the single-level multigrid on the whole coarse level */
    vo_mgstat cs;
    vo_cc_solve_ab(&M[0].res, &M[0].e, M[0].alpha, beta_base ? beta_base : beta, dx, ellbc[0], 0.0, -1.0, -1, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, 0, &cs);     /* (a nested-iteration start of the FIRST correction saves no FAC iteration here: measured, 10 -> 10) */
    for (int n = 1; n < nlev; n++) {                    /* up: the coarser correction prolonged, post-relaxation with it beyond the interface */
      prolong_add(&M[n], &M[n - 1], &M[n].e, &M[n - 1].e, n);
      cf_interp(&M[n], &M[n - 1], &M[n].e, &M[n - 1].e, M[n].ge);
      relax(&M[n], &M[n].res, &M[n].e, M[n].ge, prm->mg_nu2);
    }
    for (int n = 0; n < nlev; n++) {
      const vo_fab *r = &M[n].res;
      for (int k = r->lo[2]; k <= r->hi[2]; k++) for (int j = r->lo[1]; j <= r->hi[1]; j++) for (int i = r->lo[0]; i <= r->hi[0]; i++)
        if (cvalid(&M[n], i, j, k)) VF(phi[n], i, j, k, 0) = VF(phi[n], i, j, k, 0) + VF(&M[n].e, i, j, k, 0);
    }
    it++;
  }
  fill_phi_ghosts(nlev, M);                            /* phi consistent under the finer levels, interface values for mkumac */
  if (ghost) for (int n = 0; n < nlev; n++) for (int q = 0; q < 6; q++) { ghost[6 * n + q] = M[n].gphi[q]; M[n].gphi[q] = NULL; }
  if (st) { st->cycles = it; st->res0 = bnorm; st->res = rn; }
  for (int n = 0; n < nlev; n++) { free(M[n].res.p); free(M[n].e.p); free(M[n].t.p); for (int q = 0; q < 6; q++) { free(M[n].gphi[q]); free(M[n].ge[q]); } }
  return conv ? 0 : 1;
}

/* ---- fields held box by box (vo_bmf) ---------------------------------------------------------------------------------------------------- */
static int level_nbox(const vo_level *L) { return lv_multi(L) ? L->nbox : 1; }
static void level_box(const vo_level *L, const vo_fab *like, int b, int lo[3], int hi[3])
{
  for (int d = 0; d < 3; d++) { lo[d] = lv_multi(L) ? L->boxes[6 * b + d] : like->lo[d]; hi[d] = lv_multi(L) ? L->boxes[6 * b + 3 + d] : like->hi[d]; }
}
static void fab_new(vo_fab *f, const int *lo, const int *hi, int ng, int face_dir, int nc, double val)
{
  int nd[3] = { 0, 0, 0 }; if (face_dir >= 0) nd[face_dir] = 1;
  vo_fab_init(f, NULL, lo, hi, ng, nd, nc);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}
static void bmf_new(vo_bmf *m, const vo_level *L, const vo_fab *like, int ng, int face_dir, int nc, double val)
{
  m->nbox = level_nbox(L); m->f = (vo_fab *)malloc(sizeof(vo_fab) * (size_t)m->nbox);
  for (int b = 0; b < m->nbox; b++) { int lo[3], hi[3]; level_box(L, like, b, lo, hi); fab_new(&m->f[b], lo, hi, ng, face_dir, nc, val); }
}
static void bmf_free(vo_bmf *m) { for (int b = 0; b < m->nbox; b++) free(m->f[b].p); free(m->f); m->f = NULL; m->nbox = 0; }
static inline int in_valid(const vo_fab *f, int i, int j, int k)
{
  return i >= f->lo[0] && i <= f->hi[0] + f->nd[0] && j >= f->lo[1] && j <= f->hi[1] + f->nd[1] && k >= f->lo[2] && k <= f->hi[2] + f->nd[2];
}
/* the field at point (i,j,k): from the first box that holds it as a VALID point, else from the first whose allocation holds it (*ok = 0: nobody) */
static double bmf_lookup(const vo_bmf *m, int i, int j, int k, int c, int *ok)
{
  *ok = 1;
  for (int b = 0; b < m->nbox; b++) if (in_valid(&m->f[b], i, j, k)) return VF(&m->f[b], i, j, k, c);
  for (int b = 0; b < m->nbox; b++) if (in_alloc(&m->f[b], i, j, k)) return VF(&m->f[b], i, j, k, c);
  *ok = 0; return 0.0;
}
/* multifab_fill_boundary: the ghost points of every box from the valid points of the others (a point two boxes hold -- the face they share -- comes
 * from the first of them: BoxLib does not say which, and the two agree except inside the upwinding's dead band) */
static void bmf_fill_boundary(vo_bmf *m, const int pmask[3], const int pdlo[3], const int pdhi[3])
{
  if (m->nbox == 1) { level_fill_boundary(&m->f[0], pmask, pdlo, pdhi); return; }
  for (int b = 0; b < m->nbox; b++) {
    vo_fab *f = &m->f[b]; const int ng = f->ng;
    for (int c = 0; c < f->nc; c++)
    for (int k = f->lo[2] - ng; k <= f->hi[2] + f->nd[2] + ng; k++) for (int j = f->lo[1] - ng; j <= f->hi[1] + f->nd[1] + ng; j++) for (int i = f->lo[0] - ng; i <= f->hi[0] + f->nd[0] + ng; i++) {
      if (in_valid(f, i, j, k)) continue;
      for (int s = 0; s < m->nbox; s++) if (s != b && in_valid(&m->f[s], i, j, k)) { VF(f, i, j, k, c) = VF(&m->f[s], i, j, k, c); break; }
    }
  }
}
/* create_umac_grown on box lists: the ghost faces of every fine box from the coarse level's boxes */
static void bmf_umac_grown(vo_bmf *fine, const vo_bmf *crse, int dir)
{
  if (fine->nbox == 1 && crse->nbox == 1) { vo_create_umac_grown(&fine->f[0], &crse->f[0], dir); return; }
  for (int fb = 0; fb < fine->nbox; fb++) {
    vo_fab *f = &fine->f[fb]; const int ng = f->ng;
    for (int k = f->lo[2] - ng; k <= f->hi[2] + f->nd[2] + ng; k++) for (int j = f->lo[1] - ng; j <= f->hi[1] + f->nd[1] + ng; j++) for (int i = f->lo[0] - ng; i <= f->hi[0] + f->nd[0] + ng; i++) {
      if (in_valid(f, i, j, k)) continue;
      const int q[3] = { i, j, k };
      int P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) }, P2[3];
      const int odd = q[dir] - 2 * P[dir];
      P2[0] = P[0]; P2[1] = P[1]; P2[2] = P[2]; P2[dir] += 1;
      int ok1, ok2 = 1;
      const double v1 = bmf_lookup(crse, P[0], P[1], P[2], 0, &ok1), v2 = odd ? bmf_lookup(crse, P2[0], P2[1], P2[2], 0, &ok2) : 0.0;
      if (!ok1 || !ok2) continue;
      VF(f, i, j, k, 0) = odd ? 0.5 * (v1 + v2) : v1;
    }
  }
}
/* ml_edge_restriction on box lists, component c: every coarse face under the fine level = mean of the four fine faces that cover it */
static void bmf_edge_restriction(vo_bmf *crse, const vo_bmf *fine, int dir, int c)
{
  const int t1 = (dir + 1) % 3, t2 = (dir + 2) % 3;
  for (int cb = 0; cb < crse->nbox; cb++) for (int fb = 0; fb < fine->nbox; fb++) {
    vo_fab *C = &crse->f[cb]; const vo_fab *Fv = &fine->f[fb];
    int lo[3], hi[3], empty = 0;
    for (int d = 0; d < 3; d++) {
      int flo = Fv->lo[d] / 2, fhi = Fv->hi[d] / 2 + (d == dir ? 1 : 0);
      lo[d] = flo > C->lo[d] ? flo : C->lo[d]; hi[d] = fhi < C->hi[d] + C->nd[d] ? fhi : C->hi[d] + C->nd[d];
      if (lo[d] > hi[d]) empty = 1;
    }
    if (empty) continue;
    for (int K = lo[2]; K <= hi[2]; K++) for (int J = lo[1]; J <= hi[1]; J++) for (int I = lo[0]; I <= hi[0]; I++) {
      int Q[3] = { I, J, K };
      double s = 0.0;
      for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) {
        int q[3]; q[dir] = 2 * Q[dir]; q[t1] = 2 * Q[t1] + a; q[t2] = 2 * Q[t2] + b;
        s = s + VF(Fv, q[0], q[1], q[2], c);
      }
      VF(C, I, J, K, c) = s * 0.25;
    }
  }
}
/* a copy of the cells lo-ng .. hi+ng of a cell-centred level array (all components) / the box's valid cells written back */
static void box_gather(vo_fab *box, const int *lo, const int *hi, int ng, const vo_fab *levarr)
{
  fab_new(box, lo, hi, ng, -1, levarr->nc, 0.0);
  for (int c = 0; c < levarr->nc; c++)
  for (int k = lo[2] - ng; k <= hi[2] + ng; k++) for (int j = lo[1] - ng; j <= hi[1] + ng; j++) for (int i = lo[0] - ng; i <= hi[0] + ng; i++)
    if (in_alloc(levarr, i, j, k)) VF(box, i, j, k, c) = VF(levarr, i, j, k, c);
}
static void box_scatter(vo_fab *levarr, const vo_fab *box, int c0, int nc)
{
  for (int c = c0; c < c0 + nc; c++)
  for (int k = box->lo[2]; k <= box->hi[2]; k++) for (int j = box->lo[1]; j <= box->hi[1]; j++) for (int i = box->lo[0]; i <= box->hi[0]; i++)
    VF(levarr, i, j, k, c) = VF(box, i, j, k, c);
}
/* the bc tables of one box of a level: the level's physical boundary where the box touches it, INTERIOR elsewhere (define_bc_tower.f90:199-335 per box) */
static void box_bc(vo_bc *out, const vo_bc *levbc, const int *lo, const int *hi, const int *pdlo, const int *pdhi, const vdn_params *prm)
{
  int phys[3][2];
  for (int d = 0; d < 3; d++) { phys[d][0] = lo[d] == pdlo[d] ? levbc->phys[d][0] : VDN_INTERIOR; phys[d][1] = hi[d] == pdhi[d] ? levbc->phys[d][1] : VDN_INTERIOR; }
  vo_bc_build(out, phys, 3, prm->nscal);
}
/* run `body` for every box of level n with the level's cell-centred arrays seen as that box's fabs.  One box: the level arrays themselves. */
#define FOR_BOXES(L, like, b, lo, hi) for (int b = 0, nb_ = level_nbox(L); b < nb_; b++) for (int lo[3], hi[3], once_ = (level_box(L, like, b, lo, hi), 1); once_; once_ = 0)

/* macproject.f90:20-133 on nlev levels.  umac: [lev*3 + d], per box (ng = 1); rho: [lev] (ghost cells filled); mac_rhs: [lev] */
void vo_ml_macproject(int nlev, vo_fab **umac, vo_fab **rho, vo_fab **mac_rhs, const double *dx, const vo_bc *bc, const int pmask[3], const int *pd,
                      const vdn_params *prm, vo_mgstat *st)
{
  vo_bmf um[3 * VO_MAXLEV];
  for (int q = 0; q < 3 * nlev; q++) { um[q].nbox = 1; um[q].f = umac[q]; }
  vo_ml_macproject_g(nlev, NULL, um, rho, mac_rhs, dx, bc, pmask, pd, prm, st);
}
void vo_ml_macproject_g(int nlev, const vo_level *const *lev, vo_bmf *umac, vo_fab **rho, vo_fab **mac_rhs, const double *dx, const vo_bc *bc, const int pmask[3], const int *pd,
                        const vdn_params *prm, vo_mgstat *st)
{
  require_periodic_ok(nlev, rho, pmask, pd, "vo_ml_macproject");
  vo_fab rh[VO_MAXLEV], phi[VO_MAXLEV], beta[3 * VO_MAXLEV], *rhp[VO_MAXLEV], *php[VO_MAXLEV], *bp[3 * VO_MAXLEV];
  int ellbc[VO_MAXLEV][3][2];
  for (int n = 0; n < nlev; n++) {
    fab_like(&rh[n], rho[n], 0, 0.0); fab_like(&phi[n], rho[n], 1, 0.0); rhp[n] = &rh[n]; php[n] = &phi[n];
    rh[n].nc = phi[n].nc = 1;
    for (int d = 0; d < 3; d++) {
      int nd[3] = { 0, 0, 0 }; nd[d] = 1;
      vo_fab_init(&beta[3 * n + d], NULL, rho[n]->lo, rho[n]->hi, 0, nd, 1);
      beta[3 * n + d].p = (double *)calloc(vo_size(&beta[3 * n + d]), sizeof(double)); bp[3 * n + d] = &beta[3 * n + d];
      for (int s = 0; s < 2; s++) ellbc[n][d][s] = bc[n].ell[d][s][bc[n].press_comp];
    }
  }
  /* divumac (macproject.f90:161-206): rh = mac_rhs - div(umac) box by box, then ml_cc_restriction */
  for (int n = 0; n < nlev; n++) {
    FOR_BOXES(LEV(lev, n), rho[n], b, lo, hi) {
      vo_fab r; fab_new(&r, lo, hi, 0, -1, 1, 0.0);
      vo_fab *u3[3] = { &umac[3 * n].f[b], &umac[3 * n + 1].f[b], &umac[3 * n + 2].f[b] };
      vo_divumac(u3, &r, dx + 3 * n);
      for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
        VF(&rh[n], i, j, k, 0) = VF(&r, i, j, k, 0) * -1.0 + VF(mac_rhs[n], i, j, k, 0);
      free(r.p);
    }
  }
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction_g(&rh[n - 1], &rh[n], LEV(lev, n), 0, 1);
  /* mk_mac_coeffs (macproject.f90:296-334): rho's fine ghosts come from the caller's ml_restrict_and_fill; edge restriction */
  for (int n = 0; n < nlev; n++) vo_mk_mac_coeffs(rho[n], bp + 3 * n);
  vo_fab b0[3], *b0p[3];                                /* level 0's own coefficients, before the edge restriction overwrites the covered faces */
  for (int d = 0; d < 3; d++) { b0[d] = beta[d]; b0[d].p = (double *)malloc(sizeof(double) * vo_size(&beta[d])); memcpy(b0[d].p, beta[d].p, sizeof(double) * vo_size(&beta[d])); b0p[d] = &b0[d]; }
  for (int n = nlev - 1; n >= 1; n--) for (int d = 0; d < 3; d++) vo_ml_edge_restriction_g(bp[3 * (n - 1) + d], bp[3 * n + d], LEV(lev, n), d);
  /* The level-0 V-cycle may run on level 0's own coefficients (round 3: the density-based kernels of the single-level solver) only where they are
   * close to the edge-restricted ones.  On averaged-down data 2/(rho_i + rho_i-1) is 1 / (mean rho), the restricted beta a mean of 1 / rho: with a sharp
   * density jump the own coefficients are softer by up to the density ratio, the correction overshoots and the FAC iteration slows down (10 : 1
   * one-cell jump: 45 iterations against 12) or diverges (100 : 1).  Round 4: own coefficients only if they agree with the restricted ones to 25 % on
   * every face (the smooth profiles of the reference's inputs at production resolution), else the restricted ones (the round-2 iteration). */
  double worst = 1.0;
  for (int d = 0; d < 3; d++) { const long sz = vo_size(&beta[d]); for (long q = 0; q < sz; q++) { const double a = beta[d].p[q], b = b0[d].p[q]; const double r1 = a / b, r2 = b / a; worst = vo_nrm_acc(worst, r1 > r2 ? r1 : r2); } }
  double *gh[6 * VO_MAXLEV];
  vo_ml_cc_solve_g(nlev, lev, rhp, php, NULL, bp, dx, (const int (*)[3][2])ellbc, pmask, pd, prm->mac_rel_eps, prm->mg_max_iter, prm, worst <= 1.25 ? b0p : NULL, st, gh);
  for (int d = 0; d < 3; d++) free(b0[d].p);
  /* mkumac box by box (macproject.f90:578-645): u -= beta (phi_hi - phi_lo)/dx with phi beyond a box face = the neighbouring box's cell, the solver's
   * closure at a domain face (Neumann: the face keeps its velocity), or the interface value of the solve */
  for (int n = 0; n < nlev; n++) {
    const int *pdlo = pd + 6 * n, *pdhi = pd + 6 * n + 3;
    const vo_level *L = LEV(lev, n);
    FOR_BOXES(L, rho[n], b, lo, hi) {
      for (int d = 0; d < 3; d++) {
        vo_fab *u = &umac[3 * n + d].f[b];
        int rhi[3] = { hi[0], hi[1], hi[2] }; rhi[d] += 1;
        for (int k = lo[2]; k <= rhi[2]; k++) for (int j = lo[1]; j <= rhi[1]; j++) for (int i = lo[0]; i <= rhi[0]; i++) {
          int q[3] = { i, j, k }, m[3] = { i, j, k }; m[d] -= 1;
          double vq, vm;
          if (q[d] > pdhi[d] && ellbc[n][d][1] == VDN_BC_PER) { int w[3] = { q[0], q[1], q[2] }; w[d] = pdlo[d]; vm = VF(&phi[n], m[0], m[1], m[2], 0); vq = VF(&phi[n], w[0], w[1], w[2], 0); }
          else if (m[d] < pdlo[d] && ellbc[n][d][0] == VDN_BC_PER) { int w[3] = { m[0], m[1], m[2] }; w[d] = pdhi[d]; vq = VF(&phi[n], q[0], q[1], q[2], 0); vm = VF(&phi[n], w[0], w[1], w[2], 0); }
          else if (q[d] > pdhi[d]) { if (ellbc[n][d][1] == VDN_BC_NEU) continue; vm = VF(&phi[n], m[0], m[1], m[2], 0); vq = -vm; }
          else if (m[d] < pdlo[d]) { if (ellbc[n][d][0] == VDN_BC_NEU) continue; vq = VF(&phi[n], q[0], q[1], q[2], 0); vm = -vq; }
          else {
            const int qv = lv_valid(L, &rh[n], q[0], q[1], q[2]), mv = lv_valid(L, &rh[n], m[0], m[1], m[2]);
            vq = qv ? VF(&phi[n], q[0], q[1], q[2], 0) : gh[6 * n + 2 * d + 1][vo_idx(&rh[n], m[0], m[1], m[2], 0)];
            vm = mv ? VF(&phi[n], m[0], m[1], m[2], 0) : gh[6 * n + 2 * d + 0][vo_idx(&rh[n], q[0], q[1], q[2], 0)];
          }
          const double gphi = (vq - vm) / dx[3 * n + d];
          VF(u, i, j, k, 0) = VF(u, i, j, k, 0) - VF(&beta[3 * n + d], i, j, k, 0) * gphi;
        }
      }
    }
  }
  for (int n = 0; n < nlev; n++) for (int q = 0; q < 6; q++) free(gh[6 * n + q]);
  /* edge restriction and the ghost faces (macproject.f90:103-119) */
  for (int n = nlev - 1; n >= 1; n--) for (int d = 0; d < 3; d++) bmf_edge_restriction(&umac[3 * (n - 1) + d], &umac[3 * n + d], d, 0);
  for (int d = 0; d < 3; d++) bmf_fill_boundary(&umac[d], pmask, pd, pd + 3);
  for (int n = 1; n < nlev; n++) for (int d = 0; d < 3; d++) { bmf_umac_grown(&umac[3 * n + d], &umac[3 * (n - 1) + d], d); bmf_fill_boundary(&umac[3 * n + d], pmask, pd + 6 * n, pd + 6 * n + 3); }
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); for (int d = 0; d < 3; d++) free(beta[3 * n + d].p); }
}

/* visc_solve (viscsolve.f90:19-306) on nlev levels: per velocity component the composite solve of (rho - div mu grad) u = rhs.
 * unew: [lev] (ghost cells filled: they carry the wall values), lapu / rho / mac_rhs: [lev] */
void vo_ml_visc_solve(int nlev, vo_fab **unew, vo_fab **lapu, vo_fab **rho, vo_fab **mac_rhs, const double *dx, double mu, const vo_bc *bc,
                      const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st)
{
  vo_ml_visc_solve_g(nlev, NULL, unew, lapu, rho, mac_rhs, dx, mu, bc, pmask, pd, prm, st);
}
void vo_ml_visc_solve_g(int nlev, const vo_level *const *lev, vo_fab **unew, vo_fab **lapu, vo_fab **rho, vo_fab **mac_rhs, const double *dx, double mu, const vo_bc *bc,
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
    vo_ml_cc_solve_g(nlev, lev, rhp, php, alp, bp, dx, (const int (*)[3][2])ellbc, pmask, pd, 1.e-12, prm->mg_max_iter, prm, NULL, st, NULL);
    for (int n = 0; n < nlev; n++)
      for (int k = unew[n]->lo[2]; k <= unew[n]->hi[2]; k++) for (int j = unew[n]->lo[1]; j <= unew[n]->hi[1]; j++) for (int i = unew[n]->lo[0]; i <= unew[n]->hi[0]; i++)
        if (lv_valid(LEV(lev, n), unew[n], i, j, k)) VF(unew[n], i, j, k, d) = VF(&phi[n], i, j, k, 0);
  }
  vo_ml_restrict_and_fill_g(nlev, lev, unew, 0, 0, 3, 0, bc, pmask, pd, prm);           /* viscsolve.f90:106 */
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); free(alpha[n].p); for (int d = 0; d < 3; d++) free(beta[3 * n + d].p); }
}

/* diff_scalar_solve (viscsolve.f90:308-515) on nlev levels: (1 - div mu grad) s = s [+ mu laps], component icomp, bc component bccomp */
void vo_ml_diff_scalar_solve(int nlev, vo_fab **snew, vo_fab **laps, const double *dx, double mu, const vo_bc *bc, const int pmask[3], const int *pd,
                             const vdn_params *prm, int icomp, int bccomp, vo_mgstat *st)
{
  vo_ml_diff_scalar_solve_g(nlev, NULL, snew, laps, dx, mu, bc, pmask, pd, prm, icomp, bccomp, st);
}
void vo_ml_diff_scalar_solve_g(int nlev, const vo_level *const *lev, vo_fab **snew, vo_fab **laps, const double *dx, double mu, const vo_bc *bc, const int pmask[3], const int *pd,
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
  vo_ml_cc_solve_g(nlev, lev, rhp, php, alp, bp, dx, (const int (*)[3][2])ellbc, pmask, pd, 1.e-12, prm->mg_max_iter, prm, NULL, st, NULL);
  for (int n = 0; n < nlev; n++)
    for (int k = snew[n]->lo[2]; k <= snew[n]->hi[2]; k++) for (int j = snew[n]->lo[1]; j <= snew[n]->hi[1]; j++) for (int i = snew[n]->lo[0]; i <= snew[n]->hi[0]; i++)
      if (lv_valid(LEV(lev, n), snew[n], i, j, k)) VF(snew[n], i, j, k, icomp) = VF(&phi[n], i, j, k, 0);
  vo_ml_restrict_and_fill_g(nlev, lev, snew, icomp, bccomp, 1, 0, bc, pmask, pd, prm);          /* viscsolve.f90:378-381 */
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); free(alpha[n].p); for (int d = 0; d < 3; d++) free(beta[3 * n + d].p); }
}

/* hgproject.f90:17-178 with nlevs > 1 (rel tolerance 1e-11 for two levels, 1e-10 for more: hgproject.f90:115-119) */
void vo_ml_hgproject(int nlev, int proj_type, vo_fab **unew, vo_fab **uold, vo_fab **rhohalf, vo_fab **p, vo_fab **gp, const double *dx, double dt,
                     const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st)
{
  vo_ml_hgproject_g(nlev, NULL, proj_type, unew, uold, rhohalf, p, gp, dx, dt, bc, pmask, pd, prm, st);
}
void vo_ml_hgproject_g(int nlev, const vo_level *const *lev, int proj_type, vo_fab **unew, vo_fab **uold, vo_fab **rhohalf, vo_fab **p, vo_fab **gp, const double *dx, double dt,
                       const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st)
{
  require_periodic_ok(nlev, unew, pmask, pd, "vo_ml_hgproject");
  vo_fab rh[VO_MAXLEV], phi[VO_MAXLEV], gphi[VO_MAXLEV], coeffs[VO_MAXLEV], *rhp[VO_MAXLEV], *php[VO_MAXLEV], *cfp[VO_MAXLEV];
  int ellbc[VO_MAXLEV][3][2];
  int nd1[3] = { 1, 1, 1 };
  for (int n = 0; n < nlev; n++) {
    const int *lo = unew[n]->lo, *hi = unew[n]->hi;
    const vo_level *L = LEV(lev, n);
    vo_fab_init(&rh[n], NULL, lo, hi, 1, nd1, 1);    rh[n].p = (double *)calloc(vo_size(&rh[n]), sizeof(double));
    vo_fab_init(&phi[n], NULL, lo, hi, 1, nd1, 1);   phi[n].p = (double *)calloc(vo_size(&phi[n]), sizeof(double));
    vo_fab_init(&gphi[n], NULL, lo, hi, 0, NULL, 3); gphi[n].p = (double *)calloc(vo_size(&gphi[n]), sizeof(double));
    vo_fab_init(&coeffs[n], NULL, lo, hi, 1, NULL, 1); coeffs[n].p = (double *)calloc(vo_size(&coeffs[n]), sizeof(double));
    rhp[n] = &rh[n]; php[n] = &phi[n]; cfp[n] = &coeffs[n];
    for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ellbc[n][d][s] = bc[n].ell[d][s][bc[n].press_comp];
    if (!lv_multi(L)) {
      vo_create_uvec(unew[n], uold[n], rhohalf[n], gp[n], dt, &bc[n], proj_type);
      level_fill_boundary(unew[n], pmask, pd + 6 * n, pd + 6 * n + 3);
    } else {
      /* create_uvec box by box (it rewrites unew on the box and one ghost ring): the cells of the box go back; of the ring the composite right-hand side reads
       * nothing on a refined level (the velocity is masked to the level's own cells there) */
      FOR_BOXES(L, unew[n], b, blo, bhi) {
        vo_fab bu, bo, br, bg; vo_bc bb;
        box_gather(&bu, blo, bhi, unew[n]->ng, unew[n]); box_gather(&bo, blo, bhi, uold[n]->ng, uold[n]); box_gather(&br, blo, bhi, rhohalf[n]->ng, rhohalf[n]); box_gather(&bg, blo, bhi, gp[n]->ng, gp[n]);
        box_bc(&bb, &bc[n], blo, bhi, pd + 6 * n, pd + 6 * n + 3, prm);
        vo_create_uvec(&bu, &bo, &br, &bg, dt, &bb, proj_type);
        box_scatter(unew[n], &bu, 0, 3);
        free(bu.p); free(bo.p); free(br.p); free(bg.p);
      }
    }
    for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
      if (lv_valid(L, unew[n], i, j, k)) VF(&coeffs[n], i, j, k, 0) = 1.0 / VF(rhohalf[n], i, j, k, 0);
    level_fill_boundary(&coeffs[n], pmask, pd + 6 * n, pd + 6 * n + 3);
  }
  /* Under a finer level the coefficient of a level is the MEAN OF THE FINE sigma (what the multigrid's own coarsening takes, round 4), not
   * 1 / (mean of the fine rho) as coeffs = 1 / rhohalf gives on averaged-down data: the level's V-cycle and relaxation precondition the fine
   * equations, and with a sharp density jump the two differ by the density ratio (one heavy child among eight: 1/mean(rho) = 8/rho_heavy against
   * mean(1/rho) = 7/8) -- the softer operator overshoots and the FAC iteration diverged for one-cell jumps of 300 : 1 and more.  Smooth fields: the
   * same to O(h^2).  The composite equations themselves read sigma only on uncovered cells and are unchanged. */
  for (int n = nlev - 1; n >= 1; n--) { vo_ml_cc_restriction_g(&coeffs[n - 1], &coeffs[n], LEV(lev, n), 0, 1); level_fill_boundary(&coeffs[n - 1], pmask, pd + 6 * (n - 1), pd + 6 * (n - 1) + 3); }
  double rel = prm->hg_rel_eps > 0.0 ? prm->hg_rel_eps : (nlev == 2 ? 1.e-11 : 1.e-10);
  double abs_eps = -1.0;
  if (proj_type == VDN_INITIAL_PROJECTION && prm->prob_type == 4) abs_eps = 1.e-12;
  vo_ml_nd_solve_g(nlev, lev, rhp, php, cfp, unew, dx, (const int (*)[3][2])ellbc, pmask, pd, rel, abs_eps, prm->hg_max_iter, prm, st);
  for (int n = 0; n < nlev; n++) {
    const vo_level *L = LEV(lev, n);
    if (!lv_multi(L)) {
      vo_mkgphi(&gphi[n], &phi[n], dx + 3 * n);
      vo_hg_update(proj_type, unew[n], uold[n], gp[n], &gphi[n], rhohalf[n], p[n], &phi[n], dt);
    } else {
      /* initial projection / divu iterations zero gp and p everywhere (hg_update's memset): here on the whole level array once */
      if (proj_type == VDN_INITIAL_PROJECTION || proj_type == VDN_DIVU_ITERS) { memset(gp[n]->p, 0, sizeof(double) * vo_size(gp[n])); memset(p[n]->p, 0, sizeof(double) * vo_size(p[n])); }
      vo_fab pin = *p[n];                               /* the incoming pressure: boxes share nodes, and the pressure iterations ADD to it */
      pin.p = (double *)malloc(sizeof(double) * vo_size(p[n])); memcpy(pin.p, p[n]->p, sizeof(double) * vo_size(p[n]));
      FOR_BOXES(L, unew[n], b, blo, bhi) {
        vo_fab bu, bo, br, bg, bgp, bp_, bph;
        box_gather(&bu, blo, bhi, 0, unew[n]); box_gather(&bo, blo, bhi, 0, uold[n]); box_gather(&br, blo, bhi, 0, rhohalf[n]); box_gather(&bg, blo, bhi, 0, gp[n]);
        fab_new(&bgp, blo, bhi, 0, -1, 3, 0.0);
        vo_fab_init(&bp_, NULL, blo, bhi, 0, nd1, 1); bp_.p = (double *)calloc(vo_size(&bp_), sizeof(double));
        vo_fab_init(&bph, NULL, blo, bhi, 0, nd1, 1); bph.p = (double *)calloc(vo_size(&bph), sizeof(double));
        for (int k = blo[2]; k <= bhi[2] + 1; k++) for (int j = blo[1]; j <= bhi[1] + 1; j++) for (int i = blo[0]; i <= bhi[0] + 1; i++) { VF(&bph, i, j, k, 0) = VF(&phi[n], i, j, k, 0); VF(&bp_, i, j, k, 0) = VF(&pin, i, j, k, 0); }
        vo_mkgphi(&bgp, &bph, dx + 3 * n);
        vo_hg_update(proj_type, &bu, &bo, &bg, &bgp, &br, &bp_, &bph, dt);
        box_scatter(unew[n], &bu, 0, 3); box_scatter(gp[n], &bg, 0, 3);
        for (int k = blo[2]; k <= bhi[2] + 1; k++) for (int j = blo[1]; j <= bhi[1] + 1; j++) for (int i = blo[0]; i <= bhi[0] + 1; i++) VF(p[n], i, j, k, 0) = VF(&bp_, i, j, k, 0);
        free(bu.p); free(bo.p); free(br.p); free(bg.p); free(bgp.p); free(bp_.p); free(bph.p);
      }
      free(pin.p);
    }
  }
  for (int n = nlev - 1; n >= 1; n--) vo_ml_cc_restriction_g(gp[n - 1], gp[n], LEV(lev, n), 0, 3);              /* hgproject.f90:355-357 */
  for (int n = 0; n < nlev; n++) { level_fill_boundary(gp[n], pmask, pd + 6 * n, pd + 6 * n + 3); level_fill_boundary(p[n], pmask, pd + 6 * n, pd + 6 * n + 3); }
  vo_ml_restrict_and_fill_g(nlev, lev, unew, 0, 0, 3, 0, bc, pmask, pd, prm);        /* hgproject.f90:364-366 */
  for (int n = 0; n < nlev; n++) { free(rh[n].p); free(phi[n].p); free(gphi[n].p); free(coeffs[n].p); }
}

/* estdt of one level: the minimum over its boxes (estdt.f90:47-78) */
double vo_estdt_g(const vo_level *L, const vo_fab *u, const vo_fab *s, const vo_fab *gp, const vo_fab *ext, const double dx[3], double dtold, const vdn_params *prm)
{
  if (!lv_multi(L)) return vo_estdt(u, s, gp, ext, dx, dtold, prm);
  double dt = 1.e300;
  FOR_BOXES(L, u, b, lo, hi) {
    vo_fab bu, bs, bg, be;
    box_gather(&bu, lo, hi, 0, u); box_gather(&bs, lo, hi, 0, s); box_gather(&bg, lo, hi, 0, gp); box_gather(&be, lo, hi, 0, ext);
    const double d1 = vo_estdt(&bu, &bs, &bg, &be, dx, dtold, prm);
    if (d1 < dt) dt = d1;
    free(bu.p); free(bs.p); free(bg.p); free(be.p);
  }
  return dt;
}

/* advance_timestep.f90:26-170 on nlev levels: the orchestration of oracle/vo_advance.c with level loops, ml_restrict_and_fill
 * in place of fill_boundary + physbc, the velpred tail of velpred.f90:102-122, the restriction of the conservative fluxes (mkflux.f90:137-146)
 * and the multilevel projections.  S: [lev], level arrays.  The per-box kernels run on copies of each box with its ghost cells. */
static void fab_new_l(vo_fab *f, const vo_fab *like, int ng, int face_dir, int nc, double val) { fab_new(f, like->lo, like->hi, ng, face_dir, nc, val); }
void vo_ml_advance_timestep(int NL, vo_state *S, const double *dx, double dt, const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm,
                            int proj_type, vo_mgstat st[2])
{
  vo_ml_advance_timestep_g(NL, NULL, S, dx, dt, bc, pmask, pd, prm, proj_type, st);
}
/* one cell-centred kernel on every box of a level: in[] are gathered with their own ghost widths, out (components c0..c0+nc-1 of outarr) scattered back */
typedef void (*boxfn)(vo_fab *out, vo_fab **in, const vo_bc *bcb, const int *lo, const int *hi, void *ctx);
static void on_boxes(const vo_level *L, vo_fab *outarr, int c0, int nc, vo_fab **inarr, int nin, const vo_bc *levbc, const int *pdlo, const int *pdhi, const vdn_params *prm, boxfn fn, void *ctx)
{
  if (!lv_multi(L)) { fn(outarr, inarr, levbc, outarr->lo, outarr->hi, ctx); return; }
  FOR_BOXES(L, outarr, b, lo, hi) {
    vo_fab bo, bi[8], *bip[8]; vo_bc bb;
    box_gather(&bo, lo, hi, outarr->ng, outarr);
    for (int q = 0; q < nin; q++) { if (inarr[q]) { box_gather(&bi[q], lo, hi, inarr[q]->ng, inarr[q]); bip[q] = &bi[q]; } else bip[q] = NULL; }
    box_bc(&bb, levbc, lo, hi, pdlo, pdhi, prm);
    fn(&bo, bip, &bb, lo, hi, ctx);
    box_scatter(outarr, &bo, c0, nc);
    free(bo.p); for (int q = 0; q < nin; q++) if (inarr[q]) free(bi[q].p);
  }
}
typedef struct { const vdn_params *prm; double fac; const double *dx; int comp, bccomp; } kctx;
static void k_velforce(vo_fab *out, vo_fab **in, const vo_bc *bcb, const int *lo, const int *hi, void *c) { (void)bcb; (void)lo; (void)hi; kctx *k = (kctx *)c; vo_mkvelforce(out, in[0], in[1], in[2], in[3], k->fac, k->prm); }
static void k_scalforce(vo_fab *out, vo_fab **in, const vo_bc *bcb, const int *lo, const int *hi, void *c) { (void)bcb; (void)lo; (void)hi; kctx *k = (kctx *)c; vo_mkscalforce(out, in[0], in[1], k->fac, k->prm); }
static void k_lap(vo_fab *out, vo_fab **in, const vo_bc *bcb, const int *lo, const int *hi, void *c) { (void)lo; (void)hi; kctx *k = (kctx *)c; vo_explicit_diffusive_term(out, in[0], k->comp, k->bccomp, k->dx, bcb); }
static void k_halftime(vo_fab *out, vo_fab **in, const vo_bc *bcb, const int *lo, const int *hi, void *c) { (void)bcb; (void)lo; (void)hi; (void)c; vo_make_at_halftime(out, 0, in[0], in[1], 0); }

void vo_ml_advance_timestep_g(int NL, const vo_level *const *lev, vo_state *S, const double *dx, double dt, const vo_bc *bc, const int pmask[3], const int *pd,
                              const vdn_params *prm, int proj_type, vo_mgstat st[2])
{
  if ((pmask[0] || pmask[1] || pmask[2]) && lev && lv_multi(lev[0])) { fprintf(stderr, "vo_ml_advance_timestep: a periodic level 0 must be ONE box in the oracle\n"); abort(); }
  { vo_fab *chk[VO_MAXLEV] = { 0 }; for (int n = 0; n < NL; n++) chk[n] = &S[n].uold; require_periodic_ok(NL, chk, pmask, pd, "vo_ml_advance_timestep"); }
  const int dm = 3, nscal = prm->nscal;
  vo_fab mac_rhs[VO_MAXLEV], rhohalf[VO_MAXLEV], vel_force[VO_MAXLEV], scal_force[VO_MAXLEV], divu[VO_MAXLEV];
  vo_bmf umac[3 * VO_MAXLEV], sedge[3 * VO_MAXLEV], sflux[3 * VO_MAXLEV], uedge[3 * VO_MAXLEV], uflux[3 * VO_MAXLEV];
  vo_fab *mrp[VO_MAXLEV], *rhp[VO_MAXLEV], *vfp[VO_MAXLEV], *sfp2[VO_MAXLEV];
  vo_fab *uoldp[VO_MAXLEV], *soldp[VO_MAXLEV], *unewp[VO_MAXLEV], *snewp[VO_MAXLEV], *gpp[VO_MAXLEV], *pp[VO_MAXLEV];
  #define PDL(n) (pd + 6 * (n)), (pd + 6 * (n) + 3)
  for (int n = 0; n < NL; n++) {
    uoldp[n] = &S[n].uold; soldp[n] = &S[n].sold; unewp[n] = &S[n].unew; snewp[n] = &S[n].snew; gpp[n] = &S[n].gp; pp[n] = &S[n].p;
    fab_new_l(&mac_rhs[n], &S[n].uold, 1, -1, 1, 0.0); mrp[n] = &mac_rhs[n];
    fab_new_l(&rhohalf[n], &S[n].uold, 1, -1, dm, 0.0); rhp[n] = &rhohalf[n];
    for (int d = 0; d < 3; d++) bmf_new(&umac[3 * n + d], LEV(lev, n), &S[n].uold, 1, d, 1, 1.e20);
    fab_new_l(&vel_force[n], &S[n].uold, 1, -1, dm, 0.0); vfp[n] = &vel_force[n];
  }
  /* lapu (advance_timestep.f90:85-93; get_explicit_diffusive_term = cc_applyop per level on the filled ghost cells, then average down) */
  const int viscous = prm->visc_coef > 0.0;
  vo_fab lapu[VO_MAXLEV], *lap[VO_MAXLEV];
  for (int n = 0; n < NL; n++) {
    fab_new_l(&lapu[n], &S[n].uold, 0, -1, dm, 0.0); lap[n] = &lapu[n];
    if (viscous) for (int c = 0; c < dm; c++) { kctx k = { prm, 0.0, dx + 3 * n, c, c }; vo_fab *in[1] = { &S[n].uold }; on_boxes(LEV(lev, n), &lapu[n], c, 1, in, 1, &bc[n], PDL(n), prm, k_lap, &k); }
  }
  if (viscous) for (int n = NL - 1; n >= 1; n--) vo_ml_cc_restriction_g(lap[n - 1], lap[n], LEV(lev, n), 0, dm);
  /* advance_premac */
  for (int n = 0; n < NL; n++) { kctx k = { prm, 1.0, NULL, 0, 0 }; vo_fab *in[4] = { &S[n].ext_vel_force, &S[n].gp, &S[n].sold, viscous ? &lapu[n] : NULL }; on_boxes(LEV(lev, n), &vel_force[n], 0, dm, in, 4, &bc[n], PDL(n), prm, k_velforce, &k); }
  vo_ml_restrict_and_fill_g(NL, lev, vfp, 0, bc[0].extrap_comp, dm, 1, bc, pmask, pd, prm);
  for (int n = 0; n < NL; n++) {
    const vo_level *L = LEV(lev, n);
    FOR_BOXES(L, &S[n].uold, b, lo, hi) {
      vo_fab *u3[3] = { &umac[3 * n].f[b], &umac[3 * n + 1].f[b], &umac[3 * n + 2].f[b] };
      if (!lv_multi(L)) { vo_velpred(&S[n].uold, u3, &vel_force[n], dx + 3 * n, dt, &bc[n], prm); continue; }
      vo_fab bu, bf; vo_bc bb;
      box_gather(&bu, lo, hi, S[n].uold.ng, &S[n].uold); box_gather(&bf, lo, hi, 1, &vel_force[n]); box_bc(&bb, &bc[n], lo, hi, PDL(n), prm);
      vo_velpred(&bu, u3, &bf, dx + 3 * n, dt, &bb, prm);
      free(bu.p); free(bf.p);
    }
  }
  for (int d = 0; d < 3; d++) bmf_fill_boundary(&umac[d], pmask, pd, pd + 3);
  for (int n = 1; n < NL; n++) for (int d = 0; d < 3; d++) { bmf_umac_grown(&umac[3 * n + d], &umac[3 * (n - 1) + d], d); bmf_fill_boundary(&umac[3 * n + d], pmask, PDL(n)); }
  for (int n = NL - 1; n >= 1; n--) for (int d = 0; d < 3; d++) bmf_edge_restriction(&umac[3 * (n - 1) + d], &umac[3 * n + d], d, 0);
  /* MAC projection */
  vo_ml_macproject_g(NL, lev, umac, soldp, mrp, dx, bc, pmask, pd, prm, &st[0]);
  /* scalar advance */
  {
    int is_cons[VO_MAXCOMP]; is_cons[0] = 1; for (int c = 1; c < nscal; c++) is_cons[c] = 0;
    const int diffusive = prm->diff_coef > 0.0;
    vo_fab laps[VO_MAXLEV], *lsp[VO_MAXLEV];
    for (int n = 0; n < NL; n++) {                                                         /* scalar_advance.f90:80-89, then average down */
      fab_new_l(&laps[n], &S[n].uold, 0, -1, nscal, 0.0); lsp[n] = &laps[n];
      if (diffusive) for (int c = 1; c < nscal; c++) { kctx k = { prm, 0.0, dx + 3 * n, c, dm + c }; vo_fab *in[1] = { &S[n].sold }; on_boxes(LEV(lev, n), &laps[n], c, 1, in, 1, &bc[n], PDL(n), prm, k_lap, &k); }
    }
    if (diffusive) for (int n = NL - 1; n >= 1; n--) vo_ml_cc_restriction_g(lsp[n - 1], lsp[n], LEV(lev, n), 1, nscal - 1);
    for (int n = 0; n < NL; n++) {
      fab_new_l(&scal_force[n], &S[n].uold, 1, -1, nscal, 0.0); sfp2[n] = &scal_force[n];
      fab_new_l(&divu[n], &S[n].uold, 1, -1, 1, 0.0);
      for (int d = 0; d < 3; d++) { bmf_new(&sflux[3 * n + d], LEV(lev, n), &S[n].uold, 0, d, nscal, 0.0); bmf_new(&sedge[3 * n + d], LEV(lev, n), &S[n].uold, 0, d, nscal, 0.0); }
      kctx k = { prm, 1.0, NULL, 0, 0 }; vo_fab *in[2] = { &S[n].ext_scal_force, diffusive ? &laps[n] : NULL };
      on_boxes(LEV(lev, n), &scal_force[n], 0, nscal, in, 2, &bc[n], PDL(n), prm, k_scalforce, &k);
    }
    vo_ml_restrict_and_fill_g(NL, lev, sfp2, 0, bc[0].extrap_comp, nscal, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) {
      const vo_level *L = LEV(lev, n);
      FOR_BOXES(L, &S[n].uold, b, lo, hi) {
        vo_fab *u3[3] = { &umac[3 * n].f[b], &umac[3 * n + 1].f[b], &umac[3 * n + 2].f[b] }, *e3[3] = { &sedge[3 * n].f[b], &sedge[3 * n + 1].f[b], &sedge[3 * n + 2].f[b] }, *f3[3] = { &sflux[3 * n].f[b], &sflux[3 * n + 1].f[b], &sflux[3 * n + 2].f[b] };
        if (!lv_multi(L)) { vo_mkflux(&S[n].sold, e3, f3, u3, &scal_force[n], &divu[n], dx + 3 * n, dt, 0, is_cons, dm, &bc[n], prm); continue; }
        vo_fab bs, bf, bd; vo_bc bb;
        box_gather(&bs, lo, hi, S[n].sold.ng, &S[n].sold); box_gather(&bf, lo, hi, 1, &scal_force[n]); box_gather(&bd, lo, hi, 1, &divu[n]); box_bc(&bb, &bc[n], lo, hi, PDL(n), prm);
        vo_mkflux(&bs, e3, f3, u3, &bf, &bd, dx + 3 * n, dt, 0, is_cons, dm, &bb, prm);
        free(bs.p); free(bf.p); free(bd.p);
      }
      kctx k = { prm, 0.0, NULL, 0, 0 }; vo_fab *in[2] = { &S[n].ext_scal_force, diffusive ? &laps[n] : NULL };
      on_boxes(L, &scal_force[n], 0, nscal, in, 2, &bc[n], PDL(n), prm, k_scalforce, &k);
    }
    /* mkflux.f90:137-146: the fluxes of the conservative components on a coarse face under a finer level = the mean of the fine fluxes -- the coarse
     * cells next to the finer level are updated with the fine level's fluxes through the interface (round 5: missing in rounds 2-4) */
    for (int n = NL - 1; n >= 1; n--) for (int c = 0; c < nscal; c++) if (is_cons[c]) for (int d = 0; d < 3; d++) bmf_edge_restriction(&sflux[3 * (n - 1) + d], &sflux[3 * n + d], d, c);
    vo_ml_restrict_and_fill_g(NL, lev, sfp2, 0, bc[0].extrap_comp, nscal, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) {
      const vo_level *L = LEV(lev, n);
      FOR_BOXES(L, &S[n].uold, b, lo, hi) {
        vo_fab *u3[3] = { &umac[3 * n].f[b], &umac[3 * n + 1].f[b], &umac[3 * n + 2].f[b] }, *e3[3] = { &sedge[3 * n].f[b], &sedge[3 * n + 1].f[b], &sedge[3 * n + 2].f[b] }, *f3[3] = { &sflux[3 * n].f[b], &sflux[3 * n + 1].f[b], &sflux[3 * n + 2].f[b] };
        if (!lv_multi(L)) { vo_update(&S[n].sold, u3, e3, f3, &scal_force[n], &S[n].snew, dx + 3 * n, dt, 0, is_cons); continue; }
        vo_fab bs, bf, bn;
        box_gather(&bs, lo, hi, 0, &S[n].sold); box_gather(&bf, lo, hi, 0, &scal_force[n]); box_gather(&bn, lo, hi, 0, &S[n].snew);
        vo_update(&bs, u3, e3, f3, &bf, &bn, dx + 3 * n, dt, 0, is_cons);
        box_scatter(&S[n].snew, &bn, 0, nscal);
        free(bs.p); free(bf.p); free(bn.p);
      }
    }
    vo_ml_restrict_and_fill_g(NL, lev, snewp, 0, dm, nscal, 0, bc, pmask, pd, prm);
    if (diffusive) {                                                                   /* scalar_advance.f90:144-162 */
      const double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->diff_coef : dt * prm->diff_coef;
      vo_mgstat sst;
      for (int c = 1; c < nscal; c++) vo_ml_diff_scalar_solve_g(NL, lev, snewp, lsp, dx, visc_mu, bc, pmask, pd, prm, c, dm + c, &sst);
    }
    for (int n = 0; n < NL; n++) free(laps[n].p);
    for (int n = 0; n < NL; n++) { free(scal_force[n].p); free(divu[n].p); for (int d = 0; d < 3; d++) { bmf_free(&sflux[3 * n + d]); bmf_free(&sedge[3 * n + d]); } }
  }
  for (int n = 0; n < NL; n++) { vo_fab *in[2] = { &S[n].sold, &S[n].snew }; on_boxes(LEV(lev, n), &rhohalf[n], 0, 1, in, 2, &bc[n], PDL(n), prm, k_halftime, NULL); }
  vo_ml_restrict_and_fill_g(NL, lev, rhp, 0, dm + 0, 1, 0, bc, pmask, pd, prm);
  if (viscous && prm->diffusion_type == 2) for (int n = 0; n < NL; n++) memset(lapu[n].p, 0, sizeof(double) * vo_size(&lapu[n]));    /* advance_timestep.f90:116-120 */
  /* velocity advance */
  {
    int is_cons[3] = { 0, 0, 0 };
    for (int n = 0; n < NL; n++) {
      for (int d = 0; d < 3; d++) { bmf_new(&uflux[3 * n + d], LEV(lev, n), &S[n].uold, 0, d, dm, 0.0); bmf_new(&uedge[3 * n + d], LEV(lev, n), &S[n].uold, 0, d, dm, 0.0); }
      kctx k = { prm, 1.0, NULL, 0, 0 }; vo_fab *in[4] = { &S[n].ext_vel_force, &S[n].gp, &S[n].sold, viscous ? &lapu[n] : NULL };
      on_boxes(LEV(lev, n), &vel_force[n], 0, dm, in, 4, &bc[n], PDL(n), prm, k_velforce, &k);
    }
    vo_ml_restrict_and_fill_g(NL, lev, vfp, 0, bc[0].extrap_comp, dm, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) {
      const vo_level *L = LEV(lev, n);
      FOR_BOXES(L, &S[n].uold, b, lo, hi) {
        vo_fab *u3[3] = { &umac[3 * n].f[b], &umac[3 * n + 1].f[b], &umac[3 * n + 2].f[b] }, *e3[3] = { &uedge[3 * n].f[b], &uedge[3 * n + 1].f[b], &uedge[3 * n + 2].f[b] }, *f3[3] = { &uflux[3 * n].f[b], &uflux[3 * n + 1].f[b], &uflux[3 * n + 2].f[b] };
        if (!lv_multi(L)) { vo_mkflux(&S[n].uold, e3, f3, u3, &vel_force[n], &mac_rhs[n], dx + 3 * n, dt, 1, is_cons, 0, &bc[n], prm); continue; }
        vo_fab bu, bf, bm; vo_bc bb;
        box_gather(&bu, lo, hi, S[n].uold.ng, &S[n].uold); box_gather(&bf, lo, hi, 1, &vel_force[n]); box_gather(&bm, lo, hi, 1, &mac_rhs[n]); box_bc(&bb, &bc[n], lo, hi, PDL(n), prm);
        vo_mkflux(&bu, e3, f3, u3, &bf, &bm, dx + 3 * n, dt, 1, is_cons, 0, &bb, prm);
        free(bu.p); free(bf.p); free(bm.p);
      }
      kctx k = { prm, 0.0, NULL, 0, 0 }; vo_fab *in[4] = { &S[n].ext_vel_force, &S[n].gp, &rhohalf[n], viscous ? &lapu[n] : NULL };
      on_boxes(L, &vel_force[n], 0, dm, in, 4, &bc[n], PDL(n), prm, k_velforce, &k);
    }
    vo_ml_restrict_and_fill_g(NL, lev, vfp, 0, bc[0].extrap_comp, dm, 1, bc, pmask, pd, prm);
    for (int n = 0; n < NL; n++) {
      const vo_level *L = LEV(lev, n);
      FOR_BOXES(L, &S[n].uold, b, lo, hi) {
        vo_fab *u3[3] = { &umac[3 * n].f[b], &umac[3 * n + 1].f[b], &umac[3 * n + 2].f[b] }, *e3[3] = { &uedge[3 * n].f[b], &uedge[3 * n + 1].f[b], &uedge[3 * n + 2].f[b] }, *f3[3] = { &uflux[3 * n].f[b], &uflux[3 * n + 1].f[b], &uflux[3 * n + 2].f[b] };
        if (!lv_multi(L)) { vo_update(&S[n].uold, u3, e3, f3, &vel_force[n], &S[n].unew, dx + 3 * n, dt, 1, is_cons); continue; }
        vo_fab bu, bf, bn;
        box_gather(&bu, lo, hi, 0, &S[n].uold); box_gather(&bf, lo, hi, 0, &vel_force[n]); box_gather(&bn, lo, hi, 0, &S[n].unew);
        vo_update(&bu, u3, e3, f3, &bf, &bn, dx + 3 * n, dt, 1, is_cons);
        box_scatter(&S[n].unew, &bn, 0, dm);
        free(bu.p); free(bf.p); free(bn.p);
      }
    }
    vo_ml_restrict_and_fill_g(NL, lev, unewp, 0, 0, dm, 0, bc, pmask, pd, prm);
    if (viscous) {                                                                     /* velocity_advance.f90:103-118 */
      const double visc_mu = (prm->diffusion_type == 1) ? 0.5 * dt * prm->visc_coef : dt * prm->visc_coef;
      vo_mgstat vst;
      vo_ml_visc_solve_g(NL, lev, unewp, lap, rhp, mrp, dx, visc_mu, bc, pmask, pd, prm, &vst);
    }
    for (int n = 0; n < NL; n++) for (int d = 0; d < 3; d++) { bmf_free(&uflux[3 * n + d]); bmf_free(&uedge[3 * n + d]); }
  }
  vo_ml_hgproject_g(NL, lev, proj_type, unewp, uoldp, rhp, pp, gpp, dx, dt, bc, pmask, pd, prm, &st[1]);
  for (int n = 0; n < NL; n++) { free(mac_rhs[n].p); free(rhohalf[n].p); free(vel_force[n].p); free(lapu[n].p); for (int d = 0; d < 3; d++) bmf_free(&umac[3 * n + d]); }
  #undef PDL
}
