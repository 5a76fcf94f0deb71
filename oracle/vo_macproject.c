/* oracle/vo_macproject.c -- MAC projection (reference src/macproject.f90) and the cell-centred
 * multigrid that stands in for FBoxLib's ml_cc_solve (called at src/mac_multigrid.f90:53-62).
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * The discrete system (fixed by macproject.f90:185-196, 376-394, 611-612):
 *      -sum_d [ b_d(i+e_d) (phi(i+e_d)-phi(i)) - b_d(i) (phi(i)-phi(i-e_d)) ] / h_d^2 = rh(i)
 * with b_d = beta on d-faces.  Domain faces:  Neumann -> face term dropped (b := 0);
 * Dirichlet (phi = 0 on the face, linear closure phi_ghost = -phi_i) -> b := 2 b with a zero
 * ghost;  periodic -> wrap.  So after the b-modification the ghost cells of phi are 0 (or the
 * periodic image) and the operator has no boundary branches.
 *
 * The ALGORITHM is ours (F_MG is not in the reference tree): V(nu1,nu2) cycles, red-black
 * Gauss-Seidel smoothing, 8-cell average restriction, piecewise-constant prolongation, coarse
 * b = average of the 4 fine faces, coarsening while every extent is even and > 2, max(nub, N^2) sweeps
 * on the coarsest level (N = its largest extent); convergence is tested on the residual the cycle computes after its
 * pre-smoothing:  ||r||_inf <= rel_eps*||rh||_inf  or  <= abs_eps.
 * The HIP solver (varden_amd/csrc/mg_cc.hip) implements the same algorithm with the same
 * expression order, so the two agree to round-off of the max-norm test (bit-exact in practice).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "vo.h"

/* macproject.f90:250-278 */
void vo_divumac(vo_fab *umac[3], vo_fab *rh, const double dx[3])
{
  const int *lo = rh->lo, *hi = rh->hi;
  double dxinv[3] = { 1.0 / dx[0], 1.0 / dx[1], 1.0 / dx[2] };
  #pragma omp parallel for
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
    VF(rh, i, j, k, 0) = (VF(umac[0], i + 1, j, k, 0) - VF(umac[0], i, j, k, 0)) * dxinv[0]
                       + (VF(umac[1], i, j + 1, k, 0) - VF(umac[1], i, j, k, 0)) * dxinv[1]
                       + (VF(umac[2], i, j, k + 1, 0) - VF(umac[2], i, j, k, 0)) * dxinv[2];
}

/* macproject.f90:361-401 */
void vo_mk_mac_coeffs(const vo_fab *rho, vo_fab *beta[3])
{
  const int *lo = rho->lo, *hi = rho->hi;
  for (int d = 0; d < 3; d++) {
    int rhi[3] = { hi[0], hi[1], hi[2] }; rhi[d] += 1;
    for (int k = lo[2]; k <= rhi[2]; k++) for (int j = lo[1]; j <= rhi[1]; j++) for (int i = lo[0]; i <= rhi[0]; i++) {
      int m[3] = { i, j, k }; m[d] -= 1;
      VF(beta[d], i, j, k, 0) = 2.0 / (VF(rho, i, j, k, 0) + VF(rho, m[0], m[1], m[2], 0));
    }
  }
}

/* macproject.f90:578-645.  Interior faces: u -= beta (phi_i - phi_{i-1})/dx (611-612).  Box faces: the
 * reference subtracts the solver's flux register (608-609, FBoxLib bndry_reg, not in the tree); that
 * flux is beta * dphi/dn of the solver's own boundary closure, i.e. 0 at Neumann faces,
 * beta*(phi_i - (-phi_i))/dx at Dirichlet faces, and the ordinary difference with the periodic /
 * neighbour ghost value otherwise.  phi's ghosts are filled accordingly before this is called. */
void vo_mkumac(vo_fab *umac[3], const vo_fab *phi, vo_fab *beta[3], const double dx[3], const int ellbc[3][2])
{
  const int *lo = phi->lo, *hi = phi->hi;
  for (int d = 0; d < 3; d++) {
    int rhi[3] = { hi[0], hi[1], hi[2] }; rhi[d] += 1;
    #pragma omp parallel for
    for (int k = lo[2]; k <= rhi[2]; k++) for (int j = lo[1]; j <= rhi[1]; j++) for (int i = lo[0]; i <= rhi[0]; i++) {
      int q[3] = { i, j, k }, m[3] = { i, j, k }; m[d] -= 1;
      int side = (q[d] == lo[d]) ? 0 : ((q[d] == hi[d] + 1) ? 1 : -1);
      if (side >= 0 && ellbc[d][side] == VDN_BC_NEU) continue;
      double gphi = (VF(phi, i, j, k, 0) - VF(phi, m[0], m[1], m[2], 0)) / dx[d];
      VF(umac[d], i, j, k, 0) = VF(umac[d], i, j, k, 0) - VF(beta[d], i, j, k, 0) * gphi;
    }
  }
}

/* ---------------------------------------------------------------------------------------------
 * cell-centred multigrid
 * ------------------------------------------------------------------------------------------- */
typedef struct cclev {
  int n[3];              /* cells */
  double h[3], hi2[3];   /* spacing, 1/h^2 */
  double *phi;           /* (n+2)^3 with one ghost layer */
  double *rh;            /* n^3 */
  double *res;           /* n^3 */
  double *b[3];          /* face coefficients, (n+e_d) extents, bc-modified */
  double *alpha;         /* cell coefficient of the (alpha - div b grad) operator, NULL when alpha = 0 */
  int dm;                /* 2: one z-plane (n[2] = 1), b[2] = 0, no coarsening along z -> the 5-point operator */
} cclev;

#define PHI(L, i, j, k) (L)->phi[((i) + 1) + ((L)->n[0] + 2) * (((j) + 1) + (long)((L)->n[1] + 2) * ((k) + 1))]
#define CC(L, a, i, j, k) (a)[(i) + (long)(L)->n[0] * ((j) + (long)(L)->n[1] * (k))]
#define BX(L, i, j, k) (L)->b[0][(i) + (long)((L)->n[0] + 1) * ((j) + (long)(L)->n[1] * (k))]
#define BY(L, i, j, k) (L)->b[1][(i) + (long)(L)->n[0] * ((j) + (long)((L)->n[1] + 1) * (k))]
#define BZ(L, i, j, k) (L)->b[2][(i) + (long)(L)->n[0] * ((j) + (long)(L)->n[1] * (k))]

static void cc_alloc(cclev *L, const int n[3], const double h[3])
{
  for (int d = 0; d < 3; d++) { L->n[d] = n[d]; L->h[d] = h[d]; L->hi2[d] = 1.0 / (h[d] * h[d]); }
  long nc = (long)n[0] * n[1] * n[2], ng = (long)(n[0] + 2) * (n[1] + 2) * (n[2] + 2);
  L->phi = (double *)calloc(ng, sizeof(double));
  L->rh = (double *)calloc(nc, sizeof(double));
  L->res = (double *)calloc(nc, sizeof(double));
  for (int d = 0; d < 3; d++) {
    long nf = 1; for (int t = 0; t < 3; t++) nf *= (n[t] + (t == d));
    L->b[d] = (double *)calloc(nf, sizeof(double));
  }
  L->alpha = NULL; L->dm = 3;
}
static void cc_free(cclev *L) { free(L->phi); free(L->rh); free(L->res); for (int d = 0; d < 3; d++) free(L->b[d]); free(L->alpha); }

static void cc_fill_periodic(cclev *L, const int per[3])
{
  int any = per[0] || per[1] || per[2]; if (!any) return;
  const int *n = L->n;
  for (int k = -1; k <= n[2]; k++) for (int j = -1; j <= n[1]; j++) for (int i = -1; i <= n[0]; i++) {
    int q[3] = { i, j, k }, s[3] = { i, j, k }, g = 0, ok = 1;
    for (int d = 0; d < 3; d++) {
      if (q[d] < 0) { g = 1; if (per[d]) s[d] = q[d] + n[d]; else ok = 0; }
      else if (q[d] >= n[d]) { g = 1; if (per[d]) s[d] = q[d] - n[d]; else ok = 0; }
    }
    if (g && ok) PHI(L, i, j, k) = PHI(L, s[0], s[1], s[2]);
  }
}

/* A phi at one cell and the diagonal, in the fixed expression order shared with the HIP kernel */
static inline void cc_apply(const cclev *L, int i, int j, int k, double *Ap, double *diag)
{
  double p0 = PHI(L, i, j, k);
  double bxm = BX(L, i, j, k), bxp = BX(L, i + 1, j, k);
  double bym = BY(L, i, j, k), byp = BY(L, i, j + 1, k);
  double bzm = BZ(L, i, j, k), bzp = BZ(L, i, j, k + 1);
  double ax = (bxp * (p0 - PHI(L, i + 1, j, k)) + bxm * (p0 - PHI(L, i - 1, j, k))) * L->hi2[0];
  double ay = (byp * (p0 - PHI(L, i, j + 1, k)) + bym * (p0 - PHI(L, i, j - 1, k))) * L->hi2[1];
  double az = (bzp * (p0 - PHI(L, i, j, k + 1)) + bzm * (p0 - PHI(L, i, j, k - 1))) * L->hi2[2];
  *Ap = ax + ay + az;
  *diag = (bxp + bxm) * L->hi2[0] + (byp + bym) * L->hi2[1] + (bzp + bzm) * L->hi2[2];
  if (L->alpha) {                       /* (alpha - div b grad): viscous / diffusive solves */
    double a0 = CC(L, L->alpha, i, j, k);
    *Ap = *Ap + a0 * p0;
    *diag = *diag + a0;
  }
}

/* A phi and the diagonal on the whole row (0 .. n0-1, j, k): cc_apply's expressions cell by cell, the loop over i with unit stride so that the compiler
 * vectorises it (round 4).  A colour pass evaluates the row for BOTH colours and stores the cells of its colour: their neighbours are of the other
 * colour and do not change during the pass, so the values are the ones the cell-by-cell loop computes. */
#define CC_ROW_MAX 2048
static inline void cc_row(const cclev *L, int j, int k, double *restrict Ap, double *restrict dg)
{
  const int n0 = L->n[0];
  const long sy = n0 + 2, sz = sy * (L->n[1] + 2);
  const double *p = &PHI(L, 0, j, k);
  const double *bx = &BX(L, 0, j, k), *by0 = &BY(L, 0, j, k), *by1 = &BY(L, 0, j + 1, k), *bz0 = &BZ(L, 0, j, k), *bz1 = &BZ(L, 0, j, k + 1);
  const double h0 = L->hi2[0], h1 = L->hi2[1], h2 = L->hi2[2];
  const double *al = L->alpha ? &CC(L, L->alpha, 0, j, k) : NULL;
  if (al) {
    #pragma omp simd
    for (int i = 0; i < n0; i++) {
      const double p0 = p[i], bxm = bx[i], bxp = bx[i + 1], bym = by0[i], byp = by1[i], bzm = bz0[i], bzp = bz1[i];
      const double ax = (bxp * (p0 - p[i + 1]) + bxm * (p0 - p[i - 1])) * h0;
      const double ay = (byp * (p0 - p[i + sy]) + bym * (p0 - p[i - sy])) * h1;
      const double az = (bzp * (p0 - p[i + sz]) + bzm * (p0 - p[i - sz])) * h2;
      const double a0 = al[i];
      Ap[i] = (ax + ay + az) + a0 * p0;
      dg[i] = ((bxp + bxm) * h0 + (byp + bym) * h1 + (bzp + bzm) * h2) + a0;
    }
  } else {
    #pragma omp simd
    for (int i = 0; i < n0; i++) {
      const double p0 = p[i], bxm = bx[i], bxp = bx[i + 1], bym = by0[i], byp = by1[i], bzm = bz0[i], bzp = bz1[i];
      const double ax = (bxp * (p0 - p[i + 1]) + bxm * (p0 - p[i - 1])) * h0;
      const double ay = (byp * (p0 - p[i + sy]) + bym * (p0 - p[i - sy])) * h1;
      const double az = (bzp * (p0 - p[i + sz]) + bzm * (p0 - p[i - sz])) * h2;
      Ap[i] = ax + ay + az;
      dg[i] = (bxp + bxm) * h0 + (byp + bym) * h1 + (bzp + bzm) * h2;
    }
  }
}
static void cc_gsrb(cclev *L, const int per[3], int nsweeps)
{
  const int *n = L->n;
  for (int s = 0; s < nsweeps; s++) for (int color = 0; color < 2; color++) {
    cc_fill_periodic(L, per);
    if (L->dm == 3 && n[0] <= CC_ROW_MAX) {
      #pragma omp parallel for collapse(2) schedule(static)
      for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) {
        double Ap[CC_ROW_MAX], dg[CC_ROW_MAX];
        cc_row(L, j, k, Ap, dg);
        double *p = &PHI(L, 0, j, k); const double *rh = &CC(L, L->rh, 0, j, k);
        for (int i = (j + k + color) & 1; i < n[0]; i += 2)
          if (dg[i] != 0.0) p[i] = p[i] + (rh[i] - Ap[i]) / dg[i];
      }
      continue;
    }
    #pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++)
      for (int i = (j + k + color) & 1; i < n[0]; i += 2) {
        double Ap, diag; cc_apply(L, i, j, k, &Ap, &diag);
        if (diag != 0.0) PHI(L, i, j, k) = PHI(L, i, j, k) + (CC(L, L->rh, i, j, k) - Ap) / diag;
      }
  }
}

static double cc_residual(cclev *L, const int per[3])
{
  const int *n = L->n; double nrm = 0.0;
  cc_fill_periodic(L, per);
  if (L->dm == 3 && n[0] <= CC_ROW_MAX) {
    #pragma omp parallel for collapse(2) schedule(static) reduction(max : nrm)
    for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) {
      double Ap[CC_ROW_MAX], dg[CC_ROW_MAX];
      cc_row(L, j, k, Ap, dg);
      const double *rh = &CC(L, L->rh, 0, j, k); double *out = &CC(L, L->res, 0, j, k);
      for (int i = 0; i < n[0]; i++) { const double r = rh[i] - Ap[i]; out[i] = r; nrm = vo_nrm_acc(nrm, r); }
    }
    return nrm;
  }
  #pragma omp parallel for collapse(2) schedule(static) reduction(max : nrm)
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
    double Ap, diag; cc_apply(L, i, j, k, &Ap, &diag);
    double r = CC(L, L->rh, i, j, k) - Ap;
    CC(L, L->res, i, j, k) = r;
    nrm = vo_nrm_acc(nrm, r);
  }
  return nrm;
}

static void cc_restrict(const cclev *F, cclev *C)
{
  const int *n = C->n;
  #pragma omp parallel for
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
    int I = 2 * i, J = 2 * j, K = 2 * k;
    if (C->dm == 2) {                        /* 4-cell average */
      double s = CC(F, F->res, I, J, 0) + CC(F, F->res, I + 1, J, 0) + CC(F, F->res, I, J + 1, 0) + CC(F, F->res, I + 1, J + 1, 0);
      CC(C, C->rh, i, j, k) = s * 0.25;
      continue;
    }
    double s = CC(F, F->res, I, J, K) + CC(F, F->res, I + 1, J, K) + CC(F, F->res, I, J + 1, K) + CC(F, F->res, I + 1, J + 1, K)
             + CC(F, F->res, I, J, K + 1) + CC(F, F->res, I + 1, J, K + 1) + CC(F, F->res, I, J + 1, K + 1) + CC(F, F->res, I + 1, J + 1, K + 1);
    CC(C, C->rh, i, j, k) = s * 0.125;
  }
}

static void cc_prolong_add(cclev *F, const cclev *C)
{
  const int *n = F->n;
  #pragma omp parallel for
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++)
    PHI(F, i, j, k) = PHI(F, i, j, k) + PHI(C, i / 2, j / 2, k / 2);
}

static void cc_coarsen_coeffs(const cclev *F, cclev *C)
{
  const int *n = C->n;
  if (C->dm == 2) {                          /* coarse face coefficient = mean of the 2 fine faces */
    for (int j = 0; j < n[1]; j++) for (int i = 0; i <= n[0]; i++) BX(C, i, j, 0) = (BX(F, 2 * i, 2 * j, 0) + BX(F, 2 * i, 2 * j + 1, 0)) * 0.5;
    for (int j = 0; j <= n[1]; j++) for (int i = 0; i < n[0]; i++) BY(C, i, j, 0) = (BY(F, 2 * i, 2 * j, 0) + BY(F, 2 * i + 1, 2 * j, 0)) * 0.5;
    return;
  }
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i <= n[0]; i++)
    BX(C, i, j, k) = (BX(F, 2 * i, 2 * j, 2 * k) + BX(F, 2 * i, 2 * j + 1, 2 * k) + BX(F, 2 * i, 2 * j, 2 * k + 1) + BX(F, 2 * i, 2 * j + 1, 2 * k + 1)) * 0.25;
  for (int k = 0; k < n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i < n[0]; i++)
    BY(C, i, j, k) = (BY(F, 2 * i, 2 * j, 2 * k) + BY(F, 2 * i + 1, 2 * j, 2 * k) + BY(F, 2 * i, 2 * j, 2 * k + 1) + BY(F, 2 * i + 1, 2 * j, 2 * k + 1)) * 0.25;
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++)
    BZ(C, i, j, k) = (BZ(F, 2 * i, 2 * j, 2 * k) + BZ(F, 2 * i + 1, 2 * j, 2 * k) + BZ(F, 2 * i, 2 * j + 1, 2 * k) + BZ(F, 2 * i + 1, 2 * j + 1, 2 * k)) * 0.25;
}

typedef struct ccmg { int nlev; cclev lev[32]; int per[3]; } ccmg;

static void ccmg_build(ccmg *M, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2])
{
  const vo_fab *b0 = beta[0];
  int n[3]; double h[3];
  const int dm = b0->dm;
  for (int d = 0; d < 3; d++) { n[d] = b0->hi[d] - b0->lo[d] + 1; h[d] = dx[d]; M->per[d] = (d < dm) && (ellbc[d][0] == VDN_BC_PER); }
  if (dm == 2) { n[2] = 1; h[2] = 1.0; }
  M->nlev = 0;
  for (;;) {
    cclev *L = &M->lev[M->nlev];
    cc_alloc(L, n, h); L->dm = dm;
    if (M->nlev == 0) {
      /* copy beta with the boundary modification */
      for (int d = 0; d < dm; d++) {
        const vo_fab *bf = beta[d];
        int e[3] = { n[0], n[1], n[2] }; e[d] += 1;
        for (int k = 0; k < e[2]; k++) for (int j = 0; j < e[1]; j++) for (int i = 0; i < e[0]; i++) {
          int q[3] = { i, j, k };
          double v = VF(bf, bf->lo[0] + i, bf->lo[1] + j, bf->lo[2] + k, 0);
          int side = (q[d] == 0) ? 0 : ((q[d] == n[d]) ? 1 : -1);
          if (side >= 0) {
            if (ellbc[d][side] == VDN_BC_NEU) v = 0.0;
            else if (ellbc[d][side] == VDN_BC_DIR) v = 2.0 * v;
          }
          L->b[d][i + (long)e[0] * (j + (long)e[1] * k)] = v;
        }
      }
    } else cc_coarsen_coeffs(&M->lev[M->nlev - 1], L);
    if (alpha) {                          /* level 0: copy; coarser: mean of the 8 children */
      long nc = (long)n[0] * n[1] * n[2];
      L->alpha = (double *)calloc(nc, sizeof(double));
      for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
        if (M->nlev == 0) CC(L, L->alpha, i, j, k) = VF(alpha, alpha->lo[0] + i, alpha->lo[1] + j, alpha->lo[2] + k, 0);
        else {
          const cclev *F = &M->lev[M->nlev - 1];
          int I = 2 * i, J = 2 * j, K = 2 * k;
          if (dm == 2) { CC(L, L->alpha, i, j, k) = (CC(F, F->alpha, I, J, 0) + CC(F, F->alpha, I + 1, J, 0) + CC(F, F->alpha, I, J + 1, 0) + CC(F, F->alpha, I + 1, J + 1, 0)) * 0.25; continue; }
          double sum = CC(F, F->alpha, I, J, K) + CC(F, F->alpha, I + 1, J, K) + CC(F, F->alpha, I, J + 1, K) + CC(F, F->alpha, I + 1, J + 1, K)
                     + CC(F, F->alpha, I, J, K + 1) + CC(F, F->alpha, I + 1, J, K + 1) + CC(F, F->alpha, I, J + 1, K + 1) + CC(F, F->alpha, I + 1, J + 1, K + 1);
          CC(L, L->alpha, i, j, k) = sum * 0.125;
        }
      }
    }
    M->nlev++;
    int can = 1;
    for (int d = 0; d < dm; d++) if ((n[d] & 1) || n[d] <= 2) can = 0;
    if (!can || M->nlev >= 31) break;
    for (int d = 0; d < dm; d++) { n[d] /= 2; h[d] *= 2.0; }
  }
}
static void ccmg_free(ccmg *M) { for (int l = 0; l < M->nlev; l++) cc_free(&M->lev[l]); }

/* coarse-grid correction below level l (error equation, zero initial guess) */
/* sweeps on the coarsest level: `nub`, raised to N^2 (N = its largest extent) so that boxes that stop
 * coarsening early (non-cubic, or extents with odd factors) still get a bottom solve of ~1e-3 accuracy
 * (red-black GS on N cells contracts like 1 - pi^2/N^2; the reference asks its bottom solver for 1e-3,
 * mac_multigrid.f90:56) */
static int cc_bottom_sweeps(const cclev *L, int nub)
{
  int N = L->n[0] > L->n[1] ? L->n[0] : L->n[1]; if (L->dm == 3 && L->n[2] > N) N = L->n[2];
  return nub > N * N ? nub : N * N;
}

static void cc_vcycle(ccmg *M, int l, int nu1, int nu2, int nub)
{
  cclev *L = &M->lev[l];
  long ng = (long)(L->n[0] + 2) * (L->n[1] + 2) * (L->n[2] + 2);
  memset(L->phi, 0, sizeof(double) * ng);
  if (l == M->nlev - 1) { cc_gsrb(L, M->per, cc_bottom_sweeps(L, nub)); return; }
  cc_gsrb(L, M->per, nu1);
  (void)cc_residual(L, M->per);
  cc_restrict(L, &M->lev[l + 1]);
  cc_vcycle(M, l + 1, nu1, nu2, nub);
  cc_prolong_add(L, &M->lev[l + 1]);
  cc_gsrb(L, M->per, nu2);
}

/* ---- nested iteration for the initial guess (vdn_params.mac_fmg; round 3) --------------------------------------------------------------
 * The right-hand side is averaged down the hierarchy (the residual's restriction), the coarsest level of at least 16^3 cells gets two V-cycles
 * (the first from zero), and every level above it takes a LINEAR interpolation of the solution of the level below and, except the finest,
 * one V-cycle of its own.  The V-cycles that follow start from an error at truncation level: 8 -> 7 cycles to 1e-10 at 64^3 and 128^3.
 * The interpolation reads face neighbours only -- fine cell = (p0 + px + py + pz)/4 with px, py, pz the coarse neighbours on the fine
 * cell's side: exact for linear functions like the tri-linear one (which saved the same cycle), and it needs no edge or corner ghost
 * cells; the piecewise-constant prolongation of the V-cycle saves nothing here.  Three dimensions only. */
static void cc_restrict_rh(const cclev *F, cclev *C)
{
  const int *n = C->n;
  #pragma omp parallel for
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
    int I = 2 * i, J = 2 * j, K = 2 * k;
    double s = CC(F, F->rh, I, J, K) + CC(F, F->rh, I + 1, J, K) + CC(F, F->rh, I, J + 1, K) + CC(F, F->rh, I + 1, J + 1, K)
             + CC(F, F->rh, I, J, K + 1) + CC(F, F->rh, I + 1, J, K + 1) + CC(F, F->rh, I, J + 1, K + 1) + CC(F, F->rh, I + 1, J + 1, K + 1);
    CC(C, C->rh, i, j, k) = s * 0.125;
  }
}
/* the coarse neighbour of cell q along d (off = -1 / +1) as the solver's boundary closure sees it: the periodic image, the cell itself at a
 * Neumann face, minus the cell at a (homogeneous) Dirichlet face */
static inline double cc_nbv(const cclev *C, int d, const int q[3], int off, const int per[3], const int ellbc[3][2])
{
  int m[3] = { q[0], q[1], q[2] }; m[d] += off;
  if (m[d] < 0) { if (per[d]) m[d] += C->n[d]; else return ellbc[d][0] == VDN_BC_DIR ? -PHI(C, q[0], q[1], q[2]) : PHI(C, q[0], q[1], q[2]); }
  else if (m[d] >= C->n[d]) { if (per[d]) m[d] -= C->n[d]; else return ellbc[d][1] == VDN_BC_DIR ? -PHI(C, q[0], q[1], q[2]) : PHI(C, q[0], q[1], q[2]); }
  return PHI(C, m[0], m[1], m[2]);
}
static void cc_prolong_linear(cclev *F, const cclev *C, const int per[3], const int ellbc[3][2])
{
  const int *n = F->n;
  #pragma omp parallel for
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
    const int q[3] = { i >> 1, j >> 1, k >> 1 };
    const double p0 = PHI(C, q[0], q[1], q[2]);
    const double px = cc_nbv(C, 0, q, (i & 1) ? 1 : -1, per, ellbc), py = cc_nbv(C, 1, q, (j & 1) ? 1 : -1, per, ellbc), pz = cc_nbv(C, 2, q, (k & 1) ? 1 : -1, per, ellbc);
    PHI(F, i, j, k) = 0.25 * (((p0 + px) + py) + pz);
  }
}
static void cc_vcycle(ccmg *M, int l, int nu1, int nu2, int nub);
/* one V-cycle at level l (with a level below it) on the level's own right-hand side, from the phi it holds */
static void cc_cycle_at(ccmg *M, int l, int nu1, int nu2, int nub)
{
  cclev *L = &M->lev[l];
  cc_gsrb(L, M->per, nu1);
  (void)cc_residual(L, M->per);
  cc_restrict(L, &M->lev[l + 1]);
  cc_vcycle(M, l + 1, nu1, nu2, nub);
  cc_prolong_add(L, &M->lev[l + 1]);
  cc_gsrb(L, M->per, nu2);
}
static void cc_fmg(ccmg *M, const int ellbc[3][2], int nu1, int nu2, int nub)
{
  int ls = -1;
  for (int l = 1; l < M->nlev; l++) if ((long)M->lev[l].n[0] * M->lev[l].n[1] * M->lev[l].n[2] >= 4096) ls = l;
  if (ls < 1 || ls + 1 >= M->nlev) return;            /* (a starting level with nothing below it: no nested iteration) */
  for (int l = 0; l < ls; l++) cc_restrict_rh(&M->lev[l], &M->lev[l + 1]);
  cc_vcycle(M, ls, nu1, nu2, nub);                     /* from zero */
  cc_cycle_at(M, ls, nu1, nu2, nub);
  for (int l = ls - 1; l >= 0; l--) {
    cc_prolong_linear(&M->lev[l], &M->lev[l + 1], M->per, ellbc);
    if (l > 0) cc_cycle_at(M, l, nu1, nu2, nub);
  }
}

/* Inhomogeneous Dirichlet data: the ghost cells of the incoming phi hold the boundary-FACE values (that is what
 * multifab_physbc's EXT_DIR fill leaves there, and how visc_solve hands unew to the solver, viscsolve.f90:270).
 * The face term 2 b (phi_i - phi_b)/h^2 is split: the phi_b part moves to the right-hand side here (order:
 * x-lo, x-hi, y-lo, y-hi, z-lo, z-hi), the solver then works with a zero ghost. */
static void cc_load(cclev *L, const vo_fab *rh, const vo_fab *phi, const int ellbc[3][2])
{
  const int *n = L->n;
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
    double r = VF(rh, rh->lo[0] + i, rh->lo[1] + j, rh->lo[2] + k, 0);
    const int gi = phi->lo[0] + i, gj = phi->lo[1] + j, gk = phi->lo[2] + k;
    if (i == 0 && ellbc[0][0] == VDN_BC_DIR)        r = r + BX(L, 0, j, k) * VF(phi, gi - 1, gj, gk, 0) * L->hi2[0];
    if (i == n[0] - 1 && ellbc[0][1] == VDN_BC_DIR) r = r + BX(L, n[0], j, k) * VF(phi, gi + 1, gj, gk, 0) * L->hi2[0];
    if (j == 0 && ellbc[1][0] == VDN_BC_DIR)        r = r + BY(L, i, 0, k) * VF(phi, gi, gj - 1, gk, 0) * L->hi2[1];
    if (j == n[1] - 1 && ellbc[1][1] == VDN_BC_DIR) r = r + BY(L, i, n[1], k) * VF(phi, gi, gj + 1, gk, 0) * L->hi2[1];
    if (L->dm == 3 && k == 0 && ellbc[2][0] == VDN_BC_DIR)        r = r + BZ(L, i, j, 0) * VF(phi, gi, gj, gk - 1, 0) * L->hi2[2];
    if (L->dm == 3 && k == n[2] - 1 && ellbc[2][1] == VDN_BC_DIR) r = r + BZ(L, i, j, n[2]) * VF(phi, gi, gj, gk + 1, 0) * L->hi2[2];
    CC(L, L->rh, i, j, k) = r;
    PHI(L, i, j, k) = VF(phi, gi, gj, gk, 0);
  }
}
/* store phi incl. the ghost layer the solver's closure implies: Neumann ghost = phi_i, Dirichlet
 * ghost = -phi_i, periodic = image (edges/corners are not needed by mkumac and left alone) */
static void cc_store(cclev *L, vo_fab *phi, const int ellbc[3][2], const int per[3])
{
  const int *n = L->n;
  cc_fill_periodic(L, per);
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++)
    VF(phi, phi->lo[0] + i, phi->lo[1] + j, phi->lo[2] + k, 0) = PHI(L, i, j, k);
  for (int d = 0; d < L->dm; d++) for (int s = 0; s < 2; s++) {
    int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    for (int b2 = 0; b2 < n[t2]; b2++) for (int b1 = 0; b1 < n[t1]; b1++) {
      int q[3], g[3]; q[t1] = g[t1] = b1; q[t2] = g[t2] = b2;
      q[d] = s ? n[d] - 1 : 0; g[d] = s ? n[d] : -1;
      double v;
      if (ellbc[d][s] == VDN_BC_NEU) v = PHI(L, q[0], q[1], q[2]);
      else if (ellbc[d][s] == VDN_BC_DIR) v = -PHI(L, q[0], q[1], q[2]);
      else v = PHI(L, g[0], g[1], g[2]);
      VF(phi, phi->lo[0] + g[0], phi->lo[1] + g[1], phi->lo[2] + g[2], 0) = v;
    }
  }
}

int vo_cc_solve(const vo_fab *rh, vo_fab *phi, vo_fab *beta[3], const double dx[3], const int ellbc[3][2],
                double rel_eps, double abs_eps, int max_iter, int nu1, int nu2, int nub, int fmg, vo_mgstat *st)
{
  return vo_cc_solve_ab(rh, phi, NULL, beta, dx, ellbc, rel_eps, abs_eps, max_iter, nu1, nu2, nub, fmg, st);
}

/* (alpha - div beta grad) phi = rh; alpha may be NULL (= 0).  Dirichlet data in phi's ghost cells.
 * fmg: nested iteration for the initial guess (cc_fmg) when phi comes in zero, ghost cells included, and max_iter >= 0 */
int vo_cc_solve_ab(const vo_fab *rh, vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2],
                   double rel_eps, double abs_eps, int max_iter, int nu1, int nu2, int nub, int fmg, vo_mgstat *st)
{
  ccmg M; ccmg_build(&M, alpha, beta, dx, ellbc);
  cclev *L0 = &M.lev[0];
  cc_load(L0, rh, phi, ellbc);
  double bnorm = 0.0;      /* norm of the right-hand side as given (before the Dirichlet data moved into it) */
  for (int k = rh->lo[2]; k <= rh->hi[2]; k++) for (int j = rh->lo[1]; j <= rh->hi[1]; j++) for (int i = rh->lo[0]; i <= rh->hi[0]; i++)
    bnorm = vo_nrm_acc(bnorm, VF(rh, i, j, k, 0));
  int cyc = 0, conv = 0; double rn = 0.0, r0 = -1.0;
  if (fmg && L0->dm == 3 && M.nlev > 1 && bnorm != 0.0 && max_iter >= 0) {
    int zero = 1;
    for (int k = phi->lo[2] - 1; k <= phi->hi[2] + 1 && zero; k++) for (int j = phi->lo[1] - 1; j <= phi->hi[1] + 1 && zero; j++) for (int i = phi->lo[0] - 1; i <= phi->hi[0] + 1; i++)
      if (VF(phi, i, j, k, 0) != 0.0) { zero = 0; break; }
    if (zero) cc_fmg(&M, ellbc, nu1, nu2, nub);
  }
  if (max_iter < 0) {            /* exactly -max_iter V-cycles, no convergence test (coarse correction of the composite solves) */
    for (int c = 0; c < -max_iter; c++) {
      if (M.nlev == 1) { cc_gsrb(L0, M.per, cc_bottom_sweeps(L0, nub)); continue; }
      cc_gsrb(L0, M.per, nu1);
      (void)cc_residual(L0, M.per);
      cc_restrict(L0, &M.lev[1]);
      cc_vcycle(&M, 1, nu1, nu2, nub);
      cc_prolong_add(L0, &M.lev[1]);
      cc_gsrb(L0, M.per, nu2);
    }
    cc_store(L0, phi, ellbc, M.per);
    if (st) { st->cycles = -max_iter; st->res0 = bnorm; st->res = 0.0; }
    ccmg_free(&M);
    return 0;
  }
  if (bnorm == 0.0) { conv = 1; r0 = 0.0; }
  while (!conv && cyc <= max_iter) {
    if (M.nlev == 1) cc_gsrb(L0, M.per, cc_bottom_sweeps(L0, nub)); else cc_gsrb(L0, M.per, nu1);
    rn = cc_residual(L0, M.per);
    if (r0 < 0.0) r0 = rn;
    if (rn <= rel_eps * bnorm || rn <= abs_eps) { conv = 1; break; }
    if (cyc == max_iter) break;
    if (M.nlev > 1) {
      cc_restrict(L0, &M.lev[1]);
      cc_vcycle(&M, 1, nu1, nu2, nub);
      cc_prolong_add(L0, &M.lev[1]);
      cc_gsrb(L0, M.per, nu2);
    }
    cyc++;
  }
  cc_store(L0, phi, ellbc, M.per);
  if (st) { st->cycles = cyc; st->res0 = bnorm; st->res = rn; }
  ccmg_free(&M);
  return conv ? 0 : 1;
}

void vo_cc_smooth_ab(const vo_fab *rh, vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2], int nsweeps)
{
  ccmg M; ccmg_build(&M, alpha, beta, dx, ellbc);
  cc_load(&M.lev[0], rh, phi, ellbc);
  cc_gsrb(&M.lev[0], M.per, nsweeps);
  cc_store(&M.lev[0], phi, ellbc, M.per);
  ccmg_free(&M);
}
/* the same sweeps with the ghost cells of phi beyond the faces that are neither domain nor periodic faces (VDN_BC_INT: the coarse-fine
 * interface of a refined level) taken as data and held fixed; phi's ghost layer is left as it came.  (The composite solve relaxes the
 * correction of a refined level with the interpolated coarse correction beyond the interface.) */
void vo_cc_smooth_ab_iface(const vo_fab *rh, vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2], int nsweeps)
{
  ccmg M; ccmg_build(&M, alpha, beta, dx, ellbc);
  cclev *L = &M.lev[0];
  cc_load(L, rh, phi, ellbc);
  const int *n = L->n;
  for (int d = 0; d < L->dm; d++) for (int s = 0; s < 2; s++) {
    if (ellbc[d][s] != VDN_BC_INT) continue;
    const int t1 = (d + 1) % 3, t2 = (d + 2) % 3;
    for (int b2 = 0; b2 < n[t2]; b2++) for (int b1 = 0; b1 < n[t1]; b1++) {
      int g[3]; g[t1] = b1; g[t2] = b2; g[d] = s ? n[d] : -1;
      PHI(L, g[0], g[1], g[2]) = VF(phi, phi->lo[0] + g[0], phi->lo[1] + g[1], phi->lo[2] + g[2], 0);
    }
  }
  cc_gsrb(L, M.per, nsweeps);
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++)
    VF(phi, phi->lo[0] + i, phi->lo[1] + j, phi->lo[2] + k, 0) = PHI(L, i, j, k);
  ccmg_free(&M);
}
/* test hook: out = A phi on the valid cells, A = (alpha - div beta grad) with the boundary closure of ccmg_build -- the operator every sweep and
 * residual of this file applies (cc_apply), once.  phi's ghost cells are not read (homogeneous closure).  tests/test_operators_assembled_cpu.py
 * compares it with a scipy matrix assembled from SURVEY.md Appendix C.1. */
void vo_cc_apply(const vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2], vo_fab *out)
{
  ccmg M; ccmg_build(&M, alpha, beta, dx, ellbc);
  cclev *L = &M.lev[0];
  const int *n = L->n;
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++)
    PHI(L, i, j, k) = VF(phi, phi->lo[0] + i, phi->lo[1] + j, phi->lo[2] + k, 0);
  cc_fill_periodic(L, M.per);
  for (int k = 0; k < n[2]; k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
    double Ap, diag; cc_apply(L, i, j, k, &Ap, &diag);
    VF(out, out->lo[0] + i, out->lo[1] + j, out->lo[2] + k, 0) = Ap;
  }
  ccmg_free(&M);
}
void vo_cc_smooth(const vo_fab *rh, vo_fab *phi, vo_fab *beta[3], const double dx[3], const int ellbc[3][2], int nsweeps)
{
  ccmg M; ccmg_build(&M, NULL, beta, dx, ellbc);
  cc_load(&M.lev[0], rh, phi, ellbc);
  cc_gsrb(&M.lev[0], M.per, nsweeps);
  cc_store(&M.lev[0], phi, ellbc, M.per);
  ccmg_free(&M);
}

/* macproject.f90:20-133 for one level / one box */
void vo_macproject(vo_fab *umac[3], vo_fab *rho, const vo_fab *mac_rhs, const double dx[3], const vo_bc *bc,
                   const int pmask[3], const vdn_params *prm, vo_mgstat *st)
{
  const int *lo = rho->lo, *hi = rho->hi;
  int nd0[3] = { 0, 0, 0 };
  vo_fab rh, phi, beta[3], *bp[3];
  vo_fab_init(&rh, NULL, lo, hi, 0, nd0, 1);  rh.p = (double *)calloc(vo_size(&rh), sizeof(double));
  vo_fab_init(&phi, NULL, lo, hi, 1, nd0, 1); phi.p = (double *)calloc(vo_size(&phi), sizeof(double));
  for (int d = 0; d < 3; d++) {
    int nd[3] = { 0, 0, 0 }; nd[d] = 1;
    vo_fab_init(&beta[d], NULL, lo, hi, 0, nd, 1); beta[d].p = (double *)calloc(vo_size(&beta[d]), sizeof(double));
    bp[d] = &beta[d];
  }
  int ellbc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ellbc[d][s] = bc->ell[d][s][bc->press_comp];

  /* divumac(before): rh = mac_rhs - div(umac)   (macproject.f90:190-196) */
  vo_divumac(umac, &rh, dx);
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
    VF(&rh, i, j, k, 0) = VF(&rh, i, j, k, 0) * -1.0 + VF(mac_rhs, i, j, k, 0);
  vo_mk_mac_coeffs(rho, bp);
  /* rel = 1e-10, abs = -1 (macproject.f90:91-93) */
  vo_cc_solve(&rh, &phi, bp, dx, ellbc, prm->mac_rel_eps, -1.0, prm->mg_max_iter, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, prm->mac_fmg, st);
  vo_mkumac(umac, &phi, bp, dx, ellbc);
  for (int d = 0; d < 3; d++) vo_fill_boundary(umac[d], pmask);     /* macproject.f90:115-119 */
  free(rh.p); free(phi.p); for (int d = 0; d < 3; d++) free(beta[d].p);
}
