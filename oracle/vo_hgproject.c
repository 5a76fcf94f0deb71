/* oracle/vo_hgproject.c -- nodal (HG) projection: reference src/hgproject.f90, src/hg_multigrid.f90,
 * and the nodal multigrid that stands in for FBoxLib's ml_nd_solve (hg_multigrid.f90:95-105).
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * Discrete system.  Fixed by the reference: phi, rh nodal (hgproject.f90:74-75);
 * sigma = 1/rhohalf on cells, ZERO in ghost cells outside the domain (hg_multigrid.f90:73-79); the RHS
 * is the nodal divergence of the cell-centred unew whose wall ghost planes were zeroed
 * (hgproject.f90:506-511); dense stencil (hgproject.f90:52; 27-point, 21 when dx=dy=dz,
 * hg_hypre.f90:100-113).  Our definition: trilinear (Q1) finite elements, sigma constant per cell,
 * equations scaled by 1/(hx hy hz):
 *      (K phi)_n = sum_{cells c touching n} sigma_c sum_{corners m of c} w(n,m) phi_m ,
 *      w = 4F (m=n), -4f_a+2f_b+2f_c (m differs from n along a only), -2f_a-2f_b+f_c (along a,b),
 *          -F (all three),   f_d = 1/(36 h_d^2),  F = fx+fy+fz ,
 * and we solve  -K phi = rh  ( = "div(sigma grad phi) = div u" ),  rh = D u :
 *      (D u)_n = sum_d ( [sum of u_d over the 4 cells on the + side] - [... - side] ) * 0.25/h_d .
 * Outflow (BC_DIR) boundary nodes carry phi = 0; wall nodes are natural (sigma = 0, u = 0 outside).
 *
 * ALGORITHM (ours): V(nu1,nu2) cycles with damped-Jacobi smoothing (one pass per sweep: the GPU-
 * friendly choice), full-weighting restriction (= P^T/8), trilinear prolongation,
 * coarse sigma = mean of the 8 children, coarsening while every extent is even and > 2,
 * max(nub, 2 N^2) Jacobi sweeps on the coarsest level (N = its largest extent).  Convergence: ||rh + K phi||_inf <= rel*||rh||_inf or <= abs,
 * tested on the residual computed after pre-smoothing, at most max_iter cycles.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "vo.h"

/* hgproject.f90:434-513 */
void vo_create_uvec(vo_fab *unew, const vo_fab *uold, const vo_fab *rhohalf, vo_fab *gp, double dt,
                    const vo_bc *bc, int proj_type)
{
  const int *lo = unew->lo, *hi = unew->hi;
  double dtinv = 1.0 / dt;
  /* gp ghost layer zeroed at INLET faces (453-458) */
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) if (bc->phys[d][s] == VDN_INLET) {
    int rlo[3] = { lo[0] - 1, lo[1] - 1, lo[2] - 1 }, rhi[3] = { hi[0] + 1, hi[1] + 1, hi[2] + 1 };
    rlo[d] = rhi[d] = s ? hi[d] + 1 : lo[d] - 1;
    for (int m = 0; m < 3; m++) for (int k = rlo[2]; k <= rhi[2]; k++) for (int j = rlo[1]; j <= rhi[1]; j++) for (int i = rlo[0]; i <= rhi[0]; i++)
      VF(gp, i, j, k, m) = 0.0;
  }
  if (proj_type == VDN_PRESSURE_ITERS) {
    for (int m = 0; m < 3; m++) for (int k = lo[2] - 1; k <= hi[2] + 1; k++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
      VF(unew, i, j, k, m) = (VF(unew, i, j, k, m) - VF(uold, i, j, k, m)) * dtinv;
  } else if (proj_type == VDN_REGULAR_TIMESTEP) {
    for (int m = 0; m < 3; m++) for (int k = lo[2] - 1; k <= hi[2] + 1; k++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
      VF(unew, i, j, k, m) = VF(unew, i, j, k, m) + dt * VF(gp, i, j, k, m) / VF(rhohalf, i, j, k, 0);
  }
  /* zero the ENTIRE first ghost plane of unew at walls (506-511: unew(lo(1)-1,:,:,:) = ZERO) */
  int ng = unew->ng;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++)
    if (bc->phys[d][s] == VDN_SLIP_WALL || bc->phys[d][s] == VDN_NO_SLIP_WALL) {
      int rlo[3] = { lo[0] - ng, lo[1] - ng, lo[2] - ng }, rhi[3] = { hi[0] + ng, hi[1] + ng, hi[2] + ng };
      rlo[d] = rhi[d] = s ? hi[d] + 1 : lo[d] - 1;
      for (int m = 0; m < 3; m++) for (int k = rlo[2]; k <= rhi[2]; k++) for (int j = rlo[1]; j <= rhi[1]; j++) for (int i = rlo[0]; i <= rhi[0]; i++)
        VF(unew, i, j, k, m) = 0.0;
    }
}

/* hgproject.f90:543-577 */
void vo_mkgphi(vo_fab *gp, const vo_fab *phi, const double dx[3])
{
  const int *lo = gp->lo, *hi = gp->hi;
  double dxinv[3] = { 1.0 / dx[0], 1.0 / dx[1], 1.0 / dx[2] };
  #pragma omp parallel for
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    #define P(a, b, c) VF(phi, i + (a), j + (b), k + (c), 0)
    VF(gp, i, j, k, 0) = 0.25 * (P(1, 0, 0) + P(1, 1, 0) + P(1, 0, 1) + P(1, 1, 1) - P(0, 0, 0) - P(0, 1, 0) - P(0, 0, 1) - P(0, 1, 1)) * dxinv[0];
    VF(gp, i, j, k, 1) = 0.25 * (P(0, 1, 0) + P(1, 1, 0) + P(0, 1, 1) + P(1, 1, 1) - P(0, 0, 0) - P(1, 0, 0) - P(0, 0, 1) - P(1, 0, 1)) * dxinv[1];
    VF(gp, i, j, k, 2) = 0.25 * (P(0, 0, 1) + P(1, 0, 1) + P(0, 1, 1) + P(1, 1, 1) - P(0, 0, 0) - P(1, 0, 0) - P(0, 1, 0) - P(1, 1, 0)) * dxinv[2];
    #undef P
  }
}

/* hgproject.f90:638-698 */
void vo_hg_update(int proj_type, vo_fab *unew, const vo_fab *uold, vo_fab *gp, const vo_fab *gphi,
                  const vo_fab *rhohalf, vo_fab *p, const vo_fab *phi, double dt)
{
  const int *lo = unew->lo, *hi = unew->hi;
  double dtinv = 1.0 / dt;
  for (int m = 0; m < 3; m++) for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    double v = VF(unew, i, j, k, m) - VF(gphi, i, j, k, m) / VF(rhohalf, i, j, k, 0);
    if (proj_type == VDN_PRESSURE_ITERS) v = VF(uold, i, j, k, m) + dt * v;
    VF(unew, i, j, k, m) = v;
  }
  if (proj_type == VDN_INITIAL_PROJECTION || proj_type == VDN_DIVU_ITERS) {
    memset(gp->p, 0, sizeof(double) * vo_size(gp));
    memset(p->p, 0, sizeof(double) * vo_size(p));
  } else if (proj_type == VDN_PRESSURE_ITERS) {
    for (int m = 0; m < 3; m++) for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
      VF(gp, i, j, k, m) = VF(gp, i, j, k, m) + VF(gphi, i, j, k, m);
    for (int k = lo[2]; k <= hi[2] + 1; k++) for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0] + 1; i++)
      VF(p, i, j, k, 0) = VF(p, i, j, k, 0) + VF(phi, i, j, k, 0);
  } else if (proj_type == VDN_REGULAR_TIMESTEP) {
    for (int m = 0; m < 3; m++) for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
      VF(gp, i, j, k, m) = dtinv * VF(gphi, i, j, k, m);
    for (int k = lo[2]; k <= hi[2] + 1; k++) for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0] + 1; i++)
      VF(p, i, j, k, 0) = dtinv * VF(phi, i, j, k, 0);
  }
}

/* ---------------------------------------------------------------------------------------------
 * nodal multigrid
 * ------------------------------------------------------------------------------------------- */
typedef struct ndlev {
  int n[3];              /* cells; nodes are 0..n[d] */
  double h[3], f[3];     /* f_d = 1/(36 h_d^2) */
  double *phi, *tmp;     /* nodes with one ghost layer: (n+3)^3 */
  double *b;             /* right-hand side of  K phi = b  (b = -rh)                      */
  double *res;           /* residual b - K phi, with ghost layer                           */
  double *sig;           /* cells with one ghost layer: (n+2)^3                            */
  unsigned char *dir;    /* Dirichlet mask on nodes (no ghost)                             */
  int dm;                /* 2: one plane of nodes (n[2] = 0), 9-point Q1 operator, f_d = 1/(6 h_d^2) */
} ndlev;

#define NN(L, i, j, k) (((i) + 1) + (long)((L)->n[0] + 3) * (((j) + 1) + (long)((L)->n[1] + 3) * ((k) + 1)))
#define NS(L, i, j, k) (((i) + 1) + (long)((L)->n[0] + 2) * (((j) + 1) + (long)((L)->n[1] + 2) * ((k) + 1)))
#define NM(L, i, j, k) ((i) + (long)((L)->n[0] + 1) * ((j) + (long)((L)->n[1] + 1) * (k)))

static void nd_alloc(ndlev *L, const int n[3], const double h[3], int dm)
{
  L->dm = dm;
  for (int d = 0; d < 3; d++) { L->n[d] = n[d]; L->h[d] = h[d]; L->f[d] = (dm == 2) ? 1.0 / (6.0 * (h[d] * h[d])) : 1.0 / (36.0 * (h[d] * h[d])); }
  long nn = (long)(n[0] + 3) * (n[1] + 3) * (n[2] + 3), ns = (long)(n[0] + 2) * (n[1] + 2) * (n[2] + 2);
  L->phi = (double *)calloc(nn, sizeof(double)); L->tmp = (double *)calloc(nn, sizeof(double));
  L->b = (double *)calloc(nn, sizeof(double));   L->res = (double *)calloc(nn, sizeof(double));
  L->sig = (double *)calloc(ns, sizeof(double));
  L->dir = (unsigned char *)calloc((long)(n[0] + 1) * (n[1] + 1) * (n[2] + 1), 1);
}
static void nd_free(ndlev *L) { free(L->phi); free(L->tmp); free(L->b); free(L->res); free(L->sig); free(L->dir); }

/* ghost nodes: periodic image, else zero.  Also makes node n[d] the alias of node 0. */
/* (round 4: rows whose j and k are interior are visited at i = -1, n, n+1 only -- the entries the rule can touch --, planes in parallel: a source node
 * is never itself a ghost or alias node, so the order of the assignments does not matter) */
static inline void nd_fill_node1(const ndlev *L, double *a, const int per[3], int i, int j, int k)
{
  const int *n = L->n;
  int q[3] = { i, j, k }, s[3] = { i, j, k }, g = 0, zero = 0;
  for (int d = 0; d < L->dm; d++) {
    if (per[d]) { if (q[d] < 0) { s[d] = q[d] + n[d]; g = 1; } else if (q[d] >= n[d]) { s[d] = q[d] - n[d]; g = 1; } }
    else if (q[d] < 0 || q[d] > n[d]) { g = 1; zero = 1; }
  }
  if (g) a[NN(L, i, j, k)] = zero ? 0.0 : a[NN(L, s[0], s[1], s[2])];
}
static void nd_fill_nodes(const ndlev *L, double *a, const int per[3])
{
  const int *n = L->n;
  /* without a periodic direction the rule only writes zeros into ghost nodes, and those are zero already: every node array of a level is calloc'ed and the
   * sweeps, residuals, restrictions and prolongations write nodes 0 .. n only (the HIP levels rely on the same: ghosts zeroed at set-up, mg_nd.hip) */
  if (!per[0] && !per[1] && !(L->dm == 3 && per[2])) return;
  const int k0 = L->dm == 2 ? 0 : -1, k1 = L->dm == 2 ? 0 : n[2] + 1;
  #pragma omp parallel for
  for (int k = k0; k <= k1; k++) for (int j = -1; j <= n[1] + 1; j++) {
    const int edge = (j < 0 || j >= n[1]) || (L->dm == 3 && (k < 0 || k >= n[2]));
    if (edge) for (int i = -1; i <= n[0] + 1; i++) nd_fill_node1(L, a, per, i, j, k);
    else { nd_fill_node1(L, a, per, -1, j, k); nd_fill_node1(L, a, per, n[0], j, k); nd_fill_node1(L, a, per, n[0] + 1, j, k); }
  }
}
static inline void nd_fill_cell1(const ndlev *L, double *a, const int per[3], int i, int j, int k)
{
  const int *n = L->n;
  int q[3] = { i, j, k }, s[3] = { i, j, k }, g = 0, zero = 0;
  for (int d = 0; d < L->dm; d++) {
    if (q[d] < 0) { g = 1; if (per[d]) s[d] = q[d] + n[d]; else zero = 1; }
    else if (q[d] >= n[d]) { g = 1; if (per[d]) s[d] = q[d] - n[d]; else zero = 1; }
  }
  if (g) a[NS(L, i, j, k)] = zero ? 0.0 : a[NS(L, s[0], s[1], s[2])];
}
static void nd_fill_cells(const ndlev *L, double *a, const int per[3])
{
  const int *n = L->n;
  const int k0 = L->dm == 2 ? 0 : -1, k1 = L->dm == 2 ? 0 : n[2];
  #pragma omp parallel for
  for (int k = k0; k <= k1; k++) for (int j = -1; j <= n[1]; j++) {
    const int edge = (j < 0 || j >= n[1]) || (L->dm == 3 && (k < 0 || k >= n[2]));
    if (edge) for (int i = -1; i <= n[0]; i++) nd_fill_cell1(L, a, per, i, j, k);
    else { nd_fill_cell1(L, a, per, -1, j, k); nd_fill_cell1(L, a, per, n[0], j, k); }
  }
}

/* the 27-point nodal operator on gathered values: p[oc][ob][oa] = phi at node offset (oa-1, ob-1, oc-1), sg[dk][dj][di] = sigma of
 * cell (i-1+di, j-1+dj, k-1+dk); f[d] = 1/(36 h_d^2).  See nd_apply for the formula. */
/* iso: 1 = the face weights w1, w2, w4 are known to be zero (hx = hy = hz), 0 = known not all zero, -1 = test them here */
static inline __attribute__((always_inline)) void nd_stencil_inl(const double f[3], const double p[3][3][3], const double sg[2][2][2], double *Kp, double *diag, const int iso)
{
  const double fx = f[0], fy = f[1], fz = f[2];
  const double F = fx + fy + fz;
  const double w0 = 4.0 * F;
  const double w1 = -4.0 * fx + 2.0 * fy + 2.0 * fz;
  const double w2 = 2.0 * fx - 4.0 * fy + 2.0 * fz;
  const double w3 = -2.0 * fx - 2.0 * fy + fz;
  const double w4 = 2.0 * fx + 2.0 * fy - 4.0 * fz;
  const double w5 = -2.0 * fx + fy - 2.0 * fz;
  const double w6 = fx - 2.0 * fy - 2.0 * fz;
  const double w7 = -F;
  /* cz[b][a] = sg[0][b][a] + sg[1][b][a]: the two cells that share an xy-diagonal neighbour; cy[dk][di] = sg[dk][0][di] + sg[dk][1][di] (xz-diagonal);
   * cx[dk][dj] = sg[dk][dj][0] + sg[dk][dj][1] (yz-diagonal) -- written out so that the row loop of nd_row has no inner loop */
  const double cz[2][2] = { { sg[0][0][0] + sg[1][0][0], sg[0][0][1] + sg[1][0][1] }, { sg[0][1][0] + sg[1][1][0], sg[0][1][1] + sg[1][1][1] } };
  const double cy[2][2] = { { sg[0][0][0] + sg[0][1][0], sg[0][0][1] + sg[0][1][1] }, { sg[1][0][0] + sg[1][1][0], sg[1][0][1] + sg[1][1][1] } };
  const double cx[2][2] = { { sg[0][0][0] + sg[0][0][1], sg[0][1][0] + sg[0][1][1] }, { sg[1][0][0] + sg[1][0][1], sg[1][1][0] + sg[1][1][1] } };
  const double S8 = (cz[0][0] + cz[0][1]) + (cz[1][0] + cz[1][1]);
  /* the weights of a row sum to zero (K 1 = 0), so K phi = sum of coefficient * (phi_neighbour - phi_node): differences first, which
   * keeps the terms at the size of the answer instead of the size of diag * phi (at 256^3 the plain sum stalls at a residual of
   * ~2e-12 |rhs|, short of the 1e-12 of hgproject.f90:113-114) */
  const double p0 = p[1][1][1];
  #define D(c, b, a) (p[c][b][a] - p0)
  double A7 = sg[0][0][0] * D(0, 0, 0);
  A7 = fma(sg[0][0][1], D(0, 0, 2), A7); A7 = fma(sg[0][1][0], D(0, 2, 0), A7); A7 = fma(sg[0][1][1], D(0, 2, 2), A7);
  A7 = fma(sg[1][0][0], D(2, 0, 0), A7); A7 = fma(sg[1][0][1], D(2, 0, 2), A7); A7 = fma(sg[1][1][0], D(2, 2, 0), A7); A7 = fma(sg[1][1][1], D(2, 2, 2), A7);
  double A3 = cz[0][0] * D(1, 0, 0); A3 = fma(cz[0][1], D(1, 0, 2), A3); A3 = fma(cz[1][0], D(1, 2, 0), A3); A3 = fma(cz[1][1], D(1, 2, 2), A3);
  double A5 = cy[0][0] * D(0, 1, 0); A5 = fma(cy[0][1], D(0, 1, 2), A5); A5 = fma(cy[1][0], D(2, 1, 0), A5); A5 = fma(cy[1][1], D(2, 1, 2), A5);
  double A6 = cx[0][0] * D(0, 0, 1); A6 = fma(cx[0][1], D(0, 2, 1), A6); A6 = fma(cx[1][0], D(2, 0, 1), A6); A6 = fma(cx[1][1], D(2, 2, 1), A6);
  const double dg = w0 * S8;
  double acc = w3 * A3;
  acc = fma(w5, A5, acc); acc = fma(w6, A6, acc); acc = fma(w7, A7, acc);
  if (iso < 0 ? !(w1 == 0.0 && w2 == 0.0 && w4 == 0.0) : !iso) {
    double A1 = (cz[0][0] + cz[1][0]) * D(1, 1, 0); A1 = fma(cz[0][1] + cz[1][1], D(1, 1, 2), A1);
    double A2 = (cz[0][0] + cz[0][1]) * D(1, 0, 1); A2 = fma(cz[1][0] + cz[1][1], D(1, 2, 1), A2);
    double A4 = (cy[0][0] + cy[0][1]) * D(0, 1, 1); A4 = fma(cy[1][0] + cy[1][1], D(2, 1, 1), A4);
    acc = fma(w1, A1, acc); acc = fma(w2, A2, acc); acc = fma(w4, A4, acc);
  }
  #undef D
  *Kp = acc; *diag = dg;
}
void vo_nd_stencil(const double f[3], const double p[3][3][3], const double sg[2][2][2], double *Kp, double *diag) { nd_stencil_inl(f, p, sg, Kp, diag, -1); }
static int nd_iso_weights(const double f[3]) {      /* the test of nd_stencil_inl on the same expressions */
  const double fx = f[0], fy = f[1], fz = f[2];
  const double w1 = -4.0 * fx + 2.0 * fy + 2.0 * fz, w2 = 2.0 * fx - 4.0 * fy + 2.0 * fz, w4 = 2.0 * fx + 2.0 * fy - 4.0 * fz;
  return w1 == 0.0 && w2 == 0.0 && w4 == 0.0;
}

/* K phi at node (i,j,k) and the diagonal; fixed expression order shared with the HIP kernel:
 * cells in the order (ck,cj,ci) ascending; inside a cell corners (mz,my,mx) ascending. */
static inline void nd_apply(const ndlev *L, const double *phi, int i, int j, int k, double *Kp, double *diag)
{
  if (L->dm == 2) {
    /* bilinear (Q1) elements in 2-D, equations scaled by 1/(hx hy): weights by which coordinates differ,
     * w = 2(fx+fy) (same node), -2fx+fy (x differs), fx-2fy (y differs), -(fx+fy) (both), f_d = 1/(6 h_d^2) */
    const double fx = L->f[0], fy = L->f[1];
    double w[4];
    w[0] = 2.0 * (fx + fy); w[1] = -2.0 * fx + fy; w[2] = fx - 2.0 * fy; w[3] = -(fx + fy);
    double acc = 0.0, ssum = 0.0;
    for (int cj = j - 1; cj <= j; cj++) for (int ci = i - 1; ci <= i; ci++) {
      double sg = L->sig[NS(L, ci, cj, 0)];
      double t = 0.0;
      for (int my = 0; my < 2; my++) for (int mx = 0; mx < 2; mx++) {
        int ni = ci + mx, nj = cj + my;
        int idx = (ni != i) | ((nj != j) << 1);
        t = t + w[idx] * phi[NN(L, ni, nj, 0)];
      }
      acc = acc + sg * t;
      ssum = ssum + sg;
    }
    *Kp = acc; *diag = w[0] * ssum;
    return;
  }
  /* 3-D, trilinear (Q1) elements, equations scaled by 1/(hx hy hz).  K phi = sum over the 8 cells c around the node of
   * sigma_c * sum over the cell's 8 corners q of w[type(q)] phi_q, type = which coordinates of q differ from the node's (bit 0 x, 1 y,
   * 2 z).  Evaluated GROUPED BY NEIGHBOUR TYPE (round 2; the same sum, 41-55 operations instead of 142): a face / edge / corner
   * neighbour is shared by 4 / 2 / 1 of the cells, so its coefficient is w[type] times the sum of those sigmas.  Fixed order, explicit
   * fma() -- oracle and HIP (nd_stencil in mg_nd.hip) run the same operation sequence, hence the same bits.  With hx = hy = hz the
   * face weights w[1], w[2], w[4] are exactly zero (the 21-point stencil, hg_hypre.f90:100-113) and their terms are skipped. */
  /* (round 4: the 27 + 8 values through one base index each and constant strides) */
  const long sy = L->n[0] + 3, sz = sy * (L->n[1] + 3), ty = L->n[0] + 2, tz = ty * (L->n[1] + 2);
  const double *pp = phi + NN(L, i - 1, j - 1, k - 1), *ps = L->sig + NS(L, i - 1, j - 1, k - 1);
  double p[3][3][3], sg[2][2][2];
  for (int c = 0; c < 3; c++) for (int b = 0; b < 3; b++) for (int a = 0; a < 3; a++) p[c][b][a] = pp[a + b * sy + c * sz];
  for (int c = 0; c < 2; c++) for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) sg[c][b][a] = ps[a + b * ty + c * tz];
  nd_stencil_inl(L->f, p, sg, Kp, diag, -1);
}

/* K phi and the diagonal on the whole row (0..n0, j, k) of a 3-D level: the same nd_stencil_inl on the same 27 + 8 values per node, the loop over i written
 * so that the compiler vectorises it (unit-stride loads; round 4 -- the oracle's nodal solve was 3.5 of the 6.9 s of the bench's CPU step) */
static inline __attribute__((always_inline)) void nd_row_(const ndlev *L, const double *phi, int j, int k, double *restrict Kp, double *restrict dg, const int iso)
{
  const long sy = L->n[0] + 3, sz = sy * (L->n[1] + 3), ty = L->n[0] + 2, tz = ty * (L->n[1] + 2);
  const double *pp = phi + NN(L, -1, j - 1, k - 1), *ps = L->sig + NS(L, -1, j - 1, k - 1);
  const double f[3] = { L->f[0], L->f[1], L->f[2] };
  const int n0 = L->n[0];
  #pragma omp simd
  for (int i = 0; i <= n0; i++) {
    /* (written out: the vectoriser does not take a loop nest with inner loops) */
    const double p[3][3][3] = { { { pp[i + 0 + 0 * sy + 0 * sz], pp[i + 1 + 0 * sy + 0 * sz], pp[i + 2 + 0 * sy + 0 * sz] }, { pp[i + 0 + 1 * sy + 0 * sz], pp[i + 1 + 1 * sy + 0 * sz], pp[i + 2 + 1 * sy + 0 * sz] }, { pp[i + 0 + 2 * sy + 0 * sz], pp[i + 1 + 2 * sy + 0 * sz], pp[i + 2 + 2 * sy + 0 * sz] } }, { { pp[i + 0 + 0 * sy + 1 * sz], pp[i + 1 + 0 * sy + 1 * sz], pp[i + 2 + 0 * sy + 1 * sz] }, { pp[i + 0 + 1 * sy + 1 * sz], pp[i + 1 + 1 * sy + 1 * sz], pp[i + 2 + 1 * sy + 1 * sz] }, { pp[i + 0 + 2 * sy + 1 * sz], pp[i + 1 + 2 * sy + 1 * sz], pp[i + 2 + 2 * sy + 1 * sz] } }, { { pp[i + 0 + 0 * sy + 2 * sz], pp[i + 1 + 0 * sy + 2 * sz], pp[i + 2 + 0 * sy + 2 * sz] }, { pp[i + 0 + 1 * sy + 2 * sz], pp[i + 1 + 1 * sy + 2 * sz], pp[i + 2 + 1 * sy + 2 * sz] }, { pp[i + 0 + 2 * sy + 2 * sz], pp[i + 1 + 2 * sy + 2 * sz], pp[i + 2 + 2 * sy + 2 * sz] } } };
    const double sg[2][2][2] = { { { ps[i + 0 + 0 * ty + 0 * tz], ps[i + 1 + 0 * ty + 0 * tz] }, { ps[i + 0 + 1 * ty + 0 * tz], ps[i + 1 + 1 * ty + 0 * tz] } }, { { ps[i + 0 + 0 * ty + 1 * tz], ps[i + 1 + 0 * ty + 1 * tz] }, { ps[i + 0 + 1 * ty + 1 * tz], ps[i + 1 + 1 * ty + 1 * tz] } } };
    double kp, d;
    nd_stencil_inl(f, p, sg, &kp, &d, iso);
    Kp[i] = kp; dg[i] = d;
  }
}
static void nd_row(const ndlev *L, const double *phi, int j, int k, double *restrict Kp, double *restrict dg)
{
  if (nd_iso_weights(L->f)) nd_row_(L, phi, j, k, Kp, dg, 1); else nd_row_(L, phi, j, k, Kp, dg, 0);
}
#define ND_ROW_MAX 2048        /* longest row the row form takes (stack buffers); longer rows and dm = 2 go node by node */
static void nd_jacobi(ndlev *L, const int per[3], int nsweeps, double omega)
{
  const int *n = L->n;
  for (int s = 0; s < nsweeps; s++) {
    nd_fill_nodes(L, L->phi, per);
    if (L->dm == 3 && n[0] + 1 <= ND_ROW_MAX) {
      #pragma omp parallel for collapse(2) schedule(static)
      for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) {
        double Kp[ND_ROW_MAX], dg[ND_ROW_MAX];
        nd_row(L, L->phi, j, k, Kp, dg);
        const double *ph = L->phi + NN(L, 0, j, k), *bb = L->b + NN(L, 0, j, k); double *out = L->tmp + NN(L, 0, j, k);
        const unsigned char *dr = L->dir + NM(L, 0, j, k);
        for (int i = 0; i <= n[0]; i++) {
          double p0 = ph[i], v = p0;
          if (!dr[i] && dg[i] != 0.0) v = p0 + omega * ((bb[i] - Kp[i]) / dg[i]);
          out[i] = v;
        }
      }
    } else {
    #pragma omp parallel for collapse(2) schedule(static)
    for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++) {
      double p0 = L->phi[NN(L, i, j, k)], v = p0;
      if (!L->dir[NM(L, i, j, k)]) {
        double Kp, diag; nd_apply(L, L->phi, i, j, k, &Kp, &diag);
        if (diag != 0.0) v = p0 + omega * ((L->b[NN(L, i, j, k)] - Kp) / diag);
      }
      L->tmp[NN(L, i, j, k)] = v;
    }
    }
    double *t = L->phi; L->phi = L->tmp; L->tmp = t;
  }
}

/* the nu1 pre-smoothing sweeps of a V-cycle.  With nu1 = 2 and a damping pair given (vdn_params.hg_omega_pre1 / 2; round 3) the first sweep is damped by
 * pre[0] and the second by pre[1]: two Jacobi sweeps with the damping factors 1.45 and 0.7 reduce the error like the polynomial (1 - 1.45 t)(1 - 0.7 t)
 * in t = an eigenvalue of D^-1 K -- a two-step Chebyshev smoother for t in about [0.45, 1.7], of modulus < 1 up to t = 2.1 -- where two sweeps at 0.9
 * give (1 - 0.9 t)^2: 11 -> 10 V-cycles at 64^3, 10 -> 9 at 256^3, the residual after ten cycles 5 x smaller at 128^3.  Three dimensions only. */
static void nd_presmooth(ndlev *L, const int per[3], int nu1, double omega, const double *pre)
{
  if (pre && nu1 == 2 && L->dm == 3) { nd_jacobi(L, per, 1, pre[0]); nd_jacobi(L, per, 1, pre[1]); }
  else nd_jacobi(L, per, nu1, omega);
}

static double nd_residual(ndlev *L, const int per[3])
{
  const int *n = L->n; double nrm = 0.0;
  nd_fill_nodes(L, L->phi, per);
  if (L->dm == 3 && n[0] + 1 <= ND_ROW_MAX) {
    #pragma omp parallel for collapse(2) schedule(static) reduction(max : nrm)
    for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) {
      double Kp[ND_ROW_MAX], dg[ND_ROW_MAX];
      nd_row(L, L->phi, j, k, Kp, dg);
      const double *bb = L->b + NN(L, 0, j, k); double *out = L->res + NN(L, 0, j, k);
      const unsigned char *dr = L->dir + NM(L, 0, j, k);
      for (int i = 0; i <= n[0]; i++) {
        const double r = dr[i] ? 0.0 : bb[i] - Kp[i];
        out[i] = r;
        nrm = vo_nrm_acc(nrm, r);
      }
    }
  } else
  #pragma omp parallel for collapse(2) schedule(static) reduction(max : nrm)
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++) {
    double r = 0.0;
    if (!L->dir[NM(L, i, j, k)]) {
      double Kp, diag; nd_apply(L, L->phi, i, j, k, &Kp, &diag);
      r = L->b[NN(L, i, j, k)] - Kp;
    }
    L->res[NN(L, i, j, k)] = r;
    nrm = vo_nrm_acc(nrm, r);
  }
  nd_fill_nodes(L, L->res, per);
  return nrm;
}

static void nd_restrict(const ndlev *Fv, ndlev *C)
{
  const int *n = C->n;
  const double wt[3] = { 0.5, 1.0, 0.5 };
  #pragma omp parallel for
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++) {
    double s = 0.0;
    if (C->dm == 2) {                        /* full weighting in the plane: P^T / 4 */
      if (!C->dir[NM(C, i, j, k)])
        for (int b = -1; b <= 1; b++) for (int a = -1; a <= 1; a++)
          s = s + (wt[a + 1] * wt[b + 1]) * Fv->res[NN(Fv, 2 * i + a, 2 * j + b, 0)];
      C->b[NN(C, i, j, k)] = s * 0.25;
      continue;
    }
    /* full weighting, separably and in this order (round 4; HIP: nd_fw27): along x on each of the nine lines, X = (0.5 r[-1] + r[0]) + 0.5 r[+1]; along z on
     * each of the three rows of X; along y last -- the order in which the HIP residual march can form the sums from the lanes and planes it holds */
    if (!C->dir[NM(C, i, j, k)]) {
      double xz[3];
      for (int b = -1; b <= 1; b++) {
        double x[3];
        for (int c = -1; c <= 1; c++) {
          const double *q = Fv->res + NN(Fv, 2 * i, 2 * j + b, 2 * k + c);
          x[c + 1] = (0.5 * q[-1] + q[0]) + 0.5 * q[1];
        }
        xz[b + 1] = (0.5 * x[0] + x[1]) + 0.5 * x[2];
      }
      s = (0.5 * xz[0] + xz[1]) + 0.5 * xz[2];
    }
    C->b[NN(C, i, j, k)] = s * 0.125;
  }
}

static void nd_prolong_add(ndlev *Fv, const ndlev *C)
{
  const int *n = Fv->n;
  #pragma omp parallel for
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++) {
    if (Fv->dir[NM(Fv, i, j, k)]) continue;
    int I = i >> 1, J = j >> 1, K = k >> 1, oi = i & 1, oj = j & 1, ok = k & 1;
    double s = 0.0;
    for (int c = 0; c <= ok; c++) for (int b = 0; b <= oj; b++) for (int a = 0; a <= oi; a++)
      s = s + C->phi[NN(C, I + a, J + b, K + c)];
    double scale = 1.0 / (double)((1 + oi) * (1 + oj) * (1 + ok));
    Fv->phi[NN(Fv, i, j, k)] = Fv->phi[NN(Fv, i, j, k)] + s * scale;
  }
}

static void nd_coarsen_sigma(const ndlev *Fv, ndlev *C, const int per[3])
{
  const int *n = C->n;
  #pragma omp parallel for
  for (int k = 0; k < (C->dm == 2 ? 1 : n[2]); k++) for (int j = 0; j < n[1]; j++) for (int i = 0; i < n[0]; i++) {
    double s = 0.0;
    if (C->dm == 2) {
      for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) s = s + Fv->sig[NS(Fv, 2 * i + a, 2 * j + b, 0)];
      C->sig[NS(C, i, j, k)] = s * 0.25;
      continue;
    }
    for (int c = 0; c < 2; c++) for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++)
      s = s + Fv->sig[NS(Fv, 2 * i + a, 2 * j + b, 2 * k + c)];
    C->sig[NS(C, i, j, k)] = s * 0.125;
  }
  nd_fill_cells(C, C->sig, per);
}

static void nd_set_mask(ndlev *L, const int ellbc[3][2])
{
  const int *n = L->n;
  #pragma omp parallel for
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++) {
    int q[3] = { i, j, k }, dflag = 0;
    for (int d = 0; d < L->dm; d++) {
      if (q[d] == 0 && ellbc[d][0] == VDN_BC_DIR) dflag = 1;
      if (q[d] == n[d] && ellbc[d][1] == VDN_BC_DIR) dflag = 1;
    }
    L->dir[NM(L, i, j, k)] = (unsigned char)dflag;
  }
}

typedef struct ndmg { int nlev; ndlev lev[32]; int per[3]; } ndmg;

/* Jacobi sweeps on the coarsest level: max(nub, 2 N^2), N = its largest extent (see cc_bottom_sweeps) */
static int nd_bottom_sweeps(const ndlev *L, int nub)
{
  int N = L->n[0] > L->n[1] ? L->n[0] : L->n[1]; if (L->dm == 3 && L->n[2] > N) N = L->n[2];
  return nub > 2 * N * N ? nub : 2 * N * N;
}

static void nd_vcycle(ndmg *M, int l, int nu1, int nu2, int nub, double omega, const double *pre)
{
  ndlev *L = &M->lev[l];
  long nn = (long)(L->n[0] + 3) * (L->n[1] + 3) * (L->n[2] + 3);
  memset(L->phi, 0, sizeof(double) * nn);
  if (l == M->nlev - 1) { nd_jacobi(L, M->per, nd_bottom_sweeps(L, nub), omega); return; }
  nd_presmooth(L, M->per, nu1, omega, pre);
  (void)nd_residual(L, M->per);
  nd_restrict(L, &M->lev[l + 1]);
  nd_vcycle(M, l + 1, nu1, nu2, nub, omega, pre);
  nd_fill_nodes(&M->lev[l + 1], M->lev[l + 1].phi, M->per);
  nd_prolong_add(L, &M->lev[l + 1]);
  nd_jacobi(L, M->per, nu2, omega);
}

/* nodal divergence, our definition (see header comment); u must have >= 1 ghost cell */
void vo_nd_divu(const vo_fab *u, vo_fab *rh, const double dx[3], const int ellbc[3][2])
{
  const int *lo = u->lo, *hi = u->hi;
  if (u->dm == 2) {      /* (D u)_n = ([u_d over the 2 cells on the + side] - [- side]) * 0.5/h_d */
    double gx = 0.5 / dx[0], gy = 0.5 / dx[1];
    for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0] + 1; i++) {
      double dux = (V2(u, i, j, 0) + V2(u, i, j - 1, 0)) - (V2(u, i - 1, j, 0) + V2(u, i - 1, j - 1, 0));
      double duy = (V2(u, i, j, 1) + V2(u, i - 1, j, 1)) - (V2(u, i, j - 1, 1) + V2(u, i - 1, j - 1, 1));
      V2(rh, i, j, 0) = V2(rh, i, j, 0) + (dux * gx + duy * gy);
    }
    return;
  }
  double fx = 0.25 / dx[0], fy = 0.25 / dx[1], fz = 0.25 / dx[2];
  (void)ellbc;
  #pragma omp parallel for
  for (int k = lo[2]; k <= hi[2] + 1; k++) for (int j = lo[1]; j <= hi[1] + 1; j++) for (int i = lo[0]; i <= hi[0] + 1; i++) {
    #define U(a, b, c, m) VF(u, i + (a), j + (b), k + (c), m)
    double dux = (((U(0, 0, 0, 0) + U(0, -1, 0, 0)) + U(0, 0, -1, 0)) + U(0, -1, -1, 0))
               - (((U(-1, 0, 0, 0) + U(-1, -1, 0, 0)) + U(-1, 0, -1, 0)) + U(-1, -1, -1, 0));
    double duy = (((U(0, 0, 0, 1) + U(-1, 0, 0, 1)) + U(0, 0, -1, 1)) + U(-1, 0, -1, 1))
               - (((U(0, -1, 0, 1) + U(-1, -1, 0, 1)) + U(0, -1, -1, 1)) + U(-1, -1, -1, 1));
    double duz = (((U(0, 0, 0, 2) + U(-1, 0, 0, 2)) + U(0, -1, 0, 2)) + U(-1, -1, 0, 2))
               - (((U(0, 0, -1, 2) + U(-1, 0, -1, 2)) + U(0, -1, -1, 2)) + U(-1, -1, -1, 2));
    #undef U
    VF(rh, i, j, k, 0) = VF(rh, i, j, k, 0) + (dux * fx + duy * fy + duz * fz);
  }
}

int vo_nd_solve(vo_fab *rh, vo_fab *phi, const vo_fab *coeffs, const vo_fab *u, const double dx[3],
                const int ellbc[3][2], const int pmask[3], double rel_eps, double abs_eps, int max_iter,
                int nu1, int nu2, int nub, double omega, int fmg, const double *om_pre, vo_mgstat *st)
{
  if (!vo_nd_isotropic(dx, coeffs->dm)) om_pre = NULL;     /* the damping pair was tuned for dx = dy = dz: plain hg_omega on stretched grids (round 4) */
  ndmg M; M.nlev = 0;
  int n[3]; double h[3];
  const int dm = coeffs->dm;
  for (int d = 0; d < 3; d++) { n[d] = coeffs->hi[d] - coeffs->lo[d] + 1; h[d] = dx[d]; M.per[d] = (d < dm) && pmask[d]; }
  if (dm == 2) { n[2] = 0; h[2] = 1.0; }
  for (;;) {
    ndlev *L = &M.lev[M.nlev];
    nd_alloc(L, n, h, dm);
    nd_set_mask(L, ellbc);
    if (M.nlev == 0) {
      for (int k = (dm == 2 ? 0 : -1); k <= (dm == 2 ? 0 : n[2]); k++) for (int j = -1; j <= n[1]; j++) for (int i = -1; i <= n[0]; i++)
        L->sig[NS(L, i, j, k)] = VF(coeffs, coeffs->lo[0] + i, coeffs->lo[1] + j, coeffs->lo[2] + k, 0);
    } else nd_coarsen_sigma(&M.lev[M.nlev - 1], L, M.per);
    M.nlev++;
    int can = 1;
    for (int d = 0; d < dm; d++) if ((n[d] & 1) || n[d] <= 2) can = 0;
    if (!can || M.nlev >= 31) break;
    for (int d = 0; d < dm; d++) { n[d] /= 2; h[d] *= 2.0; }
  }
  ndlev *L0 = &M.lev[0];
  const int *n0 = L0->n;
  if (u) vo_nd_divu(u, rh, dx, ellbc);                 /* add_divu = .true. (hg_multigrid.f90:96) */
  double bnorm = 0.0;
  for (int k = 0; k <= n0[2]; k++) for (int j = 0; j <= n0[1]; j++) for (int i = 0; i <= n0[0]; i++) {
    double r = VF(rh, rh->lo[0] + i, rh->lo[1] + j, rh->lo[2] + k, 0);
    if (L0->dir[NM(L0, i, j, k)]) r = 0.0;
    L0->b[NN(L0, i, j, k)] = -r;
    L0->phi[NN(L0, i, j, k)] = L0->dir[NM(L0, i, j, k)] ? 0.0 : VF(phi, phi->lo[0] + i, phi->lo[1] + j, phi->lo[2] + k, 0);
    bnorm = vo_nrm_acc(bnorm, r);
  }
  int cyc = 0, conv = 0; double rn = 0.0;
  if (bnorm == 0.0) conv = 1;
  /* Nested iteration for the initial guess (round 3; hg_fmg): when the incoming phi is zero everywhere, the right-hand side is restricted
   * down the hierarchy (full weighting, the residual's operator), the coarsest level with more than 9^3 nodes -- below it a level is a few
   * hundred unknowns -- gets TWO V-cycles from zero, and every level above it takes the interpolated solution of the level below (the
   * prolongation of the cycle) and, except the finest, one V-cycle of its own.  The V-cycles that follow start from an error at truncation
   * level instead of 100 %: 13 -> 11 cycles to 1e-12 at 64^3 (12 with one cycle on the starting level), for the price of ~0.2 of a
   * fine-level cycle.  Three dimensions only.  fmg = 1: only with a convergence test (max_iter >= 0); fmg = 2: also before a fixed number of
   * cycles (the first coarse correction of the composite solve). */
  if (fmg && dm == 3 && M.nlev > 1 && (max_iter >= 0 || fmg == 2) && !conv) {
    int zero = 1;
    for (int k = 0; k <= n0[2] && zero; k++) for (int j = 0; j <= n0[1] && zero; j++) for (int i = 0; i <= n0[0]; i++)
      if (L0->phi[NN(L0, i, j, k)] != 0.0) { zero = 0; break; }
    int ls = -1;
    for (int l = 1; l < M.nlev; l++) if ((long)(M.lev[l].n[0] + 1) * (M.lev[l].n[1] + 1) * (M.lev[l].n[2] + 1) > 729) ls = l;
    if (zero && ls >= 1 && ls + 1 < M.nlev) {      /* (a starting level with nothing below it: no nested iteration) */
      for (int l = 0; l < ls; l++) {                   /* b_{l+1} = R b_l */
        ndlev *Lf = &M.lev[l];
        for (int k = 0; k <= Lf->n[2]; k++) for (int j = 0; j <= Lf->n[1]; j++) for (int i = 0; i <= Lf->n[0]; i++)
          Lf->res[NN(Lf, i, j, k)] = Lf->b[NN(Lf, i, j, k)];
        nd_fill_nodes(Lf, Lf->res, M.per);
        nd_restrict(Lf, &M.lev[l + 1]);
      }
      nd_vcycle(&M, ls, nu1, nu2, nub, omega, om_pre);        /* from zero */
      for (int l = ls; l >= 0; l--) {
        ndlev *Lf = &M.lev[l];
        if (l < ls) {                                  /* the interpolated solution of the level below */
          long nn = (long)(Lf->n[0] + 3) * (Lf->n[1] + 3) * (Lf->n[2] + 3);
          memset(Lf->phi, 0, sizeof(double) * nn);
          nd_fill_nodes(&M.lev[l + 1], M.lev[l + 1].phi, M.per);
          nd_prolong_add(Lf, &M.lev[l + 1]);
        }
        if (l > 0) {                                   /* one V-cycle on that guess (the starting level: its second) */
          nd_presmooth(Lf, M.per, nu1, omega, om_pre);
          (void)nd_residual(Lf, M.per);
          nd_restrict(Lf, &M.lev[l + 1]);
          nd_vcycle(&M, l + 1, nu1, nu2, nub, omega, om_pre);
          nd_fill_nodes(&M.lev[l + 1], M.lev[l + 1].phi, M.per);
          nd_prolong_add(Lf, &M.lev[l + 1]);
          nd_jacobi(Lf, M.per, nu2, omega);
        }
      }
    }
  }
  if (max_iter < 0) {            /* exactly -max_iter V-cycles, no convergence test (coarse correction of the composite solve) */
    for (int c = 0; c < -max_iter; c++) {
      if (M.nlev == 1) { nd_jacobi(L0, M.per, nd_bottom_sweeps(L0, nub), omega); continue; }
      nd_presmooth(L0, M.per, nu1, omega, om_pre);
      (void)nd_residual(L0, M.per);
      nd_restrict(L0, &M.lev[1]);
      nd_vcycle(&M, 1, nu1, nu2, nub, omega, om_pre);
      nd_fill_nodes(&M.lev[1], M.lev[1].phi, M.per);
      nd_prolong_add(L0, &M.lev[1]);
      nd_jacobi(L0, M.per, nu2, omega);
      cyc++;
    }
    conv = 1;
  }
  while (!conv) {
    if (M.nlev == 1) nd_jacobi(L0, M.per, nd_bottom_sweeps(L0, nub), omega); else nd_presmooth(L0, M.per, nu1, omega, om_pre);
    rn = nd_residual(L0, M.per);
    if (rn <= rel_eps * bnorm || rn <= abs_eps) { conv = 1; break; }
    if (cyc >= max_iter) break;
    if (M.nlev > 1) {
      nd_restrict(L0, &M.lev[1]);
      nd_vcycle(&M, 1, nu1, nu2, nub, omega, om_pre);
      nd_fill_nodes(&M.lev[1], M.lev[1].phi, M.per);
      nd_prolong_add(L0, &M.lev[1]);
      nd_jacobi(L0, M.per, nu2, omega);
    }
    cyc++;
  }
  nd_fill_nodes(L0, L0->phi, M.per);
  for (int k = (dm == 2 ? 0 : -1); k <= (dm == 2 ? 0 : n0[2] + 1); k++) for (int j = -1; j <= n0[1] + 1; j++) for (int i = -1; i <= n0[0] + 1; i++)
    VF(phi, phi->lo[0] + i, phi->lo[1] + j, phi->lo[2] + k, 0) = L0->phi[NN(L0, i, j, k)];
  if (st) { st->cycles = cyc; st->res0 = bnorm; st->res = rn; }
  for (int l = 0; l < M.nlev; l++) nd_free(&M.lev[l]);
  return conv ? 0 : 1;
}

/* test hook: out = K phi on the nodes of one level (0 on Dirichlet nodes), K the operator of nd_apply -- what every sweep and residual of the nodal
 * solver applies -- once.  tests/test_operators_assembled_cpu.py compares it with the Q1 stiffness matrix assembled element by element. */
void vo_nd_apply(const vo_fab *phi, const vo_fab *coeffs, const double dx[3], const int ellbc[3][2], const int pmask[3], vo_fab *out)
{
  ndlev Lv, *L = &Lv; int n[3], per[3];
  for (int d = 0; d < 3; d++) { n[d] = coeffs->hi[d] - coeffs->lo[d] + 1; per[d] = pmask[d]; }
  nd_alloc(L, n, dx, 3); nd_set_mask(L, ellbc);
  for (int k = -1; k <= n[2]; k++) for (int j = -1; j <= n[1]; j++) for (int i = -1; i <= n[0]; i++)
    L->sig[NS(L, i, j, k)] = VF(coeffs, coeffs->lo[0] + i, coeffs->lo[1] + j, coeffs->lo[2] + k, 0);
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++)
    L->phi[NN(L, i, j, k)] = L->dir[NM(L, i, j, k)] ? 0.0 : VF(phi, phi->lo[0] + i, phi->lo[1] + j, phi->lo[2] + k, 0);
  nd_fill_nodes(L, L->phi, per);
  for (int k = 0; k <= n[2]; k++) for (int j = 0; j <= n[1]; j++) for (int i = 0; i <= n[0]; i++) {
    double Kp = 0.0, diag;
    if (!L->dir[NM(L, i, j, k)]) nd_apply(L, L->phi, i, j, k, &Kp, &diag);
    VF(out, out->lo[0] + i, out->lo[1] + j, out->lo[2] + k, 0) = Kp;
  }
  nd_free(L);
}

/* hgproject.f90:17-178 with hg_multigrid.f90:18-119, one level / one box */
void vo_hgproject(int proj_type, vo_fab *unew, const vo_fab *uold, vo_fab *rhohalf, vo_fab *p, vo_fab *gp,
                  const double dx[3], double dt, const vo_bc *bc, const int pmask[3], const vdn_params *prm,
                  vo_mgstat *st)
{
  const int *lo = unew->lo, *hi = unew->hi;
  int nd0[3] = { 0, 0, 0 }, nd1[3] = { 1, 1, 1 };
  vo_fab rh, phi, gphi, coeffs;
  vo_fab_init(&rh, NULL, lo, hi, 1, nd1, 1);   rh.p = (double *)calloc(vo_size(&rh), sizeof(double));
  vo_fab_init(&phi, NULL, lo, hi, 1, nd1, 1);  phi.p = (double *)calloc(vo_size(&phi), sizeof(double));
  vo_fab_init(&gphi, NULL, lo, hi, 0, nd0, 3); gphi.p = (double *)calloc(vo_size(&gphi), sizeof(double));
  vo_fab_init(&coeffs, NULL, lo, hi, 1, nd0, 1); coeffs.p = (double *)calloc(vo_size(&coeffs), sizeof(double));
  int ellbc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ellbc[d][s] = bc->ell[d][s][bc->press_comp];

  vo_create_uvec(unew, uold, rhohalf, gp, dt, bc, proj_type);
  vo_fill_boundary(unew, pmask);                                        /* hgproject.f90:232 */

  double rel = prm->hg_rel_eps > 0.0 ? prm->hg_rel_eps : 1.e-12;       /* nlevs == 1: hgproject.f90:113-114 */
  double abs_eps = -1.0;
  if (proj_type == VDN_INITIAL_PROJECTION && prm->prob_type == 4) abs_eps = 1.e-12;   /* 125-127 */

  /* coeffs = 1/rhohalf on valid cells, ghosts 0, then fill_boundary (hg_multigrid.f90:68-79) */
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
    VF(&coeffs, i, j, k, 0) = 1.0 / VF(rhohalf, i, j, k, 0);
  vo_fill_boundary(&coeffs, pmask);

  vo_nd_solve(&rh, &phi, &coeffs, unew, dx, ellbc, pmask, rel, abs_eps, prm->hg_max_iter,
              prm->hg_nu1, prm->hg_nu2, prm->hg_nub, prm->hg_omega, prm->hg_fmg, vo_om_pre(prm), st);

  vo_mkgphi(&gphi, &phi, dx);
  vo_hg_update(proj_type, unew, uold, gp, &gphi, rhohalf, p, &phi, dt);
  vo_fill_boundary(gp, pmask); vo_fill_boundary(p, pmask);              /* hgproject.f90:359-362 */
  free(rh.p); free(phi.p); free(gphi.p); free(coeffs.p);
}

/* =============================================================================================================================
 * Two-level composite nodal solve and the multilevel hgproject (hgproject.f90:17-178, hg_multigrid.f90:18-119 with nlevs = 2).
 * FBoxLib's ml_nd_solve is not in the reference tree; our definition (finite elements on the composite mesh):
 *   unknowns      coarse nodes outside / on the boundary of the fine box, fine nodes strictly inside it; fine nodes ON the
 *                 coarse-fine interface are slaves, phi_f = P phi_c (trilinear)
 *   equations     fine interior nodes: the fine 27-point equation; other coarse nodes: coarse-side part (sigma = 0 and u = 0 in
 *                 covered cells) + full-weighting restriction (P^T/8) of the fine-side parts (sigma = 0 and u = 0 outside the
 *                 fine box) -- i.e. the Galerkin equation of the coarse hat function on the composite mesh
 *   algorithm     FAC: composite residual; one V-cycle of the single-level nodal multigrid on the whole coarse level (sigma under
 *                 the fine box = the averaged-down fine sigma); trilinear prolongation of the correction; damped-Jacobi
 *                 relaxation of the fine interior nodes with the interface held fixed.
 * One box per level.
 * ============================================================================================================================= */
static void nd_level_from_fab(ndlev *L, const vo_fab *coeffs, const double dx[3])
{
  int n[3];
  for (int d = 0; d < 3; d++) n[d] = coeffs->hi[d] - coeffs->lo[d] + 1;
  nd_alloc(L, n, dx, 3);
  for (int k = -1; k <= n[2]; k++) for (int j = -1; j <= n[1]; j++) for (int i = -1; i <= n[0]; i++)
    L->sig[NS(L, i, j, k)] = VF(coeffs, coeffs->lo[0] + i, coeffs->lo[1] + j, coeffs->lo[2] + k, 0);
}
/* copy of u with one ghost layer, zero outside the level's own cells (levels >= 1: `own`; level 0 keeps everything, its ghost cells carry the inflow
 * data) and zero in the cells covered by the next finer level */
static void masked_u(vo_fab *out, const vo_fab *u, const vo_level *Lown, int own, const vo_level *Lfine, const vo_fab *fine)
{
  vo_fab_init(out, NULL, u->lo, u->hi, 1, NULL, 3);
  out->p = (double *)calloc(vo_size(out), sizeof(double));
  for (int c = 0; c < 3; c++) for (int k = u->lo[2] - 1; k <= u->hi[2] + 1; k++) for (int j = u->lo[1] - 1; j <= u->hi[1] + 1; j++) for (int i = u->lo[0] - 1; i <= u->hi[0] + 1; i++) {
    int keep = 1;
    if (own && !vo_lv_valid(Lown, u, i, j, k)) keep = 0;
    if (fine && vo_lv_valid(Lfine, fine, 2 * i, 2 * j, 2 * k)) keep = 0;
    VF(out, i, j, k, c) = keep ? VF(u, i, j, k, c) : 0.0;
  }
}
#define ND_MAXLEV 4
typedef struct mlnd {
  int nlev;
  ndlev L[ND_MAXLEV];                 /* sig = MASKED sigma (zero in the cells covered by the next finer level) */
  double *sigfull[ND_MAXLEV];         /* full sigma (relaxation of the levels >= 1; level 0's correction solve takes the fab) */
  int per[3];                         /* all zero: a refined level stays clear of the periodic faces (vo_amr.c: require_periodic_ok) ... */
  int per0[3];                        /* ... and level 0, which spans the domain, wraps (round 6) */
  int org[ND_MAXLEV][3];              /* global index of local node 0 */
  unsigned char *under[ND_MAXLEV];    /* per node of level n, of the 8 cells around it: 1 = some are covered by level n+1 (its equation takes the restricted fine-side
                                       * parts), 2 = all are (no equation of its own: left out of the norm), 0 = none */
  unsigned char *cf[ND_MAXLEV];       /* nodes slaved to level n-1 (coarse-fine interface) */
  unsigned char *pdir[ND_MAXLEV];     /* physical Dirichlet nodes, and the nodes of the level array that touch no cell of the level (a level of several boxes) */
} mlnd;

/* trilinear interpolation of the coarse array `cp` (level n-1, local indexing of Lc) at local node (i,j,k) of level n */
static double ml_nd_interp(const mlnd *M, int n, const ndlev *Lc, const double *cp, int i, int j, int k)
{
  int gi = M->org[n][0] + i, gj = M->org[n][1] + j, gk = M->org[n][2] + k;
  int I = (gi >> 1) - M->org[n - 1][0], J = (gj >> 1) - M->org[n - 1][1], K = (gk >> 1) - M->org[n - 1][2], oi = gi & 1, oj = gj & 1, ok = gk & 1;
  double s = 0.0;
  for (int c = 0; c <= ok; c++) for (int b = 0; b <= oj; b++) for (int a = 0; a <= oi; a++) s = s + cp[NN(Lc, I + a, J + b, K + c)];
  return s * (1.0 / (double)((1 + oi) * (1 + oj) * (1 + ok)));
}
static void ml_nd_interface(mlnd *M, int n)
{
  ndlev *F = &M->L[n], *Cc = &M->L[n - 1];
  for (int k = 0; k <= F->n[2]; k++) for (int j = 0; j <= F->n[1]; j++) for (int i = 0; i <= F->n[0]; i++)
    if (M->cf[n][NM(F, i, j, k)]) F->phi[NN(F, i, j, k)] = ml_nd_interp(M, n, Cc, Cc->phi, i, j, k);
}
/* composite residual: L[n].res on every level (partial sums at the interfaces); returns the composite max-norm */
static double ml_nd_residual(mlnd *M)
{
  double nrm = 0.0;
  const double wt[3] = { 0.5, 1.0, 0.5 };
  for (int n = 1; n < M->nlev; n++) ml_nd_interface(M, n);
  for (int n = 0; n < M->nlev; n++) nd_fill_nodes(&M->L[n], M->L[n].phi, n == 0 ? M->per0 : M->per);
  for (int n = M->nlev - 1; n >= 0; n--) {
    ndlev *L = &M->L[n];
    const int has_fine = n < M->nlev - 1;
    memset(L->res, 0, sizeof(double) * (size_t)(L->n[0] + 3) * (L->n[1] + 3) * (L->n[2] + 3));
    for (int k = 0; k <= L->n[2]; k++) for (int j = 0; j <= L->n[1]; j++) for (int i = 0; i <= L->n[0]; i++) {
      double r = 0.0;
      if (!M->pdir[n][NM(L, i, j, k)]) {
        double Kp, diag; nd_apply(L, L->phi, i, j, k, &Kp, &diag);
        r = L->b[NN(L, i, j, k)] - Kp;
        if (has_fine && M->under[n][NM(L, i, j, k)]) {
          const ndlev *F = &M->L[n + 1];
          int fi = 2 * (i + M->org[n][0]) - M->org[n + 1][0], fj = 2 * (j + M->org[n][1]) - M->org[n + 1][1], fk = 2 * (k + M->org[n][2]) - M->org[n + 1][2];
          double s = 0.0;
          for (int c = -1; c <= 1; c++) for (int b = -1; b <= 1; b++) for (int a = -1; a <= 1; a++) {
            int ii = fi + a, jj = fj + b, kk = fk + c;
            if (ii < 0 || ii > F->n[0] || jj < 0 || jj > F->n[1] || kk < 0 || kk > F->n[2]) continue;
            s = s + (wt[a + 1] * wt[b + 1] * wt[c + 1]) * F->res[NN(F, ii, jj, kk)];
          }
          r = r + s * 0.125;
        }
      }
      L->res[NN(L, i, j, k)] = r;
      int skip = M->cf[n] && M->cf[n][NM(L, i, j, k)];
      if (has_fine && M->under[n][NM(L, i, j, k)] == 2) skip = 1;
      if (!skip) nrm = vo_nrm_acc(nrm, r);
    }
  }
  return nrm;
}
/* rh[lev] nodal (ng 1; in: extra source, normally 0), phi[lev] nodal (ng 1, in/out), coeffs[lev] cells (ng 1, ghost 0 outside the
 * level), u[lev] cells with >= 1 ghost (wall ghosts zeroed by create_uvec); dx: [lev*3+d]; ellbc per level (of its bounding box).
 * lev (may be NULL): the box lists -- the fields are level arrays (vo.h) and every test "inside the box" becomes one on the cells around a node:
 * a node with cells of the level AND cells of the domain that are not the level's around it is a slave of the next coarser level; a node of level n
 * with cells covered by level n+1 around it takes the restricted fine-side parts, with all eight covered it has no equation. */
int vo_ml_nd_solve(int nlev, vo_fab **rh, vo_fab **phi, vo_fab **coeffs, vo_fab **u, const double *dx, const int ellbc[][3][2], const int pmask[3],
                   double rel_eps, double abs_eps, int max_iter, const vdn_params *prm, vo_mgstat *st)
{
  /* the domain of each level as far as the one-box form needs it: a face of the box is a domain face unless its bc says "interior" */
  int pd[6 * ND_MAXLEV];
  for (int n = 0; n < nlev && n < ND_MAXLEV; n++) for (int d = 0; d < 3; d++) {
    pd[6 * n + d] = ellbc[n][d][0] == VDN_BC_INT ? coeffs[n]->lo[d] - (1 << 20) : coeffs[n]->lo[d];
    pd[6 * n + 3 + d] = ellbc[n][d][1] == VDN_BC_INT ? coeffs[n]->hi[d] + (1 << 20) : coeffs[n]->hi[d];
  }
  return vo_ml_nd_solve_g(nlev, NULL, rh, phi, coeffs, u, dx, ellbc, pmask, pd, rel_eps, abs_eps, max_iter, prm, st);
}
int vo_ml_nd_solve_g(int nlev, const vo_level *const *lev, vo_fab **rh, vo_fab **phi, vo_fab **coeffs, vo_fab **u, const double *dx, const int ellbc[][3][2], const int pmask[3],
                     const int *pd, double rel_eps, double abs_eps, int max_iter, const vdn_params *prm, vo_mgstat *st)
{
  if (nlev < 2 || nlev > ND_MAXLEV) { fprintf(stderr, "vo_ml_nd_solve: 2..%d levels\n", ND_MAXLEV); abort(); }
  mlnd M; memset(&M, 0, sizeof M);
  M.nlev = nlev;
  for (int d = 0; d < 3; d++) {
    M.per0[d] = pmask[d];
    if (pmask[d]) for (int n = 1; n < nlev; n++)
      if (coeffs[n]->lo[d] - 2 < pd[6 * n + d] || coeffs[n]->hi[d] + 2 > pd[6 * n + 3 + d]) { fprintf(stderr, "vo_ml_nd_solve: level %d reaches the periodic faces of direction %d: not supported by the oracle\n", n, d); abort(); }
  }
  double *scratch[ND_MAXLEV] = { 0 };
  #define LV(n) (lev ? lev[n] : NULL)
  for (int n = 0; n < nlev; n++) {
    for (int d = 0; d < 3; d++) M.org[n][d] = coeffs[n]->lo[d];
    const int has_fine = n < nlev - 1;
    const int *pdlo = pd + 6 * n, *pdhi = pd + 6 * n + 3;
    /* masked sigma level (zero outside the level's cells and under the next finer level); the full sigma kept aside */
    vo_fab cm; vo_fab_init(&cm, NULL, coeffs[n]->lo, coeffs[n]->hi, 1, NULL, 1); cm.p = (double *)malloc(sizeof(double) * vo_size(&cm));
    memcpy(cm.p, coeffs[n]->p, sizeof(double) * vo_size(&cm));
    const int *clo = coeffs[n]->lo, *chi = coeffs[n]->hi;
    for (int k = clo[2] - 1; k <= chi[2] + 1; k++) for (int j = clo[1] - 1; j <= chi[1] + 1; j++) for (int i = clo[0] - 1; i <= chi[0] + 1; i++) {
      if (vo_lv_multi(LV(n)) && !vo_lv_valid(LV(n), coeffs[n], i, j, k)) VF(&cm, i, j, k, 0) = 0.0;
      if (has_fine && vo_lv_valid(LV(n + 1), coeffs[n + 1], 2 * i, 2 * j, 2 * k)) VF(&cm, i, j, k, 0) = 0.0;
    }
    ndlev *L = &M.L[n];
    nd_level_from_fab(L, &cm, dx + 3 * n); nd_set_mask(L, ellbc[n]);
    free(cm.p);
    long ns = (long)(L->n[0] + 2) * (L->n[1] + 2) * (L->n[2] + 2), nn = (long)(L->n[0] + 3) * (L->n[1] + 3) * (L->n[2] + 3);
    M.sigfull[n] = (double *)malloc(sizeof(double) * ns);
    for (int k = -1; k <= L->n[2]; k++) for (int j = -1; j <= L->n[1]; j++) for (int i = -1; i <= L->n[0]; i++)
      M.sigfull[n][NS(L, i, j, k)] = (vo_lv_multi(LV(n)) && !vo_lv_valid(LV(n), coeffs[n], clo[0] + i, clo[1] + j, clo[2] + k)) ? 0.0 : VF(coeffs[n], clo[0] + i, clo[1] + j, clo[2] + k, 0);
    scratch[n] = (double *)calloc(nn, sizeof(double));
    long nnm = (long)(L->n[0] + 1) * (L->n[1] + 1) * (L->n[2] + 1);
    M.pdir[n] = (unsigned char *)malloc(nnm); memcpy(M.pdir[n], L->dir, nnm);
    M.cf[n] = n >= 1 ? (unsigned char *)calloc(nnm, 1) : NULL;
    M.under[n] = has_fine ? (unsigned char *)calloc(nnm, 1) : NULL;
    for (int k = 0; k <= L->n[2]; k++) for (int j = 0; j <= L->n[1]; j++) for (int i = 0; i <= L->n[0]; i++) {
      int own = 0, other = 0, cov = 0;          /* of the 8 cells around the node: cells of the level / cells of the domain that are not / cells under level n+1 */
      for (int c = -1; c <= 0; c++) for (int b = -1; b <= 0; b++) for (int a = -1; a <= 0; a++) {
        const int q[3] = { clo[0] + i + a, clo[1] + j + b, clo[2] + k + c };
        const int indom = q[0] >= pdlo[0] && q[0] <= pdhi[0] && q[1] >= pdlo[1] && q[1] <= pdhi[1] && q[2] >= pdlo[2] && q[2] <= pdhi[2];
        const int v = indom && vo_lv_valid(LV(n), coeffs[n], q[0], q[1], q[2]);
        own += v; other += indom && !v;
        if (has_fine && v && vo_lv_valid(LV(n + 1), coeffs[n + 1], 2 * q[0], 2 * q[1], 2 * q[2])) cov++;
      }
      const long nm = NM(L, i, j, k);
      if (!own) { M.pdir[n][nm] = 1; L->dir[nm] = 1; }                                         /* touches no cell of the level: inert */
      if (n >= 1 && own && other && !M.pdir[n][nm]) { M.cf[n][nm] = 1; L->dir[nm] = 1; }         /* slave: fixed during the relaxation */
      if (has_fine) M.under[n][nm] = cov == 8 ? 2 : (cov ? 1 : 0);
    }
    /* right-hand side: rh += D u with the masked velocity (vo_nd_divu), b = -rh */
    vo_fab um;
    masked_u(&um, u[n], LV(n), n >= 1, has_fine ? LV(n + 1) : NULL, has_fine ? coeffs[n + 1] : NULL);
    vo_nd_divu(&um, rh[n], dx + 3 * n, ellbc[n]);
    free(um.p);
    for (int k = 0; k <= L->n[2]; k++) for (int j = 0; j <= L->n[1]; j++) for (int i = 0; i <= L->n[0]; i++) {
      double r = VF(rh[n], rh[n]->lo[0] + i, rh[n]->lo[1] + j, rh[n]->lo[2] + k, 0);
      if (M.pdir[n][NM(L, i, j, k)]) r = 0.0;
      L->b[NN(L, i, j, k)] = -r;
      L->phi[NN(L, i, j, k)] = M.pdir[n][NM(L, i, j, k)] ? 0.0 : VF(phi[n], phi[n]->lo[0] + i, phi[n]->lo[1] + j, phi[n]->lo[2] + k, 0);
    }
  }
  #undef LV
  /* norm of the composite right-hand side = composite residual of phi = 0 */
  double bnorm;
  {
    double *save[ND_MAXLEV];
    for (int n = 0; n < nlev; n++) { ndlev *L = &M.L[n]; save[n] = L->phi; L->phi = (double *)calloc((size_t)(L->n[0] + 3) * (L->n[1] + 3) * (L->n[2] + 3), sizeof(double)); }
    bnorm = ml_nd_residual(&M);
    for (int n = 0; n < nlev; n++) { free(M.L[n].phi); M.L[n].phi = save[n]; }
  }
  /* scratch fabs for the coarse correction solve */
  vo_fab er, ee; int nd1[3] = { 1, 1, 1 };
  vo_fab_init(&er, NULL, coeffs[0]->lo, coeffs[0]->hi, 1, nd1, 1); er.p = (double *)calloc(vo_size(&er), sizeof(double));
  vo_fab_init(&ee, NULL, coeffs[0]->lo, coeffs[0]->hi, 1, nd1, 1); ee.p = (double *)calloc(vo_size(&ee), sizeof(double));
  int it = 0, conv = (bnorm == 0.0); double rn = 0.0;
  const int nu_f = prm->hg_nu1 + prm->hg_nu2;
  ndlev *Cc = &M.L[0];
  while (!conv) {
    rn = ml_nd_residual(&M);
    if (rn <= rel_eps * bnorm || rn <= abs_eps) { conv = 1; break; }
    if (it >= max_iter) break;
    /* coarse correction: K_0 e = r_0, one V-cycle (vo_nd_solve takes rh with b = -rh) */
    memset(ee.p, 0, sizeof(double) * vo_size(&ee));
    for (int k = 0; k <= Cc->n[2]; k++) for (int j = 0; j <= Cc->n[1]; j++) for (int i = 0; i <= Cc->n[0]; i++)
      VF(&er, er.lo[0] + i, er.lo[1] + j, er.lo[2] + k, 0) = -Cc->res[NN(Cc, i, j, k)];
    vo_mgstat cs;
    /* (the first correction solve starts from the nested iteration of vo_nd_solve when hg_fmg is set: 13 -> 10 FAC iterations on the
     * refined bubble, base 64^3) */
    vo_nd_solve(&er, &ee, coeffs[0], NULL, dx, ellbc[0], pmask, 0.0, -1.0, -1, prm->hg_nu1, prm->hg_nu2, prm->hg_nub, prm->hg_omega,
                (it == 0 && prm->hg_fmg) ? 2 : 0, vo_om_pre(prm), &cs);
    for (int k = 0; k <= Cc->n[2]; k++) for (int j = 0; j <= Cc->n[1]; j++) for (int i = 0; i <= Cc->n[0]; i++)
      scratch[0][NN(Cc, i, j, k)] = VF(&ee, ee.lo[0] + i, ee.lo[1] + j, ee.lo[2] + k, 0);
    for (int k = 0; k <= Cc->n[2]; k++) for (int j = 0; j <= Cc->n[1]; j++) for (int i = 0; i <= Cc->n[0]; i++)
      Cc->phi[NN(Cc, i, j, k)] = Cc->phi[NN(Cc, i, j, k)] + scratch[0][NN(Cc, i, j, k)];
    /* the finer levels, coarsest first, in correction form (round 4; rounds 2-3 applied every correction to phi on all finer levels at once
     * and formed the composite residual again before each relaxation): e_n = P e_{n-1} (trilinear, not on physical Dirichlet nodes), then
     * the damped-Jacobi sweeps of K_n e_n = r_n -- r_n the residual from the top of the iteration -- with the interface nodes held at
     * P e_{n-1}, then phi_n += e_n.  On the finest level K_n is the operator of the residual, so that two levels make the same iterates
     * as before in exact arithmetic; an intermediate level relaxes with its full sigma against r_n, whose part under the next finer level
     * is the restricted fine residual, where the earlier form recomputed that part with the prolonged correction applied. */
    for (int n = 1; n < nlev; n++) {
      ndlev *F = &M.L[n];
      double *sphi = F->phi, *sb = F->b, *ssig = F->sig;
      long nnf = (long)(F->n[0] + 3) * (F->n[1] + 3) * (F->n[2] + 3);
      double *e = (double *)calloc(nnf, sizeof(double)), *rb = (double *)malloc(sizeof(double) * nnf);
      memcpy(rb, F->res, sizeof(double) * nnf);
      for (int k = 0; k <= F->n[2]; k++) for (int j = 0; j <= F->n[1]; j++) for (int i = 0; i <= F->n[0]; i++)
        if (!M.pdir[n][NM(F, i, j, k)]) e[NN(F, i, j, k)] = ml_nd_interp(&M, n, &M.L[n - 1], scratch[n - 1], i, j, k);
      F->phi = e; F->b = rb; F->sig = M.sigfull[n];
      if (nu_f == 3 && prm->hg_omega_fac1 > 0.0 && prm->hg_omega_fac2 > 0.0 && prm->hg_omega_fac3 > 0.0 && vo_nd_isotropic(dx + 3 * n, 3)) {      /* three-step damping set (round 3): 15 -> 14 and 14 -> 13 FAC iterations on the tagged hierarchies */
        nd_jacobi(F, M.per, 1, prm->hg_omega_fac1); nd_jacobi(F, M.per, 1, prm->hg_omega_fac2); nd_jacobi(F, M.per, 1, prm->hg_omega_fac3);
      } else
      nd_jacobi(F, M.per, nu_f, prm->hg_omega);           /* ping-pongs between F->phi and F->tmp */
      e = F->phi;
      F->b = sb; F->sig = ssig;
      double *other = F->tmp;
      F->phi = sphi;
      /* after the sweeps one of the two scratch buffers holds e and the other is F->tmp: keep them apart from sphi */
      for (int k = 0; k <= F->n[2]; k++) for (int j = 0; j <= F->n[1]; j++) for (int i = 0; i <= F->n[0]; i++)
        F->phi[NN(F, i, j, k)] = F->phi[NN(F, i, j, k)] + e[NN(F, i, j, k)];
      memcpy(scratch[n], e, sizeof(double) * nnf);
      if (other == sphi) { F->tmp = e; } else { free(e); }
      free(rb);
    }
    it++;
  }
  for (int n = 1; n < nlev; n++) ml_nd_interface(&M, n);
  for (int n = 0; n < nlev; n++) {
    ndlev *L = &M.L[n];
    nd_fill_nodes(L, L->phi, n == 0 ? M.per0 : M.per);
    for (int k = -1; k <= L->n[2] + 1; k++) for (int j = -1; j <= L->n[1] + 1; j++) for (int i = -1; i <= L->n[0] + 1; i++)
      VF(phi[n], phi[n]->lo[0] + i, phi[n]->lo[1] + j, phi[n]->lo[2] + k, 0) = L->phi[NN(L, i, j, k)];
  }
  if (st) { st->cycles = it; st->res0 = bnorm; st->res = rn; }
  free(er.p); free(ee.p);
  for (int n = 0; n < nlev; n++) { free(M.cf[n]); free(M.pdir[n]); free(M.under[n]); free(M.sigfull[n]); free(scratch[n]); nd_free(&M.L[n]); }
  return conv ? 0 : 1;
}
