/* oracle/vo_viscous.c -- explicit diffusive term and the implicit viscous / diffusive solves.
 * reference src/explicit_diffusive_term.f90:16-88, src/viscsolve.f90:19-306 (visc_solve), 308-515 (diff_scalar_solve).
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * The operator applications and solves are FBoxLib's (cc_applyop, ml_cc_solve; not in the tree).  Our definition,
 * consistent with the MAC solver of vo_macproject.c: second-order cell-centred differences; Neumann faces carry zero
 * flux; at Dirichlet faces the ghost cell of the incoming field holds the boundary-FACE value phi_b (what
 * multifab_physbc's EXT_DIR fill leaves there) and the face gradient is (phi_i - phi_b)/(h/2); periodic faces wrap.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "vo.h"

void vo_explicit_diffusive_term(vo_fab *lap, const vo_fab *data, int comp, int bccomp, const double dx[3], const vo_bc *bc)
{
  const int *lo = data->lo, *hi = data->hi;
  double hi2[3] = { 1.0 / (dx[0] * dx[0]), 1.0 / (dx[1] * dx[1]), 1.0 / (dx[2] * dx[2]) };
  #pragma omp parallel for
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    const int q[3] = { i, j, k };
    const double p0 = VF(data, i, j, k, comp);
    double sum = 0.0;
    for (int d = 0; d < 3; d++) {
      int m[3] = { i, j, k }, p[3] = { i, j, k }; m[d] -= 1; p[d] += 1;
      double fm = p0 - VF(data, m[0], m[1], m[2], comp);        /* phi_i - phi_{i-1} */
      double fp = VF(data, p[0], p[1], p[2], comp) - p0;        /* phi_{i+1} - phi_i */
      if (q[d] == lo[d]) { int e = bc->ell[d][0][bccomp]; if (e == VDN_BC_NEU) fm = 0.0; else if (e == VDN_BC_DIR) fm = 2.0 * fm; }
      if (q[d] == hi[d]) { int e = bc->ell[d][1][bccomp]; if (e == VDN_BC_NEU) fp = 0.0; else if (e == VDN_BC_DIR) fp = 2.0 * fp; }
      sum = sum + (fp - fm) * hi2[d];
    }
    VF(lap, i, j, k, comp) = sum;
  }
}

static void fab_alloc(vo_fab *f, const int *lo, const int *hi, int ng, int dir, int nc, double val)
{
  int nd[3] = { 0, 0, 0 }; if (dir >= 0) nd[dir] = 1;
  vo_fab_init(f, NULL, lo, hi, ng, nd, nc);
  long n = vo_size(f);
  f->p = (double *)malloc(sizeof(double) * n);
  for (long i = 0; i < n; i++) f->p[i] = val;
}

/* viscsolve.f90:19-306: per velocity component (alpha = rho, beta = mu) solve, rel 1e-12 */
void vo_visc_solve(vo_fab *unew, const vo_fab *lapu, const vo_fab *rho, const vo_fab *mac_rhs, const double dx[3], double mu,
                   const vo_bc *bc, const int pmask[3], const vdn_params *prm, vo_mgstat *st)
{
  const int *lo = unew->lo, *hi = unew->hi;
  vo_fab rh, phi, alpha, beta[3], *bp[3];
  fab_alloc(&rh, lo, hi, 0, -1, 1, 0.0); fab_alloc(&phi, lo, hi, 1, -1, 1, 0.0); fab_alloc(&alpha, lo, hi, 0, -1, 1, 0.0);
  for (int d = 0; d < 3; d++) { fab_alloc(&beta[d], lo, hi, 0, d, 1, mu); bp[d] = &beta[d]; }
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
    VF(&alpha, i, j, k, 0) = VF(rho, i, j, k, 0);
  const double third = 1.0 / 3.0;
  for (int d = 0; d < 3; d++) {
    /* mkrhs_3d, viscsolve.f90:264-302 */
    double visc_mu_dt = (prm->diffusion_type == 1) ? 2.0 * mu : mu;
    for (int k = lo[2] - 1; k <= hi[2] + 1; k++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
      VF(&phi, i, j, k, 0) = VF(unew, i, j, k, d);
    for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
      double r = VF(unew, i, j, k, d) * VF(rho, i, j, k, 0);
      if (prm->diffusion_type == 1) r = r + mu * VF(lapu, i, j, k, d);
      int p[3] = { i, j, k }, m[3] = { i, j, k }; p[d] += 1; m[d] -= 1;
      r = r + third * visc_mu_dt * (VF(mac_rhs, p[0], p[1], p[2], 0) - VF(mac_rhs, m[0], m[1], m[2], 0)) / dx[d];
      VF(&rh, i, j, k, 0) = r;
    }
    int ellbc[3][2];
    for (int a = 0; a < 3; a++) for (int s = 0; s < 2; s++) ellbc[a][s] = bc->ell[a][s][d];
    vo_cc_solve_ab(&rh, &phi, &alpha, bp, dx, ellbc, 1.e-12, -1.0, prm->mg_max_iter, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, 0, st);
    for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
      VF(unew, i, j, k, d) = VF(&phi, i, j, k, 0);
  }
  /* ml_restrict_and_fill(unew) (viscsolve.f90:106) */
  vo_fill_boundary(unew, pmask);
  vo_physbc(unew, 0, 0, 3, bc, prm);
  free(rh.p); free(phi.p); free(alpha.p); for (int d = 0; d < 3; d++) free(beta[d].p);
}

/* viscsolve.f90:308-515: alpha = 1, beta = mu */
void vo_diff_scalar_solve(vo_fab *snew, const vo_fab *laps, const double dx[3], double mu, const vo_bc *bc, const int pmask[3],
                          const vdn_params *prm, int icomp, int bccomp, vo_mgstat *st)
{
  const int *lo = snew->lo, *hi = snew->hi;
  vo_fab rh, phi, alpha, beta[3], *bp[3];
  fab_alloc(&rh, lo, hi, 0, -1, 1, 0.0); fab_alloc(&phi, lo, hi, 1, -1, 1, 0.0); fab_alloc(&alpha, lo, hi, 0, -1, 1, 1.0);
  for (int d = 0; d < 3; d++) { fab_alloc(&beta[d], lo, hi, 0, d, 1, mu); bp[d] = &beta[d]; }
  for (int k = lo[2] - 1; k <= hi[2] + 1; k++) for (int j = lo[1] - 1; j <= hi[1] + 1; j++) for (int i = lo[0] - 1; i <= hi[0] + 1; i++)
    VF(&phi, i, j, k, 0) = VF(snew, i, j, k, icomp);
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++) {
    double r = VF(snew, i, j, k, icomp);
    if (prm->diffusion_type == 1) r = r + mu * VF(laps, i, j, k, icomp);
    VF(&rh, i, j, k, 0) = r;
  }
  int ellbc[3][2];
  for (int a = 0; a < 3; a++) for (int s = 0; s < 2; s++) ellbc[a][s] = bc->ell[a][s][bccomp];
  vo_cc_solve_ab(&rh, &phi, &alpha, bp, dx, ellbc, 1.e-12, -1.0, prm->mg_max_iter, prm->mg_nu1, prm->mg_nu2, prm->mg_nub, 0, st);
  for (int k = lo[2]; k <= hi[2]; k++) for (int j = lo[1]; j <= hi[1]; j++) for (int i = lo[0]; i <= hi[0]; i++)
    VF(snew, i, j, k, icomp) = VF(&phi, i, j, k, 0);
  vo_fill_boundary(snew, pmask);                        /* viscsolve.f90:378-381 */
  vo_physbc(snew, icomp, bccomp, 1, bc, prm);
  free(rh.p); free(phi.p); free(alpha.p); for (int d = 0; d < 3; d++) free(beta[d].p);
}
