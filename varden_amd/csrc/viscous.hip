// viscous.hip -- explicit diffusive term and the implicit viscous / diffusive solves.
//   get_explicit_diffusive_term   reference src/explicit_diffusive_term.f90:16-88 (FBoxLib cc_applyop, alpha = 0, beta = -1)
//   visc_solve                    reference src/viscsolve.f90:19-306
//   diff_scalar_solve             reference src/viscsolve.f90:308-515
// Same definitions and expression order as oracle/vo_viscous.c: second-order cell-centred differences, zero flux on
// Neumann faces, Dirichlet faces use the ghost cell as the boundary-FACE value with a half-cell gradient, periodic wrap.
// The solves reuse the cell-centred multigrid of mg_cc.hip with the alpha term switched on (56 B/cell/colour pass).
#include "vdn_dev.h"

struct LapArgs { int lo[3], hi[3]; int ebc[3][2]; double hi2[3]; int comp; };
// one descriptor per box, all boxes of a level in one launch (launch_cells: a level of a thousand boxes was a thousand launches, 36 ms of a viscous three-level step)
struct lap_K { FV lap; FV data; LapArgs A;
  __device__ void cell(int i, int j, int k) const {
  const int q[3] = { i, j, k };
  const double p0 = fv_get(data, i, j, k, A.comp);
  double sum = 0.0;
  #pragma unroll
  for (int d = 0; d < 3; d++) {
    const double pm = fv_get(data, i - (d == 0), j - (d == 1), k - (d == 2), A.comp);
    const double pp = fv_get(data, i + (d == 0), j + (d == 1), k + (d == 2), A.comp);
    double fm = p0 - pm, fp = pp - p0;
    if (q[d] == A.lo[d]) { const int e = A.ebc[d][0]; if (e == VDN_BC_NEU) fm = 0.0; else if (e == VDN_BC_DIR) fm = 2.0 * fm; }
    if (q[d] == A.hi[d]) { const int e = A.ebc[d][1]; if (e == VDN_BC_NEU) fp = 0.0; else if (e == VDN_BC_DIR) fp = 2.0 * fp; }
    sum = sum + (fp - fm) * A.hi2[d];
  }
  fv_at(lap, i, j, k, A.comp) = sum;
} };

// lap(comp) = laplacian(data(comp)); bccomp0 = 0-based ell bc component.  data must have its ghost cells filled.
void k_explicit_diffusive_term(vdn_multifab *lap, const vdn_multifab *data, int comp, int bccomp0, const double *dx, const vdn_bc_tower *bct) {
  if (ctx().prm.dm == 2) { k2_explicit_diffusive_term(lap, data, comp, bccomp0, dx, bct); return; }
  REQUIRE(data->ng >= 1, "explicit diffusive term: data needs a filled ghost cell");
  std::vector<std::pair<lap_K, Range3>> v;
  for (int i = 0; i < data->nfabs(); i++) {
    LapArgs A; Range3 r;
    for (int d = 0; d < 3; d++) {
      A.lo[d] = r.lo[d] = data->vbox[i].lo[d]; A.hi[d] = r.hi[d] = data->vbox[i].hi[d]; A.hi2[d] = 1.0 / (dx[d] * dx[d]);
      for (int s = 0; s < 2; s++) A.ebc[d][s] = bct->ell_bc(data->lev, i + 1, d, s, bccomp0);     // BC_INT on interior box faces
    }
    A.comp = comp;
    v.push_back({ lap_K{ lap->fabs[i], data->fabs[i], A }, r });
  }
  launch_cells(v, ctx().stream);
}

struct VrhsArgs { int comp, dtype; double mu, third_vmd_over_dx; };
// mkrhs_3d (viscsolve.f90:264-302): phi = unew(comp) on the grown box, rh = rho*unew + mu*lapu [+ (1/3) visc_mu_dt d(mac_rhs)/dx_comp]
struct visc_rhs_K { FV rh; FV phi; FV unew; FV lapu; FV rho; FV macrhs; VrhsArgs A; int lo0, lo1, lo2, hi0, hi1, hi2; double dxc, third, visc_mu_dt;
  __device__ void cell(int i, int j, int k) const {
  const double u = fv_get(unew, i, j, k, A.comp);
  fv_at(phi, i, j, k) = u;
  if (i < lo0 || i > hi0 || j < lo1 || j > hi1 || k < lo2 || k > hi2) return;
  double r = u * fv_get(rho, i, j, k, 0);
  if (A.dtype == 1) r = r + A.mu * fv_get(lapu, i, j, k, A.comp);
  const int c = A.comp;
  const double mp = fv_get(macrhs, i + (c == 0), j + (c == 1), k + (c == 2)), mm = fv_get(macrhs, i - (c == 0), j - (c == 1), k - (c == 2));
  r = r + third * visc_mu_dt * (mp - mm) / dxc;
  fv_at(rh, i, j, k) = r;
} };
struct diff_rhs_K { FV rh; FV phi; FV snew; FV laps; int comp, dtype; double mu; int lo0, lo1, lo2, hi0, hi1, hi2;
  __device__ void cell(int i, int j, int k) const {
  const double s = fv_get(snew, i, j, k, comp);
  fv_at(phi, i, j, k) = s;
  if (i < lo0 || i > hi0 || j < lo1 || j > hi1 || k < lo2 || k > hi2) return;
  double r = s;
  if (dtype == 1) r = r + mu * fv_get(laps, i, j, k, comp);
  fv_at(rh, i, j, k) = r;
} };
// the right-hand sides of one level, all its boxes in one launch
static void visc_rhs_level(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *unew, const vdn_multifab *lapu, const vdn_multifab *rho, const vdn_multifab *mac_rhs,
                           int comp, double mu, double dxc, double visc_mu_dt) {
  std::vector<std::pair<visc_rhs_K, Range3>> v;
  for (int i = 0; i < unew->nfabs(); i++) {
    const vdn_box &bx = unew->vbox[i];
    Range3 rg; for (int a = 0; a < 3; a++) { rg.lo[a] = bx.lo[a] - 1; rg.hi[a] = bx.hi[a] + 1; }
    VrhsArgs A; A.comp = comp; A.dtype = ctx().prm.diffusion_type; A.mu = mu; A.third_vmd_over_dx = 0.0;
    v.push_back({ visc_rhs_K{ rh->fabs[i], phi->fabs[i], unew->fabs[i], lapu->fabs[i], rho->fabs[i], mac_rhs->fabs[i], A, bx.lo[0], bx.lo[1], bx.lo[2], bx.hi[0], bx.hi[1], bx.hi[2],
                              dxc, 1.0 / 3.0, visc_mu_dt }, rg });
  }
  launch_cells(v, ctx().stream);
}
static void diff_rhs_level(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *snew, const vdn_multifab *laps, int icomp, double mu) {
  std::vector<std::pair<diff_rhs_K, Range3>> v;
  for (int i = 0; i < snew->nfabs(); i++) {
    const vdn_box &bx = snew->vbox[i];
    Range3 rg; for (int a = 0; a < 3; a++) { rg.lo[a] = bx.lo[a] - 1; rg.hi[a] = bx.hi[a] + 1; }
    v.push_back({ diff_rhs_K{ rh->fabs[i], phi->fabs[i], snew->fabs[i], laps->fabs[i], icomp, ctx().prm.diffusion_type, mu, bx.lo[0], bx.lo[1], bx.lo[2], bx.hi[0], bx.hi[1], bx.hi[2] }, rg });
  }
  launch_cells(v, ctx().stream);
}

static void ell_of(const vdn_bc_tower *bct, int lev, int comp0, int ebc[3][2]) {
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc[d][s] = bct->ell_bc(lev, 0, d, s, comp0);
}

void do_visc_solve(vdn_layout *mla, vdn_multifab *unew, const vdn_multifab *lapu, const vdn_multifab *rho, const vdn_multifab *mac_rhs,
                   const double *dx, double mu, const vdn_bc_tower *bct) {
  if (ctx().prm.dm == 2) { do2_visc_solve(mla, unew, lapu, rho, mac_rhs, dx, mu, bct); return; }
  const int n = 0;
  size_t mark = arena_mark();
  vdn_multifab *rh = mf_temp(mla, n, 1, 0, -1, false, 0.0);
  vdn_multifab *phi = mf_temp(mla, n, 1, 1, -1, true, 0.0);
  vdn_multifab *alpha = mf_temp(mla, n, 1, 0, -1, false, 0.0);
  vdn_multifab *beta[3];
  for (int d = 0; d < 3; d++) beta[d] = mf_temp(mla, n, 1, 0, d, true, mu);           // setval(beta, mu), viscsolve.f90:58-60
  mf_copy(alpha, 0, rho, 0, 1, 0);                                                     // alpha = rho, viscsolve.f90:57
  const double visc_mu_dt = (ctx().prm.diffusion_type == 1) ? 2.0 * mu : mu;
  for (int d = 0; d < 3; d++) {
    visc_rhs_level(rh, phi, unew, lapu, rho, mac_rhs, d, mu, dx[d], visc_mu_dt);
    int ebc[3][2]; ell_of(bct, n, d, ebc);                                             // bc_comp = d, viscsolve.f90:99
    int cyc; double r0, rr;
    int rc = cc_solve(rh, phi, beta, dx, ebc, 1.e-12, -1.0, ctx().prm.mg_max_iter, &cyc, &r0, &rr, alpha, nullptr, nullptr, nullptr, 0, false, nullptr, mu);   // viscsolve.f90:88-89 (beta = mu on every face)
    solver_check(rc, "viscous solve", cyc, rr, r0, d);
    mf_copy(unew, d, phi, 0, 1, 0);                                                    // viscsolve.f90:103
  }
  mf_restrict_and_fill(unew, 0, 0, 3, false, bct);                                     // viscsolve.f90:106
  for (int d = 0; d < 3; d++) mf_temp_free(beta[d]);
  mf_temp_free(alpha); mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}

// visc_solve (viscsolve.f90:19-306) on several levels: per velocity component the composite solve of (rho - div mu grad) u = rhs
// (amr.hip: ml_cc_solve with the alpha term; the wall values sit in the ghost cells of unew and go into the right-hand side)
void do_ml_visc_solve(vdn_layout *mla, vdn_multifab **unew, vdn_multifab **lapu, vdn_multifab **rho, vdn_multifab **mac_rhs,
                      const double *dx, double mu, const vdn_bc_tower *bct) {
  const int L = mla->nlev;
  size_t mark = arena_mark();
  vdn_multifab *rh[VDN_MAXLEV], *phi[VDN_MAXLEV], *alpha[VDN_MAXLEV], *beta[3 * VDN_MAXLEV];
  for (int n = 0; n < L; n++) {
    rh[n] = mf_temp(mla, n, 1, 0, -1, false, 0.0); phi[n] = mf_temp(mla, n, 1, 1, -1, true, 0.0); alpha[n] = mf_temp(mla, n, 1, 0, -1, false, 0.0);
    for (int d = 0; d < 3; d++) beta[3 * n + d] = mf_temp(mla, n, 1, 0, d, true, mu);            // setval(beta, mu), viscsolve.f90:58-60
    mf_copy(alpha[n], 0, rho[n], 0, 1, 0);                                                        // alpha = rho, viscsolve.f90:57
  }
  const double visc_mu_dt = (ctx().prm.diffusion_type == 1) ? 2.0 * mu : mu;
  for (int d = 0; d < 3; d++) {
    for (int n = 0; n < L; n++) visc_rhs_level(rh[n], phi[n], unew[n], lapu[n], rho[n], mac_rhs[n], d, mu, dx[3 * n + d], visc_mu_dt);
    int it; double r0, rr;
    int rc = ml_cc_solve(mla, rh, phi, beta, dx, bct, d, 1.e-12, ctx().prm.mg_max_iter, &it, &r0, &rr, alpha, nullptr, nullptr, nullptr, mu);      // bc_comp = d, viscsolve.f90:88-99
    solver_check(rc, "composite viscous solve", it, rr, r0, d);
    for (int n = 0; n < L; n++) mf_copy(unew[n], d, phi[n], 0, 1, 0);                            // viscsolve.f90:103
  }
  ml_restrict_and_fill(L, unew, 0, 0, 3, false, bct);                                            // viscsolve.f90:106
  for (int n = L - 1; n >= 0; n--) { for (int d = 2; d >= 0; d--) mf_temp_free(beta[3 * n + d]); mf_temp_free(alpha[n]); mf_temp_free(phi[n]); mf_temp_free(rh[n]); }
  arena_release(mark);
}

void do_diff_scalar_solve(vdn_layout *mla, vdn_multifab *snew, const vdn_multifab *laps, const double *dx, double mu,
                          const vdn_bc_tower *bct, int icomp, int bccomp0) {
  if (ctx().prm.dm == 2) { do2_diff_scalar_solve(mla, snew, laps, dx, mu, bct, icomp, bccomp0); return; }
  const int n = 0;
  size_t mark = arena_mark();
  vdn_multifab *rh = mf_temp(mla, n, 1, 0, -1, false, 0.0);
  vdn_multifab *phi = mf_temp(mla, n, 1, 1, -1, true, 0.0);
  vdn_multifab *alpha = mf_temp(mla, n, 1, 0, -1, true, 1.0);                          // viscsolve.f90:349
  vdn_multifab *beta[3];
  for (int d = 0; d < 3; d++) beta[d] = mf_temp(mla, n, 1, 0, d, true, mu);
  diff_rhs_level(rh, phi, snew, laps, icomp, mu);
  int ebc[3][2]; ell_of(bct, n, bccomp0, ebc);
  int cyc; double r0, rr;
  int rc = cc_solve(rh, phi, beta, dx, ebc, 1.e-12, -1.0, ctx().prm.mg_max_iter, &cyc, &r0, &rr, alpha, nullptr, nullptr, nullptr, 0, false, nullptr, mu);
  solver_check(rc, "diffusive solve", cyc, rr, r0);
  mf_copy(snew, icomp, phi, 0, 1, 0);                                                  // viscsolve.f90:374
  mf_fill_boundary(snew);                                                              // 378-381 (all comps: a superset of fill_boundary_c)
  mf_physbc(snew, icomp, bccomp0, 1, bct, false);
  for (int d = 0; d < 3; d++) mf_temp_free(beta[d]);
  mf_temp_free(alpha); mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}

// diff_scalar_solve (viscsolve.f90:308-515) on several levels: (1 - div mu grad) s = s [+ mu laps] for component icomp
void do_ml_diff_scalar_solve(vdn_layout *mla, vdn_multifab **snew, vdn_multifab **laps, const double *dx, double mu,
                             const vdn_bc_tower *bct, int icomp, int bccomp0) {
  const int L = mla->nlev;
  size_t mark = arena_mark();
  vdn_multifab *rh[VDN_MAXLEV], *phi[VDN_MAXLEV], *alpha[VDN_MAXLEV], *beta[3 * VDN_MAXLEV];
  for (int n = 0; n < L; n++) {
    rh[n] = mf_temp(mla, n, 1, 0, -1, false, 0.0); phi[n] = mf_temp(mla, n, 1, 1, -1, true, 0.0); alpha[n] = mf_temp(mla, n, 1, 0, -1, true, 1.0);   // viscsolve.f90:349
    for (int d = 0; d < 3; d++) beta[3 * n + d] = mf_temp(mla, n, 1, 0, d, true, mu);
    diff_rhs_level(rh[n], phi[n], snew[n], laps[n], icomp, mu);
  }
  int it; double r0, rr;
  int rc = ml_cc_solve(mla, rh, phi, beta, dx, bct, bccomp0, 1.e-12, ctx().prm.mg_max_iter, &it, &r0, &rr, alpha, nullptr, nullptr, nullptr, mu);
  solver_check(rc, "composite diffusive solve", it, rr, r0);
  for (int n = 0; n < L; n++) mf_copy(snew[n], icomp, phi[n], 0, 1, 0);                          // viscsolve.f90:374
  ml_restrict_and_fill(L, snew, icomp, bccomp0, 1, false, bct);                                  // 378-381
  for (int n = L - 1; n >= 0; n--) { for (int d = 2; d >= 0; d--) mf_temp_free(beta[3 * n + d]); mf_temp_free(alpha[n]); mf_temp_free(phi[n]); mf_temp_free(rh[n]); }
  arena_release(mark);
}
