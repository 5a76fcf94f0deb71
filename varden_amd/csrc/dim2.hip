// dim2.hip -- the dm = 2 path of advance_timestep (BASELINE.json configs[0], the reference's CPU-runnable case).
//
// Reference routines restated (single level; one box -- the 2-D configuration is plumbing, not a performance target):
//   mkvelforce_2d / mkscalforce_2d   src/mkforce.f90:82-142, 290-331
//   update_2d                        src/update.f90:113-184
//   estdt_2d                         src/estdt.f90:89-129
//   divumac_2d, mk_mac_coeffs_2d, mkumac_2d      src/macproject.f90:226-248, 338-359, 538-576
//   create_uvec_2d, mkgphi_2d, hg_update_2d      src/hgproject.f90:374-432, 517-541, 581-636
//   explicit diffusive term / visc_solve / diff_scalar_solve with the 2-D right-hand sides (viscsolve.f90:226-262)
// velpred_2d / mkflux_2d live in godunov.hip.  The two elliptic solvers are compact 2-D versions of the algorithms of
// mg_cc.hip / mg_nd.hip (5-point red-black Gauss-Seidel V-cycles; 9-point Q1 damped-Jacobi V-cycles), in the expression
// order of the oracle's dm = 2 mode (oracle/vo_macproject.c, oracle/vo_hgproject.c).
//
// Device layout of a 2-D fab: the 3-D layout with ONE valid z-plane (k = 0); its z-ghost planes exist but are never
// read or written here.  vdn_multifab_copy_to/from_host present the BoxLib 2-D layout p(lo1-ng:hi1+ng, lo2-ng:hi2+ng, nc).
#include "vdn_dev.h"
#include <vector>
#include <algorithm>

#define G2(f, i, j, c) fv_get(f, i, j, 0, c)
#define P2(f, i, j, c) fv_at(f, i, j, 0, c)
static const dim3 B2(64, 4, 1);
static Range3 rng2(int lo0, int hi0, int lo1, int hi1) { Range3 r; r.lo[0] = lo0; r.hi[0] = hi0; r.lo[1] = lo1; r.hi[1] = hi1; r.lo[2] = r.hi[2] = 0; return r; }
static void require_2d(const vdn_multifab *mf, const char *who) {
  REQUIRE(mf->la->nlev == 1 && mf->nfabs() == 1 && mf->la->boxes[0].size() == 1, "%s: the dm = 2 path supports one level with one box", who);
}

// ---- forcing, update, estdt -------------------------------------------------------------------------------------------
struct F2Args { int lo[2], hi[2]; double coef, fac; int boussinesq, nscal; };
__global__ void kk2_mkvelforce(FV vf, FV ext, FV gp, FV s, FV lapu, int has_lapu, F2Args A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const int out = (i < A.lo[0]) + (i > A.hi[0]) + (j < A.lo[1]) + (j > A.hi[1]);
  if (out > 1) return;                               // the four edge halos only (mkforce.f90:118-139)
  const int ic = min(max(i, A.lo[0]), A.hi[0]), jc = min(max(j, A.lo[1]), A.hi[1]);
  const double rho = G2(s, i, j, 0);
  #pragma unroll
  for (int m = 0; m < 2; m++) {
    const double l = has_lapu ? G2(lapu, ic, jc, m) : 0.0;
    const double lapu_local = A.coef * A.fac * l;
    double e = G2(ext, i, j, m);
    if (out == 0 && A.boussinesq == 1) e = G2(s, i, j, 1) * e;
    P2(vf, i, j, m) = e + (lapu_local - G2(gp, i, j, m)) / rho;
  }
}
__global__ void kk2_mkscalforce(FV sf, FV ext, FV laps, int has_laps, F2Args A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const int out = (i < A.lo[0]) + (i > A.hi[0]) + (j < A.lo[1]) + (j > A.hi[1]);
  if (out > 1) return;
  const int ic = min(max(i, A.lo[0]), A.hi[0]), jc = min(max(j, A.lo[1]), A.hi[1]);
  for (int m = 1; m < A.nscal; m++) {
    const double l = has_laps ? G2(laps, ic, jc, m) : 0.0;
    P2(sf, i, j, m) = G2(ext, i, j, m) + A.coef * A.fac * l;
  }
}
void k2_mkvelforce(vdn_multifab *vf, const vdn_multifab *ext, const vdn_multifab *s, const vdn_multifab *gp, const vdn_multifab *lapu, double visc_fac) {
  mf_setval(vf, 0.0, 0, vf->nc, true);
  for (int b = 0; b < vf->nfabs(); b++) {
    F2Args A; const vdn_box &bx = vf->vbox[b];
    for (int d = 0; d < 2; d++) { A.lo[d] = bx.lo[d]; A.hi[d] = bx.hi[d]; }
    A.coef = ctx().prm.visc_coef; A.fac = visc_fac; A.boussinesq = ctx().prm.boussinesq; A.nscal = ctx().prm.nscal;
    Range3 r = rng2(A.lo[0] - 1, A.hi[0] + 1, A.lo[1] - 1, A.hi[1] + 1);
    hipLaunchKernelGGL(kk2_mkvelforce, grid_for(r), B2, 0, ctx().stream, vf->fabs[b], ext->fabs[b], gp->fabs[b], s->fabs[b], lapu ? lapu->fabs[b] : vf->fabs[b], lapu ? 1 : 0, A, r);
  }
}
void k2_mkscalforce(vdn_multifab *sf, const vdn_multifab *ext, const vdn_multifab *laps, double diff_fac) {
  mf_setval(sf, 0.0, 0, sf->nc, true);
  for (int b = 0; b < sf->nfabs(); b++) {
    F2Args A; const vdn_box &bx = sf->vbox[b];
    for (int d = 0; d < 2; d++) { A.lo[d] = bx.lo[d]; A.hi[d] = bx.hi[d]; }
    A.coef = ctx().prm.diff_coef; A.fac = diff_fac; A.boussinesq = 0; A.nscal = ctx().prm.nscal;
    Range3 r = rng2(A.lo[0] - 1, A.hi[0] + 1, A.lo[1] - 1, A.hi[1] + 1);
    hipLaunchKernelGGL(kk2_mkscalforce, grid_for(r), B2, 0, ctx().stream, sf->fabs[b], ext->fabs[b], laps ? laps->fabs[b] : sf->fabs[b], laps ? 1 : 0, A, r);
  }
}

struct U2Args { double dx[2], dt; int ncomp; int cons[VDN_MAXCOMP]; };
__global__ void kk2_update(FV sold, FV snew, FV um, FV vm, FV sx, FV sy, FV fx, FV fy, FV force, U2Args A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double ubar = 0.5 * (G2(um, i, j, 0) + G2(um, i + 1, j, 0));
  const double vbar = 0.5 * (G2(vm, i, j, 0) + G2(vm, i, j + 1, 0));
  for (int c = 0; c < A.ncomp; c++) {
    const double so = G2(sold, i, j, c), f = G2(force, i, j, c);
    double v;
    if (A.cons[c]) {
      const double divsu = (G2(fx, i + 1, j, c) - G2(fx, i, j, c)) / A.dx[0] + (G2(fy, i, j + 1, c) - G2(fy, i, j, c)) / A.dx[1];
      v = so - A.dt * divsu + A.dt * f;
    } else {
      const double ug = ubar * (G2(sx, i + 1, j, c) - G2(sx, i, j, c)) / A.dx[0] + vbar * (G2(sy, i, j + 1, c) - G2(sy, i, j, c)) / A.dx[1];
      v = so - A.dt * ug + A.dt * f;
    }
    P2(snew, i, j, c) = v;
  }
}
void k2_update(const vdn_multifab *sold, vdn_multifab **umac, vdn_multifab **sedge, vdn_multifab **flux, const vdn_multifab *force, vdn_multifab *snew,
               const double *dx, double dt, bool is_vel, const int *is_cons) {
  for (int b = 0; b < sold->nfabs(); b++) {
    U2Args A; A.dx[0] = dx[0]; A.dx[1] = dx[1]; A.dt = dt; A.ncomp = sold->nc;
    for (int c = 0; c < sold->nc; c++) A.cons[c] = (!is_vel && is_cons[c]) ? 1 : 0;
    const vdn_box &bx = sold->vbox[b];
    Range3 r = rng2(bx.lo[0], bx.hi[0], bx.lo[1], bx.hi[1]);
    hipLaunchKernelGGL(kk2_update, grid_for(r), B2, 0, ctx().stream, sold->fabs[b], snew->fabs[b], umac[0]->fabs[b], umac[1]->fabs[b],
                       sedge[0]->fabs[b], sedge[1]->fabs[b], flux[0]->fabs[b], flux[1]->fabs[b], force->fabs[b], A, r);
  }
}
__global__ void kk2_estdt(FV u, FV s, FV gp, FV ext, Range3 r, double *out6) {
  REDUCE_IJ(r)
  double m[4] = { 0, 0, 0, 0 };
  if (in_ij) {
    const double rho = G2(s, i, j, 0);
    #pragma unroll
    for (int c = 0; c < 2; c++) { m[c] = fabs(G2(u, i, j, c)); m[2 + c] = fabs(G2(gp, i, j, c) / rho - G2(ext, i, j, c)); }
  }
  block_atomic_max(out6 + 0, m[0]); block_atomic_max(out6 + 1, m[1]); block_atomic_max(out6 + 3, m[2]); block_atomic_max(out6 + 4, m[3]);
}
void k2_estdt_max(const vdn_multifab *u, const vdn_multifab *s, const vdn_multifab *gp, const vdn_multifab *ext, double out6[6]) {
  VdnCtx &c = ctx();
  HIPCHK(hipMemsetAsync(c.d_scal, 0, 6 * sizeof(double), c.stream));
  for (int b = 0; b < u->nfabs(); b++) {
    const vdn_box &bx = u->vbox[b];
    Range3 r = rng2(bx.lo[0], bx.hi[0], bx.lo[1], bx.hi[1]);
    hipLaunchKernelGGL(kk2_estdt, reduce_grid(r), B2, 0, c.stream, u->fabs[b], s->fabs[b], gp->fabs[b], ext->fabs[b], r, c.d_scal);
  }
  HIPCHK(hipMemcpyAsync(c.h_scal, c.d_scal, 6 * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  HIPCHK(hipStreamSynchronize(c.stream));
  for (int k = 0; k < 6; k++) out6[k] = c.h_scal[k];
}

// =====================================================================================================================
// cell-centred multigrid, 5-point (alpha - div b grad) phi = rh
// =====================================================================================================================
struct C2 { int n0, n1, P; double hi2[2]; double *phi, *rh, *res, *bx, *by, *alpha; };
DEVI long c2i(const C2 &L, int i, int j) { return (long)(i + 1) + (long)L.P * (j + 1); }
static long c2_size(int n0, int n1) { return (long)(n0 + 2) * (n1 + 2); }
DEVI void c2_apply(const C2 &L, int i, int j, double &Ap, double &diag) {
  const long c = c2i(L, i, j);
  const double p0 = L.phi[c];
  const double bxm = L.bx[c], bxp = L.bx[c + 1], bym = L.by[c], byp = L.by[c + L.P];
  const double ax = (bxp * (p0 - L.phi[c + 1]) + bxm * (p0 - L.phi[c - 1])) * L.hi2[0];
  const double ay = (byp * (p0 - L.phi[c + L.P]) + bym * (p0 - L.phi[c - L.P])) * L.hi2[1];
  Ap = ax + ay;
  diag = (bxp + bxm) * L.hi2[0] + (byp + bym) * L.hi2[1];
  if (L.alpha) { const double a0 = L.alpha[c]; Ap = Ap + a0 * p0; diag = diag + a0; }
}
__global__ void kk_c2_gsrb(C2 L, int color) {
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int i = 2 * (int)(blockIdx.x * blockDim.x + threadIdx.x) + ((j + color) & 1);
  if (i >= L.n0 || j >= L.n1) return;
  double Ap, diag; c2_apply(L, i, j, Ap, diag);
  const long c = c2i(L, i, j);
  if (diag != 0.0) L.phi[c] = L.phi[c] + (L.rh[c] - Ap) / diag;
}
__global__ void kk_c2_residual(C2 L, double *nrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  double r = 0.0;
  if (i < L.n0 && j < L.n1) { double Ap, diag; c2_apply(L, i, j, Ap, diag); r = L.rh[c2i(L, i, j)] - Ap; L.res[c2i(L, i, j)] = r; }
  if (nrm) block_atomic_max(nrm, fabs(r));
}
__global__ void kk_c2_restrict(C2 F, C2 C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= C.n0 || j >= C.n1) return;
  const long f = c2i(F, 2 * i, 2 * j);
  C.rh[c2i(C, i, j)] = (F.res[f] + F.res[f + 1] + F.res[f + F.P] + F.res[f + F.P + 1]) * 0.25;
}
__global__ void kk_c2_prolong(C2 F, C2 C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= F.n0 || j >= F.n1) return;
  F.phi[c2i(F, i, j)] = F.phi[c2i(F, i, j)] + C.phi[c2i(C, i / 2, j / 2)];
}
__global__ void kk_c2_coarsen(C2 F, C2 C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > C.n0 || j > C.n1) return;
  if (j < C.n1) C.bx[c2i(C, i, j)] = (F.bx[c2i(F, 2 * i, 2 * j)] + F.bx[c2i(F, 2 * i, 2 * j + 1)]) * 0.5;
  if (i < C.n0) C.by[c2i(C, i, j)] = (F.by[c2i(F, 2 * i, 2 * j)] + F.by[c2i(F, 2 * i + 1, 2 * j)]) * 0.5;
  if (C.alpha && i < C.n0 && j < C.n1) {
    const long f = c2i(F, 2 * i, 2 * j);
    C.alpha[c2i(C, i, j)] = (F.alpha[f] + F.alpha[f + 1] + F.alpha[f + F.P] + F.alpha[f + F.P + 1]) * 0.25;
  }
}
__global__ void kk_c2_periodic(C2 L, int per0, int per1) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  if (i > L.n0 || j > L.n1) return;
  int si = i, sj = j; bool g = false, ok = true;
  if (i < 0) { g = true; if (per0) si = i + L.n0; else ok = false; } else if (i >= L.n0) { g = true; if (per0) si = i - L.n0; else ok = false; }
  if (j < 0) { g = true; if (per1) sj = j + L.n1; else ok = false; } else if (j >= L.n1) { g = true; if (per1) sj = j - L.n1; else ok = false; }
  if (g && ok) L.phi[c2i(L, i, j)] = L.phi[c2i(L, si, sj)];
}
struct C2Bc { int e[2][2]; int lo[2]; };
// level 0: face coefficients with the boundary folding (Neumann b := 0, Dirichlet b := 2b), alpha
__global__ void kk_c2_load_b(C2 L, FV bxf, FV byf, FV af, int has_alpha, C2Bc B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > L.n0 || j > L.n1) return;
  if (j < L.n1) {
    double v = G2(bxf, B.lo[0] + i, B.lo[1] + j, 0);
    const int side = (i == 0) ? 0 : ((i == L.n0) ? 1 : -1);
    if (side >= 0) { if (B.e[0][side] == VDN_BC_NEU) v = 0.0; else if (B.e[0][side] == VDN_BC_DIR) v = 2.0 * v; }
    L.bx[c2i(L, i, j)] = v;
  }
  if (i < L.n0) {
    double v = G2(byf, B.lo[0] + i, B.lo[1] + j, 0);
    const int side = (j == 0) ? 0 : ((j == L.n1) ? 1 : -1);
    if (side >= 0) { if (B.e[1][side] == VDN_BC_NEU) v = 0.0; else if (B.e[1][side] == VDN_BC_DIR) v = 2.0 * v; }
    L.by[c2i(L, i, j)] = v;
  }
  if (has_alpha && i < L.n0 && j < L.n1) L.alpha[c2i(L, i, j)] = G2(af, B.lo[0] + i, B.lo[1] + j, 0);
}
// right-hand side with the Dirichlet data (ghost cells of the incoming phi = boundary-face values) moved into it; phi
__global__ void kk_c2_load_rh(C2 L, FV rh, FV phi, C2Bc B, double *nrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  double r0 = 0.0;
  if (i < L.n0 && j < L.n1) {
    const int gi = B.lo[0] + i, gj = B.lo[1] + j;
    double r = G2(rh, gi, gj, 0);
    r0 = r;
    const long c = c2i(L, i, j);
    if (i == 0 && B.e[0][0] == VDN_BC_DIR)        r = r + L.bx[c] * G2(phi, gi - 1, gj, 0) * L.hi2[0];
    if (i == L.n0 - 1 && B.e[0][1] == VDN_BC_DIR) r = r + L.bx[c + 1] * G2(phi, gi + 1, gj, 0) * L.hi2[0];
    if (j == 0 && B.e[1][0] == VDN_BC_DIR)        r = r + L.by[c] * G2(phi, gi, gj - 1, 0) * L.hi2[1];
    if (j == L.n1 - 1 && B.e[1][1] == VDN_BC_DIR) r = r + L.by[c + L.P] * G2(phi, gi, gj + 1, 0) * L.hi2[1];
    L.rh[c] = r;
    L.phi[c] = G2(phi, gi, gj, 0);
  }
  block_atomic_max(nrm, fabs(r0));
}
// phi back into the fab incl. the ghost layer the closure implies (Neumann: phi_i, Dirichlet: -phi_i, periodic: image)
__global__ void kk_c2_store(C2 L, FV phi, C2Bc B) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  if (i > L.n0 || j > L.n1) return;
  const bool gi = (i < 0 || i >= L.n0), gj = (j < 0 || j >= L.n1);
  if (gi && gj) return;                          // corners are not needed
  double v;
  if (!gi && !gj) v = L.phi[c2i(L, i, j)];
  else {
    const int d = gi ? 0 : 1, s = gi ? (i < 0 ? 0 : 1) : (j < 0 ? 0 : 1);
    const int qi = gi ? (s ? L.n0 - 1 : 0) : i, qj = gj ? (s ? L.n1 - 1 : 0) : j;
    if (B.e[d][s] == VDN_BC_NEU) v = L.phi[c2i(L, qi, qj)];
    else if (B.e[d][s] == VDN_BC_DIR) v = -L.phi[c2i(L, qi, qj)];
    else v = L.phi[c2i(L, i, j)];
  }
  P2(phi, B.lo[0] + i, B.lo[1] + j, 0) = v;
}
static dim3 g2(int nx, int ny) { return dim3((nx + 63) / 64, (ny + 3) / 4, 1); }
static double read_scal(double *d) {
  return read_scalar1(d);
}
struct CC2MG { std::vector<C2> lev; int per[2]; double *d_nrm; };
static void c2_gsrb(const CC2MG &M, const C2 &L, int ns) {
  hipStream_t st = ctx().stream;
  for (int s = 0; s < ns; s++) for (int col = 0; col < 2; col++) {
    if (M.per[0] || M.per[1]) hipLaunchKernelGGL(kk_c2_periodic, g2(L.n0 + 2, L.n1 + 2), B2, 0, st, L, M.per[0], M.per[1]);
    hipLaunchKernelGGL(kk_c2_gsrb, g2((L.n0 + 1) / 2, L.n1), B2, 0, st, L, col);
  }
}
static void c2_residual(const CC2MG &M, const C2 &L, bool norm) {
  hipStream_t st = ctx().stream;
  if (M.per[0] || M.per[1]) hipLaunchKernelGGL(kk_c2_periodic, g2(L.n0 + 2, L.n1 + 2), B2, 0, st, L, M.per[0], M.per[1]);
  if (norm) HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), st));
  hipLaunchKernelGGL(kk_c2_residual, g2(L.n0, L.n1), B2, 0, st, L, norm ? M.d_nrm : (double *)nullptr);
}
static int c2_bottom(const C2 &L) { const int N = std::max(L.n0, L.n1); return std::max(ctx().prm.mg_nub, N * N); }
static void c2_vcycle(const CC2MG &M, int l) {
  const vdn_params &P = ctx().prm;
  const C2 &L = M.lev[l];
  HIPCHK(hipMemsetAsync(L.phi, 0, sizeof(double) * c2_size(L.n0, L.n1), ctx().stream));
  if (l == (int)M.lev.size() - 1) { c2_gsrb(M, L, c2_bottom(L)); return; }
  const C2 &C = M.lev[l + 1];
  c2_gsrb(M, L, P.mg_nu1);
  c2_residual(M, L, false);
  hipLaunchKernelGGL(kk_c2_restrict, g2(C.n0, C.n1), B2, 0, ctx().stream, L, C);
  c2_vcycle(M, l + 1);
  hipLaunchKernelGGL(kk_c2_prolong, g2(L.n0, L.n1), B2, 0, ctx().stream, L, C);
  c2_gsrb(M, L, P.mg_nu2);
}
int cc2_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int bc[3][2], double rel_eps, double abs_eps, int max_iter,
              int *cycles, double *res0, double *res, const vdn_multifab *alpha) {
  require_2d(rh, "cc_solve");
  REQUIRE(phi->ng >= 1, "cc multigrid: phi needs one ghost cell");
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  const vdn_box &bx = rh->vbox[0];
  CC2MG M; M.per[0] = bc[0][0] == VDN_BC_PER; M.per[1] = bc[1][0] == VDN_BC_PER;
  M.d_nrm = (double *)arena_alloc(256);
  int n0 = bx.hi[0] - bx.lo[0] + 1, n1 = bx.hi[1] - bx.lo[1] + 1; double h0 = dx[0], h1 = dx[1];
  for (;;) {
    C2 L; L.n0 = n0; L.n1 = n1; L.P = n0 + 2; L.hi2[0] = 1.0 / (h0 * h0); L.hi2[1] = 1.0 / (h1 * h1);
    const long sz = c2_size(n0, n1);
    double *base = (double *)arena_alloc(sizeof(double) * sz * (alpha ? 6 : 5));
    HIPCHK(hipMemsetAsync(base, 0, sizeof(double) * sz * (alpha ? 6 : 5), st));
    L.phi = base; L.rh = base + sz; L.res = base + 2 * sz; L.bx = base + 3 * sz; L.by = base + 4 * sz; L.alpha = alpha ? base + 5 * sz : nullptr;
    M.lev.push_back(L);
    if ((n0 & 1) || (n1 & 1) || n0 <= 2 || n1 <= 2 || M.lev.size() >= 31) break;
    n0 /= 2; n1 /= 2; h0 *= 2.0; h1 *= 2.0;
  }
  C2Bc B; for (int d = 0; d < 2; d++) { B.lo[d] = bx.lo[d]; for (int s = 0; s < 2; s++) B.e[d][s] = bc[d][s]; }
  const C2 &L0 = M.lev[0];
  hipLaunchKernelGGL(kk_c2_load_b, g2(L0.n0 + 1, L0.n1 + 1), B2, 0, st, L0, beta[0]->fabs[0], beta[1]->fabs[0], alpha ? alpha->fabs[0] : rh->fabs[0], alpha ? 1 : 0, B);
  for (size_t l = 1; l < M.lev.size(); l++) hipLaunchKernelGGL(kk_c2_coarsen, g2(M.lev[l].n0 + 1, M.lev[l].n1 + 1), B2, 0, st, M.lev[l - 1], M.lev[l]);
  HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), st));
  hipLaunchKernelGGL(kk_c2_load_rh, g2(L0.n0, L0.n1), B2, 0, st, L0, rh->fabs[0], phi->fabs[0], B, M.d_nrm);
  const double bnorm = read_scal(M.d_nrm);
  const vdn_params &P = ctx().prm;
  int cyc = 0; bool conv = false; double rn = 0.0;
  if (bnorm == 0.0) conv = true;
  while (!conv && cyc <= max_iter) {
    c2_gsrb(M, L0, M.lev.size() == 1 ? c2_bottom(L0) : P.mg_nu1);
    c2_residual(M, L0, true);
    rn = read_scal(M.d_nrm);
    if ((rn <= rel_eps * bnorm && bnorm < HUGE_VAL) || rn <= abs_eps) { conv = true; break; }
    if (cyc == max_iter || !(rn < HUGE_VAL) || !(bnorm < HUGE_VAL)) break;
    if (M.lev.size() > 1) {
      hipLaunchKernelGGL(kk_c2_restrict, g2(M.lev[1].n0, M.lev[1].n1), B2, 0, st, L0, M.lev[1]);
      c2_vcycle(M, 1);
      hipLaunchKernelGGL(kk_c2_prolong, g2(L0.n0, L0.n1), B2, 0, st, L0, M.lev[1]);
      c2_gsrb(M, L0, P.mg_nu2);
    }
    cyc++;
  }
  if (M.per[0] || M.per[1]) hipLaunchKernelGGL(kk_c2_periodic, g2(L0.n0 + 2, L0.n1 + 2), B2, 0, st, L0, M.per[0], M.per[1]);
  hipLaunchKernelGGL(kk_c2_store, g2(L0.n0 + 2, L0.n1 + 2), B2, 0, st, L0, phi->fabs[0], B);
  if (cycles) *cycles = cyc; if (res0) *res0 = bnorm; if (res) *res = rn;
  HIPCHK(hipStreamSynchronize(st));
  arena_release(mark);
  return conv ? 0 : 1;
}

// ---- MAC projection ---------------------------------------------------------------------------------------------------
__global__ void kk2_divumac(FV um, FV vm, FV macrhs, FV rh, double dxi0, double dxi1, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double d = (G2(um, i + 1, j, 0) - G2(um, i, j, 0)) * dxi0 + (G2(vm, i, j + 1, 0) - G2(vm, i, j, 0)) * dxi1;
  P2(rh, i, j, 0) = d * -1.0 + G2(macrhs, i, j, 0);
}
__global__ void kk2_mac_coeffs(FV rho, FV bx, FV by, Range3 r, int hi0, int hi1) {
  THREAD_IJK(r)
  if (!in_range) return;
  if (j <= hi1) P2(bx, i, j, 0) = 2.0 / (G2(rho, i, j, 0) + G2(rho, i - 1, j, 0));
  if (i <= hi0) P2(by, i, j, 0) = 2.0 / (G2(rho, i, j, 0) + G2(rho, i, j - 1, 0));
}
struct Um2Args { int lo[2], hi[2]; double dx[2]; int ebc[2][2]; };
__global__ void kk2_mkumac(FV um, FV vm, FV phi, FV bx, FV by, Um2Args A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  if (j <= A.hi[1]) {
    const int side = (i == A.lo[0]) ? 0 : ((i == A.hi[0] + 1) ? 1 : -1);
    if (!(side >= 0 && A.ebc[0][side] == VDN_BC_NEU)) {
      const double g = (G2(phi, i, j, 0) - G2(phi, i - 1, j, 0)) / A.dx[0];
      P2(um, i, j, 0) = G2(um, i, j, 0) - G2(bx, i, j, 0) * g;
    }
  }
  if (i <= A.hi[0]) {
    const int side = (j == A.lo[1]) ? 0 : ((j == A.hi[1] + 1) ? 1 : -1);
    if (!(side >= 0 && A.ebc[1][side] == VDN_BC_NEU)) {
      const double g = (G2(phi, i, j, 0) - G2(phi, i, j - 1, 0)) / A.dx[1];
      P2(vm, i, j, 0) = G2(vm, i, j, 0) - G2(by, i, j, 0) * g;
    }
  }
}
void do2_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs, const double *dx, const vdn_bc_tower *bct, int bc_comp0) {
  require_2d(rho[0], "macproject");
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  vdn_multifab *rh = mf_temp(mla, 0, 1, 0, -1, false, 0.0), *phi = mf_temp(mla, 0, 1, 1, -1, true, 0.0);
  vdn_multifab *beta[2] = { mf_temp(mla, 0, 1, 0, 0, false, 0.0), mf_temp(mla, 0, 1, 0, 1, false, 0.0) };
  const vdn_box &bx = rh->vbox[0];
  Range3 rv = rng2(bx.lo[0], bx.hi[0], bx.lo[1], bx.hi[1]), rf = rng2(bx.lo[0], bx.hi[0] + 1, bx.lo[1], bx.hi[1] + 1);
  hipLaunchKernelGGL(kk2_divumac, grid_for(rv), B2, 0, st, umac[0]->fabs[0], umac[1]->fabs[0], mac_rhs[0]->fabs[0], rh->fabs[0], 1.0 / dx[0], 1.0 / dx[1], rv);
  hipLaunchKernelGGL(kk2_mac_coeffs, grid_for(rf), B2, 0, st, rho[0]->fabs[0], beta[0]->fabs[0], beta[1]->fabs[0], rf, bx.hi[0], bx.hi[1]);
  int ebc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc[d][s] = d < 2 ? bct->ell_bc(0, 0, d, s, bc_comp0) : VDN_BC_INT;
  int cyc; double r0, rr;
  int rc = cc2_solve(rh, phi, beta, dx, ebc, ctx().prm.mac_rel_eps, -1.0, ctx().prm.mg_max_iter, &cyc, &r0, &rr, nullptr);
  ctx().solver_cycles[0] = cyc; ctx().solver_res0[0] = r0; ctx().solver_res[0] = rr;
  solver_check(rc, "MAC multigrid (2-D)", cyc, rr, r0);
  Um2Args A;
  for (int d = 0; d < 2; d++) { A.lo[d] = bx.lo[d]; A.hi[d] = bx.hi[d]; A.dx[d] = dx[d]; for (int s = 0; s < 2; s++) A.ebc[d][s] = ebc[d][s]; }
  hipLaunchKernelGGL(kk2_mkumac, grid_for(rf), B2, 0, st, umac[0]->fabs[0], umac[1]->fabs[0], phi->fabs[0], beta[0]->fabs[0], beta[1]->fabs[0], A, rf);
  mf_fill_boundary(umac[0]); mf_fill_boundary(umac[1]);
  mf_temp_free(beta[0]); mf_temp_free(beta[1]); mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}

// ---- explicit diffusive term and the implicit viscous / diffusive solves ------------------------------------------------
struct Lap2Args { int lo[2], hi[2]; int e[2][2]; double hi2[2]; int comp; };
__global__ void kk2_lap(FV lap, FV data, Lap2Args A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const int q[2] = { i, j };
  const double p0 = G2(data, i, j, A.comp);
  double sum = 0.0;
  #pragma unroll
  for (int d = 0; d < 2; d++) {
    const int mi = i - (d == 0), mj = j - (d == 1), pi = i + (d == 0), pj = j + (d == 1);
    double fm = p0 - G2(data, mi, mj, A.comp), fp = G2(data, pi, pj, A.comp) - p0;
    if (q[d] == A.lo[d]) { if (A.e[d][0] == VDN_BC_NEU) fm = 0.0; else if (A.e[d][0] == VDN_BC_DIR) fm = 2.0 * fm; }
    if (q[d] == A.hi[d]) { if (A.e[d][1] == VDN_BC_NEU) fp = 0.0; else if (A.e[d][1] == VDN_BC_DIR) fp = 2.0 * fp; }
    sum = sum + (fp - fm) * A.hi2[d];
  }
  P2(lap, i, j, A.comp) = sum;
}
void k2_explicit_diffusive_term(vdn_multifab *lap, const vdn_multifab *data, int comp, int bccomp0, const double *dx, const vdn_bc_tower *bct) {
  for (int b = 0; b < lap->nfabs(); b++) {
    Lap2Args A; const vdn_box &bx = lap->vbox[b];
    for (int d = 0; d < 2; d++) { A.lo[d] = bx.lo[d]; A.hi[d] = bx.hi[d]; A.hi2[d] = 1.0 / (dx[d] * dx[d]); for (int s = 0; s < 2; s++) A.e[d][s] = bct->ell_bc(lap->lev, b + 1, d, s, bccomp0); }
    A.comp = comp;
    Range3 r = rng2(bx.lo[0], bx.hi[0], bx.lo[1], bx.hi[1]);
    hipLaunchKernelGGL(kk2_lap, grid_for(r), B2, 0, ctx().stream, lap->fabs[b], data->fabs[b], A, r);
  }
}
struct Vr2Args { int d, dtype; double mu, third_mudt, dxd; };
__global__ void kk2_visc_rhs(FV rh, FV phi, FV unew, FV rho, FV lapu, FV macrhs, Vr2Args A, Range3 rg, int lo0, int hi0, int lo1, int hi1) {
  THREAD_IJK(rg)                        // rg = valid box grown by 1: phi takes unew incl. the ghost layer
  if (!in_range) return;
  P2(phi, i, j, 0) = G2(unew, i, j, A.d);
  if (i < lo0 || i > hi0 || j < lo1 || j > hi1) return;
  double r = G2(unew, i, j, A.d) * G2(rho, i, j, 0);
  if (A.dtype == 1) r = r + A.mu * G2(lapu, i, j, A.d);
  const int pi = i + (A.d == 0), pj = j + (A.d == 1), mi = i - (A.d == 0), mj = j - (A.d == 1);
  r = r + A.third_mudt * (G2(macrhs, pi, pj, 0) - G2(macrhs, mi, mj, 0)) / A.dxd;
  P2(rh, i, j, 0) = r;
}
__global__ void kk2_diff_rhs(FV rh, FV phi, FV snew, FV laps, int comp, int dtype, double mu, Range3 rg, int lo0, int hi0, int lo1, int hi1) {
  THREAD_IJK(rg)
  if (!in_range) return;
  P2(phi, i, j, 0) = G2(snew, i, j, comp);
  if (i < lo0 || i > hi0 || j < lo1 || j > hi1) return;
  double r = G2(snew, i, j, comp);
  if (dtype == 1) r = r + mu * G2(laps, i, j, comp);
  P2(rh, i, j, 0) = r;
}
void do2_visc_solve(vdn_layout *mla, vdn_multifab *unew, const vdn_multifab *lapu, const vdn_multifab *rho, const vdn_multifab *mac_rhs,
                    const double *dx, double mu, const vdn_bc_tower *bct) {
  require_2d(unew, "visc_solve");
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  vdn_multifab *rh = mf_temp(mla, 0, 1, 0, -1, false, 0.0), *phi = mf_temp(mla, 0, 1, 1, -1, true, 0.0), *alpha = mf_temp(mla, 0, 1, 0, -1, false, 0.0);
  vdn_multifab *beta[2] = { mf_temp(mla, 0, 1, 0, 0, true, mu), mf_temp(mla, 0, 1, 0, 1, true, mu) };
  mf_copy(alpha, 0, rho, 0, 1, 0);
  const vdn_box &bx = unew->vbox[0];
  Range3 rg = rng2(bx.lo[0] - 1, bx.hi[0] + 1, bx.lo[1] - 1, bx.hi[1] + 1);
  const int dtype = ctx().prm.diffusion_type;
  for (int d = 0; d < 2; d++) {
    Vr2Args A; A.d = d; A.dtype = dtype; A.mu = mu; A.third_mudt = (1.0 / 3.0) * ((dtype == 1) ? 2.0 * mu : mu); A.dxd = dx[d];
    hipLaunchKernelGGL(kk2_visc_rhs, grid_for(rg), B2, 0, st, rh->fabs[0], phi->fabs[0], unew->fabs[0], rho->fabs[0], lapu ? lapu->fabs[0] : unew->fabs[0], mac_rhs->fabs[0], A, rg,
                       bx.lo[0], bx.hi[0], bx.lo[1], bx.hi[1]);
    int ebc[3][2];
    for (int a = 0; a < 3; a++) for (int s = 0; s < 2; s++) ebc[a][s] = a < 2 ? bct->ell_bc(0, 0, a, s, d) : VDN_BC_INT;
    int cyc; double r0, rr;
    solver_check(cc2_solve(rh, phi, beta, dx, ebc, 1.e-12, -1.0, ctx().prm.mg_max_iter, &cyc, &r0, &rr, alpha), "viscous solve (2-D)", cyc, rr, r0, d);
    mf_copy(unew, d, phi, 0, 1, 0);
  }
  mf_restrict_and_fill(unew, 0, 0, 2, false, bct);
  mf_temp_free(beta[0]); mf_temp_free(beta[1]); mf_temp_free(alpha); mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}
void do2_diff_scalar_solve(vdn_layout *mla, vdn_multifab *snew, const vdn_multifab *laps, const double *dx, double mu, const vdn_bc_tower *bct, int icomp, int bccomp0) {
  require_2d(snew, "diff_scalar_solve");
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  vdn_multifab *rh = mf_temp(mla, 0, 1, 0, -1, false, 0.0), *phi = mf_temp(mla, 0, 1, 1, -1, true, 0.0), *alpha = mf_temp(mla, 0, 1, 0, -1, true, 1.0);
  vdn_multifab *beta[2] = { mf_temp(mla, 0, 1, 0, 0, true, mu), mf_temp(mla, 0, 1, 0, 1, true, mu) };
  const vdn_box &bx = snew->vbox[0];
  Range3 rg = rng2(bx.lo[0] - 1, bx.hi[0] + 1, bx.lo[1] - 1, bx.hi[1] + 1);
  hipLaunchKernelGGL(kk2_diff_rhs, grid_for(rg), B2, 0, st, rh->fabs[0], phi->fabs[0], snew->fabs[0], laps ? laps->fabs[0] : snew->fabs[0], icomp, ctx().prm.diffusion_type, mu, rg,
                     bx.lo[0], bx.hi[0], bx.lo[1], bx.hi[1]);
  int ebc[3][2];
  for (int a = 0; a < 3; a++) for (int s = 0; s < 2; s++) ebc[a][s] = a < 2 ? bct->ell_bc(0, 0, a, s, bccomp0) : VDN_BC_INT;
  int cyc; double r0, rr;
  solver_check(cc2_solve(rh, phi, beta, dx, ebc, 1.e-12, -1.0, ctx().prm.mg_max_iter, &cyc, &r0, &rr, alpha), "diffusive solve (2-D)", cyc, rr, r0);
  mf_copy(snew, icomp, phi, 0, 1, 0);
  mf_restrict_and_fill(snew, icomp, bccomp0, 1, false, bct);
  mf_temp_free(beta[0]); mf_temp_free(beta[1]); mf_temp_free(alpha); mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}

// =====================================================================================================================
// nodal multigrid, 9-point Q1:  K phi = b,  b = -rh
// =====================================================================================================================
struct N2 { int n0, n1, PN, PS; double f[2]; double *phi, *tmp, *b, *res, *sig; int dirlo[2], dirhi[2]; };
DEVI long n2i(const N2 &L, int i, int j) { return (long)(i + 1) + (long)L.PN * (j + 1); }
DEVI long s2i(const N2 &L, int i, int j) { return (long)(i + 1) + (long)L.PS * (j + 1); }
DEVI bool n2_dir(const N2 &L, int i, int j) {
  return (i == 0 && L.dirlo[0]) || (i == L.n0 && L.dirhi[0]) || (j == 0 && L.dirlo[1]) || (j == L.n1 && L.dirhi[1]);
}
// cells (cj,ci) ascending, corners (my,mx) ascending -- the order of nd_apply's dm = 2 branch in oracle/vo_hgproject.c
DEVI void n2_apply(const N2 &L, const double *__restrict__ phi, int i, int j, double &Kp, double &diag) {
  const double fx = L.f[0], fy = L.f[1];
  double w[4];
  w[0] = 2.0 * (fx + fy); w[1] = -2.0 * fx + fy; w[2] = fx - 2.0 * fy; w[3] = -(fx + fy);
  double acc = 0.0, ssum = 0.0;
  #pragma unroll
  for (int dj = 0; dj < 2; dj++)
    #pragma unroll
    for (int di = 0; di < 2; di++) {
      const int ci = i - 1 + di, cj = j - 1 + dj;
      const double sg = L.sig[s2i(L, ci, cj)];
      double t = 0.0;
      #pragma unroll
      for (int my = 0; my < 2; my++)
        #pragma unroll
        for (int mx = 0; mx < 2; mx++) {
          const int ni = ci + mx, nj = cj + my;
          const int idx = (ni != i) | ((nj != j) << 1);
          t = t + w[idx] * phi[n2i(L, ni, nj)];
        }
      acc = acc + sg * t;
      ssum = ssum + sg;
    }
  Kp = acc; diag = w[0] * ssum;
}
__global__ void kk_n2_fill_nodes(N2 L, double *a, int per0, int per1) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  if (i > L.n0 + 1 || j > L.n1 + 1) return;
  int si = i, sj = j; bool g = false, zero = false;
  if (per0) { if (i < 0) { si = i + L.n0; g = true; } else if (i >= L.n0) { si = i - L.n0; g = true; } } else if (i < 0 || i > L.n0) { g = true; zero = true; }
  if (per1) { if (j < 0) { sj = j + L.n1; g = true; } else if (j >= L.n1) { sj = j - L.n1; g = true; } } else if (j < 0 || j > L.n1) { g = true; zero = true; }
  if (g) a[n2i(L, i, j)] = zero ? 0.0 : a[n2i(L, si, sj)];
}
__global__ void kk_n2_fill_cells(N2 L, int per0, int per1) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  if (i > L.n0 || j > L.n1) return;
  int si = i, sj = j; bool g = false, zero = false;
  if (i < 0) { g = true; if (per0) si = i + L.n0; else zero = true; } else if (i >= L.n0) { g = true; if (per0) si = i - L.n0; else zero = true; }
  if (j < 0) { g = true; if (per1) sj = j + L.n1; else zero = true; } else if (j >= L.n1) { g = true; if (per1) sj = j - L.n1; else zero = true; }
  if (g) L.sig[s2i(L, i, j)] = zero ? 0.0 : L.sig[s2i(L, si, sj)];
}
__global__ void kk_n2_jacobi(N2 L, const double *__restrict__ phi, double *__restrict__ out, double omega) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > L.n0 || j > L.n1) return;
  const long c = n2i(L, i, j);
  const double p0 = phi[c];
  double v = p0;
  if (!n2_dir(L, i, j)) { double Kp, diag; n2_apply(L, phi, i, j, Kp, diag); if (diag != 0.0) v = p0 + omega * ((L.b[c] - Kp) / diag); }
  out[c] = v;
}
__global__ void kk_n2_residual(N2 L, double *nrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  double r = 0.0;
  if (i <= L.n0 && j <= L.n1) {
    if (!n2_dir(L, i, j)) { double Kp, diag; n2_apply(L, L.phi, i, j, Kp, diag); r = L.b[n2i(L, i, j)] - Kp; }
    L.res[n2i(L, i, j)] = r;
  }
  if (nrm) block_atomic_max(nrm, fabs(r));
}
__global__ void kk_n2_restrict(N2 F, N2 C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > C.n0 || j > C.n1) return;
  const double wt[3] = { 0.5, 1.0, 0.5 };
  double s = 0.0;
  if (!n2_dir(C, i, j))
    for (int b = -1; b <= 1; b++) for (int a = -1; a <= 1; a++) s = s + (wt[a + 1] * wt[b + 1]) * F.res[n2i(F, 2 * i + a, 2 * j + b)];
  C.b[n2i(C, i, j)] = s * 0.25;
}
__global__ void kk_n2_prolong(N2 F, N2 C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i > F.n0 || j > F.n1 || n2_dir(F, i, j)) return;
  const int I = i >> 1, J = j >> 1, oi = i & 1, oj = j & 1;
  double s = 0.0;
  for (int b = 0; b <= oj; b++) for (int a = 0; a <= oi; a++) s = s + C.phi[n2i(C, I + a, J + b)];
  const double scale = 1.0 / (double)((1 + oi) * (1 + oj));
  F.phi[n2i(F, i, j)] = F.phi[n2i(F, i, j)] + s * scale;
}
__global__ void kk_n2_coarsen_sigma(N2 F, N2 C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= C.n0 || j >= C.n1) return;
  double s = 0.0;
  for (int b = 0; b < 2; b++) for (int a = 0; a < 2; a++) s = s + F.sig[s2i(F, 2 * i + a, 2 * j + b)];
  C.sig[s2i(C, i, j)] = s * 0.25;
}
__global__ void kk_n2_load_sigma(N2 L, FV coeffs, int lo0, int lo1) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  if (i > L.n0 || j > L.n1) return;
  L.sig[s2i(L, i, j)] = G2(coeffs, lo0 + i, lo1 + j, 0);
}
__global__ void kk_n2_load(N2 L, FV rh, FV phi, int lo0, int lo1, double *nrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  double r = 0.0;
  if (i <= L.n0 && j <= L.n1) {
    const bool dir = n2_dir(L, i, j);
    r = dir ? 0.0 : G2(rh, lo0 + i, lo1 + j, 0);
    L.b[n2i(L, i, j)] = -r;
    L.phi[n2i(L, i, j)] = dir ? 0.0 : G2(phi, lo0 + i, lo1 + j, 0);
  }
  block_atomic_max(nrm, fabs(r));
}
__global__ void kk_n2_store(N2 L, FV phi, int lo0, int lo1) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  if (i > L.n0 + 1 || j > L.n1 + 1) return;
  P2(phi, lo0 + i, lo1 + j, 0) = L.phi[n2i(L, i, j)];
}
__global__ void kk2_nd_divu(FV u, FV rh, double gx, double gy, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double dux = (G2(u, i, j, 0) + G2(u, i, j - 1, 0)) - (G2(u, i - 1, j, 0) + G2(u, i - 1, j - 1, 0));
  const double duy = (G2(u, i, j, 1) + G2(u, i - 1, j, 1)) - (G2(u, i, j - 1, 1) + G2(u, i - 1, j - 1, 1));
  P2(rh, i, j, 0) = G2(rh, i, j, 0) + (dux * gx + duy * gy);
}
struct ND2MG { std::vector<N2> lev; int per[2]; double *d_nrm; };
static long n2_nsize(int n0, int n1) { return (long)(n0 + 3) * (n1 + 3); }
static void n2_fill(const ND2MG &M, const N2 &L, double *a) { hipLaunchKernelGGL(kk_n2_fill_nodes, g2(L.n0 + 3, L.n1 + 3), B2, 0, ctx().stream, L, a, M.per[0], M.per[1]); }
static void n2_jacobi(const ND2MG &M, N2 &L, int ns) {
  for (int s = 0; s < ns; s++) {
    n2_fill(M, L, L.phi);
    hipLaunchKernelGGL(kk_n2_jacobi, g2(L.n0 + 1, L.n1 + 1), B2, 0, ctx().stream, L, (const double *)L.phi, L.tmp, ctx().prm.hg_omega);
    std::swap(L.phi, L.tmp);
  }
}
static void n2_residual(const ND2MG &M, N2 &L, bool norm) {
  n2_fill(M, L, L.phi);
  if (norm) HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), ctx().stream));
  hipLaunchKernelGGL(kk_n2_residual, g2(L.n0 + 1, L.n1 + 1), B2, 0, ctx().stream, L, norm ? M.d_nrm : (double *)nullptr);
  n2_fill(M, L, L.res);
}
static int n2_bottom(const N2 &L) { const int N = std::max(L.n0, L.n1); return std::max(ctx().prm.hg_nub, 2 * N * N); }
static void n2_vcycle(ND2MG &M, int l) {
  const vdn_params &P = ctx().prm;
  N2 &L = M.lev[l];
  HIPCHK(hipMemsetAsync(L.phi, 0, sizeof(double) * n2_nsize(L.n0, L.n1), ctx().stream));
  if (l == (int)M.lev.size() - 1) { n2_jacobi(M, L, n2_bottom(L)); return; }
  N2 &C = M.lev[l + 1];
  n2_jacobi(M, L, P.hg_nu1);
  n2_residual(M, L, false);
  hipLaunchKernelGGL(kk_n2_restrict, g2(C.n0 + 1, C.n1 + 1), B2, 0, ctx().stream, L, C);
  n2_vcycle(M, l + 1);
  n2_fill(M, C, C.phi);
  hipLaunchKernelGGL(kk_n2_prolong, g2(L.n0 + 1, L.n1 + 1), B2, 0, ctx().stream, L, C);
  n2_jacobi(M, L, P.hg_nu2);
}
int nd2_solve(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *coeffs, const vdn_multifab *u, const double *dx, const int bc[3][2],
              double rel_eps, double abs_eps, int max_iter, int *cycles, double *res0, double *res) {
  require_2d(coeffs, "nd_solve");
  REQUIRE(phi->ng >= 1 && rh->ng >= 1 && coeffs->ng >= 1, "nodal multigrid: phi, rh, coeffs need one ghost layer");
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  const vdn_box &bx = coeffs->vbox[0];
  ND2MG M; M.per[0] = coeffs->la->pmask[0]; M.per[1] = coeffs->la->pmask[1];
  M.d_nrm = (double *)arena_alloc(256);
  int n0 = bx.hi[0] - bx.lo[0] + 1, n1 = bx.hi[1] - bx.lo[1] + 1; double h0 = dx[0], h1 = dx[1];
  for (;;) {
    N2 L; L.n0 = n0; L.n1 = n1; L.PN = n0 + 3; L.PS = n0 + 2; L.f[0] = 1.0 / (6.0 * (h0 * h0)); L.f[1] = 1.0 / (6.0 * (h1 * h1));
    for (int d = 0; d < 2; d++) { L.dirlo[d] = bc[d][0] == VDN_BC_DIR; L.dirhi[d] = bc[d][1] == VDN_BC_DIR; }
    const long nn = n2_nsize(n0, n1), ns = (long)(n0 + 2) * (n1 + 2);
    double *base = (double *)arena_alloc(sizeof(double) * (4 * nn + ns));
    HIPCHK(hipMemsetAsync(base, 0, sizeof(double) * (4 * nn + ns), st));
    L.phi = base; L.tmp = base + nn; L.b = base + 2 * nn; L.res = base + 3 * nn; L.sig = base + 4 * nn;
    M.lev.push_back(L);
    if ((n0 & 1) || (n1 & 1) || n0 <= 2 || n1 <= 2 || M.lev.size() >= 31) break;
    n0 /= 2; n1 /= 2; h0 *= 2.0; h1 *= 2.0;
  }
  N2 &L0 = M.lev[0];
  hipLaunchKernelGGL(kk_n2_load_sigma, g2(L0.n0 + 2, L0.n1 + 2), B2, 0, st, L0, coeffs->fabs[0], bx.lo[0], bx.lo[1]);
  for (size_t l = 1; l < M.lev.size(); l++) {
    hipLaunchKernelGGL(kk_n2_coarsen_sigma, g2(M.lev[l].n0, M.lev[l].n1), B2, 0, st, M.lev[l - 1], M.lev[l]);
    hipLaunchKernelGGL(kk_n2_fill_cells, g2(M.lev[l].n0 + 2, M.lev[l].n1 + 2), B2, 0, st, M.lev[l], M.per[0], M.per[1]);
  }
  if (u) {
    Range3 rn = rng2(bx.lo[0], bx.hi[0] + 1, bx.lo[1], bx.hi[1] + 1);
    hipLaunchKernelGGL(kk2_nd_divu, grid_for(rn), B2, 0, st, u->fabs[0], rh->fabs[0], 0.5 / dx[0], 0.5 / dx[1], rn);
  }
  HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), st));
  hipLaunchKernelGGL(kk_n2_load, g2(L0.n0 + 1, L0.n1 + 1), B2, 0, st, L0, rh->fabs[0], phi->fabs[0], bx.lo[0], bx.lo[1], M.d_nrm);
  const double bnorm = read_scal(M.d_nrm);
  const vdn_params &P = ctx().prm;
  int cyc = 0; bool conv = (bnorm == 0.0); double rn = 0.0;
  while (!conv) {
    n2_jacobi(M, L0, M.lev.size() == 1 ? n2_bottom(L0) : P.hg_nu1);
    n2_residual(M, L0, true);
    rn = read_scal(M.d_nrm);
    if ((rn <= rel_eps * bnorm && bnorm < HUGE_VAL) || rn <= abs_eps) { conv = true; break; }
    if (cyc >= max_iter || !(rn < HUGE_VAL) || !(bnorm < HUGE_VAL)) break;     // also: a NaN / inf norm (the reductions turn NaN into +inf)
    if (M.lev.size() > 1) {
      hipLaunchKernelGGL(kk_n2_restrict, g2(M.lev[1].n0 + 1, M.lev[1].n1 + 1), B2, 0, st, L0, M.lev[1]);
      n2_vcycle(M, 1);
      n2_fill(M, M.lev[1], M.lev[1].phi);
      hipLaunchKernelGGL(kk_n2_prolong, g2(L0.n0 + 1, L0.n1 + 1), B2, 0, st, L0, M.lev[1]);
      n2_jacobi(M, L0, P.hg_nu2);
    }
    cyc++;
  }
  n2_fill(M, L0, L0.phi);
  hipLaunchKernelGGL(kk_n2_store, g2(L0.n0 + 3, L0.n1 + 3), B2, 0, st, L0, phi->fabs[0], bx.lo[0], bx.lo[1]);
  if (cycles) *cycles = cyc; if (res0) *res0 = bnorm; if (res) *res = rn;
  HIPCHK(hipStreamSynchronize(st));
  arena_release(mark);
  return conv ? 0 : 1;
}

// ---- HG projection ------------------------------------------------------------------------------------------------------
struct Uv2Args { int lo[2], hi[2]; int phys[2][2]; int proj_type; double dt, dtinv; };
__global__ void kk2_create_uvec(FV unew, FV uold, FV rhohalf, FV gp, Uv2Args A, Range3 r) {
  THREAD_IJK(r)                          // r = box grown by the ghost width of unew
  if (!in_range) return;
  const int q[2] = { i, j };
  bool g1 = true, wall_plane = false, inlet_plane = false;
  #pragma unroll
  for (int d = 0; d < 2; d++) {
    if (q[d] < A.lo[d] - 1 || q[d] > A.hi[d] + 1) g1 = false;
    if (q[d] == A.lo[d] - 1) { const int p = A.phys[d][0]; if (p == VDN_SLIP_WALL || p == VDN_NO_SLIP_WALL) wall_plane = true; if (p == VDN_INLET) inlet_plane = true; }
    if (q[d] == A.hi[d] + 1) { const int p = A.phys[d][1]; if (p == VDN_SLIP_WALL || p == VDN_NO_SLIP_WALL) wall_plane = true; if (p == VDN_INLET) inlet_plane = true; }
  }
  #pragma unroll
  for (int m = 0; m < 2; m++) {
    double gpv = 0.0;
    if (g1) { gpv = G2(gp, i, j, m); if (inlet_plane) { gpv = 0.0; P2(gp, i, j, m) = 0.0; } }
    if (wall_plane) { P2(unew, i, j, m) = 0.0; continue; }
    if (!g1) continue;
    double v = G2(unew, i, j, m);
    if (A.proj_type == VDN_PRESSURE_ITERS) v = (v - G2(uold, i, j, m)) * A.dtinv;
    else if (A.proj_type == VDN_REGULAR_TIMESTEP) v = v + A.dt * gpv / G2(rhohalf, i, j, 0);
    P2(unew, i, j, m) = v;
  }
}
__global__ void kk2_coeffs(FV coeffs, FV rhohalf, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  P2(coeffs, i, j, 0) = 1.0 / G2(rhohalf, i, j, 0);
}
struct Hg2Args { int hi[2]; double dt, dtinv, dxi[2]; int proj_type; };
__global__ void kk2_hg_update(FV unew, FV uold, FV gp, FV rhohalf, FV p, FV phi, Hg2Args A, Range3 r) {
  THREAD_IJK(r)                          // r covers nodes lo..hi+1
  if (!in_range) return;
  if (i <= A.hi[0] && j <= A.hi[1]) {
    const double gph[2] = { 0.5 * (G2(phi, i + 1, j, 0) + G2(phi, i + 1, j + 1, 0) - G2(phi, i, j, 0) - G2(phi, i, j + 1, 0)) * A.dxi[0],      // mkgphi_2d
                            0.5 * (G2(phi, i, j + 1, 0) + G2(phi, i + 1, j + 1, 0) - G2(phi, i, j, 0) - G2(phi, i + 1, j, 0)) * A.dxi[1] };
    const double rho = G2(rhohalf, i, j, 0);
    #pragma unroll
    for (int m = 0; m < 2; m++) {
      double v = G2(unew, i, j, m) - gph[m] / rho;
      if (A.proj_type == VDN_PRESSURE_ITERS) v = G2(uold, i, j, m) + A.dt * v;
      P2(unew, i, j, m) = v;
      if (A.proj_type == VDN_PRESSURE_ITERS) P2(gp, i, j, m) = G2(gp, i, j, m) + gph[m];
      else if (A.proj_type == VDN_REGULAR_TIMESTEP) P2(gp, i, j, m) = A.dtinv * gph[m];
    }
  }
  if (A.proj_type == VDN_PRESSURE_ITERS) P2(p, i, j, 0) = G2(p, i, j, 0) + G2(phi, i, j, 0);
  else if (A.proj_type == VDN_REGULAR_TIMESTEP) P2(p, i, j, 0) = A.dtinv * G2(phi, i, j, 0);
}
void do2_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold, vdn_multifab **rhohalf, vdn_multifab **p, vdn_multifab **gp,
                   const double *dx, double dt, const vdn_bc_tower *bct, int press_comp0) {
  vdn_multifab *un = unew[0], *uo = uold[0], *rhh = rhohalf[0], *pp = p[0], *gpp = gp[0];
  require_2d(un, "hgproject");
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  vdn_multifab *rh = mf_temp(mla, 0, 1, 1, 3, true, 0.0), *phi = mf_temp(mla, 0, 1, 1, 3, true, 0.0), *coeffs = mf_temp(mla, 0, 1, 1, -1, true, 0.0);
  const vdn_box &bx = un->vbox[0];
  BoxP bp = make_boxp(un, 0, bct);
  Uv2Args A;
  for (int d = 0; d < 2; d++) { A.lo[d] = bx.lo[d]; A.hi[d] = bx.hi[d]; A.phys[d][0] = bp.phys[d][0]; A.phys[d][1] = bp.phys[d][1]; }
  A.proj_type = proj_type; A.dt = dt; A.dtinv = 1.0 / dt;
  Range3 rgn = rng2(bx.lo[0] - un->ng, bx.hi[0] + un->ng, bx.lo[1] - un->ng, bx.hi[1] + un->ng), rv = rng2(bx.lo[0], bx.hi[0], bx.lo[1], bx.hi[1]);
  hipLaunchKernelGGL(kk2_create_uvec, grid_for(rgn), B2, 0, st, un->fabs[0], uo->fabs[0], rhh->fabs[0], gpp->fabs[0], A, rgn);
  hipLaunchKernelGGL(kk2_coeffs, grid_for(rv), B2, 0, st, coeffs->fabs[0], rhh->fabs[0], rv);
  mf_fill_boundary(un); mf_fill_boundary(coeffs);
  double rel = ctx().prm.hg_rel_eps > 0.0 ? ctx().prm.hg_rel_eps : 1.e-12;
  double abs_eps = -1.0;
  if (proj_type == VDN_INITIAL_PROJECTION && ctx().prm.prob_type == 4) abs_eps = 1.e-12;
  int ebc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc[d][s] = d < 2 ? bct->ell_bc(0, 0, d, s, press_comp0) : VDN_BC_INT;
  int cyc; double r0, rr;
  int rc = nd2_solve(rh, phi, coeffs, un, dx, ebc, rel, abs_eps, ctx().prm.hg_max_iter, &cyc, &r0, &rr);
  ctx().solver_cycles[1] = cyc; ctx().solver_res0[1] = r0; ctx().solver_res[1] = rr;
  solver_check(rc, "nodal multigrid (2-D)", cyc, rr, r0);
  if (proj_type == VDN_INITIAL_PROJECTION || proj_type == VDN_DIVU_ITERS) { mf_setval(gpp, 0.0, 0, gpp->nc, true); mf_setval(pp, 0.0, 0, 1, true); }
  Hg2Args H; H.hi[0] = bx.hi[0]; H.hi[1] = bx.hi[1]; H.dt = dt; H.dtinv = 1.0 / dt; H.dxi[0] = 1.0 / dx[0]; H.dxi[1] = 1.0 / dx[1]; H.proj_type = proj_type;
  Range3 rn = rng2(bx.lo[0], bx.hi[0] + 1, bx.lo[1], bx.hi[1] + 1);
  hipLaunchKernelGGL(kk2_hg_update, grid_for(rn), B2, 0, st, un->fabs[0], uo->fabs[0], gpp->fabs[0], rhh->fabs[0], pp->fabs[0], phi->fabs[0], H, rn);
  mf_fill_boundary(gpp); mf_fill_boundary(pp);
  mf_temp_free(coeffs); mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}
