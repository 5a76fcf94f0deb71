// amr.hip -- multi-level (fixed grids) AMR pieces of the hot path (BASELINE.json configs[3], [4]).
//
// The FBoxLib multi-level operators the reference calls are not in its tree; the call sites fix what they must do, the
// definitions are ours (stated in oracle/vo_amr.c, which this file mirrors operation for operation):
//   ml_cc_restriction     coarse cell = mean of its 8 fine cells                 (macproject.f90:204-206, hgproject.f90:355-357)
//   ml_edge_restriction   coarse face = mean of the 4 fine faces covering it     (velpred.f90:115-119, macproject.f90:330-333, 497-500)
//   fill_ghost_cells      fine ghost cell = coarse parent + MC-limited linear slopes   (macproject.f90:304-310, ml_restrict_and_fill)
//   create_umac_grown     fine ghost face = coarse face (even index) / mean of the two coarse faces around it (odd)
//   ml_cc_solve           composite solve by FAC iteration: composite residual (quadratic coarse-fine ghost cells, coarse flux
//                         through an interface face = mean of the four fine fluxes), nu1 red-black sweeps on the levels finest to 1,
//                         one V-cycle of the single-level multigrid on the whole level 0, piecewise-constant prolongation, nu2 sweeps
//                         on the levels 1 to finest
// This round: up to VDN_MAXLEV levels, refinement ratio 2, every box on this rank (nranks = 1); the levels must be properly nested
// (a level-n box keeps at least two level-(n-1) cells between itself and the edge of level n-1, or touches the domain boundary).
#include "vdn_dev.h"
#include <vector>
#include <algorithm>

static const dim3 AB(64, 4, 1);
static void require_amr(const vdn_layout *la) {
  REQUIRE(la->nlev >= 2 && la->nlev <= VDN_MAXLEV, "AMR path: 2..%d levels are implemented (nlevel = %d)", VDN_MAXLEV, la->nlev);
  REQUIRE(ctx().nranks == 1, "AMR path: single rank only in this round");
  REQUIRE(ctx().prm.dm == 3, "AMR path: dm = 3 only");
  for (size_t d = 0; d < la->rr.size(); d++) REQUIRE(la->rr[d] == 2, "AMR path: refinement ratio 2 only");
}
DEVI int fdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
static int hfdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
static bool isect(const int alo[3], const int ahi[3], const int blo[3], const int bhi[3], Range3 &r) {
  for (int d = 0; d < 3; d++) { r.lo[d] = std::max(alo[d], blo[d]); r.hi[d] = std::min(ahi[d], bhi[d]); if (r.lo[d] > r.hi[d]) return false; }
  return true;
}

// ---- restriction ----------------------------------------------------------------------------------------------------
__global__ void kk_ml_restrict(FV crse, FV fine, Range3 r, int icomp, int nc) {
  THREAD_IJK(r)
  if (!in_range) return;
  for (int c = icomp; c < icomp + nc; c++) {
    double s = 0.0;
    #pragma unroll
    for (int kk = 0; kk < 2; kk++)
      #pragma unroll
      for (int jj = 0; jj < 2; jj++)
        #pragma unroll
        for (int ii = 0; ii < 2; ii++) s = s + fv_get(fine, 2 * i + ii, 2 * j + jj, 2 * k + kk, c);
    fv_at(crse, i, j, k, c) = s * 0.125;
  }
}
void ml_cc_restriction(vdn_multifab *crse, const vdn_multifab *fine, int icomp, int nc) {
  for (int f = 0; f < fine->nfabs(); f++) for (int c = 0; c < crse->nfabs(); c++) {
    int clo[3], chi[3]; Range3 r;
    for (int d = 0; d < 3; d++) { clo[d] = fine->vbox[f].lo[d] / 2; chi[d] = fine->vbox[f].hi[d] / 2; }
    if (!isect(clo, chi, crse->vbox[c].lo, crse->vbox[c].hi, r)) continue;
    hipLaunchKernelGGL(kk_ml_restrict, grid_for(r), AB, 0, ctx().stream, crse->fabs[c], fine->fabs[f], r, icomp, nc);
  }
}
__global__ void kk_ml_edge_restrict(FV crse, FV fine, Range3 r, int dir) {
  THREAD_IJK(r)
  if (!in_range) return;
  const int Q[3] = { i, j, k };
  const int t1 = (dir + 1) % 3, t2 = (dir + 2) % 3;
  double s = 0.0;
  #pragma unroll
  for (int b = 0; b < 2; b++)
    #pragma unroll
    for (int a = 0; a < 2; a++) {
      int q[3]; q[dir] = 2 * Q[dir]; q[t1] = 2 * Q[t1] + a; q[t2] = 2 * Q[t2] + b;
      s = s + fv_get(fine, q[0], q[1], q[2]);
    }
  fv_at(crse, i, j, k) = s * 0.25;
}
void ml_edge_restriction(vdn_multifab *crse, const vdn_multifab *fine, int dir) {
  for (int f = 0; f < fine->nfabs(); f++) for (int c = 0; c < crse->nfabs(); c++) {
    int clo[3], chi[3], blo[3], bhi[3]; Range3 r;
    for (int d = 0; d < 3; d++) { clo[d] = fine->vbox[f].lo[d] / 2; chi[d] = fine->vbox[f].hi[d] / 2; blo[d] = crse->vbox[c].lo[d]; bhi[d] = crse->vbox[c].hi[d]; }
    chi[dir] += 1; bhi[dir] += 1;
    if (!isect(clo, chi, blo, bhi, r)) continue;
    hipLaunchKernelGGL(kk_ml_edge_restrict, grid_for(r), AB, 0, ctx().stream, crse->fabs[c], fine->fabs[f], r, dir);
  }
}

// ---- coarse -> fine ghost interpolation ---------------------------------------------------------------------------------
struct InterpArgs { int flo[3], fhi[3]; int plo[3], phi[3]; int alo[3], ahi[3]; int icomp, nc; };
DEVI double mc_limited(double del, double sm, double s0, double sp) {
  const double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
  double slim = fmin(fabs(dpls), fabs(dmin));
  slim = (dpls * dmin > 0.0) ? slim : 0.0;
  return copysign(1.0, del) * fmin(slim, fabs(del));
}
// r: fine cells (the grown fine box); a thread writes its cell if it is a ghost cell whose parent lies in [plo,phi];
// [alo,ahi]: the allocation of the coarse fab (slopes need both neighbours inside it)
__global__ void kk_ml_interp_ghost(FV fine, FV crse, InterpArgs A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  if (i >= A.flo[0] && i <= A.fhi[0] && j >= A.flo[1] && j <= A.fhi[1] && k >= A.flo[2] && k <= A.fhi[2]) return;
  const int q[3] = { i, j, k }, P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
  #pragma unroll
  for (int d = 0; d < 3; d++) if (P[d] < A.plo[d] || P[d] > A.phi[d]) return;
  for (int c = A.icomp; c < A.icomp + A.nc; c++) {
    const double c0 = fv_get(crse, P[0], P[1], P[2], c);
    double v = c0;
    #pragma unroll
    for (int d = 0; d < 3; d++) {
      double sl = 0.0;
      if (P[d] - 1 >= A.alo[d] && P[d] + 1 <= A.ahi[d]) {
        const double cm = fv_get(crse, P[0] - (d == 0), P[1] - (d == 1), P[2] - (d == 2), c), cp = fv_get(crse, P[0] + (d == 0), P[1] + (d == 1), P[2] + (d == 2), c);
        sl = mc_limited(0.5 * (cp - cm), cm, c0, cp);
      }
      const double sg = (q[d] - 2 * P[d]) ? 0.25 : -0.25;
      v = v + sg * sl;
    }
    fv_at(fine, i, j, k, c) = v;
  }
}
// parents inside a coarse box's VALID region come from that box; parents outside the domain (physical / periodic ghost
// cells of the coarse level) come from the first coarse box whose allocation holds them.  A one-box coarse level that does not
// cover the domain: parents outside the box (its own ghost cells, filled from the next coarser level) count as "outside"
void ml_fill_ghost_cells(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc) {
  if (fine->ng == 0) return;
  const vdn_box &pdc = crse->nfabs() == 1 ? crse->vbox[0] : crse->la->pd[crse->lev];
  for (int f = 0; f < fine->nfabs(); f++) {
    InterpArgs A; Range3 r;
    int glo[3], ghi[3];
    for (int d = 0; d < 3; d++) { A.flo[d] = fine->vbox[f].lo[d]; A.fhi[d] = fine->vbox[f].hi[d]; r.lo[d] = A.flo[d] - fine->ng; r.hi[d] = A.fhi[d] + fine->ng;
      glo[d] = hfdiv2(r.lo[d]); ghi[d] = hfdiv2(r.hi[d]); }
    A.icomp = icomp; A.nc = nc;
    for (int pass = 0; pass < 2; pass++)
      for (int c = 0; c < crse->nfabs(); c++) {
        int blo[3], bhi[3]; Range3 pr;
        for (int d = 0; d < 3; d++) { A.alo[d] = crse->vbox[c].lo[d] - crse->ng; A.ahi[d] = crse->vbox[c].hi[d] + crse->ng;
          blo[d] = pass == 0 ? crse->vbox[c].lo[d] : A.alo[d]; bhi[d] = pass == 0 ? crse->vbox[c].hi[d] : A.ahi[d]; }
        if (!isect(glo, ghi, blo, bhi, pr)) continue;
        if (pass == 1) {
          // only parents outside the domain; handled per direction slab to stay disjoint from pass 0: use the whole
          // allocation but skip parents inside the domain in the kernel via plo/phi restricted below
          bool any = false;
          for (int d = 0; d < 3; d++) if (pr.lo[d] < pdc.lo[d] || pr.hi[d] > pdc.hi[d]) any = true;
          if (!any) continue;
        }
        for (int d = 0; d < 3; d++) { A.plo[d] = pr.lo[d]; A.phi[d] = pr.hi[d]; }
        if (pass == 1) {
          // launch one slab per (direction, side) that sticks out of the domain
          for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
            InterpArgs B = A;
            if (s == 0) { if (pr.lo[d] >= pdc.lo[d]) continue; B.phi[d] = std::min(pr.hi[d], pdc.lo[d] - 1); }
            else        { if (pr.hi[d] <= pdc.hi[d]) continue; B.plo[d] = std::max(pr.lo[d], pdc.hi[d] + 1); }
            for (int e = 0; e < d; e++) { B.plo[e] = std::max(B.plo[e], pdc.lo[e]); B.phi[e] = std::min(B.phi[e], pdc.hi[e]); if (B.plo[e] > B.phi[e]) goto next; }
            hipLaunchKernelGGL(kk_ml_interp_ghost, grid_for(r), AB, 0, ctx().stream, fine->fabs[f], crse->fabs[c], B, r);
            next:;
          }
          if (crse->nfabs() > 0) break;      // first coarse box whose allocation holds them
        } else
          hipLaunchKernelGGL(kk_ml_interp_ghost, grid_for(r), AB, 0, ctx().stream, fine->fabs[f], crse->fabs[c], A, r);
      }
  }
}
struct GrownArgs { int flo[3], fhi[3]; int plo[3], phi[3]; int dir; };
__global__ void kk_ml_umac_grown(FV fine, FV crse, GrownArgs A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  if (i >= A.flo[0] && i <= A.fhi[0] && j >= A.flo[1] && j <= A.fhi[1] && k >= A.flo[2] && k <= A.fhi[2]) return;     // valid faces
  const int q[3] = { i, j, k };
  int P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
  const int odd = q[A.dir] - 2 * P[A.dir];
  #pragma unroll
  for (int d = 0; d < 3; d++) if (P[d] < A.plo[d] || P[d] + ((d == A.dir) ? odd : 0) > A.phi[d]) return;
  const double a = fv_get(crse, P[0], P[1], P[2]);
  double v = a;
  if (odd) { P[A.dir] += 1; v = 0.5 * (a + fv_get(crse, P[0], P[1], P[2])); }
  fv_at(fine, i, j, k) = v;
}
// parents on VALID faces of a coarse box come from that box; the remaining ones (outside the domain, or -- for a one-box coarse
// level that does not cover the domain -- in that box's own ghost faces) from the first coarse box whose allocation holds them
void ml_create_umac_grown(vdn_multifab *fine, const vdn_multifab *crse, int dir) {
  for (int f = 0; f < fine->nfabs(); f++) {
    GrownArgs A; Range3 r; A.dir = dir;
    for (int d = 0; d < 3; d++) { A.flo[d] = fine->vbox[f].lo[d]; A.fhi[d] = fine->vbox[f].hi[d] + (d == dir); r.lo[d] = A.flo[d] - fine->ng; r.hi[d] = A.fhi[d] + fine->ng; }
    // the first box with its ghost faces (filled by fill_boundary), then every box's valid faces on top (same values where both hold one)
    for (int d = 0; d < 3; d++) { A.plo[d] = crse->vbox[0].lo[d] - crse->ng; A.phi[d] = crse->vbox[0].hi[d] + (d == dir) + crse->ng; }
    hipLaunchKernelGGL(kk_ml_umac_grown, grid_for(r), AB, 0, ctx().stream, fine->fabs[f], crse->fabs[0], A, r);
    for (int c = 0; c < crse->nfabs() && crse->nfabs() > 1; c++) {
      for (int d = 0; d < 3; d++) { A.plo[d] = crse->vbox[c].lo[d]; A.phi[d] = crse->vbox[c].hi[d] + (d == dir); }
      hipLaunchKernelGGL(kk_ml_umac_grown, grid_for(r), AB, 0, ctx().stream, fine->fabs[f], crse->fabs[c], A, r);
    }
  }
}
void ml_restrict_and_fill(int nlev, vdn_multifab **mf, int icomp, int bcomp, int nc, bool same_boundary, const vdn_bc_tower *bct) {
  for (int n = nlev - 1; n >= 1; n--) ml_cc_restriction(mf[n - 1], mf[n], icomp, nc);
  for (int n = 0; n < nlev; n++) {
    if (n > 0) ml_fill_ghost_cells(mf[n], mf[n - 1], icomp, nc);
    mf_fill_boundary(mf[n]);
    mf_physbc(mf[n], icomp, bcomp, nc, bct, same_boundary);
  }
}

// ---- composite cell-centred solve -------------------------------------------------------------------------------------------
struct FaceBc { int lo[3], hi[3]; int e[3][2]; };
__global__ void kk_phi_closure(FV phi, FaceBc B, int d, int s, Range3 r) {
  THREAD_IJK(r)                              // r: the boundary cells of face (d,s)
  if (!in_range) return;
  const double v = fv_get(phi, i, j, k);
  const int gi = i + (d == 0 ? (s ? 1 : -1) : 0), gj = j + (d == 1 ? (s ? 1 : -1) : 0), gk = k + (d == 2 ? (s ? 1 : -1) : 0);
  fv_at(phi, gi, gj, gk) = (B.e[d][s] == VDN_BC_NEU) ? v : -v;
}
static void phi_closure(vdn_multifab *phi, const vdn_bc_tower *bct, int bc_comp0) {
  for (int b = 0; b < phi->nfabs(); b++) {
    FaceBc B;
    for (int d = 0; d < 3; d++) { B.lo[d] = phi->vbox[b].lo[d]; B.hi[d] = phi->vbox[b].hi[d]; for (int s = 0; s < 2; s++) B.e[d][s] = bct->ell_bc(phi->lev, b + 1, d, s, bc_comp0); }
    for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
      if (B.e[d][s] != VDN_BC_NEU && B.e[d][s] != VDN_BC_DIR) continue;
      Range3 r; for (int t = 0; t < 3; t++) { r.lo[t] = B.lo[t]; r.hi[t] = B.hi[t]; }
      r.lo[d] = r.hi[d] = s ? B.hi[d] : B.lo[d];
      hipLaunchKernelGGL(kk_phi_closure, grid_for(r), AB, 0, ctx().stream, phi->fabs[b], B, d, s, r);
    }
  }
  mf_fill_boundary(phi);
}
struct CfArgs { int d, s; int plo[3], phi[3]; };
__global__ void kk_cf_interp(FV pf, FV pc, CfArgs A, Range3 r) {
  THREAD_IJK(r)                              // r: the ghost cells just outside face (d,s) of the fine box
  if (!in_range) return;
  const int g[3] = { i, j, k };
  const int P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
  #pragma unroll
  for (int t = 0; t < 3; t++) if (P[t] < A.plo[t] || P[t] > A.phi[t]) return;
  const int in = A.s ? -1 : 1;
  int f1[3] = { i, j, k }, f2[3] = { i, j, k }; f1[A.d] += in; f2[A.d] += 2 * in;
  double pcs = fv_get(pc, P[0], P[1], P[2]);
  #pragma unroll
  for (int t = 0; t < 3; t++) {
    if (t == A.d) continue;
    const double sg = (g[t] - 2 * P[t]) ? 0.125 : -0.125;
    pcs = pcs + sg * (fv_get(pc, P[0] + (t == 0), P[1] + (t == 1), P[2] + (t == 2)) - fv_get(pc, P[0] - (t == 0), P[1] - (t == 1), P[2] - (t == 2)));
  }
  fv_at(pf, i, j, k) = (8.0 / 15.0) * pcs + (2.0 / 3.0) * fv_get(pf, f1[0], f1[1], f1[2]) - 0.2 * fv_get(pf, f2[0], f2[1], f2[2]);
}
// ghost cells of the fine phi: coarse-fine interpolation on every face that is not a domain face, then the same-level exchange
// (which overwrites the cells that another fine box covers), then nothing else: domain faces were closed by phi_closure
static void cf_interp(vdn_multifab *pf, const vdn_multifab *pc, const vdn_bc_tower *bct, int bc_comp0) {
  for (int f = 0; f < pf->nfabs(); f++) for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
    if (bct->ell_bc(pf->lev, f + 1, d, s, bc_comp0) != VDN_BC_INT) continue;
    Range3 r; for (int t = 0; t < 3; t++) { r.lo[t] = pf->vbox[f].lo[t]; r.hi[t] = pf->vbox[f].hi[t]; }
    r.lo[d] = r.hi[d] = s ? pf->vbox[f].hi[d] + 1 : pf->vbox[f].lo[d] - 1;
    for (int c = 0; c < pc->nfabs(); c++) {
      CfArgs A; A.d = d; A.s = s;
      for (int t = 0; t < 3; t++) { A.plo[t] = pc->vbox[c].lo[t]; A.phi[t] = pc->vbox[c].hi[t]; }
      hipLaunchKernelGGL(kk_cf_interp, grid_for(r), AB, 0, ctx().stream, pf->fabs[f], pc->fabs[c], A, r);
    }
  }
  mf_fill_boundary(pf);
}
struct ResArgs { double hi2[3]; };
__global__ void kk_amr_residual(FV rh, FV phi, FV bx, FV by, FV bz, FV res, FV mask, int has_mask, ResArgs A, Range3 r, double *nrm) {
  REDUCE_IJ(r)
  double rmax = 0.0;
  if (in_ij) REDUCE_KLOOP(r) {
    const double p0 = fv_get(phi, i, j, k);
    const double ax = (fv_get(bx, i + 1, j, k) * (p0 - fv_get(phi, i + 1, j, k)) + fv_get(bx, i, j, k) * (p0 - fv_get(phi, i - 1, j, k))) * A.hi2[0];
    const double ay = (fv_get(by, i, j + 1, k) * (p0 - fv_get(phi, i, j + 1, k)) + fv_get(by, i, j, k) * (p0 - fv_get(phi, i, j - 1, k))) * A.hi2[1];
    const double az = (fv_get(bz, i, j, k + 1) * (p0 - fv_get(phi, i, j, k + 1)) + fv_get(bz, i, j, k) * (p0 - fv_get(phi, i, j, k - 1))) * A.hi2[2];
    const double rr = fv_get(rh, i, j, k) - (ax + ay + az);
    fv_at(res, i, j, k) = rr;
    if (!(has_mask && fv_get(mask, i, j, k) != 0.0)) rmax = fmax(rmax, fabs(rr));
  }
  if (nrm) block_atomic_max(nrm, rmax);
}
__global__ void kk_absmax_masked(FV a, FV mask, int has_mask, Range3 r, double *nrm) {
  REDUCE_IJ(r)
  double m = 0.0;
  if (in_ij) REDUCE_KLOOP(r) if (!(has_mask && fv_get(mask, i, j, k) != 0.0)) m = fmax(m, fabs(fv_get(a, i, j, k)));
  block_atomic_max(nrm, m);
}
struct RefluxArgs { int d, s; double dxf, dxc; };
// r: coarse faces (index along d fixed = the interface); the uncovered cell is on the outside of the fine box
__global__ void kk_reflux(FV res_c, FV phi_c, FV beta_c, FV mask, FV phi_f, FV beta_f, RefluxArgs A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const int Q[3] = { i, j, k };
  int M[3] = { i, j, k }; M[A.d] -= 1;
  const int *out = A.s == 0 ? M : Q;                    // the cell outside the fine box
  if (fv_get(mask, out[0], out[1], out[2]) != 0.0) return;   // covered by another fine box: not a coarse-fine interface
  const int t1 = (A.d + 1) % 3, t2 = (A.d + 2) % 3;
  double sum = 0.0;
  #pragma unroll
  for (int b = 0; b < 2; b++)
    #pragma unroll
    for (int a = 0; a < 2; a++) {
      int q[3], m[3]; q[A.d] = 2 * Q[A.d]; q[t1] = 2 * Q[t1] + a; q[t2] = 2 * Q[t2] + b; m[0] = q[0]; m[1] = q[1]; m[2] = q[2]; m[A.d] -= 1;
      sum = sum + fv_get(beta_f, q[0], q[1], q[2]) * (fv_get(phi_f, q[0], q[1], q[2]) - fv_get(phi_f, m[0], m[1], m[2])) / A.dxf;
    }
  const double Ff = sum * 0.25;
  const double Fc = fv_get(beta_c, Q[0], Q[1], Q[2]) * (fv_get(phi_c, Q[0], Q[1], Q[2]) - fv_get(phi_c, M[0], M[1], M[2])) / A.dxc;
  if (A.s == 0) fv_at(res_c, M[0], M[1], M[2]) = fv_get(res_c, M[0], M[1], M[2]) + (Ff - Fc) / A.dxc;
  else          fv_at(res_c, Q[0], Q[1], Q[2]) = fv_get(res_c, Q[0], Q[1], Q[2]) - (Ff - Fc) / A.dxc;
}
struct GsArgs { int lo[3], hi[3]; int e[3][2]; double hi2[3]; };
// red-black Gauss-Seidel on the fabs of the fine level: ghost cells of e are 0 at the coarse-fine interface and at Dirichlet
// faces (b := 2b), Neumann faces carry b := 0 -- the folding of mg_cc.hip applied on the fly; colour by global index
__global__ void kk_amr_gsrb(FV e, FV rh, FV bx, FV by, FV bz, GsArgs A, int color, Range3 r) {
  const int j = r.lo[1] + (int)(blockIdx.y * blockDim.y + threadIdx.y), k = r.lo[2] + (int)blockIdx.z;
  const int i = r.lo[0] + 2 * (int)(blockIdx.x * blockDim.x + threadIdx.x) + ((r.lo[0] + j + k + color) & 1);
  if (i > r.hi[0] || j > r.hi[1] || k > r.hi[2]) return;
  double bxm = fv_get(bx, i, j, k), bxp = fv_get(bx, i + 1, j, k), bym = fv_get(by, i, j, k), byp = fv_get(by, i, j + 1, k), bzm = fv_get(bz, i, j, k), bzp = fv_get(bz, i, j, k + 1);
  #define FOLD(b, dd, ss) { const int t = A.e[dd][ss]; if (t == VDN_BC_NEU) b = 0.0; else if (t == VDN_BC_DIR) b = 2.0 * b; }
  if (i == A.lo[0]) FOLD(bxm, 0, 0) if (i == A.hi[0]) FOLD(bxp, 0, 1)
  if (j == A.lo[1]) FOLD(bym, 1, 0) if (j == A.hi[1]) FOLD(byp, 1, 1)
  if (k == A.lo[2]) FOLD(bzm, 2, 0) if (k == A.hi[2]) FOLD(bzp, 2, 1)
  #undef FOLD
  const double p0 = fv_get(e, i, j, k);
  const double ax = (bxp * (p0 - fv_get(e, i + 1, j, k)) + bxm * (p0 - fv_get(e, i - 1, j, k))) * A.hi2[0];
  const double ay = (byp * (p0 - fv_get(e, i, j + 1, k)) + bym * (p0 - fv_get(e, i, j - 1, k))) * A.hi2[1];
  const double az = (bzp * (p0 - fv_get(e, i, j, k + 1)) + bzm * (p0 - fv_get(e, i, j, k - 1))) * A.hi2[2];
  const double Ap = ax + ay + az;
  const double diag = (bxp + bxm) * A.hi2[0] + (byp + bym) * A.hi2[1] + (bzp + bzm) * A.hi2[2];
  if (diag != 0.0) fv_at(e, i, j, k) = p0 + (fv_get(rh, i, j, k) - Ap) / diag;
}
__global__ void kk_add(FV a, FV b, Range3 r) { THREAD_IJK(r) if (!in_range) return; fv_at(a, i, j, k) = fv_get(a, i, j, k) + fv_get(b, i, j, k); }
// af += the parent's increment; keep: also store that increment in sc (the next finer level prolongs it in turn)
__global__ void kk_add_prolong(FV af, FV sc, int keep, FV ec, Range3 r, int plo0, int plo1, int plo2, int phi0, int phi1, int phi2) {
  THREAD_IJK(r)
  if (!in_range) return;
  const int I = i / 2, J = j / 2, K = k / 2;
  if (I < plo0 || I > phi0 || J < plo1 || J > phi1 || K < plo2 || K > phi2) return;
  const double v = fv_get(ec, I, J, K);
  if (keep) fv_at(sc, i, j, k) = v;
  fv_at(af, i, j, k) = fv_get(af, i, j, k) + v;
}
__global__ void kk_setbox(FV a, Range3 r, double v) { THREAD_IJK(r) if (!in_range) return; fv_at(a, i, j, k) = v; }

static Range3 valid_range(const vdn_multifab *mf, int b) { Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = mf->vbox[b].lo[d]; r.hi[d] = mf->vbox[b].hi[d]; } return r; }
static double read_dev(double *d) { double h; HIPCHK(hipMemcpyAsync(&h, d, sizeof(double), hipMemcpyDeviceToHost, ctx().stream)); HIPCHK(hipStreamSynchronize(ctx().stream)); return h; }

struct MLCC { int nlev; vdn_layout *la; vdn_multifab **rh, **phi, **beta; vdn_multifab *res[VDN_MAXLEV], *e[VDN_MAXLEV], *scr[VDN_MAXLEV], *mask[VDN_MAXLEV];
              const double *dx; const vdn_bc_tower *bct; int bcc; double *d_nrm; };
static void level_residual(MLCC &S, int n, bool norm) {
  ResArgs A; for (int d = 0; d < 3; d++) A.hi2[d] = 1.0 / (S.dx[3 * n + d] * S.dx[3 * n + d]);
  for (int b = 0; b < S.rh[n]->nfabs(); b++) {
    Range3 r = valid_range(S.rh[n], b);
    hipLaunchKernelGGL(kk_amr_residual, reduce_grid(r), AB, 0, ctx().stream, S.rh[n]->fabs[b], S.phi[n]->fabs[b], S.beta[3 * n]->fabs[b], S.beta[3 * n + 1]->fabs[b], S.beta[3 * n + 2]->fabs[b],
                       S.res[n]->fabs[b], S.rh[n]->fabs[b], 0, A, r, norm ? S.d_nrm : (double *)nullptr);
  }
}
static void fill_phi_ghosts(MLCC &S) {
  for (int n = S.nlev - 1; n >= 1; n--) ml_cc_restriction(S.phi[n - 1], S.phi[n], 0, 1);
  for (int n = 0; n < S.nlev; n++) phi_closure(S.phi[n], S.bct, S.bcc);
  for (int n = 1; n < S.nlev; n++) cf_interp(S.phi[n], S.phi[n - 1], S.bct, S.bcc);
}
static double composite_residual(MLCC &S) {
  hipStream_t st = ctx().stream;
  const int L = S.nlev;
  fill_phi_ghosts(S);
  HIPCHK(hipMemsetAsync(S.d_nrm, 0, sizeof(double), st));
  for (int n = 0; n < L; n++) level_residual(S, n, n == L - 1);
  // flux matching on the cells of level n-1 next to the boxes of level n: lo faces then hi faces of every direction (one update
  // per cell and launch, hence deterministic)
  for (int n = 1; n < L; n++)
    for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++)
      for (int f = 0; f < S.phi[n]->nfabs(); f++) {
        if (S.bct->ell_bc(n, f + 1, d, s, S.bcc) != VDN_BC_INT) continue;
        const vdn_box &fb = S.phi[n]->vbox[f];
        int clo[3], chi[3];
        for (int t = 0; t < 3; t++) { clo[t] = fb.lo[t] / 2; chi[t] = fb.hi[t] / 2; }
        clo[d] = chi[d] = (s ? fb.hi[d] + 1 : fb.lo[d]) / 2;
        for (int c = 0; c < S.phi[n - 1]->nfabs(); c++) {
          int blo[3], bhi[3]; Range3 r;
          for (int t = 0; t < 3; t++) { blo[t] = S.phi[n - 1]->vbox[c].lo[t]; bhi[t] = S.phi[n - 1]->vbox[c].hi[t]; }
          // the coarse cell that gets the correction must be a valid cell of box c
          if (s == 0) { blo[d] += 1; bhi[d] += 1; }
          if (!isect(clo, chi, blo, bhi, r)) continue;
          RefluxArgs A; A.d = d; A.s = s; A.dxf = S.dx[3 * n + d]; A.dxc = S.dx[3 * (n - 1) + d];
          hipLaunchKernelGGL(kk_reflux, grid_for(r), AB, 0, st, S.res[n - 1]->fabs[c], S.phi[n - 1]->fabs[c], S.beta[3 * (n - 1) + d]->fabs[c], S.mask[n - 1]->fabs[c],
                             S.phi[n]->fabs[f], S.beta[3 * n + d]->fabs[f], A, r);
        }
      }
  for (int n = L - 1; n >= 1; n--) ml_cc_restriction(S.res[n - 1], S.res[n], 0, 1);
  for (int n = 0; n < L - 1; n++)
    for (int b = 0; b < S.res[n]->nfabs(); b++) {
      Range3 r = valid_range(S.res[n], b);
      hipLaunchKernelGGL(kk_absmax_masked, reduce_grid(r), AB, 0, st, S.res[n]->fabs[b], S.mask[n]->fabs[b], 1, r, S.d_nrm);
    }
  return read_dev(S.d_nrm);
}
// nsweeps red-black sweeps of A_n e = res_n from e = 0 (homogeneous coarse-fine interface)
static void level_relax(MLCC &S, int n, int nsweeps) {
  hipStream_t st = ctx().stream;
  vdn_multifab *e = S.e[n];
  mf_setval(e, 0.0, 0, 1, true);
  const bool exchange = e->nfabs() > 1 || S.la->pmask[0] || S.la->pmask[1] || S.la->pmask[2];
  for (int s = 0; s < nsweeps; s++) for (int col = 0; col < 2; col++) {
    if (exchange) mf_fill_boundary(e);
    for (int b = 0; b < e->nfabs(); b++) {
      GsArgs A; Range3 r = valid_range(e, b);
      for (int d = 0; d < 3; d++) { A.lo[d] = r.lo[d]; A.hi[d] = r.hi[d]; A.hi2[d] = 1.0 / (S.dx[3 * n + d] * S.dx[3 * n + d]); for (int sd = 0; sd < 2; sd++) A.e[d][sd] = S.bct->ell_bc(n, b + 1, d, sd, S.bcc); }
      const int nx = r.hi[0] - r.lo[0] + 1;
      dim3 g(((nx + 1) / 2 + 63) / 64, (r.hi[1] - r.lo[1] + 4) / 4, r.hi[2] - r.lo[2] + 1);
      hipLaunchKernelGGL(kk_amr_gsrb, g, AB, 0, st, e->fabs[b], S.res[n]->fabs[b], S.beta[3 * n]->fabs[b], S.beta[3 * n + 1]->fabs[b], S.beta[3 * n + 2]->fabs[b], A, col, r);
    }
  }
}
// phi_n += e_n, and the piecewise-constant prolongation of that correction on every finer level
static void apply_correction(MLCC &S, int n) {
  hipStream_t st = ctx().stream;
  for (int b = 0; b < S.phi[n]->nfabs(); b++) { Range3 r = valid_range(S.phi[n], b); hipLaunchKernelGGL(kk_add, grid_for(r), AB, 0, st, S.phi[n]->fabs[b], S.e[n]->fabs[b], r); }
  vdn_multifab *src = S.e[n];
  for (int m = n + 1; m < S.nlev; m++) {
    const bool keep = m < S.nlev - 1;                        // a finer level still needs this level's increment
    for (int f = 0; f < S.phi[m]->nfabs(); f++) for (int c = 0; c < src->nfabs(); c++) {
      Range3 r = valid_range(S.phi[m], f); const vdn_box &cb = src->vbox[c];
      hipLaunchKernelGGL(kk_add_prolong, grid_for(r), AB, 0, st, S.phi[m]->fabs[f], keep ? S.scr[m]->fabs[f] : S.phi[m]->fabs[f], keep ? 1 : 0, src->fabs[c], r,
                         cb.lo[0], cb.lo[1], cb.lo[2], cb.hi[0], cb.hi[1], cb.hi[2]);
    }
    src = S.scr[m];
  }
}
// rh, phi: [lev];  beta: [lev*3 + d];  dx: [lev*3 + d]
int ml_cc_solve(vdn_layout *la, vdn_multifab **rh, vdn_multifab **phi, vdn_multifab **beta, const double *dx, const vdn_bc_tower *bct, int bc_comp0,
                double rel_eps, int max_iter, int *iters, double *res0, double *res) {
  require_amr(la);
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  const int L = la->nlev;
  MLCC S; S.nlev = L; S.la = la; S.rh = rh; S.phi = phi; S.beta = beta; S.dx = dx; S.bct = bct; S.bcc = bc_comp0;
  S.d_nrm = (double *)arena_alloc(256);
  for (int n = 0; n < L; n++) {
    S.res[n] = mf_temp(la, n, 1, 0, -1, true, 0.0); S.e[n] = mf_temp(la, n, 1, 1, -1, true, 0.0);
    S.scr[n] = (n >= 1 && n < L - 1) ? mf_temp(la, n, 1, 0, -1, true, 0.0) : nullptr;
    S.mask[n] = nullptr;
    if (n < L - 1) {                                         // cells of level n covered by level n+1
      S.mask[n] = mf_temp(la, n, 1, 0, -1, true, 0.0);
      for (int f = 0; f < phi[n + 1]->nfabs(); f++) for (int c = 0; c < S.mask[n]->nfabs(); c++) {
        int clo[3], chi[3]; Range3 r;
        for (int d = 0; d < 3; d++) { clo[d] = phi[n + 1]->vbox[f].lo[d] / 2; chi[d] = phi[n + 1]->vbox[f].hi[d] / 2; }
        if (isect(clo, chi, S.mask[n]->vbox[c].lo, S.mask[n]->vbox[c].hi, r)) hipLaunchKernelGGL(kk_setbox, grid_for(r), AB, 0, st, S.mask[n]->fabs[c], r, 1.0);
      }
    }
  }
  HIPCHK(hipMemsetAsync(S.d_nrm, 0, sizeof(double), st));
  for (int n = 0; n < L; n++) for (int b = 0; b < rh[n]->nfabs(); b++) {
    Range3 r = valid_range(rh[n], b);
    hipLaunchKernelGGL(kk_absmax_masked, reduce_grid(r), AB, 0, st, rh[n]->fabs[b], n < L - 1 ? S.mask[n]->fabs[b] : rh[n]->fabs[b], n < L - 1 ? 1 : 0, r, S.d_nrm);
  }
  const double bnorm = read_dev(S.d_nrm);
  const vdn_params &P = ctx().prm;
  int it = 0; bool conv = (bnorm == 0.0); double rn = 0.0;
  int ebc0[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc0[d][s] = bct->ell_bc(0, 0, d, s, bc_comp0);
  while (!conv) {
    rn = composite_residual(S);
    if (rn <= rel_eps * bnorm) { conv = true; break; }
    if (it >= max_iter) break;
    // pre-relaxation, finest level first (homogeneous interface), then the residual the next coarser level will see
    for (int n = L - 1; n >= 1; n--) {
      level_relax(S, n, P.mg_nu1);
      apply_correction(S, n);
      (void)composite_residual(S);
    }
    // coarse correction: ONE V-cycle of the single-level multigrid on the whole level 0
    mf_setval(S.e[0], 0.0, 0, 1, true);
    int cyc; double r0, rr;
    cc_solve(S.res[0], S.e[0], beta, dx, ebc0, 0.0, -1.0, -1, &cyc, &r0, &rr);
    apply_correction(S, 0);
    // post-relaxation on the new residual, coarsest level first
    for (int n = 1; n < L; n++) {
      if (n < L - 1) (void)composite_residual(S);
      else { fill_phi_ghosts(S); level_residual(S, n, false); }
      level_relax(S, n, P.mg_nu2);
      apply_correction(S, n);
    }
    it++;
  }
  fill_phi_ghosts(S);
  if (iters) *iters = it; if (res0) *res0 = bnorm; if (res) *res = rn;
  for (int n = L - 1; n >= 0; n--) { if (S.mask[n]) mf_temp_free(S.mask[n]); if (S.scr[n]) mf_temp_free(S.scr[n]); mf_temp_free(S.e[n]); mf_temp_free(S.res[n]); }
  HIPCHK(hipStreamSynchronize(st));
  arena_release(mark);
  return conv ? 0 : 1;
}

// macproject.f90:20-133 on several levels.  umac: [lev*3 + d]
void do_ml_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs, const double *dx, const vdn_bc_tower *bct, int bc_comp0) {
  require_amr(mla);
  const size_t mark = arena_mark();
  const int L = mla->nlev;
  vdn_multifab *rh[VDN_MAXLEV], *phi[VDN_MAXLEV], *beta[3 * VDN_MAXLEV];
  for (int n = 0; n < L; n++) {
    REQUIRE(rho[n]->ng >= 1, "macproject: rho needs a filled ghost cell");
    rh[n] = mf_temp(mla, n, 1, 0, -1, false, 0.0); phi[n] = mf_temp(mla, n, 1, 1, -1, true, 0.0);
    for (int d = 0; d < 3; d++) beta[3 * n + d] = mf_temp(mla, n, 1, 0, d, false, 0.0);
    mac_level_rhs(umac + 3 * n, mac_rhs[n], rh[n], dx + 3 * n);                  // divumac, macproject.f90:161-196
    mac_level_coeffs(rho[n], beta + 3 * n);                                      // mk_mac_coeffs, 296-328
  }
  for (int n = L - 1; n >= 1; n--) ml_cc_restriction(rh[n - 1], rh[n], 0, 1);     // 204-206
  for (int n = L - 1; n >= 1; n--) for (int d = 0; d < 3; d++) ml_edge_restriction(beta[3 * (n - 1) + d], beta[3 * n + d], d);       // 330-333
  int it; double r0, rr;
  int rc = ml_cc_solve(mla, rh, phi, beta, dx, bct, bc_comp0, ctx().prm.mac_rel_eps, ctx().prm.mg_max_iter, &it, &r0, &rr);
  ctx().solver_cycles[0] = it; ctx().solver_res0[0] = r0; ctx().solver_res[0] = rr;
  if (rc != 0 && ctx().prm.verbose) fprintf(stderr, "varden_amd: composite MAC solve did not converge in %d iterations (res %g / %g)\n", it, rr, r0);
  for (int n = 0; n < L; n++) mac_level_mkumac(umac + 3 * n, phi[n], beta + 3 * n, dx + 3 * n, bct, bc_comp0);   // 103
  for (int n = L - 1; n >= 1; n--) for (int d = 0; d < 3; d++) ml_edge_restriction(umac[3 * (n - 1) + d], umac[3 * n + d], d);       // 497-500
  for (int d = 0; d < 3; d++) mf_fill_boundary(umac[d]);
  for (int n = 1; n < L; n++) for (int d = 0; d < 3; d++) { ml_create_umac_grown(umac[3 * n + d], umac[3 * (n - 1) + d], d); mf_fill_boundary(umac[3 * n + d]); }   // 107-119
  for (int n = L - 1; n >= 0; n--) { for (int d = 2; d >= 0; d--) mf_temp_free(beta[3 * n + d]); mf_temp_free(phi[n]); mf_temp_free(rh[n]); }
  arena_release(mark);
}

// ---- C-ABI ------------------------------------------------------------------------------------------------------------------
extern "C" int vdn_ml_cc_restriction(vdn_multifab *crse, const vdn_multifab *fine, int icomp, int nc) { VDN_TRY ml_cc_restriction(crse, fine, icomp, nc); VDN_CATCH }
extern "C" int vdn_ml_edge_restriction(vdn_multifab *crse, const vdn_multifab *fine, int dir) { VDN_TRY ml_edge_restriction(crse, fine, dir); VDN_CATCH }
extern "C" int vdn_multifab_fill_ghost_cells(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc) { VDN_TRY ml_fill_ghost_cells(fine, crse, icomp, nc); VDN_CATCH }
extern "C" int vdn_create_umac_grown(vdn_multifab *fine, const vdn_multifab *crse, int dir) { VDN_TRY ml_create_umac_grown(fine, crse, dir); VDN_CATCH }
extern "C" int vdn_ml_restrict_and_fill(int nlev, vdn_multifab **mf, int icomp, int bcomp, int nc, int same_boundary, const vdn_bc_tower *bct) {
  VDN_TRY ml_restrict_and_fill(nlev, mf, icomp, bcomp, nc, same_boundary != 0, bct); VDN_CATCH
}
