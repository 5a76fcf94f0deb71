// pointwise.hip -- streaming kernels of the hot path: forcing terms, conservative/convective update,
// rho at half time, CFL maxima.  All HBM-bound, one thread per cell, x fastest (coalesced 512 B per
// wave-row), expression order identical to the reference Fortran (fp-contract off).
//
//   k_mkvelforce       reference src/mkforce.f90:144-236
//   k_mkscalforce      reference src/mkforce.f90:333-402
//   k_update           reference src/update.f90:186-278
//   k_make_at_halftime reference src/make_at_halftime.f90:95-115
//   k_estdt_max        reference src/estdt.f90:131-181 (the maxima; the scalar tail runs on the host)
#include "vdn_dev.h"

struct ForceArgs { int lo[3], hi[3]; double visc_coef, fac; int boussinesq, nscal; };

DEVI void mkvelforce_cell(const FV &vf, const FV &ext, const FV &gp, const FV &s, const FV &lapu, int has_lapu, const ForceArgs &A, int i, int j, int k) {
  const int out = (i < A.lo[0]) + (i > A.hi[0]) + (j < A.lo[1]) + (j > A.hi[1]) + (k < A.lo[2]) + (k > A.hi[2]);
  if (out > 1) {                                    // the six face halos only (mkforce.f90:186-234); edges and corners keep the 0 of :52
    #pragma unroll
    for (int m = 0; m < 3; m++) fv_at(vf, i, j, k, m) = 0.0;
    return;
  }
  const int ic = min(max(i, A.lo[0]), A.hi[0]), jc = min(max(j, A.lo[1]), A.hi[1]), kc = min(max(k, A.lo[2]), A.hi[2]);
  const double rho = fv_get(s, i, j, k, 0);
  #pragma unroll
  for (int m = 0; m < 3; m++) {
    double l = has_lapu ? fv_get(lapu, ic, jc, kc, m) : 0.0;   // 0th-order extrapolation of lapu
    double lapu_local = A.visc_coef * A.fac * l;
    double e = fv_get(ext, i, j, k, m);
    if (out == 0 && A.boussinesq == 1) e = fv_get(s, i, j, k, 1) * e;
    fv_at(vf, i, j, k, m) = e + (lapu_local - fv_get(gp, i, j, k, m)) / rho;
  }
}
__global__ void kk_mkvelforce(FV vf, FV ext, FV gp, FV s, FV lapu, int has_lapu, ForceArgs A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  mkvelforce_cell(vf, ext, gp, s, lapu, has_lapu, A, i, j, k);
}
// several boxes: one launch for the level (vdn_dev.h, kk_batched)
struct VelForceB { Range3 r; int g[3]; FV vf, ext, gp, s, lapu; int has_lapu; ForceArgs A;
  static __device__ double body(const VelForceB &q, int i, int j, int k, int) { mkvelforce_cell(q.vf, q.ext, q.gp, q.s, q.lapu, q.has_lapu, q.A, i, j, k); return 0.0; } };

void k_mkvelforce(vdn_multifab *vf, const vdn_multifab *ext, const vdn_multifab *s, const vdn_multifab *gp,
                  const vdn_multifab *lapu, double visc_fac) {
  Prof prof_("mkvelforce");
  if (ctx().prm.dm == 2) { k2_mkvelforce(vf, ext, s, gp, lapu, visc_fac); return; }
  REQUIRE(vf->ng >= 1 && ext->ng >= 1 && gp->ng >= 1 && s->ng >= 1, "mkvelforce: operands need a ghost cell");
  if (vf->ng != 1 || vf->nc != 3) mf_setval(vf, 0.0, 0, vf->nc, true);              // mkforce.f90:52; with one ghost layer the kernel writes every point itself
  std::vector<VelForceB> v;
  for (int i = 0; i < vf->nfabs(); i++) {
    ForceArgs A; Range3 r;
    for (int d = 0; d < 3; d++) { A.lo[d] = vf->vbox[i].lo[d]; A.hi[d] = vf->vbox[i].hi[d]; r.lo[d] = A.lo[d] - 1; r.hi[d] = A.hi[d] + 1; }
    A.visc_coef = ctx().prm.visc_coef; A.fac = visc_fac; A.boussinesq = ctx().prm.boussinesq; A.nscal = ctx().prm.nscal;
    if (vf->nfabs() == 1)
      hipLaunchKernelGGL(kk_mkvelforce, grid_for(r), dim3(64, 4, 1), 0, ctx().stream, vf->fabs[i], ext->fabs[i], gp->fabs[i],
                         s->fabs[i], lapu ? lapu->fabs[i] : vf->fabs[i], lapu ? 1 : 0, A, r);
    else { VelForceB q; q.r = r; q.vf = vf->fabs[i]; q.ext = ext->fabs[i]; q.gp = gp->fabs[i]; q.s = s->fabs[i]; q.lapu = lapu ? lapu->fabs[i] : vf->fabs[i]; q.has_lapu = lapu ? 1 : 0; q.A = A; v.push_back(q); }
  }
  launch_batched(v, 0, (double *)nullptr, 0, ctx().stream);
}

DEVI void mkscalforce_cell(const FV &sf, const FV &ext, const FV &laps, int has_laps, const ForceArgs &A, int i, int j, int k) {
  const int out = (i < A.lo[0]) + (i > A.hi[0]) + (j < A.lo[1]) + (j > A.hi[1]) + (k < A.lo[2]) + (k > A.hi[2]);
  fv_at(sf, i, j, k, 0) = 0.0;                      // density does not diffuse: its force is the 0 of setval(scal_force, 0), mkforce.f90:267
  if (out > 1) { for (int m = 1; m < A.nscal; m++) fv_at(sf, i, j, k, m) = 0.0; return; }
  const int ic = min(max(i, A.lo[0]), A.hi[0]), jc = min(max(j, A.lo[1]), A.hi[1]), kc = min(max(k, A.lo[2]), A.hi[2]);
  for (int m = 1; m < A.nscal; m++) {
    double l = has_laps ? fv_get(laps, ic, jc, kc, m) : 0.0;
    double laps_local = A.visc_coef * A.fac * l;    // here visc_coef carries diff_coef
    fv_at(sf, i, j, k, m) = fv_get(ext, i, j, k, m) + laps_local;
  }
}
__global__ void kk_mkscalforce(FV sf, FV ext, FV laps, int has_laps, ForceArgs A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  mkscalforce_cell(sf, ext, laps, has_laps, A, i, j, k);
}
struct ScalForceB { Range3 r; int g[3]; FV sf, ext, laps; int has_laps; ForceArgs A;
  static __device__ double body(const ScalForceB &q, int i, int j, int k, int) { mkscalforce_cell(q.sf, q.ext, q.laps, q.has_laps, q.A, i, j, k); return 0.0; } };

void k_mkscalforce(vdn_multifab *sf, const vdn_multifab *ext, const vdn_multifab *laps, double diff_fac) {
  Prof prof_("mkscalforce");
  if (ctx().prm.dm == 2) { k2_mkscalforce(sf, ext, laps, diff_fac); return; }
  if (sf->ng != 1 || sf->nc != ctx().prm.nscal) mf_setval(sf, 0.0, 0, sf->nc, true);              // mkforce.f90:267 / 346; with one ghost layer the kernel writes every point itself
  std::vector<ScalForceB> vb;
  for (int i = 0; i < sf->nfabs(); i++) {
    ForceArgs A; Range3 r;
    for (int d = 0; d < 3; d++) { A.lo[d] = sf->vbox[i].lo[d]; A.hi[d] = sf->vbox[i].hi[d]; r.lo[d] = A.lo[d] - 1; r.hi[d] = A.hi[d] + 1; }
    A.visc_coef = ctx().prm.diff_coef; A.fac = diff_fac; A.boussinesq = 0; A.nscal = ctx().prm.nscal;
    if (sf->nfabs() == 1)
      hipLaunchKernelGGL(kk_mkscalforce, grid_for(r), dim3(64, 4, 1), 0, ctx().stream, sf->fabs[i], ext->fabs[i],
                         laps ? laps->fabs[i] : sf->fabs[i], laps ? 1 : 0, A, r);
    else { ScalForceB q; q.r = r; q.sf = sf->fabs[i]; q.ext = ext->fabs[i]; q.laps = laps ? laps->fabs[i] : sf->fabs[i]; q.has_laps = laps ? 1 : 0; q.A = A; vb.push_back(q); }
  }
  launch_batched(vb, 0, (double *)nullptr, 0, ctx().stream);
}

// ---- update -------------------------------------------------------------------------------------
struct UpdArgs { double dx[3], dt; int ncomp; int cons[VDN_MAXCOMP]; };

DEVI void update_cell(const FV &sold, const FV &snew, const FV &um, const FV &vm, const FV &wm, const FV &sx, const FV &sy, const FV &sz,
                      const FV &fx, const FV &fy, const FV &fz, const FV &force, const UpdArgs &A, int i, int j, int k) {
  const double ubar = 0.5 * (fv_get(um, i, j, k) + fv_get(um, i + 1, j, k));
  const double vbar = 0.5 * (fv_get(vm, i, j, k) + fv_get(vm, i, j + 1, k));
  const double wbar = 0.5 * (fv_get(wm, i, j, k) + fv_get(wm, i, j, k + 1));
  for (int c = 0; c < A.ncomp; c++) {
    double so = fv_get(sold, i, j, k, c), f = fv_get(force, i, j, k, c), v;
    if (A.cons[c]) {                                 // update.f90:250-253
      double divsu = (fv_get(fx, i + 1, j, k, c) - fv_get(fx, i, j, k, c)) / A.dx[0]
                   + (fv_get(fy, i, j + 1, k, c) - fv_get(fy, i, j, k, c)) / A.dx[1]
                   + (fv_get(fz, i, j, k + 1, c) - fv_get(fz, i, j, k, c)) / A.dx[2];
      v = so - A.dt * divsu + A.dt * f;
    } else {                                         // update.f90:220-238, 263-269
      double ugrads = ubar * (fv_get(sx, i + 1, j, k, c) - fv_get(sx, i, j, k, c)) / A.dx[0]
                    + vbar * (fv_get(sy, i, j + 1, k, c) - fv_get(sy, i, j, k, c)) / A.dx[1]
                    + wbar * (fv_get(sz, i, j, k + 1, c) - fv_get(sz, i, j, k, c)) / A.dx[2];
      v = so - A.dt * ugrads + A.dt * f;
    }
    fv_at(snew, i, j, k, c) = v;
  }
}
__global__ void kk_update(FV sold, FV snew, FV um, FV vm, FV wm, FV sx, FV sy, FV sz, FV fx, FV fy, FV fz, FV force,
                          UpdArgs A, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  update_cell(sold, snew, um, vm, wm, sx, sy, sz, fx, fy, fz, force, A, i, j, k);
}
struct UpdateB { Range3 r; int g[3]; FV sold, snew, um, vm, wm, sx, sy, sz, fx, fy, fz, force; UpdArgs A;
  static __device__ double body(const UpdateB &q, int i, int j, int k, int) { update_cell(q.sold, q.snew, q.um, q.vm, q.wm, q.sx, q.sy, q.sz, q.fx, q.fy, q.fz, q.force, q.A, i, j, k); return 0.0; } };

void k_update(const vdn_multifab *sold, vdn_multifab **umac, vdn_multifab **sedge, vdn_multifab **flux,
              const vdn_multifab *force, vdn_multifab *snew, const double *dx, double dt, bool is_vel, const int *is_cons) {
  Prof prof_("update");
  if (ctx().prm.dm == 2) { k2_update(sold, umac, sedge, flux, force, snew, dx, dt, is_vel, is_cons); return; }
  std::vector<UpdateB> vb;
  for (int i = 0; i < sold->nfabs(); i++) {
    UpdArgs A; Range3 r;
    for (int d = 0; d < 3; d++) { A.dx[d] = dx[d]; r.lo[d] = sold->vbox[i].lo[d]; r.hi[d] = sold->vbox[i].hi[d]; }
    A.dt = dt; A.ncomp = sold->nc;
    for (int c = 0; c < sold->nc; c++) A.cons[c] = (!is_vel && is_cons[c]) ? 1 : 0;
    if (sold->nfabs() == 1)
      hipLaunchKernelGGL(kk_update, grid_for(r), dim3(64, 4, 1), 0, ctx().stream, sold->fabs[i], snew->fabs[i],
                         umac[0]->fabs[i], umac[1]->fabs[i], umac[2]->fabs[i], sedge[0]->fabs[i], sedge[1]->fabs[i],
                         sedge[2]->fabs[i], flux[0]->fabs[i], flux[1]->fabs[i], flux[2]->fabs[i], force->fabs[i], A, r);
    else { UpdateB q; q.r = r; q.sold = sold->fabs[i]; q.snew = snew->fabs[i]; q.um = umac[0]->fabs[i]; q.vm = umac[1]->fabs[i]; q.wm = umac[2]->fabs[i];
      q.sx = sedge[0]->fabs[i]; q.sy = sedge[1]->fabs[i]; q.sz = sedge[2]->fabs[i]; q.fx = flux[0]->fabs[i]; q.fy = flux[1]->fabs[i]; q.fz = flux[2]->fabs[i]; q.force = force->fabs[i]; q.A = A; vb.push_back(q); }
  }
  launch_batched(vb, 0, (double *)nullptr, 0, ctx().stream);
}

// The velocity update with its forcing term formed in place (one level): velocity_advance.f90:78-83 builds vel_force = mkvelforce(ext, gp,
// rhohalf, lapu, visc_fac = 0) on the grown box, fills its ghost cells, and update_3d then reads it on the valid cells only -- a pass over ten
// fields (0.32 ms at 256^3) for a value each cell can form from seven loads.  mkvelforce_cell's expression for a valid cell, update_cell's
// for a non-conservative component; same bits.
struct update_vf_K { FV uold, unew, um, vm, wm, sx, sy, sz, ext, gp, s, lapu; int has_lapu; ForceArgs F; UpdArgs A;
  __device__ void cell(int i, int j, int k) const {
    const double ubar = 0.5 * (fv_get(um, i, j, k) + fv_get(um, i + 1, j, k));
    const double vbar = 0.5 * (fv_get(vm, i, j, k) + fv_get(vm, i, j + 1, k));
    const double wbar = 0.5 * (fv_get(wm, i, j, k) + fv_get(wm, i, j, k + 1));
    const double rho = fv_get(s, i, j, k, 0);
    #pragma unroll
    for (int c = 0; c < 3; c++) {
      double l = has_lapu ? fv_get(lapu, i, j, k, c) : 0.0;
      double lapu_local = F.visc_coef * F.fac * l;
      double e = fv_get(ext, i, j, k, c);
      if (F.boussinesq == 1) e = fv_get(s, i, j, k, 1) * e;
      const double f = e + (lapu_local - fv_get(gp, i, j, k, c)) / rho;
      const double so = fv_get(uold, i, j, k, c);
      double ugrads = ubar * (fv_get(sx, i + 1, j, k, c) - fv_get(sx, i, j, k, c)) / A.dx[0]
                    + vbar * (fv_get(sy, i, j + 1, k, c) - fv_get(sy, i, j, k, c)) / A.dx[1]
                    + wbar * (fv_get(sz, i, j, k + 1, c) - fv_get(sz, i, j, k, c)) / A.dx[2];
      fv_at(unew, i, j, k, c) = so - A.dt * ugrads + A.dt * f;
    }
  } };
void k_update_velforce(const vdn_multifab *uold, vdn_multifab **umac, vdn_multifab **uedge, const vdn_multifab *ext, const vdn_multifab *s,
                       const vdn_multifab *gp, const vdn_multifab *lapu, double visc_fac, vdn_multifab *unew, const double *dx, double dt) {
  Prof prof_("update");
  REQUIRE(ctx().prm.dm == 3 && uold->nc == 3, "k_update_velforce: three dimensions, three components");
  std::vector<std::pair<update_vf_K, Range3>> v;
  for (int i = 0; i < uold->nfabs(); i++) {
    update_vf_K K; Range3 r;
    for (int d = 0; d < 3; d++) { K.A.dx[d] = dx[d]; r.lo[d] = K.F.lo[d] = uold->vbox[i].lo[d]; r.hi[d] = K.F.hi[d] = uold->vbox[i].hi[d]; }
    K.A.dt = dt; K.A.ncomp = 3; for (int c = 0; c < VDN_MAXCOMP; c++) K.A.cons[c] = 0;
    K.F.visc_coef = ctx().prm.visc_coef; K.F.fac = visc_fac; K.F.boussinesq = ctx().prm.boussinesq; K.F.nscal = ctx().prm.nscal;
    K.uold = uold->fabs[i]; K.unew = unew->fabs[i]; K.um = umac[0]->fabs[i]; K.vm = umac[1]->fabs[i]; K.wm = umac[2]->fabs[i];
    K.sx = uedge[0]->fabs[i]; K.sy = uedge[1]->fabs[i]; K.sz = uedge[2]->fabs[i];
    K.ext = ext->fabs[i]; K.gp = gp->fabs[i]; K.s = s->fabs[i]; K.lapu = lapu ? lapu->fabs[i] : uold->fabs[i]; K.has_lapu = lapu ? 1 : 0;
    v.push_back({ K, r });
  }
  launch_cells(v, ctx().stream);
}

// ---- rho at half time -----------------------------------------------------------------------------
__global__ void kk_halftime(FV rh, int oc, FV so, FV sn, int ic, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  fv_at(rh, i, j, k, oc) = 0.5 * (fv_get(so, i, j, k, ic) + fv_get(sn, i, j, k, ic));
}
struct HalftimeB { Range3 r; int g[3]; FV rh, so, sn; int oc, ic;
  static __device__ double body(const HalftimeB &q, int i, int j, int k, int) { fv_at(q.rh, i, j, k, q.oc) = 0.5 * (fv_get(q.so, i, j, k, q.ic) + fv_get(q.sn, i, j, k, q.ic)); return 0.0; } };
void k_make_at_halftime(vdn_multifab *rhohalf, const vdn_multifab *sold, const vdn_multifab *snew, int in_comp, int out_comp) {
  std::vector<HalftimeB> vb;
  for (int i = 0; i < rhohalf->nfabs(); i++) {
    Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = rhohalf->vbox[i].lo[d]; r.hi[d] = rhohalf->vbox[i].hi[d]; }
    if (rhohalf->nfabs() == 1) hipLaunchKernelGGL(kk_halftime, grid_for(r), dim3(64, 4, 1), 0, ctx().stream, rhohalf->fabs[i], out_comp, sold->fabs[i], snew->fabs[i], in_comp, r);
    else { HalftimeB q; q.r = r; q.rh = rhohalf->fabs[i]; q.so = sold->fabs[i]; q.sn = snew->fabs[i]; q.oc = out_comp; q.ic = in_comp; vb.push_back(q); }
  }
  launch_batched(vb, 0, (double *)nullptr, 0, ctx().stream);
}

// ---- estdt maxima: wave-level reduction (64 lanes, shuffles) + one atomic per wave ----------------
__global__ void kk_estdt(FV u, FV s, FV gp, FV ext, Range3 r, double *out6) {
  REDUCE_IJ(r)
  double m[6] = { 0, 0, 0, 0, 0, 0 };
  if (in_ij) REDUCE_KLOOP(r) {
    const double rho = fv_get(s, i, j, k, 0);
    #pragma unroll
    for (int c = 0; c < 3; c++) {
      m[c] = nmax(m[c], fabs(fv_get(u, i, j, k, c)));
      m[3 + c] = nmax(m[3 + c], fabs(fv_get(gp, i, j, k, c) / rho - fv_get(ext, i, j, k, c)));
    }
  }
  #pragma unroll
  for (int c = 0; c < 6; c++) block_atomic_max(out6 + c, m[c]);
}
// the same for all boxes of a level in one launch
struct EstB { Range3 r; int g[3]; FV u, s, gp, ext; };
__global__ void __launch_bounds__(256) kk_estdt_b(const EstB *args, const int *start, int nbox, double *out6) {
  int lo = 0, hi = nbox - 1;
  const int bid = (int)blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (as_constant(start + mid) <= bid) lo = mid; else hi = mid - 1; }
  const EstB &a = as_constant(args + lo);
  const int lb = bid - as_constant(start + lo);
  const int bx = lb % a.g[0], by = (lb / a.g[0]) % a.g[1], bz = lb / (a.g[0] * a.g[1]);
  const int i = a.r.lo[0] + bx * 64 + (int)threadIdx.x, j = a.r.lo[1] + by * 4 + (int)threadIdx.y;
  double m[6] = { 0, 0, 0, 0, 0, 0 };
  if (i <= a.r.hi[0] && j <= a.r.hi[1]) for (int k = a.r.lo[2] + bz; k <= a.r.hi[2]; k += a.g[2]) {
    const double rho = fv_get(a.s, i, j, k, 0);
    #pragma unroll
    for (int c = 0; c < 3; c++) {
      m[c] = nmax(m[c], fabs(fv_get(a.u, i, j, k, c)));
      m[3 + c] = nmax(m[3 + c], fabs(fv_get(a.gp, i, j, k, c) / rho - fv_get(a.ext, i, j, k, c)));
    }
  }
  #pragma unroll
  for (int c = 0; c < 6; c++) block_atomic_max(out6 + c, m[c]);
}
void k_estdt_max(const vdn_multifab *u, const vdn_multifab *s, const vdn_multifab *gp, const vdn_multifab *ext, double out6[6]) {
  Prof prof_("estdt");
  if (ctx().prm.dm == 2) { k2_estdt_max(u, s, gp, ext, out6); return; }
  VdnCtx &c = ctx();
  HIPCHK(hipMemsetAsync(c.d_scal, 0, 6 * sizeof(double), c.stream));
  if (u->nfabs() > 1) {
    const int nb = u->nfabs();
    std::vector<EstB> v(nb); std::vector<int> start(nb); int tot = 0;
    for (int i = 0; i < nb; i++) {
      EstB &a = v[i];
      for (int d = 0; d < 3; d++) { a.r.lo[d] = u->vbox[i].lo[d]; a.r.hi[d] = u->vbox[i].hi[d]; }
      const dim3 g = reduce_grid(a.r);
      a.g[0] = g.x; a.g[1] = g.y; a.g[2] = g.z; a.u = u->fabs[i]; a.s = s->fabs[i]; a.gp = gp->fabs[i]; a.ext = ext->fabs[i];
      start[i] = tot; tot += (int)(g.x * g.y * g.z);
    }
    EstB *d_args = (EstB *)desc_scratch(sizeof(EstB) * nb); int *d_start = (int *)desc_scratch(sizeof(int) * nb);
    upload_staged(d_args, v.data(), sizeof(EstB) * nb); upload_staged(d_start, start.data(), sizeof(int) * nb);
    hipLaunchKernelGGL(kk_estdt_b, dim3(tot), dim3(64, 4, 1), 0, c.stream, (const EstB *)d_args, (const int *)d_start, nb, c.d_scal);
  } else
  for (int i = 0; i < u->nfabs(); i++) {
    Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = u->vbox[i].lo[d]; r.hi[d] = u->vbox[i].hi[d]; }
    hipLaunchKernelGGL(kk_estdt, reduce_grid(r), dim3(64, 4, 1), 0, c.stream, u->fabs[i], s->fabs[i], gp->fabs[i], ext->fabs[i], r, c.d_scal);
  }
  comm_allreduce_max_dev(c.d_scal, 6);        // MAX of the maxima == the reference's MIN over ranks of dt_proc (estdt.f90:69)
  const double *h = read_scalars(c.d_scal, 6);
  for (int k = 0; k < 6; k++) out6[k] = h[k];
}

// ====================================================================================================
// derived plot quantities (makevort.f90): vorticity and velocity magnitude of write_plotfile (varden.f90:532-540)
// ====================================================================================================
struct VortArgs { int lo[3], hi[3]; int phys[3][2]; double dx[3]; int comp, dm; };
// d(u_c)/dx_D: centred (uycen & co., makevort.f90:568-572), one-sided next to an inflow / no-slip face (uylo / uyhi, :574-584)
template <int D> DEVI double vort_der(const FV &u, int c, int side, int i, int j, int k, double dxd) {
  const double up = fv_get(u, i + (D == 0), j + (D == 1), k + (D == 2), c), u0 = fv_get(u, i, j, k, c), um = fv_get(u, i - (D == 0), j - (D == 1), k - (D == 2), c);
  if (side < 0) return (up + 3.0 * u0 - 4.0 * um) / (3.0 * dxd);
  if (side > 0) return -(um + 3.0 * u0 - 4.0 * up) / (3.0 * dxd);
  return 0.5 * (up - um) / dxd;
}
DEVI bool vort_fix3(int p) { return p == VDN_INLET || p == VDN_NO_SLIP_WALL; }                           // makevort.f90:188-195
DEVI bool vort_fix2(int p) { return p == VDN_INLET || p == VDN_SLIP_WALL || p == VDN_NO_SLIP_WALL; }     // makevort.f90:116-117
struct vort_K { FV vort, u; VortArgs A;
  __device__ void cell(int i, int j, int k) const {
    if (A.dm == 2) {
      // makevort_2d (makevort.f90:93-156): one-sided forms over dx (not 3 dx), slip walls included; the y-face loops run last and
      // overwrite, so at a corner the x derivative is the centred one
      double vx = (fv_get(u, i + 1, j, 0, 1) - fv_get(u, i - 1, j, 0, 1)) / (2.0 * A.dx[0]);
      double uy = (fv_get(u, i, j + 1, 0, 0) - fv_get(u, i, j - 1, 0, 0)) / (2.0 * A.dx[1]);
      const bool ylo = j == A.lo[1] && vort_fix2(A.phys[1][0]), yhi = j == A.hi[1] && vort_fix2(A.phys[1][1]);
      if (!(ylo || yhi)) {
        if (i == A.lo[0] && vort_fix2(A.phys[0][0])) vx = (fv_get(u, i + 1, j, 0, 1) + 3.0 * fv_get(u, i, j, 0, 1) - 4.0 * fv_get(u, i - 1, j, 0, 1)) / A.dx[0];
        if (i == A.hi[0] && vort_fix2(A.phys[0][1])) vx = -(fv_get(u, i - 1, j, 0, 1) + 3.0 * fv_get(u, i, j, 0, 1) - 4.0 * fv_get(u, i + 1, j, 0, 1)) / A.dx[0];
      }
      if (ylo) uy = (fv_get(u, i, j + 1, 0, 0) + 3.0 * fv_get(u, i, j, 0, 0) - 4.0 * fv_get(u, i, j - 1, 0, 0)) / A.dx[1];
      if (yhi) uy = -(fv_get(u, i, j - 1, 0, 0) + 3.0 * fv_get(u, i, j, 0, 0) - 4.0 * fv_get(u, i, j + 1, 0, 0)) / A.dx[1];
      fv_at(vort, i, j, 0, A.comp) = vx - uy;
      return;
    }
    // makevort_3d (makevort.f90:158-682): faces, edges and corners follow one rule per direction
    const int q[3] = { i, j, k };
    int side[3];
    #pragma unroll
    for (int d = 0; d < 3; d++) {
      side[d] = 0;
      if (q[d] == A.lo[d] && vort_fix3(A.phys[d][0])) side[d] = -1;
      if (q[d] == A.hi[d] && vort_fix3(A.phys[d][1])) side[d] = 1;
    }
    const double uy = vort_der<1>(u, 0, side[1], i, j, k, A.dx[1]), uz = vort_der<2>(u, 0, side[2], i, j, k, A.dx[2]);
    const double vx = vort_der<0>(u, 1, side[0], i, j, k, A.dx[0]), vz = vort_der<2>(u, 1, side[2], i, j, k, A.dx[2]);
    const double wx = vort_der<0>(u, 2, side[0], i, j, k, A.dx[0]), wy = vort_der<1>(u, 2, side[1], i, j, k, A.dx[1]);
    fv_at(vort, i, j, k, A.comp) = sqrt((wy - vz) * (wy - vz) + (uz - wx) * (uz - wx) + (vx - uy) * (vx - uy));     // vorfun, :676-680
  } };
struct magvel_K { FV mv, u; int comp, dm;
  __device__ void cell(int i, int j, int k) const {                                                           // makevort.f90:684-724
    double s = fv_get(u, i, j, k, 0) * fv_get(u, i, j, k, 0) + fv_get(u, i, j, k, 1) * fv_get(u, i, j, k, 1);
    if (dm == 3) s = s + fv_get(u, i, j, k, 2) * fv_get(u, i, j, k, 2);
    fv_at(mv, i, j, k, comp) = sqrt(s);
  } };
// make_vorticity(vort, comp, u, dx, bc): fills the ghost cells of u (fill_boundary + physbc, makevort.f90:34-38) and writes component comp
void k_make_vorticity(vdn_multifab *vort, int comp, vdn_multifab *u, const double *dx, const vdn_bc_tower *bct) {
  const int dm = ctx().prm.dm;
  REQUIRE(u->ng >= 1 && u->nc >= dm && comp >= 0 && comp < vort->nc && vort->nfabs() == u->nfabs(), "make_vorticity: operand shapes");
  mf_fill_boundary(u);
  mf_physbc(u, 0, 0, dm, bct, false);
  std::vector<std::pair<vort_K, Range3>> v;
  for (int i = 0; i < u->nfabs(); i++) {
    VortArgs A; Range3 r; BoxP bp = make_boxp(u, i, bct);
    for (int d = 0; d < 3; d++) { A.lo[d] = r.lo[d] = bp.lo[d]; A.hi[d] = r.hi[d] = bp.hi[d]; A.dx[d] = d < dm ? dx[d] : 1.0;
      for (int s = 0; s < 2; s++) A.phys[d][s] = bp.phys[d][s]; }
    A.comp = comp; A.dm = dm;
    v.push_back({ vort_K{ vort->fabs[i], u->fabs[i], A }, r });
  }
  launch_cells(v, ctx().stream);
}
void k_make_magvel(vdn_multifab *mv, int comp, vdn_multifab *u) {
  const int dm = ctx().prm.dm;
  REQUIRE(u->nc >= dm && comp >= 0 && comp < mv->nc && mv->nfabs() == u->nfabs(), "make_magvel: operand shapes");
  mf_fill_boundary(u);                                                                                         // makevort.f90:74
  std::vector<std::pair<magvel_K, Range3>> v;
  for (int i = 0; i < u->nfabs(); i++) {
    Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = u->vbox[i].lo[d]; r.hi[d] = u->vbox[i].hi[d]; }
    v.push_back({ magvel_K{ mv->fabs[i], u->fabs[i], comp, dm }, r });
  }
  launch_cells(v, ctx().stream);
}
extern "C" int vdn_make_vorticity(vdn_multifab *vort, int comp, vdn_multifab *u, const double *dx, const vdn_bc_tower *bct) {
  VDN_TRY REQUIRE(ctx().inited, "vdn_init has not been called"); k_make_vorticity(vort, comp, u, dx, bct); VDN_CATCH
}
extern "C" int vdn_make_magvel(vdn_multifab *magvel, int comp, vdn_multifab *u) {
  VDN_TRY REQUIRE(ctx().inited, "vdn_init has not been called"); k_make_magvel(magvel, comp, u); VDN_CATCH
}
