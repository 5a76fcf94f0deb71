// godunov.hip -- unsplit Godunov predictors on CDNA4: limited slopes, MAC-velocity prediction
// (velpred) and edge states / fluxes (mkflux).
//
// Reference arithmetic restated (expression order kept; built with -ffp-contract=off):
//   slopes   src/slope.f90:148-588
//   velpred  src/velpred.f90:1776-2765  (velpred_3d, the default rolling kernel, incl. its hi-x
//            OUTLET min() at :2075)
//   mkflux   src/mkflux.f90:1186-2567   (mkflux_3d)
//
// MI355X mapping (round 1): the reference marches k-planes serially with ~30 live 2-D planes.  Here
// every stage of the data-flow DAG (A slopes, B normal predictor, C transverse
// states, D edge states) is one kernel over the whole box, one thread per cell handling that cell's
// three lower faces, x fastest so each wave row is a 512-byte coalesced segment; the one-dimensional
// predictor pairs (s_L, s_R) are recomputed where needed instead of being stored (flops are free, HBM
// is not), and only the upwinded stage results (3 + 6 fields per component) travel through HBM.
// The per-box dead-band eps needs a max-reduction: wave shuffles + one atomic per wave, consumed from
// device memory by the next kernel (no host round trip).
#include "vdn_dev.h"

// ---------------------------------------------------------------------------------------------------
struct GArgs {
  int lo[3], hi[3];
  int phys[3][2];
  int adv[3][2][3];          // adv bc of the (up to 3) advected components, for the slope specials
  double dx[3], dt;
  int ncomp, is_vel, use_minion, slope_order;
  int cons[3];
};

template <int D> DEVI double ld(const FV &f, int i, int j, int k, int off, int c = 0) {
  return fv_get(f, i + (D == 0 ? off : 0), j + (D == 1 ? off : 0), k + (D == 2 ? off : 0), c);
}
template <int D> DEVI int coord(int i, int j, int k) { return D == 0 ? i : (D == 1 ? j : k); }
DEVI double sgn1(double x) { return copysign(1.0, x); }

// ---- stage A: slopes --------------------------------------------------------------------------------
struct Fromm { double cen, lim, flag, fromm; };
DEVI Fromm fromm_of(double sm, double s0, double sp) {     // slope.f90:226-235
  Fromm f;
  f.cen = 0.5 * (sp - sm);
  double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
  double l = fmin(fabs(dmin), fabs(dpls));
  f.lim = (dpls * dmin > 0.0) ? l : 0.0;
  f.flag = sgn1(f.cen);
  f.fromm = f.flag * fmin(f.lim, fabs(f.cen));
  return f;
}
DEVI double limited(double del, double sm, double s0, double sp) {   // the one-sided boundary slope limiter
  double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
  double slim = fmin(fabs(dpls), fabs(dmin));
  slim = (dpls * dmin > 0.0) ? slim : 0.0;
  return sgn1(del) * fmin(slim, fabs(del));
}

// slope of component c along D at cell (i,j,k); needs s at offsets -2..+2 along D
template <int D> DEVI double slope_at(const FV &s, int c, int i, int j, int k, int is, int ie, bool lo_sp, bool hi_sp, int order) {
  if (order == 0) return 0.0;
  const int pos = coord<D>(i, j, k);
  const double two3rd = 2.0 / 3.0, sixth = 1.0 / 6.0, third = 1.0 / 3.0, tenth = 0.1;
  const double m2 = ld<D>(s, i, j, k, -2, c), m1 = ld<D>(s, i, j, k, -1, c), s0 = ld<D>(s, i, j, k, 0, c),
               p1 = ld<D>(s, i, j, k, 1, c), p2 = ld<D>(s, i, j, k, 2, c);
  if (lo_sp && pos == is - 1) return 0.0;
  if (hi_sp && pos == ie + 1) return 0.0;
  if (order == 2) {
    if (lo_sp && pos == is) return limited((p1 + 3.0 * s0 - 4.0 * m1) * third, m1, s0, p1);      // slope.f90:192-199
    if (hi_sp && pos == ie) return limited(-(m1 + 3.0 * s0 - 4.0 * p1) * third, m1, s0, p1);     // slope.f90:205-213
    return limited(0.5 * (p1 - m1), m1, s0, p1);                                                  // 181-187
  }
  // 4th order
  if (lo_sp && pos == is) return limited(-16.0 / 15.0 * m1 + 0.5 * s0 + two3rd * p1 - tenth * p2, m1, s0, p1);       // 247-254
  if (hi_sp && pos == ie) return limited(-(-16.0 / 15.0 * p1 + 0.5 * s0 + two3rd * m1 - tenth * m2), m1, s0, p1);    // 268-275
  Fromm f0 = fromm_of(m1, s0, p1);
  double fp, fm;
  if (hi_sp && pos == ie - 1)       // revised fromm(ie) = slope(ie)   (slope.f90:277-281)
    fp = limited(-(-16.0 / 15.0 * p2 + 0.5 * p1 + two3rd * s0 - tenth * m1), s0, p1, p2);
  else {
    // fromm(i+1) needs s(i+3)?  no: fromm(i+1) uses s(i), s(i+1), s(i+2)
    fp = fromm_of(s0, p1, p2).fromm;
  }
  if (lo_sp && pos == is + 1)       // revised fromm(is) = slope(is)   (slope.f90:256-260)
    fm = limited(-16.0 / 15.0 * m2 + 0.5 * m1 + two3rd * s0 - tenth * p1, m2, m1, s0);
  else
    fm = fromm_of(m2, m1, s0).fromm;
  double ds = 2.0 * two3rd * f0.cen - sixth * (fp + fm);
  return f0.flag * fmin(fabs(ds), f0.lim);
}

__global__ void __launch_bounds__(256) kk_slopes(FV s, FV sl0, FV sl1, FV sl2, GArgs A, Range3 r, int dirmask) {
  THREAD_IJK(r)
  if (!in_range) return;
  for (int c = 0; c < A.ncomp; c++) {
    #define SPEC(d, sd) (A.adv[d][sd][c] == VDN_EXT_DIR || A.adv[d][sd][c] == VDN_HOEXTRAP)
    if (dirmask & 1) fv_at(sl0, i, j, k, c) = slope_at<0>(s, c, i, j, k, A.lo[0], A.hi[0], SPEC(0, 0), SPEC(0, 1), A.slope_order);
    if (dirmask & 2) fv_at(sl1, i, j, k, c) = slope_at<1>(s, c, i, j, k, A.lo[1], A.hi[1], SPEC(1, 0), SPEC(1, 1), A.slope_order);
    if (dirmask & 4) fv_at(sl2, i, j, k, c) = slope_at<2>(s, c, i, j, k, A.lo[2], A.hi[2], SPEC(2, 0), SPEC(2, 1), A.slope_order);
    #undef SPEC
  }
}

// ---- boundary rule for a (left,right) pair on a domain face (velpred.f90:2044-2079, 2200-2224;
//      mkflux.f90:1463-1515, ...) ------------------------------------------------------------------------
DEVI void bc_pair(double &L, double &R, int phys, int side, bool is_vel, bool normal, double ghost, bool quirk) {
  if (phys == VDN_INLET) { L = ghost; R = ghost; }
  else if (phys == VDN_SLIP_WALL) {
    if (is_vel && normal) { L = 0.0; R = 0.0; }
    else if (side == 0) L = R; else R = L;
  } else if (phys == VDN_NO_SLIP_WALL) {
    if (is_vel) { L = 0.0; R = 0.0; }
    else if (side == 0) L = R; else R = L;
  } else if (phys == VDN_OUTLET) {
    if (is_vel && normal) {
      double v;
      if (side == 0) v = fmin(R, 0.0);
      else v = quirk ? fmin(L, 0.0) : fmax(L, 0.0);
      L = v; R = v;
    } else if (side == 0) L = R; else R = L;
  }
}
template <int D> DEVI int face_side(const GArgs &A, int i, int j, int k) {
  const int f = coord<D>(i, j, k);
  return f == A.lo[D] ? 0 : (f == A.hi[D] + 1 ? 1 : -1);
}
DEVI double eps_from(const double *umax_p) { double um = *umax_p; return (um == 0.0) ? 1.0e-8 : 1.0e-8 * um; }

// ====================================================================================================
// mkflux
// ====================================================================================================
DEVI double upwind_mac(double L, double R, double um, double eps) {   // mkflux.f90:1520-1522
  double v = (um > 0.0) ? L : R;
  double savg = 0.5 * (L + R);
  return (fabs(um) > eps) ? v : savg;
}

// one-dimensional predictor pair of component c on the lower D-face of cell (i,j,k), after the bc
// (mkflux.f90:1440-1515 x, 1527-1602 y, 1779-1858 z)
template <int D> DEVI void mk_pair(const GArgs &A, const FV &s, const FV &slp, const FV &mac, const FV &force, const FV &macrhs,
                                   int c, int i, int j, int k, double &L, double &R) {
  const double um = fv_get(mac, i, j, k);
  const double dt2 = 0.5 * A.dt;
  const double sl = ld<D>(s, i, j, k, -1, c), sr = fv_get(s, i, j, k, c);
  L = sl + (0.5 - dt2 * um / A.dx[D]) * ld<D>(slp, i, j, k, -1, c);
  R = sr - (0.5 + dt2 * um / A.dx[D]) * fv_get(slp, i, j, k, c);
  if (A.use_minion) {
    L = L + dt2 * ld<D>(force, i, j, k, -1, c); R = R + dt2 * fv_get(force, i, j, k, c);
    if (A.cons[c]) { L = L - dt2 * sl * ld<D>(macrhs, i, j, k, -1); R = R - dt2 * sr * fv_get(macrhs, i, j, k); }
  }
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) bc_pair(L, R, A.phys[D][side], side, A.is_vel != 0, c == D, side == 0 ? sl : sr, false);
}

// stage B: simh_D on the lower faces of cell (i,j,k);  SI has 3*ncomp comps: [D*ncomp + c]
__global__ void __launch_bounds__(256) kk_mk_B(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SI, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  for (int c = 0; c < A.ncomp; c++) {
    double L, R;
    if (i >= A.lo[0]) { mk_pair<0>(A, s, sl0, um, force, macrhs, c, i, j, k, L, R); fv_at(SI, i, j, k, 0 * A.ncomp + c) = upwind_mac(L, R, fv_get(um, i, j, k), eps); }
    if (j >= A.lo[1]) { mk_pair<1>(A, s, sl1, vm, force, macrhs, c, i, j, k, L, R); fv_at(SI, i, j, k, 1 * A.ncomp + c) = upwind_mac(L, R, fv_get(vm, i, j, k), eps); }
    if (k >= A.lo[2]) { mk_pair<2>(A, s, sl2, wm, force, macrhs, c, i, j, k, L, R); fv_at(SI, i, j, k, 2 * A.ncomp + c) = upwind_mac(L, R, fv_get(wm, i, j, k), eps); }
  }
}

// transverse correction of a state extrapolated from cell (ci,cj,ck) by direction T
// (mkflux.f90:1620-1626 etc.):  conservative  (dt3/hT)(simhT(+)*macT(+) - simhT*macT)
//                               convective    (dt6/hT)(macT(+)+macT)(simhT(+)-simhT)
template <int T> DEVI double trans_term(const GArgs &A, const FV &SI, int comp_idx, const FV &macT, bool cons,
                                        int ci, int cj, int ck, double fcons, double fconv) {
  const double sp = ld<T>(SI, ci, cj, ck, 1, comp_idx), s0 = fv_get(SI, ci, cj, ck, comp_idx);
  const double mp = ld<T>(macT, ci, cj, ck, 1), m0 = fv_get(macT, ci, cj, ck);
  if (cons) return (fcons / A.dx[T]) * (sp * mp - s0 * m0);
  return (fconv / A.dx[T]) * (mp + m0) * (sp - s0);
}

// stage C for one (D,T): SC[(D,T)] on the lower D-face of cell (i,j,k)
template <int D, int T> DEVI void mk_C_one(const GArgs &A, const FV &s, const FV &slp, const FV &macD, const FV &macT, const FV &force,
                                           const FV &macrhs, const FV &SI, const FV &SC, int c, int i, int j, int k, double eps) {
  constexpr int O = 3 - D - T;
  const int qd = coord<D>(i, j, k), qt = coord<T>(i, j, k);
  if (qd < A.lo[D] || qt < A.lo[T] || qt > A.hi[T]) return;       // normal faces lo..hi+1, T valid, O grown
  (void)O;
  double L, R;
  mk_pair<D>(A, s, slp, macD, force, macrhs, c, i, j, k, L, R);
  const double dt3 = A.dt / 3.0, dt6 = A.dt / 6.0;
  const int ci = i - (D == 0), cj = j - (D == 1), ck = k - (D == 2);
  L = L - trans_term<T>(A, SI, T * A.ncomp + c, macT, A.cons[c] != 0, ci, cj, ck, dt3, dt6);
  R = R - trans_term<T>(A, SI, T * A.ncomp + c, macT, A.cons[c] != 0, i, j, k, dt3, dt6);
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) bc_pair(L, R, A.phys[D][side], side, A.is_vel != 0, c == D, side == 0 ? ld<D>(s, i, j, k, -1, c) : fv_get(s, i, j, k, c), false);
  // SC component index: (D*2 + (T > D ? T-1 : T)) * ncomp + c
  fv_at(SC, i, j, k, (D * 2 + (T > D ? T - 1 : T)) * A.ncomp + c) = upwind_mac(L, R, fv_get(macD, i, j, k), eps);
}
DEVI int sc_idx(int D, int T, int ncomp, int c) { return (D * 2 + (T > D ? T - 1 : T)) * ncomp + c; }

__global__ void __launch_bounds__(256) kk_mk_C(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SI, FV SC, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  for (int c = 0; c < A.ncomp; c++) {
    mk_C_one<0, 1>(A, s, sl0, um, vm, force, macrhs, SI, SC, c, i, j, k, eps);
    mk_C_one<0, 2>(A, s, sl0, um, wm, force, macrhs, SI, SC, c, i, j, k, eps);
    mk_C_one<1, 0>(A, s, sl1, vm, um, force, macrhs, SI, SC, c, i, j, k, eps);
    mk_C_one<1, 2>(A, s, sl1, vm, wm, force, macrhs, SI, SC, c, i, j, k, eps);
    mk_C_one<2, 0>(A, s, sl2, wm, um, force, macrhs, SI, SC, c, i, j, k, eps);
    mk_C_one<2, 1>(A, s, sl2, wm, vm, force, macrhs, SI, SC, c, i, j, k, eps);
  }
}

// stage D for direction D (mkflux.f90:2307-2408 x, 2411-2511 y, 1867-1972 z)
template <int D> DEVI void mk_D_one(const GArgs &A, const FV &s, const FV &slp, const FV &macD, const FV &macT1, const FV &macT2,
                                    const FV &force, const FV &macrhs, const FV &SC, const FV &sedge, const FV &flux,
                                    int c, int i, int j, int k, double eps) {
  constexpr int T1 = (D == 0) ? 1 : 0, T2 = (D == 2) ? 1 : 2;
  if (coord<T1>(i, j, k) > A.hi[T1] || coord<T2>(i, j, k) > A.hi[T2]) return;
  double L, R;
  mk_pair<D>(A, s, slp, macD, force, macrhs, c, i, j, k, L, R);
  const double dt2 = 0.5 * A.dt, dt4 = A.dt / 4.0;
  const bool cons = A.cons[c] != 0;
  double LR[2] = { L, R };
  #pragma unroll
  for (int sd = 0; sd < 2; sd++) {
    const int ci = i - ((sd == 0) && D == 0), cj = j - ((sd == 0) && D == 1), ck = k - ((sd == 0) && D == 2);
    // the T1 term uses the T1-face state corrected by T2 (e.g. sedgex: simhyz), the T2 term the T2-face
    // state corrected by T1 (simhzy)
    double v = LR[sd];
    v = v - trans_term<T1>(A, SC, sc_idx(T1, T2, A.ncomp, c), macT1, cons, ci, cj, ck, dt2, dt4);
    v = v - trans_term<T2>(A, SC, sc_idx(T2, T1, A.ncomp, c), macT2, cons, ci, cj, ck, dt2, dt4);
    const double s0 = fv_get(s, ci, cj, ck, c);
    if (cons) {
      v = v + (dt2 / A.dx[T1]) * s0 * (ld<T1>(macT1, ci, cj, ck, 1) - fv_get(macT1, ci, cj, ck));
      v = v + (dt2 / A.dx[T2]) * s0 * (ld<T2>(macT2, ci, cj, ck, 1) - fv_get(macT2, ci, cj, ck));
    }
    if (!A.use_minion) {
      v = v + dt2 * fv_get(force, ci, cj, ck, c);
      if (cons) v = v - dt2 * s0 * fv_get(macrhs, ci, cj, ck);
    }
    LR[sd] = v;
  }
  const double um = fv_get(macD, i, j, k);
  double e = upwind_mac(LR[0], LR[1], um, eps);
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) {                                   // mkflux.f90:2369-2402
    const int ph = A.phys[D][side];
    const double in = (side == 0) ? LR[1] : LR[0];
    const bool vel = A.is_vel != 0;
    if (ph == VDN_INLET) e = (side == 0) ? ld<D>(s, i, j, k, -1, c) : fv_get(s, i, j, k, c);
    else if (ph == VDN_SLIP_WALL) e = (vel && c == D) ? 0.0 : in;
    else if (ph == VDN_NO_SLIP_WALL) e = vel ? 0.0 : in;
    else if (ph == VDN_OUTLET) e = (vel && c == D) ? ((side == 0) ? fmin(in, 0.0) : fmax(in, 0.0)) : in;
  }
  fv_at(sedge, i, j, k, c) = e;
  if (cons) fv_at(flux, i, j, k, c) = e * um;        // mkflux.f90:1969, 2405, 2508
}

__global__ void __launch_bounds__(256) kk_mk_D(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SC,
                        FV sex, FV sey, FV sez, FV flx, FV fly, FV flz, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  for (int c = 0; c < A.ncomp; c++) {
    mk_D_one<0>(A, s, sl0, um, vm, wm, force, macrhs, SC, sex, flx, c, i, j, k, eps);
    mk_D_one<1>(A, s, sl1, vm, um, wm, force, macrhs, SC, sey, fly, c, i, j, k, eps);
    mk_D_one<2>(A, s, sl2, wm, um, vm, force, macrhs, SC, sez, flz, c, i, j, k, eps);
  }
}

// max |umac| over the valid faces of the three MAC components (mkflux.f90:1374-1396)
__global__ void kk_macmax(FV um, FV vm, FV wm, GArgs A, Range3 r, double *out) {
  REDUCE_IJ(r)
  double m = 0.0;
  if (in_ij) REDUCE_KLOOP(r) {
    if (j <= A.hi[1] && k <= A.hi[2]) m = fmax(m, fabs(fv_get(um, i, j, k)));
    if (i <= A.hi[0] && k <= A.hi[2]) m = fmax(m, fabs(fv_get(vm, i, j, k)));
    if (i <= A.hi[0] && j <= A.hi[1]) m = fmax(m, fabs(fv_get(wm, i, j, k)));
  }
  block_atomic_max(out, m);
}
__global__ void kk_velmax(FV u, Range3 r, double *out) {             // velpred.f90:1965-1975
  REDUCE_IJ(r)
  double m = 0.0;
  if (in_ij) REDUCE_KLOOP(r) m = fmax(m, fmax(fmax(fabs(fv_get(u, i, j, k, 0)), fabs(fv_get(u, i, j, k, 1))), fabs(fv_get(u, i, j, k, 2))));
  block_atomic_max(out, m);
}

static void fill_gargs(GArgs &A, const vdn_multifab *s, int ibox, const vdn_bc_tower *bct, int bccomp, int ncomp, const double *dx, double dt) {
  memset(&A, 0, sizeof A);
  BoxP bp = make_boxp(s, ibox, bct);
  for (int d = 0; d < 3; d++) {
    A.lo[d] = bp.lo[d]; A.hi[d] = bp.hi[d]; A.dx[d] = dx ? dx[d] : 1.0;
    for (int sd = 0; sd < 2; sd++) {
      A.phys[d][sd] = bp.phys[d][sd];
      for (int c = 0; c < ncomp && c < 3; c++) A.adv[d][sd][c] = bct->adv_bc(s->lev, ibox + 1, d, sd, bccomp + c);
    }
  }
  A.dt = dt; A.ncomp = ncomp; A.use_minion = ctx().prm.use_minion; A.slope_order = ctx().prm.slope_order;
}
static FV work_fv(double *p, const BoxP &b, int nc_unused) {
  (void)nc_unused;
  FV f; f.p = p; f.a0 = b.lo[0] - 1; f.a1 = b.lo[1] - 1; f.a2 = b.lo[2] - 1;
  f.n0 = b.hi[0] - b.lo[0] + 3; f.n1 = b.hi[1] - b.lo[1] + 3; f.n2 = b.hi[2] - b.lo[2] + 3;
  f.sc = (long)f.n0 * f.n1 * f.n2;
  return f;
}

void k_slope(const vdn_multifab *s, vdn_multifab *slope, int dir, int bccomp, const vdn_bc_tower *bct) {
  REQUIRE(s->ng >= 3 && slope->ng == 1 && slope->nc >= s->nc && s->nc <= 3, "k_slope: need s.ng>=3, slope.ng==1, nc<=3");
  for (int i = 0; i < s->nfabs(); i++) {
    GArgs A; fill_gargs(A, s, i, bct, bccomp, s->nc, nullptr, 0.0);
    Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = A.lo[d] - 1; r.hi[d] = A.hi[d] + 1; }
    hipLaunchKernelGGL(kk_slopes, grid_for(r), dim3(64, 4, 1), 0, ctx().stream, s->fabs[i], slope->fabs[i], slope->fabs[i], slope->fabs[i], A, r, 1 << dir);
  }
}

void k_mkflux(const vdn_multifab *s, vdn_multifab **sedge, vdn_multifab **flux, vdn_multifab **umac,
              const vdn_multifab *force, const vdn_multifab *mac_rhs, const double *dx, double dt,
              const vdn_bc_tower *bct, bool is_vel, const int *is_cons) {
  const int ncomp = s->nc;
  REQUIRE(ncomp <= 3, "mkflux: at most 3 components per call (got %d)", ncomp);
  REQUIRE(s->ng >= 3 && umac[0]->ng >= 1 && force->ng >= 1 && mac_rhs->ng >= 1, "mkflux: ghost widths");
  const int bccomp = is_vel ? 0 : bct->dm;            // mkflux.f90:62-66
  hipStream_t st = ctx().stream;
  for (int ib = 0; ib < s->nfabs(); ib++) {
    size_t mark = arena_mark();
    GArgs A; fill_gargs(A, s, ib, bct, bccomp, ncomp, dx, dt);
    A.is_vel = is_vel ? 1 : 0;
    for (int c = 0; c < ncomp; c++) A.cons[c] = is_cons[c] ? 1 : 0;
    BoxP bp = make_boxp(s, ib, bct);
    FV w = work_fv(nullptr, bp, 0);
    const size_t fld = (size_t)w.sc * sizeof(double);
    FV sl[3], SI = w, SC = w;
    for (int d = 0; d < 3; d++) { sl[d] = w; sl[d].p = (double *)arena_alloc(fld * ncomp); }
    SI.p = (double *)arena_alloc(fld * 3 * ncomp);
    SC.p = (double *)arena_alloc(fld * 6 * ncomp);
    double *umax = (double *)arena_alloc(256);
    HIPCHK(hipMemsetAsync(umax, 0, sizeof(double), st));
    Range3 rg, rf;
    for (int d = 0; d < 3; d++) { rg.lo[d] = A.lo[d] - 1; rg.hi[d] = A.hi[d] + 1; rf.lo[d] = A.lo[d]; rf.hi[d] = A.hi[d] + 1; }
    const FV &um = umac[0]->fabs[ib], &vm = umac[1]->fabs[ib], &wm = umac[2]->fabs[ib];
    hipLaunchKernelGGL(kk_macmax, reduce_grid(rf), dim3(64, 4, 1), 0, st, um, vm, wm, A, rf, umax);
    hipLaunchKernelGGL(kk_slopes, grid_for(rg), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[2], A, rg, 7);
    hipLaunchKernelGGL(kk_mk_B, grid_for(rg), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SI, A, rg, umax);
    hipLaunchKernelGGL(kk_mk_C, grid_for(rg), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SI, SC, A, rg, umax);
    hipLaunchKernelGGL(kk_mk_D, grid_for(rf), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SC,
                       sedge[0]->fabs[ib], sedge[1]->fabs[ib], sedge[2]->fabs[ib], flux[0]->fabs[ib], flux[1]->fabs[ib], flux[2]->fabs[ib], A, rf, umax);
    arena_release(mark);
  }
}

// ====================================================================================================
// velpred
// ====================================================================================================
// predictor pair of ALL three velocity components on the lower D-face of cell (i,j,k), after bc
// (velpred.f90:2019-2079 x, 2105-2165 y, 2283-2343 z)
template <int D> DEVI void vp_pair(const GArgs &A, const FV &u, const FV &slp, const FV &force, int i, int j, int k, double L[3], double R[3]) {
  const double dt2 = 0.5 * A.dt;
  const double ul = ld<D>(u, i, j, k, -1, D), ur = fv_get(u, i, j, k, D);
  double cfl_l;
  if (D == 1) cfl_l = dt2 * fmax(0.0, ul / A.dx[1]);       // velpred.f90:2108: division inside max()
  else cfl_l = dt2 * fmax(0.0, ul) / A.dx[D];
  const double cfl_r = dt2 * fmin(0.0, ur) / A.dx[D];
  #pragma unroll
  for (int c = 0; c < 3; c++) {
    L[c] = ld<D>(u, i, j, k, -1, c) + (0.5 - cfl_l) * ld<D>(slp, i, j, k, -1, c);
    R[c] = fv_get(u, i, j, k, c) - (0.5 + cfl_r) * fv_get(slp, i, j, k, c);
    if (A.use_minion) { L[c] = L[c] + dt2 * ld<D>(force, i, j, k, -1, c); R[c] = R[c] + dt2 * fv_get(force, i, j, k, c); }
  }
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) {
    #pragma unroll
    for (int c = 0; c < 3; c++) {
      const double ghost = (side == 0) ? ld<D>(u, i, j, k, -1, c) : fv_get(u, i, j, k, c);
      bc_pair(L[c], R[c], A.phys[D][side], side, true, c == D, ghost, D == 0 && side == 1);   // :2075 quirk
    }
  }
}

// stage B: UI[(D*3 + c)] = uimh_D component c  (velpred.f90:2081-2098 etc.)
template <int D> DEVI void vp_B_one(const GArgs &A, const FV &u, const FV &slp, const FV &force, const FV &UI, int i, int j, int k, double eps) {
  if (coord<D>(i, j, k) < A.lo[D]) return;
  double L[3], R[3];
  vp_pair<D>(A, u, slp, force, i, j, k, L, R);
  const double uavg = 0.5 * (L[D] + R[D]);
  const bool test = ((L[D] <= 0.0 && R[D] >= 0.0) || (fabs(L[D] + R[D]) < eps));
  double un = (uavg > 0.0) ? L[D] : R[D];
  un = test ? 0.0 : un;
  #pragma unroll
  for (int c = 0; c < 3; c++) {
    double out;
    if (c == D) out = un;
    else {
      double v = (un > 0.0) ? L[c] : R[c];
      double av = 0.5 * (L[c] + R[c]);
      out = (fabs(un) < eps) ? av : v;
    }
    fv_at(UI, i, j, k, D * 3 + c) = out;
  }
}
__global__ void __launch_bounds__(256) kk_vp_B(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  vp_B_one<0>(A, u, sl0, force, UI, i, j, k, eps);
  vp_B_one<1>(A, u, sl1, force, UI, i, j, k, eps);
  vp_B_one<2>(A, u, sl2, force, UI, i, j, k, eps);
}

// stage C: component C on D-faces corrected by the third direction O = 3-C-D
// (uimhyz = (C=0,D=1) velpred.f90:2466-2503, wimhxy = (C=2,D=0) 2189-2229, ...)
// XC component index: C*2 + (D > C ? D-1 : D)
DEVI int xc_idx(int C, int D) { return C * 2 + (D > C ? D - 1 : D); }
template <int C, int D> DEVI void vp_C_one(const GArgs &A, const FV &u, const FV &slp, const FV &force, const FV &UI, const FV &XC,
                                           int i, int j, int k, double eps) {
  constexpr int O = 3 - C - D;
  const int qd = coord<D>(i, j, k), qo = coord<O>(i, j, k);
  if (qd < A.lo[D] || qo < A.lo[O] || qo > A.hi[O]) return;
  double Lv[3], Rv[3];
  vp_pair<D>(A, u, slp, force, i, j, k, Lv, Rv);
  const double dt6 = A.dt / 6.0;
  const int ci = i - (D == 0), cj = j - (D == 1), ck = k - (D == 2);
  double L = Lv[C] - (dt6 / A.dx[O]) * (ld<O>(UI, ci, cj, ck, 1, O * 3 + O) + fv_get(UI, ci, cj, ck, O * 3 + O))
                                     * (ld<O>(UI, ci, cj, ck, 1, O * 3 + C) - fv_get(UI, ci, cj, ck, O * 3 + C));
  double R = Rv[C] - (dt6 / A.dx[O]) * (ld<O>(UI, i, j, k, 1, O * 3 + O) + fv_get(UI, i, j, k, O * 3 + O))
                                     * (ld<O>(UI, i, j, k, 1, O * 3 + C) - fv_get(UI, i, j, k, O * 3 + C));
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) bc_pair(L, R, A.phys[D][side], side, true, false, side == 0 ? ld<D>(u, i, j, k, -1, C) : fv_get(u, i, j, k, C), false);
  const double un = fv_get(UI, i, j, k, D * 3 + D);
  const double v = (un > 0.0) ? L : R;
  const double av = 0.5 * (L + R);
  fv_at(XC, i, j, k, xc_idx(C, D)) = (fabs(un) < eps) ? av : v;
}
__global__ void __launch_bounds__(256) kk_vp_C(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, FV XC, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  vp_C_one<0, 1>(A, u, sl1, force, UI, XC, i, j, k, eps);
  vp_C_one<0, 2>(A, u, sl2, force, UI, XC, i, j, k, eps);
  vp_C_one<1, 0>(A, u, sl0, force, UI, XC, i, j, k, eps);
  vp_C_one<1, 2>(A, u, sl2, force, UI, XC, i, j, k, eps);
  vp_C_one<2, 0>(A, u, sl0, force, UI, XC, i, j, k, eps);
  vp_C_one<2, 1>(A, u, sl1, force, UI, XC, i, j, k, eps);
}

// stage D: the MAC velocity on valid D-faces (velpred.f90:2616-2660, 2666-2710, 2372-2416)
template <int D> DEVI void vp_D_one(const GArgs &A, const FV &u, const FV &slp, const FV &force, const FV &UI, const FV &XC, const FV &umac,
                                    int i, int j, int k, double eps) {
  constexpr int T1 = (D == 0) ? 1 : 0, T2 = (D == 2) ? 1 : 2;
  if (coord<T1>(i, j, k) > A.hi[T1] || coord<T2>(i, j, k) > A.hi[T2]) return;
  double Lv[3], Rv[3];
  vp_pair<D>(A, u, slp, force, i, j, k, Lv, Rv);
  const double dt2 = 0.5 * A.dt, dt4 = A.dt / 4.0;
  double LR[2] = { Lv[D], Rv[D] };
  #pragma unroll
  for (int sd = 0; sd < 2; sd++) {
    const int ci = i - ((sd == 0) && D == 0), cj = j - ((sd == 0) && D == 1), ck = k - ((sd == 0) && D == 2);
    double v = LR[sd]
      - (dt4 / A.dx[T1]) * (ld<T1>(UI, ci, cj, ck, 1, T1 * 3 + T1) + fv_get(UI, ci, cj, ck, T1 * 3 + T1))
                         * (ld<T1>(XC, ci, cj, ck, 1, xc_idx(D, T1)) - fv_get(XC, ci, cj, ck, xc_idx(D, T1)))
      - (dt4 / A.dx[T2]) * (ld<T2>(UI, ci, cj, ck, 1, T2 * 3 + T2) + fv_get(UI, ci, cj, ck, T2 * 3 + T2))
                         * (ld<T2>(XC, ci, cj, ck, 1, xc_idx(D, T2)) - fv_get(XC, ci, cj, ck, xc_idx(D, T2)));
    if (!A.use_minion) v = v + dt2 * fv_get(force, ci, cj, ck, D);
    LR[sd] = v;
  }
  const double L = LR[0], R = LR[1];
  const double uavg = 0.5 * (L + R);
  const bool test = ((L <= 0.0 && R >= 0.0) || (fabs(L + R) < eps));
  double v = (uavg > 0.0) ? L : R;
  v = test ? 0.0 : v;
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) {                                   // velpred.f90:2642-2659
    const int ph = A.phys[D][side];
    if (ph == VDN_SLIP_WALL || ph == VDN_NO_SLIP_WALL) v = 0.0;
    else if (ph == VDN_INLET) v = (side == 0) ? ld<D>(u, i, j, k, -1, D) : fv_get(u, i, j, k, D);
    else if (ph == VDN_OUTLET) v = (side == 0) ? fmin(R, 0.0) : fmax(L, 0.0);
  }
  fv_at(umac, i, j, k) = v;
}
__global__ void __launch_bounds__(256) kk_vp_D(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, FV XC, FV um, FV vm, FV wm, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  vp_D_one<0>(A, u, sl0, force, UI, XC, um, i, j, k, eps);
  vp_D_one<1>(A, u, sl1, force, UI, XC, vm, i, j, k, eps);
  vp_D_one<2>(A, u, sl2, force, UI, XC, wm, i, j, k, eps);
}

void k_velpred(const vdn_multifab *u, vdn_multifab **umac, const vdn_multifab *force, const double *dx, double dt,
               const vdn_bc_tower *bct) {
  REQUIRE(u->nc == 3 && u->ng >= 3 && force->ng >= 1 && umac[0]->ng >= 1, "velpred: operand shapes");
  hipStream_t st = ctx().stream;
  for (int ib = 0; ib < u->nfabs(); ib++) {
    size_t mark = arena_mark();
    GArgs A; fill_gargs(A, u, ib, bct, 0, 3, dx, dt);
    A.is_vel = 1;
    BoxP bp = make_boxp(u, ib, bct);
    FV w = work_fv(nullptr, bp, 0);
    const size_t fld = (size_t)w.sc * sizeof(double);
    FV sl[3], UI = w, XC = w;
    for (int d = 0; d < 3; d++) { sl[d] = w; sl[d].p = (double *)arena_alloc(fld * 3); }
    UI.p = (double *)arena_alloc(fld * 9);
    XC.p = (double *)arena_alloc(fld * 6);
    double *umax = (double *)arena_alloc(256);
    HIPCHK(hipMemsetAsync(umax, 0, sizeof(double), st));
    Range3 rv, rg, rf;
    for (int d = 0; d < 3; d++) { rv.lo[d] = A.lo[d]; rv.hi[d] = A.hi[d]; rg.lo[d] = A.lo[d] - 1; rg.hi[d] = A.hi[d] + 1; rf.lo[d] = A.lo[d]; rf.hi[d] = A.hi[d] + 1; }
    hipLaunchKernelGGL(kk_velmax, reduce_grid(rv), dim3(64, 4, 1), 0, st, u->fabs[ib], rv, umax);
    hipLaunchKernelGGL(kk_slopes, grid_for(rg), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[2], A, rg, 7);
    hipLaunchKernelGGL(kk_vp_B, grid_for(rg), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, A, rg, umax);
    hipLaunchKernelGGL(kk_vp_C, grid_for(rg), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC, A, rg, umax);
    hipLaunchKernelGGL(kk_vp_D, grid_for(rf), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC,
                       umac[0]->fabs[ib], umac[1]->fabs[ib], umac[2]->fabs[ib], A, rf, umax);
    arena_release(mark);
  }
}
