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
#include <algorithm>

// ---------------------------------------------------------------------------------------------------
struct GArgs {
  int lo[3], hi[3];
  int phys[3][2];
  int adv[3][2][3];          // adv bc of the (up to 3) advected components, for the slope specials
  double dx[3], dt;
  int ncomp, is_vel, use_minion, slope_order;
  int cons[3];
  int outlet2d;              // 1: velpred's hi-x OUTLET takes velpred_2d's rule (max, velpred.f90:305) instead of velpred_3d's (min, :2075): vdn_set_extruded_2d
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

// the limited slope of a cell from its five values along one direction (m2 .. p2 = offsets -2 .. +2); pos: the cell's index along that
// direction, is / ie: first / last valid index, lo_sp / hi_sp: EXT_DIR or HOEXTRAP on that face (the one-sided formulas of slope.f90)
DEVI double slope_vals(double m2, double m1, double s0, double p1, double p2, int pos, int is, int ie, bool lo_sp, bool hi_sp, int order) {
  if (order == 0) return 0.0;
  const double two3rd = 2.0 / 3.0, sixth = 1.0 / 6.0, third = 1.0 / 3.0, tenth = 0.1;
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
  else fp = fromm_of(s0, p1, p2).fromm;          // fromm(i+1) uses s(i), s(i+1), s(i+2)
  if (lo_sp && pos == is + 1)       // revised fromm(is) = slope(is)   (slope.f90:256-260)
    fm = limited(-16.0 / 15.0 * m2 + 0.5 * m1 + two3rd * s0 - tenth * p1, m2, m1, s0);
  else
    fm = fromm_of(m2, m1, s0).fromm;
  double ds = 2.0 * two3rd * f0.cen - sixth * (fp + fm);
  return f0.flag * fmin(fabs(ds), f0.lim);
}
// slope of component c along D at cell (i,j,k); needs s at offsets -2..+2 along D
template <int D> DEVI double slope_at(const FV &s, int c, int i, int j, int k, int is, int ie, bool lo_sp, bool hi_sp, int order) {
  if (order == 0) return 0.0;
  const double m2 = ld<D>(s, i, j, k, -2, c), m1 = ld<D>(s, i, j, k, -1, c), s0 = ld<D>(s, i, j, k, 0, c),
               p1 = ld<D>(s, i, j, k, 1, c), p2 = ld<D>(s, i, j, k, 2, c);
  return slope_vals(m2, m1, s0, p1, p2, coord<D>(i, j, k), is, ie, lo_sp, hi_sp, order);
}

DEVI void slopes_cell(const FV &s, const FV &sl0, const FV &sl1, const FV &sl2, const GArgs &A, int dirmask, int i, int j, int k) {
  for (int c = 0; c < A.ncomp; c++) {
    #define SPEC(d, sd) (A.adv[d][sd][c] == VDN_EXT_DIR || A.adv[d][sd][c] == VDN_HOEXTRAP)
    if (dirmask & 1) fv_at(sl0, i, j, k, c) = slope_at<0>(s, c, i, j, k, A.lo[0], A.hi[0], SPEC(0, 0), SPEC(0, 1), A.slope_order);
    if (dirmask & 2) fv_at(sl1, i, j, k, c) = slope_at<1>(s, c, i, j, k, A.lo[1], A.hi[1], SPEC(1, 0), SPEC(1, 1), A.slope_order);
    if (dirmask & 4) fv_at(sl2, i, j, k, c) = slope_at<2>(s, c, i, j, k, A.lo[2], A.hi[2], SPEC(2, 0), SPEC(2, 1), A.slope_order);
    #undef SPEC
  }
}
// vmax (velpred only): max |u| over the valid cells of the three components rides along (kk_velmax, velpred.f90:1965-1975) -- the
// kernel reads every cell of u anyway
__global__ void __launch_bounds__(256) kk_slopes(FV s, FV sl0, FV sl1, FV sl2, GArgs A, Range3 r, int dirmask, double *vmax = nullptr) {
  THREAD_IJK(r)
  if (in_range) slopes_cell(s, sl0, sl1, sl2, A, dirmask, i, j, k);
  if (vmax) {                                     // uniform
    double m = 0.0;
    if (in_range && i >= A.lo[0] && i <= A.hi[0] && j >= A.lo[1] && j <= A.hi[1] && k >= A.lo[2] && k <= A.hi[2])
      m = fmax(m, fmax(fmax(fabs(fv_get(s, i, j, k, 0)), fabs(fv_get(s, i, j, k, 1))), fabs(fv_get(s, i, j, k, 2))));
    block_atomic_max(vmax, m);
  }
}

// k-marching form of kk_slopes for all three directions at once: kk_slopes reads the five values of every direction through the L1 (13
// loads per cell and component; texture addresser busy 73 %, 2.1 GB fetched for 0.43 GB of s).  Here a thread keeps the five planes
// k-2 .. k+2 of its column in registers (one load per plane and component), takes x-neighbours from the lanes next to it and y-neighbours
// from the rows next to it (LDS); tiles overlap by two cells either side (60 x 12 of 64 x 16 cells owned).  slope_vals on the same values.
constexpr int SNY = 16;
// (Measured and not kept: sharing the Fromm slopes between neighbours -- one fromm_of per cell and direction instead of three, exchanged like
// the values -- needs more than the 128 VGPRs of a 1024-thread workgroup for three components: 3.1 ms spilling, 1.37 ms one component at a
// time with two barriers each, against 0.69 ms for this form; two components: 0.46 against 0.47 ms.)
template <int NC> __global__ void __launch_bounds__(64 * SNY) kk_slopes_m(FV s, FV sl0, FV sl1, FV sl2, GArgs A, Range3 r, int klen, double *vmax) {
  __shared__ double ly[NC][SNY][64];
  const int lane = threadIdx.x, row = threadIdx.y;
  const int i = r.lo[0] - 2 + (int)blockIdx.x * 60 + lane, j = r.lo[1] - 2 + (int)blockIdx.y * (SNY - 4) + row;
  const bool own_ij = lane >= 2 && lane <= 61 && row >= 2 && row <= SNY - 3 && i <= r.hi[0] && j <= r.hi[1];
  const int ic = min(max(i, A.lo[0] - 3), A.hi[0] + 3), jc = min(max(j, A.lo[1] - 3), A.hi[1] + 3);
  const int k0 = r.lo[2] + (int)blockIdx.z * klen, k1 = min(k0 + klen - 1, r.hi[2]);
  const long sp = (long)s.n0 * s.n1;
  const double *ps[NC];
  #pragma unroll
  for (int c = 0; c < NC; c++) ps[c] = s.p + fv_idx(s, ic, jc, s.a2) + s.sc * c;            // plane a2 of the column
  #define SPL(c, kk) ps[c][(long)(min(max((kk), A.lo[2] - 3), A.hi[2] + 3) - s.a2) * sp]
  #define SPEC(d, sd, c) (A.adv[d][sd][c] == VDN_EXT_DIR || A.adv[d][sd][c] == VDN_HOEXTRAP)
  double w[NC][5];                                                    // the column's planes k-2 .. k+2
  #pragma unroll
  for (int c = 0; c < NC; c++) { w[c][0] = 0.0; for (int q = 1; q < 5; q++) w[c][q] = SPL(c, k0 - 3 + q); }
  double m = 0.0;
  for (int k = k0; k <= k1; k++) {
    #pragma unroll
    for (int c = 0; c < NC; c++) { w[c][0] = w[c][1]; w[c][1] = w[c][2]; w[c][2] = w[c][3]; w[c][3] = w[c][4]; w[c][4] = SPL(c, k + 2); }
    #pragma unroll
    for (int c = 0; c < NC; c++) ly[c][row][lane] = w[c][2];
    __syncthreads();
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      const double s0 = w[c][2];
      const double xm1 = lane_prev(s0), xp1 = lane_next(s0), xm2 = lane_prev(xm1), xp2 = lane_next(xp1);
      if (own_ij) {
        const double ym2 = ly[c][row - 2][lane], ym1 = ly[c][row - 1][lane], yp1 = ly[c][row + 1][lane], yp2 = ly[c][row + 2][lane];
        fv_at(sl0, i, j, k, c) = slope_vals(xm2, xm1, s0, xp1, xp2, i, A.lo[0], A.hi[0], SPEC(0, 0, c), SPEC(0, 1, c), A.slope_order);
        fv_at(sl1, i, j, k, c) = slope_vals(ym2, ym1, s0, yp1, yp2, j, A.lo[1], A.hi[1], SPEC(1, 0, c), SPEC(1, 1, c), A.slope_order);
        fv_at(sl2, i, j, k, c) = slope_vals(w[c][0], w[c][1], s0, w[c][3], w[c][4], k, A.lo[2], A.hi[2], SPEC(2, 0, c), SPEC(2, 1, c), A.slope_order);
        if (vmax && i >= A.lo[0] && i <= A.hi[0] && j >= A.lo[1] && j <= A.hi[1] && k >= A.lo[2] && k <= A.hi[2]) m = fmax(m, fabs(s0));
      }
    }
    __syncthreads();      // (round 5, measured: double-buffering ly to drop this barrier made the march slower, 0.681 -> 0.736 ms: 48 KB of LDS per workgroup)
  }
  #undef SPL
  #undef SPEC
  if (vmax) block_atomic_max(vmax, m);
}
// The same march without the row exchange (round 5): the y-neighbours of plane k come straight from memory (they are the values the rows next door loaded
// two planes ago: L2 hits), so there is no LDS, no barrier and no row overlap -- 64 x 4 threads own 60 x 4 cells, several workgroups share a CU.
// (Counters: 1.77 GB fetched per three-component launch for 0.43 GB of s -- the row and plane halos of neighbouring tiles rarely meet in one L2.  Measured and not
// kept: the XCD-aware tile order of the other marches (xcd_tile) 0.645 -> 0.674 ms, with 64 x 8 workgroups 0.87 ms; fewer k-chunks no better.)
// (Measured on this form and not kept: one fromm_of per cell along x and z -- the neighbours' from the lanes next door / carried from plane to plane -- instead of
// three: 0.659 -> 0.737 ms for three components, 0.390 -> 0.412 for two; the kernel is not bound by its arithmetic.)
template <int NC> __global__ void __launch_bounds__(256) kk_slopes_my(FV s, FV sl0, FV sl1, FV sl2, GArgs A, Range3 r, int klen, double *vmax) {
  const int lane = threadIdx.x, row = threadIdx.y;
  const int i = r.lo[0] - 2 + (int)blockIdx.x * 60 + lane, j = r.lo[1] + (int)blockIdx.y * 4 + row;
  const bool own_ij = lane >= 2 && lane <= 61 && i <= r.hi[0] && j <= r.hi[1];
  const int ic = min(max(i, A.lo[0] - 3), A.hi[0] + 3), jc = min(max(j, A.lo[1] - 3), A.hi[1] + 3);
  const int k0 = r.lo[2] + (int)blockIdx.z * klen, k1 = min(k0 + klen - 1, r.hi[2]);
  const long sp = (long)s.n0 * s.n1;
  long oy[4];                                                         // rows j-2, j-1, j+1, j+2 relative to the column (kept inside the allocation like the planes)
  { const int dj[4] = { -2, -1, 1, 2 };
    #pragma unroll
    for (int q = 0; q < 4; q++) oy[q] = (long)(min(max(j + dj[q], A.lo[1] - 3), A.hi[1] + 3) - jc) * s.n0; }
  const double *ps[NC];
  #pragma unroll
  for (int c = 0; c < NC; c++) ps[c] = s.p + fv_idx(s, ic, jc, s.a2) + s.sc * c;
  #define SPL(c, kk) ps[c][(long)(min(max((kk), A.lo[2] - 3), A.hi[2] + 3) - s.a2) * sp]
  #define SPY(c, kk, q) ps[c][(long)(min(max((kk), A.lo[2] - 3), A.hi[2] + 3) - s.a2) * sp + oy[q]]
  #define SPEC(d, sd, c) (A.adv[d][sd][c] == VDN_EXT_DIR || A.adv[d][sd][c] == VDN_HOEXTRAP)
  double w[NC][5];
  #pragma unroll
  for (int c = 0; c < NC; c++) { w[c][0] = 0.0; for (int q = 1; q < 5; q++) w[c][q] = SPL(c, k0 - 3 + q); }
  double m = 0.0;
  for (int k = k0; k <= k1; k++) {
    double y[NC][4];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      w[c][0] = w[c][1]; w[c][1] = w[c][2]; w[c][2] = w[c][3]; w[c][3] = w[c][4]; w[c][4] = SPL(c, k + 2);
      #pragma unroll
      for (int q = 0; q < 4; q++) y[c][q] = SPY(c, k, q);
    }
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      const double s0 = w[c][2];
      const double xm1 = lane_prev(s0), xp1 = lane_next(s0), xm2 = lane_prev(xm1), xp2 = lane_next(xp1);
      if (own_ij) {
        fv_at(sl0, i, j, k, c) = slope_vals(xm2, xm1, s0, xp1, xp2, i, A.lo[0], A.hi[0], SPEC(0, 0, c), SPEC(0, 1, c), A.slope_order);
        fv_at(sl2, i, j, k, c) = slope_vals(w[c][0], w[c][1], s0, w[c][3], w[c][4], k, A.lo[2], A.hi[2], SPEC(2, 0, c), SPEC(2, 1, c), A.slope_order);
        fv_at(sl1, i, j, k, c) = slope_vals(y[c][0], y[c][1], s0, y[c][2], y[c][3], j, A.lo[1], A.hi[1], SPEC(1, 0, c), SPEC(1, 1, c), A.slope_order);
      }
      if (own_ij && vmax && i >= A.lo[0] && i <= A.hi[0] && j >= A.lo[1] && j <= A.hi[1] && k >= A.lo[2] && k <= A.hi[2]) m = fmax(m, fabs(s0));
    }
  }
  #undef SPL
  #undef SPY
  #undef SPEC
  if (vmax) block_atomic_max(vmax, m);
}
static void launch_slopes(const FV &s, const FV sl[3], const GArgs &A, const Range3 &rg, int ncomp, double *vmax, hipStream_t st) {
  static const bool marching = !(vdn_env("VDN_SLOPES_MARCH") && atoi(vdn_env("VDN_SLOPES_MARCH")) == 0);
  static const bool yglobal = !(vdn_env("VDN_SLOPES_Y") && atoi(vdn_env("VDN_SLOPES_Y")) == 0);        // 0: the row exchange through LDS (kk_slopes_m: 0.691 / 0.470 ms)
  if (marching && yglobal && (ncomp == 2 || ncomp == 3) && s.a0 <= A.lo[0] - 3 && s.a1 <= A.lo[1] - 3 && s.a2 <= A.lo[2] - 3) {
    const int nx = rg.hi[0] - rg.lo[0] + 1, ny = rg.hi[1] - rg.lo[1] + 1, nz = rg.hi[2] - rg.lo[2] + 1;
    const int tiles = ((nx + 59) / 60) * ((ny + 3) / 4);
    int chunks = std::max(1, std::min(nz / 8, (32 * 256 + tiles - 1) / tiles));      // (measured at 256^3, three / two components: 4 x 256 workgroups 0.741 / 0.477 ms, 8 x 0.667 / 0.418, 16 x 0.661 / 0.406, 32 x 0.645 / 0.384)
    const int klen = (nz + chunks - 1) / chunks;
    const dim3 g((nx + 59) / 60, (ny + 3) / 4, (nz + klen - 1) / klen), blk(64, 4, 1);
    if (ncomp == 3) hipLaunchKernelGGL(kk_slopes_my<3>, g, blk, 0, st, s, sl[0], sl[1], sl[2], A, rg, klen, vmax);
    else hipLaunchKernelGGL(kk_slopes_my<2>, g, blk, 0, st, s, sl[0], sl[1], sl[2], A, rg, klen, vmax);
    return;
  }
  if (marching && (ncomp == 2 || ncomp == 3) && s.a0 <= A.lo[0] - 3 && s.a1 <= A.lo[1] - 3 && s.a2 <= A.lo[2] - 3) {
    const int nx = rg.hi[0] - rg.lo[0] + 1, ny = rg.hi[1] - rg.lo[1] + 1, nz = rg.hi[2] - rg.lo[2] + 1;
    const int tiles = ((nx + 59) / 60) * ((ny + SNY - 5) / (SNY - 4));
    int chunks = std::max(1, std::min(nz / 8, (4 * 256 + tiles - 1) / tiles));        // ~4 workgroups per CU, chunks of >= 8 planes
    const int klen = (nz + chunks - 1) / chunks;
    const dim3 g((nx + 59) / 60, (ny + SNY - 5) / (SNY - 4), (nz + klen - 1) / klen), blk(64, SNY, 1);
    if (ncomp == 3) hipLaunchKernelGGL(kk_slopes_m<3>, g, blk, 0, st, s, sl[0], sl[1], sl[2], A, rg, klen, vmax);
    else hipLaunchKernelGGL(kk_slopes_m<2>, g, blk, 0, st, s, sl[0], sl[1], sl[2], A, rg, klen, vmax);
    return;
  }
  hipLaunchKernelGGL(kk_slopes, grid_for(rg), dim3(64, 4, 1), 0, st, s, sl[0], sl[1], sl[2], A, rg, 7, vmax);
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
struct MkPlain { FV s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SI, SC, sex, sey, sez, flx, fly, flz; GArgs A; const double *umax; };
DEVI void mk_B_cell(const FV &s, const FV &sl0, const FV &sl1, const FV &sl2, const FV &um, const FV &vm, const FV &wm, const FV &force, const FV &macrhs, const FV &SI,
                    const GArgs &A, int i, int j, int k, double eps, int dmask = 7) {
  for (int c = 0; c < A.ncomp; c++) {
    double L, R;
    if ((dmask & 1) && i >= A.lo[0]) { mk_pair<0>(A, s, sl0, um, force, macrhs, c, i, j, k, L, R); fv_at(SI, i, j, k, 0 * A.ncomp + c) = upwind_mac(L, R, fv_get(um, i, j, k), eps); }
    if ((dmask & 2) && j >= A.lo[1]) { mk_pair<1>(A, s, sl1, vm, force, macrhs, c, i, j, k, L, R); fv_at(SI, i, j, k, 1 * A.ncomp + c) = upwind_mac(L, R, fv_get(vm, i, j, k), eps); }
    if ((dmask & 4) && k >= A.lo[2]) { mk_pair<2>(A, s, sl2, wm, force, macrhs, c, i, j, k, L, R); fv_at(SI, i, j, k, 2 * A.ncomp + c) = upwind_mac(L, R, fv_get(wm, i, j, k), eps); }
  }
}
__global__ void __launch_bounds__(256) kk_mk_B(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SI, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  mk_B_cell(s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SI, A, i, j, k, eps_from(umax));
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

DEVI void mk_C_cell(const FV &s, const FV &sl0, const FV &sl1, const FV &sl2, const FV &um, const FV &vm, const FV &wm, const FV &force, const FV &macrhs, const FV &SI, const FV &SC,
                    const GArgs &A, int i, int j, int k, double eps, int dmask = 7) {
  for (int c = 0; c < A.ncomp; c++) {
    if (dmask & 1) mk_C_one<0, 1>(A, s, sl0, um, vm, force, macrhs, SI, SC, c, i, j, k, eps);
    if (dmask & 1) mk_C_one<0, 2>(A, s, sl0, um, wm, force, macrhs, SI, SC, c, i, j, k, eps);
    if (dmask & 2) mk_C_one<1, 0>(A, s, sl1, vm, um, force, macrhs, SI, SC, c, i, j, k, eps);
    if (dmask & 2) mk_C_one<1, 2>(A, s, sl1, vm, wm, force, macrhs, SI, SC, c, i, j, k, eps);
    if (dmask & 4) mk_C_one<2, 0>(A, s, sl2, wm, um, force, macrhs, SI, SC, c, i, j, k, eps);
    if (dmask & 4) mk_C_one<2, 1>(A, s, sl2, wm, vm, force, macrhs, SI, SC, c, i, j, k, eps);
  }
}
__global__ void __launch_bounds__(256) kk_mk_C(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SI, FV SC, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  mk_C_cell(s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SI, SC, A, i, j, k, eps_from(umax));
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

DEVI void mk_D_cell(const FV &s, const FV &sl0, const FV &sl1, const FV &sl2, const FV &um, const FV &vm, const FV &wm, const FV &force, const FV &macrhs, const FV &SC,
                    const FV &sex, const FV &sey, const FV &sez, const FV &flx, const FV &fly, const FV &flz, const GArgs &A, int i, int j, int k, double eps, int dmask = 7) {
  for (int c = 0; c < A.ncomp; c++) {
    if (dmask & 1) mk_D_one<0>(A, s, sl0, um, vm, wm, force, macrhs, SC, sex, flx, c, i, j, k, eps);
    if (dmask & 2) mk_D_one<1>(A, s, sl1, vm, um, wm, force, macrhs, SC, sey, fly, c, i, j, k, eps);
    if (dmask & 4) mk_D_one<2>(A, s, sl2, wm, um, vm, force, macrhs, SC, sez, flz, c, i, j, k, eps);
  }
}
__global__ void __launch_bounds__(256) kk_mk_D(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SC,
                        FV sex, FV sey, FV sez, FV flx, FV fly, FV flz, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  mk_D_cell(s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SC, sex, sey, sez, flx, fly, flz, A, i, j, k, eps_from(umax));
}

// ---- boundary slabs --------------------------------------------------------------------------------------------------------------
// Round 2: the marching kernels below carry NO boundary code.  The physical boundary rules (bc_pair and the edge-state rules) only
// change states ON the faces of a box that lie on a physical domain boundary: a two-dimensional set.  The marching kernels compute
// every face with the interior formulas, and each stage is followed by ONE launch of the face-centred code above over the (at most
// six) one-cell-thick slabs that hold those faces.  The face-centred and the marching forms are bit-identical
// (tests/test_kernels_gpu.py::test_godunov_marching_equals_face_centred), so the result is what the round-1 kernels produced, while the
// marches lose their branches, the register copies behind them (47 % of the instructions of kk_vp_B_m were v_mov) and ~100 VGPRs.
struct Slabs { int n; int dir[6], pos[6]; Range3 r; };
// slabs of range r (a stage's index range: lo-1..hi+1 for B and C, lo..hi+1 for D) that hold the physical boundary faces of the box
static Slabs boundary_slabs(const GArgs &A, const Range3 &r) {
  Slabs S; S.n = 0; S.r = r;
  for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) {
    const int ph = A.phys[d][sd];
    if (ph == VDN_INTERIOR || ph == VDN_PERIODIC) continue;
    S.dir[S.n] = d; S.pos[S.n] = sd == 0 ? A.lo[d] : A.hi[d] + 1; S.n++;
  }
  return S;
}
template <class F> __global__ void __launch_bounds__(256) kk_slabs(Slabs S, F f) {
  const int sl = blockIdx.z;
  const int d = S.dir[sl], da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
  int q[3];
  q[d] = S.pos[sl];
  q[da] = S.r.lo[da] + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  q[db] = S.r.lo[db] + (int)(blockIdx.y * blockDim.y + threadIdx.y);
  if (q[da] > S.r.hi[da] || q[db] > S.r.hi[db]) return;
  f(q[0], q[1], q[2], 1 << d);          // only the states on the faces normal to d change
}
template <class F> static void launch_slabs(const Slabs &S, const F &f, hipStream_t st) {
  if (S.n == 0) return;
  int m = 1;
  for (int d = 0; d < 3; d++) m = std::max(m, S.r.hi[d] - S.r.lo[d] + 1);
  hipLaunchKernelGGL((kk_slabs<F>), dim3((m + 63) / 64, (m + 3) / 4, S.n), dim3(64, 4, 1), 0, st, S, f);
}
struct MkBFix { MkPlain P; __device__ void operator()(int i, int j, int k, int dmask) const { mk_B_cell(P.s, P.sl0, P.sl1, P.sl2, P.um, P.vm, P.wm, P.force, P.macrhs, P.SI, P.A, i, j, k, eps_from(P.umax), dmask); } };
struct MkCFix { MkPlain P; __device__ void operator()(int i, int j, int k, int dmask) const { mk_C_cell(P.s, P.sl0, P.sl1, P.sl2, P.um, P.vm, P.wm, P.force, P.macrhs, P.SI, P.SC, P.A, i, j, k, eps_from(P.umax), dmask); } };
struct MkDFix { MkPlain P; __device__ void operator()(int i, int j, int k, int dmask) const { mk_D_cell(P.s, P.sl0, P.sl1, P.sl2, P.um, P.vm, P.wm, P.force, P.macrhs, P.SC, P.sex, P.sey, P.sez, P.flx, P.fly, P.flz, P.A, i, j, k, eps_from(P.umax), dmask); } };

// ====================================================================================================
// Cell-centred, k-marching form of the stage B / C / D kernels (the default path)
// ====================================================================================================
// Measured on MI355X (rocprofv3 PMC, 256^3): the face-centred kernels above issue ~230 loads per cell in stage D, most of
// them re-reading, for the LEFT side of a face, what the neighbouring thread reads for the RIGHT side of its own face;
// the per-CU vector L1 (TCP) is the bound (1.1e9 tag accesses, 1.8e9 pending-stall cycles per launch), not HBM.
// Every left/right state of a face is a function of ONE cell (the cell on that side), so here a thread owns a CELL,
// computes that cell's contribution to its lower faces (right states) and to its upper faces (left states) once, and
// hands the left states to the owner of the upper face:
//   x: to lane+1 by a wave shuffle      (tiles overlap by one cell: lane 0 is halo-only)
//   y: to row+1 through LDS             (tiles overlap by one row: row 0 is halo-only, one barrier per plane)
//   z: to itself at the next k-plane    (the workgroup marches in k; a register carry)
// The arithmetic of every state is unchanged (same expressions in the same order), so results are bit-identical.
// Boundary rule: on a physical boundary face the state of the OUTSIDE cell is dead (bc_pair overwrites it from the
// inside state or the ghost value) and the inside cell can apply its half of bc_pair alone; bc_pair is idempotent, so
// the face owner then applies the complete rule to the pair it has assembled.
// Code shape: every plane starts with ONE unconditional batch of loads (indices clamped into the arrays; threads outside
// the computable region produce values nobody uses), the rare boundary work sits in one branch after it.
// VDN_GODUNOV_BATCH=1 launches the descriptor (box-batched) kernels also for a level of one box. Measured at 256^3: they need
// fewer VGPRs (mk_D<1> 116 vs 174) yet run slower there (scalar 6.6 vs 5.6 ms, velocity 10.0 vs 8.2 ms), so one box keeps the by-value kernels
static bool batch_always() { static const bool b = vdn_env("VDN_GODUNOV_BATCH") && atoi(vdn_env("VDN_GODUNOV_BATCH")) != 0; return b; }
// VDN_GOD_SLAB_BC=0: the round-1 marches with the boundary code inside (kept for comparison); default: interior marches + boundary slabs
static bool slab_bc() { static const bool b = !(vdn_env("VDN_GOD_SLAB_BC") && atoi(vdn_env("VDN_GOD_SLAB_BC")) == 0); return b; }
// one launch per stage for all boxes (descriptors) or one set of launches per box (arguments by value)?  A level of an adaptive
// hierarchy (hundreds of 16^3 .. 32^3 boxes) is launch-bound box by box; a level of a few large boxes -- 512^3 cut into eight 256^3
// boxes -- runs faster box by box: the by-value kernels carry no boundary code (boundary slabs) and need fewer registers (measured,
// eight 256^3 boxes on one GPU: scalar 52 -> 42 ms, velocity 76 -> 54 ms per step).  VDN_GODUNOV_BATCH=1 forces the descriptors.
static bool use_batched(const vdn_multifab *s) {
  if (s->nfabs() == 0) return false;
  if (batch_always()) return true;
  if (s->nfabs() == 1) return false;
  long cells = 0;
  for (int b = 0; b < s->nfabs(); b++) cells += (long)(s->vbox[b].hi[0] - s->vbox[b].lo[0] + 1) * (s->vbox[b].hi[1] - s->vbox[b].lo[1] + 1) * (s->vbox[b].hi[2] - s->vbox[b].lo[2] + 1);
  return !(s->nfabs() <= 16 && cells / s->nfabs() >= 96L * 96 * 96);
}
bool god_per_box(const vdn_multifab *s) { return s->nfabs() >= 1 && !use_batched(s); }
static bool plain_godunov() { static const bool p = vdn_env("VDN_GODUNOV_PLAIN") != nullptr; return p; }
// parameters of the marching bodies: by value under the by-value kernels, references into the (constant) descriptor under the batched ones
template <class T, bool R> struct Prm { typedef T type; };
template <class T> struct Prm<T, true> { typedef const T &type; };
// tile order of the marching kernels: XCD-aware (vdn_dev.h xcd_tile)
__constant__ int g_god_xcd = 1;
DEVI void xcd_remap(int &bx, int &by, int &bz) { if (g_god_xcd) xcd_tile(bx, by, bz); else { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; } }
static void god_xcd_init() {
  static bool done = false;
  if (done) return;
  done = true;
}
constexpr int TNY = 8;              // rows per tile: workgroup = 64 x TNY threads
static int march_chunks() { return 12; }
static dim3 march_grid(const Range3 &r, int &klen) {
  const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
  klen = (nz + march_chunks() - 1) / march_chunks(); if (klen < 1) klen = 1;
  return dim3((nx + 62) / 63, (ny + TNY - 2) / (TNY - 1), (nz + klen - 1) / klen);
}
// i, j: the thread's cell;  ic, jc: the same clamped into the grown box (load indices);  own_ij: this thread emits
#define MARCH_SETUP(r)                                                                   \
  const int lane = threadIdx.x, row = threadIdx.y;                                       \
  const int i = (r).lo[0] - 1 + BX * 63 + lane;                                                                 \
  const int j = (r).lo[1] - 1 + BY * (TNY - 1) + row;                                     \
  const bool own_ij = lane >= 1 && row >= 1 && i <= (r).hi[0] && j <= (r).hi[1];         \
  const int ic = min(max(i, A.lo[0] - 1), A.hi[0] + 1), jc = min(max(j, A.lo[1] - 1), A.hi[1] + 1); \
  const int ip = min(ic + 1, A.hi[0] + 1), jp = min(jc + 1, A.hi[1] + 1);                \
  const bool edge_ij = ic == A.lo[0] || ic >= A.hi[0] || jc == A.lo[1] || jc >= A.hi[1]; \
  const bool vx = in_valid(A, 0, i), vy = in_valid(A, 1, j);                             \
  const int k0 = (r).lo[2] + BZ * klen, k1 = min(k0 + klen - 1, (r).hi[2]);
#define MARCH_PLANE                                                                      \
    const int kc = min(max(k, A.lo[2] - 1), A.hi[2] + 1), kp = min(kc + 1, A.hi[2] + 1); \
    const bool emit = own_ij && k >= k0;                                                 \
    const bool vz = in_valid(A, 2, k);                                                   \
    const bool edge = edge_ij || kc == A.lo[2] || kc >= A.hi[2];                         \
    const int buf = k & 1;                                                               \
    (void)kp; (void)vz; (void)edge; (void)ip; (void)jp; (void)vx; (void)vy;

DEVI double shfl_prev(double v) { return lane_prev(v); }
DEVI bool in_valid(const GArgs &A, int d, int q) { return q >= A.lo[d] && q <= A.hi[d]; }
DEVI double tv(bool cons, double fcons, double fconv, double dxT, double sp, double s0, double mp, double m0) {   // = trans_term on values
  if (cons) return (fcons / dxT) * (sp * mp - s0 * m0);
  return (fconv / dxT) * (mp + m0) * (sp - s0);
}

// one-dimensional predictor bases of component c in a cell along D: Lb = left state of the cell's UPPER D-face,
// Rb = right state of its LOWER D-face (mk_pair without the boundary rule)
template <int D> DEVI void mk_bases(const GArgs &A, int c, double s0, double slD, double mlo, double mup, double fterm, double mterm, double &Lb, double &Rb) {
  const double dt2 = 0.5 * A.dt;
  Lb = s0 + (0.5 - dt2 * mup / A.dx[D]) * slD;
  Rb = s0 - (0.5 + dt2 * mlo / A.dx[D]) * slD;
  if (A.use_minion) {
    Lb = Lb + fterm; Rb = Rb + fterm;
    if (A.cons[c]) { Lb = Lb - mterm; Rb = Rb - mterm; }
  }
}
// this cell's half of the boundary rule on its bases (cells next to a box face only)
template <int D> DEVI void mk_premod(const GArgs &A, const FV &s, int c, int i, int j, int k, double &Lb, double &Rb) {
  const int q = coord<D>(i, j, k);
  if (q == A.lo[D]) { double o = Rb; bc_pair(o, Rb, A.phys[D][0], 0, A.is_vel != 0, c == D, ld<D>(s, i, j, k, -1, c), false); }
  if (q == A.hi[D]) { double o = Lb; bc_pair(Lb, o, A.phys[D][1], 1, A.is_vel != 0, c == D, ld<D>(s, i, j, k, 1, c), false); }
}
// the complete boundary rule on an assembled pair of the lower D-face of cell (i,j,k)
template <int D> DEVI void mk_face_bc(const GArgs &A, const FV &s, int c, int i, int j, int k, double s0, double &L, double &R) {
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) bc_pair(L, R, A.phys[D][side], side, A.is_vel != 0, c == D, side == 0 ? ld<D>(s, i, j, k, -1, c) : s0, false);
}

// the per-plane load batch shared by the three mkflux stages
#define MK_LOAD_CELL                                                                                                   \
    const double m_lo[3] = { fv_get(um, ic, jc, kc), fv_get(vm, ic, jc, kc), fv_get(wm, ic, jc, kc) };                   \
    const double m_up[3] = { fv_get(um, ic + 1, jc, kc), fv_get(vm, ic, jc + 1, kc), fv_get(wm, ic, jc, kc + 1) };       \
    double s0[NC], sl[NC][3];                                                                                           \
    _Pragma("unroll") for (int c = 0; c < NC; c++) {                                                                    \
      s0[c] = fv_get(s, ic, jc, kc, c0 + c);                                                                                  \
      sl[c][0] = fv_get(sl0, ic, jc, kc, c0 + c); sl[c][1] = fv_get(sl1, ic, jc, kc, c0 + c); sl[c][2] = fv_get(sl2, ic, jc, kc, c0 + c); \
    }

// ---- stage B ----------------------------------------------------------------------------------------------------
template <int NC, bool R, bool BC> __device__ __forceinline__ void mk_B_m_body(typename Prm<FV, R>::type s, typename Prm<FV, R>::type sl0, typename Prm<FV, R>::type sl1, typename Prm<FV, R>::type sl2, typename Prm<FV, R>::type um, typename Prm<FV, R>::type vm, typename Prm<FV, R>::type wm, typename Prm<FV, R>::type force, typename Prm<FV, R>::type macrhs, typename Prm<FV, R>::type SI, typename Prm<GArgs, R>::type A, typename Prm<Range3, R>::type r, int klen, const double *umax, int c0, int ns, const int BX, const int BY, const int BZ) {
  __shared__ double ly[2][NC][TNY][64];
  MARCH_SETUP(r)
  const double eps = eps_from(umax);
  const double dt2 = 0.5 * A.dt;
  double Lz[NC];
  #pragma unroll
  for (int c = 0; c < NC; c++) Lz[c] = 0.0;
  for (int k = k0 - 1; k <= k1; k++) {
    MARCH_PLANE
    MK_LOAD_CELL
    double ft[NC], mt[NC];
    #pragma unroll
    for (int c = 0; c < NC; c++) { ft[c] = 0.0; mt[c] = 0.0; }
    if (A.use_minion) {
      const double mr = fv_get(macrhs, ic, jc, kc);
      #pragma unroll
      for (int c = 0; c < NC; c++) { ft[c] = dt2 * fv_get(force, ic, jc, kc, c0 + c); mt[c] = dt2 * s0[c] * mr; }
    }
    double Lb[NC][3], Rb[NC][3];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      mk_bases<0>(A, c0 + c, s0[c], sl[c][0], m_lo[0], m_up[0], ft[c], mt[c], Lb[c][0], Rb[c][0]);
      mk_bases<1>(A, c0 + c, s0[c], sl[c][1], m_lo[1], m_up[1], ft[c], mt[c], Lb[c][1], Rb[c][1]);
      mk_bases<2>(A, c0 + c, s0[c], sl[c][2], m_lo[2], m_up[2], ft[c], mt[c], Lb[c][2], Rb[c][2]);
    }
    // (no premod needed in stage B: the owner's complete rule is applied to the untouched pair)
    #pragma unroll
    for (int c = 0; c < NC; c++) ly[buf][c][row][lane] = Lb[c][1];
    __syncthreads();
    double Lx[NC], Ly[NC], Lzc[NC];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      Lx[c] = shfl_prev(Lb[c][0]);
      Ly[c] = ly[buf][c][row >= 1 ? row - 1 : 0][lane];
      Lzc[c] = Lz[c]; Lz[c] = Lb[c][2];
    }
    if (emit) {
      if (BC && edge) {
        #pragma unroll
        for (int c = 0; c < NC; c++) {
          mk_face_bc<0>(A, s, c0 + c, i, j, k, s0[c], Lx[c], Rb[c][0]); mk_face_bc<1>(A, s, c0 + c, i, j, k, s0[c], Ly[c], Rb[c][1]); mk_face_bc<2>(A, s, c0 + c, i, j, k, s0[c], Lzc[c], Rb[c][2]);
        }
      }
      #pragma unroll
      for (int c = 0; c < NC; c++) {
        if (i >= A.lo[0]) fv_at(SI, i, j, k, 0 * ns + c0 + c) = upwind_mac(Lx[c], Rb[c][0], m_lo[0], eps);
        if (j >= A.lo[1]) fv_at(SI, i, j, k, 1 * ns + c0 + c) = upwind_mac(Ly[c], Rb[c][1], m_lo[1], eps);
        if (k >= A.lo[2]) fv_at(SI, i, j, k, 2 * ns + c0 + c) = upwind_mac(Lzc[c], Rb[c][2], m_lo[2], eps);
      }
    }
  }
}
template <int NC, bool BC = true> __global__ void __launch_bounds__(64 * TNY) kk_mk_B_m(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SI, GArgs A, Range3 r, int klen, const double *umax, int c0, int ns) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  mk_B_m_body<NC, false, BC>(s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SI, A, r, klen, umax, c0, ns, bx_, by_, bz_);
}


// ---- stage C ----------------------------------------------------------------------------------------------------
// per cell and component: the three transverse terms t_T (from SI_T, mac_T) and the six chains base_D - t_T
template <int NC, bool R, bool BC> __device__ __forceinline__ void mk_C_m_body(typename Prm<FV, R>::type s, typename Prm<FV, R>::type sl0, typename Prm<FV, R>::type sl1, typename Prm<FV, R>::type sl2, typename Prm<FV, R>::type um, typename Prm<FV, R>::type vm, typename Prm<FV, R>::type wm, typename Prm<FV, R>::type force, typename Prm<FV, R>::type macrhs, typename Prm<FV, R>::type SI, typename Prm<FV, R>::type SC, typename Prm<GArgs, R>::type A, typename Prm<Range3, R>::type r, int klen, const double *umax, int c0, int ns, const int BX, const int BY, const int BZ) {
  __shared__ double ly[2][NC][2][TNY][64];
  MARCH_SETUP(r)
  const double eps = eps_from(umax);
  const double dt2 = 0.5 * A.dt, dt3 = A.dt / 3.0, dt6 = A.dt / 6.0;
  double Lz[NC][2];
  #pragma unroll
  for (int c = 0; c < NC; c++) { Lz[c][0] = 0.0; Lz[c][1] = 0.0; }
  for (int k = k0 - 1; k <= k1; k++) {
    MARCH_PLANE
    MK_LOAD_CELL
    double si0[NC][3], si1[NC][3];          // SI_T at the cell's lower / upper T-face
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      si0[c][0] = fv_get(SI, ic, jc, kc, 0 * ns + c0 + c); si1[c][0] = fv_get(SI, ip, jc, kc, 0 * ns + c0 + c);
      si0[c][1] = fv_get(SI, ic, jc, kc, 1 * ns + c0 + c); si1[c][1] = fv_get(SI, ic, jp, kc, 1 * ns + c0 + c);
      si0[c][2] = fv_get(SI, ic, jc, kc, 2 * ns + c0 + c); si1[c][2] = fv_get(SI, ic, jc, kp, 2 * ns + c0 + c);
    }
    double ft[NC], mt[NC];
    #pragma unroll
    for (int c = 0; c < NC; c++) { ft[c] = 0.0; mt[c] = 0.0; }
    if (A.use_minion) {
      const double mr = fv_get(macrhs, ic, jc, kc);
      #pragma unroll
      for (int c = 0; c < NC; c++) { ft[c] = dt2 * fv_get(force, ic, jc, kc, c0 + c); mt[c] = dt2 * s0[c] * mr; }
    }
    double Lb[NC][3], Rb[NC][3];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      mk_bases<0>(A, c0 + c, s0[c], sl[c][0], m_lo[0], m_up[0], ft[c], mt[c], Lb[c][0], Rb[c][0]);
      mk_bases<1>(A, c0 + c, s0[c], sl[c][1], m_lo[1], m_up[1], ft[c], mt[c], Lb[c][1], Rb[c][1]);
      mk_bases<2>(A, c0 + c, s0[c], sl[c][2], m_lo[2], m_up[2], ft[c], mt[c], Lb[c][2], Rb[c][2]);
    }
    if (BC && edge) {
      #pragma unroll
      for (int c = 0; c < NC; c++) { mk_premod<0>(A, s, c0 + c, ic, jc, kc, Lb[c][0], Rb[c][0]); mk_premod<1>(A, s, c0 + c, ic, jc, kc, Lb[c][1], Rb[c][1]); mk_premod<2>(A, s, c0 + c, ic, jc, kc, Lb[c][2], Rb[c][2]); }
    }
    // VL[c][D][n] / VR[c][D][n]: n = 0,1 -> the two transverse directions of D in increasing order
    double VL[NC][3][2], VR[NC][3][2];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      const bool cons = A.cons[c0 + c] != 0;
      double t[3];
      t[0] = tv(cons, dt3, dt6, A.dx[0], si1[c][0], si0[c][0], m_up[0], m_lo[0]);
      t[1] = tv(cons, dt3, dt6, A.dx[1], si1[c][1], si0[c][1], m_up[1], m_lo[1]);
      t[2] = tv(cons, dt3, dt6, A.dx[2], si1[c][2], si0[c][2], m_up[2], m_lo[2]);
      // D = 0: T = 1, 2;  D = 1: T = 0, 2;  D = 2: T = 0, 1
      VL[c][0][0] = Lb[c][0] - t[1]; VR[c][0][0] = Rb[c][0] - t[1]; VL[c][0][1] = Lb[c][0] - t[2]; VR[c][0][1] = Rb[c][0] - t[2];
      VL[c][1][0] = Lb[c][1] - t[0]; VR[c][1][0] = Rb[c][1] - t[0]; VL[c][1][1] = Lb[c][1] - t[2]; VR[c][1][1] = Rb[c][1] - t[2];
      VL[c][2][0] = Lb[c][2] - t[0]; VR[c][2][0] = Rb[c][2] - t[0]; VL[c][2][1] = Lb[c][2] - t[1]; VR[c][2][1] = Rb[c][2] - t[1];
    }
    #pragma unroll
    for (int c = 0; c < NC; c++) { ly[buf][c][0][row][lane] = VL[c][1][0]; ly[buf][c][1][row][lane] = VL[c][1][1]; }
    __syncthreads();
    double Lx[NC][2], Ly[NC][2], Lzc[NC][2];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      Lx[c][0] = shfl_prev(VL[c][0][0]); Lx[c][1] = shfl_prev(VL[c][0][1]);
      Ly[c][0] = ly[buf][c][0][row >= 1 ? row - 1 : 0][lane]; Ly[c][1] = ly[buf][c][1][row >= 1 ? row - 1 : 0][lane];
      Lzc[c][0] = Lz[c][0]; Lzc[c][1] = Lz[c][1];
      Lz[c][0] = VL[c][2][0]; Lz[c][1] = VL[c][2][1];
    }
    if (emit) {
      if (BC && edge) {
        #pragma unroll
        for (int c = 0; c < NC; c++) {
          #pragma unroll
          for (int n = 0; n < 2; n++) {
            mk_face_bc<0>(A, s, c0 + c, i, j, k, s0[c], Lx[c][n], VR[c][0][n]); mk_face_bc<1>(A, s, c0 + c, i, j, k, s0[c], Ly[c][n], VR[c][1][n]); mk_face_bc<2>(A, s, c0 + c, i, j, k, s0[c], Lzc[c][n], VR[c][2][n]);
          }
        }
      }
      // SC index (D,T): (D*2 + (T > D ? T-1 : T)) * NC + c;  valid where the D-face index >= lo[D] and the T index is valid
      #pragma unroll
      for (int c = 0; c < NC; c++) {
        if (i >= A.lo[0]) {
          if (vy) fv_at(SC, i, j, k, 0 * ns + c0 + c) = upwind_mac(Lx[c][0], VR[c][0][0], m_lo[0], eps);
          if (vz) fv_at(SC, i, j, k, 1 * ns + c0 + c) = upwind_mac(Lx[c][1], VR[c][0][1], m_lo[0], eps);
        }
        if (j >= A.lo[1]) {
          if (vx) fv_at(SC, i, j, k, 2 * ns + c0 + c) = upwind_mac(Ly[c][0], VR[c][1][0], m_lo[1], eps);
          if (vz) fv_at(SC, i, j, k, 3 * ns + c0 + c) = upwind_mac(Ly[c][1], VR[c][1][1], m_lo[1], eps);
        }
        if (k >= A.lo[2]) {
          if (vx) fv_at(SC, i, j, k, 4 * ns + c0 + c) = upwind_mac(Lzc[c][0], VR[c][2][0], m_lo[2], eps);
          if (vy) fv_at(SC, i, j, k, 5 * ns + c0 + c) = upwind_mac(Lzc[c][1], VR[c][2][1], m_lo[2], eps);
        }
      }
    }
  }
}
template <int NC, bool BC = true> __global__ void __launch_bounds__(64 * TNY) kk_mk_C_m(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SI, FV SC, GArgs A, Range3 r, int klen, const double *umax, int c0, int ns) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  mk_C_m_body<NC, false, BC>(s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SI, SC, A, r, klen, umax, c0, ns, bx_, by_, bz_);
}


// ---- stage D ----------------------------------------------------------------------------------------------------
template <int D> DEVI double mk_edge_bc(const GArgs &A, const FV &s, int c, int i, int j, int k, double s0, double L, double R, double e) {
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) {                                   // mkflux.f90:2369-2402
    const int ph = A.phys[D][side];
    const double in = (side == 0) ? R : L;
    const bool vel = A.is_vel != 0;
    if (ph == VDN_INLET) e = (side == 0) ? ld<D>(s, i, j, k, -1, c) : s0;
    else if (ph == VDN_SLIP_WALL) e = (vel && c == D) ? 0.0 : in;
    else if (ph == VDN_NO_SLIP_WALL) e = vel ? 0.0 : in;
    else if (ph == VDN_OUTLET) e = (vel && c == D) ? ((side == 0) ? fmin(in, 0.0) : fmax(in, 0.0)) : in;
  }
  return e;
}
// the per-plane inputs of stage D for one cell
//   q0[c][n] / q1[c][n]: SC component n at the cell's lower / upper face of the direction that component lives on
//   n: 0 = (0,1) 1 = (0,2) on x-faces, 2 = (1,0) 3 = (1,2) on y-faces, 4 = (2,0) 5 = (2,1) on z-faces
template <int NC> struct DPlane { double m_lo[3], m_up[3], s0[NC], sl[NC][3], mr, f[NC], q0[NC][6], q1[NC][6]; };
template <int NC> DEVI void d_load_plane(DPlane<NC> &P, const FV &s, const FV &sl0, const FV &sl1, const FV &sl2, const FV &um, const FV &vm, const FV &wm,
                                         const FV &force, const FV &macrhs, const FV &SC, const GArgs &A, int ic, int jc, int ip, int jp, int k, int c0, int ns) {
  const int kc = min(max(k, A.lo[2] - 1), A.hi[2] + 1), kp = min(kc + 1, A.hi[2] + 1);
  P.m_lo[0] = fv_get(um, ic, jc, kc); P.m_lo[1] = fv_get(vm, ic, jc, kc); P.m_lo[2] = fv_get(wm, ic, jc, kc);
  P.m_up[0] = fv_get(um, ic + 1, jc, kc); P.m_up[1] = fv_get(vm, ic, jc + 1, kc); P.m_up[2] = fv_get(wm, ic, jc, kc + 1);
  P.mr = fv_get(macrhs, ic, jc, kc);
  #pragma unroll
  for (int c = 0; c < NC; c++) {
    P.s0[c] = fv_get(s, ic, jc, kc, c0 + c);
    P.sl[c][0] = fv_get(sl0, ic, jc, kc, c0 + c); P.sl[c][1] = fv_get(sl1, ic, jc, kc, c0 + c); P.sl[c][2] = fv_get(sl2, ic, jc, kc, c0 + c);
    P.f[c] = fv_get(force, ic, jc, kc, c0 + c);
    #pragma unroll
    for (int n = 0; n < 6; n++) P.q0[c][n] = fv_get(SC, ic, jc, kc, n * ns + c0 + c);
    P.q1[c][0] = fv_get(SC, ip, jc, kc, 0 * ns + c0 + c); P.q1[c][1] = fv_get(SC, ip, jc, kc, 1 * ns + c0 + c);
    P.q1[c][2] = fv_get(SC, ic, jp, kc, 2 * ns + c0 + c); P.q1[c][3] = fv_get(SC, ic, jp, kc, 3 * ns + c0 + c);
    P.q1[c][4] = fv_get(SC, ic, jc, kp, 4 * ns + c0 + c); P.q1[c][5] = fv_get(SC, ic, jc, kp, 5 * ns + c0 + c);
  }
}
template <int NC, bool R, bool PF, bool BC> __device__ __forceinline__ void mk_D_m_body(typename Prm<FV, R>::type s, typename Prm<FV, R>::type sl0, typename Prm<FV, R>::type sl1, typename Prm<FV, R>::type sl2, typename Prm<FV, R>::type um, typename Prm<FV, R>::type vm, typename Prm<FV, R>::type wm, typename Prm<FV, R>::type force, typename Prm<FV, R>::type macrhs, typename Prm<FV, R>::type SC, typename Prm<FV, R>::type sex, typename Prm<FV, R>::type sey, typename Prm<FV, R>::type sez, typename Prm<FV, R>::type flx, typename Prm<FV, R>::type fly, typename Prm<FV, R>::type flz, typename Prm<GArgs, R>::type A, typename Prm<Range3, R>::type r, int klen, const double *umax, int c0, int ns, const int BX, const int BY, const int BZ) {
  __shared__ double ly[2][NC][TNY][64];
  MARCH_SETUP(r)
  const double eps = eps_from(umax);
  const double dt2 = 0.5 * A.dt, dt4 = A.dt / 4.0;
  double Lz[NC];
  #pragma unroll
  for (int c = 0; c < NC; c++) Lz[c] = 0.0;
  DPlane<NC> Pn;
  if (PF) d_load_plane<NC>(Pn, s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SC, A, ic, jc, ip, jp, k0 - 1, c0, ns);
  for (int k = k0 - 1; k <= k1; k++) {
    MARCH_PLANE
    // the plane's loads: with PF they were issued one step ahead (while the previous plane was being computed), so that a workgroup
    // keeps two planes of loads in flight -- at 2 waves per SIMD the marches are bound by load latency, not by bandwidth
    DPlane<NC> P;
    if (PF) { P = Pn; if (k < k1) d_load_plane<NC>(Pn, s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SC, A, ic, jc, ip, jp, k + 1, c0, ns); }
    else d_load_plane<NC>(P, s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SC, A, ic, jc, ip, jp, k, c0, ns);
    const double (&m_lo)[3] = P.m_lo, (&m_up)[3] = P.m_up; const double (&s0)[NC] = P.s0; const double (&sl)[NC][3] = P.sl;
    const double (&q0)[NC][6] = P.q0, (&q1)[NC][6] = P.q1;
    const double mr = P.mr;
    double ft[NC], mt[NC];
    #pragma unroll
    for (int c = 0; c < NC; c++) { ft[c] = dt2 * P.f[c]; mt[c] = dt2 * s0[c] * mr; }
    double Lb[NC][3], Rb[NC][3];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      mk_bases<0>(A, c0 + c, s0[c], sl[c][0], m_lo[0], m_up[0], ft[c], mt[c], Lb[c][0], Rb[c][0]);
      mk_bases<1>(A, c0 + c, s0[c], sl[c][1], m_lo[1], m_up[1], ft[c], mt[c], Lb[c][1], Rb[c][1]);
      mk_bases<2>(A, c0 + c, s0[c], sl[c][2], m_lo[2], m_up[2], ft[c], mt[c], Lb[c][2], Rb[c][2]);
    }
    if (BC && edge) {
      #pragma unroll
      for (int c = 0; c < NC; c++) { mk_premod<0>(A, s, c0 + c, ic, jc, kc, Lb[c][0], Rb[c][0]); mk_premod<1>(A, s, c0 + c, ic, jc, kc, Lb[c][1], Rb[c][1]); mk_premod<2>(A, s, c0 + c, ic, jc, kc, Lb[c][2], Rb[c][2]); }
    }
    double VL[NC][3], VR[NC][3];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      const bool cons = A.cons[c0 + c] != 0;
      // a_T = (dt2/dx_T) * s0 * (mac_T(+) - mac_T)
      const double a[3] = { (dt2 / A.dx[0]) * s0[c] * (m_up[0] - m_lo[0]), (dt2 / A.dx[1]) * s0[c] * (m_up[1] - m_lo[1]), (dt2 / A.dx[2]) * s0[c] * (m_up[2] - m_lo[2]) };
      #define CHAIN(Dd, T1, T2, n1, n2)                                                                    \
        { const double t1 = tv(cons, dt2, dt4, A.dx[T1], q1[c][n1], q0[c][n1], m_up[T1], m_lo[T1]);          \
          const double t2 = tv(cons, dt2, dt4, A.dx[T2], q1[c][n2], q0[c][n2], m_up[T2], m_lo[T2]);          \
          double vl = Lb[c][Dd], vr = Rb[c][Dd];                                                             \
          vl = vl - t1; vr = vr - t1; vl = vl - t2; vr = vr - t2;                                            \
          if (cons) { vl = vl + a[T1]; vr = vr + a[T1]; vl = vl + a[T2]; vr = vr + a[T2]; }                  \
          if (!A.use_minion) { vl = vl + ft[c]; vr = vr + ft[c]; if (cons) { vl = vl - mt[c]; vr = vr - mt[c]; } } \
          VL[c][Dd] = vl; VR[c][Dd] = vr; }
      CHAIN(0, 1, 2, 3, 5)      // D = 0: T1 = 1 with SC(1,2) = simhyz,  T2 = 2 with SC(2,1) = simhzy
      CHAIN(1, 0, 2, 1, 4)      // D = 1: T1 = 0 with SC(0,2),           T2 = 2 with SC(2,0)
      CHAIN(2, 0, 1, 0, 2)      // D = 2: T1 = 0 with SC(0,1),           T2 = 1 with SC(1,0)
      #undef CHAIN
    }
    #pragma unroll
    for (int c = 0; c < NC; c++) ly[buf][c][row][lane] = VL[c][1];
    __syncthreads();
    double L[NC][3];
    #pragma unroll
    for (int c = 0; c < NC; c++) {
      L[c][0] = shfl_prev(VL[c][0]);
      L[c][1] = ly[buf][c][row >= 1 ? row - 1 : 0][lane];
      L[c][2] = Lz[c]; Lz[c] = VL[c][2];
    }
    if (emit) {
      double e[NC][3];
      #pragma unroll
      for (int c = 0; c < NC; c++) { e[c][0] = upwind_mac(L[c][0], VR[c][0], m_lo[0], eps); e[c][1] = upwind_mac(L[c][1], VR[c][1], m_lo[1], eps); e[c][2] = upwind_mac(L[c][2], VR[c][2], m_lo[2], eps); }
      if (BC && edge) {
        #pragma unroll
        for (int c = 0; c < NC; c++) {
          e[c][0] = mk_edge_bc<0>(A, s, c0 + c, i, j, k, s0[c], L[c][0], VR[c][0], e[c][0]);
          e[c][1] = mk_edge_bc<1>(A, s, c0 + c, i, j, k, s0[c], L[c][1], VR[c][1], e[c][1]);
          e[c][2] = mk_edge_bc<2>(A, s, c0 + c, i, j, k, s0[c], L[c][2], VR[c][2], e[c][2]);
        }
      }
      #pragma unroll
      for (int c = 0; c < NC; c++) {
        const bool cons = A.cons[c0 + c] != 0;
        if (vy && vz) { fv_at(sex, i, j, k, c0 + c) = e[c][0]; if (cons) fv_at(flx, i, j, k, c0 + c) = e[c][0] * m_lo[0]; }
        if (vx && vz) { fv_at(sey, i, j, k, c0 + c) = e[c][1]; if (cons) fv_at(fly, i, j, k, c0 + c) = e[c][1] * m_lo[1]; }
        if (vx && vy) { fv_at(sez, i, j, k, c0 + c) = e[c][2]; if (cons) fv_at(flz, i, j, k, c0 + c) = e[c][2] * m_lo[2]; }
      }
    }
  }
}
// (measured and rejected: capping the one-component kernel at 128 VGPRs -- 11 spilled -- so that two workgroups share a CU: 0.82 -> 1.04 ms)
template <int NC, bool PF = false, bool BC = true> __global__ void __launch_bounds__(64 * TNY) kk_mk_D_m(FV s, FV sl0, FV sl1, FV sl2, FV um, FV vm, FV wm, FV force, FV macrhs, FV SC, FV sex, FV sey, FV sez, FV flx, FV fly, FV flz, GArgs A, Range3 r, int klen, const double *umax, int c0, int ns) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  mk_D_m_body<NC, false, PF, BC>(s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SC, sex, sey, sez, flx, fly, flz, A, r, klen, umax, c0, ns, bx_, by_, bz_);
}


// ---- stages B + C + D in one march (one component per launch) ------------------------------------------------------------------
// The three stages above exchange SI (3 fields per component) and SC (6 per component) through HBM: at 256^3 the velocity mkflux moves
// 18 GB against 2.5 GB of algorithmic traffic (profiles/r02_godunov_pmc.md).  Here a workgroup keeps them in registers: iteration kk of
// the march loads plane kk and runs stage B on plane kk, stage C on plane kk-1 (its upper z-face SI is B's output of this iteration) and
// stage D on plane kk-2 (its upper z-face SC is C's output of this iteration).  A cell's upper x-face value is the next lane's lower-face
// value (DPP), its upper y-face value the next row's (LDS, written one iteration earlier), its own left states come from the previous
// lane / row / plane exactly as in the separate stages.  A tile therefore owns lanes 1..62 and rows 1..TNY-2: the transverse couplings
// alternate direction from stage to stage, so the footprint of an output face is one cell either side, not one per stage.
// Every expression is the one of mk_B_m_body / mk_C_m_body / mk_D_m_body in the same order: bit-identical results
// (tests/test_kernels_gpu.py::test_mkflux, test_godunov_marching_equals_face_centred).  Reads per component: s, three slopes, three MAC
// components (+ their upper faces), force, mac_rhs; writes: three edge states (+ fluxes of conservative components).
// Memory access: the launch carries, per field, the address of plane KB = lo_z - 1 of the component (bytes) and the bytes per plane; a thread
// keeps ONE 32-bit byte offset per field layout for its (i,j) column and the workgroup advances uniform plane pointers (scalar registers)
// as the march proceeds -- no per-load index arithmetic (15 fields x fv_idx per plane cost ~150 VALU / ~180 SALU instructions per plane in
// the first version, and the 40-byte FV descriptors 170 spilled SGPRs).  The quotients of launch constants that the separate stages
// compute per cell -- dt/3/dx, dt/6/dx, ... -- are evaluated once on the host (the same IEEE operations on the same operands).
struct FGeo { int a0, a1, n0; };
struct FArgs {
  const char *p[9];       // s, sl0, sl1, sl2, um, vm, wm, force, macrhs: plane KB of the component
  long sp[7];             // bytes per plane: s, slopes, um, vm, wm, force, macrhs
  FGeo g[7];
  char *q[6];             // sex, sey, sez, flx, fly, flz
  long sq[3]; FGeo h[3];  // x-, y-, z-face outputs (edge state and flux of one direction share a layout)
  long s_row, vm_row;     // bytes per row of s / vm
  double dt2, dx[3], tC[3], tD[3], aD[3];   // tC[T] = (cons ? dt/3 : dt/6) / dx[T],  tD[T] = (cons ? dt/2 : dt/4) / dx[T],  aD[T] = (dt/2) / dx[T]
  double idx[3]; int p2;                    // 1 / dx and "every dx is a power of two": x / dx is then the exact scaling x * (1 / dx), bit for bit (P2 kernels)
  int lo[3], hi[3], phys[3][2];
  int cons, use_minion, is_vel, c;
  // UPD (round 3): update_3d of the component inside the march -- the edge states never reach memory
  char *qn; long sqn; FGeo hn;      // snew / unew of the component: plane KB, bytes per plane, layout
  const char *pfu[3]; int fmode;    // the update's forcing term: fmode 0 = the force of this call (p[7]); 1 = ext (pfu[0]) + (lapu0 - gp (pfu[1])) / rho (pfu[2]), force layout
  double dt, lapu0;
};
static bool bc_mode_host(int phys) { return phys == VDN_INLET || phys == VDN_SLIP_WALL || phys == VDN_NO_SLIP_WALL || phys == VDN_OUTLET; }
static bool same_geom(const FV &a, const FV &b) { return a.a0 == b.a0 && a.a1 == b.a1 && a.a2 == b.a2 && a.n0 == b.n0 && a.n1 == b.n1; }
DEVI unsigned fg_off(const FGeo &G, int i, int j) { return 8u * (unsigned)((i - G.a0) + G.n0 * (j - G.a1)); }
DEVI double ldd(const char *q, unsigned o) { return *(const double *)(q + o); }
DEVI void std_(char *q, unsigned o, double v) { *(double *)(q + o) = v; }
struct FCell { double m_lo[3], m_up[3], s0, f, mr, Lb[3], Rb[3]; };      // a cell's loads and its bases (before any boundary rule)
// mk_bases with the launch constants of FArgs: the bases of stage B (force / mac_rhs terms inside only with use_minion); stages C and D of
// the separate kernels recompute the same ones -- six f64 divisions per cell and stage -- here they ride along with the cell
// A division by a power of two is an exact scaling: x / dx and x * (1 / dx) are the same double for every x (zeros, infinities and subnormal
// results included: both are the correctly rounded value of the same real number), so where every dx is a power of two -- the unit cube on
// 2^n cells, every level of a hierarchy over it -- the P2 kernels multiply.  The ten f64 divisions per cell and plane of the fused march were a
// third of its f64 instructions (v_div_scale x 2, v_rcp, eight fma, v_div_fmas, v_div_fixup each).  Other spacings keep the division.
static bool no_p2() { static const bool off = vdn_env("VDN_GOD_P2") && atoi(vdn_env("VDN_GOD_P2")) == 0; return off; }      // (the variants test: division path on power-of-two grids)
static bool is_pow2(double x) { int e; return x > 0.0 && std::frexp(x, &e) == 0.5; }
#define DIVDX(x, d) (PW2 ? (x) * F.idx[d] : (x) / F.dx[d])
template <bool PW2> DEVI void f_bases(const FArgs &F, FCell &P, const double sl[3]) {
  double ft = 0.0, mt = 0.0;
  if (F.use_minion) { ft = F.dt2 * P.f; mt = F.dt2 * P.s0 * P.mr; }
  #pragma unroll
  for (int d = 0; d < 3; d++) {
    double Lb = P.s0 + (0.5 - DIVDX(F.dt2 * P.m_up[d], d)) * sl[d];
    double Rb = P.s0 - (0.5 + DIVDX(F.dt2 * P.m_lo[d], d)) * sl[d];
    if (F.use_minion) {
      Lb = Lb + ft; Rb = Rb + ft;
      if (F.cons) { Lb = Lb - mt; Rb = Rb - mt; }
    }
    P.Lb[d] = Lb; P.Rb[d] = Rb;
  }
}
DEVI double tvq(bool cons, double q, double sp, double s0, double mp, double m0) {     // tv with its quotient (fcons / dxT or fconv / dxT) given
  if (cons) return q * (sp * mp - s0 * m0);
  return q * (mp + m0) * (sp - s0);
}
static bool god_oneb() { static const bool on = !(vdn_env("VDN_GOD_1B") && atoi(vdn_env("VDN_GOD_1B")) == 0); return on; }
constexpr int FNX = 62, FNY = TNY - 2;      // cells a tile owns per row / rows it owns
static dim3 fused_grid(const Range3 &r, int &klen) {
  // one 512-thread workgroup per CU at a time: the chunk count is the one that fills the last round of workgroups best
  const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
  const int tiles = ((nx + FNX - 1) / FNX) * ((ny + FNY - 1) / FNY);
  static const int env = vdn_env("VDN_FUSED_KCHUNKS") ? std::max(1, atoi(vdn_env("VDN_FUSED_KCHUNKS"))) : 0;
  int best = 1; double best_cost = 1e300;
  for (int ch = 1; ch <= 16 && ch <= nz; ch++) {
    const int kl = (nz + ch - 1) / ch, nch = (nz + kl - 1) / kl;
    const double cost = std::ceil((double)tiles * nch / 256.0) * (kl + 4);      // rounds x iterations per workgroup
    if (cost < best_cost) { best_cost = cost; best = ch; }
  }
  const int chunks = env ? env : best;
  klen = (nz + chunks - 1) / chunks; if (klen < 1) klen = 1;
  return dim3((nx + FNX - 1) / FNX, (ny + FNY - 1) / FNY, (nz + klen - 1) / klen);
}
// the grid of a fused march whose remainder tile column runs in narrow segments (kk_mk_F_mc, kk_vp_F_mc): `full` 62-cell tile columns + one column of
// segments of segw lanes; false when the box has no remainder column worth it.  The chunk count is chosen for the workgroups that do work.
static bool fused_grid_cols(const Range3 &rf, dim3 &g, int &klen, int &full, int &segw) {
  static const bool narrow_env = !(vdn_env("VDN_GOD_NARROW") && atoi(vdn_env("VDN_GOD_NARROW")) == 0);
  const int nx = rf.hi[0] - rf.lo[0] + 1, ny = rf.hi[1] - rf.lo[1] + 1, nz = rf.hi[2] - rf.lo[2] + 1;
  full = nx / FNX;
  const int rem = nx - full * FNX;
  if (!narrow_env || full < 1 || rem <= 0 || rem + 2 > 32) return false;
  segw = rem + 2;
  const int rows = TNY * (64 / segw) - 2;
  const int tiles = full * ((ny + FNY - 1) / FNY) + (ny + rows - 1) / rows;
  int best = 1; double best_cost = 1e300;
  for (int ch = 1; ch <= 16 && ch <= nz; ch++) {
    const int kl = (nz + ch - 1) / ch, nch = (nz + kl - 1) / kl;
    const double cost = std::ceil((double)tiles * nch / 256.0) * (kl + 4);
    if (cost < best_cost) { best_cost = cost; best = ch; }
  }
  klen = (nz + best - 1) / best;
  g = dim3(full + 1, (ny + FNY - 1) / FNY, (nz + klen - 1) / klen);
  return true;
}
// The boundary rules in compact form.  bc_pair always leaves L = R = v, and upwind_mac(v, v, .) = v, so on a physical boundary face every
// stage's output IS v -- the inflow value, zero, the inner state, or the inner state clamped (bc_pair / mk_edge_bc: the same four cases):
//   mode 1: ghost value   2: zero   3: inner state   4: inner state, min(.,0) on a lo face / max(.,0) on a hi face
// and the "premod" of stages C and D is the same v applied to the base of the cell next to the face.
DEVI int bc_mode(int phys, bool is_vel, bool normal) {
  if (phys == VDN_INLET) return 1;
  if (phys == VDN_SLIP_WALL) return (is_vel && normal) ? 2 : 3;
  if (phys == VDN_NO_SLIP_WALL) return is_vel ? 2 : 3;
  if (phys == VDN_OUTLET) return (is_vel && normal) ? 4 : 3;
  return 0;
}
DEVI double bc_v(int m, int side, double in, double ghost) {
  double v = in;
  if (m == 4) v = side ? fmax(in, 0.0) : fmin(in, 0.0);
  if (m == 2) v = 0.0;
  if (m == 1) v = ghost;
  return v;
}
// NAR (box-batched launches, boxes narrower than a tile): a wave carries 64 / W node... cell rows of W lanes each (lanes 1 .. W-2 of a segment own cells, the
// two outer ones feed their neighbours, exactly as lanes 0 and 63 of the full-width tile), a workgroup TNY * (64 / W) rows of which all but the first and
// last own cells; W is the descriptor's.  The shared arrays keep their size and are addressed flat (row * W + lane); lanes left over after a wave's last
// segment own nothing, load clamped addresses and park their stores in the slots the rows do not use.  A level of 997 boxes of which 670 are 8 cells wide
// marches 0.52 x the wave-planes of the full-width tiling (tools/box_histogram.py).
struct SegGeo { int W, ROWS, lane, row, rowm, rowp, o0, om, op; bool live; };
template <bool NAR> DEVI SegGeo seg_geo(int W_) {
  SegGeo G;
  if (!NAR) {
    G.W = 64; G.ROWS = TNY; G.lane = threadIdx.x; G.row = threadIdx.y; G.live = true;
    G.rowm = G.row >= 1 ? G.row - 1 : 0; G.rowp = G.row + 1 < TNY ? G.row + 1 : TNY - 1;
    G.o0 = G.om = G.op = 0;
    return G;
  }
  const int tid = threadIdx.x, rpw = 64 / W_, seg = tid / W_;
  G.W = W_; G.ROWS = TNY * rpw; G.live = seg < rpw; G.lane = tid - seg * W_;
  G.row = G.live ? (int)threadIdx.y * rpw + seg : G.ROWS - 1;
  G.rowm = G.row >= 1 ? G.row - 1 : 0; G.rowp = G.row + 1 < G.ROWS ? G.row + 1 : G.ROWS - 1;
  G.o0 = G.live ? G.row * W_ + G.lane : G.ROWS * W_ + (int)threadIdx.y * (64 - rpw * W_) + (tid - rpw * W_);
  G.om = G.rowm * W_ + G.lane; G.op = G.rowp * W_ + G.lane;
  return G;
}
// this thread's slot / the slot of the row below / above in a [TNY][64] shared array
#define LS(A) (*(NAR ? &(&A[0][0])[G.o0] : &A[row][lane]))
#define LM(A) (NAR ? (&A[0][0])[G.om] : A[rowm][lane])
#define LP(A) (NAR ? (&A[0][0])[G.op] : A[rowp][lane])
// UPD: the conservative / convective update of the component (update.f90:220-269) rides along.  A cell's update needs the edge states on its
// six faces: the lower three are this thread's stage-D outputs, the upper x one the next lane's (DPP, same iteration), the upper y one the
// next row's (LDS, read one iteration later, like SI and SC), the upper z one this thread's output for the next plane.  So the x-term of
// plane k is formed with stage D of plane k, the y- and z-terms one iteration later, and seven doubles travel in between; a k-chunk runs
// one plane further (the lower z-face of its successor's first plane).  sedge and flux are not stored.  Same expressions as update_cell.
// ONEB (round 5): ONE workgroup barrier per plane instead of three.  The three stages exchange their y-direction left states through LDS (lB, lC, lD) and
// each had its own write -> barrier -> read.  What a stage WRITES depends on no LDS value of the same iteration -- stage C's lC needs t[0] (x: DPP) and
// t[2] (z: registers), stage D's lD the chain of direction y, whose transverse terms are x (DPP) and z (stage C's z-faces, registers) -- so all three
// writes move in front of one barrier, with every x- and z-direction upwind that needs no LDS value, and everything that reads LDS (this iteration's
// lB / lC / lD, last iteration's lSI / lSC / lE) behind it.  Same expressions, same operands, same bits; the buffers stay double: a wave that has
// passed barrier kk reads buffer kk & 1 and last iteration's lSI / lSC / lE while the fastest wave writes buffer (kk + 1) & 1 and this iteration's.
// the y-exchange arrays of the march, ONE allocation per kernel shared by the bodies inlined into it (boundary / lean, full-width / narrow)
template <bool UPD> struct FShared { double lB[2][TNY][64], lSI[2][TNY][64], lC[2][2][TNY][64], lSC[2][2][TNY][64], lD[2][TNY][64], lE[UPD ? 2 : 1][UPD ? TNY : 1][64]; };
template <bool BC, bool INL, bool UPD, bool PW2, bool NAR = false, bool ONEB = true> __device__ __forceinline__ void mk_F_m_body(FShared<UPD> &S, const FArgs &F, const Range3 &r, int klen, const double *umax, const int BX, const int BY, const int BZ, const int segw = 64) {
  double (&lB)[2][TNY][64] = S.lB, (&lSI)[2][TNY][64] = S.lSI, (&lC)[2][2][TNY][64] = S.lC, (&lSC)[2][2][TNY][64] = S.lSC, (&lD)[2][TNY][64] = S.lD;
  double (&lE)[UPD ? 2 : 1][UPD ? TNY : 1][64] = S.lE;
  const SegGeo G = seg_geo<NAR>(segw);
  const int lane = G.lane, row = G.row;
  const int ownx = NAR ? G.W - 2 : FNX, owny = NAR ? G.ROWS - 2 : FNY;
  const int i = r.lo[0] - 1 + BX * ownx + lane, j = r.lo[1] - 1 + BY * owny + row;
  const bool own_ij = G.live && lane >= 1 && lane <= ownx && row >= 1 && row <= owny && i <= r.hi[0] && j <= r.hi[1];
  const int ic = min(max(i, F.lo[0] - 1), F.hi[0] + 1), jc = min(max(j, F.lo[1] - 1), F.hi[1] + 1);
  const bool ing_ij = i == ic && j == jc;                            // a cell of the grown box
  const bool vx = i >= F.lo[0] && i <= F.hi[0], vy = j >= F.lo[1] && j <= F.hi[1];
  const int rowm = G.rowm, rowp = G.rowp;
  const int k0 = r.lo[2] + BZ * klen, k1 = min(k0 + klen - 1, r.hi[2]);
  const int KB = F.lo[2] - 1, KT = F.hi[2] + 1;                       // the planes loads are clamped to
  const double eps = eps_from(umax);
  const bool cons = F.cons != 0;
  // the thread's column in every field layout
  const unsigned o_s = fg_off(F.g[0], ic, jc), o_sl = fg_off(F.g[1], ic, jc), o_um = fg_off(F.g[2], ic, jc), o_vm = fg_off(F.g[3], ic, jc), o_wm = fg_off(F.g[4], ic, jc);
  const unsigned o_f = fg_off(F.g[5], ic, jc), o_mr = fg_off(F.g[6], ic, jc);
  const unsigned o_ex = fg_off(F.h[0], i, j), o_ey = fg_off(F.h[1], i, j), o_ez = fg_off(F.h[2], i, j);
  // boundary modes of this component: the face rule of the cell's lower x / y face (fm*, side fs*), the rule on the bases of the cells next
  // to a face (pm*0: lowest valid cell, its R base; pm*1: highest valid cell, its L base); z: per plane, below
  int mode[3][2];
  #pragma unroll
  for (int d = 0; d < 3; d++) { mode[d][0] = BC ? bc_mode(F.phys[d][0], F.is_vel != 0, F.c == d) : 0; mode[d][1] = BC ? bc_mode(F.phys[d][1], F.is_vel != 0, F.c == d) : 0; }
  // The rules that apply to this thread, packed into one word of 4-bit fields holding a code (0 none, 1 ghost value, 2 zero, 3 inner state,
  // 4 min(inner, 0), 5 max(inner, 0)):  field 0: the cell's lower x-face is a boundary face (+8: the hi one), 1: the cell is the lowest valid cell
  // along x (rule on its R base), 2: the highest (L base); fields 3..5: the same along y.  The z direction has the same word per plane, uniform.
  // Inside a flagged (rare, divergent) branch the code is decoded with a few selects; the word is laundered through an empty asm in every
  // stage so that the decode is not hoisted out of the march (as 64-bit masks: 186 spilled SGPRs in the first version).
  int cd[3][2];
  #pragma unroll
  for (int d = 0; d < 3; d++) { cd[d][0] = mode[d][0] == 4 ? 4 : mode[d][0]; cd[d][1] = mode[d][1] == 4 ? 5 : mode[d][1]; }
  int bcw = 0;
  if (BC) {
    if (ing_ij && i == F.lo[0]) bcw |= cd[0][0];
    if (ing_ij && i == F.hi[0] + 1 && cd[0][1]) bcw |= cd[0][1] | 8;
    if (ic == F.lo[0]) bcw |= cd[0][0] << 4;
    if (ic == F.hi[0]) bcw |= cd[0][1] << 8;
    if (ing_ij && j == F.lo[1]) bcw |= cd[1][0] << 12;
    if (ing_ij && j == F.hi[1] + 1 && cd[1][1]) bcw |= (cd[1][1] | 8) << 12;
    if (jc == F.lo[1]) bcw |= cd[1][0] << 16;
    if (jc == F.hi[1]) bcw |= cd[1][1] << 20;
  }
  // per-thread plane pointers (the thread's column in each field): inputs at the plane the next loads take (kcur), outputs at the plane stage D
  // emits next.  Fifteen uniform pointers plus their strides do not fit the scalar register file next to everything else (they were
  // spilled to VGPR lanes and read back every plane); per-thread pointers cost the one 64-bit add per load that forming the address cost anyway.
  int kcur = min(max(k0 - 2, KB), KT);
  const char *ps = F.p[0] + (long)(kcur - KB) * F.sp[0] + o_s, *psl0 = F.p[1] + (long)(kcur - KB) * F.sp[1] + o_sl, *psl1 = F.p[2] + (long)(kcur - KB) * F.sp[1] + o_sl, *psl2 = F.p[3] + (long)(kcur - KB) * F.sp[1] + o_sl;
  const char *pum = F.p[4] + (long)(kcur - KB) * F.sp[2] + o_um, *pvm = F.p[5] + (long)(kcur - KB) * F.sp[3] + o_vm, *pwm = F.p[6] + (long)(kcur - KB) * F.sp[4] + o_wm;
  const char *pf = F.p[7] + (long)(kcur - KB) * F.sp[5] + o_f, *pmr = F.p[8] + (long)(kcur - KB) * F.sp[6] + o_mr;
  char *qex = F.q[0] + (long)(k0 - KB) * F.sq[0] + o_ex, *qey = F.q[1] + (long)(k0 - KB) * F.sq[1] + o_ey, *qez = F.q[2] + (long)(k0 - KB) * F.sq[2] + o_ez;
  char *qfx = F.q[3] + (long)(k0 - KB) * F.sq[0] + o_ex, *qfy = F.q[4] + (long)(k0 - KB) * F.sq[1] + o_ey, *qfz = F.q[5] + (long)(k0 - KB) * F.sq[2] + o_ez;
  FCell P0, P1, P2;
  #define FZERO(P) { P.s0 = 0.0; P.f = 0.0; P.mr = 0.0; for (int d = 0; d < 3; d++) { P.m_lo[d] = 0.0; P.m_up[d] = 0.0; P.Lb[d] = 0.0; P.Rb[d] = 0.0; } }
  FZERO(P0) FZERO(P1)
  #undef FZERO
  double LzB = 0.0, LzC[2] = { 0.0, 0.0 }, LzD = 0.0;
  double c_tx = 0.0, c_e1 = 0.0, c_e2 = 0.0, c_vbar = 0.0, c_wbar = 0.0, c_so = 0.0, c_fu = 0.0;       // UPD: what plane k-1 hands to the next iteration
  char *qn = UPD ? F.qn + (long)(k0 - KB) * F.sqn + fg_off(F.hn, i, j) : nullptr;
  double si1[3] = { 0.0, 0.0, 0.0 };                                 // SI on the lower faces of the cell in plane kk-1
  double qp[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };                   // SC on the lower faces of the cell in plane kk-2
  // ghost values of s for the inflow rule (mode 1), read only where it applies: plane kpl, dbytes along x / y from the thread's (clamped) cell
  // (INL = false: the launch has no inflow face, the reads and their address arithmetic are compiled out)
  #define S_AT(kpl, dbytes) (INL ? ldd(ps + (long)((kpl) - kcur) * F.sp[0] + (dbytes), 0u) : 0.0)
  // v of the rule with code cd_ on the inner state `in`; the ghost value is evaluated only for the inflow rule
  #define BC_V(v, code, in, ghost_expr) { const int cd_ = (code); const double in_ = (in); double v_ = in_;                      \
      if (cd_ == 4) v_ = fmin(in_, 0.0); if (cd_ == 5) v_ = fmax(in_, 0.0); if (cd_ == 2) v_ = 0.0; if (cd_ == 1) v_ = (ghost_expr); v = v_; }
  // the face rule on the upwinded value `out` of a lower face: field sh of word w; left state L, right state Rr, this cell's s, ghost of a lo face
  #define FACE_BC(out, w, sh, L, Rr, s0_, ghost_expr) { const int f_ = ((w) >> (sh)) & 15;                                        \
      if (f_) { const bool hi_ = (f_ & 8) != 0; double o_; BC_V(o_, f_ & 7, hi_ ? (L) : (Rr), hi_ ? (s0_) : (ghost_expr)) out = o_; } }
  // the rule on the bases of the cells next to a face: fields shlo (R base of the lowest cell) and shhi (L base of the highest)
  #define PREMOD(Lb_, Rb_, w, shlo, shhi, gl_expr, gh_expr) { const int a_ = ((w) >> (shlo)) & 7; if (a_) { double o_; BC_V(o_, a_, Rb_, gl_expr) Rb_ = o_; }  \
                                                              const int b_ = ((w) >> (shhi)) & 7; if (b_) { double o_; BC_V(o_, b_, Lb_, gh_expr) Lb_ = o_; } }
  // the word of the z direction for stage plane k (clamped kc): uniform
  #define Z_WORD(k, kc) (BC ? ((((k) == (kc) && (k) == F.lo[2]) ? cd[2][0] : 0) | (((k) == (kc) && (k) == KT && cd[2][1]) ? (cd[2][1] | 8) : 0) | \
                                (((kc) == F.lo[2]) ? cd[2][0] << 4 : 0) | (((kc) == F.hi[2]) ? cd[2][1] << 8 : 0)) : 0)
  // a plane's twelve loads are issued one iteration ahead (N): at two waves per SIMD nothing else hides their latency, and an iteration's
  // arithmetic (~500 VALU instructions per wave) is longer than the round trip
  struct FRaw { double m_lo[3], m_up[3], s0, sl[3], f, mr; } N;
  #define F_LOAD {                                                                                                \
      N.m_lo[0] = ldd(pum, 0u); N.m_up[0] = ldd(pum, 8u);                                                         \
      N.m_lo[1] = ldd(pvm, 0u); N.m_up[1] = ldd(pvm + F.vm_row, 0u);                                              \
      N.m_lo[2] = ldd(pwm, 0u); N.m_up[2] = ldd(pwm + F.sp[4], 0u);                                               \
      N.s0 = ldd(ps, 0u); N.sl[0] = ldd(psl0, 0u); N.sl[1] = ldd(psl1, 0u); N.sl[2] = ldd(psl2, 0u);               \
      N.f = ldd(pf, 0u); N.mr = ldd(pmr, 0u); }
  // move the input pointers to the (clamped) plane of march index kn
  #define F_ADVANCE(kn) { const int kc_ = min(max((kn), KB), KT);                                                 \
      if (kc_ != kcur) { ps += F.sp[0]; psl0 += F.sp[1]; psl1 += F.sp[1]; psl2 += F.sp[1]; pum += F.sp[2]; pvm += F.sp[3]; pwm += F.sp[4]; pf += F.sp[5]; pmr += F.sp[6]; kcur = kc_; } }
  F_LOAD
  F_ADVANCE(k0 - 1)
  if (ONEB) {
  for (int kk = k0 - 2; kk <= k1 + 2 + (UPD ? 1 : 0); kk++) {
    const int buf = kk & 1;
    P2 = P1; P1 = P0;
    {
      #pragma unroll
      for (int d = 0; d < 3; d++) { P0.m_lo[d] = N.m_lo[d]; P0.m_up[d] = N.m_up[d]; }
      P0.s0 = N.s0; P0.f = N.f; P0.mr = N.mr;
      const double sl[3] = { N.sl[0], N.sl[1], N.sl[2] };
      if (kk < k1 + 2) { F_LOAD  F_ADVANCE(kk + 2) }
      f_bases<PW2>(F, P0, sl);
    }
    const bool doC = kk - 1 >= k0 - 1, doD = kk - 2 >= k0 - 1;
    // ================= in front of the barrier: the three LDS writes and every upwind that reads no LDS value =================
    // ---- stage B, plane kk: x and z faces
    double si0[3];
    const int kB = kk, kcB = min(max(kB, KB), KT), zwB = Z_WORD(kB, kcB);
    double LxB;
    {
      LS(lB[buf]) = P0.Lb[1];
      LxB = shfl_prev(P0.Lb[0]);
      const double Lzc = LzB;
      LzB = P0.Lb[2];
      si0[0] = upwind_mac(LxB, P0.Rb[0], P0.m_lo[0], eps);
      si0[2] = upwind_mac(Lzc, P0.Rb[2], P0.m_lo[2], eps);
      if (BC && kB == kcB) {
        int w = bcw; asm volatile("" : "+v"(w));
        if (w & 0xF) { FACE_BC(si0[0], w, 0, LxB, P0.Rb[0], P0.s0, S_AT(kcB, -8L)) }
        if ((zwB & 15) && ing_ij) { FACE_BC(si0[2], zwB, 0, Lzc, P0.Rb[2], P0.s0, P1.s0) }
      }
    }
    // ---- stage C, plane kk-1: t[0], t[2], the y left states, and the faces whose transverse term is not the y one (x-face / z term, z-face / x term)
    double qc[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    double cLb[3] = { 0.0, 0.0, 0.0 }, cRb[3] = { 0.0, 0.0, 0.0 }, ct0 = 0.0, ct2 = 0.0, cLz1 = 0.0;
    const int kC = kk - 1, kcC = min(max(kC, KB), KT), zwC = Z_WORD(kC, kcC);
    int wc = 0;
    if (doC) {
      #pragma unroll
      for (int d = 0; d < 3; d++) { cLb[d] = P1.Lb[d]; cRb[d] = P1.Rb[d]; }
      if (BC) {
        wc = bcw; asm volatile("" : "+v"(wc));
        if (wc & 0x770770) { PREMOD(cLb[0], cRb[0], wc, 4, 8, S_AT(kcC, -8L), S_AT(kcC, 8L)) PREMOD(cLb[1], cRb[1], wc, 16, 20, S_AT(kcC, -F.s_row), S_AT(kcC, F.s_row)) }
        if (zwC & 0x770) { PREMOD(cLb[2], cRb[2], zwC, 4, 8, S_AT(kcC - 1, 0L), S_AT(kcC + 1, 0L)) }
      }
      ct0 = tvq(cons, F.tC[0], lane_next(si1[0]), si1[0], P1.m_up[0], P1.m_lo[0]);
      ct2 = tvq(cons, F.tC[2], si0[2], si1[2], P1.m_up[2], P1.m_lo[2]);
      LS(lC[buf][0]) = cLb[1] - ct0; LS(lC[buf][1]) = cLb[1] - ct2;
      const double VL01 = cLb[0] - ct2, VR01 = cRb[0] - ct2, VL20 = cLb[2] - ct0, VR20 = cRb[2] - ct0;
      const double Lx1 = shfl_prev(VL01), Lz0 = LzC[0];
      cLz1 = LzC[1];
      LzC[0] = VL20;
      qc[1] = upwind_mac(Lx1, VR01, P1.m_lo[0], eps);
      qc[4] = upwind_mac(Lz0, VR20, P1.m_lo[2], eps);
      if (BC && kC == kcC) {
        if (wc & 0xF) { FACE_BC(qc[1], wc, 0, Lx1, VR01, P1.s0, S_AT(kcC, -8L)) }
        if ((zwC & 15) && ing_ij) { FACE_BC(qc[4], zwC, 0, Lz0, VR20, P1.s0, P2.s0) }
      }
    }
    // ---- stage D, plane kk-2: the chain of direction y (transverse terms x and z) and its left state
    double dLb[3] = { 0.0, 0.0, 0.0 }, dRb[3] = { 0.0, 0.0, 0.0 }, dVR1 = 0.0, da[3] = { 0.0, 0.0, 0.0 }, dft = 0.0, dmt = 0.0;
    const int kD = kk - 2, kcD = min(max(kD, KB), KT), zwD = Z_WORD(kD, kcD);
    int wd = 0;
    #define CHAIN1(vl_, vr_, Dd, T1, T2, q1a, q0a, q1b, q0b)                                                   \
        { const double t1 = tvq(cons, F.tD[T1], q1a, q0a, P2.m_up[T1], P2.m_lo[T1]);                             \
          const double t2 = tvq(cons, F.tD[T2], q1b, q0b, P2.m_up[T2], P2.m_lo[T2]);                             \
          double vl = dLb[Dd], vr = dRb[Dd];                                                                     \
          vl = vl - t1; vr = vr - t1; vl = vl - t2; vr = vr - t2;                                                \
          if (cons) { vl = vl + da[T1]; vr = vr + da[T1]; vl = vl + da[T2]; vr = vr + da[T2]; }                  \
          if (!F.use_minion) { vl = vl + dft; vr = vr + dft; if (cons) { vl = vl - dmt; vr = vr - dmt; } }       \
          vl_ = vl; vr_ = vr; }
    if (doD) {
      dft = F.dt2 * P2.f; dmt = F.dt2 * P2.s0 * P2.mr;
      #pragma unroll
      for (int d = 0; d < 3; d++) { dLb[d] = P2.Lb[d]; dRb[d] = P2.Rb[d]; da[d] = F.aD[d] * P2.s0 * (P2.m_up[d] - P2.m_lo[d]); }
      if (BC) {
        wd = bcw; asm volatile("" : "+v"(wd));
        if (wd & 0x770770) { PREMOD(dLb[0], dRb[0], wd, 4, 8, S_AT(kcD, -8L), S_AT(kcD, 8L)) PREMOD(dLb[1], dRb[1], wd, 16, 20, S_AT(kcD, -F.s_row), S_AT(kcD, F.s_row)) }
        if (zwD & 0x770) { PREMOD(dLb[2], dRb[2], zwD, 4, 8, S_AT(kcD - 1, 0L), S_AT(kcD + 1, 0L)) }
      }
      double vl1;
      CHAIN1(vl1, dVR1, 1, 0, 2, lane_next(qp[1]), qp[1], qc[4], qp[4])
      LS(lD[buf]) = vl1;
    }
    __syncthreads();
    // ================= behind the barrier: everything that reads LDS =================
    // ---- stage B: the y face
    {
      const double Ly = LM(lB[buf]);
      si0[1] = upwind_mac(Ly, P0.Rb[1], P0.m_lo[1], eps);
      if (BC && kB == kcB) {
        int w = bcw; asm volatile("" : "+v"(w));
        if (w & 0xF000) { FACE_BC(si0[1], w, 12, Ly, P0.Rb[1], P0.s0, S_AT(kcB, -F.s_row)) }
      }
      LS(lSI[buf]) = si0[1];                                   // read by the row below in the next iteration
    }
    // ---- stage C: t[1] (SI on the upper y face: the row above, last iteration) and the faces that need it or the y left states
    if (doC) {
      const double ct1 = tvq(cons, F.tC[1], LP(lSI[buf ^ 1]), si1[1], P1.m_up[1], P1.m_lo[1]);
      const double VL00 = cLb[0] - ct1, VR00 = cRb[0] - ct1, VR10 = cRb[1] - ct0, VR11 = cRb[1] - ct2, VL21 = cLb[2] - ct1, VR21 = cRb[2] - ct1;
      const double Lx0 = shfl_prev(VL00), Ly0 = LM(lC[buf][0]), Ly1 = LM(lC[buf][1]);
      LzC[1] = VL21;
      qc[0] = upwind_mac(Lx0, VR00, P1.m_lo[0], eps);
      qc[2] = upwind_mac(Ly0, VR10, P1.m_lo[1], eps); qc[3] = upwind_mac(Ly1, VR11, P1.m_lo[1], eps);
      qc[5] = upwind_mac(cLz1, VR21, P1.m_lo[2], eps);
      if (BC && kC == kcC) {
        if (wc & 0xF00F) {
          FACE_BC(qc[0], wc, 0, Lx0, VR00, P1.s0, S_AT(kcC, -8L))
          FACE_BC(qc[2], wc, 12, Ly0, VR10, P1.s0, S_AT(kcC, -F.s_row)) FACE_BC(qc[3], wc, 12, Ly1, VR11, P1.s0, S_AT(kcC, -F.s_row))
        }
        if ((zwC & 15) && ing_ij) { FACE_BC(qc[5], zwC, 0, cLz1, VR21, P1.s0, P2.s0) }
      }
      LS(lSC[buf][0]) = qc[2]; LS(lSC[buf][1]) = qc[3];            // read by the row below in the next iteration
    }
    // ---- stage D: the chains of directions x and z, the three edge states, the update
    if (doD) {
      const int k = kD, kc = kcD, zw = zwD;
      const bool vz = k >= F.lo[2] && k <= F.hi[2];
      const double (&m_lo)[3] = P2.m_lo, (&m_up)[3] = P2.m_up;
      const double s0 = P2.s0;
      double VL[3], VR[3];
      VR[1] = dVR1;
      CHAIN1(VL[0], VR[0], 0, 1, 2, LP(lSC[buf ^ 1][1]), qp[3], qc[5], qp[5])
      CHAIN1(VL[2], VR[2], 2, 0, 1, lane_next(qp[0]), qp[0], LP(lSC[buf ^ 1][0]), qp[2])
      double L[3];
      L[0] = shfl_prev(VL[0]); L[1] = LM(lD[buf]); L[2] = LzD; LzD = VL[2];
      if (k >= k0) {
        double e[3] = { 0.0, 0.0, 0.0 };
        if (UPD || own_ij) {
          e[0] = upwind_mac(L[0], VR[0], m_lo[0], eps); e[1] = upwind_mac(L[1], VR[1], m_lo[1], eps); e[2] = upwind_mac(L[2], VR[2], m_lo[2], eps);
          if (BC) {
            if (wd & 0xF00F) { FACE_BC(e[0], wd, 0, L[0], VR[0], s0, S_AT(kc, -8L)) FACE_BC(e[1], wd, 12, L[1], VR[1], s0, S_AT(kc, -F.s_row)) }
            if (zw & 15) { FACE_BC(e[2], zw, 0, L[2], VR[2], s0, S_AT(kc - 1, 0L)) }
          }
        }
        if (!UPD) {
          if (own_ij) {
            if (vy && vz) { std_(qex, 0u, e[0]); if (cons) std_(qfx, 0u, e[0] * m_lo[0]); }
            if (vx && vz) { std_(qey, 0u, e[1]); if (cons) std_(qfy, 0u, e[1] * m_lo[1]); }
            if (vx && vy) { std_(qez, 0u, e[2]); if (cons) std_(qfz, 0u, e[2] * m_lo[2]); }
          }
          qex += F.sq[0]; qfx += F.sq[0]; qey += F.sq[1]; qfy += F.sq[1]; qez += F.sq[2]; qfz += F.sq[2];
        } else {
          const double ex = cons ? e[0] * m_lo[0] : e[0], ey = cons ? e[1] * m_lo[1] : e[1], ez = cons ? e[2] * m_lo[2] : e[2];
          if (k - 1 >= k0) {                                           // finish plane k-1: its upper y face from the row above, its upper z face = ez
            const double eyu = LP(lE[buf ^ 1]);
            const double ty = cons ? DIVDX(eyu - c_e1, 1) : DIVDX(c_vbar * (eyu - c_e1), 1);
            const double tz = cons ? DIVDX(ez - c_e2, 2) : DIVDX(c_wbar * (ez - c_e2), 2);
            const double ug = c_tx + ty + tz;
            const bool vzp = k - 1 >= F.lo[2] && k - 1 <= F.hi[2];
            if (own_ij && vx && vy && vzp) std_(qn, 0u, c_so - F.dt * ug + F.dt * c_fu);
            qn += F.sqn;
          }
          const double exu = lane_next(ex);                            // the upper x face: the next lane's lower one
          c_tx = cons ? DIVDX(exu - ex, 0) : DIVDX((0.5 * (m_lo[0] + m_up[0])) * (exu - ex), 0);
          c_vbar = 0.5 * (m_lo[1] + m_up[1]); c_wbar = 0.5 * (m_lo[2] + m_up[2]);
          c_e1 = ey; c_e2 = ez; c_so = s0;
          if (F.fmode == 0) c_fu = P2.f;
          else {
            const long po = (long)(kc - KB) * F.sp[5] + o_f;
            c_fu = ldd(F.pfu[0] + po, 0u) + (F.lapu0 - ldd(F.pfu[1] + po, 0u)) / ldd(F.pfu[2] + po, 0u);
          }
          LS(lE[buf]) = ey;                                     // read by the row below in the next iteration
        }
      }
    }
    #undef CHAIN1
    si1[0] = si0[0]; si1[1] = si0[1]; si1[2] = si0[2];
    #pragma unroll
    for (int n = 0; n < 6; n++) qp[n] = qc[n];
  }
  } else
  for (int kk = k0 - 2; kk <= k1 + 2 + (UPD ? 1 : 0); kk++) {
    const int buf = kk & 1;
    P2 = P1; P1 = P0;
    {
      #pragma unroll
      for (int d = 0; d < 3; d++) { P0.m_lo[d] = N.m_lo[d]; P0.m_up[d] = N.m_up[d]; }
      P0.s0 = N.s0; P0.f = N.f; P0.mr = N.mr;
      const double sl[3] = { N.sl[0], N.sl[1], N.sl[2] };
      if (kk < k1 + 2) { F_LOAD  F_ADVANCE(kk + 2) }
      f_bases<PW2>(F, P0, sl);
    }
    // ---------------- stage B, plane kk ----------------
    double si0[3];
    {
      const int k = kk, kc = min(max(k, KB), KT);
      const int zw = Z_WORD(k, kc);                                   // uniform
      const double (&Lb)[3] = P0.Lb, (&Rb)[3] = P0.Rb;
      LS(lB[buf]) = Lb[1];
      __syncthreads();
      const double Lx = shfl_prev(Lb[0]), Ly = LM(lB[buf]), Lzc = LzB;
      LzB = Lb[2];
      si0[0] = upwind_mac(Lx, Rb[0], P0.m_lo[0], eps);
      si0[1] = upwind_mac(Ly, Rb[1], P0.m_lo[1], eps);
      si0[2] = upwind_mac(Lzc, Rb[2], P0.m_lo[2], eps);
      if (BC && k == kc) {
        int w = bcw; asm volatile("" : "+v"(w));
        if (w & 0xF00F) { FACE_BC(si0[0], w, 0, Lx, Rb[0], P0.s0, S_AT(kc, -8L)) FACE_BC(si0[1], w, 12, Ly, Rb[1], P0.s0, S_AT(kc, -F.s_row)) }
        if ((zw & 15) && ing_ij) { FACE_BC(si0[2], zw, 0, Lzc, Rb[2], P0.s0, P1.s0) }
      }
      LS(lSI[buf]) = si0[1];                                   // read by the row below in the next iteration
    }
    // ---------------- stage C, plane kk-1 ----------------
    double qc[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    if (kk - 1 >= k0 - 1) {
      const int k = kk - 1, kc = min(max(k, KB), KT);
      const int zw = Z_WORD(k, kc);
      double Lb[3] = { P1.Lb[0], P1.Lb[1], P1.Lb[2] }, Rb[3] = { P1.Rb[0], P1.Rb[1], P1.Rb[2] };
      int wc = 0;
      if (BC) {
        wc = bcw; asm volatile("" : "+v"(wc));
        if (wc & 0x770770) { PREMOD(Lb[0], Rb[0], wc, 4, 8, S_AT(kc, -8L), S_AT(kc, 8L)) PREMOD(Lb[1], Rb[1], wc, 16, 20, S_AT(kc, -F.s_row), S_AT(kc, F.s_row)) }
        if (zw & 0x770) { PREMOD(Lb[2], Rb[2], zw, 4, 8, S_AT(kc - 1, 0L), S_AT(kc + 1, 0L)) }
      }
      const double su[3] = { lane_next(si1[0]), LP(lSI[buf ^ 1]), si0[2] };       // SI on the upper faces
      double t[3];
      t[0] = tvq(cons, F.tC[0], su[0], si1[0], P1.m_up[0], P1.m_lo[0]);
      t[1] = tvq(cons, F.tC[1], su[1], si1[1], P1.m_up[1], P1.m_lo[1]);
      t[2] = tvq(cons, F.tC[2], su[2], si1[2], P1.m_up[2], P1.m_lo[2]);
      double VL[3][2], VR[3][2];
      VL[0][0] = Lb[0] - t[1]; VR[0][0] = Rb[0] - t[1]; VL[0][1] = Lb[0] - t[2]; VR[0][1] = Rb[0] - t[2];
      VL[1][0] = Lb[1] - t[0]; VR[1][0] = Rb[1] - t[0]; VL[1][1] = Lb[1] - t[2]; VR[1][1] = Rb[1] - t[2];
      VL[2][0] = Lb[2] - t[0]; VR[2][0] = Rb[2] - t[0]; VL[2][1] = Lb[2] - t[1]; VR[2][1] = Rb[2] - t[1];
      LS(lC[buf][0]) = VL[1][0]; LS(lC[buf][1]) = VL[1][1];
      __syncthreads();
      double Lx[2], Ly[2], Lzc[2];
      Lx[0] = shfl_prev(VL[0][0]); Lx[1] = shfl_prev(VL[0][1]);
      Ly[0] = LM(lC[buf][0]); Ly[1] = LM(lC[buf][1]);
      Lzc[0] = LzC[0]; Lzc[1] = LzC[1];
      LzC[0] = VL[2][0]; LzC[1] = VL[2][1];
      qc[0] = upwind_mac(Lx[0], VR[0][0], P1.m_lo[0], eps); qc[1] = upwind_mac(Lx[1], VR[0][1], P1.m_lo[0], eps);
      qc[2] = upwind_mac(Ly[0], VR[1][0], P1.m_lo[1], eps); qc[3] = upwind_mac(Ly[1], VR[1][1], P1.m_lo[1], eps);
      qc[4] = upwind_mac(Lzc[0], VR[2][0], P1.m_lo[2], eps); qc[5] = upwind_mac(Lzc[1], VR[2][1], P1.m_lo[2], eps);
      if (BC && k == kc) {
        if (wc & 0xF00F) {
          FACE_BC(qc[0], wc, 0, Lx[0], VR[0][0], P1.s0, S_AT(kc, -8L)) FACE_BC(qc[1], wc, 0, Lx[1], VR[0][1], P1.s0, S_AT(kc, -8L))
          FACE_BC(qc[2], wc, 12, Ly[0], VR[1][0], P1.s0, S_AT(kc, -F.s_row)) FACE_BC(qc[3], wc, 12, Ly[1], VR[1][1], P1.s0, S_AT(kc, -F.s_row))
        }
        if ((zw & 15) && ing_ij) { FACE_BC(qc[4], zw, 0, Lzc[0], VR[2][0], P1.s0, P2.s0) FACE_BC(qc[5], zw, 0, Lzc[1], VR[2][1], P1.s0, P2.s0) }
      }
      LS(lSC[buf][0]) = qc[2]; LS(lSC[buf][1]) = qc[3];            // read by the row below in the next iteration
    }
    // ---------------- stage D, plane kk-2 ----------------
    if (kk - 2 >= k0 - 1) {
      const int k = kk - 2, kc = min(max(k, KB), KT);
      const int zw = Z_WORD(k, kc);
      const bool vz = k >= F.lo[2] && k <= F.hi[2];
      const double (&m_lo)[3] = P2.m_lo, (&m_up)[3] = P2.m_up;
      const double s0 = P2.s0;
      const double ft = F.dt2 * P2.f, mt = F.dt2 * s0 * P2.mr;
      double Lb[3] = { P2.Lb[0], P2.Lb[1], P2.Lb[2] }, Rb[3] = { P2.Rb[0], P2.Rb[1], P2.Rb[2] };
      int wd = 0;
      if (BC) {
        wd = bcw; asm volatile("" : "+v"(wd));
        if (wd & 0x770770) { PREMOD(Lb[0], Rb[0], wd, 4, 8, S_AT(kc, -8L), S_AT(kc, 8L)) PREMOD(Lb[1], Rb[1], wd, 16, 20, S_AT(kc, -F.s_row), S_AT(kc, F.s_row)) }
        if (zw & 0x770) { PREMOD(Lb[2], Rb[2], zw, 4, 8, S_AT(kc - 1, 0L), S_AT(kc + 1, 0L)) }
      }
      // SC on the upper faces: x-faces from the next lane, y-faces from the next row (written in the previous iteration), z-faces from stage C above
      const double q1[6] = { lane_next(qp[0]), lane_next(qp[1]), LP(lSC[buf ^ 1][0]), LP(lSC[buf ^ 1][1]), qc[4], qc[5] };
      const double (&q0)[6] = qp;
      double VL[3], VR[3];
      const double a[3] = { F.aD[0] * s0 * (m_up[0] - m_lo[0]), F.aD[1] * s0 * (m_up[1] - m_lo[1]), F.aD[2] * s0 * (m_up[2] - m_lo[2]) };
      #define CHAIN(Dd, T1, T2, n1, n2)                                                                    \
        { const double t1 = tvq(cons, F.tD[T1], q1[n1], q0[n1], m_up[T1], m_lo[T1]);                         \
          const double t2 = tvq(cons, F.tD[T2], q1[n2], q0[n2], m_up[T2], m_lo[T2]);                         \
          double vl = Lb[Dd], vr = Rb[Dd];                                                                   \
          vl = vl - t1; vr = vr - t1; vl = vl - t2; vr = vr - t2;                                            \
          if (cons) { vl = vl + a[T1]; vr = vr + a[T1]; vl = vl + a[T2]; vr = vr + a[T2]; }                  \
          if (!F.use_minion) { vl = vl + ft; vr = vr + ft; if (cons) { vl = vl - mt; vr = vr - mt; } }       \
          VL[Dd] = vl; VR[Dd] = vr; }
      CHAIN(0, 1, 2, 3, 5)
      CHAIN(1, 0, 2, 1, 4)
      CHAIN(2, 0, 1, 0, 2)
      #undef CHAIN
      LS(lD[buf]) = VL[1];
      __syncthreads();
      double L[3];
      L[0] = shfl_prev(VL[0]); L[1] = LM(lD[buf]); L[2] = LzD; LzD = VL[2];
      if (k >= k0) {
        double e[3] = { 0.0, 0.0, 0.0 };
        if (UPD || own_ij) {
          e[0] = upwind_mac(L[0], VR[0], m_lo[0], eps); e[1] = upwind_mac(L[1], VR[1], m_lo[1], eps); e[2] = upwind_mac(L[2], VR[2], m_lo[2], eps);
          if (BC) {
            if (wd & 0xF00F) { FACE_BC(e[0], wd, 0, L[0], VR[0], s0, S_AT(kc, -8L)) FACE_BC(e[1], wd, 12, L[1], VR[1], s0, S_AT(kc, -F.s_row)) }
            if (zw & 15) { FACE_BC(e[2], zw, 0, L[2], VR[2], s0, S_AT(kc - 1, 0L)) }
          }
        }
        if (!UPD) {
          if (own_ij) {
            if (vy && vz) { std_(qex, 0u, e[0]); if (cons) std_(qfx, 0u, e[0] * m_lo[0]); }
            if (vx && vz) { std_(qey, 0u, e[1]); if (cons) std_(qfy, 0u, e[1] * m_lo[1]); }
            if (vx && vy) { std_(qez, 0u, e[2]); if (cons) std_(qfz, 0u, e[2] * m_lo[2]); }
          }
          qex += F.sq[0]; qfx += F.sq[0]; qey += F.sq[1]; qfy += F.sq[1]; qez += F.sq[2]; qfz += F.sq[2];
        } else {
          // what update_3d differences: fluxes of a conservative component (flux = sedge * umac, mkflux.f90:1969), edge states otherwise
          const double ex = cons ? e[0] * m_lo[0] : e[0], ey = cons ? e[1] * m_lo[1] : e[1], ez = cons ? e[2] * m_lo[2] : e[2];
          if (k - 1 >= k0) {                                           // finish plane k-1: its upper y face from the row above, its upper z face = ez
            const double eyu = LP(lE[buf ^ 1]);
            const double ty = cons ? DIVDX(eyu - c_e1, 1) : DIVDX(c_vbar * (eyu - c_e1), 1);
            const double tz = cons ? DIVDX(ez - c_e2, 2) : DIVDX(c_wbar * (ez - c_e2), 2);
            const double ug = c_tx + ty + tz;
            const bool vzp = k - 1 >= F.lo[2] && k - 1 <= F.hi[2];
            if (own_ij && vx && vy && vzp) std_(qn, 0u, c_so - F.dt * ug + F.dt * c_fu);
            qn += F.sqn;
          }
          const double exu = lane_next(ex);                            // the upper x face: the next lane's lower one
          c_tx = cons ? DIVDX(exu - ex, 0) : DIVDX((0.5 * (m_lo[0] + m_up[0])) * (exu - ex), 0);
          c_vbar = 0.5 * (m_lo[1] + m_up[1]); c_wbar = 0.5 * (m_lo[2] + m_up[2]);
          c_e1 = ey; c_e2 = ez; c_so = s0;
          if (F.fmode == 0) c_fu = P2.f;
          else {
            const long po = (long)(kc - KB) * F.sp[5] + o_f;
            c_fu = ldd(F.pfu[0] + po, 0u) + (F.lapu0 - ldd(F.pfu[1] + po, 0u)) / ldd(F.pfu[2] + po, 0u);
          }
          LS(lE[buf]) = ey;                                     // read by the row below in the next iteration
        }
      }
    }
    si1[0] = si0[0]; si1[1] = si0[1]; si1[2] = si0[2];
    #pragma unroll
    for (int n = 0; n < 6; n++) qp[n] = qc[n];
  }
  #undef F_LOAD
  #undef F_ADVANCE
  #undef S_AT
  #undef BC_V
  #undef FACE_BC
  #undef PREMOD
  #undef Z_WORD
}
template <bool BC = true, bool INL = true, bool UPD = false, bool PW2 = false, bool ONEB = true> __global__ void __launch_bounds__(64 * TNY) kk_mk_F_m(FArgs F, Range3 r, int klen, const double *umax) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  // a workgroup whose tile and k-chunk stay clear of every face that carries a rule runs the body without the boundary code (same values:
  // none of its cells is flagged); compiled into one kernel the lean path keeps its own register allocation (0.69 against 0.96 ms per launch)
  bool touch = false;
  if (BC) {
    const int i0 = r.lo[0] - 1 + bx_ * FNX, i1 = i0 + 63, j0 = r.lo[1] - 1 + by_ * FNY, j1 = j0 + TNY - 1;
    const int k0 = r.lo[2] + bz_ * klen - 2, k1 = min(r.lo[2] + bz_ * klen + klen - 1, r.hi[2]) + 2 + (UPD ? 1 : 0);
    const int a0[3] = { i0, j0, k0 }, a1[3] = { i1, j1, k1 };
    #pragma unroll
    for (int d = 0; d < 3; d++) {
      if (bc_mode(F.phys[d][0], F.is_vel != 0, F.c == d) && a0[d] <= F.lo[d] + 1) touch = true;
      if (bc_mode(F.phys[d][1], F.is_vel != 0, F.c == d) && a1[d] >= F.hi[d] - 1) touch = true;
    }
  }
  __shared__ FShared<UPD> S;
  if (BC && touch) mk_F_m_body<BC, INL, UPD, PW2, false, ONEB>(S, F, r, klen, umax, bx_, by_, bz_);
  else mk_F_m_body<false, false, UPD, PW2, false, ONEB>(S, F, r, klen, umax, bx_, by_, bz_);
}
// The remainder column of a one-box level in narrow segments (round 5).  A row of 256 cells is four 62-cell tiles and 8 cells: the fifth tile column
// marched 64 lanes for 8 cells (a fifth of all workgroups at 256^3).  With NCOL that column runs in segments of (cells + 2) lanes, 64 / W rows per wave
// (NAR, see seg_geo): 10 lanes x 6 rows, a workgroup 48 rows of which 46 own cells -- 6 workgroups per k-chunk instead of 43; the other workgroups of the
// column's grid slots leave at once.  ONE launch (as a second launch the 42 workgroups were a round of their own: measured, slower than the full tiles).
// Same body, same values; the update rides along as in the full-width tiles (the shared arrays are addressed flat).
template <bool BC, bool INL, bool UPD, bool PW2> __global__ void __launch_bounds__(64 * TNY) kk_mk_F_mc(FArgs F, Range3 r, int klen, const double *umax, int full, int segw) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  const bool narrow = bx_ == full;
  Range3 rr = r;
  int tw = 64, th = TNY, ownx = FNX;
  if (narrow) {
    tw = segw; th = TNY * (64 / segw); ownx = segw - 2;
    rr.lo[0] = r.lo[0] + full * FNX;
    if (by_ * (th - 2) > r.hi[1] - r.lo[1]) return;                 // the narrow column needs fewer workgroups along y than the grid has
    bx_ = 0;
  } else rr.hi[0] = r.lo[0] + full * FNX - 1;
  bool touch = false;
  if (BC) {
    const int i0 = rr.lo[0] - 1 + bx_ * ownx, i1 = i0 + tw - 1, j0 = rr.lo[1] - 1 + by_ * (th - 2), j1 = j0 + th - 1;
    const int k0 = rr.lo[2] + bz_ * klen - 2, k1 = min(rr.lo[2] + bz_ * klen + klen - 1, rr.hi[2]) + 2 + (UPD ? 1 : 0);
    const int a0[3] = { i0, j0, k0 }, a1[3] = { i1, j1, k1 };
    #pragma unroll
    for (int d = 0; d < 3; d++) {
      if (bc_mode(F.phys[d][0], F.is_vel != 0, F.c == d) && a0[d] <= F.lo[d] + 1) touch = true;
      if (bc_mode(F.phys[d][1], F.is_vel != 0, F.c == d) && a1[d] >= F.hi[d] - 1) touch = true;
    }
  }
  __shared__ FShared<UPD> S;
  if (narrow) {
    if (BC && touch) mk_F_m_body<BC, INL, UPD, PW2, true>(S, F, rr, klen, umax, bx_, by_, bz_, segw);
    else mk_F_m_body<false, false, UPD, PW2, true>(S, F, rr, klen, umax, bx_, by_, bz_, segw);
  } else {
    if (BC && touch) mk_F_m_body<BC, INL, UPD, PW2>(S, F, rr, klen, umax, bx_, by_, bz_);
    else mk_F_m_body<false, false, UPD, PW2>(S, F, rr, klen, umax, bx_, by_, bz_);
  }
}
// the launch arguments of the fused march for component c; false when the field layouts do not allow the shared offsets
static bool fused_args(FArgs &F, const GArgs &A, int c, const FV &s, const FV sl[3], const FV &um, const FV &vm, const FV &wm, const FV &force, const FV &macrhs,
                       const FV &sex, const FV &sey, const FV &sez, const FV &flx, const FV &fly, const FV &flz) {
  if (!same_geom(sl[0], sl[1]) || !same_geom(sl[0], sl[2]) || !same_geom(sex, flx) || !same_geom(sey, fly) || !same_geom(sez, flz)) return false;
  const int KB = A.lo[2] - 1;
  const FV *in[9] = { &s, &sl[0], &sl[1], &sl[2], &um, &vm, &wm, &force, &macrhs };
  const int comp[9] = { c, c, c, c, 0, 0, 0, c, 0 }, grp[9] = { 0, 1, 1, 1, 2, 3, 4, 5, 6 };
  for (int f = 0; f < 9; f++) {
    const FV &v = *in[f];
    if (KB < v.a2 || A.hi[2] + 1 + (f == 6 ? 1 : 0) >= v.a2 + v.n2) return false;      // the clamped planes must exist (wm: one face more)
    F.p[f] = (const char *)(v.p + v.sc * comp[f] + (long)v.n0 * v.n1 * (KB - v.a2));
    F.sp[grp[f]] = 8L * v.n0 * v.n1; F.g[grp[f]] = FGeo{ v.a0, v.a1, v.n0 };
  }
  const FV *out[6] = { &sex, &sey, &sez, &flx, &fly, &flz };
  for (int f = 0; f < 6; f++) {
    const FV &v = *out[f];
    F.q[f] = (char *)(v.p + v.sc * c + (long)v.n0 * v.n1 * (KB - v.a2));
    if (f < 3) { F.sq[f] = 8L * v.n0 * v.n1; F.h[f] = FGeo{ v.a0, v.a1, v.n0 }; }
  }
  F.s_row = 8L * s.n0; F.vm_row = 8L * vm.n0;
  const bool cons = A.cons[c] != 0;
  const double dt2 = 0.5 * A.dt, dt3 = A.dt / 3.0, dt4 = A.dt / 4.0, dt6 = A.dt / 6.0;
  F.dt2 = dt2;
  for (int d = 0; d < 3; d++) {
    F.dx[d] = A.dx[d]; F.idx[d] = 1.0 / A.dx[d]; F.tC[d] = (cons ? dt3 : dt6) / A.dx[d]; F.tD[d] = (cons ? dt2 : dt4) / A.dx[d]; F.aD[d] = dt2 / A.dx[d];
    F.lo[d] = A.lo[d]; F.hi[d] = A.hi[d]; F.phys[d][0] = A.phys[d][0]; F.phys[d][1] = A.phys[d][1];
  }
  F.cons = cons ? 1 : 0; F.use_minion = A.use_minion; F.is_vel = A.is_vel; F.c = c;
  F.p2 = (is_pow2(A.dx[0]) && is_pow2(A.dx[1]) && is_pow2(A.dx[2]) && !no_p2()) ? 1 : 0;
  F.qn = nullptr; F.sqn = 0; F.hn = FGeo{ 0, 0, 0 }; F.pfu[0] = F.pfu[1] = F.pfu[2] = nullptr; F.fmode = 0; F.dt = A.dt; F.lapu0 = 0.0;
  return true;
}
// the update of component c inside the march (UPD): snew, and the forcing term of the update -- the force of this call (fmode 0) or formed in place
// from ext, gp and rho, which must share the layout of `force` (fmode 1)
static bool fused_update_args(FArgs &F, const GArgs &A, int c, const FV &snew, const FV &force, const MkUpdate &U, int ib) {
  const int KB = A.lo[2] - 1;
  if (KB < snew.a2 || A.hi[2] + 1 >= snew.a2 + snew.n2) return false;
  F.qn = (char *)(snew.p + snew.sc * c + (long)snew.n0 * snew.n1 * (KB - snew.a2));
  F.sqn = 8L * snew.n0 * snew.n1; F.hn = FGeo{ snew.a0, snew.a1, snew.n0 };
  F.fmode = U.fmode; F.lapu0 = U.lapu0;
  if (U.fmode == 1) {
    const FV *in[3] = { &U.ext->fabs[ib], &U.gp->fabs[ib], &U.rho->fabs[ib] };
    const int comp[3] = { c, c, 0 };
    for (int f = 0; f < 3; f++) {
      const FV &v = *in[f];
      if (!same_geom(v, force) || v.n2 != force.n2) return false;
      F.pfu[f] = (const char *)(v.p + v.sc * comp[f] + (long)v.n0 * v.n1 * (KB - v.a2));
    }
  }
  return true;
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


// ====================================================================================================
// one launch per stage for ALL boxes of a level: a descriptor per box, workgroups bisect a prefix sum for theirs.
// The descriptors are read through the constant address space (scalar loads, never invalidated by the kernel's stores),
// exactly like kernel arguments.
// ====================================================================================================
struct MkD { FV s, sl0, sl1, sl2, um, vm, wm, force, macrhs, SI, SC, sex, sey, sez, flx, fly, flz; GArgs A; Range3 rm, rg, rf; int klg, klf, gs[3], gm[3], gg[3], gf[3]; double *umax; };
struct VpD { FV s, sl0, sl1, sl2, force, UI, XC, um, vm, wm;                                     GArgs A; Range3 rm, rg, rf; int klg, klf, gs[3], gm[3], gg[3], gf[3]; double *umax; };
DEVI int locate_box(const int *start, int nbox, int bid) {
  int lo = 0, hi = nbox - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (as_constant(start + mid) <= bid) lo = mid; else hi = mid - 1; }
  return lo;
}
#define BATCH_LOCATE(D, G)                                                              \
  const int ib_ = locate_box(start, nbox, (int)blockIdx.x);                             \
  const D &q = as_constant(descs + ib_);                                                \
  const int l_ = (int)blockIdx.x - as_constant(start + ib_);                            \
  const int gy_ = q.G[1] > 0 ? q.G[1] : 1;                                               \
  const int BX = l_ % q.G[0], BY = (l_ / q.G[0]) % gy_, BZ = l_ / (q.G[0] * gy_);
template <class D> __global__ void __launch_bounds__(256) kk_slopes_b(const D *descs, const int *start, int nbox, int dirmask) {
  BATCH_LOCATE(D, gs)
  // gs[1] == 0: the plane flattened over the workgroup's 256 threads (gs[0] workgroups per plane) -- a 14 x 14 plane of a grown 8^3 box is one
  // workgroup at 77 % instead of four 64 x 4 tiles at 19 %
  int i, j, k;
  if (q.gs[1] == 0) {
    const int nx = q.rg.hi[0] - q.rg.lo[0] + 1, l_2 = (int)blockIdx.x - as_constant(start + ib_);
    const int t = (l_2 % q.gs[0]) * 256 + (int)threadIdx.y * 64 + (int)threadIdx.x;
    i = q.rg.lo[0] + t % nx; j = q.rg.lo[1] + t / nx; k = q.rg.lo[2] + l_2 / q.gs[0];
  } else { i = q.rg.lo[0] + BX * 64 + (int)threadIdx.x; j = q.rg.lo[1] + BY * 4 + (int)threadIdx.y; k = q.rg.lo[2] + BZ; }
  if (i > q.rg.hi[0] || j > q.rg.hi[1] || k > q.rg.hi[2]) return;
  slopes_cell(q.s, q.sl0, q.sl1, q.sl2, q.A, dirmask, i, j, k);
}
// per-box maxima (eps of the upwind tests is per box, as in the reference's per-fab calls)
__global__ void kk_macmax_b(const MkD *descs, const int *start, int nbox) {
  BATCH_LOCATE(MkD, gm)
  const int i = q.rf.lo[0] + BX * 64 + (int)threadIdx.x, j = q.rf.lo[1] + BY * 4 + (int)threadIdx.y;
  double m = 0.0;
  if (i <= q.rf.hi[0] && j <= q.rf.hi[1]) for (int k = q.rf.lo[2] + BZ; k <= q.rf.hi[2]; k += q.gm[2]) {
    if (j <= q.A.hi[1] && k <= q.A.hi[2]) m = fmax(m, fabs(fv_get(q.um, i, j, k)));
    if (i <= q.A.hi[0] && k <= q.A.hi[2]) m = fmax(m, fabs(fv_get(q.vm, i, j, k)));
    if (i <= q.A.hi[0] && j <= q.A.hi[1]) m = fmax(m, fabs(fv_get(q.wm, i, j, k)));
  }
  block_atomic_max(q.umax, m);
}
__global__ void kk_velmax_b(const VpD *descs, const int *start, int nbox) {
  BATCH_LOCATE(VpD, gm)
  const int i = q.rm.lo[0] + BX * 64 + (int)threadIdx.x, j = q.rm.lo[1] + BY * 4 + (int)threadIdx.y;
  double m = 0.0;
  if (i <= q.rm.hi[0] && j <= q.rm.hi[1]) for (int k = q.rm.lo[2] + BZ; k <= q.rm.hi[2]; k += q.gm[2])
    m = fmax(m, fmax(fmax(fabs(fv_get(q.s, i, j, k, 0)), fabs(fv_get(q.s, i, j, k, 1))), fabs(fv_get(q.s, i, j, k, 2))));
  block_atomic_max(q.umax, m);
}
template <int NC> __global__ void __launch_bounds__(64 * TNY) kk_mk_B_mb(const MkD *descs, const int *start, int nbox, int c0, int ns) {
  BATCH_LOCATE(MkD, gg)
  mk_B_m_body<NC, true, true>(q.s, q.sl0, q.sl1, q.sl2, q.um, q.vm, q.wm, q.force, q.macrhs, q.SI, q.A, q.rg, q.klg, q.umax, c0, ns, BX, BY, BZ);
}
template <int NC> __global__ void __launch_bounds__(64 * TNY) kk_mk_C_mb(const MkD *descs, const int *start, int nbox, int c0, int ns) {
  BATCH_LOCATE(MkD, gg)
  mk_C_m_body<NC, true, true>(q.s, q.sl0, q.sl1, q.sl2, q.um, q.vm, q.wm, q.force, q.macrhs, q.SI, q.SC, q.A, q.rg, q.klg, q.umax, c0, ns, BX, BY, BZ);
}
template <int NC> __global__ void __launch_bounds__(64 * TNY) kk_mk_D_mb(const MkD *descs, const int *start, int nbox, int c0, int ns) {
  BATCH_LOCATE(MkD, gf)
  mk_D_m_body<NC, true, false, true>(q.s, q.sl0, q.sl1, q.sl2, q.um, q.vm, q.wm, q.force, q.macrhs, q.SC, q.sex, q.sey, q.sez, q.flx, q.fly, q.flz, q.A, q.rf, q.klf, q.umax, c0, ns, BX, BY, BZ);
}
// host side of a batch: grids of the four launch shapes per box, their prefix sums, and the upload
// the fused march for every box of a level in one launch (a descriptor per box and component; see kk_batched for the scheme)
struct FBatchD { FArgs F; Range3 r; int klen; const double *umax; int g[3], sw; };      // sw: lanes of a row segment (NAR launches; seg_geo)
template <bool INL, bool PW2 = false, bool NAR = false> __global__ void __launch_bounds__(64 * TNY) kk_mk_F_mb(const FBatchD *descs, const int *start, int nbox) {
  BATCH_LOCATE(FBatchD, g)
  bool touch = false;                                // see kk_mk_F_m
  {
    const int tw = NAR ? q.sw : 64, th = NAR ? TNY * (64 / q.sw) : TNY;
    const int i0 = q.r.lo[0] - 1 + BX * (tw - 2), i1 = i0 + tw - 1, j0 = q.r.lo[1] - 1 + BY * (th - 2), j1 = j0 + th - 1;
    const int k0 = q.r.lo[2] + BZ * q.klen - 2, k1 = min(q.r.lo[2] + BZ * q.klen + q.klen - 1, q.r.hi[2]) + 2;
    const int a0[3] = { i0, j0, k0 }, a1[3] = { i1, j1, k1 };
    #pragma unroll
    for (int d = 0; d < 3; d++) {
      if (bc_mode(q.F.phys[d][0], q.F.is_vel != 0, q.F.c == d) && a0[d] <= q.F.lo[d] + 1) touch = true;
      if (bc_mode(q.F.phys[d][1], q.F.is_vel != 0, q.F.c == d) && a1[d] >= q.F.hi[d] - 1) touch = true;
    }
  }
  __shared__ FShared<false> S;
  if (touch) mk_F_m_body<true, INL, false, PW2, NAR>(S, q.F, q.r, q.klen, q.umax, BX, BY, BZ, q.sw);
  else mk_F_m_body<false, false, false, PW2, NAR>(S, q.F, q.r, q.klen, q.umax, BX, BY, BZ, q.sw);
}
// k-chunks of a box in the batched launch: the boxes of a level fill the device together, so a box is cut only when it is tall
// sw: 0 = full-width tiles (64 x TNY threads own 62 x 6 cells), else the width of the row segments that take the fewest workgroups (seg_geo: a
// workgroup owns (sw - 2) x (TNY * (64 / sw) - 2) cells)
static void fused_grid_small(const Range3 &r, int &klen, int g[3], int &sw) {
  const int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
  const int chunks = std::max(1, (nz + 47) / 48);
  klen = (nz + chunks - 1) / chunks;
  g[0] = (nx + FNX - 1) / FNX; g[1] = (ny + FNY - 1) / FNY; g[2] = (nz + klen - 1) / klen;
  sw = 0;
  static const bool narrow_on = !(vdn_env("VDN_GOD_SEGW") && atoi(vdn_env("VDN_GOD_SEGW")) == 0);
  if (!narrow_on) return;
  int best = g[0] * g[1];
  for (int w = 6; w <= 32; w++) {
    const int ox = w - 2, oy = TNY * (64 / w) - 2, c = ((nx + ox - 1) / ox) * ((ny + oy - 1) / oy);
    if (c < best) { best = c; sw = w; g[0] = (nx + ox - 1) / ox; g[1] = (ny + oy - 1) / oy; }
  }
}

template <class D> struct GodBatch {
  std::vector<D> d; const D *dev = nullptr; const int *st[4] = { nullptr, nullptr, nullptr, nullptr }; int tot[4] = { 0, 0, 0, 0 };
  void finish() {
    const int nb = (int)d.size();
    std::vector<int> start(4 * nb);
    for (int b = 0; b < nb; b++) {
      D &q = d[b];
      const dim3 gs = grid_for(q.rg), gm = reduce_grid(q.rm);
      const dim3 gg = march_grid(q.rg, q.klg), gf = march_grid(q.rf, q.klf);
      const dim3 *g[4] = { &gs, &gm, &gg, &gf }; int *o[4] = { q.gs, q.gm, q.gg, q.gf };
      for (int t = 0; t < 4; t++) { o[t][0] = g[t]->x; o[t][1] = g[t]->y; o[t][2] = g[t]->z; start[t * nb + b] = tot[t]; tot[t] += (int)(g[t]->x * g[t]->y * g[t]->z); }
      // the slope launch: the plane flattened over the workgroup when that takes fewer workgroups (kk_slopes_b: gs[1] = 0)
      static const bool flat_on = !(vdn_env("VDN_BATCH_FLAT") && atoi(vdn_env("VDN_BATCH_FLAT")) == 0);
      const int pnx = q.rg.hi[0] - q.rg.lo[0] + 1, pny = q.rg.hi[1] - q.rg.lo[1] + 1;
      if (flat_on && pnx > 0 && pny > 0 && (pnx * pny + 255) / 256 < (int)(gs.x * gs.y)) {
        const int gfl = (pnx * pny + 255) / 256;
        tot[0] += (gfl - (int)(gs.x * gs.y)) * (int)gs.z;
        q.gs[0] = gfl; q.gs[1] = 0;
      }
    }
    D *dd = (D *)desc_scratch(sizeof(D) * nb); int *ds = (int *)desc_scratch(sizeof(int) * 4 * nb);
    upload_staged(dd, d.data(), sizeof(D) * nb); upload_staged(ds, start.data(), sizeof(int) * 4 * nb);
    dev = dd; for (int t = 0; t < 4; t++) st[t] = ds + t * nb;
  }
};

static void fill_gargs(GArgs &A, const vdn_multifab *s, int ibox, const vdn_bc_tower *bct, int bccomp, int ncomp, const double *dx, double dt) {
  memset(&A, 0, sizeof A);
  BoxP bp = make_boxp(s, ibox, bct);
  for (int d = 0; d < 3; d++) {
    A.lo[d] = bp.lo[d]; A.hi[d] = bp.hi[d]; A.dx[d] = (dx && d < ctx().prm.dm) ? dx[d] : 1.0;      // dx holds dm entries
    for (int sd = 0; sd < 2; sd++) {
      A.phys[d][sd] = bp.phys[d][sd];
      for (int c = 0; c < ncomp && c < 3; c++) A.adv[d][sd][c] = bct->adv_bc(s->lev, ibox + 1, d, sd, bccomp + c);
    }
  }
  A.dt = dt; A.ncomp = ncomp; A.use_minion = ctx().prm.use_minion; A.slope_order = ctx().prm.slope_order; A.outlet2d = ctx().extruded2d ? 1 : 0;
}
static void k2_velpred(const vdn_multifab *u, vdn_multifab **umac, const vdn_multifab *force, const double *dx, double dt, const vdn_bc_tower *bct);
static void k2_mkflux(const vdn_multifab *s, vdn_multifab **sedge, vdn_multifab **flux, vdn_multifab **umac, const vdn_multifab *force,
                      const vdn_multifab *mac_rhs, const double *dx, double dt, const vdn_bc_tower *bct, bool is_vel, const int *is_cons);
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

// upd: the caller would run update_3d on the result next (one level): where the fused march is taken -- a level of one box -- the update of
// every component runs inside it, sedge / flux stay unwritten and the call returns true; otherwise false and the caller updates as usual
bool k_mkflux(const vdn_multifab *s, vdn_multifab **sedge, vdn_multifab **flux, vdn_multifab **umac,
              const vdn_multifab *force, const vdn_multifab *mac_rhs, const double *dx, double dt,
              const vdn_bc_tower *bct, bool is_vel, const int *is_cons, const MkUpdate *upd) {
  Prof prof_("mkflux");
  god_xcd_init();
  if (ctx().prm.dm == 2) { k2_mkflux(s, sedge, flux, umac, force, mac_rhs, dx, dt, bct, is_vel, is_cons); return false; }
  bool updated = false;
  const int ncomp = s->nc;
  REQUIRE(ncomp <= 3, "mkflux: at most 3 components per call (got %d)", ncomp);
  REQUIRE(s->ng >= 3 && umac[0]->ng >= 1 && force->ng >= 1 && mac_rhs->ng >= 1, "mkflux: ghost widths");
  const int bccomp = is_vel ? 0 : bct->dm;            // mkflux.f90:62-66
  hipStream_t st = ctx().stream;
  if (use_batched(s) && !plain_godunov()) {            // every stage once for all boxes of the level
    size_t mark = arena_mark();
    const int nb = s->nfabs();
    GodBatch<MkD> B; B.d.resize(nb);
    double *umax = (double *)arena_alloc(sizeof(double) * nb);
    HIPCHK(hipMemsetAsync(umax, 0, sizeof(double) * nb, st));
    for (int ib = 0; ib < nb; ib++) {
      MkD &q = B.d[ib];
      fill_gargs(q.A, s, ib, bct, bccomp, ncomp, dx, dt);
      q.A.is_vel = is_vel ? 1 : 0;
      for (int c = 0; c < ncomp; c++) q.A.cons[c] = is_cons[c] ? 1 : 0;
      BoxP bp = make_boxp(s, ib, bct);
      FV w = work_fv(nullptr, bp, 0);
      const size_t fld = (size_t)w.sc * sizeof(double);
      q.sl0 = q.sl1 = q.sl2 = q.SI = q.SC = w;
      q.sl0.p = (double *)arena_alloc(fld * ncomp); q.sl1.p = (double *)arena_alloc(fld * ncomp); q.sl2.p = (double *)arena_alloc(fld * ncomp);
      q.SI.p = (double *)arena_alloc(fld * 3 * ncomp); q.SC.p = (double *)arena_alloc(fld * 6 * ncomp);
      for (int d = 0; d < 3; d++) { q.rg.lo[d] = q.A.lo[d] - 1; q.rg.hi[d] = q.A.hi[d] + 1; q.rf.lo[d] = q.A.lo[d]; q.rf.hi[d] = q.A.hi[d] + 1; }
      q.rm = q.rf;
      q.s = s->fabs[ib]; q.um = umac[0]->fabs[ib]; q.vm = umac[1]->fabs[ib]; q.wm = umac[2]->fabs[ib]; q.force = force->fabs[ib]; q.macrhs = mac_rhs->fabs[ib];
      q.sex = sedge[0]->fabs[ib]; q.sey = sedge[1]->fabs[ib]; q.sez = sedge[2]->fabs[ib]; q.flx = flux[0]->fabs[ib]; q.fly = flux[1]->fabs[ib]; q.flz = flux[2]->fabs[ib];
      q.umax = umax + ib;
    }
    B.finish();
    const dim3 blk(64, TNY, 1);
    hipLaunchKernelGGL(kk_macmax_b, dim3(B.tot[1]), dim3(64, 4, 1), 0, st, B.dev, B.st[1], nb);
    hipLaunchKernelGGL(kk_slopes_b<MkD>, dim3(B.tot[0]), dim3(64, 4, 1), 0, st, B.dev, B.st[0], nb, 7);
    static const bool fused_env = !(vdn_env("VDN_GOD_FUSED") && atoi(vdn_env("VDN_GOD_FUSED")) == 0);
    if (fused_env) {               // stages B + C + D in one march per box and component (mk_F_m_body)
      std::vector<FBatchD> fd((size_t)nb * ncomp);
      std::vector<int> fstart((size_t)nb * ncomp);
      bool ok = true, inflow = false;
      int tot = 0;
      for (int ib = 0; ib < nb && ok; ib++) {
        const MkD &q = B.d[ib];
        const FV sl[3] = { q.sl0, q.sl1, q.sl2 };
        for (int d = 0; d < 3; d++) inflow = inflow || q.A.phys[d][0] == VDN_INLET || q.A.phys[d][1] == VDN_INLET;
        for (int c0 = 0; c0 < ncomp && ok; c0++) {
          FBatchD &f = fd[(size_t)c0 * nb + ib];
          ok = fused_args(f.F, q.A, c0, q.s, sl, q.um, q.vm, q.wm, q.force, q.macrhs, q.sex, q.sey, q.sez, q.flx, q.fly, q.flz);
          f.r = q.rf; f.umax = q.umax;
          fused_grid_small(f.r, f.klen, f.g, f.sw);
        }
      }
      if (ok) {
        // two launches: the boxes in full-width tiles, the narrow ones in row segments (seg_geo)
        std::stable_partition(fd.begin(), fd.end(), [](const FBatchD &f) { return f.sw == 0; });
        const size_t nwide = std::count_if(fd.begin(), fd.end(), [](const FBatchD &f) { return f.sw == 0; });
        int totn = 0;
        for (size_t t = 0; t < fd.size(); t++) { int &T = t < nwide ? tot : totn; fstart[t] = T; T += fd[t].g[0] * fd[t].g[1] * fd[t].g[2]; }
        FBatchD *dd = (FBatchD *)desc_scratch(sizeof(FBatchD) * fd.size()); int *ds = (int *)desc_scratch(sizeof(int) * fstart.size());
        upload_staged(dd, fd.data(), sizeof(FBatchD) * fd.size()); upload_staged(ds, fstart.data(), sizeof(int) * fstart.size());
        const bool p2 = fd[0].F.p2 != 0;
        if (nwide) {
          if (inflow) hipLaunchKernelGGL(kk_mk_F_mb<true>, dim3(tot), blk, 0, st, dd, ds, (int)nwide);
          else if (p2) hipLaunchKernelGGL((kk_mk_F_mb<false, true>), dim3(tot), blk, 0, st, dd, ds, (int)nwide);
          else hipLaunchKernelGGL(kk_mk_F_mb<false>, dim3(tot), blk, 0, st, dd, ds, (int)nwide);
        }
        if (nwide < fd.size()) {
          const int nn = (int)(fd.size() - nwide);
          if (inflow) hipLaunchKernelGGL((kk_mk_F_mb<true, false, true>), dim3(totn), blk, 0, st, dd + nwide, ds + nwide, nn);
          else if (p2) hipLaunchKernelGGL((kk_mk_F_mb<false, true, true>), dim3(totn), blk, 0, st, dd + nwide, ds + nwide, nn);
          else hipLaunchKernelGGL((kk_mk_F_mb<false, false, true>), dim3(totn), blk, 0, st, dd + nwide, ds + nwide, nn);
        }
        arena_release(mark);
        return false;
      }
    }
    const int split = ncomp >= 2 ? 4 : 0;
    #define MKB_STAGE(K, t, bit)                                                                                               \
      if ((split >> bit) & 1) { for (int c0 = 0; c0 < ncomp; c0++) hipLaunchKernelGGL(K<1>, dim3(B.tot[t]), blk, 0, st, B.dev, B.st[t], nb, c0, ncomp); } \
      else if (ncomp == 3) hipLaunchKernelGGL(K<3>, dim3(B.tot[t]), blk, 0, st, B.dev, B.st[t], nb, 0, ncomp);                  \
      else if (ncomp == 2) hipLaunchKernelGGL(K<2>, dim3(B.tot[t]), blk, 0, st, B.dev, B.st[t], nb, 0, ncomp);                  \
      else hipLaunchKernelGGL(K<1>, dim3(B.tot[t]), blk, 0, st, B.dev, B.st[t], nb, 0, ncomp);
    MKB_STAGE(kk_mk_B_mb, 2, 0)
    MKB_STAGE(kk_mk_C_mb, 2, 1)
    MKB_STAGE(kk_mk_D_mb, 3, 2)
    #undef MKB_STAGE
    arena_release(mark);
    return false;
  }
  for (int ib = 0; ib < s->nfabs(); ib++) {
    size_t mark = arena_mark();
    GArgs A; fill_gargs(A, s, ib, bct, bccomp, ncomp, dx, dt);
    A.is_vel = is_vel ? 1 : 0;
    for (int c = 0; c < ncomp; c++) A.cons[c] = is_cons[c] ? 1 : 0;
    BoxP bp = make_boxp(s, ib, bct);
    FV w = work_fv(nullptr, bp, 0);
    const size_t fld = (size_t)w.sc * sizeof(double);
    FV sl[3], SI = w, SC = w;
    const bool cached = is_vel && ncomp == 3 && (int)ctx().slope_src.size() == s->nfabs() && ctx().slope_src[ib] == s->fabs[ib].p;
    for (int d = 0; d < 3; d++) { sl[d] = w; sl[d].p = cached ? ctx().slope_cache[d][ib] : (double *)arena_alloc(fld * ncomp); }
    SI.p = (double *)arena_alloc(fld * 3 * ncomp);
    SC.p = (double *)arena_alloc(fld * 6 * ncomp);
    Range3 rg, rf;
    for (int d = 0; d < 3; d++) { rg.lo[d] = A.lo[d] - 1; rg.hi[d] = A.hi[d] + 1; rf.lo[d] = A.lo[d]; rf.hi[d] = A.hi[d] + 1; }
    const FV &um = umac[0]->fabs[ib], &vm = umac[1]->fabs[ib], &wm = umac[2]->fabs[ib];
    const bool mm_keep = (int)ctx().macmax_cache.size() == s->nfabs();       // advance_timestep: one reduction serves both mkflux calls of the step
    double *umax = mm_keep ? ctx().macmax_cache[ib] : (double *)arena_alloc(256);
    if (!(mm_keep && ctx().macmax_src[ib] == um.p)) {
      HIPCHK(hipMemsetAsync(umax, 0, sizeof(double), st));
      hipLaunchKernelGGL(kk_macmax, reduce_grid(rf), dim3(64, 4, 1), 0, st, um, vm, wm, A, rf, umax);
      if (mm_keep) ctx().macmax_src[ib] = um.p;
    }
    if (!cached) launch_slopes(s->fabs[ib], sl, A, rg, ncomp, nullptr, st);
    if (plain_godunov()) {
      hipLaunchKernelGGL(kk_mk_B, grid_for(rg), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SI, A, rg, umax);
      hipLaunchKernelGGL(kk_mk_C, grid_for(rg), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SI, SC, A, rg, umax);
      hipLaunchKernelGGL(kk_mk_D, grid_for(rf), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SC,
                         sedge[0]->fabs[ib], sedge[1]->fabs[ib], sedge[2]->fabs[ib], flux[0]->fabs[ib], flux[1]->fabs[ib], flux[2]->fabs[ib], A, rf, umax);
    } else {
      int klg, klf;
      const dim3 gg = march_grid(rg, klg), gf = march_grid(rf, klf), blk(64, TNY, 1);
      // components per launch: the 3-component stage D needs more than 256 VGPRs, so it runs one component at a time
      #define MK_ARGS_B(c0) s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SI, A, rg, klg, umax, c0, ncomp
      #define MK_ARGS_C(c0) s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SI, SC, A, rg, klg, umax, c0, ncomp
      #define MK_ARGS_D(c0) s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SC, \
                            sedge[0]->fabs[ib], sedge[1]->fabs[ib], sedge[2]->fabs[ib], flux[0]->fabs[ib], flux[1]->fabs[ib], flux[2]->fabs[ib], A, rf, klf, umax, c0, ncomp
      // measured at 256^3: D fused <3> 3.18 ms (61 spilled VGPRs) vs 3 x <1> 2.3 ms; D <2> 1.66 ms vs 2 x <1> 1.55 ms; B and C are faster fused
      const int split = ncomp >= 2 ? 4 : 0;          // bit 0: B, 1: C, 2: D per component
      #define MK_STAGE(K, BCF, ARGS, g, bit)                                                                               \
        if ((split >> bit) & 1) { for (int c0 = 0; c0 < ncomp; c0++) hipLaunchKernelGGL((K<1, BCF>), g, blk, 0, st, ARGS(c0)); } \
        else if (ncomp == 3) hipLaunchKernelGGL((K<3, BCF>), g, blk, 0, st, ARGS(0));                                        \
        else if (ncomp == 2) hipLaunchKernelGGL((K<2, BCF>), g, blk, 0, st, ARGS(0));                                        \
        else hipLaunchKernelGGL((K<1, BCF>), g, blk, 0, st, ARGS(0));
      #define MK_STAGE_D(BCF)                                                                                              \
        if ((split >> 2) & 1) { for (int c0 = 0; c0 < ncomp; c0++) hipLaunchKernelGGL((kk_mk_D_m<1, false, BCF>), gf, blk, 0, st, MK_ARGS_D(c0)); } \
        else if (ncomp == 3) hipLaunchKernelGGL((kk_mk_D_m<3, false, BCF>), gf, blk, 0, st, MK_ARGS_D(0));                     \
        else if (ncomp == 2) hipLaunchKernelGGL((kk_mk_D_m<2, false, BCF>), gf, blk, 0, st, MK_ARGS_D(0));                     \
        else hipLaunchKernelGGL((kk_mk_D_m<1, false, BCF>), gf, blk, 0, st, MK_ARGS_D(0));
      static const bool fused_env = !(vdn_env("VDN_GOD_FUSED") && atoi(vdn_env("VDN_GOD_FUSED")) == 0);
      FArgs FA[3];
      bool fused = fused_env;
      for (int c0 = 0; c0 < ncomp && fused; c0++)
        fused = fused_args(FA[c0], A, c0, s->fabs[ib], sl, um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], sedge[0]->fabs[ib], sedge[1]->fabs[ib], sedge[2]->fabs[ib],
                           flux[0]->fabs[ib], flux[1]->fabs[ib], flux[2]->fabs[ib]);
      static const bool upd_env = !(vdn_env("VDN_GOD_UPDATE") && atoi(vdn_env("VDN_GOD_UPDATE")) == 0);
      bool do_upd = fused && upd && upd_env;        // (every box of a box-by-box level: the conditions are geometric and the same for all of them -- checked below)
      for (int c0 = 0; c0 < ncomp && do_upd; c0++) do_upd = fused_update_args(FA[c0], A, c0, upd->snew->fabs[ib], force->fabs[ib], *upd, ib);
      if (fused) {                 // stages B + C + D in one march per component, boundary rules inside (see mk_F_m_body)
        int klF;
        const dim3 gF = fused_grid(rf, klF);
        for (int c0 = 0; c0 < ncomp; c0++) {
          bool inflow = false;
          for (int d = 0; d < 3; d++) inflow = inflow || A.phys[d][0] == VDN_INLET || A.phys[d][1] == VDN_INLET;
          bool any = false;
          for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) any = any || bc_mode_host(A.phys[d][sd]);
          if (do_upd) {
            const bool p2 = FA[c0].p2 != 0;                              // (the inflow variants keep the division: fewer instantiations)
            if (!any) { if (p2) hipLaunchKernelGGL((kk_mk_F_m<false, false, true, true>), gF, blk, 0, st, FA[c0], rf, klF, umax); else hipLaunchKernelGGL((kk_mk_F_m<false, false, true>), gF, blk, 0, st, FA[c0], rf, klF, umax); }
            else if (inflow) hipLaunchKernelGGL((kk_mk_F_m<true, true, true>), gF, blk, 0, st, FA[c0], rf, klF, umax);
            else if (p2 && !god_oneb()) hipLaunchKernelGGL((kk_mk_F_m<true, false, true, true, false>), gF, blk, 0, st, FA[c0], rf, klF, umax);     // the three-barrier loop, kept for the variants test
            else if (p2) {
              // full 62-cell tiles, then the remainder column in narrow segments (kk_mk_F_mn) where that saves workgroups
              dim3 gM; int klM, full, segw;
              if (fused_grid_cols(rf, gM, klM, full, segw)) {
                hipLaunchKernelGGL((kk_mk_F_mc<true, false, true, true>), gM, blk, 0, st, FA[c0], rf, klM, umax, full, segw);
              } else hipLaunchKernelGGL((kk_mk_F_m<true, false, true, true>), gF, blk, 0, st, FA[c0], rf, klF, umax);
            }
            else hipLaunchKernelGGL((kk_mk_F_m<true, false, true>), gF, blk, 0, st, FA[c0], rf, klF, umax);
            continue;
          }
          const bool p2 = FA[c0].p2 != 0;
          if (!any) { if (p2) hipLaunchKernelGGL((kk_mk_F_m<false, false, false, true>), gF, blk, 0, st, FA[c0], rf, klF, umax); else hipLaunchKernelGGL((kk_mk_F_m<false, false>), gF, blk, 0, st, FA[c0], rf, klF, umax); }      // no physical face on this box
          else if (inflow) hipLaunchKernelGGL((kk_mk_F_m<true, true>), gF, blk, 0, st, FA[c0], rf, klF, umax);
          else if (p2) hipLaunchKernelGGL((kk_mk_F_m<true, false, false, true>), gF, blk, 0, st, FA[c0], rf, klF, umax);
          else hipLaunchKernelGGL((kk_mk_F_m<true, false>), gF, blk, 0, st, FA[c0], rf, klF, umax);
        }
        REQUIRE(ib == 0 || updated == do_upd, "mkflux: the update would ride along on some boxes of the level only");
        updated = do_upd;
      } else if (slab_bc()) {      // interior marches, then the face-centred code on the boundary slabs (see kk_slabs)
        REQUIRE(!updated, "mkflux: the update rode along on an earlier box of the level but cannot on box %d", ib);      // (every box or none)
        const MkPlain P{ s->fabs[ib], sl[0], sl[1], sl[2], um, vm, wm, force->fabs[ib], mac_rhs->fabs[ib], SI, SC,
                         sedge[0]->fabs[ib], sedge[1]->fabs[ib], sedge[2]->fabs[ib], flux[0]->fabs[ib], flux[1]->fabs[ib], flux[2]->fabs[ib], A, umax };
        const Slabs Sg = boundary_slabs(A, rg), Sf = boundary_slabs(A, rf);
        MK_STAGE(kk_mk_B_m, false, MK_ARGS_B, gg, 0)
        launch_slabs(Sg, MkBFix{ P }, st);
        MK_STAGE(kk_mk_C_m, false, MK_ARGS_C, gg, 1)
        launch_slabs(Sg, MkCFix{ P }, st);
        MK_STAGE_D(false)
        launch_slabs(Sf, MkDFix{ P }, st);
      } else {
        REQUIRE(!updated, "mkflux: the update rode along on an earlier box of the level but cannot on box %d", ib);
        MK_STAGE(kk_mk_B_m, true, MK_ARGS_B, gg, 0)
        MK_STAGE(kk_mk_C_m, true, MK_ARGS_C, gg, 1)
        MK_STAGE_D(true)
      }
      #undef MK_STAGE_D
      #undef MK_STAGE
      #undef MK_ARGS_B
      #undef MK_ARGS_C
      #undef MK_ARGS_D
    }
    arena_release(mark);
  }
  return updated;
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
      bc_pair(L[c], R[c], A.phys[D][side], side, true, c == D, ghost, D == 0 && side == 1 && !A.outlet2d);   // :2075 quirk
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
struct VpPlain { FV u, sl0, sl1, sl2, force, UI, XC, um, vm, wm; GArgs A; const double *umax; };
DEVI void vp_B_cell(const FV &u, const FV &sl0, const FV &sl1, const FV &sl2, const FV &force, const FV &UI, const GArgs &A, int i, int j, int k, double eps, int dmask = 7) {
  if (dmask & 1) vp_B_one<0>(A, u, sl0, force, UI, i, j, k, eps);
  if (dmask & 2) vp_B_one<1>(A, u, sl1, force, UI, i, j, k, eps);
  if (dmask & 4) vp_B_one<2>(A, u, sl2, force, UI, i, j, k, eps);
}
__global__ void __launch_bounds__(256) kk_vp_B(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  vp_B_cell(u, sl0, sl1, sl2, force, UI, A, i, j, k, eps_from(umax));
}
struct VpBFix { VpPlain P; __device__ void operator()(int i, int j, int k, int dmask) const { vp_B_cell(P.u, P.sl0, P.sl1, P.sl2, P.force, P.UI, P.A, i, j, k, eps_from(P.umax), dmask); } };

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
DEVI void vp_C_cell(const FV &u, const FV &sl0, const FV &sl1, const FV &sl2, const FV &force, const FV &UI, const FV &XC, const GArgs &A, int i, int j, int k, double eps, int dmask = 7) {
  if (dmask & 2) vp_C_one<0, 1>(A, u, sl1, force, UI, XC, i, j, k, eps);
  if (dmask & 4) vp_C_one<0, 2>(A, u, sl2, force, UI, XC, i, j, k, eps);
  if (dmask & 1) vp_C_one<1, 0>(A, u, sl0, force, UI, XC, i, j, k, eps);
  if (dmask & 4) vp_C_one<1, 2>(A, u, sl2, force, UI, XC, i, j, k, eps);
  if (dmask & 1) vp_C_one<2, 0>(A, u, sl0, force, UI, XC, i, j, k, eps);
  if (dmask & 2) vp_C_one<2, 1>(A, u, sl1, force, UI, XC, i, j, k, eps);
}
__global__ void __launch_bounds__(256) kk_vp_C(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, FV XC, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  vp_C_cell(u, sl0, sl1, sl2, force, UI, XC, A, i, j, k, eps_from(umax));
}
struct VpCFix { VpPlain P; __device__ void operator()(int i, int j, int k, int dmask) const { vp_C_cell(P.u, P.sl0, P.sl1, P.sl2, P.force, P.UI, P.XC, P.A, i, j, k, eps_from(P.umax), dmask); } };

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
DEVI void vp_D_cell(const FV &u, const FV &sl0, const FV &sl1, const FV &sl2, const FV &force, const FV &UI, const FV &XC, const FV &um, const FV &vm, const FV &wm,
                    const GArgs &A, int i, int j, int k, double eps, int dmask = 7) {
  if (dmask & 1) vp_D_one<0>(A, u, sl0, force, UI, XC, um, i, j, k, eps);
  if (dmask & 2) vp_D_one<1>(A, u, sl1, force, UI, XC, vm, i, j, k, eps);
  if (dmask & 4) vp_D_one<2>(A, u, sl2, force, UI, XC, wm, i, j, k, eps);
}
__global__ void __launch_bounds__(256) kk_vp_D(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, FV XC, FV um, FV vm, FV wm, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  vp_D_cell(u, sl0, sl1, sl2, force, UI, XC, um, vm, wm, A, i, j, k, eps_from(umax));
}
struct VpDFix { VpPlain P; __device__ void operator()(int i, int j, int k, int dmask) const { vp_D_cell(P.u, P.sl0, P.sl1, P.sl2, P.force, P.UI, P.XC, P.um, P.vm, P.wm, P.A, i, j, k, eps_from(P.umax), dmask); } };

// ---- velpred, cell-centred k-marching form -------------------------------------------------------------------------
// predictor bases of all three velocity components in a cell along D (vp_pair split by side, no boundary rule)
template <int D> DEVI void vp_bases(const GArgs &A, const double uc[3], const double sl[3], const double ft[3], double Lb[3], double Rb[3]) {
  const double dt2 = 0.5 * A.dt;
  double cfl_l;
  if (D == 1) cfl_l = dt2 * fmax(0.0, uc[D] / A.dx[1]);       // velpred.f90:2108: division inside max()
  else cfl_l = dt2 * fmax(0.0, uc[D]) / A.dx[D];
  const double cfl_r = dt2 * fmin(0.0, uc[D]) / A.dx[D];
  #pragma unroll
  for (int c = 0; c < 3; c++) {
    Lb[c] = uc[c] + (0.5 - cfl_l) * sl[c];
    Rb[c] = uc[c] - (0.5 + cfl_r) * sl[c];
    if (A.use_minion) { Lb[c] = Lb[c] + ft[c]; Rb[c] = Rb[c] + ft[c]; }
  }
}
// this cell's half of the boundary rule on component c of its D-bases
template <int D> DEVI void vp_premod(const GArgs &A, const FV &u, int c, int i, int j, int k, double &Lb, double &Rb) {
  const int q = coord<D>(i, j, k);
  if (q == A.lo[D]) { double o = Rb; bc_pair(o, Rb, A.phys[D][0], 0, true, c == D, ld<D>(u, i, j, k, -1, c), false); }
  if (q == A.hi[D]) { double o = Lb; bc_pair(Lb, o, A.phys[D][1], 1, true, c == D, ld<D>(u, i, j, k, 1, c), D == 0 && !A.outlet2d); }   // :2075 quirk
}
template <int D> DEVI void vp_face_bc(const GArgs &A, const FV &u, int c, bool normal, bool quirk_ok, int i, int j, int k, double uc, double &L, double &R) {
  const int side = face_side<D>(A, i, j, k);
  if (side >= 0) bc_pair(L, R, A.phys[D][side], side, true, normal, side == 0 ? ld<D>(u, i, j, k, -1, c) : uc, quirk_ok && D == 0 && side == 1 && !A.outlet2d);
}
template <int D> DEVI void vp_B_emit(const FV &UI, int i, int j, int k, const double L[3], const double R[3], double eps) {
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
#define VP_LOAD_CELL                                                                                                    \
    double uc[3], s0[3], s1[3], s2[3], ft[3] = { 0.0, 0.0, 0.0 };                                                         \
    _Pragma("unroll") for (int c = 0; c < 3; c++) {                                                                      \
      uc[c] = fv_get(u, ic, jc, kc, c); s0[c] = fv_get(sl0, ic, jc, kc, c); s1[c] = fv_get(sl1, ic, jc, kc, c); s2[c] = fv_get(sl2, ic, jc, kc, c); \
    }

template <bool R, bool BC> __device__ __forceinline__ void vp_B_m_body(typename Prm<FV, R>::type u, typename Prm<FV, R>::type sl0, typename Prm<FV, R>::type sl1, typename Prm<FV, R>::type sl2, typename Prm<FV, R>::type force, typename Prm<FV, R>::type UI, typename Prm<GArgs, R>::type A, typename Prm<Range3, R>::type r, int klen, const double *umax, const int BX, const int BY, const int BZ) {
  __shared__ double ly[2][3][TNY][64];
  MARCH_SETUP(r)
  const double eps = eps_from(umax);
  const double dt2 = 0.5 * A.dt;
  double Lz[3] = { 0.0, 0.0, 0.0 };
  for (int k = k0 - 1; k <= k1; k++) {
    MARCH_PLANE
    VP_LOAD_CELL
    if (A.use_minion) {
      #pragma unroll
      for (int c = 0; c < 3; c++) ft[c] = dt2 * fv_get(force, ic, jc, kc, c);
    }
    double Lb[3][3], Rb[3][3];
    vp_bases<0>(A, uc, s0, ft, Lb[0], Rb[0]);
    vp_bases<1>(A, uc, s1, ft, Lb[1], Rb[1]);
    vp_bases<2>(A, uc, s2, ft, Lb[2], Rb[2]);
    #pragma unroll
    for (int c = 0; c < 3; c++) ly[buf][c][row][lane] = Lb[1][c];
    __syncthreads();
    double Lx[3], Ly[3], Lzc[3];
    #pragma unroll
    for (int c = 0; c < 3; c++) {
      Lx[c] = shfl_prev(Lb[0][c]);
      Ly[c] = ly[buf][c][row >= 1 ? row - 1 : 0][lane];
      Lzc[c] = Lz[c]; Lz[c] = Lb[2][c];
    }
    if (emit) {
      if (BC && edge) {
        #pragma unroll
        for (int c = 0; c < 3; c++) {
          vp_face_bc<0>(A, u, c, c == 0, true, i, j, k, uc[c], Lx[c], Rb[0][c]);
          vp_face_bc<1>(A, u, c, c == 1, true, i, j, k, uc[c], Ly[c], Rb[1][c]);
          vp_face_bc<2>(A, u, c, c == 2, true, i, j, k, uc[c], Lzc[c], Rb[2][c]);
        }
      }
      if (i >= A.lo[0]) vp_B_emit<0>(UI, i, j, k, Lx, Rb[0], eps);
      if (j >= A.lo[1]) vp_B_emit<1>(UI, i, j, k, Ly, Rb[1], eps);
      if (k >= A.lo[2]) vp_B_emit<2>(UI, i, j, k, Lzc, Rb[2], eps);
    }
  }
}
template <bool BC> __global__ void __launch_bounds__(64 * TNY) kk_vp_B_m(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, GArgs A, Range3 r, int klen, const double *umax) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  vp_B_m_body<false, BC>(u, sl0, sl1, sl2, force, UI, A, r, klen, umax, bx_, by_, bz_);
}


// stage C: XC(C,D) = component C on D-faces corrected by the third direction O
DEVI double vp_up(double un, double L, double R, double eps) {
  const double v = (un > 0.0) ? L : R;
  const double av = 0.5 * (L + R);
  return (fabs(un) < eps) ? av : v;
}
template <bool R, bool BC> __device__ __forceinline__ void vp_C_m_body(typename Prm<FV, R>::type u, typename Prm<FV, R>::type sl0, typename Prm<FV, R>::type sl1, typename Prm<FV, R>::type sl2, typename Prm<FV, R>::type force, typename Prm<FV, R>::type UI, typename Prm<FV, R>::type XC, typename Prm<GArgs, R>::type A, typename Prm<Range3, R>::type r, int klen, const double *umax, const int BX, const int BY, const int BZ) {
  __shared__ double ly[2][2][TNY][64];
  MARCH_SETUP(r)
  const double eps = eps_from(umax);
  const double dt2 = 0.5 * A.dt, dt6 = A.dt / 6.0;
  double Lz[2] = { 0.0, 0.0 };
  for (int k = k0 - 1; k <= k1; k++) {
    MARCH_PLANE
    VP_LOAD_CELL
    // w0[O][c] / w1[O][c]: UI[O*3+c] at the cell's lower / upper O-face
    double w0[3][3], w1[3][3];
    #pragma unroll
    for (int c = 0; c < 3; c++) {
      w0[0][c] = fv_get(UI, ic, jc, kc, 0 + c); w1[0][c] = fv_get(UI, ip, jc, kc, 0 + c);
      w0[1][c] = fv_get(UI, ic, jc, kc, 3 + c); w1[1][c] = fv_get(UI, ic, jp, kc, 3 + c);
      w0[2][c] = fv_get(UI, ic, jc, kc, 6 + c); w1[2][c] = fv_get(UI, ic, jc, kp, 6 + c);
    }
    if (A.use_minion) {
      #pragma unroll
      for (int c = 0; c < 3; c++) ft[c] = dt2 * fv_get(force, ic, jc, kc, c);
    }
    double Lb[3][3], Rb[3][3];
    vp_bases<0>(A, uc, s0, ft, Lb[0], Rb[0]);
    vp_bases<1>(A, uc, s1, ft, Lb[1], Rb[1]);
    vp_bases<2>(A, uc, s2, ft, Lb[2], Rb[2]);
    if (BC && edge) {
      #pragma unroll
      for (int c = 0; c < 3; c++) { vp_premod<0>(A, u, c, ic, jc, kc, Lb[0][c], Rb[0][c]); vp_premod<1>(A, u, c, ic, jc, kc, Lb[1][c], Rb[1][c]); vp_premod<2>(A, u, c, ic, jc, kc, Lb[2][c], Rb[2][c]); }
    }
    // t[C][O] = (dt6/dx_O) * (UI[OO](+) + UI[OO]) * (UI[OC](+) - UI[OC])
    double t[3][3];
    #pragma unroll
    for (int O = 0; O < 3; O++) {
      const double sm = w1[O][O] + w0[O][O];
      #pragma unroll
      for (int C = 0; C < 3; C++) t[C][O] = (dt6 / A.dx[O]) * sm * (w1[O][C] - w0[O][C]);
    }
    // VL[D][n], VR[D][n]: D = 0: C = 1, 2;  D = 1: C = 0, 2;  D = 2: C = 0, 1
    double VL[3][2], VR[3][2];
    VL[0][0] = Lb[0][1] - t[1][2]; VR[0][0] = Rb[0][1] - t[1][2];   // D=0 C=1 O=2
    VL[0][1] = Lb[0][2] - t[2][1]; VR[0][1] = Rb[0][2] - t[2][1];   // D=0 C=2 O=1
    VL[1][0] = Lb[1][0] - t[0][2]; VR[1][0] = Rb[1][0] - t[0][2];   // D=1 C=0 O=2
    VL[1][1] = Lb[1][2] - t[2][0]; VR[1][1] = Rb[1][2] - t[2][0];   // D=1 C=2 O=0
    VL[2][0] = Lb[2][0] - t[0][1]; VR[2][0] = Rb[2][0] - t[0][1];   // D=2 C=0 O=1
    VL[2][1] = Lb[2][1] - t[1][0]; VR[2][1] = Rb[2][1] - t[1][0];   // D=2 C=1 O=0
    ly[buf][0][row][lane] = VL[1][0]; ly[buf][1][row][lane] = VL[1][1];
    __syncthreads();
    double L[3][2];
    L[0][0] = shfl_prev(VL[0][0]); L[0][1] = shfl_prev(VL[0][1]);
    L[1][0] = ly[buf][0][row >= 1 ? row - 1 : 0][lane]; L[1][1] = ly[buf][1][row >= 1 ? row - 1 : 0][lane];
    L[2][0] = Lz[0]; L[2][1] = Lz[1];
    Lz[0] = VL[2][0]; Lz[1] = VL[2][1];
    if (emit) {
      if (BC && edge) {
        vp_face_bc<0>(A, u, 1, false, false, i, j, k, uc[1], L[0][0], VR[0][0]); vp_face_bc<0>(A, u, 2, false, false, i, j, k, uc[2], L[0][1], VR[0][1]);
        vp_face_bc<1>(A, u, 0, false, false, i, j, k, uc[0], L[1][0], VR[1][0]); vp_face_bc<1>(A, u, 2, false, false, i, j, k, uc[2], L[1][1], VR[1][1]);
        vp_face_bc<2>(A, u, 0, false, false, i, j, k, uc[0], L[2][0], VR[2][0]); vp_face_bc<2>(A, u, 1, false, false, i, j, k, uc[1], L[2][1], VR[2][1]);
      }
      // the advecting normal velocity on the cell's lower faces: UI[DD] = w0[D][D]
      if (i >= A.lo[0]) { if (vz) fv_at(XC, i, j, k, xc_idx(1, 0)) = vp_up(w0[0][0], L[0][0], VR[0][0], eps); if (vy) fv_at(XC, i, j, k, xc_idx(2, 0)) = vp_up(w0[0][0], L[0][1], VR[0][1], eps); }
      if (j >= A.lo[1]) { if (vz) fv_at(XC, i, j, k, xc_idx(0, 1)) = vp_up(w0[1][1], L[1][0], VR[1][0], eps); if (vx) fv_at(XC, i, j, k, xc_idx(2, 1)) = vp_up(w0[1][1], L[1][1], VR[1][1], eps); }
      if (k >= A.lo[2]) { if (vy) fv_at(XC, i, j, k, xc_idx(0, 2)) = vp_up(w0[2][2], L[2][0], VR[2][0], eps); if (vx) fv_at(XC, i, j, k, xc_idx(1, 2)) = vp_up(w0[2][2], L[2][1], VR[2][1], eps); }
    }
  }
}
template <bool BC> __global__ void __launch_bounds__(64 * TNY) kk_vp_C_m(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, FV XC, GArgs A, Range3 r, int klen, const double *umax) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  vp_C_m_body<false, BC>(u, sl0, sl1, sl2, force, UI, XC, A, r, klen, umax, bx_, by_, bz_);
}


// stage D: the MAC velocity on valid D-faces
template <int D> DEVI double vp_D_face(const GArgs &A, const FV &u, int i, int j, int k, double ucD, double L, double R, double eps, bool edge) {
  const double uavg = 0.5 * (L + R);
  const bool test = ((L <= 0.0 && R >= 0.0) || (fabs(L + R) < eps));
  double v = (uavg > 0.0) ? L : R;
  v = test ? 0.0 : v;
  if (edge) {
    const int side = face_side<D>(A, i, j, k);
    if (side >= 0) {                                   // velpred.f90:2642-2659
      const int ph = A.phys[D][side];
      if (ph == VDN_SLIP_WALL || ph == VDN_NO_SLIP_WALL) v = 0.0;
      else if (ph == VDN_INLET) v = (side == 0) ? ld<D>(u, i, j, k, -1, D) : ucD;
      else if (ph == VDN_OUTLET) v = (side == 0) ? fmin(R, 0.0) : fmax(L, 0.0);
    }
  }
  return v;
}
template <bool R, bool BC> __device__ __forceinline__ void vp_D_m_body(typename Prm<FV, R>::type u, typename Prm<FV, R>::type sl0, typename Prm<FV, R>::type sl1, typename Prm<FV, R>::type sl2, typename Prm<FV, R>::type force, typename Prm<FV, R>::type UI, typename Prm<FV, R>::type XC, typename Prm<FV, R>::type um, typename Prm<FV, R>::type vm, typename Prm<FV, R>::type wm, typename Prm<GArgs, R>::type A, typename Prm<Range3, R>::type r, int klen, const double *umax, const int BX, const int BY, const int BZ) {
  __shared__ double ly[2][TNY][64];
  MARCH_SETUP(r)
  const double eps = eps_from(umax);
  const double dt2 = 0.5 * A.dt, dt4 = A.dt / 4.0;
  double Lz = 0.0;
  for (int k = k0 - 1; k <= k1; k++) {
    MARCH_PLANE
    double uc[3], sd[3], ft[3];
    #pragma unroll
    for (int c = 0; c < 3; c++) { uc[c] = fv_get(u, ic, jc, kc, c); ft[c] = dt2 * fv_get(force, ic, jc, kc, c); }
    sd[0] = fv_get(sl0, ic, jc, kc, 0); sd[1] = fv_get(sl1, ic, jc, kc, 1); sd[2] = fv_get(sl2, ic, jc, kc, 2);
    // g[T] = UI[TT](+) + UI[TT];  x0/x1[n]: XC component n at the lower / upper face of its direction
    //   XC index C*2 + (D > C ? D-1 : D): n 0 = (C0,D1) 1 = (C0,D2) 2 = (C1,D0) 3 = (C1,D2) 4 = (C2,D0) 5 = (C2,D1)
    double g[3];
    g[0] = fv_get(UI, ip, jc, kc, 0) + fv_get(UI, ic, jc, kc, 0);
    g[1] = fv_get(UI, ic, jp, kc, 4) + fv_get(UI, ic, jc, kc, 4);
    g[2] = fv_get(UI, ic, jc, kp, 8) + fv_get(UI, ic, jc, kc, 8);
    double x0[6], x1[6];
    #pragma unroll
    for (int n = 0; n < 6; n++) x0[n] = fv_get(XC, ic, jc, kc, n);
    x1[0] = fv_get(XC, ic, jp, kc, 0); x1[1] = fv_get(XC, ic, jc, kp, 1); x1[2] = fv_get(XC, ip, jc, kc, 2);
    x1[3] = fv_get(XC, ic, jc, kp, 3); x1[4] = fv_get(XC, ip, jc, kc, 4); x1[5] = fv_get(XC, ic, jp, kc, 5);
    // base of the NORMAL component only (vp_pair component D along D)
    double Lb[3], Rb[3];
    #define NBASE(Dd)                                                                                              \
      { double cl; if (Dd == 1) cl = dt2 * fmax(0.0, uc[Dd] / A.dx[1]); else cl = dt2 * fmax(0.0, uc[Dd]) / A.dx[Dd];  \
        const double cr = dt2 * fmin(0.0, uc[Dd]) / A.dx[Dd];                                                        \
        Lb[Dd] = uc[Dd] + (0.5 - cl) * sd[Dd]; Rb[Dd] = uc[Dd] - (0.5 + cr) * sd[Dd];                                \
        if (A.use_minion) { Lb[Dd] = Lb[Dd] + ft[Dd]; Rb[Dd] = Rb[Dd] + ft[Dd]; } }
    NBASE(0) NBASE(1) NBASE(2)
    #undef NBASE
    if (BC && edge) { vp_premod<0>(A, u, 0, ic, jc, kc, Lb[0], Rb[0]); vp_premod<1>(A, u, 1, ic, jc, kc, Lb[1], Rb[1]); vp_premod<2>(A, u, 2, ic, jc, kc, Lb[2], Rb[2]); }
    double VL[3], VR[3];
    {   // D = 0: T1 = 1 with XC(0,1) = n0,  T2 = 2 with XC(0,2) = n1
      const double a1 = (dt4 / A.dx[1]) * g[1] * (x1[0] - x0[0]);
      const double a2 = (dt4 / A.dx[2]) * g[2] * (x1[1] - x0[1]);
      double vl = Lb[0] - a1 - a2, vr = Rb[0] - a1 - a2;
      if (!A.use_minion) { vl = vl + ft[0]; vr = vr + ft[0]; }
      VL[0] = vl; VR[0] = vr;
    }
    {   // D = 1: T1 = 0 with XC(1,0) = n2,  T2 = 2 with XC(1,2) = n3
      const double a1 = (dt4 / A.dx[0]) * g[0] * (x1[2] - x0[2]);
      const double a2 = (dt4 / A.dx[2]) * g[2] * (x1[3] - x0[3]);
      double vl = Lb[1] - a1 - a2, vr = Rb[1] - a1 - a2;
      if (!A.use_minion) { vl = vl + ft[1]; vr = vr + ft[1]; }
      VL[1] = vl; VR[1] = vr;
    }
    {   // D = 2: T1 = 0 with XC(2,0) = n4,  T2 = 1 with XC(2,1) = n5
      const double a1 = (dt4 / A.dx[0]) * g[0] * (x1[4] - x0[4]);
      const double a2 = (dt4 / A.dx[1]) * g[1] * (x1[5] - x0[5]);
      double vl = Lb[2] - a1 - a2, vr = Rb[2] - a1 - a2;
      if (!A.use_minion) { vl = vl + ft[2]; vr = vr + ft[2]; }
      VL[2] = vl; VR[2] = vr;
    }
    ly[buf][row][lane] = VL[1];
    __syncthreads();
    const double Lx = shfl_prev(VL[0]);
    const double Ly = ly[buf][row >= 1 ? row - 1 : 0][lane];
    const double Lzc = Lz;
    Lz = VL[2];
    if (emit) {
      if (vy && vz) fv_at(um, i, j, k) = vp_D_face<0>(A, u, i, j, k, uc[0], Lx, VR[0], eps, BC && edge);
      if (vx && vz) fv_at(vm, i, j, k) = vp_D_face<1>(A, u, i, j, k, uc[1], Ly, VR[1], eps, BC && edge);
      if (vx && vy) fv_at(wm, i, j, k) = vp_D_face<2>(A, u, i, j, k, uc[2], Lzc, VR[2], eps, BC && edge);
    }
  }
}
template <bool BC> __global__ void __launch_bounds__(64 * TNY) kk_vp_D_m(FV u, FV sl0, FV sl1, FV sl2, FV force, FV UI, FV XC, FV um, FV vm, FV wm, GArgs A, Range3 r, int klen, const double *umax) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  vp_D_m_body<false, BC>(u, sl0, sl1, sl2, force, UI, XC, um, vm, wm, A, r, klen, umax, bx_, by_, bz_);
}


// ---- velpred: stages B + C + D in one march (all three components; the scheme of mk_F_m_body) --------------------------------------------
// Iteration kk loads plane kk (u, nine slopes, force) and runs stage B on plane kk (UI: the nine upwinded states on the cell's lower faces),
// stage C on plane kk-1 (XC: six transverse-corrected states) and stage D on plane kk-2 (the MAC velocity on the lower faces).  UI and XC
// stay in registers; upper-face values come from the next lane (DPP), the next row (LDS, written one iteration earlier) and, along z, from the
// stage that ran earlier in the same iteration.  A cell carries its eighteen bases from stage B to stage C and the three normal ones on to
// stage D (the separate stages recompute them: six f64 divisions per cell and stage).  Every expression is the one of vp_B_m_body /
// vp_C_m_body / vp_D_m_body in the same order (tests/test_kernels_gpu.py::test_velpred, test_godunov_marching_equals_face_centred).
// Boundary rules as packed per-thread codes (mk_F_m_body): bc_pair leaves L = R = v, v one of {ghost, 0, inner, clamped inner}; the
// hi-x OUTLET quirk of velpred.f90:2075 (min instead of max on the normal component) is the code 4 in place of 5 in stages B / premod.
struct VArgs {
  const char *pu, *psl[3], *pf;       // component 0 of u, the three slope fabs and the force at plane KB
  long sc_u, sc_sl, sc_f;             // bytes between components
  long sp_u, sp_sl, sp_f; FGeo g_u, g_sl, g_f;
  char *q[3]; long sq[3]; FGeo h[3];  // umac, vmac, wmac
  long u_row;
  double dt2, dx[3], tC[3], tD[3];    // tC[O] = (dt/6) / dx[O],  tD[T] = (dt/4) / dx[T]
  double idx[3]; int p2;              // as in FArgs: 1 / dx, every dx a power of two (P2 kernels multiply)
  int lo[3], hi[3], phys[3][2], use_minion, outlet2d;
};
DEVI int vp_code_B(int phys, int D, int side, int c, int outlet2d) {
  if (phys == VDN_INLET) return 1;
  if (phys == VDN_SLIP_WALL) return c == D ? 2 : 3;
  if (phys == VDN_NO_SLIP_WALL) return 2;
  if (phys == VDN_OUTLET) return c == D ? ((side == 0 || (D == 0 && !outlet2d)) ? 4 : 5) : 3;
  return 0;
}
DEVI int vp_code_D(int phys, int side) {
  if (phys == VDN_SLIP_WALL || phys == VDN_NO_SLIP_WALL) return 2;
  if (phys == VDN_INLET) return 1;
  if (phys == VDN_OUTLET) return side ? 5 : 4;
  return 0;
}
template <int D> DEVI void vpf_emit(double ui[3], const double L[3], const double R[3], double eps) {      // vp_B_emit into registers
  const double uavg = 0.5 * (L[D] + R[D]);
  const bool test = ((L[D] <= 0.0 && R[D] >= 0.0) || (fabs(L[D] + R[D]) < eps));
  double un = (uavg > 0.0) ? L[D] : R[D];
  un = test ? 0.0 : un;
  #pragma unroll
  for (int c = 0; c < 3; c++) {
    if (c == D) ui[c] = un;
    else {
      const double v = (un > 0.0) ? L[c] : R[c];
      const double av = 0.5 * (L[c] + R[c]);
      ui[c] = (fabs(un) < eps) ? av : v;
    }
  }
}
DEVI double vpf_riemann(double L, double R, double eps) {            // the state of vp_D_face before its boundary rule
  const double uavg = 0.5 * (L + R);
  const bool test = ((L <= 0.0 && R >= 0.0) || (fabs(L + R) < eps));
  const double v = (uavg > 0.0) ? L : R;
  return test ? 0.0 : v;
}
template <bool BC, bool INL, bool PW2, bool NAR = false> __device__ __forceinline__ void vp_F_m_body(const VArgs &F, const Range3 &r, int klen, const double *umax, const int BX, const int BY, const int BZ, const int segw = 64) {
  __shared__ double lB[3][TNY][64], lUI[3][TNY][64], lC[2][TNY][64], lXC[2][TNY][64], lD[TNY][64];
  const SegGeo G = seg_geo<NAR>(segw);
  const int lane = G.lane, row = G.row;
  const int ownx = NAR ? G.W - 2 : FNX, owny = NAR ? G.ROWS - 2 : FNY;
  const int i = r.lo[0] - 1 + BX * ownx + lane, j = r.lo[1] - 1 + BY * owny + row;
  const bool own_ij = G.live && lane >= 1 && lane <= ownx && row >= 1 && row <= owny && i <= r.hi[0] && j <= r.hi[1];
  const int ic = min(max(i, F.lo[0] - 1), F.hi[0] + 1), jc = min(max(j, F.lo[1] - 1), F.hi[1] + 1);
  const bool ing_ij = i == ic && j == jc;
  const bool vx = i >= F.lo[0] && i <= F.hi[0], vy = j >= F.lo[1] && j <= F.hi[1];
  const int rowm = G.rowm, rowp = G.rowp;
  const int k0 = r.lo[2] + BZ * klen, k1 = min(k0 + klen - 1, r.hi[2]);
  const int KB = F.lo[2] - 1, KT = F.hi[2] + 1;
  const double eps = eps_from(umax);
  // packed boundary codes of this thread: wf: x-face (3 comps x 4 bits, +8 = hi side), y-face (bits 12..23), stage-D code of the x-face (24..27) and
  // of the y-face (28..31);  wpx / wpy: rule on the bases of the lowest valid cell (3 x 4 bits) and of the highest (bits 12..23)
  unsigned wf = 0, wpx = 0, wpy = 0;
  if (BC) {
    #pragma unroll
    for (int c = 0; c < 3; c++) {
      if (ing_ij && i == F.lo[0]) wf |= (unsigned)vp_code_B(F.phys[0][0], 0, 0, c, F.outlet2d) << (4 * c);
      if (ing_ij && i == F.hi[0] + 1) { const int cd = vp_code_B(F.phys[0][1], 0, 1, c, F.outlet2d); if (cd) wf |= (unsigned)(cd | 8) << (4 * c); }
      if (ing_ij && j == F.lo[1]) wf |= (unsigned)vp_code_B(F.phys[1][0], 1, 0, c, F.outlet2d) << (12 + 4 * c);
      if (ing_ij && j == F.hi[1] + 1) { const int cd = vp_code_B(F.phys[1][1], 1, 1, c, F.outlet2d); if (cd) wf |= (unsigned)(cd | 8) << (12 + 4 * c); }
      if (ic == F.lo[0]) wpx |= (unsigned)vp_code_B(F.phys[0][0], 0, 0, c, F.outlet2d) << (4 * c);
      if (ic == F.hi[0]) wpx |= (unsigned)vp_code_B(F.phys[0][1], 0, 1, c, F.outlet2d) << (12 + 4 * c);
      if (jc == F.lo[1]) wpy |= (unsigned)vp_code_B(F.phys[1][0], 1, 0, c, F.outlet2d) << (4 * c);
      if (jc == F.hi[1]) wpy |= (unsigned)vp_code_B(F.phys[1][1], 1, 1, c, F.outlet2d) << (12 + 4 * c);
    }
    if (ing_ij && i == F.lo[0]) wf |= (unsigned)vp_code_D(F.phys[0][0], 0) << 24;
    if (ing_ij && i == F.hi[0] + 1) { const int cd = vp_code_D(F.phys[0][1], 1); if (cd) wf |= (unsigned)(cd | 8) << 24; }
    if (ing_ij && j == F.lo[1]) wf |= (unsigned)vp_code_D(F.phys[1][0], 0) << 28;
    if (ing_ij && j == F.hi[1] + 1) { const int cd = vp_code_D(F.phys[1][1], 1); if (cd) wf |= (unsigned)(cd | 8) << 28; }
  }
  // per-thread plane pointers (the thread's column): inputs at the plane the next loads take (kcur), outputs at the plane stage D emits next
  int kcur = min(max(k0 - 2, KB), KT);
  const char *pu = F.pu + (long)(kcur - KB) * F.sp_u + fg_off(F.g_u, ic, jc);
  const char *ps0 = F.psl[0] + (long)(kcur - KB) * F.sp_sl + fg_off(F.g_sl, ic, jc), *ps1 = F.psl[1] + (long)(kcur - KB) * F.sp_sl + fg_off(F.g_sl, ic, jc), *ps2 = F.psl[2] + (long)(kcur - KB) * F.sp_sl + fg_off(F.g_sl, ic, jc);
  const char *pf = F.pf + (long)(kcur - KB) * F.sp_f + fg_off(F.g_f, ic, jc);
  char *qu = F.q[0] + (long)(k0 - KB) * F.sq[0] + fg_off(F.h[0], i, j), *qv = F.q[1] + (long)(k0 - KB) * F.sq[1] + fg_off(F.h[1], i, j), *qw = F.q[2] + (long)(k0 - KB) * F.sq[2] + fg_off(F.h[2], i, j);
  // state carried from plane to plane
  double Lb1[3][3], Rb1[3][3], ui1[3][3];          // plane kk-1: bases [D][c], UI [D][c] on the lower faces
  double LbD[3], RbD[3], sm2[3], xc2[6];           // plane kk-2: normal bases, UI[T][T](+) + UI[T][T], XC on the lower faces
  #pragma unroll
  for (int d = 0; d < 3; d++) { LbD[d] = 0.0; RbD[d] = 0.0; sm2[d] = 0.0; for (int c = 0; c < 3; c++) { Lb1[d][c] = 0.0; Rb1[d][c] = 0.0; ui1[d][c] = 0.0; } }
  #pragma unroll
  for (int n = 0; n < 6; n++) xc2[n] = 0.0;
  double LzB[3] = { 0.0, 0.0, 0.0 }, LzC[2] = { 0.0, 0.0 }, LzD = 0.0;
  #define U_AT(kpl, dbytes, c) (INL ? ldd(pu + (long)((kpl) - kcur) * F.sp_u + (dbytes) + (long)(c) * F.sc_u, 0u) : 0.0)
  #define BC_V(v, code, in, ghost_expr) { const int cd_ = (code); const double in_ = (in); double v_ = in_;                      \
      if (cd_ == 4) v_ = fmin(in_, 0.0); if (cd_ == 5) v_ = fmax(in_, 0.0); if (cd_ == 2) v_ = 0.0; if (cd_ == 1) v_ = (ghost_expr); v = v_; }
  // the face rule on a pair (4-bit field f4: code, +8 = hi face): both states become v
  #define PAIR_BC(L, Rr, f4, own_expr, ghost_expr) { const int f_ = (int)(f4) & 15;                                              \
      if (f_) { const bool hi_ = (f_ & 8) != 0; double o_; BC_V(o_, f_ & 7, hi_ ? (L) : (Rr), hi_ ? (own_expr) : (ghost_expr)) L = o_; Rr = o_; } }
  #define OUT_BC(out, f4, L, Rr, own_expr, ghost_expr) { const int f_ = (int)(f4) & 15;                                          \
      if (f_) { const bool hi_ = (f_ & 8) != 0; double o_; BC_V(o_, f_ & 7, hi_ ? (L) : (Rr), hi_ ? (own_expr) : (ghost_expr)) out = o_; } }
  #define PREMOD(Lb_, Rb_, a4, b4, gl_expr, gh_expr) { const int a_ = (int)(a4) & 7; if (a_) { double o_; BC_V(o_, a_, Rb_, gl_expr) Rb_ = o_; }  \
                                                       const int b_ = (int)(b4) & 7; if (b_) { double o_; BC_V(o_, b_, Lb_, gh_expr) Lb_ = o_; } }
  // z words of a stage plane (uniform): face codes of the three components (bits 0..11), stage-D code (24..27); premod lo (0..11) / hi (12..23)
  #define ZF_WORD(k, kc, zf) { zf = 0u; if (BC && (k) == (kc)) {                                                                  \
      if ((k) == F.lo[2]) { for (int c_ = 0; c_ < 3; c_++) zf |= (unsigned)vp_code_B(F.phys[2][0], 2, 0, c_, F.outlet2d) << (4 * c_); zf |= (unsigned)vp_code_D(F.phys[2][0], 0) << 24; } \
      if ((k) == KT) { for (int c_ = 0; c_ < 3; c_++) { const int cd_ = vp_code_B(F.phys[2][1], 2, 1, c_, F.outlet2d); if (cd_) zf |= (unsigned)(cd_ | 8) << (4 * c_); }             \
                       const int cd_ = vp_code_D(F.phys[2][1], 1); if (cd_) zf |= (unsigned)(cd_ | 8) << 24; } } }
  #define ZP_WORD(kc, zp) { zp = 0u; if (BC) {                                                                                     \
      if ((kc) == F.lo[2]) for (int c_ = 0; c_ < 3; c_++) zp |= (unsigned)vp_code_B(F.phys[2][0], 2, 0, c_, F.outlet2d) << (4 * c_);           \
      if ((kc) == F.hi[2]) for (int c_ = 0; c_ < 3; c_++) zp |= (unsigned)vp_code_B(F.phys[2][1], 2, 1, c_, F.outlet2d) << (12 + 4 * c_); } }
  struct VRaw { double u[3], sx[3], sy[3], sz[3], f[3]; } N;
  #define V_LOAD {                                                                                                \
      _Pragma("unroll") for (int c = 0; c < 3; c++) {                                                              \
        N.u[c] = ldd(pu + c * F.sc_u, 0u); N.sx[c] = ldd(ps0 + c * F.sc_sl, 0u); N.sy[c] = ldd(ps1 + c * F.sc_sl, 0u); \
        N.sz[c] = ldd(ps2 + c * F.sc_sl, 0u); N.f[c] = ldd(pf + c * F.sc_f, 0u); } }
  #define V_ADVANCE(kn) { const int kc_ = min(max((kn), KB), KT);                                                 \
      if (kc_ != kcur) { pu += F.sp_u; ps0 += F.sp_sl; ps1 += F.sp_sl; ps2 += F.sp_sl; pf += F.sp_f; kcur = kc_; } }
  double fa[3] = { 0.0, 0.0, 0.0 }, fb[3] = { 0.0, 0.0, 0.0 };       // dt/2 force of planes kk-1, kk-2 (stage D adds it when !use_minion)
  for (int kk = k0 - 2; kk <= k1 + 2; kk++) {
    // ---------------- plane kk: loads, bases ----------------
    double uc[3], Lb[3][3], Rb[3][3], ft[3];
    {
      V_LOAD  V_ADVANCE(kk + 1)
      #pragma unroll
      for (int c = 0; c < 3; c++) { uc[c] = N.u[c]; ft[c] = F.dt2 * N.f[c]; }
      const double *sl[3] = { N.sx, N.sy, N.sz };
      #pragma unroll
      for (int d = 0; d < 3; d++) {
        double cfl_l;
        if (d == 1) cfl_l = F.dt2 * fmax(0.0, DIVDX(uc[d], 1));       // velpred.f90:2108: division inside max()
        else cfl_l = DIVDX(F.dt2 * fmax(0.0, uc[d]), d);
        const double cfl_r = DIVDX(F.dt2 * fmin(0.0, uc[d]), d);
        #pragma unroll
        for (int c = 0; c < 3; c++) {
          Lb[d][c] = uc[c] + (0.5 - cfl_l) * sl[d][c];
          Rb[d][c] = uc[c] - (0.5 + cfl_r) * sl[d][c];
          if (F.use_minion) { Lb[d][c] = Lb[d][c] + ft[c]; Rb[d][c] = Rb[d][c] + ft[c]; }
        }
      }
    }
    // values of the next row written in the previous iteration: read before this iteration overwrites them
    double uiy_up[3], xcy_up[2];
    #pragma unroll
    for (int c = 0; c < 3; c++) uiy_up[c] = LP(lUI[c]);
    xcy_up[0] = LP(lXC[0]); xcy_up[1] = LP(lXC[1]);
    // ---------------- stage B, plane kk ----------------
    double ui0[3][3];
    {
      const int k = kk, kc = min(max(k, KB), KT);
      unsigned zf; ZF_WORD(k, kc, zf)
      #pragma unroll
      for (int c = 0; c < 3; c++) LS(lB[c]) = Lb[1][c];
      __syncthreads();
      double Lx[3], Ly[3], Lzc[3], Rx[3], Ry[3], Rz[3];
      #pragma unroll
      for (int c = 0; c < 3; c++) {
        Lx[c] = shfl_prev(Lb[0][c]); Ly[c] = LM(lB[c]); Lzc[c] = LzB[c]; LzB[c] = Lb[2][c];
        Rx[c] = Rb[0][c]; Ry[c] = Rb[1][c]; Rz[c] = Rb[2][c];
      }
      if (BC && k == kc) {
        unsigned w = wf; asm volatile("" : "+v"(w));
        if (w & 0xFFFFFFu) {
          #pragma unroll
          for (int c = 0; c < 3; c++) {
            PAIR_BC(Lx[c], Rx[c], w >> (4 * c), uc[c], U_AT(kc, -8L, c))
            PAIR_BC(Ly[c], Ry[c], w >> (12 + 4 * c), uc[c], U_AT(kc, -F.u_row, c))
          }
        }
        if ((zf & 0xFFFu) && ing_ij) {
          #pragma unroll
          for (int c = 0; c < 3; c++) PAIR_BC(Lzc[c], Rz[c], zf >> (4 * c), uc[c], U_AT(kc - 1, 0L, c))
        }
      }
      vpf_emit<0>(ui0[0], Lx, Rx, eps); vpf_emit<1>(ui0[1], Ly, Ry, eps); vpf_emit<2>(ui0[2], Lzc, Rz, eps);
      #pragma unroll
      for (int c = 0; c < 3; c++) LS(lUI[c]) = ui0[1][c];      // read by the row below at the top of the next iteration
    }
    // ---------------- stage C, plane kk-1 ----------------
    double xc1[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 }, sm1[3] = { 0.0, 0.0, 0.0 };
    if (kk - 1 >= k0 - 1) {
      const int k = kk - 1, kc = min(max(k, KB), KT);
      unsigned zf, zp; ZF_WORD(k, kc, zf) ZP_WORD(kc, zp)
      // w0[O][c] = ui1[O][c]; w1[O][c]: the upper O-face
      double w1[3][3];
      #pragma unroll
      for (int c = 0; c < 3; c++) { w1[0][c] = lane_next(ui1[0][c]); w1[1][c] = uiy_up[c]; w1[2][c] = ui0[2][c]; }
      double Lc[3][3], Rc[3][3];
      #pragma unroll
      for (int d = 0; d < 3; d++) for (int c = 0; c < 3; c++) { Lc[d][c] = Lb1[d][c]; Rc[d][c] = Rb1[d][c]; }
      unsigned wfc = 0;
      if (BC) {
        unsigned px = wpx, py = wpy; asm volatile("" : "+v"(px)); asm volatile("" : "+v"(py));
        wfc = wf; asm volatile("" : "+v"(wfc));
        if (px | py) {
          #pragma unroll
          for (int c = 0; c < 3; c++) {
            PREMOD(Lc[0][c], Rc[0][c], px >> (4 * c), px >> (12 + 4 * c), U_AT(kc, -8L, c), U_AT(kc, 8L, c))
            PREMOD(Lc[1][c], Rc[1][c], py >> (4 * c), py >> (12 + 4 * c), U_AT(kc, -F.u_row, c), U_AT(kc, F.u_row, c))
          }
        }
        if (zp) {
          #pragma unroll
          for (int c = 0; c < 3; c++) PREMOD(Lc[2][c], Rc[2][c], zp >> (4 * c), zp >> (12 + 4 * c), U_AT(kc - 1, 0L, c), U_AT(kc + 1, 0L, c))
        }
      }
      double t[3][3];
      #pragma unroll
      for (int O = 0; O < 3; O++) {
        sm1[O] = w1[O][O] + ui1[O][O];
        #pragma unroll
        for (int C = 0; C < 3; C++) t[C][O] = F.tC[O] * sm1[O] * (w1[O][C] - ui1[O][C]);
      }
      double VL[3][2], VR[3][2];
      VL[0][0] = Lc[0][1] - t[1][2]; VR[0][0] = Rc[0][1] - t[1][2];   // D=0 C=1 O=2
      VL[0][1] = Lc[0][2] - t[2][1]; VR[0][1] = Rc[0][2] - t[2][1];   // D=0 C=2 O=1
      VL[1][0] = Lc[1][0] - t[0][2]; VR[1][0] = Rc[1][0] - t[0][2];   // D=1 C=0 O=2
      VL[1][1] = Lc[1][2] - t[2][0]; VR[1][1] = Rc[1][2] - t[2][0];   // D=1 C=2 O=0
      VL[2][0] = Lc[2][0] - t[0][1]; VR[2][0] = Rc[2][0] - t[0][1];   // D=2 C=0 O=1
      VL[2][1] = Lc[2][1] - t[1][0]; VR[2][1] = Rc[2][1] - t[1][0];   // D=2 C=1 O=0
      LS(lC[0]) = VL[1][0]; LS(lC[1]) = VL[1][1];
      __syncthreads();
      double L[3][2];
      L[0][0] = shfl_prev(VL[0][0]); L[0][1] = shfl_prev(VL[0][1]);
      L[1][0] = LM(lC[0]); L[1][1] = LM(lC[1]);
      L[2][0] = LzC[0]; L[2][1] = LzC[1];
      LzC[0] = VL[2][0]; LzC[1] = VL[2][1];
      if (BC && k == kc) {
        // components: x-faces carry C = 1, 2; y-faces C = 0, 2; z-faces C = 0, 1 (own value for a hi face: the cell's u, re-read)
        if (wfc & 0xFFFFFFu) {
          PAIR_BC(L[0][0], VR[0][0], wfc >> 4, U_AT(kc, 0L, 1), U_AT(kc, -8L, 1)) PAIR_BC(L[0][1], VR[0][1], wfc >> 8, U_AT(kc, 0L, 2), U_AT(kc, -8L, 2))
          PAIR_BC(L[1][0], VR[1][0], wfc >> 12, U_AT(kc, 0L, 0), U_AT(kc, -F.u_row, 0)) PAIR_BC(L[1][1], VR[1][1], wfc >> 20, U_AT(kc, 0L, 2), U_AT(kc, -F.u_row, 2))
        }
        if ((zf & 0xFFFu) && ing_ij) {
          PAIR_BC(L[2][0], VR[2][0], zf, U_AT(kc, 0L, 0), U_AT(kc - 1, 0L, 0)) PAIR_BC(L[2][1], VR[2][1], zf >> 4, U_AT(kc, 0L, 1), U_AT(kc - 1, 0L, 1))
        }
      }
      // XC index n: 0 = (C0,D1) 1 = (C0,D2) 2 = (C1,D0) 3 = (C1,D2) 4 = (C2,D0) 5 = (C2,D1); the advecting velocity is UI[D][D] of the lower face
      xc1[2] = vp_up(ui1[0][0], L[0][0], VR[0][0], eps); xc1[4] = vp_up(ui1[0][0], L[0][1], VR[0][1], eps);
      xc1[0] = vp_up(ui1[1][1], L[1][0], VR[1][0], eps); xc1[5] = vp_up(ui1[1][1], L[1][1], VR[1][1], eps);
      xc1[1] = vp_up(ui1[2][2], L[2][0], VR[2][0], eps); xc1[3] = vp_up(ui1[2][2], L[2][1], VR[2][1], eps);
      LS(lXC[0]) = xc1[0]; LS(lXC[1]) = xc1[5];           // the y-face fields: read by the row below at the top of the next iteration
    } else {
      __syncthreads();                                              // keeps lB / lUI single-buffered while stage C is idle (start of a chunk)
    }
    // ---------------- stage D, plane kk-2 ----------------
    if (kk - 2 >= k0 - 1) {
      const int k = kk - 2, kc = min(max(k, KB), KT);
      unsigned zf, zp; ZF_WORD(k, kc, zf) ZP_WORD(kc, zp)
      const bool vz = k >= F.lo[2] && k <= F.hi[2];
      double Ld[3] = { LbD[0], LbD[1], LbD[2] }, Rd[3] = { RbD[0], RbD[1], RbD[2] };
      unsigned wfd = 0;
      if (BC) {
        unsigned px = wpx, py = wpy; asm volatile("" : "+v"(px)); asm volatile("" : "+v"(py));
        wfd = wf; asm volatile("" : "+v"(wfd));
        if (px | py) {
          PREMOD(Ld[0], Rd[0], px, px >> 12, U_AT(kc, -8L, 0), U_AT(kc, 8L, 0))
          PREMOD(Ld[1], Rd[1], py >> 4, py >> 16, U_AT(kc, -F.u_row, 1), U_AT(kc, F.u_row, 1))
        }
        if (zp) { PREMOD(Ld[2], Rd[2], zp >> 8, zp >> 20, U_AT(kc - 1, 0L, 2), U_AT(kc + 1, 0L, 2)) }
      }
      // XC on the upper faces: n0, n5 (y-faces) from the next row, n2, n4 (x-faces) from the next lane, n1, n3 (z-faces) from stage C above
      const double x1[6] = { xcy_up[0], xc1[1], lane_next(xc2[2]), xc1[3], lane_next(xc2[4]), xcy_up[1] };
      const double (&x0)[6] = xc2;
      const double (&g)[3] = sm2;
      double VL[3], VR[3];
      {   // D = 0: T1 = 1 with XC(0,1) = n0,  T2 = 2 with XC(0,2) = n1
        const double a1 = F.tD[1] * g[1] * (x1[0] - x0[0]);
        const double a2 = F.tD[2] * g[2] * (x1[1] - x0[1]);
        double vl = Ld[0] - a1 - a2, vr = Rd[0] - a1 - a2;
        if (!F.use_minion) { vl = vl + fb[0]; vr = vr + fb[0]; }
        VL[0] = vl; VR[0] = vr;
      }
      {   // D = 1: T1 = 0 with XC(1,0) = n2,  T2 = 2 with XC(1,2) = n3
        const double a1 = F.tD[0] * g[0] * (x1[2] - x0[2]);
        const double a2 = F.tD[2] * g[2] * (x1[3] - x0[3]);
        double vl = Ld[1] - a1 - a2, vr = Rd[1] - a1 - a2;
        if (!F.use_minion) { vl = vl + fb[1]; vr = vr + fb[1]; }
        VL[1] = vl; VR[1] = vr;
      }
      {   // D = 2: T1 = 0 with XC(2,0) = n4,  T2 = 1 with XC(2,1) = n5
        const double a1 = F.tD[0] * g[0] * (x1[4] - x0[4]);
        const double a2 = F.tD[1] * g[1] * (x1[5] - x0[5]);
        double vl = Ld[2] - a1 - a2, vr = Rd[2] - a1 - a2;
        if (!F.use_minion) { vl = vl + fb[2]; vr = vr + fb[2]; }
        VL[2] = vl; VR[2] = vr;
      }
      LS(lD) = VL[1];
      __syncthreads();
      const double Lx = shfl_prev(VL[0]), Ly = LM(lD), Lzc = LzD;
      LzD = VL[2];
      if (k >= k0) {
        if (own_ij) {
          double e0 = vpf_riemann(Lx, VR[0], eps), e1 = vpf_riemann(Ly, VR[1], eps), e2 = vpf_riemann(Lzc, VR[2], eps);
          if (BC) {
            if (wfd >> 24) { OUT_BC(e0, wfd >> 24, Lx, VR[0], U_AT(kc, 0L, 0), U_AT(kc, -8L, 0)) OUT_BC(e1, wfd >> 28, Ly, VR[1], U_AT(kc, 0L, 1), U_AT(kc, -F.u_row, 1)) }
            if (zf >> 24) { OUT_BC(e2, zf >> 24, Lzc, VR[2], U_AT(kc, 0L, 2), U_AT(kc - 1, 0L, 2)) }
          }
          if (vy && vz) std_(qu, 0u, e0);
          if (vx && vz) std_(qv, 0u, e1);
          if (vx && vy) std_(qw, 0u, e2);
        }
        qu += F.sq[0]; qv += F.sq[1]; qw += F.sq[2];
      }
    } else {
      __syncthreads();
    }
    // hand the planes on
    #pragma unroll
    for (int d = 0; d < 3; d++) { LbD[d] = Lb1[d][d]; RbD[d] = Rb1[d][d]; sm2[d] = sm1[d]; fb[d] = fa[d]; fa[d] = ft[d]; }
    #pragma unroll
    for (int n = 0; n < 6; n++) xc2[n] = xc1[n];
    #pragma unroll
    for (int d = 0; d < 3; d++) for (int c = 0; c < 3; c++) { Lb1[d][c] = Lb[d][c]; Rb1[d][c] = Rb[d][c]; ui1[d][c] = ui0[d][c]; }
  }
  #undef U_AT
  #undef BC_V
  #undef PAIR_BC
  #undef OUT_BC
  #undef PREMOD
  #undef ZF_WORD
  #undef ZP_WORD
  #undef V_LOAD
  #undef V_ADVANCE
}
// (the interior / boundary workgroup dispatch of kk_mk_F_m was measured here too: 1.198 -> 1.227 ms, not kept)
template <bool BC = true, bool INL = true, bool PW2 = false> __global__ void __launch_bounds__(64 * TNY) kk_vp_F_m(VArgs F, Range3 r, int klen, const double *umax) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  vp_F_m_body<BC, INL, PW2>(F, r, klen, umax, bx_, by_, bz_);
}
// the remainder tile column in narrow segments inside the same launch (see kk_mk_F_mc)
template <bool BC, bool INL, bool PW2> __global__ void __launch_bounds__(64 * TNY) kk_vp_F_mc(VArgs F, Range3 r, int klen, const double *umax, int full, int segw) {
  int bx_, by_, bz_; xcd_remap(bx_, by_, bz_);
  Range3 rr = r;
  if (bx_ == full) {
    rr.lo[0] = r.lo[0] + full * FNX;
    if (by_ * (TNY * (64 / segw) - 2) > r.hi[1] - r.lo[1]) return;
    vp_F_m_body<BC, INL, PW2, true>(F, rr, klen, umax, 0, by_, bz_, segw);
  } else {
    rr.hi[0] = r.lo[0] + full * FNX - 1;
    vp_F_m_body<BC, INL, PW2>(F, rr, klen, umax, bx_, by_, bz_);
  }
}
static bool vfused_args(VArgs &F, const GArgs &A, const FV &u, const FV sl[3], const FV &force, const FV &um, const FV &vm, const FV &wm) {
  if (!same_geom(sl[0], sl[1]) || !same_geom(sl[0], sl[2]) || sl[0].sc != sl[1].sc || sl[0].sc != sl[2].sc) return false;
  const int KB = A.lo[2] - 1;
  const FV *in[5] = { &u, &sl[0], &sl[1], &sl[2], &force };
  for (int f = 0; f < 5; f++) if (KB < in[f]->a2 || A.hi[2] + 1 >= in[f]->a2 + in[f]->n2) return false;      // the clamped planes must exist
  F.pu = (const char *)(u.p + (long)u.n0 * u.n1 * (KB - u.a2)); F.sc_u = 8L * u.sc; F.sp_u = 8L * u.n0 * u.n1; F.g_u = FGeo{ u.a0, u.a1, u.n0 };
  for (int d = 0; d < 3; d++) F.psl[d] = (const char *)(sl[d].p + (long)sl[d].n0 * sl[d].n1 * (KB - sl[d].a2));
  F.sc_sl = 8L * sl[0].sc; F.sp_sl = 8L * sl[0].n0 * sl[0].n1; F.g_sl = FGeo{ sl[0].a0, sl[0].a1, sl[0].n0 };
  F.pf = (const char *)(force.p + (long)force.n0 * force.n1 * (KB - force.a2)); F.sc_f = 8L * force.sc; F.sp_f = 8L * force.n0 * force.n1; F.g_f = FGeo{ force.a0, force.a1, force.n0 };
  const FV *out[3] = { &um, &vm, &wm };
  for (int f = 0; f < 3; f++) {
    const FV &v = *out[f];
    F.q[f] = (char *)(v.p + (long)v.n0 * v.n1 * (KB - v.a2)); F.sq[f] = 8L * v.n0 * v.n1; F.h[f] = FGeo{ v.a0, v.a1, v.n0 };
  }
  F.u_row = 8L * u.n0;
  const double dt2 = 0.5 * A.dt, dt4 = A.dt / 4.0, dt6 = A.dt / 6.0;
  F.dt2 = dt2;
  for (int d = 0; d < 3; d++) {
    F.dx[d] = A.dx[d]; F.idx[d] = 1.0 / A.dx[d]; F.tC[d] = dt6 / A.dx[d]; F.tD[d] = dt4 / A.dx[d];
    F.lo[d] = A.lo[d]; F.hi[d] = A.hi[d]; F.phys[d][0] = A.phys[d][0]; F.phys[d][1] = A.phys[d][1];
  }
  F.use_minion = A.use_minion; F.outlet2d = A.outlet2d;
  F.p2 = (is_pow2(A.dx[0]) && is_pow2(A.dx[1]) && is_pow2(A.dx[2]) && !no_p2()) ? 1 : 0;
  return true;
}

struct VBatchD { VArgs F; Range3 r; int klen; const double *umax; int g[3], sw; };      // sw: lanes of a row segment (NAR launches; seg_geo)
template <bool BC, bool INL, bool PW2 = false, bool NAR = false> __global__ void __launch_bounds__(64 * TNY) kk_vp_F_mb(const VBatchD *descs, const int *start, int nbox) {
  BATCH_LOCATE(VBatchD, g)
  vp_F_m_body<BC, INL, PW2, NAR>(q.F, q.r, q.klen, q.umax, BX, BY, BZ, q.sw);
}

// ====================================================================================================
// dm = 2 (BASELINE.json configs[0], the reference's CPU-runnable case): velpred_2d (velpred.f90:125-524) and mkflux_2d
// (mkflux.f90:152-691).  Two stages only -- the transverse terms use the stage-B states directly.  One thread per cell
// of the grown box, plane k = 0; fabs of a 2-D run are one z-plane on the device (z-ghost planes exist but are unused).
// ====================================================================================================
// velpred_2d pair on the lower D-face (D = 0, 1) of cell (i,j): NB every CFL factor divides INSIDE max()/min()
// (velpred.f90:255-261, 332-338), unlike velpred_3d
template <int D> DEVI void vp2_pair(const GArgs &A, const FV &u, const FV &slp, const FV &force, int i, int j, double L[2], double R[2]) {
  const double dt2 = 0.5 * A.dt;
  const double ul = ld<D>(u, i, j, 0, -1, D), ur = fv_get(u, i, j, 0, D);
  const double cl = dt2 * fmax(0.0, ul / A.dx[D]), cr = dt2 * fmin(0.0, ur / A.dx[D]);
  #pragma unroll
  for (int c = 0; c < 2; c++) {
    L[c] = ld<D>(u, i, j, 0, -1, c) + (0.5 - cl) * ld<D>(slp, i, j, 0, -1, c);
    R[c] = fv_get(u, i, j, 0, c) - (0.5 + cr) * fv_get(slp, i, j, 0, c);
    if (A.use_minion) { L[c] = L[c] + dt2 * ld<D>(force, i, j, 0, -1, c); R[c] = R[c] + dt2 * fv_get(force, i, j, 0, c); }
  }
  const int side = face_side<D>(A, i, j, 0);
  if (side >= 0) {
    #pragma unroll
    for (int c = 0; c < 2; c++) bc_pair(L[c], R[c], A.phys[D][side], side, true, c == D, (side == 0) ? ld<D>(u, i, j, 0, -1, c) : fv_get(u, i, j, 0, c), false);
  }
}
DEVI double riemann0(double L, double R, double eps) {
  const double uavg = 0.5 * (L + R);
  const bool test = ((L <= 0.0 && R >= 0.0) || (fabs(L + R) < eps));
  double un = (uavg > 0.0) ? L : R;
  return test ? 0.0 : un;
}
// UI: [0] uimhx(.,1) (normal), [1] uimhx(.,2), [2] uimhy(.,1), [3] uimhy(.,2) (normal)
__global__ void __launch_bounds__(256) kk_vp2_B(FV u, FV sl0, FV sl1, FV force, FV UI, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  double L[2], R[2];
  if (i >= A.lo[0]) {
    vp2_pair<0>(A, u, sl0, force, i, j, L, R);
    const double un = riemann0(L[0], R[0], eps);
    fv_at(UI, i, j, 0, 0) = un;
    const double v = (un > 0.0) ? L[1] : R[1], av = 0.5 * (L[1] + R[1]);
    fv_at(UI, i, j, 0, 1) = (fabs(un) < eps) ? av : v;
  }
  if (j >= A.lo[1]) {
    vp2_pair<1>(A, u, sl1, force, i, j, L, R);
    const double un = riemann0(L[1], R[1], eps);
    fv_at(UI, i, j, 0, 3) = un;
    const double v = (un > 0.0) ? L[0] : R[0], av = 0.5 * (L[0] + R[0]);
    fv_at(UI, i, j, 0, 2) = (fabs(un) < eps) ? av : v;
  }
}
// the MAC velocities (velpred.f90:402-443 vmac, 455-496 umac)
template <int D> DEVI void vp2_D_one(const GArgs &A, const FV &u, const FV &slp, const FV &force, const FV &UI, const FV &umac, int i, int j, double eps) {
  constexpr int T = 1 - D;
  if (coord<T>(i, j, 0) > A.hi[T]) return;
  double Lv[2], Rv[2];
  vp2_pair<D>(A, u, slp, force, i, j, Lv, Rv);
  const double dt2 = 0.5 * A.dt, dt4 = A.dt / 4.0;
  // transverse data live on T-faces: normal comp UI[T*2+T'] ... index: x-faces 0 (normal), 1;  y-faces 2, 3 (normal)
  constexpr int nrm = (T == 0) ? 0 : 3, trn = (T == 0) ? 1 : 2;      // for D = 0 (T = 1): advect with uimhy(.,2) = 3, difference uimhy(.,1) = 2
  double LR[2] = { Lv[D], Rv[D] };
  #pragma unroll
  for (int sd = 0; sd < 2; sd++) {
    const int ci = i - ((sd == 0) && D == 0), cj = j - ((sd == 0) && D == 1);
    double v = LR[sd] - (dt4 / A.dx[T]) * (ld<T>(UI, ci, cj, 0, 1, nrm) + fv_get(UI, ci, cj, 0, nrm)) * (ld<T>(UI, ci, cj, 0, 1, trn) - fv_get(UI, ci, cj, 0, trn));
    if (!A.use_minion) v = v + dt2 * fv_get(force, ci, cj, 0, D);
    LR[sd] = v;
  }
  double v = riemann0(LR[0], LR[1], eps);
  const int side = face_side<D>(A, i, j, 0);
  if (side >= 0) {
    const int ph = A.phys[D][side];
    if (ph == VDN_SLIP_WALL || ph == VDN_NO_SLIP_WALL) v = 0.0;
    else if (ph == VDN_INLET) v = (side == 0) ? ld<D>(u, i, j, 0, -1, D) : fv_get(u, i, j, 0, D);
    else if (ph == VDN_OUTLET) v = (side == 0) ? fmin(LR[1], 0.0) : fmax(LR[0], 0.0);
  }
  fv_at(umac, i, j, 0) = v;
}
__global__ void __launch_bounds__(256) kk_vp2_D(FV u, FV sl0, FV sl1, FV force, FV UI, FV um, FV vm, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  vp2_D_one<0>(A, u, sl0, force, UI, um, i, j, eps);
  vp2_D_one<1>(A, u, sl1, force, UI, vm, i, j, eps);
}
__global__ void kk_velmax2(FV u, Range3 r, double *out) {             // velpred.f90:216-222
  REDUCE_IJ(r)
  double m = 0.0;
  if (in_ij) m = fmax(fabs(fv_get(u, i, j, 0, 0)), fabs(fv_get(u, i, j, 0, 1)));
  block_atomic_max(out, m);
}
__global__ void kk_macmax2(FV um, FV vm, GArgs A, Range3 r, double *out) {   // mkflux.f90:248-259
  REDUCE_IJ(r)
  double m = 0.0;
  if (in_ij) {
    if (j <= A.hi[1]) m = fmax(m, fabs(fv_get(um, i, j, 0)));
    if (i <= A.hi[0]) m = fmax(m, fabs(fv_get(vm, i, j, 0)));
  }
  block_atomic_max(out, m);
}
// SI: [0*nc + c] simhx, [1*nc + c] simhy
__global__ void __launch_bounds__(256) kk_mk2_B(FV s, FV sl0, FV sl1, FV um, FV vm, FV force, FV macrhs, FV SI, GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  for (int c = 0; c < A.ncomp; c++) {
    double L, R;
    if (i >= A.lo[0]) { mk_pair<0>(A, s, sl0, um, force, macrhs, c, i, j, 0, L, R); fv_at(SI, i, j, 0, 0 * A.ncomp + c) = upwind_mac(L, R, fv_get(um, i, j, 0), eps); }
    if (j >= A.lo[1]) { mk_pair<1>(A, s, sl1, vm, force, macrhs, c, i, j, 0, L, R); fv_at(SI, i, j, 0, 1 * A.ncomp + c) = upwind_mac(L, R, fv_get(vm, i, j, 0), eps); }
  }
}
// edge states and fluxes (mkflux.f90:472-558 sedgey, 570-660 sedgex)
template <int D> DEVI void mk2_D_one(const GArgs &A, const FV &s, const FV &slp, const FV &macD, const FV &macT, const FV &force, const FV &macrhs,
                                     const FV &SI, const FV &sedge, const FV &flux, int c, int i, int j, double eps) {
  constexpr int T = 1 - D;
  if (coord<T>(i, j, 0) > A.hi[T]) return;
  double L, R;
  mk_pair<D>(A, s, slp, macD, force, macrhs, c, i, j, 0, L, R);
  const double dt2 = 0.5 * A.dt, dt4 = A.dt / 4.0;
  const bool cons = A.cons[c] != 0;
  double LR[2] = { L, R };
  #pragma unroll
  for (int sd = 0; sd < 2; sd++) {
    const int ci = i - ((sd == 0) && D == 0), cj = j - ((sd == 0) && D == 1);
    const double s0 = fv_get(s, ci, cj, 0, c);
    double v = LR[sd] - trans_term<T>(A, SI, T * A.ncomp + c, macT, cons, ci, cj, 0, dt2, dt4);
    if (cons) v = v + (dt2 / A.dx[T]) * s0 * (ld<T>(macT, ci, cj, 0, 1) - fv_get(macT, ci, cj, 0));
    if (!A.use_minion) {
      v = v + dt2 * fv_get(force, ci, cj, 0, c);
      if (cons) v = v - dt2 * s0 * fv_get(macrhs, ci, cj, 0);
    }
    LR[sd] = v;
  }
  const double um = fv_get(macD, i, j, 0);
  double e = upwind_mac(LR[0], LR[1], um, eps);
  const int side = face_side<D>(A, i, j, 0);
  if (side >= 0) {
    const int ph = A.phys[D][side];
    const double in = (side == 0) ? LR[1] : LR[0];
    const bool vel = A.is_vel != 0;
    if (ph == VDN_INLET) e = (side == 0) ? ld<D>(s, i, j, 0, -1, c) : fv_get(s, i, j, 0, c);
    else if (ph == VDN_SLIP_WALL) e = (vel && c == D) ? 0.0 : in;
    else if (ph == VDN_NO_SLIP_WALL) e = vel ? 0.0 : in;
    else if (ph == VDN_OUTLET) e = (vel && c == D) ? ((side == 0) ? fmin(in, 0.0) : fmax(in, 0.0)) : in;
  }
  fv_at(sedge, i, j, 0, c) = e;
  if (cons) fv_at(flux, i, j, 0, c) = e * um;
}
__global__ void __launch_bounds__(256) kk_mk2_D(FV s, FV sl0, FV sl1, FV um, FV vm, FV force, FV macrhs, FV SI, FV sex, FV sey, FV flx, FV fly,
                                                GArgs A, Range3 r, const double *umax) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double eps = eps_from(umax);
  for (int c = 0; c < A.ncomp; c++) {
    mk2_D_one<0>(A, s, sl0, um, vm, force, macrhs, SI, sex, flx, c, i, j, eps);
    mk2_D_one<1>(A, s, sl1, vm, um, force, macrhs, SI, sey, fly, c, i, j, eps);
  }
}
static FV work_fv2(const BoxP &b) {        // one plane over [lo-1, hi+1]^2
  FV f; f.p = nullptr; f.a0 = b.lo[0] - 1; f.a1 = b.lo[1] - 1; f.a2 = 0;
  f.n0 = b.hi[0] - b.lo[0] + 3; f.n1 = b.hi[1] - b.lo[1] + 3; f.n2 = 1;
  f.sc = (long)f.n0 * f.n1;
  return f;
}
static void ranges2(const GArgs &A, Range3 &rv, Range3 &rg, Range3 &rf) {
  for (int d = 0; d < 2; d++) { rv.lo[d] = A.lo[d]; rv.hi[d] = A.hi[d]; rg.lo[d] = A.lo[d] - 1; rg.hi[d] = A.hi[d] + 1; rf.lo[d] = A.lo[d]; rf.hi[d] = A.hi[d] + 1; }
  rv.lo[2] = rv.hi[2] = rg.lo[2] = rg.hi[2] = rf.lo[2] = rf.hi[2] = 0;
}
static void k2_velpred(const vdn_multifab *u, vdn_multifab **umac, const vdn_multifab *force, const double *dx, double dt, const vdn_bc_tower *bct) {
  REQUIRE(u->nc == 2 && u->ng >= 3 && force->ng >= 1 && umac[0]->ng >= 1, "velpred (dm = 2): operand shapes");
  hipStream_t st = ctx().stream;
  for (int ib = 0; ib < u->nfabs(); ib++) {
    size_t mark = arena_mark();
    GArgs A; fill_gargs(A, u, ib, bct, 0, 2, dx, dt);
    A.is_vel = 1;
    BoxP bp = make_boxp(u, ib, bct);
    FV w = work_fv2(bp);
    const size_t fld = (size_t)w.sc * sizeof(double);
    FV sl[2] = { w, w }, UI = w;
    sl[0].p = (double *)arena_alloc(fld * 2); sl[1].p = (double *)arena_alloc(fld * 2); UI.p = (double *)arena_alloc(fld * 4);
    double *umax = (double *)arena_alloc(256);
    HIPCHK(hipMemsetAsync(umax, 0, sizeof(double), st));
    Range3 rv, rg, rf; ranges2(A, rv, rg, rf);
    hipLaunchKernelGGL(kk_velmax2, reduce_grid(rv), dim3(64, 4, 1), 0, st, u->fabs[ib], rv, umax);
    hipLaunchKernelGGL(kk_slopes, grid_for(rg), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[1], A, rg, 3);
    hipLaunchKernelGGL(kk_vp2_B, grid_for(rg), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], force->fabs[ib], UI, A, rg, umax);
    hipLaunchKernelGGL(kk_vp2_D, grid_for(rf), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], force->fabs[ib], UI, umac[0]->fabs[ib], umac[1]->fabs[ib], A, rf, umax);
    arena_release(mark);
  }
}
static void k2_mkflux(const vdn_multifab *s, vdn_multifab **sedge, vdn_multifab **flux, vdn_multifab **umac, const vdn_multifab *force,
                      const vdn_multifab *mac_rhs, const double *dx, double dt, const vdn_bc_tower *bct, bool is_vel, const int *is_cons) {
  const int ncomp = s->nc;
  REQUIRE(ncomp <= 3 && s->ng >= 3 && umac[0]->ng >= 1 && force->ng >= 1 && mac_rhs->ng >= 1, "mkflux (dm = 2): operand shapes");
  const int bccomp = is_vel ? 0 : bct->dm;
  hipStream_t st = ctx().stream;
  for (int ib = 0; ib < s->nfabs(); ib++) {
    size_t mark = arena_mark();
    GArgs A; fill_gargs(A, s, ib, bct, bccomp, ncomp, dx, dt);
    A.is_vel = is_vel ? 1 : 0;
    for (int c = 0; c < ncomp; c++) A.cons[c] = is_cons[c] ? 1 : 0;
    BoxP bp = make_boxp(s, ib, bct);
    FV w = work_fv2(bp);
    const size_t fld = (size_t)w.sc * sizeof(double);
    FV sl[2] = { w, w }, SI = w;
    sl[0].p = (double *)arena_alloc(fld * ncomp); sl[1].p = (double *)arena_alloc(fld * ncomp); SI.p = (double *)arena_alloc(fld * 2 * ncomp);
    double *umax = (double *)arena_alloc(256);
    HIPCHK(hipMemsetAsync(umax, 0, sizeof(double), st));
    Range3 rv, rg, rf; ranges2(A, rv, rg, rf);
    const FV &um = umac[0]->fabs[ib], &vm = umac[1]->fabs[ib];
    hipLaunchKernelGGL(kk_macmax2, reduce_grid(rf), dim3(64, 4, 1), 0, st, um, vm, A, rf, umax);
    hipLaunchKernelGGL(kk_slopes, grid_for(rg), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], sl[1], A, rg, 3);
    hipLaunchKernelGGL(kk_mk2_B, grid_for(rg), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], um, vm, force->fabs[ib], mac_rhs->fabs[ib], SI, A, rg, umax);
    hipLaunchKernelGGL(kk_mk2_D, grid_for(rf), dim3(64, 4, 1), 0, st, s->fabs[ib], sl[0], sl[1], um, vm, force->fabs[ib], mac_rhs->fabs[ib], SI,
                       sedge[0]->fabs[ib], sedge[1]->fabs[ib], flux[0]->fabs[ib], flux[1]->fabs[ib], A, rf, umax);
    arena_release(mark);
  }
}

__global__ void __launch_bounds__(64 * TNY) kk_vp_B_mb(const VpD *descs, const int *start, int nbox) {
  BATCH_LOCATE(VpD, gg)
  vp_B_m_body<true, true>(q.s, q.sl0, q.sl1, q.sl2, q.force, q.UI, q.A, q.rg, q.klg, q.umax, BX, BY, BZ);
}
__global__ void __launch_bounds__(64 * TNY) kk_vp_C_mb(const VpD *descs, const int *start, int nbox) {
  BATCH_LOCATE(VpD, gg)
  vp_C_m_body<true, true>(q.s, q.sl0, q.sl1, q.sl2, q.force, q.UI, q.XC, q.A, q.rg, q.klg, q.umax, BX, BY, BZ);
}
__global__ void __launch_bounds__(64 * TNY) kk_vp_D_mb(const VpD *descs, const int *start, int nbox) {
  BATCH_LOCATE(VpD, gf)
  vp_D_m_body<true, true>(q.s, q.sl0, q.sl1, q.sl2, q.force, q.UI, q.XC, q.um, q.vm, q.wm, q.A, q.rf, q.klf, q.umax, BX, BY, BZ);
}
void k_velpred(const vdn_multifab *u, vdn_multifab **umac, const vdn_multifab *force, const double *dx, double dt,
               const vdn_bc_tower *bct) {
  Prof prof_("velpred");
  god_xcd_init();
  if (ctx().prm.dm == 2) { k2_velpred(u, umac, force, dx, dt, bct); return; }
  REQUIRE(u->nc == 3 && u->ng >= 3 && force->ng >= 1 && umac[0]->ng >= 1, "velpred: operand shapes");
  hipStream_t st = ctx().stream;
  if (use_batched(u) && !plain_godunov()) {
    size_t mark = arena_mark();
    const int nb = u->nfabs();
    GodBatch<VpD> B; B.d.resize(nb);
    double *umax = (double *)arena_alloc(sizeof(double) * nb);
    HIPCHK(hipMemsetAsync(umax, 0, sizeof(double) * nb, st));
    for (int ib = 0; ib < nb; ib++) {
      VpD &q = B.d[ib];
      fill_gargs(q.A, u, ib, bct, 0, 3, dx, dt);
      q.A.is_vel = 1;
      BoxP bp = make_boxp(u, ib, bct);
      FV w = work_fv(nullptr, bp, 0);
      const size_t fld = (size_t)w.sc * sizeof(double);
      q.sl0 = q.sl1 = q.sl2 = q.UI = q.XC = w;
      q.sl0.p = (double *)arena_alloc(fld * 3); q.sl1.p = (double *)arena_alloc(fld * 3); q.sl2.p = (double *)arena_alloc(fld * 3);
      q.UI.p = (double *)arena_alloc(fld * 9); q.XC.p = (double *)arena_alloc(fld * 6);
      for (int d = 0; d < 3; d++) { q.rm.lo[d] = q.A.lo[d]; q.rm.hi[d] = q.A.hi[d]; q.rg.lo[d] = q.A.lo[d] - 1; q.rg.hi[d] = q.A.hi[d] + 1; q.rf.lo[d] = q.A.lo[d]; q.rf.hi[d] = q.A.hi[d] + 1; }
      q.s = u->fabs[ib]; q.force = force->fabs[ib]; q.um = umac[0]->fabs[ib]; q.vm = umac[1]->fabs[ib]; q.wm = umac[2]->fabs[ib];
      q.umax = umax + ib;
    }
    B.finish();
    const dim3 blk(64, TNY, 1);
    hipLaunchKernelGGL(kk_velmax_b, dim3(B.tot[1]), dim3(64, 4, 1), 0, st, B.dev, B.st[1], nb);
    hipLaunchKernelGGL(kk_slopes_b<VpD>, dim3(B.tot[0]), dim3(64, 4, 1), 0, st, B.dev, B.st[0], nb, 7);
    static const bool fused_env = !(vdn_env("VDN_GOD_FUSED") && atoi(vdn_env("VDN_GOD_FUSED")) == 0);
    if (fused_env) {               // stages B + C + D in one march per box (vp_F_m_body)
      std::vector<VBatchD> fd(nb);
      std::vector<int> fstart(nb);
      bool ok = true, inflow = false, any = false;
      int tot = 0;
      for (int ib = 0; ib < nb && ok; ib++) {
        const VpD &q = B.d[ib];
        const FV sl[3] = { q.sl0, q.sl1, q.sl2 };
        for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) { inflow = inflow || q.A.phys[d][sd] == VDN_INLET; any = any || bc_mode_host(q.A.phys[d][sd]); }
        ok = vfused_args(fd[ib].F, q.A, q.s, sl, q.force, q.um, q.vm, q.wm);
        fd[ib].r = q.rf; fd[ib].umax = q.umax;
        fused_grid_small(fd[ib].r, fd[ib].klen, fd[ib].g, fd[ib].sw);
      }
      if (ok) {
        // two launches: the boxes in full-width tiles, the narrow ones in row segments (seg_geo)
        std::stable_partition(fd.begin(), fd.end(), [](const VBatchD &f) { return f.sw == 0; });
        const int nwide = (int)std::count_if(fd.begin(), fd.end(), [](const VBatchD &f) { return f.sw == 0; });
        int totn = 0;
        for (int t = 0; t < nb; t++) { int &T = t < nwide ? tot : totn; fstart[t] = T; T += fd[t].g[0] * fd[t].g[1] * fd[t].g[2]; }
        VBatchD *dd = (VBatchD *)desc_scratch(sizeof(VBatchD) * nb); int *ds = (int *)desc_scratch(sizeof(int) * nb);
        upload_staged(dd, fd.data(), sizeof(VBatchD) * nb); upload_staged(ds, fstart.data(), sizeof(int) * nb);
        const bool p2 = fd[0].F.p2 != 0;
        #define VP_LAUNCH(NARROW, T, D0, S0, N)                                                                                                          \
          if (!any) { if (p2) hipLaunchKernelGGL((kk_vp_F_mb<false, false, true, NARROW>), dim3(T), blk, 0, st, D0, S0, N); else hipLaunchKernelGGL((kk_vp_F_mb<false, false, false, NARROW>), dim3(T), blk, 0, st, D0, S0, N); } \
          else if (inflow) hipLaunchKernelGGL((kk_vp_F_mb<true, true, false, NARROW>), dim3(T), blk, 0, st, D0, S0, N);                                      \
          else if (p2) hipLaunchKernelGGL((kk_vp_F_mb<true, false, true, NARROW>), dim3(T), blk, 0, st, D0, S0, N);                                          \
          else hipLaunchKernelGGL((kk_vp_F_mb<true, false, false, NARROW>), dim3(T), blk, 0, st, D0, S0, N);
        if (nwide) { VP_LAUNCH(false, tot, dd, ds, nwide) }
        if (nwide < nb) { VP_LAUNCH(true, totn, dd + nwide, ds + nwide, nb - nwide) }
        #undef VP_LAUNCH
        arena_release(mark);
        return;
      }
    }
    hipLaunchKernelGGL(kk_vp_B_mb, dim3(B.tot[2]), blk, 0, st, B.dev, B.st[2], nb);
    hipLaunchKernelGGL(kk_vp_C_mb, dim3(B.tot[2]), blk, 0, st, B.dev, B.st[2], nb);
    hipLaunchKernelGGL(kk_vp_D_mb, dim3(B.tot[3]), blk, 0, st, B.dev, B.st[3], nb);
    arena_release(mark);
    return;
  }
  for (int ib = 0; ib < u->nfabs(); ib++) {
    size_t mark = arena_mark();
    GArgs A; fill_gargs(A, u, ib, bct, 0, 3, dx, dt);
    A.is_vel = 1;
    BoxP bp = make_boxp(u, ib, bct);
    FV w = work_fv(nullptr, bp, 0);
    const size_t fld = (size_t)w.sc * sizeof(double);
    FV sl[3], UI = w, XC = w;
    const bool keep = (int)ctx().slope_src.size() == u->nfabs();           // advance_timestep: mkflux(uold) reuses these slopes
    for (int d = 0; d < 3; d++) { sl[d] = w; sl[d].p = keep ? ctx().slope_cache[d][ib] : (double *)arena_alloc(fld * 3); }
    if (keep) ctx().slope_src[ib] = u->fabs[ib].p;
    UI.p = (double *)arena_alloc(fld * 9);
    XC.p = (double *)arena_alloc(fld * 6);
    double *umax = (double *)arena_alloc(256);
    HIPCHK(hipMemsetAsync(umax, 0, sizeof(double), st));
    Range3 rv, rg, rf;
    for (int d = 0; d < 3; d++) { rv.lo[d] = A.lo[d]; rv.hi[d] = A.hi[d]; rg.lo[d] = A.lo[d] - 1; rg.hi[d] = A.hi[d] + 1; rf.lo[d] = A.lo[d]; rf.hi[d] = A.hi[d] + 1; }
    launch_slopes(u->fabs[ib], sl, A, rg, 3, umax, st);      // + max |u| (kk_velmax)
    (void)rv;
    if (plain_godunov()) {
      hipLaunchKernelGGL(kk_vp_B, grid_for(rg), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, A, rg, umax);
      hipLaunchKernelGGL(kk_vp_C, grid_for(rg), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC, A, rg, umax);
      hipLaunchKernelGGL(kk_vp_D, grid_for(rf), dim3(64, 4, 1), 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC,
                         umac[0]->fabs[ib], umac[1]->fabs[ib], umac[2]->fabs[ib], A, rf, umax);
    } else {
      int klg, klf;
      const dim3 gg = march_grid(rg, klg), gf = march_grid(rf, klf), blk(64, TNY, 1);
      static const bool fused_env = !(vdn_env("VDN_GOD_FUSED") && atoi(vdn_env("VDN_GOD_FUSED")) == 0);
      VArgs VA;
      if (fused_env && vfused_args(VA, A, u->fabs[ib], sl, force->fabs[ib], umac[0]->fabs[ib], umac[1]->fabs[ib], umac[2]->fabs[ib])) {
        int klF;                   // stages B + C + D in one march, boundary rules inside (see vp_F_m_body)
        const dim3 gF = fused_grid(rf, klF);
        bool any = false, inflow = false;
        for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) { any = any || bc_mode_host(A.phys[d][sd]); inflow = inflow || A.phys[d][sd] == VDN_INLET; }
        const bool p2 = VA.p2 != 0;
        if (!any) { if (p2) hipLaunchKernelGGL((kk_vp_F_m<false, false, true>), gF, blk, 0, st, VA, rf, klF, umax); else hipLaunchKernelGGL((kk_vp_F_m<false, false>), gF, blk, 0, st, VA, rf, klF, umax); }
        else if (inflow) hipLaunchKernelGGL((kk_vp_F_m<true, true>), gF, blk, 0, st, VA, rf, klF, umax);
        else if (p2) {
          dim3 gM; int klM, full, segw;
          if (fused_grid_cols(rf, gM, klM, full, segw)) hipLaunchKernelGGL((kk_vp_F_mc<true, false, true>), gM, blk, 0, st, VA, rf, klM, umax, full, segw);
          else hipLaunchKernelGGL((kk_vp_F_m<true, false, true>), gF, blk, 0, st, VA, rf, klF, umax);
        }
        else hipLaunchKernelGGL((kk_vp_F_m<true, false>), gF, blk, 0, st, VA, rf, klF, umax);
      } else if (slab_bc()) {      // interior marches, then the face-centred code on the boundary slabs (see kk_slabs)
        const VpPlain P{ u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC, umac[0]->fabs[ib], umac[1]->fabs[ib], umac[2]->fabs[ib], A, umax };
        const Slabs Sg = boundary_slabs(A, rg), Sf = boundary_slabs(A, rf);
        hipLaunchKernelGGL(kk_vp_B_m<false>, gg, blk, 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, A, rg, klg, umax);
        launch_slabs(Sg, VpBFix{ P }, st);
        hipLaunchKernelGGL(kk_vp_C_m<false>, gg, blk, 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC, A, rg, klg, umax);
        launch_slabs(Sg, VpCFix{ P }, st);
        hipLaunchKernelGGL(kk_vp_D_m<false>, gf, blk, 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC,
                           umac[0]->fabs[ib], umac[1]->fabs[ib], umac[2]->fabs[ib], A, rf, klf, umax);
        launch_slabs(Sf, VpDFix{ P }, st);
      } else {
      hipLaunchKernelGGL(kk_vp_B_m<true>, gg, blk, 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, A, rg, klg, umax);
      hipLaunchKernelGGL(kk_vp_C_m<true>, gg, blk, 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC, A, rg, klg, umax);
      hipLaunchKernelGGL(kk_vp_D_m<true>, gf, blk, 0, st, u->fabs[ib], sl[0], sl[1], sl[2], force->fabs[ib], UI, XC,
                         umac[0]->fabs[ib], umac[1]->fabs[ib], umac[2]->fabs[ib], A, rf, klf, umax);
      }
    }
    arena_release(mark);
  }
}
