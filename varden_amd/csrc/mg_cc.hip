// mg_cc.hip -- MAC projection (reference src/macproject.f90:20-133) and the cell-centred multigrid
// that replaces FBoxLib's ml_cc_solve (called from reference src/mac_multigrid.f90:53-62).
//
// System:  -sum_d [b_d(i+e_d)(phi(i+e_d)-phi(i)) - b_d(i)(phi(i)-phi(i-e_d))]/h_d^2 = rh
// with Neumann domain faces folded into b (b := 0), Dirichlet faces into b := 2b with a zero ghost,
// periodic faces through a ghost image.  Algorithm: V(nu1,nu2), red-black Gauss-Seidel, 8-cell
// average restriction, piecewise-constant prolongation, coarse b = mean of the 4 fine faces; the
// same algorithm and expression order as oracle/vo_macproject.c.
//
// HBM layout of one level (all six fields phi, rh, res, bx, by, bz share it): index (i,j,k) in
// [-1..n] maps to  (i + 16) + PX*((j+1) + PY*(k+1)),  PX a multiple of 16 doubles, so cell i = 0 of
// every row starts a 128-byte line and a wave's row segment is made of whole lines.
// The smoother is the kernel the north-star roofline is quoted on: ALGORITHMIC traffic per colour
// pass = 48 B/cell (phi r+w 16, rh 8, bx/by/bz 24), see DESIGN.md.
#include "vdn_dev.h"
#include <chrono>
#include <tuple>
#include <algorithm>

struct CLev {
  int n[3]; int PX, PY; long sz;
  double hi2[3];
  double *phi, *rh, *res, *b[3];
  double *alpha;        // cell coefficient of (alpha - div b grad); nullptr when alpha = 0 (MAC projection)
  const double *rho;    // finest level of the MAC solve: density with one ghost layer, the face coefficients 2/(rho_i + rho_i-1) are
  int fold[3][2];       //   recomputed from it (8 B/cell instead of 24); fold = bc type of the box faces that are domain faces
  double cmu;           // > 0 (round 6, the viscous / diffusive solves, viscsolve.f90:57-60: beta = mu on every face, alpha = rho or 1): the finest level's face
                        //   coefficients are this constant, folded at the domain faces like the stored ones -- the level by colour reads phi, rhs and alpha only
};
DEVI long cidx(const CLev &L, int i, int j, int k) { return (long)(i + 16) + (long)L.PX * ((long)(j + 1) + (long)L.PY * (long)(k + 1)); }

// the face coefficient of the MAC projection between two cells (macproject.f90:376-394) with the boundary folding of kk_cc_load:
// the same expression and the same bits as the stored array
DEVI double beta_of(double ra, double rb, bool at_face, int e) {
  double v = 2.0 / (ra + rb);
  if (at_face) { if (e == VDN_BC_NEU) v = 0.0; else if (e == VDN_BC_DIR) v = 2.0 * v; }
  return v;
}
// A phi and diag in the fixed expression order shared with the oracle (cc_apply in vo_macproject.c).  RHO: the six face coefficients
// come from the density (six divisions per cell instead of 24 B/cell of HBM traffic per pass; the kernels are HBM-bound)
template <bool RHO = false> DEVI void cc_apply(const CLev &L, long c, double &Ap, double &diag, int i = 0, int j = 0, int k = 0) {
  const long sy = L.PX, sz = (long)L.PX * L.PY;
  const double p0 = L.phi[c];
  const double pxm = L.phi[c - 1], pxp = L.phi[c + 1];
  double bxm, bxp, bym, byp, bzm, bzp;
  if (RHO) {
    const double r0 = L.rho[c];
    bxm = beta_of(r0, L.rho[c - 1], i == 0, L.fold[0][0]);  bxp = beta_of(L.rho[c + 1], r0, i == L.n[0] - 1, L.fold[0][1]);
    bym = beta_of(r0, L.rho[c - sy], j == 0, L.fold[1][0]); byp = beta_of(L.rho[c + sy], r0, j == L.n[1] - 1, L.fold[1][1]);
    bzm = beta_of(r0, L.rho[c - sz], k == 0, L.fold[2][0]); bzp = beta_of(L.rho[c + sz], r0, k == L.n[2] - 1, L.fold[2][1]);
  } else {
    bxm = L.b[0][c]; bxp = L.b[0][c + 1];
    bym = L.b[1][c]; byp = L.b[1][c + sy];
    bzm = L.b[2][c]; bzp = L.b[2][c + sz];
  }
  const double ax = (bxp * (p0 - pxp) + bxm * (p0 - pxm)) * L.hi2[0];
  const double ay = (byp * (p0 - L.phi[c + sy]) + bym * (p0 - L.phi[c - sy])) * L.hi2[1];
  const double az = (bzp * (p0 - L.phi[c + sz]) + bzm * (p0 - L.phi[c - sz])) * L.hi2[2];
  Ap = ax + ay + az;
  diag = (bxp + bxm) * L.hi2[0] + (byp + bym) * L.hi2[1] + (bzp + bzm) * L.hi2[2];
  if (L.alpha) {                        // viscous / diffusive solves (+8 B/cell of traffic)
    const double a0 = L.alpha[c];
    Ap = Ap + a0 * p0;
    diag = diag + a0;
  }
}

// one colour pass of red-black Gauss-Seidel: thread t of a row updates cell i = 2t + ((j+k+color)&1)
// (measured at 256^3 and rejected: several k-planes per workgroup 0.161 ms; the x-triplet as one 16-byte load per lane plus lane
// exchange 0.230 ms; all rows staged through LDS as aligned 16-byte pairs, even and odd cells in separate LDS rows so that every global
// load is a full line and every LDS read unit-stride, 0.157 ms -- against 0.141-0.144 ms for this form)
// Overlap of the halo exchange with the pass ("interior kernel, halos on a second stream, boundary kernel"): the cells of a box
// split into the SHELL (the outermost layer: the only cells that read ghost values) and the INTERIOR.  interior_only = 1: a pass skips
// the shell cells; they are updated afterwards, once the halo has landed, by kk_cc_gsrb_shell.  Cells of one colour do not read each
// other, so the order inside a pass is free and the bits are those of the unsplit pass.
// `hm`: the faces of the box whose ghost cells come from the exchange (bit 2d + side; physical non-periodic faces read none)
DEVI bool cc_is_shell(const CLev &L, int i, int j, int k, int hm) {
  return ((hm & 1) && i == 0) || ((hm & 2) && i == L.n[0] - 1) || ((hm & 4) && j == 0) || ((hm & 8) && j == L.n[1] - 1) ||
         ((hm & 16) && k == 0) || ((hm & 32) && k == L.n[2] - 1);
}
template <bool RHO> DEVI void cc_update_cell(const CLev &L, int i, int j, int k) {
  const long c = cidx(L, i, j, k);
  double Ap, diag; cc_apply<RHO>(L, c, Ap, diag, i, j, k);
  if (diag != 0.0) L.phi[c] = L.phi[c] + (L.rh[c] - Ap) / diag;
}
template <bool RHO> DEVI void cc_gsrb_cell(const CLev &L, int color, int interior_only) {
  int bx, by, bz; xcd_block(bx, by, bz);
  const int j = by * blockDim.y + threadIdx.y;
  const int k = bz;
  const int i = 2 * (int)(bx * blockDim.x + threadIdx.x) + ((j + k + color) & 1);
  if (i >= L.n[0] || j >= L.n[1]) return;
  if (interior_only && cc_is_shell(L, i, j, k, interior_only)) return;
  cc_update_cell<RHO>(L, i, j, k);
}
// the shell cells of one colour: blockIdx.z = face (x-lo, x-hi, y-lo, y-hi, z-lo, z-hi); the x faces own their edges and corners, the y
// faces the remaining edges, so that every shell cell is updated exactly once
template <bool RHO> __global__ void __launch_bounds__(256) kk_cc_gsrb_shell(CLev L, int color, int hm) {
  int f = 0;
  for (int z = blockIdx.z;; f++) if ((hm >> f) & 1) { if (z == 0) break; z--; }      // blockIdx.z-th face of the mask
  const int d = f >> 1, side = f & 1;
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
  int q[3];
  const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
  q[d] = side ? L.n[d] - 1 : 0; q[da] = a; q[db] = b;
  if (side && L.n[d] == 1 && ((hm >> (2 * d)) & 1)) return;          // a one-cell-thick box: its single layer belongs to the lo face
  if (q[da] >= L.n[da] || q[db] >= L.n[db]) return;
  if (d >= 1 && (((hm & 1) && q[0] == 0) || ((hm & 2) && q[0] == L.n[0] - 1))) return;           // owned by an x face
  if (d == 2 && (((hm & 4) && q[1] == 0) || ((hm & 8) && q[1] == L.n[1] - 1))) return;           // owned by a y face
  if ((q[0] + q[1] + q[2] + color) & 1) return;
  cc_update_cell<RHO>(L, q[0], q[1], q[2]);
}
// cc_apply<true> on values already in registers (the paired colour pass): same expressions, same order
DEVI void cc_apply_rho_vals(const CLev &L, int i, int j, int k, const double p[7], const double r[7], double &Ap, double &diag) {
  // p, r: centre, x-, x+, y-, y+, z-, z+
  const double p0 = p[0], r0 = r[0];
  const double bxm = beta_of(r0, r[1], i == 0, L.fold[0][0]), bxp = beta_of(r[2], r0, i == L.n[0] - 1, L.fold[0][1]);
  const double bym = beta_of(r0, r[3], j == 0, L.fold[1][0]), byp = beta_of(r[4], r0, j == L.n[1] - 1, L.fold[1][1]);
  const double bzm = beta_of(r0, r[5], k == 0, L.fold[2][0]), bzp = beta_of(r[6], r0, k == L.n[2] - 1, L.fold[2][1]);
  const double ax = (bxp * (p0 - p[2]) + bxm * (p0 - p[1])) * L.hi2[0];
  const double ay = (byp * (p0 - p[4]) + bym * (p0 - p[3])) * L.hi2[1];
  const double az = (bzp * (p0 - p[6]) + bzm * (p0 - p[5])) * L.hi2[2];
  Ap = ax + ay + az;
  diag = (bxp + bxm) * L.hi2[0] + (byp + bym) * L.hi2[1] + (bzp + bzm) * L.hi2[2];
}
// ... and the operator of the viscous / diffusive solves on values in registers: the face coefficients are the constant L.cmu, folded at the domain faces exactly as
// kk_cc_load folds the stored ones (Neumann 0, Dirichlet 2 mu), the alpha term as cc_apply / cc_apply_vals add it -- the same expressions in the same order, the same bits
DEVI double beta_const(double mu, bool at_face, int e) {
  double v = mu;
  if (at_face) { if (e == VDN_BC_NEU) v = 0.0; else if (e == VDN_BC_DIR) v = 2.0 * v; }
  return v;
}
DEVI void cc_apply_cmu_vals(const CLev &L, int i, int j, int k, const double p[7], double a0, double &Ap, double &diag) {
  const double p0 = p[0];
  const double bxm = beta_const(L.cmu, i == 0, L.fold[0][0]), bxp = beta_const(L.cmu, i == L.n[0] - 1, L.fold[0][1]);
  const double bym = beta_const(L.cmu, j == 0, L.fold[1][0]), byp = beta_const(L.cmu, j == L.n[1] - 1, L.fold[1][1]);
  const double bzm = beta_const(L.cmu, k == 0, L.fold[2][0]), bzp = beta_const(L.cmu, k == L.n[2] - 1, L.fold[2][1]);
  const double ax = (bxp * (p0 - p[2]) + bxm * (p0 - p[1])) * L.hi2[0];
  const double ay = (byp * (p0 - p[4]) + bym * (p0 - p[3])) * L.hi2[1];
  const double az = (bzp * (p0 - p[6]) + bzm * (p0 - p[5])) * L.hi2[2];
  Ap = ax + ay + az;
  diag = (bxp + bxm) * L.hi2[0] + (byp + bym) * L.hi2[1] + (bzp + bzm) * L.hi2[2];
  Ap = Ap + a0 * p0;
  diag = diag + a0;
}
// The colour pass is bound by the texture addresser (TA busy 255 k of ~310 k cycles at 256^3, profiles/r01_smoother_rho_pmc.json):
// 16 memory instructions per updated cell, each an 8-byte access with stride 2 over the lanes.  Paired form: a thread owns the 2 x 2
// block (columns 2t, 2t+1; rows j, j+1) of a k-plane, which holds exactly two cells of the colour (a diagonal).  All loads are aligned
// 16-byte pairs, unit-stride over the lanes: the two own rows give both centres, their in-pair x-neighbours and each other's y-neighbour;
// rows j-1, j+2 and the planes k-1, k+1 give the rest; the x-neighbour outside the pair comes from the adjacent lane (DPP), from memory
// at the two ends of the wave (one branch-free load per field).  8 + 1 loads per field for two cells instead of 2 x 7.  Same arithmetic.
struct Pair7 { double a[7], b[7]; };
DEVI double sel2(const double2 &q, int hi) { return hi ? q.y : q.x; }
// (round 3, measured and rejected: rho and rhs loaded with the non-temporal hint, so that phi -- 134 MB, read and written by every pass --
// might stay in the 256 MB Infinity Cache: 0.1178 -> 0.1170 ms at 256^3, 0.0154 -> 0.0182 ms at 128^3, where everything was cached before)
DEVI void pair_gather(const double *v, const CLev &L, long cpA, int par, int lane, Pair7 &o) {
  const long sy = L.PX, sz = (long)L.PX * L.PY;
  // the cells just outside the pair along x: previous lane's odd cell / next lane's even cell; the two ends of the wave read memory
  // (issued first, two active lanes).  par = 0: A (row j, even column) looks left and B (row j+1, odd column) looks right; par = 1 the
  // other way round.
  double e = 0.0;
  if (lane == 0 || lane == 63) e = v[cpA + ((lane == 0) ? (par == 0 ? -1 : sy - 1) : (par == 0 ? sy + 2 : 2))];
  #define LD2(off) (*reinterpret_cast<const double2 *>(v + cpA + (off)))
  const double2 PA = LD2(0), PB = LD2(sy), PAm = LD2(-sy), PBp = LD2(2 * sy);
  const double2 ZAm = LD2(-sz), ZAp = LD2(sz), ZBm = LD2(sy - sz), ZBp = LD2(sy + sz);
  #undef LD2
  const double prevv = par == 0 ? lane_prev(PA.y) : lane_prev(PB.y);
  const double nextv = par == 0 ? lane_next(PB.x) : lane_next(PA.x);
  const double outl = lane == 0 ? e : prevv, outr = lane == 63 ? e : nextv;
  o.a[0] = sel2(PA, par);      o.b[0] = sel2(PB, 1 - par);
  o.a[3] = sel2(PAm, par);     o.a[4] = sel2(PB, par);
  o.b[3] = sel2(PA, 1 - par);  o.b[4] = sel2(PBp, 1 - par);
  o.a[5] = sel2(ZAm, par);     o.a[6] = sel2(ZAp, par);
  o.b[5] = sel2(ZBm, 1 - par); o.b[6] = sel2(ZBp, 1 - par);
  if (par == 0) { o.a[1] = outl; o.a[2] = PA.y; o.b[1] = PB.x; o.b[2] = outr; }
  else          { o.a[1] = PA.x; o.a[2] = outr; o.b[1] = outl; o.b[2] = PB.y; }
}
// ADD (the sweep that follows a prolongation, one box without periodic faces): the piecewise-constant correction of the coarse level C is
// added on the fly instead of by a kk_cc_prolong pass over the level (phi r+w 268 MB, 59 us at 256^3).  ADD = 1, the first colour: every
// value the pass reads becomes phi + e(parent) -- a 2 x 2 block shares its parent, the six blocks around it give the rest (coarse ghost
// cells are zero, so box-boundary ghosts stay as they are); the updated cells are stored corrected.  ADD = 2, the second colour: its
// neighbours are final, only the cell itself still lacks its correction.  The sum phi + e is the one kk_cc_prolong forms.
template <int ADD> __global__ void __launch_bounds__(256) kk_cc_gsrb_rho_pair_t(CLev L, int color, int interior_only, CLev C, int kdown) {
  int bx, by, bz; xcd_block(bx, by, bz);
  const int lane = threadIdx.x, k = kdown ? L.n[2] - 1 - bz : bz;
  const int t = bx * 64 + lane, jA = 2 * (by * 4 + (int)threadIdx.y);
  const bool act = 2 * t + 1 < L.n[0] && jA + 1 < L.n[1];
  const int par = (jA + k + color) & 1;                              // uniform over the wave
  const long cpA = cidx(L, 2 * min(t, L.n[0] / 2), min(jA, L.n[1] - 2), k);      // clamped: every lane takes part in the lane exchange
  Pair7 P, R;
  pair_gather(L.phi, L, cpA, par, lane, P);
  pair_gather(L.rho, L, cpA, par, lane, R);
  const double2 RA = *reinterpret_cast<const double2 *>(L.rh + cpA), RB = *reinterpret_cast<const double2 *>(L.rh + cpA + L.PX);
  if (!act) return;
  if (ADD) {
    const int J = jA >> 1, K = k >> 1;
    const long cc = cidx(C, t, J, K), csy = C.PX, csz = (long)C.PX * C.PY;
    const double e0 = C.phi[cc];
    P.a[0] = P.a[0] + e0; P.b[0] = P.b[0] + e0;
    if (ADD == 1) {
      const double exm = C.phi[cc - 1], exp_ = C.phi[cc + 1], eym = C.phi[cc - csy], eyp = C.phi[cc + csy], ez = C.phi[(k & 1) ? cc + csz : cc - csz];
      const double ezm = (k & 1) ? e0 : ez, ezp = (k & 1) ? ez : e0;
      if (par == 0) { P.a[1] = P.a[1] + exm; P.a[2] = P.a[2] + e0; P.b[1] = P.b[1] + e0; P.b[2] = P.b[2] + exp_; }
      else          { P.a[1] = P.a[1] + e0; P.a[2] = P.a[2] + exp_; P.b[1] = P.b[1] + exm; P.b[2] = P.b[2] + e0; }
      P.a[3] = P.a[3] + eym; P.a[4] = P.a[4] + e0; P.b[3] = P.b[3] + e0; P.b[4] = P.b[4] + eyp;
      P.a[5] = P.a[5] + ezm; P.a[6] = P.a[6] + ezp; P.b[5] = P.b[5] + ezm; P.b[6] = P.b[6] + ezp;
    }
  }
  const int iA = 2 * t + par, iB = 2 * t + 1 - par;
  double Ap, diag;
  cc_apply_rho_vals(L, iA, jA, k, P.a, R.a, Ap, diag);
  if (diag != 0.0 && !(interior_only && cc_is_shell(L, iA, jA, k, interior_only))) L.phi[cpA + par] = P.a[0] + (sel2(RA, par) - Ap) / diag;
  else if (ADD) L.phi[cpA + par] = P.a[0];
  cc_apply_rho_vals(L, iB, jA + 1, k, P.b, R.b, Ap, diag);
  if (diag != 0.0 && !(interior_only && cc_is_shell(L, iB, jA + 1, k, interior_only))) L.phi[cpA + L.PX + 1 - par] = P.b[0] + (sel2(RB, 1 - par) - Ap) / diag;
  else if (ADD) L.phi[cpA + L.PX + 1 - par] = P.b[0];
}
// kdown: the planes from the top (launch_gsrb: the second colour of a sweep -- it starts on what the first colour's pass touched last, still in the Infinity Cache)
__global__ void __launch_bounds__(256) kk_cc_gsrb_rho_pair(CLev L, int color, int interior_only, int kdown) {
  int bx, by, bz; xcd_block(bx, by, bz);
  const int lane = threadIdx.x, k = kdown ? L.n[2] - 1 - bz : bz;
  const int t = bx * 64 + lane, jA = 2 * (by * 4 + (int)threadIdx.y);
  const bool act = 2 * t + 1 < L.n[0] && jA + 1 < L.n[1];
  const int par = (jA + k + color) & 1;                              // uniform over the wave
  const long cpA = cidx(L, 2 * min(t, L.n[0] / 2), min(jA, L.n[1] - 2), k);      // clamped: every lane takes part in the lane exchange
  Pair7 P, R;
  pair_gather(L.phi, L, cpA, par, lane, P);
  pair_gather(L.rho, L, cpA, par, lane, R);
  const double2 RA = *reinterpret_cast<const double2 *>(L.rh + cpA), RB = *reinterpret_cast<const double2 *>(L.rh + cpA + L.PX);
  if (!act) return;
  const int iA = 2 * t + par, iB = 2 * t + 1 - par;
  double Ap, diag;
  cc_apply_rho_vals(L, iA, jA, k, P.a, R.a, Ap, diag);
  if (diag != 0.0 && !(interior_only && cc_is_shell(L, iA, jA, k, interior_only))) L.phi[cpA + par] = P.a[0] + (sel2(RA, par) - Ap) / diag;
  cc_apply_rho_vals(L, iB, jA + 1, k, P.b, R.b, Ap, diag);
  if (diag != 0.0 && !(interior_only && cc_is_shell(L, iB, jA + 1, k, interior_only))) L.phi[cpA + L.PX + 1 - par] = P.b[0] + (sel2(RB, 1 - par) - Ap) / diag;
}
// ---- round 5: the finest level of macproject's solve STORED BY COLOUR ---------------------------------------------------------------------------
// The paired pass above still moves whole lines of phi and rhs although half of every line belongs to the other colour: 572 MB per pass at 256^3
// (profiles/r05_smoother_rho_pmc.json) where the cells it touches hold 402 MB.  Split storage: cells with (i + j + k) & 1 == c live in arrays [c],
// entry ih of row (j, k) is cell i = 2 ih + ((j + k + c) & 1); rows, planes and the ghost layer as in the level array, 16 entries (one line) of padding in front
// of a row.  A colour pass then reads its own phi / rhs / rho and the OTHER colour's phi / rho -- nothing it does not use -- as aligned 16-byte
// pairs: the y and z neighbours of entry ih are entry ih of the other colour's rows j -+ 1 / planes k -+ 1, the x neighbours its entries ih - 1, ih
// (row parity 0) or ih, ih + 1 (parity 1).  Same expressions, same order, same bits as the interleaved pass (cc_apply_rho_vals);
// tools/probes/split_colour_probe.hip measured the form first (0.0865 against 0.1135 ms per pass at 256^3).  VDN_MAC_SPLIT=0 keeps the level interleaved.
struct CSplit { int PXH, off; long sy, sz, tot; double *phi[2], *rh[2], *rho[2]; };
DEVI long sidx(const CSplit &S, int ih, int j, int k) { return (long)(ih + S.off) + S.sy * (long)(j + 1) + S.sz * (long)(k + 1); }
// interleaved -> split, every entry of the padded rows (what lies outside cells -2 .. n+1 becomes zero): one aligned pair (cells 2 ih, 2 ih + 1) feeds both colours
__global__ void __launch_bounds__(256) kk_cc_to_split(CLev L, CSplit S, int what) {       // what: 1 phi, 2 rhs, 4 rho
  const int e = blockIdx.x * 64 + threadIdx.x, j = (int)(blockIdx.y * 4 + threadIdx.y) - 1, k = (int)blockIdx.z - 1;
  if (e >= S.PXH || j > L.n[1]) return;
  const int ih = e - S.off;
  const bool in = ih >= -1 && ih <= L.n[0] / 2;
  const long src = cidx(L, in ? 2 * ih : 0, j, k), dst = sidx(S, ih, j, k);
  const int c0 = (j + k) & 1;                    // the colour of the even cells of this row
  const double2 z = make_double2(0.0, 0.0);
  if (what & 1) { const double2 v = in ? *reinterpret_cast<const double2 *>(L.phi + src) : z; S.phi[c0][dst] = v.x; S.phi[1 - c0][dst] = v.y; }
  if (what & 2) { const double2 v = in ? *reinterpret_cast<const double2 *>(L.rh + src) : z;  S.rh[c0][dst] = v.x;  S.rh[1 - c0][dst] = v.y; }
  if (what & 4) { const double *rsrc = L.cmu > 0.0 ? L.alpha : L.rho; const double2 v = in ? *reinterpret_cast<const double2 *>(rsrc + src) : z; S.rho[c0][dst] = v.x; S.rho[1 - c0][dst] = v.y; }
}
// split -> interleaved: phi on the valid cells
__global__ void __launch_bounds__(256) kk_cc_from_split(CLev L, CSplit S) {
  const int ih = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (ih >= L.n[0] / 2 || j >= L.n[1]) return;
  const int c0 = (j + k) & 1;
  const long s = sidx(S, ih, j, k);
  *reinterpret_cast<double2 *>(L.phi + cidx(L, 2 * ih, j, k)) = make_double2(S.phi[c0][s], S.phi[1 - c0][s]);
}
// a thread owns entries ih, ih + 1 of a row (A, B); a wave a row segment of 128 entries = 256 cells; 8 rows per workgroup.  ADD as in kk_cc_gsrb_rho_pair_t:
// the parent of entry ih is coarse cell ih.
// planes k0 .. k0 + gridDim.z - 1 (kdown: from the top of that range)
// hm != 0 (levels with a halo, the exchange in flight on the halo stream): the cells of the one-cell shell behind the faces of `hm` keep their value; kk_cc_gsrb_split_shell
// updates them once the halo has landed (ADD passes run whole, after their exchange)
// VISC (round 6): the level of a viscous / diffusive solve -- constant face coefficients (L.cmu), alpha in the arrays that hold rho for the MAC solve (S.rho: the cell's own
// entry is all the pass reads of them): phi own r + w, rhs own, alpha own, phi other = 20 B per cell of the level where the stored-coefficient pass moves 56
template <int ADD, bool VISC = false> __global__ void __launch_bounds__(512) kk_cc_gsrb_rho_split(CLev L, CSplit S, int color, CLev C, int kdown, int k0, int hm) {
  const int lane = threadIdx.x, k = k0 + (kdown ? (int)gridDim.z - 1 - (int)blockIdx.z : (int)blockIdx.z);
  const int t = blockIdx.x * 64 + lane, nh = L.n[0] / 2;
  const int jr = blockIdx.y * 8 + threadIdx.y, j = min(jr, L.n[1] - 1);
  const bool act = 2 * t + 1 < nh && jr < L.n[1];
  const int p = (j + k + color) & 1;                                 // parity of i on this row (uniform over the wave): i = 2 ih + p
  const int ih = 2 * min(t, nh / 2);                                 // clamped: every lane takes part in the lane exchange, the pair (nh, nh + 1) lies inside the row
  const long c = sidx(S, ih, j, k);
  const double *po = S.phi[color], *px = S.phi[1 - color], *ro = S.rho[color], *rx = S.rho[1 - color];
  // the other colour's entry outside the wave's span: ih - 1 (p = 0, first lane) or ih + 2 (p = 1, last lane).  At the two ends of a row that is a ghost cell of the
  // box: behind a Neumann face phi is zero there and the coefficient is zero whatever rho holds (beta_of), so its line is not fetched; behind a Dirichlet face (rho enters
  // the coefficient), a neighbouring box or a periodic image (round 6: levels with a halo) it is
  double ep = 0.0, er = 0.0;
  if ((p == 0 && lane == 0 && (t > 0 || L.fold[0][0] != VDN_BC_NEU)) || (p == 1 && lane == 63 && (ih + 2 < nh || L.fold[0][1] != VDN_BC_NEU))) { const long o = c + (p ? 2 : -1); ep = px[o]; if (!VISC) er = rx[o]; }
  #define LDS2(v, off) (*reinterpret_cast<const double2 *>((v) + c + (off)))
  const double2 PO = LDS2(po, 0), RH = LDS2(S.rh[color], 0), RO = LDS2(ro, 0);
  const double2 PX = LDS2(px, 0), PYm = LDS2(px, -S.sy), PYp = LDS2(px, S.sy), PZm = LDS2(px, -S.sz), PZp = LDS2(px, S.sz);
  const double2 zz = make_double2(0.0, 0.0);
  const double2 RX = VISC ? zz : LDS2(rx, 0), RYm = VISC ? zz : LDS2(rx, -S.sy), RYp = VISC ? zz : LDS2(rx, S.sy), RZm = VISC ? zz : LDS2(rx, -S.sz), RZp = VISC ? zz : LDS2(rx, S.sz);
  #undef LDS2
  const double pl = lane_prev(PX.y), pr = lane_next(PX.x), rl = VISC ? 0.0 : lane_prev(RX.y), rr = VISC ? 0.0 : lane_next(RX.x);
  if (!act) return;
  // centre, x-, x+, y-, y+, z-, z+
  double pa[7] = { PO.x, p ? PX.x : (lane == 0 ? ep : pl), p ? PX.y : PX.x, PYm.x, PYp.x, PZm.x, PZp.x };
  double pb[7] = { PO.y, p ? PX.y : PX.x, p ? (lane == 63 ? ep : pr) : PX.y, PYm.y, PYp.y, PZm.y, PZp.y };
  const double ra[7] = { RO.x, p ? RX.x : (lane == 0 ? er : rl), p ? RX.y : RX.x, RYm.x, RYp.x, RZm.x, RZp.x };
  const double rb[7] = { RO.y, p ? RX.y : RX.x, p ? (lane == 63 ? er : rr) : RX.y, RYm.y, RYp.y, RZm.y, RZp.y };
  if (ADD) {
    const int J = j >> 1, K = k >> 1;
    const long cc = cidx(C, ih, J, K), csy = C.PX, csz = (long)C.PX * C.PY;
    const double eA = C.phi[cc], eB = C.phi[cc + 1];
    pa[0] = pa[0] + eA; pb[0] = pb[0] + eB;
    if (ADD == 1) {
      const double xo = C.phi[cc + (p ? 2 : -1)];
      const long oy = (j & 1) ? csy : -csy, oz = (k & 1) ? csz : -csz;
      const double yA = C.phi[cc + oy], yB = C.phi[cc + 1 + oy], zA = C.phi[cc + oz], zB = C.phi[cc + 1 + oz];
      if (p == 0) { pa[1] = pa[1] + xo; pa[2] = pa[2] + eA; pb[1] = pb[1] + eA; pb[2] = pb[2] + eB; }
      else        { pa[1] = pa[1] + eA; pa[2] = pa[2] + eB; pb[1] = pb[1] + eB; pb[2] = pb[2] + xo; }
      pa[3] = pa[3] + ((j & 1) ? eA : yA); pa[4] = pa[4] + ((j & 1) ? yA : eA); pb[3] = pb[3] + ((j & 1) ? eB : yB); pb[4] = pb[4] + ((j & 1) ? yB : eB);
      pa[5] = pa[5] + ((k & 1) ? eA : zA); pa[6] = pa[6] + ((k & 1) ? zA : eA); pb[5] = pb[5] + ((k & 1) ? eB : zB); pb[6] = pb[6] + ((k & 1) ? zB : eB);
    }
  }
  double Ap, diag;
  double2 out = make_double2(pa[0], pb[0]);
  if (VISC) cc_apply_cmu_vals(L, 2 * ih + p, j, k, pa, ra[0], Ap, diag); else cc_apply_rho_vals(L, 2 * ih + p, j, k, pa, ra, Ap, diag);
  if (diag != 0.0 && !(hm && cc_is_shell(L, 2 * ih + p, j, k, hm))) out.x = pa[0] + (RH.x - Ap) / diag;
  if (VISC) cc_apply_cmu_vals(L, 2 * ih + 2 + p, j, k, pb, rb[0], Ap, diag); else cc_apply_rho_vals(L, 2 * ih + 2 + p, j, k, pb, rb, Ap, diag);
  if (diag != 0.0 && !(hm && cc_is_shell(L, 2 * ih + 2 + p, j, k, hm))) out.y = pb[0] + (RH.y - Ap) / diag;
  *reinterpret_cast<double2 *>(S.phi[color] + c) = out;
}
// one value of a field stored by colour: cell (i, j, k), ghost layer included (i = -1: entry -1 of an odd row)
DEVI double split_get(double *const v[2], const CSplit &S, int i, int j, int k) {
  const int p = i & 1, c = (i + j + k) & 1;
  return v[c][sidx(S, (i - p) >> 1, j, k)];
}
// the shell cells of one colour on the split level (kk_cc_gsrb_shell's job and face ownership): cell by cell through split_get, cc_apply_rho_vals' expressions
__global__ void __launch_bounds__(256) kk_cc_gsrb_split_shell(CLev L, CSplit S, int color, int hm) {
  int f = 0;
  for (int z = blockIdx.z;; f++) if ((hm >> f) & 1) { if (z == 0) break; z--; }      // blockIdx.z-th face of the mask
  const int d = f >> 1, side = f & 1;
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
  int q[3];
  const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
  q[d] = side ? L.n[d] - 1 : 0; q[da] = a; q[db] = b;
  if (side && L.n[d] == 1 && ((hm >> (2 * d)) & 1)) return;
  if (q[da] >= L.n[da] || q[db] >= L.n[db]) return;
  if (d >= 1 && (((hm & 1) && q[0] == 0) || ((hm & 2) && q[0] == L.n[0] - 1))) return;           // owned by an x face
  if (d == 2 && (((hm & 4) && q[1] == 0) || ((hm & 8) && q[1] == L.n[1] - 1))) return;           // owned by a y face
  if ((q[0] + q[1] + q[2] + color) & 1) return;
  const int i = q[0], j = q[1], k = q[2];
  const double p[7] = { split_get(S.phi, S, i, j, k), split_get(S.phi, S, i - 1, j, k), split_get(S.phi, S, i + 1, j, k), split_get(S.phi, S, i, j - 1, k),
                        split_get(S.phi, S, i, j + 1, k), split_get(S.phi, S, i, j, k - 1), split_get(S.phi, S, i, j, k + 1) };
  const double r[7] = { split_get(S.rho, S, i, j, k), split_get(S.rho, S, i - 1, j, k), split_get(S.rho, S, i + 1, j, k), split_get(S.rho, S, i, j - 1, k),
                        split_get(S.rho, S, i, j + 1, k), split_get(S.rho, S, i, j, k - 1), split_get(S.rho, S, i, j, k + 1) };
  double Ap, diag;
  if (L.cmu > 0.0) cc_apply_cmu_vals(L, i, j, k, p, r[0], Ap, diag); else cc_apply_rho_vals(L, i, j, k, p, r, Ap, diag);
  if (diag != 0.0) S.phi[color][sidx(S, (i - (i & 1)) >> 1, j, k)] = p[0] + (split_get(S.rh, S, i, j, k) - Ap) / diag;
}
// residual + restriction on the split level (kk_cc_residual_rho_pair_rst's job): a thread owns entries ih, ih + 1 of BOTH colours in rows 2J, 2J + 1 of planes
// 2K, 2K + 1 = cells 2 ih .. 2 ih + 3 of each row = the children of coarse cells (ih, J, K) and (ih + 1, J, K).  Per plane: E = the row's even cells (2 ih, 2 ih + 2),
// O = its odd cells (2 ih + 1, 2 ih + 3), each one aligned pair of the colour that holds them ((parity + j + k) & 1); q[jj][m]: cell 2 ih + m of row 2J + jj.
DEVI void split_gather(double *const v[2], const CSplit &S, long c, int e, int lane, double q[2][4][7], bool ldl, bool ldr) {
  // e: the colour of the even cells of row 2J in this plane;  ldl / ldr: the entry left of the first / right of the last lane is read (a cell, or the ghost cell of a Dirichlet face)
  const long sy = S.sy, sz = S.sz;
  double a0 = 0.0, a1 = 0.0;                                       // outside the wave's span along x, two active lanes
  if (lane == 0 && ldl)  { a0 = v[1 - e][c - 1]; a1 = v[e][c + sy - 1]; }
  if (lane == 63 && ldr) { a0 = v[e][c + 2];     a1 = v[1 - e][c + sy + 2]; }
  #define LD(col, off) (*reinterpret_cast<const double2 *>(v[col] + c + (off)))
  const double2 E0 = LD(e, 0), O0 = LD(1 - e, 0), E1 = LD(1 - e, sy), O1 = LD(e, sy);
  const double2 Em = LD(1 - e, -sy), Om = LD(e, -sy), Ep = LD(e, 2 * sy), Op = LD(1 - e, 2 * sy);
  const double2 E0m = LD(1 - e, -sz), O0m = LD(e, -sz), E0p = LD(1 - e, sz), O0p = LD(e, sz);
  const double2 E1m = LD(e, sy - sz), O1m = LD(1 - e, sy - sz), E1p = LD(e, sy + sz), O1p = LD(1 - e, sy + sz);
  #undef LD
  const double l0 = lane_prev(O0.y), r0 = lane_next(E0.x), l1 = lane_prev(O1.y), r1 = lane_next(E1.x);
  // centre, x-, x+, y-, y+, z-, z+
  q[0][0][0] = E0.x; q[0][0][1] = lane == 0 ? a0 : l0; q[0][0][2] = O0.x; q[0][0][3] = Em.x; q[0][0][4] = E1.x; q[0][0][5] = E0m.x; q[0][0][6] = E0p.x;
  q[0][1][0] = O0.x; q[0][1][1] = E0.x; q[0][1][2] = E0.y; q[0][1][3] = Om.x; q[0][1][4] = O1.x; q[0][1][5] = O0m.x; q[0][1][6] = O0p.x;
  q[0][2][0] = E0.y; q[0][2][1] = O0.x; q[0][2][2] = O0.y; q[0][2][3] = Em.y; q[0][2][4] = E1.y; q[0][2][5] = E0m.y; q[0][2][6] = E0p.y;
  q[0][3][0] = O0.y; q[0][3][1] = E0.y; q[0][3][2] = lane == 63 ? a0 : r0; q[0][3][3] = Om.y; q[0][3][4] = O1.y; q[0][3][5] = O0m.y; q[0][3][6] = O0p.y;
  q[1][0][0] = E1.x; q[1][0][1] = lane == 0 ? a1 : l1; q[1][0][2] = O1.x; q[1][0][3] = E0.x; q[1][0][4] = Ep.x; q[1][0][5] = E1m.x; q[1][0][6] = E1p.x;
  q[1][1][0] = O1.x; q[1][1][1] = E1.x; q[1][1][2] = E1.y; q[1][1][3] = O0.x; q[1][1][4] = Op.x; q[1][1][5] = O1m.x; q[1][1][6] = O1p.x;
  q[1][2][0] = E1.y; q[1][2][1] = O1.x; q[1][2][2] = O1.y; q[1][2][3] = E0.y; q[1][2][4] = Ep.y; q[1][2][5] = E1m.y; q[1][2][6] = E1p.y;
  q[1][3][0] = O1.y; q[1][3][1] = E1.y; q[1][3][2] = lane == 63 ? a1 : r1; q[1][3][3] = O0.y; q[1][3][4] = Op.y; q[1][3][5] = O1m.y; q[1][3][6] = O1p.y;
}
template <bool VISC = false> __global__ void __launch_bounds__(256) kk_cc_residual_rho_split_rst(CLev L, CSplit S, double *nrm, CLev C, int Ka, int Kb) {       // coarse planes Ka .. Kb - 1
  const int lane = threadIdx.x, nh = L.n[0] / 2;
  const int u = blockIdx.x * 64 + lane, Jr = blockIdx.y * 4 + threadIdx.y, J = min(Jr, L.n[1] / 2 - 1);
  const bool act = 2 * u + 1 < nh && Jr < L.n[1] / 2;
  const int ih = 2 * min(u, nh / 2);
  double rmax = 0.0;
  for (int K = Ka + blockIdx.z; K < Kb; K += gridDim.z) {
    double s0 = 0.0, s1 = 0.0;
    #pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int k = 2 * K + kk, e = kk;                            // (0 + 2J + k) & 1
      const long c = sidx(S, ih, 2 * J, k);
      double P[2][4][7], R[2][4][7];
      const bool ldl = u > 0 || L.fold[0][0] != VDN_BC_NEU, ldr = ih + 2 < nh || L.fold[0][1] != VDN_BC_NEU;
      split_gather(S.phi, S, c, e, lane, P, ldl, ldr);
      if (VISC) {                                                  // alpha of the eight cells themselves: four aligned pairs
        const double2 A0e = *reinterpret_cast<const double2 *>(S.rho[e] + c), A0o = *reinterpret_cast<const double2 *>(S.rho[1 - e] + c);
        const double2 A1e = *reinterpret_cast<const double2 *>(S.rho[1 - e] + c + S.sy), A1o = *reinterpret_cast<const double2 *>(S.rho[e] + c + S.sy);
        R[0][0][0] = A0e.x; R[0][1][0] = A0o.x; R[0][2][0] = A0e.y; R[0][3][0] = A0o.y;
        R[1][0][0] = A1e.x; R[1][1][0] = A1o.x; R[1][2][0] = A1e.y; R[1][3][0] = A1o.y;
      } else
      split_gather(S.rho, S, c, e, lane, R, ldl, ldr);
      const double2 H0e = *reinterpret_cast<const double2 *>(S.rh[e] + c), H0o = *reinterpret_cast<const double2 *>(S.rh[1 - e] + c);
      const double2 H1e = *reinterpret_cast<const double2 *>(S.rh[1 - e] + c + S.sy), H1o = *reinterpret_cast<const double2 *>(S.rh[e] + c + S.sy);
      if (act) {
        const double rhs[2][4] = { { H0e.x, H0o.x, H0e.y, H0o.y }, { H1e.x, H1o.x, H1e.y, H1o.y } };
        #pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          #pragma unroll
          for (int m = 0; m < 4; m++) {
            double Ap, diag;
            if (VISC) cc_apply_cmu_vals(L, 2 * ih + m, 2 * J + jj, k, P[jj][m], R[jj][m][0], Ap, diag); else cc_apply_rho_vals(L, 2 * ih + m, 2 * J + jj, k, P[jj][m], R[jj][m], Ap, diag);
            const double r = rhs[jj][m] - Ap;
            rmax = nmax(rmax, fabs(r));
            if (m < 2) s0 = (kk == 0 && jj == 0 && m == 0) ? r : s0 + r;
            else       s1 = (kk == 0 && jj == 0 && m == 2) ? r : s1 + r;
          }
        }
      }
    }
    if (act) {
      const long cc = cidx(C, ih, J, K);
      *reinterpret_cast<double2 *>(C.rh + cc) = make_double2(s0 * 0.125, s1 * 0.125);
      *reinterpret_cast<double2 *>(C.phi + cc) = make_double2(0.0, 0.0);
    }
  }
  if (nrm) block_atomic_max(nrm, rmax);
}
// (measured and rejected: one coarse cell per thread -- unit-stride 8-byte loads, half the registers, twice the waves: MAC solve 10.02 -> 10.18 ms per step)
// the same pairing for the stored-coefficient pass (viscous / diffusive solves, the 128^3 level of the MAC solve, level 0 of the
// composite solves): the face coefficients of the two cells come from nine aligned pairs (bx: rows j, j+1; by: rows j, j+1, j+2;
// bz: planes k, k+1 of both rows), the x+ coefficient of the odd cell from the next lane.  cc_apply's expressions, same order.
DEVI void cc_apply_vals(const CLev &L, const double p[7], const double b[6], double a0, bool has_alpha, double &Ap, double &diag) {
  // p: centre, x-, x+, y-, y+, z-, z+;  b: bxm, bxp, bym, byp, bzm, bzp
  const double p0 = p[0];
  const double ax = (b[1] * (p0 - p[2]) + b[0] * (p0 - p[1])) * L.hi2[0];
  const double ay = (b[3] * (p0 - p[4]) + b[2] * (p0 - p[3])) * L.hi2[1];
  const double az = (b[5] * (p0 - p[6]) + b[4] * (p0 - p[5])) * L.hi2[2];
  Ap = ax + ay + az;
  diag = (b[1] + b[0]) * L.hi2[0] + (b[3] + b[2]) * L.hi2[1] + (b[5] + b[4]) * L.hi2[2];
  if (has_alpha) { Ap = Ap + a0 * p0; diag = diag + a0; }
}
__global__ void __launch_bounds__(256) kk_cc_gsrb_pair(CLev L, int color, int interior_only, int kdown) {
  int bx, by, bz; xcd_block(bx, by, bz);
  const int lane = threadIdx.x, k = kdown ? L.n[2] - 1 - bz : bz;
  const int t = bx * 64 + lane, jA = 2 * (by * 4 + (int)threadIdx.y);
  const bool act = 2 * t + 1 < L.n[0] && jA + 1 < L.n[1];
  const int par = (jA + k + color) & 1;                              // uniform over the wave
  const long cpA = cidx(L, 2 * min(t, L.n[0] / 2), min(jA, L.n[1] - 2), k);
  const long sy = L.PX, sz = (long)L.PX * L.PY;
  Pair7 P;
  pair_gather(L.phi, L, cpA, par, lane, P);
  // x+ face of the odd cell: bx of the next pair's even face (row j+1 when par = 0, row j when par = 1); the last lane reads memory
  const long rowo = par == 0 ? sy : 0;
  double e = 0.0;
  if (lane == 63) e = L.b[0][cpA + rowo + 2];
  #define LDB(d, off) (*reinterpret_cast<const double2 *>(L.b[d] + cpA + (off)))
  const double2 XA = LDB(0, 0), XB = LDB(0, sy);
  const double2 YA = LDB(1, 0), YB = LDB(1, sy), YC = LDB(1, 2 * sy);
  const double2 ZA0 = LDB(2, 0), ZA1 = LDB(2, sz), ZB0 = LDB(2, sy), ZB1 = LDB(2, sy + sz);
  #undef LDB
  const double2 RA = *reinterpret_cast<const double2 *>(L.rh + cpA), RB = *reinterpret_cast<const double2 *>(L.rh + cpA + sy);
  double2 AA = make_double2(0.0, 0.0), AB = AA;
  if (L.alpha) { AA = *reinterpret_cast<const double2 *>(L.alpha + cpA); AB = *reinterpret_cast<const double2 *>(L.alpha + cpA + sy); }
  const double nx = lane_next(par == 0 ? XB.x : XA.x);
  const double xnext = lane == 63 ? e : nx;
  if (!act) return;
  double bA[6], bB[6];
  if (par == 0) { bA[0] = XA.x; bA[1] = XA.y; bB[0] = XB.y; bB[1] = xnext; }      // A at column 2t, B at 2t+1
  else          { bA[0] = XA.y; bA[1] = xnext; bB[0] = XB.x; bB[1] = XB.y; }      // A at column 2t+1, B at 2t
  bA[2] = sel2(YA, par); bA[3] = sel2(YB, par); bB[2] = sel2(YB, 1 - par); bB[3] = sel2(YC, 1 - par);
  bA[4] = sel2(ZA0, par); bA[5] = sel2(ZA1, par); bB[4] = sel2(ZB0, 1 - par); bB[5] = sel2(ZB1, 1 - par);
  double Ap, diag;
  const int iA = 2 * t + par, iB = 2 * t + 1 - par;
  cc_apply_vals(L, P.a, bA, sel2(AA, par), L.alpha != nullptr, Ap, diag);
  if (diag != 0.0 && !(interior_only && cc_is_shell(L, iA, jA, k, interior_only))) L.phi[cpA + par] = P.a[0] + (sel2(RA, par) - Ap) / diag;
  cc_apply_vals(L, P.b, bB, sel2(AB, 1 - par), L.alpha != nullptr, Ap, diag);
  if (diag != 0.0 && !(interior_only && cc_is_shell(L, iB, jA + 1, k, interior_only))) L.phi[cpA + sy + 1 - par] = P.b[0] + (sel2(RB, 1 - par) - Ap) / diag;
}
__global__ void __launch_bounds__(256) kk_cc_gsrb(CLev L, int color, int interior_only) { cc_gsrb_cell<false>(L, color, interior_only); }
__global__ void __launch_bounds__(256) kk_cc_gsrb_rho(CLev L, int color, int interior_only) { cc_gsrb_cell<true>(L, color, interior_only); }
static inline void launch_gsrb_shell(const CLev &L, int color, hipStream_t st, int hm) {
  if (!hm) return;
  const int m = std::max(L.n[0], std::max(L.n[1], L.n[2]));
  const dim3 g((unsigned)((m + 63) / 64), (unsigned)((m + 3) / 4), (unsigned)__builtin_popcount(hm));
  if (L.rho) hipLaunchKernelGGL(kk_cc_gsrb_shell<true>, g, dim3(64, 4, 1), 0, st, L, color, hm);
  else hipLaunchKernelGGL(kk_cc_gsrb_shell<false>, g, dim3(64, 4, 1), 0, st, L, color, hm);
}
static bool mac_kflip() { static const bool b = !(vdn_env("VDN_MAC_KFLIP") && atoi(vdn_env("VDN_MAC_KFLIP")) == 0); return b; }
static inline void launch_gsrb(const CLev &L, int color, hipStream_t st, int interior_only = 0) {
  const dim3 blk(64, 4, 1), g((unsigned)(((L.n[0] + 1) / 2 + 63) / 64), (unsigned)((L.n[1] + 3) / 4), (unsigned)L.n[2]);
  static const bool paired = !(vdn_env("VDN_GSRB_PAIR") && atoi(vdn_env("VDN_GSRB_PAIR")) == 0);
  const int kdown = (mac_kflip() && (color & 1)) ? 1 : 0;
  if (L.rho && paired && L.n[0] % 2 == 0 && L.n[1] % 2 == 0 && L.n[0] >= 128)
    hipLaunchKernelGGL(kk_cc_gsrb_rho_pair, dim3((unsigned)((L.n[0] / 2 + 63) / 64), (unsigned)((L.n[1] / 2 + 3) / 4), (unsigned)L.n[2]), blk, 0, st, L, color, interior_only, kdown);
  else if (L.rho) hipLaunchKernelGGL(kk_cc_gsrb_rho, g, blk, 0, st, L, color, interior_only);
  else if (paired && L.n[0] % 2 == 0 && L.n[1] % 2 == 0 && L.n[0] >= 128)
    hipLaunchKernelGGL(kk_cc_gsrb_pair, dim3((unsigned)((L.n[0] / 2 + 63) / 64), (unsigned)((L.n[1] / 2 + 3) / 4), (unsigned)L.n[2]), blk, 0, st, L, color, interior_only, kdown);
  else hipLaunchKernelGGL(kk_cc_gsrb, g, blk, 0, st, L, color, interior_only);
}

// (Rounds 1-2 measured two fused red+black sweeps -- an LDS plane ring and register / DPP column pairs: 0.30 and 0.33 ms per sweep at 256^3 against
// 2 x 0.116 ms for two colour passes; both were bit-identical and slower, and were removed in round 4.  DESIGN.md section 9.)
template <bool RHO> DEVI void cc_residual_body(const CLev &L, double *nrm) {
  int bx, by, bz; xcd_block(bx, by, bz);
  const int i = bx * blockDim.x + threadIdx.x;
  const int j = by * blockDim.y + threadIdx.y;
  double rmax = 0.0;
  if (i < L.n[0] && j < L.n[1])
    for (int k = bz; k < L.n[2]; k += gridDim.z) {
      const long c = cidx(L, i, j, k);
      double Ap, diag; cc_apply<RHO>(L, c, Ap, diag, i, j, k);
      const double r = L.rh[c] - Ap;
      L.res[c] = r;
      rmax = nmax(rmax, fabs(r));
    }
  if (nrm) block_atomic_max(nrm, rmax);
}
// residual of the finest MAC level with the access pattern of kk_cc_gsrb_rho_pair: a thread owns all four cells of a 2 x 2 block,
// eight aligned 16-byte loads per field serve four cells (5 memory instructions per cell instead of 16)
DEVI void quad_gather(const double *v, const CLev &L, long cpA, int lane, double q[4][7]) {
  const long sy = L.PX, sz = (long)L.PX * L.PY;
  double eA = 0.0, eB = 0.0;                                       // the cells outside the wave's span along x, two active lanes
  if (lane == 0 || lane == 63) { const long o = lane == 0 ? -1 : 2; eA = v[cpA + o]; eB = v[cpA + sy + o]; }
  #define LD2(off) (*reinterpret_cast<const double2 *>(v + cpA + (off)))
  const double2 PA = LD2(0), PB = LD2(sy), PAm = LD2(-sy), PBp = LD2(2 * sy);
  const double2 ZAm = LD2(-sz), ZAp = LD2(sz), ZBm = LD2(sy - sz), ZBp = LD2(sy + sz);
  #undef LD2
  const double pA = lane_prev(PA.y), nA = lane_next(PA.x), pB = lane_prev(PB.y), nB = lane_next(PB.x);
  // order: centre, x-, x+, y-, y+, z-, z+;   cells: (2t, j), (2t+1, j), (2t, j+1), (2t+1, j+1)
  q[0][0] = PA.x; q[0][1] = lane == 0 ? eA : pA; q[0][2] = PA.y; q[0][3] = PAm.x; q[0][4] = PB.x;  q[0][5] = ZAm.x; q[0][6] = ZAp.x;
  q[1][0] = PA.y; q[1][1] = PA.x; q[1][2] = lane == 63 ? eA : nA; q[1][3] = PAm.y; q[1][4] = PB.y;  q[1][5] = ZAm.y; q[1][6] = ZAp.y;
  q[2][0] = PB.x; q[2][1] = lane == 0 ? eB : pB; q[2][2] = PB.y; q[2][3] = PA.x;  q[2][4] = PBp.x; q[2][5] = ZBm.x; q[2][6] = ZBp.x;
  q[3][0] = PB.y; q[3][1] = PB.x; q[3][2] = lane == 63 ? eB : nB; q[3][3] = PA.y;  q[3][4] = PBp.y; q[3][5] = ZBm.y; q[3][6] = ZBp.y;
}
__global__ void __launch_bounds__(256) kk_cc_residual_rho_pair(CLev L, double *nrm) {
  int bx, by, bz; xcd_block(bx, by, bz);
  const int lane = threadIdx.x;
  const int t = bx * 64 + lane, jA = 2 * (by * 4 + (int)threadIdx.y);
  const bool act = 2 * t + 1 < L.n[0] && jA + 1 < L.n[1];
  double rmax = 0.0;
  for (int k = bz; k < L.n[2]; k += gridDim.z) {
    const long cpA = cidx(L, 2 * min(t, L.n[0] / 2), min(jA, L.n[1] - 2), k);      // clamped: every lane takes part in the lane exchange
    double P[4][7], R[4][7];
    quad_gather(L.phi, L, cpA, lane, P);
    quad_gather(L.rho, L, cpA, lane, R);
    const double2 RA = *reinterpret_cast<const double2 *>(L.rh + cpA), RB = *reinterpret_cast<const double2 *>(L.rh + cpA + L.PX);
    if (act) {
      double Ap, diag, r[4];
      const double rhs[4] = { RA.x, RA.y, RB.x, RB.y };
      #pragma unroll
      for (int m = 0; m < 4; m++) {
        cc_apply_rho_vals(L, 2 * t + (m & 1), jA + (m >> 1), k, P[m], R[m], Ap, diag);
        r[m] = rhs[m] - Ap;
        rmax = nmax(rmax, fabs(r[m]));
      }
      *reinterpret_cast<double2 *>(L.res + cpA) = make_double2(r[0], r[1]);
      *reinterpret_cast<double2 *>(L.res + cpA + L.PX) = make_double2(r[2], r[3]);
    }
  }
  if (nrm) block_atomic_max(nrm, rmax);
}
// the same residual with the restriction inside: the thread's 2 x 2 block of planes 2K and 2K+1 is exactly the eight children of coarse cell
// (t, J, K); their residuals are summed in the order of kk_cc_restrict and stored as the coarse right-hand side (coarse phi = 0), the fine
// residual itself is not stored -- nothing else reads it (saves its 134 MB write and the 27 us restriction pass at 256^3)
__global__ void __launch_bounds__(256) kk_cc_residual_rho_pair_rst(CLev L, double *nrm, CLev C) {
  int bx, by, bz; xcd_block(bx, by, bz);
  const int lane = threadIdx.x;
  const int t = bx * 64 + lane, jA = 2 * (by * 4 + (int)threadIdx.y);
  const bool act = 2 * t + 1 < L.n[0] && jA + 1 < L.n[1];
  double rmax = 0.0;
  for (int K = bz; K < L.n[2] / 2; K += gridDim.z) {
    double s = 0.0;
    #pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int k = 2 * K + kk;
      const long cpA = cidx(L, 2 * min(t, L.n[0] / 2), min(jA, L.n[1] - 2), k);
      double P[4][7], R[4][7];
      quad_gather(L.phi, L, cpA, lane, P);
      quad_gather(L.rho, L, cpA, lane, R);
      const double2 RA = *reinterpret_cast<const double2 *>(L.rh + cpA), RB = *reinterpret_cast<const double2 *>(L.rh + cpA + L.PX);
      if (act) {
        double Ap, diag;
        const double rhs[4] = { RA.x, RA.y, RB.x, RB.y };
        #pragma unroll
        for (int m = 0; m < 4; m++) {
          cc_apply_rho_vals(L, 2 * t + (m & 1), jA + (m >> 1), k, P[m], R[m], Ap, diag);
          const double r = rhs[m] - Ap;
          rmax = nmax(rmax, fabs(r));
          s = (kk == 0 && m == 0) ? r : s + r;
        }
      }
    }
    if (act) {
      const long cc = cidx(C, t, jA >> 1, K);
      C.rh[cc] = s * 0.125;
      C.phi[cc] = 0.0;
    }
  }
  if (nrm) block_atomic_max(nrm, rmax);
}
__global__ void __launch_bounds__(256) kk_cc_residual(CLev L, double *nrm) { cc_residual_body<false>(L, nrm); }
__global__ void __launch_bounds__(256) kk_cc_residual_rho(CLev L, double *nrm) { cc_residual_body<true>(L, nrm); }

__global__ void kk_cc_restrict(CLev F, CLev C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= C.n[0] || j >= C.n[1]) return;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const long f = cidx(F, 2 * i, 2 * j, 2 * k);
  const double *r = F.res;
  double s = r[f] + r[f + 1] + r[f + sy] + r[f + sy + 1] + r[f + sz] + r[f + sz + 1] + r[f + sz + sy] + r[f + sz + sy + 1];
  const long cc = cidx(C, i, j, k);
  C.rh[cc] = s * 0.125;
  C.phi[cc] = 0.0;                     // the error equation starts from zero: saves a memset launch per level and cycle (ghost cells stay zero / are refreshed)
}

__global__ void kk_cc_prolong(CLev F, CLev C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= F.n[0] || j >= F.n[1]) return;
  const long f = cidx(F, i, j, k);
  F.phi[f] = F.phi[f] + C.phi[cidx(C, i >> 1, j >> 1, k >> 1)];
}

__global__ void kk_cc_coarsen_b(CLev F, CLev C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i > C.n[0] || j > C.n[1] || k > C.n[2]) return;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const long f = cidx(F, 2 * i, 2 * j, 2 * k), c = cidx(C, i, j, k);
  if (j < C.n[1] && k < C.n[2]) C.b[0][c] = (F.b[0][f] + F.b[0][f + sy] + F.b[0][f + sz] + F.b[0][f + sz + sy]) * 0.25;
  if (i < C.n[0] && k < C.n[2]) C.b[1][c] = (F.b[1][f] + F.b[1][f + 1] + F.b[1][f + sz] + F.b[1][f + sz + 1]) * 0.25;
  if (i < C.n[0] && j < C.n[1]) C.b[2][c] = (F.b[2][f] + F.b[2][f + 1] + F.b[2][f + sy] + F.b[2][f + sy + 1]) * 0.25;
}

// periodic images of the face ghosts of phi (edges/corners are not read by the 7-point operator)
__global__ void kk_cc_periodic(CLev L, int per0, int per1, int per2) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y * blockDim.y + threadIdx.y;
  const int face = blockIdx.z;                       // 0..5
  const int d = face >> 1, s = face & 1;
  const int per = d == 0 ? per0 : (d == 1 ? per1 : per2);
  if (!per) return;
  const int t1 = (d == 0) ? 1 : 0, t2 = (d == 2) ? 1 : 2;
  if (a >= L.n[t1] || b >= L.n[t2]) return;
  int g[3], q[3];
  g[t1] = q[t1] = a; g[t2] = q[t2] = b;
  g[d] = s ? L.n[d] : -1; q[d] = s ? 0 : L.n[d] - 1;
  L.phi[cidx(L, g[0], g[1], g[2])] = L.phi[cidx(L, q[0], q[1], q[2])];
}

// bottom solve: all sweeps of the coarsest level in ONE launch by one workgroup (the level is tiny and
// L2-resident; a launch per colour pass would be pure launch latency).  __syncthreads() orders the
// global-memory writes of one colour pass before the reads of the next within the workgroup.
__global__ void __launch_bounds__(1024) kk_cc_bottom(CLev L, int nsweeps, int per0, int per1, int per2) {
  const int nx = L.n[0], ny = L.n[1], nz = L.n[2];
  const int half = (nx + 1) / 2;
  const int tot = half * ny * nz;
  const bool anyper = per0 || per1 || per2;
  for (int s = 0; s < nsweeps; s++) for (int color = 0; color < 2; color++) {
    if (anyper) {
      const int m = max(nx, max(ny, nz));
      for (int t = threadIdx.x; t < 6 * m * m; t += blockDim.x) {
        const int face = t / (m * m), a = (t / m) % m, b = t % m;
        const int d = face >> 1, sd = face & 1;
        const int per = d == 0 ? per0 : (d == 1 ? per1 : per2);
        const int t1 = (d == 0) ? 1 : 0, t2 = (d == 2) ? 1 : 2;
        if (per && a < L.n[t1] && b < L.n[t2]) {
          int g[3], q[3];
          g[t1] = q[t1] = a; g[t2] = q[t2] = b;
          g[d] = sd ? L.n[d] : -1; q[d] = sd ? 0 : L.n[d] - 1;
          L.phi[cidx(L, g[0], g[1], g[2])] = L.phi[cidx(L, q[0], q[1], q[2])];
        }
      }
      __syncthreads();
    }
    for (int t = threadIdx.x; t < tot; t += blockDim.x) {
      const int k = t / (half * ny), j = (t / half) % ny;
      const int i = 2 * (t % half) + ((j + k + color) & 1);
      if (i < nx) {
        const long c = cidx(L, i, j, k);
        double Ap, diag; cc_apply(L, c, Ap, diag);
        if (diag != 0.0) L.phi[c] = L.phi[c] + (L.rh[c] - Ap) / diag;
      }
    }
    __syncthreads();
  }
}

// ---- the small end of a V-cycle in one launch (see kk_nd_tailcycle in mg_nd.hip) -------------------------------------------------------------
// Levels of at most 8^3 cells: smoothing, residual, restriction, bottom sweeps and prolongation by ONE workgroup, the per-cell code of
// kk_cc_bottom / cc_residual_body / kk_cc_restrict / kk_cc_prolong in the same order with barriers where the launches ended.
#define CC_TAIL_MAX 4
struct CcTailArgs { CLev L[CC_TAIL_MAX]; int nlev, nu1, nu2, nbot, per[3]; };
DEVI void wg_cc_periodic(const CLev &L, const int per[3]) {
  if (!(per[0] || per[1] || per[2])) return;
  const int m = max(L.n[0], max(L.n[1], L.n[2]));
  for (int t = threadIdx.x; t < 6 * m * m; t += blockDim.x) {
    const int face = t / (m * m), a = (t / m) % m, b = t % m;
    const int d = face >> 1, sd = face & 1;
    const int t1 = (d == 0) ? 1 : 0, t2 = (d == 2) ? 1 : 2;
    if (per[d] && a < L.n[t1] && b < L.n[t2]) {
      int g[3], q[3];
      g[t1] = q[t1] = a; g[t2] = q[t2] = b;
      g[d] = sd ? L.n[d] : -1; q[d] = sd ? 0 : L.n[d] - 1;
      L.phi[cidx(L, g[0], g[1], g[2])] = L.phi[cidx(L, q[0], q[1], q[2])];
    }
  }
  __syncthreads();
}
DEVI void wg_cc_gsrb(const CLev &L, int nsweeps, const int per[3]) {
  const int nx = L.n[0], ny = L.n[1], nz = L.n[2];
  const int half = (nx + 1) / 2, tot = half * ny * nz;
  for (int s = 0; s < nsweeps; s++) for (int color = 0; color < 2; color++) {
    wg_cc_periodic(L, per);
    for (int t = threadIdx.x; t < tot; t += blockDim.x) {
      const int k = t / (half * ny), j = (t / half) % ny;
      const int i = 2 * (t % half) + ((j + k + color) & 1);
      if (i < nx) {
        const long c = cidx(L, i, j, k);
        double Ap, diag; cc_apply(L, c, Ap, diag);
        if (diag != 0.0) L.phi[c] = L.phi[c] + (L.rh[c] - Ap) / diag;
      }
    }
    __syncthreads();
  }
}
DEVI void wg_cc_down(const CLev &F, const CLev &C, const int per[3]) {
  wg_cc_periodic(F, per);
  const int tot = F.n[0] * F.n[1] * F.n[2];
  for (int t = threadIdx.x; t < tot; t += blockDim.x) {
    const int i = t % F.n[0], j = (t / F.n[0]) % F.n[1], k = t / (F.n[0] * F.n[1]);
    const long c = cidx(F, i, j, k);
    double Ap, diag; cc_apply(F, c, Ap, diag, i, j, k);
    F.res[c] = F.rh[c] - Ap;
  }
  __syncthreads();
  const int ct = C.n[0] * C.n[1] * C.n[2];
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const double *r = F.res;
  for (int t = threadIdx.x; t < ct; t += blockDim.x) {
    const int i = t % C.n[0], j = (t / C.n[0]) % C.n[1], k = t / (C.n[0] * C.n[1]);
    const long f = cidx(F, 2 * i, 2 * j, 2 * k);
    double s = r[f] + r[f + 1] + r[f + sy] + r[f + sy + 1] + r[f + sz] + r[f + sz + 1] + r[f + sz + sy] + r[f + sz + sy + 1];
    const long cc = cidx(C, i, j, k);
    C.rh[cc] = s * 0.125;
    C.phi[cc] = 0.0;
  }
  __syncthreads();
}
DEVI void wg_cc_up(const CLev &F, const CLev &C) {
  const int tot = F.n[0] * F.n[1] * F.n[2];
  for (int t = threadIdx.x; t < tot; t += blockDim.x) {
    const int i = t % F.n[0], j = (t / F.n[0]) % F.n[1], k = t / (F.n[0] * F.n[1]);
    const long f = cidx(F, i, j, k);
    F.phi[f] = F.phi[f] + C.phi[cidx(C, i >> 1, j >> 1, k >> 1)];
  }
  __syncthreads();
}
__global__ void __launch_bounds__(1024) kk_cc_tailcycle(CcTailArgs T) {
  #pragma unroll
  for (int l = 0; l < CC_TAIL_MAX - 1; l++)
    if (l < T.nlev - 1) { wg_cc_gsrb(T.L[l], T.nu1, T.per); wg_cc_down(T.L[l], T.L[l + 1], T.per); }
  #pragma unroll
  for (int l = 0; l < CC_TAIL_MAX; l++)
    if (l == T.nlev - 1) wg_cc_gsrb(T.L[l], T.nbot, T.per);
  #pragma unroll
  for (int l = CC_TAIL_MAX - 2; l >= 0; l--)
    if (l < T.nlev - 1) { wg_cc_up(T.L[l], T.L[l + 1]); wg_cc_gsrb(T.L[l], T.nu2, T.per); }
}

// ---- levels of 16^3 .. 64^3 cells: one launch down, one launch up (round 3) ------------------------------------------------------------------
// Below the 128^3 level a V-cycle is a chain of ~5 us launches: eleven per level and cycle (four colour passes, residual, restriction;
// prolongation, four colour passes), each waiting for the one before -- ~300 launches per MAC solve on the 64^3, 32^3 and 16^3 levels of a 256^3
// problem, 2.3 ms of a 14 ms projection for 1/500 of its cells.  The bytes are irrelevant there (a 64^3 level is 2 MB per field, L2 resident);
// what costs is the number of dependent launches.  Both halves of a level's visit therefore run as ONE launch each, tiled through LDS with
// redundant halo work instead of synchronisation between workgroups ("temporal blocking"):
//   kk_cc_lds_down  a workgroup owns an 8^3 tile.  phi (zero on entry: the error equation) lives in LDS on the tile grown by 5 cells; colour
//                   pass p = 0..3 updates the cells of its colour within 4 - p cells of the tile -- what the tile's own cells will need by
//                   the last pass -- then the residual on the tile, whose 4^3 parents it restricts itself.  The smoothed phi goes to L.res
//                   (nothing else uses the residual array of such a level), the coarse right-hand side and a zero coarse phi to the next level.
//   kk_cc_lds_up    loads L.res + the piecewise-constant correction on the tile grown by 4 cells, runs the four post-smoothing passes the same
//                   way (reach 3, 2, 1, 0) and writes phi of the tile.
// A workgroup reads only what no workgroup of the same launch writes (down: rhs and coefficients; up: L.res and the coarse phi), so there is
// nothing to order between workgroups.  Halo cells are recomputed by every tile that needs them, from the same operands with the same
// expressions (cc_apply_vals, the operator of every other pass), hence the same bits: the results are those of the separate launches
// (tests/test_kernels_gpu.py::test_multigrid_launch_variants_agree_bit_for_bit runs both).  A thread keeps the six face coefficients, the
// right-hand side (and alpha) of its four cells in registers for the whole launch.  Conditions: one box, no periodic face (ghost cells of phi
// are zero, the boundary conditions sit in the coefficients), extents multiples of 8, nu1 = nu2 = 2; anything else takes the separate launches.
constexpr int LT = 8;                       // tile width
constexpr int LW = LT + 8;                  // the tile grown by 4: the cells a workgroup holds coefficients for (16^3 = 4 per thread)
struct LdsCell { double b[6], rh, a0; };
template <bool ALPHA> DEVI void lds_load_cell(const CLev &L, long c, LdsCell &q) {
  const long sy = L.PX, sz = (long)L.PX * L.PY;
  q.b[0] = L.b[0][c]; q.b[1] = L.b[0][c + 1]; q.b[2] = L.b[1][c]; q.b[3] = L.b[1][c + sy]; q.b[4] = L.b[2][c]; q.b[5] = L.b[2][c + sz];
  q.rh = L.rh[c]; q.a0 = ALPHA ? L.alpha[c] : 0.0;
}
// one colour pass on the four cells (x, y, z0 .. z0+3) of this thread -- coordinates in the grown tile, the tile itself at 4 .. 4+LT-1; s = phi in
// LDS (row stride SX, plane stride SP, `at` = the thread's first cell); par0 = parity of the first cell's global index sum; reach = how far from the
// tile this pass still has to be right
template <bool ALPHA, int SX, int SP> DEVI void lds_pass(const CLev &L, double *s, const LdsCell (&q)[4], const bool (&inside)[4], int at, int x, int y, int z0, int par0, int color, int reach) {
  const bool xy_ok = x >= 4 - reach && x < 4 + LT + reach && y >= 4 - reach && y < 4 + LT + reach;
  #pragma unroll
  for (int m = 0; m < 4; m++) {
    const int z = z0 + m;
    if (!(xy_ok && inside[m] && z >= 4 - reach && z < 4 + LT + reach && ((par0 + m + color) & 1) == 0)) continue;
    const int a = at + m * SP;
    const double p[7] = { s[a], s[a - 1], s[a + 1], s[a - SX], s[a + SX], s[a - SP], s[a + SP] };
    double Ap, diag; cc_apply_vals(L, p, q[m].b, q[m].a0, ALPHA, Ap, diag);
    if (diag != 0.0) s[a] = p[0] + (q[m].rh - Ap) / diag;
  }
}
template <bool ALPHA> __global__ void __launch_bounds__(1024) kk_cc_lds_down(CLev L, CLev C) {
  constexpr int SX = LW + 2, SP = SX * SX;                    // phi on the tile grown by 5: the outermost layer stays at its initial zero
  __shared__ double s[SX * SX * SX];
  __shared__ double r[LT * LT * LT];
  const int t = threadIdx.x, x = t & (LW - 1), y = (t >> 4) & (LW - 1), z0 = 4 * (t >> 8);
  const int i = (int)blockIdx.x * LT - 4 + x, j = (int)blockIdx.y * LT - 4 + y, k0 = (int)blockIdx.z * LT - 4 + z0;
  for (int a = t; a < SX * SX * SX; a += 1024) s[a] = 0.0;      // phi = 0 on entry (written by the restriction that fed this level)
  LdsCell q[4]; bool inside[4];
  const bool ij_in = i >= 0 && i < L.n[0] && j >= 0 && j < L.n[1];
  #pragma unroll
  for (int m = 0; m < 4; m++) {
    inside[m] = ij_in && k0 + m >= 0 && k0 + m < L.n[2];
    if (inside[m]) lds_load_cell<ALPHA>(L, cidx(L, i, j, k0 + m), q[m]);
    else { for (int e = 0; e < 6; e++) q[m].b[e] = 0.0; q[m].rh = 0.0; q[m].a0 = 0.0; }
  }
  const int at = (x + 1) + SX * (y + 1) + SP * (z0 + 1), par0 = (i + j + k0) & 1;
  __syncthreads();
  #pragma unroll
  for (int pass = 0; pass < 4; pass++) { lds_pass<ALPHA, SX, SP>(L, s, q, inside, at, x, y, z0, par0, pass & 1, 4 - pass); __syncthreads(); }
  // residual of the tile's cells (cc_residual_body), smoothed phi to L.res
  const bool xy_tile = x >= 4 && x < 4 + LT && y >= 4 && y < 4 + LT;
  #pragma unroll
  for (int m = 0; m < 4; m++) {
    const int z = z0 + m;
    if (!(xy_tile && z >= 4 && z < 4 + LT)) continue;
    const int a = at + m * SP;
    const double p[7] = { s[a], s[a - 1], s[a + 1], s[a - SX], s[a + SX], s[a - SP], s[a + SP] };
    double Ap, diag; cc_apply_vals(L, p, q[m].b, q[m].a0, ALPHA, Ap, diag);
    r[(x - 4) + LT * ((y - 4) + LT * (z - 4))] = q[m].rh - Ap;
    L.res[cidx(L, i, j, k0 + m)] = p[0];
  }
  __syncthreads();
  if (t < (LT / 2) * (LT / 2) * (LT / 2)) {                      // kk_cc_restrict on the 4^3 parents of the tile
    const int ci = t & 3, cj = (t >> 2) & 3, ck = t >> 4;
    const int f = 2 * ci + LT * (2 * cj + LT * 2 * ck);
    const double sum = r[f] + r[f + 1] + r[f + LT] + r[f + LT + 1] + r[f + LT * LT] + r[f + LT * LT + 1] + r[f + LT * LT + LT] + r[f + LT * LT + LT + 1];
    const long cc = cidx(C, (int)blockIdx.x * (LT / 2) + ci, (int)blockIdx.y * (LT / 2) + cj, (int)blockIdx.z * (LT / 2) + ck);
    C.rh[cc] = sum * 0.125;
    C.phi[cc] = 0.0;
  }
}
template <bool ALPHA> __global__ void __launch_bounds__(1024) kk_cc_lds_up(CLev L, CLev C) {
  constexpr int SX = LW, SP = SX * SX;
  __shared__ double s[SX * SX * SX];
  const int t = threadIdx.x, x = t & (LW - 1), y = (t >> 4) & (LW - 1), z0 = 4 * (t >> 8);
  const int i = (int)blockIdx.x * LT - 4 + x, j = (int)blockIdx.y * LT - 4 + y, k0 = (int)blockIdx.z * LT - 4 + z0;
  LdsCell q[4]; bool inside[4];
  const bool ij_in = i >= 0 && i < L.n[0] && j >= 0 && j < L.n[1];
  const int at = x + SX * y + SP * z0, par0 = (i + j + k0) & 1;
  #pragma unroll
  for (int m = 0; m < 4; m++) {
    inside[m] = ij_in && k0 + m >= 0 && k0 + m < L.n[2];
    double v = 0.0;
    if (inside[m]) {
      const long c = cidx(L, i, j, k0 + m);
      lds_load_cell<ALPHA>(L, c, q[m]);
      v = L.res[c] + C.phi[cidx(C, i >> 1, j >> 1, (k0 + m) >> 1)];        // kk_cc_prolong on the phi kk_cc_lds_down left in L.res
    } else { for (int e = 0; e < 6; e++) q[m].b[e] = 0.0; q[m].rh = 0.0; q[m].a0 = 0.0; }
    s[at + m * SP] = v;
  }
  __syncthreads();
  #pragma unroll
  for (int pass = 0; pass < 4; pass++) { lds_pass<ALPHA, SX, SP>(L, s, q, inside, at, x, y, z0, par0, pass & 1, 3 - pass); __syncthreads(); }
  if (x >= 4 && x < 4 + LT && y >= 4 && y < 4 + LT) {
    #pragma unroll
    for (int m = 0; m < 4; m++) if (z0 + m >= 4 && z0 + m < 4 + LT) L.phi[cidx(L, i, j, k0 + m)] = s[at + m * SP];
  }
}

// ---- transfers between BoxLib-layout multifabs and level 0 ---------------------------------------------
__global__ void kk_cc_load(CLev L, FV rh, FV phi, FV alpha, FV bx, FV by, FV bz, int lo0, int lo1, int lo2, int ebc00, int ebc01, int ebc10, int ebc11, int ebc20, int ebc21) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i > L.n[0] || j > L.n[1] || k > L.n[2]) return;
  const long c = cidx(L, i, j, k);
  const bool ci = i < L.n[0], cj = j < L.n[1], ck = k < L.n[2];
  if (ci && cj && ck) { L.phi[c] = fv_get(phi, lo0 + i, lo1 + j, lo2 + k); if (L.alpha) L.alpha[c] = fv_get(alpha, lo0 + i, lo1 + j, lo2 + k); }
  if (cj && ck) {
    double v = fv_get(bx, lo0 + i, lo1 + j, lo2 + k);
    int e = (i == 0) ? ebc00 : (i == L.n[0] ? ebc01 : VDN_BC_INT);
    if (e == VDN_BC_NEU) v = 0.0; else if (e == VDN_BC_DIR) v = 2.0 * v;
    L.b[0][c] = v;
  }
  if (ci && ck) {
    double v = fv_get(by, lo0 + i, lo1 + j, lo2 + k);
    int e = (j == 0) ? ebc10 : (j == L.n[1] ? ebc11 : VDN_BC_INT);
    if (e == VDN_BC_NEU) v = 0.0; else if (e == VDN_BC_DIR) v = 2.0 * v;
    L.b[1][c] = v;
  }
  if (ci && cj) {
    double v = fv_get(bz, lo0 + i, lo1 + j, lo2 + k);
    int e = (k == 0) ? ebc20 : (k == L.n[2] ? ebc21 : VDN_BC_INT);
    if (e == VDN_BC_NEU) v = 0.0; else if (e == VDN_BC_DIR) v = 2.0 * v;
    L.b[2][c] = v;
  }
}
// ---- macproject's fast path on one level (round 3) --------------------------------------------------------------------------------------
// The multifabs rh, phi and beta of macproject.f90:63-76 only carry -div(umac) + mac_rhs, zeros and 2 / (rho_i + rho_i-1) into the solver and
// phi out of it.  With the finest level's coefficients recomputed from rho anyway, the level takes its right-hand side straight from the MAC
// field (kk_cc_load_divumac: divumac_K's expression, the max norm of macproject.f90:77-89 on the way), the 128^3 level its coefficients from
// rho (kk_cc_coarsen_b_rho: beta_of = the folded value kk_cc_load stores, kk_cc_coarsen_b's mean), and mkumac reads phi from the level array
// and forms beta from rho (mkumac_rho_K).  1.04 -> 0.4 ms of fills, copies and passes per 256^3 projection; VDN_MAC_FAST=0 restores the multifabs.
__global__ void kk_cc_load_divumac(CLev L, FV um, FV vm, FV wm, FV macrhs, double dxi0, double dxi1, double dxi2, int lo0, int lo1, int lo2,
                                   int ebc00, int ebc01, int ebc10, int ebc11, int ebc20, int ebc21, double *nrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  double rmax = 0.0;
  if (i < L.n[0] && j < L.n[1])
    for (int k = blockIdx.z; k < L.n[2]; k += gridDim.z) {
      const int gi = lo0 + i, gj = lo1 + j, gk = lo2 + k;
      const double div = (fv_get(um, gi + 1, gj, gk) - fv_get(um, gi, gj, gk)) * dxi0
                       + (fv_get(vm, gi, gj + 1, gk) - fv_get(vm, gi, gj, gk)) * dxi1
                       + (fv_get(wm, gi, gj, gk + 1) - fv_get(wm, gi, gj, gk)) * dxi2;
      double r = div * -1.0 + fv_get(macrhs, gi, gj, gk);
      rmax = nmax(rmax, fabs(r));
      // kk_cc_load_rh with the zero initial guess of macproject: the Dirichlet term is b * 0 * h^-2 = +0 (b is finite): only the sign of a zero can change
      if (i == 0 && ebc00 == VDN_BC_DIR) r = r + 0.0;
      if (i == L.n[0] - 1 && ebc01 == VDN_BC_DIR) r = r + 0.0;
      if (j == 0 && ebc10 == VDN_BC_DIR) r = r + 0.0;
      if (j == L.n[1] - 1 && ebc11 == VDN_BC_DIR) r = r + 0.0;
      if (k == 0 && ebc20 == VDN_BC_DIR) r = r + 0.0;
      if (k == L.n[2] - 1 && ebc21 == VDN_BC_DIR) r = r + 0.0;
      L.rh[cidx(L, i, j, k)] = r;
    }
  block_atomic_max(nrm, rmax);
}
// coarse face coefficients from the fine level's DENSITY: the folded fine value (kk_cc_load) of face index I along d is beta_of(rho_I, rho_I-1)
DEVI double cc_face_rho(const CLev &F, int d, int i, int j, int k) {
  const long c = cidx(F, i, j, k);
  const long st = d == 0 ? 1 : (d == 1 ? (long)F.PX : (long)F.PX * F.PY);
  const int q = d == 0 ? i : (d == 1 ? j : k);
  return beta_of(F.rho[c], F.rho[c - st], q == 0 || q == F.n[d], q == 0 ? F.fold[d][0] : F.fold[d][1]);
}
__global__ void kk_cc_coarsen_b_rho(CLev F, CLev C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i > C.n[0] || j > C.n[1] || k > C.n[2]) return;
  const long c = cidx(C, i, j, k);
  const int I = 2 * i, J = 2 * j, K = 2 * k;
  if (j < C.n[1] && k < C.n[2]) C.b[0][c] = (cc_face_rho(F, 0, I, J, K) + cc_face_rho(F, 0, I, J + 1, K) + cc_face_rho(F, 0, I, J, K + 1) + cc_face_rho(F, 0, I, J + 1, K + 1)) * 0.25;
  if (i < C.n[0] && k < C.n[2]) C.b[1][c] = (cc_face_rho(F, 1, I, J, K) + cc_face_rho(F, 1, I + 1, J, K) + cc_face_rho(F, 1, I, J, K + 1) + cc_face_rho(F, 1, I + 1, J, K + 1)) * 0.25;
  if (i < C.n[0] && j < C.n[1]) C.b[2][c] = (cc_face_rho(F, 2, I, J, K) + cc_face_rho(F, 2, I + 1, J, K) + cc_face_rho(F, 2, I, J + 1, K) + cc_face_rho(F, 2, I + 1, J + 1, K)) * 0.25;
}
// phi alone (a kept hierarchy takes a new right-hand side and initial guess: cc_reload)
__global__ void kk_cc_load_phi(CLev L, FV phi, int lo0, int lo1, int lo2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= L.n[0] || j >= L.n[1] || k >= L.n[2]) return;
  L.phi[cidx(L, i, j, k)] = fv_get(phi, lo0 + i, lo1 + j, lo2 + k);
}
// the density with its ghost layer, for the on-the-fly face coefficients of the finest level
__global__ void kk_cc_load_rho(CLev L, double *dst, FV rho, int lo0, int lo1, int lo2) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] || j > L.n[1] || k > L.n[2]) return;
  dst[cidx(L, i, j, k)] = fv_get(rho, lo0 + i, lo1 + j, lo2 + k);
}
// right-hand side, with the inhomogeneous Dirichlet data moved into it: the ghost cells of the incoming phi hold the
// boundary-FACE values (multifab_physbc EXT_DIR; visc_solve hands unew over that way, viscsolve.f90:270); the face term
// 2b(phi_i - phi_b)/h^2 keeps its phi_i part in the operator (b := 2b, zero ghost) and its phi_b part goes here, in the
// order x-lo, x-hi, y-lo, y-hi, z-lo, z-hi (same as the oracle).  Runs after kk_cc_load (needs the folded b).
// zg: the initial guess is zero, ghost cells included, and `phi` is not read (the products with 0.0 are kept: r + 0.0 is not always r)
__global__ void kk_cc_load_rh(CLev L, FV rh, FV phi, int lo0, int lo1, int lo2, int ebc00, int ebc01, int ebc10, int ebc11, int ebc20, int ebc21, int zg = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= L.n[0] || j >= L.n[1]) return;
  const long c = cidx(L, i, j, k);
  const long sy = L.PX, sz = (long)L.PX * L.PY;
  const int gi = lo0 + i, gj = lo1 + j, gk = lo2 + k;
  double r = fv_get(rh, gi, gj, gk);
  if (i == 0 && ebc00 == VDN_BC_DIR)            r = r + L.b[0][c] * (zg ? 0.0 : fv_get(phi, gi - 1, gj, gk)) * L.hi2[0];
  if (i == L.n[0] - 1 && ebc01 == VDN_BC_DIR)   r = r + L.b[0][c + 1] * (zg ? 0.0 : fv_get(phi, gi + 1, gj, gk)) * L.hi2[0];
  if (j == 0 && ebc10 == VDN_BC_DIR)            r = r + L.b[1][c] * (zg ? 0.0 : fv_get(phi, gi, gj - 1, gk)) * L.hi2[1];
  if (j == L.n[1] - 1 && ebc11 == VDN_BC_DIR)   r = r + L.b[1][c + sy] * (zg ? 0.0 : fv_get(phi, gi, gj + 1, gk)) * L.hi2[1];
  if (k == 0 && ebc20 == VDN_BC_DIR)            r = r + L.b[2][c] * (zg ? 0.0 : fv_get(phi, gi, gj, gk - 1)) * L.hi2[2];
  if (k == L.n[2] - 1 && ebc21 == VDN_BC_DIR)   r = r + L.b[2][c + sz] * (zg ? 0.0 : fv_get(phi, gi, gj, gk + 1)) * L.hi2[2];
  L.rh[c] = r;
}
// mean of the 8 children of a cell field (alpha)
__global__ void kk_cc_coarsen_cell(CLev F, const double *src, CLev C, double *dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= C.n[0] || j >= C.n[1]) return;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const long f = cidx(F, 2 * i, 2 * j, 2 * k);
  double s = src[f] + src[f + 1] + src[f + sy] + src[f + sy + 1] + src[f + sz] + src[f + sz + 1] + src[f + sz + sy] + src[f + sz + sy + 1];
  dst[cidx(C, i, j, k)] = s * 0.125;
}

// phi back, incl. the face ghost layer the closure implies (Neumann: phi_i, Dirichlet: -phi_i, periodic: image)
// add: acc += the solution on the cells of the box as well (the composite solve adds its coarse correction to phi of level 0)
__global__ void kk_cc_store(CLev L, FV phi, int lo0, int lo1, int lo2, int ebc00, int ebc01, int ebc10, int ebc11, int ebc20, int ebc21, int add = 0, FV acc = FV()) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] || j > L.n[1] || k > L.n[2]) return;
  const int out = (i < 0) + (i == L.n[0]) + (j < 0) + (j == L.n[1]) + (k < 0) + (k == L.n[2]);
  if (out > 1) return;
  double v;
  if (out == 0) v = L.phi[cidx(L, i, j, k)];
  else {
    int e; int qi = i, qj = j, qk = k;
    if (i < 0) { e = ebc00; qi = 0; } else if (i == L.n[0]) { e = ebc01; qi = L.n[0] - 1; }
    else if (j < 0) { e = ebc10; qj = 0; } else if (j == L.n[1]) { e = ebc11; qj = L.n[1] - 1; }
    else if (k < 0) { e = ebc20; qk = 0; } else { e = ebc21; qk = L.n[2] - 1; }
    if (e == VDN_BC_NEU) v = L.phi[cidx(L, qi, qj, qk)];
    else if (e == VDN_BC_DIR) v = -L.phi[cidx(L, qi, qj, qk)];
    else v = L.phi[cidx(L, i, j, k)];          // periodic image already in the ghost slot
  }
  fv_at(phi, lo0 + i, lo1 + j, lo2 + k) = v;
  if (add && out == 0) fv_at(acc, lo0 + i, lo1 + j, lo2 + k) = fv_get(acc, lo0 + i, lo1 + j, lo2 + k) + v;
}

// ---- gather of the first agglomerated level ------------------------------------------------------------------
// Below a box extent of 4 the per-box levels stop; the next coarser level of the WHOLE domain (and everything
// under it) is held redundantly by every rank as one box ("agglomerated tail"): latency-bound work is not worth
// distributing over xGMI.  Each rank restricts its boxes' residuals into a packed buffer, one all-gather makes
// every rank's buffer visible everywhere, and an unpack kernel assembles the global coarse right-hand side.
// The face coefficients of the tail's first level travel the same way once per solve.
struct GBox { int c0[3]; int n[3]; long off; };     // where a box's coarse cells sit in the tail level / in the buffer

__global__ void kk_cc_restrict_pack(CLev F, const double *r, double *buf, long off, int nx, int ny, int nz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= nx || j >= ny) return;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const long f = cidx(F, 2 * i, 2 * j, 2 * k);
  double s = r[f] + r[f + 1] + r[f + sy] + r[f + sy + 1] + r[f + sz] + r[f + sz + 1] + r[f + sz + sy] + r[f + sz + sy + 1];
  buf[off + i + (long)nx * (j + (long)ny * k)] = s * 0.125;
}
// coarse face coefficients of one box into the buffer: [bx (nx+1,ny,nz) | by (nx,ny+1,nz) | bz (nx,ny,nz+1)]
__global__ void kk_cc_coarsen_b_pack(CLev F, double *buf, long off, int nx, int ny, int nz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i > nx || j > ny || k > nz) return;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const long f = cidx(F, 2 * i, 2 * j, 2 * k);
  const long o1 = off + (long)(nx + 1) * ny * nz, o2 = o1 + (long)nx * (ny + 1) * nz;
  if (j < ny && k < nz) buf[off + i + (long)(nx + 1) * (j + (long)ny * k)] = (F.b[0][f] + F.b[0][f + sy] + F.b[0][f + sz] + F.b[0][f + sz + sy]) * 0.25;
  if (i < nx && k < nz) buf[o1 + i + (long)nx * (j + (long)(ny + 1) * k)] = (F.b[1][f] + F.b[1][f + 1] + F.b[1][f + sz] + F.b[1][f + sz + 1]) * 0.25;
  if (i < nx && j < ny) buf[o2 + i + (long)nx * (j + (long)ny * k)] = (F.b[2][f] + F.b[2][f + 1] + F.b[2][f + sy] + F.b[2][f + sy + 1]) * 0.25;
}
__global__ void kk_cc_unpack_rh(CLev T, double *dst, const double *buf, const GBox *gb) {
  const GBox g = gb[blockIdx.z];
  const int tot = g.n[0] * g.n[1] * g.n[2];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += gridDim.x * blockDim.x) {
    const int i = t % g.n[0], j = (t / g.n[0]) % g.n[1], k = t / (g.n[0] * g.n[1]);
    const long cc = cidx(T, g.c0[0] + i, g.c0[1] + j, g.c0[2] + k);
    dst[cc] = buf[g.off + t];
    T.phi[cc] = 0.0;                    // zero initial guess of the tail's first level (see kk_cc_restrict)
  }
}
__global__ void kk_cc_unpack_b(CLev T, const double *buf, const GBox *gb) {
  const GBox g = gb[blockIdx.z];
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  const long n0 = (long)(nx + 1) * ny * nz, n1 = (long)nx * (ny + 1) * nz, n2 = (long)nx * ny * (nz + 1);
  for (long t = blockIdx.x * blockDim.x + threadIdx.x; t < n0 + n1 + n2; t += gridDim.x * blockDim.x) {
    int d; long u = t;
    if (u < n0) d = 0; else if (u < n0 + n1) { d = 1; u -= n0; } else { d = 2; u -= n0 + n1; }
    const int ex = nx + (d == 0), ey = ny + (d == 1);
    const int i = (int)(u % ex), j = (int)((u / ex) % ey), k = (int)(u / ((long)ex * ey));
    T.b[d][cidx(T, g.c0[0] + i, g.c0[1] + j, g.c0[2] + k)] = buf[g.off + t];
  }
}
// prolongation from the (global, replicated) tail level into a box of the last distributed level
__global__ void kk_cc_prolong_tail(CLev F, CLev T, int c00, int c01, int c02) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= F.n[0] || j >= F.n[1]) return;
  const long f = cidx(F, i, j, k);
  F.phi[f] = F.phi[f] + T.phi[cidx(T, c00 + (i >> 1), c01 + (j >> 1), c02 + (k >> 1))];
}

// nested iteration (cc_fmg): phi_F = LINEAR interpolation of the coarse solution from face neighbours only -- (p0 + px + py + pz)/4 with px, py, pz
// the coarse neighbours on the fine cell's side (oracle: cc_prolong_linear).  c0: coarse index of the fine box's cell 0 inside C;  e: what lies
// behind each face of C -- Neumann: the cell itself, Dirichlet: minus the cell, anything else: the ghost cell (halo / periodic image, filled by the caller)
struct ProlongLinArgs { int c0[3]; int e[3][2]; };
__global__ void kk_cc_prolong_lin(CLev F, CLev C, ProlongLinArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= F.n[0] || j >= F.n[1]) return;
  const int I = A.c0[0] + (i >> 1), J = A.c0[1] + (j >> 1), K = A.c0[2] + (k >> 1);
  const long sy = C.PX, sz = (long)C.PX * C.PY;
  const long c = cidx(C, I, J, K);
  const double p0 = C.phi[c];
  double px, py, pz;
  #define NBV(out, q, odd, n, elo, ehi, stride)                                                             \
    { const int m = (q) + ((odd) ? 1 : -1);                                                                 \
      const int t = m < 0 ? (elo) : (m >= (n) ? (ehi) : VDN_BC_INT);                                         \
      out = (t == VDN_BC_NEU) ? p0 : ((t == VDN_BC_DIR) ? -p0 : C.phi[c + ((odd) ? (stride) : -(stride))]); }
  NBV(px, I, i & 1, C.n[0], A.e[0][0], A.e[0][1], 1L)
  NBV(py, J, j & 1, C.n[1], A.e[1][0], A.e[1][1], sy)
  NBV(pz, K, k & 1, C.n[2], A.e[2][0], A.e[2][1], sz)
  #undef NBV
  F.phi[cidx(F, i, j, k)] = 0.25 * (((p0 + px) + py) + pz);
}

// ---- host side ------------------------------------------------------------------------------------------
struct CBox { CLev L; int lo[3]; int gidx; int hmask = 63; /* faces whose ghost cells come from the halo exchange */
              CSplit sp; /* the box by colour (CDLev::split) */ };                    // one local box on one distributed level; lo = global index of its cell 0
struct CDLev { std::vector<CBox> boxes; XPlan *halo = nullptr; int ng[3]; /* global extents of the level */ bool single_box = false;
               bool res_restricted = false; /* the last residual pass already restricted into the next level */
               bool split = false;      /* macproject's finest level: phi, rhs, rho of every box by colour (cc_split_setup); phi of the level arrays is stale between cc_to_split / cc_from_split */
               XPlan *shalo[2] = { nullptr, nullptr };      /* round 6: the ghost exchange of phi[colour] on the split arrays (levels with a halo) */
               std::vector<XBoxInfo> xb; vdn_box lpd; unsigned long la_uid = 0; int la_lev = 0;      /* the level's global box list, kept for cc_split_setup's plans */ };
struct CCMG {
  std::vector<CDLev> dlev;          // distributed levels (finest first)
  std::vector<CLev> tail;           // agglomerated levels, whole domain, replicated on every rank
  int per[3];
  double *d_nrm;
  // gather machinery
  std::vector<GBox> gb_rh, gb_b; GBox *d_gb_rh = nullptr, *d_gb_b = nullptr;
  double *sendbuf = nullptr, *recvbuf = nullptr; size_t cnt_rh = 0, cnt_b = 0;   // per-rank counts (doubles)
  std::vector<long> loc_off_rh, loc_off_b;                                       // offsets of my boxes inside my send buffer
};

static dim3 g3(int nx, int ny, int nz, dim3 b) { return dim3((nx + b.x - 1) / b.x, (ny + b.y - 1) / b.y, nz); }
static const dim3 BLK(64, 4, 1);

// ---- macproject's finest level by colour (kk_cc_gsrb_rho_split): host side --------------------------------------------------------------------
// The level array keeps rhs and rho (level 1's coefficients, the nested iteration and the residual read them there); phi lives in the split arrays
// from cc_to_split (after the nested iteration) to cc_from_split (before the residual pass, which reads the level array, and at the end of the solve).
static bool mac_split_on() { static const bool b = !(vdn_env("VDN_MAC_SPLIT") && atoi(vdn_env("VDN_MAC_SPLIT")) == 0); return b; }
// hm: the shell behind these faces is left to kk_cc_gsrb_split_shell (cc_gsrb_d's overlap of the exchange with the pass)
template <int ADD> static inline void launch_gsrb_split(const CBox &B, int color, hipStream_t st, const CLev &C, int k0 = 0, int k1 = -1, int hm = 0) {
  const CLev &L = B.L;
  if (k1 < 0) k1 = L.n[2];
  const dim3 g((unsigned)((L.n[0] / 4 + 63) / 64), (unsigned)((L.n[1] + 7) / 8), (unsigned)(k1 - k0));
  // the second colour walks the planes downwards: what the first colour's pass touched last is what it reads first (Infinity Cache; VDN_MAC_KFLIP=0: both upwards)
  if (L.cmu > 0.0) hipLaunchKernelGGL((kk_cc_gsrb_rho_split<ADD, true>), g, dim3(64, 8, 1), 0, st, L, B.sp, color, C, (mac_kflip() && color) ? 1 : 0, k0, hm);
  else hipLaunchKernelGGL((kk_cc_gsrb_rho_split<ADD, false>), g, dim3(64, 8, 1), 0, st, L, B.sp, color, C, (mac_kflip() && color) ? 1 : 0, k0, hm);
}
static inline void launch_residual_split_rst(const CLev &L, const CSplit &S, double *nrm, const CLev &C, int Ka, int Kb, hipStream_t st) {
  const dim3 g((unsigned)((L.n[0] / 4 + 63) / 64), (unsigned)((L.n[1] / 2 + 3) / 4), (unsigned)std::min(Kb - Ka, 16));
  if (L.cmu > 0.0) hipLaunchKernelGGL(kk_cc_residual_rho_split_rst<true>, g, BLK, 0, st, L, S, nrm, C, Ka, Kb);
  else hipLaunchKernelGGL(kk_cc_residual_rho_split_rst<false>, g, BLK, 0, st, L, S, nrm, C, Ka, Kb);
}
static inline void launch_gsrb_split_shell(const CBox &B, int color, hipStream_t st, int hm) {
  if (!hm) return;
  const CLev &L = B.L;
  const int m = std::max(L.n[0], std::max(L.n[1], L.n[2]));
  hipLaunchKernelGGL(kk_cc_gsrb_split_shell, dim3((unsigned)((m + 63) / 64), (unsigned)((m + 3) / 4), (unsigned)__builtin_popcount(hm)), dim3(64, 4, 1), 0, st, L, B.sp, color, hm);
}
static void cc_to_split(const CDLev &DL, int what) {
  for (const CBox &B : DL.boxes) {
    const CLev &L = B.L;
    const dim3 g((unsigned)((B.sp.PXH + 63) / 64), (unsigned)((L.n[1] + 2 + 3) / 4), (unsigned)(L.n[2] + 2));
    hipLaunchKernelGGL(kk_cc_to_split, g, BLK, 0, ctx().stream, L, B.sp, what);
  }
}
static void cc_from_split(const CDLev &DL) {
  for (const CBox &B : DL.boxes) hipLaunchKernelGGL(kk_cc_from_split, g3(B.L.n[0] / 2, B.L.n[1], B.L.n[2], BLK), BLK, 0, ctx().stream, B.L, B.sp);
}
// the ghost entries of phi[colour] from the neighbouring boxes / periodic images (no plan: one box without periodic faces)
static void split_halo(const CDLev &DL, int colour, hipStream_t st = nullptr) { if (DL.shalo[colour]) xplan_run(DL.shalo[colour], st); }
// The finest level of macproject's solve goes by colour when: the density form, the default launch forms, a second distributed level with the same boxes; every local box
// with the extents the paired kernels take and even corner indices (its colours are then the global ones and a row's parity is the same on both sides of a box face),
// and enough cells on this rank that the level does not live in the caches anyway.  Round 5 took one box without periodic faces only; round 6: any box list, periodic faces,
// several ranks -- the ghost exchange runs on the split arrays themselves (cc_split_setup).
static bool cc_split_ok(const CCMG &M) {
  static const bool dflt = !(vdn_env("VDN_GSRB_PAIR") && atoi(vdn_env("VDN_GSRB_PAIR")) == 0) && !(vdn_env("VDN_MG_RESTRICT_FUSED") && atoi(vdn_env("VDN_MG_RESTRICT_FUSED")) == 0) &&
                           !(vdn_env("VDN_MG_PROLONG_FUSED") && atoi(vdn_env("VDN_MG_PROLONG_FUSED")) == 0);
  if (!mac_split_on() || !dflt || M.dlev.size() < 2 || ctx().prm.mg_nu1 < 1 || ctx().prm.mg_nu2 < 1) return false;
  const CDLev &D0 = M.dlev[0];
  if (D0.boxes.empty() || D0.boxes.size() != M.dlev[1].boxes.size()) return false;
  static const bool halo_ok = !(vdn_env("VDN_MAC_SPLIT_HALO") && atoi(vdn_env("VDN_MAC_SPLIT_HALO")) == 0);      // 0: round 5's rule (one box, no exchange)
  if (D0.halo && !halo_ok) return false;
  long cells = 0;
  for (const CBox &B : D0.boxes) {
    const CLev &L = B.L;
    if (!((L.rho || (L.cmu > 0.0 && L.alpha)) && L.n[0] % 4 == 0 && L.n[1] % 2 == 0 && L.n[2] % 2 == 0 && L.n[0] >= 128)) return false;
    for (int d = 0; d < 3; d++) if (B.lo[d] & 1) return false;
    cells += (long)L.n[0] * L.n[1] * L.n[2];
  }
  for (int d = 0; d < 3; d++) if (D0.ng[d] & 1) return false;
  // (128^3 stays interleaved: its arrays live in the caches, the conversions cost more than the passes gain -- 6.98 against 7.21 ms per step)
  static const long nmin = vdn_env("VDN_MAC_SPLIT_MIN") ? atol(vdn_env("VDN_MAC_SPLIT_MIN")) : (1L << 23);
  return cells >= nmin;
}
static int g_mac_level_form = 0;
extern "C" int vdn_last_mac_level_form(void) { return g_mac_level_form; }
static XPlan *cc_split_plan(const CDLev &D0, int colour, const int per[3]);
static void cc_split_setup(CCMG &M) {
  CDLev &D0 = M.dlev[0];
  for (CBox &B : D0.boxes) {
    const CLev &L = B.L;
    CSplit &S = B.sp;
    // rows start on a 128-byte line and are whole lines long (measured at 256^3: 8 entries in front, rows of 144: FETCH_SIZE 204.8 MB raw, 0.0895 ms per pass; 16 / 160: 189.3 MB, 0.0848 ms)
    const int off = 16, rnd = 16;
    S.off = off;
    S.PXH = ((L.n[0] / 2 + off + 2 + rnd - 1) / rnd) * rnd; S.sy = S.PXH; S.sz = (long)S.PXH * (L.n[1] + 2); S.tot = S.sz * (L.n[2] + 2);
    double *base = (double *)arena_alloc(sizeof(double) * S.tot * 6);
    for (int c = 0; c < 2; c++) { S.phi[c] = base + c * S.tot; S.rh[c] = base + (2 + c) * S.tot; S.rho[c] = base + (4 + c) * S.tot; }
  }
  D0.split = true;
  if (D0.halo) for (int c = 0; c < 2; c++) D0.shalo[c] = cc_split_plan(D0, c, M.per);
  cc_to_split(D0, 2 | 4);
}

static CLev cc_alloc_lev(const int n[3], const double h[3], bool has_alpha) {
  CLev L;
  for (int d = 0; d < 3; d++) { L.n[d] = n[d]; L.hi2[d] = 1.0 / (h[d] * h[d]); }
  L.PX = ((n[0] + 17 + 15) / 16) * 16; L.PY = n[1] + 2;
  L.sz = (long)L.PX * L.PY * (n[2] + 2);
  const int nf = has_alpha ? 7 : 6;
  double *base = (double *)arena_alloc(sizeof(double) * L.sz * nf);
  // zero fill: phi needs its ghost layer at zero (non-periodic faces); rh, res and the face coefficients are written on every entry that is ever
  // read (kk_cc_load / kk_cc_load_rh / the residual / kk_cc_coarsen_b) -- on the large levels only phi (and alpha) are cleared: a 256^3 level
  // holds 920 MB in its six arrays, 144 us of fill per solve
  const bool lean = (long)n[0] * n[1] * n[2] >= (1L << 21);
  HIPCHK(hipMemsetAsync(base, 0, sizeof(double) * L.sz * (lean ? 1 : nf), ctx().stream));
  if (lean && has_alpha) HIPCHK(hipMemsetAsync(base + 6 * L.sz, 0, sizeof(double) * L.sz, ctx().stream));
  L.phi = base; L.rh = base + L.sz; L.res = base + 2 * L.sz;
  for (int d = 0; d < 3; d++) L.b[d] = base + (3 + d) * L.sz;
  L.alpha = has_alpha ? base + 6 * L.sz : nullptr;
  L.rho = nullptr; L.cmu = 0.0;
  for (int d = 0; d < 3; d++) L.fold[d][0] = L.fold[d][1] = VDN_BC_INT;
  return L;
}
static FV cc_phi_view(const CLev &L, const int lo[3]) {
  FV f; f.p = L.phi; f.a0 = lo[0] - 16; f.a1 = lo[1] - 1; f.a2 = lo[2] - 1; f.n0 = L.PX; f.n1 = L.PY; f.n2 = L.n[2] + 2; f.sc = L.sz;
  return f;
}

// the 7-point operator reads no edge or corner ghost cell: the halo of phi carries the face cells only (VDN_CC_HALO_FACES=0: the whole shell)
static bool cc_faces_only() { static const bool f = !(vdn_env("VDN_CC_HALO_FACES") && atoi(vdn_env("VDN_CC_HALO_FACES")) == 0); return f; }
// plans for the per-level phi halos are cached across solves: arena addresses repeat from step to step
// sig: a hash of EVERY local box's phi address -- the plan bakes those addresses in, and two solves on one level that start at the same
// arena offset but differ in layout (6 arrays per box without alpha, 7 with) agree on the first box only
struct HaloKey { unsigned long uid; const void *p0; int lev, l, per; unsigned long long sig; bool operator<(const HaloKey &o) const { return std::tie(uid, p0, lev, l, per, sig) < std::tie(o.uid, o.p0, o.lev, o.l, o.per, o.sig); } };
static std::map<HaloKey, XPlan *> g_halo_cache;
void cc_halo_cache_purge(unsigned long uid) {        // the plans themselves are freed by exchange.hip (halo_cache_register)
  for (auto it = g_halo_cache.begin(); it != g_halo_cache.end();) { if (it->first.uid == uid) it = g_halo_cache.erase(it); else ++it; }
}

// the exchange of phi[colour] in the index space (ih, j, k) of the split arrays: boxes and domain halved along x.  Extents and corner indices are even (cc_split_ok), so a
// row has the same parity on both sides of a box face or a periodic image: entry ih of a ghost row / plane is entry ih of the neighbour's row / plane, the ghost entry -1 (nh)
// of a row its entry nh - 1 (0) -- cell i = -1 (i = nx) on the rows of odd (even) parity, an entry nobody reads on the others: a plain one-entry-wide face exchange, built,
// cached and run like the level arrays' (local copies, packed buffers to other ranks)
static XPlan *cc_split_plan(const CDLev &D0, int colour, const int per[3]) {
  std::vector<XBoxInfo> xb = D0.xb;
  size_t li = 0;
  GraphKey hk;
  for (XBoxInfo &x : xb) {
    const int nh = (x.vhi[0] - x.vlo[0] + 1) / 2;
    x.vlo[0] /= 2; x.vhi[0] = x.vlo[0] + nh - 1;
    if (x.owner == ctx().rank) {
      const CBox &B = D0.boxes[li++];
      const CSplit &S = B.sp;
      FV f; f.p = S.phi[colour]; f.a0 = x.vlo[0] - S.off; f.a1 = x.vlo[1] - 1; f.a2 = x.vlo[2] - 1; f.n0 = S.PXH; f.n1 = B.L.n[1] + 2; f.n2 = B.L.n[2] + 2; f.sc = S.tot;
      x.fv = f; hk.put(f.p); hk.put(S.tot);
    }
  }
  vdn_box lpd = D0.lpd; lpd.hi[0] = (lpd.hi[0] + 1) / 2 - 1;
  HaloKey key{ D0.la_uid, (const void *)D0.boxes[0].sp.phi[colour], D0.la_lev, 1000 + colour, per[0] | (per[1] << 1) | (per[2] << 2), hk.h };
  auto it = g_halo_cache.find(key);
  if (it == g_halo_cache.end()) { XPlan *P = xplan_build(xb, lpd, per, 1, 1, cc_faces_only()); halo_cache_register(D0.la_uid, P); it = g_halo_cache.emplace(key, P).first; }
  return it->second;
}

// Several boxes: a level stays distributed box by box while its boxes, halved, are at least this wide; below that the level of the WHOLE domain is gathered and every rank
// runs the rest of the hierarchy on one box.  64 where the boxes live on several ranks: every level that stays distributed costs ~10 latency-bound halo exchanges per V-cycle,
// the replicated levels below 64^3 per box cost microseconds per pass.  Round 6: 128 where every box of the level is this rank's and no transport is up (configs[2] on one GPU):
// eight 64^3 boxes are eight 5-us launches and a ghost-copy kernel per pass -- three times the ONE pass over the gathered 128^3 level -- and the gather is a device copy.
// The arithmetic is that of the single-box hierarchy wherever the cut is made (global colours, global bottom-sweep counts): tests/test_multirank_gpu.py compares ranks that
// cut at 64 with one rank that cuts at 128, bit for bit.  VDN_MG_AGGLOM (testing build): a fixed value.
int mg_agglom(const vdn_layout *la, int lev) {
  static const int env = vdn_env("VDN_MG_AGGLOM") ? atoi(vdn_env("VDN_MG_AGGLOM")) : 0;
  if (env > 0) return env;
  bool all_local = true;
  for (int o : la->owner[lev]) if (o != ctx().rank) all_local = false;
  return (all_local && !comm_active()) ? 128 : 64;
}

static void cc_build(CCMG &M, const vdn_multifab *rh, const double *dx, const int bc[3][2], bool has_alpha) {
  Prof prof_("cc_build");
  const vdn_layout *la = rh->la; const int lev = rh->lev;
  const auto &gboxes = la->boxes[lev];
  const int nb = (int)gboxes.size();
  for (int d = 0; d < 3; d++) M.per[d] = (bc[d][0] == VDN_BC_PER);
  // all boxes must share one size and be aligned to it (2x2x2-style decompositions); the general case needs
  // per-box gather sizes and is left for the AMR round
  int bn[3]; for (int d = 0; d < 3; d++) bn[d] = gboxes[0].hi[d] - gboxes[0].lo[d] + 1;
  for (const auto &b : gboxes) for (int d = 0; d < 3; d++) {
    REQUIRE(b.hi[d] - b.lo[d] + 1 == bn[d], "cc multigrid: all boxes of a level must have the same size");
    REQUIRE((b.lo[d] - la->pd[lev].lo[d]) % bn[d] == 0, "cc multigrid: boxes must be aligned to their size");
  }
  std::vector<int> nloc_of(ctx().nranks, 0);
  for (int g = 0; g < nb; g++) nloc_of[la->owner[lev][g]]++;
  int maxloc = 0; for (int r = 0; r < ctx().nranks; r++) maxloc = std::max(maxloc, nloc_of[r]);
  // ---- distributed levels: while every extent of the boxes is even and >= 4 --------------------------------
  int n[3] = { bn[0], bn[1], bn[2] }; double h[3] = { dx[0], dx[1], dx[2] };
  int scale = 1;
  for (;;) {
    CDLev DL;
    std::vector<XBoxInfo> xb;
    for (int g = 0; g < nb; g++) {
      XBoxInfo x; memset(&x, 0, sizeof x);
      int lo[3];
      for (int d = 0; d < 3; d++) { lo[d] = (gboxes[g].lo[d] - la->pd[lev].lo[d]) / scale; x.vlo[d] = lo[d]; x.vhi[d] = lo[d] + n[d] - 1; }
      x.owner = la->owner[lev][g];
      if (x.owner == ctx().rank) {
        CBox B; B.L = cc_alloc_lev(n, h, has_alpha); B.gidx = g;
        for (int d = 0; d < 3; d++) B.lo[d] = lo[d];
        x.fv = cc_phi_view(B.L, lo);
        DL.boxes.push_back(B);
      }
      xb.push_back(x);
    }
    vdn_box lpd; for (int d = 0; d < 3; d++) { lpd.lo[d] = 0; lpd.hi[d] = (la->pd[lev].hi[d] - la->pd[lev].lo[d] + 1) / scale - 1; DL.ng[d] = lpd.hi[d] + 1; }
    for (CBox &B : DL.boxes) {
      B.hmask = 0;
      for (int d = 0; d < 3; d++) {
        if (B.lo[d] > 0 || M.per[d]) B.hmask |= 1 << (2 * d);
        if (B.lo[d] + B.L.n[d] < DL.ng[d] || M.per[d]) B.hmask |= 2 << (2 * d);
      }
    }
    if (nb > 1 || M.per[0] || M.per[1] || M.per[2]) {
      GraphKey hk; for (const CBox &B : DL.boxes) { hk.put(B.L.phi); hk.put(B.L.sz); }
      HaloKey key{ la->uid, DL.boxes.empty() ? nullptr : (const void *)DL.boxes[0].L.phi, lev, (int)M.dlev.size(), M.per[0] | (M.per[1] << 1) | (M.per[2] << 2), hk.h };
      auto it = g_halo_cache.find(key);
      if (it == g_halo_cache.end()) { XPlan *P = xplan_build(xb, lpd, M.per, 1, 1, cc_faces_only()); halo_cache_register(la->uid, P); it = g_halo_cache.emplace(key, P).first; }
      DL.halo = it->second;
    }
    DL.single_box = (nb == 1);
    DL.xb = xb; DL.lpd = lpd; DL.la_uid = la->uid; DL.la_lev = lev;
    M.dlev.push_back(DL);
    // the hierarchy of GLOBAL levels is the single-box one (oracle rule: coarsen while every global extent is
    // even and > 2); a level stays distributed while the boxes halve cleanly to extents >= 4
    bool can = true, next_dist = true;
    for (int d = 0; d < 3; d++) {
      const int N = lpd.hi[d] + 1;
      if ((N & 1) || N <= 2) can = false;
    }
    if (can) for (int d = 0; d < 3; d++) REQUIRE(!(n[d] & 1), "cc multigrid: box extent %d is odd while the domain can still be coarsened", n[d]);
    // several boxes: stop exchanging halos once the boxes get small (below 64 cells) -- every level that stays distributed costs
    // ~10 latency-bound halo exchanges per V-cycle, the replicated tail below a 64^3-per-box level costs microseconds per pass
    const int agglom = mg_agglom(la, lev);
    const int min_dist = nb > 1 ? agglom : 4;
    for (int d = 0; d < 3; d++) if (n[d] / 2 < min_dist || ((n[d] / 2) & 1)) next_dist = false;
    if (!can) break;                                   // the domain cannot be coarsened: this level is the bottom
    if (!next_dist) {
      // ---- agglomerated tail starting at the half of this level ---------------------------------------------
      int tn[3]; double th[3]; int cn[3];
      for (int d = 0; d < 3; d++) { cn[d] = n[d] / 2; tn[d] = (lpd.hi[d] + 1) / 2; th[d] = h[d] * 2.0; }
      for (;;) {
        M.tail.push_back(cc_alloc_lev(tn, th, has_alpha));
        bool c2 = true;
        for (int d = 0; d < 3; d++) if ((tn[d] & 1) || tn[d] <= 2) c2 = false;
        if (!c2 || M.tail.size() >= 31) break;
        for (int d = 0; d < 3; d++) { tn[d] /= 2; th[d] *= 2.0; }
      }
      // gather descriptors: rank r's buffer holds its boxes in local order, padded to maxloc boxes
      const long per_rh = (long)cn[0] * cn[1] * cn[2];
      const long per_b = (long)(cn[0] + 1) * cn[1] * cn[2] + (long)cn[0] * (cn[1] + 1) * cn[2] + (long)cn[0] * cn[1] * (cn[2] + 1);
      M.cnt_rh = (size_t)per_rh * maxloc; M.cnt_b = (size_t)per_b * maxloc;
      std::vector<int> seen(ctx().nranks, 0);
      for (int g = 0; g < nb; g++) {
        const int r = la->owner[lev][g], l = seen[r]++;
        GBox a, b2;
        for (int d = 0; d < 3; d++) { a.c0[d] = b2.c0[d] = (gboxes[g].lo[d] - la->pd[lev].lo[d]) / scale / 2; a.n[d] = b2.n[d] = cn[d]; }
        a.off = (long)r * M.cnt_rh + (long)l * per_rh; b2.off = (long)r * M.cnt_b + (long)l * per_b;
        M.gb_rh.push_back(a); M.gb_b.push_back(b2);
        if (r == ctx().rank) { M.loc_off_rh.push_back((long)l * per_rh); M.loc_off_b.push_back((long)l * per_b); }
      }
      M.d_gb_rh = (GBox *)arena_alloc(nb * sizeof(GBox)); M.d_gb_b = (GBox *)arena_alloc(nb * sizeof(GBox));
      HIPCHK(hipMemcpyAsync(M.d_gb_rh, M.gb_rh.data(), nb * sizeof(GBox), hipMemcpyHostToDevice, ctx().stream));
      HIPCHK(hipMemcpyAsync(M.d_gb_b, M.gb_b.data(), nb * sizeof(GBox), hipMemcpyHostToDevice, ctx().stream));
      HIPCHK(hipStreamSynchronize(ctx().stream));     // the host vectors above are read by the copies
      M.sendbuf = (double *)arena_alloc(sizeof(double) * M.cnt_b);
      M.recvbuf = (double *)arena_alloc(sizeof(double) * M.cnt_b * ctx().nranks);
      break;
    }
    for (int d = 0; d < 3; d++) { n[d] /= 2; h[d] *= 2.0; }
    scale *= 2;
    if (M.dlev.size() >= 31) break;
  }
  M.d_nrm = (double *)arena_alloc(256);
}

static void cc_halo(CCMG &M, CDLev &DL) { if (DL.halo) xplan_run(DL.halo); }
// levels of at most 8^3 cells held in ONE box are smoothed by a single workgroup in one launch (all sweeps, both
// colours, periodic images included): such levels are launch-latency bound, not bandwidth bound
static const long SMALL_LEVEL_CELLS = 8L * 8 * 8;
static void cc_gsrb_d(CCMG &M, CDLev &DL, int nsweeps) {
  if (DL.single_box && DL.boxes.size() == 1 && (long)DL.ng[0] * DL.ng[1] * DL.ng[2] <= SMALL_LEVEL_CELLS) {
    hipLaunchKernelGGL(kk_cc_bottom, dim3(1), dim3(1024), 0, ctx().stream, DL.boxes[0].L, nsweeps, M.per[0], M.per[1], M.per[2]);
    return;
  }
  // halo exchange next to the pass: when part of the halo comes from another rank (or VDN_OVERLAP=1, the one-GPU rehearsal) the packed
  // traffic -- pack kernels, the ncclSend / ncclRecv group, box-to-box copies, unpack kernels -- runs on ctx().halo_stream while the
  // launch stream updates the cells that read no ghost value; the one-cell shell follows when the halo has landed
  // Only where the pass is long enough to hide something: boxes of at least 2^20 cells (a 64^3 pass takes 5 us).
  static const int ov_env = vdn_env("VDN_OVERLAP") ? atoi(vdn_env("VDN_OVERLAP")) : -1;
  static const long ov_min = 1L << 20;
  bool overlap = DL.halo && (ov_env == 1 || (ov_env != 0 && xplan_has_remote(DL.halo)));
  if (overlap) {
    long cells = 0;
    for (const CBox &B : DL.boxes) cells = std::max(cells, (long)B.L.n[0] * B.L.n[1] * B.L.n[2]);
    if (cells < ov_min && ov_env != 1) overlap = false;
  }
  VdnCtx &c = ctx();
  if (DL.split) {          // by colour: before a pass the OTHER colour's ghost entries are exchanged (half the volume of the level array's exchange)
    for (int s = 0; s < nsweeps; s++) for (int color = 0; color < 2; color++) {
      if (!overlap) {
        split_halo(DL, 1 - color);
        for (const CBox &B : DL.boxes) launch_gsrb_split<0>(B, color, c.stream, B.L);
        continue;
      }
      HIPCHK(hipEventRecord(c.ev_main, c.stream));
      HIPCHK(hipStreamWaitEvent(c.halo_stream, c.ev_main, 0));
      split_halo(DL, 1 - color, c.halo_stream);
      HIPCHK(hipEventRecord(c.ev_halo, c.halo_stream));
      for (const CBox &B : DL.boxes) launch_gsrb_split<0>(B, color, c.stream, B.L, 0, -1, B.hmask);
      HIPCHK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
      for (const CBox &B : DL.boxes) launch_gsrb_split_shell(B, color, c.stream, B.hmask);
    }
    return;
  }
  for (int s = 0; s < nsweeps; s++) for (int color = 0; color < 2; color++) {
    if (!overlap) {
      cc_halo(M, DL);
      for (const CBox &B : DL.boxes) launch_gsrb(B.L, (color + B.lo[0] + B.lo[1] + B.lo[2]) & 1, c.stream);       // colour by GLOBAL cell index
      continue;
    }
    HIPCHK(hipEventRecord(c.ev_main, c.stream));                    // phi of the previous pass is complete ...
    HIPCHK(hipStreamWaitEvent(c.halo_stream, c.ev_main, 0));        // ... before its outermost layer is packed
    xplan_run(DL.halo, c.halo_stream);
    HIPCHK(hipEventRecord(c.ev_halo, c.halo_stream));
    for (const CBox &B : DL.boxes) launch_gsrb(B.L, (color + B.lo[0] + B.lo[1] + B.lo[2]) & 1, c.stream, B.hmask);
    HIPCHK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
    for (const CBox &B : DL.boxes) launch_gsrb_shell(B.L, (color + B.lo[0] + B.lo[1] + B.lo[2]) & 1, c.stream, B.hmask);
  }
}
static void cc_residual_d(CCMG &M, CDLev &DL, bool norm, bool reduce = true) {       // reduce = false: the norm stays rank-local (norm history, mg_predict)
  if (norm) HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), ctx().stream));
  DL.res_restricted = false;
  if (DL.split) {
    static const bool split_res = !(vdn_env("VDN_MAC_SPLIT") && atoi(vdn_env("VDN_MAC_SPLIT")) == 2);      // 2: only the colour passes run on the split arrays
    if (split_res) {
      // the residual reads both colours' ghost entries: colour 0's were exchanged before the last pass (of colour 1) and have not changed since
      split_halo(DL, 1);
      for (size_t b = 0; b < DL.boxes.size(); b++) {
        const CBox &B = DL.boxes[b]; const CLev &L = B.L;
        launch_residual_split_rst(L, B.sp, norm ? M.d_nrm : nullptr, M.dlev[1].boxes[b].L, 0, L.n[2] / 2, ctx().stream);
      }
      DL.res_restricted = true;
      if (norm && reduce) comm_allreduce_max_dev(M.d_nrm, 1);
      return;
    }
    cc_from_split(DL);
  }
  cc_halo(M, DL);
  {   // the finest level of a MAC solve in one box: residual and restriction in one pass (kk_cc_residual_rho_pair_rst); cc_restrict_down then skips
    static const bool fuse = !(vdn_env("VDN_MG_RESTRICT_FUSED") && atoi(vdn_env("VDN_MG_RESTRICT_FUSED")) == 0);
    static const bool paired0 = !(vdn_env("VDN_GSRB_PAIR") && atoi(vdn_env("VDN_GSRB_PAIR")) == 0);
    const size_t l = &DL - &M.dlev[0];
    if (fuse && paired0 && DL.single_box && DL.boxes.size() == 1 && l + 1 < M.dlev.size() && M.dlev[l + 1].boxes.size() == 1) {
      const CLev &L = DL.boxes[0].L;
      if (L.rho && L.n[0] % 2 == 0 && L.n[1] % 2 == 0 && L.n[2] % 2 == 0 && L.n[0] >= 128) {
        const dim3 g((unsigned)((L.n[0] / 2 + 63) / 64), (unsigned)((L.n[1] / 2 + 3) / 4), (unsigned)std::min(L.n[2] / 2, 16));
        hipLaunchKernelGGL(kk_cc_residual_rho_pair_rst, g, BLK, 0, ctx().stream, L, norm ? M.d_nrm : nullptr, M.dlev[l + 1].boxes[0].L);
        DL.res_restricted = true;
        if (norm && reduce) comm_allreduce_max_dev(M.d_nrm, 1);
        return;
      }
    }
  }
  for (const CBox &B : DL.boxes) {
    const dim3 g = g3(B.L.n[0], B.L.n[1], norm ? std::min(B.L.n[2], 16) : B.L.n[2], BLK);
    static const bool paired = !(vdn_env("VDN_GSRB_PAIR") && atoi(vdn_env("VDN_GSRB_PAIR")) == 0);
    if (B.L.rho && paired && B.L.n[0] % 2 == 0 && B.L.n[1] % 2 == 0 && B.L.n[0] >= 128)
      hipLaunchKernelGGL(kk_cc_residual_rho_pair, dim3((unsigned)((B.L.n[0] / 2 + 63) / 64), (unsigned)((B.L.n[1] / 2 + 3) / 4), g.z), BLK, 0, ctx().stream, B.L, norm ? M.d_nrm : nullptr);
    else if (B.L.rho) hipLaunchKernelGGL(kk_cc_residual_rho, g, BLK, 0, ctx().stream, B.L, norm ? M.d_nrm : nullptr);
    else hipLaunchKernelGGL(kk_cc_residual, g, BLK, 0, ctx().stream, B.L, norm ? M.d_nrm : nullptr);
  }
  if (norm && reduce) comm_allreduce_max_dev(M.d_nrm, 1);
}
static double read_scalar(double *d) {
  return read_scalar1(d);
}

// ---- the replicated tail: single-box V-cycle --------------------------------------------------------------------
static void cc_periodic_t(const CCMG &M, const CLev &L) {
  if (!(M.per[0] || M.per[1] || M.per[2])) return;
  int m = std::max(L.n[0], std::max(L.n[1], L.n[2]));
  hipLaunchKernelGGL(kk_cc_periodic, g3(m, m, 6, BLK), BLK, 0, ctx().stream, L, M.per[0], M.per[1], M.per[2]);
}
static void cc_gsrb_t(const CCMG &M, const CLev &L, int nsweeps) {
  if ((long)L.n[0] * L.n[1] * L.n[2] <= SMALL_LEVEL_CELLS) {
    hipLaunchKernelGGL(kk_cc_bottom, dim3(1), dim3(1024), 0, ctx().stream, L, nsweeps, M.per[0], M.per[1], M.per[2]);
    return;
  }
  for (int s = 0; s < nsweeps; s++) for (int color = 0; color < 2; color++) {
    cc_periodic_t(M, L);
    launch_gsrb(L, color, ctx().stream);
  }
}
static void cc_bottom_t(const CCMG &M, const CLev &L) {      // max(nub, N^2) sweeps, N = largest extent (same rule as the oracle)
  const int N = std::max(L.n[0], std::max(L.n[1], L.n[2]));
  const int ns = std::max(ctx().prm.mg_nub, N * N);
  hipLaunchKernelGGL(kk_cc_bottom, dim3(1), dim3(1024), 0, ctx().stream, L, ns, M.per[0], M.per[1], M.per[2]);
}
// The small end of the hierarchy in one launch (kk_cc_tailcycle): distributed levels dl .. end when they are one box of at most 8^3 cells
// each (dl < 0: none), then the replicated tail levels tl .. end (one rank and one box: the gather between the two is the plain restriction).
static bool cc_small_end(const CCMG &M, int dl, int tl) {
  static const bool on = !(vdn_env("VDN_MG_TAILCYCLE") && atoi(vdn_env("VDN_MG_TAILCYCLE")) == 0);
  if (!on) return false;
  static const long tail_cells = SMALL_LEVEL_CELLS;     // largest level the one-workgroup cycle takes (measured: 16^3 no gain, MAC 15.33 -> 15.39 ms)
  const vdn_params &P = ctx().prm;
  CcTailArgs T; memset(&T, 0, sizeof T);
  int nl = 0;
  if (dl >= 0) {
    if (!M.tail.empty() && !(ctx().nranks == 1 && M.dlev.back().single_box)) return false;
    for (int m = dl; m < (int)M.dlev.size(); m++) {
      const CDLev &D = M.dlev[m];
      if (!(D.single_box && D.boxes.size() == 1 && !D.boxes[0].L.rho && (long)D.ng[0] * D.ng[1] * D.ng[2] <= tail_cells) || nl == CC_TAIL_MAX) return false;
      T.L[nl++] = D.boxes[0].L;
    }
  }
  for (int m = tl; m < (int)M.tail.size(); m++) {
    const CLev &L = M.tail[m];
    if ((long)L.n[0] * L.n[1] * L.n[2] > tail_cells || nl == CC_TAIL_MAX) return false;
    T.L[nl++] = L;
  }
  if (nl < 2) return false;
  const CLev &B = T.L[nl - 1];
  const int N = std::max(B.n[0], std::max(B.n[1], B.n[2]));
  T.nlev = nl; T.nu1 = P.mg_nu1; T.nu2 = P.mg_nu2; T.nbot = std::max(P.mg_nub, N * N);      // cc_bottom_t / cc_vcycle_d
  for (int d = 0; d < 3; d++) T.per[d] = M.per[d];
  hipLaunchKernelGGL(kk_cc_tailcycle, dim3(1), dim3(1024), 0, ctx().stream, T);
  return true;
}
// may tail level l run as kk_cc_lds_down / kk_cc_lds_up?  (round 6: the replicated levels of 16^3 .. 64^3 cells take the two-launch form of the one-box hierarchy too --
// they were eleven launches per level and cycle; cc_lds_level's conditions)
static bool cc_lds_tail_level(const CCMG &M, int l) {
  static const bool on = !(vdn_env("VDN_MG_LDS") && atoi(vdn_env("VDN_MG_LDS")) == 0);
  const vdn_params &P = ctx().prm;
  if (!on || l + 1 >= (int)M.tail.size() || P.mg_nu1 != 2 || P.mg_nu2 != 2 || M.per[0] || M.per[1] || M.per[2]) return false;
  const CLev &L = M.tail[l], &C = M.tail[l + 1];
  if (L.rho) return false;
  for (int d = 0; d < 3; d++) if (L.n[d] % LT || L.n[d] < 2 * LT || L.n[d] > 64 || C.n[d] * 2 != L.n[d]) return false;
  return true;
}
static void cc_vcycle_t(const CCMG &M, int l) {
  const vdn_params &P = ctx().prm;
  if (cc_small_end(M, -1, l)) return;
  const CLev &L = M.tail[l];                 // phi = 0 on entry: written by the restriction that feeds this level
  if (l == (int)M.tail.size() - 1) { cc_bottom_t(M, L); return; }
  const CLev &C = M.tail[l + 1];
  if (cc_lds_tail_level(M, l)) {
    const dim3 g((unsigned)(L.n[0] / LT), (unsigned)(L.n[1] / LT), (unsigned)(L.n[2] / LT));
    if (L.alpha) hipLaunchKernelGGL(kk_cc_lds_down<true>, g, dim3(1024), 0, ctx().stream, L, C);
    else hipLaunchKernelGGL(kk_cc_lds_down<false>, g, dim3(1024), 0, ctx().stream, L, C);
    cc_vcycle_t(M, l + 1);
    if (L.alpha) hipLaunchKernelGGL(kk_cc_lds_up<true>, g, dim3(1024), 0, ctx().stream, L, C);
    else hipLaunchKernelGGL(kk_cc_lds_up<false>, g, dim3(1024), 0, ctx().stream, L, C);
    return;
  }
  cc_gsrb_t(M, L, P.mg_nu1);
  cc_periodic_t(M, L);
  hipLaunchKernelGGL(kk_cc_residual, g3(L.n[0], L.n[1], L.n[2], BLK), BLK, 0, ctx().stream, L, (double *)nullptr);
  hipLaunchKernelGGL(kk_cc_restrict, g3(C.n[0], C.n[1], C.n[2], BLK), BLK, 0, ctx().stream, L, C);
  cc_vcycle_t(M, l + 1);
  hipLaunchKernelGGL(kk_cc_prolong, g3(L.n[0], L.n[1], L.n[2], BLK), BLK, 0, ctx().stream, L, C);
  cc_gsrb_t(M, L, P.mg_nu2);
}

// restrict the residual of distributed level l into level l+1 (distributed) or into the tail (gather)
static void cc_restrict_down(CCMG &M, int l) {
  CDLev &DL = M.dlev[l];
  if (DL.res_restricted) { DL.res_restricted = false; return; }      // done inside the residual pass
  if (l + 1 < (int)M.dlev.size()) {
    CDLev &DC = M.dlev[l + 1];
    for (size_t b = 0; b < DL.boxes.size(); b++) {
      const CLev &C = DC.boxes[b].L;
      hipLaunchKernelGGL(kk_cc_restrict, g3(C.n[0], C.n[1], C.n[2], BLK), BLK, 0, ctx().stream, DL.boxes[b].L, C);
    }
  } else {
    const CLev &T = M.tail[0];
    for (size_t b = 0; b < DL.boxes.size(); b++) {
      const CLev &F = DL.boxes[b].L;
      const int nx = F.n[0] / 2, ny = F.n[1] / 2, nz = F.n[2] / 2;
      hipLaunchKernelGGL(kk_cc_restrict_pack, g3(nx, ny, nz, BLK), BLK, 0, ctx().stream, F, (const double *)F.res, M.sendbuf, M.loc_off_rh[b], nx, ny, nz);
    }
    comm_allgather_dev(M.sendbuf, M.recvbuf, M.cnt_rh);
    hipLaunchKernelGGL(kk_cc_unpack_rh, dim3(4, 1, (unsigned)M.gb_rh.size()), dim3(256), 0, ctx().stream, T, T.rh, M.recvbuf, M.d_gb_rh);
  }
}
static void cc_prolong_up(CCMG &M, int l) {
  CDLev &DL = M.dlev[l];
  for (size_t b = 0; b < DL.boxes.size(); b++) {
    const CBox &B = DL.boxes[b];
    if (l + 1 < (int)M.dlev.size())
      hipLaunchKernelGGL(kk_cc_prolong, g3(B.L.n[0], B.L.n[1], B.L.n[2], BLK), BLK, 0, ctx().stream, B.L, M.dlev[l + 1].boxes[b].L);
    else
      hipLaunchKernelGGL(kk_cc_prolong_tail, g3(B.L.n[0], B.L.n[1], B.L.n[2], BLK), BLK, 0, ctx().stream, B.L, M.tail[0], B.lo[0] / 2, B.lo[1] / 2, B.lo[2] / 2);
  }
}
// prolongation of level l+1 into level l followed by nsweeps of smoothing.  On the finest level of a MAC solve in one box without periodic
// faces (the paired density pass) the correction is added inside the first sweep (kk_cc_gsrb_rho_pair_t); otherwise kk_cc_prolong first.
static void cc_prolong_smooth(CCMG &M, int l, int nsweeps) {
  CDLev &DL = M.dlev[l];
  if (DL.split) {
    // the correction rides in the first sweep.  With a halo: the ghost cells of the first colour's neighbours must become phi + e(parent) like the cells they mirror --
    // phi's ghost entries are current (exchanged before the residual), e's come from one exchange of the coarse level; the sum is the neighbour's own phi + e
    if (DL.halo) cc_halo(M, M.dlev[l + 1]);
    for (size_t b = 0; b < DL.boxes.size(); b++) launch_gsrb_split<1>(DL.boxes[b], 0, ctx().stream, M.dlev[l + 1].boxes[b].L);
    split_halo(DL, 0);
    for (size_t b = 0; b < DL.boxes.size(); b++) launch_gsrb_split<2>(DL.boxes[b], 1, ctx().stream, M.dlev[l + 1].boxes[b].L);
    if (nsweeps > 1) cc_gsrb_d(M, DL, nsweeps - 1);
    return;
  }
  static const bool fuse = !(vdn_env("VDN_MG_PROLONG_FUSED") && atoi(vdn_env("VDN_MG_PROLONG_FUSED")) == 0);
  static const bool paired = !(vdn_env("VDN_GSRB_PAIR") && atoi(vdn_env("VDN_GSRB_PAIR")) == 0);
  const bool ok = fuse && paired && nsweeps >= 1 && DL.single_box && DL.boxes.size() == 1 && !DL.halo && l + 1 < (int)M.dlev.size() && M.dlev[l + 1].boxes.size() == 1 &&
                  !(M.per[0] || M.per[1] || M.per[2]) && DL.boxes[0].L.rho &&
                  DL.boxes[0].L.n[0] % 2 == 0 && DL.boxes[0].L.n[1] % 2 == 0 && DL.boxes[0].L.n[2] % 2 == 0 && DL.boxes[0].L.n[0] >= 128;
  if (!ok) { cc_prolong_up(M, l); cc_gsrb_d(M, DL, nsweeps); return; }
  const CLev &L = DL.boxes[0].L, &C = M.dlev[l + 1].boxes[0].L;
  const dim3 g((unsigned)((L.n[0] / 2 + 63) / 64), (unsigned)((L.n[1] / 2 + 3) / 4), (unsigned)L.n[2]), blk(64, 4, 1);
  hipLaunchKernelGGL(kk_cc_gsrb_rho_pair_t<1>, g, blk, 0, ctx().stream, L, 0, 0, C, 0);
  hipLaunchKernelGGL(kk_cc_gsrb_rho_pair_t<2>, g, blk, 0, ctx().stream, L, 1, 0, C, mac_kflip() ? 1 : 0);
  if (nsweeps > 1) cc_gsrb_d(M, DL, nsweeps - 1);
}
// The finest level's part of a cycle -- [prolongation inside the first sweep,] nsweeps red-black sweeps, residual + restriction -- on the SPLIT level in plane
// slabs (time skewing): pass p runs on the planes pass p - 1 has left behind by at least two (plane k needs pass p - 1 done on k + 1, and must not touch
// plane k' - 1 before pass p - 1 has read it for k'), so a slab of VDN_MAC_SLAB planes goes through ALL passes while it sits in the 256 MB Infinity
// Cache: nine launches over the whole level read each of its six arrays nine times from HBM, the slabs read them about once.  Cells of a colour do not
// read each other and the residual only reads: the order changes no bit.  VDN_MAC_SLAB=0: whole-level launches.
// (The residual also WRITES the next level -- its right-hand side and a zero phi -- while the first two passes of a later slab still READ that phi, the correction
// they add: coarse plane K is zeroed once the fine planes up to 2K + 2 have seen the LAST pass, R - 1 >= 3 planes behind the first; the first pass on plane k
// reads coarse planes (k >> 1) - 1 .. (k >> 1) + 1 with k at least R - 1 planes ahead of that, i.e. coarse planes the residual has not reached.)
// Measured at 256^3 (MAC solve per step): whole-level launches 10.03 ms; slabs of 32 / 48 / 64 / 96 / 128 planes 10.66 / 10.28 / 9.86 / 9.65 / 9.44 ms -- a pass
// served from the cache takes 0.069 ms per 256 planes against 0.090 from HBM, and every launch costs its ramp and tail.  The default:
// the planes whose pass traffic (24 B per cell of the level) adds up to ~200 MB -- what stays in the cache between two passes over it; at most half the level.
// 512^3 (plane = 6.3 MB; MAC solve per step): whole-level launches 69.8 ms; slabs of 16 / 24 / 28 / 32 / 40 / 64 / 256 planes 67.9 / 65.2 / 64.5 / 64.7 / 65.2 / 67.5 / 69.1 ms.
static int mac_slab(const CLev &L) {
  static const int k = vdn_env("VDN_MAC_SLAB") ? atoi(vdn_env("VDN_MAC_SLAB")) : -1;
  if (k >= 0) return k;
  const long fit = (long)(200.0e6 / (24.0 * L.n[0] * L.n[1]));
  return (int)std::max(8L, std::min((long)(L.n[2] + 1) / 2, fit));
}
static void cc_split_run(CCMG &M, CDLev &DL, bool prolong, int nsweeps, bool residual, bool norm, bool reduce) {
  const CBox &B0 = DL.boxes[0];
  const CLev &L = B0.L, &C = M.dlev[1].boxes[0].L;
  hipStream_t st = ctx().stream;
  const int R = 2 * nsweeps, n2 = L.n[2], nK = n2 / 2, slab = mac_slab(L);
  if (residual && norm) HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), st));
  DL.res_restricted = false;
  std::vector<int> done(R + 1, 0);                  // planes [0, done[p]) have seen pass p; done[R]: coarse planes of the residual
  for (int top = slab + R; ; top += slab) {         // (the first slab longer by the skew: the last one is then not a sliver)
    for (int p = 0; p < R; p++) {
      const int hi = p == 0 ? std::min(top, n2) : (done[p - 1] == n2 ? n2 : done[p - 1] - 1);
      if (hi <= done[p]) continue;
      const int add = prolong ? (p == 0 ? 1 : p == 1 ? 2 : 0) : 0;
      if (add == 1) launch_gsrb_split<1>(B0, p & 1, st, C, done[p], hi);
      else if (add == 2) launch_gsrb_split<2>(B0, p & 1, st, C, done[p], hi);
      else launch_gsrb_split<0>(B0, p & 1, st, L, done[p], hi);
      done[p] = hi;
    }
    if (residual) {                                  // coarse plane K reads the fine planes 2K - 1 .. 2K + 2
      const int hiK = done[R - 1] == n2 ? nK : std::max(0, (done[R - 1] - 1) / 2);
      if (hiK > done[R]) {
        launch_residual_split_rst(L, B0.sp, norm ? M.d_nrm : nullptr, C, done[R], hiK, st);
        done[R] = hiK;
      }
    }
    if (done[R - 1] == n2 && (!residual || done[R] == nK)) break;
  }
  if (residual) { DL.res_restricted = true; if (norm && reduce) comm_allreduce_max_dev(M.d_nrm, 1); }
}
// the finest level between two coarse corrections: post-smoothing (after_coarse: with the prolongation), the next cycle's pre-smoothing, residual
static void cc_fine_seq(CCMG &M, bool after_coarse, bool residual, bool norm, bool reduce = true) {
  const vdn_params &P = ctx().prm;
  CDLev &D0 = M.dlev[0];
  static const bool split_res = !(vdn_env("VDN_MAC_SPLIT") && atoi(vdn_env("VDN_MAC_SPLIT")) == 2);
  // (the slab schedule: one box without an exchange between the passes)
  if (D0.split && !D0.halo && D0.boxes.size() == 1 && mac_slab(D0.boxes[0].L) > 0 && split_res && residual) { cc_split_run(M, D0, after_coarse, (after_coarse ? P.mg_nu2 : 0) + P.mg_nu1, true, norm, reduce); return; }
  if (after_coarse) cc_prolong_smooth(M, 0, P.mg_nu2);
  cc_gsrb_d(M, D0, P.mg_nu1);
  if (residual) cc_residual_d(M, D0, norm, reduce);
}
// may level l >= 1 of a V-cycle run as kk_cc_lds_down / kk_cc_lds_up?  (VDN_MG_LDS=0: never; extents up to 64)
static bool cc_lds_level(const CCMG &M, int l) {
  static const bool on = !(vdn_env("VDN_MG_LDS") && atoi(vdn_env("VDN_MG_LDS")) == 0);
  static const int nmax_ = 64;                 // (measured in round 5: the 128^3 level of a 256^3 solve as LDS tiles too, MAC 8.98 -> 10.33 ms per step)
  const vdn_params &P = ctx().prm;
  if (!on || l < 1 || l + 1 >= (int)M.dlev.size() || P.mg_nu1 != 2 || P.mg_nu2 != 2 || M.per[0] || M.per[1] || M.per[2]) return false;
  const CDLev &D = M.dlev[l], &DC = M.dlev[l + 1];
  if (!(D.single_box && D.boxes.size() == 1 && !D.halo && DC.single_box && DC.boxes.size() == 1)) return false;
  const CLev &L = D.boxes[0].L;
  if (L.rho) return false;
  for (int d = 0; d < 3; d++) if (L.n[d] % LT || L.n[d] < 2 * LT || L.n[d] > nmax_ || D.boxes[0].lo[d] != 0 || DC.boxes[0].L.n[d] * 2 != L.n[d]) return false;
  return true;
}
// error-equation V-cycle on distributed level l (zero initial guess)
static void cc_vcycle_d(CCMG &M, int l) {
  const vdn_params &P = ctx().prm;
  CDLev &DL = M.dlev[l];
  const bool last = (l == (int)M.dlev.size() - 1);      // phi = 0 on entry: written by the restriction that feeds this level
  if (cc_small_end(M, l, 0)) return;
  if (last && M.tail.empty()) {         // nothing below: bottom sweeps on the distributed level itself
    const int N = std::max(DL.ng[0], std::max(DL.ng[1], DL.ng[2]));     // largest GLOBAL extent, as in the oracle
    cc_gsrb_d(M, DL, std::max(P.mg_nub, N * N));
    return;
  }
  if (!last && cc_lds_level(M, l)) {        // 16^3 .. 64^3: smoothing + residual + restriction in one launch, prolongation + smoothing in another
    const CLev &L = DL.boxes[0].L, &C = M.dlev[l + 1].boxes[0].L;
    const dim3 g((unsigned)(L.n[0] / LT), (unsigned)(L.n[1] / LT), (unsigned)(L.n[2] / LT));
    if (L.alpha) hipLaunchKernelGGL(kk_cc_lds_down<true>, g, dim3(1024), 0, ctx().stream, L, C);
    else hipLaunchKernelGGL(kk_cc_lds_down<false>, g, dim3(1024), 0, ctx().stream, L, C);
    cc_vcycle_d(M, l + 1);
    if (L.alpha) hipLaunchKernelGGL(kk_cc_lds_up<true>, g, dim3(1024), 0, ctx().stream, L, C);
    else hipLaunchKernelGGL(kk_cc_lds_up<false>, g, dim3(1024), 0, ctx().stream, L, C);
    return;
  }
  cc_gsrb_d(M, DL, P.mg_nu1);
  cc_residual_d(M, DL, false);
  cc_restrict_down(M, l);
  if (last) cc_vcycle_t(M, 0); else cc_vcycle_d(M, l + 1);
  cc_prolong_smooth(M, l, P.mg_nu2);
}

// ---- nested iteration for the initial guess (vdn_params.mac_fmg; oracle: cc_fmg in vo_macproject.c) ------------------------------------------
// Levels are numbered globally: g < nd distributed, then the tail.
static long cc_level_cells(const CCMG &M, int g) {
  const int nd = (int)M.dlev.size();
  if (g < nd) return (long)M.dlev[g].ng[0] * M.dlev[g].ng[1] * M.dlev[g].ng[2];
  const CLev &T = M.tail[g - nd];
  return (long)T.n[0] * T.n[1] * T.n[2];
}
// one V-cycle on the phi a level holds (phi is NOT zeroed); level g has a level below it
static void cc_cycle_at(CCMG &M, int g) {
  const vdn_params &P = ctx().prm;
  const int nd = (int)M.dlev.size();
  if (g < nd) {
    CDLev &DL = M.dlev[g];
    cc_gsrb_d(M, DL, P.mg_nu1);
    cc_residual_d(M, DL, false);
    cc_restrict_down(M, g);
    if (g + 1 < nd) cc_vcycle_d(M, g + 1); else cc_vcycle_t(M, 0);
    cc_prolong_up(M, g); cc_gsrb_d(M, DL, P.mg_nu2);
    return;
  }
  const CLev &L = M.tail[g - nd], &C = M.tail[g - nd + 1];
  cc_gsrb_t(M, L, P.mg_nu1);
  cc_periodic_t(M, L);
  hipLaunchKernelGGL(kk_cc_residual, g3(L.n[0], L.n[1], L.n[2], BLK), BLK, 0, ctx().stream, L, (double *)nullptr);
  hipLaunchKernelGGL(kk_cc_restrict, g3(C.n[0], C.n[1], C.n[2], BLK), BLK, 0, ctx().stream, L, C);
  cc_vcycle_t(M, g - nd + 1);
  hipLaunchKernelGGL(kk_cc_prolong, g3(L.n[0], L.n[1], L.n[2], BLK), BLK, 0, ctx().stream, L, C);
  cc_gsrb_t(M, L, P.mg_nu2);
}
static void cc_fmg(CCMG &M, const int bc[3][2]) {
  const int nd = (int)M.dlev.size(), ntot = nd + (int)M.tail.size();
  int ls = -1;
  for (int g = 1; g < ntot; g++) if (cc_level_cells(M, g) >= 4096) ls = g;
  if (ls < 1 || ls + 1 >= ntot) return;                  // (a starting level with nothing below it: no nested iteration)
  hipStream_t st = ctx().stream;
  for (int g = 0; g < ls; g++) {                          // rh_{g+1} = mean of the children of rh_g; phi_{g+1} = 0 (written by the same kernels)
    if (g + 1 < nd) {
      for (size_t b = 0; b < M.dlev[g].boxes.size(); b++) {
        CLev F = M.dlev[g].boxes[b].L; F.res = F.rh;      // (kk_cc_restrict reads F.res)
        const CLev &C = M.dlev[g + 1].boxes[b].L;
        hipLaunchKernelGGL(kk_cc_restrict, g3(C.n[0], C.n[1], C.n[2], BLK), BLK, 0, st, F, C);
      }
    } else if (g + 1 == nd) {
      const CDLev &DL = M.dlev[g]; const CLev &T = M.tail[0];
      for (size_t b = 0; b < DL.boxes.size(); b++) {
        const CLev &F = DL.boxes[b].L;
        const int nx = F.n[0] / 2, ny = F.n[1] / 2, nz = F.n[2] / 2;
        hipLaunchKernelGGL(kk_cc_restrict_pack, g3(nx, ny, nz, BLK), BLK, 0, st, F, (const double *)F.rh, M.sendbuf, M.loc_off_rh[b], nx, ny, nz);
      }
      comm_allgather_dev(M.sendbuf, M.recvbuf, M.cnt_rh);
      hipLaunchKernelGGL(kk_cc_unpack_rh, dim3(4, 1, (unsigned)M.gb_rh.size()), dim3(256), 0, st, T, T.rh, M.recvbuf, M.d_gb_rh);
    } else {
      CLev F = M.tail[g - nd]; F.res = F.rh;
      const CLev &C = M.tail[g - nd + 1];
      hipLaunchKernelGGL(kk_cc_restrict, g3(C.n[0], C.n[1], C.n[2], BLK), BLK, 0, st, F, C);
    }
  }
  if (ls < nd) cc_vcycle_d(M, ls); else cc_vcycle_t(M, ls - nd);      // from zero
  cc_cycle_at(M, ls);
  for (int g = ls - 1; g >= 0; g--) {
    // phi_g = linear interpolation of phi_{g+1}: the coarse level's ghost cells behind box-box and periodic faces first
    if (g + 1 < nd) {
      CDLev &DC = M.dlev[g + 1];
      cc_halo(M, DC);
      for (size_t b = 0; b < M.dlev[g].boxes.size(); b++) {
        const CLev &F = M.dlev[g].boxes[b].L; const CBox &CB = DC.boxes[b];
        ProlongLinArgs A;
        for (int d = 0; d < 3; d++) { A.c0[d] = 0; A.e[d][0] = (CB.lo[d] == 0) ? bc[d][0] : VDN_BC_INT; A.e[d][1] = (CB.lo[d] + CB.L.n[d] == DC.ng[d]) ? bc[d][1] : VDN_BC_INT; }
        hipLaunchKernelGGL(kk_cc_prolong_lin, g3(F.n[0], F.n[1], F.n[2], BLK), BLK, 0, st, F, CB.L, A);
      }
    } else {
      const CLev &T = M.tail[g + 1 - nd];
      cc_periodic_t(M, T);
      ProlongLinArgs A;
      for (int d = 0; d < 3; d++) { A.c0[d] = 0; A.e[d][0] = bc[d][0]; A.e[d][1] = bc[d][1]; }
      if (g >= nd) { const CLev &F = M.tail[g - nd]; hipLaunchKernelGGL(kk_cc_prolong_lin, g3(F.n[0], F.n[1], F.n[2], BLK), BLK, 0, st, F, T, A); }
      else for (const CBox &B : M.dlev[g].boxes) {
        for (int d = 0; d < 3; d++) A.c0[d] = B.lo[d] / 2;
        hipLaunchKernelGGL(kk_cc_prolong_lin, g3(B.L.n[0], B.L.n[1], B.L.n[2], BLK), BLK, 0, st, B.L, T, A);
      }
    }
    if (g > 0) cc_cycle_at(M, g);
  }
}

struct CcKeep { bool built = false; CCMG M; };
CcKeep *cc_keep_new() { return new CcKeep; }
void cc_keep_free(CcKeep *k) { delete k; }

// ---- one cycle as a hipGraph ------------------------------------------------------------------------------------------------------
// A V-cycle at 256^3 is ~110 launches, most of them 3-15 us kernels on the levels <= 64^3: issued one by one the host cannot keep
// the GPU busy (r1: 8 ms of a 50 ms step were launch gaps).  The launch sequence of a cycle depends only on the level structures, so
// it is captured once and replayed; the key hashes every value the launches read from the host side.
static void cc_key_lev(GraphKey &k, const CLev &L) {
  k.put(L.n); k.put(L.PX); k.put(L.PY); k.put(L.sz); k.put(L.hi2); k.put(L.phi); k.put(L.rh); k.put(L.res); k.put(L.b); k.put(L.alpha);
  k.put(L.rho); k.put(L.fold); k.put(L.cmu);
}
static unsigned long long cc_graph_key(const CCMG &M, int what) {
  const vdn_params &P = ctx().prm;
  GraphKey k; k.put(what); k.put(P.mg_nu1); k.put(P.mg_nu2); k.put(P.mg_nub); k.put(M.per); k.put(M.d_nrm);
  k.put(M.sendbuf); k.put(M.recvbuf); k.put(M.d_gb_rh); k.put(M.d_gb_b); k.put(M.cnt_rh); k.put(M.cnt_b);
  for (const CDLev &DL : M.dlev) {
    k.put(xplan_serial(DL.halo)); k.put(DL.ng); k.put(DL.single_box); k.put(DL.res_restricted);
    k.put(DL.split); k.put(xplan_serial(DL.shalo[0])); k.put(xplan_serial(DL.shalo[1]));
    for (const CBox &B : DL.boxes) { cc_key_lev(k, B.L); k.put(B.lo); if (DL.split) { k.put(B.sp.PXH); k.put(B.sp.off); k.put(B.sp.phi); k.put(B.sp.rh); k.put(B.sp.rho); } }
  }
  for (const CLev &L : M.tail) cc_key_lev(k, L);
  for (long o : M.loc_off_rh) k.put(o);
  return k.h;
}
static bool cc_graphable(const CCMG &M) {
  (void)M;
  return graphs_enabled();
}
template <class Body> static void cc_run_cycle(CCMG &M, int what, Body body) {
  if (!cc_graphable(M)) { body(); return; }
  const unsigned long long key = cc_graph_key(M, what);
  if (graph_replay(key)) return;
  graph_begin();
  try { body(); } catch (...) { graph_abort(); throw; }
  graph_end(key);
}

// graph id of the nested iteration: its interpolation kernels take the boundary types as arguments, so they are part of the key
static int cc_fmg_what(const int bc[3][2]) {
  int code = 0;
  for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) code = code * 4 + (bc[d][sd] + 1);
  return 3 + 4 * code;
}
// VDN_MAC_STORED_BETA=1: the finest level reads the stored face coefficients like the others (the measured alternative of DESIGN.md section 4)
static bool beta_from_rho() { static const bool b = !(vdn_env("VDN_MAC_STORED_BETA") && atoi(vdn_env("VDN_MAC_STORED_BETA")) != 0); return b; }
static void cc_setup(CCMG &M, vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *alpha, vdn_multifab **beta, const double *dx, const int bc[3][2],
                     const vdn_multifab *rho = nullptr, double const_beta = 0.0) {
  REQUIRE(phi->ng >= 1, "cc multigrid: phi needs one ghost cell");
  cc_build(M, rh, dx, bc, alpha != nullptr);
  const vdn_layout *la = rh->la; const int lev = rh->lev;
  CDLev &D0 = M.dlev[0];
  // beta = 2 / (rho_i + rho_i-1) (the MAC projection): the finest level recomputes it from rho; not with the fused sweeps (they read b)
  const bool from_rho = rho && !alpha && beta_from_rho() && rho->ng >= 1;
  for (size_t b = 0; b < D0.boxes.size(); b++) {
    CLev &L0 = D0.boxes[b].L;
    const vdn_box &bx = rh->vbox[b];
    // boundary folding only on faces that are DOMAIN faces
    int e[3][2];
    for (int d = 0; d < 3; d++) {
      e[d][0] = (bx.lo[d] == la->pd[lev].lo[d]) ? bc[d][0] : VDN_BC_INT;
      e[d][1] = (bx.hi[d] == la->pd[lev].hi[d]) ? bc[d][1] : VDN_BC_INT;
    }
    hipLaunchKernelGGL(kk_cc_load, g3(L0.n[0] + 1, L0.n[1] + 1, L0.n[2] + 1, BLK), BLK, 0, ctx().stream, L0, rh->fabs[b], phi->fabs[b],
                       alpha ? alpha->fabs[b] : rh->fabs[b], beta[0]->fabs[b], beta[1]->fabs[b], beta[2]->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2],
                       e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1]);
    hipLaunchKernelGGL(kk_cc_load_rh, g3(L0.n[0], L0.n[1], L0.n[2], BLK), BLK, 0, ctx().stream, L0, rh->fabs[b], phi->fabs[b],
                       bx.lo[0], bx.lo[1], bx.lo[2], e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1]);
    if (from_rho) {
      double *r = (double *)arena_alloc(sizeof(double) * L0.sz);
      hipLaunchKernelGGL(kk_cc_load_rho, g3(L0.n[0] + 2, L0.n[1] + 2, L0.n[2] + 2, BLK), BLK, 0, ctx().stream, L0, r, rho->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
      L0.rho = r;
      for (int d = 0; d < 3; d++) { L0.fold[d][0] = e[d][0]; L0.fold[d][1] = e[d][1]; }
    }
    if (const_beta > 0.0 && alpha) {        // the caller vouches that every face coefficient is this constant (visc_solve / diff_scalar_solve: setval(beta, mu))
      L0.cmu = const_beta;
      for (int d = 0; d < 3; d++) { L0.fold[d][0] = e[d][0]; L0.fold[d][1] = e[d][1]; }
    }
  }
  for (size_t l = 1; l < M.dlev.size(); l++)
    for (size_t b = 0; b < M.dlev[l].boxes.size(); b++) {
      const CLev &C = M.dlev[l].boxes[b].L;
      hipLaunchKernelGGL(kk_cc_coarsen_b, g3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1, BLK), BLK, 0, ctx().stream, M.dlev[l - 1].boxes[b].L, C);
      if (alpha) hipLaunchKernelGGL(kk_cc_coarsen_cell, g3(C.n[0], C.n[1], C.n[2], BLK), BLK, 0, ctx().stream, M.dlev[l - 1].boxes[b].L, (const double *)M.dlev[l - 1].boxes[b].L.alpha, C, C.alpha);
    }
  if (!M.tail.empty()) {
    CDLev &DL = M.dlev.back();
    for (size_t b = 0; b < DL.boxes.size(); b++) {
      const CLev &F = DL.boxes[b].L;
      const int nx = F.n[0] / 2, ny = F.n[1] / 2, nz = F.n[2] / 2;
      hipLaunchKernelGGL(kk_cc_coarsen_b_pack, g3(nx + 1, ny + 1, nz + 1, BLK), BLK, 0, ctx().stream, F, M.sendbuf, M.loc_off_b[b], nx, ny, nz);
    }
    comm_allgather_dev(M.sendbuf, M.recvbuf, M.cnt_b);
    hipLaunchKernelGGL(kk_cc_unpack_b, dim3(4, 1, (unsigned)M.gb_b.size()), dim3(256), 0, ctx().stream, M.tail[0], M.recvbuf, M.d_gb_b);
    if (alpha) {                       // alpha of the first tail level: per-box 8-cell means, gathered like the residual
      for (size_t b = 0; b < DL.boxes.size(); b++) {
        const CLev &F = DL.boxes[b].L;
        const int nx = F.n[0] / 2, ny = F.n[1] / 2, nz = F.n[2] / 2;
        hipLaunchKernelGGL(kk_cc_restrict_pack, g3(nx, ny, nz, BLK), BLK, 0, ctx().stream, F, (const double *)F.alpha, M.sendbuf, M.loc_off_rh[b], nx, ny, nz);
      }
      comm_allgather_dev(M.sendbuf, M.recvbuf, M.cnt_rh);
      hipLaunchKernelGGL(kk_cc_unpack_rh, dim3(4, 1, (unsigned)M.gb_rh.size()), dim3(256), 0, ctx().stream, M.tail[0], M.tail[0].alpha, M.recvbuf, M.d_gb_rh);
    }
    for (size_t l = 1; l < M.tail.size(); l++) {
      const CLev &C = M.tail[l];
      hipLaunchKernelGGL(kk_cc_coarsen_b, g3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1, BLK), BLK, 0, ctx().stream, M.tail[l - 1], C);
      if (alpha) hipLaunchKernelGGL(kk_cc_coarsen_cell, g3(C.n[0], C.n[1], C.n[2], BLK), BLK, 0, ctx().stream, M.tail[l - 1], (const double *)M.tail[l - 1].alpha, C, C.alpha);
    }
  }
}
// set-up of macproject's fast path: levels from the layout of rho, level 0 = rho + the right-hand side from the MAC field, level 1 from rho,
// the rest as cc_setup; returns max |rh| (all ranks)
static double cc_setup_fast(CCMG &M, CcFast *fast, const double *dx, const int bc[3][2]) {
  const vdn_multifab *rho = fast->rho;
  cc_build(M, rho, dx, bc, false);
  REQUIRE(M.dlev.size() >= 2, "cc_setup_fast: needs a second distributed level");
  const vdn_layout *la = rho->la; const int lev = rho->lev;
  CDLev &D0 = M.dlev[0];
  HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), ctx().stream));
  for (size_t b = 0; b < D0.boxes.size(); b++) {
    CLev &L0 = D0.boxes[b].L;
    const vdn_box &bx = rho->vbox[b];
    int e[3][2];
    for (int d = 0; d < 3; d++) {
      e[d][0] = (bx.lo[d] == la->pd[lev].lo[d]) ? bc[d][0] : VDN_BC_INT;
      e[d][1] = (bx.hi[d] == la->pd[lev].hi[d]) ? bc[d][1] : VDN_BC_INT;
    }
    double *r = (double *)arena_alloc(sizeof(double) * L0.sz);
    hipLaunchKernelGGL(kk_cc_load_rho, g3(L0.n[0] + 2, L0.n[1] + 2, L0.n[2] + 2, BLK), BLK, 0, ctx().stream, L0, r, rho->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
    L0.rho = r;
    for (int d = 0; d < 3; d++) { L0.fold[d][0] = e[d][0]; L0.fold[d][1] = e[d][1]; }
    hipLaunchKernelGGL(kk_cc_load_divumac, g3(L0.n[0], L0.n[1], std::min(L0.n[2], 16), BLK), BLK, 0, ctx().stream, L0, fast->um[0]->fabs[b], fast->um[1]->fabs[b], fast->um[2]->fabs[b],
                       fast->mac_rhs->fabs[b], 1.0 / dx[0], 1.0 / dx[1], 1.0 / dx[2], bx.lo[0], bx.lo[1], bx.lo[2], e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1], M.d_nrm);
  }
  comm_allreduce_max_dev(M.d_nrm, 1);
  const double bnorm = read_scalar1(M.d_nrm);
  for (size_t l = 1; l < M.dlev.size(); l++)
    for (size_t b = 0; b < M.dlev[l].boxes.size(); b++) {
      const CLev &C = M.dlev[l].boxes[b].L;
      if (l == 1) hipLaunchKernelGGL(kk_cc_coarsen_b_rho, g3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1, BLK), BLK, 0, ctx().stream, M.dlev[0].boxes[b].L, C);
      else hipLaunchKernelGGL(kk_cc_coarsen_b, g3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1, BLK), BLK, 0, ctx().stream, M.dlev[l - 1].boxes[b].L, C);
    }
  if (!M.tail.empty()) {
    CDLev &DL = M.dlev.back();
    for (size_t b = 0; b < DL.boxes.size(); b++) {
      const CLev &F = DL.boxes[b].L;
      const int nx = F.n[0] / 2, ny = F.n[1] / 2, nz = F.n[2] / 2;
      hipLaunchKernelGGL(kk_cc_coarsen_b_pack, g3(nx + 1, ny + 1, nz + 1, BLK), BLK, 0, ctx().stream, F, M.sendbuf, M.loc_off_b[b], nx, ny, nz);
    }
    comm_allgather_dev(M.sendbuf, M.recvbuf, M.cnt_b);
    hipLaunchKernelGGL(kk_cc_unpack_b, dim3(4, 1, (unsigned)M.gb_b.size()), dim3(256), 0, ctx().stream, M.tail[0], M.recvbuf, M.d_gb_b);
    for (size_t l = 1; l < M.tail.size(); l++) {
      const CLev &C = M.tail[l];
      hipLaunchKernelGGL(kk_cc_coarsen_b, g3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1, BLK), BLK, 0, ctx().stream, M.tail[l - 1], C);
    }
  }
  return bnorm;
}
// a kept hierarchy (coefficients on every level stay): the finest level takes a new right-hand side and initial guess
static void cc_reload(CCMG &M, vdn_multifab *rh, vdn_multifab *phi, const int bc[3][2], bool zero_guess = false) {
  const vdn_layout *la = rh->la; const int lev = rh->lev;
  CDLev &D0 = M.dlev[0];
  for (size_t b = 0; b < D0.boxes.size(); b++) {
    CLev &L0 = D0.boxes[b].L;
    const vdn_box &bx = rh->vbox[b];
    int e[3][2];
    for (int d = 0; d < 3; d++) {
      e[d][0] = (bx.lo[d] == la->pd[lev].lo[d]) ? bc[d][0] : VDN_BC_INT;
      e[d][1] = (bx.hi[d] == la->pd[lev].hi[d]) ? bc[d][1] : VDN_BC_INT;
    }
    if (zero_guess) { if (!D0.split) HIPCHK(hipMemsetAsync(L0.phi, 0, sizeof(double) * L0.sz, ctx().stream)); }      // (the caller's phi is neither zero-filled nor read; by colour: the split arrays hold phi)
    else hipLaunchKernelGGL(kk_cc_load_phi, g3(L0.n[0], L0.n[1], L0.n[2], BLK), BLK, 0, ctx().stream, L0, phi->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
    hipLaunchKernelGGL(kk_cc_load_rh, g3(L0.n[0], L0.n[1], L0.n[2], BLK), BLK, 0, ctx().stream, L0, rh->fabs[b], phi->fabs[b],
                       bx.lo[0], bx.lo[1], bx.lo[2], e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1], zero_guess ? 1 : 0);
  }
  if (D0.split) {              // a kept hierarchy whose finest level lives by colour (the level-0 V-cycles of the composite MAC solve): the new right-hand side and guess
    cc_to_split(D0, zero_guess ? 2 : 3);
    if (zero_guess) for (const CBox &B : D0.boxes) HIPCHK(hipMemsetAsync(B.sp.phi[0], 0, sizeof(double) * 2 * B.sp.tot, ctx().stream));      // (phi[0], phi[1] are adjacent: cc_split_setup)
  }
}
static void cc_store(CCMG &M, vdn_multifab *phi, const int bc[3][2], vdn_multifab *add_to = nullptr) {
  CDLev &D0 = M.dlev[0];
  cc_halo(M, D0);
  const vdn_layout *la = phi->la; const int lev = phi->lev;
  for (size_t b = 0; b < D0.boxes.size(); b++) {
    const CLev &L0 = D0.boxes[b].L;
    const vdn_box &bx = phi->vbox[b];
    int e[3][2];
    for (int d = 0; d < 3; d++) {
      e[d][0] = (bx.lo[d] == la->pd[lev].lo[d]) ? bc[d][0] : VDN_BC_INT;
      e[d][1] = (bx.hi[d] == la->pd[lev].hi[d]) ? bc[d][1] : VDN_BC_INT;
    }
    hipLaunchKernelGGL(kk_cc_store, g3(L0.n[0] + 2, L0.n[1] + 2, L0.n[2] + 2, BLK), BLK, 0, ctx().stream, L0, phi->fabs[b],
                       bx.lo[0], bx.lo[1], bx.lo[2], e[0][0], e[0][1], e[1][0], e[1][1], e[2][0], e[2][1], add_to ? 1 : 0, add_to ? add_to->fabs[b] : phi->fabs[b]);
  }
}

// fast: macproject's single-level call (CcFast in vdn_internal.h): rh, phi and beta are not used (may be null); phi comes back as views of the
// finest level's array (ghost cells exchanged) and the level arrays stay allocated -- the CALLER releases the arena
int cc_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int bc[3][2],
             double rel_eps, double abs_eps, int max_iter, int *cycles, double *res0, double *res, const vdn_multifab *alpha, const vdn_multifab *rho, CcKeep *keep,
             CcFast *fast, int fmg, bool zero_guess, vdn_multifab *add_to, double const_beta) {
  Prof prof_("mac_multigrid");
  if (ctx().prm.dm == 2) return cc2_solve(rh, phi, beta, dx, bc, rel_eps, abs_eps, max_iter, cycles, res0, res, alpha);
  const vdn_params &P = ctx().prm;
  if (fast) REQUIRE(!keep && !alpha && max_iter >= 0 && fast->rho && fast->rho->ng >= 1, "cc_solve: bad use of the fast path");
  size_t mark = arena_mark();
  // keep: the hierarchy (arrays in the caller's arena scope, coefficients on every level) survives the call; the next call with the
  // same `keep` loads only its right-hand side and phi (the composite solves: one V-cycle per FAC iteration on the same coefficients)
  CCMG M_local;
  CCMG &M = keep ? keep->M : M_local;
  double bnorm_fast = 0.0;
  if (fast) {
    bnorm_fast = cc_setup_fast(M, fast, dx, bc);
    if (cc_split_ok(M)) cc_split_setup(M);
    g_mac_level_form = !M.dlev[0].split ? 0 : (vdn_env("VDN_MAC_SPLIT") && atoi(vdn_env("VDN_MAC_SPLIT")) == 2) ? 2 : 1;
  }
  else if (keep && keep->built) { cc_reload(M, rh, phi, bc, zero_guess); g_mac_level_form = M.dlev[0].split ? 1 : 0; }
  else {
    cc_setup(M, rh, phi, alpha, beta, dx, bc, rho, const_beta);
    g_mac_level_form = 0;
    // round 6: the V-cycles a composite MAC solve runs on its level 0 (a kept hierarchy, one cycle per FAC iteration, the density form) take the level by colour
    // too -- for the tagged 256^3 hierarchies that is a whole 256^3 level, 23 cycles per step
    if (keep && max_iter < 0 && (!alpha || const_beta > 0.0) && cc_split_ok(M)) { cc_split_setup(M); cc_to_split(M.dlev[0], 1); g_mac_level_form = 1; }      // (also the composite viscous solves: constant coefficients)
    // ... and the viscous / diffusive solves (constant face coefficients: the caller's const_beta): every 3-D input of exec/test runs three of them per step, at 256^3 they were
    // 24 ms of a 51 ms step on the stored-coefficient passes (56 B per cell and pass; by colour, without coefficient arrays: 20)
    else if (!keep && max_iter >= 0 && alpha && const_beta > 0.0 && cc_split_ok(M)) { cc_split_setup(M); g_mac_level_form = 1; }      // (phi goes over after the nested iteration, below)
  }
  if (keep) keep->built = true;
  CDLev &D0 = M.dlev[0];
  const bool single = (M.dlev.size() == 1 && M.tail.empty());
  // fmg: the caller's phi is zero, ghost cells included, and the solve starts from a nested iteration (not before a fixed number of cycles);
  // the fast path: vdn_params.mac_fmg
  if (fast) fmg = P.mac_fmg ? 1 : 0;
  if (max_iter < 0) {            // exactly -max_iter V-cycles, no norms, no convergence test (the coarse correction of the composite solves)
    for (int c = 0; c < -max_iter; c++) {
      if (single) { const int N = std::max(D0.ng[0], std::max(D0.ng[1], D0.ng[2])); cc_gsrb_d(M, D0, std::max(P.mg_nub, N * N)); continue; }
      cc_run_cycle(M, 2, [&] {
        const bool slabs = D0.split && !D0.halo && D0.boxes.size() == 1 && mac_slab(D0.boxes[0].L) > 0;
        if (slabs) cc_split_run(M, D0, false, P.mg_nu1, true, false, false);
        else { cc_gsrb_d(M, D0, P.mg_nu1); cc_residual_d(M, D0, false); }
        cc_restrict_down(M, 0);
        if (M.dlev.size() > 1) cc_vcycle_d(M, 1); else cc_vcycle_t(M, 0);
        if (slabs) cc_split_run(M, D0, true, P.mg_nu2, false, false, false);
        else cc_prolong_smooth(M, 0, P.mg_nu2);
      });
    }
    if (D0.split) cc_from_split(D0);
    cc_store(M, phi, bc, add_to);
    if (cycles) *cycles = -max_iter; if (res0) *res0 = 0.0; if (res) *res = 0.0;
    if (!keep) arena_release(mark);
    return 0;
  }
  const double bnorm = fast ? bnorm_fast : mf_norm_inf(rh, 0, 1);
  int cyc = 0; bool conv = (bnorm == 0.0); double rn = 0.0;
  // pre-smoothing + residual, then per cycle: [coarse correction, post-smoothing, the next cycle's pre-smoothing, residual + norm] as ONE
  // replayed graph and one 8-byte read-back -- the same launch sequence as testing the residual the cycle computes after pre-smoothing
  const int nbot = std::max(P.mg_nub, std::max(D0.ng[0], std::max(D0.ng[1], D0.ng[2])) * std::max(D0.ng[0], std::max(D0.ng[1], D0.ng[2])));
  if (fmg && !conv && !single && bnorm < HUGE_VAL) cc_run_cycle(M, cc_fmg_what(bc), [&] { cc_fmg(M, bc); });
  if (D0.split) cc_to_split(D0, 1);
  // vdn_params.mg_predict (macproject's call: zero guess): see nd_solve in mg_nd.hip -- the norms of the cycles before the one the previous solve of this
  // size stopped at, minus one, go into the device-side history and are read in one go; a history that shows an earlier stop repeats the solve
  const int gn[3] = { D0.ng[0], D0.ng[1], D0.ng[2] };
  const int pred = (fast && !single && !conv && bnorm < HUGE_VAL) ? std::min(mg_predict_get(0, gn), std::min(max_iter, 63)) : 0;
  if (!conv) {
    if (single) cc_gsrb_d(M, D0, nbot); else cc_fine_seq(M, false, pred >= 2, true, false);
    if (pred >= 2) {
      norm_hist_reset(); norm_hist_push(M.d_nrm);
      cc_run_cycle(M, 4 + 4 * pred, [&] {      // all blind cycles as ONE graph (ids: 1 and 2 the plain cycles, 3 + 4 code the nested iterations, multiples of 4 these)
        for (int c = 1; c <= pred - 1; c++) {
          cc_restrict_down(M, 0);
          if (M.dlev.size() > 1) cc_vcycle_d(M, 1); else cc_vcycle_t(M, 0);
          cc_fine_seq(M, true, true, true, false);
          norm_hist_push(M.d_nrm);
        }
      });
      const double *h = norm_hist_read(pred);
      int first = -1;
      for (int c = 0; c < pred && first < 0; c++)
        if (((h[c] <= rel_eps * bnorm && bnorm < HUGE_VAL) || h[c] <= abs_eps) || !(h[c] < HUGE_VAL)) first = c;
      if (first >= 0 && first < pred - 1) {                      // overshot: repeat without the prediction
        arena_release(mark);
        struct Off { Off() { g_mg_predict_off++; } ~Off() { g_mg_predict_off--; } } off_;
        return cc_solve(rh, phi, beta, dx, bc, rel_eps, abs_eps, max_iter, cycles, res0, res, alpha, rho, keep, fast, fmg, zero_guess, add_to, const_beta);
      }
      cyc = pred - 1; rn = h[pred - 1];
    } else {
      cc_residual_d(M, D0, true); rn = read_scalar(M.d_nrm);
    }
  }
  while (!conv) {
    if ((rn <= rel_eps * bnorm && bnorm < HUGE_VAL) || rn <= abs_eps) { conv = true; break; }
    if (cyc >= max_iter || !(rn < HUGE_VAL) || !(bnorm < HUGE_VAL)) break;     // also: a NaN / inf norm (the reductions turn NaN into +inf)
    if (single) { cc_gsrb_d(M, D0, nbot); cc_residual_d(M, D0, true); }
    else cc_run_cycle(M, 1, [&] {
      cc_restrict_down(M, 0);
      if (M.dlev.size() > 1) cc_vcycle_d(M, 1); else cc_vcycle_t(M, 0);
      cc_fine_seq(M, true, true, true);
    });
    cyc++;
    rn = read_scalar(M.d_nrm);
  }
  if (fast) {
    CDLev &DF = M.dlev[0];
    if (DF.split) cc_from_split(DF);
    cc_halo(M, DF);
    fast->phi_view.clear();
    for (size_t b = 0; b < DF.boxes.size(); b++) fast->phi_view.push_back(cc_phi_view(DF.boxes[b].L, fast->rho->vbox[b].lo));
  } else { if (M.dlev[0].split) cc_from_split(M.dlev[0]); cc_store(M, phi, bc); }
  if (cycles) *cycles = cyc; if (res0) *res0 = bnorm; if (res) *res = rn;
  if (conv && fast && !single && cyc >= 1) mg_predict_set(0, gn, cyc);
  if (!keep && !fast) arena_release(mark);
  return conv ? 0 : 1;
}

void cc_smooth(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int bc[3][2], int nsweeps) {
  size_t mark = arena_mark();
  CCMG M; cc_setup(M, rh, phi, nullptr, beta, dx, bc);
  cc_gsrb_d(M, M.dlev[0], nsweeps);
  cc_store(M, phi, bc);
  arena_release(mark);
}

// slab_sweeps > 0 (the level by colour in one box): the launches run as they do INSIDE a solve -- sweeps of two colour passes time-skewed over plane slabs
// (cc_split_run), the slab served from the Infinity Cache from its second pass on; avg_ms is then the time per pass over the whole level
void cc_bench_smoother(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const vdn_multifab *rho, const double *dx, const int bc[3][2],
                       int nlaunch, double *avg_ms, long *cells, int slab_sweeps) {
  size_t mark = arena_mark();
  CCMG M; cc_setup(M, rh, phi, nullptr, beta, dx, bc, rho);
  REQUIRE(M.dlev[0].boxes.size() == 1, "smoother probe: one local box expected");
  const CLev &L = M.dlev[0].boxes[0].L;
  hipStream_t st = ctx().stream;
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  CDLev &D0 = M.dlev[0];
  if (cc_split_ok(M)) { cc_split_setup(M); cc_to_split(D0, 1); }            // the form macproject's solve runs (VDN_MAC_SPLIT=0: the interleaved pass)
  g_mac_level_form = D0.split ? 1 : 0;                                      // (bench.py reads it back: which kernel the probe timed)
  auto pass = [&](int w) { if (D0.split) launch_gsrb_split<0>(D0.boxes[0], w & 1, st, L); else launch_gsrb(L, w & 1, st); };
  const bool slabs = slab_sweeps > 0 && D0.split && !D0.halo && mac_slab(L) > 0;
  if (slab_sweeps > 0 && !slabs) { *avg_ms = 0.0; *cells = 0; HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1)); arena_release(mark); return; }      // (no slab schedule on this level)
  const int nrun = slabs ? std::max(1, nlaunch / (2 * slab_sweeps)) : 0;
  if (slabs) nlaunch = nrun * 2 * slab_sweeps;
  for (int w = 0; w < 4; w++) pass(w);
  HIPCHK(hipEventRecord(e0, st));
  if (slabs) for (int w = 0; w < nrun; w++) cc_split_run(M, D0, false, slab_sweeps, false, false, false);
  else for (int w = 0; w < nlaunch; w++) pass(w);
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_ms = (double)ms / nlaunch; *cells = (long)L.n[0] * L.n[1] * L.n[2];
  HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
  arena_release(mark);
}

// ====================================================================================================
// MAC projection pieces (macproject.f90)
// ====================================================================================================
struct divumac_K { FV um; FV vm; FV wm; FV macrhs; FV rh; double dxi0; double dxi1; double dxi2;
  __device__ void cell(int i, int j, int k) const {
    // macproject.f90:270-272 then 190-196:  rh = -div + mac_rhs
    double div = (fv_get(um, i + 1, j, k) - fv_get(um, i, j, k)) * dxi0
               + (fv_get(vm, i, j + 1, k) - fv_get(vm, i, j, k)) * dxi1
               + (fv_get(wm, i, j, k + 1) - fv_get(wm, i, j, k)) * dxi2;
    fv_at(rh, i, j, k) = div * -1.0 + fv_get(macrhs, i, j, k);
  } };

struct mk_mac_coeffs_K { FV rho; FV bx; FV by; FV bz; int h0; int h1; int h2;
  __device__ void cell(int i, int j, int k) const {
    const double r0 = fv_get(rho, i, j, k);
    if (j <= h1 && k <= h2) fv_at(bx, i, j, k) = 2.0 / (r0 + fv_get(rho, i - 1, j, k));     // macproject.f90:376
    if (i <= h0 && k <= h2) fv_at(by, i, j, k) = 2.0 / (r0 + fv_get(rho, i, j - 1, k));     // 385
    if (i <= h0 && j <= h1) fv_at(bz, i, j, k) = 2.0 / (r0 + fv_get(rho, i, j, k - 1));     // 394
  } };

struct UmacArgs { int lo[3], hi[3]; int ebc[3][2]; double dx[3]; };
struct mkumac_K { FV um; FV vm; FV wm; FV phi; FV bx; FV by; FV bz; UmacArgs A;
  __device__ void cell(int i, int j, int k) const {
    const double p0 = fv_get(phi, i, j, k);
    // macproject.f90:608-612; box faces use the ghost value the solver's closure implies, Neumann faces keep umac
    if (j <= A.hi[1] && k <= A.hi[2]) {
      int side = (i == A.lo[0]) ? 0 : (i == A.hi[0] + 1 ? 1 : -1);
      if (!(side >= 0 && A.ebc[0][side] == VDN_BC_NEU)) {
        double g = (p0 - fv_get(phi, i - 1, j, k)) / A.dx[0];
        fv_at(um, i, j, k) = fv_get(um, i, j, k) - fv_get(bx, i, j, k) * g;
      }
    }
    if (i <= A.hi[0] && k <= A.hi[2]) {
      int side = (j == A.lo[1]) ? 0 : (j == A.hi[1] + 1 ? 1 : -1);
      if (!(side >= 0 && A.ebc[1][side] == VDN_BC_NEU)) {
        double g = (p0 - fv_get(phi, i, j - 1, k)) / A.dx[1];
        fv_at(vm, i, j, k) = fv_get(vm, i, j, k) - fv_get(by, i, j, k) * g;
      }
    }
    if (i <= A.hi[0] && j <= A.hi[1]) {
      int side = (k == A.lo[2]) ? 0 : (k == A.hi[2] + 1 ? 1 : -1);
      if (!(side >= 0 && A.ebc[2][side] == VDN_BC_NEU)) {
        double g = (p0 - fv_get(phi, i, j, k - 1)) / A.dx[2];
        fv_at(wm, i, j, k) = fv_get(wm, i, j, k) - fv_get(bz, i, j, k) * g;
      }
    }
  } };


// mkumac on the level array: phi's ghost cell beyond a Dirichlet face is what kk_cc_store would have written (-phi of the cell inside), beyond a box
// face with a neighbour or a periodic image the exchanged value; beta = mk_mac_coeffs_K's expression
DEVI double phi_closed(const FV &phi, const UmacArgs &A, int i, int j, int k) {
  if (i < A.lo[0]) { if (A.ebc[0][0] == VDN_BC_DIR) return -fv_get(phi, A.lo[0], j, k); }
  else if (i > A.hi[0]) { if (A.ebc[0][1] == VDN_BC_DIR) return -fv_get(phi, A.hi[0], j, k); }
  else if (j < A.lo[1]) { if (A.ebc[1][0] == VDN_BC_DIR) return -fv_get(phi, i, A.lo[1], k); }
  else if (j > A.hi[1]) { if (A.ebc[1][1] == VDN_BC_DIR) return -fv_get(phi, i, A.hi[1], k); }
  else if (k < A.lo[2]) { if (A.ebc[2][0] == VDN_BC_DIR) return -fv_get(phi, i, j, A.lo[2]); }
  else if (k > A.hi[2]) { if (A.ebc[2][1] == VDN_BC_DIR) return -fv_get(phi, i, j, A.hi[2]); }
  return fv_get(phi, i, j, k);
}
struct mkumac_rho_K { FV um; FV vm; FV wm; FV phi; FV rho; UmacArgs A;
  static constexpr bool in_constant = true;       // phi_closed walks A.ebc / A.lo / A.hi: a by-value copy in a batched launch lands in scratch
  // m: max |value| of the valid faces this cell owns AFTER the update (kk_macmax's masks: mkflux's eps, mkflux.f90:1374-1401)
  __device__ void cell_m(int i, int j, int k, double &m) const {
    const double p0 = phi_closed(phi, A, i, j, k);
    const double r0 = fv_get(rho, i, j, k);
    if (j <= A.hi[1] && k <= A.hi[2]) {
      int side = (i == A.lo[0]) ? 0 : (i == A.hi[0] + 1 ? 1 : -1);
      double v = fv_get(um, i, j, k);
      if (!(side >= 0 && A.ebc[0][side] == VDN_BC_NEU)) {
        double g = (p0 - phi_closed(phi, A, i - 1, j, k)) / A.dx[0];
        v = v - (2.0 / (r0 + fv_get(rho, i - 1, j, k))) * g;
        fv_at(um, i, j, k) = v;
      }
      m = fmax(m, fabs(v));
    }
    if (i <= A.hi[0] && k <= A.hi[2]) {
      int side = (j == A.lo[1]) ? 0 : (j == A.hi[1] + 1 ? 1 : -1);
      double v = fv_get(vm, i, j, k);
      if (!(side >= 0 && A.ebc[1][side] == VDN_BC_NEU)) {
        double g = (p0 - phi_closed(phi, A, i, j - 1, k)) / A.dx[1];
        v = v - (2.0 / (r0 + fv_get(rho, i, j - 1, k))) * g;
        fv_at(vm, i, j, k) = v;
      }
      m = fmax(m, fabs(v));
    }
    if (i <= A.hi[0] && j <= A.hi[1]) {
      int side = (k == A.lo[2]) ? 0 : (k == A.hi[2] + 1 ? 1 : -1);
      double v = fv_get(wm, i, j, k);
      if (!(side >= 0 && A.ebc[2][side] == VDN_BC_NEU)) {
        double g = (p0 - phi_closed(phi, A, i, j, k - 1)) / A.dx[2];
        v = v - (2.0 / (r0 + fv_get(rho, i, j, k - 1))) * g;
        fv_at(wm, i, j, k) = v;
      }
      m = fmax(m, fabs(v));
    }
  }
  __device__ void cell(int i, int j, int k) const { double m = 0.0; cell_m(i, j, k, m); } };
// the same with max |umac| of the box riding along (one box, one level: the value mkflux asks for twice per step -- kk_macmax read the three arrays again for it)
__global__ void __launch_bounds__(256) kk_mkumac_rho_max(mkumac_rho_K K, Range3 r, double *umax) {
  REDUCE_IJ(r)
  double m = 0.0;
  if (in_ij) REDUCE_KLOOP(r) K.cell_m(i, j, k, m);
  block_atomic_max(umax, m);
}
static void mac_level_mkumac_rho(vdn_multifab **um, const std::vector<FV> &phi_view, const vdn_multifab *rho, const double *dx, const vdn_bc_tower *bct, int bc_comp0) {
  const int n = rho->lev;
  std::vector<std::pair<mkumac_rho_K, Range3>> v;
  for (int i = 0; i < rho->nfabs(); i++) {
    UmacArgs A; Range3 rf;
    for (int d = 0; d < 3; d++) { A.lo[d] = rf.lo[d] = rho->vbox[i].lo[d]; A.hi[d] = rho->vbox[i].hi[d]; rf.hi[d] = A.hi[d] + 1; A.dx[d] = dx[d];
      for (int s = 0; s < 2; s++) A.ebc[d][s] = bct->ell_bc(n, i + 1, d, s, bc_comp0); }
    v.push_back({ mkumac_rho_K{ um[0]->fabs[i], um[1]->fabs[i], um[2]->fabs[i], phi_view[i], rho->fabs[i], A }, rf });
  }
  // advance_timestep on one level, one box: the step's cache of max |umac| (godunov.hip: macmax_cache) is filled here
  static const bool fuse_max = !(vdn_env("VDN_MAC_UMAX") && atoi(vdn_env("VDN_MAC_UMAX")) == 0);
  VdnCtx &c = ctx();
  if (fuse_max && v.size() == 1 && rho->la->nlev == 1 && c.macmax_cache.size() == 1 && c.macmax_cache[0]) {
    HIPCHK(hipMemsetAsync(c.macmax_cache[0], 0, sizeof(double), c.stream));
    dim3 g = grid_for(v[0].second, dim3(64, 4, 1));
    if (g.z > 64) g.z = 64;                                       // (one atomic per workgroup, each walking its share of the planes; 8 / 32 / 64 / 258 chunks: MAC 9.085 / 9.062 / 9.051 / 9.13 ms.
                                                                  //  The velocity update costs 0.08 ms more this way, the scalar advance 0.11 ms less: a small net gain)
    hipLaunchKernelGGL(kk_mkumac_rho_max, g, dim3(64, 4, 1), 0, c.stream, v[0].first, v[0].second, c.macmax_cache[0]);
    c.macmax_src[0] = um[0]->fabs[0].p;
    return;
  }
  launch_cells(v, ctx().stream);
}

// per-level pieces of macproject, shared by the single-level driver below and the multilevel one in amr.hip
void mac_level_rhs(vdn_multifab **um, const vdn_multifab *mac_rhs, vdn_multifab *rh, const double *dx) {      // divumac + (190-196)
  std::vector<std::pair<divumac_K, Range3>> v;
  for (int i = 0; i < rh->nfabs(); i++) {
    Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = rh->vbox[i].lo[d]; r.hi[d] = rh->vbox[i].hi[d]; }
    v.push_back({ divumac_K{ um[0]->fabs[i], um[1]->fabs[i], um[2]->fabs[i], mac_rhs->fabs[i], rh->fabs[i], 1.0 / dx[0], 1.0 / dx[1], 1.0 / dx[2] }, r });
  }
  launch_cells(v, ctx().stream);
}
void mac_level_coeffs(const vdn_multifab *rho, vdn_multifab **beta) {                                          // mk_mac_coeffs_3d
  std::vector<std::pair<mk_mac_coeffs_K, Range3>> v;
  for (int i = 0; i < rho->nfabs(); i++) {
    Range3 rf; for (int d = 0; d < 3; d++) { rf.lo[d] = rho->vbox[i].lo[d]; rf.hi[d] = rho->vbox[i].hi[d] + 1; }
    v.push_back({ mk_mac_coeffs_K{ rho->fabs[i], beta[0]->fabs[i], beta[1]->fabs[i], beta[2]->fabs[i], rf.hi[0] - 1, rf.hi[1] - 1, rf.hi[2] - 1 }, rf });
  }
  launch_cells(v, ctx().stream);
}
void mac_level_mkumac(vdn_multifab **um, const vdn_multifab *phi, vdn_multifab **beta, const double *dx, const vdn_bc_tower *bct, int bc_comp0) {
  const int n = phi->lev;
  std::vector<std::pair<mkumac_K, Range3>> v;
  for (int i = 0; i < phi->nfabs(); i++) {
    UmacArgs A; Range3 rf;
    for (int d = 0; d < 3; d++) { A.lo[d] = rf.lo[d] = phi->vbox[i].lo[d]; A.hi[d] = phi->vbox[i].hi[d]; rf.hi[d] = A.hi[d] + 1; A.dx[d] = dx[d];
      for (int s = 0; s < 2; s++) A.ebc[d][s] = bct->ell_bc(n, i + 1, d, s, bc_comp0); }
    v.push_back({ mkumac_K{ um[0]->fabs[i], um[1]->fabs[i], um[2]->fabs[i], phi->fabs[i], beta[0]->fabs[i], beta[1]->fabs[i], beta[2]->fabs[i], A }, rf });
  }
  launch_cells(v, ctx().stream);
}

void do_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs, const double *dx,
                   const vdn_bc_tower *bct, int bc_comp0) {
  if (ctx().prm.dm == 2) { do2_macproject(mla, umac, rho, mac_rhs, dx, bct, bc_comp0); return; }
  if (mla->nlev > 1) { do_ml_macproject(mla, umac, rho, mac_rhs, dx, bct, bc_comp0); return; }
  const int n = 0;
  size_t mark = arena_mark();
  {
    static const bool fast_on = !(vdn_env("VDN_MAC_FAST") && atoi(vdn_env("VDN_MAC_FAST")) == 0);
    // the second level must exist (its coefficients come from the first level's rho): boxes that halve cleanly to >= 4 cells, as cc_build asks
    bool ok = fast_on && beta_from_rho() && rho[n]->ng >= 1;
    {   // cc_build's rule for a second DISTRIBUTED level: the domain coarsens, the boxes halve cleanly and stay at least min_dist wide
      const int agglom = mg_agglom(mla, n);
      const int min_dist = mla->boxes[n].size() > 1 ? agglom : 4;
      for (const vdn_box &b : mla->boxes[n]) for (int d = 0; d < 3; d++) { const int w = b.hi[d] - b.lo[d] + 1; if ((w & 1) || w / 2 < min_dist || ((w / 2) & 1)) ok = false; }
      for (int d = 0; d < 3; d++) { const int N = mla->pd[n].hi[d] - mla->pd[n].lo[d] + 1; if ((N & 1) || N <= 2) ok = false; }
    }
    if (ok) {
      vdn_multifab *um[3] = { umac[0], umac[1], umac[2] };
      int ebc[3][2];
      for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc[d][s] = bct->ell_bc(n, 0, d, s, bc_comp0);
      CcFast F; F.um = um; F.mac_rhs = mac_rhs[n]; F.rho = rho[n];
      int cyc; double r0, rr;
      int rc = cc_solve(nullptr, nullptr, nullptr, dx, ebc, ctx().prm.mac_rel_eps, -1.0, ctx().prm.mg_max_iter, &cyc, &r0, &rr, nullptr, rho[n], nullptr, &F);
      ctx().solver_cycles[0] = cyc; ctx().solver_res0[0] = r0; ctx().solver_res[0] = rr;
      solver_check(rc, "MAC multigrid", cyc, rr, r0);
      mac_level_mkumac_rho(um, F.phi_view, rho[n], dx, bct, bc_comp0);
      for (int d = 0; d < 3; d++) mf_fill_boundary(um[d]);        // macproject.f90:115-119
      arena_release(mark);
      return;
    }
  }
  vdn_multifab *rh = mf_temp(mla, n, 1, 0, -1, false, 0.0);
  vdn_multifab *phi = mf_temp(mla, n, 1, 1, -1, true, 0.0);
  vdn_multifab *beta[3];
  for (int d = 0; d < 3; d++) beta[d] = mf_temp(mla, n, 1, 0, d, false, 0.0);
  vdn_multifab *um[3] = { umac[0], umac[1], umac[2] };
  REQUIRE(rho[n]->ng >= 1, "macproject: rho needs a filled ghost cell");
  mac_level_rhs(um, mac_rhs[n], rh, dx);
  mac_level_coeffs(rho[n], beta);
  int ebc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc[d][s] = bct->ell_bc(n, 0, d, s, bc_comp0);   // grid 0 = whole domain
  int cyc; double r0, rr;
  int rc = cc_solve(rh, phi, beta, dx, ebc, ctx().prm.mac_rel_eps, -1.0, ctx().prm.mg_max_iter, &cyc, &r0, &rr, nullptr, rho[n], nullptr, nullptr,
                    ctx().prm.mac_fmg ? 1 : 0);   // macproject.f90:91-93 (phi was created zero above)
  ctx().solver_cycles[0] = cyc; ctx().solver_res0[0] = r0; ctx().solver_res[0] = rr;
  solver_check(rc, "MAC multigrid", cyc, rr, r0);
  mac_level_mkumac(um, phi, beta, dx, bct, bc_comp0);
  for (int d = 0; d < 3; d++) mf_fill_boundary(um[d]);          // macproject.f90:115-119
  for (int d = 0; d < 3; d++) mf_temp_free(beta[d]);
  mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}
