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
// Up to VDN_MAXLEV levels, refinement ratio 2, the boxes of every level on any rank (the other level is seen through SrcView windows,
// exchange.hip); the levels must be properly nested
// (a level-n box keeps at least two level-(n-1) cells between itself and the edge of level n-1, or touches the domain boundary).
#include "vdn_dev.h"
#include <vector>
#include <algorithm>


static bool isect(const int alo[3], const int ahi[3], const int blo[3], const int bhi[3], Range3 &r);
// proper nesting (what initialize.f90:116-118 checks for fixed grids): every box of level n, coarsened and grown by two cells, lies
// inside the union of the boxes of level n-1 (or outside the domain) -- the coarse-fine interpolations read that far
static void check_nesting(const vdn_layout *la) {
  static std::vector<unsigned long> done;
  if (std::find(done.begin(), done.end(), la->uid) != done.end()) return;
  for (int n = 1; n < la->nlev; n++)
    for (size_t f = 0; f < la->boxes[n].size(); f++) {
      int glo[3], ghi[3]; long want = 1, have = 0;
      for (int d = 0; d < 3; d++) {
        glo[d] = std::max(la->boxes[n][f].lo[d] / 2 - 2, la->pd[n - 1].lo[d]); ghi[d] = std::min(la->boxes[n][f].hi[d] / 2 + 2, la->pd[n - 1].hi[d]);
        want *= (ghi[d] - glo[d] + 1);
      }
      for (size_t c = 0; c < la->boxes[n - 1].size(); c++) {
        Range3 r; if (isect(glo, ghi, la->boxes[n - 1][c].lo, la->boxes[n - 1][c].hi, r)) have += (long)(r.hi[0] - r.lo[0] + 1) * (r.hi[1] - r.lo[1] + 1) * (r.hi[2] - r.lo[2] + 1);
      }
      REQUIRE(have == want, "AMR path: box %d of level %d is not properly nested in level %d (two coarse cells of margin are required)", (int)f, n, n - 1);
    }
  done.push_back(la->uid);
}
static void require_amr(const vdn_layout *la) {
  REQUIRE(la->nlev >= 2 && la->nlev <= VDN_MAXLEV, "AMR path: 2..%d levels are implemented (nlevel = %d)", VDN_MAXLEV, la->nlev);
  REQUIRE(ctx().prm.dm == 3, "AMR path: dm = 3 only");
  for (size_t d = 0; d < la->rr.size(); d++) REQUIRE(la->rr[d] == 2, "AMR path: refinement ratio 2 only");
  check_nesting(la);
}
DEVI int fdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
static int hfdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
static bool isect(const int alo[3], const int ahi[3], const int blo[3], const int bhi[3], Range3 &r) {
  for (int d = 0; d < 3; d++) { r.lo[d] = std::max(alo[d], blo[d]); r.hi[d] = std::min(ahi[d], bhi[d]); if (r.lo[d] > r.hi[d]) return false; }
  return true;
}


// ---- views of the other level (several ranks: vdn_internal.h SrcView) ------------------------------------------------------------
// footprint tags keep the cached views of one multifab apart
enum { VT_REFINE = 1, VT_REFINE_FACE0 = 2, VT_COARSEN_G = 8, VT_COARSEN_1 = 20, VT_REFINE_G1 = 21, VT_COARSEN_0 = 22, VT_NODE_C2F = 23, VT_NODE_F2C = 24, VT_COARSEN_0G1 = 25 };
static const std::vector<vdn_box> &level_boxes(const vdn_multifab *mf) { return mf->la->boxes[mf->lev]; }
static const std::vector<int> &level_owner(const vdn_multifab *mf) { return mf->la->owner[mf->lev]; }
static int global_index(const vdn_multifab *mf, int li) { return mf->la->local[mf->lev][li]; }
// what the fab pointers of a multifab follow from (keys of kept descriptor sets, vdn_internal.h)
static void key_mf(GraphKey &k, const vdn_multifab *mf) {
  k.put(mf->la->uid); k.put(mf->lev); k.put(mf->nc); k.put(mf->ng); k.put(mf->nodal[0] | (mf->nodal[1] << 1) | (mf->nodal[2] << 2)); k.put((const void *)mf->base);
}
// the index region of the FINER level over the cells (faces: +1 in dir `face`, -1 = cells; nodes: face = 3) of every box of `crse_side`, grown by g fine points
static std::vector<vdn_box> refined_footprints(const vdn_multifab *crse_side, int face, int g) {
  std::vector<vdn_box> fp;
  for (const vdn_box &b : level_boxes(crse_side)) {
    vdn_box o;
    for (int d = 0; d < 3; d++) { const int extra = (face == 3 || face == d) ? 1 : 0; o.lo[d] = 2 * b.lo[d] - g; o.hi[d] = (extra ? 2 * (b.hi[d] + 1) : 2 * b.hi[d] + 1) + g; }
    fp.push_back(o);
  }
  return fp;
}
// the index region of the COARSER level under every box of `fine_side` grown by gf fine points, then grown by gc coarse points
static std::vector<vdn_box> coarsened_footprints(const vdn_multifab *fine_side, int gf, int face, int gc) {
  std::vector<vdn_box> fp;
  for (const vdn_box &b : level_boxes(fine_side)) {
    vdn_box o;
    for (int d = 0; d < 3; d++) { const int extra = (face == 3 || face == d) ? 1 : 0; o.lo[d] = hfdiv2(b.lo[d] - gf) - gc; o.hi[d] = hfdiv2(b.hi[d] + extra + gf) + extra + gc; }
    fp.push_back(o);
  }
  return fp;
}

// ---- restriction ----------------------------------------------------------------------------------------------------
// (all multi-box loops of this file go through launch_batched / BatchSet: one launch per operation and level, vdn_dev.h)
struct RestrictB { Range3 r; int g[3]; FV crse, fine; int icomp, nc, fc0;   // fine holds the components from fc0 on
  static __device__ double body(const RestrictB &a, int i, int j, int k, int) {
    for (int c = a.icomp; c < a.icomp + a.nc; c++) {
      double s = 0.0;
      #pragma unroll
      for (int kk = 0; kk < 2; kk++)
        #pragma unroll
        for (int jj = 0; jj < 2; jj++)
          #pragma unroll
          for (int ii = 0; ii < 2; ii++) s = s + fv_get(a.fine, 2 * i + ii, 2 * j + jj, 2 * k + kk, c - a.fc0);
      fv_at(a.crse, i, j, k, c) = s * 0.125;
    }
    return 0.0;
  } };
void ml_cc_restriction(vdn_multifab *crse, const vdn_multifab *fine, int icomp, int nc) {
  const SrcView F = make_view(fine, refined_footprints(crse, -1, 0), level_owner(crse), icomp, nc, VT_REFINE);
  F.refresh();
  GraphKey key; key.put(0x7201); key_mf(key, crse); key_mf(key, fine); key.put(icomp); key.put(nc);
  launch_batched_kept<RestrictB>(key.h, crse->la->uid, [&](std::vector<RestrictB> &v) {
    const BoxBins cb(crse->vbox);
    for (int f = 0; f < F.nboxes(); f++) {
      if (!F.have[f]) continue;
      int clo[3], chi[3];
      for (int d = 0; d < 3; d++) { clo[d] = hfdiv2(F.vbox[f].lo[d]); chi[d] = hfdiv2(F.vbox[f].hi[d]); }
      for (int c : cb.near(clo, chi, 1)) {
        RestrictB a;
        if (!isect(clo, chi, crse->vbox[c].lo, crse->vbox[c].hi, a.r)) continue;
        a.crse = crse->fabs[c]; a.fine = F.fv[f]; a.icomp = icomp; a.nc = nc; a.fc0 = icomp;
        v.push_back(a);
      }
    }
  }, 0, (double *)nullptr, 0, ctx().stream);
}
struct EdgeRestrictB { Range3 r; int g[3]; FV crse, fine; int dir;
  static __device__ double body(const EdgeRestrictB &a, int i, int j, int k, int) {
    const int Q[3] = { i, j, k };
    const int dir = a.dir, t1 = (dir + 1) % 3, t2 = (dir + 2) % 3;
    double s = 0.0;
    #pragma unroll
    for (int b = 0; b < 2; b++)
      #pragma unroll
      for (int aa = 0; aa < 2; aa++) {
        int q[3]; q[dir] = 2 * Q[dir]; q[t1] = 2 * Q[t1] + aa; q[t2] = 2 * Q[t2] + b;
        s = s + fv_get(a.fine, q[0], q[1], q[2]);
      }
    fv_at(a.crse, i, j, k) = s * 0.25;
    return 0.0;
  } };
// comp: the component restricted (ml_edge_restriction_c of mkflux.f90:137-146: the conservative fluxes; 0 for the one-component MAC fields)
void ml_edge_restriction(vdn_multifab *crse, const vdn_multifab *fine, int dir, int comp) {
  REQUIRE(comp >= 0 && comp < crse->nc && comp < fine->nc, "ml_edge_restriction: component %d of a %d / %d-component pair", comp, crse->nc, fine->nc);
  const SrcView F = make_view(fine, refined_footprints(crse, dir, 0), level_owner(crse), comp, 1, VT_REFINE_FACE0 + dir);
  F.refresh();
  GraphKey key; key.put(0x7202); key_mf(key, crse); key_mf(key, fine); key.put(dir); key.put(comp);
  // Two launches: the plane of faces on the HIGH side of every fine box first, everything else second.  Two fine boxes that share a plane both hold its faces, and
  // their copies need not be equal -- velpred's dead band is per box (velpred.f90:215-226), the copies can be 1e-9 apart -- so one launch over both boxes let the
  // scheduler decide which copy a coarse face under the shared plane received: runs of a three-level hierarchy with many boxes differed from process to process
  // at 1e-11 (profiles/r06_determinism.txt; found with VDN_PHASE_HASH).  Now the box on the plane's high side (whose LOW plane it is) always wins.
  for (int pass = 0; pass < 2; pass++) {
    GraphKey k2 = key; k2.put(0x51 + pass);
    launch_batched_kept<EdgeRestrictB>(k2.h, crse->la->uid, [&](std::vector<EdgeRestrictB> &v) {
    const BoxBins cb(crse->vbox);
    for (int f = 0; f < F.nboxes(); f++) {
      if (!F.have[f]) continue;
      int qlo[3], qhi[3];
      for (int d = 0; d < 3; d++) { qlo[d] = hfdiv2(F.vbox[f].lo[d]); qhi[d] = hfdiv2(F.vbox[f].hi[d]); }
      for (int c : cb.near(qlo, qhi, 2)) {
      int clo[3], chi[3], blo[3], bhi[3]; EdgeRestrictB a;
      for (int d = 0; d < 3; d++) { clo[d] = qlo[d]; chi[d] = qhi[d]; blo[d] = crse->vbox[c].lo[d]; bhi[d] = crse->vbox[c].hi[d]; }
      if (pass == 0) clo[dir] = chi[dir] = qhi[dir] + 1;       // the high plane alone
      bhi[dir] += 1;
      if (!isect(clo, chi, blo, bhi, a.r)) continue;
      a.crse = crse->fabs[c]; a.crse.p += (long)a.crse.sc * comp; a.fine = F.fv[f]; a.dir = dir;
      v.push_back(a);
      }
    }
    }, 0, (double *)nullptr, 0, ctx().stream);
  }
}

// ---- coarse -> fine ghost interpolation ---------------------------------------------------------------------------------
struct InterpArgs { int flo[3], fhi[3]; int plo[3], phi[3]; int alo[3], ahi[3]; int icomp, nc, cc0; };   // crse holds the components from cc0 on
DEVI double mc_limited(double del, double sm, double s0, double sp) {
  const double dmin = 2.0 * (s0 - sm), dpls = 2.0 * (sp - s0);
  double slim = fmin(fabs(dpls), fabs(dmin));
  slim = (dpls * dmin > 0.0) ? slim : 0.0;
  return copysign(1.0, del) * fmin(slim, fabs(del));
}
// r: fine cells (the grown fine box); a thread writes its cell if it is a ghost cell whose parent lies in [plo,phi];
// [alo,ahi]: the allocation of the coarse fab (slopes need both neighbours inside it)
struct InterpB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV fine, crse; InterpArgs A;
  static __device__ double body(const InterpB &a_, int i, int j, int k, int) {
  const FV &fine = a_.fine, &crse = a_.crse; const InterpArgs &A = a_.A;
  if (i >= A.flo[0] && i <= A.fhi[0] && j >= A.flo[1] && j <= A.fhi[1] && k >= A.flo[2] && k <= A.fhi[2]) return 0.0;
  const int q[3] = { i, j, k }, P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
  #pragma unroll
  for (int d = 0; d < 3; d++) if (P[d] < A.plo[d] || P[d] > A.phi[d]) return 0.0;
  for (int c = A.icomp; c < A.icomp + A.nc; c++) {
    const int cq = c - A.cc0;
    const double c0 = fv_get(crse, P[0], P[1], P[2], cq);
    double v = c0;
    #pragma unroll
    for (int d = 0; d < 3; d++) {
      double sl = 0.0;
      if (P[d] - 1 >= A.alo[d] && P[d] + 1 <= A.ahi[d]) {
        const double cm = fv_get(crse, P[0] - (d == 0), P[1] - (d == 1), P[2] - (d == 2), cq), cp = fv_get(crse, P[0] + (d == 0), P[1] + (d == 1), P[2] + (d == 2), cq);
        sl = mc_limited(0.5 * (cp - cm), cm, c0, cp);
      }
      const double sg = (q[d] - 2 * P[d]) ? 0.25 : -0.25;
      v = v + sg * sl;
    }
    fv_at(fine, i, j, k, c) = v;
  }
  return 0.0;
} };
// parents inside a coarse box's VALID region come from that box; parents outside the domain (physical / periodic ghost
// cells of the coarse level) come from the first coarse box whose allocation holds them.  A one-box coarse level that does not
// cover the domain: parents outside the box (its own ghost cells, filled from the next coarser level) count as "outside"
void ml_fill_ghost_cells(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc) {
  if (fine->ng == 0) return;
  // the coarse data this rank's fine boxes read: parents of the grown boxes and one more cell for the slopes
  const SrcView Cv = make_view(crse, coarsened_footprints(fine, fine->ng, -1, 1), level_owner(fine), icomp, nc, VT_COARSEN_G + fine->ng);
  Cv.refresh();
  GraphKey key; key.put(0x7203); key_mf(key, fine); key_mf(key, crse); key.put(icomp); key.put(nc);
  launch_batched_kept<InterpB>(key.h, fine->la->uid, [&](std::vector<InterpB> &v) {
  const vdn_box &pdc = Cv.nboxes() == 1 ? Cv.vbox[0] : crse->la->pd[crse->lev];
  const int pmk[3] = { Cv.nboxes() == 1 ? 0 : crse->la->pmask[0], Cv.nboxes() == 1 ? 0 : crse->la->pmask[1], Cv.nboxes() == 1 ? 0 : crse->la->pmask[2] };
  const BoxBins cb(Cv.vbox, &Cv.have);
  for (int f = 0; f < fine->nfabs(); f++) {
    InterpArgs A; Range3 r;
    int glo[3], ghi[3];
    for (int d = 0; d < 3; d++) { A.flo[d] = fine->vbox[f].lo[d]; A.fhi[d] = fine->vbox[f].hi[d]; r.lo[d] = A.flo[d] - fine->ng; r.hi[d] = A.fhi[d] + fine->ng;
      glo[d] = hfdiv2(r.lo[d]); ghi[d] = hfdiv2(r.hi[d]); }
    A.icomp = icomp; A.nc = nc; A.cc0 = icomp;
    for (int pass = 0; pass < 2; pass++)
      for (int c : cb.near(glo, ghi, Cv.ng + 1)) {
        int blo[3], bhi[3]; Range3 pr;
        for (int d = 0; d < 3; d++) { A.alo[d] = Cv.vbox[c].lo[d] - Cv.ng; A.ahi[d] = Cv.vbox[c].hi[d] + Cv.ng;
          blo[d] = pass == 0 ? Cv.vbox[c].lo[d] : A.alo[d]; bhi[d] = pass == 0 ? Cv.vbox[c].hi[d] : A.ahi[d]; }
        if (!isect(glo, ghi, blo, bhi, pr)) continue;
        if (pass == 1) {
          // only parents outside the domain; handled per direction slab to stay disjoint from pass 0.  Beyond a PERIODIC face the parents are valid cells of the
          // view's periodic images and pass 0 has them: taking them from a box's ghost cells here as well gave such fine ghost cells two writers whose limited
          // slopes differ where a stencil meets the edge of an allocation -- the scheduler chose (inputs_RayleighTaylor_2d differed from run to run)
          bool any = false;
          for (int d = 0; d < 3; d++) if (!pmk[d] && (pr.lo[d] < pdc.lo[d] || pr.hi[d] > pdc.hi[d])) any = true;
          if (!any) continue;
        }
        for (int d = 0; d < 3; d++) { A.plo[d] = pr.lo[d]; A.phi[d] = pr.hi[d]; }
        if (pass == 1) {
          // one slab per (direction, side) that sticks out of the domain
          for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
            if (pmk[d]) continue;
            InterpArgs B = A;
            if (s == 0) { if (pr.lo[d] >= pdc.lo[d]) continue; B.phi[d] = std::min(pr.hi[d], pdc.lo[d] - 1); }
            else        { if (pr.hi[d] <= pdc.hi[d]) continue; B.plo[d] = std::max(pr.lo[d], pdc.hi[d] + 1); }
            for (int e = 0; e < d; e++) { if (pmk[e]) continue; B.plo[e] = std::max(B.plo[e], pdc.lo[e]); B.phi[e] = std::min(B.phi[e], pdc.hi[e]); if (B.plo[e] > B.phi[e]) goto next; }
            { InterpB e; e.r = r; e.fine = fine->fabs[f]; e.crse = Cv.fv[c]; e.A = B; v.push_back(e); }
            next:;
          }
          break;      // the first coarse box (of those present here) whose allocation holds them
        } else
          { InterpB e; e.r = r; e.fine = fine->fabs[f]; e.crse = Cv.fv[c]; e.A = A; v.push_back(e); }
      }
  }
  }, 0, (double *)nullptr, 0, ctx().stream);       // every ghost cell has exactly one parent range: order-free
}
// fillpatch(fine, crse, ng = 0, ...) of src/regrid.f90:311-325: every VALID cell of the fine level from the coarse one, by the
// interpolation of multifab_fill_ghost_cells (the coarse ghost cells must be filled; the fine level is properly nested)
void ml_fillpatch(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc) {
  const SrcView Cv = make_view(crse, coarsened_footprints(fine, 0, -1, 1), level_owner(fine), icomp, nc, VT_COARSEN_1);
  Cv.refresh();
  std::vector<InterpB> v;
  const BoxBins cb(Cv.vbox, &Cv.have);
  for (int f = 0; f < fine->nfabs(); f++) {
    int qlo[3], qhi[3];
    for (int d = 0; d < 3; d++) { qlo[d] = hfdiv2(fine->vbox[f].lo[d]); qhi[d] = hfdiv2(fine->vbox[f].hi[d]); }
    for (int c : cb.near(qlo, qhi, 1)) {
      InterpB e; int plo[3], phi[3]; Range3 pr;
      for (int d = 0; d < 3; d++) {
        e.r.lo[d] = fine->vbox[f].lo[d]; e.r.hi[d] = fine->vbox[f].hi[d];
        e.A.flo[d] = 1; e.A.fhi[d] = 0;                                   // no cell is skipped as "valid"
        plo[d] = hfdiv2(e.r.lo[d]); phi[d] = hfdiv2(e.r.hi[d]);
        e.A.alo[d] = Cv.vbox[c].lo[d] - Cv.ng; e.A.ahi[d] = Cv.vbox[c].hi[d] + Cv.ng;
      }
      if (!isect(plo, phi, Cv.vbox[c].lo, Cv.vbox[c].hi, pr)) continue;
      for (int d = 0; d < 3; d++) { e.A.plo[d] = pr.lo[d]; e.A.phi[d] = pr.hi[d]; }
      e.A.icomp = icomp; e.A.nc = nc; e.A.cc0 = icomp; e.fine = fine->fabs[f]; e.crse = Cv.fv[c];
      v.push_back(e);
    }
  }
  launch_batched(v, 0, (double *)nullptr, 0, ctx().stream);
}
// ml_nodal_prolongation(fine, crse, rr) of src/regrid.f90:327: trilinear interpolation of a nodal field on every node of the fine level
struct NodalProlongB { Range3 r; int g[3]; FV pf, pc; int clo[3], chi[3];
  static __device__ double body(const NodalProlongB &q, int i, int j, int k, int) {
    const int I = fdiv2(i), J = fdiv2(j), K = fdiv2(k), oi = i - 2 * I, oj = j - 2 * J, ok = k - 2 * K;
    if (I < q.clo[0] || I > q.chi[0] || J < q.clo[1] || J > q.chi[1] || K < q.clo[2] || K > q.chi[2]) return 0.0;
    double s = 0.0;
    for (int c = 0; c <= ok; c++) for (int b = 0; b <= oj; b++) for (int a = 0; a <= oi; a++) s = s + fv_get(q.pc, I + a, J + b, K + c);
    fv_at(q.pf, i, j, k) = s * (1.0 / (double)((1 + oi) * (1 + oj) * (1 + ok)));
    return 0.0;
  } };
void ml_nodal_prolongation(vdn_multifab *fine, vdn_multifab *crse) {
  REQUIRE(fine->nodal[0] && fine->nodal[1] && fine->nodal[2] && crse->nodal[0] && crse->nodal[1] && crse->nodal[2], "ml_nodal_prolongation: nodal multifabs expected");
  mf_fill_boundary(crse);                                   // a parent node may sit in a coarse box's ghost layer
  const SrcView Cv = make_view(crse, coarsened_footprints(fine, 0, 3, 1), level_owner(fine), 0, 1, VT_NODE_C2F);
  Cv.refresh();
  std::vector<NodalProlongB> v;
  const BoxBins cb(Cv.vbox, &Cv.have);
  for (int f = 0; f < fine->nfabs(); f++) {
    int qlo[3], qhi[3];
    for (int d = 0; d < 3; d++) { qlo[d] = hfdiv2(fine->vbox[f].lo[d]); qhi[d] = hfdiv2(fine->vbox[f].hi[d] + 1); }
    for (int c : cb.near(qlo, qhi, 2)) {
      NodalProlongB q; bool empty = false;
      for (int d = 0; d < 3; d++) {
        q.clo[d] = Cv.vbox[c].lo[d]; q.chi[d] = Cv.vbox[c].hi[d] + 1;
        q.r.lo[d] = std::max(fine->vbox[f].lo[d], 2 * q.clo[d]); q.r.hi[d] = std::min(fine->vbox[f].hi[d] + 1, 2 * q.chi[d]);
        if (q.r.lo[d] > q.r.hi[d]) empty = true;
      }
      if (empty) continue;
      q.pf = fine->fabs[f]; q.pc = Cv.fv[c];
      v.push_back(q);
    }
  }
  launch_batched(v, 0, (double *)nullptr, 0, ctx().stream);
}
// multifab_copy_c between two multifabs of the SAME index space whose box lists differ (src/regrid.f90:333-337): valid points of
// dst that are valid points of src
struct CopyLB { Range3 r; int g[3]; FV d, s; int dc, sc, nc;
  static __device__ double body(const CopyLB &q, int i, int j, int k, int) { for (int c = 0; c < q.nc; c++) fv_at(q.d, i, j, k, q.dc + c) = fv_get(q.s, i, j, k, q.sc + c); return 0.0; } };
void mf_copy_layouts(vdn_multifab *dst, int dcomp, const vdn_multifab *src, int scomp, int nc) {
  for (int d = 0; d < 3; d++) REQUIRE(dst->nodal[d] == src->nodal[d], "copy between layouts: nodal flags differ");
  std::vector<vdn_box> fp;                                   // what every box of dst's list reads: its own valid points
  for (const vdn_box &b : level_boxes(dst)) { vdn_box o; for (int d = 0; d < 3; d++) { o.lo[d] = b.lo[d]; o.hi[d] = b.hi[d] + dst->nodal[d]; } fp.push_back(o); }
  const SrcView Sv = make_view(src, fp, level_owner(dst), scomp, nc, 40 + dst->la->uid * 64);
  Sv.refresh();
  std::vector<CopyLB> v;
  const BoxBins sb(Sv.vbox, &Sv.have);
  for (int a = 0; a < dst->nfabs(); a++)
    for (int b : sb.near(dst->vbox[a].lo, dst->vbox[a].hi, 2)) {
      CopyLB q; bool empty = false;
      for (int d = 0; d < 3; d++) {
        q.r.lo[d] = std::max(dst->vbox[a].lo[d], Sv.vbox[b].lo[d]);
        q.r.hi[d] = std::min(dst->vbox[a].hi[d], Sv.vbox[b].hi[d]) + dst->nodal[d];
        if (q.r.lo[d] > q.r.hi[d]) empty = true;
      }
      if (empty) continue;
      q.d = dst->fabs[a]; q.s = Sv.fv[b]; q.dc = dcomp; q.sc = 0; q.nc = nc;
      v.push_back(q);
    }
  launch_batched(v, 0, (double *)nullptr, 0, ctx().stream);
}
struct GrownArgs { int flo[3], fhi[3]; int plo[3], phi[3]; int dir; };
struct GrownB { Range3 r; int g[3]; FV fine, crse; GrownArgs A;
  static __device__ double body(const GrownB &a_, int i, int j, int k, int) {
  const FV &fine = a_.fine, &crse = a_.crse; const GrownArgs &A = a_.A;
  if (i >= A.flo[0] && i <= A.fhi[0] && j >= A.flo[1] && j <= A.fhi[1] && k >= A.flo[2] && k <= A.fhi[2]) return 0.0;     // valid faces
  const int q[3] = { i, j, k };
  int P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
  const int odd = q[A.dir] - 2 * P[A.dir];
  #pragma unroll
  for (int d = 0; d < 3; d++) if (P[d] < A.plo[d] || P[d] + ((d == A.dir) ? odd : 0) > A.phi[d]) return 0.0;
  const double a = fv_get(crse, P[0], P[1], P[2]);
  double v = a;
  if (odd) { P[A.dir] += 1; v = 0.5 * (a + fv_get(crse, P[0], P[1], P[2])); }
  fv_at(fine, i, j, k) = v;
  return 0.0;
} };
// parents on VALID faces of a coarse box come from that box; the remaining ones (outside the domain, or -- for a one-box coarse
// level that does not cover the domain -- in that box's own ghost faces) from the first coarse box whose allocation holds them
void ml_create_umac_grown(vdn_multifab *fine, const vdn_multifab *crse, int dir) {
  // every coarse box present here with its ghost faces (filled by fill_boundary / the level below: equal values where several hold
  // one), then the valid faces on top: two launches so that the second pass wins
  const SrcView Cv = make_view(crse, coarsened_footprints(fine, fine->ng, dir, 0), level_owner(fine), 0, 1, VT_COARSEN_0 + 4 * (dir + 1) + fine->ng * 64);
  Cv.refresh();
  GraphKey key; key.put(0x7204); key_mf(key, fine); key_mf(key, crse); key.put(dir);
  if (kept_family_enabled(2)) {
    KeptSet *k0 = kept_find(key.h), *k1 = kept_find(key.h + 1), *k2 = kept_find(key.h + 2);
    if (k0 && k1 && k2) {
      for (KeptSet *k : { k0, k1, k2 })
        if (k->nbox) hipLaunchKernelGGL((kk_batched<GrownB, int>), dim3(k->tot), dim3(64, 4, 1), 0, ctx().stream, (const GrownB *)k->d_args, (const int *)k->d_start, k->nbox, 0, (double *)nullptr);
      return;
    }
  }
  // v0: parents among a box's ghost faces; v1a / v1b: parents among the VALID faces of the coarse boxes -- the high plane of every box first, the rest second: two boxes that
  // share a plane hold copies of its faces that can be 1e-9 apart (velpred's per-box dead band), and one launch over both let the scheduler choose (profiles/r06_determinism.txt)
  std::vector<GrownB> v0, v1, v1a;
  const BoxBins cb(Cv.vbox, &Cv.have);
  int first_here = -1;
  for (int c = 0; c < Cv.nboxes() && first_here < 0; c++) if (Cv.have[c]) first_here = c;
  for (int f = 0; f < fine->nfabs(); f++) {
    GrownB e; e.A.dir = dir; e.fine = fine->fabs[f];
    Range3 rfull;
    for (int d = 0; d < 3; d++) { e.A.flo[d] = fine->vbox[f].lo[d]; e.A.fhi[d] = fine->vbox[f].hi[d] + (d == dir); rfull.lo[d] = e.A.flo[d] - fine->ng; rfull.hi[d] = e.A.fhi[d] + fine->ng; }
    // a (fine box, coarse box) pair only covers the fine faces whose parents [2 plo, 2 phi + 1] the coarse box can hold: with every pair
    // launched over the whole grown fine box a level of 997 boxes over 263 took 48 ms per call (262 000 box-sized sub-launches)
    auto clip = [&](GrownB &q) {
      for (int d = 0; d < 3; d++) { q.r.lo[d] = std::max(rfull.lo[d], 2 * q.A.plo[d]); q.r.hi[d] = std::min(rfull.hi[d], 2 * q.A.phi[d] + 1); if (q.r.lo[d] > q.r.hi[d]) return false; }
      return true;
    };
    if (first_here >= 0) {        // the first box present here, with its ghost faces
      const int c = first_here;
      e.crse = Cv.fv[c];
      for (int d = 0; d < 3; d++) { e.A.plo[d] = Cv.vbox[c].lo[d] - Cv.ng; e.A.phi[d] = Cv.vbox[c].hi[d] + (d == dir) + Cv.ng; }
      e.r = rfull;      // (kept whole: it is the one that also serves parents no box holds as valid faces)
      v0.push_back(e);
    }
    int qlo[3], qhi[3];
    for (int d = 0; d < 3; d++) { qlo[d] = hfdiv2(rfull.lo[d]); qhi[d] = hfdiv2(rfull.hi[d]); }
    for (int c : cb.near(qlo, qhi, 2)) {
      e.crse = Cv.fv[c];
      if (Cv.nboxes() > 1) {
        for (int d = 0; d < 3; d++) { e.A.plo[d] = Cv.vbox[c].lo[d]; e.A.phi[d] = Cv.vbox[c].hi[d] + (d == dir); }
        if (clip(e)) {                                                                    // without the fine faces ON the high plane (an odd face below it still reads the plane)
          e.r.hi[dir] = std::min(e.r.hi[dir], 2 * (Cv.vbox[c].hi[dir] + 1) - 1);
          if (e.r.lo[dir] <= e.r.hi[dir]) v1.push_back(e);
        }
        e.A.plo[dir] = e.A.phi[dir] = Cv.vbox[c].hi[dir] + 1;
        if (clip(e)) v1a.push_back(e);                                                    // the fine faces on the high plane alone
      }
    }
  }
  launch_batched_kept<GrownB>(key.h, fine->la->uid, [&](std::vector<GrownB> &v) { v = v0; }, 0, (double *)nullptr, 0, ctx().stream);
  launch_batched_kept<GrownB>(key.h + 1, fine->la->uid, [&](std::vector<GrownB> &v) { v = v1a; }, 0, (double *)nullptr, 0, ctx().stream);
  launch_batched_kept<GrownB>(key.h + 2, fine->la->uid, [&](std::vector<GrownB> &v) { v = v1; }, 0, (double *)nullptr, 0, ctx().stream);
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
enum { CLOSURE_ZERO = -99 };
struct ClosureB { Range3 r; int g[3]; FV phi; int bc, d, s;          // r: the boundary cells of face (d,s)
  static __device__ double body(const ClosureB &a, int i, int j, int k, int) {
    const double v = fv_get(a.phi, i, j, k);
    const int d = a.d, s = a.s;
    const int gi = i + (d == 0 ? (s ? 1 : -1) : 0), gj = j + (d == 1 ? (s ? 1 : -1) : 0), gk = k + (d == 2 ? (s ? 1 : -1) : 0);
    fv_at(a.phi, gi, gj, gk) = (a.bc == VDN_BC_NEU) ? v : (a.bc == CLOSURE_ZERO ? 0.0 : -v);
    return 0.0;
  } };
// zero: the ghost cells beyond the domain faces := 0 (what the relaxation, with the closure folded into its coefficients, wants there)
static void closure_descs(vdn_multifab *phi, const vdn_bc_tower *bct, int bc_comp0, std::vector<ClosureB> &v, bool zero = false) {
  for (int b = 0; b < phi->nfabs(); b++)
    for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
      const int e = bct->ell_bc(phi->lev, b + 1, d, s, bc_comp0);
      if (e != VDN_BC_NEU && e != VDN_BC_DIR) continue;
      ClosureB a; a.phi = phi->fabs[b]; a.bc = zero ? CLOSURE_ZERO : e; a.d = d; a.s = s;
      for (int t = 0; t < 3; t++) { a.r.lo[t] = phi->vbox[b].lo[t]; a.r.hi[t] = phi->vbox[b].hi[t]; }
      a.r.lo[d] = a.r.hi[d] = s ? phi->vbox[b].hi[d] : phi->vbox[b].lo[d];
      v.push_back(a);
    }
}
struct CfArgs { int d, s; int plo[3], phi[3]; };
struct CfB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV pf, pc; CfArgs A;                // r: the ghost cells just outside face (d,s) of the fine box
  static __device__ double body(const CfB &a, int i, int j, int k, int) {
    const CfArgs &A = a.A; const FV &pf = a.pf, &pc = a.pc;
    const int gq[3] = { i, j, k };
    const int P[3] = { fdiv2(i), fdiv2(j), fdiv2(k) };
    #pragma unroll
    for (int t = 0; t < 3; t++) if (P[t] < A.plo[t] || P[t] > A.phi[t]) return 0.0;
    const int in = A.s ? -1 : 1;
    int f1[3] = { i, j, k }, f2[3] = { i, j, k }; f1[A.d] += in; f2[A.d] += 2 * in;
    double pcs = fv_get(pc, P[0], P[1], P[2]);
    #pragma unroll
    for (int t = 0; t < 3; t++) {
      if (t == A.d) continue;
      const double sg = (gq[t] - 2 * P[t]) ? 0.125 : -0.125;
      pcs = pcs + sg * (fv_get(pc, P[0] + (t == 0), P[1] + (t == 1), P[2] + (t == 2)) - fv_get(pc, P[0] - (t == 0), P[1] - (t == 1), P[2] - (t == 2)));
    }
    fv_at(pf, i, j, k) = (8.0 / 15.0) * pcs + (2.0 / 3.0) * fv_get(pf, f1[0], f1[1], f1[2]) - 0.2 * fv_get(pf, f2[0], f2[1], f2[2]);
    return 0.0;
  } };
// ghost cells of the fine phi: coarse-fine interpolation on every face that is not a domain face, then the same-level exchange
// (which overwrites the cells that another fine box covers); domain faces were closed by the closure
// idx (optional): the (fine box, coarse view entry) of every descriptor -- a second field on the same boxes takes the list with its own pointers
static void cf_descs(vdn_multifab *pf, const SrcView &pc, const vdn_bc_tower *bct, int bc_comp0, std::vector<CfB> &v, std::vector<std::pair<int, int>> *idx = nullptr) {
  const BoxBins cb(pc.vbox, &pc.have);
  v.reserve((size_t)pf->nfabs() * 8);
  for (int f = 0; f < pf->nfabs(); f++) {
    int glo[3], ghi[3];
    for (int t = 0; t < 3; t++) { glo[t] = hfdiv2(pf->vbox[f].lo[t] - 1); ghi[t] = hfdiv2(pf->vbox[f].hi[t] + 1); }
    const std::vector<int> &cand = cb.near(glo, ghi, 1);     // one query per box: the parents of all its six ghost slabs
    for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
      const int eb = bct->ell_bc(pf->lev, f + 1, d, s, bc_comp0);
      if (eb != VDN_BC_INT && eb != VDN_BC_PER) continue;        // periodic faces too: the same-level exchange then overwrites what the level itself covers
      Range3 r; for (int t = 0; t < 3; t++) { r.lo[t] = pf->vbox[f].lo[t]; r.hi[t] = pf->vbox[f].hi[t]; }
      r.lo[d] = r.hi[d] = s ? pf->vbox[f].hi[d] + 1 : pf->vbox[f].lo[d] - 1;
      int plo[3], phi[3]; Range3 dummy;
      for (int t = 0; t < 3; t++) { plo[t] = hfdiv2(r.lo[t]); phi[t] = hfdiv2(r.hi[t]); }
      for (int c : cand) {
        // only coarse boxes that hold a parent of this slab
        if (!isect(plo, phi, pc.vbox[c].lo, pc.vbox[c].hi, dummy)) continue;
        CfB a; a.r = r; a.pf = pf->fabs[f]; a.pc = pc.fv[c]; a.A.d = d; a.A.s = s;
        for (int t = 0; t < 3; t++) { a.A.plo[t] = pc.vbox[c].lo[t]; a.A.phi[t] = pc.vbox[c].hi[t]; }
        v.push_back(a);
        if (idx) idx->push_back({ f, c });
      }
    }
  }
}
// fuse (the finest level): the pass also writes what the FIRST colour pass of the relaxation that follows makes of e = 0 and this residual --
// e = 0 + r / diag on the cells with (i + j + k) even (diag from the face coefficients folded at the domain faces as GsrbB folds them), 0 on
// the others: the same bits as GsrbB's pass from a zero-filled e, without the pass and without the zero fill
struct ResArgs { double hi2[3]; int lo[3], hi[3], e[3][2]; };
// use_rho (the finest level of the MAC projection): the face coefficients 2 / (rho_i + rho_i-1) of mk_mac_coeffs formed on the fly -- the bits of the
// stored ones, one cell array read instead of three face arrays;  use_rho == 2 (the viscous / diffusive solves: beta = mu on every face of every level,
// viscsolve.f90:58-60): the constant itself, no coefficient read at all -- the same bits again (the arrays hold exactly mu)
#define MLCC_FACE_COEFFS(a, i, j, k)                                                                                                                 \
  double bxm, bxp, bym, byp, bzm, bzp;                                                                                                               \
  if (a.use_rho == 2) { bxm = bxp = bym = byp = bzm = bzp = *a.cmu_p; }                                                                               \
  else if (a.use_rho) {                                                                                                                               \
    const double r0_ = fv_get(a.rho, i, j, k);                                                                                                       \
    bxm = 2.0 / (r0_ + fv_get(a.rho, i - 1, j, k)); bxp = 2.0 / (fv_get(a.rho, i + 1, j, k) + r0_);                                                  \
    bym = 2.0 / (r0_ + fv_get(a.rho, i, j - 1, k)); byp = 2.0 / (fv_get(a.rho, i, j + 1, k) + r0_);                                                  \
    bzm = 2.0 / (r0_ + fv_get(a.rho, i, j, k - 1)); bzp = 2.0 / (fv_get(a.rho, i, j, k + 1) + r0_);                                                  \
  } else {                                                                                                                                           \
    bxm = fv_get(a.bx, i, j, k); bxp = fv_get(a.bx, i + 1, j, k); bym = fv_get(a.by, i, j, k); byp = fv_get(a.by, i, j + 1, k);                     \
    bzm = fv_get(a.bz, i, j, k); bzp = fv_get(a.bz, i, j, k + 1);                                                                                    \
  }
struct ResidualB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV rh, phi, bx, by, bz, res, alpha, e, rho; int has_alpha, fuse, use_rho; const double *cmu_p; ResArgs A;
  static __device__ double body(const ResidualB &a, int i, int j, int k, int) {
    const FV &phi = a.phi;
    const double p0 = fv_get(phi, i, j, k);
    MLCC_FACE_COEFFS(a, i, j, k)
    const double ax = (bxp * (p0 - fv_get(phi, i + 1, j, k)) + bxm * (p0 - fv_get(phi, i - 1, j, k))) * a.A.hi2[0];
    const double ay = (byp * (p0 - fv_get(phi, i, j + 1, k)) + bym * (p0 - fv_get(phi, i, j - 1, k))) * a.A.hi2[1];
    const double az = (bzp * (p0 - fv_get(phi, i, j, k + 1)) + bzm * (p0 - fv_get(phi, i, j, k - 1))) * a.A.hi2[2];
    double Ap = ax + ay + az;
    if (a.has_alpha) Ap = Ap + fv_get(a.alpha, i, j, k) * p0;
    const double rr = fv_get(a.rh, i, j, k) - Ap;
    fv_at(a.res, i, j, k) = rr;
    if (a.fuse) {
      double ev = 0.0;
      if (((i + j + k) & 1) == 0) {
        const ResArgs &A = a.A;
        #define FOLD(b, dd, ss) { const int t = A.e[dd][ss]; if (t == VDN_BC_NEU) b = 0.0; else if (t == VDN_BC_DIR) b = 2.0 * b; }
        if (i == A.lo[0]) FOLD(bxm, 0, 0) if (i == A.hi[0]) FOLD(bxp, 0, 1)
        if (j == A.lo[1]) FOLD(bym, 1, 0) if (j == A.hi[1]) FOLD(byp, 1, 1)
        if (k == A.lo[2]) FOLD(bzm, 2, 0) if (k == A.hi[2]) FOLD(bzp, 2, 1)
        #undef FOLD
        double diag = (bxp + bxm) * A.hi2[0] + (byp + bym) * A.hi2[1] + (bzp + bzm) * A.hi2[2];
        if (a.has_alpha) diag = diag + fv_get(a.alpha, i, j, k);
        if (diag != 0.0) ev = 0.0 + rr / diag;
      }
      fv_at(a.e, i, j, k) = ev;
    }
    return fabs(rr);
  } };
struct AbsmaxB { Range3 r; int g[3]; FV a, mask; int has_mask;
  static __device__ double body(const AbsmaxB &q, int i, int j, int k, int) {
    if (q.has_mask && fv_get(q.mask, i, j, k) != 0.0) return 0.0;
    return fabs(fv_get(q.a, i, j, k));
  } };
struct RefluxArgs { int d, s; double dxf, dxc; };
// r: coarse faces (index along d fixed = the interface); the uncovered cell is on the outside of the fine box
// nocrs: the coarse field is zero (the correction of the coarser level before its relaxation): no coarse flux
struct RefluxB { Range3 r; int g[3]; FV res_c, phi_c, beta_c, mask, phi_f, beta_f; RefluxArgs A; int nocrs;
  static __device__ double body(const RefluxB &a, int i, int j, int k, int) {
    const RefluxArgs &A = a.A;
    const int Q[3] = { i, j, k };
    int M[3] = { i, j, k }; M[A.d] -= 1;
    const int *out = A.s == 0 ? M : Q;                    // the cell outside the fine box
    if (fv_get(a.mask, out[0], out[1], out[2]) != 0.0) return 0.0;   // covered by another fine box: not a coarse-fine interface
    const int t1 = (A.d + 1) % 3, t2 = (A.d + 2) % 3;
    double sum = 0.0;
    #pragma unroll
    for (int b = 0; b < 2; b++)
      #pragma unroll
      for (int aa = 0; aa < 2; aa++) {
        int q[3], m[3]; q[A.d] = 2 * Q[A.d]; q[t1] = 2 * Q[t1] + aa; q[t2] = 2 * Q[t2] + b; m[0] = q[0]; m[1] = q[1]; m[2] = q[2]; m[A.d] -= 1;
        sum = sum + fv_get(a.beta_f, q[0], q[1], q[2]) * (fv_get(a.phi_f, q[0], q[1], q[2]) - fv_get(a.phi_f, m[0], m[1], m[2])) / A.dxf;
      }
    const double Ff = sum * 0.25;
    const double Fc = a.nocrs ? 0.0 : fv_get(a.beta_c, Q[0], Q[1], Q[2]) * (fv_get(a.phi_c, Q[0], Q[1], Q[2]) - fv_get(a.phi_c, M[0], M[1], M[2])) / A.dxc;
    if (A.s == 0) fv_at(a.res_c, M[0], M[1], M[2]) = fv_get(a.res_c, M[0], M[1], M[2]) + (Ff - Fc) / A.dxc;
    else          fv_at(a.res_c, Q[0], Q[1], Q[2]) = fv_get(a.res_c, Q[0], Q[1], Q[2]) - (Ff - Fc) / A.dxc;
    return 0.0;
  } };
struct GsArgs { int lo[3], hi[3]; int e[3][2]; double hi2[3]; };
__global__ void k_set1(double *p, double v) { *p = v; }
// red-black Gauss-Seidel on the fabs of a level: ghost cells of e are 0 at the coarse-fine interface and at Dirichlet faces
// (b := 2b), Neumann faces carry b := 0 -- the folding of mg_cc.hip applied on the fly; colour by global index.
// r: lo[0] .. lo[0] + ceil(nx/2) - 1 along x (half the cells of a row), the colour picks which half
struct GsrbB { Range3 r; int g[3]; FV e, rh, bx, by, bz, alpha, rho; int has_alpha, use_rho; const double *cmu_p; GsArgs A;
  static __device__ double body(const GsrbB &a, int ih, int j, int k, int color) {
    const GsArgs &A = a.A; const FV &e = a.e;
    const int i = A.lo[0] + 2 * (ih - A.lo[0]) + ((A.lo[0] + j + k + color) & 1);
    if (i > A.hi[0]) return 0.0;
    MLCC_FACE_COEFFS(a, i, j, k)
    #define FOLD(b, dd, ss) { const int t = A.e[dd][ss]; if (t == VDN_BC_NEU) b = 0.0; else if (t == VDN_BC_DIR) b = 2.0 * b; }
    if (i == A.lo[0]) FOLD(bxm, 0, 0) if (i == A.hi[0]) FOLD(bxp, 0, 1)
    if (j == A.lo[1]) FOLD(bym, 1, 0) if (j == A.hi[1]) FOLD(byp, 1, 1)
    if (k == A.lo[2]) FOLD(bzm, 2, 0) if (k == A.hi[2]) FOLD(bzp, 2, 1)
    #undef FOLD
    const double p0 = fv_get(e, i, j, k);
    const double ax = (bxp * (p0 - fv_get(e, i + 1, j, k)) + bxm * (p0 - fv_get(e, i - 1, j, k))) * A.hi2[0];
    const double ay = (byp * (p0 - fv_get(e, i, j + 1, k)) + bym * (p0 - fv_get(e, i, j - 1, k))) * A.hi2[1];
    const double az = (bzp * (p0 - fv_get(e, i, j, k + 1)) + bzm * (p0 - fv_get(e, i, j, k - 1))) * A.hi2[2];
    double Ap = ax + ay + az;
    double diag = (bxp + bxm) * A.hi2[0] + (byp + bym) * A.hi2[1] + (bzp + bzm) * A.hi2[2];
    if (a.has_alpha) { const double a0 = fv_get(a.alpha, i, j, k); Ap = Ap + a0 * p0; diag = diag + a0; }
    if (diag != 0.0) fv_at(e, i, j, k) = p0 + (fv_get(a.rh, i, j, k) - Ap) / diag;
    return 0.0;
  } };
// inhomogeneous Dirichlet data into the right-hand side (kk_cc_load_rh of mg_cc.hip on fabs): rh += 2b * phi_ghost / h^2, faces in the
// order x-lo, x-hi, y-lo, y-hi, z-lo, z-hi
struct DirRhsB { Range3 r; int g[3]; FV rh, phi, bx, by, bz; GsArgs A;
  static __device__ double body(const DirRhsB &a, int i, int j, int k, int) {
    const GsArgs &A = a.A;
    double r = fv_get(a.rh, i, j, k);
    if (i == A.lo[0] && A.e[0][0] == VDN_BC_DIR) r = r + (2.0 * fv_get(a.bx, i, j, k)) * fv_get(a.phi, i - 1, j, k) * A.hi2[0];
    if (i == A.hi[0] && A.e[0][1] == VDN_BC_DIR) r = r + (2.0 * fv_get(a.bx, i + 1, j, k)) * fv_get(a.phi, i + 1, j, k) * A.hi2[0];
    if (j == A.lo[1] && A.e[1][0] == VDN_BC_DIR) r = r + (2.0 * fv_get(a.by, i, j, k)) * fv_get(a.phi, i, j - 1, k) * A.hi2[1];
    if (j == A.hi[1] && A.e[1][1] == VDN_BC_DIR) r = r + (2.0 * fv_get(a.by, i, j + 1, k)) * fv_get(a.phi, i, j + 1, k) * A.hi2[1];
    if (k == A.lo[2] && A.e[2][0] == VDN_BC_DIR) r = r + (2.0 * fv_get(a.bz, i, j, k)) * fv_get(a.phi, i, j, k - 1) * A.hi2[2];
    if (k == A.hi[2] && A.e[2][1] == VDN_BC_DIR) r = r + (2.0 * fv_get(a.bz, i, j, k + 1)) * fv_get(a.phi, i, j, k + 1) * A.hi2[2];
    fv_at(a.rh, i, j, k) = r;
    return 0.0;
  } };
struct AddB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV a, b;
  static __device__ double body(const AddB &q, int i, int j, int k, int) { fv_at(q.a, i, j, k) = fv_get(q.a, i, j, k) + fv_get(q.b, i, j, k); return 0.0; } };
// af += the prolonged correction of the parent level.
// lin = 0: the parent's value (piecewise constant);  lin = 1: (p0 + px + py + pz)/4 with px, py, pz the parent's neighbours on the fine cell's
// side, read through the source fab's ghost cells (the caller has put the neighbouring boxes' / periodic values there, and the parent's own
// value where the level ends) -- oracle: prolong_add in vo_amr.c
struct AddProlongB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV af, ec; int plo[3], phi[3];
  static __device__ double body(const AddProlongB &q, int i, int j, int k, int lin) {
    const int I = i / 2, J = j / 2, K = k / 2;
    if (I < q.plo[0] || I > q.phi[0] || J < q.plo[1] || J > q.phi[1] || K < q.plo[2] || K > q.phi[2]) return 0.0;
    double v = fv_get(q.ec, I, J, K);
    if (lin) {
      const double px = fv_get(q.ec, I + ((i & 1) ? 1 : -1), J, K), py = fv_get(q.ec, I, J + ((j & 1) ? 1 : -1), K), pz = fv_get(q.ec, I, J, K + ((k & 1) ? 1 : -1));
      v = 0.25 * (((v + px) + py) + pz);
    }
    fv_at(q.af, i, j, k) = fv_get(q.af, i, j, k) + v;
    return 0.0;
  } };
struct SetboxB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV a; double v;
  static __device__ double body(const SetboxB &q, int i, int j, int k, int) { fv_at(q.a, i, j, k) = q.v; return 0.0; } };

static Range3 valid_range(const vdn_multifab *mf, int b) { Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = mf->vbox[b].lo[d]; r.hi[d] = mf->vbox[b].hi[d]; } return r; }
static double read_dev(double *d) { return read_scalar1(d); }

// descriptor sets are built once per solve: the fields of a solve do not move
struct MLCC { int nlev; vdn_layout *la; bool fuse_first = false;   /* the finest level's residual pass also writes the first colour pass of its relaxation (ResidualB) */
              vdn_multifab **rh, **phi, **beta, **alpha; vdn_multifab *res[VDN_MAXLEV], *e[VDN_MAXLEV], *t[VDN_MAXLEV], *mask[VDN_MAXLEV];   // t[n] = res[n] - A_n e[n] (levels >= 1)
              const double *dx; const vdn_bc_tower *bct; int bcc; double *d_nrm; double cmu = 0.0; double *d_cmu = nullptr; /* cmu > 0: every face coefficient of every level is this constant, read from *d_cmu (it changes with dt: the kept sets do not) */ const vdn_multifab *fine_rho = nullptr;   // the finest level's density when its face coefficients are mk_mac_coeffs of it
              BatchSet<ClosureB> closure[VDN_MAXLEV]; BatchSet<CfB> cf[VDN_MAXLEV]; BatchSet<ResidualB> resid[VDN_MAXLEV];
              BatchSet<RefluxB> reflux[VDN_MAXLEV][6];          // [fine level][d*2+s]: one launch per side so that a coarse cell is updated once per launch
              BatchSet<AbsmaxB> absmax[VDN_MAXLEV]; BatchSet<GsrbB> gsrb[VDN_MAXLEV]; BatchSet<AddB> add[VDN_MAXLEV];
              BatchSet<AddProlongB> prolong[VDN_MAXLEV];               // [target level m]: e[m] += P e[m-1]
              BatchSet<RestrictB> rphi[VDN_MAXLEV], rres[VDN_MAXLEV], rres_t[VDN_MAXLEV];  // [fine level]: phi, res, t onto the level below
              BatchSet<ClosureB> edge_e[VDN_MAXLEV];                   // ghost cells of e[n] := the adjacent cell, all six faces of every box (before a linear prolongation)
              // the correction form of the iteration: ghost cells of e[n] (closure at the domain faces / zero there / interpolation beyond the interface /
              // zero everywhere), t[n] = res[n] - A_n e[n], the flux matching of res[n-1] with the fluxes of e[n]
              BatchSet<ClosureB> closure_e[VDN_MAXLEV], zero_e[VDN_MAXLEV]; BatchSet<CfB> cf_e[VDN_MAXLEV]; BatchSet<SetboxB> zero_ghost_e[VDN_MAXLEV];
              BatchSet<ResidualB> resid_e[VDN_MAXLEV]; BatchSet<RefluxB> reflux_e[VDN_MAXLEV][6];
              // the other level's fields as seen from this rank (several ranks: windows of remote boxes, refreshed before each use)
              SrcView vc_phi[VDN_MAXLEV], vf_phi[VDN_MAXLEV], vf_res[VDN_MAXLEV], vf_beta[VDN_MAXLEV][3], vc_src[VDN_MAXLEV];
              SrcView vc_e[VDN_MAXLEV], vf_e[VDN_MAXLEV], vf_t[VDN_MAXLEV];   // [fine level n]: e[n-1] under the grown boxes of level n; e[n] (with ghost cells) and t[n] over the boxes of level n-1
};
static void restrict_descs(vdn_multifab *crse, const SrcView &fine, std::vector<RestrictB> &v, std::vector<std::pair<int, int>> *idx = nullptr) {
  const BoxBins cb(crse->vbox);
  v.reserve((size_t)fine.nboxes() * 2);
  for (int f = 0; f < fine.nboxes(); f++) {
    if (!fine.have[f]) continue;
    int clo[3], chi[3];
    for (int d = 0; d < 3; d++) { clo[d] = hfdiv2(fine.vbox[f].lo[d]); chi[d] = hfdiv2(fine.vbox[f].hi[d]); }
    for (int c : cb.near(clo, chi, 1)) {
      RestrictB a;
      if (!isect(clo, chi, crse->vbox[c].lo, crse->vbox[c].hi, a.r)) continue;
      a.crse = crse->fabs[c]; a.fine = fine.fv[f]; a.icomp = 0; a.nc = 1; a.fc0 = 0;
      v.push_back(a);
      if (idx) idx->push_back({ f, c });
    }
  }
}
// a box face that is not on the domain boundary (what ell_bc == BC_INT says for a local box), for ANY box of the level
static bool face_is_interior(const vdn_layout *la, int lev, const vdn_box &b, int d, int s) {
  if (la->pmask[d]) return true;                           // a periodic direction has no physical boundary
  return s ? b.hi[d] != la->pd[lev].hi[d] : b.lo[d] != la->pd[lev].lo[d];
}
static void mlcc_build_sets(MLCC &S) {
  Prof prof_("mlcc_build_sets");
  S.d_cmu = (double *)set_alloc(256);
  hipStream_t st = ctx().stream;
  const int L = S.nlev;
  for (int n = 0; n < L; n++) {
    { std::vector<ClosureB> v; closure_descs(S.phi[n], S.bct, S.bcc, v); S.closure[n].build(v, 0, st); }
    { std::vector<ClosureB> v; closure_descs(S.e[n], S.bct, S.bcc, v); S.closure_e[n].build(v, 0, st); }
    if (n >= 1) { std::vector<ClosureB> v; closure_descs(S.e[n], S.bct, S.bcc, v, true); S.zero_e[n].build(v, 0, st); }
    if (n >= 1 && n < L - 1)               // levels that are the SOURCE of a linear prolongation (into a level >= 2)
      {
        vdn_multifab *mf = S.e[n];
        std::vector<ClosureB> v;
        for (int b = 0; b < mf->nfabs(); b++) for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) {
          ClosureB a; a.phi = mf->fabs[b]; a.bc = VDN_BC_NEU; a.d = d; a.s = sd;
          for (int t = 0; t < 3; t++) { a.r.lo[t] = mf->vbox[b].lo[t]; a.r.hi[t] = mf->vbox[b].hi[t]; }
          a.r.lo[d] = a.r.hi[d] = sd ? mf->vbox[b].hi[d] : mf->vbox[b].lo[d];
          v.push_back(a);
        }
        S.edge_e[n].build(v, 0, st);
      }
    if (n >= 1) {
      S.vc_phi[n] = make_view(S.phi[n - 1], coarsened_footprints(S.phi[n], 1, -1, 1), level_owner(S.phi[n]), 0, 1, VT_COARSEN_1);
      S.vf_phi[n] = make_view(S.phi[n], refined_footprints(S.phi[n - 1], -1, 2), level_owner(S.phi[n - 1]), 0, 1, VT_REFINE_G1);
      S.vf_res[n] = make_view(S.res[n], refined_footprints(S.res[n - 1], -1, 0), level_owner(S.res[n - 1]), 0, 1, VT_REFINE);
      for (int d = 0; d < 3; d++) { S.vf_beta[n][d] = make_view(S.beta[3 * n + d], refined_footprints(S.phi[n - 1], d, 2), level_owner(S.phi[n - 1]), 0, 1, VT_REFINE_G1); S.vf_beta[n][d].refresh(); }
      std::vector<CfB> vcf; std::vector<std::pair<int, int>> icf;
      cf_descs(S.phi[n], S.vc_phi[n], S.bct, S.bcc, vcf, &icf);
      std::vector<RestrictB> vrr; std::vector<std::pair<int, int>> irr;
      restrict_descs(S.res[n - 1], S.vf_res[n], vrr, &irr);
      S.vc_e[n] = make_view(S.e[n - 1], coarsened_footprints(S.e[n], 1, -1, 1), level_owner(S.e[n]), 0, 1, VT_COARSEN_1);
      S.vf_e[n] = make_view(S.e[n], refined_footprints(S.e[n - 1], -1, 2), level_owner(S.e[n - 1]), 0, 1, VT_REFINE_G1);
      S.vf_t[n] = make_view(S.t[n], refined_footprints(S.res[n - 1], -1, 0), level_owner(S.res[n - 1]), 0, 1, VT_REFINE);
      // the same boxes carry phi / e and res / t / phi: the lists above with the other fields' pointers (the views of one pair of levels have the same entries;
      // an entry a wider footprint adds touches no valid cell of a coarse box)
      { std::vector<CfB> v = vcf; for (size_t q = 0; q < v.size(); q++) { REQUIRE(S.vc_e[n].have[icf[q].second], "composite solve: views of one level pair differ"); v[q].pf = S.e[n]->fabs[icf[q].first]; v[q].pc = S.vc_e[n].fv[icf[q].second]; } S.cf_e[n].build(v, 0, st); }
      S.cf[n].build(vcf, 0, st);
      { std::vector<RestrictB> v = vrr; for (size_t q = 0; q < v.size(); q++) { REQUIRE(S.vf_t[n].have[irr[q].first], "composite solve: views of one level pair differ"); v[q].fine = S.vf_t[n].fv[irr[q].first]; } S.rres_t[n].build(v, 0, st); }
      { std::vector<RestrictB> v = vrr; for (size_t q = 0; q < v.size(); q++) { REQUIRE(S.vf_phi[n].have[irr[q].first], "composite solve: views of one level pair differ"); v[q].crse = S.phi[n - 1]->fabs[irr[q].second]; v[q].fine = S.vf_phi[n].fv[irr[q].first]; } S.rphi[n].build(v, 0, st); }
      S.rres[n].build(vrr, 0, st);
      { std::vector<SetboxB> v;                            // every ghost cell of e[n] := 0 (six slabs per box)
        for (int b = 0; b < S.e[n]->nfabs(); b++) for (int d = 0; d < 3; d++) for (int sd = 0; sd < 2; sd++) {
          SetboxB q; q.a = S.e[n]->fabs[b]; q.v = 0.0;
          for (int t = 0; t < 3; t++) { q.r.lo[t] = S.e[n]->vbox[b].lo[t] - 1; q.r.hi[t] = S.e[n]->vbox[b].hi[t] + 1; }
          q.r.lo[d] = q.r.hi[d] = sd ? S.e[n]->vbox[b].hi[d] + 1 : S.e[n]->vbox[b].lo[d] - 1;
          v.push_back(q);
        }
        S.zero_ghost_e[n].build(v, 0, st); }
    }
    std::vector<ResidualB> vr, vre; std::vector<AbsmaxB> va; std::vector<GsrbB> vg; std::vector<AddB> vadd;
    for (int b = 0; b < S.rh[n]->nfabs(); b++) {
      const Range3 r = valid_range(S.rh[n], b);
      ResidualB q; q.r = r; q.rh = S.rh[n]->fabs[b]; q.phi = S.phi[n]->fabs[b]; q.bx = S.beta[3 * n]->fabs[b]; q.by = S.beta[3 * n + 1]->fabs[b]; q.bz = S.beta[3 * n + 2]->fabs[b];
      q.res = S.res[n]->fabs[b]; for (int d = 0; d < 3; d++) q.A.hi2[d] = 1.0 / (S.dx[3 * n + d] * S.dx[3 * n + d]);
      q.has_alpha = S.alpha ? 1 : 0; q.alpha = S.alpha ? S.alpha[n]->fabs[b] : q.rh;
      q.use_rho = (S.fine_rho && n == L - 1) ? 1 : 0; q.rho = q.use_rho ? S.fine_rho->fabs[b] : q.rh;
      q.cmu_p = S.d_cmu; if (S.cmu > 0.0) q.use_rho = 2;
      q.fuse = (S.fuse_first && n == L - 1) ? 1 : 0; q.e = n >= 1 ? S.e[n]->fabs[b] : q.res;
      for (int d = 0; d < 3; d++) { q.A.lo[d] = r.lo[d]; q.A.hi[d] = r.hi[d]; for (int sd = 0; sd < 2; sd++) q.A.e[d][sd] = S.bct->ell_bc(n, b + 1, d, sd, S.bcc); }
      vr.push_back(q);
      if (n >= 1) { ResidualB qe = q; qe.rh = S.res[n]->fabs[b]; qe.phi = S.e[n]->fabs[b]; qe.res = S.t[n]->fabs[b]; qe.fuse = 0; vre.push_back(qe); }
      if (n < L - 1) { AbsmaxB m; m.r = r; m.a = S.res[n]->fabs[b]; m.mask = S.mask[n]->fabs[b]; m.has_mask = 1; va.push_back(m); }
      if (n >= 1) {
        GsrbB gq; gq.e = S.e[n]->fabs[b]; gq.rh = S.res[n]->fabs[b]; gq.bx = q.bx; gq.by = q.by; gq.bz = q.bz; gq.has_alpha = q.has_alpha; gq.alpha = q.alpha; gq.use_rho = q.use_rho; gq.rho = q.rho; gq.cmu_p = q.cmu_p;
        for (int d = 0; d < 3; d++) { gq.A.lo[d] = r.lo[d]; gq.A.hi[d] = r.hi[d]; gq.A.hi2[d] = q.A.hi2[d]; for (int sd = 0; sd < 2; sd++) gq.A.e[d][sd] = S.bct->ell_bc(n, b + 1, d, sd, S.bcc); }
        gq.r = r; gq.r.hi[0] = r.lo[0] + (r.hi[0] - r.lo[0] + 2) / 2 - 1;
        vg.push_back(gq);
      }
      AddB ad; ad.r = r; ad.a = S.phi[n]->fabs[b]; ad.b = S.e[n]->fabs[b]; vadd.push_back(ad);
    }
    S.resid_e[n].build(vre, 0, st);
    S.resid[n].build(vr, 0, st);          // (contiguous chunks of planes per workgroup: the k-1 / k+1 planes of phi stay in cache; its norm's atomics are rare, vdn_dev.h)
    S.absmax[n].build(va, 16, st); S.gsrb[n].build(vg, 0, st); S.add[n].build(vadd, 0, st);
    // flux matching on the cells of level n-1 next to the boxes of level n
    std::vector<std::vector<int>> cand;                     // per fine view entry: the coarse boxes around it (one query for its six faces)
    if (n >= 1) {
      const BoxBins cbins(S.phi[n - 1]->vbox);
      const SrcView &Fv = S.vf_phi[n];
      cand.resize(Fv.nboxes());
      for (int f = 0; f < Fv.nboxes(); f++) {
        if (!Fv.have[f]) continue;
        int glo[3], ghi[3];
        for (int t = 0; t < 3; t++) { glo[t] = hfdiv2(Fv.vbox[f].lo[t]); ghi[t] = hfdiv2(Fv.vbox[f].hi[t] + 1); }
        cand[f] = cbins.near(glo, ghi, 2);
      }
    }
    if (n >= 1)
      for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) {
        std::vector<RefluxB> v, ve;
        const SrcView &Fv = S.vf_phi[n];                     // entries: fine boxes and their periodic images
        for (int f = 0; f < Fv.nboxes(); f++) {
          if (!Fv.have[f] || !S.vf_beta[n][d].have[f]) continue;      // no coarse box of this rank reaches that fine box
          const vdn_box &fb = Fv.vbox[f];
          if (!face_is_interior(S.la, n, fb, d, s)) continue;
          int clo[3], chi[3];
          for (int t = 0; t < 3; t++) { clo[t] = hfdiv2(fb.lo[t]); chi[t] = hfdiv2(fb.hi[t]); }
          clo[d] = chi[d] = hfdiv2(s ? fb.hi[d] + 1 : fb.lo[d]);
          for (int c : cand[f]) {
            int blo[3], bhi[3]; RefluxB q;
            for (int t = 0; t < 3; t++) { blo[t] = S.phi[n - 1]->vbox[c].lo[t]; bhi[t] = S.phi[n - 1]->vbox[c].hi[t]; }
            if (s == 0) { blo[d] += 1; bhi[d] += 1; }          // the coarse cell that gets the correction must be a valid cell of box c
            if (!isect(clo, chi, blo, bhi, q.r)) continue;
            q.A.d = d; q.A.s = s; q.A.dxf = S.dx[3 * n + d]; q.A.dxc = S.dx[3 * (n - 1) + d];
            q.res_c = S.res[n - 1]->fabs[c]; q.phi_c = S.phi[n - 1]->fabs[c]; q.beta_c = S.beta[3 * (n - 1) + d]->fabs[c]; q.mask = S.mask[n - 1]->fabs[c];
            q.phi_f = S.vf_phi[n].fv[f]; q.beta_f = S.vf_beta[n][d].fv[f]; q.nocrs = 0;
            v.push_back(q);
            if (S.vf_e[n].have[f]) { q.phi_f = S.vf_e[n].fv[f]; q.phi_c = S.e[n - 1]->fabs[c]; q.nocrs = 1; ve.push_back(q); }
          }
        }
        S.reflux[n][2 * d + s].build(v, 0, st); S.reflux_e[n][2 * d + s].build(ve, 0, st);
      }
    // prolongation of the correction of level n-1 into e[n]
    if (n >= 1) {
      const int m = n;
      vdn_multifab *srcmf = S.e[m - 1];
      // (into the levels >= 2 the prolongation is linear and reads the parent's face neighbours: one more ring of the source level)
      if (m >= 2) S.vc_src[m] = make_view(srcmf, coarsened_footprints(S.phi[m], 0, -1, 1), level_owner(S.phi[m]), 0, 1, VT_COARSEN_0G1);
      else S.vc_src[m] = make_view(srcmf, coarsened_footprints(S.phi[m], 0, -1, 0), level_owner(S.phi[m]), 0, 1, VT_COARSEN_0);
      const SrcView &src = S.vc_src[m];
      std::vector<AddProlongB> v;
      const BoxBins sb(src.vbox, &src.have);
      for (int f = 0; f < S.e[m]->nfabs(); f++) {
        int qlo[3], qhi[3];
        for (int d = 0; d < 3; d++) { qlo[d] = S.e[m]->vbox[f].lo[d] / 2; qhi[d] = S.e[m]->vbox[f].hi[d] / 2; }
        for (int c : sb.near(qlo, qhi, 2)) {
        AddProlongB q; q.r = valid_range(S.e[m], f); q.af = S.e[m]->fabs[f]; q.ec = src.fv[c];
        for (int d = 0; d < 3; d++) { q.plo[d] = src.vbox[c].lo[d]; q.phi[d] = src.vbox[c].hi[d]; }
        int plo[3], phi[3]; Range3 dummy;
        for (int d = 0; d < 3; d++) { plo[d] = q.r.lo[d] / 2; phi[d] = q.r.hi[d] / 2; }
        if (!isect(plo, phi, q.plo, q.phi, dummy)) continue;
        v.push_back(q);
        }
      }
      S.prolong[m].build(v, 0, st);
    }
  }
}
static void fill_phi_ghosts(MLCC &S) {
  hipStream_t st = ctx().stream;
  for (int n = S.nlev - 1; n >= 1; n--) { S.vf_phi[n].refresh(); S.rphi[n].run(0, (double *)nullptr, st); }
  // (levels >= 1: the exchange follows the coarse-fine interpolation below, which reads no ghost cell of its own level -- one exchange per level)
  for (int n = 0; n < S.nlev; n++) { S.closure[n].run(0, (double *)nullptr, st); if (n == 0) mf_fill_boundary(S.phi[n], true); }
  for (int n = 1; n < S.nlev; n++) { S.vc_phi[n].refresh(); S.cf[n].run(0, (double *)nullptr, st); mf_fill_boundary(S.phi[n], true); }
}
// the composite residual on every level and its norm over the composite grid
static double composite_residual(MLCC &S) {
  hipStream_t st = ctx().stream;
  const int L = S.nlev;
  fill_phi_ghosts(S);
  HIPCHK(hipMemsetAsync(S.d_nrm, 0, sizeof(double), st));
  for (int n = 0; n < L; n++) S.resid[n].run(0, n == L - 1 ? S.d_nrm : (double *)nullptr, st);
  // flux matching: lo faces then hi faces of every direction (one update per coarse cell and launch, hence deterministic)
  for (int n = 1; n < L; n++) { S.vf_phi[n].refresh(); for (int ds = 0; ds < 6; ds++) S.reflux[n][ds].run(0, (double *)nullptr, st); }     // fine phi incl. its ghost cells
  for (int n = L - 1; n >= 1; n--) { S.vf_res[n].refresh(); S.rres[n].run(0, (double *)nullptr, st); }
  for (int n = 0; n < L - 1; n++) S.absmax[n].run(0, S.d_nrm, st);
  comm_allreduce_max_dev(S.d_nrm, 1);
  return read_dev(S.d_nrm);
}
// nsweeps red-black sweeps of A_n e = res_n on the e at hand; its ghost cells beyond the interface are data (zero on the way down, interpolated from
// the coarser correction on the way up), those beyond the domain faces zero (the closure is folded into the coefficients), the neighbouring boxes'
// are exchanged before every pass but the first (ghosts_current: the caller has just filled them; otherwise e = 0 with ghost cells)
static void level_relax(MLCC &S, int n, int nsweeps, bool first_done) {
  vdn_multifab *e = S.e[n];
  const bool exchange = level_boxes(e).size() > 1 || S.la->pmask[0] || S.la->pmask[1] || S.la->pmask[2];     // boxes of the level anywhere, not just here: every rank must take part
  for (int s = 0; s < nsweeps; s++) for (int col = 0; col < 2; col++) {
    if (first_done && s == 0 && col == 0) continue;     // (the finest level's residual pass has written the first colour pass from e = 0)
    if (exchange && (s > 0 || col > 0)) mf_fill_boundary(e, true);
    S.gsrb[n].run(col, (double *)nullptr, ctx().stream);
  }
}
// ghost cells of e[n] as the operator of the composite residual reads them: closure at the domain faces, interpolation from e[n-1] beyond the
// interface, the neighbouring boxes' values (fill_phi_ghosts for one level of the correction)
static void fill_e_ghosts(MLCC &S, int n) {
  hipStream_t st = ctx().stream;
  S.closure_e[n].run(0, (double *)nullptr, st);
  if (n >= 1) { S.vc_e[n].refresh(); S.cf_e[n].run(0, (double *)nullptr, st); }
  mf_fill_boundary(S.e[n], true);
}
// the sets of a solve kept across solves (vdn_internal.h: kept descriptor sets): the MLCC of the solve that built them, reused with the per-call fields replaced
struct MLCCKept { KeeperMem mem; unsigned long uid = 0; MLCC S; };
static std::map<unsigned long long, MLCCKept *> g_mlcc_kept;
void mlcc_kept_purge(unsigned long uid) {
  for (auto it = g_mlcc_kept.begin(); it != g_mlcc_kept.end();) {
    if (uid == 0 || it->second->uid == uid) { keeper_free(&it->second->mem); delete it->second; it = g_mlcc_kept.erase(it); } else ++it;
  }
}
// ---- a singular composite system (no Dirichlet face, no alpha term) needs a right-hand side whose volume-weighted sum over the composite grid is zero.  The MAC
// projection's is div(umac), a telescoping sum -- but the reference's velpred upwinds with a dead band eps = 1e-8 * (the BOX's own max |u|) (velpred.f90:215-226,
// 1965-1980: decomposition-dependent by design), so two boxes can give their shared face values that differ by O(1e-9) where the normal velocity passes through zero -- a
// periodic face on a symmetry plane of inputs_RayleighTaylor_2d -- and the residual then cannot fall below that defect (it sat at 1.3e-9 for 100 iterations, bit for
// bit).  FBoxLib's solvers deal with singular systems themselves; here: when the mean defect exceeds 1e-11 of the right-hand side's norm it is subtracted (below that the
// solve converges anyway and nothing is touched: every run that converged before keeps its bits).  The sums are deterministic: one workgroup per box with a fixed tree,
// the boxes added on the host in order, ranks in rank order.
struct SumD { FV a, mask; Range3 r; int has_mask; };
__global__ void __launch_bounds__(256) kk_box_sums(const SumD *descs, double *partial) {
  const SumD &D = descs[blockIdx.x];
  const int nx = D.r.hi[0] - D.r.lo[0] + 1, ny = D.r.hi[1] - D.r.lo[1] + 1, nz = D.r.hi[2] - D.r.lo[2] + 1;
  const long tot = (long)nx * ny * nz;
  double acc = 0.0;
  for (long t = threadIdx.x; t < tot; t += 256) {
    const int i = D.r.lo[0] + (int)(t % nx), j = D.r.lo[1] + (int)((t / nx) % ny), k = D.r.lo[2] + (int)(t / ((long)nx * ny));
    if (D.has_mask && fv_get(D.mask, i, j, k) != 0.0) continue;
    acc = acc + fv_get(D.a, i, j, k);
  }
  __shared__ double sh[256];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
struct add_const_K { FV a; double v; __device__ void cell(int i, int j, int k) const { fv_at(a, i, j, k) = fv_get(a, i, j, k) + v; } };
// volume-weighted mean of rh over the cells of the composite grid (mask[n] != 0: covered by level n + 1), all ranks
static double composite_mean(int L, vdn_multifab **rh, vdn_multifab **mask, const double *dx, const vdn_layout *la) {
  hipStream_t st = ctx().stream;
  double local = 0.0;
  for (int n = 0; n < L; n++) {
    const int nb = rh[n]->nfabs();
    if (nb == 0) continue;
    std::vector<SumD> v;                                   // a descriptor (= a workgroup) per slab of at most ~64 K cells of a box: the one 256^3 box of a base level is 256 of them
    for (int b = 0; b < nb; b++) {
      SumD q; q.a = rh[n]->fabs[b]; q.has_mask = (n < L - 1 && mask[n]) ? 1 : 0; q.mask = q.has_mask ? mask[n]->fabs[b] : rh[n]->fabs[b]; q.r = valid_range(rh[n], b);
      const long plane = (long)(q.r.hi[0] - q.r.lo[0] + 1) * (q.r.hi[1] - q.r.lo[1] + 1);
      const int kper = (int)std::max<long>(1, 65536 / plane), k0 = q.r.lo[2], k1 = q.r.hi[2];
      for (int k = k0; k <= k1; k += kper) { SumD c = q; c.r.lo[2] = k; c.r.hi[2] = std::min(k + kper - 1, k1); v.push_back(c); }
    }
    const int nd = (int)v.size();
    const size_t mark = arena_mark();
    SumD *d_desc = (SumD *)arena_alloc(sizeof(SumD) * nd);
    double *d_part = (double *)arena_alloc(sizeof(double) * nd);
    upload_staged(d_desc, v.data(), sizeof(SumD) * nd);
    hipLaunchKernelGGL(kk_box_sums, dim3(nd), dim3(256), 0, st, (const SumD *)d_desc, d_part);
    std::vector<double> h(nd);
    HIPCHK(hipMemcpyAsync(h.data(), d_part, sizeof(double) * nd, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    arena_release(mark);
    double lev = 0.0;
    for (int b = 0; b < nd; b++) lev = lev + h[b];
    local = local + lev * (dx[3 * n] * dx[3 * n + 1] * dx[3 * n + 2]);
  }
  double total = local;
  if (comm_active()) {
    const size_t mark = arena_mark();
    const int nr = ctx().nranks;
    double *d_all = (double *)arena_alloc(sizeof(double) * (nr + 1));
    HIPCHK(hipMemcpyAsync(d_all + nr, &local, sizeof(double), hipMemcpyHostToDevice, st));
    comm_allgather_dev(d_all + nr, d_all, 1);
    std::vector<double> h(nr);
    HIPCHK(hipMemcpyAsync(h.data(), d_all, sizeof(double) * nr, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    arena_release(mark);
    total = 0.0;
    for (int r = 0; r < nr; r++) total = total + h[r];
  }
  double vol = dx[0] * dx[1] * dx[2];
  for (int d = 0; d < 3; d++) vol = vol * (double)(la->pd[0].hi[d] - la->pd[0].lo[d] + 1);
  return total / vol;
}
static void add_constant(vdn_multifab *mf, double v) {
  std::vector<std::pair<add_const_K, Range3>> w;
  for (int b = 0; b < mf->nfabs(); b++) w.push_back({ add_const_K{ mf->fabs[b], v }, valid_range(mf, b) });
  launch_cells(w, ctx().stream);
}

// rh, phi: [lev];  beta: [lev*3 + d];  dx: [lev*3 + d]
// alpha: [lev] cell coefficients of (alpha - div beta grad), or nullptr.  The ghost cells of the incoming phi carry inhomogeneous
// Dirichlet data (boundary-face values); they are moved into rh, which is modified
int ml_cc_solve(vdn_layout *la, vdn_multifab **rh, vdn_multifab **phi, vdn_multifab **beta, const double *dx, const vdn_bc_tower *bct, int bc_comp0,
                double rel_eps, int max_iter, int *iters, double *res0, double *res, vdn_multifab **alpha, vdn_multifab **base_beta, const vdn_multifab *base_rho, const vdn_multifab *fine_rho,
                double const_beta) {
  require_amr(la);
  hipStream_t st = ctx().stream;
  const size_t mark = arena_mark();
  const int L = la->nlev;
  static const bool fuse_first_on = !(vdn_env("VDN_MLCC_FUSE1") && atoi(vdn_env("VDN_MLCC_FUSE1")) == 0);
  static const bool rho_form = !(vdn_env("VDN_MLCC_RHO") && atoi(vdn_env("VDN_MLCC_RHO")) == 0);
  const bool fuse_first = fuse_first_on && ctx().prm.mg_nu1 >= 1 && ctx().prm.mg_nu2 >= 1;
  const vdn_multifab *frho = (rho_form && !alpha) ? fine_rho : nullptr;
  if (frho) REQUIRE(frho->ng >= 1 && frho->lev == L - 1, "composite solve: the finest level's density with a filled ghost cell expected");
  // everything the descriptor sets of this solve follow from: the key of the kept ones (vdn_internal.h)
  GraphKey key; key.put(0x7301); key.put(la->uid); key.put(L); key.put(bct->serial); key.put(bc_comp0); key.put(fuse_first); key.put((const void *)(frho ? frho->base : nullptr)); key.put(rho_form && const_beta > 0.0);
  for (int n = 0; n < L; n++) {
    key_mf(key, rh[n]); key_mf(key, phi[n]); for (int d = 0; d < 3; d++) { key_mf(key, beta[3 * n + d]); key.put(dx[3 * n + d]); }
    key.put(alpha != nullptr); if (alpha) key_mf(key, alpha[n]);
  }
  for (int n = 0; n < L; n++) {
    GraphKey k2 = key; k2.put(0x11); k2.put(n);
    launch_batched_kept<DirRhsB>(k2.h, la->uid, [&](std::vector<DirRhsB> &v) {
      for (int b = 0; b < rh[n]->nfabs(); b++) {
        DirRhsB q; q.rh = rh[n]->fabs[b]; q.phi = phi[n]->fabs[b]; q.bx = beta[3 * n]->fabs[b]; q.by = beta[3 * n + 1]->fabs[b]; q.bz = beta[3 * n + 2]->fabs[b];
        bool any = false;
        for (int d = 0; d < 3; d++) { q.r.lo[d] = q.A.lo[d] = rh[n]->vbox[b].lo[d]; q.r.hi[d] = q.A.hi[d] = rh[n]->vbox[b].hi[d]; q.A.hi2[d] = 1.0 / (dx[3 * n + d] * dx[3 * n + d]);
          for (int sd = 0; sd < 2; sd++) { q.A.e[d][sd] = bct->ell_bc(n, b + 1, d, sd, bc_comp0); any = any || q.A.e[d][sd] == VDN_BC_DIR; } }
        if (any) v.push_back(q);
      }
    }, 0, (double *)nullptr, 0, ctx().stream);
  }
  double *d_nrm = (double *)arena_alloc(256);
  vdn_multifab *t_res[VDN_MAXLEV], *t_e[VDN_MAXLEV], *t_t[VDN_MAXLEV], *t_mask[VDN_MAXLEV];
  for (int n = 0; n < L; n++) {
    t_res[n] = mf_temp(la, n, 1, 0, -1, true, 0.0); t_e[n] = mf_temp(la, n, 1, 1, -1, true, 0.0);
    t_t[n] = n >= 1 ? mf_temp(la, n, 1, 0, -1, true, 0.0) : nullptr;
    t_mask[n] = nullptr;
    key.put((const void *)t_res[n]->base); key.put((const void *)t_e[n]->base); key.put((const void *)(t_t[n] ? t_t[n]->base : nullptr));
    if (n < L - 1) {                                         // cells of level n covered by level n+1
      t_mask[n] = mf_temp(la, n, 1, 0, -1, true, 0.0);
      key.put((const void *)t_mask[n]->base);
      GraphKey k2 = key; k2.put(0x12);
      launch_batched_kept<SetboxB>(k2.h, la->uid, [&](std::vector<SetboxB> &v) {
        const std::vector<vdn_box> &fb = level_boxes(phi[n + 1]);
        const BoxBins mb(t_mask[n]->vbox);
        for (int f = 0; f < (int)fb.size(); f++) {
          int clo[3], chi[3];
          for (int d = 0; d < 3; d++) { clo[d] = fb[f].lo[d] / 2; chi[d] = fb[f].hi[d] / 2; }
          for (int c : mb.near(clo, chi, 1)) {
            SetboxB q;
            if (!isect(clo, chi, t_mask[n]->vbox[c].lo, t_mask[n]->vbox[c].hi, q.r)) continue;
            q.a = t_mask[n]->fabs[c]; q.v = 1.0; v.push_back(q);
          }
        }
      }, 0, (double *)nullptr, 0, st);
    }
  }
  // the sets of the iteration: kept across solves under the key (a hit skips every pair loop and upload), else built into the arena
  MLCC S_local; MLCC *Sp = &S_local; bool hit = false; MLCCKept *kept = nullptr;
  if (kept_family_enabled(4)) {
    auto itk = g_mlcc_kept.find(key.h);
    if (itk != g_mlcc_kept.end()) { kept = itk->second; hit = true; }
    else { if ((int)g_mlcc_kept.size() >= kept_bound(64)) { HIPCHK(hipStreamSynchronize(st)); mlcc_kept_purge(0); } kept = new MLCCKept; kept->uid = la->uid; g_mlcc_kept[key.h] = kept; }
    Sp = &kept->S;
  }
  MLCC &S = *Sp;
  S.nlev = L; S.la = la; S.rh = rh; S.phi = phi; S.beta = beta; S.alpha = alpha; S.dx = dx; S.bct = bct; S.bcc = bc_comp0; S.fuse_first = fuse_first; S.fine_rho = frho; S.d_nrm = d_nrm; S.cmu = rho_form ? const_beta : 0.0;
  for (int n = 0; n < L; n++) { S.res[n] = t_res[n]; S.e[n] = t_e[n]; S.t[n] = t_t[n]; S.mask[n] = t_mask[n]; }
  if (!hit) {
    if (kept) keeper_begin(&kept->mem);
    try { mlcc_build_sets(S); }
    catch (...) { if (kept) { keeper_end(); HIPCHK(hipStreamSynchronize(st)); keeper_free(&kept->mem); g_mlcc_kept.erase(key.h); delete kept; } throw; }
    if (kept) keeper_end();
  } else
    for (int n = 1; n < L; n++) for (int d = 0; d < 3; d++) S.vf_beta[n][d].refresh();      // (the coefficients change from solve to solve: the windows of remote boxes)
  if (S.cmu > 0.0) hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, S.d_cmu, S.cmu);
  HIPCHK(hipMemsetAsync(S.d_nrm, 0, sizeof(double), st));
  for (int n = 0; n < L; n++) {
    GraphKey k2 = key; k2.put(0x13); k2.put(n);
    launch_batched_kept<AbsmaxB>(k2.h, la->uid, [&](std::vector<AbsmaxB> &v) {
      for (int b = 0; b < rh[n]->nfabs(); b++) { AbsmaxB q; q.r = valid_range(rh[n], b); q.a = rh[n]->fabs[b]; q.mask = n < L - 1 ? S.mask[n]->fabs[b] : rh[n]->fabs[b]; q.has_mask = n < L - 1 ? 1 : 0; v.push_back(q); }
    }, 0, S.d_nrm, 16, st);
  }
  comm_allreduce_max_dev(S.d_nrm, 1);
  const double bnorm = read_dev(S.d_nrm);
  const vdn_params &P = ctx().prm;
  {   // solvability of a singular system (see composite_mean)
    bool singular = !alpha;
    for (int d = 0; d < 3 && singular; d++) for (int sd = 0; sd < 2; sd++) if (bct->ell_bc(0, 0, d, sd, bc_comp0) == VDN_BC_DIR) singular = false;
    if (singular && bnorm > 0.0 && bnorm < HUGE_VAL) {
      const double mean = composite_mean(L, rh, S.mask, dx, la);
      if (fabs(mean) > 1.e-11 * bnorm) for (int n = 0; n < L; n++) add_constant(rh[n], -mean);
    }
  }
  int it = 0; bool conv = (bnorm == 0.0); double rn = 0.0;
  struct KeepGuard { CcKeep *k; ~KeepGuard() { cc_keep_free(k); } } keep_guard{ cc_keep_new() };      // freed on every exit, exceptions included
  CcKeep *coarse_keep = keep_guard.k;        // the level-0 multigrid hierarchy is built once for all FAC iterations
  int ebc0[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc0[d][s] = bct->ell_bc(0, 0, d, s, bc_comp0);
  while (!conv) {
    rn = composite_residual(S);
    { static const bool trace = vdn_env("VDN_MLCC_TRACE") && atoi(vdn_env("VDN_MLCC_TRACE")) != 0;
      if (trace) fprintf(stderr, "  ml_cc_solve: iteration %d, composite residual %.6e (right-hand side %.6e, target %.3e)\n", it, rn, bnorm, rel_eps * bnorm); }
    if (rn <= rel_eps * bnorm && bnorm < HUGE_VAL) { conv = true; break; }
    if (it >= max_iter || !(rn < HUGE_VAL) || !(bnorm < HUGE_VAL)) break;
    // one V-cycle over the levels in correction form (oracle: vo_ml_cc_solve).  Down, finest level first: e_n = 0, nu1 sweeps, t = res_n - A_n e_n,
    // res_{n-1} := restriction of t under level n and the flux matching with e_n's fluxes next to it
    for (int n = 1; n < L; n++) {
      if (S.fuse_first && n == L - 1) S.zero_ghost_e[n].run(0, (double *)nullptr, st);      // (its cells hold the first colour pass already)
      else mf_setval(S.e[n], 0.0, 0, 1, true);
    }
    mf_setval(S.e[0], 0.0, 0, 1, true);
    for (int n = L - 1; n >= 1; n--) {
      level_relax(S, n, P.mg_nu1, S.fuse_first && n == L - 1);
      fill_e_ghosts(S, n);                                 // (e[n-1] = 0 here)
      S.resid_e[n].run(0, (double *)nullptr, st);
      S.vf_e[n].refresh(); for (int ds = 0; ds < 6; ds++) S.reflux_e[n][ds].run(0, (double *)nullptr, st);
      S.vf_t[n].refresh(); S.rres_t[n].run(0, (double *)nullptr, st);
    }
    // coarse correction: ONE V-cycle of the single-level multigrid on the whole level 0
    static const bool glue = !(vdn_env("VDN_MLCC_GLUE") && atoi(vdn_env("VDN_MLCC_GLUE")) == 0);
    const bool zg = glue && it > 0;           // (the first call builds the kept hierarchy and loads phi as the generic solver does)
    int cyc; double r0, rr;
    // (base_beta / base_rho, the MAC projection: the V-cycle runs on level 0's OWN coefficients 2/(rho_i + rho_i-1) -- `beta` carries the edge
    // restriction of the finer level's on the covered faces -- and so on the density-based kernels of the single-level solver; it is a
    // preconditioner, the composite residual above is formed with `beta`.  Oracle: beta_base of vo_ml_cc_solve)
    cc_solve(S.res[0], S.e[0], base_beta ? base_beta : beta, dx, ebc0, 0.0, -1.0, -1, &cyc, &r0, &rr, alpha ? alpha[0] : nullptr, base_beta ? base_rho : nullptr, coarse_keep, nullptr, 0, zg, glue ? S.phi[0] : nullptr, const_beta);     // (no nested-iteration start here: it saves no FAC iteration, measured)
    if (!glue) S.add[0].run(0, (double *)nullptr, st);     // (glue: phi_0 += e_0 was done where e_0 was stored)
    fill_e_ghosts(S, 0);
    // up, coarsest level first: e_n += P e_{n-1}, the interface ghost cells from e_{n-1}, nu2 sweeps, phi_n += e_n
    for (int n = 1; n < L; n++) {
      const int lin = n >= 2 ? 1 : 0;                     // piecewise constant into level 1, linear into the finer ones (oracle: prolong_add)
      if (lin) { S.edge_e[n - 1].run(0, (double *)nullptr, st); mf_fill_boundary(S.e[n - 1], true); }     // the source's ghost cells: the cell itself where the level ends, then the neighbouring boxes' / periodic values
      S.vc_src[n].refresh();
      S.prolong[n].run(lin, (double *)nullptr, st);
      if (lin) fill_e_ghosts(S, n - 1);                   // (back to what the interface interpolation reads)
      S.zero_e[n].run(0, (double *)nullptr, st);
      S.vc_e[n].refresh(); S.cf_e[n].run(0, (double *)nullptr, st);
      mf_fill_boundary(S.e[n], true);
      level_relax(S, n, P.mg_nu2, false);
      S.add[n].run(0, (double *)nullptr, st);
    }
    it++;
  }
  fill_phi_ghosts(S);
  if (iters) *iters = it; if (res0) *res0 = bnorm; if (res) *res = rn;
  for (int n = L - 1; n >= 0; n--) { if (S.mask[n]) mf_temp_free(S.mask[n]); if (S.t[n]) mf_temp_free(S.t[n]); mf_temp_free(S.e[n]); mf_temp_free(S.res[n]); }
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
  vdn_multifab *beta0[3];                    // level 0's own coefficients (the covered faces not overwritten): what the coarse correction's V-cycle runs on
  for (int d = 0; d < 3; d++) beta0[d] = mf_temp(mla, 0, 1, 0, d, false, 0.0);
  mac_level_coeffs(rho[0], beta0);
  // ... only where they agree with the edge-restricted ones to 25 % on every face (oracle/vo_amr.c, vo_ml_macproject: on averaged-down data the own
  // coefficients are 1 / (mean rho), the restricted ones a mean of 1 / rho; across a sharp density jump the softer operator makes the correction
  // overshoot -- 45 FAC iterations instead of 12 at a one-cell jump of 10 : 1, divergence at 100 : 1).  One reduction and read-back per solve.
  const bool own = mf_max_ratio3(beta, beta0) <= 1.25;
  int rc = ml_cc_solve(mla, rh, phi, beta, dx, bct, bc_comp0, ctx().prm.mac_rel_eps, ctx().prm.mg_max_iter, &it, &r0, &rr, nullptr, own ? beta0 : nullptr, own ? rho[0] : nullptr, rho[L - 1]);
  for (int d = 2; d >= 0; d--) mf_temp_free(beta0[d]);
  ctx().solver_cycles[0] = it; ctx().solver_res0[0] = r0; ctx().solver_res[0] = rr;
  solver_check(rc, "composite MAC solve", it, rr, r0);
  for (int n = 0; n < L; n++) mac_level_mkumac(umac + 3 * n, phi[n], beta + 3 * n, dx + 3 * n, bct, bc_comp0);   // 103
  for (int n = L - 1; n >= 1; n--) for (int d = 0; d < 3; d++) ml_edge_restriction(umac[3 * (n - 1) + d], umac[3 * n + d], d);       // 497-500
  for (int d = 0; d < 3; d++) mf_fill_boundary(umac[d]);
  for (int n = 1; n < L; n++) for (int d = 0; d < 3; d++) { ml_create_umac_grown(umac[3 * n + d], umac[3 * (n - 1) + d], d); mf_fill_boundary(umac[3 * n + d]); }   // 107-119
  for (int n = L - 1; n >= 0; n--) { for (int d = 2; d >= 0; d--) mf_temp_free(beta[3 * n + d]); mf_temp_free(phi[n]); mf_temp_free(rh[n]); }
  arena_release(mark);
}

// ---- C-ABI ------------------------------------------------------------------------------------------------------------------
extern "C" int vdn_ml_cc_restriction(vdn_multifab *crse, const vdn_multifab *fine, int icomp, int nc) { VDN_TRY ml_cc_restriction(crse, fine, icomp, nc); VDN_CATCH }
extern "C" int vdn_ml_edge_restriction(vdn_multifab *crse, const vdn_multifab *fine, int dir) { VDN_TRY ml_edge_restriction(crse, fine, dir, 0); VDN_CATCH }
extern "C" int vdn_multifab_fill_ghost_cells(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc) { VDN_TRY ml_fill_ghost_cells(fine, crse, icomp, nc); VDN_CATCH }
extern "C" int vdn_create_umac_grown(vdn_multifab *fine, const vdn_multifab *crse, int dir) { VDN_TRY ml_create_umac_grown(fine, crse, dir); VDN_CATCH }
extern "C" int vdn_fillpatch(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc) { VDN_TRY ml_fillpatch(fine, crse, icomp, nc); VDN_CATCH }
extern "C" int vdn_ml_nodal_prolongation(vdn_multifab *fine, vdn_multifab *crse) { VDN_TRY ml_nodal_prolongation(fine, crse); VDN_CATCH }
extern "C" int vdn_multifab_copy_layouts(vdn_multifab *dst, int dcomp, const vdn_multifab *src, int scomp, int nc) {
  VDN_TRY
  REQUIRE(dcomp >= 0 && dcomp + nc <= dst->nc && scomp >= 0 && scomp + nc <= src->nc, "copy between layouts: component range");
  mf_copy_layouts(dst, dcomp, src, scomp, nc);
  VDN_CATCH
}
extern "C" int vdn_ml_restrict_and_fill(int nlev, vdn_multifab **mf, int icomp, int bcomp, int nc, int same_boundary, const vdn_bc_tower *bct) {
  VDN_TRY ml_restrict_and_fill(nlev, mf, icomp, bcomp, nc, same_boundary != 0, bct); VDN_CATCH
}
