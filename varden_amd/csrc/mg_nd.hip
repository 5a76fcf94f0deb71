// mg_nd.hip -- nodal (HG) projection (reference src/hgproject.f90:17-178, src/hg_multigrid.f90:18-119)
// and the nodal multigrid that replaces FBoxLib's ml_nd_solve (hg_multigrid.f90:95-105).
//
// Discrete system and algorithm: see oracle/vo_hgproject.c (same definitions, same expression order):
// Q1 finite-element stiffness with cell-constant sigma scaled by 1/(hx hy hz) ("dense" 27-point
// stencil, 21-point when dx=dy=dz), RHS = nodal divergence, outflow nodes phi = 0, walls natural.
// V(nu1,nu2) cycles, damped Jacobi (one streaming pass per sweep -- a lexicographic Gauss-Seidel, as
// a CPU code would use, has no parallel order; an 8-colour one costs 8 passes over the level),
// full-weighting restriction, trilinear prolongation, coarse sigma = mean of the 8 children.
//
// HBM layout of a level: node (i,j,k) in [-1..n+1] and cell (i,j,k) in [-1..n] both map to
// (i + 16) + PX*((j+1) + PY*(k+1)), PX a multiple of 16 doubles (rows start on a 128-byte line).
// Algorithmic traffic of one Jacobi sweep: phi read 8 + phi write 8 + rhs 8 + sigma 8 = 32 B/node.
#include "vdn_dev.h"
#include <tuple>
#include <algorithm>

void cc_halo_cache_purge(unsigned long uid);
void nd_halo_cache_purge(unsigned long uid);
void mg_halo_cache_purge(unsigned long uid) { cc_halo_cache_purge(uid); nd_halo_cache_purge(uid); }

struct NLev {
  int n[3]; int PX, PY; long sz;
  double f[3];                    // 1/(36 h^2)
  double *phi, *tmp, *b, *res, *sig;
  int dirlo[3], dirhi[3];         // Dirichlet (outflow) faces
  int per[3];
};
DEVI long nidx(const NLev &L, int i, int j, int k) { return (long)(i + 16) + (long)L.PX * ((long)(j + 1) + (long)L.PY * (long)(k + 1)); }
DEVI bool nd_is_dir(const NLev &L, int i, int j, int k) {
  return (i == 0 && L.dirlo[0]) || (i == L.n[0] && L.dirhi[0]) || (j == 0 && L.dirlo[1]) || (j == L.n[1] && L.dirhi[1]) ||
         (k == 0 && L.dirlo[2]) || (k == L.n[2] && L.dirhi[2]);
}
// nodes ON the faces of the box: the only ones whose stencil reaches ghost nodes
// `hm`: the faces of the box whose ghost nodes come from the exchange (bit 2d + side; the ghost nodes outside a physical face stay zero)
DEVI bool nd_is_shell(const NLev &L, int i, int j, int k, int hm) {
  return ((hm & 1) && i == 0) || ((hm & 2) && i == L.n[0]) || ((hm & 4) && j == 0) || ((hm & 8) && j == L.n[1]) ||
         ((hm & 16) && k == 0) || ((hm & 32) && k == L.n[2]);
}

// ---- the 27-point nodal operator -----------------------------------------------------------------------------------------------------
// K phi = sum over the 8 cells c around the node of sigma_c * sum over the cell's 8 corners q of w[type(q)] phi_q, type = which
// coordinates of q differ from the node's (bit 0 x, 1 y, 2 z); trilinear (Q1) elements, equations scaled by 1/(hx hy hz).
// Round 2: evaluated GROUPED BY NEIGHBOUR TYPE -- a face / edge / corner neighbour is shared by 4 / 2 / 1 of the cells, so its
// coefficient is w[type] times the sum of those sigmas: 41 (hx = hy = hz: the face weights are exactly zero and their terms are
// skipped, the 21-point stencil of hg_hypre.f90:100-113) to 55 f64 operations with explicit fma, instead of the 142 of the
// cell-by-cell accumulation of round 1, which kept the sweeps VALU-bound (249 VALU instructions per node, VALU busy 75 %).
// Same operation sequence as vo_nd_stencil in oracle/vo_hgproject.c, hence the same bits.
// p[oc][ob][oa] = phi at node offset (oa-1, ob-1, oc-1); sg[dk][dj][di] = sigma of cell (i-1+di, j-1+dj, k-1+dk)
struct NdW { double w0, w1, w2, w3, w4, w5, w6, w7; int iso; };
DEVI NdW nd_weights(const double f[3]) {
  const double fx = f[0], fy = f[1], fz = f[2];
  const double F = fx + fy + fz;
  NdW W;
  W.w0 = 4.0 * F;
  W.w1 = -4.0 * fx + 2.0 * fy + 2.0 * fz;
  W.w2 = 2.0 * fx - 4.0 * fy + 2.0 * fz;
  W.w3 = -2.0 * fx - 2.0 * fy + fz;
  W.w4 = 2.0 * fx + 2.0 * fy - 4.0 * fz;
  W.w5 = -2.0 * fx + fy - 2.0 * fz;
  W.w6 = fx - 2.0 * fy - 2.0 * fz;
  W.w7 = -F;
  W.iso = (W.w1 == 0.0 && W.w2 == 0.0 && W.w4 == 0.0);
  return W;
}
DEVI void nd_stencil(const NdW &W, const double p[3][3][3], const double sg[2][2][2], double &Kp, double &diag) {
  double cz[2][2], cy[2][2], cx[2][2];
  #pragma unroll
  for (int b = 0; b < 2; b++)
    #pragma unroll
    for (int a = 0; a < 2; a++) {
      cz[b][a] = sg[0][b][a] + sg[1][b][a];          // [dj][di]: the two cells that share an xy-diagonal neighbour
      cy[b][a] = sg[b][0][a] + sg[b][1][a];          // [dk][di]: xz-diagonal
      cx[b][a] = sg[b][a][0] + sg[b][a][1];          // [dk][dj]: yz-diagonal
    }
  const double S8 = (cz[0][0] + cz[0][1]) + (cz[1][0] + cz[1][1]);
  // the weights of a row sum to zero (K 1 = 0), so K phi = sum of coefficient * (phi_neighbour - phi_node): differences first, which
  // keeps the terms at the size of the answer instead of the size of diag * phi (at 256^3 the plain sum stalls at a residual of
  // ~2e-12 |rhs|, short of the 1e-12 of hgproject.f90:113-114)
  const double p0 = p[1][1][1];
  #define D(c, b, a) (p[c][b][a] - p0)
  double A7 = sg[0][0][0] * D(0, 0, 0);
  A7 = fma(sg[0][0][1], D(0, 0, 2), A7); A7 = fma(sg[0][1][0], D(0, 2, 0), A7); A7 = fma(sg[0][1][1], D(0, 2, 2), A7);
  A7 = fma(sg[1][0][0], D(2, 0, 0), A7); A7 = fma(sg[1][0][1], D(2, 0, 2), A7); A7 = fma(sg[1][1][0], D(2, 2, 0), A7); A7 = fma(sg[1][1][1], D(2, 2, 2), A7);
  double A3 = cz[0][0] * D(1, 0, 0); A3 = fma(cz[0][1], D(1, 0, 2), A3); A3 = fma(cz[1][0], D(1, 2, 0), A3); A3 = fma(cz[1][1], D(1, 2, 2), A3);
  double A5 = cy[0][0] * D(0, 1, 0); A5 = fma(cy[0][1], D(0, 1, 2), A5); A5 = fma(cy[1][0], D(2, 1, 0), A5); A5 = fma(cy[1][1], D(2, 1, 2), A5);
  double A6 = cx[0][0] * D(0, 0, 1); A6 = fma(cx[0][1], D(0, 2, 1), A6); A6 = fma(cx[1][0], D(2, 0, 1), A6); A6 = fma(cx[1][1], D(2, 2, 1), A6);
  const double dg = W.w0 * S8;
  double acc = W.w3 * A3;
  acc = fma(W.w5, A5, acc); acc = fma(W.w6, A6, acc); acc = fma(W.w7, A7, acc);
  if (!W.iso) {                                      // uniform over the launch
    double A1 = (cz[0][0] + cz[1][0]) * D(1, 1, 0); A1 = fma(cz[0][1] + cz[1][1], D(1, 1, 2), A1);
    double A2 = (cz[0][0] + cz[0][1]) * D(1, 0, 1); A2 = fma(cz[1][0] + cz[1][1], D(1, 2, 1), A2);
    double A4 = (cy[0][0] + cy[0][1]) * D(0, 1, 1); A4 = fma(cy[1][0] + cy[1][1], D(2, 1, 1), A4);
    acc = fma(W.w1, A1, acc); acc = fma(W.w2, A2, acc); acc = fma(W.w4, A4, acc);
  }
  #undef D
  Kp = acc; diag = dg;
}
// from a level in the multigrid layout
DEVI void nd_apply(const NLev &L, const double *__restrict__ phi, int i, int j, int k, double &Kp, double &diag) {
  const NdW W = nd_weights(L.f);
  const long sy = L.PX, sz = (long)L.PX * L.PY;
  const long c0 = nidx(L, i, j, k);
  double p[3][3][3], sg[2][2][2];
  #pragma unroll
  for (int c = 0; c < 3; c++)
    #pragma unroll
    for (int b = 0; b < 3; b++)
      #pragma unroll
      for (int a = 0; a < 3; a++) p[c][b][a] = phi[c0 + (a - 1) + (b - 1) * sy + (c - 1) * sz];
  #pragma unroll
  for (int c = 0; c < 2; c++)
    #pragma unroll
    for (int b = 0; b < 2; b++)
      #pragma unroll
      for (int a = 0; a < 2; a++) sg[c][b][a] = L.sig[c0 + (a - 1) + (b - 1) * sy + (c - 1) * sz];
  nd_stencil(W, p, sg, Kp, diag);
}

#define NODE_IJK(L)                                                    \
  const int i = blockIdx.x * blockDim.x + threadIdx.x;                 \
  const int j = blockIdx.y * blockDim.y + threadIdx.y;                 \
  const int k = blockIdx.z;                                            \
  const bool in_range = (i <= (L).n[0]) && (j <= (L).n[1]) && (k <= (L).n[2]);

// ---- k-marching forms of the smoother and the residual ----------------------------------------------------------
// A workgroup owns a 64 x 4 patch of (i,j) and marches through a slab of k planes keeping the three phi planes and
// the two sigma planes of the 27-point stencil in registers: per node 9 + 4 (+1 rhs) loads instead of 27 + 8 (+1),
// which is what bounds the plain kernels (they run at ~1.6 TB/s algorithmic, limited by L1/TA transactions,
// not by HBM).  The operator is nd_stencil on the register planes.
// MODE 0: Jacobi sweep (out = phi + omega (b - K phi)/diag);  MODE 1: residual (res = b - K phi, max-norm)
// Per plane a thread loads only its own column (phi at rows j-1..j+1, sigma at rows j-1..j, rhs) in one unconditional
// batch and takes the i-1 / i+1 columns from the neighbouring lanes (wave shuffles): 6 loads per node instead of 14.
// Tiles overlap by two columns: lanes 0 and 63 only feed their neighbours (62 nodes per wave row).
// (Tried: rotating the three phi planes and two sigma planes by NAME over six unrolled plane steps, to save the 22 register moves of the
// hand-over among 204 instructions per plane: the scheduler then hoists the loads of later steps, 214 VGPRs, occupancy 2, HG 17.8 -> 24.3 ms;
// capped at 128 VGPRs it spills 340 B per lane, 38 ms.  The moves stay.)
template <int MODE>
__global__ void __launch_bounds__(256) kk_nd_march(NLev L, const double *__restrict__ phi, double *__restrict__ out, double omega, int kchunk, double *nrm, int shell_later) {
  const int lane = threadIdx.x;
  const int i = (int)blockIdx.x * 62 + lane - 1;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k0 = blockIdx.z * kchunk, k1 = min(k0 + kchunk - 1, L.n[2]);
  const bool active = lane >= 1 && lane <= 62 && i <= L.n[0] && j <= L.n[1];
  const int ic = min(i, L.n[0] + 1), jc = min(j, L.n[1]);
  double rmax = 0.0;
  if (k0 <= k1) {          // uniform over the workgroup
    const long sy = L.PX, sz = (long)L.PX * L.PY;
    long c = nidx(L, ic, jc, k0);
    double p[3][3][3], sg[2][2][2];
    #pragma unroll
    for (int b = 0; b < 3; b++) { p[0][b][1] = phi[c - sz + (b - 1) * sy]; p[1][b][1] = phi[c + (b - 1) * sy]; }
    #pragma unroll
    for (int dj = 0; dj < 2; dj++) sg[0][dj][1] = L.sig[c - sz + (dj - 1) * sy];
    #pragma unroll
    for (int b = 0; b < 3; b++) {
      p[0][b][0] = lane_prev(p[0][b][1]); p[0][b][2] = lane_next(p[0][b][1]);
      p[1][b][0] = lane_prev(p[1][b][1]); p[1][b][2] = lane_next(p[1][b][1]);
    }
    #pragma unroll
    for (int dj = 0; dj < 2; dj++) sg[0][dj][0] = lane_prev(sg[0][dj][1]);
    const bool dir_ij = (i == 0 && L.dirlo[0]) || (i == L.n[0] && L.dirhi[0]) || (j == 0 && L.dirlo[1]) || (j == L.n[1] && L.dirhi[1]);
    const NdW W = nd_weights(L.f);
    for (int k = k0; k <= k1; k++, c += sz) {
      #pragma unroll
      for (int b = 0; b < 3; b++) p[2][b][1] = phi[c + sz + (b - 1) * sy];
      #pragma unroll
      for (int dj = 0; dj < 2; dj++) sg[1][dj][1] = L.sig[c + (dj - 1) * sy];
      const double rhs = L.b[c];
      #pragma unroll
      for (int b = 0; b < 3; b++) { p[2][b][0] = lane_prev(p[2][b][1]); p[2][b][2] = lane_next(p[2][b][1]); }
      #pragma unroll
      for (int dj = 0; dj < 2; dj++) sg[1][dj][0] = lane_prev(sg[1][dj][1]);
      const bool dir = dir_ij || (k == 0 && L.dirlo[2]) || (k == L.n[2] && L.dirhi[2]);
      const double p0 = p[1][1][1];
      double Kp, diag; nd_stencil(W, p, sg, Kp, diag);
      if (MODE == 0) {
        double v = p0;
        if (!dir && diag != 0.0) v = p0 + omega * ((rhs - Kp) / diag);
        if (active) out[c] = v;
      } else {
        const double r = dir ? 0.0 : rhs - Kp;
        if (active) { out[c] = r; if (!(shell_later && nd_is_shell(L, i, j, k, shell_later))) rmax = nmax(rmax, fabs(r)); }
      }
      #pragma unroll
      for (int b = 0; b < 3; b++)
        #pragma unroll
        for (int a = 0; a < 3; a++) { p[0][b][a] = p[1][b][a]; p[1][b][a] = p[2][b][a]; }
      #pragma unroll
      for (int dj = 0; dj < 2; dj++)
        #pragma unroll
        for (int di = 0; di < 2; di++) sg[0][dj][di] = sg[1][dj][di];
    }
  }
  if (MODE == 1 && nrm) block_atomic_max(nrm, rmax);
}

// ---- paired form of the march (levels at least 128 nodes wide) -------------------------------------------------------------------------
// After the operator was regrouped (nd_stencil) the march at 257^3 stayed at 0.145 ms: what binds it is the texture addresser, as it did
// the cell-centred colour pass (profiles/r01_smoother_rho_pmc.json) -- seven 8-byte memory instructions per node.  Here a thread owns the
// two nodes (2t, 2t+1) of a row: every access is an aligned 16-byte pair, the columns 2t-1 / 2t+2 come from the neighbouring lanes
// (DPP), so a node costs 3.5 memory instructions and half the lane exchanges.  Same arithmetic per node (nd_stencil), same bits.
DEVI double2 ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
// Stores and loads share one in-order counter (vmcnt) on gfx9: a wave that waits for a load also waits for every OLDER store, and when
// the number of stores in flight is not known at compile time (a store inside a divergent branch) the compiler waits for ALL of them --
// `s_waitcnt vmcnt(0)` at the loop latch, i.e. the full store round trip exposed once per plane.  Hence: exactly ONE store instruction
// per plane, executed by every lane (lanes that own no node write a 16-byte slot of a scratch line instead), issued BEFORE the loads
// of the next plane, so that the store's latency hides under theirs.
__device__ double g_nd_sink[128];
__device__ int g_nd_dbg = 0;      // (probe of round 3, always 0 now: 1 = no stencil arithmetic, 2 = no loads inside the march)
// Round 3 -- the shape of the launch.  (i) A row of 257 nodes is 129 pairs = two full wave rows (62 owned pairs each) and FIVE pairs more:
// with one tile shape the third tile of every row ran 57 of its 62 pair lanes idle -- a third of all waves of the sweep issued the loads of
// 8 % of the nodes, and the sweep is bound by the loads it has in flight.  The remainder columns are now covered by waves that pack several
// rows: a lane segment of 2^lw lanes (lw = 2..6, the smallest that holds the remainder; its first and last lane feed their neighbours
// as before -- the DPP shifts cross segment borders only into those two lanes) carries one row, a wave 64 >> lw rows, a workgroup four
// times that.  Main tiles and remainder tiles are workgroups of ONE launch (1-D grid: main tiles first, in the XCD-aware order).
// (ii) The k-slabs are balanced (sizes differ by at most one plane): 257 planes in 16 slabs of 17 left a last slab of two planes.
struct NdPairGrid { int gxm, gy, gz, nmain, lwr, gyr, rev; };       // main tiles gxm x gy x gz (lw = 6), then gyr x gz remainder tiles of segment 2^lwr
// (round 3, measured and rejected at 257^3: sigma and rhs loaded with the non-temporal hint 0.1392 -> 0.1420 ms; the kernel held to 128 VGPRs
// -- four waves per SIMD, 116 bytes of scratch per lane -- 0.341 ms)
template <int MODE, int ROWS>
__global__ void __launch_bounds__(64 * ROWS) kk_nd_march_pair(NLev L, const double *__restrict__ phi, double *__restrict__ out, double omega, NdPairGrid G, double *nrm, int shell_later) {
  const int lane = threadIdx.x;
  const int id = (int)blockIdx.x;
  int lw, pair0, j, bz;
  if (id < G.nmain) {                                                     // xcd_tile's order over the main tiles
    const int q = G.nmain >> 3, r = G.nmain & 7, x = id & 7, slot = id >> 3;
    int t = (x < r) ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
    if (G.rev) t = G.nmain - 1 - t;                                       // every other march of a level walks the tiles backwards (nd_pair_grid)
    lw = 6; pair0 = (t % G.gxm) * 62; j = ((t / G.gxm) % G.gy) * ROWS + (int)threadIdx.y; bz = t / (G.gxm * G.gy);
  } else {
    const int t = G.rev ? G.gyr * G.gz - 1 - (id - G.nmain) : id - G.nmain;
    lw = G.lwr; pair0 = G.gxm * 62; j = (((t % G.gyr) * ROWS + (int)threadIdx.y) << (6 - lw)) + (lane >> lw); bz = t / G.gyr;
  }
  const int seg = 1 << lw, sl = lane & (seg - 1);
  // (measured and rejected: all 64 lanes owning a pair -- whole 128-byte lines per wave row -- with the two outside columns from an extra
  // two-lane load per row: 0.154 -> 0.202 ms per sweep at 257^3)
  const int ia = 2 * (pair0 + sl - 1);                                  // nodes ia, ia + 1; the first and last lane of a segment only feed their neighbours
  const int nzp = L.n[2] + 1;
  const int k0 = (int)(((long)bz * nzp) / G.gz), k1 = (int)(((long)(bz + 1) * nzp) / G.gz) - 1;
  const bool own = sl >= 1 && sl <= seg - 2 && j <= L.n[1];
  const bool actA = own && ia <= L.n[0], actB = own && ia + 1 <= L.n[0];
  const int iac = min(ia, L.PX - 18), jc = min(j, L.n[1]);                // load address kept inside the (zero-padded) row
  double rmax = 0.0;
  if (k0 <= k1) {          // uniform over the workgroup
    const long sy = L.PX, sz = (long)L.PX * L.PY;
    long c = nidx(L, iac, jc, k0);
    // a lane with node A writes its pair (node B of the last pair of a row may be the ghost node n+1: it gets its old value back, resp. a
    // zero residual); the other lanes write to the sink.  One pointer, advanced by the plane stride (0 for the sink).
    double *op = actA ? out + c : g_nd_sink + 2 * lane;       // (rows beyond the level, pairs beyond its width)
    const long ostep = actA ? sz : 0;
    // q[plane][row][col]: col 0..3 = nodes ia-1 .. ia+2;  sg[dk][dj][col]: col 0..2 = cells ia-1 .. ia+1
    double q[3][3][4], sg[2][2][3];
    #define LOADP(pl, off) { _Pragma("unroll") for (int b = 0; b < 3; b++) { const double2 v = ld2(phi + (off) + (b - 1) * sy); q[pl][b][1] = v.x; q[pl][b][2] = v.y; } }
    #define EXCHP(pl) { _Pragma("unroll") for (int b = 0; b < 3; b++) { q[pl][b][0] = lane_prev(q[pl][b][2]); q[pl][b][3] = lane_next(q[pl][b][1]); } }
    #define LOADS(dk, off) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) { const double2 v = ld2(L.sig + (off) + (dj - 1) * sy); sg[dk][dj][1] = v.x; sg[dk][dj][2] = v.y; } }
    #define EXCHS(dk) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) sg[dk][dj][0] = lane_prev(sg[dk][dj][2]); }
    LOADP(0, c - sz) LOADP(1, c) LOADS(0, c - sz)
    EXCHP(0) EXCHP(1) EXCHS(0)
    const bool dirj = (j == 0 && L.dirlo[1]) || (j == L.n[1] && L.dirhi[1]);
    const bool dirA_ij = dirj || (ia == 0 && L.dirlo[0]) || (ia == L.n[0] && L.dirhi[0]);
    const bool dirB_ij = dirj || (ia + 1 == L.n[0] && L.dirhi[0]);
    const NdW W = nd_weights(L.f);
    const int dbg = g_nd_dbg;
    for (int k = k0; k <= k1; k++, c += sz, op += ostep) {
      double2 rhs = make_double2(1.0, 1.0);
      if (!(dbg & 2)) { LOADP(2, c + sz) LOADS(1, c) rhs = ld2(L.b + c); }
      EXCHP(2) EXCHS(1)
      const bool dirk = (k == 0 && L.dirlo[2]) || (k == L.n[2] && L.dirhi[2]);
      double pa[3][3][3], pb[3][3][3], sa[2][2][2], sb[2][2][2];
      #pragma unroll
      for (int pl = 0; pl < 3; pl++)
        #pragma unroll
        for (int b = 0; b < 3; b++)
          #pragma unroll
          for (int a = 0; a < 3; a++) { pa[pl][b][a] = q[pl][b][a]; pb[pl][b][a] = q[pl][b][a + 1]; }
      #pragma unroll
      for (int dk = 0; dk < 2; dk++)
        #pragma unroll
        for (int dj = 0; dj < 2; dj++)
          #pragma unroll
          for (int a = 0; a < 2; a++) { sa[dk][dj][a] = sg[dk][dj][a]; sb[dk][dj][a] = sg[dk][dj][a + 1]; }
      double KpA, dgA, KpB, dgB;
      if (dbg & 1) { KpA = pa[2][0][1] + pa[2][1][1] + pa[2][2][1] + sa[1][0][1] + sa[1][1][1]; KpB = pb[2][0][1] + pb[2][1][1] + pb[2][2][1] + sb[1][0][1] + sb[1][1][1]; dgA = dgB = 1.0; }
      else {
      nd_stencil(W, pa, sa, KpA, dgA);
      nd_stencil(W, pb, sb, KpB, dgB);
      }
      const double p0A = q[1][1][1], p0B = q[1][1][2];
      double2 o;
      if (MODE == 0) {
        o.x = p0A; o.y = p0B;
        if (!(dirA_ij || dirk) && dgA != 0.0) o.x = p0A + omega * ((rhs.x - KpA) / dgA);
        if (actB && !(dirB_ij || dirk) && dgB != 0.0) o.y = p0B + omega * ((rhs.y - KpB) / dgB);
      } else {
        o.x = (dirA_ij || dirk) ? 0.0 : rhs.x - KpA;
        o.y = (!actB || dirB_ij || dirk) ? 0.0 : rhs.y - KpB;
        if (actA && !(shell_later && nd_is_shell(L, ia, j, k, shell_later))) rmax = nmax(rmax, fabs(o.x));
        if (actB && !(shell_later && nd_is_shell(L, ia + 1, j, k, shell_later))) rmax = nmax(rmax, fabs(o.y));
      }
      *reinterpret_cast<double2 *>(op) = o;
      #pragma unroll
      for (int b = 0; b < 3; b++)
        #pragma unroll
        for (int a = 0; a < 4; a++) { q[0][b][a] = q[1][b][a]; q[1][b][a] = q[2][b][a]; }
      #pragma unroll
      for (int dj = 0; dj < 2; dj++)
        #pragma unroll
        for (int a = 0; a < 3; a++) sg[0][dj][a] = sg[1][dj][a];
    }
    #undef LOADP
    #undef EXCHP
    #undef LOADS
    #undef EXCHS
  }
  if (MODE == 1 && nrm) block_atomic_max(nrm, rmax);
}

// Full weighting (P^T / 8 up to the factor 0.125 applied by the caller) of the 27 residuals around r[0], SEPARABLY and in this order (round 4; oracle:
// nd_restrict): along x on each of the nine lines, X = (0.5 r[-1] + r[0]) + 0.5 r[+1]; along z on each of the three rows, (0.5 X[k-1] + X[k]) + 0.5 X[k+1];
// along y last.  The residual march of a wide one-box level forms the x- and z-sums from the lanes and planes it holds anyway and never stores the
// residual (kk_nd_march_pair_rst); this function is the same arithmetic from the stored residual (narrow, multi-box, periodic levels; the nested iteration).
DEVI double nd_fw3(double m, double c, double p) { return (0.5 * m + c) + 0.5 * p; }
DEVI double nd_fw27(const double *__restrict__ r, long sy, long sz) {
  double xz[3];
  #pragma unroll
  for (int b = -1; b <= 1; b++) {
    double x[3];
    #pragma unroll
    for (int c = -1; c <= 1; c++) { const double *q = r + b * sy + c * sz; x[c + 1] = nd_fw3(q[-1], q[0], q[1]); }
    xz[b + 1] = nd_fw3(x[0], x[1], x[2]);
  }
  return nd_fw3(xz[0], xz[1], xz[2]);
}
// ---- residual + full weighting in one march (round 4) -------------------------------------------------------------------------------------
// The V-cycle's residual is read once more by the restriction and by nothing else.  Here the residual march of a wide ONE-BOX, non-periodic level keeps
// it in registers: per plane a thread forms X = (0.5 r[ia-1] + r[ia]) + 0.5 r[ia+1] for its coarse column I = ia / 2 (r[ia-1] is the previous lane's
// node B -- also the feed lane's, whose node-B stencil is complete), every second plane Z = (0.5 X[2K-1] + X[2K]) + 0.5 X[2K+1], and stores Z for
// every FINE row j into T(I, j, K) -- a quarter of the level, kept in the level's residual array, which is otherwise unused now; kk_nd_rst_y
// finishes along y.  The order of nd_fw27 / the oracle's nd_restrict, hence the same bits as the unfused pair (tests: the variants test).
// Slabs hold whole coarse planes Ka .. Kb, i.e. fine planes 2Ka-1 .. 2Kb+1: neighbouring slabs both evaluate the odd plane between them.
// Saves the 8 B/node store and the restriction's 8 B/node load + launch: 0.143 + 0.052 -> ms per cycle at 257^3 see DESIGN section 11.
template <int ROWS>
__global__ void __launch_bounds__(64 * ROWS) kk_nd_march_pair_rst(NLev L, const double *__restrict__ phi, double *__restrict__ T, NdPairGrid G, double *nrm) {
  const int lane = threadIdx.x;
  const int id = (int)blockIdx.x;
  int lw, pair0, j, bz;
  if (id < G.nmain) {
    const int q = G.nmain >> 3, r = G.nmain & 7, x = id & 7, slot = id >> 3;
    int t = (x < r) ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
    if (G.rev) t = G.nmain - 1 - t;                                       // every other march of a level walks the tiles backwards (nd_pair_grid)
    lw = 6; pair0 = (t % G.gxm) * 62; j = ((t / G.gxm) % G.gy) * ROWS + (int)threadIdx.y; bz = t / (G.gxm * G.gy);
  } else {
    const int t = G.rev ? G.gyr * G.gz - 1 - (id - G.nmain) : id - G.nmain;
    lw = G.lwr; pair0 = G.gxm * 62; j = (((t % G.gyr) * ROWS + (int)threadIdx.y) << (6 - lw)) + (lane >> lw); bz = t / G.gyr;
  }
  const int seg = 1 << lw, sl = lane & (seg - 1);
  const int ia = 2 * (pair0 + sl - 1);
  const int ncp = L.n[2] / 2 + 1;                                        // coarse planes
  const int Ka = (int)(((long)bz * ncp) / G.gz), Kb = (int)(((long)(bz + 1) * ncp) / G.gz) - 1;
  const bool own = sl >= 1 && sl <= seg - 2 && j <= L.n[1];
  const bool actA = own && ia <= L.n[0], actB = own && ia + 1 <= L.n[0];
  const bool feedB = sl <= seg - 2 && j <= L.n[1] && ia + 1 >= 0 && ia + 1 <= L.n[0];      // node B's residual is complete on this lane (the feed lane 0 included)
  const int iac = min(max(ia, -16), L.PX - 18), jc = min(j, L.n[1]);
  double rmax = 0.0;
  if (Ka <= Kb) {
    const int kf = max(2 * Ka - 1, 0), kl = min(2 * Kb + 1, L.n[2]);
    const long sy = L.PX, sz = (long)L.PX * L.PY;
    long c = nidx(L, iac, jc, kf);
    const int NI = L.n[0] / 2 + 1;
    double *tp = T + ((long)Ka * (L.n[1] + 1) + jc) * NI + min(max(ia, 0), L.n[0]) / 2;
    const long tstep = (long)(L.n[1] + 1) * NI;
    double q[3][3][4], sg[2][2][3];
    #define LOADP(pl, off) { _Pragma("unroll") for (int b = 0; b < 3; b++) { const double2 v = ld2(phi + (off) + (b - 1) * sy); q[pl][b][1] = v.x; q[pl][b][2] = v.y; } }
    #define EXCHP(pl) { _Pragma("unroll") for (int b = 0; b < 3; b++) { q[pl][b][0] = lane_prev(q[pl][b][2]); q[pl][b][3] = lane_next(q[pl][b][1]); } }
    #define LOADS(dk, off) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) { const double2 v = ld2(L.sig + (off) + (dj - 1) * sy); sg[dk][dj][1] = v.x; sg[dk][dj][2] = v.y; } }
    #define EXCHS(dk) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) sg[dk][dj][0] = lane_prev(sg[dk][dj][2]); }
    LOADP(0, c - sz) LOADP(1, c) LOADS(0, c - sz)
    EXCHP(0) EXCHP(1) EXCHS(0)
    const bool dirj = (j == 0 && L.dirlo[1]) || (j == L.n[1] && L.dirhi[1]);
    const bool dirA_ij = dirj || (ia == 0 && L.dirlo[0]) || (ia == L.n[0] && L.dirhi[0]);
    const bool dirB_ij = dirj || (ia + 1 == L.n[0] && L.dirhi[0]);
    const NdW W = nd_weights(L.f);
    double Xm = 0.0, X0 = 0.0;                                           // X on the planes 2K-1 and 2K of the coarse plane being assembled
    for (int k = kf; k <= kl; k++, c += sz) {
      LOADP(2, c + sz) LOADS(1, c)
      const double2 rhs = ld2(L.b + c);
      EXCHP(2) EXCHS(1)
      const bool dirk = (k == 0 && L.dirlo[2]) || (k == L.n[2] && L.dirhi[2]);
      double pa[3][3][3], pb[3][3][3], sa[2][2][2], sb[2][2][2];
      #pragma unroll
      for (int pl = 0; pl < 3; pl++)
        #pragma unroll
        for (int b = 0; b < 3; b++)
          #pragma unroll
          for (int a = 0; a < 3; a++) { pa[pl][b][a] = q[pl][b][a]; pb[pl][b][a] = q[pl][b][a + 1]; }
      #pragma unroll
      for (int dk = 0; dk < 2; dk++)
        #pragma unroll
        for (int dj = 0; dj < 2; dj++)
          #pragma unroll
          for (int a = 0; a < 2; a++) { sa[dk][dj][a] = sg[dk][dj][a]; sb[dk][dj][a] = sg[dk][dj][a + 1]; }
      double KpA, dgA, KpB, dgB;
      nd_stencil(W, pa, sa, KpA, dgA);
      nd_stencil(W, pb, sb, KpB, dgB);
      const double rA = (!actA || dirA_ij || dirk) ? 0.0 : rhs.x - KpA;
      const double rB = (!feedB || dirB_ij || dirk) ? 0.0 : rhs.y - KpB;
      if (actA) rmax = nmax(rmax, fabs(rA));
      if (actB) rmax = nmax(rmax, fabs(rB));
      const double X = nd_fw3(lane_prev(rB), rA, rB);
      if (k & 1) {                                                       // uniform: plane 2K+1 completes coarse plane K = (k - 1) / 2 and opens K + 1
        if (((k - 1) >> 1) >= Ka) { if (actA) *tp = nd_fw3(Xm, X0, X); tp += tstep; }
        Xm = X;
      } else X0 = X;
      #pragma unroll
      for (int b = 0; b < 3; b++)
        #pragma unroll
        for (int a = 0; a < 4; a++) { q[0][b][a] = q[1][b][a]; q[1][b][a] = q[2][b][a]; }
      #pragma unroll
      for (int dj = 0; dj < 2; dj++)
        #pragma unroll
        for (int a = 0; a < 3; a++) sg[0][dj][a] = sg[1][dj][a];
    }
    if (!(kl & 1) && actA) *tp = nd_fw3(Xm, X0, 0.0);                     // the level's last plane n2 = 2 Kb is even: plane n2 + 1 holds no residual
    #undef LOADP
    #undef EXCHP
    #undef LOADS
    #undef EXCHS
  }
  if (nrm) block_atomic_max(nrm, rmax);
}
// the y-sums of the fused residual + restriction: T(I, j, K) -> b of the coarse level, phi of the coarse level := 0 (as kk_nd_restrict)
__global__ void kk_nd_rst_y(NLev F, const double *__restrict__ T, NLev C) {
  NODE_IJK(C)
  if (!in_range) return;
  double s = 0.0;
  if (!nd_is_dir(C, i, j, k)) {
    const int NI = F.n[0] / 2 + 1;
    const double *t = T + ((long)k * (F.n[1] + 1) + 2 * j) * NI + i;
    const double tm = (2 * j - 1 >= 0) ? t[-NI] : 0.0, tp = (2 * j + 1 <= F.n[1]) ? t[NI] : 0.0;
    s = nd_fw3(tm, t[0], tp);
  }
  const long cn = nidx(C, i, j, k);
  C.b[cn] = s * 0.125;
  C.phi[cn] = 0.0;
}

// (measured and rejected: a plane-per-workgroup form of the paired sweep -- a workgroup owns a 124 x 4 patch of ONE k-plane, the planes
// shared through the XCD's L2 like the cell-centred colour pass, 14 sixteen-byte loads per pair of nodes: 0.224 ms against 0.150 ms)
// ---- halo exchange next to a sweep ----------------------------------------------------------------------------
// Only the nodes ON the faces of a box read ghost nodes.  A sweep is out of place (phi -> tmp, phi -> res), so the march may run over the
// whole box while the halo of phi is still in flight on ctx().halo_stream -- its face nodes come out wrong -- and this kernel then
// recomputes exactly those nodes once the halo has landed (the x faces own their edges and corners, the y faces the remaining edges).
// With `shell_later` the march leaves the face nodes out of its residual norm; this kernel contributes theirs.
template <int MODE>
__global__ void __launch_bounds__(256) kk_nd_shell(NLev L, const double *__restrict__ phi, double *__restrict__ out, double omega, double *nrm, int hm) {
  int f = 0;
  for (int z = blockIdx.z;; f++) if ((hm >> f) & 1) { if (z == 0) break; z--; }      // blockIdx.z-th face of the mask
  const int d = f >> 1, side = f & 1;
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y;
  const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
  int q[3];
  q[d] = side ? L.n[d] : 0; q[da] = a; q[db] = b;
  bool act = q[da] <= L.n[da] && q[db] <= L.n[db];
  if (d >= 1 && (((hm & 1) && q[0] == 0) || ((hm & 2) && q[0] == L.n[0]))) act = false;           // owned by an x face
  if (d == 2 && (((hm & 4) && q[1] == 0) || ((hm & 8) && q[1] == L.n[1]))) act = false;           // owned by a y face
  double rmax = 0.0;
  if (act) {
    const long c = nidx(L, q[0], q[1], q[2]);
    const double p0 = phi[c];
    const bool dir = nd_is_dir(L, q[0], q[1], q[2]);
    double Kp, diag; nd_apply(L, phi, q[0], q[1], q[2], Kp, diag);
    if (MODE == 0) {
      double v = p0;
      if (!dir && diag != 0.0) v = p0 + omega * ((L.b[c] - Kp) / diag);
      out[c] = v;
    } else {
      const double r = dir ? 0.0 : L.b[c] - Kp;
      out[c] = r; rmax = fabs(r);
    }
  }
  if (MODE == 1 && nrm) block_atomic_max(nrm, rmax);
}
// the damping of the sweep being launched when it is not vdn_params.hg_omega (nd_jacobi_d / nd_jacobi_t set it per sweep: NdOm)
static double g_nd_omega_now = 0.0;
static double nd_cur_omega() { return g_nd_omega_now > 0.0 ? g_nd_omega_now : ctx().prm.hg_omega; }
struct NdOmegaScope { ~NdOmegaScope() { g_nd_omega_now = 0.0; } };      // the override does not outlive the run of sweeps that set it, exceptions included
template <int MODE> static void nd_launch_shell(const NLev &L, const double *phi, double *out, double *nrm, int hm) {
  if (!hm) return;
  const int m = std::max(L.n[0], std::max(L.n[1], L.n[2])) + 1;
  hipLaunchKernelGGL((kk_nd_shell<MODE>), dim3((m + 63) / 64, (m + 3) / 4, (unsigned)__builtin_popcount(hm)), dim3(64, 4, 1), 0, ctx().stream, L, phi, out, nd_cur_omega(), nrm, hm);
}

// ghost nodes (and the periodic alias node n): periodic image, else zero
__global__ void kk_nd_fill_nodes(NLev L, double *a) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] + 1 || j > L.n[1] + 1 || k > L.n[2] + 1) return;
  int q[3] = { i, j, k }, s[3] = { i, j, k }; bool g = false, zero = false;
  #pragma unroll
  for (int d = 0; d < 3; d++) {
    if (L.per[d]) { if (q[d] < 0) { s[d] = q[d] + L.n[d]; g = true; } else if (q[d] >= L.n[d]) { s[d] = q[d] - L.n[d]; g = true; } }
    else if (q[d] < 0 || q[d] > L.n[d]) { g = true; zero = true; }
  }
  if (g) a[nidx(L, i, j, k)] = zero ? 0.0 : a[nidx(L, s[0], s[1], s[2])];
}
__global__ void kk_nd_fill_cells(NLev L, double *a) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] || j > L.n[1] || k > L.n[2]) return;
  int q[3] = { i, j, k }, s[3] = { i, j, k }; bool g = false, zero = false;
  #pragma unroll
  for (int d = 0; d < 3; d++) {
    if (q[d] < 0) { g = true; if (L.per[d]) s[d] = q[d] + L.n[d]; else zero = true; }
    else if (q[d] >= L.n[d]) { g = true; if (L.per[d]) s[d] = q[d] - L.n[d]; else zero = true; }
  }
  if (g) a[nidx(L, i, j, k)] = zero ? 0.0 : a[nidx(L, s[0], s[1], s[2])];
}
__global__ void kk_nd_restrict(NLev F, NLev C) {
  NODE_IJK(C)
  if (!in_range) return;
  double s = 0.0;
  if (!nd_is_dir(C, i, j, k)) {
    const long sy = F.PX, sz = (long)F.PX * F.PY;
    const long f0 = nidx(F, 2 * i, 2 * j, 2 * k);
    s = nd_fw27(F.res + f0, sy, sz);
  }
  const long cn = nidx(C, i, j, k);
  C.b[cn] = s * 0.125;
  C.phi[cn] = 0.0;                     // the error equation starts from zero: saves a memset launch per level and cycle (ghost nodes stay zero / are refreshed)
}
// (round 3, measured and rejected: the restriction of wide levels with aligned 16-byte pairs -- a lane owns coarse node I, loads the fine pair
// (2I, 2I+1) of each of the nine lines and takes column 2I-1 from the previous lane: 59.8 us against 55.5 us at 257^3 -> 129^3; the 27
// strided reads of the plain kernel are served by L2 lines the neighbouring threads share anyway)
// trilinear interpolation of the coarse field at fine node offsets (oi,oj,ok) of coarse node (I,J,K): the eight coarse values are
// loaded in one unconditional batch and added under predicates in the order (c,b,a) ascending of the oracle's loops
DEVI double nd_interp8(const NLev &C, const double *__restrict__ cp, int I, int J, int K, int oi, int oj, int ok) {
  const long c0 = nidx(C, I, J, K), sy = C.PX, sz = (long)C.PX * C.PY;
  const double v000 = cp[c0], v100 = cp[c0 + 1], v010 = cp[c0 + sy], v110 = cp[c0 + sy + 1];
  const double v001 = cp[c0 + sz], v101 = cp[c0 + sz + 1], v011 = cp[c0 + sz + sy], v111 = cp[c0 + sz + sy + 1];
  double s = 0.0;
  s = s + v000;
  s = oi ? s + v100 : s;
  s = oj ? s + v010 : s;
  s = (oi && oj) ? s + v110 : s;
  s = ok ? s + v001 : s;
  s = (ok && oi) ? s + v101 : s;
  s = (ok && oj) ? s + v011 : s;
  s = (ok && oi && oj) ? s + v111 : s;
  return s * (1.0 / (double)((1 + oi) * (1 + oj) * (1 + ok)));
}
__global__ void kk_nd_prolong(NLev F, NLev C) {
  NODE_IJK(F)
  if (!in_range) return;
  if (nd_is_dir(F, i, j, k)) return;
  const long f = nidx(F, i, j, k);
  F.phi[f] = F.phi[f] + nd_interp8(C, C.phi, i >> 1, j >> 1, k >> 1, i & 1, j & 1, k & 1);
}
// k-marching form of kk_nd_prolong / kk_nd_prolong_tail: a thread owns a fine (i,j) column of a slab of planes and keeps the four
// coarse values of planes K and K+1 in registers -- 4 coarse loads per TWO fine planes instead of 8 per node; same sums, same order
__global__ void __launch_bounds__(256) kk_nd_prolong_m(NLev F, NLev C, int c00, int c01, int c02, int kchunk) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  const int k0 = (int)blockIdx.z * kchunk, k1 = min(k0 + kchunk - 1, F.n[2]);          // kchunk is even: k0 is even
  if (i > F.n[0] || j > F.n[1] || k0 > k1) return;
  const int oi = i & 1, oj = j & 1;
  const long sy = C.PX, sz = (long)C.PX * C.PY;
  const double *__restrict__ cp = C.phi;
  long c0 = nidx(C, c00 + (i >> 1), c01 + (j >> 1), c02 + (k0 >> 1));
  double a00 = cp[c0], a10 = cp[c0 + 1], a01 = cp[c0 + sy], a11 = cp[c0 + sy + 1];
  const bool dir_ij = (i == 0 && F.dirlo[0]) || (i == F.n[0] && F.dirhi[0]) || (j == 0 && F.dirlo[1]) || (j == F.n[1] && F.dirhi[1]);
  const double w0 = 1.0 / (double)((1 + oi) * (1 + oj)), w1 = 1.0 / (double)((1 + oi) * (1 + oj) * 2);
  for (int k = k0; k <= k1; k += 2) {
    const double b00 = cp[c0 + sz], b10 = cp[c0 + sz + 1], b01 = cp[c0 + sz + sy], b11 = cp[c0 + sz + sy + 1];
    const long f = nidx(F, i, j, k);
    const double pe = F.phi[f];
    const bool odd_in = k + 1 <= k1;
    const long f2 = f + (long)F.PX * F.PY;
    const double po = odd_in ? F.phi[f2] : 0.0;
    double s = 0.0;
    s = s + a00;
    s = oi ? s + a10 : s;
    s = oj ? s + a01 : s;
    s = (oi && oj) ? s + a11 : s;
    double t = s;
    t = t + b00;
    t = oi ? t + b10 : t;
    t = oj ? t + b01 : t;
    t = (oi && oj) ? t + b11 : t;
    if (!(dir_ij || (k == 0 && F.dirlo[2]) || (k == F.n[2] && F.dirhi[2]))) F.phi[f] = pe + s * w0;
    if (odd_in && !(dir_ij || (k + 1 == F.n[2] && F.dirhi[2]))) F.phi[f2] = po + t * w1;
    a00 = b00; a10 = b10; a01 = b01; a11 = b11; c0 += sz;
  }
}
static void nd_launch_prolong(const NLev &F, const NLev &C, int c00, int c01, int c02) {
  const int nzp = F.n[2] + 1;
  const int kchunk = nzp > 64 ? 16 : (nzp > 16 ? 8 : 2);
  hipLaunchKernelGGL(kk_nd_prolong_m, dim3((F.n[0] + 64) / 64, (F.n[1] + 4) / 4, (nzp + kchunk - 1) / kchunk), dim3(64, 4, 1), 0, ctx().stream, F, C, c00, c01, c02, kchunk);
}
__global__ void kk_nd_coarsen_sigma(NLev F, NLev C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= C.n[0] || j >= C.n[1] || k >= C.n[2]) return;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const long f = nidx(F, 2 * i, 2 * j, 2 * k);
  double s = 0.0;
  #pragma unroll
  for (int c = 0; c < 2; c++)
    #pragma unroll
    for (int b = 0; b < 2; b++)
      #pragma unroll
      for (int a = 0; a < 2; a++) s = s + F.sig[f + a + b * sy + c * sz];
  C.sig[nidx(C, i, j, k)] = s * 0.125;
}

// bottom solve: all Jacobi sweeps of the coarsest level in one launch by one workgroup (ping-pong between
// phi and tmp; ghost nodes refreshed by the same workgroup between sweeps).  Returns with the result in
// `a` if nsweeps is even, in `b` otherwise (the host swaps accordingly).
// (om1, om2, nsp: the first nsp sweeps -- the two pre-smoothing sweeps of a V-cycle -- are damped by om1, om2 instead of omega; NdOm below)
__global__ void __launch_bounds__(1024) kk_nd_bottom(NLev L, double *a, double *b, int nsweeps, double omega, double om1 = 0.0, double om2 = 0.0, int nsp = 0) {
  const int ex = L.n[0] + 3, ey = L.n[1] + 3, ez = L.n[2] + 3;
  const int nx = L.n[0] + 1, ny = L.n[1] + 1, nz = L.n[2] + 1;
  double *src = a, *dst = b;
  for (int s = 0; s < nsweeps; s++) {
    for (int t = threadIdx.x; t < ex * ey * ez; t += blockDim.x) {       // nd_fill_nodes
      const int i = t % ex - 1, j = (t / ex) % ey - 1, k = t / (ex * ey) - 1;
      int q[3] = { i, j, k }, sidx[3] = { i, j, k }; bool g = false, zero = false;
      for (int d = 0; d < 3; d++) {
        if (L.per[d]) { if (q[d] < 0) { sidx[d] = q[d] + L.n[d]; g = true; } else if (q[d] >= L.n[d]) { sidx[d] = q[d] - L.n[d]; g = true; } }
        else if (q[d] < 0 || q[d] > L.n[d]) { g = true; zero = true; }
      }
      if (g) src[nidx(L, i, j, k)] = zero ? 0.0 : src[nidx(L, sidx[0], sidx[1], sidx[2])];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nx * ny * nz; t += blockDim.x) {
      const int i = t % nx, j = (t / nx) % ny, k = t / (nx * ny);
      const long c = nidx(L, i, j, k);
      const double p0 = src[c];
      double v = p0;
      if (!nd_is_dir(L, i, j, k)) {
        double Kp, diag; nd_apply(L, src, i, j, k, Kp, diag);
        if (diag != 0.0) v = p0 + (s < nsp ? (s == 0 ? om1 : om2) : omega) * ((L.b[c] - Kp) / diag);
      }
      dst[c] = v;
    }
    __syncthreads();
    double *tsw = src; src = dst; dst = tsw;
  }
}

// ---- the small end of a V-cycle in one launch ---------------------------------------------------------------------------------------------
// Levels of at most 9^3 nodes (9^3, 5^3, 3^3 under a 257^3 solve) took 13 launches of 5-14 us per cycle -- smoothing, residual, restriction,
// bottom sweeps, prolongation, each a one-workgroup grid -- 1.5 ms of an HG projection.  One workgroup runs the whole recursion here:
// the same per-node code (nd_apply, the sums of kk_nd_restrict, nd_interp8) in the same order, barriers instead of launch boundaries.
// L[0] is entered like any level of nd_vcycle_d (b set, phi = 0) and left with its correction in the array the host's swap parity names.
#define ND_TAIL_MAX 4
struct NdTailArgs { NLev L[ND_TAIL_MAX]; int nlev, nu1, nu2, nbot; double omega, om1, om2; int nsp; };      // om1, om2, nsp: the pre-smoothing sweeps' damping (NdOm)
DEVI void wg_nd_fill(const NLev &L, double *a) {                     // nd_fill_nodes; ghost nodes outside a physical face stay zero
  if (!(L.per[0] || L.per[1] || L.per[2])) return;
  const int ex = L.n[0] + 3, ey = L.n[1] + 3, ez = L.n[2] + 3;
  for (int t = threadIdx.x; t < ex * ey * ez; t += blockDim.x) {
    const int i = t % ex - 1, j = (t / ex) % ey - 1, k = t / (ex * ey) - 1;
    int q[3] = { i, j, k }, sidx[3] = { i, j, k }; bool g = false, zero = false;
    for (int d = 0; d < 3; d++) {
      if (L.per[d]) { if (q[d] < 0) { sidx[d] = q[d] + L.n[d]; g = true; } else if (q[d] >= L.n[d]) { sidx[d] = q[d] - L.n[d]; g = true; } }
      else if (q[d] < 0 || q[d] > L.n[d]) { g = true; zero = true; }
    }
    if (g) a[nidx(L, i, j, k)] = zero ? 0.0 : a[nidx(L, sidx[0], sidx[1], sidx[2])];
  }
  __syncthreads();
}
DEVI void wg_nd_jacobi(const NLev &L, double *&src, double *&dst, int nsweeps, double omega, double om1 = 0.0, double om2 = 0.0, int nsp = 0) {
  const int nx = L.n[0] + 1, ny = L.n[1] + 1, nz = L.n[2] + 1;
  for (int s = 0; s < nsweeps; s++) {
    wg_nd_fill(L, src);
    for (int t = threadIdx.x; t < nx * ny * nz; t += blockDim.x) {
      const int i = t % nx, j = (t / nx) % ny, k = t / (nx * ny);
      const long c = nidx(L, i, j, k);
      const double p0 = src[c];
      double v = p0;
      if (!nd_is_dir(L, i, j, k)) {
        double Kp, diag; nd_apply(L, src, i, j, k, Kp, diag);
        if (diag != 0.0) v = p0 + (s < nsp ? (s == 0 ? om1 : om2) : omega) * ((L.b[c] - Kp) / diag);
      }
      dst[c] = v;
    }
    __syncthreads();
    double *tsw = src; src = dst; dst = tsw;
  }
}
DEVI void wg_nd_down(const NLev &F, double *fphi, const NLev &C, double *cphi) {     // residual of F, full weighting into C.b, cphi = 0
  const int nx = F.n[0] + 1, ny = F.n[1] + 1, nz = F.n[2] + 1;
  wg_nd_fill(F, fphi);
  for (int t = threadIdx.x; t < nx * ny * nz; t += blockDim.x) {
    const int i = t % nx, j = (t / nx) % ny, k = t / (nx * ny);
    const long c = nidx(F, i, j, k);
    double r = 0.0;
    if (!nd_is_dir(F, i, j, k)) { double Kp, diag; nd_apply(F, fphi, i, j, k, Kp, diag); r = F.b[c] - Kp; }
    F.res[c] = r;
  }
  __syncthreads();
  wg_nd_fill(F, F.res);
  const int cx = C.n[0] + 1, cy = C.n[1] + 1, cz = C.n[2] + 1;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  for (int t = threadIdx.x; t < cx * cy * cz; t += blockDim.x) {
    const int i = t % cx, j = (t / cx) % cy, k = t / (cx * cy);
    double s = 0.0;
    if (!nd_is_dir(C, i, j, k)) {
      const long f0 = nidx(F, 2 * i, 2 * j, 2 * k);
      s = nd_fw27(F.res + f0, sy, sz);
    }
    const long cn = nidx(C, i, j, k);
    C.b[cn] = s * 0.125;
    cphi[cn] = 0.0;
  }
  __syncthreads();
}
DEVI void wg_nd_up(const NLev &F, double *fphi, const NLev &C, double *cphi) {       // fphi += trilinear interpolation of cphi
  wg_nd_fill(C, cphi);
  const int nx = F.n[0] + 1, ny = F.n[1] + 1, nz = F.n[2] + 1;
  for (int t = threadIdx.x; t < nx * ny * nz; t += blockDim.x) {
    const int i = t % nx, j = (t / nx) % ny, k = t / (nx * ny);
    if (nd_is_dir(F, i, j, k)) continue;
    const long f = nidx(F, i, j, k);
    fphi[f] = fphi[f] + nd_interp8(C, cphi, i >> 1, j >> 1, k >> 1, i & 1, j & 1, k & 1);
  }
  __syncthreads();
}
// (The same with the levels copied into LDS for the duration was measured at 66 us per cycle against 57 us on the L2-resident arrays: a phase costs
// ~2.5 us of dependent instructions of one wave per SIMD, not memory latency.)
__global__ void __launch_bounds__(1024) kk_nd_tailcycle(NdTailArgs T) {
  double *ph[ND_TAIL_MAX], *tm[ND_TAIL_MAX];
  #pragma unroll
  for (int l = 0; l < ND_TAIL_MAX; l++) { ph[l] = T.L[l].phi; tm[l] = T.L[l].tmp; }
  #pragma unroll
  for (int l = 0; l < ND_TAIL_MAX - 1; l++)
    if (l < T.nlev - 1) { wg_nd_jacobi(T.L[l], ph[l], tm[l], T.nu1, T.omega, T.om1, T.om2, T.nsp); wg_nd_down(T.L[l], ph[l], T.L[l + 1], ph[l + 1]); }
  #pragma unroll
  for (int l = 0; l < ND_TAIL_MAX; l++)
    if (l == T.nlev - 1) wg_nd_jacobi(T.L[l], ph[l], tm[l], T.nbot, T.omega);
  #pragma unroll
  for (int l = ND_TAIL_MAX - 2; l >= 0; l--)
    if (l < T.nlev - 1) { wg_nd_up(T.L[l], ph[l], T.L[l + 1], ph[l + 1]); wg_nd_jacobi(T.L[l], ph[l], tm[l], T.nu2, T.omega); }
}

// (round 3, built, measured and removed: the 17^3 .. 65^3 levels as one LDS-tiled launch down -- two Jacobi sweeps on a 16^3 region around a
// 10^3 tile, residual, full weighting -- and one up, the scheme of kk_cc_lds_down / kk_cc_lds_up in mg_cc.hip.  Bit-identical, and no faster:
// 17-38 us down and 14-20 us up per level against six launches of 5-10 us; HG 15.75 ms either way.  The 27-point operator costs ~150 f64
// instructions per node, a tile recomputes 2.9x its own nodes, and a 17^3 level keeps 8 of 256 CUs busy for three dependent sweeps.)
// ---- load / store / divergence -------------------------------------------------------------------------
__global__ void kk_nd_load_sigma(NLev L, FV coeffs, int lo0, int lo1, int lo2) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] || j > L.n[1] || k > L.n[2]) return;
  L.sig[nidx(L, i, j, k)] = fv_get(coeffs, lo0 + i, lo1 + j, lo2 + k);
}
// rh(node) += D u  (definition in oracle/vo_hgproject.c::vo_nd_divu)
DEVI void nd_divu_node(const FV &u, const FV &rh, double fx, double fy, double fz, int i, int j, int k) {
  #define U(a, b, c, m) fv_get(u, i + (a), j + (b), k + (c), m)
  const double dux = (((U(0, 0, 0, 0) + U(0, -1, 0, 0)) + U(0, 0, -1, 0)) + U(0, -1, -1, 0))
                   - (((U(-1, 0, 0, 0) + U(-1, -1, 0, 0)) + U(-1, 0, -1, 0)) + U(-1, -1, -1, 0));
  const double duy = (((U(0, 0, 0, 1) + U(-1, 0, 0, 1)) + U(0, 0, -1, 1)) + U(-1, 0, -1, 1))
                   - (((U(0, -1, 0, 1) + U(-1, -1, 0, 1)) + U(0, -1, -1, 1)) + U(-1, -1, -1, 1));
  const double duz = (((U(0, 0, 0, 2) + U(-1, 0, 0, 2)) + U(0, -1, 0, 2)) + U(-1, -1, 0, 2))
                   - (((U(0, 0, -1, 2) + U(-1, 0, -1, 2)) + U(0, -1, -1, 2)) + U(-1, -1, -1, 2));
  #undef U
  fv_at(rh, i, j, k) = fv_get(rh, i, j, k) + (dux * fx + duy * fy + duz * fz);
}
struct nd_divu_K { FV u; FV rh; double fx; double fy; double fz;
  __device__ void cell(int i, int j, int k) const {
    nd_divu_node(u, rh, fx, fy, fz, i, j, k);
  } };

// nrm[0] = max |rhs|, nrm[1] = max |phi| of the initial guess (zero: the solve may start from a nested iteration, nd_fmg)
__global__ void kk_nd_load(NLev L, FV rh, FV phi, int lo0, int lo1, int lo2, double *nrm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  double rmax = 0.0, pmax = 0.0;
  if (i <= L.n[0] && j <= L.n[1])
    for (int k = blockIdx.z; k <= L.n[2]; k += gridDim.z) {
      const bool dir = nd_is_dir(L, i, j, k);
      const double r = dir ? 0.0 : fv_get(rh, lo0 + i, lo1 + j, lo2 + k);
      const long c = nidx(L, i, j, k);
      const double p0 = dir ? 0.0 : fv_get(phi, lo0 + i, lo1 + j, lo2 + k);
      L.b[c] = -r;
      L.phi[c] = p0;
      rmax = nmax(rmax, fabs(r)); pmax = nmax(pmax, fabs(p0));
    }
  block_atomic_max(nrm, rmax);
  block_atomic_max(nrm + 1, pmax);
}
// the composite solve's coarse correction: the fab holds b itself (the composite residual), the guess is zero -- what kk_nd_load makes of
// rh = -b and phi = 0, bit for bit (a Dirichlet node gets b = -0.0 there: -(0.0)), without the negated copy and the zero-filled phi
__global__ void kk_nd_load_b(NLev L, FV rb, int lo0, int lo1, int lo2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i <= L.n[0] && j <= L.n[1])
    for (int k = blockIdx.z; k <= L.n[2]; k += gridDim.z) {
      const long c = nidx(L, i, j, k);
      L.b[c] = nd_is_dir(L, i, j, k) ? -0.0 : fv_get(rb, lo0 + i, lo1 + j, lo2 + k);
      L.phi[c] = 0.0;
    }
}
// res := b on the nodes of the level (the carrier of the right-hand side's restriction, nd_fmg)
__global__ void kk_nd_copy_b_res(NLev L) {
  NODE_IJK(L)
  if (!in_range) return;
  const long c = nidx(L, i, j, k);
  L.res[c] = L.b[c];
}
__global__ void kk_nd_store(NLev L, FV phi, int lo0, int lo1, int lo2) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] + 1 || j > L.n[1] + 1 || k > L.n[2] + 1) return;
  fv_at(phi, lo0 + i, lo1 + j, lo2 + k) = L.phi[nidx(L, i, j, k)];
}
// the same, and acc += the solution on the nodes of the box (the composite solve adds its coarse correction to phi of level 0)
__global__ void kk_nd_store_add(NLev L, FV phi, FV acc, int lo0, int lo1, int lo2) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] + 1 || j > L.n[1] + 1 || k > L.n[2] + 1) return;
  const double v = L.phi[nidx(L, i, j, k)];
  fv_at(phi, lo0 + i, lo1 + j, lo2 + k) = v;
  if (i >= 0 && i <= L.n[0] && j >= 0 && j <= L.n[1] && k >= 0 && k <= L.n[2]) fv_at(acc, lo0 + i, lo1 + j, lo2 + k) = fv_get(acc, lo0 + i, lo1 + j, lo2 + k) + v;
}

// hgproject's fast path (one level): sigma = 1 / rhohalf written straight into the level on the cells of the box, zero on its ghost cells
// (hg_multigrid.f90:73-79); the ghost cells that have a neighbour or a periodic image then come from the level's sigma halo, exactly as
// multifab_fill_boundary(coeffs) filled them -- instead of coeffs = 1 / rhohalf, its ghost fill and a copy
__global__ void kk_nd_load_sigma_rho(NLev L, FV rhohalf, int lo0, int lo1, int lo2) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1;
  const int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int k = (int)blockIdx.z - 1;
  if (i > L.n[0] || j > L.n[1] || k > L.n[2]) return;
  const bool in = i >= 0 && i < L.n[0] && j >= 0 && j < L.n[1] && k >= 0 && k < L.n[2];
  L.sig[nidx(L, i, j, k)] = in ? 1.0 / fv_get(rhohalf, lo0 + i, lo1 + j, lo2 + k, 0) : 0.0;     // coeffs_K's expression
}
// b = -(0 + D u), phi = 0, max |rhs|: nd_divu_node on a zero rh followed by kk_nd_load on a zero phi, without the two multifabs in between
// Round 4: a k-march.  A thread keeps the values of its column (cells (i, j) and (i, j-1)) of plane k-1 in registers, loads the two of plane k and takes
// the cells (i-1, .) from the lane before it (lane 0 of a row loads them): 6 loads per node instead of 24 (0.38 -> ms at 257^3, the gather form was bound
// by the texture addresser like every other stencil kernel that reads its neighbours through the L1).  The sums are formed in the order of nd_divu_K.
__global__ void __launch_bounds__(256) kk_nd_load_divu(NLev L, FV u, double fx, double fy, double fz, int lo0, int lo1, int lo2, double *nrm) {
  const int lane = threadIdx.x;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int nzp = L.n[2] + 1;
  const int k0 = (int)(((long)blockIdx.z * nzp) / gridDim.z), k1 = (int)(((long)(blockIdx.z + 1) * nzp) / gridDim.z) - 1;
  double rmax = 0.0;
  if (k0 <= k1) {                                                       // uniform
    const int ic = min(i, L.n[0]), jc = min(j, L.n[1]);                 // (lanes beyond the level load a valid column and store nothing)
    const bool act = i <= L.n[0] && j <= L.n[1];
    const long sp = (long)u.n0 * u.n1;
    const double *pc = u.p + fv_idx(u, lo0 + ic, lo1 + jc, lo2 + k0 - 1);   // cell (i, j, k0 - 1) of component 0
    const long dj = -(long)u.n0;
    // [m][0] = u(i, j), [m][1] = u(i, j-1), [m][2] = u(i-1, j), [m][3] = u(i-1, j-1) of the plane below
    double lo[3][4], hi[3][4];
    #define LOADPL(dst, q) { _Pragma("unroll") for (int m = 0; m < 3; m++) { dst[m][0] = (q)[m * u.sc]; dst[m][1] = (q)[m * u.sc + dj]; }                                   \
                             _Pragma("unroll") for (int m = 0; m < 3; m++) { dst[m][2] = lane_prev(dst[m][0]); dst[m][3] = lane_prev(dst[m][1]); }                            \
                             if (lane == 0) { _Pragma("unroll") for (int m = 0; m < 3; m++) { dst[m][2] = (q)[m * u.sc - 1]; dst[m][3] = (q)[m * u.sc + dj - 1]; } } }
    LOADPL(lo, pc)
    const bool dir_ij = (i == 0 && L.dirlo[0]) || (i == L.n[0] && L.dirhi[0]) || (j == 0 && L.dirlo[1]) || (j == L.n[1] && L.dirhi[1]);
    long c = nidx(L, ic, jc, k0);
    const long cz = (long)L.PX * L.PY;
    for (int k = k0; k <= k1; k++, c += cz) {
      pc += sp;
      LOADPL(hi, pc)
      // U(a, b, c, m): a = 0 / -1 -> index 0,1 / 2,3;  b = 0 / -1 -> even / odd index;  c = 0 -> hi, -1 -> lo
      const double dux = (((hi[0][0] + hi[0][1]) + lo[0][0]) + lo[0][1]) - (((hi[0][2] + hi[0][3]) + lo[0][2]) + lo[0][3]);
      const double duy = (((hi[1][0] + hi[1][2]) + lo[1][0]) + lo[1][2]) - (((hi[1][1] + hi[1][3]) + lo[1][1]) + lo[1][3]);
      const double duz = (((hi[2][0] + hi[2][2]) + hi[2][1]) + hi[2][3]) - (((lo[2][0] + lo[2][2]) + lo[2][1]) + lo[2][3]);
      const double rhv = 0.0 + (dux * fx + duy * fy + duz * fz);               // rh (zero) + D u
      const bool dir = dir_ij || (k == 0 && L.dirlo[2]) || (k == L.n[2] && L.dirhi[2]);
      const double r = dir ? 0.0 : rhv;
      if (act) { L.b[c] = -r; L.phi[c] = 0.0; rmax = nmax(rmax, fabs(r)); }
      #pragma unroll
      for (int m = 0; m < 3; m++)
        #pragma unroll
        for (int t = 0; t < 4; t++) lo[m][t] = hi[m][t];
    }
    #undef LOADPL
  }
  block_atomic_max(nrm, rmax);
}

// ---- gather of the first agglomerated level (see mg_cc.hip: same scheme, nodes instead of cells) --------------------
struct NGBox { int c0[3]; int n[3]; long off; };     // n = coarse CELLS of the box; nodes are n+1

__global__ void kk_nd_restrict_pack(NLev F, NLev Cf /* flags + extents of the coarse box */, double *buf, long off) {
  NODE_IJK(Cf)
  if (!in_range) return;
  double s = 0.0;
  if (!nd_is_dir(Cf, i, j, k)) {
    const long sy = F.PX, sz = (long)F.PX * F.PY;
    const long f0 = nidx(F, 2 * i, 2 * j, 2 * k);
    s = nd_fw27(F.res + f0, sy, sz);
  }
  buf[off + i + (long)(Cf.n[0] + 1) * (j + (long)(Cf.n[1] + 1) * k)] = s * 0.125;
}
__global__ void kk_nd_coarsen_sigma_pack(NLev F, double *buf, long off, int nx, int ny, int nz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  const int k = blockIdx.z;
  if (i >= nx || j >= ny || k >= nz) return;
  const long sy = F.PX, sz = (long)F.PX * F.PY;
  const long f = nidx(F, 2 * i, 2 * j, 2 * k);
  double s = 0.0;
  #pragma unroll
  for (int c = 0; c < 2; c++)
    #pragma unroll
    for (int b = 0; b < 2; b++)
      #pragma unroll
      for (int a = 0; a < 2; a++) s = s + F.sig[f + a + b * sy + c * sz];
  buf[off + i + (long)nx * (j + (long)ny * k)] = s * 0.125;
}
__global__ void kk_nd_unpack(NLev T, double *dst, const double *buf, const NGBox *gb, int nodal) {
  const NGBox g = gb[blockIdx.z];
  const int ex = g.n[0] + nodal, ey = g.n[1] + nodal, ez = g.n[2] + nodal;
  const int tot = ex * ey * ez;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < tot; t += gridDim.x * blockDim.x) {
    const int i = t % ex, j = (t / ex) % ey, k = t / (ex * ey);
    dst[nidx(T, g.c0[0] + i, g.c0[1] + j, g.c0[2] + k)] = buf[g.off + t];
  }
}
__global__ void kk_nd_prolong_tail(NLev F, NLev T, int c00, int c01, int c02, int f00, int f01, int f02) {
  NODE_IJK(F)
  if (!in_range) return;
  if (nd_is_dir(F, i, j, k)) return;
  // global fine node = f0 + (i,j,k); the box origin is even, so parity and halving are local
  (void)f00; (void)f01; (void)f02;
  const long f = nidx(F, i, j, k);
  F.phi[f] = F.phi[f] + nd_interp8(T, T.phi, c00 + (i >> 1), c01 + (j >> 1), c02 + (k >> 1), i & 1, j & 1, k & 1);
}

// ---- host ---------------------------------------------------------------------------------------------------
static const dim3 NBLK(64, 4, 1);
static dim3 ng3(int nx, int ny, int nz) { return dim3((nx + 63) / 64, (ny + 3) / 4, nz); }

// tiles and k-slabs of the paired march over `nzu` k-units (planes of the level for a sweep, coarse planes for the fused residual + restriction)
static NdPairGrid nd_pair_grid(const NLev &L, int rows, int nzu, bool use_rem, int minwg, int kc_env) {
  const int npair = (L.n[0] + 2) / 2;
  NdPairGrid G;
  G.gxm = npair / 62; G.gy = (L.n[1] + rows) / rows;
  const int rem = npair - 62 * G.gxm;
  G.lwr = 6; G.gyr = 0;
  if (rem > 0 && use_rem) { G.lwr = 2; while ((1 << G.lwr) - 2 < rem) G.lwr++; G.gyr = (L.n[1] + (rows << (6 - G.lwr))) / (rows << (6 - G.lwr)); }
  else if (rem > 0) G.gxm++;
  const int tiles = G.gxm * G.gy + G.gyr;
  int kc = nzu;
  while (kc > 8 && tiles * ((nzu + kc - 1) / kc) < minwg) kc = (kc + 1) / 2;
  if (kc_env > 0) kc = std::min(kc_env, nzu);
  G.gz = std::max(1, (nzu + kc - 1) / kc);      // balanced slabs of at most kc planes (257 planes: 16 slabs of 16 or 17 -- measured 0.1396 ms against 0.1443 with 15 slabs)
  G.nmain = G.gxm * G.gy * G.gz;
  G.rev = 0;
  return G;
}
// slab thickness: enough workgroups to fill 256 CUs several times over, yet long enough marches to amortise the
// two warm-up planes (overhead 2/kchunk)
// rev: the tiles in reverse order.  Consecutive marches of a level alternate (NDLev::rev): a sweep reads what the previous one wrote and the same sigma and
// right-hand side, and the planes that one touched last are the ones still in the 256 MB Infinity Cache (same bits: a Jacobi sweep has no order)
template <int MODE> static void nd_launch_march(const NLev &L, const double *phi, double *out, double *nrm, int shell_later = 0, int rev = 0) {
  const int nzp = L.n[2] + 1;
  const int tiles = ((L.n[0] + 62) / 62) * ((L.n[1] + 4) / 4);
  int kchunk = nzp;
  while (kchunk > 8 && tiles * ((nzp + kchunk - 1) / kchunk) < 2048) kchunk = (kchunk + 1) / 2;
  const int nch = (nzp + kchunk - 1) / kchunk;
  static const bool paired = !(vdn_env("VDN_ND_PAIR") && atoi(vdn_env("VDN_ND_PAIR")) == 0);
  if (paired && L.n[0] >= 127) {                   // 124 nodes per wave row
    const int rows = 4;                          // (measured: 8 rows per workgroup 17.1 -> 18.7 ms of HG per step, 16 rows spill)
    const bool use_rem = true; const int minwg = 2048, kc_env = 0;
    NdPairGrid G = nd_pair_grid(L, rows, nzp, use_rem, minwg, kc_env);
    static const bool flip = !(vdn_env("VDN_ND_REV") && atoi(vdn_env("VDN_ND_REV")) == 0);
    G.rev = flip ? rev : 0;
    hipLaunchKernelGGL((kk_nd_march_pair<MODE, 4>), dim3(G.nmain + G.gyr * G.gz), NBLK, 0, ctx().stream, L, phi, out, nd_cur_omega(), G, nrm, shell_later);
    return;
  }
  hipLaunchKernelGGL(kk_nd_march<MODE>, dim3((L.n[0] + 62) / 62, (L.n[1] + 4) / 4, nch), NBLK, 0, ctx().stream, L, phi, out, nd_cur_omega(), kchunk, nrm, shell_later);
}

struct NBox { NLev L; int lo[3]; int hmask = 63; XPlan *hA = nullptr, *hB = nullptr; double *A = nullptr, *B = nullptr; };
struct NDLev { std::vector<NBox> boxes; XPlan *halo_A = nullptr, *halo_B = nullptr, *halo_res = nullptr, *halo_sig = nullptr; int ng[3]; bool flip = false; bool single_box = false; int per[3] = {0, 0, 0};
               int rev = 0; /* tile order of the next march (nd_launch_march) */
               bool res_restricted = false; /* the last residual pass left the x- and z-sums of the full weighting in res (kk_nd_march_pair_rst): nd_restrict_down finishes along y */ };
struct NDMG {
  std::vector<NDLev> dlev; std::vector<NLev> tail; int per[3]; double *d_nrm;
  std::vector<NGBox> gb; NGBox *d_gb = nullptr;
  double *sendbuf = nullptr, *recvbuf = nullptr; size_t cnt_nodes = 0, cnt_cells = 0;
  std::vector<long> loc_off_nodes, loc_off_cells;
};

struct NdHaloKey { unsigned long uid; const void *p0; int lev, l, which, per; unsigned long long sig; bool operator<(const NdHaloKey &o) const {
  return std::tie(uid, p0, lev, l, which, per, sig) < std::tie(o.uid, o.p0, o.lev, o.l, o.which, o.per, o.sig); } };      // sig: every local box's address (see mg_cc.hip HaloKey)
static std::map<NdHaloKey, XPlan *> g_nd_halo_cache;
void nd_halo_cache_purge(unsigned long uid) {
  for (auto it = g_nd_halo_cache.begin(); it != g_nd_halo_cache.end();) { if (it->first.uid == uid) it = g_nd_halo_cache.erase(it); else ++it; }
}

// ghost nodes (index -1 or n+1 in some direction) of up to three node arrays := 0
__global__ void kk_nd_zero_shell(NLev L, double *a0, double *a1, double *a2) {
  const int u = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, v = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  const int face = blockIdx.z, d = face >> 1, side = face & 1;
  const int da = d == 0 ? 1 : 0, db = d == 2 ? 1 : 2;
  if (u > L.n[da] + 1 || v > L.n[db] + 1) return;
  int q[3]; q[d] = side ? L.n[d] + 1 : -1; q[da] = u; q[db] = v;
  const long c = nidx(L, q[0], q[1], q[2]);
  a0[c] = 0.0; if (a1) a1[c] = 0.0; if (a2) a2[c] = 0.0;
}
static NLev nd_alloc_lev(const int n[3], const double h[3]) {
  NLev L;
  for (int d = 0; d < 3; d++) { L.n[d] = n[d]; L.f[d] = 1.0 / (36.0 * (h[d] * h[d])); L.dirlo[d] = L.dirhi[d] = L.per[d] = 0; }
  L.PX = ((n[0] + 18 + 15) / 16) * 16; L.PY = n[1] + 3; L.sz = (long)L.PX * L.PY * (n[2] + 3);
  double *base = (double *)arena_alloc(sizeof(double) * L.sz * 5);
  L.phi = base; L.tmp = base + L.sz; L.b = base + 2 * L.sz; L.res = base + 3 * L.sz; L.sig = base + 4 * L.sz;
  // What must be zero is the ghost layer of phi, tmp and res (nodes outside a physical face; the exchange overwrites the others) and of sigma
  // (cells beyond a physical face: a COARSE level's sigma is written on the cells 0..n-1 only): every node 0..n of phi, tmp, res is written
  // before it is read (phi by the load / the restriction, tmp by the first sweep, res by the residual) and b is read on nodes 0..n only.  On a
  // big level that is a 6-face kernel and one array fill instead of a fill of five arrays (780 MB at 257^3, 0.16 ms per solve).
  // Row padding beyond the ghost nodes only ever reaches lanes whose results are discarded.
  // (The first version left sigma to its load as well -- true on the finest level only: the 129^3 level of a 257^3 solve then read
  // whatever the arena held beyond the walls, which happened to be zeros until hgproject stopped allocating its multifabs in front of it.)
  static const bool lean_on = !(vdn_env("VDN_ND_LEAN") && atoi(vdn_env("VDN_ND_LEAN")) == 0);
  if (lean_on && (long)(n[0] + 1) * (n[1] + 1) * (n[2] + 1) >= (1L << 21)) {
    const int m = std::max(n[0], std::max(n[1], n[2])) + 3;
    hipLaunchKernelGGL(kk_nd_zero_shell, dim3((m + 63) / 64, (m + 3) / 4, 6), dim3(64, 4, 1), 0, ctx().stream, L, L.phi, L.tmp, L.res);
    HIPCHK(hipMemsetAsync(L.sig, 0, sizeof(double) * L.sz, ctx().stream));
  } else HIPCHK(hipMemsetAsync(base, 0, sizeof(double) * L.sz * 5, ctx().stream));
  return L;
}
static FV nd_view(const NLev &L, double *p, const int lo[3], int extra /* 3 for nodes, 2 for cells */) {
  FV f; f.p = p; f.a0 = lo[0] - 16; f.a1 = lo[1] - 1; f.a2 = lo[2] - 1; f.n0 = L.PX; f.n1 = L.PY; f.n2 = L.n[2] + extra; f.sc = L.sz;
  return f;
}

static void nd_build(NDMG &M, const vdn_multifab *coeffs, const double *dx, const int bc[3][2]) {
  Prof prof_("nd_build");
  const vdn_layout *la = coeffs->la; const int lev = coeffs->lev;
  const auto &gboxes = la->boxes[lev];
  const int nb = (int)gboxes.size();
  for (int d = 0; d < 3; d++) M.per[d] = (bc[d][0] == VDN_BC_PER);
  int bn[3]; for (int d = 0; d < 3; d++) bn[d] = gboxes[0].hi[d] - gboxes[0].lo[d] + 1;
  for (const auto &b : gboxes) for (int d = 0; d < 3; d++) {
    REQUIRE(b.hi[d] - b.lo[d] + 1 == bn[d], "nodal multigrid: all boxes of a level must have the same size");
    REQUIRE((b.lo[d] - la->pd[lev].lo[d]) % bn[d] == 0, "nodal multigrid: boxes must be aligned to their size");
  }
  std::vector<int> nloc_of(ctx().nranks, 0);
  for (int g = 0; g < nb; g++) nloc_of[la->owner[lev][g]]++;
  int maxloc = 0; for (int r = 0; r < ctx().nranks; r++) maxloc = std::max(maxloc, nloc_of[r]);
  const int perbits = M.per[0] | (M.per[1] << 1) | (M.per[2] << 2);
  int n[3] = { bn[0], bn[1], bn[2] }; double h[3] = { dx[0], dx[1], dx[2] };
  int scale = 1;
  for (;;) {
    NDLev DL;
    std::vector<XBoxInfo> xa, xb2, xr, xs;
    vdn_box lpd; for (int d = 0; d < 3; d++) { lpd.lo[d] = 0; lpd.hi[d] = (la->pd[lev].hi[d] - la->pd[lev].lo[d] + 1) / scale - 1; DL.ng[d] = lpd.hi[d] + 1; }
    for (int g = 0; g < nb; g++) {
      XBoxInfo x; memset(&x, 0, sizeof x);
      int lo[3];
      for (int d = 0; d < 3; d++) { lo[d] = (gboxes[g].lo[d] - la->pd[lev].lo[d]) / scale; x.vlo[d] = lo[d]; x.vhi[d] = lo[d] + n[d]; }   // nodes lo..lo+n
      x.owner = la->owner[lev][g];
      XBoxInfo xc = x; for (int d = 0; d < 3; d++) xc.vhi[d] = lo[d] + n[d] - 1;                                                   // cells
      XBoxInfo xA = x, xB = x, xR = x;
      if (x.owner == ctx().rank) {
        NBox B; B.L = nd_alloc_lev(n, h); for (int d = 0; d < 3; d++) B.lo[d] = lo[d];
        for (int d = 0; d < 3; d++) {          // Dirichlet (outflow) flags only on DOMAIN faces
          B.L.dirlo[d] = (lo[d] == 0 && bc[d][0] == VDN_BC_DIR); B.L.dirhi[d] = (lo[d] + n[d] == lpd.hi[d] + 1 && bc[d][1] == VDN_BC_DIR);
        }
        B.hmask = 0;
        for (int d = 0; d < 3; d++) {
          if (lo[d] > 0 || M.per[d]) B.hmask |= 1 << (2 * d);
          if (lo[d] + n[d] < lpd.hi[d] + 1 || M.per[d]) B.hmask |= 2 << (2 * d);
        }
        B.A = B.L.phi; B.B = B.L.tmp;
        xA.fv = nd_view(B.L, B.A, lo, 3); xB.fv = nd_view(B.L, B.B, lo, 3); xR.fv = nd_view(B.L, B.L.res, lo, 3); xc.fv = nd_view(B.L, B.L.sig, lo, 3);
        DL.boxes.push_back(B);
      }
      xa.push_back(xA); xb2.push_back(xB); xr.push_back(xR); xs.push_back(xc);
    }
    if (nb > 1 || perbits) {
      const void *p0 = DL.boxes.empty() ? nullptr : (const void *)DL.boxes[0].A;
      GraphKey hk; for (const NBox &B : DL.boxes) { hk.put(B.A); hk.put(B.L.sz); }
      auto get = [&](int which, const std::vector<XBoxInfo> &xv, const vdn_box &pdm) {
        NdHaloKey key{ la->uid, p0, lev, (int)M.dlev.size(), which, perbits, hk.h };
        auto it = g_nd_halo_cache.find(key);
        if (it == g_nd_halo_cache.end()) { XPlan *P = xplan_build(xv, pdm, M.per, 1, 1); halo_cache_register(la->uid, P); it = g_nd_halo_cache.emplace(key, P).first; }
        return it->second;
      };
      // the node "domain" for periodic shifts has the CELL period (node n is node 0): use the cell domain box
      DL.halo_A = get(0, xa, lpd); DL.halo_B = get(1, xb2, lpd); DL.halo_res = get(2, xr, lpd); DL.halo_sig = get(3, xs, lpd);
    }
    DL.single_box = (nb == 1); for (int d = 0; d < 3; d++) DL.per[d] = M.per[d];
    M.dlev.push_back(DL);
    bool can = true, next_dist = true;
    for (int d = 0; d < 3; d++) { const int N = lpd.hi[d] + 1; if ((N & 1) || N <= 2) can = false; }
    if (can) for (int d = 0; d < 3; d++) REQUIRE(!(n[d] & 1), "nodal multigrid: box extent %d is odd while the domain can still be coarsened", n[d]);
    // several boxes: stop exchanging halos once the boxes get small (below 64 cells) -- every level that stays distributed costs
    // ~10 latency-bound halo exchanges per V-cycle, the replicated tail below a 64^3-per-box level costs microseconds per pass
    const int agglom = mg_agglom(la, lev);            // (mg_cc.hip: 64 across ranks, 128 where every box is this rank's)
    const int min_dist = nb > 1 ? agglom : 4;
    for (int d = 0; d < 3; d++) if (n[d] / 2 < min_dist || ((n[d] / 2) & 1)) next_dist = false;
    if (!can) break;
    if (!next_dist) {
      int tn[3]; double th[3]; int cn[3];
      for (int d = 0; d < 3; d++) { cn[d] = n[d] / 2; tn[d] = (lpd.hi[d] + 1) / 2; th[d] = h[d] * 2.0; }
      for (;;) {
        NLev T = nd_alloc_lev(tn, th);
        for (int d = 0; d < 3; d++) { T.dirlo[d] = (bc[d][0] == VDN_BC_DIR); T.dirhi[d] = (bc[d][1] == VDN_BC_DIR); T.per[d] = M.per[d]; }
        M.tail.push_back(T);
        bool c2 = true;
        for (int d = 0; d < 3; d++) if ((tn[d] & 1) || tn[d] <= 2) c2 = false;
        if (!c2 || M.tail.size() >= 31) break;
        for (int d = 0; d < 3; d++) { tn[d] /= 2; th[d] *= 2.0; }
      }
      const long per_nodes = (long)(cn[0] + 1) * (cn[1] + 1) * (cn[2] + 1), per_cells = (long)cn[0] * cn[1] * cn[2];
      M.cnt_nodes = (size_t)per_nodes * maxloc; M.cnt_cells = (size_t)per_cells * maxloc;
      std::vector<int> seen(ctx().nranks, 0);
      std::vector<NGBox> gbn, gbc;
      for (int g = 0; g < nb; g++) {
        const int r = la->owner[lev][g], l = seen[r]++;
        NGBox a, c;
        for (int d = 0; d < 3; d++) { a.c0[d] = c.c0[d] = (gboxes[g].lo[d] - la->pd[lev].lo[d]) / scale / 2; a.n[d] = c.n[d] = cn[d]; }
        a.off = (long)r * M.cnt_nodes + (long)l * per_nodes; c.off = (long)r * M.cnt_cells + (long)l * per_cells;
        gbn.push_back(a); gbc.push_back(c);
        if (r == ctx().rank) { M.loc_off_nodes.push_back((long)l * per_nodes); M.loc_off_cells.push_back((long)l * per_cells); }
      }
      M.gb = gbn; M.gb.insert(M.gb.end(), gbc.begin(), gbc.end());      // [nodes... | cells...]
      M.d_gb = (NGBox *)arena_alloc(2 * nb * sizeof(NGBox));
      HIPCHK(hipMemcpyAsync(M.d_gb, M.gb.data(), 2 * nb * sizeof(NGBox), hipMemcpyHostToDevice, ctx().stream));
      HIPCHK(hipStreamSynchronize(ctx().stream));
      M.sendbuf = (double *)arena_alloc(sizeof(double) * M.cnt_nodes);
      M.recvbuf = (double *)arena_alloc(sizeof(double) * M.cnt_nodes * ctx().nranks);
      break;
    }
    for (int d = 0; d < 3; d++) { n[d] /= 2; h[d] *= 2.0; }
    scale *= 2;
    if (M.dlev.size() >= 31) break;
  }
  M.d_nrm = (double *)arena_alloc(256);
}

// ---- distributed levels ------------------------------------------------------------------------------------------
static void nd_halo_phi(NDLev &DL) { XPlan *P = DL.flip ? DL.halo_B : DL.halo_A; if (P) xplan_run(P); }
// the halo of phi next to the sweep that reads it (see kk_nd_shell): when part of it comes from another rank (or VDN_OVERLAP=1, the
// one-GPU rehearsal) the exchange runs on ctx().halo_stream and nd_halo_begin returns true -- the caller launches its marches, calls
// nd_halo_end (the launch stream waits for the halo) and recomputes the face nodes; otherwise the exchange is done here, in line
static bool nd_halo_begin(NDLev &DL) {
  XPlan *P = DL.flip ? DL.halo_B : DL.halo_A;
  if (!P) return false;
  static const int ov_env = vdn_env("VDN_OVERLAP") ? atoi(vdn_env("VDN_OVERLAP")) : -1;
  static const long ov_min = 1L << 20;     // see cc_gsrb_d
  long nodes = 0;
  for (const NBox &B : DL.boxes) nodes = std::max(nodes, (long)B.L.n[0] * B.L.n[1] * B.L.n[2]);
  if (!(ov_env == 1 || (ov_env != 0 && xplan_has_remote(P) && nodes >= ov_min))) { xplan_run(P); return false; }
  VdnCtx &c = ctx();
  HIPCHK(hipEventRecord(c.ev_main, c.stream));                      // the phi the halo is packed from is complete
  HIPCHK(hipStreamWaitEvent(c.halo_stream, c.ev_main, 0));
  xplan_run(P, c.halo_stream);
  HIPCHK(hipEventRecord(c.ev_halo, c.halo_stream));
  return true;
}
static void nd_halo_end() { VdnCtx &c = ctx(); HIPCHK(hipStreamWaitEvent(c.stream, c.ev_halo, 0)); }
// levels of at most 9^3 nodes held in ONE box: all sweeps in a single one-workgroup launch (launch-latency bound otherwise)
static const long SMALL_LEVEL_NODES = 9L * 9 * 9;
// The damping of a run of sweeps.  `pre` = the nu1 pre-smoothing sweeps of a V-cycle: with nu1 = 2 and vdn_params.hg_omega_pre1 / 2 > 0 the first is
// damped by pre1 and the second by pre2 (oracle: nd_presmooth); every other sweep by hg_omega.
struct NdOm { double om, om1, om2; int nsp; double at(int s) const { return s < nsp ? (s == 0 ? om1 : om2) : om; } };
// The multi-step sets were tuned for dx = dy = dz; with spacings more than a quarter apart they lose to the plain hg_omega or diverge (oracle, round 4:
// 50 against 38 cycles at dz = 2 dx), so a solve on such a grid keeps hg_omega (oracle: vo_nd_isotropic).  Set on entry of nd_solve / ml_nd_solve.
static bool g_nd_iso = true;
static bool nd_isotropic(const double *dx) {
  const double lo = std::min(dx[0], std::min(dx[1], dx[2])), hi = std::max(dx[0], std::max(dx[1], dx[2]));
  return hi <= 1.25 * lo;
}
static NdOm nd_om(bool pre, int nsweeps) {
  const vdn_params &P = ctx().prm;
  NdOm o{ P.hg_omega, 0.0, 0.0, 0 };
  if (g_nd_iso && pre && nsweeps == 2 && P.hg_nu1 == 2 && P.hg_omega_pre1 > 0.0 && P.hg_omega_pre2 > 0.0) { o.om1 = P.hg_omega_pre1; o.om2 = P.hg_omega_pre2; o.nsp = 2; }
  return o;
}
static void nd_jacobi_d(NDLev &DL, int nsweeps, bool pre = false) {
  const NdOm om = nd_om(pre, nsweeps);
  NdOmegaScope scope_;
  if (DL.single_box && DL.boxes.size() == 1 && (long)(DL.ng[0] + 1) * (DL.ng[1] + 1) * (DL.ng[2] + 1) <= SMALL_LEVEL_NODES) {
    NBox &B = DL.boxes[0];
    NLev Lp = B.L;                      // the kernel refreshes periodic images itself: give it the periodicity flags
    for (int d = 0; d < 3; d++) Lp.per[d] = DL.per[d];
    hipLaunchKernelGGL(kk_nd_bottom, dim3(1), dim3(1024), 0, ctx().stream, Lp, B.L.phi, B.L.tmp, nsweeps, om.om, om.om1, om.om2, om.nsp);
    if (nsweeps & 1) { std::swap(B.L.phi, B.L.tmp); DL.flip = !DL.flip; }
    return;
  }
  for (int s = 0; s < nsweeps; s++) {
    g_nd_omega_now = om.at(s);
    const bool ov = nd_halo_begin(DL);
    for (NBox &B : DL.boxes) nd_launch_march<0>(B.L, B.L.phi, B.L.tmp, nullptr, 0, DL.rev);
    DL.rev ^= 1;
    if (ov) {
      nd_halo_end();
      for (NBox &B : DL.boxes) nd_launch_shell<0>(B.L, B.L.phi, B.L.tmp, nullptr, B.hmask);
    }
    for (NBox &B : DL.boxes) std::swap(B.L.phi, B.L.tmp);
    DL.flip = !DL.flip;
  }
}
static void nd_residual_d(NDMG &M, NDLev &DL, bool norm, bool reduce = true) {     // reduce = false: the norm stays rank-local (it goes into the norm history, made global when that is read)
  if (norm) HIPCHK(hipMemsetAsync(M.d_nrm, 0, sizeof(double), ctx().stream));
  DL.res_restricted = false;
  {   // a wide one-box level without periodic images whose next level is one box too: residual and the x / z part of the restriction in one march
    static const bool fuse = !(vdn_env("VDN_ND_RESTRICT_FUSED") && atoi(vdn_env("VDN_ND_RESTRICT_FUSED")) == 0);
    static const bool paired = !(vdn_env("VDN_ND_PAIR") && atoi(vdn_env("VDN_ND_PAIR")) == 0);
    const size_t l = &DL - &M.dlev[0];
    if (fuse && paired && DL.single_box && DL.boxes.size() == 1 && !DL.halo_res && !(DL.per[0] || DL.per[1] || DL.per[2]) && l + 1 < M.dlev.size() && M.dlev[l + 1].boxes.size() == 1) {
      const NLev &L = DL.boxes[0].L;
      if (L.n[0] >= 127 && L.n[0] % 2 == 0 && L.n[1] % 2 == 0 && L.n[2] % 2 == 0) {
        nd_halo_phi(DL);                                                  // (no neighbour, no image: nothing to exchange; kept for symmetry with the plain path)
        NdPairGrid G = nd_pair_grid(L, 4, L.n[2] / 2 + 1, true, 2048, 0);
        static const bool flip = !(vdn_env("VDN_ND_REV") && atoi(vdn_env("VDN_ND_REV")) == 0);
        G.rev = flip ? DL.rev : 0; DL.rev ^= 1;
        hipLaunchKernelGGL((kk_nd_march_pair_rst<4>), dim3(G.nmain + G.gyr * G.gz), NBLK, 0, ctx().stream, L, (const double *)L.phi, L.res, G, norm ? M.d_nrm : nullptr);
        DL.res_restricted = true;
        if (norm && reduce) comm_allreduce_max_dev(M.d_nrm, 1);
        return;
      }
    }
  }
  const bool ov = nd_halo_begin(DL);
  for (NBox &B : DL.boxes)
    nd_launch_march<1>(B.L, B.L.phi, B.L.res, norm ? M.d_nrm : nullptr, ov ? B.hmask : 0, DL.rev);
  DL.rev ^= 1;
  if (ov) {
    nd_halo_end();
    for (NBox &B : DL.boxes) nd_launch_shell<1>(B.L, B.L.phi, B.L.res, norm ? M.d_nrm : nullptr, B.hmask);
  }
  if (DL.halo_res) xplan_run(DL.halo_res);
  if (norm && reduce) comm_allreduce_max_dev(M.d_nrm, 1);
}

// ---- replicated tail ------------------------------------------------------------------------------------------------
static void nd_fill_nodes(const NLev &L, double *a) {
  if (!(L.per[0] || L.per[1] || L.per[2])) return;     // ghosts stay zero: set once by the setup memset, never written
  hipLaunchKernelGGL(kk_nd_fill_nodes, ng3(L.n[0] + 3, L.n[1] + 3, L.n[2] + 3), NBLK, 0, ctx().stream, L, a);
}
static void nd_jacobi_t(NLev &L, int nsweeps, bool pre = false) {
  const NdOm om = nd_om(pre, nsweeps);
  NdOmegaScope scope_;
  if ((long)(L.n[0] + 1) * (L.n[1] + 1) * (L.n[2] + 1) <= SMALL_LEVEL_NODES) {
    hipLaunchKernelGGL(kk_nd_bottom, dim3(1), dim3(1024), 0, ctx().stream, L, L.phi, L.tmp, nsweeps, om.om, om.om1, om.om2, om.nsp);
    if (nsweeps & 1) std::swap(L.phi, L.tmp);
    return;
  }
  for (int s = 0; s < nsweeps; s++) {
    g_nd_omega_now = om.at(s);
    nd_fill_nodes(L, L.phi);
    nd_launch_march<0>(L, L.phi, L.tmp, nullptr);
    std::swap(L.phi, L.tmp);
  }
}
static void nd_bottom_t(NLev &L) {          // max(nub, 2 N^2) sweeps (same rule as the oracle)
  const int N = std::max(L.n[0], std::max(L.n[1], L.n[2]));
  const int ns = std::max(ctx().prm.hg_nub, 2 * N * N);
  hipLaunchKernelGGL(kk_nd_bottom, dim3(1), dim3(1024), 0, ctx().stream, L, L.phi, L.tmp, ns, ctx().prm.hg_omega);
  if (ns & 1) std::swap(L.phi, L.tmp);
}
static bool nd_small_end(NDMG &M, int dl, int tl);
static void nd_vcycle_t(NDMG &M, int l) {
  const vdn_params &P = ctx().prm;
  NLev &L = M.tail[l];
  // phi = 0 on entry: deeper tail levels get it from kk_nd_restrict, the first one is filled by the generic gather/unpack
  if (l == 0) HIPCHK(hipMemsetAsync(L.phi, 0, sizeof(double) * L.sz, ctx().stream));
  if (nd_small_end(M, -1, l)) return;
  if (l == (int)M.tail.size() - 1) { nd_bottom_t(L); return; }
  NLev &C = M.tail[l + 1];
  nd_jacobi_t(L, P.hg_nu1, true);
  nd_fill_nodes(L, L.phi);
  nd_launch_march<1>(L, L.phi, L.res, nullptr);
  nd_fill_nodes(L, L.res);
  hipLaunchKernelGGL(kk_nd_restrict, ng3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1), NBLK, 0, ctx().stream, L, C);
  nd_vcycle_t(M, l + 1);
  nd_fill_nodes(C, C.phi);
  nd_launch_prolong(L, C, 0, 0, 0);
  nd_jacobi_t(L, P.hg_nu2);
}

static void nd_restrict_down(NDMG &M, int l) {
  NDLev &DL = M.dlev[l];
  if (l + 1 < (int)M.dlev.size()) {
    NDLev &DC = M.dlev[l + 1];
    for (size_t b = 0; b < DL.boxes.size(); b++) {
      NLev &C = DC.boxes[b].L;
      if (DL.res_restricted) hipLaunchKernelGGL(kk_nd_rst_y, ng3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1), NBLK, 0, ctx().stream, DL.boxes[b].L, (const double *)DL.boxes[b].L.res, C);
      else hipLaunchKernelGGL(kk_nd_restrict, ng3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1), NBLK, 0, ctx().stream, DL.boxes[b].L, C);
    }
    DL.res_restricted = false;
  } else {
    NLev &T = M.tail[0];
    const int nb = (int)M.gb.size() / 2;
    for (size_t b = 0; b < DL.boxes.size(); b++) {
      const NLev &F = DL.boxes[b].L;
      NLev Cf = F;                       // extents + Dirichlet flags of the coarse box
      for (int d = 0; d < 3; d++) Cf.n[d] = F.n[d] / 2;
      hipLaunchKernelGGL(kk_nd_restrict_pack, ng3(Cf.n[0] + 1, Cf.n[1] + 1, Cf.n[2] + 1), NBLK, 0, ctx().stream, F, Cf, M.sendbuf, M.loc_off_nodes[b]);
    }
    comm_allgather_dev(M.sendbuf, M.recvbuf, M.cnt_nodes);
    hipLaunchKernelGGL(kk_nd_unpack, dim3(4, 1, (unsigned)nb), dim3(256), 0, ctx().stream, T, T.b, M.recvbuf, M.d_gb, 1);
  }
}
static void nd_prolong_up(NDMG &M, int l) {
  NDLev &DL = M.dlev[l];
  if (l + 1 < (int)M.dlev.size()) { NDLev &DC = M.dlev[l + 1]; nd_halo_phi(DC); }
  else nd_fill_nodes(M.tail[0], M.tail[0].phi);
  for (size_t b = 0; b < DL.boxes.size(); b++) {
    NBox &B = DL.boxes[b];
    if (l + 1 < (int)M.dlev.size())
      nd_launch_prolong(B.L, M.dlev[l + 1].boxes[b].L, 0, 0, 0);
    else
      nd_launch_prolong(B.L, M.tail[0], B.lo[0] / 2, B.lo[1] / 2, B.lo[2] / 2);
  }
}
static int nd_bottom_sweeps_global(const NDLev &DL) {
  const int N = std::max(DL.ng[0], std::max(DL.ng[1], DL.ng[2]));
  return std::max(ctx().prm.hg_nub, 2 * N * N);
}
// The small end of the hierarchy in one launch (kk_nd_tailcycle): distributed levels dl .. end when they are one box of at most 9^3 nodes
// each (dl < 0: none), then the replicated tail levels tl .. end (one rank and one box: the gather between the two is the plain restriction).
static bool nd_small_end(NDMG &M, int dl, int tl) {
  static const bool on = !(vdn_env("VDN_MG_TAILCYCLE") && atoi(vdn_env("VDN_MG_TAILCYCLE")) == 0);
  if (!on) return false;
  static const long tail_nodes = SMALL_LEVEL_NODES;    // largest level the one-workgroup cycle takes (measured: 17^3 is slower, HG 16.9 -> 17.6 ms)
  const vdn_params &P = ctx().prm;
  NdTailArgs T; memset(&T, 0, sizeof T);
  int nl = 0;
  if (dl >= 0) {
    if (!M.tail.empty() && !(ctx().nranks == 1 && M.dlev.back().single_box)) return false;
    for (int m = dl; m < (int)M.dlev.size(); m++) {
      const NDLev &D = M.dlev[m];
      if (!(D.single_box && D.boxes.size() == 1 && (long)(D.ng[0] + 1) * (D.ng[1] + 1) * (D.ng[2] + 1) <= tail_nodes) || nl == ND_TAIL_MAX) return false;
      T.L[nl] = D.boxes[0].L; for (int d = 0; d < 3; d++) T.L[nl].per[d] = D.per[d];
      nl++;
    }
  }
  for (int m = tl; m < (int)M.tail.size(); m++) {
    const NLev &L = M.tail[m];
    if ((long)(L.n[0] + 1) * (L.n[1] + 1) * (L.n[2] + 1) > tail_nodes || nl == ND_TAIL_MAX) return false;
    T.L[nl++] = L;
  }
  if (nl < 2) return false;
  const NLev &B = T.L[nl - 1];
  const int N = std::max(B.n[0], std::max(B.n[1], B.n[2]));
  T.nlev = nl; T.nu1 = P.hg_nu1; T.nu2 = P.hg_nu2; T.nbot = std::max(P.hg_nub, 2 * N * N); T.omega = P.hg_omega; { const NdOm o = nd_om(true, P.hg_nu1); T.om1 = o.om1; T.om2 = o.om2; T.nsp = o.nsp; }     // nd_bottom_t / nd_bottom_sweeps_global
  hipLaunchKernelGGL(kk_nd_tailcycle, dim3(1), dim3(1024), 0, ctx().stream, T);
  int m = 0;                                           // the ping-pong state the sweeps leave behind (nd_jacobi_d / nd_jacobi_t)
  if (dl >= 0) for (int q = dl; q < (int)M.dlev.size(); q++, m++) {
    const int sweeps = (m == nl - 1) ? T.nbot : T.nu1 + T.nu2;
    if (sweeps & 1) { NDLev &D = M.dlev[q]; std::swap(D.boxes[0].L.phi, D.boxes[0].L.tmp); D.flip = !D.flip; }
  }
  for (int q = tl; q < (int)M.tail.size(); q++, m++) {
    const int sweeps = (m == nl - 1) ? T.nbot : T.nu1 + T.nu2;
    if (sweeps & 1) std::swap(M.tail[q].phi, M.tail[q].tmp);
  }
  return true;
}
static void nd_vcycle_d(NDMG &M, int l) {
  const vdn_params &P = ctx().prm;
  NDLev &DL = M.dlev[l];                               // phi = 0 on entry: written by kk_nd_restrict
  if (nd_small_end(M, l, 0)) return;
  const bool last = (l == (int)M.dlev.size() - 1);
  if (last && M.tail.empty()) { nd_jacobi_d(DL, nd_bottom_sweeps_global(DL)); return; }
  nd_jacobi_d(DL, P.hg_nu1, true);
  nd_residual_d(M, DL, false);
  nd_restrict_down(M, l);
  if (last) nd_vcycle_t(M, 0); else nd_vcycle_d(M, l + 1);
  nd_prolong_up(M, l);
  nd_jacobi_d(DL, P.hg_nu2);
}
// ---- nested iteration for the initial guess (round 3; vdn_params.hg_fmg; the algorithm is stated with vo_nd_solve in oracle/vo_hgproject.c) ----
// Only for a solve that starts from phi = 0.  Levels are counted globally: the distributed ones, then the replicated tail.  The right-hand side
// travels down through the residual arrays (res := b, halo, the cycle's own restriction -- which also zeroes the coarse phi), the coarsest level
// with more than 9^3 nodes gets two V-cycles from zero, every level above it the interpolated solution of the level below and, except the
// finest, one V-cycle.  Launched eagerly, once per solve (~0.2 of a fine-level cycle); the V-cycles that follow are the replayed graphs.
static long nd_level_nodes(const NDMG &M, int g) {
  const int nd = (int)M.dlev.size();
  if (g < nd) return (long)(M.dlev[g].ng[0] + 1) * (M.dlev[g].ng[1] + 1) * (M.dlev[g].ng[2] + 1);
  const NLev &T = M.tail[g - nd];
  return (long)(T.n[0] + 1) * (T.n[1] + 1) * (T.n[2] + 1);
}
// one V-cycle on the phi a level holds (not the error equation: phi is NOT zeroed); level g > 0 with a level below it
static void nd_cycle_at(NDMG &M, int g) {
  const vdn_params &P = ctx().prm;
  const int nd = (int)M.dlev.size();
  if (g < nd) {
    nd_jacobi_d(M.dlev[g], P.hg_nu1, true);
    nd_residual_d(M, M.dlev[g], false);
    nd_restrict_down(M, g);
    if (g + 1 < nd) nd_vcycle_d(M, g + 1); else nd_vcycle_t(M, 0);
    nd_prolong_up(M, g);
    nd_jacobi_d(M.dlev[g], P.hg_nu2);
    return;
  }
  const int t = g - nd;
  nd_jacobi_t(M.tail[t], P.hg_nu1, true);
  NLev &L = M.tail[t]; NLev &C = M.tail[t + 1];
  nd_fill_nodes(L, L.phi);
  nd_launch_march<1>(L, L.phi, L.res, nullptr);
  nd_fill_nodes(L, L.res);
  hipLaunchKernelGGL(kk_nd_restrict, ng3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1), NBLK, 0, ctx().stream, L, C);
  nd_vcycle_t(M, t + 1);
  nd_fill_nodes(M.tail[t + 1], M.tail[t + 1].phi);
  nd_launch_prolong(M.tail[t], M.tail[t + 1], 0, 0, 0);
  nd_jacobi_t(M.tail[t], P.hg_nu2);
}
static void nd_fmg(NDMG &M) {
  const int nd = (int)M.dlev.size(), ntot = nd + (int)M.tail.size();
  int ls = -1;
  for (int g = 1; g < ntot; g++) if (nd_level_nodes(M, g) > 729) ls = g;
  if (ls < 1 || ls + 1 >= ntot) return;                 // (a starting level with nothing below it: no nested iteration)
  hipStream_t st = ctx().stream;
  for (int g = 0; g < ls; g++) {                         // b_{g+1} = R b_g
    if (g < nd) {
      NDLev &DL = M.dlev[g];
      for (NBox &B : DL.boxes) hipLaunchKernelGGL(kk_nd_copy_b_res, ng3(B.L.n[0] + 1, B.L.n[1] + 1, B.L.n[2] + 1), NBLK, 0, st, B.L);
      DL.res_restricted = false;                                          // (res holds b itself: the plain restriction)
      if (DL.halo_res) xplan_run(DL.halo_res);
      nd_restrict_down(M, g);
      if (g + 1 == nd) HIPCHK(hipMemsetAsync(M.tail[0].phi, 0, sizeof(double) * M.tail[0].sz, st));      // (the gather fills b only)
    } else {
      NLev &L = M.tail[g - nd]; NLev &C = M.tail[g - nd + 1];
      hipLaunchKernelGGL(kk_nd_copy_b_res, ng3(L.n[0] + 1, L.n[1] + 1, L.n[2] + 1), NBLK, 0, st, L);
      nd_fill_nodes(L, L.res);
      hipLaunchKernelGGL(kk_nd_restrict, ng3(C.n[0] + 1, C.n[1] + 1, C.n[2] + 1), NBLK, 0, st, L, C);
    }
  }
  if (ls < nd) nd_vcycle_d(M, ls); else nd_vcycle_t(M, ls - nd);      // from zero
  for (int g = ls; g >= 0; g--) {
    if (g < ls) {                                        // the interpolated solution of the level below (phi_g is zero: the restriction left it so)
      if (g < nd) nd_prolong_up(M, g);
      else { nd_fill_nodes(M.tail[g - nd + 1], M.tail[g - nd + 1].phi); nd_launch_prolong(M.tail[g - nd], M.tail[g - nd + 1], 0, 0, 0); }
    }
    if (g > 0) nd_cycle_at(M, g);
  }
}

// ---- one cycle as a hipGraph (see mg_cc.hip) ----------------------------------------------------------------------------------------
// The Jacobi sweeps ping-pong phi / tmp on the host side, so a cycle changes the host state: the cache keeps, next to the graph, the
// state the body left behind (a function of the state it started from, which the key hashes), and a replay installs it.
static void nd_key_lev(GraphKey &k, const NLev &L) {
  k.put(L.n); k.put(L.PX); k.put(L.PY); k.put(L.sz); k.put(L.f); k.put(L.phi); k.put(L.tmp); k.put(L.b); k.put(L.res); k.put(L.sig);
  k.put(L.dirlo); k.put(L.dirhi); k.put(L.per);
}
static unsigned long long nd_graph_key(const NDMG &M, int what) {
  const vdn_params &P = ctx().prm;
  GraphKey k; k.put(what); k.put(P.hg_nu1); k.put(P.hg_nu2); k.put(P.hg_nub); k.put(P.hg_omega); k.put(P.hg_omega_pre1); k.put(P.hg_omega_pre2); k.put(M.per); k.put(M.d_nrm);
  k.put(M.sendbuf); k.put(M.recvbuf); k.put(M.d_gb); k.put(M.cnt_nodes); k.put(M.cnt_cells);
  for (const NDLev &DL : M.dlev) {
    k.put(xplan_serial(DL.halo_A)); k.put(xplan_serial(DL.halo_B)); k.put(xplan_serial(DL.halo_res)); k.put(xplan_serial(DL.halo_sig)); k.put(DL.ng); k.put(DL.flip); k.put(DL.single_box); k.put(DL.per); k.put(DL.res_restricted);
    for (const NBox &B : DL.boxes) { nd_key_lev(k, B.L); k.put(B.lo); k.put(B.A); k.put(B.B); k.put(B.hmask); k.put(xplan_serial(B.hA)); k.put(xplan_serial(B.hB)); }
  }
  for (const NLev &L : M.tail) nd_key_lev(k, L);
  for (long o : M.loc_off_nodes) k.put(o);
  return k.h;
}
static std::map<unsigned long long, NDMG> g_nd_post;
static unsigned long g_nd_post_gen = 0;
template <class Body> static void nd_run_cycle(NDMG &M, int what, Body body) {
  if (!graphs_enabled()) { body(); return; }
  if (g_nd_post_gen != graph_generation()) { g_nd_post.clear(); g_nd_post_gen = graph_generation(); }     // the graphs went: so does the state kept next to them
  const unsigned long long key = nd_graph_key(M, what);
  auto it = g_nd_post.find(key);
  if (it != g_nd_post.end() && graph_replay(key)) { M = it->second; return; }
  graph_begin();
  try { body(); } catch (...) { graph_abort(); throw; }
  graph_end(key);
  if (g_nd_post_gen != graph_generation()) { g_nd_post.clear(); g_nd_post_gen = graph_generation(); }     // graph_end cleared the cache to make room
  if (g_nd_post.size() >= 256) g_nd_post.clear();
  g_nd_post[key] = M;
}

static double nd_read(double *d) {
  return read_scalar1(d);
}

// keep: the level hierarchy (arrays in the caller's arena scope, sigma on every level) survives the call and the next call with the same
// `keep` only loads its right-hand side and phi -- the composite solves run one V-cycle of this solver per FAC iteration on the same
// coefficients (19 iterations per step on the tagged 256^3 hierarchy: 19 set-ups of 0.6 ms each before)
struct NdKeep { bool built = false; NDMG M; };
NdKeep *nd_keep_new() { return new NdKeep; }
void nd_keep_free(NdKeep *k) { delete k; }
// fast: hgproject's single-level call (NdFast in vdn_internal.h) -- rh and phi are known to be zero and are not touched (may be null), sigma comes
// from fast->rhohalf (coeffs may be null), and instead of storing phi into a multifab the call returns views of the finest level's phi
// (ghost nodes exchanged) in fast->phi_view; the level arrays then stay allocated: the CALLER releases the arena (mark taken before the call)
int nd_solve(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *coeffs, const vdn_multifab *u, const double *dx,
             const int bc[3][2], double rel_eps, double abs_eps, int max_iter, int *cycles, double *res0, double *res, NdKeep *keep, NdFast *fast, bool fmg_start, bool rh_is_b, vdn_multifab *add_to) {
  Prof prof_("hg_multigrid");
  if (ctx().prm.dm == 2) return nd2_solve(rh, phi, coeffs, u, dx, bc, rel_eps, abs_eps, max_iter, cycles, res0, res);
  const vdn_params &P = ctx().prm;
  g_nd_iso = nd_isotropic(dx);
  if (fast) { REQUIRE(fast->rhohalf && u && !keep, "nodal multigrid: the fast path needs rhohalf and u"); /* rhohalf is read on valid cells only: no ghost layer needed */ coeffs = fast->rhohalf; }
  else REQUIRE(rh->ng >= 1 && phi->ng >= 1 && coeffs->ng >= 1, "nodal multigrid: rh, phi, coeffs need one ghost layer");
  hipStream_t st = ctx().stream;
  size_t mark = arena_mark();
  NDMG M_local;
  NDMG &M = keep ? keep->M : M_local;
  const bool rebuild = !(keep && keep->built);
  if (rebuild) nd_build(M, coeffs, dx, bc);          // (of `coeffs` only the layout, the level and the boxes are used)
  NDLev &D0 = M.dlev[0];
  if (rebuild) {
  // sigma: level 0 from the (ghost-filled) coeffs multifab; coarser distributed levels by averaging + halo exchange
  for (size_t b = 0; b < D0.boxes.size(); b++) {
    NLev &L0 = D0.boxes[b].L; const vdn_box &bx = coeffs->vbox[b];
    if (fast) {
      hipLaunchKernelGGL(kk_nd_load_sigma_rho, ng3(L0.n[0] + 2, L0.n[1] + 2, L0.n[2] + 2), NBLK, 0, st, L0, fast->rhohalf->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
    } else
    hipLaunchKernelGGL(kk_nd_load_sigma, ng3(L0.n[0] + 2, L0.n[1] + 2, L0.n[2] + 2), NBLK, 0, st, L0, coeffs->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
  }
  if (fast && D0.halo_sig) xplan_run(D0.halo_sig);          // neighbours' and periodic images of sigma (what fill_boundary(coeffs) did)
  for (size_t l = 1; l < M.dlev.size(); l++) {
    for (size_t b = 0; b < M.dlev[l].boxes.size(); b++) {
      NLev &C = M.dlev[l].boxes[b].L;
      hipLaunchKernelGGL(kk_nd_coarsen_sigma, ng3(C.n[0], C.n[1], C.n[2]), NBLK, 0, st, M.dlev[l - 1].boxes[b].L, C);
    }
    if (M.dlev[l].halo_sig) xplan_run(M.dlev[l].halo_sig);        // interior / periodic ghost cells; domain ghosts stay 0
  }
  if (!M.tail.empty()) {
    NDLev &DL = M.dlev.back();
    const int nb = (int)M.gb.size() / 2;
    for (size_t b = 0; b < DL.boxes.size(); b++) {
      const NLev &F = DL.boxes[b].L;
      hipLaunchKernelGGL(kk_nd_coarsen_sigma_pack, ng3(F.n[0] / 2, F.n[1] / 2, F.n[2] / 2), NBLK, 0, st, F, M.sendbuf, M.loc_off_cells[b], F.n[0] / 2, F.n[1] / 2, F.n[2] / 2);
    }
    comm_allgather_dev(M.sendbuf, M.recvbuf, M.cnt_cells);
    hipLaunchKernelGGL(kk_nd_unpack, dim3(4, 1, (unsigned)nb), dim3(256), 0, st, M.tail[0], M.tail[0].sig, M.recvbuf, M.d_gb + nb, 0);
    hipLaunchKernelGGL(kk_nd_fill_cells, ng3(M.tail[0].n[0] + 2, M.tail[0].n[1] + 2, M.tail[0].n[2] + 2), NBLK, 0, st, M.tail[0], M.tail[0].sig);
    for (size_t l = 1; l < M.tail.size(); l++) {
      NLev &C = M.tail[l];
      hipLaunchKernelGGL(kk_nd_coarsen_sigma, ng3(C.n[0], C.n[1], C.n[2]), NBLK, 0, st, M.tail[l - 1], C);
      hipLaunchKernelGGL(kk_nd_fill_cells, ng3(C.n[0] + 2, C.n[1] + 2, C.n[2] + 2), NBLK, 0, st, C, C.sig);
    }
  }
  }
  if (keep) keep->built = true;
  if (u && !fast) {                                         // add_divu = .true., hg_multigrid.f90:96
    REQUIRE(u->ng >= 1 && u->nc >= 3, "nodal multigrid: u needs a ghost cell");
    std::vector<std::pair<nd_divu_K, Range3>> v;
    for (size_t b = 0; b < D0.boxes.size(); b++) {
      const vdn_box &bx = coeffs->vbox[b];
      Range3 r; for (int d = 0; d < 3; d++) { r.lo[d] = bx.lo[d]; r.hi[d] = bx.hi[d] + 1; }
      v.push_back({ nd_divu_K{ u->fabs[b], rh->fabs[b], 0.25 / dx[0], 0.25 / dx[1], 0.25 / dx[2] }, r });
    }
    launch_cells(v, st);
  }
  HIPCHK(hipMemsetAsync(M.d_nrm, 0, 2 * sizeof(double), st));
  for (size_t b = 0; b < D0.boxes.size(); b++) {
    NLev &L0 = D0.boxes[b].L; const vdn_box &bx = coeffs->vbox[b];
    if (fast) {
      REQUIRE(u->ng >= 1 && u->nc >= 3, "nodal multigrid: u needs a ghost cell");
      hipLaunchKernelGGL(kk_nd_load_divu, ng3(L0.n[0] + 1, L0.n[1] + 1, std::min(L0.n[2] + 1, 16)), NBLK, 0, st, L0, u->fabs[b], 0.25 / dx[0], 0.25 / dx[1], 0.25 / dx[2],
                         bx.lo[0], bx.lo[1], bx.lo[2], M.d_nrm);
    } else if (rh_is_b) {
      REQUIRE(max_iter < 0 && !u && ctx().prm.dm == 3, "nodal multigrid: rh_is_b is the fixed-cycle correction solve's");
      hipLaunchKernelGGL(kk_nd_load_b, ng3(L0.n[0] + 1, L0.n[1] + 1, std::min(L0.n[2] + 1, 16)), NBLK, 0, st, L0, rh->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
    } else
    hipLaunchKernelGGL(kk_nd_load, ng3(L0.n[0] + 1, L0.n[1] + 1, std::min(L0.n[2] + 1, 16)), NBLK, 0, st, L0, rh->fabs[b], phi->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2], M.d_nrm);
  }
  comm_allreduce_max_dev(M.d_nrm, 2);
  const bool single = (M.dlev.size() == 1 && M.tail.empty());
  const bool fixed_cycles = max_iter < 0;     // exactly -max_iter V-cycles, no norms, no convergence test (composite coarse correction)
  double bnorm = 1.0, p0max = 1.0;
  if (!fixed_cycles) { const double *sc = read_scalars(M.d_nrm, 2); bnorm = sc[0]; p0max = sc[1]; }
  int cyc = 0; bool conv = (bnorm == 0.0); double rn = 0.0;
  if (P.hg_fmg && !conv && !single && (fixed_cycles ? fmg_start : (p0max == 0.0 && bnorm < HUGE_VAL))) nd_fmg(M);
  for (int c = 0; fixed_cycles && c < -max_iter; c++) {
    if (single) { nd_jacobi_d(M.dlev[0], nd_bottom_sweeps_global(M.dlev[0])); continue; }
    nd_run_cycle(M, 2, [&] {
      NDLev &D = M.dlev[0];
      nd_jacobi_d(D, P.hg_nu1, true);
      nd_residual_d(M, D, false);
      nd_restrict_down(M, 0);
      if (M.dlev.size() > 1) nd_vcycle_d(M, 1); else nd_vcycle_t(M, 0);
      nd_prolong_up(M, 0);
      nd_jacobi_d(D, P.hg_nu2);
    });
    cyc++;
  }
  if (fixed_cycles) conv = true;
  // pre-smoothing + residual, then per cycle [coarse correction, post-smoothing, next pre-smoothing, residual + norm] as one replayed
  // graph and one read-back: the same launch sequence as testing the residual the cycle computes after its pre-smoothing
  // vdn_params.mg_predict (hgproject's call, zero guess): the previous solve of this size stopped after `pred` cycles, so the norms of the cycles
  // before pred - 1 are not waited for -- they go into the device-side history and are read in one go after cycle pred - 1.  Should the history show
  // that an earlier cycle had already met the tolerance, this solve is thrown away and repeated with a read-back per cycle (rare: the count
  // dropped by two or more from one solve to the next), so the result is the one the plain loop gives, whatever the prediction was.
  int gn[3] = { M.dlev[0].ng[0], M.dlev[0].ng[1], M.dlev[0].ng[2] };
  int pred = (fast && !fixed_cycles && !single && !conv && p0max == 0.0) ? std::min(mg_predict_get(1, gn), std::min(max_iter, 63)) : 0;
  if (!conv) {
    nd_jacobi_d(M.dlev[0], single ? nd_bottom_sweeps_global(M.dlev[0]) : P.hg_nu1, !single);
    if (pred >= 2) {
      nd_residual_d(M, M.dlev[0], true, false);
      norm_hist_reset(); norm_hist_push(M.d_nrm);
      nd_run_cycle(M, 4 + 4 * pred, [&] {                  // all blind cycles as ONE graph (ids 1, 2: the plain cycles; nothing else is a multiple of 4)
        for (int c = 1; c <= pred - 1; c++) {
          NDLev &D = M.dlev[0];
          nd_restrict_down(M, 0);
          if (M.dlev.size() > 1) nd_vcycle_d(M, 1); else nd_vcycle_t(M, 0);
          nd_prolong_up(M, 0);
          nd_jacobi_d(D, P.hg_nu2);
          nd_jacobi_d(D, P.hg_nu1, true);
          nd_residual_d(M, D, true, false);
          norm_hist_push(M.d_nrm);
        }
      });
      const double *h = norm_hist_read(pred);
      int first = -1;                                            // the first cycle count at which the plain loop would have stopped
      for (int c = 0; c < pred && first < 0; c++)
        if (((h[c] <= rel_eps * bnorm && bnorm < HUGE_VAL) || h[c] <= abs_eps) || !(h[c] < HUGE_VAL)) first = c;
      if (first >= 0 && first < pred - 1) {                      // overshot: repeat without the prediction
        arena_release(mark);
        struct Off { Off() { g_mg_predict_off++; } ~Off() { g_mg_predict_off--; } } off_;
        return nd_solve(rh, phi, nullptr, u, dx, bc, rel_eps, abs_eps, max_iter, cycles, res0, res, keep, fast, fmg_start, rh_is_b, add_to);
      }
      cyc = pred - 1; rn = h[pred - 1];
    } else {
      nd_residual_d(M, M.dlev[0], true); rn = nd_read(M.d_nrm);
    }
  }
  while (!conv) {
    if ((rn <= rel_eps * bnorm && bnorm < HUGE_VAL) || rn <= abs_eps) { conv = true; break; }
    if (cyc >= max_iter || !(rn < HUGE_VAL) || !(bnorm < HUGE_VAL)) break;     // also: a NaN / inf norm (the reductions turn NaN into +inf)
    if (single) { nd_jacobi_d(M.dlev[0], nd_bottom_sweeps_global(M.dlev[0])); nd_residual_d(M, M.dlev[0], true); }
    else nd_run_cycle(M, 1, [&] {
      NDLev &D = M.dlev[0];
      nd_restrict_down(M, 0);
      if (M.dlev.size() > 1) nd_vcycle_d(M, 1); else nd_vcycle_t(M, 0);
      nd_prolong_up(M, 0);
      nd_jacobi_d(D, P.hg_nu2);
      nd_jacobi_d(D, P.hg_nu1, true);
      nd_residual_d(M, D, true);
    });
    cyc++;
    rn = nd_read(M.d_nrm);
  }
  NDLev &DF = M.dlev[0];                    // (a replayed cycle re-assigns M: take the reference afresh)
  nd_halo_phi(DF);
  if (fast) fast->phi_view.clear();
  for (size_t b = 0; b < DF.boxes.size(); b++) {
    NLev &L0 = DF.boxes[b].L; const vdn_box &bx = coeffs->vbox[b];
    if (fast) { fast->phi_view.push_back(nd_view(L0, L0.phi, bx.lo, 3)); continue; }
    if (add_to) hipLaunchKernelGGL(kk_nd_store_add, ng3(L0.n[0] + 3, L0.n[1] + 3, L0.n[2] + 3), NBLK, 0, st, L0, phi->fabs[b], add_to->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
    else
    hipLaunchKernelGGL(kk_nd_store, ng3(L0.n[0] + 3, L0.n[1] + 3, L0.n[2] + 3), NBLK, 0, st, L0, phi->fabs[b], bx.lo[0], bx.lo[1], bx.lo[2]);
  }
  if (cycles) *cycles = cyc; if (res0) *res0 = bnorm; if (res) *res = rn;
  if (conv && fast && !fixed_cycles && !single && cyc >= 1) mg_predict_set(1, gn, cyc);
  if (!keep && !fast) arena_release(mark);  // with `keep` / `fast` the hierarchy stays in the caller's arena scope
  return conv ? 0 : 1;
}

// ====================================================================================================
// hgproject pieces (hgproject.f90)
// ====================================================================================================
struct UvecArgs { int lo[3], hi[3], ng; int phys[3][2]; double dt, dtinv; int proj_type; };
// gp ghost zeroing at INLET, the projected quantity on the grown box, wall ghost planes zeroed
// (hgproject.f90:453-511), fused in one pass over the ghosted fab
struct create_uvec_K { FV unew; FV uold; FV rhohalf; FV gp; UvecArgs A;
  __device__ void cell(int i, int j, int k) const {
    const int q[3] = { i, j, k };
    bool g1 = true;                         // inside the box grown by 1
  #pragma unroll
    for (int d = 0; d < 3; d++) if (q[d] < A.lo[d] - 1 || q[d] > A.hi[d] + 1) g1 = false;
    bool wall_plane = false, inlet_plane = false;
  #pragma unroll
    for (int d = 0; d < 3; d++) {
      if (q[d] == A.lo[d] - 1) { int p = A.phys[d][0]; if (p == VDN_SLIP_WALL || p == VDN_NO_SLIP_WALL) wall_plane = true; if (p == VDN_INLET) inlet_plane = true; }
      if (q[d] == A.hi[d] + 1) { int p = A.phys[d][1]; if (p == VDN_SLIP_WALL || p == VDN_NO_SLIP_WALL) wall_plane = true; if (p == VDN_INLET) inlet_plane = true; }
    }
  #pragma unroll
    for (int m = 0; m < 3; m++) {
      double gpv = 0.0;
      if (g1) { gpv = fv_get(gp, i, j, k, m); if (inlet_plane) { gpv = 0.0; fv_at(gp, i, j, k, m) = 0.0; } }
      if (wall_plane) { fv_at(unew, i, j, k, m) = 0.0; continue; }
      if (!g1) continue;
      double v = fv_get(unew, i, j, k, m);
      if (A.proj_type == VDN_PRESSURE_ITERS) v = (v - fv_get(uold, i, j, k, m)) * A.dtinv;
      else if (A.proj_type == VDN_REGULAR_TIMESTEP) v = v + A.dt * gpv / fv_get(rhohalf, i, j, k, 0);
      fv_at(unew, i, j, k, m) = v;
    }
  } };

struct coeffs_K { FV coeffs; FV rhohalf;
  __device__ void cell(int i, int j, int k) const {
    fv_at(coeffs, i, j, k) = 1.0 / fv_get(rhohalf, i, j, k, 0);      // hg_multigrid.f90:76-77
  } };

// cell-centred gradient of the nodal phi (mkgphi, hgproject.f90:538-580), component m at cell (i,j,k)
DEVI double gphi_of(const FV &phi, int i, int j, int k, int m, double dxi) {
  #define P(a, b, c) fv_get(phi, i + (a), j + (b), k + (c))
  if (m == 0) return 0.25 * (P(1, 0, 0) + P(1, 1, 0) + P(1, 0, 1) + P(1, 1, 1) - P(0, 0, 0) - P(0, 1, 0) - P(0, 0, 1) - P(0, 1, 1)) * dxi;
  if (m == 1) return 0.25 * (P(0, 1, 0) + P(1, 1, 0) + P(0, 1, 1) + P(1, 1, 1) - P(0, 0, 0) - P(1, 0, 0) - P(0, 0, 1) - P(1, 0, 1)) * dxi;
  return 0.25 * (P(0, 0, 1) + P(1, 0, 1) + P(0, 1, 1) + P(1, 1, 1) - P(0, 0, 0) - P(1, 0, 0) - P(0, 1, 0) - P(1, 1, 0)) * dxi;
  #undef P
}

struct HgUpdArgs { int hi[3]; double dt, dtinv, dxi[3]; int proj_type; };
// mkgphi and the update of hgproject.f90:659-676 in one pass: grad phi is used where it is computed, no gphi multifab in between
struct hg_update_K { FV unew; FV uold; FV gp; FV rhohalf; FV p; FV phi; HgUpdArgs A;
  __device__ void cell(int i, int j, int k) const {
    const bool cell = i <= A.hi[0] && j <= A.hi[1] && k <= A.hi[2];
    if (cell) {
      const double rho = fv_get(rhohalf, i, j, k, 0);
    #pragma unroll
      for (int m = 0; m < 3; m++) {
        const double gph = gphi_of(phi, i, j, k, m, A.dxi[m]);
        double v = fv_get(unew, i, j, k, m) - gph / rho;                    // hgproject.f90:659-667
        if (A.proj_type == VDN_PRESSURE_ITERS) v = fv_get(uold, i, j, k, m) + A.dt * v;
        fv_at(unew, i, j, k, m) = v;
        if (A.proj_type == VDN_PRESSURE_ITERS) fv_at(gp, i, j, k, m) = fv_get(gp, i, j, k, m) + gph;
        else if (A.proj_type == VDN_REGULAR_TIMESTEP) fv_at(gp, i, j, k, m) = A.dtinv * gph;
      }
    }
    if (A.proj_type == VDN_PRESSURE_ITERS) fv_at(p, i, j, k) = fv_get(p, i, j, k) + fv_get(phi, i, j, k);
    else if (A.proj_type == VDN_REGULAR_TIMESTEP) fv_at(p, i, j, k) = A.dtinv * fv_get(phi, i, j, k);
  } };


// per-level pieces of hgproject, shared by the single-level driver and the two-level one
// coeffs == nullptr: the caller takes sigma straight from rhohalf (nd_solve's fast path)
static void hg_level_pre(int proj_type, vdn_multifab *un, const vdn_multifab *uo, const vdn_multifab *rhh, vdn_multifab *gpp, vdn_multifab *coeffs,
                         double dt, const vdn_bc_tower *bct) {
  hipStream_t st = ctx().stream;
  REQUIRE(un->ng >= 1 && gpp->ng >= 1 && rhh->ng >= 1, "hgproject: ghost widths");
  std::vector<std::pair<create_uvec_K, Range3>> vu; std::vector<std::pair<coeffs_K, Range3>> vc;
  for (int i = 0; i < un->nfabs(); i++) {
    UvecArgs A; Range3 r; BoxP bp = make_boxp(un, i, bct);
    for (int d = 0; d < 3; d++) { A.lo[d] = bp.lo[d]; A.hi[d] = bp.hi[d]; r.lo[d] = bp.lo[d] - un->ng; r.hi[d] = bp.hi[d] + un->ng;
      for (int s = 0; s < 2; s++) A.phys[d][s] = bp.phys[d][s]; }
    A.ng = un->ng; A.dt = dt; A.dtinv = 1.0 / dt; A.proj_type = proj_type;
    vu.push_back({ create_uvec_K{ un->fabs[i], uo->fabs[i], rhh->fabs[i], gpp->fabs[i], A }, r });
    Range3 rv; for (int d = 0; d < 3; d++) { rv.lo[d] = bp.lo[d]; rv.hi[d] = bp.hi[d]; }
    if (coeffs) vc.push_back({ coeffs_K{ coeffs->fabs[i], rhh->fabs[i] }, rv });
  }
  launch_cells(vu, st); launch_cells(vc, st);
  mf_fill_boundary(un);                                               // hgproject.f90:232
  if (coeffs) mf_fill_boundary(coeffs);                               // hg_multigrid.f90:79
}
// phi_view: phi of box i as a view of the solver's level array (the fast path) instead of phi->fabs[i]
static void hg_level_post(int proj_type, vdn_multifab *un, const vdn_multifab *uo, const vdn_multifab *rhh, vdn_multifab *gpp, vdn_multifab *pp,
                          vdn_multifab *gphi, const vdn_multifab *phi, const double *dx, double dt, const std::vector<FV> *phi_view = nullptr) {
  hipStream_t st = ctx().stream;
  if (proj_type == VDN_INITIAL_PROJECTION || proj_type == VDN_DIVU_ITERS) { mf_setval(gpp, 0.0, 0, gpp->nc, true); mf_setval(pp, 0.0, 0, 1, true); }   // 673-676
  std::vector<std::pair<hg_update_K, Range3>> vh;
  for (int i = 0; i < un->nfabs(); i++) {
    Range3 rv, rn; HgUpdArgs H;
    for (int d = 0; d < 3; d++) { rv.lo[d] = rn.lo[d] = un->vbox[i].lo[d]; rv.hi[d] = un->vbox[i].hi[d]; rn.hi[d] = rv.hi[d] + 1; H.hi[d] = rv.hi[d]; }
    H.dt = dt; H.dtinv = 1.0 / dt; H.proj_type = proj_type;
    for (int d = 0; d < 3; d++) H.dxi[d] = 1.0 / dx[d];
    vh.push_back({ hg_update_K{ un->fabs[i], uo->fabs[i], gpp->fabs[i], rhh->fabs[i], pp->fabs[i], phi_view ? (*phi_view)[i] : phi->fabs[i], H }, rn });
  }
  (void)gphi;
  launch_cells(vh, st);
}
static void do_ml_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold, vdn_multifab **rhohalf,
                            vdn_multifab **p, vdn_multifab **gp, const double *dx, double dt, const vdn_bc_tower *bct, int press_comp0);

void do_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold, vdn_multifab **rhohalf,
                  vdn_multifab **p, vdn_multifab **gp, const double *dx, double dt, const vdn_bc_tower *bct, int press_comp0) {
  if (ctx().prm.dm == 2) { do2_hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, bct, press_comp0); return; }
  REQUIRE(proj_type >= VDN_INITIAL_PROJECTION && proj_type <= VDN_REGULAR_TIMESTEP, "No proj_type by this number");
  if (mla->nlev > 1) { do_ml_hgproject(proj_type, mla, unew, uold, rhohalf, p, gp, dx, dt, bct, press_comp0); return; }
  const int n = 0;
  size_t mark = arena_mark();
  vdn_multifab *un = unew[n], *uo = uold[n], *rhh = rhohalf[n], *gpp = gp[n], *pp = p[n];
  REQUIRE(pp->ng >= 1, "hgproject: ghost widths");
  double rel = ctx().prm.hg_rel_eps > 0.0 ? ctx().prm.hg_rel_eps : 1.e-12;   // hgproject.f90:113-119 (nlevs = 1)
  double abs_eps = -1.0;
  if (proj_type == VDN_INITIAL_PROJECTION && ctx().prm.prob_type == 4) abs_eps = 1.e-12;   // 125-127
  int ebc[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc[d][s] = bct->ell_bc(n, 0, d, s, press_comp0);
  int cyc; double r0, rr;
  // Round 3: rh, phi and coeffs (hgproject.f90:70-76, hg_multigrid.f90:68-80) exist only to carry zeros, D u and 1 / rhohalf into the solver
  // and phi out of it: the solver takes sigma from rhohalf, forms b = -D u while it loads, and hg_update reads phi from the level array
  // (0.5 ms of fills, copies and passes per 256^3 projection; VDN_HG_FAST=0: the multifabs as the reference has them -- same values)
  static const bool fast_on = !(vdn_env("VDN_HG_FAST") && atoi(vdn_env("VDN_HG_FAST")) == 0);
  if (fast_on) {
    hg_level_pre(proj_type, un, uo, rhh, gpp, nullptr, dt, bct);
    NdFast F; F.rhohalf = rhh;
    int rc = nd_solve(nullptr, nullptr, nullptr, un, dx, ebc, rel, abs_eps, ctx().prm.hg_max_iter, &cyc, &r0, &rr, nullptr, &F);
    ctx().solver_cycles[1] = cyc; ctx().solver_res0[1] = r0; ctx().solver_res[1] = rr;
    solver_check(rc, "nodal multigrid", cyc, rr, r0);
    hg_level_post(proj_type, un, uo, rhh, gpp, pp, nullptr, nullptr, dx, dt, &F.phi_view);
    mf_fill_boundary(gpp); mf_fill_boundary(pp);                      // hgproject.f90:359-362
    arena_release(mark);
    return;
  }
  vdn_multifab *rh = mf_temp(mla, n, 1, 1, 3, true, 0.0);
  vdn_multifab *phi = mf_temp(mla, n, 1, 1, 3, true, 0.0);
  vdn_multifab *gphi = mf_temp(mla, n, 3, 0, -1, false, 0.0);
  vdn_multifab *coeffs = mf_temp(mla, n, 1, 1, -1, true, 0.0);        // ghosts 0: hg_multigrid.f90:73
  hg_level_pre(proj_type, un, uo, rhh, gpp, coeffs, dt, bct);
  int rc = nd_solve(rh, phi, coeffs, un, dx, ebc, rel, abs_eps, ctx().prm.hg_max_iter, &cyc, &r0, &rr);
  ctx().solver_cycles[1] = cyc; ctx().solver_res0[1] = r0; ctx().solver_res[1] = rr;
  solver_check(rc, "nodal multigrid", cyc, rr, r0);
  hg_level_post(proj_type, un, uo, rhh, gpp, pp, gphi, phi, dx, dt);
  mf_fill_boundary(gpp); mf_fill_boundary(pp);                        // hgproject.f90:359-362
  mf_temp_free(coeffs); mf_temp_free(gphi); mf_temp_free(phi); mf_temp_free(rh);
  arena_release(mark);
}

// =====================================================================================================================
// composite nodal solve on two levels (the algorithm stated with vo_ml_nd_solve in the CPU restatement) and the multilevel hgproject
// =====================================================================================================================
// Works directly on the multifab fabs (nodal, one ghost layer).  This round: one fine box (the coarse level may be any
// decomposition the single-level multigrid accepts), single rank.
DEVI void ndf_apply(const FV &phi, const FV &sig, const double f[3], int i, int j, int k, double &Kp, double &diag) {
  const NdW W = nd_weights(f);
  double p[3][3][3], sg[2][2][2];
  #pragma unroll
  for (int c = 0; c < 3; c++)
    #pragma unroll
    for (int b = 0; b < 3; b++)
      #pragma unroll
      for (int a = 0; a < 3; a++) p[c][b][a] = fv_get(phi, i + a - 1, j + b - 1, k + c - 1);
  #pragma unroll
  for (int c = 0; c < 2; c++)
    #pragma unroll
    for (int b = 0; b < 2; b++)
      #pragma unroll
      for (int a = 0; a < 2; a++) sg[c][b][a] = fv_get(sig, i + a - 1, j + b - 1, k + c - 1);
  nd_stencil(W, p, sg, Kp, diag);
}
struct NdfArgs { double f[3]; int lo[3], hi[3]; int dirlo[3], dirhi[3]; int cflo[3], cfhi[3]; int ilo[3], ihi[3]; };
DEVI bool ndf_pdir(const NdfArgs &A, int i, int j, int k) {
  return (i == A.lo[0] && A.dirlo[0]) || (i == A.hi[0] && A.dirhi[0]) || (j == A.lo[1] && A.dirlo[1]) || (j == A.hi[1] && A.dirhi[1]) ||
         (k == A.lo[2] && A.dirlo[2]) || (k == A.hi[2] && A.dirhi[2]);
}
DEVI bool ndf_cf(const NdfArgs &A, int i, int j, int k) {          // node on a coarse-fine face of the (single) fine box
  return (i == A.lo[0] && A.cflo[0]) || (i == A.hi[0] && A.cfhi[0]) || (j == A.lo[1] && A.cflo[1]) || (j == A.hi[1] && A.cfhi[1]) ||
         (k == A.lo[2] && A.cflo[2]) || (k == A.hi[2] && A.cfhi[2]);
}
// A.lo/hi: node range of the box.  excl: 0 none, 1 exclude interface nodes from the norm (fine), 2 exclude coarse nodes strictly
// inside the fine box (A.ilo/ihi, coarse node indices)
__global__ void kk_ndf_residual(FV b, FV phi, FV sig, FV res, NdfArgs A, int excl, Range3 r, double *nrm) {
  REDUCE_IJ(r)
  double rmax = 0.0;
  if (in_ij) REDUCE_KLOOP(r) {
    double rr = 0.0;
    if (!ndf_pdir(A, i, j, k)) { double Kp, diag; ndf_apply(phi, sig, A.f, i, j, k, Kp, diag); rr = fv_get(b, i, j, k) - Kp; }
    fv_at(res, i, j, k) = rr;
    bool skip = false;
    if (excl == 1) skip = ndf_cf(A, i, j, k) && !ndf_pdir(A, i, j, k);
    if (excl == 2) skip = i > A.ilo[0] && i < A.ihi[0] && j > A.ilo[1] && j < A.ihi[1] && k > A.ilo[2] && k < A.ihi[2];
    if (!skip) rmax = nmax(rmax, fabs(rr));
  }
  if (nrm) block_atomic_max(nrm, rmax);
}
__global__ void kk_ndf_jacobi(FV ein, FV eout, FV rb, FV sig, NdfArgs A, double omega, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  const double p0 = fv_get(ein, i, j, k);
  double v = p0;
  if (!ndf_pdir(A, i, j, k) && !ndf_cf(A, i, j, k)) { double Kp, diag; ndf_apply(ein, sig, A.f, i, j, k, Kp, diag); if (diag != 0.0) v = p0 + omega * ((fv_get(rb, i, j, k) - Kp) / diag); }
  fv_at(eout, i, j, k) = v;
}

// k-marching forms of kk_ndf_jacobi / kk_ndf_residual (same structure as kk_nd_march: own-column loads, i-1 / i+1 columns by wave
// shuffles, three phi planes and two sigma planes in registers).  r: the node range of the box; tiles of 62 nodes along i.
// MODE 0: eout = ein + omega (rb - K ein)/diag on free nodes;  MODE 1: res = b - K phi (0 on physical Dirichlet nodes), max-norm
// over the nodes that are not interface nodes (excl = 1) / all nodes (excl = 0)
struct MarchB { FV phi, out, rb, sig, slave; int has_slave; NdfArgs A; Range3 r; int g[3], kchunk, sw; };      // sw: lanes of the segment that carries one node row (4 .. 64)
template <int MODE>
__global__ void __launch_bounds__(256) kk_ndf_march(const MarchB *args, const int *start, int nbox, double omega, int excl, double *nrm) {
  int lo_ = 0, hi_ = nbox - 1;
  const int bid = (int)blockIdx.x;
  while (lo_ < hi_) { const int mid = (lo_ + hi_ + 1) >> 1; if (as_constant(start + mid) <= bid) lo_ = mid; else hi_ = mid - 1; }
  const MarchB B = as_constant(args + lo_);                  // by value: the k loop would re-read a constant-space descriptor every plane (243 -> 357 us)
  const FV phi = B.phi, out = B.out, rb = B.rb, sig = B.sig, slave = B.slave;
  const int has_slave = B.has_slave, kchunk = B.kchunk;
  const NdfArgs A = B.A; const Range3 r = B.r;
  const int lb = bid - as_constant(start + lo_);
  const int bx = lb % B.g[0], by = (lb / B.g[0]) % B.g[1], bz = lb / (B.g[0] * B.g[1]);
  // a wave holds 64 / sw node rows of sw lanes each (first and last lane of a row segment only feed their neighbours; sw need not be a power of
  // two -- the lanes left over after the last segment own nothing): narrow boxes put several rows into a wave instead of leaving most of it idle;
  // the lane exchange never crosses a segment for an active lane
  const int sw = B.sw, rows = 64 / sw;
  const int lane = (int)threadIdx.x % sw, seg = (int)threadIdx.x / sw;
  const int i = r.lo[0] + bx * (sw - 2) + lane - 1;
  const int j = r.lo[1] + (by * (int)blockDim.y + (int)threadIdx.y) * rows + seg;
  const int k0 = r.lo[2] + bz * kchunk, k1 = min(k0 + kchunk - 1, r.hi[2]);
  const bool active = seg < rows && lane >= 1 && lane <= sw - 2 && i <= r.hi[0] && j <= r.hi[1];
  const int ic = min(i, r.hi[0] + 1), jc = min(j, r.hi[1]);
  double rmax = 0.0;
  if (k0 <= k1) {
    double p[3][3][3], sg[2][2][2];
    #pragma unroll
    for (int b = 0; b < 3; b++) { p[0][b][1] = fv_get(phi, ic, jc + b - 1, k0 - 1); p[1][b][1] = fv_get(phi, ic, jc + b - 1, k0); }
    #pragma unroll
    for (int dj = 0; dj < 2; dj++) sg[0][dj][1] = fv_get(sig, ic, jc + dj - 1, k0 - 1);
    #pragma unroll
    for (int b = 0; b < 3; b++) {
      p[0][b][0] = lane_prev(p[0][b][1]); p[0][b][2] = lane_next(p[0][b][1]);
      p[1][b][0] = lane_prev(p[1][b][1]); p[1][b][2] = lane_next(p[1][b][1]);
    }
    #pragma unroll
    for (int dj = 0; dj < 2; dj++) sg[0][dj][0] = lane_prev(sg[0][dj][1]);
    const NdW W = nd_weights(A.f);
    for (int k = k0; k <= k1; k++) {
      #pragma unroll
      for (int b = 0; b < 3; b++) p[2][b][1] = fv_get(phi, ic, jc + b - 1, k + 1);
      #pragma unroll
      for (int dj = 0; dj < 2; dj++) sg[1][dj][1] = fv_get(sig, ic, jc + dj - 1, k);
      const double rhs = fv_get(rb, ic, jc, k);
      #pragma unroll
      for (int b = 0; b < 3; b++) { p[2][b][0] = lane_prev(p[2][b][1]); p[2][b][2] = lane_next(p[2][b][1]); }
      #pragma unroll
      for (int dj = 0; dj < 2; dj++) sg[1][dj][0] = lane_prev(sg[1][dj][1]);
      double Kp, diag; nd_stencil(W, p, sg, Kp, diag);
      const double p0 = p[1][1][1];
      const bool pdir = ndf_pdir(A, i, j, k);
      // slaved to the coarser level: has_slave = 1 node mask (any union of boxes), 2 = the level is ONE box: its non-physical faces (no mask traffic)
      const bool cf = (has_slave == 2) ? (ndf_cf(A, i, j, k) && !pdir) : ((has_slave == 1 && (MODE == 0 || excl == 1)) ? (fv_get(slave, ic, jc, k) != 0.0) : false);
      if (MODE == 0) {
        double v = p0;
        if (!pdir && !cf && diag != 0.0) v = p0 + omega * ((rhs - Kp) / diag);
        if (active) fv_at(out, i, j, k) = v;
      } else {
        const double rr = pdir ? 0.0 : rhs - Kp;
        if (active) { fv_at(out, i, j, k) = rr; if (!(excl == 1 && cf && !pdir)) rmax = nmax(rmax, fabs(rr)); }
      }
      #pragma unroll
      for (int b = 0; b < 3; b++)
        #pragma unroll
        for (int a = 0; a < 3; a++) { p[0][b][a] = p[1][b][a]; p[1][b][a] = p[2][b][a]; }
      #pragma unroll
      for (int dj = 0; dj < 2; dj++)
        #pragma unroll
        for (int di = 0; di < 2; di++) sg[0][dj][di] = sg[1][dj][di];
    }
  }
  if (MODE == 1 && nrm) block_atomic_max(nrm, rmax);
}
// ---- paired form of the box-batched march (round 3) -----------------------------------------------------------------------------------
// The boxes of a hierarchy are 9 .. 41 nodes wide: with one node per lane and the first and last lane of a row segment feeding their
// neighbours, a 33-node row fills half of a 64-lane segment and a 17-node row half of a 32-lane one -- the march of a 997-box level ran at
// 2.5 x the time of its traffic.  Here a lane owns the nodes (ia, ia + 1), ia = lo + 2p, as in kk_nd_march_pair: a row of 33 nodes is 17
// lanes (+ 2 feeders) of a 32-lane segment, two rows per wave, and every access is a 16-byte pair.  A fab row has no padding: the pair of
// a lane at the end of a row is read from inside the row and shifted (fv_pair), its unused half is junk; rows are 8-byte aligned only.
// Same arithmetic per node (nd_stencil), same bits.
typedef double vdn_d2u __attribute__((ext_vector_type(2), aligned(8)));
struct PairAt { long idx; int sh; };         // linear index of the (clamped) pair start in a fab, and ia - that start (-1, 0, +1; beyond: junk)
DEVI PairAt pair_at(const FV &f, int i, int j, int k) {
  const int ic = min(max(i, f.a0), f.a0 + f.n0 - 2);
  PairAt q; q.idx = fv_idx(f, ic, j, k); q.sh = i - ic; return q;
}
DEVI void ld_pair(const double *p, long idx, int sh, double &a, double &b) {
  const vdn_d2u v = *reinterpret_cast<const vdn_d2u *>(p + idx);
  a = sh > 0 ? v.y : v.x; b = sh < 0 ? v.x : v.y;
}
template <int MODE>
__global__ void __launch_bounds__(256) kk_ndf_march2(const MarchB *args, const int *start, int nbox, double omega, int excl, double *nrm) {
  int lo_ = 0, hi_ = nbox - 1;
  const int bid = (int)blockIdx.x;
  while (lo_ < hi_) { const int mid = (lo_ + hi_ + 1) >> 1; if (as_constant(start + mid) <= bid) lo_ = mid; else hi_ = mid - 1; }
  const MarchB B = as_constant(args + lo_);
  const FV phi = B.phi, out = B.out, rb = B.rb, sig = B.sig, slave = B.slave;
  const int has_slave = B.has_slave, kchunk = B.kchunk;
  const NdfArgs A = B.A; const Range3 r = B.r;
  const int lb = bid - as_constant(start + lo_);
  const int bx = lb % B.g[0], by = (lb / B.g[0]) % B.g[1], bz = lb / (B.g[0] * B.g[1]);
  const int sw = B.sw, rows = 64 / sw;
  const int lane = (int)threadIdx.x % sw, seg = (int)threadIdx.x / sw;
  const int ia = r.lo[0] + 2 * (bx * (sw - 2) + lane - 1);                  // nodes ia, ia + 1
  const int j = r.lo[1] + (by * (int)blockDim.y + (int)threadIdx.y) * rows + seg;
  const int k0 = r.lo[2] + bz * kchunk, k1 = min(k0 + kchunk - 1, r.hi[2]);
  const bool own = seg < rows && lane >= 1 && lane <= sw - 2 && j <= r.hi[1];
  const bool actA = own && ia <= r.hi[0], actB = own && ia + 1 <= r.hi[0];
  const int jc = min(j, r.hi[1]);
  double rmax = 0.0;
  if (k0 <= k1) {
    // per fab: where this lane's pair of row jc, plane k0 sits, the row and plane strides
    const PairAt ap = pair_at(phi, ia, jc, k0), as = pair_at(sig, ia, jc, k0), ab = pair_at(rb, ia, jc, k0);
    const long syp = phi.n0, szp = (long)phi.n0 * phi.n1, sys = sig.n0, szs = (long)sig.n0 * sig.n1, szb = (long)rb.n0 * rb.n1;
    PairAt al = ab; long szl = 0;
    const bool use_mask = has_slave == 1 && (MODE == 0 || excl == 1);
    if (use_mask) { al = pair_at(slave, ia, jc, k0); szl = (long)slave.n0 * slave.n1; }
    // the pair store: lanes with both nodes write 16 bytes, the lane whose second node lies beyond the box writes node A alone; every lane
    // executes both store instructions (the others into a sink): the number of stores in flight is known at compile time (see kk_nd_march_pair)
    const long io = fv_idx(out, min(ia, r.hi[0]), jc, k0), szo = (long)out.n0 * out.n1;
    double *op2 = actB ? out.p + io : g_nd_sink + 2 * (int)threadIdx.x;       // (g_nd_sink: 128 doubles, one 16-byte slot per lane of a wave)
    double *op1 = (actA && !actB) ? out.p + io : g_nd_sink + 2 * (int)threadIdx.x;
    const long st2 = actB ? szo : 0, st1 = (actA && !actB) ? szo : 0;
    long cp = ap.idx, cs = as.idx, cb = ab.idx, cl = al.idx;
    double q[3][3][4], sg[2][2][3];
    #define LOADP(pl, off) { _Pragma("unroll") for (int b = 0; b < 3; b++) ld_pair(phi.p, (off) + (b - 1) * syp, ap.sh, q[pl][b][1], q[pl][b][2]); }
    #define EXCHP(pl) { _Pragma("unroll") for (int b = 0; b < 3; b++) { q[pl][b][0] = lane_prev(q[pl][b][2]); q[pl][b][3] = lane_next(q[pl][b][1]); } }
    #define LOADS(dk, off) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) ld_pair(sig.p, (off) + (dj - 1) * sys, as.sh, sg[dk][dj][1], sg[dk][dj][2]); }
    #define EXCHS(dk) { _Pragma("unroll") for (int dj = 0; dj < 2; dj++) sg[dk][dj][0] = lane_prev(sg[dk][dj][2]); }
    LOADP(0, cp - szp) LOADP(1, cp) LOADS(0, cs - szs)
    EXCHP(0) EXCHP(1) EXCHS(0)
    const NdW W = nd_weights(A.f);
    for (int k = k0; k <= k1; k++, cp += szp, cs += szs, cb += szb, cl += szl, op2 += st2, op1 += st1) {
      LOADP(2, cp + szp) LOADS(1, cs)
      double rhsA, rhsB; ld_pair(rb.p, cb, ab.sh, rhsA, rhsB);
      double mA = 0.0, mB = 0.0;
      if (use_mask) ld_pair(slave.p, cl, al.sh, mA, mB);
      EXCHP(2) EXCHS(1)
      double pa[3][3][3], pb[3][3][3], sa[2][2][2], sb[2][2][2];
      #pragma unroll
      for (int pl = 0; pl < 3; pl++)
        #pragma unroll
        for (int b = 0; b < 3; b++)
          #pragma unroll
          for (int a = 0; a < 3; a++) { pa[pl][b][a] = q[pl][b][a]; pb[pl][b][a] = q[pl][b][a + 1]; }
      #pragma unroll
      for (int dk = 0; dk < 2; dk++)
        #pragma unroll
        for (int dj = 0; dj < 2; dj++)
          #pragma unroll
          for (int a = 0; a < 2; a++) { sa[dk][dj][a] = sg[dk][dj][a]; sb[dk][dj][a] = sg[dk][dj][a + 1]; }
      double KpA, dgA, KpB, dgB;
      nd_stencil(W, pa, sa, KpA, dgA);
      nd_stencil(W, pb, sb, KpB, dgB);
      const double p0A = q[1][1][1], p0B = q[1][1][2];
      const bool pdirA = ndf_pdir(A, ia, j, k), pdirB = ndf_pdir(A, ia + 1, j, k);
      // slaved to the coarser level: has_slave = 1 node mask (any union of boxes), 2 = the level is ONE box: its non-physical faces
      const bool cfA = (has_slave == 2) ? (ndf_cf(A, ia, j, k) && !pdirA) : (use_mask ? (mA != 0.0) : false);
      const bool cfB = (has_slave == 2) ? (ndf_cf(A, ia + 1, j, k) && !pdirB) : (use_mask ? (mB != 0.0) : false);
      double oA, oB;
      if (MODE == 0) {
        oA = p0A; oB = p0B;
        if (!pdirA && !cfA && dgA != 0.0) oA = p0A + omega * ((rhsA - KpA) / dgA);
        if (!pdirB && !cfB && dgB != 0.0) oB = p0B + omega * ((rhsB - KpB) / dgB);
      } else {
        oA = pdirA ? 0.0 : rhsA - KpA;
        oB = pdirB ? 0.0 : rhsB - KpB;
        if (actA && !(excl == 1 && cfA && !pdirA)) rmax = nmax(rmax, fabs(oA));
        if (actB && !(excl == 1 && cfB && !pdirB)) rmax = nmax(rmax, fabs(oB));
      }
      vdn_d2u o2; o2.x = oA; o2.y = oB;
      *reinterpret_cast<vdn_d2u *>(op2) = o2;
      *op1 = oA;
      #pragma unroll
      for (int b = 0; b < 3; b++)
        #pragma unroll
        for (int a = 0; a < 4; a++) { q[0][b][a] = q[1][b][a]; q[1][b][a] = q[2][b][a]; }
      #pragma unroll
      for (int dj = 0; dj < 2; dj++)
        #pragma unroll
        for (int a = 0; a < 3; a++) sg[0][dj][a] = sg[1][dj][a];
    }
    #undef LOADP
    #undef EXCHP
    #undef LOADS
    #undef EXCHS
  }
  if (MODE >= 1 && nrm) block_atomic_max(nrm, rmax);
}
// one launch for all boxes of a level
struct MarchSet { MarchB *d_args = nullptr; int *d_start = nullptr; int nbox = 0, tot = 0; };
static MarchSet ndf_build_march(std::vector<MarchB> &v) {
  MarchSet S; S.nbox = (int)v.size();
  if (v.empty()) return S;
  std::vector<int> start(v.size());
  for (size_t b = 0; b < v.size(); b++) {
    MarchB &B = v[b];
    const int nx = B.r.hi[0] - B.r.lo[0] + 1, ny = B.r.hi[1] - B.r.lo[1] + 1, nz = B.r.hi[2] - B.r.lo[2] + 1;
    static const bool paired = !(vdn_env("VDN_NDF_PAIR") && atoi(vdn_env("VDN_NDF_PAIR")) == 0);
    // the segment width that needs the fewest waves per node row (kk_ndf_march2: a lane carries two nodes): tiles along x / rows per wave; a box of
    // 33 nodes takes 19 lanes (17 pairs + the two feeding lanes), three rows per wave, where the power-of-two segments of round 3 gave it 32 and two
    static const bool any_width = !(vdn_env("VDN_NDF_SEGW") && atoi(vdn_env("VDN_NDF_SEGW")) == 0);
    const int per_lane = paired ? 2 : 1;
    int best = 64; double best_cost = 1e30;
    for (int sw = 4; sw <= 64; sw++) {
      if (!any_width && (sw & (sw - 1))) continue;
      const double cost = (double)((nx + per_lane * (sw - 2) - 1) / (per_lane * (sw - 2))) / (double)(64 / sw);
      if (cost < best_cost - 1e-12) { best_cost = cost; best = sw; }
    }
    B.sw = best;
    const int act = per_lane * (B.sw - 2), rows = 4 * (64 / B.sw);
    const int tiles = ((nx + act - 1) / act) * ((ny + rows - 1) / rows);
    int kchunk = nz;
    while (kchunk > 8 && tiles * ((nz + kchunk - 1) / kchunk) < 2048) kchunk = (kchunk + 1) / 2;
    if (v.size() > 16) { const int nch = (nz + 63) / 64; kchunk = (nz + nch - 1) / nch; }      // many boxes fill the chip together: whole boxes, tall ones in chunks of <= 64 planes (two warm-up planes each)
    B.kchunk = kchunk; B.g[0] = (nx + act - 1) / act; B.g[1] = (ny + rows - 1) / rows; B.g[2] = (nz + kchunk - 1) / kchunk;
    start[b] = S.tot; S.tot += B.g[0] * B.g[1] * B.g[2];
  }
  S.d_args = (MarchB *)arena_alloc(sizeof(MarchB) * v.size());
  S.d_start = (int *)arena_alloc(sizeof(int) * v.size());
  upload_staged(S.d_args, v.data(), sizeof(MarchB) * v.size());
  upload_staged(S.d_start, start.data(), sizeof(int) * v.size());
  return S;
}
template <int MODE> static void ndf_run_march(const MarchSet &S, double omega, int excl, double *nrm) {
  if (S.nbox == 0) return;
  static const bool paired = !(vdn_env("VDN_NDF_PAIR") && atoi(vdn_env("VDN_NDF_PAIR")) == 0);
  if (paired) hipLaunchKernelGGL(kk_ndf_march2<MODE>, dim3(S.tot), NBLK, 0, ctx().stream, (const MarchB *)S.d_args, (const int *)S.d_start, S.nbox, omega, excl, nrm);
  else hipLaunchKernelGGL(kk_ndf_march<MODE>, dim3(S.tot), NBLK, 0, ctx().stream, (const MarchB *)S.d_args, (const int *)S.d_start, S.nbox, omega, excl, nrm);
}

static double ndf_read(double *d) { return read_scalar1(d); }
// ---- composite nodal solve on arbitrary unions of boxes: node masks instead of per-face flags ----------------------------------
// (batched kernels: one launch per operation and level, vdn_dev.h)
// cell mask -> node mask.  mode 0 ("slave"): 1 on the nodes that are not physical Dirichlet nodes and touch a cell INSIDE the
// domain whose mask is 0 (a cell the level does not cover): the nodes of the coarse-fine interface.  mode 1 ("inside"): 1 on the
// nodes whose eight cells are all inside the domain and masked (strictly inside the region the next finer level covers)
struct MarkArgs { int dlo[3], dhi[3], per[3]; };      // a periodic direction has no outside: the mask's ghost cells hold the periodic images
struct NdmMarkB { Range3 r; int g[3]; FV out, cmask; NdfArgs A; MarkArgs D; int mode;
  static __device__ double body(const NdmMarkB &q, int i, int j, int k, int) {
    bool any_open = false, all_in = true;
    #pragma unroll
    for (int c = -1; c <= 0; c++)
      #pragma unroll
      for (int b = -1; b <= 0; b++)
        #pragma unroll
        for (int a = -1; a <= 0; a++) {
          const int ci = i + a, cj = j + b, ck = k + c;
          const bool in_dom = (q.D.per[0] || (ci >= q.D.dlo[0] && ci <= q.D.dhi[0])) && (q.D.per[1] || (cj >= q.D.dlo[1] && cj <= q.D.dhi[1])) && (q.D.per[2] || (ck >= q.D.dlo[2] && ck <= q.D.dhi[2]));
          const bool m = in_dom && fv_get(q.cmask, ci, cj, ck) != 0.0;
          if (in_dom && !m) any_open = true;
          if (!m) all_in = false;
        }
    if (q.mode == 0) { if (any_open && !ndf_pdir(q.A, i, j, k)) fv_at(q.out, i, j, k) = 1.0; }
    else if (all_in) fv_at(q.out, i, j, k) = 1.0;
    return 0.0;
  } };
struct NdmMulB { Range3 r; int g[3]; FV out, a, keep0;               // out = a where keep0 == 0, else 0   (cells)
  static __device__ double body(const NdmMulB &q, int i, int j, int k, int) { fv_at(q.out, i, j, k) = (fv_get(q.keep0, i, j, k) != 0.0) ? 0.0 : fv_get(q.a, i, j, k); return 0.0; } };
struct NdmMaskUB { Range3 r; int g[3]; FV out, u, inlev, cov; int has_inlev, has_cov;      // masked velocity, 3 components
  static __device__ double body(const NdmMaskUB &q, int i, int j, int k, int) {
    const bool keep = (!q.has_inlev || fv_get(q.inlev, i, j, k) != 0.0) && !(q.has_cov && fv_get(q.cov, i, j, k) != 0.0);
    #pragma unroll
    for (int c = 0; c < 3; c++) fv_at(q.out, i, j, k, c) = keep ? fv_get(q.u, i, j, k, c) : 0.0;
    return 0.0;
  } };
struct NdDivuB { Range3 r; int g[3]; FV u, rh; double f0, f1, f2;    // rh += D u (kk_nd_divu)
  static __device__ double body(const NdDivuB &q, int i, int j, int k, int) { nd_divu_node(q.u, q.rh, q.f0, q.f1, q.f2, i, j, k); return 0.0; } };
struct NdfNegB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV out, in; NdfArgs A;          // b = -rh, zero on physical Dirichlet nodes
  static __device__ double body(const NdfNegB &q, int i, int j, int k, int) { fv_at(q.out, i, j, k) = ndf_pdir(q.A, i, j, k) ? 0.0 : -fv_get(q.in, i, j, k); return 0.0; } };
struct NdfAddB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV a, b;
  static __device__ double body(const NdfAddB &q, int i, int j, int k, int) { fv_at(q.a, i, j, k) = fv_get(q.a, i, j, k) + fv_get(q.b, i, j, k); return 0.0; } };
struct NdfSetB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV a; double v;
  static __device__ double body(const NdfSetB &q, int i, int j, int k, int) { fv_at(q.a, i, j, k) = q.v; return 0.0; } };
struct NdfAbsmaxB { Range3 r; int g[3]; FV a, mask;
  static __device__ double body(const NdfAbsmaxB &q, int i, int j, int k, int) { return fv_get(q.mask, i, j, k) == 0.0 ? fabs(fv_get(q.a, i, j, k)) : 0.0; } };
// mode 0: phi_f = P phi_c on the slave nodes;  mode 1: phi_f += P e_c on every node that is not a physical Dirichlet node;
// mode 2: phi_f = P e_c there (0 + P e_c).  Only nodes whose coarse parent (i>>1, j>>1, k>>1) is a valid node of this coarse box
struct NdmProlongB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV pf, pc, slave, own_c; int has_own; NdfArgs Af; int clo[3], chi[3];
  static __device__ double body(const NdmProlongB &q, int i, int j, int k, int mode) {
    if (ndf_pdir(q.Af, i, j, k)) return 0.0;
    if (mode == 0 && fv_get(q.slave, i, j, k) == 0.0) return 0.0;
    const int I = i >> 1, J = j >> 1, K = k >> 1, oi = i & 1, oj = j & 1, ok = k & 1;
    if (I < q.clo[0] || I > q.chi[0] || J < q.clo[1] || J > q.chi[1] || K < q.clo[2] || K > q.chi[2]) return 0.0;
    if (q.has_own && fv_get(q.own_c, I, J, K) == 0.0) return 0.0;          // a coarse node shared by several boxes: its owner does the work, once
    double s = 0.0;
    for (int c = 0; c <= ok; c++) for (int b = 0; b <= oj; b++) for (int a = 0; a <= oi; a++) s = s + fv_get(q.pc, I + a, J + b, K + c);
    const double v = s * ((oi ? 0.5 : 1.0) * (oj ? 0.5 : 1.0) * (ok ? 0.5 : 1.0));      // = s * (1.0 / (double)((1 + oi) * (1 + oj) * (1 + ok))): the same power of two, without the division
    fv_at(q.pf, i, j, k) = (mode == 1) ? fv_get(q.pf, i, j, k) + v : v;
    return 0.0;
  } };
// modes 1 and 2 of NdmProlongB with a thread per COARSE node: it reads the eight corners of its cell once and writes its (up to) eight children
// 2I + a, 2J + b, 2K + c -- 8 gathers for 8 nodes where a thread per fine node issues 27; the sums run in NdmProlongB's order, same bits.
// r: coarse nodes whose children may lie in rf (the fine range of NdmProlongB)
struct NdmProlong8B { Range3 r; int g[3]; static constexpr int planes_per_wg = 4; FV pf, pc, own_c; int has_own; NdfArgs Af; Range3 rf;
  static __device__ double body(const NdmProlong8B &q, int I, int J, int K, int mode) {
    if (q.has_own && fv_get(q.own_c, I, J, K) == 0.0) return 0.0;          // a coarse node shared by several boxes: its owner does the work, once
    double p[2][2][2];                                                      // [c][b][a]
    #pragma unroll
    for (int c = 0; c < 2; c++)
      #pragma unroll
      for (int b = 0; b < 2; b++)
        #pragma unroll
        for (int a = 0; a < 2; a++) p[c][b][a] = fv_get(q.pc, I + a, J + b, K + c);
    #pragma unroll
    for (int ok = 0; ok < 2; ok++) {
      const int k = 2 * K + ok;
      if (k < q.rf.lo[2] || k > q.rf.hi[2]) continue;
      #pragma unroll
      for (int oj = 0; oj < 2; oj++) {
        const int j = 2 * J + oj;
        if (j < q.rf.lo[1] || j > q.rf.hi[1]) continue;
        #pragma unroll
        for (int oi = 0; oi < 2; oi++) {
          const int i = 2 * I + oi;
          if (i < q.rf.lo[0] || i > q.rf.hi[0] || ndf_pdir(q.Af, i, j, k)) continue;
          double s = 0.0;
          #pragma unroll
          for (int c = 0; c <= ok; c++)
            #pragma unroll
            for (int b = 0; b <= oj; b++)
              #pragma unroll
              for (int a = 0; a <= oi; a++) s = s + p[c][b][a];
          const double v = s * ((oi ? 0.5 : 1.0) * (oj ? 0.5 : 1.0) * (ok ? 0.5 : 1.0));
          fv_at(q.pf, i, j, k) = (mode == 1) ? fv_get(q.pf, i, j, k) + v : v;
        }
      }
    }
    return 0.0;
  } };
// res_c += full weighting of the fine residual around the fine node (2i,2j,2k), taken from the fine box that OWNS that node (its
// ghost nodes hold the neighbouring boxes' values, zero outside the level)
struct NdmRestrictB { Range3 r; int g[3]; FV res_c, res_f, own_f; NdfArgs Af, Ac;
  static __device__ double body(const NdmRestrictB &q, int i, int j, int k, int) {
    if (ndf_pdir(q.Ac, i, j, k)) return 0.0;
    if (fv_get(q.own_f, 2 * i, 2 * j, 2 * k) == 0.0) return 0.0;
    const double wt[3] = { 0.5, 1.0, 0.5 };
    const NdfArgs &Af = q.Af;
    double s = 0.0;
    for (int c = -1; c <= 1; c++) for (int b = -1; b <= 1; b++) for (int a = -1; a <= 1; a++) {
      const int ii = 2 * i + a, jj = 2 * j + b, kk = 2 * k + c;
      if (ii < Af.lo[0] - 1 || ii > Af.hi[0] + 1 || jj < Af.lo[1] - 1 || jj > Af.hi[1] + 1 || kk < Af.lo[2] - 1 || kk > Af.hi[2] + 1) continue;
      s = s + (wt[a + 1] * wt[b + 1] * wt[c + 1]) * fv_get(q.res_f, ii, jj, kk);
    }
    fv_at(q.res_c, i, j, k) = fv_get(q.res_c, i, j, k) + s * 0.125;
    return 0.0;
  } };
// node-space footprints for the views of the other level (vdn_internal.h SrcView): the nodes of a box are lo .. hi+1
static int nd_fdiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }
static std::vector<vdn_box> nd_coarse_footprints(const vdn_layout *la, int fine_lev) {          // coarse nodes the fine nodes of every fine box interpolate from
  std::vector<vdn_box> fp;
  for (const vdn_box &b : la->boxes[fine_lev]) { vdn_box o; for (int d = 0; d < 3; d++) { o.lo[d] = nd_fdiv2(b.lo[d] - 1) - 1; o.hi[d] = nd_fdiv2(b.hi[d] + 2) + 2; } fp.push_back(o); }
  return fp;
}
static std::vector<vdn_box> nd_fine_footprints(const vdn_layout *la, int crse_lev) {            // fine nodes the coarse nodes of every coarse box restrict from
  std::vector<vdn_box> fp;
  for (const vdn_box &b : la->boxes[crse_lev]) { vdn_box o; for (int d = 0; d < 3; d++) { o.lo[d] = 2 * b.lo[d] - 2; o.hi[d] = 2 * (b.hi[d] + 1) + 2; } fp.push_back(o); }
  return fp;
}
enum { NVT_C2F = 31, NVT_F2C = 32 };
static bool nd_isect(const Range3 &a, const Range3 &b, Range3 &r) {
  for (int d = 0; d < 3; d++) { r.lo[d] = std::max(a.lo[d], b.lo[d]); r.hi[d] = std::min(a.hi[d], b.hi[d]); if (r.lo[d] > r.hi[d]) return false; }
  return true;
}
// every level may be any union of boxes (properly nested in the next coarser one).  The per-iteration kernels run from descriptor
// sets built once per solve.
struct MLND {
  int nlev;
  vdn_multifab *phi[VDN_MAXLEV], *b[VDN_MAXLEV], *res[VDN_MAXLEV];
  vdn_multifab *sig[VDN_MAXLEV];                     // MASKED sigma (zero under the next finer level); the finest level's own sigma
  vdn_multifab *sigfull[VDN_MAXLEV];                 // the caller's coefficients
  vdn_multifab *slave[VDN_MAXLEV];                   // nodes slaved to the next coarser level (levels >= 1)
  vdn_multifab *own[VDN_MAXLEV];                     // 1 in the fab that owns a node shared by several boxes (lowest box index)
  vdn_multifab *skip[VDN_MAXLEV];                    // nodes left out of the norm: slaves and nodes strictly inside the finer level
  vdn_multifab *ea[VDN_MAXLEV], *eb[VDN_MAXLEV];   // Jacobi ping-pong of the correction
  std::vector<NdfArgs> A[VDN_MAXLEV]; std::vector<Range3> r[VDN_MAXLEV];      // per box: operator weights, node range, physical Dirichlet faces
  bool multi[VDN_MAXLEV]; double *d_nrm;
  MarchSet m_res[VDN_MAXLEV], m_res0[VDN_MAXLEV];    // residual of phi / of the zero field (norm of the right-hand side)
  MarchSet m_jac[VDN_MAXLEV][2];                     // Jacobi ea -> eb, eb -> ea
  BatchSet<NdmRestrictB> rst[VDN_MAXLEV];            // [fine level]
  const vdn_layout *la = nullptr;
  SrcView vf_res[VDN_MAXLEV], vf_own[VDN_MAXLEV], vc_own[VDN_MAXLEV];   // [fine level]: fine residual / fine ownership seen from the coarse boxes, coarse ownership seen from the fine boxes
  BatchSet<NdfAbsmaxB> amax[VDN_MAXLEV];
  // prolongations and additions of the iteration: the descriptor set of a (level, destination, source, kind) is built at its first use in the solve
  // (the pair loop over fine x coarse boxes and the upload cost the host 0.7-1 ms per call on a level of a thousand boxes, the GPU idle meanwhile)
  struct ProlongSet { SrcView Cv; BatchSet<NdmProlongB> s; BatchSet<NdmProlong8B> s8; };
  std::map<std::tuple<int, const void *, const void *, int>, ProlongSet> pro;
  unsigned long long base_key = 0;                   // what every descriptor of the solve follows from besides its fields: layout, boundary conditions, spacings (kept descriptor sets, vdn_internal.h)
};
// what the fab pointers and strides of a multifab follow from (keys of kept descriptor sets; amr.hip: key_mf)
static void nd_key_mf(GraphKey &k, const vdn_multifab *mf) {
  k.put(mf->la->uid); k.put(mf->lev); k.put(mf->nc); k.put(mf->ng); k.put(mf->nodal[0] | (mf->nodal[1] << 1) | (mf->nodal[2] << 2)); k.put((const void *)mf->base);
}
// the prolongation sets of a (level, destination, source, kind) kept across solves
struct NdProKept { KeeperMem mem; unsigned long uid = 0; BatchSet<NdmProlongB> s; BatchSet<NdmProlong8B> s8; };
static std::map<unsigned long long, NdProKept *> g_ndpro_kept;
void mlnd_kept_purge(unsigned long uid) {
  for (auto it = g_ndpro_kept.begin(); it != g_ndpro_kept.end();) {
    if (uid == 0 || it->second->uid == uid) { keeper_free(&it->second->mem); delete it->second; it = g_ndpro_kept.erase(it); } else ++it;
  }
}
// mode 0: slaves of level n <- P phi_{n-1};  mode 1: dst_n += P src_{n-1};  mode 2: dst_n = P src_{n-1} (dst zeroed first by the caller)
static void ml_nd_prolong(MLND &S, int n, vdn_multifab *dst, vdn_multifab *src, int mode) {
  static const bool faces_only = !(vdn_env("VDN_NDM_IFACE_FACES") && atoi(vdn_env("VDN_NDM_IFACE_FACES")) == 0);
  const auto key = std::make_tuple(n, (const void *)dst, (const void *)src, mode == 0 ? 0 : 1);
  auto hit = S.pro.find(key);
  if (hit != S.pro.end()) { hit->second.Cv.refresh(); hit->second.s.run(mode, (double *)nullptr, ctx().stream); hit->second.s8.run(mode, (double *)nullptr, ctx().stream); return; }
  MLND::ProlongSet &PS = S.pro[key];
  PS.Cv = make_view(src, nd_coarse_footprints(S.la, n), S.la->owner[n], 0, 1, NVT_C2F);
  const SrcView &Cv = PS.Cv;
  Cv.refresh();
  GraphKey gk; gk.put(0x7401); gk.put(S.base_key); gk.put(n); nd_key_mf(gk, dst); nd_key_mf(gk, src); gk.put(mode == 0 ? 0 : 1);
  gk.put((const void *)(S.slave[n] ? S.slave[n]->base : nullptr)); gk.put((const void *)(S.own[n - 1] ? S.own[n - 1]->base : nullptr));
  NdProKept *kept = nullptr;
  if (kept_family_enabled(8)) {
    auto itk = g_ndpro_kept.find(gk.h);
    if (itk != g_ndpro_kept.end()) { PS.s = itk->second->s; PS.s8 = itk->second->s8; PS.s.run(mode, (double *)nullptr, ctx().stream); PS.s8.run(mode, (double *)nullptr, ctx().stream); return; }
    kept = new NdProKept; kept->uid = S.la->uid;     // (the table is bounded at the entry of ml_nd_solve: sets already bound to this solve must not be freed here)
  }
  static const bool by_parent = !(vdn_env("VDN_NDM_PROLONG8") && atoi(vdn_env("VDN_NDM_PROLONG8")) == 0);
  std::vector<NdmProlongB> v; std::vector<NdmProlong8B> v8;
  const BoxBins cb(Cv.vbox, &Cv.have);
  for (size_t f = 0; f < S.A[n].size(); f++) {
    int qlo[3], qhi[3];
    for (int d = 0; d < 3; d++) { qlo[d] = nd_fdiv2(S.r[n][f].lo[d]); qhi[d] = nd_fdiv2(S.r[n][f].hi[d]); }
    for (int c : cb.near(qlo, qhi, 2)) {
      NdmProlongB q;
      for (int d = 0; d < 3; d++) { q.clo[d] = Cv.vbox[c].lo[d]; q.chi[d] = Cv.vbox[c].hi[d] + 1; q.r.lo[d] = std::max(S.r[n][f].lo[d], 2 * q.clo[d]); q.r.hi[d] = std::min(S.r[n][f].hi[d], 2 * q.chi[d] + 1); }
      if (q.r.lo[0] > q.r.hi[0] || q.r.lo[1] > q.r.hi[1] || q.r.lo[2] > q.r.hi[2]) continue;
      q.pf = dst->fabs[f]; q.pc = Cv.fv[c]; q.slave = S.slave[n]->fabs[f];
      const bool ho = S.own[n - 1] != nullptr && S.vc_own[n].have[c];
      q.own_c = ho ? S.vc_own[n].fv[c] : Cv.fv[c]; q.has_own = ho ? 1 : 0; q.Af = S.A[n][f];
      if (mode != 0 && by_parent) {
        NdmProlong8B t; t.pf = q.pf; t.pc = q.pc; t.own_c = q.own_c; t.has_own = q.has_own; t.Af = q.Af; t.rf = q.r;
        for (int d = 0; d < 3; d++) { t.r.lo[d] = nd_fdiv2(q.r.lo[d]); t.r.hi[d] = nd_fdiv2(q.r.hi[d]); }
        v8.push_back(t); continue;
      }
      if (mode != 0 || !faces_only) { v.push_back(q); continue; }
      // mode 0 writes slave nodes only, and a slave node lies ON a face of its box (a node inside touches eight cells of the box): the six
      // faces of the box's node range instead of the whole box (the x faces own their edges and corners, the y faces the remaining edges)
      const Range3 &bx = S.r[n][f];
      Range3 in = q.r;
      for (int d = 0; d < 3; d++) {
        for (int sd = 0; sd < 2; sd++) {
          const int at = sd ? bx.hi[d] : bx.lo[d];
          if (at < q.r.lo[d] || at > q.r.hi[d] || (sd == 1 && bx.hi[d] == bx.lo[d])) continue;
          NdmProlongB t = q; t.r = in; t.r.lo[d] = t.r.hi[d] = at;
          if (t.r.lo[0] <= t.r.hi[0] && t.r.lo[1] <= t.r.hi[1] && t.r.lo[2] <= t.r.hi[2]) v.push_back(t);
        }
        in.lo[d] = std::max(in.lo[d], bx.lo[d] + 1); in.hi[d] = std::min(in.hi[d], bx.hi[d] - 1);      // the next directions leave these faces out
      }
    }
  }
  if (kept) keeper_begin(&kept->mem);
  try { PS.s.build(v, 0, ctx().stream); PS.s8.build(v8, 0, ctx().stream); }
  catch (...) { if (kept) { keeper_end(); keeper_free(&kept->mem); delete kept; } throw; }
  if (kept) { keeper_end(); kept->s = PS.s; kept->s8 = PS.s8; g_ndpro_kept[gk.h] = kept; }
  PS.s.run(mode, (double *)nullptr, ctx().stream); PS.s8.run(mode, (double *)nullptr, ctx().stream);
}
static void ml_nd_add(MLND &S, int n, vdn_multifab *dst, vdn_multifab *src) {
  // (the key carries the shapes as well as the addresses: a multifab re-created at the same address with another ghost width must not replay stale strides -- ADVICE r4)
  GraphKey gk; gk.put(0x7402); gk.put(S.base_key); gk.put(S.la->uid); gk.put(n); nd_key_mf(gk, dst); nd_key_mf(gk, src);
  launch_batched_kept<NdfAddB>(gk.h, S.la->uid, [&](std::vector<NdfAddB> &v) {
    for (size_t f = 0; f < S.A[n].size(); f++) { NdfAddB q; q.r = S.r[n][f]; q.a = dst->fabs[f]; q.b = src->fabs[f]; v.push_back(q); }
  }, 0, (double *)nullptr, 0, ctx().stream);
}
static void ml_nd_interface(MLND &S, int n) {
  if (S.multi[n - 1]) mf_fill_boundary(S.phi[n - 1]);        // the parents of a fine node may sit in a coarse box's ghost nodes
  ml_nd_prolong(S, n, S.phi[n], S.phi[n - 1], 0);
  if (S.multi[n]) mf_fill_boundary(S.phi[n]);
}
// the damping of sweep s of the relaxation of a refined level in the composite solve: with hg_nu1 + hg_nu2 = 3 and vdn_params.hg_omega_fac1..3 > 0 the
// three sweeps take those (a three-step Chebyshev set, 1.6 / 0.9 / 0.65: one FAC iteration fewer on the tagged hierarchies), otherwise hg_omega
static double ndf_relax_omega(int s) {
  const vdn_params &P = ctx().prm;
  const double o[3] = { P.hg_omega_fac1, P.hg_omega_fac2, P.hg_omega_fac3 };
  return (g_nd_iso && P.hg_nu1 + P.hg_nu2 == 3 && s < 3 && o[0] > 0.0 && o[1] > 0.0 && o[2] > 0.0) ? o[s] : P.hg_omega;
}
// the composite residual on every level and its norm.  zero_field: the residual of phi = 0 (the norm of the right-hand side)
static double ml_nd_residual(MLND &S, bool zero_field = false) {
  hipStream_t st = ctx().stream;
  const int L = S.nlev;
  if (!zero_field) { if (S.multi[0]) mf_fill_boundary(S.phi[0]); for (int n = 1; n < L; n++) ml_nd_interface(S, n); }
  HIPCHK(hipMemsetAsync(S.d_nrm, 0, sizeof(double), st));
  for (int n = L - 1; n >= 0; n--) {
    const bool finest = n == L - 1;
    ndf_run_march<1>(zero_field ? S.m_res0[n] : S.m_res[n], 0.0, (finest && S.slave[n]) ? 1 : 0, finest ? S.d_nrm : (double *)nullptr);
    if (S.multi[n]) mf_fill_boundary(S.res[n]);
    if (finest) continue;
    S.vf_res[n + 1].refresh();
    S.rst[n + 1].run(0, (double *)nullptr, st);
    if (S.multi[n] && n > 0) mf_fill_boundary(S.res[n]);      // the ghost nodes must see the restricted part too before level n-1 restricts them
    S.amax[n].run(0, S.d_nrm, st);
  }
  comm_allreduce_max_dev(S.d_nrm, 1);
  return ndf_read(S.d_nrm);
}
// rh, phi: nodal ng 1 per level; coeffs: cells ng 1 (ghost 0 outside the level); u: cells (>= 1 ghost); dx: [lev*3+d]
static int ml_nd_solve(vdn_layout *la, vdn_multifab **rh, vdn_multifab **phi, vdn_multifab **coeffs, vdn_multifab **u, const double *dx,
                       const vdn_bc_tower *bct, int press_comp0, double rel_eps, double abs_eps, int max_iter, int *iters, double *res0, double *res) {
  const int L = la->nlev;
  REQUIRE(L >= 2 && L <= VDN_MAXLEV, "composite nodal solve: 2..%d levels", VDN_MAXLEV);
  g_nd_iso = nd_isotropic(dx);                         // (the ratio of the spacings is the same on every level)
  hipStream_t st = ctx().stream;
  if ((int)g_ndpro_kept.size() >= kept_bound(512)) { HIPCHK(hipStreamSynchronize(st)); mlnd_kept_purge(0); }     // before any kept set is bound to this solve
  const size_t mark = arena_mark();
  const vdn_params &P = ctx().prm;
  MLND S; S.nlev = L; S.la = la; S.d_nrm = (double *)arena_alloc(256);
  std::vector<vdn_multifab *> temps;
  auto T = [&](vdn_multifab *m) { temps.push_back(m); return m; };
  for (int n = 0; n < L; n++) {
    const int nb = phi[n]->nfabs();
    S.multi[n] = la->boxes[n].size() > 1 || la->pmask[0] || la->pmask[1] || la->pmask[2];      // several boxes (anywhere) or periodic images: exchanges are needed
    S.A[n].resize(nb); S.r[n].resize(nb);
    for (int f = 0; f < nb; f++) {
      const vdn_box &bx = phi[n]->vbox[f];
      NdfArgs &A = S.A[n][f];
      for (int d = 0; d < 3; d++) {
        A.f[d] = 1.0 / (36.0 * (dx[3 * n + d] * dx[3 * n + d]));
        A.lo[d] = bx.lo[d]; A.hi[d] = bx.hi[d] + 1; A.ilo[d] = A.ihi[d] = 0;
        S.r[n][f].lo[d] = A.lo[d]; S.r[n][f].hi[d] = A.hi[d];
        A.dirlo[d] = bct->ell_bc(n, f + 1, d, 0, press_comp0) == VDN_BC_DIR;
        A.dirhi[d] = bct->ell_bc(n, f + 1, d, 1, press_comp0) == VDN_BC_DIR;
        // a one-box level: its interface nodes are those of its non-physical faces (the marching kernels then skip the node mask)
        A.cflo[d] = (n > 0 && !S.multi[n] && bct->ell_bc(n, f + 1, d, 0, press_comp0) == VDN_BC_INT) ? 1 : 0;
        A.cfhi[d] = (n > 0 && !S.multi[n] && bct->ell_bc(n, f + 1, d, 1, press_comp0) == VDN_BC_INT) ? 1 : 0;
      }
    }
  }
  { GraphKey gk; gk.put(la->uid); gk.put(bct->serial); gk.put(press_comp0); gk.put(L); for (int q = 0; q < 3 * L; q++) gk.put(dx[q]); S.base_key = gk.h; }
  vdn_multifab *zero[VDN_MAXLEV], *inlev[VDN_MAXLEV], *cov[VDN_MAXLEV];
  for (int n = 0; n < L; n++) {
    S.phi[n] = phi[n]; S.b[n] = T(mf_temp(la, n, 1, 1, 3, true, 0.0)); S.res[n] = T(mf_temp(la, n, 1, 1, 3, true, 0.0));
    zero[n] = T(mf_temp(la, n, 1, 1, 3, true, 0.0));
    S.sigfull[n] = coeffs[n]; S.sig[n] = coeffs[n]; S.skip[n] = S.slave[n] = S.own[n] = nullptr;
    S.ea[n] = n >= 1 ? T(mf_temp(la, n, 1, 1, 3, true, 0.0)) : nullptr; S.eb[n] = n >= 1 ? T(mf_temp(la, n, 1, 1, 3, true, 0.0)) : nullptr;
    inlev[n] = cov[n] = nullptr;
  }
  // cell masks: inlev[n] = the cells of level n (its ghost cells included where another box of the level covers them),
  // cov[n] = the cells of level n under level n+1; node masks: own, slave, skip
  for (int n = 0; n < L; n++) {
    MarkArgs D; for (int d = 0; d < 3; d++) { D.dlo[d] = la->pd[n].lo[d]; D.dhi[d] = la->pd[n].hi[d]; D.per[d] = la->pmask[d] ? 1 : 0; }
    const int nb = phi[n]->nfabs();
    if (n >= 1 || S.multi[n]) {                         // (multi: several boxes or periodic images)
      S.own[n] = T(mf_temp(la, n, 1, 1, 3, true, 0.0));
      std::vector<NdfSetB> v1, v0;
      const BoxBins lb(la->boxes[n]);
      for (int f = 0; f < nb; f++) {
        NdfSetB q; q.r = S.r[n][f]; q.a = S.own[n]->fabs[f]; q.v = 1.0; v1.push_back(q);
        const int gf = la->local[n][f];
        // a node shared by several boxes -- or by a box and a periodic image -- belongs to the copy with the lowest global box index;
        // among the images of ONE box to the one at the lower coordinates
        int per[3], ns[3];
        for (int d = 0; d < 3; d++) { per[d] = la->pd[n].hi[d] - la->pd[n].lo[d] + 1; ns[d] = la->pmask[d] ? 1 : 0; }
        for (int sz = -ns[2]; sz <= ns[2]; sz++) for (int sy = -ns[1]; sy <= ns[1]; sy++) for (int sx = -ns[0]; sx <= ns[0]; sx++) {
          const int sh[3] = { sx * per[0], sy * per[1], sz * per[2] };
          int qlo[3], qhi[3];
          for (int d = 0; d < 3; d++) { qlo[d] = S.r[n][f].lo[d] - sh[d]; qhi[d] = S.r[n][f].hi[d] - sh[d]; }
          for (int g : lb.near(qlo, qhi, 2)) {                 // (a set of zero fills: their order is free)
            if (g > gf) break;
            if (g == gf) {                                   // my own images: only those at lower coordinates take nodes from me
              const int first = sx ? sx : (sy ? sy : sz);
              if (first >= 0) continue;
            }
            Range3 rg2; for (int d = 0; d < 3; d++) { rg2.lo[d] = la->boxes[n][g].lo[d] + sh[d]; rg2.hi[d] = la->boxes[n][g].hi[d] + 1 + sh[d]; }
            NdfSetB z; if (nd_isect(S.r[n][f], rg2, z.r)) { z.a = S.own[n]->fabs[f]; z.v = 0.0; v0.push_back(z); }
          }
        }
      }
      launch_batched(v1, 0, (double *)nullptr, 0, st);
      launch_batched(v0, 0, (double *)nullptr, 0, st);
    }
    if (n >= 1) {
      inlev[n] = T(mf_temp(la, n, 1, 1, -1, true, 0.0));
      mf_setval(inlev[n], 1.0, 0, 1, false);
      mf_fill_boundary(inlev[n]);
      S.slave[n] = T(mf_temp(la, n, 1, 1, 3, true, 0.0));
      std::vector<NdmMarkB> v;
      for (int f = 0; f < nb; f++) { NdmMarkB q; q.r = S.r[n][f]; q.out = S.slave[n]->fabs[f]; q.cmask = inlev[n]->fabs[f]; q.A = S.A[n][f]; q.D = D; q.mode = 0; v.push_back(q); }
      launch_batched(v, 0, (double *)nullptr, 0, st);
      if (S.multi[n]) mf_fill_boundary(S.slave[n]);
    }
    if (n < L - 1) {
      cov[n] = T(mf_temp(la, n, 1, 1, -1, true, 0.0));
      std::vector<NdfSetB> vc;
      const BoxBins pb(phi[n]->vbox);
      for (size_t f = 0; f < la->boxes[n + 1].size(); f++) {
        const vdn_box &fb = la->boxes[n + 1][f];
        Range3 rcov; for (int d = 0; d < 3; d++) { rcov.lo[d] = fb.lo[d] / 2; rcov.hi[d] = fb.hi[d] / 2; }
        for (int c : pb.near(rcov.lo, rcov.hi, 2)) {
          Range3 rb2; NdfSetB q; for (int d = 0; d < 3; d++) { rb2.lo[d] = phi[n]->vbox[c].lo[d] - 1; rb2.hi[d] = phi[n]->vbox[c].hi[d] + 1; }
          if (nd_isect(rcov, rb2, q.r)) { q.a = cov[n]->fabs[c]; q.v = 1.0; vc.push_back(q); }
        }
      }
      launch_batched(vc, 0, (double *)nullptr, 0, st);
      if (la->pmask[0] || la->pmask[1] || la->pmask[2]) mf_fill_boundary(cov[n]);       // ghost cells across a periodic boundary
      S.sig[n] = T(mf_temp(la, n, 1, 1, -1, true, 0.0));
      S.skip[n] = T(mf_temp(la, n, 1, 1, 3, true, 0.0));
      if (S.slave[n]) mf_copy(S.skip[n], 0, S.slave[n], 0, 1, 0);
      std::vector<NdmMulB> vm; std::vector<NdmMarkB> vk;
      for (int c = 0; c < nb; c++) {
        NdmMulB q; for (int d = 0; d < 3; d++) { q.r.lo[d] = phi[n]->vbox[c].lo[d] - 1; q.r.hi[d] = phi[n]->vbox[c].hi[d] + 1; }
        q.out = S.sig[n]->fabs[c]; q.a = coeffs[n]->fabs[c]; q.keep0 = cov[n]->fabs[c]; vm.push_back(q);
        NdmMarkB k2; k2.r = S.r[n][c]; k2.out = S.skip[n]->fabs[c]; k2.cmask = cov[n]->fabs[c]; k2.A = S.A[n][c]; k2.D = D; k2.mode = 1; vk.push_back(k2);
      }
      launch_batched(vm, 0, (double *)nullptr, 0, st);
      launch_batched(vk, 0, (double *)nullptr, 0, st);
    } else S.skip[n] = S.slave[n];
  }
  // right-hand side b = -(rh + D u) with the masked velocity (zero outside the level and under the next finer one)
  for (int n = 0; n < L; n++) {
    vdn_multifab *um = T(mf_temp(la, n, 3, 1, -1, false, 0.0));
    std::vector<NdmMaskUB> vu; std::vector<NdDivuB> vd; std::vector<NdfNegB> vn;
    for (int f = 0; f < phi[n]->nfabs(); f++) {
      NdmMaskUB q; for (int d = 0; d < 3; d++) { q.r.lo[d] = phi[n]->vbox[f].lo[d] - 1; q.r.hi[d] = phi[n]->vbox[f].hi[d] + 1; }
      q.out = um->fabs[f]; q.u = u[n]->fabs[f]; q.inlev = inlev[n] ? inlev[n]->fabs[f] : u[n]->fabs[f]; q.has_inlev = inlev[n] ? 1 : 0;
      q.cov = cov[n] ? cov[n]->fabs[f] : u[n]->fabs[f]; q.has_cov = cov[n] ? 1 : 0; vu.push_back(q);
      NdDivuB dv; dv.r = S.r[n][f]; dv.u = um->fabs[f]; dv.rh = rh[n]->fabs[f]; dv.f0 = 0.25 / dx[3 * n]; dv.f1 = 0.25 / dx[3 * n + 1]; dv.f2 = 0.25 / dx[3 * n + 2]; vd.push_back(dv);
      NdfNegB ng; ng.r = S.r[n][f]; ng.out = S.b[n]->fabs[f]; ng.in = rh[n]->fabs[f]; ng.A = S.A[n][f]; vn.push_back(ng);
    }
    launch_batched(vu, 0, (double *)nullptr, 0, st);
    launch_batched(vd, 0, (double *)nullptr, 0, st);
    launch_batched(vn, 0, (double *)nullptr, 0, st);
  }
  // descriptor sets of the per-iteration kernels
  for (int n = 0; n < L; n++) {
    std::vector<MarchB> vr, vr0, vj0, vj1; std::vector<NdfAbsmaxB> va;
    for (size_t f = 0; f < S.A[n].size(); f++) {
      MarchB q; q.phi = phi[n]->fabs[f]; q.out = S.res[n]->fabs[f]; q.rb = S.b[n]->fabs[f]; q.sig = S.sig[n]->fabs[f];
      q.slave = S.slave[n] ? S.slave[n]->fabs[f] : phi[n]->fabs[f]; q.has_slave = S.slave[n] ? (S.multi[n] ? 1 : 2) : 0; q.A = S.A[n][f]; q.r = S.r[n][f];
      vr.push_back(q);
      q.phi = zero[n]->fabs[f]; vr0.push_back(q);
      if (n >= 1) {
        MarchB j0 = q; j0.phi = S.ea[n]->fabs[f]; j0.out = S.eb[n]->fabs[f]; j0.rb = S.res[n]->fabs[f]; j0.sig = S.sigfull[n]->fabs[f]; vj0.push_back(j0);
        MarchB j1 = j0; j1.phi = S.eb[n]->fabs[f]; j1.out = S.ea[n]->fabs[f]; vj1.push_back(j1);
      }
      if (n < L - 1) { NdfAbsmaxB m; m.r = S.r[n][f]; m.a = S.res[n]->fabs[f]; m.mask = S.skip[n]->fabs[f]; va.push_back(m); }
    }
    S.m_res[n] = ndf_build_march(vr); S.m_res0[n] = ndf_build_march(vr0);
    S.m_jac[n][0] = ndf_build_march(vj0); S.m_jac[n][1] = ndf_build_march(vj1);
    S.amax[n].build(va, 16, st);
    if (n >= 1) {
      S.vf_res[n] = make_view(S.res[n], nd_fine_footprints(la, n - 1), la->owner[n - 1], 0, 1, NVT_F2C);
      S.vf_own[n] = make_view(S.own[n], nd_fine_footprints(la, n - 1), la->owner[n - 1], 0, 1, NVT_F2C); S.vf_own[n].refresh();
      if (S.own[n - 1]) { S.vc_own[n] = make_view(S.own[n - 1], nd_coarse_footprints(la, n), la->owner[n], 0, 1, NVT_C2F); S.vc_own[n].refresh(); }
      std::vector<NdmRestrictB> v;
      const BoxBins cbn(phi[n - 1]->vbox);
      for (int f = 0; f < S.vf_res[n].nboxes(); f++) {
        if (!S.vf_res[n].have[f] || !S.vf_own[n].have[f]) continue;
        NdfArgs Af; for (int d = 0; d < 3; d++) { Af.lo[d] = S.vf_res[n].vbox[f].lo[d]; Af.hi[d] = S.vf_res[n].vbox[f].hi[d] + 1; Af.dirlo[d] = Af.dirhi[d] = Af.cflo[d] = Af.cfhi[d] = Af.ilo[d] = Af.ihi[d] = 0; Af.f[d] = 0.0; }
        Range3 rf; for (int d = 0; d < 3; d++) { rf.lo[d] = nd_fdiv2(Af.lo[d] + 1); rf.hi[d] = nd_fdiv2(Af.hi[d]); }       // coarse nodes whose fine twin is a node of box f
        for (int c : cbn.near(rf.lo, rf.hi, 2)) {
          NdmRestrictB q; if (!nd_isect(rf, S.r[n - 1][c], q.r)) continue;
          q.res_c = S.res[n - 1]->fabs[c]; q.res_f = S.vf_res[n].fv[f]; q.own_f = S.vf_own[n].fv[f]; q.Af = Af; q.Ac = S.A[n - 1][c];
          v.push_back(q);
        }
      }
      S.rst[n].build(v, 0, st);
    }
  }
  // norm of the composite right-hand side = composite residual of phi = 0
  const double bnorm = ml_nd_residual(S, true);
  vdn_multifab *er = zero[0], *ee = T(mf_temp(la, 0, 1, 1, 3, true, 0.0));      // scratch of the coarse correction solve
  int ebc0[3][2];
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) ebc0[d][s] = bct->ell_bc(0, 0, d, s, press_comp0);
  int it = 0; bool conv = (bnorm == 0.0); double rn = 0.0;
  NdKeep coarse_keep;                        // the level-0 multigrid hierarchy is built once for all FAC iterations
  static const bool neg_copy = vdn_env("VDN_NDM_NEG") && atoi(vdn_env("VDN_NDM_NEG")) != 0;
  while (!conv) {
    rn = ml_nd_residual(S);
    if ((rn <= rel_eps * bnorm && bnorm < HUGE_VAL) || rn <= abs_eps) { conv = true; break; }
    if (it >= max_iter || !(rn < HUGE_VAL) || !(bnorm < HUGE_VAL)) break;
    // coarse correction K_0 e = r_0: one V-cycle of the single-level solver from e = 0, the composite residual loaded as its b
    // (VDN_NDM_NEG=1: through a negated copy and a zero-filled e, as rounds 2 built it -- same bits)
    int cyc; double r0, rr;
    if (!neg_copy) nd_solve(S.res[0], ee, coeffs[0], nullptr, dx, ebc0, 0.0, -1.0, -1, &cyc, &r0, &rr, &coarse_keep, nullptr, it == 0, true, S.phi[0]);
    else {
      mf_setval(ee, 0.0, 0, 1, true);                   // (er: every node is overwritten below, its ghost nodes are never written and stay zero)
      std::vector<NdfNegB> v;
      for (size_t c = 0; c < S.A[0].size(); c++) {
        NdfNegB q; q.r = S.r[0][c]; q.out = er->fabs[c]; q.in = S.res[0]->fabs[c]; q.A = S.A[0][c]; for (int d = 0; d < 3; d++) { q.A.dirlo[d] = q.A.dirhi[d] = 0; }
        v.push_back(q);
      }
      launch_batched(v, 0, (double *)nullptr, 0, st);
      nd_solve(er, ee, coeffs[0], nullptr, dx, ebc0, 0.0, -1.0, -1, &cyc, &r0, &rr, &coarse_keep, nullptr, it == 0);     // (first correction: from the nested iteration, hg_fmg)
    }
    if (neg_copy) ml_nd_add(S, 0, S.phi[0], ee);           // (otherwise phi_0 += e_0 was done where e_0 was stored)
    // the finer levels, coarsest first, in correction form (oracle: vo_ml_nd_solve): e_n = P e_{n-1} (trilinear, not on physical Dirichlet nodes),
    // the sweeps of K_n e_n = r_n -- r_n from the top of the iteration -- with the interface nodes held at P e_{n-1}, phi_n += e_n.  (Rounds 2-3 applied
    // every correction to phi on all finer levels at once and recomputed the composite residual before each relaxation: three residual passes over
    // the finest of three levels per iteration where this form makes one.)
    vdn_multifab *src = ee;
    for (int n = 1; n < L; n++) {
      if (S.multi[n - 1]) mf_fill_boundary(src);         // a parent node may sit in a coarse box's ghost layer
      mf_setval(S.ea[n], 0.0, 0, 1, true);
      ml_nd_prolong(S, n, S.ea[n], src, 2);
      vdn_multifab *a = S.ea[n], *b2 = S.eb[n];
      for (int s = 0; s < P.hg_nu1 + P.hg_nu2; s++) {
        if (S.multi[n]) mf_fill_boundary(a);
        ndf_run_march<0>(S.m_jac[n][s & 1], ndf_relax_omega(s), 0, (double *)nullptr);
        std::swap(a, b2);
      }
      ml_nd_add(S, n, S.phi[n], a);
      src = a;
    }
    it++;
  }
  if (S.multi[0]) mf_fill_boundary(S.phi[0]);
  for (int n = 1; n < L; n++) ml_nd_interface(S, n);
  if (iters) *iters = it; if (res0) *res0 = bnorm; if (res) *res = rn;
  HIPCHK(hipStreamSynchronize(st));
  for (size_t i = temps.size(); i-- > 0;) mf_temp_free(temps[i]);
  arena_release(mark);
  return conv ? 0 : 1;
}
// hgproject.f90:17-178 with nlevs > 1 (rel 1e-11 for two levels, 1e-10 for more: hgproject.f90:115-119)
static void do_ml_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold, vdn_multifab **rhohalf,
                            vdn_multifab **p, vdn_multifab **gp, const double *dx, double dt, const vdn_bc_tower *bct, int press_comp0) {
  const size_t mark = arena_mark();
  const int L = mla->nlev;
  REQUIRE(L <= VDN_MAXLEV, "hgproject: at most %d levels", VDN_MAXLEV);
  vdn_multifab *rh[VDN_MAXLEV], *phi[VDN_MAXLEV], *gphi[VDN_MAXLEV], *coeffs[VDN_MAXLEV];
  for (int n = 0; n < L; n++) {
    rh[n] = mf_temp(mla, n, 1, 1, 3, true, 0.0); phi[n] = mf_temp(mla, n, 1, 1, 3, true, 0.0);
    gphi[n] = mf_temp(mla, n, 3, 0, -1, false, 0.0); coeffs[n] = mf_temp(mla, n, 1, 1, -1, true, 0.0);
    hg_level_pre(proj_type, unew[n], uold[n], rhohalf[n], gp[n], coeffs[n], dt, bct);
  }
  // under a finer level a level's coefficient is the mean of the fine sigma, not 1 / (averaged-down rho): see oracle/vo_amr.c (vo_ml_hgproject) --
  // the FAC iteration diverged on one-cell density jumps of 300 : 1 and more with the softer operator.  The composite equations read sigma on
  // uncovered cells only; this changes the preconditioner (level V-cycle, relaxation of intermediate levels), not the system.
  for (int n = L - 1; n >= 1; n--) { ml_cc_restriction(coeffs[n - 1], coeffs[n], 0, 1); mf_fill_boundary(coeffs[n - 1]); }
  double rel = ctx().prm.hg_rel_eps > 0.0 ? ctx().prm.hg_rel_eps : (L == 2 ? 1.e-11 : 1.e-10);
  double abs_eps = -1.0;
  if (proj_type == VDN_INITIAL_PROJECTION && ctx().prm.prob_type == 4) abs_eps = 1.e-12;
  int it; double r0, rr;
  int rc = ml_nd_solve(mla, rh, phi, coeffs, unew, dx, bct, press_comp0, rel, abs_eps, ctx().prm.hg_max_iter, &it, &r0, &rr);
  ctx().solver_cycles[1] = it; ctx().solver_res0[1] = r0; ctx().solver_res[1] = rr;
  solver_check(rc, "composite nodal solve", it, rr, r0);
  for (int n = 0; n < L; n++) hg_level_post(proj_type, unew[n], uold[n], rhohalf[n], gp[n], p[n], gphi[n], phi[n], dx + 3 * n, dt);
  for (int n = L - 1; n >= 1; n--) ml_cc_restriction(gp[n - 1], gp[n], 0, 3);      // hgproject.f90:355-357
  for (int n = 0; n < L; n++) { mf_fill_boundary(gp[n]); mf_fill_boundary(p[n]); }
  ml_restrict_and_fill(L, unew, 0, 0, 3, false, bct);                  // hgproject.f90:364-366
  for (int n = L - 1; n >= 0; n--) { mf_temp_free(coeffs[n]); mf_temp_free(gphi[n]); mf_temp_free(phi[n]); mf_temp_free(rh[n]); }
  arena_release(mark);
}
