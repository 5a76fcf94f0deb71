/* oracle/vo_plot.c -- derived plot quantities (makevort.f90): vorticity and velocity magnitude.
 * TEST INFRASTRUCTURE ONLY (see vo.h).  parity unpinned.
 *
 * The caller fills the ghost cells of u first (fill_boundary + physbc on the velocity components,
 * makevort.f90:34-38); only one ghost cell is read. */
#include <math.h>
#include "vo.h"

static int fix3(int p) { return p == VDN_INLET || p == VDN_NO_SLIP_WALL; }                          /* makevort.f90:188-195 */
static int fix2(int p) { return p == VDN_INLET || p == VDN_SLIP_WALL || p == VDN_NO_SLIP_WALL; }    /* makevort.f90:116-117 etc. */

/* d(u_c)/dx_d at cell q of a 3-D box: centred (uycen, makevort.f90:568-572, and its permutations), or the one-sided three-point
 * forms next to an inflow / no-slip face (uylo / uyhi, :574-584), side = -1 low face, +1 high face, 0 interior */
static double der3(const vo_fab *u, int c, int d, int side, int i, int j, int k, double dxd)
{
  const int e[3] = { d == 0, d == 1, d == 2 };
  const double up = VF(u, i + e[0], j + e[1], k + e[2], c), u0 = VF(u, i, j, k, c), um = VF(u, i - e[0], j - e[1], k - e[2], c);
  if (side < 0) return (up + 3.0 * u0 - 4.0 * um) / (3.0 * dxd);
  if (side > 0) return -(um + 3.0 * u0 - 4.0 * up) / (3.0 * dxd);
  return 0.5 * (up - um) / dxd;
}

/* makevort_3d (makevort.f90:158-682): faces, edges and corners all follow one rule per direction */
void vo_makevort(vo_fab *vort, int comp, const vo_fab *u, const double dx[3], const vo_bc *bc)
{
  if (u->dm == 2) {
    /* makevort_2d (makevort.f90:93-156): the one-sided forms divide by dx, not 3 dx, and slip walls count too;
     * the four face loops run lo-x, hi-x, lo-y, hi-y and each OVERWRITES the cell, so at a corner the y rule wins
     * and the x derivative there is the centred one */
    for (int j = u->lo[1]; j <= u->hi[1]; j++)
      for (int i = u->lo[0]; i <= u->hi[0]; i++) {
        double vx = (VF(u, i + 1, j, 0, 1) - VF(u, i - 1, j, 0, 1)) / (2.0 * dx[0]);
        double uy = (VF(u, i, j + 1, 0, 0) - VF(u, i, j - 1, 0, 0)) / (2.0 * dx[1]);
        if (i == u->lo[0] && fix2(bc->phys[0][0])) vx = (VF(u, i + 1, j, 0, 1) + 3.0 * VF(u, i, j, 0, 1) - 4.0 * VF(u, i - 1, j, 0, 1)) / dx[0];
        if (i == u->hi[0] && fix2(bc->phys[0][1])) vx = -(VF(u, i - 1, j, 0, 1) + 3.0 * VF(u, i, j, 0, 1) - 4.0 * VF(u, i + 1, j, 0, 1)) / dx[0];
        int ylo = j == u->lo[1] && fix2(bc->phys[1][0]), yhi = j == u->hi[1] && fix2(bc->phys[1][1]);
        if (ylo || yhi) vx = (VF(u, i + 1, j, 0, 1) - VF(u, i - 1, j, 0, 1)) / (2.0 * dx[0]);
        if (ylo) uy = (VF(u, i, j + 1, 0, 0) + 3.0 * VF(u, i, j, 0, 0) - 4.0 * VF(u, i, j - 1, 0, 0)) / dx[1];
        if (yhi) uy = -(VF(u, i, j - 1, 0, 0) + 3.0 * VF(u, i, j, 0, 0) - 4.0 * VF(u, i, j + 1, 0, 0)) / dx[1];
        VF(vort, i, j, 0, comp) = vx - uy;
      }
    return;
  }
  for (int k = u->lo[2]; k <= u->hi[2]; k++)
    for (int j = u->lo[1]; j <= u->hi[1]; j++)
      for (int i = u->lo[0]; i <= u->hi[0]; i++) {
        const int q[3] = { i, j, k };
        int side[3];
        for (int d = 0; d < 3; d++) {
          side[d] = 0;
          if (q[d] == u->lo[d] && fix3(bc->phys[d][0])) side[d] = -1;
          if (q[d] == u->hi[d] && fix3(bc->phys[d][1])) side[d] = 1;       /* the hi loops run after the lo loops */
        }
        const double uy = der3(u, 0, 1, side[1], i, j, k, dx[1]), uz = der3(u, 0, 2, side[2], i, j, k, dx[2]);
        const double vx = der3(u, 1, 0, side[0], i, j, k, dx[0]), vz = der3(u, 1, 2, side[2], i, j, k, dx[2]);
        const double wx = der3(u, 2, 0, side[0], i, j, k, dx[0]), wy = der3(u, 2, 1, side[1], i, j, k, dx[1]);
        VF(vort, i, j, k, comp) = sqrt((wy - vz) * (wy - vz) + (uz - wx) * (uz - wx) + (vx - uy) * (vx - uy));   /* vorfun, :676-680 */
      }
}

/* makemagvel_2d / _3d (makevort.f90:684-724) */
void vo_makemagvel(vo_fab *magvel, int comp, const vo_fab *u)
{
  for (int k = u->lo[2]; k <= u->hi[2]; k++)
    for (int j = u->lo[1]; j <= u->hi[1]; j++)
      for (int i = u->lo[0]; i <= u->hi[0]; i++) {
        double s = VF(u, i, j, k, 0) * VF(u, i, j, k, 0) + VF(u, i, j, k, 1) * VF(u, i, j, k, 1);
        if (u->dm == 3) s = s + VF(u, i, j, k, 2) * VF(u, i, j, k, 2);
        VF(magvel, i, j, k, comp) = sqrt(s);
      }
}
