/*
 * oracle/vo.h -- CPU restatement ("oracle") of VARDEN's advance_timestep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under varden_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the reported CPU baseline.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (BoxLib-Codes/VARDEN, pure Fortran 90) ships
 * no golden vectors, fixtures or known-answer tests for this path, and it
 * cannot be built here: it needs FBoxLib (external, absent, unpinned) and a generated
 * probin.f90.  Building it against hand-written stand-ins for those modules is not allowed
 * by this round's rules, so this restatement is checked only against (i) a line-by-line
 * reading of the reference sources it cites and (ii) the known-answer invariants listed in
 * SURVEY.md section 8(c) (tests/test_oracle_invariants.py).  Both multigrid solvers live in FBoxLib and
 * have NO reference text at all: oracle/vo_mg_cc.c and vo_mg_nd.c define the discrete systems
 *  and an algorithm of our own.
 *
 * Arithmetic: IEEE f64, compiled with -ffp-contract=off so that expression order is the
 * only thing that determines the bits (the reference CPU build has no FMA contraction on
 * baseline x86-64).  Every function cites the reference file:line it follows.
 */
#ifndef VO_H
#define VO_H
#include <stddef.h>
#include <math.h>
#include "../include/varden_amd.h"   /* vdn_params and the bc enums: the product's PUBLIC header */

/* fmin / fmax as inline selects.  Without -ffinite-math-only gcc leaves them as calls into libm (16 call sites in the Godunov loops, none
 * of them vectorisable).  These return exactly what glibc's do on this image, signed zeros and NaNs included (probed: for two zeros and for
 * a NaN first operand the SECOND operand comes back; a NaN second operand gives the first), so no value changes anywhere. */
static inline double vo_fmin(double x, double y) { return (x < y || y != y) ? x : y; }
static inline double vo_fmax(double x, double y) { return (x > y || y != y) ? x : y; }
#define fmin(x, y) vo_fmin((x), (y))
#define fmax(x, y) vo_fmax((x), (y))

#ifdef __cplusplus
extern "C" {
#endif

/* one fab in BoxLib layout: p(lo-ng:hi+ng+nd, ..., 0:nc-1), x fastest */
typedef struct vo_fab {
  double *p;
  int lo[3], hi[3];   /* valid CELL box */
  int ng;
  int nd[3];          /* nodal flags */
  int nc;
  long n[3];          /* allocated extents */
  long sc;            /* component stride */
  int gz;             /* ghost width along z (= ng in 3-D, 0 for the one-plane fabs of a 2-D run) */
  int dm;             /* 2 or 3 */
} vo_fab;

static inline void vo_fab_init(vo_fab *f, double *p, const int *lo, const int *hi, int ng,
                               const int *nd, int nc) {
  f->p = p; f->ng = ng; f->nc = nc; f->gz = ng; f->dm = 3;
  for (int d = 0; d < 3; d++) {
    f->lo[d] = lo[d]; f->hi[d] = hi[d]; f->nd[d] = nd ? nd[d] : 0;
    f->n[d] = hi[d] - lo[d] + 1 + f->nd[d] + 2 * ng;
  }
  f->sc = f->n[0] * f->n[1] * f->n[2];
}
/* a fab of a 2-D run: BoxLib layout p(lo1-ng:hi1+ng, lo2-ng:hi2+ng, nc) held as one z-plane (lo[2] = hi[2] = 0) */
static inline void vo_fab_init2d(vo_fab *f, double *p, const int *lo, const int *hi, int ng, const int *nd, int nc) {
  f->p = p; f->ng = ng; f->nc = nc; f->gz = 0; f->dm = 2;
  for (int d = 0; d < 3; d++) { f->lo[d] = d < 2 ? lo[d] : 0; f->hi[d] = d < 2 ? hi[d] : 0; f->nd[d] = (nd && d < 2) ? nd[d] : 0; }
  f->n[0] = f->hi[0] - f->lo[0] + 1 + f->nd[0] + 2 * ng; f->n[1] = f->hi[1] - f->lo[1] + 1 + f->nd[1] + 2 * ng; f->n[2] = 1;
  f->sc = f->n[0] * f->n[1];
}
static inline long vo_size(const vo_fab *f) { return f->sc * f->nc; }
static inline long vo_idx(const vo_fab *f, int i, int j, int k, int c) {
  return (long)(i - f->lo[0] + f->ng) + f->n[0] * ((long)(j - f->lo[1] + f->ng) +
         f->n[1] * (long)(k - f->lo[2] + f->gz)) + f->sc * c;
}
#define VF(f, i, j, k, c) ((f)->p[vo_idx((f), (i), (j), (k), (c))])

/* bc tables for ONE box: phys[dir][side]; adv[dir][side][comp] with comp 0-based in the
 * reference order vel(0..dm-1), rho, tracers, press, extrap (define_bc_tower.f90:171-193);
 * ell[dir][side][comp] */
#define VO_MAXCOMP 16
typedef struct vo_bc {
  int phys[3][2];
  int adv[3][2][VO_MAXCOMP];
  int ell[3][2][VO_MAXCOMP];
  int ncomp_adv, ncomp_ell;
  int press_comp, extrap_comp;   /* 0-based */
} vo_bc;

void vo_bc_build(vo_bc *bc, const int phys[3][2], int dm, int nscal);

/* ---- ghost cells ---------------------------------------------------------------------- */
/* single-box multifab_fill_boundary: periodic wrap only (pmask), cell/face/nodal aware */
void vo_fill_boundary(vo_fab *f, const int pmask[3]);
/* multifab_physbc.f90:238-561, component scomp (0-based) with adv bc component bccomp (0-based) */
void vo_physbc(vo_fab *f, int scomp, int bccomp, int nc, const vo_bc *bc, const vdn_params *prm);

/* ---- Godunov pieces --------------------------------------------------------------------- */
/* slope.f90:148-588; out: slope fab with ng=1 (cells [lo-1,hi+1]), comps [0,nc) of s using
 * adv bc comps bccomp+c; dir = 0,1,2 */
void vo_slope(const vo_fab *s, vo_fab *slope, int dir, int nc, int bccomp, const vo_bc *bc,
              int slope_order);
/* velpred.f90:1776-2765 (kernel only, no ghost fill) */
void vo_velpred(const vo_fab *u, vo_fab *umac[3], const vo_fab *force, const double dx[3],
                double dt, const vo_bc *bc, const vdn_params *prm);
/* mkflux.f90:1186-2567 (kernel only) */
void vo_mkflux(const vo_fab *s, vo_fab *sedge[3], vo_fab *flux[3], vo_fab *umac[3],
               const vo_fab *force, const vo_fab *mac_rhs, const double dx[3], double dt,
               int is_vel, const int *is_cons, int bccomp, const vo_bc *bc, const vdn_params *prm);
/* update.f90:186-278 (kernel only) */
void vo_update(const vo_fab *sold, vo_fab *umac[3], vo_fab *sedge[3], vo_fab *flux[3],
               const vo_fab *force, vo_fab *snew, const double dx[3], double dt, int is_vel,
               const int *is_cons);
/* mkforce.f90:144-236 / 333-402 (kernels only) */
void vo_mkvelforce(vo_fab *vel_force, const vo_fab *ext, const vo_fab *gp, const vo_fab *s,
                   const vo_fab *lapu, double visc_fac, const vdn_params *prm);
void vo_mkscalforce(vo_fab *scal_force, const vo_fab *ext, const vo_fab *laps, double diff_fac,
                    const vdn_params *prm);
/* make_at_halftime.f90:95-115 */
void vo_make_at_halftime(vo_fab *rhohalf, int out_comp, const vo_fab *sold, const vo_fab *snew, int in_comp);
/* estdt.f90:131-181 + 69-78 */
double vo_estdt(const vo_fab *u, const vo_fab *s, const vo_fab *gp, const vo_fab *ext, const double dx[3],
                double dtold, const vdn_params *prm);

/* ---- MAC projection ------------------------------------------------------------------------ */
/* macproject.f90:250-278 */
void vo_divumac(vo_fab *umac[3], vo_fab *rh, const double dx[3]);
/* macproject.f90:361-401 */
void vo_mk_mac_coeffs(const vo_fab *rho, vo_fab *beta[3]);
/* macproject.f90:578-645 restated with ghost-phi gradients at box faces (see vo_macproject.c) */
void vo_mkumac(vo_fab *umac[3], const vo_fab *phi, vo_fab *beta[3], const double dx[3], const int ellbc[3][2]);
/* our cell-centred multigrid (replaces FBoxLib ml_cc_solve, mac_multigrid.f90:53) */
/* max-norm accumulation of the solvers' stopping tests: a NaN turns the norm into +inf (fmax alone would drop it and a solve that blew up
 * would pass for converged with a zero residual) -- the HIP reductions do the same, and solver_check fails the call on a non-finite norm */
static inline double vo_nrm_acc(double nrm, double x) { x = x < 0.0 ? -x : x; return (x != x) ? HUGE_VAL : (x > nrm ? x : nrm); }
typedef struct vo_mgstat { int cycles; double res0, res; } vo_mgstat;
int  vo_cc_solve(const vo_fab *rh, vo_fab *phi, vo_fab *beta[3], const double dx[3], const int ellbc[3][2],
                 double rel_eps, double abs_eps, int max_iter, int nu1, int nu2, int nub, int fmg, vo_mgstat *st);
int  vo_cc_solve_ab(const vo_fab *rh, vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2],
                    double rel_eps, double abs_eps, int max_iter, int nu1, int nu2, int nub, int fmg, vo_mgstat *st);
/* explicit_diffusive_term.f90:16-88 = FBoxLib cc_applyop with alpha = 0, beta = -1: out(comp) = laplacian(data(comp)) with the
 * ell bc of bc component bccomp; Dirichlet faces use the ghost-cell value as the face value */
void vo_explicit_diffusive_term(vo_fab *lap, const vo_fab *data, int comp, int bccomp, const double dx[3], const vo_bc *bc);
/* viscsolve.f90:19-306 and 308-515, one level / one box */
void vo_visc_solve(vo_fab *unew, const vo_fab *lapu, const vo_fab *rho, const vo_fab *mac_rhs, const double dx[3], double mu,
                   const vo_bc *bc, const int pmask[3], const vdn_params *prm, vo_mgstat *st);
void vo_diff_scalar_solve(vo_fab *snew, const vo_fab *laps, const double dx[3], double mu, const vo_bc *bc, const int pmask[3],
                          const vdn_params *prm, int icomp, int bccomp, vo_mgstat *st);
void vo_cc_smooth_ab(const vo_fab *rh, vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2], int nsweeps);
void vo_cc_smooth_ab_iface(const vo_fab *rh, vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2], int nsweeps);
/* vo_plot.c: derived plot quantities (makevort.f90) */
void vo_makevort(vo_fab *vort, int comp, const vo_fab *u, const double dx[3], const vo_bc *bc);
void vo_makemagvel(vo_fab *magvel, int comp, const vo_fab *u);
void vo_cc_smooth(const vo_fab *rh, vo_fab *phi, vo_fab *beta[3], const double dx[3], const int ellbc[3][2],
                  int nsweeps);
/* test hooks: the operators the two solvers iterate on, applied once (compared with independently assembled scipy matrices) */
void vo_cc_apply(const vo_fab *phi, const vo_fab *alpha, vo_fab *beta[3], const double dx[3], const int ellbc[3][2], vo_fab *out);
void vo_nd_apply(const vo_fab *phi, const vo_fab *coeffs, const double dx[3], const int ellbc[3][2], const int pmask[3], vo_fab *out);
/* macproject.f90:20-133, single level */
void vo_macproject(vo_fab *umac[3], vo_fab *rho, const vo_fab *mac_rhs, const double dx[3], const vo_bc *bc,
                   const int pmask[3], const vdn_params *prm, vo_mgstat *st);

/* ---- HG projection -------------------------------------------------------------------------- */
/* hgproject.f90:434-513 */
void vo_create_uvec(vo_fab *unew, const vo_fab *uold, const vo_fab *rhohalf, vo_fab *gp, double dt,
                    const vo_bc *bc, int proj_type);
/* hgproject.f90:543-577 */
void vo_mkgphi(vo_fab *gphi, const vo_fab *phi, const double dx[3]);
/* hgproject.f90:638-698 */
void vo_hg_update(int proj_type, vo_fab *unew, const vo_fab *uold, vo_fab *gp, const vo_fab *gphi,
                  const vo_fab *rhohalf, vo_fab *p, const vo_fab *phi, double dt);
/* our nodal multigrid (replaces FBoxLib ml_nd_solve, hg_multigrid.f90:95) */
void vo_nd_divu(const vo_fab *u, vo_fab *rh, const double dx[3], const int ellbc[3][2]);
int  vo_nd_solve(vo_fab *rh, vo_fab *phi, const vo_fab *coeffs, const vo_fab *u, const double dx[3],
                 const int ellbc[3][2], const int pmask[3], double rel_eps, double abs_eps, int max_iter,
                 int nu1, int nu2, int nub, double omega, int fmg, const double *om_pre, vo_mgstat *st);
/* the multi-step damping sets of the nodal solver (hg_omega_pre1/2, hg_omega_fac1..3) were tuned on dx = dy = dz and lose to (or diverge against)
 * the plain hg_omega when the spacings differ by more than a quarter: they are used only on grids with max(dx) <= 1.25 min(dx) (round 4) */
static inline int vo_nd_isotropic(const double *dx, int dm) {
  double lo = dx[0], hi = dx[0];
  for (int d = 1; d < dm; d++) { if (dx[d] < lo) lo = dx[d]; if (dx[d] > hi) hi = dx[d]; }
  return hi <= 1.25 * lo;
}
/* the damping pair of the two pre-smoothing sweeps of the nodal V-cycle (NULL: hg_omega for both) */
static inline const double *vo_om_pre(const vdn_params *prm) { return (prm->hg_omega_pre1 > 0.0 && prm->hg_omega_pre2 > 0.0) ? &prm->hg_omega_pre1 : NULL; }
/* hgproject.f90:17-178 + hg_multigrid.f90:18-119, single level */
void vo_hgproject(int proj_type, vo_fab *unew, const vo_fab *uold, vo_fab *rhohalf, vo_fab *p, vo_fab *gp,
                  const double dx[3], double dt, const vo_bc *bc, const int pmask[3], const vdn_params *prm,
                  vo_mgstat *st);

/* ---- advance_timestep.f90:26-170, single level, single box ---------------------------------- */
typedef struct vo_state {
  vo_fab uold, sold, unew, snew, gp, p, ext_vel_force, ext_scal_force;
} vo_state;
void vo_advance_timestep(vo_state *S, const double dx[3], double dt, const vo_bc *bc, const int pmask[3],
                         const vdn_params *prm, int proj_type, vo_mgstat st[2], double phase_sec[4]);

/* ---- initdata.f90:201-311 (prob_type 1 and 2) ----------------------------------------------- */
void vo_initdata(vo_fab *u, vo_fab *s, const double dx[3], int prob_type);

/* ---- multi-level AMR (oracle/vo_amr.c, vo_hgproject.c): FBoxLib's multi-level operators (our definitions) and the multilevel projections / advance ----
 * A level of a hierarchy is a LIST OF BOXES (round 5).  Its cell- and node-centred fields are LEVEL ARRAYS: one vo_fab over the bounding box of the
 * boxes (lo / hi = the bounding box) with `valid` marking the cells of the union -- every such field is single-valued on the level, and a cell outside
 * the union but inside the allocation holds what a ghost cell there holds in BoxLib (the coarse interpolation, a physical boundary value), whichever box
 * it is a ghost cell of.  Face-centred MAC velocities, edge states and fluxes are held PER BOX (vo_bmf): the upwinding's dead band is per box
 * (velpred.f90:1965-1980), so two boxes may disagree on the face they share, as in BoxLib.  A NULL vo_level (or nbox = 1) is the one-box level of
 * rounds 2-4: the fab's own lo..hi.  Level 0 is one box, the domain.  Periodic sides are not supported on hierarchies here (pmask must be 0). */
typedef struct vo_level {
  int nbox;
  const int *boxes;            /* [nbox][2][3]: lo, hi of every box, cell indices of the level */
  int blo[3], bhi[3];          /* bounding box */
  int mg;                      /* `valid` covers blo - mg .. bhi + mg */
  unsigned char *valid;        /* 1 on the cells of the union; x fastest */
} vo_level;
void vo_level_build(vo_level *L, int nbox, const int *boxes);
void vo_level_free(vo_level *L);
static inline int vo_valid(const vo_level *L, int i, int j, int k) {
  const int g = L->mg, nx = L->bhi[0] - L->blo[0] + 1 + 2 * g, ny = L->bhi[1] - L->blo[1] + 1 + 2 * g, nz = L->bhi[2] - L->blo[2] + 1 + 2 * g;
  const int a = i - L->blo[0] + g, b = j - L->blo[1] + g, c = k - L->blo[2] + g;
  if (a < 0 || a >= nx || b < 0 || b >= ny || c < 0 || c >= nz) return 0;
  return L->valid[(size_t)a + (size_t)nx * ((size_t)b + (size_t)ny * (size_t)c)];
}
static inline int vo_lv_multi(const vo_level *L) { return L && L->nbox > 1; }
/* cell (i,j,k) is a cell of the level whose cell-centred level array is f */
static inline int vo_lv_valid(const vo_level *L, const vo_fab *f, int i, int j, int k) {
  if (vo_lv_multi(L)) return vo_valid(L, i, j, k);
  return i >= f->lo[0] && i <= f->hi[0] && j >= f->lo[1] && j <= f->hi[1] && k >= f->lo[2] && k <= f->hi[2];
}
/* a face- (or cell-) centred field held box by box */
typedef struct vo_bmf { int nbox; vo_fab *f; } vo_bmf;

/* the `_g` forms take the box lists (lev: [nlev], an entry may be NULL = one box); the plain forms are the one-box hierarchies of rounds 2-4 */
void vo_ml_cc_restriction(vo_fab *crse, const vo_fab *fine, int icomp, int nc);
void vo_ml_cc_restriction_g(vo_fab *crse, const vo_fab *fine, const vo_level *Lf, int icomp, int nc);
void vo_ml_edge_restriction(vo_fab *crse, const vo_fab *fine, int dir);
void vo_ml_edge_restriction_g(vo_fab *crse, const vo_fab *fine, const vo_level *Lf, int dir);
void vo_fill_ghost_cells(vo_fab *fine, const vo_fab *crse, int icomp, int nc);
void vo_fill_ghost_cells_g(vo_fab *fine, const vo_fab *crse, const vo_level *Lf, int icomp, int nc);
void vo_create_umac_grown(vo_fab *fine, const vo_fab *crse, int dir);
void vo_ml_restrict_and_fill(int nlev, vo_fab **mf, int icomp, int bcomp, int nc, int same_boundary, const vo_bc *bc, const int pmask[3],
                             const int *pd, const vdn_params *prm);
void vo_ml_restrict_and_fill_g(int nlev, const vo_level *const *lev, vo_fab **mf, int icomp, int bcomp, int nc, int same_boundary, const vo_bc *bc, const int pmask[3],
                               const int *pd, const vdn_params *prm);
int  vo_ml_cc_solve(int nlev, vo_fab **rh, vo_fab **phi, vo_fab **alpha, vo_fab **beta, const double *dx, const int ellbc[][3][2], const int pmask[3], const int *pd,
                    double rel_eps, int max_iter, const vdn_params *prm, vo_fab **beta_base, vo_mgstat *st);
/* ghost: [lev*6 + 2 d + side] (may be NULL): on return the value of phi beyond the coarse-fine interface next to each valid cell (arrays indexed like rh) */
int  vo_ml_cc_solve_g(int nlev, const vo_level *const *lev, vo_fab **rh, vo_fab **phi, vo_fab **alpha, vo_fab **beta, const double *dx, const int ellbc[][3][2], const int pmask[3],
                      const int *pd, double rel_eps, int max_iter, const vdn_params *prm, vo_fab **beta_base, vo_mgstat *st, double **ghost);
/* umac: [lev*3 + d], one fab per box of the level */
void vo_ml_macproject(int nlev, vo_fab **umac, vo_fab **rho, vo_fab **mac_rhs, const double *dx, const vo_bc *bc, const int pmask[3], const int *pd,
                      const vdn_params *prm, vo_mgstat *st);
void vo_ml_macproject_g(int nlev, const vo_level *const *lev, vo_bmf *umac, vo_fab **rho, vo_fab **mac_rhs, const double *dx, const vo_bc *bc, const int pmask[3], const int *pd,
                        const vdn_params *prm, vo_mgstat *st);

int  vo_ml_nd_solve(int nlev, vo_fab **rh, vo_fab **phi, vo_fab **coeffs, vo_fab **u, const double *dx, const int ellbc[][3][2], const int pmask[3],
                    double rel_eps, double abs_eps, int max_iter, const vdn_params *prm, vo_mgstat *st);
int  vo_ml_nd_solve_g(int nlev, const vo_level *const *lev, vo_fab **rh, vo_fab **phi, vo_fab **coeffs, vo_fab **u, const double *dx, const int ellbc[][3][2], const int pmask[3],
                      const int *pd, double rel_eps, double abs_eps, int max_iter, const vdn_params *prm, vo_mgstat *st);
void vo_ml_visc_solve(int nlev, vo_fab **unew, vo_fab **lapu, vo_fab **rho, vo_fab **mac_rhs, const double *dx, double mu, const vo_bc *bc,
                      const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st);
void vo_ml_visc_solve_g(int nlev, const vo_level *const *lev, vo_fab **unew, vo_fab **lapu, vo_fab **rho, vo_fab **mac_rhs, const double *dx, double mu, const vo_bc *bc,
                        const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st);
void vo_ml_diff_scalar_solve(int nlev, vo_fab **snew, vo_fab **laps, const double *dx, double mu, const vo_bc *bc, const int pmask[3], const int *pd,
                             const vdn_params *prm, int icomp, int bccomp, vo_mgstat *st);
void vo_ml_diff_scalar_solve_g(int nlev, const vo_level *const *lev, vo_fab **snew, vo_fab **laps, const double *dx, double mu, const vo_bc *bc, const int pmask[3], const int *pd,
                               const vdn_params *prm, int icomp, int bccomp, vo_mgstat *st);
void vo_ml_hgproject(int nlev, int proj_type, vo_fab **unew, vo_fab **uold, vo_fab **rhohalf, vo_fab **p, vo_fab **gp, const double *dx, double dt,
                     const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st);
void vo_ml_hgproject_g(int nlev, const vo_level *const *lev, int proj_type, vo_fab **unew, vo_fab **uold, vo_fab **rhohalf, vo_fab **p, vo_fab **gp, const double *dx, double dt,
                       const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm, vo_mgstat *st);

void vo_ml_advance_timestep(int nlev, vo_state *S, const double *dx, double dt, const vo_bc *bc, const int pmask[3], const int *pd, const vdn_params *prm,
                            int proj_type, vo_mgstat st[2]);
void vo_ml_advance_timestep_g(int nlev, const vo_level *const *lev, vo_state *S, const double *dx, double dt, const vo_bc *bc, const int pmask[3], const int *pd,
                              const vdn_params *prm, int proj_type, vo_mgstat st[2]);
/* estdt (estdt.f90:15-78) of one level of a hierarchy: the minimum over its boxes */
double vo_estdt_g(const vo_level *L, const vo_fab *u, const vo_fab *s, const vo_fab *gp, const vo_fab *ext, const double dx[3], double dtold, const vdn_params *prm);

/* ---- the 2-D path (oracle/vo_2d.c): velpred_2d, mkflux_2d, update_2d, mkforce 2-D, estdt_2d, macproject / hgproject
 *      2-D kernels, our 5-point cell-centred and 9-point nodal multigrids, advance_timestep with dm = 2 ------------ */
#define V2(f, i, j, c) VF(f, i, j, 0, c)
void vo2_velpred(const vo_fab *u, vo_fab *umac[2], const vo_fab *force, const double dx[2], double dt, const vo_bc *bc, const vdn_params *prm);
void vo2_mkflux(const vo_fab *s, vo_fab *sedge[2], vo_fab *flux[2], vo_fab *umac[2], const vo_fab *force, const vo_fab *mac_rhs,
                const double dx[2], double dt, int is_vel, const int *is_cons, int bccomp, const vo_bc *bc, const vdn_params *prm);
void vo2_update(const vo_fab *sold, vo_fab *umac[2], vo_fab *sedge[2], vo_fab *flux[2], const vo_fab *force, vo_fab *snew,
                const double dx[2], double dt, int is_vel, const int *is_cons);
void vo2_mkvelforce(vo_fab *vf, const vo_fab *ext, const vo_fab *gp, const vo_fab *s, const vo_fab *lapu, double visc_fac, const vdn_params *prm);
void vo2_mkscalforce(vo_fab *sf, const vo_fab *ext, const vo_fab *laps, double diff_fac, const vdn_params *prm);
double vo2_estdt(const vo_fab *u, const vo_fab *s, const vo_fab *gp, const vo_fab *ext, const double dx[2], double dtold, const vdn_params *prm);
int  vo2_cc_solve(const vo_fab *rh, vo_fab *phi, const vo_fab *alpha, vo_fab *beta[2], const double dx[2], const int ellbc[3][2],
                  double rel_eps, double abs_eps, int max_iter, int nu1, int nu2, int nub, vo_mgstat *st);
void vo2_macproject(vo_fab *umac[2], vo_fab *rho, const vo_fab *mac_rhs, const double dx[2], const vo_bc *bc, const int pmask[3],
                    const vdn_params *prm, vo_mgstat *st);
int  vo2_nd_solve(vo_fab *rh, vo_fab *phi, const vo_fab *coeffs, const vo_fab *u, const double dx[2], const int ellbc[3][2], const int pmask[3],
                  double rel_eps, double abs_eps, int max_iter, int nu1, int nu2, int nub, double omega, vo_mgstat *st);
void vo2_hgproject(int proj_type, vo_fab *unew, const vo_fab *uold, vo_fab *rhohalf, vo_fab *p, vo_fab *gp, const double dx[2], double dt,
                   const vo_bc *bc, const int pmask[3], const vdn_params *prm, vo_mgstat *st);
void vo2_explicit_diffusive_term(vo_fab *lap, const vo_fab *data, int comp, int bccomp, const double dx[2], const vo_bc *bc);
void vo2_advance_timestep(vo_state *S, const double dx[2], double dt, const vo_bc *bc, const int pmask[3], const vdn_params *prm,
                          int proj_type, vo_mgstat st[2], double phase_sec[4]);
void vo2_initdata(vo_fab *u, vo_fab *s, const double dx[2], int prob_type);


/* regridding (vo_amr.c): tag_boxes (src/tag_boxes.f90:128-216), fillpatch and ml_nodal_prolongation as src/regrid.f90:311-327 calls them */
int  vo_tag_boxes(const vo_fab *s, int lev, int prob_type, unsigned char *tags);
void vo_fillpatch(vo_fab *fine, const vo_fab *crse, int icomp, int nc);
void vo_nodal_prolongation(vo_fab *fine, const vo_fab *crse);

/* the nodal 27-point operator on gathered values (see vo_hgproject.c) */
void vo_nd_stencil(const double f[3], const double p[3][3][3], const double sg[2][2][2], double *Kp, double *diag);

#ifdef __cplusplus
}
#endif
#endif
