/*
 * varden_amd.h -- C-ABI of the MI355X-native VARDEN hot path.
 *
 * This is the drop-in boundary for the reference's per-timestep path
 *     advance_module::advance_timestep      (reference src/advance_timestep.f90:26-44)
 *     estdt_module::estdt                   (reference src/estdt.f90:15)
 *     hgproject_module::hgproject           (reference src/hgproject.f90:17-18, called by the
 *                                            driver for the initial projection, varden.f90:134)
 * and for the BoxLib (FBoxLib, external to the reference tree) containers those
 * routines take: box / ml_layout / multifab / bc_tower.  A Fortran ISO_C_BINDING
 * module that re-exports these entry points under the BoxLib names lives in
 * varden_amd/fortran/varden_amd_mod.f90; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available from vdn_last_error() (reference: bl_error aborts; the Fortran
 *     shim turns non-zero into `error stop`).  A HIP error the CALLER left pending on the
 *     calling thread is cleared at entry (noted once per process on stderr) and not restored
 *     -- HIP offers no way to put it back --: every entry point then reports any launch
 *     failure of its own, whatever its code.  A host that polls hipGetLastError() after
 *     calling in asks vdn_last_stale_hip_error() for what was taken off its thread.
 *   - plain pointers and sizes only.  "device" pointers are hipMalloc'ed HBM.
 *   - all floating point data is IEEE f64; arrays use the BoxLib fab layout:
 *         p(lo1-ng:hi1+ng[+nodal1], lo2-ng:..., lo3-ng:..., 1:nc)   column-major,
 *     x fastest, component slowest (reference evidence: src/mkflux.f90:74-92).
 *   - indices (lo/hi) are 0-based cell indices of the level's index space, as in
 *     BoxLib; direction index d = 0,1,2 ; side 0 = lo, 1 = hi.
 */
#ifndef VARDEN_AMD_H
#define VARDEN_AMD_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bc_module constants (FBoxLib values; reference use: src/define_bc_tower.f90:199-335,
 *      exec/test/inputs_*: -1 periodic, 11 inlet, 12 outlet, 14 slip, 15 no-slip) -------- */
enum {
  VDN_PERIODIC = -1, VDN_INTERIOR = 0, VDN_INLET = 11, VDN_OUTLET = 12, VDN_SYMMETRY = 13,
  VDN_SLIP_WALL = 14, VDN_NO_SLIP_WALL = 15,
  VDN_REFLECT_ODD = 20, VDN_REFLECT_EVEN = 21, VDN_FOEXTRAP = 22, VDN_EXT_DIR = 23, VDN_HOEXTRAP = 24,
  VDN_BC_PER = -1, VDN_BC_INT = 0, VDN_BC_DIR = 1, VDN_BC_NEU = 2
};

/* ---- proj_parameters (reference src/proj_parameters.f90:5-8) ------------------------------ */
enum { VDN_INITIAL_PROJECTION = 1, VDN_DIVU_ITERS = 2, VDN_PRESSURE_ITERS = 3, VDN_REGULAR_TIMESTEP = 4 };

/* ---- runtime parameters read implicitly by the reference kernels through probin_module
 *      (reference src/_parameters, src/probin.template).  One POD passed once. ------------- */
typedef struct vdn_params {
  int    dm;              /* dim_in: 3, or 2 (one level, one box: BASELINE configs[0])         */
  int    nscal;           /* nscal (2)                                                        */
  int    slope_order;     /* 0, 2 or 4 (default 4)                                            */
  int    use_minion;      /* logical use_minion (default 0)                                   */
  int    boussinesq;      /* default 0                                                        */
  int    stencil_order;   /* 2                                                                */
  int    diffusion_type;  /* 1 = Crank-Nicolson, 2 = backward Euler                           */
  int    verbose;
  int    mg_verbose;
  int    prob_type;       /* only used for the hgproject abs-eps special case (prob_type 4)   */
  double visc_coef;       /* >= 0; > 0 enables the viscous solve (viscsolve.f90:19-306)          */
  double diff_coef;       /* >= 0; > 0 enables diff_scalar_solve (viscsolve.f90:308-515)        */
  double cflfac;          /* 0.8                                                              */
  double max_dt_growth;   /* 1.1                                                              */
  /* inflow data used by multifab_physbc EXT_DIR fills: [dir][side]                           */
  double u_bc[3][2], v_bc[3][2], w_bc[3][2], rho_bc[3][2], trac_bc[3][2];
  /* multigrid controls (the solvers are external to the reference tree; these are ours)      */
  int    mg_nu1, mg_nu2;          /* pre/post smoothing sweeps (2,2)                          */
  int    mg_nub;                  /* bottom sweeps on the coarsest level                      */
  int    mg_max_iter;             /* max V-cycles for the MAC solve (100)                     */
  int    hg_max_iter;             /* max V-cycles for the nodal solve (reference: 100)        */
  int    hg_nu1, hg_nu2, hg_nub;
  double hg_omega;                /* nodal Jacobi damping                                     */
  double mac_rel_eps;             /* 1e-10: reference src/macproject.f90:91-93                */
  double hg_rel_eps;              /* <=0: use 1e-12/1e-11/1e-10 by nlevs, hgproject.f90:113-119 */
  int    abort_on_max_iter;       /* 1 (default): a solve that reaches its iteration cap or meets a non-finite norm fails the call,
                                   * as FBoxLib's solvers abort (bl_error); 0: report through vdn_last_solver_stats and go on */
  int    hg_fmg;                  /* 1 (default): a nodal solve that starts from phi = 0 takes its initial guess from a nested iteration
                                   * (right-hand side restricted down, one V-cycle per level on the way up): two V-cycles fewer at 1e-12;
                                   * 0: V-cycles from the zero guess (rounds 1 and 2)                                              */
  int    mac_fmg;                 /* 1 (default): the same for the cell-centred solve of the MAC projection (whose phi starts from zero):
                                   * right-hand side averaged down to the coarsest level of at least 16^3 cells, two V-cycles there, then
                                   * per level a linear interpolation of the solution and one V-cycle; one V-cycle fewer at 1e-10; 0: off */
  double hg_omega_pre1, hg_omega_pre2;   /* 1.45, 0.7 (adjacent in memory, in this order): with hg_nu1 = 2 the two pre-smoothing sweeps of the nodal V-cycle
                                   * are damped by these instead of hg_omega -- a two-step Chebyshev pair, one V-cycle fewer at 256^3; either <= 0: hg_omega */
  double hg_omega_fac1, hg_omega_fac2, hg_omega_fac3;   /* 1.6, 0.9, 0.65: with hg_nu1 + hg_nu2 = 3 the three relaxation sweeps on a refined level of the
                                   * composite nodal solve (nlevs > 1) are damped by these -- a three-step Chebyshev set; any <= 0: hg_omega */
  int    mg_predict;              /* 1 (default): a projection's multigrid (one level, zero initial guess) does not read its residual norm back before the
                                   * cycle at which the previous solve of the same size stopped, minus one: the norms of the cycles in between are kept on the
                                   * device and read in one go (one read-back -- at N > 1 one all-reduce -- instead of one per V-cycle).  The stopping cycle is
                                   * decided by the same norms: if the history shows an earlier cycle had already converged, the solve is REPEATED with a
                                   * read-back per cycle, so results never depend on the prediction; 0: read every cycle (rounds 1-3) */
} vdn_params;

/* fills *p with the reference defaults (src/_parameters) */
void vdn_params_default(vdn_params *p);

typedef struct vdn_box { int lo[3]; int hi[3]; } vdn_box;

typedef struct vdn_layout   vdn_layout;    /* ml_layout (+ per-level layout, box->rank map)    */
typedef struct vdn_multifab vdn_multifab;  /* multifab of one level                            */
typedef struct vdn_bc_tower vdn_bc_tower;  /* define_bc_module::bc_tower                       */

/* ------------------------------------------------------------------------------------------- */
/* runtime                                                                                     */
/* ------------------------------------------------------------------------------------------- */
int  vdn_init(const vdn_params *prm, int rank, int nranks, int device);
/* Debug / measurement switches.  The library reads environment variables VDN_* only through one table (varden_amd/csrc/runtime.hip, g_switches): each
 * selects between launch forms that the test suite holds bit-for-bit equal, or is a probe; none changes a result, none is needed in production.  This call
 * returns the table as text -- one line per switch: name, current value, what it does.  vdn_init warns on stderr about VDN_* variables that are not in it.
 * Groups: transport rehearsal (VDN_FORCE_PACKED, VDN_RCCL_LIB + VDN_TESTING); runtime (VDN_ARENA_POISON, VDN_POLL, VDN_NO_GRAPHS,
 * VDN_NO_ROCTX, VDN_KEEP_SETS, VDN_KEPT_BOUND); advance (VDN_NO_SLOPE_CACHE, VDN_NO_FORCE_REUSE); Godunov launch forms (VDN_GOD_*, VDN_GODUNOV_*, VDN_SLOPES_MARCH,
 * VDN_FUSED_KCHUNKS); cell-centred multigrid (VDN_GSRB_PAIR, VDN_CC_HALO_FACES, VDN_MG_*, VDN_MAC_*, VDN_OVERLAP*); nodal
 * multigrid (VDN_ND_*, VDN_HG_FAST); composite solves (VDN_NDF_*, VDN_NDM_*, VDN_MLCC_*, VDN_FB_FACES); box-batched kernels (VDN_BATCH_*). */
const char *vdn_debug_switches(void);
/* "release": libvarden_amd.so -- reads NO environment variable (the switches are compiled out, RCCL is the only transport); "testing": libvarden_amd_testing.so,
 * the same objects with the switch table and the test-transport seam compiled in (the test suite, the A/B tools, the one-GPU transport rehearsal) */
const char *vdn_build_flavour(void);
int  vdn_finalize(void);
const char *vdn_last_error(void);
/* the hipError_t (as int; 0 = none) most recently found pending at the entry of a call and cleared there (see Conventions); clear != 0 resets it */
int  vdn_last_stale_hip_error(int clear);
/* Launch stream.  Default: a PRIVATE non-blocking stream created by vdn_init -- not ordered against the legacy null stream or any
 * stream of the host application.  Calls that only enqueue work (setval, copy_c, fill_boundary, physbc, the vdn_k_* hooks) return
 * before it has run; vdn_multifab_dataptr drains the private stream before handing a device pointer out, vdn_device_synchronize
 * drains it on request.  vdn_set_stream(s) makes every launch go to the caller's stream s instead (the caller then orders its own work
 * on s; nothing is drained for it); vdn_set_stream(NULL) returns to the private stream. */
int  vdn_set_stream(void *hip_stream);
int  vdn_device_synchronize(void);
/* arena of per-step temporaries (the multifabs advance_timestep.f90:65-80 allocates and frees every step): bytes backed by device memory (1 GB chunks mapped
 * into a reserved address range as the high-water mark moves), high-water mark */
int  vdn_arena_stats(size_t *reserved_bytes, size_t *peak_bytes);
/* A 2-D problem run as its z-uniform, z-periodic 3-D copy (the hierarchies of the reference's 2-D inputs: varden_amd/driver.py, VardenAMR(extrude2d = nz)): with
 * w = 0 and nothing varying along z the 3-D scheme IS the 2-D one -- velpred_3d / mkflux_3d reduce to velpred_2d / mkflux_2d, the 7-point and 27-point operators to
 * the 5-point and 9-point ones -- except where the reference's two restatements differ: velpred_3d clamps the normal velocity of a hi-x OUTLET face with min()
 * (src/velpred.f90:2075), velpred_2d with max() (:305).  on != 0 selects velpred_2d's rule in the 3-D kernels.  Default 0; reset by vdn_init. */
int  vdn_set_extruded_2d(int on);
int  vdn_get_params(vdn_params *out);

/* ------------------------------------------------------------------------------------------- */
/* rank-to-rank transport (replaces FBoxLib's `parallel` MPI wrapper on the hot path): one rank  */
/* per GPU, RCCL point-to-point for ghost exchange, ncclAllReduce(MAX) for norms / estdt.          */
/* Rank 0 calls vdn_comm_get_unique_id, the host broadcasts the 128 bytes (MPI_Bcast in a Fortran   */
/* driver, torch.distributed in bench.py), every rank calls vdn_comm_init.  No-ops when nranks = 1. */
/* ------------------------------------------------------------------------------------------- */
int  vdn_comm_get_unique_id(char *id128);
int  vdn_comm_init(const char *id128);
int  vdn_comm_finalize(void);
/* ranks of the live RCCL communicator (ncclCommCount); 1 when none is up -- bench.py reports it as rccl_nranks */
int  vdn_comm_nranks(int *n);
/* MAX over the ranks of n host doubles, in place (parallel_reduce(..., MPI_MAX) of a Fortran driver; the file writers use it for the
 * per-box minima / maxima and as their barrier).  No-op when nranks = 1. */
int  vdn_comm_allreduce_max(double *host, int n);
/* traffic counters since the last reset (24 longs): [0] ghost exchanges with remote traffic (pack + one ncclGroup + unpack each),
 * [1] ncclSend calls, [2] doubles sent, [3] all-reduces, [4] all-gathers, [5] doubles contributed to them, [6] refreshes of
 * inter-level views, [7] doubles they sent, [8+b] exchanges that sent [2^(10+b), 2^(11+b)) bytes.  The exchange sites of the reference:
 * src/velpred.f90:102-119, src/macproject.f90:117,492, src/estdt.f90:69; everything inside ml_cc_solve / ml_nd_solve is FBoxLib's. */
int  vdn_comm_stats(long *out24, int reset);
/* "rccl", "test-double" (tests/fake_rccl, only with VDN_TESTING=1 -- see exchange.hip) or "none" (no transport loaded: one rank) */
const char *vdn_comm_transport(void);
/* host-only introspection of the ghost-exchange plan (used by the CPU multi-process tests): the remote  */
/* copies rank `as_rank` performs for a multifab (nc, ng, nodal) on the given boxes; rows of 14 longs:  */
/* [kind 0=send 1=recv, peer, lo[3], hi[3], shift[3], buffer offset (doubles), dst box, src box]          */
int  vdn_plan_describe(const vdn_box *pd, const int *pmask, int nboxes, const vdn_box *boxes, const int *owner,
                       int nc, int ng, const int *nodal, int as_rank, long *rows, int maxrows,
                       int *nrows, int *nlocal_descs);
/* host-only introspection of the box index behind the box-pair loops of the inter-level operators (CPU tests): the boxes of the list that the  */
/* index names as candidates for touching [qlo - margin, qhi + margin], ascending; *ncand their number (out holds the first maxout)            */
int  vdn_box_candidates(int nboxes, const vdn_box *boxes, const int *qlo, const int *qhi, int margin, int *out, int maxout, int *ncand);

/* ------------------------------------------------------------------------------------------- */
/* ml_layout: nlev levels; rr[nlev-1][3] refinement ratios; pd[nlev] problem domains;          */
/* boxes of all levels concatenated (nboxes[l] each); owner[] = rank of each box;              */
/* pmask[3] periodicity.  (BoxLib: ml_layout_build / layout_build_ba)                          */
/* ------------------------------------------------------------------------------------------- */
int  vdn_layout_create(int nlev, const int *rr, const vdn_box *pd, const int *nboxes,
                       const vdn_box *boxes, const int *owner, const int *pmask, vdn_layout **out);
int  vdn_layout_destroy(vdn_layout *la);
int  vdn_layout_nlevel(const vdn_layout *la);
int  vdn_layout_nboxes(const vdn_layout *la, int lev);          /* global number of boxes     */
int  vdn_layout_nlocal(const vdn_layout *la, int lev);          /* boxes owned by this rank    */
int  vdn_layout_global_index(const vdn_layout *la, int lev, int local_i);
int  vdn_layout_get_box(const vdn_layout *la, int lev, int global_i, vdn_box *out);

/* ------------------------------------------------------------------------------------------- */
/* bc_tower: built from the domain phys_bc[dir][side] exactly as define_bc_tower.f90 does       */
/* (phys 129-156, adv 158-252, ell 254-340).  Arrays are addressed [grid][dir][side][comp]      */
/* with grid 0 = whole domain, grids 1..nlocal = local boxes, comp 0-based.                     */
/* ------------------------------------------------------------------------------------------- */
int  vdn_bc_tower_create(const vdn_layout *la, const int *phys_bc /*[3][2]*/, vdn_bc_tower **out);
int  vdn_bc_tower_destroy(vdn_bc_tower *bct);
int  vdn_bc_tower_phys(const vdn_bc_tower *b, int lev, int grid, int dir, int side);
int  vdn_bc_tower_adv (const vdn_bc_tower *b, int lev, int grid, int dir, int side, int comp);
int  vdn_bc_tower_ell (const vdn_bc_tower *b, int lev, int grid, int dir, int side, int comp);

/* ------------------------------------------------------------------------------------------- */
/* multifab (BoxLib multifab_module names in comments)                                          */
/* ------------------------------------------------------------------------------------------- */
int  vdn_multifab_create(const vdn_layout *la, int lev, int nc, int ng, const int *nodal /*[3] or NULL*/,
                         vdn_multifab **out);                                   /* multifab_build[_edge] */
int  vdn_multifab_destroy(vdn_multifab *mf);                                    /* multifab_destroy      */
int  vdn_multifab_nfabs(const vdn_multifab *mf);                                /* nfabs                 */
int  vdn_multifab_ncomp(const vdn_multifab *mf);
int  vdn_multifab_nghost(const vdn_multifab *mf);
int  vdn_multifab_get_box(const vdn_multifab *mf, int local_i, vdn_box *out);   /* get_box (valid cells) */
long vdn_multifab_fab_size(const vdn_multifab *mf, int local_i);                /* doubles in the fab    */
int  vdn_multifab_dataptr(const vdn_multifab *mf, int local_i, double **dev);   /* dataptr (DEVICE ptr)  */
int  vdn_multifab_copy_to_host(const vdn_multifab *mf, int local_i, double *host);
int  vdn_multifab_copy_from_host(vdn_multifab *mf, int local_i, const double *host);
int  vdn_multifab_setval(vdn_multifab *mf, double val, int comp, int nc, int all);     /* setval       */
int  vdn_multifab_copy_c(vdn_multifab *dst, int dcomp, const vdn_multifab *src, int scomp,
                         int nc, int ng);                                               /* copy_c       */
int  vdn_multifab_norm_inf(const vdn_multifab *mf, int comp, int nc, double *out);      /* norm_inf     */
int  vdn_multifab_min_max(const vdn_multifab *mf, int comp, double *mn, double *mx);    /* min_c/max_c  */
int  vdn_multifab_fill_boundary(vdn_multifab *mf);                       /* multifab_fill_boundary    */
int  vdn_multifab_physbc(vdn_multifab *mf, int scomp, int bccomp, int nc,
                         const vdn_bc_tower *bct);                        /* multifab_physbc.f90:17    */

/* ------------------------------------------------------------------------------------------- */
/* the hot path.  Arrays of per-level multifab handles (length nlevel).                         */
/* dx is [nlevel][3].  press_comp is the 1-based bc component (dm+nscal+1) as in the reference. */
/* ------------------------------------------------------------------------------------------- */
int  vdn_advance_timestep(int istep, vdn_layout *mla,
                          vdn_multifab **sold, vdn_multifab **uold,
                          vdn_multifab **snew, vdn_multifab **unew,
                          vdn_multifab **gp,   vdn_multifab **p,
                          vdn_multifab **ext_vel_force, vdn_multifab **ext_scal_force,
                          const vdn_bc_tower *the_bc_tower,
                          double dt, double time, const double *dx,
                          int press_comp, int proj_type);
/* estdt(lev,u,s,gp,ext_vel_force,dx,dtold,dt)  (reference src/estdt.f90:15-87) */
int  vdn_estdt(int lev, const vdn_multifab *u, const vdn_multifab *s, const vdn_multifab *gp,
               const vdn_multifab *ext_vel_force, const double *dx /*[3]*/, double dtold, double *dt);
/* hgproject(proj_type,mla,unew,uold,rhohalf,p,gp,dx,dt,the_bc_tower,press_comp)
 * (reference src/hgproject.f90:17) */
int  vdn_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold,
                   vdn_multifab **rhohalf, vdn_multifab **p, vdn_multifab **gp,
                   const double *dx, double dt, const vdn_bc_tower *bct, int press_comp);
/* macproject(mla,umac,rho,mac_rhs,dx,the_bc_tower,bc_comp)  (reference src/macproject.f90:20);
 * umac is [nlevel][3] */
int  vdn_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs,
                    const double *dx, const vdn_bc_tower *bct, int bc_comp);

/* ---- multi-level operators (FBoxLib; up to 4 levels, refinement ratio 2, boxes on any rank) ---------------------------------
 * ml_cc_restriction(crse, fine, rr)            reference call sites src/macproject.f90:204-206, src/hgproject.f90:355-357
 * ml_edge_restriction(crse, fine, rr, dir)     src/velpred.f90:115-119, src/macproject.f90:330-333, 497-500
 * multifab_fill_ghost_cells(fine, crse, ...)   src/macproject.f90:304-310 (and inside ml_restrict_and_fill)
 * create_umac_grown(fine, crse, ...)           src/velpred.f90:102-107, src/macproject.f90:107-113
 * ml_restrict_and_fill(nlevs, mf, rr, bc, icomp, bcomp, nc, same_boundary)   src/update.f90:103-107, src/mkforce.f90:75-76, ...
 * components 0-based.  vdn_macproject / vdn_hgproject / vdn_advance_timestep accept nlevel = 2..4 (composite solves); umac is then [lev*3 + dir]. */
int vdn_ml_cc_restriction(vdn_multifab *crse, const vdn_multifab *fine, int icomp, int nc);
int vdn_ml_edge_restriction(vdn_multifab *crse, const vdn_multifab *fine, int dir);
int vdn_multifab_fill_ghost_cells(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc);
int vdn_create_umac_grown(vdn_multifab *fine, const vdn_multifab *crse, int dir);
int vdn_ml_restrict_and_fill(int nlev, vdn_multifab **mf, int icomp, int bcomp, int nc, int same_boundary, const vdn_bc_tower *bct);

/* regridding (src/regrid.f90:280-337, build_and_fill_data): fillpatch(fine, crse, ng = 0, ...) fills every valid cell of a new fine
 * level from the coarser one (coarse ghost cells filled; interpolation of multifab_fill_ghost_cells); ml_nodal_prolongation does the
 * same for the nodal pressure (trilinear); multifab_copy_c between multifabs of one level whose box lists differ copies the points
 * valid in both (old data over the interpolated data). */
int vdn_fillpatch(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc);
int vdn_ml_nodal_prolongation(vdn_multifab *fine, vdn_multifab *crse);
int vdn_multifab_copy_layouts(vdn_multifab *dst, int dcomp, const vdn_multifab *src, int scomp, int nc);

/* ---- grid generation in front of the AMR path ------------------------------------------------------------------------------------
 * tag_boxes(mf, tagboxes, dx, lev)                               src/tag_boxes.f90:17-48 (rules :142-210: rho > 1.01 / 1.1 / 1.5 by level
 *                                                                for prob_type 1, 2; 1.2 < rho < 1.8 for prob_type 3)
 * make_new_grids(new_grid, la_crse, la_fine, mf, dx, buf_wid, ref_ratio, lev, max_grid_size)   [FBoxLib]   src/initialize.f90:247-248,
 *                                                                src/regrid.f90:148-149; cluster_* parameters src/_parameters:37-39
 * s: the state of level `lev1` (1-based, as in tag_boxes) -- component 0 is tagged; the boxes of level lev1+1 are returned in that
 * level's index space (*nboxes_out = 0: no cell tagged, "new_grid = .false.").  nest: cells of level lev1 kept between the new
 * level and the edge of level lev1 (proper nesting).  Collective: every rank passes its part of the level and gets the same boxes. */
/* tag_boxes(tagboxes, mf, dx, lev) of src/tag_boxes.f90:17-216 alone: one byte per cell of the level's domain (x fastest), 1 = tagged;
 * lev1 is the 1-based level the thresholds are chosen by (tag_boxes.f90:65-94, 142-178) */
int vdn_tag_boxes(const vdn_multifab *s, int lev1, unsigned char *tags_host);
int vdn_make_new_grids(const vdn_multifab *s, int lev1, int buf_wid, int nest, double min_eff, int min_width, int blocking,
                       int max_grid_size, int maxboxes, vdn_box *boxes_out, int *nboxes_out, long *ntagged);

/* ---- derived plot quantities of write_plotfile (src/varden.f90:532-540) ------------------------------------------------------------
 * make_vorticity(vort, comp, u, dx, bc)   src/makevort.f90:16-57  (fills the ghost cells of u first, as the reference does;
 *                                         3-D: |curl u| with one-sided differences next to inflow / no-slip faces, 2-D: v_x - u_y)
 * make_magvel(magvel, comp, u)            src/makevort.f90:59-91
 * comp 0-based. */
int vdn_make_vorticity(vdn_multifab *vort, int comp, vdn_multifab *u, const double *dx /*[dm]*/, const vdn_bc_tower *bct);
int vdn_make_magvel(vdn_multifab *magvel, int comp, vdn_multifab *u);

/* per-phase wall seconds of the last vdn_advance_timestep (reference prints them,
 * advance_timestep.f90:159-166): [0]=scalar [1]=velocity [2]=MAC [3]=HG [4]=total              */
int  vdn_last_step_timing(double *sec5);
/* diagnostics of the last MAC / HG solves: cycles, initial and final residual norms             */
int  vdn_last_solver_stats(int which /*0=MAC,1=HG*/, int *cycles, double *res0, double *res);
/* how the last macproject solve on one box kept its finest level: 0 interleaved (the level array), 1 by colour (passes and residual on the
 * split arrays), 2 by colour for the passes only (VDN_MAC_SPLIT=2).  No reference counterpart: the tests use it to know which kernels they exercised. */
int  vdn_last_mac_level_form(void);

/* ------------------------------------------------------------------------------------------- */
/* unit-test hooks: one per reference kernel, single-level, all local boxes.                    */
/* ------------------------------------------------------------------------------------------- */
/* slope_module (src/slope.f90): slopes of comps [0,nc) in direction dir on [lo-1,hi+1]^3;
 * slope multifab must have ng=1, nc comps; bccomp = 0-based first adv_bc component             */
int  vdn_k_slope(const vdn_multifab *s, vdn_multifab *slope, int dir, int bccomp, const vdn_bc_tower *bct);
/* velpred (src/velpred.f90:16): u(3,ng3), force(3,ng1) -> umac[3] (face, ng1), incl. fill_boundary */
int  vdn_k_velpred(const vdn_multifab *u, vdn_multifab **umac, const vdn_multifab *force,
                   const double *dx, double dt, const vdn_bc_tower *bct);
/* mkflux (src/mkflux.f90:16) */
int  vdn_k_mkflux(const vdn_multifab *s, vdn_multifab **sedge, vdn_multifab **flux,
                  vdn_multifab **umac, const vdn_multifab *force, const vdn_multifab *mac_rhs,
                  const double *dx, double dt, const vdn_bc_tower *bct, int is_vel, const int *is_cons);
/* update (src/update.f90:16), incl. the ghost fill of snew */
int  vdn_k_update(const vdn_multifab *sold, vdn_multifab **umac, vdn_multifab **sedge, vdn_multifab **flux,
                  const vdn_multifab *force, vdn_multifab *snew, const double *dx, double dt,
                  int is_vel, const int *is_cons, const vdn_bc_tower *bct);
/* mkvelforce / mkscalforce (src/mkforce.f90:18,238), incl. ghost fill */
int  vdn_k_mkvelforce(vdn_multifab *vel_force, const vdn_multifab *ext_vel_force, const vdn_multifab *s,
                      const vdn_multifab *gp, const vdn_multifab *lapu /*may be NULL*/, double visc_fac,
                      const vdn_bc_tower *bct);
int  vdn_k_mkscalforce(vdn_multifab *scal_force, const vdn_multifab *ext_scal_force,
                       const vdn_multifab *laps /*may be NULL*/, double diff_fac, const vdn_bc_tower *bct);
/* make_at_halftime (src/make_at_halftime.f90:18) */
int  vdn_k_make_at_halftime(vdn_multifab *rhohalf, const vdn_multifab *sold, const vdn_multifab *snew,
                            int in_comp, int out_comp, const vdn_bc_tower *bct);

/* cell-centred multigrid: solves (-div beta grad) phi = rh with the boundary types
 * bc[dir][side] in {VDN_BC_NEU, VDN_BC_DIR, VDN_BC_PER}; replaces ml_cc_solve as called from
 * reference src/mac_multigrid.f90:53-62 (alpha = 0).  phi has ng=1, rh ng=0, beta[3] faces.      */
int  vdn_cc_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx,
                  const int *bc /*[3][2]*/, double rel_eps, double abs_eps, int max_iter,
                  int *cycles, double *res0, double *res);
/* one red-black Gauss-Seidel sweep pair (nsweeps times) of that operator on the finest level --
 * the kernel the smoother roofline is quoted on                                                   */
int  vdn_cc_smooth(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx,
                   const int *bc /*[3][2]*/, int nsweeps);
/* nodal multigrid: solves div(sigma grad phi) = div(u) (+rh in) with sigma = coeffs (cell, ng=1,
 * zero outside the domain), dense (Q1) stencil; replaces ml_nd_solve as called from reference
 * src/hg_multigrid.f90:95-105 (add_divu=.true., u=unew).  phi, rh nodal ng=1.                     */
int  vdn_nd_solve(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *coeffs, const vdn_multifab *u,
                  const double *dx, const int *bc /*[3][2]*/, double rel_eps, double abs_eps, int max_iter,
                  int *cycles, double *res0, double *res);

/* ---- kernel timing (HIP events on the launch stream) for bench.py's roofline object ---------- */
/* times `nlaunch` back-to-back launches of ONE colour pass of the cc smoother on the finest level
 * and returns the average milliseconds per launch and the number of cells one launch covers.
 * rho (may be NULL): the density behind beta = 2/(rho_i + rho_i-1); given, the pass is the one macproject runs (face coefficients
 * recomputed from rho), otherwise the stored-coefficient pass of the viscous solves and the coarser levels       */
int  vdn_bench_cc_smoother(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const vdn_multifab *rho, const double *dx,
                           const int *bc, int nlaunch, double *avg_ms, long *cells);
/* the same passes launched as they are INSIDE a solve: `nsweeps` red-black sweeps time-skewed over plane slabs (the level stored by colour, one box; a slab is served
 * from the Infinity Cache from its second pass on), about `nlaunch` passes in all; avg_ms = time per pass over the whole level, cells = 0 when the level has no slab
 * schedule (smaller than 2^23 cells, several boxes, periodic faces).  bench.py: roofline.in_solve_ms_per_level_pass                                              */
int  vdn_bench_cc_smoother_in_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const vdn_multifab *rho, const double *dx,
                                    const int *bc, int nsweeps, int nlaunch, double *avg_ms, long *cells);

#ifdef __cplusplus
}
#endif
#endif /* VARDEN_AMD_H */
