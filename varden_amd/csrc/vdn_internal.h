// vdn_internal.h -- host-side data model of the MI355X-native VARDEN hot path (private to csrc/).
//
// box / layout / multifab / bc_tower mirror the FBoxLib containers the reference's hot path takes
// (reference evidence of the layout: src/mkflux.f90:74-92, src/define_bc_tower.f90:9-33).
// State lives in HBM for the whole run; per-step temporaries come from a persistent arena so a step
// performs no hipMalloc/hipFree (the reference allocates ~25 multifabs per step,
// src/advance_timestep.f90:65-80,141-148).
#pragma once
#define VDN_MAXLEV 4          // deepest hierarchy the multi-level operators take (arrays sized [lev], [lev*3 + d])
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <map>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <stdexcept>
#include <cmath>
#include "../../include/varden_amd.h"

// ---- device-visible views ---------------------------------------------------------------------
// one fab: p(alo0:, alo1:, alo2:, 0:nc-1), x fastest.  alo = valid lo - ng.
struct FV {
  double *p;
  int a0, a1, a2;     // lowest allocated index per direction
  int n0, n1, n2;     // allocated extents
  long sc;            // component stride
};
// index-space description of one box and its physical bcs, passed by value to kernels
struct BoxP {
  int lo[3], hi[3];   // valid cell box
  int phys[3][2];     // phys_bc_level_array(i,:,:)  (INTERIOR on faces inside the domain)
};

#define VDN_MAXCOMP 16

// ---- host structures ---------------------------------------------------------------------------
struct vdn_layout {
  unsigned long uid = 0;                     // unique per created layout: cache keys must survive pointer reuse
  int nlev = 0;
  std::vector<int> rr;                       // [nlev-1][3]
  std::vector<vdn_box> pd;                   // [nlev]
  std::vector<std::vector<vdn_box>> boxes;   // [lev][global box]
  std::vector<std::vector<int>> owner;       // [lev][global box]
  std::vector<std::vector<int>> local;       // [lev][local i] -> global index
  int pmask[3] = {0, 0, 0};
};

struct vdn_multifab {
  const vdn_layout *la = nullptr;
  int lev = 0, nc = 1, ng = 0;
  int nodal[3] = {0, 0, 0};
  double *base = nullptr;                    // one allocation for all local fabs
  bool owns = true;                          // false: memory belongs to the arena
  size_t bytes = 0;
  std::vector<FV> fabs;                      // per local box
  std::vector<vdn_box> vbox;                 // valid cell box per local box
  int nfabs() const { return (int)fabs.size(); }
  long fab_size(int i) const { return fabs[i].sc * nc; }
};

struct vdn_bc_tower {
  const vdn_layout *la = nullptr;
  unsigned long serial = 0;                                  // unique per tower ever created (keys of kept descriptor sets)
  int dm = 3, nscal = 2;
  int ncomp_adv = 0, ncomp_ell = 0;
  // [lev][grid (0 = domain, 1.. = local boxes)]
  std::vector<std::vector<BoxP>> phys;                       // only .phys used
  std::vector<std::vector<std::vector<int>>> adv;            // [lev][grid][ (d*2+s)*ncomp_adv + c ]
  std::vector<std::vector<std::vector<int>>> ell;
  int domain_bc[3][2];
  int press_comp0() const { return dm + nscal; }             // 0-based
  int extrap_comp0() const { return dm + nscal + 1; }
  int adv_bc(int lev, int grid, int d, int s, int c) const { return adv[lev][grid][(d * 2 + s) * ncomp_adv + c]; }
  int ell_bc(int lev, int grid, int d, int s, int c) const { return ell[lev][grid][(d * 2 + s) * ncomp_ell + c]; }
};

// ---- global context ----------------------------------------------------------------------------
struct VdnCtx {
  bool inited = false;
  vdn_params prm;
  int rank = 0, nranks = 1, device = 0;
  bool extruded2d = false;    // vdn_set_extruded_2d: the 3-D kernels run a z-uniform, z-periodic copy of a 2-D problem (godunov.hip: velpred's hi-x OUTLET rule)
  hipStream_t stream = 0;                    // the launch stream: our own non-blocking stream unless vdn_set_stream names another
  hipStream_t own_stream = 0, halo_stream = 0;    // halo_stream: packed ghost traffic next to interior compute (exchange.hip)
  hipEvent_t ev_main = nullptr, ev_halo = nullptr; // ordering between the two (no timing)
  // persistent arena for per-step temporaries (bump allocator, reset at the start of each public call)
  char *arena = nullptr; size_t arena_bytes = 0, arena_off = 0, arena_peak = 0;
  // small device scratch for reductions + pinned host mirror
  double *d_scal = nullptr; double *h_scal = nullptr;       // 64 doubles each
  double *h_scal_dev = nullptr;                             // device view of the pinned mirror (k_publish writes it)
  double *d_hist = nullptr;                                 // 64 norms of consecutive V-cycles + (slot 64, as an integer) their count: norm_hist_*
  double step_sec[5] = {0, 0, 0, 0, 0};
  int solver_cycles[2] = {0, 0}; double solver_res0[2] = {0, 0}, solver_res[2] = {0, 0};
  // per local box of a level handled box by box (one box, or a few large ones): the slopes of uold computed by velpred are used again by the velocity
  // mkflux of the same advance_timestep, max |umac| of the scalar mkflux by the velocity mkflux (empty vectors: off)
  std::vector<double *> slope_cache[3]; std::vector<const double *> slope_src;
  std::vector<double *> macmax_cache; std::vector<const double *> macmax_src;
  void drop_step_caches() { for (int d = 0; d < 3; d++) slope_cache[d].clear(); slope_src.clear(); macmax_cache.clear(); macmax_src.clear(); }
};
VdnCtx &ctx();

// the one way the library reads its environment: getenv for a name declared in the switch table of runtime.hip (fails for any other name)
const char *vdn_env(const char *name);
// error handling: C-ABI functions wrap their body in VDN_TRY/VDN_CATCH
void vdn_set_error(const char *fmt, ...);
struct VdnErr : std::runtime_error { using std::runtime_error::runtime_error; };
[[noreturn]] void vdn_fail(const char *fmt, ...);
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) vdn_fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
// entry: an error the HOST application left pending on this thread (a failed launch of its own; torch's event queries leave hipErrorNotReady) is
// taken off the thread (HIP has no way to look past it) so that EVERY launch failure inside the call is seen on the way out, whatever its code
// (comparing codes hid a failure of ours that happened to carry the stale code).  A stale error other than hipErrorNotReady is reported
// once per process on stderr -- it is the host's, the call goes on -- and is NOT restored (HIP cannot put it back): the host reads the last one taken this
// way through vdn_last_stale_hip_error(); include/varden_amd.h says so under "Error convention", INTEGRATION.md section 4.
void vdn_note_stale_error(hipError_t e);       // runtime.hip: remembers it (vdn_last_stale_hip_error) and prints the note ONCE per process
#define VDN_TRY try { { const hipError_t entry_err_ = hipGetLastError(); \
    if (entry_err_ != hipSuccess && entry_err_ != hipErrorNotReady) vdn_note_stale_error(entry_err_); }
// the success path of every C-ABI call also asks HIP for a pending launch error (a kernel launch with a bad grid fails silently otherwise);
// hipErrorNotReady is the benign residue of our own hipStreamQuery / hipEventQuery polls
#define VDN_CATCH   if (ctx().inited) { const hipError_t le_ = hipGetLastError(); \
    if (le_ != hipSuccess && le_ != hipErrorNotReady) vdn_fail("a HIP launch failed inside this call: %s", hipGetErrorString(le_)); } \
  } catch (const std::exception &e) { vdn_set_error("%s", e.what()); return 1; } return 0;
// outcome of an elliptic solve: FBoxLib's solvers abort on max_iter (bl_error); so do we unless prm.abort_on_max_iter = 0.
// A non-finite norm (the reductions turn NaN into +inf) is a failure whatever rc says.  comp >= 0: the component being solved
void solver_check(int rc, const char *what, int iters, double res, double res0, int comp = -1);
#define REQUIRE(c, ...) do { if (!(c)) vdn_fail(__VA_ARGS__); } while (0)

// bl_prof_timer of the reference (build(bpt, "name") / destroy(bpt), e.g. src/advance_timestep.f90:60,99-101) as roctx ranges: rocprofv3
// --marker-trace shows the same names.  The roctx library is bound with dlopen at vdn_init; without it the ranges cost one branch.
struct Prof { explicit Prof(const char *name); ~Prof(); Prof(const Prof &) = delete; bool on; };
void prof_load();

// n (<= 64) device doubles -> the pinned host mirror, then a stream synchronisation: the one way scalars (norms, estdt maxima)
// come back.  A 64-thread kernel stores straight into the mapped host buffer: ~3 us instead of the ~20 us blit of an 8-byte
// hipMemcpyAsync (profiles/r01_bench_kernel_stats.csv: 1120 __amd_rocclr_copyBuffer calls = 4.9 % of a step)
const double *read_scalars(const double *dev, int n);
inline double read_scalar1(const double *dev) { return read_scalars(dev, 1)[0]; }
// Residual norms of consecutive V-cycles kept on the device (vdn_params.mg_predict): norm_hist_push appends *d_nrm (a one-thread kernel on the launch
// stream, capturable in a cycle's graph), norm_hist_read makes the first n entries global (ONE all-reduce when several ranks run) and reads them back.
void norm_hist_reset();
void norm_hist_push(const double *d_nrm);
const double *norm_hist_read(int n);
// The V-cycle count of the previous solve of one kind and size (0: none yet) -- a performance hint only, results never depend on it
int  mg_predict_get(int solver, const int n[3]);
void mg_predict_set(int solver, const int n[3], int cycles);
extern int g_mg_predict_off;          // > 0: inside the repeat of a solve whose prediction overshot

// ---- hipGraph replay of fixed launch sequences (one multigrid cycle = ~100 launches of 3-15 us) ---------------------------------------
// Usage:  GraphKey k; k.put(...every value the launches depend on...);  if (!graph_replay(k.h)) { graph_begin(); body(); graph_end(k.h); }
// graph_end instantiates, caches and launches.  graph_generation() changes whenever the cache is cleared.  Bodies may only enqueue work on ctx().stream (kernels, memsets, device copies).
// Off when the transport is active (RCCL calls and the test double are not captured) or VDN_NO_GRAPHS is set.
struct GraphKey {
  unsigned long long h = 1469598103934665603ull;
  void add(const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; } }
  template <class T> void put(const T &v) { add(&v, sizeof v); }
};
// ---- descriptor sets kept across calls -------------------------------------------------------------------------------------------------
// The inter-level operators and the composite solves describe their box-batched launches by descriptor arrays built on the host (pair loops over
// the boxes of two levels) and uploaded: 0.3-1 ms per call on a level of a thousand boxes, with the GPU idle meanwhile, a few dozen times per step.
// State fields live at fixed addresses and temporaries come back at the same arena offsets every step, so the arrays of one call site are the same
// bytes step after step: they are kept on the device under a key made of everything they depend on (GraphKey over the layout uid, levels, base
// pointers, components, ...) and dropped with the layout (xplan_cache_purge) or when the table outgrows its bound.  VDN_KEEP_SETS=0: rebuilt every call.
struct GraphKey;
struct KeptSet { void *d_args = nullptr; int *d_start = nullptr; int nbox = 0, tot = 0; unsigned long uid = 0; };
bool     kept_sets_enabled();
bool     kept_family_enabled(int fam);
void     dbg_sync(int bit);          // VDN_SYNC_POINTS (testing build): hipDeviceSynchronize at the points whose bit is set -- the search for a host / device race
KeptSet *kept_find(unsigned long long key);
KeptSet *kept_store(unsigned long long key, unsigned long uid, const void *args, size_t arg_bytes, const int *start, int nbox, int tot);
void     kept_purge(unsigned long uid);            // uid 0: every entry (plain sets AND the groups of the composite solves: never from inside a solve)
int      kept_bound(int dflt);                     // size bound of a kept table (VDN_KEPT_BOUND overrides)
// ... and whole groups of BatchSets (the composite solves): while a KeeperMem is open, BatchSet::build takes its device memory from it instead of the arena
struct KeeperMem { std::vector<void *> chunks; char *cur = nullptr; size_t left = 0; };
void  keeper_begin(KeeperMem *m);
void  keeper_end();
void  keeper_free(KeeperMem *m);
void *set_alloc(size_t bytes);                     // the open KeeperMem, or the arena
void  mlcc_kept_purge(unsigned long uid);          // amr.hip
void  mlnd_kept_purge(unsigned long uid);          // mg_nd.hip
bool graphs_enabled();
bool graph_replay(unsigned long long key);     // true: the cached graph was launched
void graph_begin();
void graph_end(unsigned long long key);
void graph_abort();                            // leave capture mode after an exception inside a body
void graph_cache_clear();
unsigned long graph_generation();      // bumped by every graph_cache_clear

// arena
void  arena_reset();
void  arena_reserve(size_t bytes);
void  arena_reserve_for(const vdn_layout *la);
void *arena_alloc(size_t bytes);
size_t arena_mark();
void  arena_release(size_t mark);
// temporary multifab living in the arena (freed by arena_release/reset); filled with val if fill
vdn_multifab *mf_temp(const vdn_layout *la, int lev, int nc, int ng, int face_dir /* -1 cell, 0..2 face, 3 nodal */,
                      bool fill, double val);
void mf_temp_free(vdn_multifab *mf);

// exchange.hip: ghost exchange plans and the RCCL transport
struct XBoxInfo { FV fv; int vlo[3], vhi[3]; int owner; };      // valid POINT range (incl. nodal points); fv only if local
struct XPlan;
XPlan *xplan_build(const std::vector<XBoxInfo> &boxes, const vdn_box &pd, const int pmask[3], int ng, int nc, bool faces_only = false, const int *src_trim = nullptr);   // src_trim[d] = 1: the sources give their points without their high plane along d
void   xplan_run(XPlan *P, hipStream_t st = nullptr);      // st: the stream the pack / transfer / copy / unpack run on (default: the launch stream)
bool   xplan_has_remote(const XPlan *P);                  // some of the traffic goes to another rank
void   xplan_free(XPlan *P);
unsigned long xplan_serial(const XPlan *P);
void   xplan_cache_purge(unsigned long layout_uid);   // drop every cached plan built for that layout
void   halo_cache_register(unsigned long layout_uid, XPlan *P);
std::vector<XBoxInfo> xboxes_of(const vdn_multifab *mf);
// ---- views of another level's (or another box list's) data, for the inter-level operators on several ranks -----------------------
// For every GLOBAL box j of `src`'s level: an FV that holds the part of box j (valid + ghost points) this rank's destination boxes
// read -- the local fab itself when this rank owns j, else a window received from j's owner.  vbox[j] is the TRUE valid box of j
// (the operators' parent-range filters need it); have[j] tells whether any data of j is present here.  The window buffers persist
// with the view (descriptor sets built from the FVs stay valid); refresh() re-sends the data.  With one rank a view is just the
// multifab's own fabs and refresh() does nothing.
struct ViewPlan;
struct SrcView {
  std::vector<char> have; std::vector<FV> fv; std::vector<vdn_box> vbox;
  int ng = 0, nc = 1, nodal[3] = {0, 0, 0};
  ViewPlan *plan = nullptr;
  int nboxes() const { return (int)vbox.size(); }
  void refresh() const;
};
// footprint[i]: for the GLOBAL destination box i (on the level / box list dst_owner describes), the region of src's index space
// its owner reads (empty: lo > hi).  comps: scomp .. scomp+nc-1 of src travel.
SrcView make_view(const vdn_multifab *src, const std::vector<vdn_box> &footprint, const std::vector<int> &dst_owner, int scomp, int nc, unsigned long cache_tag);
void view_cache_purge(unsigned long layout_uid);
bool   comm_active();
void   comm_allreduce_max_dev(double *d, int n);
void   comm_allreduce_max_u8_dev(unsigned char *d, size_t n);
void   comm_allgather_dev(const double *send, double *recv, size_t count);

// candidates of a box-pair loop: the boxes of a list (those with have[i] != 0) that may touch a query region, in ascending order -- a level of a
// thousand boxes against one of a few hundred is 262 000 pairs per descriptor build, 0.5-2 ms of host time with the GPU idle, a few dozen
// times per step; with the bins a build visits the handful of neighbours of every box
#include <algorithm>
struct BoxBins {
  int lo[3] = {0, 0, 0}, w[3] = {1, 1, 1}, n[3] = {0, 0, 0};
  std::vector<int> first, items; mutable std::vector<int> out;        // bin b holds items[first[b] .. first[b + 1])
  explicit BoxBins(const std::vector<vdn_box> &b, const std::vector<char> *have = nullptr) {
    int hi[3] = {0, 0, 0}; bool any = false;
    for (size_t i = 0; i < b.size(); i++) {
      if (have && !(*have)[i]) continue;
      for (int d = 0; d < 3; d++) {
        if (!any || b[i].lo[d] < lo[d]) lo[d] = b[i].lo[d];
        if (!any || b[i].hi[d] > hi[d]) hi[d] = b[i].hi[d];
        w[d] = std::max(w[d], b[i].hi[d] - b[i].lo[d] + 1);
      }
      any = true;
    }
    if (!any) return;
    for (int d = 0; d < 3; d++) n[d] = (hi[d] - lo[d]) / w[d] + 1;
    const size_t nb = (size_t)n[0] * n[1] * n[2];
    first.assign(nb + 1, 0);
    for (int pass = 0; pass < 2; pass++) {                 // count, prefix sums, fill
      for (size_t i = 0; i < b.size(); i++) {
        if (have && !(*have)[i]) continue;
        int a[3], z[3];
        for (int d = 0; d < 3; d++) { a[d] = (b[i].lo[d] - lo[d]) / w[d]; z[d] = (b[i].hi[d] - lo[d]) / w[d]; }
        for (int k = a[2]; k <= z[2]; k++) for (int j = a[1]; j <= z[1]; j++) for (int q = a[0]; q <= z[0]; q++) {
          const size_t bin = ((size_t)k * n[1] + j) * n[0] + q;
          if (pass == 0) first[bin + 1]++; else items[first[bin]++] = (int)i;
        }
      }
      if (pass == 0) { for (size_t q = 0; q < nb; q++) first[q + 1] += first[q]; items.resize(first[nb]); }
      else { for (size_t q = nb; q > 0; q--) first[q] = first[q - 1]; first[0] = 0; }      // (the fill advanced every start to its end)
    }
  }
  // boxes that may intersect [qlo - margin, qhi + margin] (a superset of those that do), ascending
  const std::vector<int> &near(const int qlo[3], const int qhi[3], int margin) const {
    out.clear();
    if (first.empty()) return out;
    int a[3], z[3];
    for (int d = 0; d < 3; d++) {
      const int l = qlo[d] - margin - lo[d], h = qhi[d] + margin - lo[d];
      if (h < 0 || l > n[d] * w[d] - 1) return out;
      a[d] = l < 0 ? 0 : l / w[d]; z[d] = std::min(h / w[d], n[d] - 1);
    }
    for (int k = a[2]; k <= z[2]; k++) for (int j = a[1]; j <= z[1]; j++) for (int q = a[0]; q <= z[0]; q++) {
      const size_t bin = ((size_t)k * n[1] + j) * n[0] + q;
      out.insert(out.end(), items.begin() + first[bin], items.begin() + first[bin + 1]);
    }
    std::sort(out.begin(), out.end()); out.erase(std::unique(out.begin(), out.end()), out.end());
    return out;
  }
};
// helpers shared between translation units
BoxP make_boxp(const vdn_multifab *mf, int i, const vdn_bc_tower *bct);
void mf_setval(vdn_multifab *mf, double val, int comp, int nc, bool all);
void mf_fill_boundary(vdn_multifab *mf, bool faces_only = false);
void mf_physbc(vdn_multifab *mf, int scomp, int bccomp, int nc, const vdn_bc_tower *bct, bool same_boundary = false);
void mf_copy(vdn_multifab *dst, int dcomp, const vdn_multifab *src, int scomp, int nc, int ng);
double mf_norm_inf(const vdn_multifab *mf, int comp, int nc);
double mf_norm_inf_grown(const vdn_multifab *mf, int comp, int nc, int grow);
double mf_max_ratio3(vdn_multifab *const *a, vdn_multifab *const *b);      // max of max(a/b, b/a) over the valid points of three pairs (component 0), all ranks
// ml_restrict_and_fill on one level = fill_boundary + physbc
void mf_restrict_and_fill(vdn_multifab *mf, int icomp, int bcomp, int nc, bool same_boundary, const vdn_bc_tower *bct);

// godunov.hip
bool god_per_box(const vdn_multifab *s);       // the level takes the box-by-box Godunov path (one box, or a few large boxes), not the box-batched one
void k_velpred(const vdn_multifab *u, vdn_multifab **umac, const vdn_multifab *force, const double *dx, double dt,
               const vdn_bc_tower *bct);
// k_mkflux with the update that follows it inside the same march (godunov.hip, UPD): snew and the update's forcing term -- fmode 0: the
// force multifab of the call; fmode 1: ext + (lapu0 - gp) / rho formed in place (mkvelforce on a valid cell, no lapu array)
struct MkUpdate { vdn_multifab *snew = nullptr; int fmode = 0; const vdn_multifab *ext = nullptr, *gp = nullptr, *rho = nullptr; double lapu0 = 0.0; };
bool k_mkflux(const vdn_multifab *s, vdn_multifab **sedge, vdn_multifab **flux, vdn_multifab **umac,
              const vdn_multifab *force, const vdn_multifab *mac_rhs, const double *dx, double dt,
              const vdn_bc_tower *bct, bool is_vel, const int *is_cons, const MkUpdate *upd = nullptr);
void k_slope(const vdn_multifab *s, vdn_multifab *slope, int dir, int bccomp, const vdn_bc_tower *bct);
// pointwise.hip
void k_update_velforce(const vdn_multifab *uold, vdn_multifab **umac, vdn_multifab **uedge, const vdn_multifab *ext, const vdn_multifab *s,
                       const vdn_multifab *gp, const vdn_multifab *lapu, double visc_fac, vdn_multifab *unew, const double *dx, double dt);
void k_update(const vdn_multifab *sold, vdn_multifab **umac, vdn_multifab **sedge, vdn_multifab **flux,
              const vdn_multifab *force, vdn_multifab *snew, const double *dx, double dt, bool is_vel, const int *is_cons);
void k_mkvelforce(vdn_multifab *vf, const vdn_multifab *ext, const vdn_multifab *s, const vdn_multifab *gp,
                  const vdn_multifab *lapu, double visc_fac);
void k_mkscalforce(vdn_multifab *sf, const vdn_multifab *ext, const vdn_multifab *laps, double diff_fac);
void k_make_at_halftime(vdn_multifab *rhohalf, const vdn_multifab *sold, const vdn_multifab *snew, int in_comp, int out_comp);
void k_estdt_max(const vdn_multifab *u, const vdn_multifab *s, const vdn_multifab *gp, const vdn_multifab *ext, double out6[6]);
// macproject.hip / mg_cc.hip
void do_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs, const double *dx,
                   const vdn_bc_tower *bct, int bc_comp0);
// cc_solve's fast path for macproject on one level (mg_cc.hip): right-hand side from the MAC field, coefficients from rho, phi handed back as views
struct CcFast { vdn_multifab **um = nullptr; const vdn_multifab *mac_rhs = nullptr, *rho = nullptr; std::vector<FV> phi_view; };
int  cc_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int bc[3][2],
              double rel_eps, double abs_eps, int max_iter, int *cycles, double *res0, double *res,
              const vdn_multifab *alpha = nullptr, const vdn_multifab *rho = nullptr,    // rho: beta = 2/(rho_i + rho_i-1), recomputed on the finest level
              struct CcKeep *keep = nullptr, CcFast *fast = nullptr, int fmg = 0, bool zero_guess = false, vdn_multifab *add_to = nullptr,
              double const_beta = 0.0);      // const_beta > 0 (with alpha): every face coefficient of `beta` is this constant (visc_solve, diff_scalar_solve) -- the finest level may then live by colour without coefficient arrays   // fmg: the caller's phi is zero and max_iter >= 0: start from the nested iteration (cc_fmg); zero_guess (a kept hierarchy's later calls, max_iter < 0): phi is not read, the guess is zero; add_to += the solution on the valid cells          // keep: see mg_cc.hip (hierarchy kept between the calls of a composite solve)
struct CcKeep *cc_keep_new(); void cc_keep_free(struct CcKeep *k);
int  mg_agglom(const vdn_layout *la, int lev);     // box width below which a multi-box multigrid level is gathered into one box (mg_cc.hip)
void cc_smooth(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int bc[3][2], int nsweeps);
void cc_bench_smoother(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const vdn_multifab *rho, const double *dx, const int bc[3][2],
                       int nlaunch, double *avg_ms, long *cells, int slab_sweeps = 0);
// viscous.hip
void k_explicit_diffusive_term(vdn_multifab *lap, const vdn_multifab *data, int comp, int bccomp0, const double *dx, const vdn_bc_tower *bct);
void do_visc_solve(vdn_layout *mla, vdn_multifab *unew, const vdn_multifab *lapu, const vdn_multifab *rho, const vdn_multifab *mac_rhs,
                   const double *dx, double mu, const vdn_bc_tower *bct);
void do_diff_scalar_solve(vdn_layout *mla, vdn_multifab *snew, const vdn_multifab *laps, const double *dx, double mu,
                          const vdn_bc_tower *bct, int icomp, int bccomp0);
// hgproject.hip / mg_nd.hip
void do_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold, vdn_multifab **rhohalf,
                  vdn_multifab **p, vdn_multifab **gp, const double *dx, double dt, const vdn_bc_tower *bct, int press_comp0);
// nd_solve's fast path for hgproject on one level (mg_nd.hip): sigma = 1 / rhohalf, rh = phi = 0 on entry, phi handed back as views
struct NdFast { const vdn_multifab *rhohalf = nullptr; std::vector<FV> phi_view; };
int  nd_solve(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *coeffs, const vdn_multifab *u, const double *dx,
              const int bc[3][2], double rel_eps, double abs_eps, int max_iter, int *cycles, double *res0, double *res, struct NdKeep *keep = nullptr,
              NdFast *fast = nullptr, bool fmg_start = false, bool rh_is_b = false, vdn_multifab *add_to = nullptr);      // fmg_start: nested iteration before a FIXED number of cycles (max_iter < 0; the caller's phi is zero); rh_is_b: `rh` holds b = -rh and phi is not read (zero guess; max_iter < 0); add_to += the solution on the valid nodes

// dim2.hip: the dm = 2 path (one level, one box)
void k2_mkvelforce(vdn_multifab *vf, const vdn_multifab *ext, const vdn_multifab *s, const vdn_multifab *gp, const vdn_multifab *lapu, double visc_fac);
void k2_mkscalforce(vdn_multifab *sf, const vdn_multifab *ext, const vdn_multifab *laps, double diff_fac);
void k2_update(const vdn_multifab *sold, vdn_multifab **umac, vdn_multifab **sedge, vdn_multifab **flux, const vdn_multifab *force, vdn_multifab *snew,
               const double *dx, double dt, bool is_vel, const int *is_cons);
void k2_estdt_max(const vdn_multifab *u, const vdn_multifab *s, const vdn_multifab *gp, const vdn_multifab *ext, double out6[6]);
int  cc2_solve(vdn_multifab *rh, vdn_multifab *phi, vdn_multifab **beta, const double *dx, const int bc[3][2], double rel_eps, double abs_eps, int max_iter,
               int *cycles, double *res0, double *res, const vdn_multifab *alpha);
void do2_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs, const double *dx, const vdn_bc_tower *bct, int bc_comp0);
void k2_explicit_diffusive_term(vdn_multifab *lap, const vdn_multifab *data, int comp, int bccomp0, const double *dx, const vdn_bc_tower *bct);
void do2_visc_solve(vdn_layout *mla, vdn_multifab *unew, const vdn_multifab *lapu, const vdn_multifab *rho, const vdn_multifab *mac_rhs,
                    const double *dx, double mu, const vdn_bc_tower *bct);
void do2_diff_scalar_solve(vdn_layout *mla, vdn_multifab *snew, const vdn_multifab *laps, const double *dx, double mu, const vdn_bc_tower *bct, int icomp, int bccomp0);
int  nd2_solve(vdn_multifab *rh, vdn_multifab *phi, const vdn_multifab *coeffs, const vdn_multifab *u, const double *dx, const int bc[3][2],
               double rel_eps, double abs_eps, int max_iter, int *cycles, double *res0, double *res);
void do2_hgproject(int proj_type, vdn_layout *mla, vdn_multifab **unew, vdn_multifab **uold, vdn_multifab **rhohalf, vdn_multifab **p, vdn_multifab **gp,
                   const double *dx, double dt, const vdn_bc_tower *bct, int press_comp0);

// per-level pieces of macproject (mg_cc.hip) and the two-level AMR operators (amr.hip)
void mac_level_rhs(vdn_multifab **um, const vdn_multifab *mac_rhs, vdn_multifab *rh, const double *dx);
void mac_level_coeffs(const vdn_multifab *rho, vdn_multifab **beta);
void mac_level_mkumac(vdn_multifab **um, const vdn_multifab *phi, vdn_multifab **beta, const double *dx, const vdn_bc_tower *bct, int bc_comp0);
void ml_cc_restriction(vdn_multifab *crse, const vdn_multifab *fine, int icomp, int nc);
void ml_edge_restriction(vdn_multifab *crse, const vdn_multifab *fine, int dir, int comp = 0);
void ml_fill_ghost_cells(vdn_multifab *fine, const vdn_multifab *crse, int icomp, int nc);
void ml_create_umac_grown(vdn_multifab *fine, const vdn_multifab *crse, int dir);
void ml_restrict_and_fill(int nlev, vdn_multifab **mf, int icomp, int bcomp, int nc, bool same_boundary, const vdn_bc_tower *bct);
int ml_cc_solve(vdn_layout *la, vdn_multifab **rh, vdn_multifab **phi, vdn_multifab **beta, const double *dx, const vdn_bc_tower *bct, int bc_comp0,
                double rel_eps, int max_iter, int *iters, double *res0, double *res, vdn_multifab **alpha, vdn_multifab **base_beta = nullptr, const vdn_multifab *base_rho = nullptr, const vdn_multifab *fine_rho = nullptr,
                double const_beta = 0.0);      // const_beta > 0: every face coefficient on every level is this constant (the viscous / diffusive solves): handed to level 0's V-cycles
void do_ml_visc_solve(vdn_layout *mla, vdn_multifab **unew, vdn_multifab **lapu, vdn_multifab **rho, vdn_multifab **mac_rhs,
                      const double *dx, double mu, const vdn_bc_tower *bct);
void do_ml_diff_scalar_solve(vdn_layout *mla, vdn_multifab **snew, vdn_multifab **laps, const double *dx, double mu,
                             const vdn_bc_tower *bct, int icomp, int bccomp0);
void do_ml_macproject(vdn_layout *mla, vdn_multifab **umac, vdn_multifab **rho, vdn_multifab **mac_rhs, const double *dx, const vdn_bc_tower *bct, int bc_comp0);
