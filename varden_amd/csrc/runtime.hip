// runtime.hip -- context, arena, ml_layout / multifab / bc_tower containers and the ghost-cell
// operators (multifab_fill_boundary, multifab_physbc) of the MI355X-native VARDEN hot path.
//
// Reference behaviour restated here:
//   define_bc_tower.f90:129-340   (phys / adv / ell tables)
//   multifab_physbc.f90:238-561   (physbc_3d)
//   FBoxLib multifab_fill_boundary / setval / copy_c / norm_inf (external to the reference tree;
//   semantics from their call sites)
#include "vdn_dev.h"
#include <algorithm>
#include <chrono>
#include <tuple>

// ================================================================================================
// context / errors / arena
// ================================================================================================
static VdnCtx g_ctx;
VdnCtx &ctx() { return g_ctx; }
static thread_local char g_err[1024] = "";

void vdn_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
void vdn_fail(const char *fmt, ...) {
  char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  throw VdnErr(buf);
}

// also drops the pointers advance_timestep keeps INTO the arena (limited slopes of uold, max |umac|): a step that ended in an exception
// (solver_check throws by default) must not leave them dangling for the next stand-alone vdn_k_mkflux / vdn_k_velpred
static bool arena_poison() { static const bool p = vdn_env("VDN_ARENA_POISON") && atoi(vdn_env("VDN_ARENA_POISON")) != 0; return p; }
// the descriptors of arena temporaries (mf_temp) that nobody freed: they die with the arena contents they describe
static std::vector<vdn_multifab *> g_temp_mfs;
void arena_reset() {
  dbg_sync(16);
  for (vdn_multifab *m : g_temp_mfs) delete m;
  g_temp_mfs.clear();
  if (arena_poison() && g_ctx.arena && g_ctx.arena_peak > 0) HIPCHK(hipMemsetAsync(g_ctx.arena, 0xFF, std::min(g_ctx.arena_peak + (size_t)(64 << 20), g_ctx.arena_bytes), g_ctx.stream));
  g_ctx.arena_off = 0;
  g_ctx.drop_step_caches();
}
// The arena is a range of device ADDRESSES reserved once (hipMemAddressReserve: the card's whole memory, rounded up) into which physical memory is mapped in
// 1 GB chunks as the high-water mark moves (hipMemCreate + hipMemMap + hipMemSetAccess: tens of microseconds per chunk, tools/probes/vmm_probe2.hip).  Rounds 1-5
// sized it up front -- 150 ghosted fields of the layout, 120 GB for the three-level 256^3 hierarchy -- with one hipMalloc, and a regrid that needed more freed it
// and allocated again: 3-7 SECONDS per hipMalloc beyond some 30 GB on this card (profiles/r06_regrid_cost.txt), every kept descriptor set and graph dropped because
// the base moved.  Now the base never moves, nothing is guessed, and only what a step touches is backed by memory.
static size_t env_mb(const char *name, size_t dflt_mb) { const char *e = vdn_env(name); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : (long)dflt_mb) << 20; }
static size_t arena_chunk() { static const size_t v = env_mb("VDN_ARENA_CHUNK_MB", 1024); return v; }
#define ARENA_CHUNK arena_chunk()
static std::vector<hipMemGenericAllocationHandle_t> g_arena_chunks;
static size_t g_arena_va = 0;
static void arena_map_to(size_t bytes) {                   // c.arena_bytes (= mapped bytes) >= bytes afterwards
  VdnCtx &c = g_ctx;
  if (bytes <= c.arena_bytes) return;
  if (!c.arena) {
    int vmm = 0; HIPCHK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, c.device));
    REQUIRE(vmm, "the device does not support hipMemAddressReserve / hipMemMap (the scratch arena needs them)");
    size_t free_b = 0, total_b = 0; HIPCHK(hipMemGetInfo(&free_b, &total_b));
    g_arena_va = ((total_b + ARENA_CHUNK - 1) / ARENA_CHUNK) * ARENA_CHUNK;
    void *base = nullptr; HIPCHK(hipMemAddressReserve(&base, g_arena_va, (size_t)2 << 20, nullptr, 0));
    c.arena = (char *)base;
  }
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = c.device;
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = c.device; acc.flags = hipMemAccessFlagsProtReadWrite;
  // (a temporary first asked for inside a graph capture: the mapping calls are no stream work, the relaxed mode says so for this thread while they run)
  struct Relax { hipStreamCaptureMode m = hipStreamCaptureModeRelaxed; Relax() { (void)hipThreadExchangeStreamCaptureMode(&m); } ~Relax() { (void)hipThreadExchangeStreamCaptureMode(&m); } } relax_;
  while (c.arena_bytes < bytes) {
    if (c.arena_bytes + ARENA_CHUNK > g_arena_va) vdn_fail("arena exhausted: %zu bytes wanted, the card holds %zu", bytes, g_arena_va);
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, ARENA_CHUNK, &prop, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); vdn_fail("arena: no device memory for another chunk (%zu bytes mapped, %zu wanted): %s", c.arena_bytes, bytes, hipGetErrorString(e)); }
    e = hipMemMap(c.arena + c.arena_bytes, ARENA_CHUNK, 0, h, 0);
    if (e == hipSuccess) e = hipMemSetAccess(c.arena + c.arena_bytes, ARENA_CHUNK, &acc, 1);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); vdn_fail("arena: mapping a chunk at offset %zu failed: %s", c.arena_bytes, hipGetErrorString(e)); }
    g_arena_chunks.push_back(h);
    // fresh device memory holds whatever its last owner left: a temporary that is read where nobody wrote (a ghost entry multiplied by a zero coefficient ...) would make
    // a run differ from process to process.  Zeros -- or, under VDN_ARENA_POISON, the NaNs that released arena bytes get, so that such a read fails loudly
    HIPCHK(hipMemsetAsync(c.arena + c.arena_bytes, arena_poison() ? 0xFF : 0x00, ARENA_CHUNK, c.stream));
    c.arena_bytes += ARENA_CHUNK;
  }
}
static void arena_destroy() {
  VdnCtx &c = g_ctx;
  if (!c.arena) return;
  if (c.arena_bytes) HIPCHK(hipMemUnmap(c.arena, c.arena_bytes));
  for (hipMemGenericAllocationHandle_t h : g_arena_chunks) HIPCHK(hipMemRelease(h));
  g_arena_chunks.clear();
  HIPCHK(hipMemAddressFree(c.arena, g_arena_va));
  c.arena = nullptr; c.arena_bytes = 0; c.arena_off = 0; g_arena_va = 0;
}
// The fields of the state (vdn_multifab_create) come the same way: an address range per field, backed by 64 MB chunks of physical memory that go back to a POOL
// when the field is destroyed, not to the driver.  A regrid frees and creates some 36 GB-sized fields of slightly different sizes; memory that comes back from the
// driver costs about 10 ms per GB on this card (hipMalloc and hipMemCreate alike -- 360 ms of a regrid, tools/probes/regrid_profile_probe.py), chunks from
// the pool cost the mapping calls, some 10 us each.  Fields below 32 MB are plain hipMalloc blocks (a chunk each would waste the card on the small cases), and so are
// the fields of a ONE-level layout, which are created once: the one-box 512^3 step runs 3 % slower on chunk-backed fields than on hipMalloc blocks (7 % on 1 GB chunks;
// a per-field offset into the chunk did not change that: profiles/r06_allocator_ab.txt) -- the boxes of a hierarchy are small and show no difference.
static size_t field_chunk() { static const size_t v = env_mb("VDN_FIELD_CHUNK_MB", 64); return v; }
static size_t field_small() { static const size_t v = vdn_env("VDN_FIELD_VMM") && atoi(vdn_env("VDN_FIELD_VMM")) == 0 ? ~(size_t)0 : (size_t)32 << 20; return v; }
#define FIELD_CHUNK field_chunk()
#define FIELD_SMALL field_small()
struct FieldAlloc { void *va; size_t va_bytes; std::vector<hipMemGenericAllocationHandle_t> chunks; };
static std::map<void *, FieldAlloc> g_field_allocs;
static std::vector<hipMemGenericAllocationHandle_t> g_chunk_pool;
static void *field_alloc(size_t bytes, bool pooled) {
  VdnCtx &c = g_ctx;
  if (!pooled || bytes < FIELD_SMALL) { void *q = nullptr; HIPCHK(hipMalloc(&q, bytes)); return q; }
  const size_t n = (bytes + FIELD_CHUNK - 1) / FIELD_CHUNK, sz = n * FIELD_CHUNK;
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = c.device;
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = c.device; acc.flags = hipMemAccessFlagsProtReadWrite;
  void *p = nullptr;
  HIPCHK(hipMemAddressReserve(&p, sz, (size_t)2 << 20, nullptr, 0));
  FieldAlloc A; A.va = p; A.va_bytes = sz;
  hipError_t e = hipSuccess;
  for (size_t i = 0; i < n && e == hipSuccess; i++) {
    hipMemGenericAllocationHandle_t h;
    if (!g_chunk_pool.empty()) { h = g_chunk_pool.back(); g_chunk_pool.pop_back(); }
    else if ((e = hipMemCreate(&h, FIELD_CHUNK, &prop, 0)) != hipSuccess) break;
    e = hipMemMap((char *)p + i * FIELD_CHUNK, FIELD_CHUNK, 0, h, 0);
    if (e == hipSuccess) A.chunks.push_back(h); else g_chunk_pool.push_back(h);
  }
  if (e == hipSuccess) e = hipMemSetAccess(p, sz, &acc, 1);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (!A.chunks.empty()) (void)hipMemUnmap(p, A.chunks.size() * FIELD_CHUNK);
    for (auto h : A.chunks) g_chunk_pool.push_back(h);
    (void)hipMemAddressFree(p, sz);
    vdn_fail("no device memory for a field of %zu bytes (%zu chunks in the pool): %s", bytes, g_chunk_pool.size(), hipGetErrorString(e));
  }
  g_field_allocs[p] = std::move(A);
  return p;
}
static void field_free(void *p) {
  auto it = g_field_allocs.find(p);
  if (it == g_field_allocs.end()) { HIPCHK(hipFree(p)); return; }      // (a small field)
  HIPCHK(hipMemUnmap(it->second.va, it->second.va_bytes));
  for (auto h : it->second.chunks) g_chunk_pool.push_back(h);
  HIPCHK(hipMemAddressFree(it->second.va, it->second.va_bytes));
  g_field_allocs.erase(it);
}
static void field_pool_release() { for (auto h : g_chunk_pool) (void)hipMemRelease(h); g_chunk_pool.clear(); }
// back at least `bytes` of the arena with memory now
void arena_reserve(size_t bytes) { arena_map_to(bytes); }
// nothing to size: the arena follows the steps' high-water mark (kept for its callers: the start of every public entry point)
void arena_reserve_for(const vdn_layout *) {}
size_t arena_mark() { return g_ctx.arena_off; }
// VDN_ARENA_POISON=1 (debugging): whatever is handed back to the arena is overwritten with NaNs (all-ones bytes), so that a kernel which reads
// an entry nobody wrote -- and only worked because the last tenant of that address left zeros there -- meets a NaN; the norms turn it into a
// failed solve.  The GPU suite is run once per round this way (tools/r3_poison.sh).
void arena_release(size_t mark) {
  if (arena_poison() && g_ctx.arena && g_ctx.arena_off > mark) HIPCHK(hipMemsetAsync(g_ctx.arena + mark, 0xFF, g_ctx.arena_off - mark, g_ctx.stream));
  g_ctx.arena_off = mark;
}
void *arena_alloc(size_t bytes) {
  VdnCtx &c = g_ctx;
  size_t off = (c.arena_off + 255) & ~(size_t)255;
  if (off + bytes > c.arena_bytes) arena_map_to(off + bytes);      // (legal with temporaries live and work in flight: the mapped part does not move)
  c.arena_off = off + bytes;
  if (c.arena_off > c.arena_peak) c.arena_peak = c.arena_off;
  return c.arena + off;
}

extern "C" void vdn_params_default(vdn_params *p) {
  memset(p, 0, sizeof *p);
  p->dm = 3; p->nscal = 2; p->slope_order = 4; p->use_minion = 0; p->boussinesq = 0;
  p->stencil_order = 2; p->diffusion_type = 1; p->verbose = 0; p->mg_verbose = 0; p->prob_type = 1;
  p->visc_coef = 0.0; p->diff_coef = 0.0; p->cflfac = 0.8; p->max_dt_growth = 1.1;
  p->mg_nu1 = 2; p->mg_nu2 = 2; p->mg_nub = 8; p->mg_max_iter = 100;
  p->hg_max_iter = 100; p->hg_nu1 = 2; p->hg_nu2 = 1; p->hg_nub = 8; p->hg_omega = 0.9;     // hg_nub: 32 until round 3 -- the coarsest level (3^3 nodes under a 2^k box) gains nothing from more than max(8, 2 N^2) sweeps (same cycle counts), and each costs ~1.2 us of a one-workgroup launch
  p->mac_rel_eps = 1.0e-10; p->hg_rel_eps = -1.0; p->abort_on_max_iter = 1; p->hg_fmg = 1; p->mac_fmg = 1; p->hg_omega_pre1 = 1.45; p->hg_omega_pre2 = 0.7; p->hg_omega_fac1 = 1.6; p->hg_omega_fac2 = 0.9; p->hg_omega_fac3 = 0.65; p->mg_predict = 1;
}

// ---- roctx ranges ------------------------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <thread>
static int (*g_roctx_push)(const char *) = nullptr;
static int (*g_roctx_pop)() = nullptr;
void prof_load() {
  static bool tried = false;
  if (tried) return;
  tried = true;
  if (vdn_env("VDN_NO_ROCTX")) return;
  for (const char *n : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" }) {
    void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    *(void **)(&g_roctx_push) = dlsym(h, "roctxRangePushA");
    *(void **)(&g_roctx_pop) = dlsym(h, "roctxRangePop");
    if (g_roctx_push && g_roctx_pop) return;
    g_roctx_push = nullptr; g_roctx_pop = nullptr;
  }
}
Prof::Prof(const char *name) : on(g_roctx_push != nullptr) { if (on) g_roctx_push(name); }
Prof::~Prof() { if (on) g_roctx_pop(); }

// ---- debug / measurement switches ------------------------------------------------------------------------------------------------
// Every environment variable the library reads, with what it does.  None changes a result: they select between launch forms that the tests hold bit-for-bit
// equal (tests/test_projection_gpu.py::test_multigrid_launch_variants_agree_bit_for_bit, test_kernels_gpu.py, test_amr_gpu.py) or are probes.  vdn_env() is the
// only way the library reads the environment: a name missing from this table fails the call, and vdn_init warns about VDN_* variables it does not know
// (a misspelt switch would otherwise be silently ignored).  vdn_debug_switches() hands the table out (include/varden_amd.h).
struct EnvSwitch { const char *name, *doc; };
static const EnvSwitch g_switches[] = {
  { "VDN_TESTING", "1: allows VDN_RCCL_LIB (the test transport of tests/fake_rccl); nothing else" },
  { "VDN_RCCL_LIB", "path of a library that stands in for librccl -- honoured only with VDN_TESTING=1 and the test double's handshake" },
  { "VDN_FORCE_PACKED", "1: box-to-box copies of one rank go through the packed per-peer buffers (device memcpy for send/recv); 2: through a 1-rank RCCL communicator (one-GPU rehearsal of the N > 1 transport)" },
  { "VDN_ARENA_POISON", "1: every byte handed back to the arena is overwritten with NaNs (a read of an entry nobody wrote fails the next solve)" },
  { "VDN_ARENA_CHUNK_MB", "size of the physical chunks mapped into the arena's address range (default 1024)" },
  { "VDN_FIELD_CHUNK_MB", "size of the pooled physical chunks behind the state fields (default 64)" },
  { "VDN_FIELD_VMM", "0: every state field is one hipMalloc block (rounds 1-5) instead of pooled chunks mapped into its own address range" },
  { "VDN_MLCC_TRACE", "1: the composite cell-centred solve prints its residual at every FAC iteration (stderr)" },
  { "VDN_KEEP_OFF", "mask of kept-descriptor families rebuilt at every call: 1 generic sets, 2 create_umac_grown, 4 composite cell-centred solve, 8 nodal prolongation" },
  { "VDN_SYNC_POINTS", "mask of points that synchronise the device (race hunting): 1 after every batched launch, 2 after every staged upload, 4 after every exchange, 8 before a scalar read-back, 16 at arena_reset, 32 after launch_cells" },
  { "VDN_PHASE_HASH", "1: advance_timestep prints a checksum of its fields at every phase boundary (stderr)" },
  { "VDN_NO_ROCTX", "do not bind the roctx library (no bl_prof ranges)" },
  { "VDN_POLL", "scalar read-back: 1 spin on the pinned sequence number, 0 hipStreamSynchronize; default: spin on one rank, synchronise on several" },
  { "VDN_NO_GRAPHS", "launch every multigrid cycle eagerly instead of replaying its hipGraph" },
  { "VDN_NO_SLOPE_CACHE", "velocity mkflux recomputes the slopes of uold that velpred computed in the same step" },
  { "VDN_NO_FORCE_REUSE", "1: every forcing term is computed where the reference computes it (advance_premac AND velocity_advance, ...)" },
  { "VDN_SLOPES_Y", "0: the slopes march exchanges rows through LDS (kk_slopes_m) instead of reading the y-neighbours from memory (kk_slopes_my)" },
  { "VDN_SLOPES_MARCH", "0: the per-cell slopes kernel instead of the k-marching one" },
  { "VDN_GODUNOV_BATCH", "1: the descriptor (box-batched) Godunov kernels also on a level of one box" },
  { "VDN_GOD_SLAB_BC", "0: (unfused marches) boundary rules inside the marches instead of the face-centred code on boundary slabs" },
  { "VDN_GODUNOV_PLAIN", "the face-centred one-thread-per-cell Godunov kernels of round 1 (the marching kernels' bit-for-bit reference)" },
  { "VDN_GOD_SEGW", "0: the box-batched fused Godunov marches use full-width (64 x 8) tiles for every box" },
  { "VDN_GOD_P2", "0: the fused marches divide by dx also where every dx is a power of two (default there: scale by 1/dx, the same doubles)" },
  { "VDN_FUSED_KCHUNKS", "k-chunks of the fused marches (default: the count that fills the last round of workgroups best)" },
  { "VDN_GOD_FUSED", "0: one march per Godunov stage (B, C, D) instead of the fused B+C+D march" },
  { "VDN_GOD_UPDATE", "0: update_3d as its own pass instead of inside the fused mkflux march" },
  { "VDN_GSRB_PAIR", "0: one cell per thread in the colour passes / residuals of wide levels instead of the 2 x 2 pair form" },
  { "VDN_MAC_SPLIT", "0: the finest level of macproject's one-level solve stays interleaved (kk_cc_gsrb_rho_pair) instead of stored by colour (kk_cc_gsrb_rho_split); 2: only the colour passes on the split arrays, the residual on the level array" },
  { "VDN_MAC_SPLIT_MIN", "fewest cells (of this rank's boxes together) of a level stored by colour (default 2^23)" },
  { "VDN_MAC_SPLIT_HALO", "0: only a one-box level without periodic faces is stored by colour (round 5); default: any box list, periodic faces and several ranks too, the ghost exchange on the split arrays" },
  { "VDN_ND_REV", "0: every march of a nodal level walks its tiles in the same order (default: consecutive marches alternate)" },
  { "VDN_MAC_SLAB", "planes per slab of the time-skewed schedule of the split level's passes (cc_split_run; default: ~200 MB of pass traffic, at most half the level); 0: whole-level launches" },
  { "VDN_MAC_UMAX", "0: max |umac| by its own pass (kk_macmax) instead of inside macproject's velocity update (kk_mkumac_rho_max)" },
  { "VDN_MAC_KFLIP", "0: both colour passes of a sweep walk the planes upwards (default: the second colour downwards; paired and split passes of the cell-centred multigrid)" },
  { "VDN_CC_HALO_FACES", "0: the cell-centred multigrid exchanges the whole ghost shell instead of the faces only" },
  { "VDN_OVERLAP", "halo exchange of multigrid passes next to interior work: 1 always, 0 never, default: when a plan has a remote peer and the box is large" },
  { "VDN_MG_AGGLOM", "box width below which a multi-box multigrid level is gathered into one box (default: 64 across ranks, 128 where every box is this rank's)" },
  { "VDN_MG_RESTRICT_FUSED", "0: cell-centred residual and restriction as two passes" },
  { "VDN_MG_TAILCYCLE", "0: the smallest levels launch by launch instead of one single-workgroup cycle" },
  { "VDN_MG_PROLONG_FUSED", "0: cell-centred prolongation as its own pass instead of inside the first post-smoothing colour pass" },
  { "VDN_MG_LDS", "0: the 16^3..64^3 cell-centred levels launch by launch instead of the LDS-tiled down / up kernels" },
  { "VDN_MAC_STORED_BETA", "1: the finest MAC level reads stored face coefficients instead of recomputing them from rho" },
  { "VDN_MAC_FAST", "0: macproject with its rh / phi / beta multifabs as the reference has them" },
  { "VDN_HG_FAST", "0: hgproject with its rh / phi / coeffs multifabs as the reference has them" },
  { "VDN_ND_PAIR", "0: one node per lane in the nodal march instead of the pair form" },
  { "VDN_ND_LEAN", "0: whole-array zero fills of the big nodal levels instead of shell-only" },
  { "VDN_ND_RESTRICT_FUSED", "0: nodal residual and full weighting as two passes" },
  { "VDN_NDF_SEGW", "0: the marches of the composite nodal solve use power-of-two lane segments per node row only" },
  { "VDN_NDF_PAIR", "0: one node per lane in the box-batched nodal march of the composite solve" },
  { "VDN_NDM_IFACE_FACES", "0: interface interpolation of the composite nodal solve over whole boxes instead of box faces" },
  { "VDN_NDM_PROLONG8", "0: correction interpolation with a thread per fine node instead of per coarse node" },
  { "VDN_NDM_NEG", "1: the composite nodal solve copies -res into the correction's right-hand side instead of loading it directly" },
  { "VDN_FB_FACES", "0: the ghost exchanges of the composite cell-centred solve fill edges and corners too" },
  { "VDN_MLCC_RHO", "0: the composite MAC solve reads stored face coefficients on its finest level too" },
  { "VDN_GOD_NARROW", "0: the remainder tile column of the fused mkflux + update march in full 64-lane tiles instead of narrow segments (kk_mk_F_mn)" },
  { "VDN_GOD_1B", "0: the fused mkflux + update march of a one-box level with three workgroup barriers per plane (round 4) instead of one (godunov.hip, ONEB)" },
  { "VDN_KEEP_SETS", "0: the descriptor arrays of the inter-level operators and composite solves are rebuilt and uploaded at every call" },
  { "VDN_KEPT_BOUND", "n > 0: the kept descriptor tables hold at most n entries each (default 4096 / 64 / 512): the eviction paths in a test" },
  { "VDN_MLCC_GLUE", "0: the level-0 correction of the composite MAC solve stored and added in separate passes" },
  { "VDN_MLCC_FUSE1", "0: the composite MAC solve's finest-level residual and first colour pass as two launches" },
  { "VDN_BATCH_YZ", "0: no (j,k) / (i,k) tiles for thin ranges in the box-batched kernels" },
  { "VDN_BATCH_PPW", "planes per workgroup of the light box-batched kernels (default 8)" },
  { "VDN_BATCH_FLAT", "0: no flattened (i,j) plane mapping for badly filling tiles" },
  { "VDN_BATCH_CHUNK", "0: box-batched workgroups take strided instead of contiguous plane chunks" }
};
// Round 6: the switches exist in the TESTING build only (libvarden_amd_testing.so: -DVDN_TESTING_BUILD on this file and exchange.hip; the suite, the A/B tools and
// the one-GPU transport rehearsal load it -- VDN_LIB_FLAVOUR=testing in the Python mirror).  The shipped libvarden_amd.so reads NO environment variable: every launch
// form is the default one, every choice that matters is a field of vdn_params; vdn_init says so once if VDN_* switches are set.
extern "C" const char *vdn_build_flavour(void) {
#ifdef VDN_TESTING_BUILD
  return "testing";
#else
  return "release";
#endif
}
const char *vdn_env(const char *name) {
  for (const EnvSwitch &e : g_switches) if (!strcmp(e.name, name)) {
#ifdef VDN_TESTING_BUILD
    return getenv(name);
#else
    return nullptr;
#endif
  }
  vdn_fail("internal: the switch %s is not declared in the table of runtime.hip", name);
}
extern char **environ;
static void env_warn_unknown() {
  static bool done = false;
  if (done) return;
  done = true;
  for (char **e = environ; e && *e; e++) {
    if (strncmp(*e, "VDN_", 4) != 0 || !strncmp(*e, "VDN_BENCH_", 10) || !strncmp(*e, "VDN_WORKER_", 11)) continue;     // (bench.py's and the test workers' own variables)
    const char *eq = strchr(*e, '=');
    const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
    bool known = false;
    for (const EnvSwitch &s : g_switches) if (strlen(s.name) == len && !strncmp(s.name, *e, len)) known = true;
    if (!strncmp(*e, "VDN_LIB_FLAVOUR", 15)) continue;                  // (read by the Python mirror: which of the two libraries to load)
    if (!known) fprintf(stderr, "varden_amd: warning: environment variable %.*s is not a switch of this library (vdn_debug_switches lists them)\n", (int)len, *e);
#ifndef VDN_TESTING_BUILD
    else fprintf(stderr, "varden_amd: note: %.*s is set but this is the release build: launch-form switches are compiled out (libvarden_amd_testing.so reads them)\n", (int)len, *e);
#endif
  }
}
extern "C" const char *vdn_debug_switches(void) {
  static std::string out;
#ifndef VDN_TESTING_BUILD
  if (out.empty()) out = "release build: the switches below are compiled out, every launch form is the default one (libvarden_amd_testing.so reads them)\n";
#endif
  if (out.find("VDN_TESTING ") == std::string::npos) for (const EnvSwitch &s : g_switches) { const char *v = getenv(s.name); out += s.name; out += v ? std::string(" = ") + v : std::string(" (unset)"); out += ": "; out += s.doc; out += "\n"; }
  return out.c_str();
}

// ---- scalar read-back ------------------------------------------------------------------------------------------------------------
bool g_capturing_now();
// The values, then (after a system-scope fence) a sequence number in slot 64 of the pinned buffer: the host spins on that number instead of
// sleeping in hipStreamSynchronize -- a V-cycle loop reads one norm per cycle, and the wake-up of a blocked host thread was most of the ~20 us
// the GPU sat idle after every k_publish (profiles/r03_bench_trace_gaps.txt).  The stream is queried now and then: an error or an idle stream
// without the number (memory that is not host-coherent) falls back to the synchronisation.  VDN_POLL=0: synchronise always.
__global__ void k_publish(double *host_view, const double *dev, int n, unsigned long long seq) {
  if ((int)threadIdx.x < n) host_view[threadIdx.x] = dev[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *reinterpret_cast<volatile unsigned long long *>(host_view + 64) = seq; }
}
const double *read_scalars(const double *dev, int n) {
  VdnCtx &c = g_ctx;
  REQUIRE(n >= 1 && n <= 64, "read_scalars: 1..64 values");
  static unsigned long long seq = 0;
  // VDN_POLL: 1 = spin, 0 = synchronise always; unset: spin on a one-rank run, synchronise when several ranks run (each rank's spinning thread would
  // take a core from RCCL's proxy threads and from the other ranks of an oversubscribed host)
  static const int poll_env = vdn_env("VDN_POLL") ? atoi(vdn_env("VDN_POLL")) : -1;
  const bool poll = poll_env >= 0 ? poll_env != 0 : c.nranks == 1;
  ++seq;
  dbg_sync(8);
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c.stream, c.h_scal_dev, dev, n, seq);
  if (poll && !g_capturing_now()) {
    volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(c.h_scal + 64);
    for (unsigned long spins = 1;; spins++) {
      if (*flag == seq) { __sync_synchronize(); return c.h_scal; }
      __builtin_ia32_pause();                               // a spin-wait hint: the sibling hyperthread keeps its issue slots
      if ((spins & 0xfff) == 0) {                           // every ~4 k reads: has the stream failed, or finished without the number showing up?
        const hipError_t q = hipStreamQuery(c.stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIPCHK(q);
        if (spins > 0x40000) std::this_thread::yield();     // a long wait (a whole solve queued ahead): let other threads of the host run
      }
    }
  }
  HIPCHK(hipStreamSynchronize(c.stream));
  return c.h_scal;
}

// ---- residual-norm history (vdn_params.mg_predict) --------------------------------------------------------------------------------
__global__ void k_hist_push(double *hist, const double *nrm) {
  int *cnt = reinterpret_cast<int *>(hist + 64);
  const int c = *cnt;
  if (c < 64) hist[c] = *nrm;
  *cnt = c + 1;
}
void norm_hist_reset() { HIPCHK(hipMemsetAsync(g_ctx.d_hist + 64, 0, sizeof(double), g_ctx.stream)); }
void norm_hist_push(const double *d_nrm) { hipLaunchKernelGGL(k_hist_push, dim3(1), dim3(1), 0, g_ctx.stream, g_ctx.d_hist, d_nrm); }
const double *norm_hist_read(int n) {
  REQUIRE(n >= 1 && n <= 64, "norm_hist_read: 1..64 entries");
  comm_allreduce_max_dev(g_ctx.d_hist, n);
  return read_scalars(g_ctx.d_hist, n);
}
int g_mg_predict_off = 0;
static std::map<unsigned long long, int> g_mg_pred;
static unsigned long long mg_pred_key(int solver, const int n[3]) { GraphKey k; k.put(solver); k.put(n[0]); k.put(n[1]); k.put(n[2]); return k.h; }
int mg_predict_get(int solver, const int n[3]) {
  if (!g_ctx.prm.mg_predict || g_mg_predict_off > 0) return 0;
  auto it = g_mg_pred.find(mg_pred_key(solver, n));
  return it == g_mg_pred.end() ? 0 : it->second;
}
void mg_predict_set(int solver, const int n[3], int cycles) { if (g_mg_pred.size() > 4096) g_mg_pred.clear(); g_mg_pred[mg_pred_key(solver, n)] = cycles; }

// ---- hipGraph cache ---------------------------------------------------------------------------------------------------------------
static std::map<unsigned long long, hipGraphExec_t> g_graphs;
static bool g_capturing = false;
bool g_capturing_now() { return g_capturing; }
bool graphs_enabled() {
  static const bool off = vdn_env("VDN_NO_GRAPHS") != nullptr;
  return !off && !comm_active() && g_ctx.stream != 0 && !g_capturing;
}
// `generation` counts the clears: solvers that keep host state next to a graph (mg_nd.hip: the ping-pong state a cycle leaves behind)
// drop it when the generation has moved on
static unsigned long g_graph_generation = 1;
unsigned long graph_generation() { return g_graph_generation; }
void graph_cache_clear() {
  // an exec may still be in flight (fixed-cycle FAC loops replay graphs back to back with no read-back in between), and HIP does not
  // promise the deferred destruction CUDA gives: drain the launch stream first
  if (!g_graphs.empty() && g_ctx.stream) (void)hipStreamSynchronize(g_ctx.stream);
  for (auto &kv : g_graphs) (void)hipGraphExecDestroy(kv.second);
  g_graphs.clear();
  g_graph_generation++;
}
bool graph_replay(unsigned long long key) {
  auto it = g_graphs.find(key);
  if (it == g_graphs.end()) return false;
  HIPCHK(hipGraphLaunch(it->second, g_ctx.stream));
  return true;
}
void graph_begin() {
  REQUIRE(!g_capturing, "graph_begin: nested capture");
  HIPCHK(hipStreamBeginCapture(g_ctx.stream, hipStreamCaptureModeThreadLocal));
  g_capturing = true;
}
void graph_abort() {
  if (!g_capturing) return;
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(g_ctx.stream, &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  g_capturing = false;
}
void graph_end(unsigned long long key) {
  REQUIRE(g_capturing, "graph_end without graph_begin");
  hipGraph_t g = nullptr;
  g_capturing = false;
  HIPCHK(hipStreamEndCapture(g_ctx.stream, &g));
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) vdn_fail("hipGraphInstantiate failed: %s", hipGetErrorString(e));
  if (g_graphs.size() >= 256) graph_cache_clear();        // stale keys (freed arenas, destroyed layouts) do not pile up
  auto old = g_graphs.find(key);                          // a re-capture of a key that still has an exec: destroy, do not leak it
  if (old != g_graphs.end()) { (void)hipStreamSynchronize(g_ctx.stream); (void)hipGraphExecDestroy(old->second); }
  g_graphs[key] = ex;
  HIPCHK(hipGraphLaunch(ex, g_ctx.stream));
}

void solver_check(int rc, const char *what, int iters, double res, double res0, int comp) {
  const bool bad = !(res < HUGE_VAL) || !(res0 < HUGE_VAL);
  if (rc == 0 && !bad) return;
  char cs[32] = ""; if (comp >= 0) snprintf(cs, sizeof cs, " (component %d)", comp);
  if (ctx().prm.abort_on_max_iter)
    vdn_fail("%s%s %s after %d iterations (residual %g, right-hand side %g)", what, cs, bad ? "met a non-finite norm" : "did not converge", iters, res, res0);
  if (ctx().prm.verbose) fprintf(stderr, "varden_amd: %s%s did not converge in %d iterations (res %g / %g)\n", what, cs, iters, res, res0);
}

extern "C" const char *vdn_last_error(void) { return g_err; }
// a HIP error the host application left pending when it called in (VDN_TRY took it off the thread): kept for the host to ask for, noted once per process
static int g_stale_hip_error = 0;
void vdn_note_stale_error(hipError_t e) {
  static bool noted = false;
  g_stale_hip_error = (int)e;
  if (!noted) { noted = true; fprintf(stderr, "varden_amd: note: the caller left a pending HIP error on this thread (%s); cleared at entry (vdn_last_stale_hip_error returns it; further ones are not printed)\n", hipGetErrorString(e)); }
}
extern "C" int vdn_last_stale_hip_error(int clear) { const int e = g_stale_hip_error; if (clear) g_stale_hip_error = 0; return e; }

extern "C" int vdn_init(const vdn_params *prm, int rank, int nranks, int device) {
  VDN_TRY
  env_warn_unknown();
  REQUIRE(prm != nullptr, "vdn_init: null params");
  g_ctx.extruded2d = false;
  REQUIRE(prm->dm == 3 || prm->dm == 2, "vdn_init: dm must be 2 or 3 (got %d)", prm->dm);
  REQUIRE(prm->nscal >= 1 && prm->nscal + 5 <= VDN_MAXCOMP, "vdn_init: bad nscal %d", prm->nscal);
  REQUIRE(prm->visc_coef >= 0.0 && prm->diff_coef >= 0.0, "vdn_init: negative visc_coef / diff_coef");
  REQUIRE(prm->diffusion_type == 1 || prm->diffusion_type == 2, "BAD DIFFUSION TYPE");      // velocity_advance.f90:113
  REQUIRE(prm->slope_order == 0 || prm->slope_order == 2 || prm->slope_order == 4, "bad slope_order");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  REQUIRE(ndev > 0, "vdn_init: no HIP device visible -- the product path has no CPU fallback");
  REQUIRE(device >= 0 && device < ndev, "vdn_init: device %d out of range (%d devices)", device, ndev);
  HIPCHK(hipSetDevice(device));
  VdnCtx &c = g_ctx;
  c.prm = *prm; c.rank = rank; c.nranks = nranks; c.device = device;
  if (!c.d_scal) {
    HIPCHK(hipMalloc((void **)&c.d_scal, 64 * sizeof(double)));
    HIPCHK(hipMalloc((void **)&c.d_hist, 65 * sizeof(double)));
    HIPCHK(hipHostMalloc((void **)&c.h_scal, 72 * sizeof(double), hipHostMallocMapped));      // 64 values + the sequence number of read_scalars
    memset(c.h_scal, 0, 72 * sizeof(double));
    HIPCHK(hipHostGetDevicePointer((void **)&c.h_scal_dev, c.h_scal, 0));
  }
  // our own launch stream: the legacy null stream cannot be captured into a hipGraph, and a second stream next to it could not overlap
  if (!c.own_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c.halo_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c.ev_main, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c.ev_halo, hipEventDisableTiming));
  }
  if (c.stream == 0) c.stream = c.own_stream;
  prof_load();
  c.inited = true;
  VDN_CATCH
}
extern "C" int vdn_finalize(void) {
  VDN_TRY
  VdnCtx &c = g_ctx;
  kept_purge(0);
  arena_destroy(); field_pool_release();
  graph_cache_clear();
  if (c.d_hist) { HIPCHK(hipFree(c.d_hist)); c.d_hist = nullptr; }
  if (c.d_scal) { HIPCHK(hipFree(c.d_scal)); c.d_scal = nullptr; HIPCHK(hipHostFree(c.h_scal)); c.h_scal = nullptr; c.h_scal_dev = nullptr; }
  c.inited = false;
  VDN_CATCH
}
extern "C" int vdn_set_stream(void *s) {            // NULL: back to the library's own stream
  VDN_TRY
  if (g_ctx.inited) HIPCHK(hipStreamSynchronize(g_ctx.stream));
  graph_cache_clear();                                // cached graphs were captured on the old stream
  g_ctx.stream = s ? (hipStream_t)s : g_ctx.own_stream;
  VDN_CATCH
}
extern "C" int vdn_arena_stats(size_t *reserved_bytes, size_t *peak_bytes) { *reserved_bytes = g_ctx.arena_bytes; *peak_bytes = g_ctx.arena_peak; return 0; }
extern "C" int vdn_device_synchronize(void) { VDN_TRY HIPCHK(hipStreamSynchronize(g_ctx.stream)); VDN_CATCH }
extern "C" int vdn_get_params(vdn_params *out) { *out = g_ctx.prm; return 0; }
extern "C" int vdn_set_extruded_2d(int on) { g_ctx.extruded2d = on != 0; return 0; }
extern "C" int vdn_last_step_timing(double *s) { for (int i = 0; i < 5; i++) s[i] = g_ctx.step_sec[i]; return 0; }
extern "C" int vdn_last_solver_stats(int w, int *cyc, double *r0, double *r) {
  if (w < 0 || w > 1) return 1;
  *cyc = g_ctx.solver_cycles[w]; *r0 = g_ctx.solver_res0[w]; *r = g_ctx.solver_res[w]; return 0;
}

// ================================================================================================
// ml_layout
// ================================================================================================
extern "C" int vdn_layout_create(int nlev, const int *rr, const vdn_box *pd, const int *nboxes, const vdn_box *boxes,
                                 const int *owner, const int *pmask, vdn_layout **out) {
  VDN_TRY
  REQUIRE(nlev >= 1, "nlev must be >= 1");
  static unsigned long next_uid = 1;
  vdn_layout *la = new vdn_layout;
  la->uid = next_uid++;
  la->nlev = nlev;
  if (nlev > 1) la->rr.assign(rr, rr + 3 * (nlev - 1));
  la->pd.assign(pd, pd + nlev);
  for (int d = 0; d < 3; d++) la->pmask[d] = pmask ? pmask[d] : 0;
  int off = 0;
  la->boxes.resize(nlev); la->owner.resize(nlev); la->local.resize(nlev);
  for (int l = 0; l < nlev; l++) {
    for (int b = 0; b < nboxes[l]; b++) {
      const vdn_box &bx = boxes[off + b];
      for (int d = 0; d < g_ctx.prm.dm; d++) REQUIRE(bx.hi[d] - bx.lo[d] + 1 >= 4, "box %d of level %d is narrower than 4 cells", b, l);
      if (g_ctx.prm.dm == 2) REQUIRE(bx.lo[2] == 0 && bx.hi[2] == 0, "dm = 2: boxes must have lo(3) = hi(3) = 0");
      la->boxes[l].push_back(bx);
      int ow = owner ? owner[off + b] : 0;
      REQUIRE(ow >= 0 && ow < g_ctx.nranks, "owner %d out of range", ow);
      la->owner[l].push_back(ow);
      if (ow == g_ctx.rank) la->local[l].push_back(b);
    }
    off += nboxes[l];
  }
  *out = la;
  VDN_CATCH
}
extern "C" int vdn_layout_destroy(vdn_layout *la) { VDN_TRY if (la) { graph_cache_clear(); xplan_cache_purge(la->uid); delete la; } VDN_CATCH }
extern "C" int vdn_layout_nlevel(const vdn_layout *la) { return la->nlev; }
extern "C" int vdn_layout_nboxes(const vdn_layout *la, int lev) { return (int)la->boxes[lev].size(); }
extern "C" int vdn_layout_nlocal(const vdn_layout *la, int lev) { return (int)la->local[lev].size(); }
extern "C" int vdn_layout_global_index(const vdn_layout *la, int lev, int i) { return la->local[lev][i]; }
extern "C" int vdn_layout_get_box(const vdn_layout *la, int lev, int g, vdn_box *out) { *out = la->boxes[lev][g]; return 0; }

// ================================================================================================
// bc_tower  (define_bc_tower.f90)
// ================================================================================================
static void build_adv_ell(int p, int d, int dm, int nscal, int *a, int *e) {
  const int press = dm + nscal, extrap = press + 1;
  if (p == VDN_SLIP_WALL) {                       // define_bc_tower.f90:199-207, 291-297
    for (int c = 0; c < dm; c++) a[c] = VDN_HOEXTRAP;
    a[d] = VDN_EXT_DIR;
    for (int n = 0; n < nscal; n++) a[dm + n] = VDN_HOEXTRAP;
    a[press] = VDN_FOEXTRAP; a[extrap] = VDN_FOEXTRAP;
    for (int c = 0; c < dm; c++) e[c] = VDN_BC_NEU;
    e[d] = VDN_BC_DIR;
    for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
    e[press] = VDN_BC_NEU;
  } else if (p == VDN_NO_SLIP_WALL) {             // 209-216, 299-304
    for (int c = 0; c < dm; c++) a[c] = VDN_EXT_DIR;
    for (int n = 0; n < nscal; n++) a[dm + n] = VDN_HOEXTRAP;
    a[press] = VDN_FOEXTRAP; a[extrap] = VDN_FOEXTRAP;
    for (int c = 0; c < dm; c++) e[c] = VDN_BC_DIR;
    for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
    e[press] = VDN_BC_NEU;
  } else if (p == VDN_INLET) {                    // 218-225, 306-311
    for (int c = 0; c < dm; c++) a[c] = VDN_EXT_DIR;
    for (int n = 0; n < nscal; n++) a[dm + n] = VDN_EXT_DIR;
    a[press] = VDN_FOEXTRAP; a[extrap] = VDN_FOEXTRAP;
    for (int c = 0; c < dm; c++) e[c] = VDN_BC_DIR;
    for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_DIR;
    e[press] = VDN_BC_NEU;
  } else if (p == VDN_OUTLET) {                   // 227-234, 313-318
    for (int c = 0; c < dm; c++) a[c] = VDN_FOEXTRAP;
    for (int n = 0; n < nscal; n++) a[dm + n] = VDN_FOEXTRAP;
    a[press] = VDN_EXT_DIR; a[extrap] = VDN_FOEXTRAP;
    for (int c = 0; c < dm; c++) e[c] = VDN_BC_NEU;
    for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
    e[press] = VDN_BC_DIR;
  } else if (p == VDN_SYMMETRY) {                 // 236-244, 320-326
    for (int c = 0; c < dm; c++) a[c] = VDN_REFLECT_EVEN;
    a[d] = VDN_REFLECT_ODD;
    for (int n = 0; n < nscal; n++) a[dm + n] = VDN_REFLECT_EVEN;
    a[press] = VDN_EXT_DIR; a[extrap] = VDN_REFLECT_EVEN;
    for (int c = 0; c < dm; c++) e[c] = VDN_BC_NEU;
    e[d] = VDN_BC_DIR;
    for (int n = 0; n < nscal; n++) e[dm + n] = VDN_BC_NEU;
    e[press] = VDN_BC_NEU;
  } else if (p == VDN_PERIODIC) {                 // 328-334 (ell only)
    for (int c = 0; c < dm + nscal + 1; c++) e[c] = VDN_BC_PER;
  }
}

extern "C" int vdn_bc_tower_create(const vdn_layout *la, const int *phys_bc, vdn_bc_tower **out) {
  VDN_TRY
  vdn_bc_tower *b = new vdn_bc_tower;
  { static unsigned long next_serial = 0; b->serial = ++next_serial; }
  b->la = la; b->dm = g_ctx.prm.dm; b->nscal = g_ctx.prm.nscal;
  b->ncomp_adv = b->dm + b->nscal + 2; b->ncomp_ell = b->dm + b->nscal + 1;
  for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++) b->domain_bc[d][s] = (d < b->dm) ? phys_bc[d * 2 + s] : VDN_INTERIOR;   // dm = 2: no z faces
  b->phys.resize(la->nlev); b->adv.resize(la->nlev); b->ell.resize(la->nlev);
  for (int l = 0; l < la->nlev; l++) {
    int ng = (int)la->local[l].size() + 1;
    b->phys[l].resize(ng); b->adv[l].resize(ng); b->ell[l].resize(ng);
    for (int g = 0; g < ng; g++) {
      BoxP &bp = b->phys[l][g];
      vdn_box bx = (g == 0) ? la->pd[l] : la->boxes[l][la->local[l][g - 1]];
      for (int d = 0; d < 3; d++) {
        bp.lo[d] = bx.lo[d]; bp.hi[d] = bx.hi[d];
        // phys_bc_level_build, define_bc_tower.f90:129-156
        bp.phys[d][0] = (bx.lo[d] == la->pd[l].lo[d]) ? b->domain_bc[d][0] : VDN_INTERIOR;
        bp.phys[d][1] = (bx.hi[d] == la->pd[l].hi[d]) ? b->domain_bc[d][1] : VDN_INTERIOR;
      }
      b->adv[l][g].assign(6 * b->ncomp_adv, VDN_INTERIOR);
      b->ell[l][g].assign(6 * b->ncomp_ell, VDN_BC_INT);
      for (int d = 0; d < 3; d++) for (int s = 0; s < 2; s++)
        build_adv_ell(bp.phys[d][s], d, b->dm, b->nscal, &b->adv[l][g][(d * 2 + s) * b->ncomp_adv], &b->ell[l][g][(d * 2 + s) * b->ncomp_ell]);
    }
  }
  *out = b;
  VDN_CATCH
}
extern "C" int vdn_bc_tower_destroy(vdn_bc_tower *b) { delete b; return 0; }
extern "C" int vdn_bc_tower_phys(const vdn_bc_tower *b, int lev, int grid, int dir, int side) { return b->phys[lev][grid].phys[dir][side]; }
extern "C" int vdn_bc_tower_adv(const vdn_bc_tower *b, int lev, int grid, int dir, int side, int comp) { return b->adv_bc(lev, grid, dir, side, comp); }
extern "C" int vdn_bc_tower_ell(const vdn_bc_tower *b, int lev, int grid, int dir, int side, int comp) { return b->ell_bc(lev, grid, dir, side, comp); }

BoxP make_boxp(const vdn_multifab *mf, int i, const vdn_bc_tower *bct) {
  BoxP bp;
  if (bct) bp = bct->phys[mf->lev][i + 1];
  else for (int d = 0; d < 3; d++) { bp.phys[d][0] = bp.phys[d][1] = VDN_INTERIOR; }
  for (int d = 0; d < 3; d++) { bp.lo[d] = mf->vbox[i].lo[d]; bp.hi[d] = mf->vbox[i].hi[d]; }
  return bp;
}

// ================================================================================================
// multifab
// ================================================================================================
static void mf_layout_fabs(vdn_multifab *mf, size_t *total_doubles) {
  const vdn_layout *la = mf->la;
  size_t off = 0;
  mf->fabs.clear(); mf->vbox.clear();
  for (int g : la->local[mf->lev]) {
    const vdn_box &bx = la->boxes[mf->lev][g];
    FV f;
    f.a0 = bx.lo[0] - mf->ng; f.a1 = bx.lo[1] - mf->ng; f.a2 = bx.lo[2] - mf->ng;
    f.n0 = bx.hi[0] - bx.lo[0] + 1 + mf->nodal[0] + 2 * mf->ng;
    f.n1 = bx.hi[1] - bx.lo[1] + 1 + mf->nodal[1] + 2 * mf->ng;
    f.n2 = bx.hi[2] - bx.lo[2] + 1 + mf->nodal[2] + 2 * mf->ng;
    f.sc = (long)f.n0 * f.n1 * f.n2;
    f.p = (double *)(uintptr_t)(off * sizeof(double));   // offset for now
    off += (size_t)f.sc * mf->nc;
    off = (off + 31) & ~(size_t)31;                      // 256-byte aligned fabs
    mf->fabs.push_back(f); mf->vbox.push_back(bx);
  }
  *total_doubles = off;
}

extern "C" int vdn_multifab_create(const vdn_layout *la, int lev, int nc, int ng, const int *nodal, vdn_multifab **out) {
  VDN_TRY
  REQUIRE(la && lev >= 0 && lev < la->nlev, "vdn_multifab_create: bad level");
  REQUIRE(nc >= 1 && ng >= 0, "vdn_multifab_create: bad nc/ng");
  vdn_multifab *mf = new vdn_multifab;
  mf->la = la; mf->lev = lev; mf->nc = nc; mf->ng = ng;
  for (int d = 0; d < 3; d++) mf->nodal[d] = nodal ? (nodal[d] != 0) : 0;
  size_t tot; mf_layout_fabs(mf, &tot);
  mf->bytes = std::max<size_t>(tot, 1) * sizeof(double);
  mf->base = (double *)field_alloc(mf->bytes, la->nlev > 1);
  for (auto &f : mf->fabs) f.p = (double *)((char *)mf->base + (uintptr_t)f.p);
  HIPCHK(hipMemsetAsync(mf->base, 0, mf->bytes, g_ctx.stream));
  *out = mf;
  VDN_CATCH
}

vdn_multifab *mf_temp(const vdn_layout *la, int lev, int nc, int ng, int face_dir, bool fill, double val) {
  vdn_multifab *mf = new vdn_multifab;
  mf->la = la; mf->lev = lev; mf->nc = nc; mf->ng = ng; mf->owns = false;
  for (int d = 0; d < 3; d++) mf->nodal[d] = (face_dir == 3) ? 1 : (face_dir == d ? 1 : 0);
  size_t tot; mf_layout_fabs(mf, &tot);
  mf->bytes = std::max<size_t>(tot, 1) * sizeof(double);
  mf->base = (double *)arena_alloc(mf->bytes);
  for (auto &f : mf->fabs) f.p = (double *)((char *)mf->base + (uintptr_t)f.p);
  {   // (VDN_PHASE_HASH: every temporary starts from zeros, so that entries nobody writes -- and nobody reads -- do not differ from process to process in the checksums)
    static const bool clr = vdn_env("VDN_PHASE_HASH") && atoi(vdn_env("VDN_PHASE_HASH")) != 0;
    if (clr) HIPCHK(hipMemsetAsync(mf->base, 0, mf->bytes, g_ctx.stream));
  }
  if (fill) mf_setval(mf, val, 0, nc, true);
  g_temp_mfs.push_back(mf);
  return mf;
}
void mf_temp_free(vdn_multifab *mf) {
  for (size_t i = g_temp_mfs.size(); i-- > 0;) if (g_temp_mfs[i] == mf) { g_temp_mfs.erase(g_temp_mfs.begin() + (long)i); break; }      // (freed in reverse order of creation: found at the end)
  delete mf;
}


// descriptor sets kept across calls (vdn_internal.h)
static std::map<unsigned long long, KeptSet> g_kept;
bool kept_sets_enabled() { static const bool on = !(vdn_env("VDN_KEEP_SETS") && atoi(vdn_env("VDN_KEEP_SETS")) == 0); return on; }
// VDN_KEEP_OFF: a mask of families switched off one by one (1 the generic launch_batched_kept sets, 2 create_umac_grown, 4 the composite cell-centred solve, 8 the nodal prolongation)
void dbg_sync(int bit) { static const int m = vdn_env("VDN_SYNC_POINTS") ? atoi(vdn_env("VDN_SYNC_POINTS")) : 0; if (m & bit) HIPCHK(hipDeviceSynchronize()); }
bool kept_family_enabled(int fam) { static const int off = vdn_env("VDN_KEEP_OFF") ? atoi(vdn_env("VDN_KEEP_OFF")) : 0; return kept_sets_enabled() && !(off & fam); }
KeptSet *kept_find(unsigned long long key) { auto it = g_kept.find(key); return it == g_kept.end() ? nullptr : &it->second; }
static void kept_free(KeptSet &k) { if (k.d_args) HIPCHK(hipFree(k.d_args)); if (k.d_start) HIPCHK(hipFree(k.d_start)); k.d_args = nullptr; k.d_start = nullptr; }
static KeeperMem *g_keeper = nullptr;
void keeper_begin(KeeperMem *m) { REQUIRE(!g_keeper, "kept descriptor sets: nested keeper"); g_keeper = m; }
void keeper_end() { g_keeper = nullptr; }
void keeper_free(KeeperMem *m) { for (void *p : m->chunks) HIPCHK(hipFree(p)); m->chunks.clear(); m->cur = nullptr; m->left = 0; }
void *set_alloc(size_t bytes) {
  if (!g_keeper) return arena_alloc(bytes);
  KeeperMem &m = *g_keeper;
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (need > m.left) {
    const size_t chunk = std::max<size_t>(need, (size_t)4 << 20);
    void *p = nullptr; HIPCHK(hipMalloc(&p, chunk));
    m.chunks.push_back(p); m.cur = (char *)p; m.left = chunk;
  }
  void *r = m.cur; m.cur += need; m.left -= need;
  return r;
}
void kept_purge(unsigned long uid) {
  if (ctx().inited) HIPCHK(hipStreamSynchronize(ctx().stream));
  mlcc_kept_purge(uid); mlnd_kept_purge(uid);
  if (g_kept.empty()) return;
  for (auto it = g_kept.begin(); it != g_kept.end();) { if (uid == 0 || it->second.uid == uid) { kept_free(it->second); it = g_kept.erase(it); } else ++it; }
}
// the size bounds of the three tables (plain sets here, the groups of the composite solves in amr.hip / mg_nd.hip); VDN_KEPT_BOUND shrinks them for
// the eviction test (tests/test_amr_gpu.py)
int kept_bound(int dflt) { static const int env = vdn_env("VDN_KEPT_BOUND") ? atoi(vdn_env("VDN_KEPT_BOUND")) : 0; return env > 0 ? env : dflt; }
KeptSet *kept_store(unsigned long long key, unsigned long uid, const void *args, size_t arg_bytes, const int *start, int nbox, int tot) {
  // Bounded (temporaries that wander through the arena), rebuilt on demand.  This runs INSIDE the composite solves (launch_batched_kept), whose own
  // groups (MLCCKept, NdProKept) are bound to the running solve: the bound drops the plain entries only -- nobody holds a KeptSet across a store --
  // and the groups are bounded at the entry of their solves, before any of them is bound
  if ((int)g_kept.size() >= kept_bound(4096)) {
    HIPCHK(hipStreamSynchronize(ctx().stream));
    for (auto &kv : g_kept) kept_free(kv.second);
    g_kept.clear();
  }
  KeptSet k; k.nbox = nbox; k.tot = tot; k.uid = uid;
  if (nbox > 0) {
    HIPCHK(hipMalloc(&k.d_args, arg_bytes)); HIPCHK(hipMalloc((void **)&k.d_start, sizeof(int) * nbox));
    upload_staged(k.d_args, args, arg_bytes); upload_staged(k.d_start, start, sizeof(int) * nbox);
  }
  return &(g_kept[key] = k);
}
// device scratch for the descriptor arrays of one-off batched launches: a ring; reuse is safe because uploads and launches are
// ordered on the one launch stream
void *desc_scratch(size_t bytes) {
  static char *ring = nullptr; static size_t cap = 0, head = 0;
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (!ring || need > cap) {
    if (ring) { HIPCHK(hipStreamSynchronize(ctx().stream)); HIPCHK(hipFree(ring)); }
    cap = std::max<size_t>((size_t)64 << 20, 2 * need); head = 0;
    HIPCHK(hipMalloc((void **)&ring, cap));
  }
  if (head + need > cap) head = 0;
  void *p = ring + head; head += need;
  return p;
}
// small host -> device uploads (descriptor arrays of the batched launches) through a pinned ring: the copy is asynchronous on
// the launch stream, the caller's buffer may be freed at once; the ring is drained (stream sync) when it wraps
void upload_staged(void *dst, const void *src, size_t bytes) {
  static char *ring = nullptr; static size_t cap = 0, head = 0;
  VdnCtx &c = ctx();
  if (bytes == 0) return;
  if (!ring || bytes > cap) {
    if (ring) { HIPCHK(hipStreamSynchronize(c.stream)); HIPCHK(hipHostFree(ring)); }
    cap = std::max<size_t>((size_t)32 << 20, 2 * bytes); head = 0;
    HIPCHK(hipHostMalloc((void **)&ring, cap, hipHostMallocDefault));
  }
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (head + need > cap) { HIPCHK(hipStreamSynchronize(c.stream)); head = 0; }
  memcpy(ring + head, src, bytes);
  HIPCHK(hipMemcpyAsync(dst, ring + head, bytes, hipMemcpyHostToDevice, c.stream));
  head += need;
  dbg_sync(2);
}

extern "C" int vdn_multifab_destroy(vdn_multifab *mf) {
  VDN_TRY
  if (mf) { if (mf->owns && mf->base) { HIPCHK(hipStreamSynchronize(g_ctx.stream)); field_free(mf->base); } delete mf; }
  VDN_CATCH
}
extern "C" int vdn_multifab_nfabs(const vdn_multifab *mf) { return mf->nfabs(); }
extern "C" int vdn_multifab_ncomp(const vdn_multifab *mf) { return mf->nc; }
extern "C" int vdn_multifab_nghost(const vdn_multifab *mf) { return mf->ng; }
extern "C" int vdn_multifab_get_box(const vdn_multifab *mf, int i, vdn_box *out) { *out = mf->vbox[i]; return 0; }
// dm = 2: the host sees the BoxLib 2-D layout p(lo1-ng:hi1+ng, lo2-ng:hi2+ng, nc) = plane k = 0 of every component
static bool host_2d(const vdn_multifab *mf) { return g_ctx.prm.dm == 2; (void)mf; }
extern "C" long vdn_multifab_fab_size(const vdn_multifab *mf, int i) {
  if (host_2d(mf)) return (long)mf->fabs[i].n0 * mf->fabs[i].n1 * mf->nc;
  return mf->fab_size(i);
}
static void copy_plane0(const vdn_multifab *mf, int i, double *host, bool to_host) {
  const FV &f = mf->fabs[i];
  const long plane = (long)f.n0 * f.n1;
  for (int c = 0; c < mf->nc; c++) {
    double *dev = f.p + f.sc * c + plane * (0 - f.a2);
    if (to_host) HIPCHK(hipMemcpyAsync(host + plane * c, dev, plane * sizeof(double), hipMemcpyDeviceToHost, g_ctx.stream));
    else HIPCHK(hipMemcpyAsync(dev, host + plane * c, plane * sizeof(double), hipMemcpyHostToDevice, g_ctx.stream));
  }
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
}
// Handing out a raw device pointer is where the host starts touching the data with its OWN stream: when the launch stream is the
// library's private (non-blocking) one nothing would order the two, so the call drains it first.  With a caller-provided stream
// (vdn_set_stream) ordering is the caller's, as for any work on their stream.
extern "C" int vdn_multifab_dataptr(const vdn_multifab *mf, int i, double **dev) {
  VDN_TRY
  if (g_ctx.inited && g_ctx.stream == g_ctx.own_stream) HIPCHK(hipStreamSynchronize(g_ctx.stream));
  *dev = mf->fabs[i].p;
  VDN_CATCH
}
extern "C" int vdn_multifab_copy_to_host(const vdn_multifab *mf, int i, double *host) {
  VDN_TRY
  if (host_2d(mf)) { copy_plane0(mf, i, host, true); return 0; }
  HIPCHK(hipMemcpyAsync(host, mf->fabs[i].p, mf->fab_size(i) * sizeof(double), hipMemcpyDeviceToHost, g_ctx.stream));
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  VDN_CATCH
}
extern "C" int vdn_multifab_copy_from_host(vdn_multifab *mf, int i, const double *host) {
  VDN_TRY
  if (host_2d(mf)) { copy_plane0(mf, i, const_cast<double *>(host), false); return 0; }
  HIPCHK(hipMemcpyAsync(mf->fabs[i].p, host, mf->fab_size(i) * sizeof(double), hipMemcpyHostToDevice, g_ctx.stream));
  HIPCHK(hipStreamSynchronize(g_ctx.stream));
  VDN_CATCH
}

// ---- setval / copy / norm ----------------------------------------------------------------------
__global__ void k_setval(FV f, Range3 r, int comp, int nc, double val) {
  THREAD_IJK(r)
  if (!in_range) return;
  for (int c = comp; c < comp + nc; c++) fv_at(f, i, j, k, c) = val;
}
static Range3 fab_range(const vdn_multifab *mf, int i, int grow) {   // valid (incl. nodal) grown by `grow`
  Range3 r;
  for (int d = 0; d < 3; d++) { r.lo[d] = mf->vbox[i].lo[d] - grow; r.hi[d] = mf->vbox[i].hi[d] + mf->nodal[d] + grow; }
  return r;
}
struct SetvalB { Range3 r; int g[3]; static constexpr int planes_per_wg = 8; FV f; int comp, nc; double val;
  static __device__ double body(const SetvalB &q, int i, int j, int k, int) { for (int c = q.comp; c < q.comp + q.nc; c++) fv_at(q.f, i, j, k, c) = q.val; return 0.0; } };
struct CopyB { Range3 r; int g[3]; FV d, s; int dc, sc, nc;
  static __device__ double body(const CopyB &q, int i, int j, int k, int) { for (int c = 0; c < q.nc; c++) fv_at(q.d, i, j, k, q.dc + c) = fv_get(q.s, i, j, k, q.sc + c); return 0.0; } };
// a run of whole components of one fab is contiguous (component slowest): unit-stride stores over the allocation instead of (i, j, k) threads over rows of
// 262 entries -- a 262^3 component in 27 instead of 39-50 us (five such fills open every step: mac_rhs, rhohalf, umac x 3, advance_timestep.f90:68-78)
__global__ void __launch_bounds__(256) k_fill_flat(double *__restrict__ p, long n, double val) {
  const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (long i = i0; i < n; i += (long)gridDim.x * 512) { p[i] = val; if (i + 1 < n) p[i + 1] = val; }
}
void mf_setval(vdn_multifab *mf, double val, int comp, int nc, bool all) {
  // zero over everything the multifab owns: one memset of its allocation (the fabs are contiguous)
  if (val == 0.0 && all && comp == 0 && nc == mf->nc && mf->nfabs() > 1) { HIPCHK(hipMemsetAsync(mf->base, 0, mf->bytes, g_ctx.stream)); return; }
  if (mf->nfabs() == 1 && all) {
    const FV &f = mf->fabs[0];
    const long n = (long)nc * f.sc;
    if (n > 0) hipLaunchKernelGGL(k_fill_flat, dim3((unsigned)std::min<long>((n + 511) / 512, 1 << 16)), dim3(256), 0, g_ctx.stream, f.p + (long)comp * f.sc, n, val);
    return;
  }
  if (mf->nfabs() == 1) {
    Range3 r = fab_range(mf, 0, all ? mf->ng : 0);
    hipLaunchKernelGGL(k_setval, grid_for(r), dim3(64, 4, 1), 0, g_ctx.stream, mf->fabs[0], r, comp, nc, val);
    return;
  }
  std::vector<SetvalB> v;
  for (int i = 0; i < mf->nfabs(); i++) { SetvalB q; q.r = fab_range(mf, i, all ? mf->ng : 0); q.f = mf->fabs[i]; q.comp = comp; q.nc = nc; q.val = val; v.push_back(q); }
  launch_batched(v, 0, (double *)nullptr, 0, g_ctx.stream);
}
extern "C" int vdn_multifab_setval(vdn_multifab *mf, double val, int comp, int nc, int all) {
  VDN_TRY
  REQUIRE(comp >= 0 && comp + nc <= mf->nc, "setval: component range");
  mf_setval(mf, val, comp, nc, all != 0);
  VDN_CATCH
}

__global__ void k_copy(FV d, int dc, FV s, int scomp, int nc, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  for (int c = 0; c < nc; c++) fv_at(d, i, j, k, dc + c) = fv_get(s, i, j, k, scomp + c);
}
void mf_copy(vdn_multifab *dst, int dcomp, const vdn_multifab *src, int scomp, int nc, int ng) {
  REQUIRE(dst->nfabs() == src->nfabs(), "copy_c: layouts differ");
  REQUIRE(ng <= dst->ng && ng <= src->ng, "copy_c: ng too large");
  if (dst->nfabs() == 1) {
    Range3 r = fab_range(dst, 0, ng);
    hipLaunchKernelGGL(k_copy, grid_for(r), dim3(64, 4, 1), 0, g_ctx.stream, dst->fabs[0], dcomp, src->fabs[0], scomp, nc, r);
    return;
  }
  std::vector<CopyB> v;
  for (int i = 0; i < dst->nfabs(); i++) { CopyB q; q.r = fab_range(dst, i, ng); q.d = dst->fabs[i]; q.s = src->fabs[i]; q.dc = dcomp; q.sc = scomp; q.nc = nc; v.push_back(q); }
  launch_batched(v, 0, (double *)nullptr, 0, g_ctx.stream);
}
extern "C" int vdn_multifab_copy_c(vdn_multifab *dst, int dcomp, const vdn_multifab *src, int scomp, int nc, int ng) {
  VDN_TRY mf_copy(dst, dcomp, src, scomp, nc, ng); VDN_CATCH
}

__global__ void k_absmax(FV f, Range3 r, int comp, int nc, double *out) {
  REDUCE_IJ(r)
  double v = 0.0;
  if (in_ij) REDUCE_KLOOP(r) for (int c = comp; c < comp + nc; c++) v = nmax(v, fabs(fv_get(f, i, j, k, c)));
  block_atomic_max(out, v);
}
__global__ void k_minmax(FV f, Range3 r, int comp, double *out /* [0]=max(-x) shifted, [1]=max(x) shifted */, double shift) {
  REDUCE_IJ(r)
  double a = 0.0, b = 0.0;   // max of (shift - x) and (x + shift), both non-negative when |x| <= shift
  if (in_ij) REDUCE_KLOOP(r) { double x = fv_get(f, i, j, k, comp); a = fmax(a, shift - x); b = fmax(b, x + shift); }
  block_atomic_max(out, a); block_atomic_max(out + 1, b);
}
double mf_norm_inf(const vdn_multifab *mf, int comp, int nc) { return mf_norm_inf_grown(mf, comp, nc, 0); }
double mf_norm_inf_grown(const vdn_multifab *mf, int comp, int nc, int grow) {       // over the valid region grown by `grow` ghost cells
  VdnCtx &c = g_ctx;
  REQUIRE(grow >= 0 && grow <= mf->ng, "norm_inf: more ghost cells asked for than the multifab has");
  HIPCHK(hipMemsetAsync(c.d_scal, 0, sizeof(double), c.stream));
  for (int i = 0; i < mf->nfabs(); i++) {
    Range3 r = fab_range(mf, i, grow);
    hipLaunchKernelGGL(k_absmax, reduce_grid(r), dim3(64, 4, 1), 0, c.stream, mf->fabs[i], r, comp, nc, c.d_scal);
  }
  comm_allreduce_max_dev(c.d_scal, 1);        // FBoxLib norm_inf is a global (all-rank) norm
  return read_scalar1(c.d_scal);
}
// max over the valid points of three pairs of multifabs (component 0) of max(a / b, b / a): how far two sets of face coefficients are apart
// (amr.hip: the composite MAC solve's choice of the level-0 V-cycle's coefficients); NaN -> +inf; all ranks
__global__ void k_maxratio(FV a, FV b, Range3 r, double *out) {
  REDUCE_IJ(r)
  double v = 0.0;
  if (in_ij) REDUCE_KLOOP(r) { const double x = fv_get(a, i, j, k, 0), y = fv_get(b, i, j, k, 0); const double r1 = x / y, r2 = y / x; v = nmax(v, r1 > r2 ? r1 : r2); }
  block_atomic_max(out, v);
}
double mf_max_ratio3(vdn_multifab *const *a, vdn_multifab *const *b) {
  VdnCtx &c = g_ctx;
  HIPCHK(hipMemsetAsync(c.d_scal, 0, sizeof(double), c.stream));
  for (int d = 0; d < 3; d++) {
    REQUIRE(a[d]->nfabs() == b[d]->nfabs(), "max_ratio: layouts differ");
    for (int i = 0; i < a[d]->nfabs(); i++) {
      Range3 r = fab_range(a[d], i, 0);
      hipLaunchKernelGGL(k_maxratio, reduce_grid(r), dim3(64, 4, 1), 0, c.stream, a[d]->fabs[i], b[d]->fabs[i], r, c.d_scal);
    }
  }
  comm_allreduce_max_dev(c.d_scal, 1);
  return read_scalar1(c.d_scal);
}
extern "C" int vdn_multifab_norm_inf(const vdn_multifab *mf, int comp, int nc, double *out) {
  VDN_TRY *out = mf_norm_inf(mf, comp, nc); VDN_CATCH
}
extern "C" int vdn_multifab_min_max(const vdn_multifab *mf, int comp, double *mn, double *mx) {
  VDN_TRY
  VdnCtx &c = g_ctx;
  double amax = mf_norm_inf(mf, comp, 1);
  double shift = amax;     // x + shift >= 0 and shift - x >= 0
  HIPCHK(hipMemsetAsync(c.d_scal, 0, 2 * sizeof(double), c.stream));
  for (int i = 0; i < mf->nfabs(); i++) {
    Range3 r = fab_range(mf, i, 0);
    hipLaunchKernelGGL(k_minmax, reduce_grid(r), dim3(64, 4, 1), 0, c.stream, mf->fabs[i], r, comp, c.d_scal, shift);
  }
  comm_allreduce_max_dev(c.d_scal, 2);
  const double *h2 = read_scalars(c.d_scal, 2);
  *mn = shift - h2[0]; *mx = h2[1] - shift;
  VDN_CATCH
}

// ================================================================================================
// multifab_physbc  (multifab_physbc.f90:238-561)
// ================================================================================================
struct PhysArgs {
  int lo[3], hi[3], ng;
  int d, s;            // face
  int t1, t2;          // transverse directions (t1 < t2)
  int r1lo, r1hi, r2lo, r2hi;
  int bc; double ev;   // bc code, EXT_DIR value
  int comp;
};
DEVI void physbc_cell(const FV &f, const PhysArgs &A, int b1, int b2) {
  int q[3]; q[A.t1] = b1; q[A.t2] = b2;
  const int edge = A.s == 0 ? A.lo[A.d] : A.hi[A.d];
  const int in = A.s == 0 ? 1 : -1;
  double v = 0.0;
  if (A.bc == VDN_FOEXTRAP) { q[A.d] = edge; v = fv_get(f, q[0], q[1], q[2], A.comp); }
  else if (A.bc == VDN_HOEXTRAP) {
    q[A.d] = edge;          double s0 = fv_get(f, q[0], q[1], q[2], A.comp);
    q[A.d] = edge + in;     double s1 = fv_get(f, q[0], q[1], q[2], A.comp);
    q[A.d] = edge + 2 * in; double s2 = fv_get(f, q[0], q[1], q[2], A.comp);
    v = (15.0 * s0 - 10.0 * s1 + 3.0 * s2) * 0.125;
  } else if (A.bc == VDN_EXT_DIR) v = A.ev;
  for (int g = 1; g <= A.ng; g++) {
    if (A.bc == VDN_REFLECT_EVEN || A.bc == VDN_REFLECT_ODD) {
      q[A.d] = edge + in * (g - 1);
      v = fv_get(f, q[0], q[1], q[2], A.comp);
      if (A.bc == VDN_REFLECT_ODD) v = -v;
    }
    q[A.d] = edge - in * g;
    fv_at(f, q[0], q[1], q[2], A.comp) = v;
  }
}
__global__ void k_physbc(FV f, PhysArgs A) {
  int b1 = A.r1lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  int b2 = A.r2lo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
  if (b1 > A.r1hi || b2 > A.r2hi) return;
  physbc_cell(f, A, b1, b2);
}
// one box: the faces of one direction (two sides x the components) in ONE launch, descriptors as kernel arguments, blockIdx.z = face
constexpr int PHYS_MULTI = 8;
struct PhysMulti { PhysArgs A[PHYS_MULTI]; };
__global__ void k_physbc_multi(FV f, PhysMulti M) {
  const PhysArgs &A = M.A[blockIdx.z];
  int b1 = A.r1lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  int b2 = A.r2lo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
  if (b1 > A.r1hi || b2 > A.r2hi) return;
  physbc_cell(f, A, b1, b2);
}
// all faces of one direction on every box and component of a level in one launch: the batch runs over (t1, t2, 0)
struct PhysB { Range3 r; int g[3]; FV f; PhysArgs A;
  static constexpr bool in_constant = true;       // PhysArgs is indexed by direction / side: by value it went to scratch (168 bytes per lane)
  static __device__ double body(const PhysB &q, int i, int j, int, int) { physbc_cell(q.f, q.A, i, j); return 0.0; } };

static bool extdir_value(int icomp1, int d, int s, double *v) {
  const vdn_params &p = g_ctx.prm;
  if (p.dm == 2) {                // multifab_physbc.f90:96-99
    switch (icomp1) { case 1: *v = p.u_bc[d][s]; return true; case 2: *v = p.v_bc[d][s]; return true; case 3: *v = p.rho_bc[d][s]; return true; case 4: *v = p.trac_bc[d][s]; return true; }
    return false;
  }
  switch (icomp1) {               // multifab_physbc.f90:282-287
    case 1: *v = p.u_bc[d][s]; return true;
    case 2: *v = p.v_bc[d][s]; return true;
    case 3: *v = p.w_bc[d][s]; return true;
    case 4: *v = p.rho_bc[d][s]; return true;
    case 5: *v = p.trac_bc[d][s]; return true;
  }
  return false;
}

void mf_physbc(vdn_multifab *mf, int scomp, int bccomp, int nc, const vdn_bc_tower *bct, bool same_boundary) {
  Prof prof_("multifab_physbc");
  if (mf->ng == 0) return;
  REQUIRE(!mf->nodal[0] && !mf->nodal[1] && !mf->nodal[2], "physbc on a nodal multifab");
  // a direction reads the ghost cells the directions before it wrote, boxes and components are independent:
  // levels of several boxes run one batched launch per direction
  const bool batch = mf->nfabs() > 1;
  std::vector<PhysB> pb;
  for (int d = 0; d < 3; d++) {
   pb.clear();
   PhysMulti pm; int npm = 0, gx = 0, gy = 0;
   for (int i = 0; i < mf->nfabs(); i++) for (int c = 0; c < nc; c++) {
    const int bcc = same_boundary ? bccomp : bccomp + c;
    REQUIRE(bcc < bct->ncomp_adv, "physbc: bc component %d out of range", bcc);
    int bc[3][2];
    for (int dd = 0; dd < 3; dd++) for (int s = 0; s < 2; s++) bc[dd][s] = bct->adv_bc(mf->lev, i + 1, dd, s, bcc);
    const int *lo = mf->vbox[i].lo, *hi = mf->vbox[i].hi;
    const int ng = mf->ng;
    for (int s = 0; s < 2; s++) {
      int b = bc[d][s];
      if (b == VDN_INTERIOR) continue;
      PhysArgs A;
      for (int t = 0; t < 3; t++) { A.lo[t] = lo[t]; A.hi[t] = hi[t]; }
      A.ng = ng; A.d = d; A.s = s; A.bc = b; A.ev = 0.0; A.comp = scomp + c;
      A.t1 = (d == 0) ? 1 : 0; A.t2 = (d == 2) ? 1 : 2;
      if (b == VDN_EXT_DIR) { if (!extdir_value(bcc + 1, d, s, &A.ev)) continue; }
      else REQUIRE(b == VDN_FOEXTRAP || b == VDN_HOEXTRAP || b == VDN_REFLECT_EVEN || b == VDN_REFLECT_ODD,
                   "physbc: bc(%d,%d) = %d NOT YET SUPPORTED", d + 1, s + 1, b);
      // transverse ranges (254-276): EXT_DIR covers everything; otherwise directions after d skip
      // their ghost layers on physical sides
      int rlo[3], rhi[3];
      for (int t = 0; t < 3; t++) {
        int nlo = ng, nhi = ng;
        if (b != VDN_EXT_DIR && t > d) { nlo = (bc[t][0] == VDN_INTERIOR) ? ng : 0; nhi = (bc[t][1] == VDN_INTERIOR) ? ng : 0; }
        rlo[t] = lo[t] - nlo; rhi[t] = hi[t] + nhi;
      }
      A.r1lo = rlo[A.t1]; A.r1hi = rhi[A.t1]; A.r2lo = rlo[A.t2]; A.r2hi = rhi[A.t2];
      if (batch) { PhysB q; q.r.lo[0] = A.r1lo; q.r.hi[0] = A.r1hi; q.r.lo[1] = A.r2lo; q.r.hi[1] = A.r2hi; q.r.lo[2] = q.r.hi[2] = 0; q.f = mf->fabs[i]; q.A = A; pb.push_back(q); continue; }
      if (npm == PHYS_MULTI) { hipLaunchKernelGGL(k_physbc_multi, dim3(gx, gy, npm), dim3(64, 4, 1), 0, g_ctx.stream, mf->fabs[0], pm); npm = gx = gy = 0; }
      pm.A[npm++] = A; gx = std::max(gx, (A.r1hi - A.r1lo + 64) / 64); gy = std::max(gy, (A.r2hi - A.r2lo + 4) / 4);
    }
   }
   if (batch) launch_batched(pb, 0, (double *)nullptr, 0, g_ctx.stream);
   else if (npm) hipLaunchKernelGGL(k_physbc_multi, dim3(gx, gy, npm), dim3(64, 4, 1), 0, g_ctx.stream, mf->fabs[0], pm);
  }
}
extern "C" int vdn_multifab_physbc(vdn_multifab *mf, int scomp, int bccomp, int nc, const vdn_bc_tower *bct) {
  VDN_TRY mf_physbc(mf, scomp, bccomp, nc, bct, false); VDN_CATCH
}

void mf_restrict_and_fill(vdn_multifab *mf, int icomp, int bcomp, int nc, bool same_boundary, const vdn_bc_tower *bct) {
  mf_fill_boundary(mf);
  mf_physbc(mf, icomp, bcomp, nc, bct, same_boundary);
}
