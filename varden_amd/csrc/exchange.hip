// exchange.hip -- ghost-point exchange (FBoxLib multifab_fill_boundary) and the rank-to-rank
// transport of the MI355X-native VARDEN hot path.
//
// One rank per GPU.  Boxes owned by this rank exchange ghosts with device-side copies; boxes owned by
// other ranks go through packed HBM buffers and RCCL point-to-point (ncclSend/ncclRecv inside one
// group per exchange -- xGMI is point-to-point, a 256^3 box has at most 26 neighbours and the seven
// face/edge/corner peers of a 2x2x2 decomposition map onto the seven links).  Reductions
// (residual norms, estdt maxima, umac_norm) are ncclAllReduce(MAX) on a device scalar.
//
// RCCL is bound with dlopen("librccl.so.1") at vdn_comm_init time instead of at link time: the host
// process (PyTorch in bench.py, an MPI Fortran driver elsewhere) may already have an RCCL loaded, and the
// dynamic linker then hands back that same copy instead of a second one.
//
// The exchange plan -- which region of which box goes where -- is the analogue of FBoxLib's cached
// `copyassoc` (reference src/main.f90:23,39-47); it is built once per (allocation, shape) and kept on
// the device.  Both sides of a remote copy enumerate (dst box, src box, periodic shift) in the same
// canonical order, so offsets into the per-peer buffers agree without any handshake.
#include "vdn_dev.h"
#include <dlfcn.h>
#include <cstdlib>
#include <algorithm>
#include <tuple>
#include <array>

void mg_halo_cache_purge(unsigned long uid);   // mg_cc.hip / mg_nd.hip keep key -> plan maps; they drop the keys

// ====================================================================================================
// RCCL through dlopen
// ====================================================================================================
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1, ncclFloat64 = 8 };       // ncclDataType_t: ncclUint8, ncclDouble
enum { ncclSum = 0, ncclMax = 2 };
struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(ncclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommCount)(ncclComm_t, int *) = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  ncclComm_t comm = nullptr;
  bool test_double = false;
};
static Rccl g_rccl;
static int packed_mode() { const char *e = vdn_env("VDN_FORCE_PACKED"); return e ? atoi(e) : 0; }
#define NCCLCHK(x) do { int r_ = (x); if (r_ != ncclSuccess) vdn_fail("%s failed: %s", #x, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); } while (0)

static void rccl_load() {
  if (g_rccl.h) return;
  // The transport is RCCL, found by name.  The ONE exception is the test double of the multi-rank tests (tests/fake_rccl: several
  // ranks on one GPU), and an environment variable alone must not be able to swap the transport of the shipping library: VDN_RCCL_LIB
  // is honoured only together with VDN_TESTING=1 AND a library that identifies itself through vdn_test_transport_magic(); anything else
  // named there fails the call.  vdn_comm_transport() says which one is in use (bench.py prints it).
#ifdef VDN_TESTING_BUILD
  const char *forced = vdn_env("VDN_RCCL_LIB");
#else
  const char *forced = nullptr;            // the release build has no seam for another transport: RCCL by name, nothing else
#endif
  // the handle and the entry points go into a local copy: g_rccl is assigned only when the handshake and every lookup succeeded, so a
  // failed load leaves no half-bound state behind (a later call would otherwise return early here and jump through null pointers)
  Rccl R;
  struct Closer { void *&h; bool armed = true; ~Closer() { if (armed && h) { dlclose(h); h = nullptr; } } } closer{ R.h };
  if (forced && *forced) {
    const char *t = vdn_env("VDN_TESTING");
    REQUIRE(t && atoi(t) == 1, "VDN_RCCL_LIB is set but VDN_TESTING=1 is not: the transport of this library is RCCL; only the test suite may replace it");
    R.h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    REQUIRE(R.h, "cannot dlopen the test transport %s: %s", forced, dlerror());
    long (*magic)() = nullptr; *(void **)(&magic) = dlsym(R.h, "vdn_test_transport_magic");
    REQUIRE(magic && magic() == 0x76646e74657374L, "VDN_RCCL_LIB names %s, which is not the test double of tests/fake_rccl", forced);
    R.test_double = true;
    fprintf(stderr, "varden_amd: TEST TRANSPORT %s in place of RCCL (VDN_TESTING=1)\n", forced);
  }
  const char *names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
  for (const char *n : names) { if (R.h) break; R.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
  REQUIRE(R.h, "cannot dlopen librccl.so.1: %s", dlerror());
  #define SYM(field, name) do { *(void **)(&R.field) = dlsym(R.h, name); REQUIRE(R.field, "RCCL symbol %s missing", name); } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount");
  SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather");
  SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
  closer.armed = false;
  g_rccl = R;
  #undef SYM
}

extern "C" int vdn_comm_get_unique_id(char *id128) {
  VDN_TRY
  rccl_load();
  ncclUniqueId id; NCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, 128);
  VDN_CATCH
}
extern "C" int vdn_comm_init(const char *id128) {
  VDN_TRY
  REQUIRE(ctx().inited, "vdn_comm_init: call vdn_init first");
  // VDN_FORCE_PACKED=2 (hardware self-test of the RCCL call path on ONE GPU): build a 1-rank communicator and send
  // the rank's own packed buffers to itself with ncclSend/ncclRecv; reductions go through ncclAllReduce too
  if (ctx().nranks == 1 && packed_mode() != 2) return 0;
  rccl_load();
  ncclUniqueId id; memcpy(id.internal, id128, 128);
  NCCLCHK(g_rccl.CommInitRank(&g_rccl.comm, ctx().nranks, id, ctx().rank));
  VDN_CATCH
}
extern "C" int vdn_comm_finalize(void) {
  VDN_TRY
  if (g_rccl.comm) { HIPCHK(hipStreamSynchronize(ctx().stream)); NCCLCHK(g_rccl.CommDestroy(g_rccl.comm)); g_rccl.comm = nullptr; }
  VDN_CATCH
}
// number of ranks of the live communicator, read back from RCCL (ncclCommCount); 1 when no communicator is up
extern "C" int vdn_comm_nranks(int *n) {
  VDN_TRY
  *n = 1;
  if (g_rccl.comm) NCCLCHK(g_rccl.CommCount(g_rccl.comm, n));
  VDN_CATCH
}
bool comm_active() { return ctx().nranks > 1 || g_rccl.comm != nullptr; }
// traffic counters (vdn_comm_stats): what a step asks of the transport -- the budget of profiles/r03_exchange_budget.md is made of these
// [0] exchanges with remote traffic (one pack + ncclGroup + unpack each)  [1] ncclSend calls  [2] doubles sent  [3] all-reduces
// [4] all-gathers  [5] doubles contributed to all-gathers  [6] view refreshes (inter-level operators)  [7] doubles sent by them
// [8 + b] exchanges whose sends add up to [2^(10+b), 2^(11+b)) bytes, b = 0 .. 15 (b = 0 also takes everything smaller)
static long g_cstat[24];
static void cstat_exchange(size_t doubles, int sends, bool view) {
  g_cstat[view ? 6 : 0]++; g_cstat[1] += sends; g_cstat[view ? 7 : 2] += (long)doubles;
  if (!view) { int b = 0; size_t bytes = doubles * 8; while (b < 15 && bytes >= ((size_t)2048 << b)) b++; g_cstat[8 + b]++; }
}
extern "C" const char *vdn_comm_transport(void) { return !g_rccl.h ? "none" : (g_rccl.test_double ? "test-double" : "rccl"); }
extern "C" int vdn_comm_stats(long *out24, int reset) {
  VDN_TRY
  if (out24) memcpy(out24, g_cstat, sizeof g_cstat);
  if (reset) memset(g_cstat, 0, sizeof g_cstat);
  VDN_CATCH
}
static void need_comm() { REQUIRE(g_rccl.comm != nullptr, "this operation spans ranks: call vdn_comm_init first (nranks = %d)", ctx().nranks); }

// all-reduce MAX of n device doubles, in place, on the launch stream
void comm_allreduce_max_dev(double *d, int n) {
  if (!comm_active()) return;
  need_comm();
  g_cstat[3]++;
  NCCLCHK(g_rccl.AllReduce(d, d, (size_t)n, ncclFloat64, ncclMax, g_rccl.comm, ctx().stream));
}
// all-reduce MAX of n device bytes (tag bitmaps of the grid generation), in place, in pieces of 8 MB
void comm_allreduce_max_u8_dev(unsigned char *d, size_t n) {
  if (!comm_active()) return;
  need_comm();
  const size_t piece = (size_t)8 << 20;
  for (size_t off = 0; off < n; off += piece)
    NCCLCHK(g_rccl.AllReduce(d + off, d + off, std::min(piece, n - off), ncclUint8, ncclMax, g_rccl.comm, ctx().stream));
}
// host values (per-box minima / maxima of the plot files, barriers of the file writers): MAX over the ranks, in place
extern "C" int vdn_comm_allreduce_max(double *host, int n) {
  VDN_TRY
  REQUIRE(ctx().inited && n >= 0, "vdn_comm_allreduce_max: not initialised");
  if (!comm_active() || n == 0) return 0;
  double *d = nullptr;
  HIPCHK(hipMalloc((void **)&d, (size_t)n * sizeof(double)));
  HIPCHK(hipMemcpyAsync(d, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx().stream));
  comm_allreduce_max_dev(d, n);
  HIPCHK(hipMemcpyAsync(host, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx().stream));
  HIPCHK(hipStreamSynchronize(ctx().stream));
  HIPCHK(hipFree(d));
  VDN_CATCH
}
// all-gather: every rank contributes `count` doubles; recv holds nranks*count
void comm_allgather_dev(const double *send, double *recv, size_t count) {
  if (!comm_active()) { if (send != recv) HIPCHK(hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, ctx().stream)); return; }
  need_comm();
  g_cstat[4]++; g_cstat[5] += (long)count;
  NCCLCHK(g_rccl.AllGather(send, recv, count, ncclFloat64, g_rccl.comm, ctx().stream));
}

// ====================================================================================================
// exchange plans
// ====================================================================================================
struct CopyDesc { FV dst, src; int lo[3], hi[3]; int sh[3]; int vlo[3], vhi[3]; };
struct PackDesc { FV fv; int lo[3], hi[3]; int sh[3]; int vlo[3], vhi[3]; long off; double *buf; };   // pack: fv = src (read at q - sh); unpack: fv = dst; buf: the peer's buffer

// box-to-box ghost copies of a level in one launch: a descriptor per (destination, source, shift) overlap, XCOPY_CHUNK points per
// workgroup, workgroups bisect the prefix sum of chunk counts (a level of a few hundred boxes has thousands of thin slabs)
constexpr int XCOPY_CHUNK = 1024;
__global__ void __launch_bounds__(256) k_xcopy(const CopyDesc *descs, const int *start, int ndesc, int nc) {
  int lo = 0, hi = ndesc - 1;
  const int bid = (int)blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (as_constant(start + mid) <= bid) lo = mid; else hi = mid - 1; }
  const CopyDesc &D = as_constant(descs + lo);
  const int nx = D.hi[0] - D.lo[0] + 1, ny = D.hi[1] - D.lo[1] + 1, nz = D.hi[2] - D.lo[2] + 1;
  const int tot = nx * ny * nz, t0 = (bid - as_constant(start + lo)) * XCOPY_CHUNK;
  #pragma unroll
  for (int u = 0; u < XCOPY_CHUNK / 256; u++) {
    const int t = t0 + u * 256 + (int)threadIdx.x;
    if (t >= tot) break;
    const int i = D.lo[0] + t % nx, j = D.lo[1] + (t / nx) % ny, k = D.lo[2] + t / (nx * ny);
    const bool inside = i >= D.vlo[0] && i <= D.vhi[0] && j >= D.vlo[1] && j <= D.vhi[1] && k >= D.vlo[2] && k <= D.vhi[2];
    if (inside) continue;                 // only ghost points are filled
    for (int c = 0; c < nc; c++) fv_at(D.dst, i, j, k, c) = fv_get(D.src, i - D.sh[0], j - D.sh[1], k - D.sh[2], c);
  }
}
constexpr unsigned XPACK_WG = 32;
__global__ void k_xpack(const PackDesc *descs, int nc, double *buf) {
  // XPACK_WG workgroups per descriptor, descriptor = blockIdx.x / XPACK_WG (gridDim.x may be 2^31 - 1; gridDim.z is capped at 65535,
  // which a level of a few thousand boxes with periodic images exceeds)
  const PackDesc &D = descs[blockIdx.x / XPACK_WG];
  const int nx = D.hi[0] - D.lo[0] + 1, ny = D.hi[1] - D.lo[1] + 1, nz = D.hi[2] - D.lo[2] + 1;
  const long tot = (long)nx * ny * nz;
  for (long t = (long)(blockIdx.x % XPACK_WG) * blockDim.x + threadIdx.x; t < tot; t += (long)XPACK_WG * blockDim.x) {
    const int i = D.lo[0] + (int)(t % nx), j = D.lo[1] + (int)((t / nx) % ny), k = D.lo[2] + (int)(t / ((long)nx * ny));
    for (int c = 0; c < nc; c++) buf[D.off + c * tot + t] = fv_get(D.fv, i - D.sh[0], j - D.sh[1], k - D.sh[2], c);
  }
}
__global__ void k_xunpack(const PackDesc *descs, int nc, const double *buf) {
  // XPACK_WG workgroups per descriptor, descriptor = blockIdx.x / XPACK_WG (gridDim.x may be 2^31 - 1; gridDim.z is capped at 65535,
  // which a level of a few thousand boxes with periodic images exceeds)
  const PackDesc &D = descs[blockIdx.x / XPACK_WG];
  const int nx = D.hi[0] - D.lo[0] + 1, ny = D.hi[1] - D.lo[1] + 1, nz = D.hi[2] - D.lo[2] + 1;
  const long tot = (long)nx * ny * nz;
  for (long t = (long)(blockIdx.x % XPACK_WG) * blockDim.x + threadIdx.x; t < tot; t += (long)XPACK_WG * blockDim.x) {
    const int i = D.lo[0] + (int)(t % nx), j = D.lo[1] + (int)((t / nx) % ny), k = D.lo[2] + (int)(t / ((long)nx * ny));
    const bool inside = i >= D.vlo[0] && i <= D.vhi[0] && j >= D.vlo[1] && j <= D.vhi[1] && k >= D.vlo[2] && k <= D.vhi[2];
    if (inside) continue;
    for (int c = 0; c < nc; c++) fv_at(D.fv, i, j, k, c) = buf[D.off + c * tot + t];
  }
}

// all peers in one launch (the descriptor names its peer's buffer): with seven peers per rank a launch per peer and direction put 14 small
// kernels around every ncclGroup
__global__ void k_xpack_all(const PackDesc *descs, int nc) {
  const PackDesc &D = descs[blockIdx.x / XPACK_WG];
  const int nx = D.hi[0] - D.lo[0] + 1, ny = D.hi[1] - D.lo[1] + 1, nz = D.hi[2] - D.lo[2] + 1;
  const long tot = (long)nx * ny * nz;
  double *buf = D.buf;
  for (long t = (long)(blockIdx.x % XPACK_WG) * blockDim.x + threadIdx.x; t < tot; t += (long)XPACK_WG * blockDim.x) {
    const int i = D.lo[0] + (int)(t % nx), j = D.lo[1] + (int)((t / nx) % ny), k = D.lo[2] + (int)(t / ((long)nx * ny));
    for (int c = 0; c < nc; c++) buf[D.off + c * tot + t] = fv_get(D.fv, i - D.sh[0], j - D.sh[1], k - D.sh[2], c);
  }
}
__global__ void k_xunpack_all(const PackDesc *descs, int nc) {
  const PackDesc &D = descs[blockIdx.x / XPACK_WG];
  const int nx = D.hi[0] - D.lo[0] + 1, ny = D.hi[1] - D.lo[1] + 1, nz = D.hi[2] - D.lo[2] + 1;
  const long tot = (long)nx * ny * nz;
  const double *buf = D.buf;
  for (long t = (long)(blockIdx.x % XPACK_WG) * blockDim.x + threadIdx.x; t < tot; t += (long)XPACK_WG * blockDim.x) {
    const int i = D.lo[0] + (int)(t % nx), j = D.lo[1] + (int)((t / nx) % ny), k = D.lo[2] + (int)(t / ((long)nx * ny));
    const bool inside = i >= D.vlo[0] && i <= D.vhi[0] && j >= D.vlo[1] && j <= D.vhi[1] && k >= D.vlo[2] && k <= D.vhi[2];
    if (inside) continue;
    for (int c = 0; c < nc; c++) fv_at(D.fv, i, j, k, c) = buf[D.off + c * tot + t];
  }
}
struct Peer {
  int rank = -1;
  std::vector<PackDesc> pack, unpack;
  size_t nsend = 0, nrecv = 0;           // doubles
  PackDesc *d_pack = nullptr, *d_unpack = nullptr;
  double *d_send = nullptr, *d_recv = nullptr;
};
struct XPlan {
  std::vector<CopyDesc> local; CopyDesc *d_local = nullptr; int *d_lstart = nullptr; int lchunks = 0;
  std::vector<Peer> peers;
  PackDesc *d_pack_all = nullptr, *d_unpack_all = nullptr; int npack_all = 0, nunpack_all = 0;      // the descriptors of all peers, one launch each way
  int nc = 1;
  unsigned long serial = 0;          // unique over the life of the process: cache keys (hipGraph replay) must not match a new plan that malloc put at a freed plan's address
};
unsigned long xplan_serial(const XPlan *P) { return P ? P->serial : 0; }

// the plan for: every box b (global list) has valid point range [vlo,vhi] (incl. nodal points) and, if
// local, an FV view; ghosts of width ng are filled from other boxes' valid points, through periodic
// shifts of the domain `pd` where pmask says so.
// faces_only: only the ghost cells that lie outside the valid box in exactly ONE direction are filled (what a 7-point operator reads): with a
// 2 x 2 x 2 decomposition a rank then exchanges with its three face neighbours instead of seven peers per colour pass
XPlan *xplan_build(const std::vector<XBoxInfo> &boxes, const vdn_box &pd, const int pmask[3], int ng, int nc, bool faces_only, const int *src_trim) {
  Prof prof_("xplan_build");
  static unsigned long next_serial = 0;
  XPlan *P = new XPlan; P->nc = nc; P->serial = ++next_serial;
  const int me = ctx().rank;
  // self-test mode: route the rank's OWN box-to-box copies through the pack -> buffer -> unpack path that remote
  // copies take (peer == me, buffer handed over with a device memcpy instead of ncclSend/ncclRecv)
  const bool force_packed = packed_mode() >= 1;
  int per[3], nshift[3];
  for (int d = 0; d < 3; d++) { per[d] = pd.hi[d] - pd.lo[d] + 1; nshift[d] = pmask[d] ? 1 : 0; }
  std::map<int, Peer> peers;
  const int nb = (int)boxes.size();
  std::vector<vdn_box> vb(nb);
  for (int i = 0; i < nb; i++) for (int d = 0; d < 3; d++) { vb[i].lo[d] = boxes[i].vlo[d]; vb[i].hi[d] = boxes[i].vhi[d]; }
  const BoxBins bins(vb);
  std::vector<int> cand;
  for (int i = 0; i < nb; i++) {
    const XBoxInfo &B = boxes[i];
    int glo[3], ghi[3];
    for (int d = 0; d < 3; d++) { glo[d] = B.vlo[d] - ng; ghi[d] = B.vhi[d] + ng; }
    // the boxes whose valid range, shifted by a period or not, can reach B's grown box: from the bins, in ascending order -- the pairs and their order are those of
    // the loop over all j (a level of a thousand boxes: 51 plans after a regrid took 140 ms of pair tests, profiles/r06_regrid_cost.txt)
    cand.clear();
    for (int sz = -nshift[2]; sz <= nshift[2]; sz++) for (int sy = -nshift[1]; sy <= nshift[1]; sy++) for (int sx = -nshift[0]; sx <= nshift[0]; sx++) {
      const int sh[3] = { sx * per[0], sy * per[1], sz * per[2] };
      int qlo[3], qhi[3]; for (int d = 0; d < 3; d++) { qlo[d] = glo[d] - sh[d]; qhi[d] = ghi[d] - sh[d]; }
      const std::vector<int> &c = bins.near(qlo, qhi, 0);
      cand.insert(cand.end(), c.begin(), c.end());
    }
    if (nshift[0] || nshift[1] || nshift[2]) { std::sort(cand.begin(), cand.end()); cand.erase(std::unique(cand.begin(), cand.end()), cand.end()); }
    for (int j : cand) {
      const XBoxInfo &S = boxes[j];
      if (B.owner != me && S.owner != me) continue;
      for (int sz = -nshift[2]; sz <= nshift[2]; sz++) for (int sy = -nshift[1]; sy <= nshift[1]; sy++) for (int sx = -nshift[0]; sx <= nshift[0]; sx++) {
        if (j == i && sx == 0 && sy == 0 && sz == 0) continue;
        const int sh[3] = { sx * per[0], sy * per[1], sz * per[2] };
        int lo[3], hi[3]; bool empty = false, all_inside = true;
        for (int d = 0; d < 3; d++) {
          const int st_ = src_trim ? src_trim[d] : 0;             // 1: the source without its high plane along d; 2: its high plane alone
          lo[d] = std::max(glo[d], (st_ == 2 ? S.vhi[d] : S.vlo[d]) + sh[d]); hi[d] = std::min(ghi[d], S.vhi[d] - (st_ == 1 ? 1 : 0) + sh[d]);
          if (lo[d] > hi[d]) empty = true;
          if (lo[d] < B.vlo[d] || hi[d] > B.vhi[d]) all_inside = false;
        }
        if (empty || all_inside) continue;
        // the sub-regions to exchange: the whole overlap, or (faces_only) its parts beyond one face of B with the other two directions cut
        // to B's valid range
        int nsub = 1, slo[6][3], shi[6][3];
        for (int d = 0; d < 3; d++) { slo[0][d] = lo[d]; shi[0][d] = hi[d]; }
        if (faces_only) {
          nsub = 0;
          for (int d = 0; d < 3; d++) for (int side = 0; side < 2; side++) {
            int a[3], b[3]; bool ok = true;
            for (int t = 0; t < 3; t++) {
              if (t == d) { a[t] = side ? std::max(lo[t], B.vhi[t] + 1) : lo[t]; b[t] = side ? hi[t] : std::min(hi[t], B.vlo[t] - 1); }
              else { a[t] = std::max(lo[t], B.vlo[t]); b[t] = std::min(hi[t], B.vhi[t]); }
              if (a[t] > b[t]) ok = false;
            }
            if (ok) { for (int t = 0; t < 3; t++) { slo[nsub][t] = a[t]; shi[nsub][t] = b[t]; } nsub++; }
          }
        }
        for (int q = 0; q < nsub; q++) {
        for (int d = 0; d < 3; d++) { lo[d] = slo[q][d]; hi[d] = shi[q][d]; }
        const long cnt = (long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) * nc;
        if (B.owner == me && S.owner == me && !force_packed) {
          CopyDesc D; memset(&D, 0, sizeof D);
          D.dst = B.fv; D.src = S.fv;
          for (int d = 0; d < 3; d++) { D.lo[d] = lo[d]; D.hi[d] = hi[d]; D.sh[d] = sh[d]; D.vlo[d] = B.vlo[d]; D.vhi[d] = B.vhi[d]; }
          P->local.push_back(D);
        } else {
          PackDesc D; memset(&D, 0, sizeof D);
          for (int d = 0; d < 3; d++) { D.lo[d] = lo[d]; D.hi[d] = hi[d]; D.sh[d] = sh[d]; D.vlo[d] = B.vlo[d]; D.vhi[d] = B.vhi[d]; }
          if (S.owner == me) {            // I send to the owner of the destination box
            Peer &pr = peers[B.owner]; pr.rank = B.owner;
            PackDesc Ds = D; Ds.fv = S.fv; Ds.off = (long)pr.nsend; pr.nsend += cnt; pr.pack.push_back(Ds);
          }
          if (B.owner == me) {            // I receive from the owner of the source box
            Peer &pr = peers[S.owner]; pr.rank = S.owner;
            PackDesc Dr = D; Dr.fv = B.fv; Dr.off = (long)pr.nrecv; pr.nrecv += cnt; pr.unpack.push_back(Dr);
          }
        }
        }
      }
    }
  }
  if (!P->local.empty()) {
    HIPCHK(hipMalloc((void **)&P->d_local, P->local.size() * sizeof(CopyDesc)));
    upload_staged(P->d_local, P->local.data(), P->local.size() * sizeof(CopyDesc));
    std::vector<int> lstart(P->local.size());
    for (size_t q = 0; q < P->local.size(); q++) {
      const CopyDesc &D = P->local[q];
      const long tot = (long)(D.hi[0] - D.lo[0] + 1) * (D.hi[1] - D.lo[1] + 1) * (D.hi[2] - D.lo[2] + 1);
      REQUIRE(tot < (1L << 31) && (long)P->lchunks + tot / XCOPY_CHUNK + 1 < (1L << 31), "fill_boundary: overlap too large for 32-bit indexing");
      lstart[q] = P->lchunks; P->lchunks += (int)((tot + XCOPY_CHUNK - 1) / XCOPY_CHUNK);
    }
    HIPCHK(hipMalloc((void **)&P->d_lstart, lstart.size() * sizeof(int)));
    upload_staged(P->d_lstart, lstart.data(), lstart.size() * sizeof(int));
  }
  for (auto &kv : peers) {
    Peer pr = kv.second;
    if (!pr.pack.empty()) { HIPCHK(hipMalloc((void **)&pr.d_pack, pr.pack.size() * sizeof(PackDesc)));
      upload_staged(pr.d_pack, pr.pack.data(), pr.pack.size() * sizeof(PackDesc));
      HIPCHK(hipMalloc((void **)&pr.d_send, pr.nsend * sizeof(double))); }
    if (!pr.unpack.empty()) { HIPCHK(hipMalloc((void **)&pr.d_unpack, pr.unpack.size() * sizeof(PackDesc)));
      upload_staged(pr.d_unpack, pr.unpack.data(), pr.unpack.size() * sizeof(PackDesc));
      HIPCHK(hipMalloc((void **)&pr.d_recv, pr.nrecv * sizeof(double))); }
    P->peers.push_back(pr);
  }
  {
    std::vector<PackDesc> ap, au;
    for (auto &pr : P->peers) {
      for (PackDesc D : pr.pack) { D.buf = pr.d_send; ap.push_back(D); }
      for (PackDesc D : pr.unpack) { D.buf = pr.d_recv; au.push_back(D); }
    }
    P->npack_all = (int)ap.size(); P->nunpack_all = (int)au.size();
    if (!ap.empty()) { HIPCHK(hipMalloc((void **)&P->d_pack_all, ap.size() * sizeof(PackDesc))); upload_staged(P->d_pack_all, ap.data(), ap.size() * sizeof(PackDesc)); }
    if (!au.empty()) { HIPCHK(hipMalloc((void **)&P->d_unpack_all, au.size() * sizeof(PackDesc))); upload_staged(P->d_unpack_all, au.data(), au.size() * sizeof(PackDesc)); }
  }
  // (no synchronisation: the uploads went through the pinned ring, asynchronous on the launch stream that every user of the plan is ordered after --
  // a regrid builds some fifty plans, and each used to drain the stream)
  return P;
}

void xplan_free(XPlan *P) {
  if (!P) return;
  HIPCHK(hipStreamSynchronize(ctx().stream));
  if (P->d_local) HIPCHK(hipFree(P->d_local));
  if (P->d_lstart) HIPCHK(hipFree(P->d_lstart));
  if (P->d_pack_all) HIPCHK(hipFree(P->d_pack_all));
  if (P->d_unpack_all) HIPCHK(hipFree(P->d_unpack_all));
  for (auto &pr : P->peers) {
    if (pr.d_pack) HIPCHK(hipFree(pr.d_pack));
    if (pr.d_unpack) HIPCHK(hipFree(pr.d_unpack));
    if (pr.d_send) HIPCHK(hipFree(pr.d_send));
    if (pr.d_recv) HIPCHK(hipFree(pr.d_recv));
  }
  delete P;
}

bool xplan_has_remote(const XPlan *P) {
  if (!P) return false;
  for (const auto &pr : P->peers) if (pr.rank != ctx().rank || packed_mode() == 2) return true;
  return false;
}
void xplan_run(XPlan *P, hipStream_t st) {
  if (!st) st = ctx().stream;
  const int nc = P->nc;
  // pack + post the remote traffic first so that it overlaps the local copies
  if (!P->peers.empty()) {
    const bool self_rccl = packed_mode() == 2 && g_rccl.comm != nullptr;    // self peer through ncclSend/ncclRecv
    bool remote = self_rccl;
    for (auto &pr : P->peers) if (pr.rank != ctx().rank) remote = true;
    if (P->npack_all) hipLaunchKernelGGL(k_xpack_all, dim3(XPACK_WG * (unsigned)P->npack_all), dim3(256), 0, st, P->d_pack_all, nc);
    if (remote) {
      need_comm();
      { size_t tot = 0; int ns = 0; for (auto &pr : P->peers) if (!(pr.rank == ctx().rank && !self_rccl) && pr.nsend) { tot += pr.nsend; ns++; } cstat_exchange(tot, ns, false); }
      NCCLCHK(g_rccl.GroupStart());
      for (auto &pr : P->peers) {
        if (pr.rank == ctx().rank && !self_rccl) continue;
        if (pr.nsend) NCCLCHK(g_rccl.Send(pr.d_send, pr.nsend, ncclFloat64, pr.rank, g_rccl.comm, st));
        if (pr.nrecv) NCCLCHK(g_rccl.Recv(pr.d_recv, pr.nrecv, ncclFloat64, pr.rank, g_rccl.comm, st));
      }
      NCCLCHK(g_rccl.GroupEnd());
    }
    for (auto &pr : P->peers)           // self-test mode (VDN_FORCE_PACKED): my own buffer is my inbox
      if (pr.rank == ctx().rank && pr.nsend && !self_rccl) { REQUIRE(pr.nsend == pr.nrecv, "self exchange: send/recv sizes differ"); HIPCHK(hipMemcpyAsync(pr.d_recv, pr.d_send, pr.nsend * sizeof(double), hipMemcpyDeviceToDevice, st)); }
  }
  if (!P->local.empty())
    hipLaunchKernelGGL(k_xcopy, dim3((unsigned)P->lchunks), dim3(256), 0, st, P->d_local, P->d_lstart, (int)P->local.size(), nc);
  if (P->nunpack_all) hipLaunchKernelGGL(k_xunpack_all, dim3(XPACK_WG * (unsigned)P->nunpack_all), dim3(256), 0, st, P->d_unpack_all, nc);
  dbg_sync(4);
}

// ====================================================================================================
// views (see vdn_internal.h): windows of remote boxes for the inter-level operators
// ====================================================================================================
struct ViewPlan {
  std::vector<Peer> peers;            // pack descriptors of what I send; nrecv doubles per peer arrive in d_recv
  PackDesc *d_pack_all = nullptr; int npack_all = 0;       // all peers' pack descriptors: one launch (k_xpack_all)
  int nc = 1;
};
struct ViewKey { unsigned long uid; const void *base; int lev, scomp, nc, ng, nd; unsigned long tag; bool operator<(const ViewKey &o) const {
  return std::tie(uid, base, lev, scomp, nc, ng, nd, tag) < std::tie(o.uid, o.base, o.lev, o.scomp, o.nc, o.ng, o.nd, o.tag); } };
static std::map<ViewKey, SrcView> g_view_cache;
static void viewplan_free(ViewPlan *P) {
  if (!P) return;
  HIPCHK(hipStreamSynchronize(ctx().stream));
  if (P->d_pack_all) HIPCHK(hipFree(P->d_pack_all));
  for (auto &pr : P->peers) { if (pr.d_pack) HIPCHK(hipFree(pr.d_pack)); if (pr.d_send) HIPCHK(hipFree(pr.d_send)); if (pr.d_recv) HIPCHK(hipFree(pr.d_recv)); }
  delete P;
}
void view_cache_purge(unsigned long uid) {
  for (auto it = g_view_cache.begin(); it != g_view_cache.end();) {
    if (it->first.uid == uid) { viewplan_free(it->second.plan); it = g_view_cache.erase(it); } else ++it;
  }
}
void SrcView::refresh() const {
  if (!plan || plan->peers.empty()) return;
  hipStream_t st = ctx().stream;
  need_comm();
  if (plan->npack_all) hipLaunchKernelGGL(k_xpack_all, dim3(XPACK_WG * (unsigned)plan->npack_all), dim3(256), 0, st, plan->d_pack_all, plan->nc);
  { size_t tot = 0; int ns = 0; for (auto &pr : plan->peers) if (pr.nsend) { tot += pr.nsend; ns++; } cstat_exchange(tot, ns, true); }
  NCCLCHK(g_rccl.GroupStart());
  for (auto &pr : plan->peers) {
    if (pr.nsend) NCCLCHK(g_rccl.Send(pr.d_send, pr.nsend, ncclFloat64, pr.rank, g_rccl.comm, st));
    if (pr.nrecv) NCCLCHK(g_rccl.Recv(pr.d_recv, pr.nrecv, ncclFloat64, pr.rank, g_rccl.comm, st));
  }
  NCCLCHK(g_rccl.GroupEnd());
}
SrcView make_view(const vdn_multifab *src, const std::vector<vdn_box> &footprint, const std::vector<int> &dst_owner, int scomp, int nc, unsigned long cache_tag) {
  const vdn_layout *la = src->la;
  const int lev = src->lev, me = ctx().rank, nranks = ctx().nranks;
  const auto &gb = la->boxes[lev];
  ViewKey key{ la->uid, src->base, lev, scomp, nc, src->ng, src->nodal[0] | (src->nodal[1] << 1) | (src->nodal[2] << 2), cache_tag };
  auto hit = g_view_cache.find(key);
  if (hit != g_view_cache.end()) return hit->second;
  // periodic images: an entry of the view is (box, shift); shift 0 first, so that a non-periodic view has one entry per box
  std::vector<std::array<int, 3>> shifts;
  {
    int per[3], ns[3];
    for (int d = 0; d < 3; d++) { per[d] = la->pd[lev].hi[d] - la->pd[lev].lo[d] + 1; ns[d] = la->pmask[d] ? 1 : 0; }
    shifts.push_back({ 0, 0, 0 });
    for (int sz = -ns[2]; sz <= ns[2]; sz++) for (int sy = -ns[1]; sy <= ns[1]; sy++) for (int sx = -ns[0]; sx <= ns[0]; sx++)
      if (sx || sy || sz) shifts.push_back({ sx * per[0], sy * per[1], sz * per[2] });
  }
  const int nsh = (int)shifts.size();
  SrcView V; V.ng = src->ng; V.nc = nc; for (int d = 0; d < 3; d++) V.nodal[d] = src->nodal[d];
  V.vbox.resize(gb.size() * nsh); V.have.assign(gb.size() * nsh, 0); V.fv.resize(gb.size() * nsh);
  std::vector<FV> localfv(gb.size());
  { int li = 0; for (size_t j = 0; j < gb.size(); j++) if (la->owner[lev][j] == me) { FV f = src->fabs[li++]; f.p += (long)f.sc * scomp; localfv[j] = f; } }
  ViewPlan *P = nranks > 1 ? new ViewPlan : nullptr;
  if (P) P->nc = nc;
  std::map<int, Peer> peers;
  struct Win { int lo[3], hi[3]; bool any; };
  for (size_t j = 0; j < gb.size(); j++) for (int si = 0; si < nsh; si++) {
    const size_t e = j * nsh + si;
    const int oj = la->owner[lev][j];
    const std::array<int, 3> &sh = shifts[si];
    int alo[3], ahi[3];
    for (int d = 0; d < 3; d++) { V.vbox[e].lo[d] = gb[j].lo[d] + sh[d]; V.vbox[e].hi[d] = gb[j].hi[d] + sh[d]; alo[d] = V.vbox[e].lo[d] - src->ng; ahi[d] = V.vbox[e].hi[d] + src->nodal[d] + src->ng; }
    // the window of this entry that rank r needs: bounding box over r's destination boxes of (footprint ∩ shifted allocation)
    std::vector<Win> win(nranks);
    for (auto &w : win) { w.any = false; for (int d = 0; d < 3; d++) { w.lo[d] = 1 << 30; w.hi[d] = -(1 << 30); } }
    for (size_t i = 0; i < footprint.size(); i++) {
      const int r = dst_owner[i];
      int lo[3], hi[3]; bool empty = false;
      for (int d = 0; d < 3; d++) { lo[d] = std::max(footprint[i].lo[d], alo[d]); hi[d] = std::min(footprint[i].hi[d], ahi[d]); if (lo[d] > hi[d]) empty = true; }
      if (empty) continue;
      Win &w = win[r]; w.any = true;
      for (int d = 0; d < 3; d++) { w.lo[d] = std::min(w.lo[d], lo[d]); w.hi[d] = std::max(w.hi[d], hi[d]); }
    }
    if (oj == me && (si == 0 || win[me].any)) {              // my own box: the fab itself, seen at its shifted position
      FV f = localfv[j]; f.a0 += sh[0]; f.a1 += sh[1]; f.a2 += sh[2];
      V.fv[e] = f; V.have[e] = 1;
    }
    for (int r = 0; r < nranks && P; r++) {
      if (!win[r].any || r == oj) continue;
      const Win &w = win[r];
      const long tot = (long)(w.hi[0] - w.lo[0] + 1) * (w.hi[1] - w.lo[1] + 1) * (w.hi[2] - w.lo[2] + 1);
      if (oj == me) {                                        // I send the window of my box j (read at window - shift) to rank r
        Peer &pr = peers[r]; pr.rank = r;
        PackDesc D; memset(&D, 0, sizeof D);
        D.fv = localfv[j];
        for (int d = 0; d < 3; d++) { D.lo[d] = w.lo[d]; D.hi[d] = w.hi[d]; D.sh[d] = sh[d]; D.vlo[d] = 1; D.vhi[d] = 0; }
        D.off = (long)pr.nsend; pr.nsend += (size_t)tot * nc; pr.pack.push_back(D);
      } else if (r == me) {                                  // I receive it from the owner of j: remember where it will sit
        Peer &pr = peers[oj]; pr.rank = oj;
        PackDesc D; memset(&D, 0, sizeof D);
        for (int d = 0; d < 3; d++) { D.lo[d] = w.lo[d]; D.hi[d] = w.hi[d]; }
        D.off = (long)pr.nrecv; pr.nrecv += (size_t)tot * nc; D.vlo[0] = (int)e;          // vlo[0] carries the entry index until the buffers exist
        pr.unpack.push_back(D);
      }
    }
  }
  if (P) {
    for (auto &kv : peers) {
      Peer pr = kv.second;
      if (!pr.pack.empty()) { HIPCHK(hipMalloc((void **)&pr.d_pack, pr.pack.size() * sizeof(PackDesc)));
        HIPCHK(hipMemcpyAsync(pr.d_pack, pr.pack.data(), pr.pack.size() * sizeof(PackDesc), hipMemcpyHostToDevice, ctx().stream));
        HIPCHK(hipMalloc((void **)&pr.d_send, pr.nsend * sizeof(double))); }
      if (pr.nrecv) {
        HIPCHK(hipMalloc((void **)&pr.d_recv, pr.nrecv * sizeof(double)));
        HIPCHK(hipMemsetAsync(pr.d_recv, 0, pr.nrecv * sizeof(double), ctx().stream));
        for (const PackDesc &D : pr.unpack) {                // the window lives in the receive buffer: same layout as k_xpack writes
          const int e = D.vlo[0];
          FV f; f.p = pr.d_recv + D.off; f.a0 = D.lo[0]; f.a1 = D.lo[1]; f.a2 = D.lo[2];
          f.n0 = D.hi[0] - D.lo[0] + 1; f.n1 = D.hi[1] - D.lo[1] + 1; f.n2 = D.hi[2] - D.lo[2] + 1; f.sc = (long)f.n0 * f.n1 * f.n2;
          V.fv[e] = f; V.have[e] = 1;
        }
      }
      P->peers.push_back(pr);
    }
    {
      std::vector<PackDesc> ap;
      for (auto &pr : P->peers) for (PackDesc D : pr.pack) { D.buf = pr.d_send; ap.push_back(D); }
      P->npack_all = (int)ap.size();
      if (!ap.empty()) { HIPCHK(hipMalloc((void **)&P->d_pack_all, ap.size() * sizeof(PackDesc))); HIPCHK(hipMemcpy(P->d_pack_all, ap.data(), ap.size() * sizeof(PackDesc), hipMemcpyHostToDevice)); }
    }
    HIPCHK(hipStreamSynchronize(ctx().stream));
    V.plan = P;
  }
  g_view_cache.emplace(key, V);
  return V;
}

// ====================================================================================================
// multifab_fill_boundary
// ====================================================================================================
std::vector<XBoxInfo> xboxes_of(const vdn_multifab *mf) {
  const vdn_layout *la = mf->la;
  std::vector<XBoxInfo> v;
  const auto &bx = la->boxes[mf->lev];
  int li = 0;
  for (size_t g = 0; g < bx.size(); g++) {
    XBoxInfo b; memset(&b, 0, sizeof b);
    for (int d = 0; d < 3; d++) { b.vlo[d] = bx[g].lo[d]; b.vhi[d] = bx[g].hi[d] + mf->nodal[d]; }
    b.owner = la->owner[mf->lev][g];
    if (b.owner == ctx().rank) b.fv = mf->fabs[li++];
    v.push_back(b);
  }
  return v;
}

// plan caches are keyed by the layout's uid (never by its address: a freed layout's address is reused) and by the
// allocation; they are purged when the layout is destroyed
struct FbKey { unsigned long uid; const void *base; int lev, nc, ng, nd; bool operator<(const FbKey &o) const {
  return std::tie(uid, base, lev, nc, ng, nd) < std::tie(o.uid, o.base, o.lev, o.nc, o.ng, o.nd); } };
static std::map<FbKey, XPlan *> g_fb_cache;
static std::multimap<unsigned long, XPlan *> g_halo_owned;     // multigrid halo plans, by layout uid
void halo_cache_register(unsigned long uid, XPlan *P) { g_halo_owned.emplace(uid, P); }
void xplan_cache_purge(unsigned long uid) {
  kept_purge(uid);                              // descriptor sets hold pointers into the views' windows
  view_cache_purge(uid);
  for (auto it = g_fb_cache.begin(); it != g_fb_cache.end();) {
    if (it->first.uid == uid) { xplan_free(it->second); it = g_fb_cache.erase(it); } else ++it;
  }
  auto rng = g_halo_owned.equal_range(uid);
  for (auto it = rng.first; it != rng.second; ++it) xplan_free(it->second);
  g_halo_owned.erase(rng.first, rng.second);
  mg_halo_cache_purge(uid);
}

// faces_only: the ghost cells beyond ONE face of a box only (what 7-point operators, the coarse-fine interpolation and the flux matching of the
// composite cell-centred solve read): a level of a thousand 32^3 boxes has 26 neighbour regions per box, 20 of them edges and corners of 32 cells
// or one -- three quarters of the copy kernel's workgroups
void mf_fill_boundary(vdn_multifab *mf, bool faces_only) {
  if (mf->ng == 0) return;
  static const bool faces_ok = !(vdn_env("VDN_FB_FACES") && atoi(vdn_env("VDN_FB_FACES")) == 0);
  faces_only = faces_only && faces_ok;
  const int ndflags = mf->nodal[0] | (mf->nodal[1] << 1) | (mf->nodal[2] << 2);
  auto plan = [&](int variant, const int *trim) -> XPlan * {
    FbKey key{ mf->la->uid, mf->base, mf->lev, mf->nc, mf->ng, ndflags + (faces_only ? 8 : 0) + 16 * variant };
    auto it = g_fb_cache.find(key);
    if (it == g_fb_cache.end()) {
      it = g_fb_cache.emplace(key, xplan_build(xboxes_of(mf), mf->la->pd[mf->lev], mf->la->pmask, mf->ng, mf->nc, faces_only, trim)).first;
      if (g_fb_cache.size() > 4096) vdn_fail("fill_boundary plan cache grew beyond 4096 entries (leaking multifabs?)");
    }
    return it->second;
  };
  auto run = [&](XPlan *P) { if (!(P->local.empty() && P->peers.empty())) xplan_run(P); };
  // (fully nodal fields -- the pressure, the nodal solvers' iterates -- keep the one exchange: their shared nodes are computed alike on every box that holds them,
  // and the solvers exchange them several times per sweep)
  if (ndflags != 1 && ndflags != 2 && ndflags != 4) { run(plan(0, nullptr)); return; }
  // Face-centred data: the boxes on both sides of a shared plane hold its points, and their copies need not be equal bit for bit (velpred's dead band is
  // per box, velpred.f90:215-226: two copies of a MAC velocity can be 1e-9 apart).  A ghost point that both copies cover received whichever the scheduler wrote
  // last when one exchange carried both -- runs of a three-level hierarchy differed from process to process (profiles/r06_determinism.txt).  So: first the HIGH
  // plane of every source along each nodal direction (thin launches), then everything else of every source; wherever a box holds a point on its low side or
  // inside, that copy is the one that stays, and the volume moved is that of the one exchange.
  for (int d = 0; d < 3; d++) if (mf->nodal[d]) { int trim[3] = { 0, 0, 0 }; trim[d] = 2; run(plan(1 + d, trim)); }
  const int trim[3] = { mf->nodal[0], mf->nodal[1], mf->nodal[2] };
  run(plan(4, trim));
}
extern "C" int vdn_multifab_fill_boundary(vdn_multifab *mf) { VDN_TRY mf_fill_boundary(mf, false); VDN_CATCH }

// plan introspection for the CPU tests of the host logic (pure host code, no GPU, no layout object): the remote
// descriptors rank `as_rank` would build for a multifab of the given shape on the given boxes, one row of 14
// longs each: [kind (0 = I send, 1 = I receive), peer, lo[3], hi[3], shift[3], buffer offset, dst box, src box]
extern "C" int vdn_plan_describe(const vdn_box *pd, const int *pmask, int nboxes, const vdn_box *boxes, const int *owner,
                                 int nc, int ng, const int *nodal, int as_rank, long *rows, int maxrows,
                                 int *nrows, int *nlocal_descs) {
  VDN_TRY
  int per[3], nshift[3];
  for (int d = 0; d < 3; d++) { per[d] = pd->hi[d] - pd->lo[d] + 1; nshift[d] = pmask[d] ? 1 : 0; }
  std::map<int, std::pair<long, long>> offs;     // peer -> (send offset, recv offset)
  int n = 0, nloc = 0;
  for (int i = 0; i < nboxes; i++) for (int j = 0; j < nboxes; j++) {
    if (owner[i] != as_rank && owner[j] != as_rank) continue;
    int bvlo[3], bvhi[3], svlo[3], svhi[3];
    for (int d = 0; d < 3; d++) {
      bvlo[d] = boxes[i].lo[d]; bvhi[d] = boxes[i].hi[d] + (nodal ? nodal[d] : 0);
      svlo[d] = boxes[j].lo[d]; svhi[d] = boxes[j].hi[d] + (nodal ? nodal[d] : 0);
    }
    for (int sz = -nshift[2]; sz <= nshift[2]; sz++) for (int sy = -nshift[1]; sy <= nshift[1]; sy++) for (int sx = -nshift[0]; sx <= nshift[0]; sx++) {
      if (j == i && sx == 0 && sy == 0 && sz == 0) continue;
      const int sh[3] = { sx * per[0], sy * per[1], sz * per[2] };
      int lo[3], hi[3]; bool empty = false, all_inside = true;
      for (int d = 0; d < 3; d++) {
        lo[d] = std::max(bvlo[d] - ng, svlo[d] + sh[d]); hi[d] = std::min(bvhi[d] + ng, svhi[d] + sh[d]);
        if (lo[d] > hi[d]) empty = true;
        if (lo[d] < bvlo[d] || hi[d] > bvhi[d]) all_inside = false;
      }
      if (empty || all_inside) continue;
      const long cnt = (long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) * nc;
      if (owner[i] == as_rank && owner[j] == as_rank) { nloc++; continue; }
      const int kind = (owner[j] == as_rank) ? 0 : 1;
      const int peer = kind == 0 ? owner[i] : owner[j];
      long &off = kind == 0 ? offs[peer].first : offs[peer].second;
      if (n < maxrows) { long *r = rows + 14 * n; r[0] = kind; r[1] = peer; for (int d = 0; d < 3; d++) { r[2 + d] = lo[d]; r[5 + d] = hi[d]; r[8 + d] = sh[d]; } r[11] = off; r[12] = i; r[13] = j; }
      off += cnt; n++;
    }
  }
  *nrows = n;
  if (nlocal_descs) *nlocal_descs = nloc;
  VDN_CATCH
}
extern "C" int vdn_box_candidates(int nboxes, const vdn_box *boxes, const int *qlo, const int *qhi, int margin, int *out, int maxout, int *ncand) {
  VDN_TRY
  const std::vector<vdn_box> b(boxes, boxes + nboxes);
  const BoxBins bins(b);
  const std::vector<int> &c = bins.near(qlo, qhi, margin);
  for (size_t i = 0; i < c.size() && (int)i < maxout; i++) out[i] = c[i];
  *ncand = (int)c.size();
  VDN_CATCH
}
