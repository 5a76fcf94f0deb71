// vdn_dev.h -- device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <type_traits>
#include <utility>
#include "vdn_internal.h"

#define DEVI __device__ __forceinline__
// neighbouring-lane reads by DPP wave shifts (gfx9 wave_shr:1 / wave_shl:1) instead of ds_bpermute: no LDS round trip.
// lane_prev: lane l gets lane l-1's value (lane 0 keeps its own), lane_next: lane l gets lane l+1's (lane 63 keeps its own) --
// the semantics of __shfl_up(v, 1, 64) / __shfl_down(v, 1, 64)
__device__ __forceinline__ double lane_prev(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

DEVI long fv_idx(const FV &f, int i, int j, int k) {
  return (long)(i - f.a0) + (long)f.n0 * ((long)(j - f.a1) + (long)f.n1 * (long)(k - f.a2));
}
DEVI double &fv_at(const FV &f, int i, int j, int k, int c = 0) { return f.p[fv_idx(f, i, j, k) + f.sc * c]; }
DEVI double fv_get(const FV &f, int i, int j, int k, int c = 0) { return f.p[fv_idx(f, i, j, k) + f.sc * c]; }

// thread -> (i,j,k) of a box [lo,hi] with blockDim = (64,4,1), grid = (ceil(nx/64), ceil(ny/4), nz)
struct Range3 { int lo[3], hi[3]; };
static inline dim3 grid_for(const Range3 &r, dim3 block = dim3(64, 4, 1)) {
  int nx = r.hi[0] - r.lo[0] + 1, ny = r.hi[1] - r.lo[1] + 1, nz = r.hi[2] - r.lo[2] + 1;
  if (nx < 1) nx = 1; if (ny < 1) ny = 1; if (nz < 1) nz = 1;
  return dim3((nx + block.x - 1) / block.x, (ny + block.y - 1) / block.y, (nz + block.z - 1) / block.z);
}
#define THREAD_IJK(r)                                                   \
  const int i = (r).lo[0] + (int)(blockIdx.x * blockDim.x + threadIdx.x); \
  const int j = (r).lo[1] + (int)(blockIdx.y * blockDim.y + threadIdx.y); \
  const int k = (r).lo[2] + (int)(blockIdx.z * blockDim.z + threadIdx.z); \
  const bool in_range = (i <= (r).hi[0]) && (j <= (r).hi[1]) && (k <= (r).hi[2]);

DEVI void block_atomic_max_fwd(double *addr, double v);
// NaN-propagating max for the norms: fmax() drops a NaN operand, so a blown-up field would read as "residual 0 = converged".
// A NaN becomes +inf, whose bit pattern is the largest among the non-negative doubles, so the u64 atomicMax keeps it.
DEVI double nmax(double a, double b) { return (b == b) ? fmax(a, b) : __builtin_huge_val(); }
// ---- XCD-aware tile order ---------------------------------------------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2 (blocks b and b + 8 share one).  With the natural order the
// (x, y) tiles of a k-plane that one XCD works on are scattered over the plane, and every tile's halo rows are fetched into that XCD's
// L2 on their own (kk_cc_gsrb at 256^3: 64 x 4 tiles, phi and rho cost 1.5x their size per pass).  This remap gives each XCD one
// contiguous band of tiles per plane, all XCDs marching through the planes together: the halo rows are shared inside the band.
// Measured (256^3 colour pass, beta from rho): L2 fetch 626 -> 445 MB per pass, time unchanged (0.144 ms): the pass is not bound by
// the fabric.  The other variant -- every XCD one contiguous eighth of the whole grid -- was slower (0.149 ms), and neither helped the
// Godunov marches or the nodal smoother, which keep the natural order.  A bijection of the grid (gridDim.x * gridDim.y divisible
// by 8, identity otherwise).
DEVI void xcd_block(int &bx, int &by, int &bz) {
  const int gx = gridDim.x, gy = gridDim.y, T = gx * gy;
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
  if ((T & 7) || T < 16) return;
  const int id = bx + gx * (by + gy * bz);
  const int per = T >> 3, slot = id >> 3;
  const int t = (id & 7) * per + slot % per;
  bz = slot / per; bx = t % gx; by = t / gx;
}

// The same idea for ANY grid (the remap above needs gridDim.x * gridDim.y divisible by 8 and is the identity otherwise -- which is what
// the marching kernels got in round 1: 5 x 37 or 3 x 65 tiles per slab): workgroup ids id, id + 8, ... share an XCD; XCD x takes the
// x-th contiguous eighth of the logical tile sequence (x fastest, then y, then z), in the order its workgroups are dispatched, so that
// tiles which read the same halo rows / planes meet in one L2 at about the same time.  A bijection of the grid for every size.
// Measured (rocprofv3 FETCH_SIZE, 257^3 nodal Jacobi march): 662 MB read per launch against 407 MB of distinct data before the remap.
DEVI void xcd_tile(int &bx, int &by, int &bz) {
  const int gx = gridDim.x, gy = gridDim.y, N = gx * gy * (int)gridDim.z;
  const int id = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
  const int q = N >> 3, r = N & 7, x = id & 7, slot = id >> 3;
  const int L = (x < r) ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
  bx = L % gx; by = (L / gx) % gy; bz = L / (gx * gy);
}

// ---- box-batched launches -------------------------------------------------------------------------------------------------------
// A level of an adaptive hierarchy can hold hundreds of small boxes; one launch per box and operation makes such levels
// launch-bound (measured: 480 000 launches of ~4 us for two steps on a 271-box level).  A batched kernel takes an array of per-box
// argument structs and a prefix sum of workgroup counts; a workgroup finds its box by bisection and then behaves exactly like the
// per-box kernel (same thread -> cell mapping, so the arithmetic and its order are unchanged).
//   struct A { Range3 r; int g[3]; ...;  static __device__ double body(const A &a, int i, int j, int k, P extra); };
// g = workgroups per direction (g[2] may be smaller than the number of planes: the workgroup then strides over k).  `body` returns a
// non-negative value that is max-reduced into *nrm when nrm is not null.
// read-only launch data (descriptor arrays) seen through the constant address space: scalar loads that the kernel's own stores never
// invalidate, so the compiler treats them like kernel arguments (re-loads instead of spilling, no re-read after every store)
template <class T> DEVI const T &as_constant(const T *p) {
  typedef const T __attribute__((address_space(4))) *CP;
  return *(const T *)(CP)p;
}
// where a batched kernel keeps its descriptor: by value in registers (small ones) or read through the constant address space.  A descriptor may
// say so itself (`static constexpr bool in_constant`): one that is indexed dynamically (arrays of boundary codes walked in a loop) goes to
// SCRATCH when it is copied by value -- mkumac_rho_K, 310 bytes: 320 bytes of scratch per lane, 19 ms instead of 2.3 ms per 8-box launch
template <class T, class = void> struct desc_in_constant : std::integral_constant<bool, (sizeof(T) > 320)> {};
template <class T> struct desc_in_constant<T, std::void_t<decltype(T::in_constant)>> : std::integral_constant<bool, T::in_constant> {};
enum { BATCH_FLAT = 14, BATCH_YZ = 15, BATCH_XZ = 16 };      // tile codes in the top byte of g[2] (otherwise log2 of the tile width along x); XZ + log2 width
template <class A, class P>
__global__ void __launch_bounds__(256) kk_batched(const A *args, const int *start, int nbox, P extra, double *nrm) {
  int lo = 0, hi = nbox - 1;
  const int bid = (int)blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (as_constant(start + mid) <= bid) lo = mid; else hi = mid - 1; }
  // small descriptors are copied into registers (kernels that loop over k re-read a constant-space one every plane: NdfNegB 106 -> 151 us),
  // large ones stay in constant space (a by-value copy of UpdateB lands in 632 bytes of scratch: 3.8 ms -> 0.5 ms)
  typename std::conditional<!desc_in_constant<A>::value, const A, const A &>::type a = as_constant(args + lo);
  int lb = bid - as_constant(start + lo);
  const int lw = a.g[2] >> 24, gz = a.g[2] & 0x7fffff;
  const bool chunks = (a.g[2] >> 23) & 1;       // k: a contiguous chunk of planes per workgroup (the stencil's k-1 / k+1 planes stay in cache) instead of a stride of gz
  {   // a box with many workgroups (the one-box base level of a hierarchy): workgroups bid, bid + 8, ... share an XCD and its L2 -- give each XCD a
      // contiguous piece of the box's tile sequence (xcd_tile's bijection, counted from the box's first workgroup) so that tiles which read the
      // same rows and planes meet in one L2
    const int N = a.g[0] * a.g[1] * gz;
    if (N >= 64) {
      const int s0 = bid - lb, t = (bid - s0) & 7, slot = lb >> 3, q = N >> 3, r = N & 7;
      lb = t * q + (t < r ? t : r) + slot;
    }
  }
  const int bx = lb % a.g[0], by = (lb / a.g[0]) % a.g[1], bz = lb / (a.g[0] * a.g[1]);
  // tile of the 256 threads: 64 x 4, 32 x 8 or 16 x 16 by the width of the box (a level of an adaptive hierarchy is full of 16- and
  // 32-wide boxes: a 64-wide tile would leave half or three quarters of every wave idle); log2(width) rides in the top byte of g[2]
  const int tid = (int)threadIdx.x + 64 * (int)threadIdx.y;
  double v = 0.0;
  if (lw == BATCH_YZ) {                   // a range one or two cells thin along x (the x faces of a box): 16 x 16 tiles of (j, k), x inside
    const int j = a.r.lo[1] + by * 16 + (tid & 15), k = a.r.lo[2] + bz * 16 + (tid >> 4);
    if (j <= a.r.hi[1] && k <= a.r.hi[2])
      for (int i = a.r.lo[0]; i <= a.r.hi[0]; i++) v = nmax(v, A::body(a, i, j, k, extra));
  } else if (lw == BATCH_FLAT) {          // the (i, j) plane of the range flattened over the workgroups' threads: a 33 x 33 plane of nodes is 5 workgroups, not 9 tiles of 64 x 4
    const int nx = a.r.hi[0] - a.r.lo[0] + 1, t = bx * 256 + tid;
    const int jj = t / nx, i = a.r.lo[0] + (t - jj * nx), j = a.r.lo[1] + jj;
    if (j <= a.r.hi[1]) {
      if (chunks) {
        const int nz = a.r.hi[2] - a.r.lo[2] + 1, ch = (nz + gz - 1) / gz, ka = a.r.lo[2] + bz * ch, kb = min(ka + ch - 1, a.r.hi[2]);
        for (int k = ka; k <= kb; k++) v = nmax(v, A::body(a, i, j, k, extra));
      } else
        for (int k = a.r.lo[2] + bz; k <= a.r.hi[2]; k += gz) v = nmax(v, A::body(a, i, j, k, extra));
    }
  } else if (lw >= BATCH_XZ) {            // a range thin along y (the y faces): tiles of (i, k) -- 2^w lanes along x, 256 >> w planes --, y inside
    const int w = lw - BATCH_XZ;
    const int i = a.r.lo[0] + (bx << w) + (tid & ((1 << w) - 1)), k = a.r.lo[2] + bz * (256 >> w) + (tid >> w);
    if (i <= a.r.hi[0] && k <= a.r.hi[2])
      for (int j = a.r.lo[1]; j <= a.r.hi[1]; j++) v = nmax(v, A::body(a, i, j, k, extra));
  } else {
    const int i = a.r.lo[0] + (bx << lw) + (tid & ((1 << lw) - 1)), j = a.r.lo[1] + by * (256 >> lw) + (tid >> lw);
    if (i <= a.r.hi[0] && j <= a.r.hi[1]) {
      if (chunks) {
        const int nz = a.r.hi[2] - a.r.lo[2] + 1, ch = (nz + gz - 1) / gz, ka = a.r.lo[2] + bz * ch, kb = min(ka + ch - 1, a.r.hi[2]);
        for (int k = ka; k <= kb; k++) v = nmax(v, A::body(a, i, j, k, extra));
      } else
        for (int k = a.r.lo[2] + bz; k <= a.r.hi[2]; k += gz) v = nmax(v, A::body(a, i, j, k, extra));
    }
  }
  if (nrm) block_atomic_max_fwd(nrm, v);
}
// wave-level max (64 lanes) then one atomic per wave on a non-negative double stored as u64 bits
DEVI double wave_max(double v) {
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
DEVI void atomic_max_nonneg(double *addr, double v) {
  // for non-negative IEEE doubles the u64 bit pattern is monotone in the value
  // The target only grows during a launch, so a workgroup whose value does not exceed what it READS there has nothing to add: one L2 read
  // instead of a read-modify-write that serialises with every other workgroup's (a stale read only costs a superfluous atomic).  A level of
  // 997 boxes reduces through 64 000 workgroups: 12 ns per serialised atomic was most of its residual pass.
  unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
  const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
  if (bits <= __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  atomicMax(a, bits);
}
// block-level max, then ONE atomic per workgroup.  Every thread of the block must call it.
// (A single device-scope atomic costs ~12 ns and atomics on one address serialise: one per wave on a
// 256^3 box is 262k atomics = 3 ms, measured; reduction kernels therefore also loop over k-planes so
// that a launch has only a few thousand workgroups -- see REDUCE_KLOOP / reduce_grid.)
DEVI void block_atomic_max(double *addr, double v) {
  __shared__ double sm_[16];
  if (!(v == v)) v = __builtin_huge_val();      // a NaN must not vanish in the fmax chain below
  v = wave_max(v);
  const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  const int nw = (blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
  if ((tid & 63) == 0) sm_[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < nw; w++) v = fmax(v, sm_[w]);
    atomic_max_nonneg(addr, v);
  }
  __syncthreads();
}
DEVI void block_atomic_max_fwd(double *addr, double v) { block_atomic_max(addr, v); }
// Planes a workgroup of a batched launch takes when the caller sets no limit.  A 16 x 16 tile of ONE plane is a few KB of traffic behind the
// workgroup's prologue (bisection over the box list, descriptor fetch): on the 263-box level of the two-level bench line the light bodies (a
// few loads per cell) gain from eight planes per workgroup -- CfB 78 -> 43 us, NdfAddB 86 -> 60, NdfNegB 100 -> 56, AddProlongB 68 -> 36,
// AddB 67 -> 54, InterpB 141 -> 119 -- while heavy bodies and one-cell-thick ranges lose from fewer, longer workgroups (RefluxB 11 -> 21,
// NdmRestrictB 82 -> 110, GsrbB 97 -> 103).  A descriptor says so itself: `static constexpr int planes_per_wg = 8;` (default 1).  The cell a
// thread works on and its arithmetic do not depend on it.
template <class T, class = void> struct batch_ppw : std::integral_constant<int, 1> {};
template <class T> struct batch_ppw<T, std::void_t<decltype(T::planes_per_wg)>> : std::integral_constant<int, T::planes_per_wg> {};
// workgroups of one box: tile width 64, 32 or 16 by the width of its range, kz = at most this many workgroups along k (0 = one per plane);
// returns their number (an empty range gets one idle workgroup)
template <class A> static inline int batch_grid(A &a, int kz) {
  const int nx = a.r.hi[0] - a.r.lo[0] + 1, ny = a.r.hi[1] - a.r.lo[1] + 1, nz = a.r.hi[2] - a.r.lo[2] + 1;
  // x faces (ghost slabs, coarse-fine faces, interface nodes): with x along the lanes a 16-wide tile runs one lane in sixteen
  static const bool yz_on = !(vdn_env("VDN_BATCH_YZ") && atoi(vdn_env("VDN_BATCH_YZ")) == 0);
  if (yz_on && nx >= 1 && nx <= 2 && ny >= 1 && nz >= 1 && (long)ny * nz >= 64) {
    a.g[0] = 1; a.g[1] = (ny + 15) / 16; a.g[2] = ((nz + 15) / 16) | (BATCH_YZ << 24);
    return a.g[1] * ((nz + 15) / 16);
  }
  const int lw = nx > 32 ? 6 : (nx > 16 ? 5 : 4), w = 1 << lw, h = 256 >> lw;
  if (yz_on && ny >= 1 && ny <= 2 && nx >= 1 && nz >= 4 && (long)nx * nz >= 64) {      // y faces: x along the lanes as always, planes instead of rows
    a.g[0] = (nx + w - 1) / w; a.g[1] = 1; a.g[2] = ((nz + h - 1) / h) | ((BATCH_XZ + lw) << 24);
    return a.g[0] * ((nz + h - 1) / h);
  }
  // planes per workgroup when the caller sets no limit (batch_ppw): VDN_BATCH_PPW overrides every descriptor's own choice
  static const int ppw_env = vdn_env("VDN_BATCH_PPW") ? std::max(1, atoi(vdn_env("VDN_BATCH_PPW"))) : 0;
  const int ppw = ppw_env > 0 ? ppw_env : batch_ppw<A>::value;
  int g0 = nx > 0 ? (nx + w - 1) / w : 0, g1 = ny > 0 ? (ny + h - 1) / h : 0, g2 = nz > 0 ? ((kz > 0 && nz > kz) ? kz : (nz + ppw - 1) / ppw) : 0;
  if (g0 == 0 || g1 == 0 || g2 == 0) { g0 = g1 = g2 = 1; a.r.hi[0] = a.r.lo[0] - 1; }
  // widths that fill the 16 / 32 / 64-wide tiles badly (node ranges: 17, 25, 33, 41): the plane flattened over the threads when that takes fewer workgroups
  static const bool flat_on = !(vdn_env("VDN_BATCH_FLAT") && atoi(vdn_env("VDN_BATCH_FLAT")) == 0);
  int code = lw;
  if (flat_on && nx > 0 && ny > 0 && (long)nx * ny < (1L << 24)) { const int gf = (nx * ny + 255) / 256; if (gf < g0 * g1) { g0 = gf; g1 = 1; code = BATCH_FLAT; } }
  static const bool chunk_on = !(vdn_env("VDN_BATCH_CHUNK") && atoi(vdn_env("VDN_BATCH_CHUNK")) == 0);
  const int chunked = (chunk_on && !(kz > 0 && nz > kz) && ppw > 1 && g2 < nz) ? 1 : 0;      // planes-per-workgroup mode: contiguous planes
  if (chunked) { const int ch = (nz + g2 - 1) / g2; g2 = (nz + ch - 1) / ch; }                 // (no workgroup without a plane)
  a.g[0] = g0; a.g[1] = g1; a.g[2] = g2 | (chunked << 23) | (code << 24);
  return g0 * g1 * g2;
}
// host side: fills g / the prefix sums, uploads and launches.  kz: at most this many workgroups along k per box (0 = one per plane)
void *arena_alloc(size_t bytes);
void *set_alloc(size_t bytes);
void upload_staged(void *dst, const void *src, size_t bytes);     // host -> device on the launch stream through a pinned ring (runtime.hip)
void *desc_scratch(size_t bytes);                                   // device ring for one-off descriptor arrays (runtime.hip)
template <class A, class P>
static inline void launch_batched(std::vector<A> &v, P extra, double *nrm, int kz, hipStream_t st) {
  if (v.empty()) return;
  std::vector<int> start(v.size());
  int tot = 0;
  for (size_t b = 0; b < v.size(); b++) {
    A &a = v[b];
    const int nx = a.r.hi[0] - a.r.lo[0] + 1, ny = a.r.hi[1] - a.r.lo[1] + 1, nz = a.r.hi[2] - a.r.lo[2] + 1;
    (void)nx; (void)ny; (void)nz;
    start[b] = tot; tot += batch_grid(a, kz);
  }
  A *d_args = (A *)desc_scratch(sizeof(A) * v.size());
  int *d_start = (int *)desc_scratch(sizeof(int) * v.size());
  upload_staged(d_args, v.data(), sizeof(A) * v.size());
  upload_staged(d_start, start.data(), sizeof(int) * v.size());
  hipLaunchKernelGGL((kk_batched<A, P>), dim3(tot), dim3(64, 4, 1), 0, st, (const A *)d_args, (const int *)d_start, (int)v.size(), extra, nrm);
  dbg_sync(1);
}
// the same launch from a descriptor set kept on the device under `key` (vdn_internal.h): build(v) fills the descriptors only when the key is new
template <class A, class P, class F>
static inline void launch_batched_kept(unsigned long long key, unsigned long uid, F &&build, P extra, double *nrm, int kz, hipStream_t st) {
  if (!kept_family_enabled(1)) { std::vector<A> v; build(v); launch_batched(v, extra, nrm, kz, st); return; }
  KeptSet *k = kept_find(key);
  if (!k) {
    std::vector<A> v; build(v);
    std::vector<int> start(v.size());
    int tot = 0;
    for (size_t b = 0; b < v.size(); b++) { start[b] = tot; tot += batch_grid(v[b], kz); }
    k = kept_store(key, uid, v.data(), sizeof(A) * v.size(), start.data(), (int)v.size(), tot);
  }
  if (k->nbox == 0) return;
  hipLaunchKernelGGL((kk_batched<A, P>), dim3(k->tot), dim3(64, 4, 1), 0, st, (const A *)k->d_args, (const int *)k->d_start, k->nbox, extra, nrm);
  dbg_sync(1);
}
// ---- cell kernels: one body, launched for one box or batched over the boxes of a level --------------------------------------------
// struct K { <arguments>; __device__ void cell(int i, int j, int k) const { ... } };   then   launch_cells(vector of (K, range))
template <class K> __global__ void __launch_bounds__(256) kk_cell(K a, Range3 r) {
  THREAD_IJK(r)
  if (!in_range) return;
  a.cell(i, j, k);
}
template <class K> struct CellB { Range3 r; int g[3]; K a;
  static constexpr bool in_constant = desc_in_constant<K>::value || sizeof(K) + sizeof(Range3) + 3 * sizeof(int) > 320;
  static __device__ double body(const CellB &q, int i, int j, int k, int) { q.a.cell(i, j, k); return 0.0; } };
template <class K> static inline void launch_cells(const std::vector<std::pair<K, Range3>> &v, hipStream_t st) {
  if (v.empty()) return;
  if (v.size() == 1) { hipLaunchKernelGGL((kk_cell<K>), grid_for(v[0].second), dim3(64, 4, 1), 0, st, v[0].first, v[0].second); return; }
  // a few LARGE boxes (512^3 cut into eight 256^3 boxes): box by box with the by-value kernel -- the descriptor form pays for its indirection there
  // (mkumac: 5.7 ms for eight boxes against 8 x 0.25, profiles/r04_bench512_kernel_stats.csv); the same cell() per cell, the same bits
  if (v.size() <= 16) {
    long cells = 0;
    for (const auto &e : v) cells += (long)(e.second.hi[0] - e.second.lo[0] + 1) * (e.second.hi[1] - e.second.lo[1] + 1) * (e.second.hi[2] - e.second.lo[2] + 1);
    if (cells / (long)v.size() >= 96L * 96 * 96) {
      for (const auto &e : v) hipLaunchKernelGGL((kk_cell<K>), grid_for(e.second), dim3(64, 4, 1), 0, st, e.first, e.second);
      return;
    }
  }
  std::vector<CellB<K>> b(v.size());
  for (size_t i = 0; i < v.size(); i++) { b[i].r = v[i].second; b[i].a = v[i].first; }
  launch_batched(b, 0, (double *)nullptr, 0, st);
  dbg_sync(32);
}
// a descriptor set that is uploaded once and launched many times (the per-iteration kernels of the composite solves)
template <class A> struct BatchSet {
  A *d_args = nullptr; int *d_start = nullptr; int nbox = 0, tot = 0;
  void build(std::vector<A> &v, int kz, hipStream_t st) {
    nbox = (int)v.size(); tot = 0;
    if (v.empty()) return;
    std::vector<int> start(v.size());
    for (size_t b = 0; b < v.size(); b++) {
      A &a = v[b];
      const int nx = a.r.hi[0] - a.r.lo[0] + 1, ny = a.r.hi[1] - a.r.lo[1] + 1, nz = a.r.hi[2] - a.r.lo[2] + 1;
      (void)nx; (void)ny; (void)nz;
      start[b] = tot; tot += batch_grid(a, kz);
    }
    d_args = (A *)set_alloc(sizeof(A) * v.size());          // (the arena, or the memory of a kept group of sets: vdn_internal.h)
    d_start = (int *)set_alloc(sizeof(int) * v.size());
    upload_staged(d_args, v.data(), sizeof(A) * v.size());
    upload_staged(d_start, start.data(), sizeof(int) * v.size());
    (void)st;
  }
  template <class P> void run(P extra, double *nrm, hipStream_t st) const {
    if (nbox == 0) return;
    hipLaunchKernelGGL((kk_batched<A, P>), dim3(tot), dim3(64, 4, 1), 0, st, (const A *)d_args, (const int *)d_start, nbox, extra, nrm);
    dbg_sync(1);
  }
};
// grid for a reduction over range r: x,y tiled by the block, at most 8 workgroups along z, each looping
// over its share of k-planes with stride gridDim.z
static inline dim3 reduce_grid(const Range3 &r, dim3 block = dim3(64, 4, 1)) {
  dim3 g = grid_for(r, block);
  if (g.z > 8) g.z = 8;
  return g;
}
#define REDUCE_IJ(r)                                                      \
  const int i = (r).lo[0] + (int)(blockIdx.x * blockDim.x + threadIdx.x); \
  const int j = (r).lo[1] + (int)(blockIdx.y * blockDim.y + threadIdx.y); \
  const bool in_ij = (i <= (r).hi[0]) && (j <= (r).hi[1]);
#define REDUCE_KLOOP(r) for (int k = (r).lo[2] + (int)blockIdx.z; k <= (r).hi[2]; k += (int)gridDim.z)
