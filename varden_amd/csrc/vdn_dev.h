// vdn_dev.h -- device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
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

// wave-level max (64 lanes) then one atomic per wave on a non-negative double stored as u64 bits
DEVI double wave_max(double v) {
  #pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}
DEVI void atomic_max_nonneg(double *addr, double v) {
  // for non-negative IEEE doubles the u64 bit pattern is monotone in the value
  atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}
// block-level max, then ONE atomic per workgroup.  Every thread of the block must call it.
// (A single device-scope atomic costs ~12 ns and atomics on one address serialise: one per wave on a
// 256^3 box is 262k atomics = 3 ms, measured; reduction kernels therefore also loop over k-planes so
// that a launch has only a few thousand workgroups -- see REDUCE_KLOOP / reduce_grid.)
DEVI void block_atomic_max(double *addr, double v) {
  __shared__ double sm_[16];
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
