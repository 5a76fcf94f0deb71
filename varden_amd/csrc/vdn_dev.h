// vdn_dev.h -- device-side helpers shared by the HIP kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include "vdn_internal.h"

#define DEVI __device__ __forceinline__

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
DEVI void block_atomic_max(double *addr, double v) {
  v = wave_max(v);
  if (((threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63) == 0) atomic_max_nonneg(addr, v);
}
