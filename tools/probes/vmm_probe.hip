// does the HIP virtual-memory API work on this box, and what does mapping cost?  (the arena of per-step temporaries: reserve the address range once,
// map physical chunks as the high-water mark moves -- instead of a hipMalloc of 150 fields' worth, 6.7 s for 120 GB: profiles/r06_regrid_cost.txt)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAILED %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(double *p, size_t n, double v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void sum(const double *p, size_t n, double *out) { double s = 0; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; atomicAdd(out, s); }
int main() {
  int dev = 0; CK(hipSetDevice(dev));
  int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
  printf("hipDeviceAttributeVirtualMemoryManagementSupported = %d\n", vmm);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  printf("granularity: recommended %zu, minimum %zu\n", gran, gmin);
  const size_t VA = (size_t)256 << 30;
  void *base = nullptr; double t0 = now();
  CK(hipMemAddressReserve(&base, VA, 0, nullptr, 0));
  printf("reserved %zu GB of addresses at %p in %.3f ms\n", VA >> 30, base, 1e3 * (now() - t0));
  hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> hs;
  size_t mapped = 0;
  for (size_t chunk : { (size_t)256 << 20, (size_t)1 << 30, (size_t)1 << 30, (size_t)4 << 30, (size_t)16 << 30, (size_t)32 << 30 }) {
    hipMemGenericAllocationHandle_t h; double a = now();
    CK(hipMemCreate(&h, chunk, &prop, 0)); double b = now();
    CK(hipMemMap((char *)base + mapped, chunk, 0, h, 0)); double c = now();
    CK(hipMemSetAccess((char *)base + mapped, chunk, &acc, 1)); double d = now();
    printf("chunk %6zu MB at offset %6zu MB: create %.3f ms, map %.3f ms, set access %.3f ms\n", chunk >> 20, mapped >> 20, 1e3 * (b - a), 1e3 * (c - b), 1e3 * (d - c));
    hs.push_back(h); mapped += chunk;
  }
  // one kernel over the whole mapped range, across chunk borders
  double *d_out; CK(hipMalloc((void **)&d_out, 8)); CK(hipMemset(d_out, 0, 8));
  const size_t n = mapped / 8; t0 = now();
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (double *)base, n, 1.0);
  CK(hipDeviceSynchronize()); double t1 = now();
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (double *)base, n, 1.0);
  CK(hipDeviceSynchronize()); double t2 = now();
  hipLaunchKernelGGL(sum, dim3(4096), dim3(256), 0, 0, (const double *)base, n, d_out);
  double s = 0; CK(hipMemcpy(&s, d_out, 8, hipMemcpyDeviceToHost));
  printf("fill of %zu MB: first touch %.2f ms, again %.2f ms (%.0f GB/s); sum %.0f (expected %zu)\n", mapped >> 20, 1e3 * (t1 - t0), 1e3 * (t2 - t1), mapped / (t2 - t1) / 1e9, s, n);
  // a plain hipMalloc of the same size, for comparison, and of 100 GB
  for (size_t sz : { mapped, (size_t)100 << 30 }) {
    void *p; t0 = now(); CK(hipMalloc(&p, sz)); t1 = now(); CK(hipFree(p)); t2 = now();
    printf("hipMalloc of %zu MB: %.1f ms, hipFree %.1f ms\n", sz >> 20, 1e3 * (t1 - t0), 1e3 * (t2 - t1));
  }
  t0 = now();
  CK(hipMemUnmap(base, mapped));
  for (auto h : hs) CK(hipMemRelease(h));
  CK(hipMemAddressFree(base, VA));
  printf("unmap + release + address free: %.2f ms\n", 1e3 * (now() - t0));
  printf("OK\n");
  return 0;
}
